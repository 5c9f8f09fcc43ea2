// Random 4-byte gathers over a large buffer, allocated three ways, to see how
// much the address translation of a big random working set costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_gather(const uint32_t *__restrict__ buf, uint64_t words, uint32_t *__restrict__ out, uint32_t per) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t z = t * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < per; ++i) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    acc += buf[z % words];
  }
  out[t] = acc;
}

static int run(const char *name, const uint32_t *buf, uint64_t bytes, uint32_t *out) {
  const uint32_t threads = 1u << 26, per = 4;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a, 0));
    k_gather<<<threads / 256, 256>>>(buf, bytes / 4, out, per);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-28s va=%p (mod 1GiB = %4llu MiB) %7.3f ms  %6.1f G gathers/s\n", name, (const void *)buf,
         (unsigned long long)(((uintptr_t)buf & ((1ull << 30) - 1)) >> 20), best, threads * (double)per / best / 1e6);
  return 0;
}

int main(int argc, char **argv) {
  const uint64_t gib = argc > 1 ? atoll(argv[1]) : 24;
  const uint64_t bytes = gib << 30;
  uint32_t *out; CK(hipMalloc(&out, (size_t)4 << 26));
  // (a) plain hipMalloc right away
  { uint32_t *p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); if (run("hipMalloc", p, bytes, out)) return 1; CK(hipFree(p)); }
  // (b) hipMalloc after an odd-sized allocation, odd size itself
  { void *odd; CK(hipMalloc(&odd, (size_t)70 << 20)); uint32_t *p; CK(hipMalloc(&p, bytes + (6u << 20))); CK(hipMemset(p, 1, bytes));
    if (run("hipMalloc after 70 MiB", p, bytes, out)) return 1; CK(hipFree(p)); CK(hipFree(odd)); }
  // (c) reserved at 1 GiB alignment, physical memory mapped at offset 0
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    void *va = nullptr; CK(hipMemAddressReserve(&va, bytes, (size_t)1 << 30, nullptr, 0));
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, bytes, &prop, 0));
    CK(hipMemMap(va, bytes, 0, h, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    CK(hipMemset(va, 1, bytes));
    if (run("reserve 1 GiB-aligned + map", (uint32_t *)va, bytes, out)) return 1;
    CK(hipMemUnmap(va, bytes)); CK(hipMemRelease(h)); CK(hipMemAddressFree(va, bytes));
  }
  return 0;
}
