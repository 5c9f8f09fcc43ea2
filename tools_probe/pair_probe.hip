// What a second load costs next to a random one: each thread makes `per` random
// 4-byte gathers over a large buffer and, depending on the mode, a second load
// (1) 64 bytes away in the same 128-byte aligned pair, (2) 128 bytes away (the
// next pair), or (3) at another random place.  Mode 0: the one load alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ void k_gather(const uint32_t *__restrict__ buf, uint64_t words, uint32_t *__restrict__ out, uint32_t per) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t z = t * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < per; ++i) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    const uint64_t w = z % (words - 64);
    acc += buf[w];
    if (MODE == 1) acc += buf[w ^ 16];
    if (MODE == 2) acc += buf[w + 32];
    if (MODE == 3) {
      uint64_t y = z * 0xD6E8FEB86659FD93ull; y ^= y >> 32;
      acc += buf[y % (words - 64)];
    }
  }
  out[t] = acc;
}

template <int MODE>
static int run(const char *name, const uint32_t *buf, uint64_t bytes, uint32_t *out) {
  const uint32_t threads = 1u << 26, per = 4;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a, 0));
    k_gather<MODE><<<threads / 256, 256>>>(buf, bytes / 4, out, per);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-44s %7.3f ms  %6.1f G iterations/s\n", name, best, threads * (double)per / best / 1e6);
  return 0;
}

int main(int argc, char **argv) {
  const uint64_t gib = argc > 1 ? atoll(argv[1]) : 24;
  const uint64_t bytes = gib << 30;
  uint32_t *buf = nullptr, *out = nullptr;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, (size_t)(1u << 26) * 4));
  CK(hipMemset(buf, 1, bytes));
  printf("%llu GiB buffer\n", (unsigned long long)gib);
  if (run<0>("one random line", buf, bytes, out)) return 1;
  if (run<1>("+ the other half of its 128-byte pair", buf, bytes, out)) return 1;
  if (run<2>("+ the line 128 bytes further", buf, bytes, out)) return 1;
  if (run<3>("+ another random line", buf, bytes, out)) return 1;
  return 0;
}
