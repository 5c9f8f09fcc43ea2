// How fast the GPU writes a 2 GiB table of 8-byte entries in the access pattern of the k-mer
// table's last pass (every thread writes 16 entries, entry q of thread g at q * threads + g:
// sixteen streams, each coalesced across the wave), against one stream (entry g of thread g,
// sixteen launches' worth in one), and against hipMemsetAsync of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_sixteen(uint2 *__restrict__ out, uint64_t threads) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= threads) return;
#pragma unroll
  for (uint32_t q = 0; q < 16; ++q) out[q * threads + g] = make_uint2((uint32_t)g + q, (uint32_t)(g >> 3) ^ q);
}

__global__ void __launch_bounds__(256) k_one(uint4 *__restrict__ out, uint64_t vecs) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= vecs) return;
  out[g] = make_uint4((uint32_t)g, 1, 2, (uint32_t)(g >> 7));
}

int main() {
  const uint64_t threads = 1ull << 24, entries = threads * 16, bytes = entries * 8;
  uint2 *d = nullptr;
  CK(hipMalloc(&d, bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_sixteen, dim3((unsigned)(threads / 256)), dim3(256), 0, nullptr, d, threads);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("sixteen streams, 8 B a lane: %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_one, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, nullptr, (uint4 *)d, bytes / 16);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("one stream, 16 B a lane:     %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    CK(hipEventRecord(a));
    CK(hipMemsetAsync(d, 0, bytes, nullptr));
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemsetAsync:              %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
  }
  return 0;
}
