// Diagnostic entry points: let the tests drive the device-wide scan and the
// radix sort directly.  They allocate their own scratch; not for hot paths.
#include "common.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

namespace bwtgpu {

// `per` dependent-free random 4-byte loads per thread over `words` words
__global__ void k_diag_gather(const uint32_t *__restrict__ buf, uint64_t words, uint32_t per,
                              uint32_t *__restrict__ sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t z = t * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < per; ++i) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    acc += buf[z % words];
  }
  if (acc == 0x9E3779B9u) sink[0] = acc;  // keeps the loads alive; all but never true
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

// What the GPU sustains for loads with no locality at all: `loads` random
// 4-byte gathers (each its own 64-byte line) over the caller's buffer, timed
// with HIP events on `stream`, best of three.  The search kernels are bound by
// this rate, not by bytes.  d_buf: at least 8 + 4 bytes; the last word is a sink.
int bwtgpu_diag_gather_rate(const void *d_buf, uint64_t bytes, uint64_t loads, double *lines_per_s,
                            void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!d_buf || !lines_per_s || bytes < 16 || loads == 0) return set_error(kErrArg, "bad argument");
  const uint32_t per = 4;
  const uint64_t threads = div_up(loads, per), words = bytes / 4 - 1;
  if (div_up(threads, 256) > 0x7FFFFFFFull) return set_error(kErrArg, "too many loads for one launch");
  uint32_t *sink = reinterpret_cast<uint32_t *>(const_cast<void *>(d_buf)) + words;
  hipEvent_t a, b;
  BWT_HIP(hipEventCreate(&a));
  BWT_HIP(hipEventCreate(&b));
  float best = 0.f;
  hipError_t e = hipSuccess;
  for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {
    e = hipEventRecord(a, s);
    hipLaunchKernelGGL(k_diag_gather, dim3((unsigned)div_up(threads, 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t *>(d_buf), words, per, sink);
    if (e == hipSuccess) e = hipEventRecord(b, s);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (e == hipSuccess && (best == 0.f || ms < best)) best = ms;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  if (e != hipSuccess || best <= 0.f) return set_error(kErrHip, "gather probe: %s", hipGetErrorString(e));
  *lines_per_s = (double)(threads * per) / (best * 1e-3);
  return kOk;
}

int bwtgpu_diag_scan(int op, const uint32_t *d_in, uint32_t *d_out, uint64_t count,
                     uint32_t *d_total, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  void *ws = nullptr;
  const size_t bytes = scan_workspace_bytes(count);
  BWT_HIP(hipMalloc(&ws, bytes));
  int rc = device_scan_u32(op ? ScanOp::kMaxInclusive : ScanOp::kAddExclusive, d_in, d_out, count,
                           d_total, ws, bytes, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(ws);
  return rc;
}

int bwtgpu_diag_radix_sort(uint64_t *d_keys, uint32_t *d_vals, uint64_t count, int key_bits,
                           void *stream) {
  hipStream_t s = (hipStream_t)stream;
  void *ws = nullptr, *ka = nullptr, *va = nullptr;
  const size_t bytes = radix_sort_workspace_bytes(count);
  hipError_t e = hipMalloc(&ws, bytes);
  if (e == hipSuccess) e = hipMalloc(&ka, (count ? count : 1) * 8);
  if (e == hipSuccess) e = hipMalloc(&va, (count ? count : 1) * 4);
  int rc = e == hipSuccess
               ? radix_sort_pairs(d_keys, d_vals, (uint64_t *)ka, (uint32_t *)va, count, key_bits, ws, bytes, s)
               : set_error(kErrHip, "diag sort buffers: %s", hipGetErrorString(e));
  (void)hipStreamSynchronize(s);
  if (ws) (void)hipFree(ws);
  if (ka) (void)hipFree(ka);
  if (va) (void)hipFree(va);
  return rc;
}

}  // extern "C"
