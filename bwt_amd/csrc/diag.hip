// Diagnostic entry points: let the tests drive the device-wide scan and the
// radix sort directly.  They allocate their own scratch; not for hot paths.
#include "common.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

using namespace bwtgpu;

extern "C" {

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
