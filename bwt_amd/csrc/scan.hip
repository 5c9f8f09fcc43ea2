// Device-wide uint32 scans: reduce per tile, scan the tile totals in one
// workgroup, re-scan each tile with its prefix.  Streaming, 8 B/element.
#include "scan.hpp"

namespace bwtgpu {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

template <bool MAX>
__device__ __forceinline__ uint32_t comb(uint32_t a, uint32_t b) {
  return MAX ? (a > b ? a : b) : a + b;
}

// inclusive scan of one value per thread across a 256-thread workgroup;
// returns the inclusive value, *total = workgroup total
template <bool MAX>
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t *wsum, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t u = __shfl_up(v, off, 64);
    if (lane >= off) v = comb<MAX>(v, u);
  }
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  uint32_t pre = 0, tot = 0;
  for (int w = 0; w < kScanThreads / 64; ++w) {
    uint32_t x = wsum[w];
    if (w < wave) pre = comb<MAX>(pre, x);
    tot = comb<MAX>(tot, x);
  }
  __syncthreads();
  *total = tot;
  return comb<MAX>(pre, v);
}

template <bool MAX>
__global__ void __launch_bounds__(kScanThreads)
k_scan_reduce(const uint32_t *__restrict__ in, uint64_t count, uint32_t *__restrict__ tile_tot) {
  __shared__ uint32_t wsum[kScanThreads / 64];
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < count) acc = comb<MAX>(acc, in[base + k]);
  uint32_t tot;
  block_scan_incl<MAX>(acc, wsum, &tot);
  if (threadIdx.x == 0) tile_tot[blockIdx.x] = tot;
}

// one workgroup: tile totals -> exclusive prefixes (in place), grand total out
template <bool MAX>
__global__ void __launch_bounds__(kScanThreads)
k_scan_tiles(uint32_t *__restrict__ tile_tot, uint64_t ntiles, uint32_t *__restrict__ d_total) {
  __shared__ uint32_t wsum[kScanThreads / 64];
  __shared__ uint32_t wlast[kScanThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint64_t base = 0; base < ntiles; base += kScanTile) {
    const uint64_t i0 = base + (uint64_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      v[k] = i0 + k < ntiles ? tile_tot[i0 + k] : 0;
      acc = comb<MAX>(acc, v[k]);
    }
    uint32_t tot;
    const uint32_t incl = block_scan_incl<MAX>(acc, wsum, &tot);
    if (lane == 63) wlast[wave] = incl;
    __syncthreads();
    const uint32_t up = __shfl_up(incl, 1, 64);
    const uint32_t prev = lane ? up : (wave ? wlast[wave - 1] : 0);  // all before this thread
    uint32_t run = comb<MAX>(carry, prev);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (i0 + k < ntiles) tile_tot[i0 + k] = run;
      run = comb<MAX>(run, v[k]);
    }
    carry = comb<MAX>(carry, tot);
    __syncthreads();
  }
  if (threadIdx.x == 0 && d_total) *d_total = carry;
}

template <bool MAX>
__global__ void __launch_bounds__(kScanThreads)
k_scan_apply(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t count,
             const uint32_t *__restrict__ tile_pre) {
  __shared__ uint32_t wsum[kScanThreads / 64];
  __shared__ uint32_t wlast[kScanThreads / 64];
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = base + k < count ? in[base + k] : 0;
    acc = comb<MAX>(acc, v[k]);
  }
  uint32_t tot;
  uint32_t incl = block_scan_incl<MAX>(acc, wsum, &tot);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) wlast[wave] = incl;
  __syncthreads();
  const uint32_t up = __shfl_up(incl, 1, 64);
  const uint32_t prev = lane ? up : (wave ? wlast[wave - 1] : 0);  // everything before this thread
  uint32_t run = comb<MAX>(tile_pre[blockIdx.x], prev);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (MAX) {
      run = comb<MAX>(run, v[k]);
      if (base + k < count) out[base + k] = run;  // inclusive
    } else {
      if (base + k < count) out[base + k] = run;  // exclusive
      run += v[k];
    }
  }
}

size_t scan_workspace_bytes(uint64_t count) {
  return align_up((size_t)div_up(count, kScanTile) * sizeof(uint32_t), 256) + 256;
}

template <bool MAX>
static int scan_impl(const uint32_t *in, uint32_t *out, uint64_t count, uint32_t *d_total,
                     void *d_ws, size_t ws_bytes, hipStream_t s) {
  if (count == 0) {
    if (d_total) BWT_HIP(hipMemsetAsync(d_total, 0, sizeof(uint32_t), s));
    return kOk;
  }
  if (ws_bytes < scan_workspace_bytes(count))
    return set_error(kErrWorkspace, "scan workspace %zu < %zu", ws_bytes, scan_workspace_bytes(count));
  const uint64_t ntiles = div_up(count, kScanTile);
  if (ntiles > 0x7FFFFFFFull) return set_error(kErrArg, "scan too large");
  uint32_t *tiles = reinterpret_cast<uint32_t *>(d_ws);
  BWT_LAUNCH(MAX ? "scan_reduce_max" : "scan_reduce_add", k_scan_reduce<MAX>, dim3((unsigned)ntiles),
             dim3(kScanThreads), 0, s, in, count, tiles);
  BWT_LAUNCH(MAX ? "scan_tiles_max" : "scan_tiles_add", k_scan_tiles<MAX>, dim3(1),
             dim3(kScanThreads), 0, s, tiles, ntiles, d_total);
  BWT_LAUNCH(MAX ? "scan_apply_max" : "scan_apply_add", k_scan_apply<MAX>, dim3((unsigned)ntiles),
             dim3(kScanThreads), 0, s, in, out, count, tiles);
  BWT_CHECK_LAUNCH();
  return kOk;
}

int device_scan_u32(ScanOp op, const uint32_t *in, uint32_t *out, uint64_t count,
                    uint32_t *d_total, void *d_ws, size_t ws_bytes, hipStream_t s) {
  if (op == ScanOp::kMaxInclusive)
    return scan_impl<true>(in, out, count, d_total, d_ws, ws_bytes, s);
  return scan_impl<false>(in, out, count, d_total, d_ws, ws_bytes, s);
}

}  // namespace bwtgpu
