// Stable LSD radix sort of (uint64 key, uint32 value) pairs, 8 bits per pass:
// per-tile digit histograms, one device-wide scan of the digit-major counter
// table, then a stable scatter that ranks keys inside a tile with wave ballots.
#include "radix_sort.hpp"

#include "scan.hpp"

namespace bwtgpu {

constexpr int kRsThreads = 256;
constexpr int kRsItems = 16;
constexpr int kRsTile = kRsThreads * kRsItems;  // 4096 pairs
constexpr int kRsWaves = kRsThreads / 64;
constexpr int kRsWaveSpan = 64 * kRsItems;      // pairs one wave owns, contiguous

__global__ void __launch_bounds__(kRsThreads)
k_rs_hist(const uint64_t *__restrict__ keys, uint64_t count, int shift, uint64_t ntiles,
          uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kRsTile;
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const uint64_t idx = base + (uint64_t)i * kRsThreads + threadIdx.x;
    if (idx < count) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// Tile order is memory order: wave-major, then item, then lane (each wave owns
// a contiguous span and walks it 64 pairs at a time).  A key's position is
//   scanned_hist[digit][tile] + (same digit in earlier waves of the tile)
//   + (same digit earlier in this wave), which keeps equal keys in input order.
__global__ void __launch_bounds__(kRsThreads)
k_rs_scatter(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
             uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint64_t count,
             int shift, uint64_t ntiles, const uint32_t *__restrict__ hist_scanned) {
  __shared__ uint32_t cnt[kRsWaves][256];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (int w = 0; w < kRsWaves; ++w) cnt[w][threadIdx.x] = 0;
  __syncthreads();
  const uint64_t chunk = (uint64_t)blockIdx.x * kRsTile + (uint64_t)wave * kRsWaveSpan;
  uint64_t key[kRsItems];
  uint32_t val[kRsItems];
  uint32_t rank[kRsItems];
  const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const uint64_t idx = chunk + (uint64_t)i * 64 + lane;
    const bool valid = idx < count;
    key[i] = valid ? keys[idx] : ~0ull;
    val[i] = valid ? vals[idx] : 0u;
  }
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const uint64_t idx = chunk + (uint64_t)i * 64 + lane;
    const bool valid = idx < count;
    const uint32_t d = (uint32_t)(key[i] >> shift) & 255u;
    uint64_t same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(same & below);
    const uint32_t prev = cnt[wave][d];  // every lane reads before the leader's write below
    rank[i] = prev + before;
    if (valid && before == 0) cnt[wave][d] = prev + (uint32_t)__popcll(same);
  }
  __syncthreads();
  {
    const uint32_t d = threadIdx.x;
    uint32_t run = hist_scanned[(uint64_t)d * ntiles + blockIdx.x];
    for (int w = 0; w < kRsWaves; ++w) {
      const uint32_t t = cnt[w][d];
      cnt[w][d] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const uint64_t idx = chunk + (uint64_t)i * 64 + lane;
    if (idx < count) {
      const uint32_t d = (uint32_t)(key[i] >> shift) & 255u;
      const uint32_t pos = cnt[wave][d] + rank[i];
      keys_out[pos] = key[i];
      vals_out[pos] = val[i];
    }
  }
}

size_t radix_sort_workspace_bytes(uint64_t count) {
  const uint64_t ntiles = div_up(count ? count : 1, kRsTile);
  return align_up((size_t)ntiles * 256 * sizeof(uint32_t), 256) + scan_workspace_bytes(ntiles * 256) + 256;
}

int radix_sort_pairs(uint64_t *keys, uint32_t *vals, uint64_t *keys_alt, uint32_t *vals_alt,
                     uint64_t count, int key_bits, void *d_ws, size_t ws_bytes, hipStream_t s) {
  if (count <= 1 || key_bits <= 0) return kOk;
  if (count >= (1ull << 32)) return set_error(kErrArg, "radix sort limited to 2^32-1 pairs");
  if (ws_bytes < radix_sort_workspace_bytes(count))
    return set_error(kErrWorkspace, "radix sort workspace %zu < %zu", ws_bytes,
                     radix_sort_workspace_bytes(count));
  const uint64_t ntiles = div_up(count, kRsTile);
  uint32_t *hist = reinterpret_cast<uint32_t *>(d_ws);
  const size_t hist_bytes = align_up((size_t)ntiles * 256 * sizeof(uint32_t), 256);
  void *scan_ws = reinterpret_cast<uint8_t *>(d_ws) + hist_bytes;
  const size_t scan_ws_bytes = ws_bytes - hist_bytes;
  uint64_t *kin = keys, *kout = keys_alt;
  uint32_t *vin = vals, *vout = vals_alt;
  int passes = 0;
  for (int shift = 0; shift < key_bits; shift += 8) {
    BWT_LAUNCH("rs_hist", k_rs_hist, dim3((unsigned)ntiles), dim3(kRsThreads), 0, s, kin, count,
               shift, ntiles, hist);
    BWT_TRY(device_scan_u32(ScanOp::kAddExclusive, hist, hist, ntiles * 256, nullptr, scan_ws,
                            scan_ws_bytes, s));
    BWT_LAUNCH("rs_scatter", k_rs_scatter, dim3((unsigned)ntiles), dim3(kRsThreads), 0, s, kin, vin,
               kout, vout, count, shift, ntiles, hist);
    BWT_CHECK_LAUNCH();
    uint64_t *tk = kin; kin = kout; kout = tk;
    uint32_t *tv = vin; vin = vout; vout = tv;
    ++passes;
  }
  if (passes & 1) {
    BWT_HIP(hipMemcpyAsync(keys, kin, count * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    BWT_HIP(hipMemcpyAsync(vals, vin, count * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  }
  return kOk;
}

}  // namespace bwtgpu
