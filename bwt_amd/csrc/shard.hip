// A build split over several GPUs by bucket range: what a rank makes of its own
// rows before anything travels, and the index assembled from every rank's part.
//
// A rank sorts the suffixes of its range of leading strings (sa_build_core with
// a range): a contiguous run of suffix-array rows, with the BWT symbol of every
// row.  The index needs of those rows only the symbols, 2 bits each, and every
// 8th row as a sample: 0.75 bytes per base instead of the 5 bytes the row and a
// byte-wide symbol take.  That is what the ranks exchange; each rank then lays
// the symbols out as bit-planes, counts them into rank blocks and has the index
// make_fmi (seqindex.c:161-189) describes.  The whole suffix array is needed on
// a rank only for the search-only position records, and travels separately if
// asked for (bwtgpu_index_attach_sa).
#include <cstdlib>

#include "common.hpp"
#include "fm_device.hpp"

namespace bwtgpu {

constexpr uint32_t kMaxShards = 64;

// four BWT symbols per byte, symbol j of a shard at bits 2 (j & 3) of byte j >> 2
__global__ void k_pack_bw2(const uint8_t *__restrict__ bw, uint64_t rows, uint8_t *__restrict__ out) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * q >= rows) return;
  uint32_t acc = 0;
  for (uint32_t k = 0; k < 4 && 4 * q + k < rows; ++k) acc |= (uint32_t)(bw[4 * q + k] & 3u) << (2 * k);
  out[q] = (uint8_t)acc;
}

struct ShardTable {
  uint32_t nshards;
  uint32_t row_begin[kMaxShards + 1];  // first row of shard g; [nshards] = n + 1
  uint64_t bw2_offset[kMaxShards];     // where shard g's packed symbols start in the gathered buffer
};

// even bits of a 16-bit value, packed into 8
__device__ __forceinline__ uint32_t even_bits8(uint32_t x) {
  x &= 0x5555u;
  x = (x | (x >> 1)) & 0x3333u;
  x = (x | (x >> 2)) & 0x0F0Fu;
  return (x | (x >> 4)) & 0xFFu;
}

// One wave per eight 64-base blocks.  BWT position p is the symbol of row p,
// one further along once past the $ row (histsortcomp.c:238-245); the row's
// symbol comes from the shard that holds the row, row 0 (the empty suffix) is
// preceded by the last base of the text.
// Nearly every wave finds its 512 rows in one shard, on one side of the $ row:
// then lane k takes eight neighbouring symbols out of three bytes of that shard
// and the planes of a block are eight lanes' bytes side by side.  The other
// waves look every row up on its own.
__global__ void __launch_bounds__(256)
k_planes_from_shards(const uint8_t *__restrict__ text, const uint8_t *__restrict__ bw2, ShardTable tab, uint32_t n,
                     uint32_t endloc, IndexBlock *__restrict__ blocks, uint64_t nblocks) {
  __shared__ uint32_t begin[kMaxShards + 1];
  __shared__ uint64_t offset[kMaxShards];
  if (threadIdx.x <= tab.nshards) begin[threadIdx.x] = tab.row_begin[threadIdx.x];
  if (threadIdx.x < tab.nshards) offset[threadIdx.x] = tab.bw2_offset[threadIdx.x];
  __syncthreads();
  constexpr int kPer = 8;
  const uint64_t b0 = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kPer;
  const uint32_t lane = threadIdx.x & 63u;
  if (b0 >= nblocks) return;  // wave-uniform
  const uint64_t p_first = b0 * 64, p_end = p_first + 64 * kPer;
  if (p_end <= n && (endloc <= p_first || endloc >= p_end)) {
    const uint64_t r_first = p_first + (endloc <= p_first ? 1 : 0);  // all of them shifted, or none
    uint32_t lo = 0, hi = tab.nshards;  // begin[lo] <= r_first < begin[hi]; the same in every lane
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (begin[mid] <= r_first) lo = mid; else hi = mid;
    }
    if (r_first >= 1 && r_first + 64 * kPer <= begin[lo + 1]) {
      const uint32_t j = (uint32_t)(r_first - begin[lo]) + 8u * lane;
      const uint8_t *at = bw2 + offset[lo] + (j >> 2);
      uint32_t v = (uint32_t)at[0] | ((uint32_t)at[1] << 8);
      if (j & 3u) v |= (uint32_t)at[2] << 16;  // eight symbols from a byte boundary take two bytes
      v >>= 2 * (j & 3u);
      // lane 8 m + i holds byte i of block m's planes
      uint64_t plo = (uint64_t)even_bits8(v) << (8 * (lane & 7u));
      uint64_t phi = (uint64_t)even_bits8(v >> 1) << (8 * (lane & 7u));
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) {
        plo |= __shfl_xor(plo, off, 64);
        phi |= __shfl_xor(phi, off, 64);
      }
      if ((lane & 7u) == 0) {
        blocks[b0 + (lane >> 3)].lo = plo;
        blocks[b0 + (lane >> 3)].hi = phi;
      }
      return;
    }
  }
  uint32_t base[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const uint64_t p = (b0 + k) * 64 + lane;
    base[k] = 0;
    if (p < n) {
      const uint32_t row = (uint32_t)(p + (p >= endloc ? 1 : 0));
      if (row == 0) {
        const uint64_t last = (uint64_t)n - 1;
        base[k] = (text[last >> 2] >> (2 * (3 - (last & 3)))) & 3u;
      } else {
        uint32_t lo = 0, hi = tab.nshards;  // begin[lo] <= row < begin[hi]
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (begin[mid] <= row) lo = mid; else hi = mid;
        }
        const uint32_t j = row - begin[lo];
        base[k] = (bw2[offset[lo] + (j >> 2)] >> (2 * (j & 3u))) & 3u;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const uint64_t lo = __ballot(base[k] & 1u);
    const uint64_t hi = __ballot(base[k] & 2u);
    if (lane == 0 && b0 + k < nblocks) {
      blocks[b0 + k].lo = lo;
      blocks[b0 + k].hi = hi;
    }
  }
}

__global__ void k_text2_words(const uint8_t *__restrict__ text, uint32_t n, uint64_t nwords,
                              uint32_t *__restrict__ text2) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint32_t out = 0;
  if (w >= 1) {
    const uint64_t first = 16 * (w - 1);
    for (uint32_t k = 0; k < 16; ++k) {
      const uint64_t p = first + k;
      if (p < n) out |= (uint32_t)((text[p >> 2] >> (2 * (3 - (p & 3)))) & 3u) << (2 * k);
    }
  }
  text2[w] = out;
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

int bwtgpu_pack_bw2(const uint8_t *d_bw, uint64_t rows, uint8_t *d_out, void *stream) {
  if (!d_bw || !d_out) return set_error(kErrArg, "null argument");
  if (rows == 0) return kOk;
  BWT_LAUNCH("shard_pack_bw2", k_pack_bw2, dim3((unsigned)div_up(div_up(rows, 4), 256)), dim3(256), 0,
             (hipStream_t)stream, d_bw, rows, d_out);
  BWT_CHECK_LAUNCH();
  return kOk;
}

bwtgpu_index *bwtgpu_index_build_from_shards(const uint8_t *d_text, uint32_t n, uint32_t endloc, uint32_t nshards,
                                             const uint64_t *h_row_begin, const uint64_t *h_rows,
                                             const uint64_t *h_bw2_offset, const uint8_t *d_bw2,
                                             const uint32_t *d_samples, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!d_text || !h_row_begin || !h_rows || !h_bw2_offset || !d_bw2 || !d_samples || n == 0) {
    set_error(kErrArg, "null argument");
    return nullptr;
  }
  if (nshards == 0 || nshards > kMaxShards) {
    set_error(kErrArg, "%u shards: between 1 and %u", nshards, kMaxShards);
    return nullptr;
  }
  if (endloc == 0 || endloc > n) {
    set_error(kErrArg, "the row of suffix 0 (%u) must lie in [1, n]", endloc);
    return nullptr;
  }
  ShardTable tab;
  tab.nshards = nshards;
  uint64_t next = 1;  // row 0 is the empty suffix; the shards cover rows 1..n in order
  for (uint32_t g = 0; g < nshards; ++g) {
    if (h_row_begin[g] != next) {
      set_error(kErrArg, "shard %u begins at row %llu, the shards before it end at %llu", g,
                (unsigned long long)h_row_begin[g], (unsigned long long)next);
      return nullptr;
    }
    tab.row_begin[g] = (uint32_t)h_row_begin[g];
    tab.bw2_offset[g] = h_bw2_offset[g];
    next += h_rows[g];
  }
  if (next != (uint64_t)n + 1) {
    set_error(kErrArg, "the shards hold %llu rows, the text has %u suffixes", (unsigned long long)(next - 1), n);
    return nullptr;
  }
  tab.row_begin[nshards] = n + 1;
  bwtgpu_index *ix = nullptr;
  if (index_alloc(&ix, n) != kOk) return nullptr;
  ix->endloc = endloc;
  BWT_LAUNCH("shard_planes", k_planes_from_shards, dim3((unsigned)div_up(div_up(ix->nblocks, 8) * 64, 256)),
             dim3(256), 0, s, d_text, d_bw2, tab, n, endloc, ix->d_blocks, ix->nblocks);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(ix->d_idxs, d_samples, (((size_t)n >> kDevSampleShift) + 1) * 4, hipMemcpyDeviceToDevice, s);
  int rc = e == hipSuccess ? index_finish_from_planes(ix, s) : set_error(kErrHip, "index from shards: %s",
                                                                         hipGetErrorString(e));
  if (rc != kOk) {
    bwtgpu_index_free(ix);
    return nullptr;
  }
  index_defer_search_arrays(ix);
  return ix;
}

int bwtgpu_index_attach_sa(bwtgpu_index *ix, const uint8_t *d_text, const uint32_t *d_sa, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ix || !d_text || !d_sa) return set_error(kErrArg, "null argument");
  if (ix->d_sa) return kOk;
  const uint32_t n = ix->n;
  if (getenv("BWT_MI355X_COMPACT") || n > kPosRecordMaxN) return kOk;  // such an index keeps no suffix array
  BWT_HIP(hipSetDevice(ix->device));
  std::lock_guard<std::mutex> lock(ix->prepare_mu);
  const uint64_t rows = (uint64_t)n + 1, t2words = (uint64_t)n / 16 + 4;
  if (ix->d_sa) return kOk;  // another thread attached it meanwhile
  // in buffers of our own until they are complete (see bwtgpu_index::prepare_mu)
  uint32_t *sa = nullptr, *text2 = nullptr;
  hipError_t ea = pool_alloc(&sa, rows * 4);
  if (ea == hipSuccess) ea = pool_alloc(&text2, t2words * 4);
  if (ea == hipSuccess) ea = hipMemcpyAsync(sa, d_sa, rows * 4, hipMemcpyDeviceToDevice, s);
  if (ea == hipSuccess) {
    BWT_LAUNCH("make_text2", k_text2_words, dim3((unsigned)div_up(t2words, 256)), dim3(256), 0, s, d_text, n,
               t2words, text2);
    ea = hipGetLastError();
  }
  if (ea == hipSuccess) ea = hipStreamSynchronize(s);
  if (ea != hipSuccess) {
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    pool_free(sa);
    pool_free(text2);
    return set_error(kErrHip, "attaching the suffix array of %u bases: %s", n, hipGetErrorString(ea));
  }
  ix->d_text2 = text2;
  publish(ix->d_sa, sa);
  ix->want_records = !ix->d_isa;
  if (ix->want_records) ix->search_ready.store(false, std::memory_order_release);
  return kOk;
}

}  // extern "C"
