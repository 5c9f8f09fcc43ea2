// FM-index object in HBM: rank blocks from a packed BWT (the rebuild that
// read_index does on the host, fileio.c:62-63 -> seq_index, seqindex.c:23-87),
// export of the stored parts, and the reference-layout rank rows.
#include <cstdlib>

#include "common.hpp"
#include "fm_device.hpp"

namespace bwtgpu {

constexpr int kTileThreads = 256;
constexpr int kBlocksPerThread = 8;
constexpr int kTileBlocks = kTileThreads * kBlocksPerThread;  // 2048 index blocks = 131072 bases

// 16 bases of reference-packed BWT (4 bytes, first base in the top bits of the
// first byte) -> 16 plane bits each.
__device__ __forceinline__ void word_to_planes(uint32_t w, uint32_t &lo16, uint32_t &hi16) {
  auto squeeze = [](uint32_t x) {
    // x holds one bit per base at even positions; pack to a nibble per byte,
    // reverse the nibble (base 0 of a byte sits in its top bits), then gather
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = ((x & 0x01010101u) << 3) | ((x & 0x02020202u) << 1) |
        ((x & 0x04040404u) >> 1) | ((x & 0x08080808u) >> 3);
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
  };
  lo16 = squeeze(w & 0x55555555u);
  hi16 = squeeze((w >> 1) & 0x55555555u);
}

// One thread per index block: 16 bytes of packed BWT -> bit-planes.  d_bwt is
// readable up to 16*nblocks bytes (the caller pads); bases >= n are cleared.
__global__ void k_bwt_to_planes(const uint4 *__restrict__ d_bwt, IndexBlock *__restrict__ blocks,
                                uint64_t nblocks, uint32_t n) {
  uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  uint4 w = d_bwt[b];
  uint32_t l0, h0, l1, h1, l2, h2, l3, h3;
  word_to_planes(w.x, l0, h0);
  word_to_planes(w.y, l1, h1);
  word_to_planes(w.z, l2, h2);
  word_to_planes(w.w, l3, h3);
  uint64_t lo = (uint64_t)l0 | ((uint64_t)l1 << 16) | ((uint64_t)l2 << 32) | ((uint64_t)l3 << 48);
  uint64_t hi = (uint64_t)h0 | ((uint64_t)h1 << 16) | ((uint64_t)h2 << 32) | ((uint64_t)h3 << 48);
  uint64_t first = b * 64;
  uint64_t valid = first >= n ? 0 : (n - first >= 64 ? 64 : n - first);
  uint64_t mask = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
  blocks[b].lo = lo & mask;
  blocks[b].hi = hi & mask;
}

__device__ __forceinline__ void block_counts(const IndexBlock &blk, uint64_t b, uint32_t n,
                                             uint32_t cnt[4]) {
  uint64_t first = b * 64;
  uint64_t valid = first >= n ? 0 : (n - first >= 64 ? 64 : n - first);
  uint64_t mask = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
  uint64_t lo = blk.lo, hi = blk.hi;
  cnt[0] = __popcll(~lo & ~hi & mask);
  cnt[1] = __popcll(lo & ~hi & mask);
  cnt[2] = __popcll(~lo & hi & mask);
  cnt[3] = __popcll(lo & hi & mask);
}

// Pass 1 of the rank-table scan: per-tile totals of the four base counts.
__global__ void __launch_bounds__(kTileThreads)
k_rank_tile_reduce(const IndexBlock *__restrict__ blocks, uint64_t nblocks, uint32_t n,
                   uint4 *__restrict__ tile_sums) {
  __shared__ uint32_t s[4][kTileThreads / kWave];
  uint64_t b0 = (uint64_t)blockIdx.x * kTileBlocks + (uint64_t)threadIdx.x * kBlocksPerThread;
  uint32_t acc[4] = {0, 0, 0, 0};
  for (int k = 0; k < kBlocksPerThread; ++k) {
    uint64_t b = b0 + k;
    if (b < nblocks) {
      uint32_t c[4];
      block_counts(blocks[b], b, n, c);
      for (int q = 0; q < 4; ++q) acc[q] += c[q];
    }
  }
  for (int q = 0; q < 4; ++q) {
    uint32_t v = acc[q];
    for (int off = 32; off; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s[q][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t[4];
    for (int q = 0; q < 4; ++q) {
      t[q] = 0;
      for (int w = 0; w < kTileThreads / kWave; ++w) t[q] += s[q][w];
    }
    tile_sums[blockIdx.x] = make_uint4(t[0], t[1], t[2], t[3]);
  }
}

// Pass 2: one workgroup turns tile totals into exclusive tile prefixes with
// C[c] folded in, and publishes C[0..4] (seqindex.c:183-187).
__global__ void __launch_bounds__(1024)
k_rank_scan_tiles(uint4 *__restrict__ tile_sums, uint32_t ntiles, uint32_t *__restrict__ d_C,
                  uint32_t *__restrict__ host_C /* pinned, or null */, const HostWords more0, const HostWords more1) {
  __shared__ uint32_t s[4][1024];
  __shared__ uint32_t carry[4];
  __shared__ uint32_t Cs[5];
  const int t = threadIdx.x;
  if (t < 4) carry[t] = 0;
  __syncthreads();
  // first sweep: grand totals, to know C before writing prefixes
  uint32_t tot[4] = {0, 0, 0, 0};
  for (uint32_t i = t; i < ntiles; i += 1024) {
    uint4 v = tile_sums[i];
    tot[0] += v.x; tot[1] += v.y; tot[2] += v.z; tot[3] += v.w;
  }
  for (int q = 0; q < 4; ++q) s[q][t] = tot[q];
  __syncthreads();
  for (int off = 512; off; off >>= 1) {
    if (t < off)
      for (int q = 0; q < 4; ++q) s[q][t] += s[q][t + off];
    __syncthreads();
  }
  if (t == 0) {
    Cs[0] = 1;
    for (int q = 0; q < 4; ++q) Cs[q + 1] = Cs[q] + s[q][0];
    for (int q = 0; q < 5; ++q) d_C[q] = Cs[q];
    if (host_C)
      for (int q = 0; q < 5; ++q) host_C[q] = Cs[q];
  }
  if ((uint32_t)t < more0.words) more0.dst[t] = more0.src[t];
  if ((uint32_t)t < more1.words) more1.dst[t] = more1.src[t];
  __syncthreads();
  // second sweep: chunked exclusive scan with a running carry
  for (uint32_t base = 0; base < ntiles; base += 1024) {
    uint32_t i = base + t;
    uint4 v = i < ntiles ? tile_sums[i] : make_uint4(0, 0, 0, 0);
    uint32_t own[4] = {v.x, v.y, v.z, v.w};
    for (int q = 0; q < 4; ++q) s[q][t] = own[q];
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      uint32_t add[4] = {0, 0, 0, 0};
      if (t >= off)
        for (int q = 0; q < 4; ++q) add[q] = s[q][t - off];
      __syncthreads();
      for (int q = 0; q < 4; ++q) s[q][t] += add[q];
      __syncthreads();
    }
    if (i < ntiles) {
      uint4 o;
      o.x = Cs[0] + carry[0] + s[0][t] - own[0];
      o.y = Cs[1] + carry[1] + s[1][t] - own[1];
      o.z = Cs[2] + carry[2] + s[2][t] - own[2];
      o.w = Cs[3] + carry[3] + s[3][t] - own[3];
      tile_sums[i] = o;
    }
    __syncthreads();
    if (t < 4) carry[t] += s[t][1023];
    __syncthreads();
  }
}

// Pass 3: exclusive scan inside each tile; writes occ[] of every block.
__global__ void __launch_bounds__(kTileThreads)
k_rank_apply(IndexBlock *__restrict__ blocks, uint64_t nblocks, uint32_t n,
             const uint4 *__restrict__ tile_prefix) {
  __shared__ uint32_t wsum[4][kTileThreads / kWave];
  uint64_t b0 = (uint64_t)blockIdx.x * kTileBlocks + (uint64_t)threadIdx.x * kBlocksPerThread;
  uint32_t cnt[kBlocksPerThread][4];
  uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < kBlocksPerThread; ++k) {
    uint64_t b = b0 + k;
    if (b < nblocks) {
      block_counts(blocks[b], b, n, cnt[k]);
    } else {
      cnt[k][0] = cnt[k][1] = cnt[k][2] = cnt[k][3] = 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += cnt[k][q];
  }
  // exclusive scan of per-thread totals across the workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t excl[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t v = acc[q];
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    if (lane == 63) wsum[q][wave] = v;
    excl[q] = v - acc[q];
  }
  __syncthreads();
  uint4 tp = tile_prefix[blockIdx.x];
  uint32_t run[4] = {tp.x, tp.y, tp.z, tp.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    for (int w = 0; w < wave; ++w) run[q] += wsum[q][w];
    run[q] += excl[q];
  }
#pragma unroll
  for (int k = 0; k < kBlocksPerThread; ++k) {
    uint64_t b = b0 + k;
    if (b < nblocks) {
      *reinterpret_cast<uint4 *>(blocks[b].occ) = make_uint4(run[0], run[1], run[2], run[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) run[q] += cnt[k][q];
    }
  }
}

// Bit-planes -> reference-packed BWT bytes, one thread per index block.
__global__ void k_planes_to_bwt(const IndexBlock *__restrict__ blocks, uint64_t nblocks,
                                uint4 *__restrict__ d_bwt) {
  uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  uint64_t lo = blocks[b].lo, hi = blocks[b].hi;
  uint32_t w[4];
  for (int q = 0; q < 4; ++q) {
    uint32_t acc = 0;
    for (int k = 0; k < 16; ++k) {
      int bit = 16 * q + k;
      uint32_t base = (uint32_t)((lo >> bit) & 1ull) | ((uint32_t)((hi >> bit) & 1ull) << 1);
      int byte = k >> 2, slot = k & 3;
      acc |= base << (8 * byte + 2 * (3 - slot));
    }
    w[q] = acc;
  }
  d_bwt[b] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Reference-layout rank rows (seqindex.c:23-87, blocksize 16): row j holds the
// counts of each base in BWT[0, min(16 j, n)).
__global__ void k_rank_rows(IndexView v, uint32_t C0, uint32_t C1, uint32_t C2, uint32_t C3,
                            uint32_t blocksize, uint64_t rows, uint4 *__restrict__ out) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= rows) return;
  uint64_t p = j * blocksize;
  if (p > v.n) p = v.n;
  const IndexBlock *b = v.blocks + (p >> 6);
  uint4 occ = *reinterpret_cast<const uint4 *>(b->occ);
  uint64_t lo = b->lo, hi = b->hi, m = low_mask((uint32_t)(p & 63));
  uint4 r;
  r.x = occ.x - C0 + __popcll(~lo & ~hi & m);
  r.y = occ.y - C1 + __popcll(lo & ~hi & m);
  r.z = occ.z - C2 + __popcll(~lo & hi & m);
  r.w = occ.w - C3 + __popcll(lo & hi & m);
  out[j] = r;
}

// Per-symbol search blocks from the finished 64-base blocks: one thread per
// 96-base block, all four symbols.
__global__ void k_build_sym(const IndexBlock *__restrict__ blocks, uint64_t nblocks, uint32_t n,
                            SymBlock *__restrict__ sym, uint64_t nsym) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nsym) return;
  const uint64_t p0 = b * kSymBases;
  const uint64_t q = p0 >> 6;
  const uint32_t off = (uint32_t)(p0 & 63u);  // 0 or 32
  const IndexBlock A = blocks[q];  // q <= n/64 always exists
  uint64_t lo_b = 0, hi_b = 0;
  if (q + 1 < nblocks) {
    lo_b = blocks[q + 1].lo;
    hi_b = blocks[q + 1].hi;
  }
  // 96 plane bits starting at p0
  uint64_t lo64, hi64;
  uint32_t lo32, hi32;
  if (off == 0) {
    lo64 = A.lo; hi64 = A.hi;
    lo32 = (uint32_t)lo_b; hi32 = (uint32_t)hi_b;
  } else {
    lo64 = (A.lo >> 32) | (lo_b << 32); hi64 = (A.hi >> 32) | (hi_b << 32);
    lo32 = (uint32_t)(lo_b >> 32); hi32 = (uint32_t)(hi_b >> 32);
  }
  const uint64_t valid = p0 >= n ? 0 : (n - p0 >= kSymBases ? kSymBases : n - p0);
  const uint64_t v64 = valid >= 64 ? ~0ull : ((1ull << valid) - 1ull);
  const uint32_t v32 = valid >= 96 ? ~0u : (valid > 64 ? ((1u << (valid - 64)) - 1u) : 0u);
  const uint64_t before = low_mask(off);  // bases of block q that precede p0
  const uint32_t occq[4] = {A.occ[0], A.occ[1], A.occ[2], A.occ[3]};
#pragma unroll
  for (uint32_t c = 0; c < 4; ++c) {
    const uint64_t nlo = (c & 1u) ? 0ull : ~0ull, nhi = (c & 2u) ? 0ull : ~0ull;
    const uint64_t m64 = (lo64 ^ nlo) & (hi64 ^ nhi) & v64;
    const uint32_t m32 = (lo32 ^ (uint32_t)nlo) & (hi32 ^ (uint32_t)nhi) & v32;
    SymBlock o;
    o.occ = occq[c] + (uint32_t)__popcll((A.lo ^ nlo) & (A.hi ^ nhi) & before);
    o.m[0] = (uint32_t)m64;
    o.m[1] = (uint32_t)(m64 >> 32);
    o.m[2] = m32;
    sym[(uint64_t)c * nsym + b] = o;
  }
}

__global__ void k_thin_samples(const uint32_t *__restrict__ dense, uint64_t nstored,
                               uint32_t *__restrict__ stored) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nstored) stored[i] = dense[i << (kFileSampleShift - kDevSampleShift)];
}

// One level of the k-mer interval table: the interval of string c.x (x of j
// bases) is one backward-search step (seqindex.c:292-306) from the interval of
// x.  Strings are numbered with their first base most significant, so
// neighbouring threads extend neighbouring intervals with the same base and the
// rank queries of a wave fall on neighbouring blocks.  An interval is kept even
// when it is empty: its start is then the row the string would be inserted at,
// which is what the step computes (the number of suffixes smaller than c.x).
// The last level stores the start alone.
// the four extensions c.x of the string whose interval is iv
__device__ __forceinline__ void kmer_extend(const IndexView &v, uint2 iv, uint2 o[4]) {
  const uint32_t a = iv.x - (iv.x > v.endloc), b = iv.y - (iv.y > v.endloc);  // seqindex.c:222-223
  const IndexBlock *ba = v.blocks + (a >> 6), *bb = v.blocks + (b >> 6);
  const uint4 oa = *reinterpret_cast<const uint4 *>(ba->occ), ob = *reinterpret_cast<const uint4 *>(bb->occ);
  const uint64_t la = ba->lo, ha = ba->hi, lb = bb->lo, hb = bb->hi;
  const uint64_t ma = low_mask(a & 63u), mb = low_mask(b & 63u);
  const uint32_t occa[4] = {oa.x, oa.y, oa.z, oa.w}, occb[4] = {ob.x, ob.y, ob.z, ob.w};
#pragma unroll
  for (uint32_t c = 0; c < 4; ++c) {
    const uint64_t nlo = (c & 1u) ? 0ull : ~0ull, nhi = (c & 2u) ? 0ull : ~0ull;
    o[c] = make_uint2(occa[c] + (uint32_t)__popcll((la ^ nlo) & (ha ^ nhi) & ma),
                      occb[c] + (uint32_t)__popcll((lb ^ nlo) & (hb ^ nhi) & mb));
  }
}

// Second word of a two-word entry (see kmer_interval / kmer_rows_with in fm_device.hpp):
// SA[start] for an interval of one row (and for intervals too large for symbols),
// the BWT symbols of its rows for an interval of 2 .. kmer_symbol_rows() rows, and
// the interval's size in the top `sizebits` bits.
__device__ __forceinline__ uint32_t kmer_second_word(const IndexView &v, uint2 iv, const uint32_t *__restrict__ sa,
                                                    uint32_t sizebits) {
  const uint32_t size = iv.y > iv.x ? iv.y - iv.x : 0u;
  uint32_t second = 0;
  if (size >= 2u && size <= kmer_symbol_rows(sizebits) && iv.y <= v.n + 1u) {
    if (v.endloc < iv.x || v.endloc >= iv.y) {  // else the $ row lies inside: no symbols, flag clear
      // the rows are up to 15 neighbouring bits of the two planes, in one block or two
      const uint32_t j0 = iv.x - (iv.x > v.endloc), off = j0 & 63u;
      const IndexBlock *b = v.blocks + (j0 >> 6);
      uint64_t lo = b->lo >> off, hi = b->hi >> off;
      if (off + size > 64u) {
        lo |= b[1].lo << (64u - off);
        hi |= b[1].hi << (64u - off);
      }
      auto spread = [](uint32_t x) {  // bit t -> bit 2t, t < 16
        x = (x | (x << 8)) & 0x00FF00FFu;
        x = (x | (x << 4)) & 0x0F0F0F0Fu;
        x = (x | (x << 2)) & 0x33333333u;
        return (x | (x << 1)) & 0x55555555u;
      };
      const uint32_t keep = (1u << size) - 1u;
      second = spread((uint32_t)lo & keep) | (spread((uint32_t)hi & keep) << 1) | (1u << (32u - sizebits - 1u));
    }
  } else if (size && iv.x <= v.n) {
    second = sa[iv.x];
  }
  if (sizebits) {  // the interval's size in the bits a text position leaves free (see bwtgpu_index)
    const uint32_t cap = (1u << sizebits) - 1u;
    second |= (size < cap ? size : cap) << (32u - sizebits);
  }
  return second;
}

template <bool LAST>
__global__ void k_kmer_level(IndexView v, const uint2 *__restrict__ src, uint2 *__restrict__ dst,
                             uint32_t *__restrict__ dst_start, uint32_t j, const uint32_t *__restrict__ sa,
                             uint32_t sizebits) {
  // one thread per string x of j bases: the two blocks its interval ends fall
  // in serve all four extensions c.x
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t parents = 1ull << (2 * j);
  if (p >= parents) return;
  const uint2 iv = src[p];
  if (LAST && sa) {
    // two-word entries: the start and, where the string occurs, SA[start].  Starts grow
    // with the string number, so neighbouring threads read neighbouring rows of the array.
    uint2 o[4];
    kmer_extend(v, iv, o);
    uint2 *entries = reinterpret_cast<uint2 *>(dst_start);
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c)
      entries[(uint64_t)c * parents + p] = make_uint2(o[c].x, kmer_second_word(v, o[c], sa, sizebits));
    return;
  }
  if (LAST) {
    const uint32_t a = iv.x - (iv.x > v.endloc);
    const IndexBlock *ba = v.blocks + (a >> 6);
    const uint4 oa = *reinterpret_cast<const uint4 *>(ba->occ);
    const uint64_t la = ba->lo, ha = ba->hi;
    const uint64_t ma = low_mask(a & 63u);
    const uint32_t occa[4] = {oa.x, oa.y, oa.z, oa.w};
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
      const uint64_t nlo = (c & 1u) ? 0ull : ~0ull, nhi = (c & 2u) ? 0ull : ~0ull;
      dst_start[(uint64_t)c * parents + p] = occa[c] + (uint32_t)__popcll((la ^ nlo) & (ha ^ nhi) & ma);
    }
    return;
  }
  uint2 o[4];
  kmer_extend(v, iv, o);
#pragma unroll
  for (uint32_t c = 0; c < 4; ++c) dst[(uint64_t)c * parents + p] = o[c];
}

// The last two levels of a two-word table in one pass: a thread takes a string g
// of K - 2 bases, extends it to its four children and each child to its four, and
// writes the sixteen finished entries; the level in between never goes to memory
// (it is half as large again as everything below it).
__global__ void __launch_bounds__(256)
k_kmer_last_two(IndexView v, const uint2 *__restrict__ src, uint2 *__restrict__ entries, uint32_t j /* = K - 2 */,
                const uint32_t *__restrict__ sa, uint32_t sizebits) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t grand = 1ull << (2 * j);
  if (g >= grand) return;
  uint2 mid[4];
  kmer_extend(v, src[g], mid);
#pragma unroll
  for (uint32_t c1 = 0; c1 < 4; ++c1) {
    uint2 o[4];
    kmer_extend(v, mid[c1], o);
    const uint64_t p = (uint64_t)c1 * grand + g;  // the child's number among the strings of K - 1 bases
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c)
      entries[(uint64_t)c * 4 * grand + p] = make_uint2(o[c].x, kmer_second_word(v, o[c], sa, sizebits));
  }
}

struct KmerFirst {
  uint32_t c[5];
};
constexpr uint32_t kKmerHeadLevels = 5;  // 4^4 strings is what one workgroup extends in one go

// Workgroup 0: levels 1..top of the table (top <= kKmerHeadLevels, top < K)
// through LDS, level `top` written to dst.  Workgroup 1: what follows the
// table's 4^K starts -- the closing start n + 1, a word of padding, and one
// (string, 1) pair per suffix shorter than K bases, the string being that
// suffix padded with A to K bases; the last K - 1 bases of the text are read
// off the index by walking LF from the row of the empty suffix
// (seqindex.c:214-219).  Unused pairs name no string.
__global__ void __launch_bounds__(256) k_kmer_head(IndexView v, KmerFirst C, uint32_t top, uint32_t K,
                                                    uint2 *__restrict__ dst, uint32_t *__restrict__ tab,
                                                    uint32_t entry_words) {
  const uint32_t t = threadIdx.x;
  if (blockIdx.x == 1) {
    if (t) return;
    uint32_t *trailer = tab + ((uint64_t)entry_words << (2 * K));
    trailer[0] = v.n + 1;
    trailer[1] = 0;
    uint2 *gaps = reinterpret_cast<uint2 *>(trailer + 2);
    uint32_t idx = 0, val = 0, L = 1;
    for (; L < K && idx != v.endloc; ++L) {
      const uint32_t jj = idx - (idx > v.endloc);
      const IndexBlock *b = v.blocks + (jj >> 6);
      const uint32_t c = (uint32_t)((b->lo >> (jj & 63u)) & 1ull) | ((uint32_t)((b->hi >> (jj & 63u)) & 1ull) << 1);
      val |= c << (2 * (L - 1));  // the suffix of L bases, first base most significant
      gaps[L - 1] = make_uint2(val << (2 * (K - L)), 1u);
      idx = lf_step(v, idx);
    }
    for (; L <= kKmerMaxGaps; ++L) gaps[L - 1] = make_uint2(0xFFFFFFFFu, 0u);
    return;
  }
  __shared__ uint2 lv[2][256];
  if (t < 4) {
    const uint2 e = make_uint2(t == 0 ? C.c[0] : t == 1 ? C.c[1] : t == 2 ? C.c[2] : C.c[3],
                               t == 0 ? C.c[1] : t == 1 ? C.c[2] : t == 2 ? C.c[3] : C.c[4]);
    if (top == 1) dst[t] = e;
    else lv[1][t] = e;
  }
  for (uint32_t j = 1; j < top; ++j) {
    __syncthreads();
    const uint32_t parents = 1u << (2 * j);
    if (t < parents) {
      uint2 o[4];
      kmer_extend(v, lv[j & 1u][t], o);
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c) {
        if (j + 1 == top) dst[c * parents + t] = o[c];
        else lv[(j + 1) & 1u][c * parents + t] = o[c];
      }
    }
  }
}

// Table size.  An index that keeps its suffix array gets two-word entries, and a
// string that occurs once then costs a search one line for the table and one for
// the position record; the share of such strings among those that occur is
// exp(-n / 4^k), so the table is made sparse: the smallest k with 4^k >= 2 n
// (46 Mbp: k = 14, 2.1 GB; 3.1 Gbp: k = 16, 34 GB -- the GPU has 288).  An index
// without its suffix array (one-word entries, every search goes on with rank
// steps) gets the smallest k with 4^k >= n.  At most 4^16 entries: a string
// number is 32 bits.  BWT_MI355X_KMER=<k> overrides (0: no table),
// BWT_MI355X_COMPACT means none; when HBM is short the next smaller k is tried.
static uint32_t kmer_table_k(uint32_t n, bool two_words) {
  if (getenv("BWT_MI355X_COMPACT")) return 0;
  if (const char *e = getenv("BWT_MI355X_KMER")) {
    const long k = strtol(e, nullptr, 10);
    return k < 2 ? 0u : (k > 16 ? 16u : (uint32_t)k);
  }
  const uint64_t want = two_words ? 2ull * n : (uint64_t)n;
  uint32_t k = 0;
  while (k < 16 && (1ull << (2 * k)) < want) ++k;
  return k < 6 ? 0u : k;
}

// Layout of the table buffer: 4^K + 1 interval starts (the last is n + 1), one
// word of padding, then kKmerMaxGaps (string, rows) pairs.
//
// start[x + 1] is the end of the interval of x unless rows of suffixes shorter
// than K bases lie between the two: a suffix s of L < K bases sorts after every
// string smaller than s and before every string s is a prefix of, so its row
// sits right before the interval of s.AA..A and nowhere else.  There are at
// most K - 1 such suffixes; the pairs list the strings s.AA..A, and a lookup
// subtracts the rows of the pairs that name the next string.
int index_build_kmer_table(bwtgpu_index *ix, hipStream_t s) {
  const uint32_t W = ix->d_sa ? 2u : 1u;  // with the suffix array at hand, entries carry SA[start] as well
  uint32_t K = kmer_table_k(ix->n, W == 2);
  if (K == 0) return kOk;
  static_assert(kKmerMaxGaps >= 15, "one pair per suffix shorter than the longest table string");
  uint64_t entries = 0;
  uint32_t *tab = nullptr;
  uint2 *tmp_a = nullptr, *tmp_b = nullptr, *kmer4 = nullptr;
  for (;; --K) {
    entries = 1ull << (2 * K);
    const size_t tab_bytes = (entries * W + 2) * sizeof(uint32_t) + kKmerMaxGaps * sizeof(uint2);
    // tmp_a holds levels K - 1, K - 3, ..; a two-word table of 7 bases or more never stores level K - 1 (k_kmer_last_two)
    const uint64_t a_entries = (W == 2 && K >= 7) ? entries / 64 : entries / 4;
    if (pool_alloc(&tab, tab_bytes) == hipSuccess && pool_alloc(&tmp_a, a_entries * sizeof(uint2)) == hipSuccess &&
        pool_alloc(&tmp_b, (entries / 16) * sizeof(uint2)) == hipSuccess)
      break;
    (void)hipGetLastError();  // short of memory: a smaller table, or none (searches then start from the first base)
    pool_free(tab);
    pool_free(tmp_a);
    pool_free(tmp_b);
    tab = nullptr;
    tmp_a = tmp_b = nullptr;
    if (K <= 6 || getenv("BWT_MI355X_KMER")) return kOk;
  }
  // bits of an entry's second word that text positions (0..n) leave free take the interval's size
  uint32_t posbits = 1;
  while (posbits < 32 && (ix->n >> posbits)) ++posbits;
  uint32_t sizebits = (W == 2 && 32 - posbits >= 2) ? (32 - posbits < 6 ? 32 - posbits : 6u) : 0u;
  // BWT_MI355X_KMER_SIZEBITS=<b> takes fewer (0: none, as for texts of 2^30 bases and more; 1 means 0 too)
  if (const char *e = getenv("BWT_MI355X_KMER_SIZEBITS")) {
    const uint32_t want = (uint32_t)atoi(e);
    if (want < sizebits) sizebits = want >= 2 ? want : 0u;
  }
  // level j < K lives in tmp_a when K - 1 - j is even, in tmp_b otherwise
  auto buf = [&](uint32_t j) { return ((K - 1 - j) & 1u) ? tmp_b : tmp_a; };
  const IndexView v = view_of(ix);
  KmerFirst first;
  for (uint32_t c = 0; c < 5; ++c) first.c[c] = ix->C[c];
  const uint32_t top = K - 1 < kKmerHeadLevels ? K - 1 : kKmerHeadLevels;
  BWT_LAUNCH("kmer_head", k_kmer_head, dim3(2), dim3(256), 0, s, v, first, top, K, buf(top), tab, W);
  hipError_t e = hipGetLastError();
  for (uint32_t j = top; j < K && e == hipSuccess; ++j) {
    const uint64_t parents = 1ull << (2 * j);
    if (W == 2 && j + 2 == K) {
      BWT_LAUNCH("kmer_level", k_kmer_last_two, dim3((unsigned)div_up(parents, 256)), dim3(256), 0, s, v, buf(j),
                 reinterpret_cast<uint2 *>(tab), j, (const uint32_t *)ix->d_sa, sizebits);
      e = hipGetLastError();
      break;
    }
    if (W == 2 && K >= 9 && j + 3 == K) {
      // level K - 4 is about to be overwritten by level K - 2: a copy stays with the index
      // (a sixteenth of the level that is kept as it is; skipped when there is no room)
      const size_t bytes = sizeof(uint2) << (2 * (K - 4));
      if (pool_alloc(&kmer4, bytes) == hipSuccess)
        e = hipMemcpyAsync(kmer4, buf(j - 1), bytes, hipMemcpyDeviceToDevice, s);
      else
        (void)hipGetLastError();
      if (e != hipSuccess) break;
    }
    if (j + 1 < K)
      BWT_LAUNCH("kmer_level", k_kmer_level<false>, dim3((unsigned)div_up(parents, 256)), dim3(256), 0, s, v,
                 buf(j), buf(j + 1), (uint32_t *)nullptr, j, (const uint32_t *)nullptr, 0u);
    else
      BWT_LAUNCH("kmer_level", k_kmer_level<true>, dim3((unsigned)div_up(parents, 256)), dim3(256), 0, s, v,
                 buf(j), (uint2 *)nullptr, tab, j, W == 2 ? (const uint32_t *)ix->d_sa : (const uint32_t *)nullptr,
                 sizebits);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) (void)hipDeviceSynchronize();
  pool_free(tmp_a);
  // the level two bases up stays with a two-word table (it was the last pass's input):
  // where a search finds that the last K bases of a read do not occur, it goes on from there
  const bool keep_level = e == hipSuccess && W == 2 && K >= 7;
  if (!keep_level) pool_free(tmp_b);
  if (e != hipSuccess) {
    pool_free(tab);
    pool_free(kmer4);
    return set_error(kErrHip, "k-mer table build: %s", hipGetErrorString(e));
  }
  ix->d_kmer2 = keep_level ? tmp_b : nullptr;
  ix->d_kmer4 = kmer4;
  ix->d_kmer = tab;
  ix->kmer_k = K;
  ix->kmer_w = W;
  ix->kmer_sizebits = sizebits;
  return kOk;
}

// Called when what make_fmi produces is in place: notes which search-only
// arrays this index can still get.
void index_defer_search_arrays(bwtgpu_index *ix) {
  ix->want_records = ix->d_sa && ix->d_text2 && !ix->d_isa;
  ix->want_kmer = !ix->d_kmer && kmer_table_k(ix->n, ix->d_sa != nullptr) != 0;
  ix->search_ready.store(!(ix->want_records || ix->want_kmer || ix->want_recover), std::memory_order_release);
}

int index_prepare_search(const bwtgpu_index *cix, hipStream_t s) {
  bwtgpu_index *ix = const_cast<bwtgpu_index *>(cix);
  std::lock_guard<std::mutex> lock(ix->prepare_mu);
  if (ix->search_ready.load(std::memory_order_acquire)) return kOk;
  int cur = -1;
  BWT_HIP(hipGetDevice(&cur));
  if (cur != ix->device) BWT_HIP(hipSetDevice(ix->device));
  int rc = kOk;
  if (ix->want_recover) {
    rc = index_recover_arrays(ix, s);
    if (rc == kOk && ix->d_sa) {
      ix->want_records = !ix->d_isa;
      ix->want_kmer = !ix->d_kmer && kmer_table_k(ix->n, true) != 0;
    }
  }
  if (rc == kOk && ix->want_records) rc = index_build_records(ix, s);
  if (rc == kOk && ix->want_kmer) rc = index_build_kmer_table(ix, s);  // synchronises the stream
  if (cur != ix->device) (void)hipSetDevice(cur);
  if (rc != kOk) return rc;
  ix->want_records = ix->want_kmer = ix->want_recover = false;
  ix->search_ready.store(true, std::memory_order_release);
  return kOk;
}

int index_alloc(bwtgpu_index **out, uint32_t n) {
  if (n == 0 || n > 0xFFFFFFF0u)
    return set_error(kErrArg, "text length %u out of range", n);
  bwtgpu_index *ix = new bwtgpu_index();
  ix->n = n;
  ix->nblocks = (uint64_t)n / 64 + 1;
  ix->nsym = (uint64_t)n / kSymBases + 1;
  BWT_HIP(hipGetDevice(&ix->device));
  hipError_t e = pool_alloc(&ix->d_blocks, ix->nblocks * sizeof(IndexBlock));
  if (e == hipSuccess) e = pool_alloc(&ix->d_sym, 4 * ix->nsym * sizeof(SymBlock));
  if (e == hipSuccess)
    e = pool_alloc(&ix->d_idxs, (((uint64_t)n >> kDevSampleShift) + 1) * sizeof(uint32_t));
  if (e != hipSuccess) {
    bwtgpu_index_free(ix);
    return set_error(kErrHip, "hipMalloc of index for n=%u failed: %s", n, hipGetErrorString(e));
  }
  *out = ix;
  return kOk;
}

// Planes of every block are in place; computes occ[] and C[].
int index_finish_from_planes(bwtgpu_index *ix, hipStream_t s, uint32_t *readback, const HostWords *more) {
  const uint64_t nb = ix->nblocks;
  const uint32_t ntiles = (uint32_t)div_up(nb, kTileBlocks);
  uint4 *d_tiles = nullptr;
  uint32_t *d_C = nullptr;
  BWT_HIP(pool_alloc(&d_tiles, (size_t)ntiles * sizeof(uint4) + 64));
  d_C = reinterpret_cast<uint32_t *>(d_tiles + ntiles);
  BWT_LAUNCH("rank_tile_reduce", k_rank_tile_reduce, dim3(ntiles), dim3(kTileThreads), 0, s,
             ix->d_blocks, nb, ix->n, d_tiles);
  const HostWords none;
  BWT_LAUNCH("rank_scan_tiles", k_rank_scan_tiles, dim3(1), dim3(1024), 0, s, d_tiles, ntiles, d_C, readback,
             readback && more ? more[0] : none, readback && more ? more[1] : none);
  BWT_LAUNCH("rank_apply", k_rank_apply, dim3(ntiles), dim3(kTileThreads), 0, s, ix->d_blocks, nb,
             ix->n, d_tiles);
  BWT_LAUNCH("rank_build_sym", k_build_sym, dim3((unsigned)div_up(ix->nsym, 256)), dim3(256), 0, s,
             ix->d_blocks, nb, ix->n, ix->d_sym, ix->nsym);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !readback) e = hipMemcpyAsync(ix->C, d_C, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) (void)hipDeviceSynchronize();
  else if (readback)
    for (int c = 0; c < 5; ++c) ix->C[c] = readback[c];
  pool_free(d_tiles);
  if (e != hipSuccess) return set_error(kErrHip, "rank table build: %s", hipGetErrorString(e));
  if (ix->C[4] != ix->n + 1)
    return set_error(kErrInternal, "rank table totals %u != n+1 = %u", ix->C[4], ix->n + 1);
  return kOk;
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

void bwtgpu_index_free(bwtgpu_index *ix) {
  if (!ix) return;
  // the buffers go back to the pool for the next index: nothing may still be using them
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != ix->device) (void)hipSetDevice(ix->device);
  (void)hipDeviceSynchronize();
  pool_free(ix->d_blocks);
  pool_free(ix->d_sym);
  pool_free(ix->d_idxs);
  pool_free(ix->d_sa);
  pool_free(ix->d_isa);
  pool_free(ix->d_text2);
  pool_free(ix->d_kmer);
  pool_free(ix->d_kmer2);
  pool_free(ix->d_kmer4);
  if (cur != ix->device) (void)hipSetDevice(cur);
  delete ix;
}

bwtgpu_index *bwtgpu_index_from_parts(const uint8_t *h_bwt, uint32_t n, uint32_t endloc,
                                      const uint32_t *h_idxs, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h_bwt) {
    set_error(kErrArg, "null input");
    return nullptr;
  }
  if (endloc > n) {
    set_error(kErrArg, "endloc %u beyond n=%u", endloc, n);
    return nullptr;
  }
  bwtgpu_index *ix = nullptr;
  if (index_alloc(&ix, n) != kOk) return nullptr;
  ix->endloc = endloc;
  const uint64_t nb = ix->nblocks;
  const size_t bwt_bytes = ((size_t)n + 3) / 4;
  uint8_t *d_bwt = nullptr;
  uint32_t *d_stored = nullptr;
  auto fail = [&](hipError_t e, const char *what) {
    set_error(kErrHip, "%s: %s", what, hipGetErrorString(e));
    if (d_bwt) (void)hipFree(d_bwt);
    if (d_stored) (void)hipFree(d_stored);
    bwtgpu_index_free(ix);
    return (bwtgpu_index *)nullptr;
  };
  hipError_t e = hipMalloc(&d_bwt, nb * 16);
  if (e != hipSuccess) return fail(e, "hipMalloc bwt staging");
  e = hipMemsetAsync(d_bwt, 0, nb * 16, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_bwt, h_bwt, bwt_bytes, hipMemcpyHostToDevice, s);
  const size_t nstored = (size_t)n / 32 + 1, ndense = ((size_t)n >> kDevSampleShift) + 1;
  if (e == hipSuccess) e = hipMalloc(&d_stored, nstored * 4);
  if (e == hipSuccess)
    e = h_idxs ? hipMemcpyAsync(d_stored, h_idxs, nstored * 4, hipMemcpyHostToDevice, s)
               : hipMemsetAsync(ix->d_idxs, 0, ndense * 4, s);
  if (e != hipSuccess) return fail(e, "index upload");
  {
    ScopedKernelTimer t("bwt_to_planes", s);
    hipLaunchKernelGGL(k_bwt_to_planes, dim3((unsigned)div_up(nb, 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint4 *>(d_bwt), ix->d_blocks, nb, n);
  }
  int rc = index_finish_from_planes(ix, s);
  ix->has_samples = h_idxs != nullptr;
  // the first search recomputes the suffix array and the text from BWT and samples,
  // then makes the search-only arrays a built index has (index_recover_arrays)
  ix->want_recover = h_idxs != nullptr && !getenv("BWT_MI355X_COMPACT") && n <= kPosRecordMaxN;
  if (rc == kOk) index_defer_search_arrays(ix);  // ... and the k-mer table
  if (rc == kOk && h_idxs) {
    // stored samples (every 32nd row) -> HBM samples (every 8th): walks along LF from the stored rows
    bool consistent = false;
    rc = chain_walk(view_of(ix), true, d_stored, nstored, ix->d_idxs, nullptr, ndense, &consistent, s);
    if (rc == kOk && !consistent)
      rc = set_error(kErrArg, "index parts are inconsistent: the walks along LF from the stored rows do not pass "
                              "every row once (wrong $ row, or BWT and sampled rows of different texts)");
  }
  if (rc == kOk && hipStreamSynchronize(s) != hipSuccess) rc = set_error(kErrHip, "index load failed");
  (void)hipFree(d_bwt);
  if (d_stored) (void)hipFree(d_stored);
  if (rc != kOk) {
    bwtgpu_index_free(ix);
    return nullptr;
  }
  return ix;
}

int bwtgpu_index_info(const bwtgpu_index *ix, uint32_t *n, uint32_t *endloc, uint32_t C[5]) {
  if (!ix) return set_error(kErrArg, "null index");
  if (n) *n = ix->n;
  if (endloc) *endloc = ix->endloc;
  if (C)
    for (int i = 0; i < 5; ++i) C[i] = ix->C[i];
  return kOk;
}

int bwtgpu_index_kmer_table(const bwtgpu_index *ix, uint32_t *k, uint32_t *entry_words) {
  if (!ix) return set_error(kErrArg, "null index");
  if (k) *k = ix->d_kmer ? ix->kmer_k : 0u;
  if (entry_words) *entry_words = ix->d_kmer ? ix->kmer_w : 0u;
  return kOk;
}

size_t bwtgpu_index_device_bytes(const bwtgpu_index *ix) {
  if (!ix) return 0;
  return ix->nblocks * sizeof(IndexBlock) + 4 * ix->nsym * sizeof(SymBlock) +
         (((size_t)ix->n >> kDevSampleShift) + 1) * 4 +
         (ix->d_sa ? ((size_t)ix->n + 1) * 4 + ((size_t)ix->n / 16 + 4) * 4 : 0) +
         (ix->d_isa ? pos_record_count(ix->n, ix->rec_wide) * sizeof(PosRecord) : 0) +
         (ix->d_kmer ? (((uint64_t)ix->kmer_w << (2 * ix->kmer_k)) + 2) * sizeof(uint32_t) + kKmerMaxGaps * sizeof(uint2)
                     : 0) +
         (ix->d_kmer2 ? (sizeof(uint2) << (2 * (ix->kmer_k - 2))) : 0) +
         (ix->d_kmer4 ? (sizeof(uint2) << (2 * (ix->kmer_k - 4))) : 0);
}

int bwtgpu_index_compact(bwtgpu_index *ix) {
  if (!ix) return set_error(kErrArg, "null index");
  BWT_HIP(hipSetDevice(ix->device));
  BWT_HIP(hipDeviceSynchronize());  // searches in flight may still read the arrays
  std::lock_guard<std::mutex> lock(ix->prepare_mu);
  // compacting is for saving memory: these go back to the driver, not to the pool
  pool_free(ix->d_sa);
  pool_free(ix->d_isa);
  pool_free(ix->d_text2);
  pool_free(ix->d_kmer);
  pool_free(ix->d_kmer2);
  pool_free(ix->d_kmer4);
  pool_release_idle();
  ix->d_sa = ix->d_isa = ix->d_text2 = nullptr;
  ix->d_kmer = nullptr;
  ix->d_kmer2 = ix->d_kmer4 = nullptr;
  ix->kmer_k = 0;
  ix->want_records = ix->want_kmer = ix->want_recover = false;
  ix->search_ready.store(true, std::memory_order_release);
  return kOk;
}

int bwtgpu_index_keep_lean(bwtgpu_index *ix) {
  if (!ix) return set_error(kErrArg, "null index");
  std::lock_guard<std::mutex> lock(ix->prepare_mu);
  if (ix->want_recover) {
    ix->want_recover = false;
    ix->search_ready.store(!(ix->want_records || ix->want_kmer), std::memory_order_release);
  }
  return kOk;
}

int bwtgpu_index_prepare_search(bwtgpu_index *ix, void *stream) {
  if (!ix) return set_error(kErrArg, "null index");
  return index_ready_for_search(ix, (hipStream_t)stream);
}

bwtgpu_index *bwtgpu_index_replicate(const bwtgpu_index *src, int device) {
  if (!src) {
    set_error(kErrArg, "null index");
    return nullptr;
  }
  int ndev = 0, cur = -1;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_error(kErrArg, "device %d out of range (have %d)", device, ndev);
    return nullptr;
  }
  if (hipGetDevice(&cur) != hipSuccess || hipSetDevice(device) != hipSuccess) {
    set_error(kErrHip, "cannot make device %d current", device);
    return nullptr;
  }
  bwtgpu_index *ix = nullptr;
  int rc = index_alloc(&ix, src->n);
  if (rc == kOk) {
    if (device != src->device) {  // direct access where the link allows it; the copies below work either way
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can)
        (void)hipDeviceEnablePeerAccess(src->device, 0);
      (void)hipGetLastError();  // "already enabled" is not an error
    }
    auto copy = [&](void *dst, const void *from, size_t bytes) {
      return device == src->device ? hipMemcpyAsync(dst, from, bytes, hipMemcpyDeviceToDevice, nullptr)
                                   : hipMemcpyPeerAsync(dst, device, from, src->device, bytes, nullptr);
    };
    const uint32_t n = src->n;
    const size_t rows = (size_t)n + 1, t2words = (size_t)n / 16 + 4;
    // the source's suffix array and text words stay as they are while we hold its lock
    std::lock_guard<std::mutex> lock(src->prepare_mu);
    hipError_t e = copy(ix->d_blocks, src->d_blocks, ix->nblocks * sizeof(IndexBlock));
    if (e == hipSuccess) e = copy(ix->d_sym, src->d_sym, 4 * ix->nsym * sizeof(SymBlock));
    if (e == hipSuccess) e = copy(ix->d_idxs, src->d_idxs, (((size_t)n >> kDevSampleShift) + 1) * sizeof(uint32_t));
    uint32_t *sa = nullptr, *text2 = nullptr;
    if (e == hipSuccess && src->d_sa && src->d_text2 && !getenv("BWT_MI355X_COMPACT")) {
      if (pool_alloc(&sa, rows * 4) == hipSuccess && pool_alloc(&text2, t2words * 4) == hipSuccess) {
        e = copy(sa, src->d_sa, rows * 4);
        if (e == hipSuccess) e = copy(text2, src->d_text2, t2words * 4);
      } else {  // no room: the copy searches on its rank blocks
        (void)hipGetLastError();
        pool_free(sa);
        pool_free(text2);
        sa = text2 = nullptr;
      }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
      (void)hipDeviceSynchronize();
      pool_free(sa);
      pool_free(text2);
      rc = set_error(kErrHip, "copying the index to device %d: %s", device, hipGetErrorString(e));
    } else {
      ix->endloc = src->endloc;
      for (int c = 0; c < 5; ++c) ix->C[c] = src->C[c];
      ix->has_samples = src->has_samples;
      ix->d_sa = sa;
      ix->d_text2 = text2;
      ix->want_recover = !sa && src->want_recover;
      index_defer_search_arrays(ix);
    }
  }
  (void)hipSetDevice(cur);
  if (rc != kOk) {
    if (ix) bwtgpu_index_free(ix);
    return nullptr;
  }
  return ix;
}

int bwtgpu_index_export(const bwtgpu_index *ix, uint8_t *h_bwt, uint32_t *h_idxs) {
  if (!ix) return set_error(kErrArg, "null index");
  BWT_HIP(hipSetDevice(ix->device));
  if (h_idxs) {
    // the stored form keeps every 32nd row (fileio.c:14)
    const uint64_t nstored = (uint64_t)ix->n / 32 + 1;
    uint32_t *d_stored = nullptr;
    BWT_HIP(hipMalloc(&d_stored, nstored * 4));
    BWT_LAUNCH("thin_samples", k_thin_samples, dim3((unsigned)div_up(nstored, 256)), dim3(256), 0, nullptr,
               ix->d_idxs, nstored, d_stored);
    hipError_t e = hipMemcpy(h_idxs, d_stored, nstored * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_stored);
    if (e != hipSuccess) return set_error(kErrHip, "sample export: %s", hipGetErrorString(e));
  }
  if (h_bwt) {
    uint4 *d_bwt = nullptr;
    BWT_HIP(hipMalloc(&d_bwt, ix->nblocks * 16));
    BWT_LAUNCH("planes_to_bwt", k_planes_to_bwt, dim3((unsigned)div_up(ix->nblocks, 256)), dim3(256),
               0, nullptr, ix->d_blocks, ix->nblocks, d_bwt);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
      e = hipMemcpy(h_bwt, d_bwt, ((size_t)ix->n + 3) / 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_bwt);
    if (e != hipSuccess) return set_error(kErrHip, "bwt export: %s", hipGetErrorString(e));
  }
  return kOk;
}

int bwtgpu_index_rank_rows16(const bwtgpu_index *ix, uint32_t *h_rows) {
  return bwtgpu_index_rank_rows(ix, 16, h_rows);
}

int bwtgpu_index_rank_rows(const bwtgpu_index *ix, uint32_t blocksize, uint32_t *h_rows) {
  if (!ix || !h_rows) return set_error(kErrArg, "null argument");
  if (blocksize == 0 || blocksize % 4) return set_error(kErrArg, "blocksize must be a multiple of 4");
  BWT_HIP(hipSetDevice(ix->device));
  const uint64_t rows = 1 + ((uint64_t)ix->n + blocksize - 1) / blocksize;
  uint4 *d_rows = nullptr;
  BWT_HIP(hipMalloc(&d_rows, rows * 16));
  BWT_LAUNCH("rank_rows", k_rank_rows, dim3((unsigned)div_up(rows, 256)), dim3(256), 0, nullptr,
             view_of(ix), ix->C[0], ix->C[1], ix->C[2], ix->C[3], blocksize, rows, d_rows);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(h_rows, d_rows, rows * 16, hipMemcpyDeviceToHost);
  (void)hipFree(d_rows);
  if (e != hipSuccess) return set_error(kErrHip, "rank rows export: %s", hipGetErrorString(e));
  return kOk;
}

}  // extern "C"
