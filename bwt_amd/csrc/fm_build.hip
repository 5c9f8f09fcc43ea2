// FM-index build on device (make_fmi, seqindex.c:161-189): suffix array, BWT
// (sprintcbwt, histsortcomp.c:220-250), sampled SA rows (seqindex.c:173-176;
// every 8th here, thinned to every 32nd on export), rank blocks and C[]
// (seq_index, seqindex.c:23-87, :183-187), and what an index keeps beyond that:
// the suffix array itself, its inverse, the text in words, the k-mer table.
// Only what write_index stores ever goes to the host.
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "fm_device.hpp"

namespace bwtgpu {

__device__ __forceinline__ uint32_t text_base(const uint8_t *__restrict__ t, uint64_t i) {
  return (t[i >> 2] >> (2 * (3 - (i & 3)))) & 3u;  // histsortcomp.c:20-23
}

// The row whose suffix starts at 0 has no preceding base: that is the $ of
// the BWT (histsortcomp.c:232-236).
__global__ void k_find_endloc(const uint32_t *__restrict__ sa, uint64_t rows,
                              uint32_t *__restrict__ endloc) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < rows && sa[j] == 0) *endloc = (uint32_t)j;
}

// One wave emits one 64-base block: lane k holds BWT position 64*b + k, whose
// SA row is one further along once past the $ row (histsortcomp.c:238-245);
// the two ballots are the block's bit-planes.
__global__ void __launch_bounds__(256)
k_emit_planes(const uint8_t *__restrict__ text, const uint32_t *__restrict__ sa, uint32_t n,
              const uint32_t *__restrict__ d_endloc, IndexBlock *__restrict__ blocks,
              uint64_t nblocks) {
  const uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u;
  if (b >= nblocks) return;  // wave-uniform
  const uint32_t endloc = *d_endloc;
  const uint64_t p = b * 64 + lane;
  uint32_t base = 0;
  if (p < n) {
    const uint64_t row = p + (p >= endloc ? 1 : 0);
    base = text_base(text, (uint64_t)sa[row] - 1);
  }
  const uint64_t lo = __ballot(base & 1u);
  const uint64_t hi = __ballot(base & 2u);
  if (lane == 0) {
    blocks[b].lo = lo;
    blocks[b].hi = hi;
  }
}

// Same, when the suffix-array build left the BWT symbol of every row behind
// (bw[r-1] for row r >= 1): a streaming pass, no gather.  Row 0 is the $ suffix,
// preceded by the last base of the text.
constexpr int kPlanesPerWave = 16;  // blocks one wave emits: its loads are issued together

__global__ void __launch_bounds__(256)
k_emit_planes_rows(const uint8_t *__restrict__ text, const uint8_t *__restrict__ bw, uint32_t n,
                   const uint32_t *__restrict__ d_endloc, IndexBlock *__restrict__ blocks,
                   uint64_t nblocks) {
  const uint64_t b0 = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kPlanesPerWave;
  const uint32_t lane = threadIdx.x & 63u;
  if (b0 >= nblocks) return;  // wave-uniform
  const uint32_t endloc = *d_endloc;
  uint32_t base[kPlanesPerWave];
#pragma unroll
  for (int k = 0; k < kPlanesPerWave; ++k) {
    const uint64_t p = (b0 + k) * 64 + lane;
    base[k] = 0;
    if (p < n) {
      const uint64_t row = p + (p >= endloc ? 1 : 0);
      base[k] = row ? bw[row - 1] : text_base(text, (uint64_t)n - 1);
    }
  }
  uint64_t mylo = 0, myhi = 0;  // lane k keeps block b0 + k: one store instruction for all of them
#pragma unroll
  for (int k = 0; k < kPlanesPerWave; ++k) {
    const uint64_t lo = __ballot(base[k] & 1u);
    const uint64_t hi = __ballot(base[k] & 2u);
    if (lane == (uint32_t)k) {
      mylo = lo;
      myhi = hi;
    }
  }
  if (lane < (uint32_t)kPlanesPerWave && b0 + lane < nblocks) {
    blocks[b0 + lane].lo = mylo;
    blocks[b0 + lane].hi = myhi;
  }
}

// Do the bit-planes that (text, SA) give equal the ones the index holds?
// Same work as k_emit_planes, compared instead of stored.
__global__ void __launch_bounds__(256)
k_check_planes(const uint8_t *__restrict__ text, const uint32_t *__restrict__ sa, uint32_t n,
               const uint32_t *__restrict__ d_endloc, const IndexBlock *__restrict__ blocks,
               uint64_t nblocks, uint32_t *__restrict__ mismatch) {
  const uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u;
  if (b >= nblocks) return;  // wave-uniform
  const uint32_t endloc = *d_endloc;
  const uint64_t p = b * 64 + lane;
  uint32_t base = 0;
  if (p < n) {
    const uint64_t row = p + (p >= endloc ? 1 : 0);
    base = text_base(text, (uint64_t)sa[row] - 1);
  }
  const uint64_t lo = __ballot(base & 1u);
  const uint64_t hi = __ballot(base & 2u);
  if (lane == 0 && (blocks[b].lo != lo || blocks[b].hi != hi)) *mismatch = 1u;
}

__global__ void k_inverse_sa(const uint32_t *__restrict__ sa, uint64_t rows, uint32_t *__restrict__ isa, int wide) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < rows) isa[isa_slot(sa[j], wide != 0)] = (uint32_t)j;
}

// Inverse of a suffix array, in two steps so that the 4-byte scatter stays
// inside a window a cache holds (one XCD's L2 up to about 10^9 rows, the
// Infinity Cache beyond): first the
// (position, row) pairs are dealt into up to 256 bins by the top bits of the
// position -- bin b owns positions [b << shift, (b+1) << shift), so its pairs
// fill exactly that range of the pair buffer and no counting pass is needed --
// then the bins are emptied in order.
constexpr int kIsaThreads = 1024;  // the 64 KB of staging allow two workgroups per CU: 1024 threads fill every wave slot
constexpr int kIsaItems = 8;
constexpr int kIsaTile = kIsaThreads * kIsaItems;  // 8192 pairs: 32-pair (256-byte) runs per bin

__global__ void k_isa_cursors1(uint32_t shift, unsigned long long *__restrict__ cursors) {
  cursors[threadIdx.x] = (unsigned long long)threadIdx.x << shift;  // bin b starts at slot b << shift
}

// Second level for large arrays: the pairs of each first-level bin (a run of
// 1 << shift1 slots, tiles never straddle two) are dealt into 256 sub-bins by the
// next eight bits of the position; sub-bin (b, c) owns slots
// [(b << shift1) + (c << shift2), ...) of the output.
__global__ void k_isa_cursors2(uint32_t shift1, uint32_t shift2, unsigned long long *__restrict__ cursors) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // 65536 of them
  cursors[i] = ((unsigned long long)(i >> 8) << shift1) + ((unsigned long long)(i & 255u) << shift2);
}

// One tile of pairs into its 256 bins: tile-local bin order in LDS, one cursor
// claim per (tile, bin), runs written out whole.
// LEVEL 1: the tile is kIsaTile rows of the suffix array, bin = position >> shift,
//          cursors[256].
// LEVEL 2: the tile is kIsaTile pairs of one first-level bin (row0 >> shift1),
//          bin = (position >> shift) & 255, cursors[256][256].
template <int LEVEL>
__global__ void __launch_bounds__(kIsaThreads)
k_isa_deal(const uint32_t *__restrict__ sa, const uint64_t *__restrict__ in, uint64_t rows, uint32_t shift1,
           uint32_t shift, unsigned long long *__restrict__ cursors, uint64_t *__restrict__ out) {
  __shared__ uint32_t hist[256], lstart[256], fill[256], wsum[4];
  __shared__ unsigned long long gbase[256];
  __shared__ uint64_t stage[kIsaTile];
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const uint64_t row0 = (uint64_t)blockIdx.x * kIsaTile;
  unsigned long long *cur = LEVEL == 1 ? cursors : cursors + (row0 >> shift1) * 256u;
  if (t < 256) {
    hist[t] = 0;
    fill[t] = 0;
  }
  __syncthreads();
  // a thread's items are neighbours in memory, 16 bytes to a load: four rows of the suffix array
  // (level 1) or two pairs (level 2); which item goes to which thread is free
  constexpr int kPer = LEVEL == 1 ? 4 : 2;
  static_assert(kIsaItems % kPer == 0, "whole loads per thread");
  auto item_row = [&](int i) {
    return row0 + (uint64_t)kPer * ((uint64_t)(i / kPer) * kIsaThreads + t) + (uint64_t)(i % kPer);
  };
  uint64_t pr[kIsaItems];
#pragma unroll
  for (int u = 0; u < kIsaItems / kPer; ++u) {
    const uint64_t j0 = item_row(kPer * u);
    if (j0 + kPer <= rows) {
      if (LEVEL == 1) {
        const uint4 q = *reinterpret_cast<const uint4 *>(sa + j0);
        pr[4 * u] = (j0 << 32) | q.x;
        pr[4 * u + 1] = ((j0 + 1) << 32) | q.y;
        pr[4 * u + 2] = ((j0 + 2) << 32) | q.z;
        pr[4 * u + 3] = ((j0 + 3) << 32) | q.w;
      } else {
        const uint4 q = *reinterpret_cast<const uint4 *>(in + j0);
        pr[2 * u] = ((uint64_t)q.y << 32) | q.x;
        pr[2 * u + 1] = ((uint64_t)q.w << 32) | q.z;
      }
    } else {
#pragma unroll
      for (int e = 0; e < kPer; ++e) {
        const uint64_t j = j0 + e;
        pr[kPer * u + e] = j < rows ? (LEVEL == 1 ? (j << 32) | sa[j] : in[j]) : 0ull;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kIsaItems; ++i)
    if (item_row(i) < rows) atomicAdd(&hist[((uint32_t)pr[i] >> shift) & 255u], 1u);
  __syncthreads();
  uint32_t c = 0, v = 0;
  if (t < 256) {
    c = hist[t];
    v = c;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    if (lane == 63) wsum[wave] = v;
  }
  __syncthreads();
  if (t < 256) {
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wave; ++w) pre += wsum[w];
    lstart[t] = pre + v - c;
    gbase[t] = c ? atomicAdd(&cur[t], (unsigned long long)c) : 0ull;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kIsaItems; ++i) {
    if (item_row(i) < rows) {
      const uint32_t b = ((uint32_t)pr[i] >> shift) & 255u;
      stage[lstart[b] + atomicAdd(&fill[b], 1u)] = pr[i];
    }
  }
  __syncthreads();
  const uint32_t placed = lstart[255] + hist[255];
  for (uint32_t j = t; j < placed; j += kIsaThreads) {
    const uint64_t p = stage[j];
    const uint32_t bin = ((uint32_t)p >> shift) & 255u;
    out[gbase[bin] + (j - lstart[bin])] = p;
  }
}

// Workgroups go to the eight XCDs round-robin, and each XCD has its own L2:
// workgroup i takes the chunk of pairs that keeps every XCD inside its own
// eighth of the pair buffer, so the 4-byte stores into one window meet in one L2
// and leave it as whole lines.
// That pays while a window fits an L2 (xcd_local); larger windows live in the
// Infinity Cache, which all XCDs share, and are then filled one after the other
// by all of them.
__global__ void k_isa_fill(const uint64_t *__restrict__ pairs, uint64_t rows,
                           uint32_t *__restrict__ isa, int xcd_local, int wide) {
  const uint32_t per_xcd = (gridDim.x + 7u) / 8u;
  const uint64_t chunk =
      xcd_local ? (uint64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : (uint64_t)blockIdx.x;
  const uint64_t j = chunk * blockDim.x + threadIdx.x;
  if (j >= rows) return;
  const uint64_t pr = pairs[j];
  isa[isa_slot((uint32_t)pr, wide != 0)] = (uint32_t)(pr >> 32);
}

// The text part of every position record: the 64 bases from the record's first
// position on, zeros before the text and from base n on.  (Written apart from
// the rows: the threads that fill in the rows hold them in no order, and having
// them fetch the text as well cost more than this pass; and after them: run
// first, its lines were still in the caches when the pairs were being dealt.)
__global__ void k_record_text(const uint32_t *__restrict__ text2, uint32_t n, uint64_t nrec,
                              uint32_t *__restrict__ recs, int wide) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nrec) return;
  if (wide) {  // 128 bases behind the eight rows
    const int64_t first = ((int64_t)b - kWideGuardRecords) * kWidePosPerRecord;
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t at = first + 16 * k;
      w[k] = (at <= -16 || at >= (int64_t)n) ? 0u : text16(text2, at);
    }
    reinterpret_cast<uint4 *>(recs)[b * 4 + 2] = make_uint4(w[0], w[1], w[2], w[3]);
    reinterpret_cast<uint4 *>(recs)[b * 4 + 3] = make_uint4(w[4], w[5], w[6], w[7]);
    return;
  }
  const int64_t first = ((int64_t)b - kPosGuardRecords) * kPosPerRecord;
  uint32_t w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t at = first + 16 * k;
    w[k] = (at <= -16 || at >= (int64_t)n) ? 0u : text16(text2, at);
  }
  reinterpret_cast<uint4 *>(recs)[pos_record_place(b) * 4 + 3] = make_uint4(w[0], w[1], w[2], w[3]);
}

// The position records at d_isa: the inverse of the n+1 rows at d_sa and the
// text from d_text2.  scratch (may be null): 8 (n+1) + 2048 bytes that nothing
// else is using.
static int inverse_suffix_array(const uint32_t *d_sa, uint32_t n, const uint32_t *d_text2, uint32_t *d_isa,
                                bool wide, void *scratch, hipStream_t s) {
  const uint64_t rows = (uint64_t)n + 1;
  const uint64_t rec_bytes = wide ? 8 : 6;  // of records a position, rounded up
  auto record_text = [&]() {
    BWT_LAUNCH("record_text", k_record_text, dim3((unsigned)div_up(pos_record_count(n, wide), 256)), dim3(256), 0, s,
               d_text2, n, pos_record_count(n, wide), d_isa, wide ? 1 : 0);
  };
  // small arrays sit in L2 whatever the order of the stores
  const char *env = getenv("BWT_MI355X_ISA_TWO_STEP");  // for tests on small texts: 1 = one level, 2 = two
  const bool force = env != nullptr, force2 = env && env[0] == '2';
  if (!scratch || (rows <= (1ull << 22) && !force)) {
    BWT_LAUNCH("inverse_sa", k_inverse_sa, dim3((unsigned)div_up(rows, 256)), dim3(256), 0, s, d_sa, rows, d_isa,
               wide ? 1 : 0);
    record_text();
    return kOk;
  }
  uint32_t shift = 0;
  while (((uint64_t)n >> shift) > 255) ++shift;
  unsigned long long *d_cur = reinterpret_cast<unsigned long long *>(scratch);
  uint64_t *pairs = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(scratch) + 2048);
  BWT_LAUNCH("isa_cursors", k_isa_cursors1, dim3(1), dim3(256), 0, s, shift, d_cur);
  BWT_LAUNCH("isa_partition", k_isa_deal<1>, dim3((unsigned)div_up(rows, kIsaTile)), dim3(kIsaThreads), 0, s,
             d_sa, (const uint64_t *)nullptr, rows, 0u, shift, d_cur, pairs);
  const uint64_t *filled = pairs;
  uint32_t window_shift = shift;
  uint64_t *pairs2 = nullptr;
  unsigned long long *d_cur2 = nullptr;
  if (((rec_bytes << shift) > (2ull << 20) || force2) && shift >= 13) {  // tiles of 2^13 pairs must not straddle bins
    // windows beyond an L2: split each bin once more, if there is room for a second pair buffer
    if (pool_alloc(&pairs2, rows * 8) == hipSuccess && pool_alloc(&d_cur2, 65536 * 8) == hipSuccess) {
      const uint32_t shift2 = shift - 8;
      BWT_LAUNCH("isa_cursors2", k_isa_cursors2, dim3(256), dim3(256), 0, s, shift, shift2, d_cur2);
      BWT_LAUNCH("isa_partition2", k_isa_deal<2>, dim3((unsigned)div_up(rows, kIsaTile)), dim3(kIsaThreads), 0, s,
                 (const uint32_t *)nullptr, pairs, rows, shift, shift2, d_cur2, pairs2);
      filled = pairs2;
      window_shift = shift2;
    } else {
      (void)hipGetLastError();
    }
  }
  const int xcd_local = (rec_bytes << window_shift) <= (2ull << 20);  // a window of records within half an L2
  BWT_LAUNCH("isa_fill", k_isa_fill, dim3((unsigned)(div_up(div_up(rows, 256), 8) * 8)), dim3(256), 0, s, filled,
             rows, d_isa, xcd_local, wide ? 1 : 0);
  record_text();
  hipError_t e = hipGetLastError();
  if (pairs2 || d_cur2) {
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // before the buffers go back to the pool
    else (void)hipDeviceSynchronize();
    pool_free(pairs2);
    pool_free(d_cur2);
  }
  if (e != hipSuccess) return set_error(kErrHip, "inverse suffix array: %s", hipGetErrorString(e));
  return kOk;
}

// text as 16-base words with base k at bits 2k (the layout of packed reads),
// one zero guard word in front, zeros from base n on
__global__ void k_make_text2(const uint8_t *__restrict__ text, uint32_t n, uint64_t nwords,
                             uint32_t *__restrict__ text2) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint32_t out = 0;
  if (w >= 1) {
    const uint64_t first = 16 * (w - 1);
    const uint64_t nbytes = (uint64_t)n / 4 + 1;
    if (first + 16 <= n && !((uintptr_t)text & 3u)) {
      // four whole bytes: one load, then the four bases of every byte change places
      uint32_t v = reinterpret_cast<const uint32_t *>(text)[w - 1];
      v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
      out = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    } else {
      for (uint32_t k = 0; k < 16; ++k) {
        const uint64_t p = first + k;
        if (p < n && (p >> 2) < nbytes) out |= text_base(text, p) << (2 * k);
      }
    }
  }
  text2[w] = out;
}

__global__ void k_sample_sa(const uint32_t *__restrict__ sa, uint64_t nsamples,
                            uint32_t *__restrict__ idxs) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nsamples) idxs[i] = sa[i << kDevSampleShift];
}

// sprintcbwt's own output format: four bases per byte, first base in the top
// bits, $ row skipped, zero padding in the last byte.  One thread per byte.
__global__ void k_emit_bwt_bytes(const uint8_t *__restrict__ text, const uint32_t *__restrict__ sa,
                                 uint32_t n, const uint32_t *__restrict__ d_endloc,
                                 uint8_t *__restrict__ bwt) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= ((uint64_t)n + 3) / 4) return;
  const uint32_t endloc = *d_endloc;
  uint32_t acc = 0;
  for (int k = 0; k < 4; ++k) {
    const uint64_t p = 4 * q + k;
    if (p < n) {
      const uint64_t row = p + (p >= endloc ? 1 : 0);
      acc |= text_base(text, (uint64_t)sa[row] - 1) << (2 * (3 - k));
    }
  }
  bwt[q] = (uint8_t)acc;
}

// The suffix array and the build workspace are transient and large (about
// 21 bytes per base); allocating and freeing them costs more than a millisecond
// per build, so the calling thread keeps the largest pair it has used on its
// device.  Every thread's cache is on a list: bwtgpu_release_cache() frees them
// all (a cache whose thread is inside a build at that moment is skipped), and a
// thread that ends hands its buffers to the idle pool, from where the next
// request of that size -- or the pool's byte cap -- takes them.
struct BuildCache;
static std::mutex g_caches_mu;
static std::vector<BuildCache *> g_caches;

// A build reads a few words back from the device (the sort's counters, the $
// row, C[]).  Copied into pageable memory each of them waits for the stream;
// copied into pinned memory they all ride on the one wait at the end.  One
// pinned page for the process, a 64-word slot of it per building thread (a
// thread beyond the 16th builds with pageable copies); the page is never freed.
static std::mutex g_readback_mu;
static uint32_t *g_readback_page = nullptr;
static uint32_t g_readback_free = 0xFFFFu;

static uint32_t *readback_acquire(int *slot) {
  std::lock_guard<std::mutex> lock(g_readback_mu);
  if (!g_readback_page) {
    void *p = nullptr;
    if (hipHostMalloc(&p, 4096, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    // kernels store into it through the same address
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || d != p) {
      (void)hipGetLastError();
      (void)hipHostFree(p);
      return nullptr;
    }
    g_readback_page = static_cast<uint32_t *>(p);
  }
  if (!g_readback_free) return nullptr;
  *slot = __builtin_ctz(g_readback_free);
  g_readback_free &= g_readback_free - 1;
  return g_readback_page + 64 * *slot;
}

static void readback_release(int slot) {
  std::lock_guard<std::mutex> lock(g_readback_mu);
  g_readback_free |= 1u << slot;
}

struct BuildCache {
  int device = -1;
  void *sa = nullptr, *ws = nullptr, *small = nullptr;
  size_t sa_bytes = 0, ws_bytes = 0;
  uint32_t *readback = nullptr;  // this thread's slot of the pinned page, or null
  int readback_slot = -1;
  std::mutex busy;  // held by the owning thread for the length of a build
  BuildCache() {
    std::lock_guard<std::mutex> lock(g_caches_mu);
    g_caches.push_back(this);
  }
  ~BuildCache() {
    {
      std::lock_guard<std::mutex> lock(g_caches_mu);
      for (size_t i = 0; i < g_caches.size(); ++i)
        if (g_caches[i] == this) {
          g_caches.erase(g_caches.begin() + (long)i);
          break;
        }
    }
    if (readback_slot >= 0) readback_release(readback_slot);
    // no HIP call here: the main thread gets here after the runtime may have shut down
    pool_adopt_idle(sa, sa_bytes, device);
    pool_adopt_idle(ws, ws_bytes, device);
    pool_adopt_idle(small, 4096, device);
  }
  void release() {
    if (sa) (void)hipFree(sa);
    if (ws) (void)hipFree(ws);
    if (small) (void)hipFree(small);
    sa = ws = small = nullptr;
    sa_bytes = ws_bytes = 0;
    device = -1;
  }
};
static thread_local BuildCache g_cache;

static int cache_reserve(size_t sa_bytes, size_t ws_bytes) {
  int dev = 0;
  BWT_HIP(hipGetDevice(&dev));
  if (g_cache.device != dev) g_cache.release();
  g_cache.device = dev;
  if (!g_cache.small) BWT_HIP(hipMalloc(&g_cache.small, 4096));
  if (!g_cache.readback) g_cache.readback = readback_acquire(&g_cache.readback_slot);
  if (g_cache.sa_bytes < sa_bytes) {
    if (g_cache.sa) (void)hipFree(g_cache.sa);
    g_cache.sa = nullptr;
    g_cache.sa_bytes = 0;
    BWT_HIP(hipMalloc(&g_cache.sa, sa_bytes));
    g_cache.sa_bytes = sa_bytes;
  }
  if (g_cache.ws_bytes < ws_bytes) {
    if (g_cache.ws) (void)hipFree(g_cache.ws);
    g_cache.ws = nullptr;
    g_cache.ws_bytes = 0;
    BWT_HIP(hipMalloc(&g_cache.ws, ws_bytes));
    g_cache.ws_bytes = ws_bytes;
  }
  return kOk;
}

// Allocates what an index keeps beside what make_fmi produces, unless asked not
// to or HBM is short: room for the whole suffix array (the sort then writes its
// rows there) and for the text in words.  The position records are made from
// the two when a search first needs them (index_build_records).
static void alloc_kept_arrays(bwtgpu_index *ix, uint32_t n) {
  if (getenv("BWT_MI355X_COMPACT") || n > kPosRecordMaxN) return;
  const uint64_t rows_all = (uint64_t)n + 1, t2words = (uint64_t)n / 16 + 4;
  hipError_t ea = pool_alloc(&ix->d_sa, rows_all * 4);
  if (ea == hipSuccess) ea = pool_alloc(&ix->d_text2, t2words * 4);
  if (ea != hipSuccess) {
    (void)hipGetLastError();
    pool_free(ix->d_sa);
    pool_free(ix->d_text2);
    ix->d_sa = ix->d_text2 = nullptr;
  }
}

// Everything after the suffix array: BWT bit-planes, sampled rows, the text in
// words, rank blocks, C[].  d_sa holds the n+1 rows (it is
// ix->d_sa when the index keeps them); bw, when not null, is the BWT symbol of
// rows 1..n and *d_endloc then already holds the $ row.
static int emit_index(bwtgpu_index *ix, const uint8_t *d_text, uint32_t n, const uint32_t *d_sa,
                      const uint8_t *bw, uint32_t *d_endloc, bool have_samples, bool have_text2, hipStream_t s,
                      uint32_t *readback = nullptr /* 16 pinned words, or null */,
                      const HostWords *counters = nullptr /* with readback: copied along, or null */) {
  const uint64_t rows = (uint64_t)n + 1, t2words = (uint64_t)n / 16 + 4;
  if (bw) {
    BWT_LAUNCH("emit_planes_rows", k_emit_planes_rows,
               dim3((unsigned)div_up(div_up(ix->nblocks, kPlanesPerWave) * 64, 256)),
               dim3(256), 0, s, d_text, bw, n, d_endloc, ix->d_blocks, ix->nblocks);
  } else {
    (void)hipMemsetAsync(d_endloc, 0xFF, 4, s);
    BWT_LAUNCH("find_endloc", k_find_endloc, dim3((unsigned)div_up(rows, 256)), dim3(256), 0, s, d_sa, rows,
               d_endloc);
    BWT_LAUNCH("emit_planes", k_emit_planes, dim3((unsigned)div_up(ix->nblocks * 64, 256)), dim3(256), 0, s,
               d_text, d_sa, n, d_endloc, ix->d_blocks, ix->nblocks);
  }
  const uint64_t ns = ((uint64_t)n >> kDevSampleShift) + 1;
  if (!have_samples)  // else the leaf sort dropped them as it wrote the rows
    BWT_LAUNCH("sample_sa", k_sample_sa, dim3((unsigned)div_up(ns, 256)), dim3(256), 0, s, d_sa, ns, ix->d_idxs);
  if (ix->d_sa && !have_text2)  // the caller's text may be gone by the time a search wants the records
    BWT_LAUNCH("make_text2", k_make_text2, dim3((unsigned)div_up(t2words, 256)), dim3(256), 0, s, d_text, n,
               t2words, ix->d_text2);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !readback) e = hipMemcpyAsync(&ix->endloc, d_endloc, 4, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return set_error(kErrHip, "BWT emission: %s", hipGetErrorString(e));
  HostWords more[2];
  more[0].dst = readback ? readback + 8 : nullptr;
  more[0].src = d_endloc;
  more[0].words = 1;
  if (counters) more[1] = *counters;
  BWT_TRY(index_finish_from_planes(ix, s, readback, more));  // synchronises the stream
  if (readback) ix->endloc = readback[8];
  if (ix->endloc > n) return set_error(kErrInternal, "no suffix-array row holds suffix 0");
  // what make_fmi produces is complete; the search-only arrays wait for the first search
  index_defer_search_arrays(ix);
  return kOk;
}

// The position records (see PosRecord) of an index that holds its suffix array
// and the text in words.  Not an error when HBM is short: the index then
// searches on its rank blocks alone.
int index_build_records(bwtgpu_index *ix, hipStream_t s) {
  if (!ix->d_sa || !ix->d_text2 || ix->d_isa) return kOk;
  const uint32_t n = ix->n;
  uint32_t *isa = nullptr;  // the index gets it when every record is written
  // Wide records (see kWidePosPerRecord) once the twelve-position kind is several times the
  // last-level cache anyway; BWT_MI355X_WIDE_RECORDS=0 / 1 says otherwise (tests)
  bool wide = n >= (1u << 27);
  if (const char *e = getenv("BWT_MI355X_WIDE_RECORDS")) wide = e[0] == '1';
  if (pool_alloc(&isa, pos_record_count(n, wide) * sizeof(PosRecord)) != hipSuccess) {
    (void)hipGetLastError();
    wide = false;
    if (pool_alloc(&isa, pos_record_count(n, wide) * sizeof(PosRecord)) != hipSuccess) {
      (void)hipGetLastError();
      return kOk;
    }
  }
  // scratch for the two-step scatter: the calling thread's idle build workspace
  // when it is large enough (the build that has just returned left it), else a
  // buffer of its own
  void *scratch = nullptr, *own = nullptr;
  int dev = -1;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> building(g_cache.busy);
  if (g_cache.device == dev && g_cache.ws && g_cache.ws_bytes >= bwtgpu_sa_workspace_bytes(n))
    scratch = sa_workspace_free_region(n, g_cache.ws);
  if (!scratch && (uint64_t)n + 1 > (1ull << 22)) {
    if (pool_alloc(&own, ((size_t)n + 1) * 8 + 2048) == hipSuccess) scratch = own;
    else (void)hipGetLastError();  // the one-step scatter needs none
  }
  int rc = inverse_suffix_array(ix->d_sa, n, ix->d_text2, isa, wide, scratch, s);
  if (rc == kOk && hipStreamSynchronize(s) != hipSuccess) rc = set_error(kErrHip, "position records failed");
  if (rc != kOk) (void)hipDeviceSynchronize();
  pool_free(own);
  if (rc != kOk) {
    pool_free(isa);
  } else {
    ix->rec_wide = wide;
    publish(ix->d_isa, isa);
  }
  return rc;
}

static int build_from_device_text(const uint8_t *d_text, uint32_t n, hipStream_t s,
                                  bwtgpu_index **out) {
  std::lock_guard<std::mutex> building(g_cache.busy);
  bwtgpu_index *ix = nullptr;
  BWT_TRY(index_alloc(&ix, n));
  const size_t ws_bytes = bwtgpu_sa_workspace_bytes(n);
  alloc_kept_arrays(ix, n);
  int rc = cache_reserve(ix->d_sa ? 256 : ((size_t)n + 1) * 4, ws_bytes);
  if (rc != kOk) {
    bwtgpu_index_free(ix);
    return rc;
  }
  uint32_t *d_sa = ix->d_sa ? ix->d_sa : reinterpret_cast<uint32_t *>(g_cache.sa);
  uint32_t *d_endloc = reinterpret_cast<uint32_t *>(g_cache.small);
  SaExtras extras;
  extras.samples = ix->d_idxs;
  uint32_t *readback = g_cache.readback;  // words 0..15: the sort's counters; 16..31: emit_index
  extras.readback = readback;
  extras.text2 = ix->d_sa ? ix->d_text2 : nullptr;
  extras.caller_copies_counters = readback != nullptr;
  rc = sa_build_impl(d_text, n, d_sa, g_cache.ws, ws_bytes, s, &extras);
  if (rc == kOk) {
    const bool assumed = extras.pending;  // ... that the leaf sort left no rows tied
    const bool have_text2 = extras.text2_done;
    HostWords counters;
    if (assumed && readback) {
      counters.dst = readback;
      counters.src = static_cast<const uint32_t *>(extras.later.ctr);
      counters.words = 16;
    }
    // the $ row: where the leaf sort left it, when it wrote the BWT symbols too
    uint32_t *endloc_at = extras.bw ? const_cast<uint32_t *>(extras.endloc) : d_endloc;
    rc = emit_index(ix, d_text, n, d_sa, extras.bw, endloc_at, extras.samples_done, have_text2, s,
                    readback ? readback + 16 : nullptr, assumed && readback ? &counters : nullptr);
    if (assumed) {
      if (rc != kOk) {  // emit_index waits, and its last kernels bring the counters along, only when all went well
        (void)hipStreamSynchronize(s);
        if (readback) (void)hipMemcpy(readback, extras.later.ctr, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost);
      }
      const int settled = sa_build_settle(&extras, s);
      if (settled < 0) rc = settled;
      else if (settled > 0)  // it had: the rows have changed, emit again from the final suffix array
        rc = emit_index(ix, d_text, n, d_sa, nullptr, d_endloc, false, have_text2, s,
                        readback ? readback + 16 : nullptr);
    }
  }
  if (rc != kOk) {
    bwtgpu_index_free(ix);
    return rc;
  }
  *out = ix;
  return kOk;
}

// 256-bin histogram of the first four bases of every suffix, straight from the
// packed text (bases past the end read as A, as in the sort).
__global__ void __launch_bounds__(256)
k_bucket_counts(const uint8_t *__restrict__ text, uint32_t n, unsigned long long *__restrict__ out) {
  __shared__ uint32_t hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t nbytes = (uint64_t)n / 4 + 1;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; 4 * q < n; q += stride) {
    // the four suffixes that start in byte q
    uint32_t w = (uint32_t)text[q] << 8;
    if (q + 1 < nbytes) w |= text[q + 1];
    // bits from base n on are not the text's: clear them
    const uint64_t left = (uint64_t)n - 4 * q;  // bases from 4q to the end, >= 1
    if (left < 8) w &= ~((1u << (16 - 2 * (uint32_t)left)) - 1u);
    for (uint32_t k = 0; k < 4 && 4 * q + k < n; ++k) atomicAdd(&hist[(w >> (8 - 2 * k)) & 255u], 1u);
  }
  __syncthreads();
  if (hist[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

int bwtgpu_release_cache(void) {
  {
    std::lock_guard<std::mutex> lock(g_caches_mu);
    for (BuildCache *c : g_caches) {
      if (!c->busy.try_lock()) continue;  // its thread is building: it keeps what it is using
      c->release();
      c->busy.unlock();
    }
  }
  pool_release_idle();
  pinned_release_idle();
  return kOk;
}

int bwtgpu_bwt_from_sa(const uint8_t *d_text, const uint32_t *d_sa, uint32_t n, uint8_t *d_bwt,
                       uint32_t *d_endloc, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!d_text || !d_sa || !d_bwt || !d_endloc || n == 0) return set_error(kErrArg, "bad argument");
  const uint64_t rows = (uint64_t)n + 1;
  BWT_HIP(hipMemsetAsync(d_endloc, 0xFF, 4, s));
  BWT_LAUNCH("find_endloc", k_find_endloc, dim3((unsigned)div_up(rows, 256)), dim3(256), 0, s, d_sa, rows,
             d_endloc);
  BWT_LAUNCH("emit_bwt_bytes", k_emit_bwt_bytes, dim3((unsigned)div_up(((uint64_t)n + 3) / 4, 256)),
             dim3(256), 0, s, d_text, d_sa, n, d_endloc, d_bwt);
  BWT_CHECK_LAUNCH();
  return kOk;
}

int bwtgpu_sa_bucket_counts(const uint8_t *d_text, uint32_t n, uint64_t h_counts[256], void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!d_text || !h_counts || n == 0) return set_error(kErrArg, "bad argument");
  unsigned long long *d_out = nullptr;
  BWT_HIP(hipMalloc(&d_out, 256 * sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d_out, 0, 256 * sizeof(unsigned long long), s);
  uint64_t groups = div_up(div_up(n, 4), 256 * 16);
  if (groups > 4096) groups = 4096;
  if (e == hipSuccess) {
    BWT_LAUNCH("sa_bucket_counts", k_bucket_counts, dim3((unsigned)groups), dim3(256), 0, s, d_text, n, d_out);
    e = hipGetLastError();
  }
  unsigned long long h[256];
  if (e == hipSuccess) e = hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d_out);
  if (e != hipSuccess) return set_error(kErrHip, "bucket counts: %s", hipGetErrorString(e));
  for (int i = 0; i < 256; ++i) h_counts[i] = h[i];
  return kOk;
}

int bwtgpu_sa_build_range(const uint8_t *d_text, uint32_t n, uint32_t bucket_lo, uint32_t bucket_hi,
                          uint32_t *d_rows, uint8_t *d_bw, uint64_t capacity, uint64_t *row_begin,
                          uint64_t *rows, uint32_t *endloc, void *d_ws, size_t ws_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!d_rows || !row_begin || !rows || !endloc) return set_error(kErrArg, "null argument");
  SaRange range;
  range.dlo = bucket_lo;
  range.dhi = bucket_hi;
  range.capacity = capacity;
  SaExtras extras;
  BWT_TRY(sa_build_core(d_text, n, d_rows, &range, d_ws, ws_bytes, s, &extras));
  if (d_bw && range.rows) {
    if (!extras.bw) return set_error(kErrInternal, "sharded build left no BWT symbols");
    BWT_HIP(hipMemcpyAsync(d_bw, extras.bw, range.rows, hipMemcpyDeviceToDevice, s));
  }
  BWT_HIP(hipStreamSynchronize(s));
  *row_begin = range.row_begin;
  *rows = range.rows;
  *endloc = range.endloc;
  return kOk;
}

bwtgpu_index *bwtgpu_index_build_from_sa(const uint8_t *d_text, uint32_t n, const uint32_t *d_sa,
                                         const uint8_t *d_bw, uint32_t endloc, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!d_text || !d_sa || n == 0 || (d_bw && (endloc == 0 || endloc > n))) {
    set_error(kErrArg, "bad argument");
    return nullptr;
  }
  std::lock_guard<std::mutex> building(g_cache.busy);
  bwtgpu_index *ix = nullptr;
  if (index_alloc(&ix, n) != kOk) return nullptr;
  alloc_kept_arrays(ix, n);
  int rc = cache_reserve(256, 256);
  uint32_t *d_endloc = reinterpret_cast<uint32_t *>(g_cache.small);
  if (rc == kOk && ix->d_sa &&
      hipMemcpyAsync(ix->d_sa, d_sa, ((size_t)n + 1) * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
    rc = set_error(kErrHip, "suffix array copy failed");
  if (rc == kOk && d_bw && hipMemcpyAsync(d_endloc, &endloc, 4, hipMemcpyHostToDevice, s) != hipSuccess)
    rc = set_error(kErrHip, "endloc upload failed");
  if (rc == kOk) rc = emit_index(ix, d_text, n, ix->d_sa ? ix->d_sa : d_sa, d_bw, d_endloc, false, false, s);
  if (rc != kOk) {
    bwtgpu_index_free(ix);
    return nullptr;
  }
  return ix;
}

int bwtgpu_index_attach_text_dev(bwtgpu_index *ix, const uint8_t *d_text, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ix || !d_text) return set_error(kErrArg, "null argument");
  if (ix->d_sa) return kOk;  // already there
  BWT_HIP(hipSetDevice(ix->device));
  const uint32_t n = ix->n;
  const uint64_t rows = (uint64_t)n + 1, t2words = (uint64_t)n / 16 + 4;
  if (n > kPosRecordMaxN) return set_error(kErrArg, "text of %u bases is too long to keep beside its index", n);
  std::lock_guard<std::mutex> lock(ix->prepare_mu);
  if (ix->d_sa) return kOk;  // another thread attached it meanwhile
  // in buffers of our own until they are complete and checked (see bwtgpu_index::prepare_mu)
  uint32_t *sa = nullptr, *text2 = nullptr;
  hipError_t ea = pool_alloc(&sa, rows * 4);
  if (ea == hipSuccess) ea = pool_alloc(&text2, t2words * 4);
  auto drop = [&]() {
    (void)hipDeviceSynchronize();
    pool_free(sa);
    pool_free(text2);
  };
  if (ea != hipSuccess) {
    (void)hipGetLastError();
    drop();
    return set_error(kErrHip, "no room for the suffix array of %u bases: %s", n, hipGetErrorString(ea));
  }
  const size_t ws_bytes = bwtgpu_sa_workspace_bytes(n);
  std::lock_guard<std::mutex> building(g_cache.busy);
  int rc = cache_reserve(256, ws_bytes);
  if (rc == kOk) rc = sa_build_impl(d_text, n, sa, g_cache.ws, ws_bytes, s, nullptr);
  if (rc == kOk) {
    // the suffix array of this text must give the BWT the index was loaded with
    uint32_t *d_endloc = reinterpret_cast<uint32_t *>(g_cache.small);
    uint32_t *d_flag = d_endloc + 1;
    uint32_t h[2] = {0xFFFFFFFFu, 1u};
    hipError_t e = hipMemsetAsync(d_endloc, 0xFF, 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, 4, s);
    if (e == hipSuccess) {
      BWT_LAUNCH("find_endloc", k_find_endloc, dim3((unsigned)div_up(rows, 256)), dim3(256), 0, s, sa, rows,
                 d_endloc);
      BWT_LAUNCH("check_planes", k_check_planes, dim3((unsigned)div_up(ix->nblocks * 64, 256)), dim3(256), 0, s,
                 d_text, sa, n, d_endloc, ix->d_blocks, ix->nblocks, d_flag);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_endloc, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = set_error(kErrHip, "text check: %s", hipGetErrorString(e));
    else if (h[0] != ix->endloc || h[1] != 0)
      rc = set_error(kErrArg, "this text is not the one the index was built from");
  }
  if (rc == kOk) {
    BWT_LAUNCH("make_text2", k_make_text2, dim3((unsigned)div_up(t2words, 256)), dim3(256), 0, s, d_text, n,
               t2words, text2);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      rc = set_error(kErrHip, "attaching the text failed");
  }
  if (rc != kOk) {
    drop();
    return rc;
  }
  ix->d_text2 = text2;
  publish(ix->d_sa, sa);
  // the position records follow when a search first wants them
  ix->want_records = true;
  ix->search_ready.store(false, std::memory_order_release);
  return kOk;
}

int bwtgpu_index_attach_text(bwtgpu_index *ix, const uint8_t *h_text, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ix || !h_text) return set_error(kErrArg, "null argument");
  if (ix->d_sa) return kOk;
  BWT_HIP(hipSetDevice(ix->device));
  const size_t nbytes = (size_t)ix->n / 4 + 1;
  uint8_t *d_text = nullptr;
  BWT_HIP(hipMalloc(&d_text, nbytes + 64));
  int rc = kOk;
  if (hipMemcpyAsync(d_text, h_text, nbytes, hipMemcpyHostToDevice, s) != hipSuccess)
    rc = set_error(kErrHip, "text upload failed");
  if (rc == kOk) rc = bwtgpu_index_attach_text_dev(ix, d_text, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(d_text);
  return rc;
}

bwtgpu_index *bwtgpu_index_build_dev(const uint8_t *d_text, uint32_t n, void *stream) {
  if (!d_text) {
    set_error(kErrArg, "null text");
    return nullptr;
  }
  bwtgpu_index *ix = nullptr;
  if (build_from_device_text(d_text, n, (hipStream_t)stream, &ix) != kOk) return nullptr;
  return ix;
}

bwtgpu_index *bwtgpu_index_build(const uint8_t *h_text, uint32_t n, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h_text || n == 0) {
    set_error(kErrArg, "bad text");
    return nullptr;
  }
  const size_t nbytes = (size_t)n / 4 + 1;
  uint8_t *d_text = nullptr;
  hipError_t e = hipMalloc(&d_text, nbytes + 64);
  if (e == hipSuccess) e = hipMemcpyAsync(d_text, h_text, nbytes, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) {
    if (d_text) (void)hipFree(d_text);
    set_error(kErrHip, "text upload: %s", hipGetErrorString(e));
    return nullptr;
  }
  bwtgpu_index *ix = nullptr;
  int rc = build_from_device_text(d_text, n, s, &ix);
  (void)hipStreamSynchronize(s);
  (void)hipFree(d_text);
  return rc == kOk ? ix : nullptr;
}

}  // extern "C"
