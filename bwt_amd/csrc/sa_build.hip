// Suffix-array construction for 2-bit packed texts on gfx950.
//
// Replaces histsort (histsortcomp.c:161-180) and csuff_arr (csacak.c:521-528).
// The reference sorts one base per level by count + scatter (histhelper,
// histsortcomp.c:35-90).  Here a level consumes four bases (256 buckets):
//   1. k_part_hist     LDS-privatised k-mer histogram per 8192-suffix tile,
//                      merged into per-segment counters
//   2. k_part_classify per-segment prefix sum -> child bucket cursors; children
//                      above kLeafMax rows go to the next level, runs of small
//                      neighbours are merged into leaves
//   3. k_part_scatter  tile-local bucket order in LDS, then runs of each
//                      bucket are written to the segment's cursors
// Leaves (<= kLeafMax suffixes sharing `depth` bases) are sorted in LDS by the
// next 16 bases (k_leaf_sort).  Suffixes that still tie, and buckets that never
// shrink (repeats), go to prefix doubling (sa_doubling below), which is what
// keeps tandem repeats from costing O(n * LCP) as they do in the reference.
//
// Order being computed: suffixes compare as their bases followed by an
// end-of-text symbol smaller than A (histsortcomp.c:54-64 puts the suffix that
// ends first).  Padding the text with A (zeros) and breaking exact ties by
// "larger start first" gives the same order, so keys never need a fifth symbol.
#include <cstring>

#include "common.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

namespace bwtgpu {

constexpr int kPartThreads = 256;                      // histogram workgroup
constexpr int kScatThreads = 512;                      // scatter workgroup
constexpr int kPartItems = 16;                         // suffixes per scatter thread
constexpr int kPartTile = kScatThreads * kPartItems;   // 8192 suffixes per tile: 32-element runs per bucket
constexpr int kScat1Threads = 1024;                    // scatter of the levels after the first: the same tile, more waves
constexpr int kScat1Items = kPartTile / kScat1Threads;
constexpr int kHistItems = kPartTile / kPartThreads;
constexpr uint32_t kLeafMax = 4096;                   // suffixes one workgroup sorts in LDS
constexpr uint32_t kWaveLeafMax = 768;                // ... and one wave on its own: small children are merged up to this
constexpr uint32_t kWaveLeafMid = 1024;               // a child of up to this many rows is still one wave's (fewer waves to a CU)
constexpr uint32_t kDigitBases = 4;                   // bases consumed per level (256 buckets)
constexpr uint32_t kMaxDepth = 24;                    // bucket levels stop here; doubling takes over
constexpr uint32_t kLeafKeyBases = 15;                // bases in a leaf sort key
constexpr uint32_t kTextPadWords = 8;

enum : uint32_t { kSrcA = 0, kSrcB = 1, kSrcImplicit = 2 };

struct Seg {
  uint32_t start, size;  // rows [start, start+size) of SA+1 (row 0 of the SA is the $ suffix)
};
struct Leaf {
  uint32_t start, size, depth;
  uint32_t src;  // low byte: kSrcA / kSrcB / kSrcImplicit; next byte: bases of text in the stored sort keys
};
__host__ __device__ inline uint32_t leaf_src_word(uint32_t src, uint32_t keybases) { return src | (keybases << 8); }

struct Counters {
  uint32_t nnext;       // segments emitted for the next level
  uint32_t nleaves;
  uint32_t unresolved;  // rows left in tie groups
  uint32_t min_depth;   // smallest number of bases any tie group is sorted by
  uint32_t overflow;    // a list ran out of room (reported as an error)
  uint32_t total_tiles;
  uint32_t active;      // doubling: rows still in groups
  uint32_t endloc;      // SA row whose suffix starts at 0 (the $ of the BWT)
  uint32_t ncur;        // segments of the level being partitioned (k_begin_level)
  uint32_t nbig;        // leaves of more than kWaveLeafMid rows (listed for the workgroup-wide leaf sort)
  uint32_t nmid;        // leaves of kWaveLeafMax + 1 .. kWaveLeafMid rows (listed for the one-wave sort with room for them)
  uint32_t ndeferred;   // leaves the one-wave leaf sort left to the workgroup-wide one when it met them
};

// ---- text -------------------------------------------------------------

// What a build sets before its first level, done by the kernel that re-deals the
// text (each as a copy or a fill of its own was eight stream operations of a few
// microseconds before the first kernel that does work).
struct BuildInit {
  Counters *ctr;         // zeroed; min_depth = none yet
  uint32_t *row0;        // row 0 of the suffix array = n (histsortcomp.c:165), or null
  uint32_t *sample0;     // the sample of row 0, or null
  uint32_t *ghead0;      // tie-group head of row 0
  Seg *seg0;             // level 0's one segment, the whole text; null when the text is a single leaf
  uint32_t *tile_first;  // ... and where its tiles begin
  uint32_t *zero;        // counters of the first two levels
  uint32_t zero_words;
  uint32_t *text2;       // the text in little-endian 16-base words, one word of zeros in front (what the
  uint64_t t2words;      // position records are made from), or null
};

// Big-endian 16-base words of the text: base i at bits 31-2*(i&15) of word
// i>>4, zeros from base n on, kTextPadWords words of zeros after the end.
__global__ void k_prepare_text(const uint8_t *__restrict__ text, uint32_t n, uint64_t nwords,
                               uint32_t *__restrict__ tb, const BuildInit init) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t i = w; i < init.zero_words; i += (uint64_t)gridDim.x * blockDim.x) init.zero[i] = 0;
  if (w == 0) {
    Counters c = {};
    c.min_depth = 0xFFFFFFFFu;
    if (init.seg0) {
      // what k_begin_level_small would make of a list of this one segment
      *init.seg0 = Seg{0, n};
      init.tile_first[0] = 0;
      c.ncur = 1;
      c.total_tiles = (n + kPartTile - 1) / kPartTile;
    }
    *init.ctr = c;
    if (init.row0) *init.row0 = n;
    if (init.sample0) *init.sample0 = n;
    *init.ghead0 = 0;
    if (init.text2) init.text2[0] = 0;
  }
  if (w >= nwords) return;
  const uint64_t nbytes = (uint64_t)n / 4 + 1;
  uint32_t v = 0;
  if (4 * w + 4 <= nbytes && !((uintptr_t)text & 3u)) {
    v = __builtin_bswap32(reinterpret_cast<const uint32_t *>(text)[w]);
  } else {
    for (int k = 0; k < 4; ++k) {
      const uint64_t b = 4 * w + k;
      v = (v << 8) | (b < nbytes ? text[b] : 0u);
    }
  }
  const uint64_t first = 16 * w;
  const uint32_t valid = first >= n ? 0u : (n - first >= 16 ? 16u : (uint32_t)(n - first));
  const uint32_t mask = valid == 16 ? ~0u : (valid == 0 ? 0u : ~((1u << (32 - 2 * valid)) - 1u));
  v &= mask;
  tb[w] = v;
  if (init.text2 && w + 1 < init.t2words) {
    // the same bases, the first one in the lowest two bits
    const uint32_t x = __builtin_bitreverse32(v);
    init.text2[w + 1] = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  }
}

// 16 bases starting at position p, first base in the top bits
__device__ __forceinline__ uint32_t key32(const uint32_t *__restrict__ tb, uint64_t p) {
  const uint64_t w = p >> 4;
  const uint32_t sh = 2u * (uint32_t)(p & 15u);
  const uint64_t both = ((uint64_t)tb[w] << 32) | tb[w + 1];
  return (uint32_t)((both << sh) >> 32);
}

// base just before position p (p >= 1)
__device__ __forceinline__ uint32_t base_before(const uint32_t *__restrict__ tb, uint64_t p) {
  const uint64_t q = p - 1;
  return (tb[q >> 4] >> (30u - 2u * (uint32_t)(q & 15u))) & 3u;
}

// What travels with a suffix id from the scatter to the leaf sort: 15 bases of
// the suffix (sort key and the digits of further levels) and, in the low two
// bits, the base that precedes the suffix (its BWT symbol), both read from the
// same line of text while the scatter has it in hand.
//
// Where the 15 bases start: level 0 reads the text in order and so can look
// ahead for free; a suffix going on to level 1 gets bases 4..18 (its level-0
// digit is implied by its bucket), one that ends in a level-0 leaf bases 0..14.
// Level 1 finds its digit in the top byte and passes the word on unchanged;
// from level 2 on the digit is the second byte and the word moves up one byte
// per level.  A word therefore serves two levels after level 0 before the text
// has to be gathered from again, and leaves sort by 15 or 11 bases of text.
__device__ __forceinline__ uint32_t carried_word(const uint32_t *__restrict__ tb, uint32_t id,
                                                 uint32_t key) {
  return (key & ~3u) | (id ? base_before(tb, id) : 0u);
}

// ---- one bucket level ----------------------------------------------------

// which segment a tile belongs to: last s with tile_first[s] <= tile
__device__ __forceinline__ uint32_t find_segment(const uint32_t *tile_first, uint32_t nsegs,
                                                 uint32_t tile) {
  uint32_t lo = 0, hi = nsegs;  // invariant: tile_first[lo] <= tile < tile_first[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (tile_first[mid] <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

struct TileInfo {
  uint32_t seg, row0, count;
};

// one thread's work: which segment a tile lies in and which rows it covers (count 0: no such tile)
__device__ __forceinline__ void tile_info(uint32_t tile, const Seg *segs, const uint32_t *tile_first,
                                          const Counters *ctr, TileInfo *out) {
  if (tile >= ctr->total_tiles) {
    out->count = 0;
    return;
  }
  const uint32_t s = find_segment(tile_first, ctr->ncur, tile);
  const Seg sg = segs[s];
  const uint32_t off = (tile - tile_first[s]) * (uint32_t)kPartTile;
  out->seg = s;
  out->row0 = sg.start + off;
  const uint32_t left = sg.size - off;
  out->count = left < (uint32_t)kPartTile ? left : (uint32_t)kPartTile;
}

__device__ __forceinline__ bool locate_tile_at(uint32_t tile, const Seg *segs,
                                               const uint32_t *tile_first, const Counters *ctr,
                                               TileInfo *sh) {
  if (threadIdx.x == 0) tile_info(tile, segs, tile_first, ctr, sh);
  __syncthreads();
  return sh->count != 0;
}

// Scatter workgroups: the hardware deals workgroups to the eight XCDs
// round-robin, each with its own L2.  Giving XCD x the x-th eighth of the tiles
// keeps the runs that land next to each other in a bucket (consecutive tiles of
// one segment) in one L2, where partial lines merge before they are written back.
__device__ __forceinline__ bool locate_tile(const Seg *segs, const uint32_t *tile_first,
                                            const Counters *ctr, TileInfo *sh) {
  const uint32_t per_xcd = (ctr->total_tiles + 7u) / 8u;
  const uint32_t seq = blockIdx.x >> 3;
  const uint32_t tile = seq < per_xcd ? (blockIdx.x & 7u) * per_xcd + seq : 0xFFFFFFFFu;
  return locate_tile_at(tile, segs, tile_first, ctr, sh);
}

constexpr uint32_t kHistTilesPerGroup = 4;

// LDS-privatised histogram of the next four bases of every suffix in a run of
// tiles; counts go to the segment's counters when the run leaves the segment.
// Level 0 has a kernel of its own (k_part_hist0).
// MODE 1: ids from src, bases from the text;
// MODE 2: the bases are bits 23..16 of the word the previous level's scatter
//         stored next to the id (its key started four bases earlier): no gather;
// MODE 3: the bases are the top byte of that word (level 1: level 0 stored
//         words that start at this level's depth).
template <int MODE>
__global__ void __launch_bounds__(kPartThreads)
k_part_hist(const uint32_t *__restrict__ tb, const uint32_t *__restrict__ src,
            const uint32_t *__restrict__ srckey, const Seg *__restrict__ segs,
            const uint32_t *__restrict__ tile_first, uint32_t depth,
            const Counters *__restrict__ ctr, uint32_t *__restrict__ seg_counts) {
  __shared__ uint32_t hist[256];
  __shared__ TileInfo tis[kHistTilesPerGroup];
  hist[threadIdx.x] = 0;
  // the group's tiles are looked up side by side: a look-up is a chain of a dozen dependent loads
  if (threadIdx.x < kHistTilesPerGroup)
    tile_info(blockIdx.x * kHistTilesPerGroup + threadIdx.x, segs, tile_first, ctr, &tis[threadIdx.x]);
  __syncthreads();
  uint32_t cur_seg = 0xFFFFFFFFu;
  for (uint32_t k = 0; k < kHistTilesPerGroup; ++k) {
    const TileInfo ti = tis[k];
    const bool live = ti.count != 0;
    const uint32_t seg = live ? ti.seg : 0xFFFFFFFFu;
    if (seg != cur_seg) {
      if (cur_seg != 0xFFFFFFFFu) {
        const uint32_t c = hist[threadIdx.x];
        if (c) atomicAdd(&seg_counts[(uint64_t)cur_seg * 256 + threadIdx.x], c);
        hist[threadIdx.x] = 0;
      }
      cur_seg = seg;
      __syncthreads();
    }
    if (!live) break;
    const uint32_t row0 = ti.row0, count = ti.count;
    if (MODE == 2 || MODE == 3) {
      // the digit comes with the row: the tile's words are read four to a load where they are aligned
      auto add = [&](uint32_t w) { atomicAdd(&hist[MODE == 2 ? (w >> 16) & 255u : w >> 24], 1u); };
      const uint32_t *p = srckey + row0;
      uint32_t head = (4u - (row0 & 3u)) & 3u;
      head = head < count ? head : count;
      if (threadIdx.x < head) add(p[threadIdx.x]);
      const uint4 *pv = reinterpret_cast<const uint4 *>(p + head);
      const uint32_t nv = (count - head) / 4u;
      constexpr int kVecs = kPartTile / 4 / kPartThreads;  // per thread in a full tile
      uint4 q[kVecs];
#pragma unroll
      for (int u = 0; u < kVecs; ++u) {
        const uint32_t i = (uint32_t)u * kPartThreads + threadIdx.x;
        q[u] = i < nv ? pv[i] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < kVecs; ++u) {
        const uint32_t i = (uint32_t)u * kPartThreads + threadIdx.x;
        if (i < nv) {
          add(q[u].x);
          add(q[u].y);
          add(q[u].z);
          add(q[u].w);
        }
      }
      const uint32_t done = head + 4u * nv;
      if (done + threadIdx.x < count) add(p[done + threadIdx.x]);
    } else {
#pragma unroll 4
      for (int i = 0; i < kHistItems; ++i) {
        const uint32_t j = (uint32_t)i * kPartThreads + threadIdx.x;
        if (j < count) atomicAdd(&hist[key32(tb, (uint64_t)src[row0 + j] + depth) >> 24], 1u);
      }
    }
    __syncthreads();
  }
  if (cur_seg != 0xFFFFFFFFu) {
    const uint32_t c = hist[threadIdx.x];
    if (c) atomicAdd(&seg_counts[(uint64_t)cur_seg * 256 + threadIdx.x], c);
  }
}

// One workgroup per segment: counts -> absolute cursors of the 256 children;
// children above kLeafMax become next-level segments, runs of smaller
// neighbours are merged into leaves (a merged leaf is sorted from the parent's
// depth, so the bases that separate the merged children are part of its key).
__global__ void __launch_bounds__(256)
k_part_classify(const Seg *__restrict__ segs, uint32_t *__restrict__ seg_counts, uint32_t depth,
                uint32_t dst_id /* leaf_src_word */, Seg *__restrict__ next_segs, uint32_t next_cap,
                Leaf *__restrict__ leaves, uint32_t leaf_cap, uint32_t *__restrict__ leafmask,
                Counters *__restrict__ ctr, int by_eighth, uint32_t *__restrict__ bigleaves,
                uint32_t *__restrict__ midleaves) {
  // by_eighth (level 0, one segment): seg_counts holds eight rows of counts, one
  // per eighth of the tiles; each becomes that eighth's own run of cursors
  // inside every child, so that what one XCD scatters lies together.
  __shared__ uint32_t sizes[256], starts[256], wsum[4];
  const uint32_t s = blockIdx.x, t = threadIdx.x;
  if (s >= ctr->ncur) return;  // the grid covers the most segments this level can have
  const Seg sg = segs[s];
  uint32_t c = 0;
  if (by_eighth) {
    for (int e = 0; e < 8; ++e) c += seg_counts[e * 256 + t];
  } else {
    c = seg_counts[(uint64_t)s * 256 + t];
  }
  uint32_t v = c;
  const uint32_t lane = t & 63u, wave = t >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  uint32_t pre = 0;
  for (uint32_t w = 0; w < wave; ++w) pre += wsum[w];
  const uint32_t start = sg.start + pre + v - c;
  if (by_eighth) {
    uint32_t run = start;
    for (int e = 0; e < 8; ++e) {
      const uint32_t part = seg_counts[e * 256 + t];
      seg_counts[e * 256 + t] = run;
      run += part;
    }
  } else {
    seg_counts[(uint64_t)s * 256 + t] = start;  // becomes the scatter cursor
  }
  sizes[t] = c;
  starts[t] = start;
  {
    // which children end in a leaf: the scatter fetches fresh sort keys for those only
    const uint64_t small = __ballot(c <= kLeafMax);
    if (lane == 0) {
      leafmask[(uint64_t)s * 8 + wave * 2] = (uint32_t)small;
      leafmask[(uint64_t)s * 8 + wave * 2 + 1] = (uint32_t)(small >> 32);
    }
  }
  __syncthreads();
  // Greedy merge of neighbouring children into leaves.  Children come in four kinds: large ones
  // (more than kLeafMax rows) go on to the next level; those of up to kWaveLeafMax rows are merged
  // into leaves of up to that many rows, which one wave sorts by itself; one of up to kWaveLeafMid
  // rows is a leaf of its own, also one wave's; the rest are merged into leaves of up to kLeafMax
  // rows for the workgroup-wide sort.  A leaf holds children of one kind.
  // A run of children between two large ones is dealt by the thread of its first child; how far
  // a leaf that starts at child d reaches is found by every thread for its own child at once (a
  // binary search over prefix sums), so the walk is one hop per leaf, not one step per child.
  __shared__ uint32_t lstart[256], lsize[256], sidx[256], reach[256], wtot[4];
  __shared__ uint64_t upto_[257], wtot64[4];
  __shared__ uint32_t nl, ns, lbase, sbase;
  constexpr uint32_t kRowsBits = 24, kRunBits = 16;  // rows (at most 2^20), runs of one kind, large children
  constexpr uint64_t kRowsMask = (1ull << kRowsBits) - 1ull;
  const bool big = c > kLeafMax;
  auto kind_of = [](uint32_t rows) { return rows > kLeafMax ? 3u : rows > kWaveLeafMid ? 2u : rows > kWaveLeafMax ? 1u : 0u; };
  const uint32_t kind = kind_of(c);
  {
    // inclusive scan of (large children so far, runs begun so far, rows in other children so far), packed
    const uint32_t kind_before = t ? kind_of(sizes[t - 1]) : 3u;
    uint64_t pk = (big ? 1ull << (kRowsBits + kRunBits) : (uint64_t)c) |
                  ((uint64_t)((big || kind != kind_before) ? 1u : 0u) << kRowsBits);
    for (int off = 1; off < 64; off <<= 1) {
      const uint64_t u = __shfl_up(pk, off, 64);
      if (lane >= off) pk += u;
    }
    if (lane == 63) wtot64[wave] = pk;
    __syncthreads();
    for (uint32_t w = 0; w < wave; ++w) pk += wtot64[w];
    upto_[t + 1] = pk;
    if (t == 0) {
      upto_[0] = 0;
      nl = 0;
    }
  }
  __syncthreads();
  if (big) {
    sidx[(uint32_t)(upto_[t] >> (kRowsBits + kRunBits))] = t;
  } else {
    // the furthest child e such that children t..e belong to t's run and hold no more rows than a leaf of its kind may
    const uint64_t run = upto_[t + 1] >> kRowsBits;  // runs (and large children) up to and including t's own
    const uint32_t from = (uint32_t)(upto_[t] & kRowsMask);
    const uint32_t most = kind == 0 ? kWaveLeafMax : kind == 1 ? kWaveLeafMid : kLeafMax;
    uint32_t lo = t, hi = 255;
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      const uint64_t upto = upto_[mid + 1];
      if ((upto >> kRowsBits) == run && (uint32_t)(upto & kRowsMask) - from <= most) lo = mid;
      else hi = mid - 1;
    }
    reach[t] = lo;
  }
  // (no neighbours to merge at all -- every two small children together exceed a wave's leaf, the
  // usual case one level above the leaves -- is settled without the walk: each child is a leaf)
  const bool alone = big || reach[t] == t;
  if (__syncthreads_and(alone)) {
    const bool is_leaf = !big && c != 0;
    const uint64_t mine = __ballot(is_leaf);
    if (lane == 0) wtot[wave] = (uint32_t)__popcll(mine);
    __syncthreads();
    uint32_t k = (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wave; ++w) k += wtot[w];
    if (is_leaf) {
      lstart[k] = start;
      lsize[k] = c;
    }
    if (t == 255) nl = k + (is_leaf ? 1u : 0u);
  } else if (!big && (t == 0 || sizes[t - 1] > kLeafMax)) {
    uint32_t d = t;
    while (d < 256 && sizes[d] <= kLeafMax) {
      const uint32_t e = reach[d];
      const uint32_t rows_in = (uint32_t)(upto_[e + 1] & kRowsMask) - (uint32_t)(upto_[d] & kRowsMask);
      if (rows_in) {
        const uint32_t k = atomicAdd(&nl, 1u);
        lstart[k] = starts[d];
        lsize[k] = rows_in;
      }
      d = e + 1;
    }
  }
  __syncthreads();
  if (t == 0) {
    const uint32_t l = nl, g = (uint32_t)(upto_[256] >> (kRowsBits + kRunBits));
    ns = g;
    lbase = l ? atomicAdd(&ctr->nleaves, l) : 0u;
    sbase = g ? atomicAdd(&ctr->nnext, g) : 0u;
    if (lbase + l > leaf_cap || sbase + g > next_cap) ctr->overflow = 1;
  }
  __syncthreads();
  {
    const bool mine = t < nl && lbase + t < leaf_cap;
    const uint32_t rows = mine ? lsize[t] : 0u;
    if (mine) leaves[lbase + t] = Leaf{lstart[t], rows, depth, dst_id};
    // more rows than the plain one-wave sort takes: on the list of the roomier one, or of the
    // workgroup-wide sort; one claim of room per wave and list (every leaf of a level may be listed)
    auto list = [&](bool want, uint32_t *counter, uint32_t *out) {
      const uint64_t such = __ballot(want);
      if (!such) return;
      const int first = __ffsll((long long)such) - 1;
      uint32_t base = 0;
      if ((int)lane == first) base = atomicAdd(counter, (uint32_t)__popcll(such));
      base = __shfl(base, first, 64);
      if (want) out[base + (uint32_t)__popcll(such & ((1ull << lane) - 1ull))] = lbase + t;
    };
    list(rows > kWaveLeafMid, &ctr->nbig, bigleaves);
    list(rows > kWaveLeafMax && rows <= kWaveLeafMid, &ctr->nmid, midleaves);
  }
  if (t < ns && sbase + t < next_cap) next_segs[sbase + t] = Seg{starts[sidx[t]], sizes[sidx[t]]};
}

// The segments the level before emitted become this level's.  `bound` is what
// the launches of this level are sized for: the exact count when the host has
// read it, else the most a level of that depth can have -- the host then goes on
// without waiting for the count.
__global__ void k_begin_level(Counters *__restrict__ ctr, uint32_t bound) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uint32_t have = ctr->nnext;
    ctr->ncur = have < bound ? have : bound;  // more only after an overflow, which is reported
    ctr->nnext = 0;
  }
}

// k_begin_level, the tiles of every segment and their exclusive scan in one
// workgroup, for levels of at most 1024 segments (the first two of any text):
// five launches less than the general path below.
__global__ void __launch_bounds__(1024)
k_begin_level_small(Counters *__restrict__ ctr, uint32_t bound, const Seg *__restrict__ segs,
                    uint32_t *__restrict__ tile_first) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t ncur;
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  if (t == 0) {
    const uint32_t have = ctr->nnext;
    ncur = have < bound ? have : bound;
  }
  __syncthreads();
  const uint32_t c = t < ncur ? (segs[t].size + kPartTile - 1) / kPartTile : 0u;
  uint32_t v = c;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  uint32_t pre = 0, total = 0;
  for (uint32_t w = 0; w < 16; ++w) {
    if (w < wave) pre += wsum[w];
    total += wsum[w];
  }
  if (t < bound) tile_first[t] = pre + v - c;
  if (t == 0) {
    ctr->ncur = ncur;
    ctr->nnext = 0;
    ctr->total_tiles = total;
  }
}

__global__ void k_seg_tile_counts(const Seg *__restrict__ segs, uint32_t bound, const Counters *__restrict__ ctr,
                                  uint32_t *__restrict__ tiles) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < bound) tiles[s] = s < ctr->ncur ? (segs[s].size + kPartTile - 1) / kPartTile : 0u;
}

// Sharded build, level 0: how many suffixes precede the digit range [dlo, dhi)
// and how many fall in it.
constexpr uint32_t kRangeSlots = 32, kRangeSlotStride = 16;  // partial sums, 128 bytes apart
constexpr size_t kRangeBytes = (size_t)kRangeSlots * kRangeSlotStride * 8;

__global__ void __launch_bounds__(256)
k_restrict_level0(const uint32_t *__restrict__ counts /* [8][256], one row per eighth of the tiles */,
                  uint32_t dlo, uint32_t dhi, const unsigned long long *__restrict__ before_slots,
                  uint64_t *__restrict__ out /* [0] before, [1] inside */) {
  // the histogram counted the digits of the range only, and how many positions lie before it
  __shared__ unsigned long long inside, before;
  const uint32_t t = threadIdx.x;
  if (t == 0) inside = before = 0;
  __syncthreads();
  uint32_t c = 0;
  for (int e = 0; e < 8; ++e) c += counts[e * 256 + t];
  if (t >= dlo && t < dhi && c) atomicAdd(&inside, (unsigned long long)c);
  if (t < kRangeSlots && before_slots[t * kRangeSlotStride]) atomicAdd(&before, before_slots[t * kRangeSlotStride]);
  __syncthreads();
  if (t == 0) {
    out[0] = before;
    out[1] = inside;
  }
}

// Tile-local bucket order in LDS, then each bucket's run goes to the
// segment's cursor for that bucket (one atomic per tile and bucket).  Order
// inside a bucket is irrelevant at this stage.
// The bucket of a suffix is read from the word the previous level stored with
// it, and the word is passed on; fresh keys are gathered from the text only when
// the stored words have run out of bases (see carried_word).  Level 0 has a
// kernel of its own (k_part_scatter0).
__global__ void __launch_bounds__(kScat1Threads)
k_part_scatter(const uint32_t *__restrict__ tb, const uint32_t *__restrict__ src,
               const uint32_t *__restrict__ srckey, uint32_t *__restrict__ dst,
               uint32_t *__restrict__ dstkey, const Seg *__restrict__ segs,
               const uint32_t *__restrict__ tile_first, uint32_t depth,
               const Counters *__restrict__ ctr, uint32_t *__restrict__ cursors,
               int gather_all, int top_digit) {
  // gather_all: the stored words have run out of bases, read fresh ones for every suffix.
  // top_digit: the source word starts at this level's depth (level 1).
  __shared__ uint32_t hist[256], lstart[256], gbase[256], fill[256], wsum[4];
  __shared__ uint32_t ids[kPartTile];
  __shared__ uint32_t kv[kPartTile];
  __shared__ TileInfo ti;
  const uint32_t t = threadIdx.x;
  if (t < 256) {
    hist[t] = 0;
    fill[t] = 0;
  }
  if (!locate_tile(segs, tile_first, ctr, &ti)) return;
  const uint32_t row0 = ti.row0, count = ti.count;
  // The order of a tile's rows is free, so a thread takes four neighbouring rows per load where the
  // rows are 16-byte aligned (4-byte loads, one row per lane, kept the kernel's loads in flight a
  // quarter as deep); the up to three rows in front of the first aligned one and behind the last whole
  // group of four are a ninth item of the first threads.
  constexpr int kVecs = kScat1Items / 4;
  static_assert(kScat1Items % 4 == 0, "whole groups of four rows per thread");
  uint32_t id[kScat1Items + 1], key[kScat1Items + 1];
  uint32_t have = 0;  // bit i: item i holds a row
  uint32_t head = (4u - (row0 & 3u)) & 3u;
  head = head < count ? head : count;
  const uint32_t nv = (count - head) / 4u, done = head + 4u * nv;
  {
    const uint4 *sv = reinterpret_cast<const uint4 *>(src + row0 + head);
    const uint4 *kvv = reinterpret_cast<const uint4 *>(srckey + row0 + head);
    uint4 a[kVecs], b[kVecs];
#pragma unroll
    for (int u = 0; u < kVecs; ++u) {
      const uint32_t i = (uint32_t)u * kScat1Threads + t;
      if (i < nv) {
        a[u] = sv[i];
        b[u] = kvv[i];
        have |= 15u << (4 * u);
      }
    }
    // odd rows: t < head takes row t, the next threads the rows from `done` on
    const uint32_t odd = t < head ? t : done + (t - head);
    if (odd < count && (t < head || odd >= done)) {
      id[kScat1Items] = src[row0 + odd];
      key[kScat1Items] = srckey[row0 + odd];
      have |= 1u << kScat1Items;
    }
#pragma unroll
    for (int u = 0; u < kVecs; ++u) {
      id[4 * u] = a[u].x, id[4 * u + 1] = a[u].y, id[4 * u + 2] = a[u].z, id[4 * u + 3] = a[u].w;
      key[4 * u] = b[u].x, key[4 * u + 1] = b[u].y, key[4 * u + 2] = b[u].z, key[4 * u + 3] = b[u].w;
    }
  }
#pragma unroll
  for (int i = 0; i <= kScat1Items; ++i) {
    if ((have >> i) & 1u) {
      const uint32_t old = key[i];
      const uint32_t d = top_digit ? old >> 24 : (old >> 16) & 255u;
      atomicAdd(&hist[d], 1u);
      if (gather_all)
        key[i] = (key32(tb, (uint64_t)id[i] + depth) & ~3u) | (old & 3u);
      else
        key[i] = top_digit ? old : ((old & ~3u) << 8) | (old & 3u);
    }
  }
  __syncthreads();
  // exclusive scan of the 256 bucket counts by the first four waves; claim global room per bucket
  const uint32_t lane = t & 63u, wave = t >> 6;
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
  // the claim of global room is in flight while the tile is put in bucket order
  uint32_t claimed = 0;
  if (t < 256) {
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wave; ++w) pre += wsum[w];
    lstart[t] = pre + v - c;
    if (c) claimed = atomicAdd(&cursors[(uint64_t)ti.seg * 256 + t], c);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i <= kScat1Items; ++i) {
    if ((have >> i) & 1u) {
      const uint32_t d = key[i] >> 24;
      const uint32_t p = lstart[d] + atomicAdd(&fill[d], 1u);
      ids[p] = id[i];
      kv[p] = key[i];
    }
  }
  if (t < 256) gbase[t] = claimed;
  __syncthreads();
  const uint32_t placed = lstart[255] + hist[255];
  for (uint32_t j = t; j < placed; j += kScat1Threads) {
    const uint32_t k = kv[j];
    const uint32_t pos = gbase[k >> 24] + (j - lstart[k >> 24]);
    dst[pos] = ids[j];
    dstkey[pos] = k;
  }
}

// ---- level 0 ---------------------------------------------------------------
//
// Level 0 sorts text positions themselves: a tile is 8192 consecutive positions,
// so a thread can take 16 (or 32) consecutive ones and read their bases out of a
// register window of three (or four) text words instead of loading a window per
// position.  In a sharded build most positions are looked at only to be dropped
// (their first four bases belong to another rank), which is then nearly free.

// 16 bases starting `o` bases (0..31) into the 48 bases of the words a, b, c
__device__ __forceinline__ uint32_t window16(uint32_t a, uint32_t b, uint32_t c, uint32_t o) {
  const uint64_t ab = ((uint64_t)a << 32) | b, bc = ((uint64_t)b << 32) | c;
  return o < 16 ? (uint32_t)((ab << (2 * o)) >> 32) : (uint32_t)((bc << (2 * (o - 16))) >> 32);
}

// Counts of the leading four bases of every position, per eighth of the tiles (see
// k_part_classify's by_eighth), restricted to digits in [dlo, dhi); the positions
// whose digit lies below dlo are only counted, into the slots of range_counts.
__global__ void __launch_bounds__(kPartThreads)
k_part_hist0(const uint32_t *__restrict__ tb, uint32_t n, const Counters *__restrict__ ctr,
             uint32_t *__restrict__ seg_counts, uint32_t dlo, uint32_t dhi,
             unsigned long long *__restrict__ range_counts) {
  __shared__ uint32_t hist[256];
  const uint32_t t = threadIdx.x;
  hist[t] = 0;
  const uint32_t total_tiles = ctr->total_tiles, per_xcd = (total_tiles + 7u) / 8u;
  uint32_t cur = 0xFFFFFFFFu, before = 0;
  __syncthreads();
  for (uint32_t k = 0; k < kHistTilesPerGroup; ++k) {
    const uint32_t tile = blockIdx.x * kHistTilesPerGroup + k;
    if (tile >= total_tiles) break;  // uniform
    const uint32_t eighth = tile / per_xcd;
    if (eighth != cur) {
      if (cur != 0xFFFFFFFFu) {
        const uint32_t c = hist[t];
        if (c) atomicAdd(&seg_counts[(uint64_t)cur * 256 + t], c);
        hist[t] = 0;
      }
      cur = eighth;
      __syncthreads();
    }
    const uint32_t row0 = tile * (uint32_t)kPartTile;
    const uint32_t count = n - row0 < (uint32_t)kPartTile ? n - row0 : (uint32_t)kPartTile;
    // kHistItems = 32 consecutive positions per thread: two words and two of lookahead
    const uint64_t w = ((uint64_t)row0 >> 4) + 2ull * t;
    const uint32_t p0 = 32u * t;
    uint32_t w0 = 0, w1 = 0, w2 = 0;
    if (p0 < count) {  // the text is padded by a few words only
      w0 = tb[w];
      w1 = tb[w + 1];
      w2 = tb[w + 2];
    }
#pragma unroll
    for (uint32_t q = 0; q < 32; ++q) {
      if (p0 + q < count) {
        const uint32_t d = window16(w0, w1, w2, q) >> 24;
        if (d >= dlo && d < dhi) atomicAdd(&hist[d], 1u);
        else before += d < dlo ? 1u : 0u;
      }
    }
    __syncthreads();
  }
  if (cur != 0xFFFFFFFFu) {
    const uint32_t c = hist[t];
    if (c) atomicAdd(&seg_counts[(uint64_t)cur * 256 + t], c);
  }
  if (range_counts) {
    // one atomic per workgroup, spread over kRangeSlots cache lines: every workgroup
    // of the launch adding to one address would take longer than the counting
    __syncthreads();
    if (t == 0) hist[0] = 0;
    __syncthreads();
    for (int off = 32; off; off >>= 1) before += __shfl_down(before, off, 64);
    if ((t & 63u) == 0 && before) atomicAdd(&hist[0], before);
    __syncthreads();
    if (t == 0 && hist[0])
      atomicAdd(&range_counts[(blockIdx.x % kRangeSlots) * kRangeSlotStride], (unsigned long long)hist[0]);
  }
}

// Level-0 scatter: as k_part_scatter, for a tile of 8192 consecutive positions.
// A suffix that goes on to level 1 carries bases 4..18 (its digit is implied by
// its bucket), one that ends in a level-0 leaf bases 0..14; both with the base
// before it.  Positions whose digit lies outside [dlo, dhi) are dropped.
__global__ void __launch_bounds__(kScat1Threads)
k_part_scatter0(const uint32_t *__restrict__ tb, uint32_t n, uint32_t *__restrict__ dst,
                uint32_t *__restrict__ dstkey, const Counters *__restrict__ ctr,
                uint32_t *__restrict__ cursors /* [8][256]: one row per eighth of the tiles */,
                const uint32_t *__restrict__ leafmask, uint32_t dlo, uint32_t dhi) {
  __shared__ uint32_t hist[256], lstart[256], gbase[256], fill[256], wsum[4], lmask[8];
  __shared__ uint32_t ids[kPartTile];
  __shared__ uint32_t kv[kPartTile];
  const uint32_t t = threadIdx.x;
  if (t < 256) {
    hist[t] = 0;
    fill[t] = 0;
  }
  if (t < 8) lmask[t] = leafmask[t];
  // the x-th eighth of the tiles goes to XCD x (see locate_tile)
  const uint32_t total_tiles = ctr->total_tiles, per_xcd = (total_tiles + 7u) / 8u;
  const uint32_t seq = blockIdx.x >> 3;
  const uint32_t tile = (blockIdx.x & 7u) * per_xcd + seq;
  if (seq >= per_xcd || tile >= total_tiles) return;  // uniform
  const uint32_t row0 = tile * (uint32_t)kPartTile;
  const uint32_t count = n - row0 < (uint32_t)kPartTile ? n - row0 : (uint32_t)kPartTile;
  __syncthreads();
  // eight consecutive positions per thread, one half of a text word: the word, the base before
  // it, two words of lookahead
  const uint32_t h8 = 8u * (t & 1u);
  const uint64_t w = ((uint64_t)row0 >> 4) + (t >> 1);
  const uint32_t p0 = 8u * t;
  uint32_t w0 = 0, w1 = 0, w2 = 0, prev = 0;
  if (p0 < count) {  // the text is padded by a few words only
    w0 = tb[w];
    w1 = tb[w + 1];
    w2 = tb[w + 2];
    prev = w ? tb[w - 1] & 3u : 0u;  // base 15 of the word before
  }
  uint32_t kept = 0;  // bit q: position p0 + q is this shard's
#pragma unroll
  for (uint32_t q = 0; q < 8; ++q) {
    if (p0 + q < count) {
      const uint32_t d = window16(w0, w1, w2, h8 + q) >> 24;
      if (d >= dlo && d < dhi) {
        kept |= 1u << q;
        atomicAdd(&hist[d], 1u);
      }
    }
  }
  __syncthreads();
  const uint32_t lane = t & 63u, wave = t >> 6;
  uint32_t c = 0, v = 0, claimed = 0;
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
    for (uint32_t q = 0; q < wave; ++q) pre += wsum[q];
    lstart[t] = pre + v - c;
    if (c) claimed = atomicAdd(&cursors[(uint64_t)(blockIdx.x & 7u) * 256 + t], c);
  }
  __syncthreads();
#pragma unroll
  for (uint32_t q = 0; q < 8; ++q) {
    if ((kept >> q) & 1u) {
      const uint32_t o = h8 + q;  // place in the word
      const uint32_t k32 = window16(w0, w1, w2, o);
      const uint32_t d = k32 >> 24;
      const bool leaf = (lmask[d >> 5] >> (d & 31u)) & 1u;
      const uint32_t key = leaf ? k32 : window16(w0, w1, w2, o + 4);
      const uint32_t bw = o ? (w0 >> (32u - 2u * o)) & 3u : prev;  // the base before position p0 + q
      const uint32_t p = lstart[d] + atomicAdd(&fill[d], 1u);
      ids[p] = (d << 24) | (p0 + q);
      kv[p] = (key & ~3u) | ((row0 + p0 + q) ? bw : 0u);
    }
  }
  if (t < 256) gbase[t] = claimed;  // the claim was in flight while the tile was put in order
  __syncthreads();
  const uint32_t placed = lstart[255] + hist[255];
  for (uint32_t j = t; j < placed; j += kScat1Threads) {
    const uint32_t id = ids[j];
    const uint32_t d = id >> 24;
    const uint32_t pos = gbase[d] + (j - lstart[d]);
    dst[pos] = row0 + (id & 0xFFFFFFu);
    dstkey[pos] = kv[j];
  }
}

// ---- leaves ---------------------------------------------------------------

constexpr int kLogLeafThreads = 9;
constexpr int kLeafThreads = 1 << kLogLeafThreads;  // per leaf: eight waves, eight rows per thread
constexpr int kLeafWaves = kLeafThreads / 64;
constexpr int kLeafPer = kLeafMax / kLeafThreads;    // rows a thread owns, at most

constexpr uint32_t kDeepRun = 8;     // tie runs up to this long are settled by direct comparison
constexpr uint32_t kDeepWords = 32;  // ... over at most this many further 16-base words

// LDS slot of element i: one slot of padding per 32 keeps 8-byte accesses at a
// stride of 16 elements (the blocked layout) free of bank conflicts
__device__ __forceinline__ uint32_t pad(uint32_t i) { return i + (i >> 5); }

// Element m of thread t when each thread holds the 2^LOGE elements whose index
// differs in bits [s, s+LOGE): s = 0 is the blocked layout.
template <int LOGE>
__device__ __forceinline__ uint32_t lay(uint32_t s, uint32_t t, uint32_t m) {
  return ((t >> s) << (s + LOGE)) | (m << s) | (t & ((1u << s) - 1u));
}

__device__ __forceinline__ void cex(uint64_t &a, uint64_t &b, bool asc) {
  const bool sw = (a > b) == asc;
  const uint64_t x = sw ? b : a, y = sw ? a : b;
  a = x;
  b = y;
}

// Bitonic sort of kLeafThreads << LOGE keys, 2^LOGE per thread, compare-exchanges done in
// registers: a stage whose stride bit lies outside the bits a thread holds
// costs one trip through LDS that re-deals the elements, and each trip then
// serves up to LOGE stages.  Ends in the blocked layout.
template <int LOGE>
__device__ __forceinline__ void leaf_bitonic(uint64_t (&r)[1 << LOGE], uint64_t *lds, uint32_t t) {
  constexpr int E = 1 << LOGE;
  constexpr int LOGP = LOGE + kLogLeafThreads;
#pragma unroll
  for (int lk = 1; lk <= LOGE; ++lk) {
#pragma unroll
    for (int lb = lk - 1; lb >= 0; --lb) {
#pragma unroll
      for (int m = 0; m < E; ++m)
        if (!(m & (1 << lb))) cex(r[m], r[m | (1 << lb)], (((t * E + m) >> lk) & 1u) == 0);
    }
  }
  uint32_t cur = 0;
  for (int lk = LOGE + 1; lk <= LOGP; ++lk) {
    int jb = lk - 1;
    while (jb >= 0) {
      const uint32_t s = jb >= LOGE - 1 ? (uint32_t)(jb - (LOGE - 1)) : 0u;
      if (s != cur) {
#pragma unroll
        for (int m = 0; m < E; ++m) lds[pad(lay<LOGE>(cur, t, m))] = r[m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < E; ++m) r[m] = lds[pad(lay<LOGE>(s, t, m))];
        __syncthreads();
        cur = s;
      }
#pragma unroll
      for (int lb = LOGE - 1; lb >= 0; --lb) {
        if ((int)s + lb > jb) continue;
#pragma unroll
        for (int m = 0; m < E; ++m)
          if (!(m & (1 << lb)))
            cex(r[m], r[m | (1 << lb)], ((lay<LOGE>(s, t, m) >> lk) & 1u) == 0);
      }
      jb = (int)s - 1;
    }
  }
}

// Sort word of a suffix: 30 key bits, then the complement of the start (equal
// keys: larger start first), then the two bits of its BWT symbol (payload).
__device__ __forceinline__ uint64_t leaf_word(uint32_t carried, uint32_t id) {
  return ((uint64_t)(carried >> 2) << 34) | ((uint64_t)(uint32_t)~id << 2) | (carried & 3u);
}
__device__ __forceinline__ uint32_t word_key(uint64_t w) { return (uint32_t)(w >> 34); }
__device__ __forceinline__ uint32_t word_id(uint64_t w) { return ~(uint32_t)(w >> 2); }

__device__ __forceinline__ uint64_t leaf_fetch(const uint32_t *__restrict__ tb,
                                               const uint32_t *__restrict__ src,
                                               const uint32_t *__restrict__ srckey, uint32_t row,
                                               uint32_t depth) {
  if (src) return leaf_word(srckey[row], src[row]);
  return leaf_word(carried_word(tb, row, key32(tb, (uint64_t)row + depth)), row);
}

template <int LOGE>
__device__ __forceinline__ void leaf_load_sort(const uint32_t *__restrict__ tb,
                                               const uint32_t *__restrict__ src,
                                               const uint32_t *__restrict__ srckey, uint32_t start,
                                               uint32_t size, uint32_t depth, uint64_t *lds,
                                               uint32_t t) {
  constexpr int E = 1 << LOGE;
  uint64_t r[E];
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const uint32_t j = t * E + m;
    r[m] = j < size ? leaf_fetch(tb, src, srckey, start + j, depth) : ~0ull;
  }
  leaf_bitonic<LOGE>(r, lds, t);
#pragma unroll
  for (int m = 0; m < E; ++m) lds[pad(t * E + m)] = r[m];
}

constexpr uint32_t kLeafBuckets = 2048;   // LDS buckets of the interpolation sort (16-bit counters, two per word)
constexpr uint32_t kLeafBucketMax = 32;   // a fuller bucket sends the leaf to the bitonic network

// Sorts a leaf whose keys are spread evenly (any text without long repeats):
// one counting pass over kLeafBuckets buckets chosen by linear interpolation
// between the smallest and largest key -- monotone in the key, so buckets are
// in order -- then every word finds its place inside its bucket (1.7 words on
// average) by counting.  Leaves the words in `lds` in the blocked layout and returns true;
// returns false (lds untouched in any way that matters) when some bucket holds
// more than kLeafBucketMax words.  All threads of the workgroup call it.
__device__ __forceinline__ bool leaf_bucket_sort(const uint32_t *__restrict__ tb,
                                                 const uint32_t *__restrict__ src,
                                                 const uint32_t *__restrict__ srckey, uint32_t start,
                                                 uint32_t size, uint32_t depth, uint64_t *lds,
                                                 uint32_t *cnt, uint32_t *red, uint32_t t) {
  constexpr int E = kLeafPer;
  static_assert(E % 4 == 0, "a thread takes its rows four neighbours at a time");
  const uint32_t lane = t & 63u, wave = t >> 6;
  uint64_t r[E];
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
  // Which row of the leaf a thread's item m is: any one-to-one assignment serves a sort, and this one
  // lets a thread read four neighbouring rows with one 16-byte load per array (the rows start at any
  // multiple of four bytes; 4-byte loads, one row per lane, cost a third of the kernel).
  auto row_of = [&](int m) { return 4u * ((uint32_t)(m / 4) * kLeafThreads + t) + (uint32_t)(m % 4); };
  typedef uint32_t u32x4_at_4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
  for (int u = 0; u < E / 4; ++u) {
    const uint32_t j0 = row_of(4 * u);
    if (src && j0 + 3u < size) {
      const u32x4_at_4 k4 = *reinterpret_cast<const u32x4_at_4 *>(srckey + start + j0);
      const u32x4_at_4 i4 = *reinterpret_cast<const u32x4_at_4 *>(src + start + j0);
      r[4 * u] = leaf_word(k4.x, i4.x);
      r[4 * u + 1] = leaf_word(k4.y, i4.y);
      r[4 * u + 2] = leaf_word(k4.z, i4.z);
      r[4 * u + 3] = leaf_word(k4.w, i4.w);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        r[4 * u + e] = j0 + e < size ? leaf_fetch(tb, src, srckey, start + j0 + e, depth) : ~0ull;
    }
  }
#pragma unroll
  for (int m = 0; m < E; ++m) {
    if (row_of(m) < size) {
      const uint32_t k = word_key(r[m]);
      kmin = k < kmin ? k : kmin;
      kmax = k > kmax ? k : kmax;
    }
  }
#pragma unroll
  for (int q = 0; q < (int)(kLeafBuckets / 2 / kLeafThreads); ++q) cnt[q * kLeafThreads + t] = 0;
  for (int off = 32; off; off >>= 1) {
    const uint32_t a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
  }
  if (lane == 0) {
    red[wave] = kmin;
    red[kLeafWaves + wave] = kmax;
  }
  __syncthreads();
  for (int w = 0; w < kLeafWaves; ++w) {
    kmin = red[w] < kmin ? red[w] : kmin;
    kmax = red[kLeafWaves + w] > kmax ? red[kLeafWaves + w] : kmax;
  }
  const float scale = (float)kLeafBuckets / (float)(kmax - kmin + 1u);
  uint32_t bucket[E], slot[E];
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const uint32_t j = row_of(m);
    if (j < size) {
      uint32_t b = (uint32_t)((float)(word_key(r[m]) - kmin) * scale);
      b = b < kLeafBuckets ? b : kLeafBuckets - 1;
      bucket[m] = b;
      const uint32_t sh = 16u * (b & 1u);
      slot[m] = (atomicAdd(&cnt[b >> 1], 1u << sh) >> sh) & 0xFFFFu;
    }
  }
  __syncthreads();
  // exclusive scan of the bucket counts; thread t owns buckets [B t, B t + B)
  constexpr int B = kLeafBuckets / kLeafThreads;
  uint32_t own[B], sum = 0, fullest = 0;
#pragma unroll
  for (int q = 0; q < B; ++q) {
    own[q] = (cnt[(t * B + q) >> 1] >> (16 * (q & 1))) & 0xFFFFu;
    sum += own[q];
    fullest = own[q] > fullest ? own[q] : fullest;
  }
  uint32_t v = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  for (int off = 32; off; off >>= 1) {
    const uint32_t u = __shfl_xor(fullest, off, 64);
    fullest = u > fullest ? u : fullest;
  }
  __syncthreads();  // red[] is read by every thread above before it is reused
  if (lane == 63) red[wave] = v;
  if (lane == 0) red[kLeafWaves + wave] = fullest;
  __syncthreads();
  uint32_t first = v - sum;
  for (uint32_t w = 0; w < wave; ++w) first += red[w];
  fullest = 0;
  for (int w = 0; w < kLeafWaves; ++w) fullest = red[kLeafWaves + w] > fullest ? red[kLeafWaves + w] : fullest;
  if (fullest > kLeafBucketMax) return false;
  {
    uint32_t run = first;
#pragma unroll
    for (int q = 0; q < B; q += 2) {
      cnt[(t * B + q) >> 1] = run | ((run + own[q]) << 16);  // start rows of the two buckets (< 65536)
      run += own[q] + own[q + 1];
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const uint32_t j = row_of(m);
    if (j < size) lds[pad(((cnt[bucket[m] >> 1] >> (16 * (bucket[m] & 1u))) & 0xFFFFu) + slot[m])] = r[m];
  }
  __syncthreads();
  // Order inside the buckets: every word counts the words of its bucket that are
  // smaller (words are distinct: they hold the suffix start) and moves to that
  // place.  The loads of a thread's eight words are independent of each other,
  // where an insertion sort per bucket is a chain of dependent LDS round trips.
  uint32_t s0[E], c[E], rank[E], most = 0;
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const uint32_t j = row_of(m);
    s0[m] = c[m] = rank[m] = 0;
    if (j < size) {
      const uint32_t b = bucket[m];
      s0[m] = (cnt[b >> 1] >> (16 * (b & 1u))) & 0xFFFFu;
      const uint32_t s1 = b + 1 < kLeafBuckets ? (cnt[(b + 1) >> 1] >> (16 * ((b + 1) & 1u))) & 0xFFFFu : size;
      c[m] = s1 - s0[m];
      most = c[m] > most ? c[m] : most;
    }
  }
  for (uint32_t i = 0; i < most; ++i) {
#pragma unroll
    for (int m = 0; m < E; ++m)
      if (i < c[m]) rank[m] += lds[pad(s0[m] + i)] < r[m] ? 1u : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const uint32_t j = row_of(m);
    if (j < size) lds[pad(s0[m] + rank[m])] = r[m];
  }
  return true;
}

// Order of two suffixes that agree on their first `off` bases: -1 if a comes
// first, +1 if b does, 0 if still undecided after kDeepWords words.  Past the
// end of the text a suffix reads as A's; two suffixes equal all the way are
// ordered shorter first.
__device__ __forceinline__ int deep_cmp(const uint32_t *__restrict__ tb, uint32_t n, uint32_t a,
                                        uint32_t b, uint32_t off) {
  for (uint32_t w = 0; w < kDeepWords; ++w) {
    const uint64_t pa = (uint64_t)a + off + 16ull * w, pb = (uint64_t)b + off + 16ull * w;
    const bool ea = pa >= n, eb = pb >= n;
    if (ea && eb) return a > b ? -1 : 1;
    const uint32_t ka = ea ? 0u : key32(tb, pa), kb = eb ? 0u : key32(tb, pb);
    if (ka != kb) return ka < kb ? -1 : 1;
  }
  return 0;
}

// One workgroup sorts one leaf in LDS by (next 16 bases ascending, start
// descending), settles short runs of equal keys by comparing the suffixes
// further, writes the rows of the suffix array, and records for every row the
// first row of its tie group (its own row when resolved).
__device__ __forceinline__ void
leaf_sort_block(const uint32_t leaf, const uint32_t *__restrict__ tb, const uint32_t *__restrict__ bufA,
                const uint32_t *__restrict__ bufB, const uint32_t *__restrict__ keyA,
                const uint32_t *__restrict__ keyB, const Leaf *__restrict__ leaves, uint32_t n,
                uint32_t *__restrict__ sa_rows /* = SA+1 */, uint32_t *__restrict__ ghead /* full rows */,
                uint8_t *__restrict__ bw /* BWT symbol of each row of SA+1 */,
                uint8_t *__restrict__ leaf_ties /* per leaf: 1 if it wrote tie-group heads */,
                uint32_t *__restrict__ samples /* SA[8 i] at samples[i], or null */,
                Counters *__restrict__ ctr) {
  __shared__ uint64_t keys[kLeafMax + kLeafMax / 32 + 1];
  __shared__ uint32_t settled[kLeafMax / 32];  // bit j: row j is ordered against row j-1
  __shared__ uint32_t wred[kLeafWaves], wmax[kLeafWaves];
  __shared__ uint32_t bcnt[kLeafBuckets / 2], bred[2 * kLeafWaves];
  const Leaf lf = leaves[leaf];
  const uint32_t t = threadIdx.x, size = lf.size, start = lf.start, depth = lf.depth;
  const uint32_t srcid = lf.src & 255u, keybases = lf.src >> 8;
  const uint32_t *src = srcid == kSrcA ? bufA : srcid == kSrcB ? bufB : nullptr;
  const uint32_t *srckey = srcid == kSrcA ? keyA : keyB;
  if (size == 1) {
    if (t == 0) {
      const uint64_t w = leaf_fetch(tb, src, srckey, start, depth);
      const uint32_t id = word_id(w);
      sa_rows[start] = id;
      if (samples && !((1 + start) & kDevSampleMask)) samples[(1 + start) >> kDevSampleShift] = id;
      leaf_ties[leaf] = 0;
      bw[start] = (uint8_t)(w & 3u);
      if (id == 0) ctr->endloc = 1 + start;
    }
    return;
  }
  if (t < kLeafMax / 32) settled[t] = 0;
  uint32_t P = 2;
  while (P < size) P <<= 1;
  if (size > kLeafThreads && leaf_bucket_sort(tb, src, srckey, start, size, depth, keys, bcnt, bred, t)) {
    // sorted by the counting pass
  } else if (kLeafPer >= 8 && P == 8 * kLeafThreads) {
    leaf_load_sort<3>(tb, src, srckey, start, size, depth, keys, t);
  } else if (kLeafPer >= 4 && P == 4 * kLeafThreads) {
    leaf_load_sort<2>(tb, src, srckey, start, size, depth, keys, t);
  } else if (P == 2 * kLeafThreads) {
    leaf_load_sort<1>(tb, src, srckey, start, size, depth, keys, t);
  } else {
    for (uint32_t j = t; j < P; j += kLeafThreads)
      keys[pad(j)] = j < size ? leaf_fetch(tb, src, srckey, start + j, depth) : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t q = t; q < (P >> 1); q += kLeafThreads) {
          const uint32_t i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
          const uint32_t x = i | j;
          const uint64_t a = keys[pad(i)], b = keys[pad(x)];
          if ((a > b) == ((i & k) == 0)) {
            keys[pad(i)] = b;
            keys[pad(x)] = a;
          }
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  // blocked from here on: thread t owns rows [t*per, t*per+per)
  const uint32_t per = (P + kLeafThreads - 1) / kLeafThreads;  // 1..kLeafPer
  const uint32_t j0 = t * per;
  // the thread's rows and one neighbour on each side, in registers: w[1 + k] is
  // row j0 + k
  uint64_t w[kLeafPer + 2];
  auto load_window = [&]() {
#pragma unroll
    for (uint32_t k = 0; k < kLeafPer + 2; ++k) {
      const uint32_t j = j0 + k - 1u;  // wraps below row 0, which then fails j < size
      w[k] = (k <= per + 1 && j < size) ? keys[pad(j)] : ~0ull;
    }
  };
  load_window();
  bool any_equal = false;
#pragma unroll
  for (uint32_t k = 0; k < kLeafPer + 1; ++k) {
    const uint32_t j = j0 + k - 1u;  // rows j and j + 1
    any_equal = any_equal || (k <= per && j < size && j + 1 < size && word_key(w[k]) == word_key(w[k + 1]));
  }
  // equal keys are rare (none in most leaves of a random text): only then are
  // runs settled by deeper comparison, in LDS, and the window read again
  bool ties = __syncthreads_or(any_equal);  // also: every thread has read its window
  if (ties) {
    // Mostly pairs (a text of 250 Mbp sorted by 19 bases: one or two in every leaf): the
    // thread of a pair's second row compares the two suffixes beyond the key and swaps
    // the words in LDS.  A run of three or an undecided pair leaves it to the code below.
    bool longer = false;
#pragma unroll
    for (uint32_t k = 0; k < kLeafPer; ++k) {
      const uint32_t j = j0 + k;
      longer = longer || (k < per && j > 0 && j + 1 < size && word_key(w[k]) == word_key(w[k + 1]) &&
                          word_key(w[k + 2]) == word_key(w[k + 1]));
    }
    if (!__syncthreads_or(longer)) {
      bool undecided = false;
#pragma unroll
      for (uint32_t k = 0; k < kLeafPer; ++k) {
        const uint32_t j = j0 + k;
        if (k < per && j > 0 && j < size && word_key(w[k]) == word_key(w[k + 1])) {
          const int c = deep_cmp(tb, n, word_id(w[k + 1]), word_id(w[k]), depth + keybases);
          undecided = undecided || c == 0;
          if (c < 0) {
            keys[pad(j - 1)] = w[k + 1];
            keys[pad(j)] = w[k];
          }
        }
      }
      if (!__syncthreads_or(undecided)) {
        load_window();
        ties = false;
      }
    }
  }
  uint64_t outv[kLeafPer];
  if (ties) {
    // short runs of equal keys: the thread that owns the first row of a run
    // orders the whole run by comparing the suffixes beyond the key
    for (uint32_t k = 0; k < per; ++k) {
      const uint32_t j = j0 + k;
      if (j + 1 >= size) break;
      const uint32_t key = word_key(keys[pad(j)]);
      if (word_key(keys[pad(j + 1)]) != key) continue;
      if (j > 0 && word_key(keys[pad(j - 1)]) == key) continue;
      uint32_t e = j + 2;
      while (e < size && e - j <= kDeepRun && word_key(keys[pad(e)]) == key) ++e;
      if (e - j > kDeepRun) continue;
      bool ok = true;
      for (uint32_t a = j + 1; a < e && ok; ++a) {
        const uint64_t v = keys[pad(a)];
        uint32_t b = a;
        while (b > j) {
          const int c = deep_cmp(tb, n, word_id(v), word_id(keys[pad(b - 1)]), depth + keybases);
          if (c == 0) ok = false;
          if (c >= 0) break;
          keys[pad(b)] = keys[pad(b - 1)];
          --b;
        }
        keys[pad(b)] = v;
      }
      if (ok)
        for (uint32_t a = j + 1; a < e; ++a) atomicOr(&settled[a >> 5], 1u << (a & 31u));
    }
    __syncthreads();
    load_window();
    uint32_t hv[kLeafPer];
    uint32_t local_max = 0, tied_rows = 0;
  #pragma unroll
    for (uint32_t k = 0; k < kLeafPer; ++k) {
      hv[k] = 0;
      const uint32_t j = j0 + k;
      if (k < per && j < size) {
        bool tie = false, tie_next = false;
        const uint64_t cur = w[k + 1];
        if (j > 0) {
          const uint64_t prv = w[k];
          const uint32_t pid = word_id(prv);
          // the previous suffix is final here if its key window ran past the
          // end of the text: a shorter suffix precedes every longer one it ties
          tie = word_key(prv) == word_key(cur) &&
                (uint64_t)pid + depth + keybases < n && !((settled[j >> 5] >> (j & 31u)) & 1u);
        }
        if (j + 1 < size) {
          const uint64_t nxt = w[k + 2];
          const uint32_t cid = word_id(cur);
          tie_next = word_key(nxt) == word_key(cur) &&
                     (uint64_t)cid + depth + keybases < n &&
                     !((settled[(j + 1) >> 5] >> ((j + 1) & 31u)) & 1u);
        }
        hv[k] = tie ? 0u : 1u + start + j;
        tied_rows += (tie || tie_next) ? 1u : 0u;
        local_max = hv[k] > local_max ? hv[k] : local_max;
      }
    }
    // inclusive max-scan of local_max over threads
    const uint32_t lane = t & 63u, wave = t >> 6;
    uint32_t v = local_max;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(v, off, 64);
      if (lane >= off) v = u > v ? u : v;
    }
    uint32_t tr = tied_rows;
    for (int off = 32; off; off >>= 1) tr += __shfl_down(tr, off, 64);
    if (lane == 63) wmax[wave] = v;
    if (lane == 0) wred[wave] = tr;
    __syncthreads();
    uint32_t before = 0;  // max over all earlier threads
    for (uint32_t q = 0; q < wave; ++q) before = wmax[q] > before ? wmax[q] : before;
    const uint32_t up = __shfl_up(v, 1, 64);
    if (lane) before = up > before ? up : before;
    if (t == 0) {
      uint32_t total = 0;
      for (int q = 0; q < kLeafWaves; ++q) total += wred[q];
      if (total) {
        atomicAdd(&ctr->unresolved, total);
        atomicMin(&ctr->min_depth, depth + keybases);
      }
    }
    // every thread has read its neighbours' keys; now reuse the LDS slots
    uint32_t run = before;
  #pragma unroll
    for (uint32_t k = 0; k < kLeafPer; ++k) {
      const uint32_t j = j0 + k;
      if (k < per && j < size) {
        const uint64_t cur = w[k + 1];
        const uint32_t id = word_id(cur);
        run = hv[k] > run ? hv[k] : run;
        // head row relative to the leaf (12 bits), BWT symbol, id: one word per row
        outv[k] = ((uint64_t)(cur & 3u) << 44) | ((uint64_t)(run - (1 + start)) << 32) | id;
        if (id == 0) ctr->endloc = 1 + start + j;
      }
    }
    __syncthreads();  // neighbours' keys have been read; the LDS slots are reused below
  } else {
    // no two neighbouring keys are equal: every row is its own group
#pragma unroll
    for (uint32_t k = 0; k < kLeafPer; ++k) {
      const uint32_t j = j0 + k;
      if (k < per && j < size) {
        const uint64_t cur = w[k + 1];
        const uint32_t id = word_id(cur);
        outv[k] = ((uint64_t)(cur & 3u) << 44) | ((uint64_t)j << 32) | id;
        if (id == 0) ctr->endloc = 1 + start + j;
      }
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < kLeafPer; ++k) {
    const uint32_t j = j0 + k;
    if (k < per && j < size) keys[pad(j)] = outv[k];
  }
  __syncthreads();
  if (t == 0) leaf_ties[leaf] = ties ? 1 : 0;
  for (uint32_t j = t; j < size; j += kLeafThreads) {
    const uint64_t o = keys[pad(j)];
    sa_rows[start + j] = (uint32_t)o;
    if (samples && !((1 + start + j) & kDevSampleMask)) samples[(1 + start + j) >> kDevSampleShift] = (uint32_t)o;
    // heads matter to prefix doubling only; a leaf without ties leaves them to k_ghead_identity
    if (ties) ghead[1 + start + j] = 1 + start + ((uint32_t)(o >> 32) & 0xFFFu);
    bw[start + j] = (uint8_t)(o >> 44);
  }
}

// The leaves of more than kWaveLeafMax rows (listed by k_part_classify): one workgroup each.
__global__ void __launch_bounds__(kLeafThreads, 2)
k_leaf_sort(const uint32_t *__restrict__ tb, const uint32_t *__restrict__ bufA,
            const uint32_t *__restrict__ bufB, const uint32_t *__restrict__ keyA,
            const uint32_t *__restrict__ keyB, const Leaf *__restrict__ leaves, uint32_t n,
            uint32_t *__restrict__ sa_rows, uint32_t *__restrict__ ghead, uint8_t *__restrict__ bw,
            uint8_t *__restrict__ leaf_ties, uint32_t *__restrict__ samples, Counters *__restrict__ ctr,
            const uint32_t *__restrict__ list, uint32_t first) {
  if (first + blockIdx.x >= ctr->nbig) return;
  leaf_sort_block(list[first + blockIdx.x], tb, bufA, bufB, keyA, keyB, leaves, n, sa_rows, ghead, bw, leaf_ties,
                  samples, ctr);
}

// The leaves the one-wave sort below left to a whole workgroup when it met them
// (keys so uneven that a counting bucket overflowed: repeats): how many is known
// on the device only, so the workgroups take them from the list in turn.
__global__ void __launch_bounds__(kLeafThreads)
k_leaf_sort_deferred(const uint32_t *__restrict__ tb, const uint32_t *__restrict__ bufA,
                     const uint32_t *__restrict__ bufB, const uint32_t *__restrict__ keyA,
                     const uint32_t *__restrict__ keyB, const Leaf *__restrict__ leaves, uint32_t n,
                     uint32_t *__restrict__ sa_rows, uint32_t *__restrict__ ghead, uint8_t *__restrict__ bw,
                     uint8_t *__restrict__ leaf_ties, uint32_t *__restrict__ samples, Counters *__restrict__ ctr,
                     const uint32_t *__restrict__ deferred) {
  const uint32_t count = ctr->ndeferred;
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    leaf_sort_block(deferred[i], tb, bufA, bufB, keyA, keyB, leaves, n, sa_rows, ghead, bw, leaf_ties, samples, ctr);
    __syncthreads();  // the next leaf reuses the LDS
  }
}

// ---- a leaf per wave ---------------------------------------------------------
//
// The workgroup-wide sort above spends most of its time waiting: ten phases
// separated by barriers of eight waves, three workgroups to a CU.  A leaf of up
// to kWaveLeafMax rows is sorted here by ONE wave with the same steps --
// interpolation buckets, a place inside the bucket by counting, the tie groups
// of neighbouring equal keys -- but no barrier at all: the lanes of a wave run in
// lockstep, so an LDS write is visible to the wave's next LDS read (wave_sync
// keeps the compiler from moving one across the other).  Sixteen waves to a CU,
// each on a leaf of its own and at a phase of its own, keep the LDS and the
// vector units busy while others wait for their loads.  The kernel is bound by
// instruction issue, so the sort proper is compiled for 4, 8, 12 and 16 rows a
// lane (a leaf takes the smallest that holds it) and only the last four rows of
// a lane are ever tested against the leaf's end.
constexpr uint32_t kWaveBuckets = 512;         // 16-bit counters, two per word
constexpr uint32_t kWaveBucketMax = 32;        // a fuller bucket sends the leaf to the workgroup-wide sort
constexpr int kWaveGroup = 1;                  // leaves (waves) per workgroup: a wave's LDS is free for the next leaf the moment it ends
// counters: kWaveBuckets / 2 words, then the word that says where a bucket past the last would start,
// then (two words on) the word that takes the increments of slots past the leaf's end
constexpr uint32_t kWaveCntWords = kWaveBuckets / 2 + 4;
constexpr uint32_t kWaveDummyCnt = kWaveBuckets / 2 + 2;
// words of a sort of up to `rows` rows: the rows, kWaveBucketMax sentinels behind the last row, and one
// slot for the words of slots past the end
constexpr uint32_t wave_dummy_key(uint32_t rows) { return rows + kWaveBucketMax + 8; }
constexpr uint32_t wave_key_slots(uint32_t rows) { return wave_dummy_key(rows) + 8; }

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct WaveLeaf {
  uint32_t leaf, size, start, depth, keybases;
  const uint32_t *src, *srckey;
};

// Sorts the leaf's words into keys[0 .. size).  Returns 0 when a bucket
// overflowed (nothing written that matters), 2 when two neighbouring rows have
// equal keys (the caller settles the tie groups from the sorted words), and 1
// when every key is distinct: the rows, their BWT symbols and samples have then
// been written.  size in (64 (E - 4), 64 E].
template <int E, uint32_t DUMMY /* wave_dummy_key of the LDS in use */>
__device__ __forceinline__ int wave_leaf_sort(const WaveLeaf &L, const uint32_t *__restrict__ tb, uint64_t *keys,
                                              uint32_t *cnt, uint32_t lane, uint32_t n,
                                              uint32_t *__restrict__ sa_rows, uint8_t *__restrict__ bw,
                                              uint32_t *__restrict__ samples, Counters *__restrict__ ctr) {
  const uint32_t size = L.size, start = L.start;
  uint64_t r[E];
  bool ok[4];  // whether the lane's last four slots are rows of the leaf; the others always are
  auto row_of = [&](int m) { return 4u * ((uint32_t)(m / 4) * 64u + lane) + (uint32_t)(m % 4); };
#pragma unroll
  for (int e = 0; e < 4; ++e) ok[e] = row_of(E - 4 + e) < size;
  auto valid = [&](int m) { return m < E - 4 ? true : ok[m - (E - 4)]; };
  typedef uint32_t u32x4_at_4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
  for (int u = 0; u < E / 4; ++u) {
    const uint32_t j0 = row_of(4 * u);
    if (L.src && (u < E / 4 - 1 || ok[3])) {
      const u32x4_at_4 k4 = *reinterpret_cast<const u32x4_at_4 *>(L.srckey + start + j0);
      const u32x4_at_4 i4 = *reinterpret_cast<const u32x4_at_4 *>(L.src + start + j0);
      r[4 * u] = leaf_word(k4.x, i4.x);
      r[4 * u + 1] = leaf_word(k4.y, i4.y);
      r[4 * u + 2] = leaf_word(k4.z, i4.z);
      r[4 * u + 3] = leaf_word(k4.w, i4.w);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        r[4 * u + e] = j0 + e < size ? leaf_fetch(tb, L.src, L.srckey, start + j0 + e, L.depth) : ~0ull;
    }
  }
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const uint32_t k = word_key(r[m]);
    const uint32_t klo = valid(m) ? k : 0xFFFFFFFFu, khi = valid(m) ? k : 0u;
    kmin = klo < kmin ? klo : kmin;
    kmax = khi > kmax ? khi : kmax;
  }
  for (int off = 32; off; off >>= 1) {
    const uint32_t a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
  }
  reinterpret_cast<uint4 *>(cnt)[lane] = make_uint4(0, 0, 0, 0);
  // Words behind the last row that compare above every word: the count of smaller words
  // below runs a fixed number of places on from a bucket's start, past its end
  if (lane < kWaveBucketMax) keys[size + lane] = ~0ull;
  wave_sync();
  // counting pass: bucket by interpolation between the smallest and the largest key
  const float scale = (float)kWaveBuckets / (float)(kmax - kmin + 1u);
  uint32_t at[E];  // bucket | place among the bucket's words << 16, then the word's place in the leaf
#pragma unroll
  for (int m = 0; m < E; ++m) {
    uint32_t b = (uint32_t)((float)(word_key(r[m]) - kmin) * scale);
    b = b < kWaveBuckets ? b : kWaveBuckets - 1;
    const uint32_t sh = 16u * (b & 1u);
    const uint32_t word = valid(m) ? b >> 1 : kWaveDummyCnt;
    at[m] = b | ((atomicAdd(&cnt[word], 1u << sh) >> sh) << 16);
  }
  wave_sync();
  uint32_t fullest = 0;
  {
    // exclusive scan of the bucket counts; lane l owns buckets [8 l, 8 l + 8)
    const uint4 q = reinterpret_cast<const uint4 *>(cnt)[lane];
    const uint32_t own[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16,
                             q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
    uint32_t sum = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      sum += own[b];
      fullest = own[b] > fullest ? own[b] : fullest;
    }
    uint32_t v = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    for (int off = 32; off; off >>= 1) {
      const uint32_t u = __shfl_xor(fullest, off, 64);
      fullest = u > fullest ? u : fullest;
    }
    fullest = __builtin_amdgcn_readfirstlane(fullest);
    if (fullest > kWaveBucketMax) return 0;
    uint32_t run = v - sum, w4[4];
#pragma unroll
    for (int b = 0; b < 8; b += 2) {
      w4[b / 2] = run | ((run + own[b]) << 16);  // start rows of the two buckets (<= 1024)
      run += own[b] + own[b + 1];
    }
    reinterpret_cast<uint4 *>(cnt)[lane] = make_uint4(w4[0], w4[1], w4[2], w4[3]);  // a lane's own words: no one else's
  }
  wave_sync();
  // the words go to their bucket in arrival order; at[] becomes the place of the bucket's first word
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const uint32_t b = at[m] & 0xFFFFu;
    const uint32_t s0 = (cnt[b >> 1] >> (16u * (b & 1u))) & 0xFFFFu;
    keys[valid(m) ? s0 + (at[m] >> 16) : DUMMY] = r[m];
    at[m] = s0;
  }
  wave_sync();
  // Order inside the buckets: every word counts the smaller words among the `fullest`
  // (the fullest bucket's count) words from its bucket's start on.  Words are distinct
  // (they hold the suffix start) and those of later buckets, like the sentinels behind
  // the last row, are larger: they count for nothing.
  uint32_t rank[E];
#pragma unroll
  for (int m = 0; m < E; ++m) rank[m] = 0;
#pragma unroll 2
  for (uint32_t i = 0; i < fullest; ++i) {
#pragma unroll
    for (int m = 0; m < E; ++m) rank[m] += keys[at[m] + i] < r[m] ? 1u : 0u;
  }
  wave_sync();
#pragma unroll
  for (int m = 0; m < E; ++m) keys[valid(m) ? at[m] + rank[m] : DUMMY] = r[m];
  wave_sync();
  // sorted.  Row j = lane + 64 i: its neighbour j - 1 is the lane before (lane 63 of the round before)
  uint64_t tied[E];  // per round, the lanes whose row has the key of the row before it (the same in every lane)
  uint64_t any_tied = 0, longer = 0, carry = 0;
  uint32_t before = 0xFFFFFFFFu;  // key of row 64 i - 1; no key has 32 bits
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const uint32_t j = lane + 64u * (uint32_t)i;
    r[i] = (i < E - 4 || j < size) ? keys[j] : ~0ull;
    const uint32_t k = word_key(r[i]);
    const uint32_t up = __shfl_up(k, 1, 64);
    const uint32_t prev = lane ? up : before;
    tied[i] = __ballot(j < size && prev == k);
    any_tied |= tied[i];
    longer |= tied[i] & ((tied[i] << 1) | carry);  // the row before is tied to its own neighbour as well
    carry = tied[i] >> 63;
    before = __shfl(k, 63, 64);
  }
  if (any_tied) {
    if (longer) return 2;  // a run of three or more equal keys
    // Pairs only (a text of 3 x 10^9 bases: a few in every leaf).  The lane of a pair's
    // second row compares the two suffixes beyond the key and swaps the words in LDS.
    bool undecided = false;
#pragma unroll
    for (int i = 0; i < E; ++i) {
      if (!tied[i]) continue;
      if ((tied[i] >> lane) & 1u) {
        const uint32_t j = lane + 64u * (uint32_t)i;
        const uint64_t pw = keys[j - 1];
        const int c = deep_cmp(tb, n, word_id(r[i]), word_id(pw), L.depth + L.keybases);
        undecided = undecided || c == 0;
        if (c < 0) {
          keys[j - 1] = r[i];
          keys[j] = pw;
        }
      }
    }
    if (__ballot(undecided)) return 2;  // equal for 512 bases more: the tie groups' business
    wave_sync();
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const uint32_t j = lane + 64u * (uint32_t)i;
      r[i] = (i < E - 4 || j < size) ? keys[j] : ~0ull;
    }
  }
  // every row is its own group
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const uint32_t j = lane + 64u * (uint32_t)i;
    if (i < E - 4 || j < size) {
      const uint32_t id = word_id(r[i]), row = start + j;
      sa_rows[row] = id;
      if (samples && !((1 + row) & kDevSampleMask)) samples[(1 + row) >> kDevSampleShift] = id;
      bw[row] = (uint8_t)(r[i] & 3u);
      if (id == 0) ctr->endloc = 1 + row;
    }
  }
  return 1;
}

// One leaf of up to MAXROWS rows, by the wave that calls it (every lane); keys / cnt: the wave's own LDS.
template <uint32_t MAXROWS>
__device__ __forceinline__ void
wave_leaf(const uint32_t leaf, const Leaf lf, const uint32_t lane, uint64_t *keys, uint32_t *cnt,
          const uint32_t *__restrict__ tb, const uint32_t *__restrict__ bufA, const uint32_t *__restrict__ bufB,
          const uint32_t *__restrict__ keyA, const uint32_t *__restrict__ keyB, uint32_t n,
          uint32_t *__restrict__ sa_rows, uint32_t *__restrict__ ghead, uint8_t *__restrict__ bw,
          uint8_t *__restrict__ leaf_ties, uint32_t *__restrict__ samples, Counters *__restrict__ ctr,
          uint32_t *__restrict__ deferred) {
  constexpr int E = MAXROWS / 64;  // rows a lane owns, at most
  constexpr uint32_t DUMMY = wave_dummy_key(MAXROWS);
  static_assert((E == 12 || E == 16) && kWaveBuckets / 2 == 4 * 64, "twelve or sixteen rows, four counter words a lane");
  WaveLeaf L;
  L.leaf = leaf;
  L.size = __builtin_amdgcn_readfirstlane(lf.size);
  L.start = __builtin_amdgcn_readfirstlane(lf.start);
  L.depth = __builtin_amdgcn_readfirstlane(lf.depth);
  const uint32_t srcw = __builtin_amdgcn_readfirstlane(lf.src), srcid = srcw & 255u;
  L.keybases = srcw >> 8;
  L.src = srcid == kSrcA ? bufA : srcid == kSrcB ? bufB : nullptr;
  L.srckey = srcid == kSrcA ? keyA : keyB;
  const uint32_t size = L.size, start = L.start, depth = L.depth, keybases = L.keybases;
  if (size > MAXROWS) return;  // on another sort's list
  if (size == 1) {
    if (lane == 0) {
      const uint64_t w = leaf_fetch(tb, L.src, L.srckey, start, depth);
      const uint32_t id = word_id(w);
      sa_rows[start] = id;
      if (samples && !((1 + start) & kDevSampleMask)) samples[(1 + start) >> kDevSampleShift] = id;
      leaf_ties[leaf] = 0;
      bw[start] = (uint8_t)(w & 3u);
      if (id == 0) ctr->endloc = 1 + start;
    }
    return;
  }
  int how;
  if (MAXROWS > kWaveLeafMax) how = wave_leaf_sort<E, DUMMY>(L, tb, keys, cnt, lane, n, sa_rows, bw, samples, ctr);  // listed: all of a size
  else if (size <= 256) how = wave_leaf_sort<4, DUMMY>(L, tb, keys, cnt, lane, n, sa_rows, bw, samples, ctr);
  else if (size <= 512) how = wave_leaf_sort<8, DUMMY>(L, tb, keys, cnt, lane, n, sa_rows, bw, samples, ctr);
  else how = wave_leaf_sort<12, DUMMY>(L, tb, keys, cnt, lane, n, sa_rows, bw, samples, ctr);
  if (how == 0) {
    if (lane == 0) deferred[atomicAdd(&ctr->ndeferred, 1u)] = leaf;
    return;
  }
  if (how == 1) {
    if (lane == 0) leaf_ties[leaf] = 0;
    return;
  }
  // ---- neighbouring rows with equal keys: the tie groups, from the sorted words.
  // Blocked: lane l owns rows [l per, l per + per)
  uint32_t *settled = cnt;  // bit j: row j is ordered against row j-1 (the counters are done with)
  wave_sync();
  if (lane < MAXROWS / 32) settled[lane] = 0;
  wave_sync();
  const uint32_t per = (size + 63u) / 64u;  // 1 .. E
  const uint32_t j0 = lane * per;
  uint64_t w[E + 2];  // the lane's rows and one neighbour on each side: w[1 + k] is row j0 + k
  // short runs of equal keys: the lane that owns the first row of a run orders the
  // whole run by comparing the suffixes beyond the key
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t j = j0 + k;
    if (j + 1 >= size) break;
    const uint32_t key = word_key(keys[j]);
    if (word_key(keys[j + 1]) != key) continue;
    if (j > 0 && word_key(keys[j - 1]) == key) continue;
    uint32_t e = j + 2;
    while (e < size && e - j <= kDeepRun && word_key(keys[e]) == key) ++e;
    if (e - j > kDeepRun) continue;
    bool ok = true;
    for (uint32_t a = j + 1; a < e && ok; ++a) {
      const uint64_t v = keys[a];
      uint32_t b = a;
      while (b > j) {
        const int c = deep_cmp(tb, n, word_id(v), word_id(keys[b - 1]), depth + keybases);
        if (c == 0) ok = false;
        if (c >= 0) break;
        keys[b] = keys[b - 1];
        --b;
      }
      keys[b] = v;
    }
    if (ok)
      for (uint32_t a = j + 1; a < e; ++a) atomicOr(&settled[a >> 5], 1u << (a & 31u));
  }
  wave_sync();
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)E + 2; ++k) {
    const uint32_t j = j0 + k - 1u;  // wraps below row 0, which then fails j < size
    w[k] = (k <= per + 1 && j < size) ? keys[j] : ~0ull;
  }
  uint32_t hv[E];
  uint32_t local_max = 0, tied_rows = 0;
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)E; ++k) {
    hv[k] = 0;
    const uint32_t j = j0 + k;
    if (k < per && j < size) {
      bool tie = false, tie_next = false;
      const uint64_t cur = w[k + 1];
      if (j > 0) {
        const uint64_t prv = w[k];
        const uint32_t pid = word_id(prv);
        // the previous suffix is final here if its key window ran past the
        // end of the text: a shorter suffix precedes every longer one it ties
        tie = word_key(prv) == word_key(cur) &&
              (uint64_t)pid + depth + keybases < n && !((settled[j >> 5] >> (j & 31u)) & 1u);
      }
      if (j + 1 < size) {
        const uint64_t nxt = w[k + 2];
        const uint32_t cid = word_id(cur);
        tie_next = word_key(nxt) == word_key(cur) &&
                   (uint64_t)cid + depth + keybases < n &&
                   !((settled[(j + 1) >> 5] >> ((j + 1) & 31u)) & 1u);
      }
      hv[k] = tie ? 0u : 1u + start + j;
      tied_rows += (tie || tie_next) ? 1u : 0u;
      local_max = hv[k] > local_max ? hv[k] : local_max;
    }
  }
  // inclusive max-scan of local_max over the lanes
  uint32_t v = local_max;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(v, off, 64);
    if (lane >= off) v = u > v ? u : v;
  }
  uint32_t tr = tied_rows;
  for (int off = 32; off; off >>= 1) tr += __shfl_xor(tr, off, 64);
  const uint32_t up = __shfl_up(v, 1, 64);
  uint32_t run = lane ? up : 0u;  // max over all earlier lanes
  if (lane == 0 && tr) {
    atomicAdd(&ctr->unresolved, tr);
    atomicMin(&ctr->min_depth, depth + keybases);
  }
  wave_sync();  // neighbours' keys have been read; the LDS slots are reused below
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)E; ++k) {
    const uint32_t j = j0 + k;
    if (k < per && j < size) {
      const uint64_t cur = w[k + 1];
      const uint32_t id = word_id(cur);
      run = hv[k] > run ? hv[k] : run;
      // head row relative to the leaf, BWT symbol, id: one word per row
      keys[j] = ((uint64_t)(cur & 3u) << 44) | ((uint64_t)(run - (1 + start)) << 32) | id;
      if (id == 0) ctr->endloc = 1 + start + j;
    }
  }
  wave_sync();
  if (lane == 0) leaf_ties[leaf] = 1;
  for (uint32_t j = lane; j < size; j += 64) {
    const uint64_t o = keys[j];
    sa_rows[start + j] = (uint32_t)o;
    if (samples && !((1 + start + j) & kDevSampleMask)) samples[(1 + start + j) >> kDevSampleShift] = (uint32_t)o;
    ghead[1 + start + j] = 1 + start + ((uint32_t)(o >> 32) & 0xFFFu);
    bw[start + j] = (uint8_t)(o >> 44);
  }
}

__global__ void __launch_bounds__(64 * kWaveGroup) __attribute__((amdgpu_waves_per_eu(5, 5)))
k_leaf_sort_wave(const uint32_t *__restrict__ tb, const uint32_t *__restrict__ bufA,
                 const uint32_t *__restrict__ bufB, const uint32_t *__restrict__ keyA,
                 const uint32_t *__restrict__ keyB, const Leaf *__restrict__ leaves, uint32_t n,
                 uint32_t *__restrict__ sa_rows, uint32_t *__restrict__ ghead, uint8_t *__restrict__ bw,
                 uint8_t *__restrict__ leaf_ties, uint32_t *__restrict__ samples, Counters *__restrict__ ctr,
                 uint32_t first_leaf, uint32_t *__restrict__ deferred) {
  __shared__ uint64_t keys_all[kWaveGroup][wave_key_slots(kWaveLeafMax)];
  __shared__ uint32_t cnt_all[kWaveGroup][kWaveCntWords];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t leaf = first_leaf + blockIdx.x * kWaveGroup + wv;
  // the grid may be sized by a guess made before the host knew the number of leaves
  if (leaf >= ctr->nleaves) return;
  wave_leaf<kWaveLeafMax>(leaf, leaves[leaf], lane, keys_all[wv], cnt_all[wv], tb, bufA, bufB, keyA, keyB, n, sa_rows,
                          ghead, bw, leaf_ties, samples, ctr, deferred);
}

// The same for the listed leaves of kWaveLeafMax + 1 .. kWaveLeafMid rows (single children just too
// large for the sort above): sixteen rows a lane and the LDS for them, four waves to a SIMD.
__global__ void __launch_bounds__(64 * kWaveGroup) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_leaf_sort_wave_mid(const uint32_t *__restrict__ tb, const uint32_t *__restrict__ bufA,
                     const uint32_t *__restrict__ bufB, const uint32_t *__restrict__ keyA,
                     const uint32_t *__restrict__ keyB, const Leaf *__restrict__ leaves, uint32_t n,
                     uint32_t *__restrict__ sa_rows, uint32_t *__restrict__ ghead, uint8_t *__restrict__ bw,
                     uint8_t *__restrict__ leaf_ties, uint32_t *__restrict__ samples, Counters *__restrict__ ctr,
                     const uint32_t *__restrict__ list, uint32_t *__restrict__ deferred) {
  __shared__ uint64_t keys_all[kWaveGroup][wave_key_slots(kWaveLeafMid)];
  __shared__ uint32_t cnt_all[kWaveGroup][kWaveCntWords];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t at = blockIdx.x * kWaveGroup + wv;
  if (at >= ctr->nmid) return;
  const uint32_t leaf = list[at];
  wave_leaf<kWaveLeafMid>(leaf, leaves[leaf], lane, keys_all[wv], cnt_all[wv], tb, bufA, bufB, keyA, keyB, n, sa_rows,
                          ghead, bw, leaf_ties, samples, ctr, deferred);
}

// Before prefix doubling: every row of a leaf that found no ties is its own group.
__global__ void __launch_bounds__(256)
k_ghead_identity(const Leaf *__restrict__ leaves, const uint8_t *__restrict__ leaf_ties,
                 uint32_t *__restrict__ ghead) {
  if (leaf_ties[blockIdx.x]) return;
  const Leaf lf = leaves[blockIdx.x];
  for (uint32_t j = threadIdx.x; j < lf.size; j += 256) ghead[1 + lf.start + j] = 1 + lf.start + j;
}

// Segments still above kLeafMax when the bucket levels stop: one tie group
// each.  Suffixes shorter than `depth` got here through the zero padding; they
// precede the rest of their segment, shortest first, and are final.  There are
// fewer than `depth` of them in the whole text.  A segment can be most of the
// text (a tandem repeat of period 7 has seven), so its rows are dealt to
// kUnresParts workgroups: k_unresolved_copy moves the rows, makes the segment one
// group and lists the short suffixes; k_unresolved_short orders those few;
// k_unresolved_heads moves the group's head behind them where there are any.
constexpr uint32_t kUnresParts = 64;
constexpr uint32_t kUnresList = kMaxDepth + 8;  // short[seg * (kUnresList + 1)]: count, then rows

__global__ void __launch_bounds__(256)
k_unresolved_copy(const Seg *__restrict__ segs, const uint32_t *__restrict__ src,
                  uint32_t *__restrict__ sa_rows, uint32_t *__restrict__ ghead, uint32_t depth, uint32_t n,
                  uint32_t *__restrict__ shortl) {
  const uint32_t seg = blockIdx.x / kUnresParts, part = blockIdx.x % kUnresParts;
  const Seg sg = segs[seg];
  const uint32_t slice = (sg.size + kUnresParts - 1) / kUnresParts;
  const uint32_t lo = part * slice, hi = lo + slice < sg.size ? lo + slice : sg.size;
  for (uint32_t j = lo + threadIdx.x; j < hi; j += 256) {
    const uint32_t row = sg.start + j;
    const uint32_t id = src[row];
    sa_rows[row] = id;
    ghead[1 + row] = 1 + sg.start;
    if ((uint64_t)id + depth > n) {
      const uint32_t k = atomicAdd(&shortl[(uint64_t)seg * (kUnresList + 1)], 1u);
      if (k < kUnresList) shortl[(uint64_t)seg * (kUnresList + 1) + 1 + k] = row;
    }
  }
}

__global__ void k_unresolved_short(const Seg *__restrict__ segs, uint32_t *__restrict__ sa_rows,
                                   uint32_t *__restrict__ ghead, uint32_t depth, uint32_t n,
                                   const uint32_t *__restrict__ shortl, uint32_t nsegs,
                                   Counters *__restrict__ ctr) {
  const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= nsegs) return;
  const Seg sg = segs[seg];
  const uint32_t *mine = shortl + (uint64_t)seg * (kUnresList + 1);
  const uint32_t E = mine[0] < kUnresList ? mine[0] : kUnresList;
  if (E) {
    // move the short suffixes to the first E rows (the order of the others is
    // free), then order them by decreasing start
    uint32_t lo = sg.start;
    for (uint32_t k = 0; k < E; ++k) {
      const uint32_t r = mine[1 + k];
      if (r < sg.start + E) continue;  // already in the front part
      while ((uint64_t)sa_rows[lo] + depth > n) ++lo;  // first front row holding a long suffix
      const uint32_t tmp = sa_rows[lo];
      sa_rows[lo] = sa_rows[r];
      sa_rows[r] = tmp;
      ++lo;
    }
    for (uint32_t a = 1; a < E; ++a) {
      const uint32_t v = sa_rows[sg.start + a];
      uint32_t b = a;
      while (b > 0 && sa_rows[sg.start + b - 1] < v) {
        sa_rows[sg.start + b] = sa_rows[sg.start + b - 1];
        --b;
      }
      sa_rows[sg.start + b] = v;
    }
    for (uint32_t j = 0; j < E; ++j) ghead[1 + sg.start + j] = 1 + sg.start + j;
  }
  if (sg.size - E >= 2) {
    atomicAdd(&ctr->unresolved, sg.size - E);
    atomicMin(&ctr->min_depth, depth);
  }
}

__global__ void __launch_bounds__(256)
k_unresolved_heads(const Seg *__restrict__ segs, uint32_t *__restrict__ ghead, const uint32_t *__restrict__ shortl) {
  const uint32_t seg = blockIdx.x / kUnresParts, part = blockIdx.x % kUnresParts;
  const uint32_t *mine = shortl + (uint64_t)seg * (kUnresList + 1);
  const uint32_t E = mine[0] < kUnresList ? mine[0] : kUnresList;
  if (!E) return;  // the head k_unresolved_copy wrote stands
  const Seg sg = segs[seg];
  const uint32_t slice = (sg.size + kUnresParts - 1) / kUnresParts;
  const uint32_t lo = part * slice, hi = lo + slice < sg.size ? lo + slice : sg.size;
  for (uint32_t j = lo + threadIdx.x; j < hi; j += 256)
    if (j >= E) ghead[1 + sg.start + j] = 1 + sg.start + E;
}

// ---- prefix doubling over the remaining tie groups -------------------------------

__global__ void k_isa_init(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ ghead,
                           uint32_t *__restrict__ isa, uint64_t rows) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < rows) isa[sa[j]] = ghead[j];
}

__global__ void k_mark_active(const uint32_t *__restrict__ ghead, uint32_t *__restrict__ flags,
                              uint64_t rows) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= rows) return;
  const bool tied_prev = ghead[j] != (uint32_t)j;
  const bool tied_next = j + 1 < rows && ghead[j + 1] == (uint32_t)j;
  flags[j] = (tied_prev || tied_next) ? 1u : 0u;
}

__global__ void k_compact_keys(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ offs,
                               const uint32_t *__restrict__ sa, const uint32_t *__restrict__ ghead,
                               const uint32_t *__restrict__ isa, uint32_t n, uint64_t h, int kb,
                               uint64_t rows, uint32_t *__restrict__ rowof,
                               uint64_t *__restrict__ K, uint32_t *__restrict__ V) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= rows || !flags[j]) return;
  const uint32_t a = offs[j];
  const uint32_t id = sa[j];
  const uint64_t nxt = (uint64_t)id + h;
  // a suffix in a tie group shares at least h bases with another, so id+h <= n;
  // row 0 (rank 0) is the end-of-text suffix and orders a suffix that ends
  // exactly there before the others
  const uint32_t key2 = nxt <= n ? isa[nxt] : 0u;
  rowof[a] = (uint32_t)j;
  K[a] = ((uint64_t)ghead[j] << kb) | key2;
  V[a] = id;
}

__global__ void k_group_heads(const uint64_t *__restrict__ K, const uint32_t *__restrict__ rowof,
                              uint32_t A, uint32_t *__restrict__ hv) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  hv[a] = (a == 0 || K[a] != K[a - 1]) ? rowof[a] : 0u;
}

__global__ void k_doubling_apply(const uint32_t *__restrict__ V, const uint32_t *__restrict__ rowof,
                                 const uint32_t *__restrict__ newhead, uint32_t A,
                                 uint32_t *__restrict__ sa, uint32_t *__restrict__ ghead,
                                 uint32_t *__restrict__ isa) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  const uint32_t row = rowof[a], id = V[a], head = newhead[a];
  sa[row] = id;
  ghead[row] = head;
  isa[id] = head;
}

struct SaStats {
  uint64_t v[8];
};
static thread_local SaStats g_stats;

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 256); }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// SA rows partly ordered, ghead[row] = first row of the row's tie group (full
// row numbering, row 0 = the $ suffix), every tie group ordered by at least h0
// bases.  Finishes the order by prefix doubling restricted to the tie groups.
static int sa_doubling(uint32_t *d_sa, uint32_t *ghead, uint32_t n, uint32_t h0, uint32_t *flags,
                       Counters *d_ctr, hipStream_t s) {
  const uint64_t rows = (uint64_t)n + 1;
  int kb = 1;
  while (kb < 32 && (n >> kb)) ++kb;  // values <= n fit in kb bits
  DevBuf isa, offs, scanws;
  hipError_t e = isa.alloc(rows * 4);
  if (e == hipSuccess) e = offs.alloc(rows * 4);
  if (e == hipSuccess) e = scanws.alloc(scan_workspace_bytes(rows));
  if (e != hipSuccess) return set_error(kErrHip, "prefix doubling buffers: %s", hipGetErrorString(e));
  const unsigned gr = (unsigned)div_up(rows, 256);
  BWT_LAUNCH("dbl_isa_init", k_isa_init, dim3(gr), dim3(256), 0, s, d_sa, ghead, isa.as<uint32_t>(), rows);
  DevBuf K, Kalt, V, Valt, rowof, hv, sortws;
  uint64_t cap = 0;
  uint64_t h = h0 ? h0 : 1;
  uint32_t prevA = 0xFFFFFFFFu;
  for (int round = 0; round < 64; ++round) {
    BWT_LAUNCH("dbl_mark_active", k_mark_active, dim3(gr), dim3(256), 0, s, ghead, flags, rows);
    BWT_TRY(device_scan_u32(ScanOp::kAddExclusive, flags, offs.as<uint32_t>(), rows, &d_ctr->active,
                            scanws.p, scan_workspace_bytes(rows), s));
    uint32_t A = 0;
    BWT_HIP(hipMemcpyAsync(&A, &d_ctr->active, 4, hipMemcpyDeviceToHost, s));
    BWT_HIP(hipStreamSynchronize(s));
    if (A == 0) {
      g_stats.v[3] = round;
      return kOk;
    }
    if (A > prevA) return set_error(kErrInternal, "prefix doubling: active rows grew %u -> %u", prevA, A);
    prevA = A;
    if (!cap) {
      cap = A;
      e = K.alloc(cap * 8);
      if (e == hipSuccess) e = Kalt.alloc(cap * 8);
      if (e == hipSuccess) e = V.alloc(cap * 4);
      if (e == hipSuccess) e = Valt.alloc(cap * 4);
      if (e == hipSuccess) e = rowof.alloc(cap * 4);
      if (e == hipSuccess) e = hv.alloc(cap * 4);
      if (e == hipSuccess) e = sortws.alloc(radix_sort_workspace_bytes(cap));
      if (e != hipSuccess) return set_error(kErrHip, "prefix doubling sort buffers: %s", hipGetErrorString(e));
    }
    BWT_LAUNCH("dbl_compact_keys", k_compact_keys, dim3(gr), dim3(256), 0, s, flags, offs.as<uint32_t>(),
               d_sa, ghead, isa.as<uint32_t>(), n, h, kb, rows, rowof.as<uint32_t>(), K.as<uint64_t>(),
               V.as<uint32_t>());
    BWT_TRY(radix_sort_pairs(K.as<uint64_t>(), V.as<uint32_t>(), Kalt.as<uint64_t>(), Valt.as<uint32_t>(),
                             A, 2 * kb, sortws.p, radix_sort_workspace_bytes(cap), s));
    const unsigned ga = (unsigned)div_up(A, 256);
    BWT_LAUNCH("dbl_group_heads", k_group_heads, dim3(ga), dim3(256), 0, s, K.as<uint64_t>(),
               rowof.as<uint32_t>(), A, hv.as<uint32_t>());
    BWT_TRY(device_scan_u32(ScanOp::kMaxInclusive, hv.as<uint32_t>(), hv.as<uint32_t>(), A, nullptr,
                            scanws.p, scan_workspace_bytes(rows), s));
    BWT_LAUNCH("dbl_apply", k_doubling_apply, dim3(ga), dim3(256), 0, s, V.as<uint32_t>(),
               rowof.as<uint32_t>(), hv.as<uint32_t>(), A, d_sa, ghead, isa.as<uint32_t>());
    BWT_CHECK_LAUNCH();
    h *= 2;
  }
  return set_error(kErrInternal, "prefix doubling did not converge");
}

// ---- prefix doubling, group by group ---------------------------------------------
//
// The tie groups are runs of neighbouring rows, so a doubling round never has to
// sort across groups: rows only move inside their group.  A round with offset h
//   1. notes, per tile of kDblTile rows, the first and last group head in it and
//      (k_dbl_next_head) the first head after it: enough to know every group's size;
//   2. gathers the key of every row still tied, rank[start + h]       (k_dbl_keys);
//   3. splits groups of more than kDblTile rows by an 8-bit digit of the key,
//      interpolated between the group's smallest and largest key, level by
//      level until the pieces are small or hold one key         (k_dbl_hist, .._scatter);
//   4. sorts every group of up to kDblTile rows in LDS and gives equal keys a
//      group of their own                                            (k_dbl_small).
// The rank of a suffix is a label of its group: some row of the group, the first
// one to begin with.  When a group splits, the piece that holds the label's row
// keeps the label and the others take their first row -- labels stay in row
// order, and a piece of more than half the group (what a repeat leaves when a
// few suffixes part from it) keeps its label wherever it lies, so that ranks
// are rewritten for the rows that parted only.  Against one stable 64-bit LSD
// sort of (group, key) over all tied rows per round, a row moves once or twice
// per round instead of seven times.

constexpr uint32_t kDblTile = 2048;           // rows per tile; a group of up to this many rows is sorted in LDS
constexpr uint32_t kDblChunk = 4096;          // rows a partition workgroup holds at a time (256 threads x 16)
constexpr uint32_t kDblChunks = 4;            // ... and chunks it takes one after the other: one claim of room for all
constexpr uint32_t kDblPart = kDblChunk * kDblChunks;
constexpr uint32_t kDblPartThreads = 512;  // two waves a SIMD more than 256 for the same LDS; the 256 bins are the first 256 threads'
constexpr uint32_t kDblItems = kDblChunk / kDblPartThreads;
constexpr uint32_t kDblNone = 0xFFFFFFFFu;

struct DblCtr {
  uint32_t active;       // rows in groups of two or more
  uint32_t nbig;         // groups of more than kDblTile rows found by k_dbl_emit_big
  uint32_t nnext;        // pieces of more than kDblTile rows the current level left
  uint32_t total_tiles;  // partition tiles of the current level
};

__global__ void k_dbl_init(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ ghead,
                           uint32_t *__restrict__ isa, uint32_t *__restrict__ glabel, uint64_t rows) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= rows) return;
  isa[sa[j]] = ghead[j];
  glabel[j] = (uint32_t)j;  // read at group heads only
}

__global__ void __launch_bounds__(256)
k_dbl_tiles(const uint32_t *__restrict__ ghead, uint64_t rows, uint32_t *__restrict__ first_head,
            uint32_t *__restrict__ last_head, uint32_t *__restrict__ tied_rows) {
  static_assert(kDblTile == 256 * 8, "eight consecutive rows per thread");
  __shared__ uint32_t lo, hi, act;
  if (threadIdx.x == 0) {
    lo = kDblNone;
    hi = 0;
    act = 0;
  }
  __syncthreads();
  const uint64_t r0 = (uint64_t)blockIdx.x * kDblTile + 8ull * threadIdx.x;
  uint32_t g[9];
  if (r0 + 9 <= rows) {  // ghead is 16-byte aligned and so is every thread's first row
    const uint4 a = reinterpret_cast<const uint4 *>(ghead + r0)[0], b = reinterpret_cast<const uint4 *>(ghead + r0)[1];
    g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w;
    g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
    g[8] = ghead[r0 + 8];
  } else {
#pragma unroll
    for (uint32_t k = 0; k < 9; ++k) g[k] = r0 + k < rows ? ghead[r0 + k] : kDblNone;  // no row has this head
  }
  uint32_t mylo = kDblNone, myhi = 0, mine = 0;
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    const uint64_t j = r0 + k;
    if (j < rows) {
      if (g[k] == (uint32_t)j) {
        mylo = mylo == kDblNone ? (uint32_t)j : mylo;
        myhi = (uint32_t)j;
        mine += g[k + 1] == (uint32_t)j ? 1u : 0u;
      } else {
        ++mine;
      }
    }
  }
  for (int off = 32; off; off >>= 1) {
    const uint32_t x = __shfl_xor(mylo, off, 64), y = __shfl_xor(myhi, off, 64);
    mylo = x < mylo ? x : mylo;
    myhi = y > myhi ? y : myhi;
    mine += __shfl_xor(mine, off, 64);
  }
  if ((threadIdx.x & 63u) == 0) {
    if (mylo != kDblNone) {
      atomicMin(&lo, mylo);
      atomicMax(&hi, myhi);
    }
    if (mine) atomicAdd(&act, mine);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    first_head[blockIdx.x] = lo;
    last_head[blockIdx.x] = lo == kDblNone ? kDblNone : hi;
    tied_rows[blockIdx.x] = act;  // summed by k_dbl_next_head: one counter for all tiles would serialise them
  }
}

// nxt[i] = the first group head in a tile after tile i, `rows` if there is none;
// also the sum of the tiles' tied rows
__global__ void __launch_bounds__(1024)
k_dbl_next_head(const uint32_t *__restrict__ first_head, const uint32_t *__restrict__ tied_rows, uint32_t ntiles,
                uint32_t rows, uint32_t *__restrict__ nxt, DblCtr *__restrict__ ctr) {
  __shared__ uint32_t wmin[16], wsum[16];
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  uint32_t carry = rows, tied = 0;
  for (int64_t base = (int64_t)((ntiles - 1) / 1024u) * 1024; base >= 0; base -= 1024) {
    const uint64_t i = (uint64_t)base + t;
    const uint32_t v = i < ntiles ? first_head[i] : kDblNone;
    tied += i < ntiles ? tied_rows[i] : 0u;
    uint32_t s = v;  // smallest of lanes lane.. of the wave
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_down(s, off, 64);
      if (lane + off < 64 && u < s) s = u;
    }
    if (lane == 0) wmin[wave] = s;
    __syncthreads();
    uint32_t after = carry;
    for (uint32_t w = wave + 1; w < 16; ++w) after = wmin[w] < after ? wmin[w] : after;
    uint32_t ex = __shfl_down(s, 1, 64);
    if (lane == 63) ex = kDblNone;
    if (i < ntiles) nxt[i] = ex < after ? ex : after;
    uint32_t all = carry;
    for (uint32_t w = 0; w < 16; ++w) all = wmin[w] < all ? wmin[w] : all;
    carry = all;
    __syncthreads();
  }
  for (int off = 32; off; off >>= 1) tied += __shfl_down(tied, off, 64);
  if (lane == 0) wsum[wave] = tied;
  __syncthreads();
  if (t == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < 16; ++w) total += wsum[w];
    ctr->active = total;
  }
}

// groups of more than kDblTile rows: only the last head of a tile can start one
__global__ void k_dbl_emit_big(const uint32_t *__restrict__ last_head, const uint32_t *__restrict__ nxt,
                               uint32_t ntiles, Seg *__restrict__ segs, uint32_t cap, DblCtr *__restrict__ ctr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntiles) return;
  const uint32_t j = last_head[i];
  if (j == kDblNone) return;
  const uint32_t size = nxt[i] - j;
  if (size <= kDblTile) return;
  const uint32_t k = atomicAdd(&ctr->nbig, 1u);
  if (k < cap) segs[k] = Seg{j, size};
}

__global__ void k_dbl_keys(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ ghead,
                           const uint32_t *__restrict__ isa, uint32_t n, uint64_t h, uint64_t rows,
                           uint32_t *__restrict__ key) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= rows) return;
  const bool tied = ghead[j] != (uint32_t)j || (j + 1 < rows && ghead[j + 1] == (uint32_t)j);
  if (!tied) return;
  // a suffix in a tie group shares at least h bases with another, so start + h <= n;
  // rank 0 is the end-of-text suffix and orders a suffix that ends exactly there first
  const uint64_t nx = (uint64_t)sa[j] + h;
  key[j] = nx <= n ? isa[nx] : 0u;
}

// --- groups of more than kDblTile rows: one 8-bit split per level ---

__global__ void k_dbl_seg_tiles(const Seg *__restrict__ segs, uint32_t nsegs, uint32_t *__restrict__ tiles,
                                uint32_t *__restrict__ mm) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsegs) return;
  tiles[s] = (segs[s].size + kDblPart - 1) / kDblPart;
  mm[2 * s] = 0xFFFFFFFFu;
  mm[2 * s + 1] = 0u;
}

// a partition tile is up to kDblPart rows of one group
__device__ __forceinline__ bool dbl_locate_tile(uint32_t tile, const Seg *segs, const uint32_t *tile_first,
                                                uint32_t nsegs, const DblCtr *ctr, TileInfo *sh) {
  if (threadIdx.x == 0) {
    if (tile >= ctr->total_tiles) {
      sh->count = 0;
    } else {
      const uint32_t s = find_segment(tile_first, nsegs, tile);
      const Seg sg = segs[s];
      const uint32_t off = (tile - tile_first[s]) * kDblPart;
      sh->seg = s;
      sh->row0 = sg.start + off;
      const uint32_t left = sg.size - off;
      sh->count = left < kDblPart ? left : kDblPart;
    }
  }
  __syncthreads();
  return sh->count != 0;
}

// f(value, valid) for every row of a[row0 .. row0 + count), in no particular order, every thread of the
// workgroup calling f the same number of times (f may be wave-wide): the rows are read four to a load where
// they are 16-byte aligned, the up to three rows in front and behind one per thread.
template <typename F>
__device__ __forceinline__ void dbl_each_row(const uint32_t *__restrict__ a, uint32_t row0, uint32_t count, F f) {
  const uint32_t t = threadIdx.x;
  uint32_t head = (4u - (row0 & 3u)) & 3u;
  head = head < count ? head : count;
  if (head) f(t < head ? a[row0 + t] : 0u, t < head);
  const uint4 *av = reinterpret_cast<const uint4 *>(a + row0 + head);
  const uint32_t nv = (count - head) / 4u;
  for (uint32_t i0 = 0; i0 < nv; i0 += kDblPartThreads) {
    const bool valid = i0 + t < nv;
    const uint4 q = valid ? av[i0 + t] : make_uint4(0, 0, 0, 0);
    f(q.x, valid);
    f(q.y, valid);
    f(q.z, valid);
    f(q.w, valid);
  }
  const uint32_t done = head + 4u * nv;
  if (done < count) f(done + t < count ? a[row0 + done + t] : 0u, done + t < count);
}

__global__ void __launch_bounds__(kDblPartThreads)
k_dbl_minmax(const Seg *__restrict__ segs, const uint32_t *__restrict__ tile_first, uint32_t nsegs,
             const DblCtr *__restrict__ ctr, const uint32_t *__restrict__ key, uint32_t *__restrict__ mm) {
  __shared__ TileInfo ti;
  __shared__ uint32_t lo, hi;
  if (threadIdx.x == 0) {
    lo = 0xFFFFFFFFu;
    hi = 0;
  }
  if (!dbl_locate_tile(blockIdx.x, segs, tile_first, nsegs, ctr, &ti)) return;
  uint32_t a = 0xFFFFFFFFu, b = 0;
  dbl_each_row(key, ti.row0, ti.count, [&](uint32_t v, bool valid) {
    if (valid) {
      a = v < a ? v : a;
      b = v > b ? v : b;
    }
  });
  for (int off = 32; off; off >>= 1) {
    const uint32_t x = __shfl_xor(a, off, 64), y = __shfl_xor(b, off, 64);
    a = x < a ? x : a;
    b = y > b ? y : b;
  }
  if ((threadIdx.x & 63u) == 0) {
    atomicMin(&lo, a);
    atomicMax(&hi, b);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&mm[2 * ti.seg], lo);
    atomicMax(&mm[2 * ti.seg + 1], hi);
  }
}

// the digit of a key inside a group whose keys span [lo, hi], lo < hi: monotone in
// the key, and lo and hi never share a digit, so every level splits its group
__device__ __forceinline__ uint32_t dbl_digit(uint32_t key, uint32_t lo, float scale) {
  const uint32_t d = (uint32_t)((float)(key - lo) * scale);
  return d < 255u ? d : 255u;
}
__device__ __forceinline__ float dbl_scale(uint32_t lo, uint32_t hi) {
  return 256.0f / ((float)(hi - lo) + 1.0f);
}

// Adds one to counter[d] for every lane with `valid`, one LDS atomic per distinct digit
// of the wave: when nearly all keys of a group are equal (what a repeat looks like) the
// lanes of a wave would otherwise queue on one address.  Returns the lane's place among
// the increments of its digit.  Every lane of the wave calls it.
__device__ __forceinline__ uint32_t dbl_count(uint32_t *counter, uint32_t d, bool valid, uint32_t lane) {
  uint64_t same = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (d >> b) & 1u;
    const uint64_t bal = __ballot(bit);
    same &= bit ? bal : ~bal;
  }
  const uint32_t before = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
  uint32_t first = 0;
  if (valid && before == 0) first = atomicAdd(&counter[d], (uint32_t)__popcll(same));
  const int leader = valid ? __ffsll((long long)same) - 1 : (int)lane;
  first = __shfl(first, leader, 64);
  return first + before;
}

__global__ void __launch_bounds__(kDblPartThreads)
k_dbl_hist(const Seg *__restrict__ segs, const uint32_t *__restrict__ tile_first, uint32_t nsegs,
           const DblCtr *__restrict__ ctr, const uint32_t *__restrict__ key, const uint32_t *__restrict__ mm,
           uint32_t *__restrict__ counts) {
  __shared__ TileInfo ti;
  __shared__ uint32_t hist[256];
  if (threadIdx.x < 256) hist[threadIdx.x] = 0;
  if (!dbl_locate_tile(blockIdx.x, segs, tile_first, nsegs, ctr, &ti)) return;
  const uint32_t lo = mm[2 * ti.seg], hi = mm[2 * ti.seg + 1];
  if (lo >= hi) return;  // one key: the group stays as it is this round
  const float scale = dbl_scale(lo, hi);
  dbl_each_row(key, ti.row0, ti.count, [&](uint32_t v, bool valid) {
    dbl_count(hist, valid ? dbl_digit(v, lo, scale) : 0u, valid, threadIdx.x & 63u);
  });
  __syncthreads();
  if (threadIdx.x < 256) {
    const uint32_t c = hist[threadIdx.x];
    if (c) atomicAdd(&counts[(uint64_t)ti.seg * 256 + threadIdx.x], c);
  }
}

// counts -> first rows of the 256 pieces (cursors for the scatter, and a copy that
// stays) and their labels; pieces of more than kDblTile rows go on to the next level
__global__ void __launch_bounds__(256)
k_dbl_classify(const Seg *__restrict__ segs, uint32_t nsegs, const uint32_t *__restrict__ mm,
               uint32_t *__restrict__ counts, uint32_t *__restrict__ piece_start, uint32_t *__restrict__ piece_label,
               uint32_t *__restrict__ glabel, Seg *__restrict__ next_segs, uint32_t cap, DblCtr *__restrict__ ctr) {
  __shared__ uint32_t wsum[4];
  const uint32_t s = blockIdx.x, t = threadIdx.x;
  if (mm[2 * s] >= mm[2 * s + 1]) return;
  const Seg sg = segs[s];
  const uint32_t label = glabel[sg.start];  // every thread reads it before the barrier, the pieces' are written after
  const uint32_t c = counts[(uint64_t)s * 256 + t];
  uint32_t v = c;
  const uint32_t lane = t & 63u, wave = t >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  uint32_t pre = 0;
  for (uint32_t w = 0; w < wave; ++w) pre += wsum[w];
  const uint32_t start = sg.start + pre + v - c;
  counts[(uint64_t)s * 256 + t] = start;
  piece_start[(uint64_t)s * 256 + t] = start;
  // the piece that holds the row the group's label names keeps the label; another takes its
  // middle row, so that it keeps that label as long as it loses less than half its rows
  // from either end (a repeat sheds a few suffixes per round, mostly from the front)
  const uint32_t mine = (label >= start && label - start < c) ? label : start + c / 2;
  piece_label[(uint64_t)s * 256 + t] = mine;
  if (c) glabel[start] = mine;
  if (c > kDblTile) {
    const uint32_t k = atomicAdd(&ctr->nnext, 1u);
    if (k < cap) next_segs[k] = Seg{start, c};
  }
}

// A workgroup takes a tile chunk by chunk: first the tile's digit counts and one
// claim of room per digit for all of it (with one group making up most of the text,
// a claim per chunk would queue tens of thousands of workgroups on one counter),
// then each chunk goes into bucket order in LDS and out.
__global__ void __launch_bounds__(kDblPartThreads)
k_dbl_scatter(const Seg *__restrict__ segs, const uint32_t *__restrict__ tile_first, uint32_t nsegs,
              const DblCtr *__restrict__ ctr, const uint32_t *__restrict__ sa, const uint32_t *__restrict__ key,
              const uint32_t *__restrict__ mm, uint32_t *__restrict__ cursors,
              const uint32_t *__restrict__ piece_start, const uint32_t *__restrict__ piece_label,
              const uint32_t *__restrict__ glabel_old /* label of the group, read before k_dbl_classify ran */,
              uint32_t *__restrict__ dst_sa, uint32_t *__restrict__ dst_key, uint32_t *__restrict__ ghead,
              uint32_t *__restrict__ isa) {
  __shared__ TileInfo ti;
  __shared__ uint32_t hist[256], lstart[256], gbase[256], pstart[256], plabel[256], wsum[4];
  __shared__ uint32_t ids[kDblChunk], kv[kDblChunk];
  __shared__ uint8_t dig[kDblChunk];
  const uint32_t t = threadIdx.x;
  const bool bin = t < 256;  // the thread that looks after digit t
  if (bin) hist[t] = 0;
  if (!dbl_locate_tile(blockIdx.x, segs, tile_first, nsegs, ctr, &ti)) return;
  const uint32_t lo = mm[2 * ti.seg], hi = mm[2 * ti.seg + 1];
  if (lo >= hi) return;
  const float scale = dbl_scale(lo, hi);
  const uint32_t row0 = ti.row0, count = ti.count;
  const uint32_t old_label = glabel_old[ti.seg];
  const uint32_t lane = t & 63u, wave = t >> 6;
  dbl_each_row(key, row0, count, [&](uint32_t v, bool valid) {
    dbl_count(hist, valid ? dbl_digit(v, lo, scale) : 0u, valid, lane);
  });
  __syncthreads();
  if (bin) {
    const uint32_t c = hist[t];
    gbase[t] = c ? atomicAdd(&cursors[(uint64_t)ti.seg * 256 + t], c) : 0u;
    pstart[t] = piece_start[(uint64_t)ti.seg * 256 + t];
    plabel[t] = piece_label[(uint64_t)ti.seg * 256 + t];
  }
  for (uint32_t c0 = 0; c0 < count; c0 += kDblChunk) {
    const uint32_t cc = count - c0 < kDblChunk ? count - c0 : kDblChunk;
    __syncthreads();  // the chunk before has been written out; the claims have landed
    if (bin) hist[t] = 0;
    __syncthreads();
    uint32_t id[kDblItems], kk[kDblItems], dd[kDblItems], place[kDblItems];
#pragma unroll
    for (uint32_t i = 0; i < kDblItems; ++i) {
      const uint32_t j = i * kDblPartThreads + t;
      const bool valid = j < cc;
      id[i] = valid ? sa[row0 + c0 + j] : 0u;
      kk[i] = valid ? key[row0 + c0 + j] : 0u;
      dd[i] = valid ? dbl_digit(kk[i], lo, scale) : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < kDblItems; ++i)
      place[i] = dbl_count(hist, dd[i], i * kDblPartThreads + t < cc, lane);  // its place among the chunk's rows of that digit
    __syncthreads();
    const uint32_t c = bin ? hist[t] : 0u;
    uint32_t v = c;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    if (bin && lane == 63) wsum[wave] = v;
    __syncthreads();
    if (bin) {
      uint32_t pre = 0;
      for (uint32_t w = 0; w < wave; ++w) pre += wsum[w];
      lstart[t] = pre + v - c;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kDblItems; ++i) {
      if (i * kDblPartThreads + t < cc) {
        const uint32_t p = lstart[dd[i]] + place[i];
        ids[p] = id[i];
        kv[p] = kk[i];
        dig[p] = (uint8_t)dd[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kDblItems; ++i) {
      const uint32_t j = i * kDblPartThreads + t;
      if (j < cc) {
        const uint32_t d = dig[j];
        const uint32_t pos = gbase[d] + (j - lstart[d]);
        dst_sa[pos] = ids[j];
        dst_key[pos] = kv[j];
        ghead[pos] = pstart[d];
        if (plabel[d] != old_label) isa[ids[j]] = plabel[d];
      }
    }
    __syncthreads();
    if (bin) gbase[t] += c;  // the chunk's share of the claim is used up
  }
}

__global__ void k_dbl_old_labels(const Seg *__restrict__ segs, uint32_t nsegs, const uint32_t *__restrict__ glabel,
                                 uint32_t *__restrict__ out) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nsegs) out[s] = glabel[segs[s].start];
}

__global__ void __launch_bounds__(kDblPartThreads)
k_dbl_copyback(const Seg *__restrict__ segs, const uint32_t *__restrict__ tile_first, uint32_t nsegs,
               const DblCtr *__restrict__ ctr, const uint32_t *__restrict__ mm, const uint32_t *__restrict__ src_sa,
               const uint32_t *__restrict__ src_key, uint32_t *__restrict__ sa, uint32_t *__restrict__ key) {
  __shared__ TileInfo ti;
  if (!dbl_locate_tile(blockIdx.x, segs, tile_first, nsegs, ctr, &ti)) return;
  if (mm[2 * ti.seg] >= mm[2 * ti.seg + 1]) return;
  // 16 bytes a lane over the aligned middle of the rows, the few rows in front and behind one per thread
  const uint32_t t = threadIdx.x, row0 = ti.row0, count = ti.count;
  uint32_t head = (4u - (row0 & 3u)) & 3u;
  head = head < count ? head : count;
  const uint32_t nv = (count - head) / 4u, done = head + 4u * nv;
  const uint32_t odd = t < head ? t : done + (t - head);
  if (odd < count && (t < head || odd >= done)) {
    sa[row0 + odd] = src_sa[row0 + odd];
    key[row0 + odd] = src_key[row0 + odd];
  }
  const uint4 *ssa = reinterpret_cast<const uint4 *>(src_sa + row0 + head);
  const uint4 *skey = reinterpret_cast<const uint4 *>(src_key + row0 + head);
  uint4 *dsa = reinterpret_cast<uint4 *>(sa + row0 + head), *dkey = reinterpret_cast<uint4 *>(key + row0 + head);
  for (uint32_t i = t; i < nv; i += kDblPartThreads) {
    dsa[i] = ssa[i];
    dkey[i] = skey[i];
  }
}

// --- groups of up to kDblTile rows: every tile's workgroup sorts the groups that
// start in its tile (they end within the next tile) by (group, key) in LDS ---

template <int LOGE>
__device__ __forceinline__ void dbl_sort_words(const uint32_t *__restrict__ ghl, const uint32_t *__restrict__ key,
                                               uint32_t base, uint32_t cnt, uint64_t *lds, uint32_t t) {
  constexpr int E = 1 << LOGE;
  uint64_t r[E];
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const uint32_t j = t * E + m;
    r[m] = ~0ull;
    if (j < cnt) {
      const uint32_t g = ghl[j];
      const bool tied = g != base + j || (j + 1 < cnt && ghl[j + 1] == base + j);
      const uint32_t k = tied ? key[base + j] : 0u;
      r[m] = ((uint64_t)(g - base) << 52) | ((uint64_t)k << 20) | j;
    }
  }
  leaf_bitonic<LOGE>(r, lds, t);
#pragma unroll
  for (int m = 0; m < E; ++m) lds[pad(t * E + m)] = r[m];
}

__global__ void __launch_bounds__(kLeafThreads, 2)
k_dbl_small(const uint32_t *__restrict__ first_head, const uint32_t *__restrict__ last_head,
            const uint32_t *__restrict__ nxt, uint32_t *__restrict__ sa, const uint32_t *__restrict__ key,
            uint32_t *__restrict__ ghead, uint32_t *__restrict__ isa, uint32_t *__restrict__ glabel) {
  constexpr uint32_t kWin = 2 * kDblTile;
  static_assert(kWin == (uint32_t)kLeafThreads * 8, "eight rows per thread");
  __shared__ uint64_t words[kWin + kWin / 32 + 1];
  __shared__ uint32_t ghl[kWin];  // heads of the window's rows, then the starts of its suffixes
  __shared__ uint32_t wmax[kLeafWaves];
  __shared__ uint32_t anytied;
  const uint32_t t = threadIdx.x;
  const uint32_t base = first_head[blockIdx.x];
  if (base == kDblNone) return;
  const uint32_t lastg = last_head[blockIdx.x], e = nxt[blockIdx.x];
  const uint32_t end = e - lastg <= kDblTile ? e : lastg;  // a longer last group is not this kernel's
  const uint32_t cnt = end - base;                        // < kWin
  if (cnt == 0) return;
  if (t == 0) anytied = 0;
  __syncthreads();
  bool tied = false;
  for (uint32_t j = t; j < cnt; j += kLeafThreads) {
    const uint32_t g = ghead[base + j];
    ghl[j] = g;
    tied = tied || g != base + j;
  }
  if (tied) anytied = 1;
  __syncthreads();
  if (!anytied) return;  // every row of the range is settled
  uint32_t P = 2 * kLeafThreads;
  while (P < cnt) P <<= 1;
  if (P == 8 * kLeafThreads) dbl_sort_words<3>(ghl, key, base, cnt, words, t);
  else if (P == 4 * kLeafThreads) dbl_sort_words<2>(ghl, key, base, cnt, words, t);
  else dbl_sort_words<1>(ghl, key, base, cnt, words, t);
  __syncthreads();
  // the suffix starts take the place of the heads (which the words now hold)
  for (uint32_t j = t; j < cnt; j += kLeafThreads) ghl[j] = sa[base + j];
  __syncthreads();
  // blocked: thread t owns sorted places [t * per, t * per + per)
  const uint32_t per = P / kLeafThreads;
  const uint32_t j0 = t * per;
  uint64_t w[9];
  uint32_t hv[8], local_max = 0;
#pragma unroll
  for (uint32_t k = 0; k < 9; ++k) {
    const uint32_t j = j0 + k - 1u;
    w[k] = (k <= per && j < cnt) ? words[pad(j)] : ~0ull;
  }
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    const uint32_t j = j0 + k;
    hv[k] = 0;
    if (k < per && j < cnt) {
      const bool same = j > 0 && (w[k] >> 20) == (w[k + 1] >> 20);  // same group, same key as the row before
      hv[k] = same ? 0u : base + j;
      local_max = hv[k] > local_max ? hv[k] : local_max;
    }
  }
  const uint32_t lane = t & 63u, wave = t >> 6;
  uint32_t v = local_max;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(v, off, 64);
    if (lane >= off) v = u > v ? u : v;
  }
  if (lane == 63) wmax[wave] = v;
  __syncthreads();  // also: every thread has read its window of words
  uint32_t run = 0;
  for (uint32_t q = 0; q < wave; ++q) run = wmax[q] > run ? wmax[q] : run;
  const uint32_t up = __shfl_up(v, 1, 64);
  if (lane) run = up > run ? up : run;
  // the new head of every place, where its neighbours can see it (in the words' room),
  // and the label of the group each row came from
  uint32_t *newhead = reinterpret_cast<uint32_t *>(words);
  uint32_t nh[8], lab[8];
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    const uint32_t j = j0 + k;
    nh[k] = lab[k] = 0;
    if (k < per && j < cnt) {
      run = hv[k] > run ? hv[k] : run;
      nh[k] = run;
      newhead[j] = run;
      lab[k] = glabel[base + (uint32_t)(w[k + 1] >> 52)];
    }
  }
  __syncthreads();  // every old label has been read before a new one is written
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    const uint32_t j = j0 + k;
    if (k < per && j < cnt) {
      const uint64_t cur = w[k + 1];
      const uint32_t id = ghl[(uint32_t)cur & 0xFFFFFu];
      const uint32_t old = base + (uint32_t)(cur >> 52);
      // the piece that holds the row the old group's label names keeps the label
      const uint32_t mine = newhead[lab[k] - base] == nh[k] ? lab[k] : nh[k];
      sa[base + j] = id;
      if (nh[k] != old) ghead[base + j] = nh[k];
      if (nh[k] == base + j) glabel[base + j] = mine;
      if (mine != lab[k]) isa[id] = mine;
    }
  }
}

static int sa_doubling_groups(uint32_t *d_sa, uint32_t *ghead, uint32_t n, uint32_t h0, hipStream_t s);

// BWT_MI355X_DOUBLING=lsd: the rounds as one 64-bit radix sort each (sa_doubling), kept for comparison
static int sa_finish_ties(uint32_t *d_sa, uint32_t *ghead, uint32_t n, uint32_t h0, uint32_t *flags,
                          Counters *d_ctr, hipStream_t s) {
  const char *e = getenv("BWT_MI355X_DOUBLING");
  if (e && !strcmp(e, "lsd")) return sa_doubling(d_sa, ghead, n, h0, flags, d_ctr, s);
  return sa_doubling_groups(d_sa, ghead, n, h0, s);
}

static int sa_doubling_groups(uint32_t *d_sa, uint32_t *ghead, uint32_t n, uint32_t h0, hipStream_t s) {
  const uint64_t rows = (uint64_t)n + 1;
  const uint32_t ntiles = (uint32_t)div_up(rows, kDblTile);
  const uint32_t cap = (uint32_t)(rows / kDblTile) + 16;  // groups of more than kDblTile rows
  DevBuf isa, key, alt_sa, alt_key, glabel, heads, segs_a, segs_b, mm, counts, pieces, plabels, oldl, tiles, scanws, dctr;
  hipError_t e = isa.alloc(rows * 4);
  if (e == hipSuccess) e = key.alloc(rows * 4);
  if (e == hipSuccess) e = alt_sa.alloc(rows * 4);
  if (e == hipSuccess) e = alt_key.alloc(rows * 4);
  if (e == hipSuccess) e = glabel.alloc(rows * 4);
  if (e == hipSuccess) e = heads.alloc((size_t)ntiles * 4 * 4);
  if (e == hipSuccess) e = segs_a.alloc((size_t)cap * sizeof(Seg));
  if (e == hipSuccess) e = segs_b.alloc((size_t)cap * sizeof(Seg));
  if (e == hipSuccess) e = mm.alloc((size_t)cap * 8);
  if (e == hipSuccess) e = counts.alloc((size_t)cap * 256 * 4);
  if (e == hipSuccess) e = pieces.alloc((size_t)cap * 256 * 4);
  if (e == hipSuccess) e = plabels.alloc((size_t)cap * 256 * 4);
  if (e == hipSuccess) e = oldl.alloc((size_t)cap * 4);
  if (e == hipSuccess) e = tiles.alloc(((size_t)cap + 1) * 4);
  if (e == hipSuccess) e = scanws.alloc(scan_workspace_bytes((uint64_t)cap + 1));
  if (e == hipSuccess) e = dctr.alloc(256);
  if (e != hipSuccess) return set_error(kErrHip, "prefix doubling buffers: %s", hipGetErrorString(e));
  uint32_t *first_head = heads.as<uint32_t>(), *last_head = first_head + ntiles, *nxt = last_head + ntiles,
           *tied_rows = nxt + ntiles;
  DblCtr *ctr = dctr.as<DblCtr>();
  const unsigned gr = (unsigned)div_up(rows, 256);
  BWT_LAUNCH("dbl_init", k_dbl_init, dim3(gr), dim3(256), 0, s, d_sa, ghead, isa.as<uint32_t>(), glabel.as<uint32_t>(),
             rows);
  auto survey = [&]() -> int {  // heads per tile, then the first head after every tile
    BWT_LAUNCH("dbl_tiles", k_dbl_tiles, dim3(ntiles), dim3(256), 0, s, ghead, rows, first_head, last_head, tied_rows);
    BWT_LAUNCH("dbl_next_head", k_dbl_next_head, dim3(1), dim3(1024), 0, s, first_head, tied_rows, ntiles,
               (uint32_t)rows, nxt, ctr);
    return kOk;
  };
  uint64_t h = h0 ? h0 : 1;
  uint32_t prev_active = 0xFFFFFFFFu;
  for (int round = 0; round < 64; ++round) {
    BWT_HIP(hipMemsetAsync(ctr, 0, sizeof(DblCtr), s));
    BWT_TRY(survey());
    BWT_LAUNCH("dbl_emit_big", k_dbl_emit_big, dim3((unsigned)div_up(ntiles, 256)), dim3(256), 0, s, last_head, nxt,
               ntiles, segs_a.as<Seg>(), cap, ctr);
    DblCtr hc;
    BWT_HIP(hipMemcpyAsync(&hc, ctr, sizeof hc, hipMemcpyDeviceToHost, s));
    BWT_HIP(hipStreamSynchronize(s));
    if (hc.active == 0) {
      g_stats.v[3] = round;
      return kOk;
    }
    if (hc.active > prev_active)
      return set_error(kErrInternal, "prefix doubling: tied rows grew %u -> %u", prev_active, hc.active);
    prev_active = hc.active;
    if (hc.nbig > cap) return set_error(kErrInternal, "prefix doubling: group list overflow");
    const uint32_t nbig = hc.nbig;
    BWT_LAUNCH("dbl_keys", k_dbl_keys, dim3(gr), dim3(256), 0, s, d_sa, ghead, isa.as<uint32_t>(), n, h, rows,
               key.as<uint32_t>());
    // groups of more than kDblTile rows, level by level
    uint32_t nsegs = nbig;
    Seg *cur = segs_a.as<Seg>(), *next = segs_b.as<Seg>();
    for (int level = 0; nsegs > 0; ++level) {
      if (level > 8) return set_error(kErrInternal, "prefix doubling: a group did not split in %d levels", level);
      BWT_LAUNCH("dbl_seg_tiles", k_dbl_seg_tiles, dim3((unsigned)div_up(nsegs, 256)), dim3(256), 0, s, cur, nsegs,
                 tiles.as<uint32_t>(), mm.as<uint32_t>());
      BWT_TRY(device_scan_u32(ScanOp::kAddExclusive, tiles.as<uint32_t>(), tiles.as<uint32_t>(), nsegs,
                              &ctr->total_tiles, scanws.p, scan_workspace_bytes((uint64_t)cap + 1), s));
      BWT_HIP(hipMemsetAsync(&ctr->nnext, 0, 4, s));
      BWT_HIP(hipMemsetAsync(counts.p, 0, (size_t)nsegs * 256 * 4, s));
      const unsigned tile_bound = (unsigned)(rows / kDblPart + nsegs + 1);
      BWT_LAUNCH("dbl_minmax", k_dbl_minmax, dim3(tile_bound), dim3(kDblPartThreads), 0, s, cur, tiles.as<uint32_t>(),
                 nsegs, ctr, key.as<uint32_t>(), mm.as<uint32_t>());
      BWT_LAUNCH("dbl_hist", k_dbl_hist, dim3(tile_bound), dim3(kDblPartThreads), 0, s, cur, tiles.as<uint32_t>(),
                 nsegs, ctr, key.as<uint32_t>(), mm.as<uint32_t>(), counts.as<uint32_t>());
      BWT_LAUNCH("dbl_old_labels", k_dbl_old_labels, dim3((unsigned)div_up(nsegs, 256)), dim3(256), 0, s, cur, nsegs,
                 glabel.as<uint32_t>(), oldl.as<uint32_t>());
      BWT_LAUNCH("dbl_classify", k_dbl_classify, dim3(nsegs), dim3(256), 0, s, cur, nsegs, mm.as<uint32_t>(),
                 counts.as<uint32_t>(), pieces.as<uint32_t>(), plabels.as<uint32_t>(), glabel.as<uint32_t>(), next, cap,
                 ctr);
      BWT_LAUNCH("dbl_scatter", k_dbl_scatter, dim3(tile_bound), dim3(kDblPartThreads), 0, s, cur,
                 tiles.as<uint32_t>(), nsegs, ctr, d_sa, key.as<uint32_t>(), mm.as<uint32_t>(), counts.as<uint32_t>(),
                 pieces.as<uint32_t>(), plabels.as<uint32_t>(), oldl.as<uint32_t>(), alt_sa.as<uint32_t>(),
                 alt_key.as<uint32_t>(), ghead, isa.as<uint32_t>());
      BWT_LAUNCH("dbl_copyback", k_dbl_copyback, dim3(tile_bound), dim3(kDblPartThreads), 0, s, cur,
                 tiles.as<uint32_t>(), nsegs, ctr, mm.as<uint32_t>(), alt_sa.as<uint32_t>(), alt_key.as<uint32_t>(), d_sa,
                 key.as<uint32_t>());
      BWT_HIP(hipMemcpyAsync(&hc, ctr, sizeof hc, hipMemcpyDeviceToHost, s));
      BWT_HIP(hipStreamSynchronize(s));
      if (hc.nnext > cap) return set_error(kErrInternal, "prefix doubling: piece list overflow");
      nsegs = hc.nnext;
      Seg *tmp = cur;
      cur = next;
      next = tmp;
    }
    if (nbig) BWT_TRY(survey());  // the pieces are groups now: the tiles' heads have changed
    BWT_LAUNCH("dbl_small", k_dbl_small, dim3(ntiles), dim3(kLeafThreads), 0, s, first_head, last_head, nxt, d_sa,
               key.as<uint32_t>(), ghead, isa.as<uint32_t>(), glabel.as<uint32_t>());
    BWT_CHECK_LAUNCH();
    h *= 2;
  }
  return set_error(kErrInternal, "prefix doubling did not converge");
}

// ---- driver -------------------------------------------------------------------

// A second stream per building thread and device, and the event `s` waits on for it.
// Not freed: a thread may end after the runtime has shut down.
struct LeafSide {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
};
static thread_local LeafSide g_leaf_side;

static hipStream_t leaf_side_stream(hipStream_t s) {
  static const bool off = [] {
    const char *e = getenv("BWT_MI355X_LEAF_STREAMS");
    return e && e[0] == '0';
  }();
  if (off || timing().enabled) return s;  // timed launch by launch: one after the other
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return s;
  LeafSide &g = g_leaf_side;
  if (g.device != dev) {
    g = LeafSide{};
    if (hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g.done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      g = LeafSide{};
      return s;
    }
    g.device = dev;
  }
  return g.stream;
}

static int leaf_side_join(hipStream_t s) {
  BWT_HIP(hipEventRecord(g_leaf_side.done, g_leaf_side.stream));
  BWT_HIP(hipStreamWaitEvent(s, g_leaf_side.done, 0));
  return kOk;
}

struct Layout {
  size_t tb, bufB, keyA, keyB, bw, ghead, segs0, segs1, counts, leafmask, tile_first, leaves, leaf_ties, bigleaves, midleaves, deferred, ctr, scanws, range, end;
  uint64_t nwords;
  uint32_t seg_cap, leaf_cap;
};

static Layout layout_for(uint32_t n) {
  Layout L;
  L.nwords = (uint64_t)n / 16 + 1 + kTextPadWords;
  L.seg_cap = n / kLeafMax + 512;       // a next-level segment holds > kLeafMax rows
  L.leaf_cap = n / 128 + 8192;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  L.tb = take(L.nwords * 4);
  L.bufB = take(((size_t)n + 1) * 4);
  L.keyA = take(((size_t)n + 1) * 4);
  L.keyB = take(((size_t)n + 1) * 4);
  L.bw = take((size_t)n + 1);
  L.ghead = take(((size_t)n + 1) * 4);
  L.segs0 = take((size_t)L.seg_cap * sizeof(Seg));
  L.segs1 = take((size_t)L.seg_cap * sizeof(Seg));
  L.counts = take((size_t)L.seg_cap * 256 * 4);
  L.leafmask = take((size_t)L.seg_cap * 8 * 4);
  L.tile_first = take(((size_t)L.seg_cap + 1) * 4);
  L.leaves = take((size_t)L.leaf_cap * sizeof(Leaf));
  L.leaf_ties = take((size_t)L.leaf_cap);
  L.bigleaves = take((size_t)L.leaf_cap * 4);
  L.midleaves = take((size_t)L.leaf_cap * 4);
  L.deferred = take((size_t)L.leaf_cap * 4);
  L.ctr = take(sizeof(Counters));
  L.scanws = take(scan_workspace_bytes(L.seg_cap + 1));
  L.range = take(kRangeBytes + 16);
  L.end = off;
  return L;
}

void *sa_workspace_free_region(uint32_t n, void *d_ws) {
  const Layout L = layout_for(n);
  // bufB, keyA, keyB follow each other: 12 (n+1) bytes and some alignment
  static_assert(sizeof(uint32_t) * 3 >= 8 + 0, "three words per row hold one pair");
  if (L.keyB + ((size_t)n + 1) * 4 - L.bufB < ((size_t)n + 1) * 8 + 2048) return nullptr;
  return reinterpret_cast<uint8_t *>(d_ws) + L.bufB;
}

int sa_build_impl(const uint8_t *d_text, uint32_t n, uint32_t *d_sa, void *d_ws, size_t ws_bytes,
                  hipStream_t s, SaExtras *extras) {
  return sa_build_core(d_text, n, d_sa, nullptr, d_ws, ws_bytes, s, extras);
}

// range == nullptr: the whole suffix array into d_sa (n+1 rows).
// Otherwise only the suffixes whose first four bases, as a number, lie in
// [range->dlo, range->dhi): their rows, in order, go to d_sa[0 ..) and the
// range's place among all rows is reported in *range.
int sa_build_core(const uint8_t *d_text, uint32_t n, uint32_t *d_sa, SaRange *range, void *d_ws,
                  size_t ws_bytes, hipStream_t s, SaExtras *extras) {
  uint32_t *samples = (extras && !range) ? extras->samples : nullptr;
  if (extras) {
    extras->bw = nullptr;
    extras->endloc = nullptr;
    extras->samples_done = false;
    extras->text2_done = false;
    extras->pending = false;
  }
  if (!d_text || !d_sa || !d_ws) return set_error(kErrArg, "null argument");
  if (range && (range->dlo >= range->dhi || range->dhi > 256))
    return set_error(kErrArg, "bucket range [%u, %u) is not inside [0, 256)", range->dlo, range->dhi);
  if (range && n <= kLeafMax)
    return set_error(kErrNeedsWholeText, "texts of up to %u bases are built whole", kLeafMax);
  const uint32_t dlo = range ? range->dlo : 0u, dhi = range ? range->dhi : 256u;
  if (n == 0 || n > 0xFFFFFFF0u) return set_error(kErrArg, "text length %u out of range", n);
  if ((uintptr_t)d_ws & 255u) return set_error(kErrArg, "workspace must be 256-byte aligned");
  const Layout L = layout_for(n);
  if (ws_bytes < L.end) return set_error(kErrWorkspace, "SA workspace %zu < %zu bytes", ws_bytes, L.end);
  uint8_t *ws = reinterpret_cast<uint8_t *>(d_ws);
  uint32_t *tb = reinterpret_cast<uint32_t *>(ws + L.tb);
  uint32_t *bufA = range ? d_sa : d_sa + 1;
  uint32_t *bufB = reinterpret_cast<uint32_t *>(ws + L.bufB);
  uint32_t *keyA = reinterpret_cast<uint32_t *>(ws + L.keyA);
  uint32_t *keyB = reinterpret_cast<uint32_t *>(ws + L.keyB);
  uint8_t *bw = ws + L.bw;
  uint32_t *ghead = reinterpret_cast<uint32_t *>(ws + L.ghead);
  Seg *segs[2] = {reinterpret_cast<Seg *>(ws + L.segs0), reinterpret_cast<Seg *>(ws + L.segs1)};
  uint32_t *counts = reinterpret_cast<uint32_t *>(ws + L.counts);
  uint32_t *leafmask = reinterpret_cast<uint32_t *>(ws + L.leafmask);
  uint32_t *tile_first = reinterpret_cast<uint32_t *>(ws + L.tile_first);
  Leaf *leaves = reinterpret_cast<Leaf *>(ws + L.leaves);
  uint8_t *leaf_ties = ws + L.leaf_ties;
  uint32_t *bigleaves = reinterpret_cast<uint32_t *>(ws + L.bigleaves);
  uint32_t *midleaves = reinterpret_cast<uint32_t *>(ws + L.midleaves);
  uint32_t *deferred = reinterpret_cast<uint32_t *>(ws + L.deferred);
  Counters *ctr = reinterpret_cast<Counters *>(ws + L.ctr);
  void *scanws = ws + L.scanws;
  const size_t scanws_bytes = scan_workspace_bytes(L.seg_cap + 1);
  g_stats = SaStats{};

  // Level 0 counts into the words behind the 256 x 256 that level 1 can use, so both are zeroed here
  static_assert(kTextPadWords >= 3, "the text in words reaches as far as the little-endian copy");
  uint32_t *counts0 = counts + 256 * 256;
  BuildInit init = {};
  init.ctr = ctr;
  init.row0 = range ? nullptr : d_sa;
  init.sample0 = samples;  // SA[0] = n is row 0's sample
  init.ghead0 = ghead;
  init.seg0 = n > kLeafMax ? segs[0] : nullptr;
  init.tile_first = tile_first;
  init.zero = counts;
  init.zero_words = n > kLeafMax ? 256 * 256 + 8 * 256 : 0;
  init.text2 = (extras && !range) ? extras->text2 : nullptr;
  init.t2words = (uint64_t)n / 16 + 4;
  BWT_LAUNCH("sa_prepare_text", k_prepare_text, dim3((unsigned)div_up(L.nwords, 256)), dim3(256), 0, s,
             d_text, n, L.nwords, tb, init);
  if (init.text2) extras->text2_done = true;
  Counters h_ctr = {};
  h_ctr.min_depth = 0xFFFFFFFFu;

  uint32_t nsegs = 0, depth = 0, src_id = kSrcImplicit;
  int cur = 0;
  uint32_t nleaves_host = 0, nbig_host = 0, nmid_host = 0;
  if (n <= kLeafMax) {
    const Leaf whole = {0, n, 0, leaf_src_word(kSrcImplicit, kLeafKeyBases)};
    BWT_HIP(hipMemcpyAsync(leaves, &whole, sizeof whole, hipMemcpyHostToDevice, s));
    const uint32_t one = 1;
    BWT_HIP(hipMemcpyAsync(&ctr->nleaves, &one, 4, hipMemcpyHostToDevice, s));
    nleaves_host = 1;
    if (n > kWaveLeafMid) {
      BWT_HIP(hipMemsetAsync(bigleaves, 0, 4, s));  // leaf 0
      BWT_HIP(hipMemcpyAsync(&ctr->nbig, &one, 4, hipMemcpyHostToDevice, s));
      nbig_host = 1;
    } else if (n > kWaveLeafMax) {
      BWT_HIP(hipMemsetAsync(midleaves, 0, 4, s));
      BWT_HIP(hipMemcpyAsync(&ctr->nmid, &one, 4, hipMemcpyHostToDevice, s));
      nmid_host = 1;
    }
  } else {
    nsegs = 1;  // the whole text: k_prepare_text has listed it
  }
  // A text without long repeats is through after the first level whose buckets
  // hold kLeafMax suffixes on average.  Up to that level the host does not wait
  // to learn how many segments a level left: it sizes the next level's launches
  // for the most there can be (256 per segment, and no more than fit the rows)
  // and the kernels take the count from the device.  After it, every level is
  // read back as before.
  int expected_levels = 1;
  for (uint64_t per = (uint64_t)n >> 8; per > kLeafMax; per >>= 8) ++expected_levels;
  bool exact = true;  // nsegs is the count the device holds, not a bound on it
  uint32_t leaves_sorted = 0;  // leaves [0, leaves_sorted) have been handed to the leaf sort already
  struct EventHolder {
    hipEvent_t ev = nullptr;
    ~EventHolder() {
      if (ev) (void)hipEventDestroy(ev);
    }
  } ahead;
  if (extras && extras->readback && hipEventCreateWithFlags(&ahead.ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    ahead.ev = nullptr;
  }
  int levels = 0;
  uint32_t reach = 0;  // the words stored with the ids hold the text up to this base (exclusive)
  while (nsegs > 0 && depth < kMaxDepth) {
    const uint32_t dst_id = src_id == kSrcA ? kSrcB : kSrcA;
    const uint32_t *src = src_id == kSrcA ? bufA : src_id == kSrcB ? bufB : nullptr;
    uint32_t *dst = dst_id == kSrcA ? bufA : bufB;
    uint32_t *dstkey = dst_id == kSrcA ? keyA : keyB;
    // this level's segments and where the tiles of each begin
    const bool implicit = src_id == kSrcImplicit;  // level 0: one segment, counts per eighth of its tiles
    if (implicit) {
      // set by k_prepare_text
    } else if (nsegs <= 1024) {
      BWT_LAUNCH("sa_begin_level", k_begin_level_small, dim3(1), dim3(1024), 0, s, ctr, nsegs, segs[cur], tile_first);
    } else {
      BWT_LAUNCH("sa_begin_level", k_begin_level, dim3(1), dim3(64), 0, s, ctr, nsegs);
      BWT_LAUNCH("sa_seg_tiles", k_seg_tile_counts, dim3((unsigned)div_up(nsegs, 256)), dim3(256), 0, s,
                 segs[cur], nsegs, ctr, tile_first);
      BWT_TRY(device_scan_u32(ScanOp::kAddExclusive, tile_first, tile_first, nsegs, &ctr->total_tiles,
                              scanws, scanws_bytes, s));
    }
    uint32_t *lcounts = implicit ? counts0 : counts;
    if (levels > 1) BWT_HIP(hipMemsetAsync(counts, 0, (size_t)nsegs * 256 * 4, s));  // else zeroed by k_prepare_text
    const unsigned tile_bound = (unsigned)(((uint64_t)n / kPartTile + nsegs + 1 + 7) / 8 * 8);  // eighths, see locate_tile
    const unsigned hist_groups = (tile_bound + kHistTilesPerGroup - 1) / kHistTilesPerGroup;
    const uint32_t *srckey = src_id == kSrcA ? keyA : keyB;
    unsigned long long *d_range = reinterpret_cast<unsigned long long *>(ws + L.range);
    if (src_id == kSrcImplicit) {
      if (range) BWT_HIP(hipMemsetAsync(d_range, 0, kRangeBytes + 16, s));
      BWT_LAUNCH("sa_part_hist_l0", k_part_hist0, dim3(hist_groups), dim3(kPartThreads), 0, s, tb, n, ctr, lcounts, dlo,
                 dhi, range ? d_range : (unsigned long long *)nullptr);
    } else if (levels == 1) {
      // level 0 stored words that start with this level's four bases
      BWT_LAUNCH("sa_part_hist", k_part_hist<3>, dim3(hist_groups), dim3(kPartThreads), 0, s, tb, src,
                 srckey, segs[cur], tile_first, depth, ctr, lcounts);
    } else {
      // the word stored by the previous level holds this level's four bases in its second byte
      BWT_LAUNCH("sa_part_hist", k_part_hist<2>, dim3(hist_groups), dim3(kPartThreads), 0, s, tb, src,
                 srckey, segs[cur], tile_first, depth, ctr, lcounts);
    }
    // Without a gather a suffix that ends in a leaf keeps bases [depth, reach) as
    // its sort key, and one that goes on needs the next level's digit, bases
    // [depth + 4, depth + 8).  Level 0 always reads the text.
    const bool gather_all = !implicit && (reach < depth + 11 || reach < depth + 2 * kDigitBases);
    const uint32_t leaf_keybases = (implicit || gather_all) ? kLeafKeyBases : reach - depth;
    if (range && depth == 0) {
      // rows of other shards leave the picture here
      uint64_t *d_out = reinterpret_cast<uint64_t *>(d_range) + kRangeSlots * kRangeSlotStride;
      uint64_t h_out[2] = {0, 0};
      BWT_LAUNCH("sa_restrict_level0", k_restrict_level0, dim3(1), dim3(256), 0, s, lcounts, dlo, dhi, d_range, d_out);
      BWT_HIP(hipMemcpyAsync(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost, s));
      BWT_HIP(hipStreamSynchronize(s));
      range->row_begin = 1 + h_out[0];
      range->rows = h_out[1];
      if (range->rows > range->capacity)
        return set_error(kErrArg, "bucket range holds %llu rows, the output has room for %llu",
                         (unsigned long long)range->rows, (unsigned long long)range->capacity);
      if (range->rows == 0) {
        range->endloc = 0;
        return kOk;
      }
    }
    BWT_LAUNCH("sa_part_classify", k_part_classify, dim3(nsegs), dim3(256), 0, s, segs[cur], lcounts, depth,
               leaf_src_word(dst_id, leaf_keybases < kLeafKeyBases ? leaf_keybases : kLeafKeyBases), segs[cur ^ 1],
               L.seg_cap, leaves, L.leaf_cap, leafmask, ctr, implicit ? 1 : 0, bigleaves,
               midleaves);
    if (src_id == kSrcImplicit) {
      BWT_LAUNCH("sa_part_scatter_l0", k_part_scatter0, dim3(tile_bound), dim3(kScat1Threads), 0, s, tb, n, dst,
                 dstkey, ctr, lcounts, leafmask, dlo, dhi);
      reach = kDigitBases + kLeafKeyBases;  // suffixes that go on carry bases 4..18
    } else {
      BWT_LAUNCH(levels == 1 ? "sa_part_scatter_l1" : "sa_part_scatter_l2+", k_part_scatter,
                 dim3(tile_bound), dim3(kScat1Threads), 0, s, tb, src, srckey, dst, dstkey, segs[cur], tile_first,
                 depth, ctr, lcounts, gather_all ? 1 : 0, levels == 1 ? 1 : 0);
      if (gather_all) reach = depth + kLeafKeyBases;
    }
    BWT_CHECK_LAUNCH();
    cur ^= 1;
    src_id = dst_id;
    depth += kDigitBases;
    ++levels;
    if (levels < expected_levels && depth < kMaxDepth) {
      // go on without the count: 256 children per segment at most, each of more than kLeafMax rows
      const uint64_t most = (uint64_t)nsegs * 256;
      nsegs = (uint32_t)(most < L.seg_cap ? most : L.seg_cap);
      exact = false;
      continue;
    }
    if (extras && extras->readback && ahead.ev && levels == expected_levels) {
      // The counters go to pinned memory and the host waits for that copy alone,
      // while the device already sorts the leaves there are so far -- about as
      // many as this guess, the rest (if any) after the last level.
      BWT_HIP(hipMemcpyAsync(extras->readback, ctr, sizeof h_ctr, hipMemcpyDeviceToHost, s));
      BWT_HIP(hipEventRecord(ahead.ev, s));
      const uint32_t guess = (n / 512 + kWaveGroup) / kWaveGroup * kWaveGroup;
      BWT_LAUNCH("sa_leaf_sort", k_leaf_sort_wave, dim3(guess / kWaveGroup), dim3(64 * kWaveGroup), 0, s, tb, bufA, bufB,
                 keyA, keyB, leaves, n, bufA, ghead, bw, leaf_ties, samples, ctr, 0u, deferred);
      BWT_HIP(hipEventSynchronize(ahead.ev));
      std::memcpy(&h_ctr, extras->readback, sizeof h_ctr);
      leaves_sorted = h_ctr.nleaves < guess ? h_ctr.nleaves : guess;
    } else {
      BWT_HIP(hipMemcpyAsync(&h_ctr, ctr, sizeof h_ctr, hipMemcpyDeviceToHost, s));
      BWT_HIP(hipStreamSynchronize(s));
    }
    if (h_ctr.overflow) return set_error(kErrInternal, "SA build: bucket list overflow at depth %u", depth - kDigitBases);
    nsegs = h_ctr.nnext;
    nleaves_host = h_ctr.nleaves;
    nbig_host = h_ctr.nbig;
    nmid_host = h_ctr.nmid;
    exact = true;
  }
  if (!exact) return set_error(kErrInternal, "SA build: level count not read back");
  g_stats.v[0] = levels;
  g_stats.v[1] = nleaves_host;
  g_stats.v[4] = depth;
  if (nleaves_host > leaves_sorted)
    BWT_LAUNCH("sa_leaf_sort", k_leaf_sort_wave, dim3((unsigned)div_up(nleaves_host - leaves_sorted, kWaveGroup)),
               dim3(64 * kWaveGroup), 0, s, tb, bufA, bufB, keyA, keyB, leaves, n, bufA, ghead, bw, leaf_ties, samples,
               ctr, leaves_sorted, deferred);
  {
    // The listed leaves are other rows than the one-wave sort's: on a stream of their own they
    // run beside it.  Everything before the one-wave sort has finished (the host has read the
    // counts), so that stream waits for nothing; `s` waits for it.
    hipStream_t ls = (nmid_host || nbig_host) ? leaf_side_stream(s) : s;
    if (nmid_host)
      BWT_LAUNCH("sa_leaf_sort_mid", k_leaf_sort_wave_mid, dim3((unsigned)div_up(nmid_host, kWaveGroup)),
                 dim3(64 * kWaveGroup), 0, ls, tb, bufA, bufB, keyA, keyB, leaves, n, bufA, ghead, bw, leaf_ties,
                 samples, ctr, midleaves, deferred);
    if (nbig_host)
      BWT_LAUNCH("sa_leaf_sort_big", k_leaf_sort, dim3(nbig_host), dim3(kLeafThreads), 0, ls, tb, bufA, bufB, keyA,
                 keyB, leaves, n, bufA, ghead, bw, leaf_ties, samples, ctr, bigleaves, 0u);
    if (ls != s) BWT_TRY(leaf_side_join(s));
  }
  // A whole-text build whose caller reads the counters at its own synchronisation learns there
  // whether any leaf was put off (sa_build_settle): none in a text without repeats
  const bool settle_later = extras && extras->readback && !range && !nsegs;
  if (nleaves_host && !settle_later) {
    // what the waves left to whole workgroups when they met it (lopsided keys): the count is on the device,
    // zero for a text without repeats
    const unsigned groups = nleaves_host < 512u ? nleaves_host : 512u;
    BWT_LAUNCH("sa_leaf_sort_deferred", k_leaf_sort_deferred, dim3(groups), dim3(kLeafThreads), 0, s, tb, bufA, bufB,
               keyA, keyB, leaves, n, bufA, ghead, bw, leaf_ties, samples, ctr, deferred);
  }
  if (nsegs) {
    const uint32_t *src = src_id == kSrcA ? bufA : bufB;
    uint32_t *shortl = counts;  // the levels are done with their counters: 256 words a segment, 33 used
    static_assert(kUnresList + 1 <= 256, "a segment's list of short suffixes fits its counters");
    BWT_HIP(hipMemsetAsync(shortl, 0, (size_t)nsegs * (kUnresList + 1) * 4, s));
    BWT_LAUNCH("sa_unresolved_segs", k_unresolved_copy, dim3(nsegs * kUnresParts), dim3(256), 0, s, segs[cur], src,
               bufA, ghead, depth, n, shortl);
    BWT_LAUNCH("sa_unresolved_segs", k_unresolved_short, dim3((unsigned)div_up(nsegs, 64)), dim3(64), 0, s, segs[cur],
               bufA, ghead, depth, n, shortl, nsegs, ctr);
    BWT_LAUNCH("sa_unresolved_segs", k_unresolved_heads, dim3(nsegs * kUnresParts), dim3(256), 0, s, segs[cur], ghead,
               shortl);
  }
  BWT_CHECK_LAUNCH();
  if (settle_later) {
    // the usual case has no tied rows: let the caller get on with it (see SaExtras)
    static_assert(sizeof(Counters) <= 16 * sizeof(uint32_t), "readback words");
    if (!extras->caller_copies_counters)
      BWT_HIP(hipMemcpyAsync(extras->readback, ctr, sizeof h_ctr, hipMemcpyDeviceToHost, s));
    extras->bw = bw;
    extras->endloc = &ctr->endloc;
    extras->samples_done = samples != nullptr;
    extras->pending = true;
    extras->later.d_sa = d_sa;
    extras->later.ghead = ghead;
    extras->later.flags = bufB;  // free from here on
    extras->later.leaves = leaves;
    extras->later.leaf_ties = leaf_ties;
    extras->later.ctr = ctr;
    extras->later.n = n;
    extras->later.nleaves = nleaves_host;
    extras->later.tb = tb;
    extras->later.bufA = bufA;
    extras->later.bufB = bufB;
    extras->later.keyA = keyA;
    extras->later.keyB = keyB;
    extras->later.deferred = deferred;
    extras->later.samples = samples;
    extras->later.bw = bw;
    return kOk;
  }
  BWT_HIP(hipMemcpyAsync(&h_ctr, ctr, sizeof h_ctr, hipMemcpyDeviceToHost, s));
  BWT_HIP(hipStreamSynchronize(s));
  g_stats.v[2] = h_ctr.unresolved;
  if (range) {
    // the tie groups that are left need the ranks of suffixes of every shard
    if (h_ctr.unresolved || nsegs)
      return set_error(kErrNeedsWholeText,
                       "%u rows of bucket range [%u, %u) tie beyond the bucket sort (repeats): build this text whole",
                       h_ctr.unresolved, dlo, dhi);
    range->endloc = h_ctr.endloc ? (uint32_t)(range->row_begin - 1 + h_ctr.endloc) : 0u;
  }
  if (extras && !h_ctr.unresolved && !nsegs) {
    // every row came out of a leaf sort: its BWT symbol and the $ row are known
    extras->bw = bw;
    extras->endloc = &ctr->endloc;
    extras->samples_done = samples != nullptr;
  }
  if (h_ctr.unresolved) {
    // bufB is free from here on: reuse it for the active-row flags
    if (nleaves_host)
      BWT_LAUNCH("sa_ghead_identity", k_ghead_identity, dim3(nleaves_host), dim3(256), 0, s, leaves, leaf_ties,
                 ghead);
    BWT_TRY(sa_finish_ties(d_sa, ghead, n, h_ctr.min_depth, bufB, ctr, s));
  }
  return kOk;
}

int sa_build_settle(SaExtras *extras, hipStream_t s) {
  if (!extras || !extras->pending) return 0;
  extras->pending = false;
  Counters h_ctr;
  std::memcpy(&h_ctr, extras->readback, sizeof h_ctr);
  const auto &L = extras->later;
  Counters *ctr = static_cast<Counters *>(L.ctr);
  const bool put_off = h_ctr.ndeferred != 0;
  if (put_off) {
    // leaves whose keys were too lopsided for one wave (repeats): a workgroup each, now
    const unsigned groups = L.nleaves < 512u ? L.nleaves : 512u;
    BWT_LAUNCH("sa_leaf_sort_deferred", k_leaf_sort_deferred, dim3(groups), dim3(kLeafThreads), 0, s, L.tb, L.bufA,
               L.bufB, L.keyA, L.keyB, static_cast<const Leaf *>(L.leaves), L.n, L.bufA, L.ghead, L.bw,
               static_cast<uint8_t *>(L.leaf_ties), L.samples, ctr, L.deferred);
    BWT_CHECK_LAUNCH();
    BWT_HIP(hipMemcpyAsync(&h_ctr, ctr, sizeof h_ctr, hipMemcpyDeviceToHost, s));
    BWT_HIP(hipStreamSynchronize(s));
  }
  g_stats.v[2] = h_ctr.unresolved;
  if (!h_ctr.unresolved && !put_off) return 0;
  extras->bw = nullptr;
  extras->endloc = nullptr;
  extras->samples_done = false;
  if (h_ctr.unresolved) {
    if (L.nleaves)
      BWT_LAUNCH("sa_ghead_identity", k_ghead_identity, dim3(L.nleaves), dim3(256), 0, s,
                 static_cast<const Leaf *>(L.leaves), static_cast<const uint8_t *>(L.leaf_ties), L.ghead);
    BWT_TRY(sa_finish_ties(L.d_sa, L.ghead, L.n, h_ctr.min_depth, L.flags, ctr, s));
  }
  BWT_HIP(hipStreamSynchronize(s));
  return 1;
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

size_t bwtgpu_sa_workspace_bytes(uint32_t n) { return layout_for(n ? n : 1).end; }

int bwtgpu_sa_build(const uint8_t *d_text, uint32_t n, uint32_t *d_sa, void *d_ws, size_t ws_bytes,
                    void *stream) {
  return sa_build_impl(d_text, n, d_sa, d_ws, ws_bytes, (hipStream_t)stream, nullptr);
}

int bwtgpu_sa_last_stats(uint64_t stats[8]) {
  for (int i = 0; i < 8; ++i) stats[i] = g_stats.v[i];
  return kOk;
}

}  // extern "C"
