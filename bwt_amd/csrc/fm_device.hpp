// Device-side FM-index primitives over the 32-byte block layout.
#pragma once

#include "common.hpp"

namespace bwtgpu {

// What a kernel needs of an index, passed by value (lives in SGPRs).
struct IndexView {
  const IndexBlock *blocks;
  const SymBlock *sym;
  const uint32_t *idxs;
  const uint32_t *sa;     // null unless the index keeps the whole suffix array
  const uint32_t *isa;    // PosRecords, 16 words each
  const uint32_t *text2;
  const uint32_t *kmer;   // null unless the index has the k-mer interval table
  const uint2 *kmer2;     // null, or the intervals of the strings of kmer_k - 2 bases
  const uint2 *kmer4;     // null, or those of the strings of kmer_k - 4 bases
  uint32_t kmer_k;
  uint32_t kmer_w;        // words per table entry: 1 (start) or 2 (start, SA[start])
  uint32_t kmer_sizebits; // top bits of a two-word entry's second word that hold the interval's size, or 0
  uint64_t nsym;
  uint32_t n;
  uint32_t endloc;
  uint32_t rec_wide;      // the position records are the wide kind
};

inline IndexView view_of(const bwtgpu_index *ix) {
  return IndexView{ix->d_blocks, ix->d_sym, ix->d_idxs, ix->d_sa,   ix->d_isa, ix->d_text2,
                   ix->d_kmer,   ix->d_kmer2, ix->d_kmer4, ix->kmer_k, ix->kmer_w, ix->kmer_sizebits, ix->nsym, ix->n, ix->endloc,
                   ix->rec_wide ? 1u : 0u};
}

// The view for entry points that do not wait for the search-only arrays: all of
// them once search_ready says so; until then the rank blocks and sampled rows
// alone (what a build or a load has finished before it returns), plus the suffix
// array if it has been published -- its pointer is stored only when every row is
// in place.  The answers are the same on either view.
inline IndexView view_published(const bwtgpu_index *ix) {
  if (ix->search_ready.load(std::memory_order_acquire)) return view_of(ix);
  const uint32_t *sa = __atomic_load_n(&ix->d_sa, __ATOMIC_ACQUIRE);
  return IndexView{ix->d_blocks, ix->d_sym, ix->d_idxs, sa, nullptr, nullptr, nullptr, nullptr, nullptr,
                   0u, 1u, 0u, ix->nsym, ix->n, ix->endloc, 0u};
}

__device__ __forceinline__ uint64_t low_mask(uint32_t r) {
  // r in [0, 63]
  return (1ull << r) - 1ull;
}

// C[c] + Occ(c, idx) for a full-BWT row number idx in [0, n+1]: the value a
// backward-search step assigns (seqindex.c:236-239 with rank, :221-225, and
// seq_rank, :91-111).  nlo/nhi are the per-symbol plane inverters:
// nlo = (c & 1) ? 0 : ~0, nhi = (c & 2) ? 0 : ~0.
__device__ __forceinline__ uint32_t occ_c(const IndexView &v, uint32_t c,
                                          uint64_t nlo, uint64_t nhi, uint32_t idx) {
  idx -= (idx > v.endloc);  // the $ row is not stored (seqindex.c:222-223)
  const IndexBlock *b = v.blocks + (idx >> 6);
  const uint32_t base = b->occ[c];
  const uint64_t lo = b->lo, hi = b->hi;
  const uint64_t match = (lo ^ nlo) & (hi ^ nhi) & low_mask(idx & 63u);
  return base + (uint32_t)__popcll(match);
}

// The SA interval [sp, return value) of the string numbered x in the k-mer
// table; empty when the two are equal.  pos (when the table has two words per
// entry, else 0xFFFFFFFF): SA[sp], the text position of the string's first
// occurrence in row order -- its only one when the interval is a single row.
// next_read: whether the start of the next string had to be read as well (4 more bytes).
__device__ __forceinline__ uint32_t kmer_interval(const IndexView &v, uint32_t x, uint32_t &sp, uint32_t &pos,
                                                 bool &next_read) {
  uint32_t ep;
  next_read = v.kmer_w == 2;
  if (v.kmer_w == 2) {
    // start and position of x, start of x + 1: three words of one 16-byte-or-less span
    // The entry is read once per search and the table is many times the size of
    // the last-level cache: the load goes past the caches, which then keep the
    // position records and rank blocks (46 Mbp: 16.0 -> 18.0 G reads/s).
    const uint32_t *e = v.kmer + 2ull * x;
    if (v.kmer_sizebits) {  // the entry says how many rows, unless they are many
      const uint64_t mine = __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(e));
      sp = (uint32_t)mine;
      pos = (uint32_t)(mine >> 32);
      const uint32_t posbits = 32u - v.kmer_sizebits, size = pos >> posbits;
      pos &= (1u << posbits) - 1u;
      if (size < (1u << v.kmer_sizebits) - 1u) {
        next_read = false;
        return sp + size;
      }
      ep = e[2];
    } else {
      // no room for the size: the entry and the start of the next one in one 16-byte load
      // (a load past the caches leaves no line behind for a second one to find)
      typedef uint32_t u32x4_at_8 __attribute__((ext_vector_type(4), aligned(8)));
      const auto q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_at_8 *>(e));
      sp = q.x;
      pos = q.y;
      ep = q.z;
    }
  } else {
    // one 8-byte load at a 4-byte aligned address: both starts in one request
    typedef uint64_t u64_at_4 __attribute__((aligned(4)));
    const uint64_t both = __builtin_nontemporal_load(reinterpret_cast<const u64_at_4 *>(v.kmer + x));
    sp = (uint32_t)both;
    ep = (uint32_t)(both >> 32);
    pos = 0xFFFFFFFFu;
  }
  // rows of suffixes shorter than the table's strings that lie between the two;
  // every pair is read (unused ones name no string): the loads are uniform and
  // issue together
  const uint2 *gaps = reinterpret_cast<const uint2 *>(v.kmer + ((uint64_t)v.kmer_w << (2 * v.kmer_k)) + 2);
#pragma unroll
  for (uint32_t g = 0; g < kKmerMaxGaps; ++g) {
    const uint2 e = gaps[g];
    ep -= (uint64_t)e.x == (uint64_t)x + 1 ? e.y : 0u;  // x + 1 does not fit 32 bits for the last string of 16 bases
  }
  return ep;
}
__device__ __forceinline__ uint32_t kmer_interval(const IndexView &v, uint32_t x, uint32_t &sp, uint32_t &pos) {
  bool next_read;
  return kmer_interval(v, x, sp, pos, next_read);
}
__device__ __forceinline__ uint32_t kmer_interval(const IndexView &v, uint32_t x, uint32_t &sp) {
  uint32_t pos;
  return kmer_interval(v, x, sp, pos);
}

// Second word of a two-word table entry whose interval has from 2 to
// kmer_symbol_rows() rows: instead of a position (of use only when the interval is
// a single row) it holds the BWT symbols of the interval's rows, row t at bits
// 2t, and a flag above them in bit F - 1, F = 32 - kmer_sizebits being the width
// of the field.  The flag is clear when the $ row lies inside the interval.  With
// the symbols a search knows which rows go on with its next base without a rank
// query: if it is exactly one, that row's suffix decides everything that follows.
__device__ __forceinline__ uint32_t kmer_symbol_rows(uint32_t sizebits) {
  const uint32_t rows = (32u - sizebits - 1u) / 2u;
  return rows < 15u ? rows : 15u;
}

// The rows of [sp, sp + size) whose BWT symbol is c, as a mask with bit 2t set for row t;
// 0 when the entry holds no symbols (see above).  `field` is the entry's second word
// without its size bits.
__device__ __forceinline__ uint32_t kmer_rows_with(uint32_t field, uint32_t size, uint32_t sizebits, uint32_t c) {
  const uint32_t F = 32u - sizebits;
  if (size < 2u || size > kmer_symbol_rows(sizebits) || !((field >> (F - 1u)) & 1u)) return 0u;
  const uint32_t x = field ^ (c * 0x55555555u);
  return ~(x | (x >> 1)) & 0x55555555u & ((1u << (2u * size)) - 1u);
}

// 16 bases of text starting at position o (o >= -15), base k of the result at
// bits 2k; text2[0] is a guard word so that o may reach 15 below zero.
__device__ __forceinline__ uint32_t text16(const uint32_t *__restrict__ text2, int64_t o) {
  const uint64_t q = (uint64_t)(o + 16);
  const uint64_t both = ((uint64_t)text2[(q >> 4) + 1] << 32) | text2[q >> 4];
  return (uint32_t)(both >> (2u * (uint32_t)(q & 15u)));
}

// p / 12 and p % 12 for a 32-bit p
__device__ __forceinline__ uint32_t div12(uint32_t p, uint32_t &rem) {
  const uint32_t q = __umulhi(p, 0xAAAAAAABu) >> 3;
  rem = p - q * kPosPerRecord;
  return q;
}

// word index of position p's row in the array of PosRecords
__device__ __forceinline__ uint64_t isa_slot(uint32_t p, bool wide) {
  if (wide) return ((uint64_t)(p >> 3) + kWideGuardRecords) * 16u + (p & 7u);
  uint32_t o;
  const uint32_t b = div12(p, o);
  return pos_record_place((uint64_t)b + kPosGuardRecords) * 16u + o;
}

__device__ __forceinline__ uint32_t isa_at(const IndexView &v, uint32_t p) { return v.isa[isa_slot(p, v.rec_wide != 0)]; }

// j / 96 and j % 96 for a 32-bit j
__device__ __forceinline__ uint32_t div96(uint32_t j, uint32_t &rem) {
  const uint32_t q = __umulhi(j >> 5, 0xAAAAAAABu) >> 1;  // (j / 32) / 3
  rem = j - q * kSymBases;
  return q;
}

// number of set bits among the first r (0..95) bits of a SymBlock mask, plus occ
__device__ __forceinline__ uint32_t sym_count(const uint4 &b, uint32_t r) {
  const uint64_t lo64 = ((uint64_t)b.z << 32) | b.y;
  const bool full = r >= 64;
  const uint64_t mlo = full ? ~0ull : low_mask(r);
  const uint32_t mhi = full ? (uint32_t)low_mask(r - 64) : 0u;
  return b.x + (uint32_t)__popcll(lo64 & mlo) + (uint32_t)__popc(b.w & mhi);
}

// LF mapping (seqindex.c:214-219) for a row in [0, n].  Branch-free so that
// several independent walks in one lane keep their loads in flight together.
// c: the BWT symbol of the row (meaningless for the $ row).
__device__ __forceinline__ uint32_t lf_step(const IndexView &v, uint32_t idx, uint32_t &c) {
  const uint32_t j = idx - (idx > v.endloc);
  const IndexBlock *b = v.blocks + (j >> 6);
  const uint4 occ = *reinterpret_cast<const uint4 *>(b->occ);
  const uint64_t lo = b->lo, hi = b->hi;
  const uint32_t r = j & 63u;
  c = (uint32_t)((lo >> r) & 1ull) | ((uint32_t)((hi >> r) & 1ull) << 1);
  const uint64_t nlo = (c & 1u) ? 0ull : ~0ull, nhi = (c & 2u) ? 0ull : ~0ull;
  const uint64_t match = (lo ^ nlo) & (hi ^ nhi) & low_mask(r);
  const uint32_t base = c == 0 ? occ.x : c == 1 ? occ.y : c == 2 ? occ.z : occ.w;
  const uint32_t next = base + (uint32_t)__popcll(match);
  return idx == v.endloc ? 0u : next;  // the $ row maps to row 0
}
__device__ __forceinline__ uint32_t lf_step(const IndexView &v, uint32_t idx) {
  uint32_t c;
  return lf_step(v, idx, c);
}

}  // namespace bwtgpu
