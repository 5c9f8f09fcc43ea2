// Device-side mms (seqindex.c:314-351) and unc_sa (seqindex.c:244-255), shared by
// the batch kernels and the aligner.
#pragma once

#include "common.hpp"
#include "fm_device.hpp"

namespace bwtgpu {

struct CTab {
  uint32_t c[5];
};

__device__ __forceinline__ uint32_t c_of(const CTab &C, uint32_t k) {
  return k == 0 ? C.c[0] : k == 1 ? C.c[1] : k == 2 ? C.c[2] : k == 3 ? C.c[3] : C.c[4];
}

// C[c] + Occ(c, idx) through the per-symbol layout
__device__ __forceinline__ uint32_t occ_sym(const IndexView &v, uint32_t c, uint32_t idx) {
  const uint32_t j = idx - (idx > v.endloc);
  uint32_t r;
  const uint32_t b = div96(j, r);
  return sym_count(reinterpret_cast<const uint4 *>(v.sym)[(uint64_t)c * v.nsym + b], r);
}

// all four C[d] + Occ(d, idx) from one 32-byte block
__device__ __forceinline__ void occ_all(const IndexView &v, uint32_t idx, uint32_t out[4]) {
  const uint32_t j = idx - (idx > v.endloc);
  const IndexBlock *b = v.blocks + (j >> 6);
  const uint4 occ = *reinterpret_cast<const uint4 *>(b->occ);
  const uint64_t lo = b->lo, hi = b->hi, m = low_mask(j & 63u);
  out[0] = occ.x + (uint32_t)__popcll(~lo & ~hi & m);
  out[1] = occ.y + (uint32_t)__popcll(lo & ~hi & m);
  out[2] = occ.z + (uint32_t)__popcll(~lo & hi & m);
  out[3] = occ.w + (uint32_t)__popcll(lo & hi & m);
}

// base p of the text (p < n) from the 16-base words kept with the suffix array
__device__ __forceinline__ uint32_t text_base2(const IndexView &v, uint32_t p) {
  return (v.text2[1 + (p >> 4)] >> (2u * (p & 15u))) & 3u;
}

// mms (seqindex.c:314-351).  Symbols 0..3 are bases, 5 is N: trailing Ns are
// skipped (:317-320), an inner N is read as the base with the most
// occurrences in the current interval, lowest base on ties (:330-340).
// A pattern made of Ns only (the reference reads before the buffer there)
// returns its length with an empty interval.
//
// Two shortcuts, both leaving the results as the loop alone gives them: the
// search starts from the k-mer table entry of the last k bases when none of
// them is N and the string occurs; and once the interval is a single row of an
// index that keeps SA, ISA and the text, the remaining bases are compared with
// the text in front of that row's suffix -- an N there always agrees, being
// read as the one base that precedes the row -- and the last non-empty
// interval is the row of the suffix that many bases earlier.
//
// Symbol i of the pattern is pat[i * step]: step = -1 reads a buffer backwards
// (the reversed read of search_reads without a second copy).  pat may point
// into LDS.
//
// hint: what the caller already knows about the last k bases (k the table's
// string length), used when no trailing N is skipped: 0 nothing, 1 no N among
// them and hx is their table index, 2 an N among them.
//
// at_least: the caller only asks whether the result reaches this length (the
// anchor test of search_reads); when the table shows that the last k bases do
// not occur and k <= at_least, the answer is "no" without a single rank query
// and 0 is returned.  0 asks for the exact value.
//
// WAVE: the 64 lanes of the calling wave run the call together on one pattern
// (the aligner); the comparison with the text then takes 64 bases a step, one
// per lane, instead of one.
template <bool WAVE = false>
__device__ __forceinline__ int mms_dev(const IndexView &v, const CTab &C, const uint8_t *pat, int step,
                                       int len, uint32_t *sp_out, uint32_t *ep_out, int at_least,
                                       int hint = 0, uint32_t hx = 0) {
  int skips = 0;
  while (len > 0 && pat[(len - 1) * step] == 5) {
    --len;
    ++skips;
  }
  if (len == 0) {
    *sp_out = 0;
    *ep_out = 0;
    return skips;
  }
  uint32_t c = pat[(len - 1) * step] & 3u;
  uint32_t start = c_of(C, c), end = c_of(C, c + 1);
  int i = len - 2;
  uint32_t table_pos = 0xFFFFFFFFu;  // SA[start] when the table gave a single row and said where it lies
  const int K = (int)v.kmer_k;
  if (K && len >= K) {
    uint32_t x = hx;
    bool plain = hint == 1;
    if (!hint || skips) {
      x = 0;
      plain = true;
      for (int k = 0; k < K; ++k) {
        const uint32_t raw = pat[(len - K + k) * step];
        plain = plain && raw != 5;
        x = (x << 2) | (raw & 3u);
      }
    }
    if (plain) {
      uint32_t e_sp, e_pos;
      const uint32_t e_ep = kmer_interval(v, x, e_sp, e_pos);
      if (e_ep > e_sp) {
        start = e_sp;
        end = e_ep;
        i = len - K - 1;
        if (e_ep - e_sp == 1) table_pos = e_pos;  // consumed by the first pass of the loop below, before any step
      } else if (K <= at_least && skips == 0) {
        // fewer than K bases match and none was skipped: below the caller's bar
        *sp_out = 0;
        *ep_out = 0;
        return 0;
      } else if (v.kmer2) {
        // the last K bases do not occur: go on from the interval of the last K - 2 (or K - 4) where
        // the index keeps those levels of the table, instead of from the last base
        const uint2 iv = v.kmer2[x & ((1u << (2 * (K - 2))) - 1u)];
        if (iv.y > iv.x) {
          start = iv.x;
          end = iv.y;
          i = len - K + 1;
        } else if (v.kmer4) {
          const uint2 iv4 = v.kmer4[x & ((1u << (2 * (K - 4))) - 1u)];
          if (iv4.y > iv4.x) {
            start = iv4.x;
            end = iv4.y;
            i = len - K + 3;
          }
        }
      }
    }
  }
  uint32_t sp = start, ep = end;
  for (; i >= 0; --i) {
    if (end <= start) break;
    sp = start;
    ep = end;
    if (v.isa && end - start == 1) {
      const uint32_t pos = table_pos != 0xFFFFFFFFu ? table_pos : __builtin_nontemporal_load(v.sa + start);
      uint32_t k = 0;  // bases that agree with the text in front of pos
      if (WAVE) {
        const uint32_t lane = threadIdx.x & 63u;
        for (;;) {
          const uint32_t kk = k + lane;
          bool agrees = false;
          if ((int)kk <= i && kk < pos) {
            const uint32_t raw = pat[(i - (int)kk) * step];
            agrees = raw == 5 || (raw & 3u) == text_base2(v, pos - 1 - kk);
          }
          const uint64_t stops = __ballot(!agrees);  // the first lane that disagrees or is past the end
          if (stops) {
            k += (uint32_t)__ffsll((unsigned long long)stops) - 1u;
            break;
          }
          k += 64;
        }
      } else {
        while ((int)k <= i && k < pos) {
          const uint32_t raw = pat[(i - (int)k) * step];
          if (raw != 5 && (raw & 3u) != text_base2(v, pos - 1 - k)) break;
          ++k;
        }
      }
      const uint32_t row = k ? isa_at(v, pos - k) : start;
      *sp_out = row;
      *ep_out = row + 1;
      return len - 1 - i + (int)k + skips;
    }
    const uint32_t raw = pat[i * step];
    if (raw == 5) {
      uint32_t a[4], b[4];
      occ_all(v, start, a);
      occ_all(v, end, b);
      c = 0;
      int64_t best = -1;
      for (uint32_t d = 0; d < 4; ++d) {
        const int64_t cnt = (int64_t)b[d] - (int64_t)a[d];
        if (cnt > best) {
          best = cnt;
          c = d;
        }
      }
      start = a[c];
      end = b[c];
    } else {
      c = raw & 3u;
      start = occ_sym(v, c, start);
      end = occ_sym(v, c, end);
    }
    table_pos = 0xFFFFFFFFu;  // the interval has moved on from the table's
  }
  if (end <= start) {
    *sp_out = sp;
    *ep_out = ep;
    return len - i - 2 + skips;
  }
  *sp_out = start;
  *ep_out = end;
  return len - i - 1 + skips;
}

// unc_sa (seqindex.c:244-255) for one row, in line
__device__ __forceinline__ uint32_t unc_sa_dev(const IndexView &v, uint32_t idx) {
  uint32_t steps = 0;
  while (idx & kDevSampleMask) {
    idx = lf_step(v, idx);
    if (++steps > v.n || idx > v.n) return 0xFFFFFFFFu;  // not the LF map of a text
  }
  uint64_t x = (uint64_t)v.idxs[idx >> kDevSampleShift] + steps;
  if (x > v.n) x -= (uint64_t)v.n + 1;
  return (uint32_t)x;
}

}  // namespace bwtgpu
