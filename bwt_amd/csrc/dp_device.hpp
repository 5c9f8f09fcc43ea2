// The aligner's dynamic programming (nw_fast, smw.c:26-111, and sw_fast,
// smw.c:117-200) on one wave.
//
// The matrix is swept in strips of 64 columns.  Lane l of a strip owns column
// j0 + l and at step t fills row t - l, so the cells of one step lie on an
// anti-diagonal and need only what the lane to the left held one step (left
// neighbour, its direction) and two steps (diagonal neighbour) earlier: three
// lane shifts, no memory; the read base of a row travels the same way.  The
// last lane of a strip leaves its column in LDS for the first lane of the next
// strip.  Directions (2 bits a cell) are packed by each lane along its own
// column, sixteen rows to a word, and go to LDS for the traceback, which the
// wave walks serially.
#pragma once

#include "common.hpp"

namespace bwtgpu {

// the value lane l - 1 holds, in lane l (lane 0 keeps its own): one DPP move
// across the whole wave instead of a trip through the LDS crossbar
__device__ __forceinline__ int wave_shift_up(int v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

struct DpLds {
  uint32_t dirs, edge_v, edge_p, ops, total;  // byte offsets from the start of the wave's area
  uint32_t words_per_col;
};

// LDS one wave needs for pieces of up to max1 x max2 bases
inline DpLds dp_lds_layout(uint32_t max1, uint32_t max2) {
  DpLds L;
  L.words_per_col = (max1 + 1 + 15) / 16;
  L.dirs = 0;
  L.edge_v = L.dirs + max2 * L.words_per_col * 4;
  L.edge_p = L.edge_v + (max1 + 1) * 4;
  L.ops = L.edge_p + (max1 + 1) * 4;
  L.total = L.ops + (uint32_t)align_up((size_t)max1 + max2 + 1, 4);
  return L;
}

// One alignment by the calling wave (all 64 lanes, one wave per workgroup, so
// __syncthreads is the wave's own barrier).  s1_at(i), i < len1: bases of the
// read piece (0..3, 5 = N, which agrees with every base), read from LDS or
// global by one lane at a time; s2_at(j), j < len2: bases of the genome piece.
// mode 0 = nw_fast: ends at the best cell of the last row (first one on ties,
// smw.c:65-70) and returns that column - 1; mode 1 = sw_fast: ends in the last
// cell, returns 0.  out receives the unit operations 'M', 'I', 'D' in the order
// the reference's stack_push calls reach the caller's stack (the traceback
// order for mode 1, smw.c:167-191; its reverse for mode 0, which collects them
// on a second stack and flips it, smw.c:86-107); *count of them.
template <class S1, class S2>
__device__ __forceinline__ int dp_wave(S1 s1_at, int len1, S2 s2_at, int len2, int mode, const DpLds &L,
                                       uint8_t *area, uint8_t *out, uint32_t *count) {
  uint32_t *dirs = reinterpret_cast<uint32_t *>(area + L.dirs);
  int *edge_v = reinterpret_cast<int *>(area + L.edge_v);
  uint32_t *edge_p = reinterpret_cast<uint32_t *>(area + L.edge_p);
  uint8_t *ops = area + L.ops;
  const uint32_t lane = threadIdx.x & 63u;
  *count = 0;
  if (mode == 0 && len1 == 0) return 0;  // smw.c:28-30
  __syncthreads();  // the area may still be read by the previous call's last steps
  // column 0 (smw.c:37, :44-45) is what the first strip finds to its left
  for (int i = (int)lane; i <= len1; i += 64) {
    edge_v[i] = i ? -5 - 3 * i : 0;
    edge_p[i] = 1;
  }
  __syncthreads();
  const uint32_t W = L.words_per_col;
  int best_v = -5 - 3 * len1, best_j = 0;  // column 0 of the last row (smw.c:45, :65-70)
  for (int j0 = 1; j0 <= len2; j0 += 64) {
    const int j = j0 + (int)lane;
    const bool col = j <= len2;
    const int lanes = len2 - j0 + 1 < 64 ? len2 - j0 + 1 : 64;
    const uint32_t c2 = col ? s2_at(j - 1) : 0u;
    int vc = -5 - 3 * j, vp = 0;  // row 0 (smw.c:38-41)
    uint32_t pc = 2, pw = 0, c1 = 0;
    int ev_prev = edge_v[0];  // the column to the left, one row up
    const int steps = len1 ? len1 + lanes - 1 : 0;  // sw_fast of an empty read has no cells
    for (int t = 1; t <= steps; ++t) {
      const int i = t - (int)lane;
      // what the lane to the left held after the previous step: its cell of this
      // row, that cell's direction, its cell one row up, and the read base of this
      // row (the same base travels down the anti-diagonal)
      int lv = wave_shift_up(vc), dv = wave_shift_up(vp);
      uint32_t lp = (uint32_t)wave_shift_up((int)pc);
      c1 = (uint32_t)wave_shift_up((int)c1);
      // the first lane takes them from the column left of the strip (one address for the whole wave)
      const int te = t <= len1 ? t : len1;
      const int ev = edge_v[te];
      const uint32_t ep = edge_p[te];
      if (lane == 0) {
        lv = ev;
        lp = ep;
        dv = ev_prev;
        c1 = s1_at(te - 1);
      }
      ev_prev = ev;
      const bool cell = col && i >= 1 && i <= len1;
      const int d = dv + ((c1 == 5u || c1 == c2) ? 0 : -6);  // smw.c:52
      const int l = lv - 3 + (lp == 2u ? 0 : -5);             // :48, :53
      const int u = vc - 3 + (pc == 1u ? 0 : -5);             // :49, :54
      int v = d > l ? d : l;
      v = v > u ? v : u;
      const uint32_t p = v == l ? 2u : v == u ? 1u : 0u;      // :55-62
      vp = cell ? vc : vp;
      vc = cell ? v : vc;
      pc = cell ? p : pc;
      pw |= cell ? p << (2 * (i & 15)) : 0u;
      if (cell && ((i & 15) == 15 || i == len1)) {
        dirs[(uint32_t)(j - 1) * W + (uint32_t)(i >> 4)] = pw;
        pw = 0;
      }
      if (cell && lane == 63) {  // rows t - 63 and below are behind the first lane, which reads row t
        edge_v[i] = v;
        edge_p[i] = p;
      }
    }
    // the last row: the first best cell wins (smw.c:65-70)
    int cv = col ? vc : -2000000, cj = j;
    for (int off = 1; off < 64; off <<= 1) {
      const int ov = __shfl_xor(cv, off, 64), oj = __shfl_xor(cj, off, 64);
      if (ov > cv || (ov == cv && oj < cj)) {
        cv = ov;
        cj = oj;
      }
    }
    if (cv > best_v) {
      best_v = cv;
      best_j = cj;
    }
    __syncthreads();
    if (lane == 0) edge_v[0] = -5 - 3 * (j0 + 63);  // row 0 of the column the next strip starts from
    __syncthreads();
  }
  // traceback (smw.c:71-107, :154-191): every lane walks the same path, lane 0 writes it down
  int i = len1, j = mode == 0 ? best_j : len2;
  uint32_t k = 0;
  while (i && j) {
    const uint32_t p = (dirs[(uint32_t)(j - 1) * W + (uint32_t)(i >> 4)] >> (2 * (i & 15))) & 3u;
    uint8_t ch;
    if (p == 1u) {
      --i;
      ch = 'I';
    } else if (p == 2u) {
      --j;
      ch = 'D';
    } else {
      --i;
      --j;
      ch = 'M';
    }
    if (lane == 0) ops[k] = ch;
    ++k;
  }
  for (; i; --i, ++k)
    if (lane == 0) ops[k] = 'I';
  for (; j; --j, ++k)
    if (lane == 0) ops[k] = 'D';
  __syncthreads();
  for (uint32_t q = lane; q < k; q += 64) out[q] = mode == 0 ? ops[k - 1 - q] : ops[q];
  *count = k;
  return mode == 0 ? best_j - 1 : 0;
}

}  // namespace bwtgpu
