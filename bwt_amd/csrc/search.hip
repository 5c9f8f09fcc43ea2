// FM-index backward search and locate: one read per lane, index resident in HBM.
//
// On rank blocks alone a read of m bases makes 2(m-1) rank queries, each one
// 16-byte load from the per-symbol layout (the two ends of an interval share
// the load once the interval is narrow), and locate walks LF steps of one
// 32-byte block each.  An index that also holds the k-mer interval table skips
// the first k steps with one 8-byte lookup; one that holds the suffix array, its
// inverse and the text finishes from the text once a single row is left and
// locates with one lookup.  Every form is bound by the rate of random cache-line
// fetches, so what these shortcuts buy is fewer lines per read, not fewer bytes.
#include <mutex>

#include "common.hpp"
#include "fm_device.hpp"

namespace bwtgpu {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // a 16-byte load the non-temporal builtin accepts

constexpr int kPackThreads = 256;
constexpr uint32_t kPackLdsBytes = 48 * 1024;

// Reads arrive as the reference passes them: one base per byte, read r at
// d_pats + r*stride (fmitest.c:60-63 uses stride 1, overlapping 12-mers).
// They are repacked once to 2 bits per base, word-major (word w of read r at
// packed[w*R + r], base k of the word at bits 2k), so the search kernel's
// pattern loads are coalesced and cost m/4 bytes per read.
__global__ void __launch_bounds__(kPackThreads)
k_pack_reads(const uint8_t *__restrict__ pats, uint32_t m, uint64_t stride, uint64_t R,
             uint32_t *__restrict__ packed, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint64_t r0 = (uint64_t)blockIdx.x * kPackThreads;
  const uint64_t r = r0 + threadIdx.x;
  const uint32_t words = (m + 15) / 16;
  if (use_lds) {
    const uint8_t *src = pats + r0 * stride;
    // bytes of this workgroup's reads; clipped to the last read of the batch
    uint64_t last = R - 1 < r0 + kPackThreads - 1 ? R - 1 : r0 + kPackThreads - 1;
    uint32_t span = (uint32_t)((last - r0) * stride + m);
    const uint32_t mis = (uint32_t)((uintptr_t)src & 15u);
    const uint8_t *asrc = src - mis;  // 16-byte aligned, still inside the allocation's first line
    // the aligned window [asrc, asrc + mis + span) may start before pats; those
    // bytes sit in the same 16-byte granule as pats[0], which hipMalloc'd or
    // torch-allocated buffers own (256-byte aligned bases), so for interior
    // workgroups they are simply earlier reads
    const uint32_t total = mis + span;
    const uint32_t vecs = total / 16;
    for (uint32_t i = threadIdx.x; i < vecs; i += kPackThreads)
      reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(asrc)[i];
    for (uint32_t i = vecs * 16 + threadIdx.x; i < total; i += kPackThreads) lds[i] = asrc[i];
    __syncthreads();
    if (r >= R) return;
    const uint8_t *mine = lds + mis + (uint32_t)(threadIdx.x * stride);
    for (uint32_t w = 0; w < words; ++w) {
      uint32_t acc = 0;
      const uint32_t kmax = m - 16 * w < 16 ? m - 16 * w : 16;
      for (uint32_t k = 0; k < kmax; ++k) acc |= (uint32_t)(mine[16 * w + k] & 3u) << (2 * k);
      packed[(uint64_t)w * R + r] = acc;
    }
  } else {
    if (r >= R) return;
    const uint8_t *mine = pats + r * stride;
    for (uint32_t w = 0; w < words; ++w) {
      uint32_t acc = 0;
      const uint32_t kmax = m - 16 * w < 16 ? m - 16 * w : 16;
      for (uint32_t k = 0; k < kmax; ++k) acc |= (uint32_t)(mine[16 * w + k] & 3u) << (2 * k);
      packed[(uint64_t)w * R + r] = acc;
    }
  }
}

struct CTable {
  uint32_t c[5];
};

// Backward search (loc_search, seqindex.c:280-309), one read per lane.
// d_locrow (optional) receives the row locate() would hand to unc_sa
// (seqindex.c:265-276): sp unless the interval emptied before the last step.
//
// VERIFY (the index keeps SA, ISA and the text): once the interval is a single
// row its remaining steps are decided by the text -- the row's suffix starts
// at SA[sp], so the bases still to be prepended either equal the text before
// that position (then every further interval is the single row ISA[pos - k])
// or differ first at some base, where one ordinary step from the row reached
// so far gives the empty interval loc_search stops with.  Same outputs, a
// handful of reads instead of two rank queries per remaining base.
//
// COUNT adds up what the batch loaded (see bwtgpu_search_profile): counters[0]
// 16-byte rank blocks, [1] 16-byte text windows, [2] 4-byte SA/ISA entries,
// [3] 4-byte pattern words, [4] 8-byte k-mer table entries.
//
// FUSED: the reads are packed inside the kernel instead of by k_pack_reads: a
// workgroup stages the bytes of its 256 reads in LDS, packs them into LDS
// words and searches from there, so the packed form never goes to HBM.  Used
// when bytes and words of a workgroup's reads (256, 128 or 64 of them) fit
// kFusedLdsBytes.
// WIDE: the position records are the wide kind (kWidePosPerRecord).
template <bool VERIFY, bool COUNT, bool FUSED, bool WIDE = false>
__global__ void __launch_bounds__(256)
k_search(IndexView v, CTable C, const uint32_t *__restrict__ packed, const uint8_t *__restrict__ pats,
         uint64_t stride, uint64_t R, uint32_t m, uint32_t *__restrict__ d_sp,
         uint32_t *__restrict__ d_ep, uint32_t *__restrict__ d_locrow, uint32_t *__restrict__ d_pos,
         unsigned long long *__restrict__ counters, uint64_t wstride, uint64_t rstride, uint32_t rounds) {
  // !FUSED: word w of read r is packed[w * wstride + r * rstride] -- (R, 1) for the
  // word-major form k_pack_reads writes, (1, words per read) for reads packed
  // one after the other (bwtgpu_search_packed)
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t t = threadIdx.x;
  const uint64_t r0 = (uint64_t)blockIdx.x * blockDim.x;
  const uint64_t r = r0 + t;
  const uint32_t *lw = nullptr;  // FUSED: word w of this lane's read at lw[w * blockDim.x]
  const uint32_t nt = blockDim.x;  // reads per workgroup: 256, fewer for long reads in the fused form
  if (FUSED) {
    // The bytes of the workgroup's reads go through LDS in `rounds` parts (nt / rounds reads
    // each, all threads copying, the part's own threads packing), so that the byte area is
    // a fraction of the reads: LDS per workgroup decides how many waves a CU holds.
    const uint32_t words = (m + 15) / 16;
    uint32_t *wbase = reinterpret_cast<uint32_t *>(lds);  // [words][nt], then the staged bytes
    uint8_t *bytes = lds + (size_t)words * nt * 4;
    const uint32_t part = nt / rounds;
    for (uint32_t q = 0; q < rounds; ++q) {
      const uint64_t p0 = r0 + (uint64_t)q * part;  // first read of the part
      if (q) __syncthreads();                       // the part before has been packed
      if (p0 < R) {
        const uint8_t *src = pats + p0 * stride;
        const uint64_t last = R - 1 < p0 + part - 1 ? R - 1 : p0 + part - 1;
        const uint32_t span = (uint32_t)((last - p0) * stride + m);
        const uint32_t mis = (uint32_t)((uintptr_t)src & 15u);
        const uint8_t *asrc = src - mis;  // see k_pack_reads
        const uint32_t total = mis + span, vecs = total / 16;
        for (uint32_t i = t; i < vecs; i += nt)
          reinterpret_cast<u32x4 *>(bytes)[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(asrc) + i);
        for (uint32_t i = vecs * 16 + t; i < total; i += nt) bytes[i] = asrc[i];
        __syncthreads();
        if (r < R && t / part == q) {
          const uint8_t *mine = bytes + mis + (uint32_t)((t - q * part) * stride);
          for (uint32_t w = 0; w < words; ++w) {
            uint32_t acc = 0;
            const uint32_t kmax = m - 16 * w < 16 ? m - 16 * w : 16;
            for (uint32_t k = 0; k < kmax; ++k) acc |= (uint32_t)(mine[16 * w + k] & 3u) << (2 * k);
            wbase[w * nt + t] = acc;
          }
        }
      }
    }
    lw = wbase + t;  // only this lane reads its own words: no barrier needed
  }
  auto word_at = [&](uint32_t w) -> uint32_t { return FUSED ? lw[w * nt] : packed[(uint64_t)w * wstride + r * rstride]; };
  uint32_t n_blk = 0, n_txt = 0, n_ent = 0, n_pat = 1, n_kmr = 0;
  if (r < R) {
  int i = -2;
  uint32_t word = 0, c, sp = 0, ep = 0;
  uint32_t table_pos = 0xFFFFFFFFu;  // SA[sp] when the k-mer table supplied it (a two-word table, a single row)
  const uint32_t K = v.kmer_k;
  if (K && m >= K) {
    // the last K bases as a base-4 number, first base most significant
    const uint32_t a = m - K, wa = a >> 4;
    const uint32_t w0 = word_at((uint32_t)(wa));
    const uint32_t w1 = wa < ((m - 1) >> 4) ? word_at((uint32_t)(wa + 1)) : 0u;
    if (COUNT) n_pat += wa < ((m - 1) >> 4);
    uint32_t x = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (2 * (a & 15u)));  // base a in the low bits
    x = __brev(x);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);  // base a in the top bits
    uint32_t e_sp, e_pos;
    bool e_next;
    const uint32_t e_ep = kmer_interval(v, x >> (32 - 2 * K), e_sp, e_pos, e_next);
    if (COUNT) {
      ++n_kmr;  // 8 bytes: two starts, or a start and the position word
      n_ent += e_next ? 1u : 0u;
    }
    if (e_ep > e_sp) {
      sp = e_sp;
      ep = e_ep;
      word = w0;
      i = (int)a - 1;
      if (e_ep - e_sp == 1) {
        table_pos = e_pos;  // the one occurrence of the string: SA[sp] without the lookup
      } else if (VERIFY && i >= 0 && v.kmer_w == 2) {
        // a few rows, and the entry holds their BWT symbols: if exactly one row goes on with the
        // read's next base, the search is down to that row's suffix without a rank query, and the
        // comparison with the text in front of it (which begins with that base) decides the rest
        const uint32_t wi = (uint32_t)i >> 4;
        const uint32_t cw = wi == wa ? w0 : word_at(wi);
        const uint32_t rows = kmer_rows_with(e_pos, e_ep - e_sp, v.kmer_sizebits, (cw >> (2 * (i & 15))) & 3u);
        if (rows && !(rows & (rows - 1u))) {
          sp = e_sp + (uint32_t)__builtin_ctz(rows) / 2u;
          ep = sp + 1;
        }
      }
    } else if (v.kmer2) {
      // the last K bases do not occur; if the last K - 2 do, the interval empties on one of the two
      // bases in front of them, and the steps from that level's interval are the ones loc_search takes
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 iv = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(v.kmer2) +
                                                  ((x >> (32 - 2 * K)) & ((1u << (2 * (K - 2))) - 1u)));
      if (COUNT) n_ent += 2;
      if (iv.y > iv.x) {
        sp = iv.x;
        ep = iv.y;
        i = (int)a + 1;
        word = (((uint32_t)i >> 4) == wa) ? w0 : w1;
      } else if (v.kmer4) {  // nor do the last K - 2: the level two bases further up
        const u32x2 iv4 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(v.kmer4) +
                                                     ((x >> (32 - 2 * K)) & ((1u << (2 * (K - 4))) - 1u)));
        if (COUNT) n_ent += 2;
        if (iv4.y > iv4.x) {
          sp = iv4.x;
          ep = iv4.y;
          i = (int)a + 3;
          word = (((uint32_t)i >> 4) == wa) ? w0 : w1;
        }
      }
    }
  }
  if (i == -2) {  // no table, a short read, or a string the text does not hold: step by step
    i = (int)m - 1;
    word = word_at((uint32_t)(i >> 4));
    c = (word >> (2 * (i & 15))) & 3u;
    sp = c == 0 ? C.c[0] : c == 1 ? C.c[1] : c == 2 ? C.c[2] : C.c[3];
    ep = c == 0 ? C.c[1] : c == 1 ? C.c[2] : c == 2 ? C.c[3] : C.c[4];
    i = (int)m - 2;
  }
  bool early = false;
  // d_pos: the text position of row sp, when the search came by it anyway -- the table's
  // position word while no step has moved sp, the place the read was compared at
  uint32_t known_pos = 0xFFFFFFFFu;
  if (VERIFY && i < 0) known_pos = table_pos;
  const uint4 *sym = reinterpret_cast<const uint4 *>(v.sym);
  for (; i >= 0; --i) {
    if ((i & 15) == 15) {
      word = word_at((uint32_t)(i >> 4));
      if (COUNT) ++n_pat;
    }
    if (ep <= sp) {  // seqindex.c:299-301
      early = true;
      break;
    }
    if (VERIFY && ep - sp == 1) break;
    c = (word >> (2 * (i & 15))) & 3u;
    // rank() first steps over the unstored $ row (seqindex.c:222-223)
    const uint32_t js = sp - (sp > v.endloc), je = ep - (ep > v.endloc);
    uint32_t rs, re;
    const uint32_t bs = div96(js, rs), be = div96(je, re);
    const uint4 *col = sym + (uint64_t)c * v.nsym;
    const uint4 blk_s = col[bs];
    uint4 blk_e = blk_s;
    if (be != bs) blk_e = col[be];  // narrow intervals share the block
    if (COUNT) n_blk += 1 + (be != bs);
    sp = sym_count(blk_s, rs);
    ep = sym_count(blk_e, re);
  }
  if (VERIFY && !early && i >= 0 && ep - sp == 1) {
    const uint32_t rem = (uint32_t)i + 1;  // read bases 0..i are still to be prepended
    // no rank step has moved sp since the table lookup exactly when the table's interval was the single row
    const uint32_t pos = table_pos != 0xFFFFFFFFu ? table_pos : __builtin_nontemporal_load(v.sa + sp);
    if (COUNT) n_ent += table_pos != 0xFFFFFFFFu ? 1 : 2;  // the SA entry unless the table had it, and the one ISA entry read below
    const uint32_t jmin = pos < rem ? rem - pos : 0u;  // bases below jmin would lie before the text
    // the highest read base j in [jmin, rem) that differs from text[pos - rem + j];
    // the record of position pos - rem + GB g holds the text for read words GW g .. GW g + GW - 1
    // (GB = 48 bases and GW = 3 words; 112 and 7 for wide records)
    constexpr uint32_t GW = WIDE ? 7u : 3u, GB = 16u * GW;
    int jbad = -1;
    uint32_t row_all = 0;  // ISA[pos - rem]: in the record read last
    bool before_text = false;
    for (int g = (int)((rem - 1) / GB); g >= 0 && jbad < 0 && !before_text; --g) {
      if (GB * g + GB <= jmin) break;
      uint32_t o;
      uint64_t b;  // guard records included: they hold GB positions
      uint32_t tx[GW + 1];
      if (WIDE) {
        const uint32_t at0 = (pos + GB) - (rem - GB * g);
        o = at0 & 7u;
        b = at0 >> 3;
        const uint4 ta = reinterpret_cast<const uint4 *>(v.isa)[b * 4u + 2u];
        const uint4 tb = reinterpret_cast<const uint4 *>(v.isa)[b * 4u + 3u];
        tx[0] = ta.x; tx[1] = ta.y; tx[2] = ta.z; tx[3] = ta.w;
        tx[4] = tb.x; tx[5] = tb.y; tx[6] = tb.z; tx[GW] = tb.w;
      } else {
        b = pos_record_place(div12((pos + GB) - (rem - GB * g), o));
        const uint4 ta = reinterpret_cast<const uint4 *>(v.isa)[b * 4u + 3u];
        tx[0] = ta.x; tx[1] = ta.y; tx[2] = ta.z; tx[GW] = ta.w;
      }
      if (g == 0 && jmin == 0) row_all = v.isa[b * 16u + o];
      if (COUNT) ++n_txt;
#pragma unroll
      for (int wl = (int)GW - 1; wl >= 0; --wl) {
        if (jbad >= 0 || before_text) continue;
        const uint32_t w = GW * g + wl;
        if (16u * w >= rem) continue;
        if (16u * w + 16u <= jmin) {
          before_text = true;
          continue;
        }
        const uint32_t lo = jmin > 16u * w ? jmin - 16u * w : 0u;
        const uint32_t hi = rem - 16u * w < 16u ? rem - 16u * w : 16u;
        const uint32_t rw = word_at(w);
        if (COUNT) ++n_pat;
        // 16 bases from base o + 16 wl of the record's text: o < 16, so they start in word wl
        const uint32_t tw = (uint32_t)((((uint64_t)tx[wl + 1] << 32) | tx[wl]) >> (2u * o));
        uint32_t d = rw ^ tw;
        d = (d | (d >> 1)) & 0x55555555u;
        const uint32_t keep = (hi == 16 ? ~0u : ((1u << (2 * hi)) - 1u)) & ~((1u << (2 * lo)) - 1u);
        d &= keep;
        if (d) jbad = 16 * (int)w + (31 - __clz((int)d)) / 2;
      }
    }
    if (jbad < 0 && jmin > 0) jbad = (int)jmin - 1;  // ran off the start of the text
    if (jbad < 0) {
      sp = row_all;
      ep = sp + 1;
      known_pos = pos - rem;  // row_all = ISA[pos - rem]
    } else {
      const uint32_t k = rem - 1 - (uint32_t)jbad;  // bases that did match
      const uint32_t row = k ? isa_at(v, pos - k) : sp;
      c = (word_at((uint32_t)(jbad >> 4)) >> (2 * (jbad & 15))) & 3u;
      const uint32_t js = row - (row > v.endloc), je = row + 1 - (row + 1 > v.endloc);
      uint32_t rs, re;
      const uint32_t bs = div96(js, rs), be = div96(je, re);
      const uint4 *col = sym + (uint64_t)c * v.nsym;
      const uint4 blk_s = col[bs];
      uint4 blk_e = blk_s;
      if (be != bs) blk_e = col[be];
      if (COUNT) n_blk += 1 + (be != bs);
      sp = sym_count(blk_s, rs);
      ep = sym_count(blk_e, re);
      early = jbad > 0;  // a further base remained, so loc_search sees the empty interval and stops
    }
  }
  // results are written once and read by nobody here: streamed past the caches
  __builtin_nontemporal_store(sp, d_sp + r);
  __builtin_nontemporal_store(ep, d_ep + r);
  if (d_locrow) __builtin_nontemporal_store(early ? 0xFFFFFFFFu : (sp > v.n ? 0xFFFFFFFEu : sp), d_locrow + r);
  if (VERIFY && d_pos) {  // locate() of the read (seqindex.c:258-277): SA[sp] unless the interval emptied early
    uint32_t p = 0xFFFFFFFFu;
    if (!early && sp <= v.n) p = known_pos != 0xFFFFFFFFu ? known_pos : v.sa[sp];
    __builtin_nontemporal_store(p, d_pos + r);
  }
  }
  if (COUNT) {
    if (r >= R) n_pat = 0;
    unsigned long long t[5] = {n_blk, n_txt, n_ent, n_pat, n_kmr};
    for (int q = 0; q < 5; ++q) {
      for (int off = 32; off; off >>= 1) t[q] += __shfl_down(t[q], off, 64);
      if ((threadIdx.x & 63u) == 0 && t[q]) atomicAdd(counters + q, t[q]);
    }
  }
}

// locate when the index keeps the whole suffix array: one lookup per row
__global__ void k_locate_sa(IndexView v, const uint32_t *__restrict__ rows, uint64_t R,
                            uint32_t *__restrict__ pos) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const uint32_t row = rows[r];
  pos[r] = row > v.n ? 0xFFFFFFFFu : v.sa[row];
}

// unc_sa (seqindex.c:244-255) for a handful of rows (the reference's scalar
// calls are batches of one): a row per lane, no queue to set up.
__global__ void __launch_bounds__(256)
k_locate_few(IndexView v, const uint32_t *__restrict__ rows, uint64_t R, uint32_t *__restrict__ pos) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  uint32_t idx = rows[r], steps = 0;
  if (idx > v.n) {
    pos[r] = 0xFFFFFFFFu;
    return;
  }
  while (idx & kDevSampleMask) {
    idx = lf_step(v, idx);
    if (++steps > v.n || idx > v.n) {  // not the LF map of a text
      pos[r] = 0xFFFFFFFFu;
      return;
    }
  }
  uint64_t x = (uint64_t)v.idxs[idx >> kDevSampleShift] + steps;
  if (x > v.n) x -= (uint64_t)v.n + 1;
  pos[r] = (uint32_t)x;
}

constexpr uint32_t kLocateFew = 4096;   // up to this many rows skip the queue
constexpr uint32_t kLocateBatch = 256;  // rows a wave takes from the queue at a time
constexpr int kLocateSlots = 2;         // independent walks per lane

// unc_sa (seqindex.c:244-255) for a batch of rows.  The walk length is
// geometric (a row is sampled when row % 8 == 0 in HBM), so lanes do not run one row
// each to the end: a slot that finishes takes the next row from its wave's
// share of a global queue, which keeps every lane walking.  Each lane runs
// kLocateSlots walks at once: a walk is a chain of dependent 32-byte reads, and
// independent chains are what hides their latency.
__global__ void __launch_bounds__(256)
k_locate(IndexView v, const uint32_t *__restrict__ rows, uint64_t R, uint32_t *__restrict__ pos,
         unsigned long long *__restrict__ steps_total, unsigned long long *__restrict__ queue_head) {
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t loc_lo = 0, loc_hi = 0;  // wave-uniform: rows this wave holds
  bool exhausted = false;           // wave-uniform: the queue has run dry
  bool active[kLocateSlots];
  uint32_t idx[kLocateSlots], steps[kLocateSlots];
  uint64_t mine[kLocateSlots];
#pragma unroll
  for (int s = 0; s < kLocateSlots; ++s) {
    active[s] = false;
    idx[s] = 0;
    steps[s] = 0;
    mine[s] = 0;
  }
  unsigned long long my_steps = 0;
  for (;;) {
    uint64_t need[kLocateSlots];
    uint32_t want = 0;
#pragma unroll
    for (int s = 0; s < kLocateSlots; ++s) {
      need[s] = __ballot(!active[s]);
      want += (uint32_t)__popcll(need[s]);
    }
    if (want) {
      if (loc_lo >= loc_hi && !exhausted) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(queue_head, (unsigned long long)kLocateBatch);
        base = __shfl(base, 0, 64);
        if (base >= R) {
          exhausted = true;
          loc_lo = loc_hi = 0;
        } else {
          loc_lo = base;
          loc_hi = base + kLocateBatch < R ? base + kLocateBatch : R;
        }
      }
      uint64_t first = loc_lo;
#pragma unroll
      for (int s = 0; s < kLocateSlots; ++s) {
        if (!active[s]) {
          const uint64_t cand = first + (uint64_t)__popcll(need[s] & low_mask(lane));
          if (cand < loc_hi) {
            const uint32_t row = rows ? rows[cand] : (uint32_t)cand;  // no list: every row in turn
            if (row > v.n) {
              pos[cand] = 0xFFFFFFFFu;
            } else {
              mine[s] = cand;
              idx[s] = row;
              steps[s] = 0;
              active[s] = true;
            }
          }
        }
        first += (uint64_t)__popcll(need[s]);
      }
      loc_lo = first < loc_hi ? first : loc_hi;
    }
    bool any = false;
#pragma unroll
    for (int s = 0; s < kLocateSlots; ++s) any = any || active[s];
    if (!__any(any)) {
      if (exhausted && loc_lo >= loc_hi) break;
      continue;
    }
    // walks that reached a sampled row finish; the others take one LF step,
    // all loads of the lane issued before any result is used
    bool step[kLocateSlots];
    uint32_t nxt[kLocateSlots];
#pragma unroll
    for (int s = 0; s < kLocateSlots; ++s) {
      step[s] = active[s] && (idx[s] & kDevSampleMask);
      nxt[s] = lf_step(v, step[s] ? idx[s] : 0u);
    }
#pragma unroll
    for (int s = 0; s < kLocateSlots; ++s) {
      if (active[s] && !step[s]) {
        uint64_t x = (uint64_t)v.idxs[idx[s] >> kDevSampleShift] + steps[s];
        if (x > v.n) x -= (uint64_t)v.n + 1;  // seqindex.c:252-253
        pos[mine[s]] = (uint32_t)x;
        my_steps += steps[s];
        active[s] = false;
      } else if (step[s]) {
        idx[s] = nxt[s];
        if (++steps[s] > v.n || nxt[s] > v.n) {  // not the LF map of a text
          pos[mine[s]] = 0xFFFFFFFFu;
          active[s] = false;
        }
      }
    }
  }
  if (steps_total) {
    for (int off = 32; off; off >>= 1) my_steps += __shfl_down(my_steps, off, 64);
    if (lane == 0 && my_steps) atomicAdd(steps_total, my_steps);
  }
}

// Walks along LF from sampled rows of the suffix array.  Row r with SA[r] = p has
// BWT symbol text[p - 1], and LF(r) is the row of suffix p - 1 (seqindex.c:214-219,
// the step unc_sa repeats, seqindex.c:244-255): walking LF from a sampled row
// names row and text position of one suffix after the other, until the next
// sampled row, where another walk has started.  The sampled rows cut the one
// cycle of LF into chains and every row lies on exactly one, so a row costs one
// LF step (walking from every row to its next sample costs half the sample
// distance).  Chain lengths are geometric: lanes take chains from a device-wide
// queue as they finish, two chains per lane in flight (cf. k_locate).
//
// DENSE = false: suffix array and text of an index that holds every 8th row
//   (`samples`): out = sa[n + 1]; a chain's text symbols are gathered into the
//   word they share and joined to text2 (zeroed beforehand) by one atomic per
//   word and chain.
// DENSE = true: every 8th row from every 32nd (`samples`, what an index file
//   stores, fileio.c:14): out = the dense samples.
// report[0] += entries of `out` written, report[1] += chains given up (not the LF map of a text) or
// arriving at a sampled row with another position than the one stored for it.
template <bool DENSE>
__global__ void __launch_bounds__(256)
k_chain_walk(IndexView v, const uint32_t *__restrict__ samples, uint64_t chains, uint32_t *__restrict__ out,
             uint32_t *__restrict__ text2, unsigned long long *__restrict__ queue_head,
             unsigned long long *__restrict__ report) {
  constexpr uint32_t kShift = DENSE ? kFileSampleShift : kDevSampleShift;  // rows between chain starts
  constexpr uint32_t kMask = (1u << kShift) - 1u;
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t loc_lo = 0, loc_hi = 0;  // wave-uniform: chains this wave holds
  bool exhausted = false;
  bool active[kLocateSlots];
  uint32_t idx[kLocateSlots], pos[kLocateSlots], steps[kLocateSlots], tw[kLocateSlots], twi[kLocateSlots];
#pragma unroll
  for (int s = 0; s < kLocateSlots; ++s) {
    active[s] = false;
    idx[s] = pos[s] = steps[s] = tw[s] = 0;
    twi[s] = 0xFFFFFFFFu;
  }
  unsigned long long written = 0, given_up = 0;
  for (;;) {
    uint64_t need[kLocateSlots];
    uint32_t want = 0;
#pragma unroll
    for (int s = 0; s < kLocateSlots; ++s) {
      need[s] = __ballot(!active[s]);
      want += (uint32_t)__popcll(need[s]);
    }
    if (want) {
      if (loc_lo >= loc_hi && !exhausted) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(queue_head, (unsigned long long)kLocateBatch);
        base = __shfl(base, 0, 64);
        if (base >= chains) {
          exhausted = true;
          loc_lo = loc_hi = 0;
        } else {
          loc_lo = base;
          loc_hi = base + kLocateBatch < chains ? base + kLocateBatch : chains;
        }
      }
      uint64_t first = loc_lo;
#pragma unroll
      for (int s = 0; s < kLocateSlots; ++s) {
        if (!active[s]) {
          const uint64_t cand = first + (uint64_t)__popcll(need[s] & low_mask(lane));
          if (cand < loc_hi) {
            const uint32_t row = (uint32_t)(cand << kShift), p = samples[cand];
            out[DENSE ? row >> kDevSampleShift : row] = p;
            ++written;
            idx[s] = row;
            pos[s] = p;
            steps[s] = 0;
            tw[s] = 0;
            twi[s] = 0xFFFFFFFFu;
            active[s] = true;
          }
        }
        first += (uint64_t)__popcll(need[s]);
      }
      loc_lo = first < loc_hi ? first : loc_hi;
    }
    bool any = false;
#pragma unroll
    for (int s = 0; s < kLocateSlots; ++s) any = any || active[s];
    if (!__any(any)) {
      if (exhausted && loc_lo >= loc_hi) break;
      continue;
    }
    uint32_t nxt[kLocateSlots], sym[kLocateSlots];
#pragma unroll
    for (int s = 0; s < kLocateSlots; ++s) nxt[s] = lf_step(v, active[s] ? idx[s] : 0u, sym[s]);
#pragma unroll
    for (int s = 0; s < kLocateSlots; ++s) {
      if (!active[s]) continue;
      bool done = false;
      if (idx[s] == v.endloc) {
        done = true;  // suffix 0: nothing in front of it, and LF leads to row 0, a sampled row
        if (pos[s] != 0) ++given_up;
      } else if (pos[s] == 0 || pos[s] > v.n || nxt[s] > v.n || ++steps[s] > v.n) {
        done = true;  // not the LF map of a text
        ++given_up;
      } else {
        const uint32_t q = pos[s] - 1u;
        if (!DENSE) {
          const uint32_t wi = q >> 4;
          if (wi != twi[s]) {
            if (tw[s]) atomicOr(&text2[1u + twi[s]], tw[s]);
            tw[s] = 0;
            twi[s] = wi;
          }
          tw[s] |= sym[s] << (2u * (q & 15u));
        }
        idx[s] = nxt[s];
        pos[s] = q;
        if ((nxt[s] & kMask) == 0) {
          done = true;  // the chain of that sampled row goes on from here, and from this position
          if (samples[nxt[s] >> kShift] != q) ++given_up;
        } else if (!DENSE) {
          out[nxt[s]] = q;
          ++written;
        } else if ((nxt[s] & kDevSampleMask) == 0) {
          out[nxt[s] >> kDevSampleShift] = q;
          ++written;
        }
      }
      if (done) {
        if (!DENSE && tw[s]) atomicOr(&text2[1u + twi[s]], tw[s]);
        tw[s] = 0;
        active[s] = false;
      }
    }
  }
  for (int off = 32; off; off >>= 1) {
    written += __shfl_down(written, off, 64);
    given_up += __shfl_down(given_up, off, 64);
  }
  if (lane == 0) {
    if (written) atomicAdd(report, written);
    if (given_up) atomicAdd(report + 1, given_up);
  }
}

// Runs the walks and waits for them.  *consistent: every one of the `expect` entries was
// written exactly once and no chain was given up.
int chain_walk(const IndexView &v, bool dense, const uint32_t *d_samples, uint64_t chains, uint32_t *d_out,
               uint32_t *d_text2, uint64_t expect, bool *consistent, hipStream_t s) {
  unsigned long long *head = nullptr;  // [0] the queue, [1] entries written, [2] chains given up
  *consistent = false;
  BWT_HIP(pool_alloc(&head, 256));
  unsigned long long report[2] = {0, 0};
  hipError_t e = hipMemsetAsync(head, 0, 256, s);
  if (e == hipSuccess) {
    uint64_t groups = div_up(div_up(chains, (uint64_t)kLocateBatch), 4);
    if (groups > 256 * 8) groups = 256 * 8;
    if (dense)
      BWT_LAUNCH("densify_samples", k_chain_walk<true>, dim3((unsigned)groups), dim3(256), 0, s, v, d_samples, chains,
                 d_out, d_text2, head, head + 1);
    else
      BWT_LAUNCH("recover_sa_text", k_chain_walk<false>, dim3((unsigned)groups), dim3(256), 0, s, v, d_samples,
                 chains, d_out, d_text2, head, head + 1);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(report, head + 1, sizeof report, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) (void)hipDeviceSynchronize();
  pool_free(head);
  if (e != hipSuccess) return set_error(kErrHip, "walking LF from the sampled rows: %s", hipGetErrorString(e));
  *consistent = report[0] == expect && report[1] == 0;
  return kOk;
}

// An index loaded from its parts (BWT, sampled rows) has everything from which
// the arrays a built index keeps can be recomputed: the suffix array and the
// text, by walking LF from the sampled rows.  With them the loaded index gets
// position records and a two-word k-mer table like a built one.  Not an error
// when HBM is short, or when the walks do not cover every row exactly once (the
// parts are not those of one text): the index then searches on its rank blocks.
// Waits for `s`.
int index_recover_arrays(bwtgpu_index *ix, hipStream_t s) {
  if (ix->d_sa || !ix->has_samples) return kOk;
  const uint32_t n = ix->n;
  const uint64_t rows = (uint64_t)n + 1, t2words = (uint64_t)n / 16 + 4;
  // filled in buffers of our own: the index gets them when the walks have covered every row
  uint32_t *sa = nullptr, *text2 = nullptr;
  auto drop = [&] {
    pool_free(sa);
    pool_free(text2);
  };
  hipError_t e = pool_alloc(&sa, rows * 4);
  if (e == hipSuccess) e = pool_alloc(&text2, t2words * 4);
  if (e == hipSuccess) e = hipMemsetAsync(text2, 0, t2words * 4, s);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    drop();
    return kOk;
  }
  IndexView v = view_of(ix);
  v.sa = nullptr;  // being written
  bool consistent = false;
  const int rc = chain_walk(v, false, ix->d_idxs, ((uint64_t)n >> kDevSampleShift) + 1, sa, text2, rows,
                            &consistent, s);  // waits for the stream
  if (rc != kOk || !consistent) {
    drop();  // inconsistent parts: stay on the rank blocks
    return rc;
  }
  ix->d_text2 = text2;
  publish(ix->d_sa, sa);
  return rc;
}

// Scratch of bwtgpu_search_static: the packed reads of a batch whose reads are
// too long to be packed inside k_search.  One copy per GPU, held by the code
// object until the process ends.  The name below is only the host's handle on
// the symbol: the address kernels get is the one hipGetSymbolAddress returns
// for the current device (static_scratch_setup), kept in g_scratch_ptr.
constexpr size_t kStaticScratchBytes = 16u << 20;
static __device__ __align__(256) uint8_t scratch[kStaticScratchBytes];
constexpr int kMaxDevices = 64;
static void *g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mu;

int static_scratch_setup(int device, void **out) {
  if (device < 0 || device >= kMaxDevices) return set_error(kErrArg, "device %d out of range", device);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[device]) {
    if (out) *out = g_scratch_ptr[device];
    return kOk;
  }
  int cur = -1;
  BWT_HIP(hipGetDevice(&cur));
  if (cur != device) return set_error(kErrArg, "device %d is not the current device (%d)", device, cur);
  void *p = nullptr;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !p)
    return set_error(kErrHip, "address of the static search scratch on device %d: %s", device,
                     hipGetErrorString(e));
  g_scratch_ptr[device] = p;
  if (out) *out = p;
  return kOk;
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

size_t bwtgpu_search_workspace_bytes(uint64_t R, uint32_t m) {
  return align_up((size_t)((m + 15) / 16) * R * sizeof(uint32_t), 256) + 256;
}

constexpr uint32_t kFusedLdsBytes = 40 * 1024;

// Reads per workgroup (256, 128 or 64) with which the bytes and packed words of
// a workgroup's reads fit kFusedLdsBytes, so that k_search packs them itself and
// the packed-read workspace is never touched; 0 when even 64 reads do not fit.
// *rounds: in how many parts the bytes pass through LDS.  One part while the workgroup then
// needs no more than kFusedLdsFull bytes (eight workgroups of 256 to a CU: every wave slot
// taken); else as many parts (up to 8) as it takes to get there, or at least under the limit.
constexpr uint32_t kFusedLdsFull = 20 * 1024;
static uint32_t fused_group_reads(uint32_t m, uint64_t stride, uint64_t R, uint64_t *lds_bytes, uint32_t *rounds) {
  for (uint32_t nt = 256; nt >= 64; nt /= 2) {
    if (div_up(R, nt) > 0x7FFFFFFFull) return 0;
    const uint64_t wordbytes = (uint64_t)((m + 15) / 16) * nt * 4;
    const uint64_t target = (uint64_t)kFusedLdsFull * nt / 256;
    uint32_t best = 0;
    uint64_t best_lds = 0;
    for (uint32_t q = 1; q <= 8 && q <= nt / 8; q *= 2) {
      const uint64_t bytes = align_up((uint64_t)(nt / q - 1) * stride + m + 16, 16);
      const uint64_t fused_lds = bytes + wordbytes;
      if (fused_lds > kFusedLdsBytes) continue;
      best = q;
      best_lds = fused_lds;
      if (fused_lds <= target) break;
    }
    if (!best) continue;
    *lds_bytes = best_lds;
    *rounds = best;
    return nt;
  }
  return 0;
}

// d_pos (optional): the position locate() returns for the read, written by the search
// kernel itself; only for an index that keeps its position records (ix->d_isa).
static int search_impl(const bwtgpu_index *ix, const IndexView &v, const uint8_t *d_pats, uint32_t m, uint64_t stride,
                       uint64_t R, uint32_t *d_sp, uint32_t *d_ep, uint32_t *d_locrow, uint32_t *d_pos,
                       void *d_ws, size_t ws_bytes, unsigned long long *d_counters, hipStream_t s) {
  if (!ix || !d_pats || !d_sp || !d_ep || !d_ws) return set_error(kErrArg, "null argument");
  if (m == 0 || m > 65535) return set_error(kErrArg, "read length %u out of range [1, 65535]", m);
  if (R == 0) return kOk;
  if (R > (1ull << 40)) return set_error(kErrArg, "too many reads");
  if (d_pos && !v.isa) return set_error(kErrArg, "positions from the search kernel need the position records");
  uint64_t fused_lds = 0;
  uint32_t fused_rounds = 1;
  const uint32_t fused_nt = d_counters ? 0u : fused_group_reads(m, stride, R, &fused_lds, &fused_rounds);
  if (!fused_nt && ws_bytes < bwtgpu_search_workspace_bytes(R, m))
    return set_error(kErrWorkspace, "search workspace %zu < %zu bytes", ws_bytes,
                     bwtgpu_search_workspace_bytes(R, m));
  if ((uintptr_t)d_ws & 255u) return set_error(kErrArg, "workspace must be 256-byte aligned");
  uint32_t *packed = reinterpret_cast<uint32_t *>(d_ws);
  const uint64_t groups = div_up(R, kPackThreads);
  if (groups > 0x7FFFFFFFull) return set_error(kErrArg, "too many reads for one launch");
  const uint64_t span = (uint64_t)(kPackThreads - 1) * stride + m + 16;
  CTable C;
  for (int i = 0; i < 5; ++i) C.c[i] = ix->C[i];
  const dim3 grid((unsigned)div_up(R, 256)), block(256);
  // bytes and packed words of a workgroup's reads in LDS: pack inside the search kernel.
  // Long reads get smaller workgroups so that they still fit.
  if (fused_nt) {
    const dim3 fgrid((unsigned)div_up(R, fused_nt)), fblock(fused_nt);
    if (v.isa && v.rec_wide)
      BWT_LAUNCH("fm_search", (k_search<true, false, true, true>), fgrid, fblock, (size_t)fused_lds, s, v, C, packed,
                 d_pats, stride, R, m, d_sp, d_ep, d_locrow, d_pos, d_counters, R, (uint64_t)1, fused_rounds);
    else if (v.isa)
      BWT_LAUNCH("fm_search", (k_search<true, false, true>), fgrid, fblock, (size_t)fused_lds, s, v, C, packed,
                 d_pats, stride, R, m, d_sp, d_ep, d_locrow, d_pos, d_counters, R, (uint64_t)1, fused_rounds);
    else
      BWT_LAUNCH("fm_search", (k_search<false, false, true>), fgrid, fblock, (size_t)fused_lds, s, v, C, packed,
                 d_pats, stride, R, m, d_sp, d_ep, d_locrow, d_pos, d_counters, R, (uint64_t)1, fused_rounds);
    BWT_CHECK_LAUNCH();
    return kOk;
  }
  const int use_lds = span <= kPackLdsBytes;
  BWT_LAUNCH("pack_reads", k_pack_reads, dim3((unsigned)groups), dim3(kPackThreads),
             use_lds ? (size_t)align_up(span, 16) : 0, s, d_pats, m, stride, R, packed, use_lds);
  BWT_CHECK_LAUNCH();
  if (d_counters) {
    if (v.isa && v.rec_wide)
      BWT_LAUNCH("fm_search_counting", (k_search<true, true, false, true>), grid, block, 0, s, v, C, packed, d_pats,
                 stride, R, m, d_sp, d_ep, d_locrow, d_pos, d_counters, R, (uint64_t)1, 1u);
    else if (v.isa)
      BWT_LAUNCH("fm_search_counting", (k_search<true, true, false>), grid, block, 0, s, v, C, packed, d_pats,
                 stride, R, m, d_sp, d_ep, d_locrow, d_pos, d_counters, R, (uint64_t)1, 1u);
    else
      BWT_LAUNCH("fm_search_counting", (k_search<false, true, false>), grid, block, 0, s, v, C, packed, d_pats,
                 stride, R, m, d_sp, d_ep, d_locrow, d_pos, d_counters, R, (uint64_t)1, 1u);
  } else if (v.isa && v.rec_wide) {
    BWT_LAUNCH("fm_search", (k_search<true, false, false, true>), grid, block, 0, s, v, C, packed, d_pats, stride, R, m,
               d_sp, d_ep, d_locrow, d_pos, d_counters, R, (uint64_t)1, 1u);
  } else if (v.isa) {
    BWT_LAUNCH("fm_search", (k_search<true, false, false>), grid, block, 0, s, v, C, packed, d_pats, stride, R, m,
               d_sp, d_ep, d_locrow, d_pos, d_counters, R, (uint64_t)1, 1u);
  } else {
    BWT_LAUNCH("fm_search", (k_search<false, false, false>), grid, block, 0, s, v, C, packed, d_pats, stride, R,
               m, d_sp, d_ep, d_locrow, d_pos, d_counters, R, (uint64_t)1, 1u);
  }
  BWT_CHECK_LAUNCH();
  return kOk;
}

int bwtgpu_search(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m, uint64_t stride,
                  uint64_t R, uint32_t *d_sp, uint32_t *d_ep, uint32_t *d_locrow, void *d_ws,
                  size_t ws_bytes, void *stream) {
  if (!ix) return set_error(kErrArg, "null argument");
  BWT_TRY(index_ready_for_search(ix, (hipStream_t)stream));
  return search_impl(ix, view_of(ix), d_pats, m, stride, R, d_sp, d_ep, d_locrow, nullptr, d_ws, ws_bytes, nullptr,
                     (hipStream_t)stream);
}

static int search_packed_impl(const bwtgpu_index *ix, const uint32_t *d_words, uint32_t m, uint64_t R,
                              uint32_t *d_sp, uint32_t *d_ep, uint32_t *d_locrow, uint32_t *d_pos, hipStream_t s) {
  CTable C;
  for (int i = 0; i < 5; ++i) C.c[i] = ix->C[i];
  const IndexView v = view_of(ix);
  const dim3 grid((unsigned)div_up(R, 256)), block(256);
  const uint64_t words = (m + 15) / 16;
  if (ix->d_isa && v.rec_wide)
    BWT_LAUNCH("fm_search", (k_search<true, false, false, true>), grid, block, 0, s, v, C, d_words,
               (const uint8_t *)nullptr, (uint64_t)0, R, m, d_sp, d_ep, d_locrow, d_pos,
               (unsigned long long *)nullptr, (uint64_t)1, words, 1u);
  else if (ix->d_isa)
    BWT_LAUNCH("fm_search", (k_search<true, false, false>), grid, block, 0, s, v, C, d_words, (const uint8_t *)nullptr,
               (uint64_t)0, R, m, d_sp, d_ep, d_locrow, d_pos, (unsigned long long *)nullptr, (uint64_t)1, words, 1u);
  else
    BWT_LAUNCH("fm_search", (k_search<false, false, false>), grid, block, 0, s, v, C, d_words,
               (const uint8_t *)nullptr, (uint64_t)0, R, m, d_sp, d_ep, d_locrow, d_pos,
               (unsigned long long *)nullptr, (uint64_t)1, words, 1u);
  BWT_CHECK_LAUNCH();
  return kOk;
}

static int packed_args_ok(const bwtgpu_index *ix, const uint32_t *d_words, uint32_t m, uint64_t R,
                          const uint32_t *d_sp, const uint32_t *d_ep) {
  if (!ix || !d_words || !d_sp || !d_ep) return set_error(kErrArg, "null argument");
  if (m == 0 || m > 65535) return set_error(kErrArg, "read length %u out of range [1, 65535]", m);
  if (div_up(R, 256) > 0x7FFFFFFFull) return set_error(kErrArg, "too many reads for one launch");
  return kOk;
}

int bwtgpu_search_packed(const bwtgpu_index *ix, const uint32_t *d_words, uint32_t m, uint64_t R, uint32_t *d_sp,
                         uint32_t *d_ep, uint32_t *d_locrow, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  BWT_TRY(packed_args_ok(ix, d_words, m, R, d_sp, d_ep));
  if (R == 0) return kOk;
  BWT_TRY(index_ready_for_search(ix, s));
  return search_packed_impl(ix, d_words, m, R, d_sp, d_ep, d_locrow, nullptr, s);
}

// Room for the rows of a batch (an index without position records locates in a second kernel)
// and the locate kernel's queue head.
static size_t locate_part_bytes(uint64_t R) { return align_up((size_t)R * sizeof(uint32_t), 256) + 256; }

size_t bwtgpu_search_locate_workspace_bytes(uint64_t R, uint32_t m) {
  return bwtgpu_search_workspace_bytes(R, m) + locate_part_bytes(R);
}

size_t bwtgpu_search_locate_packed_workspace_bytes(uint64_t R) { return locate_part_bytes(R); }

// the rows at the front of `part`, the queue head behind them
static int locate_from_rows(const bwtgpu_index *ix, uint8_t *part, uint64_t R, uint32_t *d_pos, hipStream_t s) {
  return bwtgpu_locate_rows(ix, reinterpret_cast<const uint32_t *>(part), R, d_pos, nullptr,
                            part + align_up((size_t)R * sizeof(uint32_t), 256), 256, (void *)s);
}

int bwtgpu_search_locate(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m, uint64_t stride, uint64_t R,
                         uint32_t *d_sp, uint32_t *d_ep, uint32_t *d_pos, void *d_ws, size_t ws_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ix || !d_pos) return set_error(kErrArg, "null argument");
  if (!d_ws || ((uintptr_t)d_ws & 255u)) return set_error(kErrArg, "workspace must be 256-byte aligned");
  if (R > (1ull << 40)) return set_error(kErrArg, "too many reads");
  BWT_TRY(index_ready_for_search(ix, s));
  if (ix->d_isa)  // the kernel knows the position of nearly every read by the time it has its interval
    return search_impl(ix, view_of(ix), d_pats, m, stride, R, d_sp, d_ep, nullptr, d_pos, d_ws, ws_bytes, nullptr, s);
  if (ws_bytes < bwtgpu_search_locate_workspace_bytes(R, m))
    return set_error(kErrWorkspace, "search-and-locate workspace %zu < %zu bytes", ws_bytes,
                     bwtgpu_search_locate_workspace_bytes(R, m));
  if (!ix->has_samples)
    return set_error(kErrArg, "this index was loaded without sampled suffix-array rows: it cannot locate");
  const size_t sws = bwtgpu_search_workspace_bytes(R, m);
  uint8_t *part = (uint8_t *)d_ws + sws;
  BWT_TRY(search_impl(ix, view_of(ix), d_pats, m, stride, R, d_sp, d_ep, reinterpret_cast<uint32_t *>(part), nullptr,
                      d_ws, sws, nullptr, s));
  if (R == 0) return kOk;
  return locate_from_rows(ix, part, R, d_pos, s);
}

int bwtgpu_search_locate_packed(const bwtgpu_index *ix, const uint32_t *d_words, uint32_t m, uint64_t R,
                                uint32_t *d_sp, uint32_t *d_ep, uint32_t *d_pos, void *d_ws, size_t ws_bytes,
                                void *stream) {
  hipStream_t s = (hipStream_t)stream;
  BWT_TRY(packed_args_ok(ix, d_words, m, R, d_sp, d_ep));
  if (!d_pos) return set_error(kErrArg, "null argument");
  if (R == 0) return kOk;
  BWT_TRY(index_ready_for_search(ix, s));
  if (ix->d_isa) return search_packed_impl(ix, d_words, m, R, d_sp, d_ep, nullptr, d_pos, s);
  if (!d_ws || ((uintptr_t)d_ws & 255u)) return set_error(kErrArg, "workspace must be 256-byte aligned");
  if (ws_bytes < locate_part_bytes(R))
    return set_error(kErrWorkspace, "search-and-locate workspace %zu < %zu bytes", ws_bytes, locate_part_bytes(R));
  if (!ix->has_samples)
    return set_error(kErrArg, "this index was loaded without sampled suffix-array rows: it cannot locate");
  uint8_t *part = (uint8_t *)d_ws;
  BWT_TRY(search_packed_impl(ix, d_words, m, R, d_sp, d_ep, reinterpret_cast<uint32_t *>(part), nullptr, s));
  return locate_from_rows(ix, part, R, d_pos, s);
}

int bwtgpu_search_static(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m, uint64_t stride,
                         uint64_t R, uint32_t *d_sp, uint32_t *d_ep, uint32_t *d_locrow, void *stream) {
  if (!ix) return set_error(kErrArg, "null argument");
  int dev = -1;
  BWT_HIP(hipGetDevice(&dev));
  if (dev != ix->device)
    return set_error(kErrArg, "the index lives on device %d, the current device is %d", ix->device, dev);
  // already resolved when bwtgpu_init ran on this device; the first call on a device does it otherwise
  void *d_scratch = nullptr;
  BWT_TRY(static_scratch_setup(dev, &d_scratch));
  // reads short enough to be packed inside the kernel use no scratch at all;
  // longer ones must fit their packed words into the static array
  uint64_t lds = 0;
  uint32_t rounds = 1;
  if (m && m <= 65535 && R && R <= (1ull << 40) && !fused_group_reads(m, stride, R, &lds, &rounds) &&
      bwtgpu_search_workspace_bytes(R, m) > kStaticScratchBytes)
    return set_error(kErrArg,
                     "%llu reads of %u bases need %zu bytes of scratch, bwtgpu_search_static has %zu: "
                     "split the batch or use bwtgpu_search with a workspace",
                     (unsigned long long)R, m, bwtgpu_search_workspace_bytes(R, m), kStaticScratchBytes);
  // never waits for, and never makes, the search-only arrays: it uses them once all are in place
  return search_impl(ix, view_published(ix), d_pats, m, stride, R, d_sp, d_ep, d_locrow, nullptr, d_scratch,
                     kStaticScratchBytes, nullptr, (hipStream_t)stream);
}

int bwtgpu_search_profile(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m, uint64_t stride,
                          uint64_t R, uint32_t *d_sp, uint32_t *d_ep, void *d_ws, size_t ws_bytes,
                          uint64_t loads[5], void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!loads) return set_error(kErrArg, "null argument");
  // the counters take the first 256 bytes of the workspace, the packed reads the rest
  if (ws_bytes < bwtgpu_search_workspace_bytes(R, m) + 256)
    return set_error(kErrWorkspace, "search profile workspace %zu < %zu bytes", ws_bytes,
                     bwtgpu_search_workspace_bytes(R, m) + 256);
  if (!d_ws || ((uintptr_t)d_ws & 255u)) return set_error(kErrArg, "workspace must be 256-byte aligned");
  unsigned long long *ctr = reinterpret_cast<unsigned long long *>(d_ws);
  if (!ix) return set_error(kErrArg, "null argument");
  BWT_TRY(index_ready_for_search(ix, s));
  BWT_HIP(hipMemsetAsync(ctr, 0, 256, s));
  BWT_TRY(search_impl(ix, view_of(ix), d_pats, m, stride, R, d_sp, d_ep, nullptr, nullptr, (uint8_t *)d_ws + 256,
                      ws_bytes - 256, ctr, s));
  unsigned long long h[5];
  BWT_HIP(hipMemcpyAsync(h, ctr, sizeof h, hipMemcpyDeviceToHost, s));
  BWT_HIP(hipStreamSynchronize(s));
  for (int q = 0; q < 5; ++q) loads[q] = h[q];
  return kOk;
}

size_t bwtgpu_locate_workspace_bytes(void) { return 256; }

int bwtgpu_locate_rows(const bwtgpu_index *ix, const uint32_t *d_rows, uint64_t R, uint32_t *d_pos,
                       unsigned long long *d_steps, void *d_ws, size_t ws_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ix || !d_rows || !d_pos || !d_ws) return set_error(kErrArg, "null argument");
  if (ws_bytes < bwtgpu_locate_workspace_bytes())
    return set_error(kErrWorkspace, "locate workspace %zu < %zu bytes", ws_bytes,
                     bwtgpu_locate_workspace_bytes());
  if ((uintptr_t)d_ws & 7u) return set_error(kErrArg, "workspace must be 8-byte aligned");
  if (R == 0) return kOk;
  // may run beside another thread's first search on this index: only what has been published
  const IndexView v = view_published(ix);
  if (!v.sa && !ix->has_samples)
    return set_error(kErrArg, "this index was loaded without sampled suffix-array rows: it cannot locate");
  if (v.sa) {
    BWT_LAUNCH("fm_locate", k_locate_sa, dim3((unsigned)div_up(R, 256)), dim3(256), 0, s, v, d_rows,
               R, d_pos);
    BWT_CHECK_LAUNCH();
    return kOk;
  }
  if (R <= kLocateFew && !d_steps) {
    BWT_LAUNCH("fm_locate", k_locate_few, dim3((unsigned)div_up(R, 256)), dim3(256), 0, s, v, d_rows, R,
               d_pos);
    BWT_CHECK_LAUNCH();
    return kOk;
  }
  unsigned long long *head = reinterpret_cast<unsigned long long *>(d_ws);
  BWT_HIP(hipMemsetAsync(head, 0, 256, s));
  // enough waves to fill the chip (256 CUs x 8 waves/SIMD x 4 SIMDs), no more
  // than there are batches of rows
  uint64_t waves = div_up(R, (uint64_t)kLocateBatch);
  uint64_t groups = div_up(waves, 4);
  if (groups > 256 * 8) groups = 256 * 8;
  BWT_LAUNCH("fm_locate", k_locate, dim3((unsigned)groups), dim3(256), 0, s, v, d_rows, R,
             d_pos, d_steps, head);
  BWT_CHECK_LAUNCH();
  return kOk;
}

}  // extern "C"

namespace bwtgpu {

int search_packed_dev(const bwtgpu_index *ix, const uint32_t *d_words, uint32_t m, uint64_t R, uint32_t *d_sp,
                      uint32_t *d_ep, uint32_t *d_locrow, uint32_t *d_pos, hipStream_t s) {
  BWT_TRY(packed_args_ok(ix, d_words, m, R, d_sp, d_ep));
  if (d_pos && !ix->d_isa) return set_error(kErrArg, "positions from the search kernel need the position records");
  if (R == 0) return kOk;
  return search_packed_impl(ix, d_words, m, R, d_sp, d_ep, d_locrow, d_pos, s);
}

}  // namespace bwtgpu
