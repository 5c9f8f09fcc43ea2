/*
 * CPU oracle: plain-C restatement of the reference's SA/BWT build and FM-index
 * search.  TEST INFRASTRUCTURE ONLY (see bwt_oracle.h for what may call it and
 * for the parity status: pinned against the compiled reference).
 */
#include "bwt_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* seeded inputs                                                       */
/* ------------------------------------------------------------------ */

uint64_t orc_rnd(uint64_t seed, uint64_t k) {
  uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

uint64_t orc_packed_bytes(uint64_t n) { return n / 4 + 1; }

static void mask_tail(uint8_t *text, uint64_t n) {
  /* base n and everything after it in the last byte read 0, which is what the
   * SACA-K path requires (csacak.c:518-520) and fmitest arranges
   * (fmitest.c:130-142) */
  static const uint8_t keep[4] = {0x00, 0xC0, 0xF0, 0xFC};
  text[n / 4] &= keep[n & 3];
}

void orc_gen_genome(uint8_t *text, uint64_t n, uint64_t seed) {
  uint64_t nb = orc_packed_bytes(n);
  for (uint64_t j = 0; j < nb; ++j)
    text[j] = (uint8_t)(orc_rnd(seed, j >> 3) >> (8 * (j & 7)));
  mask_tail(text, n);
}

static void setbase(uint8_t *s, uint64_t i, unsigned b) {
  unsigned sh = 2 * (3 - (i & 3));
  s[i >> 2] = (uint8_t)((s[i >> 2] & ~(3u << sh)) | (b << sh));
}

void orc_gen_tandem(uint8_t *text, uint64_t n, uint64_t period,
                    uint64_t unit_seed, uint64_t mut_every, uint64_t mut_seed) {
  uint64_t nb = orc_packed_bytes(n);
  uint8_t *unit = malloc(orc_packed_bytes(period));
  orc_gen_genome(unit, period, unit_seed);
  memset(text, 0, nb);
  for (uint64_t i = 0; i < n; ++i) {
    unsigned b = orc_getbase(unit, i % period);
    if (mut_every) {
      uint64_t r = orc_rnd(mut_seed, i);
      if (r % mut_every == 0) b = (b + 1 + (unsigned)((r >> 32) % 3)) & 3;
    }
    setbase(text, i, b);
  }
  free(unit);
}

void orc_gen_reads(const uint8_t *text, uint64_t n, uint8_t *reads, uint64_t R,
                   uint32_t m, uint64_t seed, uint32_t *offsets) {
  uint64_t words = (m + 31) / 32;
  for (uint64_t r = 0; r < R; ++r) {
    uint8_t *out = reads + r * m;
    if (orc_rnd(seed, 2 * r + 1) % 100 == 0) {
      for (uint32_t k = 0; k < m; ++k)
        out[k] = (uint8_t)((orc_rnd(seed ^ 0xABCDull, r * words + k / 32) >>
                            (2 * (k % 32))) & 3);
      if (offsets) offsets[r] = 0xFFFFFFFFu;
    } else {
      uint64_t j = orc_rnd(seed, 2 * r) % (n - m + 1);
      for (uint32_t k = 0; k < m; ++k) out[k] = (uint8_t)orc_getbase(text, j + k);
      if (offsets) offsets[r] = (uint32_t)j;
    }
  }
}

/* ------------------------------------------------------------------ */
/* suffix array: MSD 2-bit bucket sort (histsortcomp.c:35-180)         */
/* ------------------------------------------------------------------ */

typedef struct {
  uint32_t lo, hi, depth;
  int src_is_primary;
} frame_t;

typedef struct {
  const uint8_t *text;
  uint32_t n;
  uint32_t *primary, *secondary;
  uint32_t max_depth;
  frame_t first;
  int failed;
} hs_job;

/* One bucket-sort job: the reference's histhelper recursion
 * (histsortcomp.c:35-90) run off an explicit stack so that deep common
 * prefixes cost heap, not call stack. */
static void *hs_run(void *arg) {
  hs_job *job = arg;
  const uint8_t *text = job->text;
  const uint32_t n = job->n;
  size_t cap = 1024, top = 0;
  frame_t *stack = malloc(cap * sizeof *stack);
  stack[top++] = job->first;
  while (top) {
    frame_t f = stack[--top];
    uint32_t *src = f.src_is_primary ? job->primary : job->secondary;
    uint32_t *dst = f.src_is_primary ? job->secondary : job->primary;
    uint32_t lo = f.lo, hi = f.hi, d = f.depth;
    if (hi == lo) continue;
    if (hi - lo == 1) {
      /* histsortcomp.c:45-50: the result must land in the caller's array */
      if (!f.src_is_primary) job->primary[lo] = job->secondary[lo];
      continue;
    }
    if (job->max_depth && d > job->max_depth) { job->failed = 1; break; }
    uint32_t cnt[4] = {0, 0, 0, 0};
    for (uint32_t i = lo; i != hi; ++i) {
      if (src[i] + d != n) {
        cnt[orc_getbase(text, (uint64_t)src[i] + d)]++;
      } else {
        /* histsortcomp.c:58-64: the suffix that ends here goes first and is
         * final in both arrays */
        dst[lo] = src[i];
        src[i] = src[lo];
        src[lo] = dst[lo];
        ++lo;
      }
    }
    uint32_t ptr[4];
    ptr[0] = lo;
    ptr[1] = ptr[0] + cnt[0];
    ptr[2] = ptr[1] + cnt[1];
    ptr[3] = ptr[2] + cnt[2];
    for (uint32_t i = lo; i != hi; ++i) {
      unsigned b = orc_getbase(text, (uint64_t)src[i] + d);
      dst[ptr[b]++] = src[i];
    }
    if (top + 4 > cap) {
      cap *= 2;
      stack = realloc(stack, cap * sizeof *stack);
    }
    /* children swap the roles of the two arrays (histsortcomp.c:82-89);
     * pushed in reverse so bucket A is processed first */
    uint32_t s = lo;
    uint32_t b0 = s, b1 = b0 + cnt[0], b2 = b1 + cnt[1], b3 = b2 + cnt[2];
    frame_t c3 = {b3, hi, d + 1, !f.src_is_primary};
    frame_t c2 = {b2, b3, d + 1, !f.src_is_primary};
    frame_t c1 = {b1, b2, d + 1, !f.src_is_primary};
    frame_t c0 = {b0, b1, d + 1, !f.src_is_primary};
    stack[top++] = c3;
    stack[top++] = c2;
    stack[top++] = c1;
    stack[top++] = c0;
  }
  free(stack);
  return NULL;
}

uint32_t *orc_histsort(const uint8_t *text, uint32_t n, int threads,
                       uint32_t max_depth) {
  uint64_t rows = (uint64_t)n + 1;
  uint32_t *arr = malloc(rows * sizeof *arr);
  uint32_t *aux = malloc(rows * sizeof *aux);
  /* histsortcomp.c:165-168 */
  arr[0] = n;
  for (uint64_t i = 1; i <= n; ++i) arr[i] = (uint32_t)(i - 1);
  int failed = 0;
  if (threads < 4 || n < 4) {
    hs_job job = {text, n, arr, aux, max_depth, {0, n + 1, 0, 1}, 0};
    hs_run(&job);
    failed = job.failed;
  } else {
    /* histzero (histsortcomp.c:113-155): split on the first base, then one
     * thread per first-base bucket */
    uint32_t cnt[4] = {0, 0, 0, 0}, ptr[4];
    for (uint64_t i = 1; i <= n; ++i) cnt[orc_getbase(text, arr[i])]++;
    ptr[0] = 1;
    ptr[1] = ptr[0] + cnt[0];
    ptr[2] = ptr[1] + cnt[1];
    ptr[3] = ptr[2] + cnt[2];
    uint32_t start[5] = {ptr[0], ptr[1], ptr[2], ptr[3], n + 1};
    aux[0] = arr[0];
    for (uint64_t i = 1; i <= n; ++i) aux[ptr[orc_getbase(text, arr[i])]++] = arr[i];
    hs_job jobs[4];
    pthread_t th[4];
    for (int t = 0; t < 4; ++t) {
      hs_job j = {text, n, arr, aux, max_depth, {start[t], start[t + 1], 1, 0}, 0};
      jobs[t] = j;
      pthread_create(&th[t], NULL, hs_run, &jobs[t]);
    }
    for (int t = 0; t < 4; ++t) {
      pthread_join(th[t], NULL);
      failed |= jobs[t].failed;
    }
  }
  free(aux);
  if (failed) {
    free(arr);
    return NULL;
  }
  return arr;
}

/* ------------------------------------------------------------------ */
/* suffix array: prefix doubling (repetition-proof second construction) */
/* ------------------------------------------------------------------ */

uint32_t *orc_sa_doubling(const uint8_t *text, uint32_t n) {
  const uint64_t N = (uint64_t)n + 1; /* suffixes 0..n; suffix n is "$" */
  uint32_t *sa = malloc(N * sizeof *sa);
  uint32_t *rk = malloc(N * sizeof *rk);
  uint32_t *tmp = malloc(N * sizeof *tmp);
  uint32_t *cnt = malloc((N + 1) * sizeof *cnt);
  /* round 0: order by first symbol ($ = 0, bases 1..4) */
  uint64_t c5[6] = {0, 0, 0, 0, 0, 0};
  for (uint64_t i = 0; i < N; ++i) {
    unsigned s = (i == n) ? 0 : 1 + orc_getbase(text, i);
    c5[s + 1]++;
  }
  for (int s = 0; s < 5; ++s) c5[s + 1] += c5[s];
  uint64_t head[5];
  for (int s = 0; s < 5; ++s) head[s] = c5[s];
  for (uint64_t i = 0; i < N; ++i) {
    unsigned s = (i == n) ? 0 : 1 + orc_getbase(text, i);
    rk[i] = (uint32_t)head[s]; /* rank = first row of the symbol's group */
    sa[c5[s]++] = (uint32_t)i;
  }
  for (uint64_t h = 1;; h *= 2) {
    /* order by second key: suffixes whose h-successor falls off the end come
     * first (they are unique already), then i = sa[j]-h in sa order */
    uint64_t p = 0;
    for (uint64_t i = N - (h < N ? h : N); i < N; ++i) tmp[p++] = (uint32_t)i;
    for (uint64_t j = 0; j < N; ++j)
      if (sa[j] >= h) tmp[p++] = (uint32_t)(sa[j] - h);
    /* stable counting sort by first key (group-head ranks, < N) */
    memset(cnt, 0, (N + 1) * sizeof *cnt);
    for (uint64_t i = 0; i < N; ++i) cnt[rk[i] + 1]++;
    for (uint64_t i = 0; i < N; ++i) cnt[i + 1] += cnt[i];
    for (uint64_t j = 0; j < N; ++j) sa[cnt[rk[tmp[j]]]++] = tmp[j];
    /* new ranks = first row of each (key1,key2) group */
    uint64_t groups = 0;
    uint32_t prev1 = 0, prev2 = 0, cur = 0;
    for (uint64_t j = 0; j < N; ++j) {
      uint32_t i = sa[j];
      uint32_t k1 = rk[i];
      uint32_t k2 = ((uint64_t)i + h < N) ? rk[i + h] + 1 : 0;
      if (j == 0 || k1 != prev1 || k2 != prev2) {
        cur = (uint32_t)j;
        ++groups;
      }
      tmp[i] = cur;
      prev1 = k1;
      prev2 = k2;
    }
    memcpy(rk, tmp, N * sizeof *rk);
    if (groups == N || h >= N) break;
  }
  free(rk);
  free(tmp);
  free(cnt);
  return sa;
}

int orc_check_sa(const uint8_t *text, uint32_t n, const uint32_t *sa) {
  const uint64_t N = (uint64_t)n + 1;
  uint32_t *isa = malloc(N * sizeof *isa);
  memset(isa, 0xFF, N * sizeof *isa);
  int bad = 0;
  for (uint64_t j = 0; j < N && !bad; ++j) {
    if (sa[j] > n || isa[sa[j]] != 0xFFFFFFFFu) bad = 1;
    else isa[sa[j]] = (uint32_t)j;
  }
  if (!bad && sa[0] != n) bad = 2;
  for (uint64_t j = 1; j + 1 < N && !bad; ++j) {
    uint32_t a = sa[j], b = sa[j + 1];
    unsigned ca = orc_getbase(text, a), cb = orc_getbase(text, b);
    if (ca > cb) bad = 3;
    else if (ca == cb && isa[a + 1] >= isa[b + 1]) bad = 4;
  }
  free(isa);
  return bad;
}

/* ------------------------------------------------------------------ */
/* BWT + FM-index                                                      */
/* ------------------------------------------------------------------ */

uint32_t orc_sprintcbwt(const uint8_t *text, const uint32_t *sa, uint32_t n,
                        uint8_t *out) {
  /* histsortcomp.c:220-250: rows before the $ row are packed at their own
   * position, rows after it one position earlier; four bases per byte,
   * first base in the top bits; a partial last byte is zero-padded */
  uint32_t endloc = 0xFFFFFFFFu;
  uint64_t pos = 0;
  uint8_t acc = 0;
  for (uint64_t i = 0; i <= n; ++i) {
    if (sa[i] == 0) {
      endloc = (uint32_t)i;
      continue;
    }
    acc |= (uint8_t)(orc_getbase(text, (uint64_t)sa[i] - 1) << (2 * (3 - (pos & 3))));
    if ((pos & 3) == 3) {
      out[pos >> 2] = acc;
      acc = 0;
    }
    ++pos;
  }
  if (pos & 3) out[pos >> 2] = acc;
  return endloc;
}

void orc_lookup_table(uint8_t *tbl) {
  /* seqindex.c:115-138: tbl[4*b + c] = occurrences of base c in byte b */
  memset(tbl, 0, 1024);
  for (unsigned b = 0; b < 256; ++b)
    for (unsigned k = 0; k < 4; ++k) tbl[4 * b + ((b >> (2 * k)) & 3)]++;
}

uint32_t *orc_seq_index(const uint8_t *bwt, uint32_t len) {
  /* seqindex.c:23-87 with blocksize 16: row j = counts of each base in
   * bwt[0, 16 j); one more row holding the totals when len % 16 != 0 */
  uint64_t full = len / 16;
  uint64_t rows = 1 + ((uint64_t)len + 15) / 16;
  uint32_t *idx = calloc(rows * 4, sizeof *idx);
  uint8_t tbl[1024];
  orc_lookup_table(tbl);
  uint64_t j;
  for (j = 1; j <= full; ++j) {
    uint32_t *row = idx + 4 * j;
    memcpy(row, row - 4, 4 * sizeof *row);
    for (int b = 0; b < 4; ++b) {
      unsigned byte = bwt[(j - 1) * 4 + b];
      for (int c = 0; c < 4; ++c) row[c] += tbl[4 * byte + c];
    }
  }
  if (len % 16) {
    uint32_t *row = idx + 4 * j;
    memcpy(row, row - 4, 4 * sizeof *row);
    for (uint64_t p = full * 16; p < len; ++p) row[orc_getbase(bwt, p)]++;
  }
  return idx;
}

orc_fmi *orc_make_fmi(const uint8_t *text, uint32_t n, const uint32_t *sa_in) {
  /* seqindex.c:161-189 */
  uint32_t *own = NULL;
  const uint32_t *sa = sa_in;
  if (!sa) {
    own = orc_histsort(text, n, n >= 10000000u ? 4 : 1, 0);
    sa = own;
  }
  orc_fmi *f = calloc(1, sizeof *f);
  uint64_t ns = 1 + (uint64_t)n / 32;
  f->idxs = malloc(ns * sizeof *f->idxs);
  for (uint64_t i = 0; i < ns; ++i) f->idxs[i] = sa[32 * i];
  f->bwt = calloc(((uint64_t)n + 3) / 4 + 1, 1);
  f->len = n;
  f->endloc = orc_sprintcbwt(text, sa, n, f->bwt);
  free(own);
  orc_lookup_table(f->lookup);
  f->rank_rows = orc_seq_index(f->bwt, n);
  const uint32_t *tot = f->rank_rows + 4 * (((uint64_t)n + 15) / 16);
  f->C[0] = 1;
  for (int c = 0; c < 4; ++c) f->C[c + 1] = f->C[c] + tot[c];
  return f;
}

void orc_destroy_fmi(orc_fmi *f) {
  if (!f) return;
  free(f->bwt);
  free(f->idxs);
  free(f->rank_rows);
  free(f);
}

static uint32_t seq_rank16(const orc_fmi *f, uint32_t idx, unsigned c) {
  /* seqindex.c:91-111 */
  uint32_t x = f->rank_rows[4 * (uint64_t)(idx / 16) + c];
  for (uint64_t i = (uint64_t)(idx / 16) * 4; i < idx / 4; ++i)
    x += f->lookup[4 * f->bwt[i] + c];
  for (uint32_t i = 0; i < idx % 4; ++i)
    if (orc_getbase(f->bwt, (uint64_t)(idx & ~3u) + i) == c) ++x;
  return x;
}

uint32_t orc_rank(const orc_fmi *f, unsigned c, uint32_t idx) {
  /* seqindex.c:221-225: the $ row is not stored, later rows shift down */
  if (idx > f->endloc) idx--;
  return seq_rank16(f, idx, c);
}

uint32_t orc_lf(const orc_fmi *f, uint32_t idx) {
  /* seqindex.c:214-219 */
  if (idx == f->endloc) return 0;
  unsigned c = orc_getbase(f->bwt, idx - (idx > f->endloc));
  return f->C[c] + orc_rank(f, c, idx);
}

uint32_t orc_unc_sa(const orc_fmi *f, uint32_t idx, uint32_t *steps) {
  /* seqindex.c:244-255 */
  uint32_t i = 0;
  while (idx & 31) {
    idx = orc_lf(f, idx);
    ++i;
  }
  uint64_t x = (uint64_t)f->idxs[idx / 32] + i;
  if (x > f->len) x -= (uint64_t)f->len + 1;
  if (steps) *steps = i;
  return (uint32_t)x;
}

int orc_reverse_search(const orc_fmi *f, const uint8_t *pat, int m) {
  /* seqindex.c:228-242; note the return value end - start + 1 */
  uint32_t start = f->C[pat[m - 1]], end = f->C[pat[m - 1] + 1];
  for (int i = m - 2; i >= 0; --i) {
    if (end <= start) return 0;
    start = f->C[pat[i]] + orc_rank(f, pat[i], start);
    end = f->C[pat[i]] + orc_rank(f, pat[i], end);
  }
  return (int)(end - start + 1);
}

void orc_loc_search(const orc_fmi *f, const uint8_t *pat, int m, uint32_t *sp,
                    uint32_t *ep) {
  /* seqindex.c:280-309 */
  uint32_t start = f->C[pat[m - 1]], end = f->C[pat[m - 1] + 1];
  for (int i = m - 2; i >= 0; --i) {
    if (end <= start) break;
    start = f->C[pat[i]] + orc_rank(f, pat[i], start);
    end = f->C[pat[i]] + orc_rank(f, pat[i], end);
  }
  *sp = start;
  *ep = end;
}

int64_t orc_locate(const orc_fmi *f, const uint8_t *pat, int m) {
  /* seqindex.c:258-277 */
  uint32_t start = f->C[pat[m - 1]], end = f->C[pat[m - 1] + 1];
  for (int i = m - 2; i >= 0; --i) {
    if (end <= start) return -1;
    start = f->C[pat[i]] + orc_rank(f, pat[i], start);
    end = f->C[pat[i]] + orc_rank(f, pat[i], end);
  }
  if (start > f->len) return -1; /* the reference reads out of bounds here */
  return orc_unc_sa(f, start, NULL);
}

int orc_mms(const orc_fmi *f, const uint8_t *pat, int m, uint32_t *sp,
            uint32_t *ep) {
  /* seqindex.c:314-351 */
  int skips = 0;
  while (pat[m - 1] == 5) {
    m--;
    skips++;
  }
  uint32_t start, end;
  *sp = start = f->C[pat[m - 1]];
  *ep = end = f->C[pat[m - 1] + 1];
  int i;
  for (i = m - 2; i >= 0; --i) {
    if (end <= start) break;
    *sp = start;
    *ep = end;
    unsigned c = pat[i];
    if (c == 5) {
      int64_t best = -1;
      for (unsigned d = 0; d < 4; ++d) {
        int64_t count = (int64_t)orc_rank(f, d, end) - (int64_t)orc_rank(f, d, start);
        if (count > best) {
          best = count;
          c = d;
        }
      }
    }
    start = f->C[c] + orc_rank(f, c, start);
    end = f->C[c] + orc_rank(f, c, end);
  }
  if (end <= start) return m - i - 2 + skips;
  *sp = start;
  *ep = end;
  return m - i - 1 + skips;
}

void orc_search_reads_one(const orc_fmi *f, const uint8_t *read, int len, int anchor,
                          uint32_t out[4]) {
  /* search_reads.c:102-160 on an encoded read (0..3, 5 = N; the shipped
   * program passes raw ASCII to mms, which indexes C[] out of bounds, so the
   * read encoding of single_align.c:450-474 is used instead); the second pass
   * runs on the plain reversal of the read (search_reads.c:112-114) */
  uint8_t *rev = malloc(len ? len : 1);
  for (int k = 0; k < len; ++k) rev[k] = read[len - 1 - k];
  out[0] = out[1] = 0;
  out[2] = out[3] = 0xFFFFFFFFu;
  for (int dir = 0; dir < 2; ++dir) {
    const uint8_t *p = dir ? rev : read;
    int l = len;
    while (l > anchor) {
      uint32_t sp, ep;
      int got = orc_mms(f, p, l, &sp, &ep);
      if (got >= anchor) {
        out[dir]++;
        l -= got;
        out[2 + dir] = sp <= f->len ? orc_unc_sa(f, sp, NULL) : 0xFFFFFFFFu;
      } else {
        l -= 1;
      }
    }
  }
  free(rev);
}

/* ---- the aligner's dynamic programming (smw.c) ---- */

/* nw_fast (smw.c:26-111; mode 0) and sw_fast (smw.c:117-200; mode 1): one
 * matrix of scores and one of directions over the read (s1: bases 0..3, 5 = N,
 * which agrees with every base) and a piece of the genome (s2), scores 0 / -6
 * for agreement / disagreement and -3 per gap base plus -5 when a gap opens
 * (:48-54), direction 2 preferred over 1 over 0 on equal scores (:55-62).
 * Mode 0 ends at the best cell of the last row (first one on ties, :65-70) and
 * returns that column - 1; mode 1 ends at the last cell and returns 0.
 * ops receives the unit operations ('M', 'I', 'D') in the order the
 * reference's stack_push calls reach the caller's stack: the traceback order
 * for mode 1 (:167-191), its reverse for mode 0 (collected on a second stack
 * and flipped, :86-107); at most len1 + len2 of them. */
int orc_dp_align(const uint8_t *s1, int len1, const uint8_t *s2, int len2, int mode,
                 uint8_t *ops, uint32_t *nops) {
  *nops = 0;
  if (mode == 0 && len1 == 0) return 0; /* smw.c:28-30 */
  const size_t w = (size_t)len2 + 1;
  int *val = malloc((size_t)(len1 + 1) * w * sizeof(int));
  uint8_t *dir = malloc((size_t)(len1 + 1) * w);
  val[0] = 0;
  dir[0] = 0; /* never read */
  for (int j = 1; j <= len2; ++j) {
    dir[j] = 2;
    val[j] = -5 - 3 * j;
  }
  for (int i = 1; i <= len1; ++i) {
    dir[i * w] = 1;
    val[i * w] = -5 - 3 * i;
    for (int j = 1; j <= len2; ++j) {
      const int skip2 = dir[i * w + j - 1] == 2 ? 0 : -5;
      const int skip1 = dir[(i - 1) * w + j] == 1 ? 0 : -5;
      const int d = val[(i - 1) * w + j - 1] + ((s1[i - 1] == 5 || s1[i - 1] == s2[j - 1]) ? 0 : -6);
      const int l = val[i * w + j - 1] - 3 + skip2;
      const int u = val[(i - 1) * w + j] - 3 + skip1;
      int v = d > l ? d : l;
      v = v > u ? v : u;
      val[i * w + j] = v;
      dir[i * w + j] = v == l ? 2 : v == u ? 1 : 0;
    }
  }
  int i = len1, j = len2, ret = 0;
  if (mode == 0) {
    int mx = -1000000, maxloc = 0;
    for (int q = 0; q <= len2; ++q)
      if (val[len1 * w + q] > mx) {
        mx = val[len1 * w + q];
        maxloc = q;
      }
    j = maxloc;
    ret = maxloc - 1;
  }
  uint32_t k = 0;
  while (i && j) {
    const uint8_t d = dir[i * w + j];
    if (d == 1) {
      --i;
      ops[k++] = 'I';
    } else if (d == 2) {
      --j;
      ops[k++] = 'D';
    } else {
      --i;
      --j;
      ops[k++] = 'M';
    }
  }
  while (i) {
    --i;
    ops[k++] = 'I';
  }
  while (j) {
    --j;
    ops[k++] = 'D';
  }
  if (mode == 0)
    for (uint32_t a = 0, b = k; a + 1 < b; ++a, --b) {
      const uint8_t t = ops[a];
      ops[a] = ops[b - 1];
      ops[b - 1] = t;
    }
  *nops = k;
  free(val);
  free(dir);
  return ret;
}

/* ---- align_read_anchored (single_align.c:163-298) ---- */

/* The caller's CIGAR stack as the list of unit operations pushed so far
 * (stack_push merges runs, stack.c:62-66, which leaves the unit sequence
 * unchanged). */
typedef struct {
  uint8_t *ops;
  uint32_t n, cap;
} opstack;

static void ops_push(opstack *s, uint8_t c, int count) {
  for (int k = 0; k < count; ++k) {
    if (s->n == s->cap) {
      s->cap = s->cap ? 2 * s->cap : 256;
      s->ops = realloc(s->ops, s->cap);
    }
    s->ops[s->n++] = c;
  }
}

static void ops_dp(opstack *s, const uint8_t *s1, int len1, const uint8_t *s2, int len2, int mode, int *ret) {
  uint8_t *tmp = malloc((size_t)len1 + len2 + 1);
  uint32_t k = 0;
  const int r = orc_dp_align(s1, len1, s2, len2, mode, tmp, &k);
  for (uint32_t q = 0; q < k; ++q) ops_push(s, tmp[q], 1);
  free(tmp);
  if (ret) *ret = r;
}

static unsigned align_base(const uint8_t *text, int64_t idx) { /* single_align.c:19-23 */
  return orc_getbase(text, idx < 0 ? 0 : (uint64_t)idx);
}

/* single_align.c:163-298.  Returns the position the function returns (0: not
 * aligned); ops[0..*nops) are the unit CIGAR operations left on the stack,
 * first pushed first.  Two places where the reference's behaviour is undefined
 * are given a definition here and reported: a read of at most anchor_len bases
 * (nmisses is read uninitialised, :281) returns 0, and an mms call on no bases
 * or fewer (len - curgap <= 0, :215) ends the alignment with ORC_ALIGN_UNDEFINED. */
int orc_align_read_anchored(const orc_fmi *f, const uint8_t *text, const uint8_t *pattern, int len,
                            int anchor_len, uint8_t *ops_out, uint32_t *nops) {
  opstack st = {NULL, 0, 0};
  const int olen = len;
  int anchmisses = len / 10, nmisses = 0;
  int curgap = 0, anchlen = 0, result = 0;
  int64_t curpos = -1;
  int done = 0;
  while (len > anchor_len && anchmisses > 0 && !done) {
    nmisses = 0;
    while (len > anchor_len && anchmisses > 0) { /* :175-206 */
      uint32_t sp, ep;
      const int seglen = orc_mms(f, pattern, len, &sp, &ep);
      if (seglen < anchor_len || (int64_t)ep - (int64_t)sp > 1) {
        anchmisses--;
        len -= 3;
        continue;
      }
      len -= seglen;
      anchlen = seglen;
      nmisses = olen / 5;
      curpos = orc_unc_sa(f, sp, NULL);
      int buflen = 10 + (olen - (len + seglen));
      if ((int64_t)buflen + curpos + seglen > (int64_t)f->len) buflen = (int)((int64_t)f->len - curpos - seglen);
      uint8_t *buf = malloc(buflen > 0 ? (size_t)buflen : 1);
      for (int i = 0; i < buflen; ++i) buf[i] = (uint8_t)align_base(text, curpos + seglen + i);
      ops_dp(&st, pattern + len + seglen, olen - (len + seglen), buf, buflen, 0, NULL);
      free(buf);
      ops_push(&st, 'M', seglen);
      break;
    }
    if (nmisses < 1) continue;
    while (len > nmisses && len > 4 && nmisses > 0 && !done) { /* :212-255 */
      for (curgap = 1; curgap < 10; ++curgap) {
        if (len - curgap <= 0) {
          result = ORC_ALIGN_UNDEFINED;
          done = 1;
          break;
        }
        uint32_t start, end;
        const int seglen = orc_mms(f, pattern, len - curgap, &start, &end);
        int matched = 0;
        for (uint32_t i = start; i < end; ++i) {
          const int64_t here = orc_unc_sa(f, i, NULL);
          int64_t dist = here + seglen - curpos;
          if (dist < 0) dist = -dist;
          if (dist - curgap <= 3) {
            nmisses -= curgap;
            matched = 1;
            const int64_t buflen = curpos - (here + seglen);
            if (buflen < 0) {
              ops_push(&st, 'I', (int)-buflen);
            } else {
              uint8_t *buf = malloc(buflen > 0 ? (size_t)buflen : 1);
              for (int64_t j = 0; j < buflen; ++j) buf[j] = (uint8_t)align_base(text, here + seglen + j);
              ops_dp(&st, pattern + (len - curgap), curgap, buf, (int)buflen, 1, NULL);
              free(buf);
            }
            ops_push(&st, 'M', seglen);
            curpos = here;
            len -= seglen + curgap;
            curgap = 0;
            break;
          }
        }
        if (matched) break;
      }
      if (done) break;
      if (curgap) nmisses = 0;
    }
    if (done) break;
    if (nmisses > 0) { /* :256-272 */
      int buflen = len + 10;
      if (buflen > curpos) buflen = (int)curpos;
      uint8_t *buf = malloc(buflen > 0 ? (size_t)buflen : 1), *buf2 = malloc(len > 0 ? (size_t)len : 1);
      for (int i = 0; i < buflen; ++i) buf[i] = (uint8_t)align_base(text, curpos - 1 - i);
      for (int i = 0; i < len; ++i) buf2[i] = pattern[len - 1 - i];
      int x = 0;
      ops_dp(&st, buf2, len, buf, buflen, 0, &x);
      free(buf);
      free(buf2);
      result = (int)(curpos - x);
      done = 2;
      break;
    }
    len -= anchlen;
    anchmisses -= anchlen / 10;
    st.n = 0; /* :279 */
  }
  if (!done) {
    if (len > nmisses || nmisses < 1) { /* :281-283 */
      result = 0;
    } else {
      int buflen = len + 10;
      if (buflen > curpos) buflen = (int)curpos;
      uint8_t *buf = malloc(buflen > 0 ? (size_t)buflen : 1), *buf2 = malloc(len > 0 ? (size_t)len : 1);
      for (int i = 0; i < buflen; ++i) buf[i] = (uint8_t)align_base(text, curpos - 1 - i);
      for (int i = 0; i < len; ++i) buf2[i] = pattern[len - 1 - i];
      ops_dp(&st, buf2, len, buf, buflen, 0, NULL);
      free(buf);
      free(buf2);
      result = (int)(curpos - len); /* :297 */
    }
  }
  for (uint32_t q = 0; q < st.n; ++q) ops_out[q] = st.ops[q];
  *nops = st.n;
  free(st.ops);
  return result;
}

/* ---- threaded batch drivers (fmitest.c:54-91: static contiguous split) ---- */

typedef struct {
  const orc_fmi *f;
  const uint8_t *pats;
  uint32_t m;
  uint64_t stride, lo, hi;
  uint32_t *sp, *ep;
  const uint32_t *rows;
  uint32_t *pos;
  uint64_t steps;
} batch_job;

static void *search_worker(void *arg) {
  batch_job *j = arg;
  for (uint64_t r = j->lo; r < j->hi; ++r)
    orc_loc_search(j->f, j->pats + r * j->stride, (int)j->m, &j->sp[r], &j->ep[r]);
  return NULL;
}

static void *locate_worker(void *arg) {
  batch_job *j = arg;
  uint64_t total = 0;
  for (uint64_t r = j->lo; r < j->hi; ++r) {
    uint32_t s;
    j->pos[r] = orc_unc_sa(j->f, j->rows[r], &s);
    total += s;
  }
  j->steps = total;
  return NULL;
}

static void run_split(batch_job *proto, uint64_t R, int threads,
                      void *(*fn)(void *), uint64_t *steps) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  batch_job jobs[256];
  pthread_t th[256];
  for (int t = 0; t < threads; ++t) {
    jobs[t] = *proto;
    jobs[t].lo = R * t / threads;
    jobs[t].hi = R * (t + 1) / threads;
    pthread_create(&th[t], NULL, fn, &jobs[t]);
  }
  uint64_t total = 0;
  for (int t = 0; t < threads; ++t) {
    pthread_join(th[t], NULL);
    total += jobs[t].steps;
  }
  if (steps) *steps = total;
}

void orc_search_batch(const orc_fmi *f, const uint8_t *pats, uint32_t m,
                      uint64_t stride, uint64_t R, uint32_t *sp, uint32_t *ep,
                      int threads) {
  batch_job p = {f, pats, m, stride, 0, 0, sp, ep, NULL, NULL, 0};
  run_split(&p, R, threads, search_worker, NULL);
}

void orc_locate_rows(const orc_fmi *f, const uint32_t *rows, uint64_t R,
                     uint32_t *pos, uint64_t *total_steps, int threads) {
  batch_job p = {f, NULL, 0, 0, 0, 0, NULL, NULL, rows, pos, 0};
  run_split(&p, R, threads, locate_worker, total_steps);
}

/* ------------------------------------------------------------------ */
/* index file (fileio.c:8-65): len, C[5], endloc, idxs, bwt; host order */
/* ------------------------------------------------------------------ */

int orc_write_index(const orc_fmi *f, FILE *fp) {
  uint64_t ns = 1 + (uint64_t)f->len / 32, nb = ((uint64_t)f->len + 3) / 4;
  int ok = fwrite(&f->len, 4, 1, fp) == 1;
  ok &= fwrite(f->C, 4, 5, fp) == 5;
  ok &= fwrite(&f->endloc, 4, 1, fp) == 1;
  ok &= fwrite(f->idxs, 4, ns, fp) == ns;
  ok &= fwrite(f->bwt, 1, nb, fp) == nb;
  return ok ? 0 : -1;
}

orc_fmi *orc_read_index(FILE *fp) {
  orc_fmi *f = calloc(1, sizeof *f);
  int ok = fread(&f->len, 4, 1, fp) == 1;
  ok = ok && fread(f->C, 4, 5, fp) == 5;
  ok = ok && fread(&f->endloc, 4, 1, fp) == 1;
  if (ok) {
    uint64_t ns = 1 + (uint64_t)f->len / 32, nb = ((uint64_t)f->len + 3) / 4;
    f->idxs = malloc(ns * 4);
    f->bwt = calloc(nb + 1, 1);
    ok = fread(f->idxs, 4, ns, fp) == ns && fread(f->bwt, 1, nb, fp) == nb;
  }
  if (!ok) {
    /* fileio.c:57-60 */
    fprintf(stderr, "Error reading index from file\n");
    orc_destroy_fmi(f);
    return NULL;
  }
  orc_lookup_table(f->lookup);
  f->rank_rows = orc_seq_index(f->bwt, f->len); /* fileio.c:62-63 */
  return f;
}

int orc_write_index_path(const orc_fmi *f, const char *path) {
  FILE *fp = fopen(path, "wb");
  if (!fp) return -1;
  int rc = orc_write_index(f, fp);
  return fclose(fp) ? -1 : rc;
}

orc_fmi *orc_read_index_path(const char *path) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return NULL;
  orc_fmi *f = orc_read_index(fp);
  fclose(fp);
  return f;
}

/* ------------------------------------------------------------------ */
/* genome packer (build_index.c:54-87)                                 */
/* ------------------------------------------------------------------ */

void orc_pack_ascii(const uint8_t *ascii, uint64_t len, uint8_t *packed) {
  for (uint64_t i = 0; i < len / 4 + 1; ++i) {
    uint8_t c = 0;
    for (int k = 0; k < 4; ++k) {
      uint64_t p = 4 * i + k;
      unsigned b = 0;
      if (p < len) {
        switch (ascii[p]) {
          case 'C': b = 1; break;
          case 'G': b = 2; break;
          case 'T': b = 3; break;
          default: b = 0;
        }
      }
      c |= (uint8_t)(b << (2 * (3 - k)));
    }
    packed[i] = c;
  }
  /* build_index.c:80-87: the tail loop's condition parses as (i < len) & 3,
   * and every fgetc in it returns EOF, so its net effect is to zero the byte
   * that holds the last len % 4 bases */
  if (len) packed[len / 4] = 0;
}
