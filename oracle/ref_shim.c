/*
 * Harness compiled INTO oracle/_ref/libbwtref.so next to the unmodified
 * reference sources.  It only calls the reference's public functions
 * (seqindex.h, fileio.h) from worker threads the way fmitest.c:54-91 does, and
 * opens FILE handles for fileio.c so that Python can drive it through ctypes.
 * TEST INFRASTRUCTURE ONLY; never linked into the product library.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "seqindex.h"
#include "fileio.h"
#include "smw.h"
#include "stack.h"

typedef struct {
  const fm_index *fmi;
  const char *pats;
  int m;
  long stride, lo, hi;
  int *sp, *ep;
  const int *rows;
  int *pos;
  long acc;
} job_t;

static void *w_loc_search(void *a) {
  job_t *j = a;
  for (long r = j->lo; r < j->hi; ++r)
    loc_search(j->fmi, j->pats + r * j->stride, j->m, &j->sp[r], &j->ep[r]);
  return NULL;
}

static void *w_reverse_search(void *a) {
  job_t *j = a;
  long acc = 0;
  for (long r = j->lo; r < j->hi; ++r)
    acc += reverse_search(j->fmi, j->pats + r * j->stride, j->m);
  j->acc = acc;
  return NULL;
}

static void *w_unc_sa(void *a) {
  job_t *j = a;
  for (long r = j->lo; r < j->hi; ++r) j->pos[r] = unc_sa(j->fmi, j->rows[r]);
  return NULL;
}

static long run(job_t *proto, long R, int threads, void *(*fn)(void *)) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  job_t jobs[256];
  pthread_t th[256];
  for (int t = 0; t < threads; ++t) {
    jobs[t] = *proto;
    jobs[t].lo = R * t / threads;
    jobs[t].hi = R * (t + 1) / threads;
    pthread_create(&th[t], NULL, fn, &jobs[t]);
  }
  long acc = 0;
  for (int t = 0; t < threads; ++t) {
    pthread_join(th[t], NULL);
    acc += jobs[t].acc;
  }
  return acc;
}

void ref_loc_search_mt(const fm_index *fmi, const char *pats, int m, long stride,
                       long R, int *sp, int *ep, int threads) {
  job_t p = {fmi, pats, m, stride, 0, 0, sp, ep, NULL, NULL, 0};
  run(&p, R, threads, w_loc_search);
}

long ref_reverse_search_mt(const fm_index *fmi, const char *pats, int m,
                           long stride, long R, int threads) {
  job_t p = {fmi, pats, m, stride, 0, 0, NULL, NULL, NULL, NULL, 0};
  return run(&p, R, threads, w_reverse_search);
}

void ref_unc_sa_mt(const fm_index *fmi, const int *rows, long R, int *pos,
                   int threads) {
  job_t p = {fmi, NULL, 0, 0, 0, 0, NULL, NULL, rows, pos, 0};
  run(&p, R, threads, w_unc_sa);
}

int ref_write_index_path(const fm_index *fmi, const char *path) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  write_index(fmi, f);
  return fclose(f);
}

fm_index *ref_read_index_path(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fm_index *fmi = read_index(NULL, f);
  fclose(f);
  return fmi;
}

/* rank rows are one malloc each in the reference (seqindex.c:38-43); copy
 * them out flat for comparison */
void ref_copy_rank_rows(const fm_index *fmi, int *out) {
  long rows = 1 + ((long)fmi->len + 15) / 16;
  for (long j = 0; j < rows; ++j)
    for (int c = 0; c < 4; ++c) out[4 * j + c] = fmi->rank_index[j][c];
}

/* nw_fast (mode 0) / sw_fast (mode 1) on a fresh stack; the stack's runs are
 * written out as unit operations, first pushed first. */
int ref_dp_align(const char *s1, int len1, const char *s2, int len2, int mode,
                 unsigned char *ops, unsigned *nops) {
  stack *s = stack_make();
  int ret = 0;
  if (mode == 0) ret = nw_fast(s1, len1, s2, len2, s);
  else sw_fast(s1, len1, s2, len2, s);
  unsigned k = 0;
  for (int e = 0; e < s->size; ++e)
    for (int c = 0; c < s->counts[e]; ++c) ops[k++] = (unsigned char)s->chars[e];
  *nops = k;
  stack_destroy(s);
  return ret;
}

/* align_read_anchored (single_align.c:163) on a fresh stack, as main() calls
 * it (single_align.c:479-483); single_align.c is compiled in with its main
 * renamed. */
int align_read_anchored(const fm_index *fmi, const char *seq, const char *pattern, int len, int anchor_len,
                        stack *s);
int ref_align_read_anchored(const fm_index *fmi, const char *seq, const char *pattern, int len,
                            int anchor_len, unsigned char *ops, unsigned *nops) {
  stack *s = stack_make();
  const int pos = align_read_anchored(fmi, seq, pattern, len, anchor_len, s);
  unsigned k = 0;
  for (int e = 0; e < s->size; ++e)
    for (int c = 0; c < s->counts[e]; ++c) ops[k++] = (unsigned char)s->chars[e];
  *nops = k;
  stack_destroy(s);
  return pos;
}
