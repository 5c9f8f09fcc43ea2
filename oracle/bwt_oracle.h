/*
 * CPU oracle for the SA/BWT build and FM-index search paths.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithms, used by tests/, the smoke check and the bench's cpu_baseline leg
 * as the checker for the HIP path.  Nothing in the product library links,
 * loads or calls it.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*)
 * against golden vectors produced by running the unmodified reference
 * (oracle/_ref/libbwtref.so, built by oracle/Makefile from the reference's own
 * sources) on seeded inputs; see tests/golden/make_golden.py.
 *
 * Differences from the reference, all deliberate:
 *   - lengths, rows and counts are uint32_t (the reference uses int), so
 *     texts up to 2^32-2 bases work; results are identical below 2^31;
 *   - orc_seq_index counts the final partial block exactly; the reference
 *     masks with 0xFFFFFFC (seqindex.c:78) and miscounts when
 *     len >= 2^28 and len % 4 != 0;
 *   - rank rows live in one flat array instead of one malloc per row.
 */
#ifndef BWT_ORACLE_H
#define BWT_ORACLE_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- seeded synthetic inputs (counter-based splitmix64) ---- */
uint64_t orc_rnd(uint64_t seed, uint64_t k);
/* bytes needed for a packed text of n bases, including the byte that holds
 * base n (which always reads 0) */
uint64_t orc_packed_bytes(uint64_t n);
void orc_gen_genome(uint8_t *text, uint64_t n, uint64_t seed);
/* text[i] = unit[i % period]; every base with rnd(mut_seed,i) % mut_every == 0
 * is replaced by a different base (mut_every == 0: no mutations) */
void orc_gen_tandem(uint8_t *text, uint64_t n, uint64_t period,
                    uint64_t unit_seed, uint64_t mut_every, uint64_t mut_seed);
/* R reads of m bases, one byte per base (0..3), row-major; about 1 % of the
 * reads are random, the rest are substrings; offsets[r] = text offset or
 * 0xFFFFFFFF for a random read (offsets may be NULL) */
void orc_gen_reads(const uint8_t *text, uint64_t n, uint8_t *reads,
                   uint64_t R, uint32_t m, uint64_t seed, uint32_t *offsets);

static inline unsigned orc_getbase(const uint8_t *s, uint64_t i) {
  /* histsortcomp.c:20-23 */
  return (s[i >> 2] >> (2 * (3 - (i & 3)))) & 3;
}

/* ---- suffix array ---- */
/* MSD 2-bit bucket sort (histsortcomp.c:35-180). threads: 1 or 4 (the
 * reference uses 4 for len >= 10^7).  Returns malloc'd SA of n+1 rows with
 * SA[0] = n.  max_depth bounds the sort depth (0 = unbounded); returns NULL if
 * the bound is hit (the reference overflows its stack on such inputs). */
uint32_t *orc_histsort(const uint8_t *text, uint32_t n, int threads,
                       uint32_t max_depth);
/* Independent O(n log n) prefix-doubling SA for repetitive inputs; same
 * output contract.  Plays the role csuff_arr (csacak.c:521-528) plays in the
 * reference: a second, repetition-proof construction of the same array. */
uint32_t *orc_sa_doubling(const uint8_t *text, uint32_t n);
/* returns 0 when SA is the suffix array of text (permutation + order) */
int orc_check_sa(const uint8_t *text, uint32_t n, const uint32_t *sa);

/* ---- BWT + FM-index ---- */
/* histsortcomp.c:220-250; out has (n+3)/4 bytes; returns the $ row */
uint32_t orc_sprintcbwt(const uint8_t *text, const uint32_t *sa, uint32_t n,
                        uint8_t *out);
/* seqindex.c:115-138; 1024 bytes */
void orc_lookup_table(uint8_t *tbl);

typedef struct {
  uint8_t  *bwt;        /* (len+3)/4 bytes, no $ */
  uint32_t *idxs;       /* 1 + len/32 entries: SA[32 i] */
  uint32_t *rank_rows;  /* (1 + (len+15)/16) rows of 4 counts */
  uint8_t   lookup[1024];
  uint32_t  endloc;
  uint32_t  C[5];
  uint32_t  len;
} orc_fmi;

/* seqindex.c:23-87 with blocksize 16; returns rows, flat */
uint32_t *orc_seq_index(const uint8_t *bwt, uint32_t len);
/* seqindex.c:161-189 (sa == NULL: build it with orc_histsort, 4 threads at
 * len >= 10^7 as the reference does) */
orc_fmi *orc_make_fmi(const uint8_t *text, uint32_t n, const uint32_t *sa);
void orc_destroy_fmi(orc_fmi *f);

uint32_t orc_rank(const orc_fmi *f, unsigned c, uint32_t idx);    /* seqindex.c:221-225, 91-111 */
uint32_t orc_lf(const orc_fmi *f, uint32_t idx);                  /* seqindex.c:214-219 */
uint32_t orc_unc_sa(const orc_fmi *f, uint32_t idx, uint32_t *steps); /* seqindex.c:244-255 */
int      orc_reverse_search(const orc_fmi *f, const uint8_t *pat, int m); /* seqindex.c:228-242 */
void     orc_loc_search(const orc_fmi *f, const uint8_t *pat, int m,
                        uint32_t *sp, uint32_t *ep);              /* seqindex.c:280-309 */
/* seqindex.c:258-277 without the stdout warning; -1 when the interval empties
 * before the last step */
int64_t  orc_locate(const orc_fmi *f, const uint8_t *pat, int m);
int      orc_mms(const orc_fmi *f, const uint8_t *pat, int m,
                 uint32_t *sp, uint32_t *ep);                     /* seqindex.c:314-351 */

/* the per-read body of search_reads (search_reads.c:102-160) on an encoded
 * read: out = forward anchors, backward anchors, last forward position, last
 * backward position (0xFFFFFFFF: none) */
void orc_search_reads_one(const orc_fmi *f, const uint8_t *read, int len, int anchor,
                          uint32_t out[4]);
/* nw_fast (mode 0) / sw_fast (mode 1) of smw.c: returns what nw_fast returns
 * (0 for mode 1); ops[0..*nops) are the unit CIGAR operations in the order they
 * reach the caller's stack. */
int orc_dp_align(const uint8_t *s1, int len1, const uint8_t *s2, int len2, int mode,
                 uint8_t *ops, uint32_t *nops);
/* align_read_anchored (single_align.c:163-298): the position it returns (0: not
 * aligned) and the unit CIGAR operations left on the stack, first pushed first
 * (room for 5 * len + 64).  ORC_ALIGN_UNDEFINED where the reference calls mms on
 * no bases or fewer (its behaviour is undefined there). */
#define ORC_ALIGN_UNDEFINED (-2147483647 - 1)
int orc_align_read_anchored(const orc_fmi *f, const uint8_t *text, const uint8_t *pattern, int len,
                            int anchor_len, uint8_t *ops_out, uint32_t *nops);

/* batch drivers used for cpu_baseline timing: threads pthreads, each a
 * contiguous slice (fmitest.c:67-91) */
void orc_search_batch(const orc_fmi *f, const uint8_t *pats, uint32_t m,
                      uint64_t stride, uint64_t R, uint32_t *sp, uint32_t *ep,
                      int threads);
void orc_locate_rows(const orc_fmi *f, const uint32_t *rows, uint64_t R,
                     uint32_t *pos, uint64_t *total_steps, int threads);

/* ---- index file (fileio.c:8-65) ---- */
int      orc_write_index(const orc_fmi *f, FILE *fp);
orc_fmi *orc_read_index(FILE *fp);
int      orc_write_index_path(const orc_fmi *f, const char *path);
orc_fmi *orc_read_index_path(const char *path);

/* ---- genome packer (build_index.c:54-87, quirk included) ---- */
/* ascii[len] -> packed (len/4+1 bytes); the last len%4 bases read as A */
void orc_pack_ascii(const uint8_t *ascii, uint64_t len, uint8_t *packed);

#ifdef __cplusplus
}
#endif
#endif
