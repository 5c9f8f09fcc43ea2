/*
 * bwt_mi355x.h -- the drop-in boundary of libbwt_mi355x.so.
 *
 * The reference has no plugin or FFI layer: its "operator API" is the set of
 * non-static C symbols in histsortcomp.h, csacak.h, seqindex.h and fileio.h,
 * resolved at link time (reference Makefile:32-60).  This header declares
 * those symbols with the reference's own names, argument order and ownership
 * rules, so that a caller of histsortcomp.o + csacak.o + seqindex.o + fileio.o
 * can link libbwt_mi355x.so instead.  Everything that is a batch of work --
 * suffix array, BWT, index build and load, the batch forms -- runs on the MI355X
 * through the device entry points of bwt_gpu.h and has no CPU path: if no gfx950
 * device is usable it prints the error to stderr and returns the reference's
 * failure value (NULL / -1 / 0), as the comments say.  The single-query
 * functions (rank, lf, unc_sa, loc_search, reverse_search, locate, mms) answer
 * one query from the host arrays of the fm_index they are given, because one
 * query is a fraction of a kernel launch; BWT_MI355X_SCALAR=device sends them to
 * the GPU as batches of one, as does an fm_index without rank_index.
 *
 * The batch forms at the end have no counterpart in the reference: its
 * callers loop over reads on the host (fmitest.c:54-65, search_reads.c:102-162)
 * and that loop is the thing that moves to the GPU.
 */
#ifndef BWT_MI355X_H
#define BWT_MI355X_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* seqindex.h:14-22; same field order, types and size (64 bytes on x86-64).
 * Lengths above 2^31-1 are carried in the same 4 bytes as unsigned values.
 * rank_index is filled (seq_index layout, blocksize 16) for len < 2^28 and is
 * NULL above that, where the reference's own rows are wrong (seqindex.c:78). */
typedef struct _fmi {
  char *bwt;
  int *idxs;
  int **rank_index;
  unsigned char *lookup;
  int endloc;
  int C[5];
  int len;
} fm_index;

/* ---- histsortcomp.h:4-20 ---- */
/* histsortcomp.h:4 / histsortcomp.c:35-90: sorts arr[start .. end), starts of
 * suffixes that agree on their first `depth` bases, by what follows; len = the
 * number of bases.  On return arr[start .. end) and aux[start .. end) both hold
 * the sorted run (the reference leaves it in one of the two, chosen by a
 * file-static only histsort sets). */
void histhelper(const char *base, int len, int *arr, int *aux, int start, int end, int depth);
/* histsortcomp.h:6 / histsortcomp.c:161-180.  malloc'd (len+1) ints, caller frees. */
int *histsort(const char *str, int len);
/* histsortcomp.h:20 / histsortcomp.c:220-250.  out holds (len+3)/4 bytes; returns the $ row. */
unsigned int sprintcbwt(const char *str, unsigned int *idxs, unsigned int len, char *out);
/* histsortcomp.h:16,18 / histsortcomp.c:195-210 */
int makecbwt(const char *str, int len, char *out);
int saca_makecbwt(const char *str, int len, char *out);
/* histsortcomp.h:14,12,10,8 / histsortcomp.c:183-191,252-287: text forms of the BWT */
char *makebwt(const char *str, int len);
void sprintbwt(char *out, const char *str, const int *bwt, int len);
void putbwt(const char *str, const int *bwt, int len);
void putsg(const char *str, int idx, int len);

/* ---- csacak.h:3 / csacak.c:521-528: same array as histsort, any input ---- */
unsigned int *csuff_arr(const char *seq, unsigned int len);

/* ---- seqindex.h:8-67 ---- */
int **seq_index(char *bwt, int len, int blocksize, const unsigned char *tbl);   /* seqindex.c:23-87 */
int seq_rank(unsigned char *bwt, int **sidx, int blocksize, int idx, char c,
             const unsigned char *lookup);                                      /* seqindex.c:91-111 */
unsigned char *lookup_table(void);                                              /* seqindex.c:115-138 */
void destroy_fmi(fm_index *fmi);                                                /* seqindex.c:140-157 */
fm_index *make_fmi(const char *str, int len);                                   /* seqindex.c:161-189 */
fm_index *make_fmi_sacak(const char *str, int len);                             /* seqindex.c:192-212 */
int rank(const fm_index *fmi, char c, int idx);                                 /* seqindex.c:221-225 */
int lf(const fm_index *fmi, int idx);                                           /* seqindex.c:214-219 */
int reverse_search(const fm_index *fmi, const char *pattern, int len);          /* seqindex.c:228-242 */
int unc_sa(const fm_index *fmi, int idx);                                       /* seqindex.c:244-255 */
int locate(const fm_index *fmi, const char *pattern, int len);                  /* seqindex.c:258-277 */
void loc_search(const fm_index *fmi, const char *pattern, int len, int *sp, int *ep); /* seqindex.c:280-309 */
int mms(const fm_index *fmi, const char *pattern, int len, int *sp, int *ep);    /* seqindex.c:314-351 */
void printseq(const char *seq, int startidx, int len);                          /* seqindex.c:354-360 */

/* ---- fileio.h:4-6 / fileio.c:8-65: byte-identical index files ---- */
void write_index(const fm_index *fmi, FILE *f);
fm_index *read_index(const char *seq, FILE *f);

/* ---- batch forms (no counterpart in the reference) ---- */
/* Read r is the len bytes at pats + r*stride, bases 0..3 one per byte.
 * All return 0, or -1 after printing the device error to stderr.
 *
 * Several GPUs: the index is replicated and the reads are partitioned, the
 * static split over worker threads of fmitest.c:67-91 with a GPU behind each
 * worker.  The GPUs are those of BWT_MI355X_DEVICES ("all", or a comma-separated
 * list of ordinals; an ordinal listed twice is two slots on one GPU), else the
 * one of BWT_MI355X_DEVICE, else every visible GPU.  make_fmi / read_index build
 * or load on the first; the first batch of at least
 * BWT_MI355X_MIN_READS_PER_DEVICE (65536) reads per GPU copies the index to the
 * others (peer-to-peer), and from then on fmi_search_batch, _reverse_search_,
 * _locate_, _search_locate_, _search_batch_packed, _unc_sa_ and _mms_batch hand
 * GPU g a contiguous share of their n reads (fmi_share_bounds) on a host
 * thread of its own and return when all shares are done, results in read order.
 * fmi_search_reads_batch and fmi_align_reads_batch run on the first GPU.  A
 * process that is one of several, one per GPU, sets BWT_MI355X_DEVICE. */
int fmi_device_slots(void); /* how many GPUs (slots) the batch forms spread over */
/* the share of slot g of `shares`: items [*lo, *hi) of `count` */
void fmi_share_bounds(long count, int shares, int g, long *lo, long *hi);
/* loc_search per read: sp[r], ep[r] */
int fmi_search_batch(const fm_index *fmi, const char *pats, int len, long stride,
                     long nreads, int *sp, int *ep);
/* reverse_search per read: its return value (0, or ep-sp+1) */
int fmi_reverse_search_batch(const fm_index *fmi, const char *pats, int len, long stride,
                             long nreads, int *result);
/* locate per read: position, or -1 */
int fmi_locate_batch(const fm_index *fmi, const char *pats, int len, long stride,
                     long nreads, int *pos);
/* loc_search and locate of every read in one pass over the reads (what filetest.c and the
 * worker loop fmitest.c:54-65 do per read): sp[r], ep[r], pos[r]; any of the three may be NULL */
int fmi_search_locate_batch(const fm_index *fmi, const char *pats, int len, long stride,
                            long nreads, int *sp, int *ep, int *pos);
/* Reads packed to 2 bits per base, for callers that hold many of them: read r
 * becomes the (len+15)/16 unsigned words at words + r * ((len+15)/16), base k at
 * bits 2 (k % 16) of word k / 16.  fmi_search_batch_packed = loc_search, and
 * locate when pos is not NULL (-1 where locate returns -1), per packed read;
 * sp, ep, pos may each be NULL.  From pinned host memory this is the fastest way
 * in: a quarter of the bytes, no packing on the way. */
void fmi_pack_reads(const char *pats, int len, long stride, long nreads, unsigned *words);
int fmi_search_batch_packed(const fm_index *fmi, const unsigned *words, int len, long nreads,
                            int *sp, int *ep, int *pos);
/* unc_sa per row */
int fmi_unc_sa_batch(const fm_index *fmi, const int *rows, long nrows, int *pos);

/* mms per read (symbols 0..3, 5 = N): matched[r], sp[r], ep[r] */
int fmi_mms_batch(const fm_index *fmi, const char *pats, int len, long stride, long nreads,
                  int *matched, int *sp, int *ep);
/* The per-read body of search_reads (search_reads.c:102-160) for nreads reads
 * stored back to back, read r at reads[offsets[r] .. offsets[r+1]), encoded
 * 0..3 / 5 (fmi_encode_read).  out[4r..4r+3] = forward anchors, backward
 * anchors, last forward position, last backward position (-1: none). */
int fmi_search_reads_batch(const fm_index *fmi, const char *reads, const unsigned long *offsets,
                           long nreads, int anchor, int *out);
/* align_read_anchored (single_align.c:163-298) for many reads at once: seq is
 * the packed sequence the index was built from, reads are stored as for
 * fmi_search_reads_batch.  pos[r] = the function's return value (0: not
 * aligned; INT_MIN where the reference calls mms on no bases, which is
 * undefined there).  The CIGAR stack comes back as unit operations 'M', 'I',
 * 'D', first pushed first (stack_print_destroy prints from the other end):
 * nops[r] of them at ops + 5 * offsets[r] + 64 * r; ops holds
 * 5 * offsets[nreads] + 64 * nreads bytes. */
int fmi_align_reads_batch(const fm_index *fmi, const char *seq, const char *reads,
                          const unsigned long *offsets, long nreads, int anchor_len, int *pos,
                          char *ops, unsigned *nops);
/* read_index (fileio.c:24) ignores the sequence it is given.  This call hands
 * the packed sequence the index was built from to the device, which checks it
 * against the index and then uses it to speed up searches and locate (same
 * results).  Returns 0, or -1 with the index unchanged (wrong text, no memory). */
int fmi_attach_text(const fm_index *fmi, const char *seq);

/* ASCII -> symbols as single_align.c:450-474 encodes reads: A C G T -> 0 1 2 3,
 * anything else -> 5 */
void fmi_encode_read(const char *ascii, long len, char *out);

/* the genome packer of build_index.c:54-87 / search_reads.c:47-80 (tail quirk
 * included: the last len%4 bases read as A); out holds len/4+1 bytes */
void fmi_pack_ascii(const char *ascii, long len, char *out);

#ifdef __cplusplus
}
#endif
#endif
