/*
 * bwt_gpu.h -- device-level C-ABI of libbwt_mi355x.so (MI355X / gfx950).
 *
 * Plain C linkage, plain pointers and sizes.  Pointers whose name starts with
 * d_ are device pointers (HBM), h_ host pointers.  `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  Every function returns 0 on
 * success and a negative code on failure; bwtgpu_last_error() returns the
 * message for the calling thread.  Nothing here falls back to the CPU: if no
 * gfx950 device is usable the call fails.
 *
 * Each entry point names the reference interface it replaces (file:line under
 * the yichizhng/bwt tree).  The reference runs these loops on the host; the
 * batch forms exist because fmitest.c:54-91 and search_reads.c:102-162 loop
 * over reads on the host and that loop is what moves to the GPU.
 */
#ifndef BWT_GPU_H
#define BWT_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bwtgpu_index bwtgpu_index; /* HBM-resident FM-index */

/* ---- library ---- */
/* Makes `device` current for the calling thread and checks that it is a
 * gfx950 part.  Must precede any other call on a thread. */
int bwtgpu_init(int device);
const char *bwtgpu_last_error(void);
int bwtgpu_device_count(void);
/* When enabled, every kernel launch of the calling thread's next calls is
 * bracketed by HIP events on its own stream; bwtgpu_timing_get returns, for a
 * kernel name, the number of launches and their summed duration since the
 * last bwtgpu_timing_reset.  Used by bench.py for the roofline figures. */
int bwtgpu_timing_enable(int on);
int bwtgpu_timing_reset(void);
int bwtgpu_timing_get(const char *kernel, uint64_t *launches, double *total_ms);
/* fills buf with "name launches total_ms\n" lines; returns bytes written */
int bwtgpu_timing_report(char *buf, size_t cap);

/* ---- suffix array + BWT (HP1) ---- */
/* Replaces histsort (histsortcomp.c:161-180) and csuff_arr (csacak.c:521-528):
 * the suffix array of text+$ as n+1 uint32 rows, d_sa[0] = n.  d_text is the
 * 2-bit packed text (base i in byte i>>2, bits 2*(3-(i&3)); histsortcomp.c:20-23),
 * n/4+1 bytes readable; bits past base n-1 are ignored.  Repetitive inputs
 * take the prefix-doubling path instead of overflowing a stack.
 * Workspace: bwtgpu_sa_workspace_bytes(n) bytes at d_ws (256-B aligned). */
size_t bwtgpu_sa_workspace_bytes(uint32_t n);
int bwtgpu_sa_build(const uint8_t *d_text, uint32_t n, uint32_t *d_sa,
                    void *d_ws, size_t ws_bytes, void *stream);
/* counters of the last bwtgpu_sa_build on this thread (for tests/bench):
 * [0] partition levels run, [1] leaf buckets, [2] rows left unresolved after
 * the bucket phase, [3] prefix-doubling rounds, [4] deepest partition depth */
int bwtgpu_sa_last_stats(uint64_t stats[8]);

/* Sharded build.  A shard is the part of the suffix array whose suffixes begin
 * with four bases that, read as a base-4 number (first base most significant,
 * bases past the end of the text read as A), lie in [bucket_lo, bucket_hi):
 * shards are contiguous row ranges, so the whole array is their concatenation
 * in bucket order behind row 0.  The reference has one whole-array build
 * (histsort, histsortcomp.c:161-180); its threads split the same way, by
 * first base (histsortcomp.c:112-158).
 * bwtgpu_sa_bucket_counts: how many suffixes begin with each of the 256
 * strings (for choosing balanced ranges).
 * bwtgpu_sa_build_range: d_rows[0 .. *rows) = rows [*row_begin, *row_begin +
 * *rows) of the suffix array; d_bw (may be NULL) receives the BWT symbol of
 * each of those rows, one per byte; *endloc = the row of suffix 0 if the shard
 * holds it, else 0.  capacity = room in d_rows and d_bw, in rows.  Returns -6
 * when the shard holds ties that only a whole-text build orders (long
 * repeats): use bwtgpu_sa_build then.  Workspace as for bwtgpu_sa_build. */
int bwtgpu_sa_bucket_counts(const uint8_t *d_text, uint32_t n, uint64_t h_counts[256],
                            void *stream);
int bwtgpu_sa_build_range(const uint8_t *d_text, uint32_t n, uint32_t bucket_lo,
                          uint32_t bucket_hi, uint32_t *d_rows, uint8_t *d_bw,
                          uint64_t capacity, uint64_t *row_begin, uint64_t *rows,
                          uint32_t *endloc, void *d_ws, size_t ws_bytes, void *stream);

/* What the ranks of a sharded build exchange is not the rows but what the index
 * needs of them: the BWT symbol of every row packed to 2 bits and every 8th row
 * (0.75 bytes per base instead of 5).
 * bwtgpu_pack_bw2: d_bw[0 .. rows) (one symbol per byte, as bwtgpu_sa_build_range
 * leaves them) -> (rows+3)/4 bytes, symbol j at bits 2 (j & 3) of byte j >> 2.
 * bwtgpu_index_build_from_shards: the index of make_fmi (seqindex.c:161-189)
 * from every shard's packed symbols: shard g holds rows [h_row_begin[g],
 * h_row_begin[g] + h_rows[g]) -- the shards cover rows 1..n in order -- and its
 * symbols start at byte h_bw2_offset[g] of d_bw2; d_samples[i] = SA[8 i] for i in
 * [0, n/8] (d_samples[0] = n); endloc = the row of suffix 0.  At most 64 shards.
 * The index keeps no suffix array (it searches on rank blocks and the k-mer
 * table) until bwtgpu_index_attach_sa hands it the n+1 assembled rows, which it
 * copies and from which the first search then makes the position records.
 * No counterpart in the reference, whose threads share one address space
 * (histsortcomp.c:112-158). */
int bwtgpu_pack_bw2(const uint8_t *d_bw, uint64_t rows, uint8_t *d_out, void *stream);
bwtgpu_index *bwtgpu_index_build_from_shards(const uint8_t *d_text, uint32_t n, uint32_t endloc,
                                             uint32_t nshards, const uint64_t *h_row_begin,
                                             const uint64_t *h_rows, const uint64_t *h_bw2_offset,
                                             const uint8_t *d_bw2, const uint32_t *d_samples,
                                             void *stream);
int bwtgpu_index_attach_sa(bwtgpu_index *ix, const uint8_t *d_text, const uint32_t *d_sa,
                           void *stream);

/* Replaces sprintcbwt (histsortcomp.c:220-250): packed BWT without the $ row
 * ((n+3)/4 bytes at d_bwt) and the $ row number at *d_endloc. */
int bwtgpu_bwt_from_sa(const uint8_t *d_text, const uint32_t *d_sa, uint32_t n,
                       uint8_t *d_bwt, uint32_t *d_endloc, void *stream);

/* ---- FM-index build (HP1 tail) ---- */
/* Replaces make_fmi / make_fmi_sacak (seqindex.c:161-212): SA build, every
 * 32nd SA row (seqindex.c:173-176), BWT, rank blocks, C[].  h_text is the
 * packed text on the host (n/4+1 bytes).  The SA never leaves the device. */
bwtgpu_index *bwtgpu_index_build(const uint8_t *h_text, uint32_t n, void *stream);
/* Same from a packed text already in HBM.  Both keep the workspace buffers
 * (about 21 bytes per base) cached on the calling thread for the next build,
 * and the buffers of freed indexes are kept for the next index of the same
 * size; bwtgpu_release_cache() returns all of that to the driver. */
bwtgpu_index *bwtgpu_index_build_dev(const uint8_t *d_text, uint32_t n, void *stream);
int bwtgpu_release_cache(void);
/* The tail of make_fmi (seqindex.c:168-188) from a suffix array already in HBM
 * (n+1 rows, d_sa[0] = n), e.g. one assembled from shards.  d_bw (may be NULL)
 * holds the BWT symbol of rows 1..n, one per byte, and endloc the $ row; when
 * NULL both are read off the text.  The arrays are copied, not adopted. */
bwtgpu_index *bwtgpu_index_build_from_sa(const uint8_t *d_text, uint32_t n,
                                         const uint32_t *d_sa, const uint8_t *d_bw,
                                         uint32_t endloc, void *stream);
/* Replaces the rebuild half of read_index (fileio.c:44-63): takes the stored
 * parts of an index (host pointers) and rebuilds the rank blocks on device.
 * h_idxs may be NULL (no locate on such an index).  The device keeps every 8th
 * row: the missing ones are found by walking LF (seqindex.c:214-219) from the
 * stored rows, one step per row of the index, and the load is refused (NULL,
 * "index parts are inconsistent") unless those walks pass every row once and
 * each arrives at the next stored row with the position stored for it -- BWT,
 * $ row and sampled rows then belong to one text.
 * BWT and sampled rows determine the suffix array and the text: the first batch
 * search on a loaded index (or bwtgpu_index_prepare_search) recomputes both on
 * the device by the same walks (row and text position of one suffix after the
 * other, the BWT symbol being the base in front) and then makes the search-only
 * arrays an index built from the text has, so a loaded index searches as fast
 * as a built one.  This costs one LF step per base once (1.5 ms at 46 Mbp,
 * 0.16 s at 3.1 Gbp) and the memory bwtgpu_index_compact describes;
 * bwtgpu_index_keep_lean, that call, or BWT_MI355X_COMPACT in the environment
 * keep the index as loaded. */
bwtgpu_index *bwtgpu_index_from_parts(const uint8_t *h_bwt, uint32_t n,
                                      uint32_t endloc, const uint32_t *h_idxs,
                                      void *stream);
/* A loaded index stays on its rank blocks, sampled rows and (one-word) k-mer
 * table: the suffix array and the text are not recomputed.  To be called before
 * the first search; no effect on an index that already holds them.  No
 * counterpart in the reference. */
int bwtgpu_index_keep_lean(bwtgpu_index *ix);
/* read_index (fileio.c:24-65) is handed the text the index was built from and
 * ignores it.  Here the text can be put to use: these calls rebuild the
 * suffix array from it on the device, check that it gives the BWT and $ row
 * the index holds (-2 and nothing changed if not), and leave the index as one
 * built from the text (suffix array, inverse, text copy; see
 * bwtgpu_index_compact).  Results of searches do not change, their speed does. */
int bwtgpu_index_attach_text(bwtgpu_index *ix, const uint8_t *h_text, void *stream);
int bwtgpu_index_attach_text_dev(bwtgpu_index *ix, const uint8_t *d_text, void *stream);
void bwtgpu_index_free(bwtgpu_index *ix);
/* The fields of fm_index (seqindex.h:14-22) that write_index stores
 * (fileio.c:11-15).  Any output pointer may be NULL. */
int bwtgpu_index_info(const bwtgpu_index *ix, uint32_t *n, uint32_t *endloc,
                      uint32_t C[5]);
int bwtgpu_index_export(const bwtgpu_index *ix, uint8_t *h_bwt, uint32_t *h_idxs);
/* Rank rows in the reference's layout (seq_index, seqindex.c:23-87, blocksize
 * 16): 1+(n+15)/16 rows of 4 counts, computed on device, copied to h_rows. */
int bwtgpu_index_rank_rows16(const bwtgpu_index *ix, uint32_t *h_rows);
/* Same for any blocksize that is a multiple of 4 (seq_index's third argument):
 * 1+(n+blocksize-1)/blocksize rows. */
int bwtgpu_index_rank_rows(const bwtgpu_index *ix, uint32_t blocksize, uint32_t *h_rows);
size_t bwtgpu_index_device_bytes(const bwtgpu_index *ix);
/* The k-mer table an index holds at this moment (it is made on first search):
 * *k = bases per string (0: none), *entry_words = 1 (interval start) or 2 (start
 * and the text position of the string's first row).  For measurement. */
int bwtgpu_index_kmer_table(const bwtgpu_index *ix, uint32_t *k, uint32_t *entry_words);
/* An index built on the device keeps, beside what make_fmi (seqindex.c:161-189)
 * produces, the whole suffix array, its inverse and the text (9.6 bytes per
 * base more) and uses them to finish searches and to locate in one lookup;
 * every index also holds a table of the SA intervals of all strings of k bases
 * (4 bytes per string and the smallest k with 4^k >= n for an index without
 * its suffix array; 8 bytes -- the interval start and the text position of its
 * first row, which is all a string that occurs once needs -- and the smallest k
 * with 4^k >= 2n for one that keeps it; at most k = 16; BWT_MI355X_KMER=<k>
 * overrides), where
 * searches of longer reads start.  Results are the same with or without them.
 * This call frees all four and leaves the rank blocks and the sampled suffix
 * array.  Setting BWT_MI355X_COMPACT in the environment makes every build and
 * load do so.  No counterpart in the reference. */
int bwtgpu_index_compact(bwtgpu_index *ix);
/* A build or load returns as soon as what make_fmi produces (seqindex.c:161-189:
 * BWT, sampled rows, rank blocks, C[], the $ row) is in place; the arrays that
 * only speed searches up -- position records, k-mer table, 5.3 (8 for texts of
 * 2^27 bases and more, whose records hold 128 bases each) + about 4 bytes
 * per base -- are made by the first bwtgpu_search / _mms / _search_reads /
 * _align_reads call on the index (once; concurrent first calls wait for each
 * other), so an index that is built only to be written to a file never pays
 * for them.  This call makes them now, on `stream`, and waits for it.  When HBM
 * is short they are left out, which is not an error.  No counterpart in the
 * reference. */
int bwtgpu_index_prepare_search(bwtgpu_index *ix, void *stream);
/* A second copy of an index in the memory of GPU `device` (which may be the one
 * it already lives on): what make_fmi produces -- rank blocks, BWT planes,
 * sampled rows -- and, when the source keeps them, the suffix array and the
 * text words travel by peer-to-peer copies (1.67 + 4.25 bytes per base); the
 * copy makes its own search-only arrays on its first search, which on these
 * GPUs is quicker than moving them (3.1 Gbp: 68 ms against 56 GB over a link).
 * A loaded source that has not recomputed its suffix array yet hands the copy
 * the same task.  The replicated-index form of a multi-GPU search: the
 * reference's analogue is the one fm_index its four worker threads share
 * (fmitest.c:67-91).  Leaves the calling thread's current device as it was.
 * NULL on failure. */
bwtgpu_index *bwtgpu_index_replicate(const bwtgpu_index *src, int device);

/* ---- certificates (no counterpart in the reference: its lengths are int, so
 * beyond 2^31 - 1 bases there is no reference output to compare with) ---- */
/* Is d_sa[0..n] the suffix array of the packed text (row 0 = n, then all
 * suffixes in order, the end of the text smallest: histsortcomp.c:161-180)?
 * Checked on the device in O(n) with an inverse built in the workspace
 * (bwtgpu_sa_certify_workspace_bytes(n) = 4(n+1) + 256 bytes, 256-byte
 * aligned): report[0] = values out of range or row 0 != n, report[1] =
 * positions that no row, or more than one, holds, report[2] = neighbouring
 * rows in the wrong order.  All zero <=> the array is correct. */
size_t bwtgpu_sa_certify_workspace_bytes(uint32_t n);
int bwtgpu_sa_certify(const uint8_t *d_text, uint32_t n, const uint32_t *d_sa, void *d_ws,
                      size_t ws_bytes, uint64_t report[8], void *stream);
/* The same for the suffix array an index built from the text keeps, and then
 * the rest of what make_fmi (seqindex.c:161-189) derives from it: report[3] =
 * 64-base blocks whose BWT differs from text[SA[row] - 1] (sprintcbwt,
 * histsortcomp.c:220-250), report[4] = sampled rows that differ from SA
 * (seqindex.c:173-176), report[5] = 1 if the $ row does not hold suffix 0.
 * Fails with -2 on an index that keeps no suffix array.  Same workspace. */
int bwtgpu_index_certify(const bwtgpu_index *ix, const uint8_t *d_text, void *d_ws,
                         size_t ws_bytes, uint64_t report[8], void *stream);

/* ---- backward search (HP2) ---- */
/* Replaces the host loop over loc_search (seqindex.c:280-309; the worker loop
 * fmitest.c:54-65): read r is the m bytes at d_pats + r*stride, one base
 * (0..3) per byte as the reference passes them; d_sp[r], d_ep[r] are the
 * half-open SA interval exactly as loc_search leaves them, including the
 * sp >= ep values of a read that stops matching.  If d_locrow is not NULL it
 * receives, per read, the row locate() (seqindex.c:258-277) would pass to
 * unc_sa: sp, or 0xFFFFFFFF where locate() returns -1 (interval empty before
 * the last step), or 0xFFFFFFFE where sp = n+1 (the reference reads out of
 * bounds there); bwtgpu_locate_rows maps both to "no position".
 * Only the low two bits of a pattern byte are used.
 * The kernels read the reads in aligned 16-byte pieces, so up to 15 bytes in
 * front of d_pats (back to the previous 16-byte boundary) and up to 16 bytes
 * past the last read are loaded and ignored: both must be readable, which they
 * are when d_pats lies inside a device allocation whose base is 16-byte aligned
 * (hipMalloc and torch give 256) and that extends 16 bytes past the last read.
 * Workspace: bwtgpu_search_workspace_bytes(R, m) bytes, 256-byte aligned. */
size_t bwtgpu_search_workspace_bytes(uint64_t R, uint32_t m);
int bwtgpu_search(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m,
                  uint64_t stride, uint64_t R, uint32_t *d_sp, uint32_t *d_ep,
                  uint32_t *d_locrow, void *d_ws, size_t ws_bytes, void *stream);
/* The same search for reads that are already packed: read r is the (m+15)/16
 * 32-bit words at d_words + r * ((m+15)/16), base k of the read at bits
 * 2 (k % 16) of word k / 16, zero bits above the last base (the format
 * bwtgpu_pack_reads_host writes).  No workspace.  No counterpart in the
 * reference, which keeps reads one base per byte. */
int bwtgpu_search_packed(const bwtgpu_index *ix, const uint32_t *d_words, uint32_t m, uint64_t R,
                         uint32_t *d_sp, uint32_t *d_ep, uint32_t *d_locrow, void *stream);
/* loc_search and locate() of every read in one call (the pair the reference's
 * callers make per read: seqindex.c:280-309 then seqindex.c:258-277, as in
 * filetest.c and the worker loop fmitest.c:54-65): d_sp, d_ep as bwtgpu_search
 * leaves them, d_pos[r] = the position locate() returns -- SA[sp], also for an
 * interval that emptied on the very last step (the reference does not look at
 * ep there), 0xFFFFFFFF where it returns -1 or sp = n+1.  Same results as
 * bwtgpu_search with d_locrow followed by bwtgpu_locate_rows.  On an index that
 * keeps its position records the search kernel writes the positions itself: it
 * compares the read with the text at a known place, so the position of all but
 * the repeated reads costs no further load.  Other indexes run the two kernels.
 * Workspace: bwtgpu_search_locate_workspace_bytes(R, m), 256-byte aligned;
 * the packed form: bwtgpu_search_locate_packed_workspace_bytes(R). */
size_t bwtgpu_search_locate_workspace_bytes(uint64_t R, uint32_t m);
int bwtgpu_search_locate(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m,
                         uint64_t stride, uint64_t R, uint32_t *d_sp, uint32_t *d_ep,
                         uint32_t *d_pos, void *d_ws, size_t ws_bytes, void *stream);
size_t bwtgpu_search_locate_packed_workspace_bytes(uint64_t R);
int bwtgpu_search_locate_packed(const bwtgpu_index *ix, const uint32_t *d_words, uint32_t m,
                                uint64_t R, uint32_t *d_sp, uint32_t *d_ep, uint32_t *d_pos,
                                void *d_ws, size_t ws_bytes, void *stream);
/* bwtgpu_search for callers that cannot allocate or hold a workspace (code
 * under HIP graph capture, tight serving loops): no counterpart in the
 * reference.  Same inputs, outputs and results as bwtgpu_search; the packed-read
 * workspace is a 16 MiB array that lives in the library's code object on every
 * GPU the library runs on (its address on a device is resolved by bwtgpu_init,
 * or by the first call on that device).  The call itself allocates nothing,
 * which includes the search-only arrays of bwtgpu_index_prepare_search: it
 * uses them when they are there and searches on the rank blocks when not.
 * It may run while another thread's first bwtgpu_search on the same index is
 * still making those arrays: it takes them only once all of them are complete
 * (the same holds for bwtgpu_locate_rows and the suffix array of a loaded
 * index); bwtgpu_index_compact and bwtgpu_index_free are the callers' to keep
 * apart from searches in flight.
 * Reads short enough to be packed inside the search kernel (up to about 1700
 * bases at stride = m) use no scratch and any R is accepted; for longer reads
 * ((m+15)/16) * 4 * R + 512 bytes must fit the 16 MiB, else the call fails with
 * -2 and a message that says so.  All calls on one GPU share that GPU's array:
 * two calls with long reads that overlap in time on one GPU race, so keep them
 * on one stream or order them yourself.  The index must live on the current
 * device.  (With launch timing enabled, bwtgpu_timing_enable, the HIP events
 * it records are created per launch.) */
int bwtgpu_search_static(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m,
                         uint64_t stride, uint64_t R, uint32_t *d_sp, uint32_t *d_ep,
                         uint32_t *d_locrow, void *stream);
/* bwtgpu_search, and the number of loads the batch made: loads[0] 16-byte rank
 * blocks, loads[1] 16-byte text windows, loads[2] 4-byte SA/ISA entries (and
 * 4-byte starts of the next string where a table lookup needed one),
 * loads[3] 4-byte packed pattern words, loads[4] 8-byte k-mer table lookups.
 * For measurement (bytes a search
 * needs); slower than bwtgpu_search and synchronises the stream.  Workspace:
 * bwtgpu_search_workspace_bytes(R, m) + 256 bytes.  No counterpart in the
 * reference. */
int bwtgpu_search_profile(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m,
                          uint64_t stride, uint64_t R, uint32_t *d_sp, uint32_t *d_ep,
                          void *d_ws, size_t ws_bytes, uint64_t loads[5], void *stream);
/* Replaces the host loop over unc_sa (seqindex.c:244-255): d_pos[r] =
 * SA[d_rows[r]]; rows equal to 0xFFFFFFFF, or beyond n, give 0xFFFFFFFF.
 * If d_steps is not NULL the total number of LF steps taken is ADDED to it
 * (zero it first).  Workspace: bwtgpu_locate_workspace_bytes() bytes. */
size_t bwtgpu_locate_workspace_bytes(void);
int bwtgpu_locate_rows(const bwtgpu_index *ix, const uint32_t *d_rows, uint64_t R,
                       uint32_t *d_pos, unsigned long long *d_steps,
                       void *d_ws, size_t ws_bytes, void *stream);
/* Replaces the host loop over mms (seqindex.c:314-351): maximum mappable
 * suffix of each read; symbols 0..3 are bases, 5 is N (trailing Ns skipped,
 * an inner N read as the most frequent base of the current interval).
 * d_matched[r] is mms's return value, d_sp/d_ep the interval it stores.  A read
 * of Ns only returns its length and an empty interval (the reference reads
 * before the buffer there).  Patterns are read byte-wise, no workspace. */
int bwtgpu_mms(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m, uint64_t stride,
               uint64_t R, int32_t *d_matched, uint32_t *d_sp, uint32_t *d_ep, void *stream);
/* Replaces the per-read body of search_reads (search_reads.c:115-153): read r
 * is the encoded bytes d_reads[d_offsets[r] .. d_offsets[r+1]) (0..3, 5 = N).
 * The reversed read of search_reads.c:112-114 is a plain reversal, so the kernel
 * reads the same bytes from the other end: d_reversed is never read and may be
 * NULL (the parameter remains for callers that already pass a reversed copy).
 * d_out[4r..4r+3] = forward anchors, backward anchors, position of the last
 * forward anchor, of the last backward anchor (0xFFFFFFFF: none); an anchor is
 * a maximum mappable suffix of at least `anchor` bases (20 in the reference). */
int bwtgpu_search_reads(const bwtgpu_index *ix, const uint8_t *d_reads,
                        const uint8_t *d_reversed, const uint64_t *d_offsets, uint64_t R,
                        int anchor, uint32_t *d_out, void *stream);
/* The aligner's dynamic programming for a batch of pairs, one wave per pair:
 * nw_fast (smw.c:26-111; mode 0) or sw_fast (smw.c:117-200; mode 1) of read
 * piece r = d_s1[d_off1[r] .. d_off1[r+1]) (bases 0..3, 5 = N, which agrees with
 * every base) against genome piece r = d_s2[d_off2[r] .. d_off2[r+1]) (bases
 * 0..3), the way align_read_anchored (single_align.c:196, :238, :267, :294)
 * calls them.  d_ret[r] = nw_fast's return value (0 for sw_fast).  The CIGAR
 * goes out as unit operations 'M', 'I', 'D' in the order the reference's
 * stack_push calls reach the caller's stack (merging runs is stack_push's own
 * rule, stack.c:62-66): d_nops[r] of them at d_ops + d_off1[r] + d_off2[r].
 * No piece may be longer than max_len1 / max_len2, which size the launch (a
 * longer one gets d_nops[r] = 0xFFFFFFFF); 64 KiB of LDS bound them (470 x 480
 * bases still fit). */
int bwtgpu_dp_align_batch(const uint8_t *d_s1, const uint64_t *d_off1, const uint8_t *d_s2,
                          const uint64_t *d_off2, uint64_t R, int mode, uint32_t max_len1,
                          uint32_t max_len2, int32_t *d_ret, uint8_t *d_ops, uint32_t *d_nops,
                          void *stream);
/* align_read_anchored (single_align.c:163-298) for a batch of reads, one wave
 * per read: anchors by mms, the anchor extended backwards over gaps of up to
 * nine bases, nw_fast on the read's tail and head, sw_fast between anchors.
 * d_text is the packed text the index was built from (the reference's `seq`
 * argument); read r is the encoded bytes d_reads[d_offsets[r] .. d_offsets[r+1])
 * (0..3, 5 = N; single_align.c:450-474), none longer than max_len (up to 479
 * bases fit the 64 KiB of LDS).  d_pos[r] = the function's return value (0:
 * not aligned; single_align.c:484-486 prints it + 1).  The CIGAR stack is
 * returned as unit operations 'M', 'I', 'D', first pushed first
 * (stack_print_destroy, stack.c:27-37, prints from the other end): d_nops[r]
 * of them at d_ops + 5 * d_offsets[r] + 64 * r.  d_nops[r] = 0xFFFFFFFF marks a
 * read longer than max_len.  Two things the reference leaves undefined are
 * defined here: a read of at most anchor_len bases returns 0 (the reference
 * reads an unset variable, single_align.c:281), and where it calls mms on no
 * bases (fewer than ten bases left in front of an anchor, :215) d_pos[r] is
 * INT_MIN. */
int bwtgpu_align_reads(const bwtgpu_index *ix, const uint8_t *d_text, const uint8_t *d_reads,
                       const uint64_t *d_offsets, uint64_t R, int anchor_len, uint32_t max_len,
                       int32_t *d_pos, uint8_t *d_ops, uint32_t *d_nops, void *stream);

/* Occ(c, idx) for R (symbol, index) pairs: seq_rank (seqindex.c:91-111) when
 * skip_dollar_row is 0, rank (seqindex.c:221-225) when it is 1.  Indices
 * beyond n give 0xFFFFFFFF. */
int bwtgpu_occ_batch(const bwtgpu_index *ix, const uint8_t *d_c, const uint32_t *d_idx,
                     uint64_t R, int skip_dollar_row, uint32_t *d_out, void *stream);
/* lf (seqindex.c:214-219) for R rows. */
int bwtgpu_lf_batch(const bwtgpu_index *ix, const uint32_t *d_rows, uint64_t R,
                    uint32_t *d_out, void *stream);

/* ---- host-pointer forms: upload, same kernels, download (synchronous) ---- */
/* histsort / csuff_arr with the reference's own calling convention
 * (histsortcomp.c:161, csacak.c:521): host text in, n+1 host rows out. */
int bwtgpu_sa_build_host(const uint8_t *h_text, uint32_t n, uint32_t *h_sa);
/* histhelper (histsortcomp.c:35-90) on host arrays: sorts h_ids[0 .. count), starts
 * of suffixes of the packed text (an id equal to n is the empty suffix), into
 * the order those suffixes have in the suffix array.  For a run of suffixes that
 * agree on their first `depth` bases that is the reference's result at any
 * depth.  Builds the whole suffix array on the device to do so. */
int bwtgpu_sort_suffixes_host(const uint8_t *h_text, uint32_t n, uint32_t *h_ids, uint64_t count);
/* sprintcbwt (histsortcomp.c:220-250) on host arrays. */
int bwtgpu_bwt_from_sa_host(const uint8_t *h_text, const uint32_t *h_sa, uint32_t n,
                            uint8_t *h_bwt, uint32_t *endloc);
/* loc_search over R host reads (seqindex.c:280-309); h_locrow as d_locrow of
 * bwtgpu_search; with h_pos also locate() (seqindex.c:258-277) per read,
 * 0xFFFFFFFF standing for its -1.  Each output pointer may be NULL. */
int bwtgpu_search_host(const bwtgpu_index *ix, const uint8_t *h_pats, uint32_t m,
                       uint64_t stride, uint64_t R, uint32_t *h_sp, uint32_t *h_ep,
                       uint32_t *h_locrow, uint32_t *h_pos);
/* Large batches (32768 reads or more that do not overlap) stream: host threads
 * (BWT_MI355X_HOST_THREADS, default 8) pack a chunk to 2 bits per base into pinned
 * memory while the chunk before is being copied and searched and the results of
 * the one before that are copied out, so a quarter of the bytes cross PCIe and
 * nothing waits for the whole batch.
 * bwtgpu_pack_reads_host: that packing alone, for callers that keep their reads
 * packed: R reads of m bases at h_pats + r * stride -> (m+15)/16 words per read
 * (format: see bwtgpu_search_packed).
 * bwtgpu_search_packed_host: bwtgpu_search_host for reads in that format; from
 * pinned memory nothing but DMA and the kernels is left. */
int bwtgpu_pack_reads_host(const uint8_t *h_pats, uint32_t m, uint64_t stride, uint64_t R,
                           uint32_t *h_words);
int bwtgpu_search_packed_host(const bwtgpu_index *ix, const uint32_t *h_words, uint32_t m,
                              uint64_t R, uint32_t *h_sp, uint32_t *h_ep, uint32_t *h_locrow,
                              uint32_t *h_pos);
/* reverse_search (seqindex.c:228-242) per read: 0 if the interval emptied before
 * the last step, else ep - sp + 1 as the reference returns it */
int bwtgpu_reverse_search_host(const bwtgpu_index *ix, const uint8_t *h_pats, uint32_t m,
                               uint64_t stride, uint64_t R, int32_t *h_result);
int bwtgpu_locate_rows_host(const bwtgpu_index *ix, const uint32_t *h_rows, uint64_t R,
                            uint32_t *h_pos);
int bwtgpu_occ_host(const bwtgpu_index *ix, const uint8_t *h_c, const uint32_t *h_idx,
                    uint64_t R, int skip_dollar_row, uint32_t *h_out);
int bwtgpu_lf_host(const bwtgpu_index *ix, const uint32_t *h_rows, uint64_t R, uint32_t *h_out);
int bwtgpu_mms_host(const bwtgpu_index *ix, const uint8_t *h_pats, uint32_t m, uint64_t stride,
                    uint64_t R, int32_t *h_matched, uint32_t *h_sp, uint32_t *h_ep);
/* reverses the reads itself */
int bwtgpu_search_reads_host(const bwtgpu_index *ix, const uint8_t *h_reads,
                             const uint64_t *h_offsets, uint64_t R, int anchor, uint32_t *h_out);
/* bwtgpu_align_reads with everything on the host: h_text is the packed text
 * (n/4+1 bytes), h_ops has room for 5 * h_offsets[R] + 64 * R bytes (read r's
 * operations at 5 * h_offsets[r] + 64 * r). */
int bwtgpu_align_reads_host(const bwtgpu_index *ix, const uint8_t *h_text, const uint8_t *h_reads,
                            const uint64_t *h_offsets, uint64_t R, int anchor_len, int32_t *h_pos,
                            uint8_t *h_ops, uint32_t *h_nops);

/* ---- seeded synthetic inputs, generated in HBM (bench/test plumbing) ---- */
/* Bit-identical to the host generators in bwt_amd/synth.py. */
int bwtgpu_gen_genome(uint8_t *d_text, uint64_t n, uint64_t seed, void *stream);
int bwtgpu_gen_reads(const uint8_t *d_text, uint64_t n, uint8_t *d_reads,
                     uint64_t R, uint32_t m, uint64_t seed, uint32_t *d_offsets,
                     void *stream);

/* reads [first, first + R) of the set of reads `seed` names, so that ranks can
 * each generate their slice of ONE set; bwtgpu_gen_reads is first = 0 */
int bwtgpu_gen_reads_slice(const uint8_t *d_text, uint64_t n, uint8_t *d_reads, uint64_t first,
                           uint64_t R, uint32_t m, uint64_t seed, uint32_t *d_offsets,
                           void *stream);

/* ---- diagnostics: drive internal device primitives from the tests ---- */
/* Rate of random 4-byte loads (one 64-byte line each) over d_buf[0 .. bytes),
 * the ceiling for loads without locality: `loads` of them, timed with HIP events
 * on `stream`, best of three launches.  The last 4 bytes of the buffer are a
 * sink.  No counterpart in the reference. */
int bwtgpu_diag_gather_rate(const void *d_buf, uint64_t bytes, uint64_t loads,
                            double *lines_per_s, void *stream);
/* No counterpart in the reference.  op 0: exclusive add scan, 1: inclusive
 * max scan, over uint32; allocate their own scratch and synchronise. */
int bwtgpu_diag_scan(int op, const uint32_t *d_in, uint32_t *d_out, uint64_t count,
                     uint32_t *d_total, void *stream);
/* stable sort of (uint64 key, uint32 value) pairs by key bits [0, key_bits) */
int bwtgpu_diag_radix_sort(uint64_t *d_keys, uint32_t *d_vals, uint64_t count,
                           int key_bits, void *stream);

#ifdef __cplusplus
}
#endif
#endif
