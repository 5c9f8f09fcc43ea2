// Shared host-side plumbing for the gfx950 kernels: error reporting, launch
// timing, the HBM index object.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bwt_gpu.h"

namespace bwtgpu {

enum : int {
  kOk = 0,
  kErrHip = -1,       // a HIP runtime call failed
  kErrArg = -2,       // bad argument (null pointer, size out of range)
  kErrWorkspace = -3, // workspace too small
  kErrDevice = -4,    // no usable gfx950 device
  kErrInternal = -5,  // invariant violated on device (reported, never ignored)
  kErrNeedsWholeText = -6,  // a sharded suffix-array build met ties only a whole-text build orders
};

int set_error(int code, const char *fmt, ...);

// runtime.hip: device buffers that are reused across indexes (exact-size
// matches); pool_release_idle() returns the idle ones to the driver.
hipError_t pool_alloc(void **out, size_t bytes);

// Small calls through the host-buffer entry points (a batch of one is what the
// reference's scalar functions become) skip the copies: the calling thread
// keeps 64 KiB of pinned host memory that the device reads and writes in
// place, so a call is a memcpy into it, the launches and one synchronisation.
struct Mailbox {
  static constexpr size_t kBytes = 64 * 1024;
  uint8_t *host = nullptr, *dev = nullptr;  // the same memory, as the host and as the device address it
  void *ws = nullptr;                        // device scratch kept beside it, grown on demand
  hipStream_t stream = nullptr;              // the thread's own: calls from several threads do not wait for each other
  size_t ws_bytes = 0;
  int device = -1;
  bool open(int device_ordinal);             // false: no pinned memory to be had, take the copying path
  bool scratch(size_t bytes);
  template <class T> T *on_device(const void *host_ptr) const {
    return reinterpret_cast<T *>(dev + (static_cast<const uint8_t *>(host_ptr) - host));
  }
  ~Mailbox();
};
Mailbox &thread_mailbox();
void pool_free(void *p);
void pool_release_idle();
void pool_adopt_idle(void *p, size_t bytes, int device);
template <class T>
inline hipError_t pool_alloc(T **out, size_t bytes) {
  return pool_alloc(reinterpret_cast<void **>(out), bytes);
}

#define BWT_HIP(expr)                                                        \
  do {                                                                       \
    hipError_t e__ = (expr);                                                 \
    if (e__ != hipSuccess)                                                   \
      return ::bwtgpu::set_error(::bwtgpu::kErrHip, "%s: %s (%s:%d)", #expr, \
                                 hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

#define BWT_TRY(expr)           \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != 0) return rc__; \
  } while (0)

// Per-thread launch timing: when enabled, each launch goes through
// ScopedKernelTimer which records two events on the launch stream; durations
// are resolved lazily (after the stream has drained) by timing_get/report.
struct TimingState {
  bool enabled = false;
  struct Pending {
    std::string name;
    hipEvent_t a, b;
  };
  std::vector<Pending> pending;
  std::map<std::string, std::pair<uint64_t, double>> totals;
};
TimingState &timing();

class ScopedKernelTimer {
 public:
  ScopedKernelTimer(const char *name, hipStream_t s);
  ~ScopedKernelTimer();

 private:
  const char *name_;
  hipStream_t s_;
  hipEvent_t a_ = nullptr, b_ = nullptr;
  bool on_ = false;
};

#define BWT_LAUNCH(name, kernel, grid, block, shmem, stream, ...)              \
  do {                                                                         \
    ::bwtgpu::ScopedKernelTimer t__(name, stream);                             \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);       \
  } while (0)

#define BWT_CHECK_LAUNCH()                                                     \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess)                                                     \
      return ::bwtgpu::set_error(::bwtgpu::kErrHip, "kernel launch: %s (%s:%d)", \
                                 hipGetErrorString(e__), __FILE__, __LINE__);  \
  } while (0)

// One 64-base block of the HBM index: 32 bytes, one sector per rank query.
//   occ[c] = C[c] + number of c in BWT[0, 64*b)  (C folded in, so a backward
//            search step is one add-free lookup)
//   lo/hi  = bit-planes of the 64 bases of the block (bit k = base 64*b+k)
struct __attribute__((aligned(32))) IndexBlock {
  uint32_t occ[4];
  uint64_t lo, hi;
};
static_assert(sizeof(IndexBlock) == 32, "one rank = one 32-byte sector");

// Backward search reads a second, per-symbol layout: for each symbol c an
// array of 16-byte blocks, block b covering BWT positions [96 b, 96 b + 96):
//   occ = C[c] + number of c in BWT[0, 96 b)
//   m   = bit k set iff BWT[96 b + k] == c
// so C[c] + Occ(c, i) is ONE 16-byte load, a mask and three popcounts, and the
// two ends of an SA interval share that load whenever they fall in one block.
struct __attribute__((aligned(16))) SymBlock {
  uint32_t occ;
  uint32_t m[3];
};
static_assert(sizeof(SymBlock) == 16, "one rank = one 16-byte load");
constexpr uint32_t kSymBases = 96;

// The index file keeps every 32nd suffix-array row (seqindex.c:173-176); in
// HBM every 8th is kept, which cuts the LF walk of unc_sa from 31 steps on
// average to 7 for 0.375 more bytes per base.  Positions returned are the
// same: a walk stops at the first sampled row either way.
constexpr uint32_t kDevSampleShift = 3;
constexpr uint32_t kDevSampleMask = (1u << kDevSampleShift) - 1u;
constexpr uint32_t kFileSampleShift = 5;
// A walk of LF steps from any row of a sound index visits every row once before
// it returns, so it meets a stored row (row 0 at the latest) within n + 1 steps
// -- 31 on average on ordinary texts, many more on some repeats, where rows a
// multiple of the period apart share their residue.  An index assembled from
// inconsistent parts (a damaged file) may hold a cycle that never does: walks
// give up after n + 1 steps and report no position.

// The inverse suffix array lives in 64-byte records of twelve positions: the
// twelve rows, then the 64 bases of text that start at the record's first
// position (base k at bits 2k of word k / 16).  A search that has narrowed to
// one row compares the rest of its read with the text and then needs the row
// of the position the read starts at: one record -- one cache line -- holds
// both for reads that start in it and are up to 53 bases long; every further
// 48 bases cost one more record (48 = 4 records, so the offset stays).  Four
// guard records of zeros come first (a read may overhang the start of the
// text), i.e. position p is slot p % 12 of record p / 12 + 4.
struct __attribute__((aligned(64))) PosRecord {
  uint32_t row[12];
  uint32_t text[4];
};
static_assert(sizeof(PosRecord) == 64, "one position record = one cache line");
constexpr uint32_t kPosPerRecord = 12;
constexpr uint32_t kPosGuardRecords = 4;
// largest text whose position + guard still fits 32 bits
constexpr uint32_t kPosRecordMaxN = 0xFFFFFF00u;
// Records are kept in groups of eight in the order 0 4 1 5 2 6 3 7: a read of more
// than 53 bases needs records q and q + 4 (48 bases further), and those are then the
// two halves of one 128-byte-aligned pair for every second q -- the second half of a
// pair costs a fifth of a line fetched from elsewhere.
__host__ __device__ inline uint64_t pos_record_place(uint64_t b) {
  return (b & ~7ull) | ((b & 3ull) << 1) | ((b >> 2) & 1ull);
}
// Wide records, for texts whose records outgrow the last-level cache anyway (where reads
// are mostly longer than 53 bases as well): eight positions and the 128 bases from the
// record's first position on, so one line serves a read of up to 121 bases that starts
// in it, and every further 112 bases (14 records: the offset stays) one more.  Fourteen
// guard records; records in their natural order.  8 bytes a base instead of 5.33.
constexpr uint32_t kWidePosPerRecord = 8;
constexpr uint32_t kWideGuardRecords = 14;
inline uint64_t pos_record_count(uint32_t n, bool wide) {
  if (wide) return (uint64_t)n / kWidePosPerRecord + kWideGuardRecords + 2;
  return ((uint64_t)n / kPosPerRecord + kPosGuardRecords + 2 + 7) / 8 * 8;  // whole groups
}

// Entries of the list that follows the k-mer table (see index_build_kmer_table):
// one per suffix shorter than the table's strings, so at most 14.
constexpr uint32_t kKmerMaxGaps = 16;

}  // namespace bwtgpu

struct bwtgpu_index {
  uint32_t n = 0;        // bases
  uint32_t endloc = 0;   // row of the (unstored) $ in the BWT
  uint32_t C[5] = {0, 0, 0, 0, 0};
  uint64_t nblocks = 0;  // n/64 + 1
  bwtgpu::IndexBlock *d_blocks = nullptr;
  uint64_t nsym = 0;     // blocks per symbol in d_sym: n/96 + 1
  bwtgpu::SymBlock *d_sym = nullptr;  // [4][nsym]
  uint32_t *d_idxs = nullptr;  // 1 + n/8 sampled SA rows: SA[8 i]
  bool has_samples = true;     // false: loaded without sampled rows, locate is refused
  // Optional, kept when the index was built from the text and HBM allows
  // (9.6 more bytes per base): the whole suffix array, its inverse, and the
  // text as 16-base little-endian words behind one guard word.  With them a
  // search finishes by comparing the read with the text once its interval is a
  // single row, and locate is one lookup.
  uint32_t *d_sa = nullptr;     // n+1 rows
  uint32_t *d_isa = nullptr;    // position -> row, as pos_record_count(n, rec_wide) PosRecords (with their text)
  bool rec_wide = false;        // the records are the wide kind (set before d_isa is published)
  uint32_t *d_text2 = nullptr;  // [0] guard, [1 + w] = bases 16w..16w+15, base k at bits 2k
  // Optional: where the SA interval of every string of kmer_k bases starts, i.e.
  // the state of a backward search after its first kmer_k steps, indexed by the
  // string read as a base-4 number (first base most significant).  The interval
  // ends where the next string's starts, less the rows the gap list that follows
  // the table names.  A search of a read of at least kmer_k
  // bases starts there.
  uint32_t *d_kmer = nullptr;
  uint32_t kmer_k = 0;
  // Beside a two-word table: the level two bases up, [start, end) of every string of
  // kmer_k - 2 bases as the build computed it (no gap list: both ends are stored).  A
  // read whose last kmer_k bases do not occur starts there instead of at its last base.
  uint2 *d_kmer2 = nullptr;
  uint2 *d_kmer4 = nullptr;  // the same four bases up (strings of kmer_k - 4 bases), for when that level has no rows either
  // Words per table entry: 1 = the interval start alone; 2 = the start and, next to it,
  // SA[start] -- the text position of the string when it occurs exactly once (half of
  // all strings of ceil(log4 n) bases that occur at all), which saves such a search the
  // suffix-array line.  Two words when the index keeps its suffix array.  For a string
  // that occurs a few times (2 .. 15 rows) the second word holds the BWT symbols of its
  // rows instead (kmer_rows_with, fm_device.hpp): a search whose next base goes on with
  // exactly one of them is down to one row without a rank query.
  uint32_t kmer_w = 1;
  // Two-word entries, when text positions leave bits of the second word free (n < 2^30):
  // its top kmer_sizebits bits hold the number of rows of the interval, capped at
  // 2^kmer_sizebits - 1 ("that many or more: see where the next string starts"), so that
  // a lookup is the entry's own 8 bytes and never the next entry's line.  0: not stored.
  uint32_t kmer_sizebits = 0;
  int device = 0;
  // What make_fmi (seqindex.c:161-189) produces is complete when a build or a
  // load returns.  The arrays that only speed searches up (position records,
  // k-mer table) are made by index_prepare_search, which the search entry
  // points call on first use: a caller that builds an index to write it to a
  // file never pays for them.  want_*: the array can be and is to be made.
  bool want_records = false;  // d_sa and d_text2 are in place
  bool want_kmer = false;
  // A loaded index (BWT + sampled rows): recompute the suffix array and the text
  // from them first, so that it gets the same search-only arrays as a built one.
  bool want_recover = false;
  mutable std::atomic<bool> search_ready{true};  // nothing left for index_prepare_search to do
  mutable std::mutex prepare_mu;
  // Entry points that neither take prepare_mu nor wait for the search-only arrays
  // (bwtgpu_search_static, bwtgpu_locate_rows) may run while another thread is
  // making those arrays.  The rule that keeps them safe: an array's pointer is
  // stored in the index only when the array is complete and the stream that
  // filled it has been waited for (bwtgpu::publish), and such readers take their
  // view with view_published (fm_device.hpp), which hands out the search-only
  // arrays only once search_ready says that all of them are in place.
};

namespace bwtgpu {

constexpr int kWave = 64;

inline uint64_t div_up(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
// Stores a finished array's pointer in the index (see bwtgpu_index::prepare_mu).
template <class T>
inline void publish(T *&slot, T *value) { __atomic_store_n(&slot, value, __ATOMIC_RELEASE); }
inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// search.hip: resolves, once per device and with that device current, the
// address of the static scratch array bwtgpu_search_static uses.
int static_scratch_setup(int device, void **out = nullptr);

// hoststream.hip: host batches through pinned staging, packed on the way
bool host_batch_prefers_packing(uint32_t m, uint64_t stride, uint64_t R);
// search.hip: walks along LF from sampled rows (k_chain_walk); dense: every 8th row from every 32nd
// (d_out = the dense samples), else suffix array and text from every 8th row.  Waits for s.
struct IndexView;
int chain_walk(const IndexView &v, bool dense, const uint32_t *d_samples, uint64_t chains, uint32_t *d_out,
               uint32_t *d_text2, uint64_t expect, bool *consistent, hipStream_t s);
// search.hip: the packed-read search with both optional outputs, rows (d_locrow) and
// positions (d_pos; only an index with position records, ix->d_isa, can fill those)
int search_packed_dev(const bwtgpu_index *ix, const uint32_t *d_words, uint32_t m, uint64_t R, uint32_t *d_sp,
                      uint32_t *d_ep, uint32_t *d_locrow, uint32_t *d_pos, hipStream_t s);
int search_host_streamed(const bwtgpu_index *ix, const uint8_t *h_pats, uint32_t m, uint64_t stride, uint64_t R,
                         uint32_t *h_sp, uint32_t *h_ep, uint32_t *h_locrow, uint32_t *h_pos);
void pinned_release_idle();

// index.hip
// `readback`: 8 words of pinned host memory for C[] to land in, or null (the copy
// into pageable memory then waits for the stream by itself).  Waits for `s`.
// A few words a kernel at the end of a build copies to pinned host memory for the
// caller (each as a copy of its own is a stream operation of some microseconds).
struct HostWords {
  uint32_t *dst = nullptr;
  const uint32_t *src = nullptr;
  uint32_t words = 0;
};
// readback: 16 words of pinned host memory that kernels can write, or null; C[] is
// then stored there by the scan itself, and `more` (two entries, or null) with it.
int index_finish_from_planes(bwtgpu_index *ix, hipStream_t s, uint32_t *readback = nullptr,
                             const HostWords *more = nullptr);
int index_build_kmer_table(bwtgpu_index *ix, hipStream_t s);  // after endloc and C[] are known
// Makes the search-only arrays an index still lacks (see bwtgpu_index); safe to
// call from several threads, returns at once when there is nothing to do, and
// waits for `s` when it did something, so that any stream may search afterwards.
int index_prepare_search(const bwtgpu_index *ix, hipStream_t s);
inline int index_ready_for_search(const bwtgpu_index *ix, hipStream_t s) {
  return ix->search_ready.load(std::memory_order_acquire) ? 0 : index_prepare_search(ix, s);
}
void index_defer_search_arrays(bwtgpu_index *ix);
// fm_build.hip: the position records of an index that holds d_sa and d_text2
int index_build_records(bwtgpu_index *ix, hipStream_t s);
// search.hip: d_sa and d_text2 of an index that was loaded from its parts
int index_recover_arrays(bwtgpu_index *ix, hipStream_t s);
int index_alloc(bwtgpu_index **out, uint32_t n);

// sa_build.hip
// By-products of a build whose rows all came out of the leaf sort (no prefix
// doubling): the BWT symbol of every row of SA+1, and the $ row.  Both live in
// the workspace; null when not available.
struct SaExtras {
  const uint8_t *bw = nullptr;
  const uint32_t *endloc = nullptr;
  // in: where the leaf sort may drop every 8th row as it writes it (SA[8 i] at
  // samples[i]), or null; out: whether all of them were written
  uint32_t *samples = nullptr;
  bool samples_done = false;
  // in: room for the text in little-endian words (bwtgpu_index::d_text2, n/16 + 4 words), or
  // null; out: whether the pass that re-deals the text for the sort filled it in
  uint32_t *text2 = nullptr;
  bool text2_done = false;
  // in: 16 words of pinned host memory, or null.  With it a whole-text build
  // does not wait to learn whether the leaf sort left rows tied: it hands out
  // the by-products as if none were, queues the copy of its counters into
  // these words and sets `pending`; the caller goes on queueing work on the
  // same stream and calls sa_build_settle after its next synchronisation.
  uint32_t *readback = nullptr;
  bool pending = false;
  // in: with `pending`, the caller copies the sixteen counter words from later.ctr to
  // `readback` itself, in a kernel of its own, before that synchronisation
  bool caller_copies_counters = false;
  struct {  // what sa_build_settle needs of the workspace
    uint32_t *d_sa = nullptr, *ghead = nullptr, *flags = nullptr;
    void *leaves = nullptr, *leaf_ties = nullptr, *ctr = nullptr;
    uint32_t n = 0, nleaves = 0;
    // the arguments of the sort of leaves the one-wave sort put off (see sa_build_settle)
    const uint32_t *tb = nullptr, *bufB = nullptr, *keyA = nullptr, *keyB = nullptr, *deferred = nullptr;
    uint32_t *bufA = nullptr, *samples = nullptr;
    uint8_t *bw = nullptr;
  } later;
};
// 0: no row was left tied, the by-products stand.  1: tied rows were sorted by
// prefix doubling just now (the stream has been waited for); the suffix array is
// final, the by-products are not to be used.  Negative: error.
int sa_build_settle(SaExtras *extras, hipStream_t s);
int sa_build_impl(const uint8_t *d_text, uint32_t n, uint32_t *d_sa, void *d_ws,
                  size_t ws_bytes, hipStream_t s, SaExtras *extras);
// After a build: a part of its workspace that no by-product lives in, at
// least 8 (n+1) + 2048 bytes, or null for tiny n.
void *sa_workspace_free_region(uint32_t n, void *d_ws);
// One shard of a build: the suffixes whose first four bases, read as a number
// with the first base most significant, lie in [dlo, dhi).
struct SaRange {
  uint32_t dlo = 0, dhi = 256;  // in
  uint64_t capacity = 0;        // in: rows d_sa has room for
  uint64_t row_begin = 0;       // out: suffix-array row of the shard's first row
  uint64_t rows = 0;            // out
  uint32_t endloc = 0;          // out: row of suffix 0 if the shard holds it, else 0
};
int sa_build_core(const uint8_t *d_text, uint32_t n, uint32_t *d_sa, SaRange *range, void *d_ws,
                  size_t ws_bytes, hipStream_t s, SaExtras *extras);

}  // namespace bwtgpu
