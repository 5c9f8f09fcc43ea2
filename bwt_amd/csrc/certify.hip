// Certificate of a finished index, computed on the device: is the suffix array
// the index keeps THE suffix array of the text, and are its BWT, sampled rows
// and $ row the ones sprintcbwt / make_fmi (histsortcomp.c:220-250,
// seqindex.c:161-189) derive from it?  For texts beyond the reference's int
// (more than 2^31 - 1 bases) nothing else can say so; below that the tests also
// compare with the reference's own output.
//
// A sequence sa[0..n] is the suffix array of text+$ iff
//   (1) it is a permutation of 0..n with sa[0] = n (the empty suffix first), and
//   (2) for neighbouring rows i, i+1 >= 1 with a = sa[i], b = sa[i+1]:
//       text[a] < text[b], or text[a] = text[b] and rank[a+1] < rank[b+1],
//       rank being the inverse of sa (rank[n] = 0).
// (2) says each suffix is smaller than the next given that the suffixes one base
// shorter are ordered as sa says, which holds for all of them by induction on
// the length of the common prefix.  The inverse used here is written by this
// file into the caller's workspace: it does not trust the position records.
#include "common.hpp"
#include "fm_device.hpp"

namespace bwtgpu {

__device__ __forceinline__ uint32_t packed_base(const uint8_t *__restrict__ t, uint64_t i) {
  return (t[i >> 2] >> (2 * (3 - (i & 3)))) & 3u;  // histsortcomp.c:20-23
}

// rank[sa[r]] = r for values in range; out-of-range values are counted
__global__ void k_cert_scatter(const uint32_t *__restrict__ sa, uint64_t rows, uint32_t n,
                               uint32_t *__restrict__ rank, unsigned long long *__restrict__ report) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const uint32_t v = sa[r];
  if (v > n || (r == 0 && v != n)) {
    atomicAdd(&report[0], 1ull);
    return;
  }
  rank[v] = (uint32_t)r;
}

// every position must point back at itself: a value that occurs twice leaves
// another one out, whose slot still holds the fill pattern or a row that holds
// a different value
__global__ void k_cert_permutation(const uint32_t *__restrict__ sa, uint64_t rows,
                                   const uint32_t *__restrict__ rank,
                                   unsigned long long *__restrict__ report) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= rows) return;
  const uint32_t r = rank[p];
  const bool bad = r >= rows || sa[r] != (uint32_t)p;
  const uint64_t any = __ballot(bad);
  if (any && (threadIdx.x & 63u) == 0) atomicAdd(&report[1], (unsigned long long)__popcll(any));
}

__global__ void k_cert_order(const uint8_t *__restrict__ text, const uint32_t *__restrict__ sa,
                             uint64_t rows, uint32_t n, const uint32_t *__restrict__ rank,
                             unsigned long long *__restrict__ report) {
  const uint64_t i = 1 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (i + 1 < rows) {
    const uint32_t a = sa[i], b = sa[i + 1];
    if (a < n && b < n) {  // anything else has been counted by the scatter
      const uint32_t ca = packed_base(text, a), cb = packed_base(text, b);
      bad = ca > cb || (ca == cb && rank[a + 1] >= rank[b + 1]);
    } else {
      bad = true;
    }
  }
  const uint64_t any = __ballot(bad);
  if (any && (threadIdx.x & 63u) == 0) atomicAdd(&report[2], (unsigned long long)__popcll(any));
}

// BWT bit-planes of the index against text[sa[row] - 1] (one wave per 64-base
// block, as the build emits them), and the $ row
__global__ void __launch_bounds__(256)
k_cert_bwt(const uint8_t *__restrict__ text, const uint32_t *__restrict__ sa, uint32_t n, uint32_t endloc,
           const IndexBlock *__restrict__ blocks, uint64_t nblocks, unsigned long long *__restrict__ report) {
  const uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u;
  if (b >= nblocks) return;  // wave-uniform
  const uint64_t p = b * 64 + lane;
  uint32_t base = 0;
  if (p < n) {
    const uint64_t row = p + (p >= endloc ? 1 : 0);  // histsortcomp.c:238-245
    const uint32_t at = sa[row];
    base = (at >= 1 && at <= n) ? packed_base(text, (uint64_t)at - 1) : 0u;
  }
  const uint64_t lo = __ballot(base & 1u);
  const uint64_t hi = __ballot(base & 2u);
  if (lane == 0 && (blocks[b].lo != lo || blocks[b].hi != hi)) atomicAdd(&report[3], 1ull);
}

// sampled rows (every 8th in HBM; the file's every 32nd is a subset) and the $ row
__global__ void k_cert_samples(const uint32_t *__restrict__ sa, uint64_t nsamples,
                               const uint32_t *__restrict__ idxs, uint32_t endloc, uint32_t n,
                               unsigned long long *__restrict__ report) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && (endloc > n || sa[endloc] != 0)) atomicAdd(&report[5], 1ull);
  if (i >= nsamples) return;
  if (idxs[i] != sa[i << kDevSampleShift]) atomicAdd(&report[4], 1ull);
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

size_t bwtgpu_sa_certify_workspace_bytes(uint32_t n) { return ((size_t)n + 1) * 4 + 256; }

static int certify_sa(const uint8_t *d_text, uint32_t n, const uint32_t *d_sa, void *d_ws, size_t ws_bytes,
                      hipStream_t s) {
  if (!d_text || !d_sa || !d_ws) return set_error(kErrArg, "null argument");
  if (n == 0 || n > 0xFFFFFFF0u) return set_error(kErrArg, "text length %u out of range", n);
  if ((uintptr_t)d_ws & 255u) return set_error(kErrArg, "workspace must be 256-byte aligned");
  if (ws_bytes < bwtgpu_sa_certify_workspace_bytes(n))
    return set_error(kErrWorkspace, "certify workspace %zu < %zu bytes", ws_bytes,
                     bwtgpu_sa_certify_workspace_bytes(n));
  const uint64_t rows = (uint64_t)n + 1;
  unsigned long long *rep = reinterpret_cast<unsigned long long *>(d_ws);
  uint32_t *rank = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(d_ws) + 256);
  BWT_HIP(hipMemsetAsync(rep, 0, 256, s));
  BWT_HIP(hipMemsetAsync(rank, 0xFF, rows * 4, s));
  const dim3 g((unsigned)div_up(rows, 256)), b(256);
  BWT_LAUNCH("cert_scatter", k_cert_scatter, g, b, 0, s, d_sa, rows, n, rank, rep);
  BWT_LAUNCH("cert_permutation", k_cert_permutation, g, b, 0, s, d_sa, rows, rank, rep);
  BWT_LAUNCH("cert_order", k_cert_order, g, b, 0, s, d_text, d_sa, rows, n, rank, rep);
  BWT_CHECK_LAUNCH();
  return kOk;
}

static int read_report(void *d_ws, uint64_t report[8], hipStream_t s) {
  unsigned long long h[8];
  BWT_HIP(hipMemcpyAsync(h, d_ws, sizeof h, hipMemcpyDeviceToHost, s));
  BWT_HIP(hipStreamSynchronize(s));
  for (int q = 0; q < 8; ++q) report[q] = h[q];
  return kOk;
}

int bwtgpu_sa_certify(const uint8_t *d_text, uint32_t n, const uint32_t *d_sa, void *d_ws, size_t ws_bytes,
                      uint64_t report[8], void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!report) return set_error(kErrArg, "null argument");
  BWT_TRY(certify_sa(d_text, n, d_sa, d_ws, ws_bytes, s));
  return read_report(d_ws, report, s);
}

int bwtgpu_index_certify(const bwtgpu_index *ix, const uint8_t *d_text, void *d_ws, size_t ws_bytes,
                         uint64_t report[8], void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!ix || !report) return set_error(kErrArg, "null argument");
  if (!ix->d_sa) return set_error(kErrArg, "this index does not keep its suffix array");
  const uint32_t n = ix->n;
  BWT_TRY(certify_sa(d_text, n, ix->d_sa, d_ws, ws_bytes, s));
  unsigned long long *rep = reinterpret_cast<unsigned long long *>(d_ws);
  BWT_LAUNCH("cert_bwt", k_cert_bwt, dim3((unsigned)div_up(ix->nblocks * 64, 256)), dim3(256), 0, s, d_text,
             ix->d_sa, n, ix->endloc, ix->d_blocks, ix->nblocks, rep);
  const uint64_t ns = ((uint64_t)n >> kDevSampleShift) + 1;
  BWT_LAUNCH("cert_samples", k_cert_samples, dim3((unsigned)div_up(ns, 256)), dim3(256), 0, s, ix->d_sa, ns,
             ix->d_idxs, ix->endloc, n, rep);
  BWT_CHECK_LAUNCH();
  return read_report(d_ws, report, s);
}

}  // extern "C"
