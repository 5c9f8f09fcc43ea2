// Maximum-mappable-suffix search (mms, seqindex.c:314-351) and the anchor loop
// of search_reads (search_reads.c:102-160), one read per lane.
#include <cstring>

#include "common.hpp"
#include "fm_device.hpp"
#include "mms_device.hpp"

namespace bwtgpu {

// The staged bytes are read through generic pointers (a pattern may lie in LDS
// or in HBM), i.e. by flat loads, and a read that walks a buffer backwards is
// compiled to `flat_load base offset:k` with base = address - k.  The hardware
// decides from `base`, not from base + offset, which aperture a flat access
// belongs to: a base one byte below the start of the LDS aperture is a memory
// aperture violation even though the byte loaded lies inside it.  So nothing is
// ever staged at the very start of LDS: the first kStageHead bytes stay unused.
constexpr uint32_t kStageHead = 64;

// Bytes [lo, hi) of a global buffer into LDS (behind kStageHead unused bytes)
// with 16-byte loads where the alignment allows; returns where byte lo landed.
// All threads of the workgroup call it; ends with a barrier.
__device__ __forceinline__ const uint8_t *stage_bytes(const uint8_t *__restrict__ g, uint64_t lo, uint64_t hi,
                                                      uint8_t *lds) {
  const uint8_t *src = g + lo;
  const uint32_t mis = (uint32_t)((uintptr_t)src & 15u);
  const uint8_t *asrc = src - mis;  // same 16-byte granule as src: inside the allocation
  const uint32_t total = mis + (uint32_t)(hi - lo), vecs = total / 16;
  uint8_t *dst = lds + kStageHead;
  for (uint32_t i = threadIdx.x; i < vecs; i += blockDim.x)
    reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(asrc)[i];
  for (uint32_t i = vecs * 16 + threadIdx.x; i < total; i += blockDim.x) dst[i] = asrc[i];
  __syncthreads();
  return dst + mis;
}

// lds_bytes != 0: the workgroup's patterns are staged in LDS first (coalesced
// loads instead of one byte per lane and step).
__global__ void __launch_bounds__(256)
k_mms_batch(IndexView v, CTab C, const uint8_t *__restrict__ pats, uint32_t m, uint64_t stride,
            uint64_t R, int32_t *__restrict__ matched, uint32_t *__restrict__ d_sp,
            uint32_t *__restrict__ d_ep, uint32_t lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint64_t r0 = (uint64_t)blockIdx.x * blockDim.x;
  const uint64_t r = r0 + threadIdx.x;
  const uint8_t *mine = pats + r * stride;
  if (lds_bytes) {
    const uint64_t last = R - 1 < r0 + blockDim.x - 1 ? R - 1 : r0 + blockDim.x - 1;
    const uint8_t *base = stage_bytes(pats, r0 * stride, last * stride + m, lds);
    mine = base + (uint32_t)(threadIdx.x * stride);
  }
  if (r >= R) return;
  uint32_t sp, ep;
  matched[r] = mms_dev(v, C, mine, 1, (int)m, &sp, &ep, 0);
  d_sp[r] = sp;
  d_ep[r] = ep;
}

// The per-read loop of search_reads (search_reads.c:115-153) on an encoded
// read and on its reverse: repeatedly take the maximum mappable suffix; an
// anchor of at least `anchor` bases counts, is located (unc_sa of the first
// row of its interval) and is cut off, otherwise one base is cut off.
// out[4r..4r+3] = forward anchors, backward anchors, last forward position,
// last backward position (positions 0xFFFFFFFF when there was no anchor).
__global__ void __launch_bounds__(256)
k_search_reads(IndexView v, CTab C, const uint8_t *__restrict__ reads, const uint64_t *__restrict__ offsets,
               uint64_t R, int anchor, uint32_t *__restrict__ out, uint32_t lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint64_t r0 = (uint64_t)blockIdx.x * blockDim.x;
  const uint64_t r = r0 + threadIdx.x;
  const uint64_t rl = r < R ? r : R - 1;  // lanes past the end stage nothing of their own
  const uint64_t o = offsets[rl];
  const int full = (int)(offsets[rl + 1] - o);
  // the reversed read (search_reads.c:112-114 reverses, it does not complement) is
  // the read's own bytes taken from the other end: no second copy is ever read
  const uint8_t *fwd = reads + o;
  if (lds_bytes) {
    // the workgroup's reads lie back to back: one staged copy serves both directions
    const uint64_t lo = offsets[r0], hi = offsets[R < r0 + blockDim.x ? R : r0 + blockDim.x];
    const uint8_t *base = stage_bytes(reads, lo, hi, lds);
    fwd = base + (uint32_t)(o - lo);
  }
  const uint8_t *bwd = fwd + (full - 1);
  const int bstep = -1;
  if (r >= R) return;
  uint32_t res[4] = {0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu};
  for (int dir = 0; dir < 2; ++dir) {
    const uint8_t *p = dir ? bwd : fwd;
    const int step = dir ? bstep : 1;
    int len = full;
    // the table index of the last K bases is kept from one attempt to the next:
    // a failed attempt shortens the read by one base, which shifts the window
    // by one (most attempts on the reversed read fail at the table)
    const int K = (int)v.kmer_k;
    int hint = 0, n_in_window = 0;
    uint32_t hx = 0;
    while (len > anchor) {  // search_reads.c:115
      if (K && len >= K && !hint) {
        hx = 0;
        n_in_window = 0;
        for (int k = 0; k < K; ++k) {
          const uint32_t raw = p[(len - K + k) * step];
          n_in_window += raw == 5;
          hx = (hx << 2) | (raw & 3u);
        }
        hint = n_in_window ? 2 : 1;
      }
      uint32_t sp, ep;
      const int got = mms_dev(v, C, p, step, len, &sp, &ep, anchor, hint, hx);
      if (got >= anchor) {  // :119-127
        res[dir]++;
        len -= got;
        res[2 + dir] = sp > v.n ? 0xFFFFFFFFu : v.sa ? __builtin_nontemporal_load(v.sa + sp) : unc_sa_dev(v, sp);
        hint = 0;
      } else {
        len -= 1;  // :130
        if (hint && len >= K) {  // base `len` leaves the window, base len - K enters it
          const uint32_t in = p[(len - K) * step];
          n_in_window += (int)(in == 5) - (int)(p[len * step] == 5);
          hx = (hx >> 2) | ((in & 3u) << (2 * (K - 1)));
          hint = n_in_window ? 2 : 1;
        } else {
          hint = 0;
        }
      }
    }
  }
  reinterpret_cast<uint4 *>(out)[r] = make_uint4(res[0], res[1], res[2], res[3]);
}

constexpr uint32_t kMmsLdsBytes = 40 * 1024;

// largest number of bytes the reads of one workgroup (nt consecutive reads) span
__global__ void k_max_group_span(const uint64_t *__restrict__ offsets, uint64_t R, uint32_t nt,
                                 unsigned long long *__restrict__ out) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t r0 = g * nt;
  if (r0 >= R) return;
  const uint64_t r1 = r0 + nt < R ? r0 + nt : R;
  atomicMax(out, (unsigned long long)(offsets[r1] - offsets[r0]));
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

int bwtgpu_mms(const bwtgpu_index *ix, const uint8_t *d_pats, uint32_t m, uint64_t stride, uint64_t R,
               int32_t *d_matched, uint32_t *d_sp, uint32_t *d_ep, void *stream) {
  if (!ix || !d_pats || !d_matched || !d_sp || !d_ep) return set_error(kErrArg, "null argument");
  if (m == 0 || m > 0x7FFFFFFFu) return set_error(kErrArg, "pattern length out of range");
  if (R == 0) return kOk;
  BWT_TRY(index_ready_for_search(ix, (hipStream_t)stream));
  CTab C;
  for (int i = 0; i < 5; ++i) C.c[i] = ix->C[i];
  // patterns of a workgroup through LDS when they fit (smaller workgroups for long patterns)
  uint32_t nt = 256, lds_bytes = 0;
  for (; nt >= 64; nt /= 2) {
    const uint64_t need = align_up((uint64_t)(nt - 1) * stride + m + 16, 16);
    if (need <= kMmsLdsBytes) {
      lds_bytes = (uint32_t)need;
      break;
    }
  }
  if (!lds_bytes) nt = 256;
  BWT_LAUNCH("fm_mms", k_mms_batch, dim3((unsigned)div_up(R, nt)), dim3(nt), lds_bytes ? lds_bytes + kStageHead : 0,
             (hipStream_t)stream,
             view_of(ix), C, d_pats, m, stride, R, d_matched, d_sp, d_ep, lds_bytes);
  BWT_CHECK_LAUNCH();
  return kOk;
}

int bwtgpu_search_reads(const bwtgpu_index *ix, const uint8_t *d_reads, const uint8_t *d_reversed,
                        const uint64_t *d_offsets, uint64_t R, int anchor, uint32_t *d_out,
                        void *stream) {
  (void)d_reversed;  // never read: see k_search_reads
  if (!ix || !d_reads || !d_offsets || !d_out) return set_error(kErrArg, "null argument");
  if (anchor < 1) return set_error(kErrArg, "anchor length must be positive");
  if (R == 0) return kOk;
  if (!ix->d_sa && !ix->has_samples)
    return set_error(kErrArg, "this index was loaded without sampled suffix-array rows: it cannot locate anchors");
  BWT_TRY(index_ready_for_search(ix, (hipStream_t)stream));
  CTab C;
  for (int i = 0; i < 5; ++i) C.c[i] = ix->C[i];
  hipStream_t s = (hipStream_t)stream;
  // reads of a workgroup through LDS when the longest group fits: needs one number from the device
  unsigned long long *d_span = nullptr, h_span = 0;
  BWT_HIP(pool_alloc(&d_span, 256));
  hipError_t e = hipMemsetAsync(d_span, 0, 8, s);
  if (e == hipSuccess) {
    const uint64_t groups = div_up(R, 256);
    BWT_LAUNCH("fm_search_reads_span", k_max_group_span, dim3((unsigned)div_up(groups, 256)), dim3(256), 0, s,
               d_offsets, R, 256u, d_span);
    e = hipMemcpyAsync(&h_span, d_span, 8, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  pool_free(d_span);
  if (e != hipSuccess) return set_error(kErrHip, "search_reads: %s", hipGetErrorString(e));
  const uint64_t need = align_up(h_span + 32, 16);
  const uint32_t lds_bytes = need <= kMmsLdsBytes ? (uint32_t)need : 0u;
  BWT_LAUNCH("fm_search_reads", k_search_reads, dim3((unsigned)div_up(R, 256)), dim3(256),
             lds_bytes ? lds_bytes + kStageHead : 0, s,
             view_of(ix), C, d_reads, d_offsets, R, anchor, d_out, lds_bytes);
  BWT_CHECK_LAUNCH();
  return kOk;
}

int bwtgpu_mms_host(const bwtgpu_index *ix, const uint8_t *h_pats, uint32_t m, uint64_t stride,
                    uint64_t R, int32_t *h_matched, uint32_t *h_sp, uint32_t *h_ep) {
  if (!ix || !h_pats || !h_matched || !h_sp || !h_ep) return set_error(kErrArg, "null argument");
  if (R == 0) return kOk;
  BWT_HIP(hipSetDevice(ix->device));
  const size_t pb = (R - 1) * stride + m;
  {
    // a few patterns (the reference's mms is a batch of one): through the thread's mailbox, no copies
    Mailbox &box = thread_mailbox();
    if (R <= 256 && pb <= 32 * 1024 && box.open(ix->device)) {
      memcpy(box.host, h_pats, pb);
      uint32_t *res = reinterpret_cast<uint32_t *>(box.host + 32 * 1024 + 256);
      uint32_t *d_res = box.on_device<uint32_t>(res);
      BWT_TRY(bwtgpu_mms(ix, box.dev, m, stride, R, reinterpret_cast<int32_t *>(d_res), d_res + 256, d_res + 512,
                         box.stream));
      BWT_HIP(hipStreamSynchronize(box.stream));
      memcpy(h_matched, res, R * 4);
      memcpy(h_sp, res + 256, R * 4);
      memcpy(h_ep, res + 512, R * 4);
      return kOk;
    }
  }
  void *pats = nullptr, *out = nullptr;
  hipError_t e = hipMalloc(&pats, pb);
  if (e == hipSuccess) e = hipMalloc(&out, R * 12);
  int rc = kOk;
  if (e != hipSuccess) {
    rc = set_error(kErrHip, "mms buffers: %s", hipGetErrorString(e));
  } else {
    int32_t *dm = (int32_t *)out;
    uint32_t *ds = (uint32_t *)out + R, *de = (uint32_t *)out + 2 * R;
    e = hipMemcpy(pats, h_pats, pb, hipMemcpyHostToDevice);
    if (e == hipSuccess) rc = bwtgpu_mms(ix, (const uint8_t *)pats, m, stride, R, dm, ds, de, nullptr);
    if (e == hipSuccess && rc == kOk) e = hipMemcpy(h_matched, dm, R * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && rc == kOk) e = hipMemcpy(h_sp, ds, R * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && rc == kOk) e = hipMemcpy(h_ep, de, R * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = set_error(kErrHip, "mms copies: %s", hipGetErrorString(e));
  }
  if (pats) (void)hipFree(pats);
  if (out) (void)hipFree(out);
  return rc;
}

int bwtgpu_search_reads_host(const bwtgpu_index *ix, const uint8_t *h_reads, const uint64_t *h_offsets,
                             uint64_t R, int anchor, uint32_t *h_out) {
  if (!ix || !h_reads || !h_offsets || !h_out) return set_error(kErrArg, "null argument");
  if (R == 0) return kOk;
  BWT_HIP(hipSetDevice(ix->device));
  const uint64_t total = h_offsets[R];
  void *d_reads = nullptr, *d_off = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc(&d_reads, total + 16);
  if (e == hipSuccess) e = hipMalloc(&d_off, (R + 1) * 8);
  if (e == hipSuccess) e = hipMalloc(&d_out, R * 16);
  int rc = kOk;
  if (e == hipSuccess) e = hipMemcpy(d_reads, h_reads, total, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_off, h_offsets, (R + 1) * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    rc = bwtgpu_search_reads(ix, (const uint8_t *)d_reads, nullptr, (const uint64_t *)d_off,
                             R, anchor, (uint32_t *)d_out, nullptr);
  if (e == hipSuccess && rc == kOk) e = hipMemcpy(h_out, d_out, R * 16, hipMemcpyDeviceToHost);
  if (e != hipSuccess) rc = set_error(kErrHip, "search_reads buffers: %s", hipGetErrorString(e));
  if (d_reads) (void)hipFree(d_reads);
  if (d_off) (void)hipFree(d_off);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

}  // extern "C"
