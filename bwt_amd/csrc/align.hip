// align_read_anchored (single_align.c:163-298) for a batch of reads, one wave
// per read: the control flow is the reference's, run by the whole wave in step
// (every lane holds the same state); the rank queries of mms and unc_sa are
// therefore one request per wave, and the dynamic programming (nw_fast /
// sw_fast) is spread over the lanes (dp_device.hpp).
#include "common.hpp"
#include "dp_device.hpp"
#include "fm_device.hpp"
#include "mms_device.hpp"

namespace bwtgpu {

constexpr int kAlignUndefined = -2147483647 - 1;

// base idx of the packed text, idx below 0 read as 0 (single_align.c:19-23)
__device__ __forceinline__ uint32_t align_base(const uint8_t *__restrict__ text, int64_t idx) {
  const uint64_t i = idx < 0 ? 0ull : (uint64_t)idx;
  return (uint32_t)(text[i >> 2] >> (2u * (3u - (uint32_t)(i & 3u)))) & 3u;
}

__device__ __forceinline__ int64_t align_unc_sa(const IndexView &v, uint32_t row) {
  return v.sa ? (int64_t)__builtin_nontemporal_load(v.sa + row) : (int64_t)unc_sa_dev(v, row);
}

// room for the unit operations of a read of len bases
__host__ __device__ inline uint64_t align_ops_capacity(uint64_t len) { return 5 * len + 64; }

__global__ void __launch_bounds__(64)
k_align_reads(IndexView v, CTab C, const uint8_t *__restrict__ text, const uint8_t *__restrict__ reads,
              const uint64_t *__restrict__ offsets, int anchor_len, uint32_t max_len, DpLds L,
              int32_t *__restrict__ pos_out, uint8_t *__restrict__ ops_all, uint32_t *__restrict__ nops_out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t *pat = lds + L.total;  // the read, behind the matrix area
  const uint64_t r = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const uint64_t o = offsets[r];
  const uint64_t full = offsets[r + 1] - o;
  if (full > max_len) {  // not what the launch was sized for
    if (lane == 0) {
      pos_out[r] = 0;
      nops_out[r] = 0xFFFFFFFFu;
    }
    return;
  }
  const int olen = (int)full;
  for (int i = (int)lane; i < olen; i += 64) pat[i] = reads[o + i];
  __syncthreads();
  uint8_t *out = ops_all + 5 * o + 64 * r;
  const uint32_t cap = (uint32_t)align_ops_capacity(full);
  uint32_t n_ops = 0;
  bool overflow = false;
  // stack_push of `count` times c (stack.c:62-86; runs merge there, the unit sequence is the same)
  auto push = [&](uint8_t c, int count) {
    if (n_ops + (uint32_t)count > cap) {
      overflow = true;
      return;
    }
    for (int q = (int)lane; q < count; q += 64) out[n_ops + q] = c;
    n_ops += (uint32_t)count;
  };
  // nw_fast / sw_fast of read bases pat[p0 + i * pstep], i < len1, against genome
  // bases g0 + j * gstep, j < len2, onto the stack
  auto dp = [&](int p0, int pstep, int len1, int64_t g0, int gstep, int len2, int mode) -> int {
    // nw_fast returns at once for an empty read piece (smw.c:28-30), whatever the genome piece: the
    // first anchor of a read has no tail, and its genome piece can come out negative when the anchor's
    // length counts trailing Ns that mms skipped past the end of the text
    if (mode == 0 && len1 == 0) return 0;
    if (len2 < 0 || n_ops + (uint32_t)(len1 + len2) > cap) {
      overflow = true;
      return 0;
    }
    uint32_t k = 0;
    const int rv = dp_wave([&](int i) -> uint32_t { return pat[p0 + i * pstep]; }, len1,
                           [&](int j) -> uint32_t { return align_base(text, g0 + (int64_t)j * gstep); }, len2,
                           mode, L, lds, out + n_ops, &k);
    n_ops += k;
    return rv;
  };

  int len = olen;
  int anchmisses = len / 10, nmisses = 0;  // nmisses: the reference reads it unset for reads of up to anchor_len bases
  int curgap = 0, anchlen = 0, result = 0;
  int64_t curpos = -1;
  int done = 0;
  while (len > anchor_len && anchmisses > 0 && !done) {  // single_align.c:173
    nmisses = 0;
    while (len > anchor_len && anchmisses > 0) {  // :175-206
      uint32_t sp, ep;
      const int seglen = mms_dev<true>(v, C, pat, 1, len, &sp, &ep, 0);
      if (seglen < anchor_len || (int64_t)ep - (int64_t)sp > 1) {
        anchmisses--;
        len -= 3;
        continue;
      }
      len -= seglen;
      anchlen = seglen;
      nmisses = olen / 5;
      curpos = align_unc_sa(v, sp);
      int buflen = 10 + (olen - (len + seglen));
      if ((int64_t)buflen + curpos + seglen > (int64_t)v.n) buflen = (int)((int64_t)v.n - curpos - seglen);
      dp(len + seglen, 1, olen - (len + seglen), curpos + seglen, 1, buflen, 0);  // the tail of the read
      push('M', seglen);
      break;
    }
    if (nmisses < 1) continue;
    while (len > nmisses && len > 4 && nmisses > 0 && !done) {  // :212-255: extend the anchor backwards
      for (curgap = 1; curgap < 10; ++curgap) {
        if (len - curgap <= 0) {  // the reference calls mms on no bases here: undefined
          result = kAlignUndefined;
          done = 1;
          break;
        }
        uint32_t start, end;
        const int seglen = mms_dev<true>(v, C, pat, 1, len - curgap, &start, &end, 0);
        bool matched = false;
        // the rows of the interval in order, 64 at a time, one per lane; the first
        // that lies within 3 + curgap of the anchor is taken (:217-218)
        for (uint64_t first = start; first < end && !matched; first += 64) {
          const uint64_t i = first + lane;
          int here32 = 0;
          bool near = false;
          if (i < end) {
            const int64_t h = align_unc_sa(v, (uint32_t)i);
            int64_t dist = h + seglen - curpos;
            if (dist < 0) dist = -dist;
            near = dist - curgap <= 3;
            here32 = (int)h;  // texts of up to 2^31 - 1 bases
          }
          const uint64_t hits = __ballot(near);
          if (hits) {
            const int64_t here = __shfl(here32, __ffsll((unsigned long long)hits) - 1, 64);
            nmisses -= curgap;
            matched = true;
            const int64_t buflen = curpos - (here + seglen);
            if (buflen < 0) push('I', (int)-buflen);
            else dp(len - curgap, 1, curgap, here + seglen, 1, (int)buflen, 1);
            push('M', seglen);
            curpos = here;
            len -= seglen + curgap;
            curgap = 0;
          }
        }
        if (matched) break;
      }
      if (done) break;
      if (curgap) nmisses = 0;
    }
    if (done) break;
    if (nmisses > 0) {  // :256-272: the head of the read, both pieces backwards
      int buflen = len + 10;
      if (buflen > curpos) buflen = (int)curpos;
      const int x = dp(len - 1, -1, len, curpos - 1, -1, buflen, 0);
      result = (int)(curpos - x);
      done = 2;
      break;
    }
    len -= anchlen;
    anchmisses -= anchlen / 10;
    n_ops = 0;  // :279
    overflow = false;  // what did not fit went with the rest of the abandoned attempt
  }
  if (!done) {
    if (len > nmisses || nmisses < 1) {  // :281-283
      result = 0;
    } else {
      int buflen = len + 10;
      if (buflen > curpos) buflen = (int)curpos;
      dp(len - 1, -1, len, curpos - 1, -1, buflen, 0);
      result = (int)(curpos - len);  // :297
    }
  }
  if (lane == 0) {
    pos_out[r] = result;
    nops_out[r] = overflow ? 0xFFFFFFFEu : n_ops;
  }
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" int bwtgpu_align_reads(const bwtgpu_index *ix, const uint8_t *d_text, const uint8_t *d_reads,
                                  const uint64_t *d_offsets, uint64_t R, int anchor_len, uint32_t max_len,
                                  int32_t *d_pos, uint8_t *d_ops, uint32_t *d_nops, void *stream) {
  if (!ix || !d_text || !d_reads || !d_offsets || !d_pos || !d_ops || !d_nops)
    return set_error(kErrArg, "null argument");
  if (anchor_len < 1) return set_error(kErrArg, "anchor length must be positive");
  if (R == 0) return kOk;
  if (R > 0x7FFFFFFFull) return set_error(kErrArg, "too many reads for one launch");
  if (ix->n > 0x7FFFFFFFu) return set_error(kErrArg, "positions are returned as int: texts of up to 2^31 - 1 bases");
  if (max_len > 65535u) return set_error(kErrArg, "reads of up to 65535 bases");
  const DpLds L = dp_lds_layout(max_len, max_len + 10);
  const uint32_t lds_bytes = L.total + (uint32_t)align_up(max_len + 1, 4);
  if (lds_bytes > 64u * 1024u)
    return set_error(kErrArg, "reads of %u bases need %u bytes of LDS (64 KiB at most)", max_len, lds_bytes);
  if (!ix->d_sa && !ix->has_samples)
    return set_error(kErrArg, "this index was loaded without sampled suffix-array rows: it cannot locate anchors");
  BWT_TRY(index_ready_for_search(ix, (hipStream_t)stream));
  CTab C;
  for (int i = 0; i < 5; ++i) C.c[i] = ix->C[i];
  BWT_LAUNCH("align_reads", k_align_reads, dim3((unsigned)R), dim3(64), lds_bytes, (hipStream_t)stream,
             view_of(ix), C, d_text, d_reads, d_offsets, anchor_len, max_len, L, d_pos, d_ops, d_nops);
  BWT_CHECK_LAUNCH();
  return kOk;
}

extern "C" int bwtgpu_align_reads_host(const bwtgpu_index *ix, const uint8_t *h_text, const uint8_t *h_reads,
                                       const uint64_t *h_offsets, uint64_t R, int anchor_len, int32_t *h_pos,
                                       uint8_t *h_ops, uint32_t *h_nops) {
  if (!ix || !h_text || !h_reads || !h_offsets || !h_pos || !h_ops || !h_nops)
    return set_error(kErrArg, "null argument");
  if (R == 0) return kOk;
  BWT_HIP(hipSetDevice(ix->device));
  const uint64_t total = h_offsets[R];
  uint64_t longest = 0;
  for (uint64_t r = 0; r < R; ++r) {
    if (h_offsets[r + 1] < h_offsets[r]) return set_error(kErrArg, "read offsets must not decrease");
    const uint64_t len = h_offsets[r + 1] - h_offsets[r];
    longest = len > longest ? len : longest;
  }
  if (longest > 65535u) return set_error(kErrArg, "reads of up to 65535 bases");
  const size_t text_bytes = (size_t)ix->n / 4 + 1, ops_bytes = 5 * total + 64 * R;
  void *d_text = nullptr, *d_reads = nullptr, *d_off = nullptr, *d_pos = nullptr, *d_ops = nullptr, *d_nops = nullptr;
  hipError_t e = hipMalloc(&d_text, text_bytes + 16);
  if (e == hipSuccess) e = hipMalloc(&d_reads, total + 16);
  if (e == hipSuccess) e = hipMalloc(&d_off, (R + 1) * 8);
  if (e == hipSuccess) e = hipMalloc(&d_pos, R * 4);
  if (e == hipSuccess) e = hipMalloc(&d_nops, R * 4);
  if (e == hipSuccess) e = hipMalloc(&d_ops, ops_bytes + 16);
  int rc = kOk;
  if (e == hipSuccess) e = hipMemcpy(d_text, h_text, text_bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_reads, h_reads, total, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_off, h_offsets, (R + 1) * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    rc = bwtgpu_align_reads(ix, (const uint8_t *)d_text, (const uint8_t *)d_reads, (const uint64_t *)d_off, R,
                            anchor_len, (uint32_t)longest, (int32_t *)d_pos, (uint8_t *)d_ops, (uint32_t *)d_nops,
                            nullptr);
  if (e == hipSuccess && rc == kOk) e = hipMemcpy(h_pos, d_pos, R * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == kOk) e = hipMemcpy(h_nops, d_nops, R * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == kOk) e = hipMemcpy(h_ops, d_ops, ops_bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) rc = set_error(kErrHip, "align_reads buffers: %s", hipGetErrorString(e));
  if (d_text) (void)hipFree(d_text);
  if (d_reads) (void)hipFree(d_reads);
  if (d_off) (void)hipFree(d_off);
  if (d_pos) (void)hipFree(d_pos);
  if (d_nops) (void)hipFree(d_nops);
  if (d_ops) (void)hipFree(d_ops);
  return rc;
}
