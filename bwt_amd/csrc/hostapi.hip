// Host-pointer forms of the device entry points: upload, run the same kernels,
// download.  These are what the C host layer (bwt_amd/host) calls to stand in
// for the reference's host functions; the compute is still all on the GPU.
#include <cstring>

#include "common.hpp"
#include "fm_device.hpp"
#include "mms_device.hpp"
#include "radix_sort.hpp"

namespace bwtgpu {

// A device buffer for the length of one host-form call.  The calls are
// synchronous (every one ends with a blocking copy to the host), so by the time
// a buffer goes back to the pool nothing is using it; repeated calls of one
// size -- the scalar rank / lf / unc_sa of the reference API -- reuse it.
struct Dev {
  void *p = nullptr;
  ~Dev() {
    if (p) {
      (void)hipDeviceSynchronize();
      pool_free(p);
    }
  }
  hipError_t alloc(size_t bytes) { return pool_alloc(&p, bytes ? bytes : 256); }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// Layout of a small call in the thread's mailbox (see Mailbox): inputs first,
// then up to kBoxReads results of each kind.
constexpr uint64_t kBoxReads = 256;
constexpr uint32_t kFewMaxLen = 64;  // reads up to this long take the one-launch kernel
constexpr size_t kBoxInput = 32 * 1024;
constexpr size_t kBoxOut = kBoxInput + 256;  // four arrays of kBoxReads words follow

// loc_search (seqindex.c:280-309) and, when pos is given, locate's unc_sa
// (seqindex.c:265-276, 244-255) for a handful of reads in one launch: a read per
// lane, two rank queries a step, nothing else -- what a scalar call of the
// reference's API turns into.  Same outputs as bwtgpu_search + bwtgpu_locate_rows.
__global__ void __launch_bounds__(256)
k_search_locate_few(IndexView v, CTab C, const uint8_t *__restrict__ pats, uint32_t m, uint64_t stride,
                    uint64_t R, uint32_t *__restrict__ sp_out, uint32_t *__restrict__ ep_out,
                    uint32_t *__restrict__ row_out, uint32_t *__restrict__ pos_out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const uint8_t *p = pats + r * stride;
  uint32_t c = p[m - 1] & 3u;
  uint32_t sp = c_of(C, c), ep = c_of(C, c + 1);
  bool early = false;
  for (int i = (int)m - 2; i >= 0; --i) {
    if (ep <= sp) {  // seqindex.c:299-301
      early = true;
      break;
    }
    c = p[i] & 3u;
    sp = occ_sym(v, c, sp);
    ep = occ_sym(v, c, ep);
  }
  sp_out[r] = sp;
  ep_out[r] = ep;
  const uint32_t row = early ? 0xFFFFFFFFu : (sp > v.n ? 0xFFFFFFFEu : sp);
  row_out[r] = row;
  if (pos_out) pos_out[r] = row > v.n ? 0xFFFFFFFFu : (v.sa ? __builtin_nontemporal_load(v.sa + row) : unc_sa_dev(v, row));
}

// Occ(c, idx) on the packed BWT (no C, no $ adjustment): seq_rank, seqindex.c:91-111
__global__ void k_occ_batch(IndexView v, uint32_t C0, uint32_t C1, uint32_t C2, uint32_t C3,
                            const uint8_t *__restrict__ cs, const uint32_t *__restrict__ idx,
                            uint64_t R, int adjust, uint32_t *__restrict__ out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const uint32_t c = cs[r] & 3u;
  uint32_t i = idx[r];
  if (adjust) i -= (i > v.endloc);  // rank(), seqindex.c:221-225
  if (i > v.n) {
    out[r] = 0xFFFFFFFFu;
    return;
  }
  uint32_t rem;
  const uint32_t b = div96(i, rem);
  const uint4 blk = reinterpret_cast<const uint4 *>(v.sym)[(uint64_t)c * v.nsym + b];
  const uint32_t Cc = c == 0 ? C0 : c == 1 ? C1 : c == 2 ? C2 : C3;
  out[r] = sym_count(blk, rem) - Cc;
}

__global__ void k_lf_batch(IndexView v, const uint32_t *__restrict__ rows, uint64_t R,
                           uint32_t *__restrict__ out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const uint32_t i = rows[r];
  out[r] = i > v.n ? 0xFFFFFFFFu : lf_step(v, i);
}

// reverse_search's return value from the interval and the locate row
// (seqindex.c:228-242): 0 if the interval emptied before the last step, else
// ep - sp + 1 (sic).  Written over sp.
__global__ void k_reverse_result(uint32_t *__restrict__ sp, const uint32_t *__restrict__ ep,
                                 const uint32_t *__restrict__ locrow, uint64_t R) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  sp[r] = locrow[r] == 0xFFFFFFFFu ? 0u : ep[r] - sp[r] + 1u;
}

// rank[sa[r]] = r: the row of every suffix
__global__ void k_rows_of_suffixes(const uint32_t *__restrict__ sa, uint64_t rows, uint32_t *__restrict__ rank) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) rank[sa[r]] = (uint32_t)r;
}

// sort key of every id handed in: its row in the suffix array
__global__ void k_keys_of_ids(const uint32_t *__restrict__ ids, uint64_t count, uint32_t n,
                              const uint32_t *__restrict__ rank, uint64_t *__restrict__ keys) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) keys[i] = ids[i] <= n ? rank[ids[i]] : 0xFFFFFFFFull;
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

// histhelper (histsortcomp.c:35-90) sorts a run of suffix starts that agree on
// their first `depth` bases by what follows.  Whatever the depth, that is the
// order the suffixes have in the suffix array of the whole text, so the device
// builds that array, turns it into the row of every suffix and sorts the ids
// handed in by their rows.  An id equal to n is the empty suffix and comes
// first; ids above n go last.
int bwtgpu_sort_suffixes_host(const uint8_t *h_text, uint32_t n, uint32_t *h_ids, uint64_t count) {
  if (!h_text || !h_ids || n == 0) return set_error(kErrArg, "bad argument");
  if (count <= 1) return kOk;
  if (count >= (1ull << 32)) return set_error(kErrArg, "too many suffixes");
  Dev text, sa, ws, rank, keys, keys_alt, ids, ids_alt, sws;
  const size_t nbytes = (size_t)n / 4 + 1, wsb = bwtgpu_sa_workspace_bytes(n), rows = (size_t)n + 1;
  const size_t swsb = radix_sort_workspace_bytes(count);
  hipError_t e = text.alloc(nbytes + 64);
  if (e == hipSuccess) e = sa.alloc(rows * 4);
  if (e == hipSuccess) e = ws.alloc(wsb);
  if (e == hipSuccess) e = rank.alloc(rows * 4);
  if (e == hipSuccess) e = keys.alloc(count * 8);
  if (e == hipSuccess) e = keys_alt.alloc(count * 8);
  if (e == hipSuccess) e = ids.alloc(count * 4);
  if (e == hipSuccess) e = ids_alt.alloc(count * 4);
  if (e == hipSuccess) e = sws.alloc(swsb);
  if (e != hipSuccess) return set_error(kErrHip, "buffers to sort %llu suffixes of %u bases: %s",
                                        (unsigned long long)count, n, hipGetErrorString(e));
  BWT_HIP(hipMemcpy(text.p, h_text, nbytes, hipMemcpyHostToDevice));
  BWT_HIP(hipMemcpy(ids.p, h_ids, count * 4, hipMemcpyHostToDevice));
  BWT_TRY(bwtgpu_sa_build(text.as<uint8_t>(), n, sa.as<uint32_t>(), ws.p, wsb, nullptr));
  BWT_LAUNCH("rows_of_suffixes", k_rows_of_suffixes, dim3((unsigned)div_up(rows, 256)), dim3(256), 0, nullptr,
             sa.as<uint32_t>(), (uint64_t)rows, rank.as<uint32_t>());
  BWT_LAUNCH("keys_of_ids", k_keys_of_ids, dim3((unsigned)div_up(count, 256)), dim3(256), 0, nullptr,
             ids.as<uint32_t>(), count, n, rank.as<uint32_t>(), keys.as<uint64_t>());
  BWT_CHECK_LAUNCH();
  BWT_TRY(radix_sort_pairs(keys.as<uint64_t>(), ids.as<uint32_t>(), keys_alt.as<uint64_t>(), ids_alt.as<uint32_t>(),
                           count, 32, sws.p, swsb, nullptr));
  BWT_HIP(hipMemcpy(h_ids, ids.p, count * 4, hipMemcpyDeviceToHost));
  return kOk;
}

int bwtgpu_occ_batch(const bwtgpu_index *ix, const uint8_t *d_c, const uint32_t *d_idx, uint64_t R,
                     int skip_dollar_row, uint32_t *d_out, void *stream) {
  if (!ix || !d_c || !d_idx || !d_out) return set_error(kErrArg, "null argument");
  if (R == 0) return kOk;
  BWT_LAUNCH("occ_batch", k_occ_batch, dim3((unsigned)div_up(R, 256)), dim3(256), 0, (hipStream_t)stream,
             view_of(ix), ix->C[0], ix->C[1], ix->C[2], ix->C[3], d_c, d_idx, R, skip_dollar_row, d_out);
  BWT_CHECK_LAUNCH();
  return kOk;
}

int bwtgpu_lf_batch(const bwtgpu_index *ix, const uint32_t *d_rows, uint64_t R, uint32_t *d_out,
                    void *stream) {
  if (!ix || !d_rows || !d_out) return set_error(kErrArg, "null argument");
  if (R == 0) return kOk;
  BWT_LAUNCH("lf_batch", k_lf_batch, dim3((unsigned)div_up(R, 256)), dim3(256), 0, (hipStream_t)stream,
             view_of(ix), d_rows, R, d_out);
  BWT_CHECK_LAUNCH();
  return kOk;
}

int bwtgpu_sa_build_host(const uint8_t *h_text, uint32_t n, uint32_t *h_sa) {
  if (!h_text || !h_sa || n == 0) return set_error(kErrArg, "bad argument");
  Dev text, sa, ws;
  const size_t nbytes = (size_t)n / 4 + 1, wsb = bwtgpu_sa_workspace_bytes(n);
  hipError_t e = text.alloc(nbytes + 64);
  if (e == hipSuccess) e = sa.alloc(((size_t)n + 1) * 4);
  if (e == hipSuccess) e = ws.alloc(wsb);
  if (e != hipSuccess) return set_error(kErrHip, "SA buffers for n=%u: %s", n, hipGetErrorString(e));
  BWT_HIP(hipMemcpy(text.p, h_text, nbytes, hipMemcpyHostToDevice));
  BWT_TRY(bwtgpu_sa_build(text.as<uint8_t>(), n, sa.as<uint32_t>(), ws.p, wsb, nullptr));
  BWT_HIP(hipMemcpy(h_sa, sa.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
  return kOk;
}

int bwtgpu_bwt_from_sa_host(const uint8_t *h_text, const uint32_t *h_sa, uint32_t n, uint8_t *h_bwt,
                            uint32_t *endloc) {
  if (!h_text || !h_sa || !h_bwt || !endloc || n == 0) return set_error(kErrArg, "bad argument");
  Dev text, sa, bwt, e4;
  const size_t nbytes = (size_t)n / 4 + 1, nb = ((size_t)n + 3) / 4;
  hipError_t e = text.alloc(nbytes + 64);
  if (e == hipSuccess) e = sa.alloc(((size_t)n + 1) * 4);
  if (e == hipSuccess) e = bwt.alloc(nb + 64);
  if (e == hipSuccess) e = e4.alloc(256);
  if (e != hipSuccess) return set_error(kErrHip, "BWT buffers for n=%u: %s", n, hipGetErrorString(e));
  BWT_HIP(hipMemcpy(text.p, h_text, nbytes, hipMemcpyHostToDevice));
  BWT_HIP(hipMemcpy(sa.p, h_sa, ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
  BWT_TRY(bwtgpu_bwt_from_sa(text.as<uint8_t>(), sa.as<uint32_t>(), n, bwt.as<uint8_t>(),
                             e4.as<uint32_t>(), nullptr));
  BWT_HIP(hipMemcpy(h_bwt, bwt.p, nb, hipMemcpyDeviceToHost));
  BWT_HIP(hipMemcpy(endloc, e4.p, 4, hipMemcpyDeviceToHost));
  if (*endloc > n) return set_error(kErrArg, "the array holds no row for suffix 0");
  return kOk;
}

// Reads on the host, results to the host; batches are cut to bound device memory.
int bwtgpu_search_host(const bwtgpu_index *ix, const uint8_t *h_pats, uint32_t m, uint64_t stride,
                       uint64_t R, uint32_t *h_sp, uint32_t *h_ep, uint32_t *h_locrow, uint32_t *h_pos) {
  if (!ix || !h_pats || m == 0) return set_error(kErrArg, "bad argument");
  if (R == 0) return kOk;
  if (h_pos && !ix->d_sa && !ix->has_samples)
    return set_error(kErrArg, "this index was loaded without sampled suffix-array rows: it cannot locate");
  BWT_HIP(hipSetDevice(ix->device));
  {
    // a few reads (the reference's scalar calls are batches of one): through the mailbox, no copies
    Mailbox &box = thread_mailbox();
    const uint64_t bytes = (R - 1) * stride + m;
    const size_t wsb = bwtgpu_search_workspace_bytes(R, m), lwsb = bwtgpu_locate_workspace_bytes();
    const bool one_launch = m <= kFewMaxLen;  // longer reads are worth staging through LDS: the search kernel proper
    if (R <= kBoxReads && bytes <= kBoxInput && box.open(ix->device) &&
        (one_launch || box.scratch(wsb + 256 + lwsb))) {
      memcpy(box.host, h_pats, bytes);
      uint32_t *out = reinterpret_cast<uint32_t *>(box.host + kBoxOut);
      uint32_t *d_out = box.on_device<uint32_t>(out);
      if (one_launch) {
        CTab C;
        for (int i = 0; i < 5; ++i) C.c[i] = ix->C[i];
        BWT_LAUNCH("fm_search_few", k_search_locate_few, dim3((unsigned)div_up(R, 256)), dim3(256), 0, box.stream,
                   view_published(ix), C, box.dev, m, stride, R, d_out, d_out + kBoxReads, d_out + 2 * kBoxReads,
                   h_pos ? d_out + 3 * kBoxReads : nullptr);
        BWT_CHECK_LAUNCH();
      } else {
        uint8_t *lws = static_cast<uint8_t *>(box.ws) + align_up(wsb, 256);
        BWT_TRY(bwtgpu_search(ix, box.dev, m, stride, R, d_out, d_out + kBoxReads,
                              (h_pos || h_locrow) ? d_out + 2 * kBoxReads : nullptr, box.ws, wsb, box.stream));
        if (h_pos)
          BWT_TRY(bwtgpu_locate_rows(ix, d_out + 2 * kBoxReads, R, d_out + 3 * kBoxReads, nullptr, lws, lwsb,
                                     box.stream));
      }
      BWT_HIP(hipStreamSynchronize(box.stream));
      if (h_sp) memcpy(h_sp, out, R * 4);
      if (h_ep) memcpy(h_ep, out + kBoxReads, R * 4);
      if (h_locrow) memcpy(h_locrow, out + 2 * kBoxReads, R * 4);
      if (h_pos) memcpy(h_pos, out + 3 * kBoxReads, R * 4);
      return kOk;
    }
  }
  // large batches of reads that do not overlap: packed by host threads, chunk by chunk, while the
  // chunks before are on the wire and in the kernels
  if (host_batch_prefers_packing(m, stride, R))
    return search_host_streamed(ix, h_pats, m, stride, R, h_sp, h_ep, h_locrow, h_pos);
  const uint64_t chunk_max = stride ? (1ull << 30) / (stride > m ? stride : m) + 1 : R;
  const uint64_t chunk = R < chunk_max ? R : chunk_max;
  Dev pats, sp, ep, lr, pos, ws, lws;
  const size_t wsb = bwtgpu_search_workspace_bytes(chunk, m);
  hipError_t e = pats.alloc((chunk - 1) * stride + m + 64);
  if (e == hipSuccess) e = sp.alloc(chunk * 4);
  if (e == hipSuccess) e = ep.alloc(chunk * 4);
  if (e == hipSuccess) e = lr.alloc(chunk * 4);
  if (e == hipSuccess) e = pos.alloc(chunk * 4);
  if (e == hipSuccess) e = ws.alloc(wsb);
  if (e == hipSuccess) e = lws.alloc(bwtgpu_locate_workspace_bytes());
  if (e != hipSuccess) return set_error(kErrHip, "search buffers: %s", hipGetErrorString(e));
  for (uint64_t r0 = 0; r0 < R; r0 += chunk) {
    const uint64_t cnt = R - r0 < chunk ? R - r0 : chunk;
    BWT_HIP(hipMemcpy(pats.p, h_pats + r0 * stride, (cnt - 1) * stride + m, hipMemcpyHostToDevice));
    BWT_TRY(bwtgpu_search(ix, pats.as<uint8_t>(), m, stride, cnt, sp.as<uint32_t>(), ep.as<uint32_t>(),
                          (h_pos || h_locrow) ? lr.as<uint32_t>() : nullptr, ws.p, wsb, nullptr));
    if (h_pos)
      BWT_TRY(bwtgpu_locate_rows(ix, lr.as<uint32_t>(), cnt, pos.as<uint32_t>(), nullptr, lws.p,
                                 bwtgpu_locate_workspace_bytes(), nullptr));
    if (h_sp) BWT_HIP(hipMemcpy(h_sp + r0, sp.p, cnt * 4, hipMemcpyDeviceToHost));
    if (h_ep) BWT_HIP(hipMemcpy(h_ep + r0, ep.p, cnt * 4, hipMemcpyDeviceToHost));
    if (h_locrow) BWT_HIP(hipMemcpy(h_locrow + r0, lr.p, cnt * 4, hipMemcpyDeviceToHost));
    if (h_pos) BWT_HIP(hipMemcpy(h_pos + r0, pos.p, cnt * 4, hipMemcpyDeviceToHost));
  }
  return kOk;
}

// reverse_search for a batch: one int per read comes back instead of three
int bwtgpu_reverse_search_host(const bwtgpu_index *ix, const uint8_t *h_pats, uint32_t m, uint64_t stride,
                               uint64_t R, int32_t *h_result) {
  if (!ix || !h_pats || !h_result || m == 0) return set_error(kErrArg, "bad argument");
  if (R == 0) return kOk;
  BWT_HIP(hipSetDevice(ix->device));
  {
    Mailbox &box = thread_mailbox();
    const uint64_t bytes = (R - 1) * stride + m;
    const size_t wsb = bwtgpu_search_workspace_bytes(R, m);
    const bool one_launch = m <= kFewMaxLen;
    if (R <= kBoxReads && bytes <= kBoxInput && box.open(ix->device) && (one_launch || box.scratch(wsb))) {
      memcpy(box.host, h_pats, bytes);
      uint32_t *out = reinterpret_cast<uint32_t *>(box.host + kBoxOut);
      uint32_t *d_out = box.on_device<uint32_t>(out);
      if (one_launch) {
        CTab C;
        for (int i = 0; i < 5; ++i) C.c[i] = ix->C[i];
        BWT_LAUNCH("fm_search_few", k_search_locate_few, dim3((unsigned)div_up(R, 256)), dim3(256), 0, box.stream,
                   view_published(ix), C, box.dev, m, stride, R, d_out, d_out + kBoxReads, d_out + 2 * kBoxReads,
                   (uint32_t *)nullptr);
        BWT_CHECK_LAUNCH();
        BWT_HIP(hipStreamSynchronize(box.stream));
        for (uint64_t r = 0; r < R; ++r)  // seqindex.c:228-242: 0 if the interval emptied early, else ep - sp + 1 (sic)
          h_result[r] = out[2 * kBoxReads + r] == 0xFFFFFFFFu ? 0 : (int32_t)(out[kBoxReads + r] - out[r] + 1u);
        return kOk;
      }
      BWT_TRY(bwtgpu_search(ix, box.dev, m, stride, R, d_out, d_out + kBoxReads, d_out + 2 * kBoxReads, box.ws,
                            wsb, box.stream));
      BWT_LAUNCH("reverse_result", k_reverse_result, dim3(1), dim3(256), 0, box.stream, d_out, d_out + kBoxReads,
                 d_out + 2 * kBoxReads, R);
      BWT_CHECK_LAUNCH();
      BWT_HIP(hipStreamSynchronize(box.stream));
      memcpy(h_result, out, R * 4);
      return kOk;
    }
  }
  const uint64_t chunk_max = stride ? (1ull << 30) / (stride > m ? stride : m) + 1 : R;
  const uint64_t chunk = R < chunk_max ? R : chunk_max;
  Dev pats, sp, ep, lr, ws;
  const size_t wsb = bwtgpu_search_workspace_bytes(chunk, m);
  hipError_t e = pats.alloc((chunk - 1) * stride + m + 64);
  if (e == hipSuccess) e = sp.alloc(chunk * 4);
  if (e == hipSuccess) e = ep.alloc(chunk * 4);
  if (e == hipSuccess) e = lr.alloc(chunk * 4);
  if (e == hipSuccess) e = ws.alloc(wsb);
  if (e != hipSuccess) return set_error(kErrHip, "search buffers: %s", hipGetErrorString(e));
  for (uint64_t r0 = 0; r0 < R; r0 += chunk) {
    const uint64_t cnt = R - r0 < chunk ? R - r0 : chunk;
    BWT_HIP(hipMemcpy(pats.p, h_pats + r0 * stride, (cnt - 1) * stride + m, hipMemcpyHostToDevice));
    BWT_TRY(bwtgpu_search(ix, pats.as<uint8_t>(), m, stride, cnt, sp.as<uint32_t>(), ep.as<uint32_t>(),
                          lr.as<uint32_t>(), ws.p, wsb, nullptr));
    BWT_LAUNCH("reverse_result", k_reverse_result, dim3((unsigned)div_up(cnt, 256)), dim3(256), 0, nullptr,
               sp.as<uint32_t>(), ep.as<uint32_t>(), lr.as<uint32_t>(), cnt);
    BWT_CHECK_LAUNCH();
    BWT_HIP(hipMemcpy(h_result + r0, sp.p, cnt * 4, hipMemcpyDeviceToHost));
  }
  return kOk;
}

int bwtgpu_locate_rows_host(const bwtgpu_index *ix, const uint32_t *h_rows, uint64_t R,
                            uint32_t *h_pos) {
  if (!ix || !h_rows || !h_pos) return set_error(kErrArg, "bad argument");
  if (R == 0) return kOk;
  BWT_HIP(hipSetDevice(ix->device));
  {
    Mailbox &box = thread_mailbox();
    const size_t lwsb = bwtgpu_locate_workspace_bytes();
    if (R <= kBoxReads && box.open(ix->device) && box.scratch(lwsb)) {
      uint32_t *io = reinterpret_cast<uint32_t *>(box.host);
      memcpy(io, h_rows, R * 4);
      BWT_TRY(bwtgpu_locate_rows(ix, box.on_device<uint32_t>(io), R, box.on_device<uint32_t>(io + kBoxReads),
                                 nullptr, box.ws, lwsb, box.stream));
      BWT_HIP(hipStreamSynchronize(box.stream));
      memcpy(h_pos, io + kBoxReads, R * 4);
      return kOk;
    }
  }
  Dev rows, pos, lws;
  hipError_t e = rows.alloc(R * 4);
  if (e == hipSuccess) e = pos.alloc(R * 4);
  if (e == hipSuccess) e = lws.alloc(bwtgpu_locate_workspace_bytes());
  if (e != hipSuccess) return set_error(kErrHip, "locate buffers: %s", hipGetErrorString(e));
  BWT_HIP(hipMemcpy(rows.p, h_rows, R * 4, hipMemcpyHostToDevice));
  BWT_TRY(bwtgpu_locate_rows(ix, rows.as<uint32_t>(), R, pos.as<uint32_t>(), nullptr, lws.p,
                             bwtgpu_locate_workspace_bytes(), nullptr));
  BWT_HIP(hipMemcpy(h_pos, pos.p, R * 4, hipMemcpyDeviceToHost));
  return kOk;
}

int bwtgpu_occ_host(const bwtgpu_index *ix, const uint8_t *h_c, const uint32_t *h_idx, uint64_t R,
                    int skip_dollar_row, uint32_t *h_out) {
  if (!ix || !h_c || !h_idx || !h_out) return set_error(kErrArg, "bad argument");
  if (R == 0) return kOk;
  BWT_HIP(hipSetDevice(ix->device));
  {
    Mailbox &box = thread_mailbox();
    if (R <= kBoxReads && box.open(ix->device)) {
      uint32_t *io = reinterpret_cast<uint32_t *>(box.host);
      uint8_t *cs = box.host + 2 * kBoxReads * 4;
      memcpy(io, h_idx, R * 4);
      memcpy(cs, h_c, R);
      BWT_TRY(bwtgpu_occ_batch(ix, box.on_device<uint8_t>(cs), box.on_device<uint32_t>(io), R, skip_dollar_row,
                               box.on_device<uint32_t>(io + kBoxReads), box.stream));
      BWT_HIP(hipStreamSynchronize(box.stream));
      memcpy(h_out, io + kBoxReads, R * 4);
      return kOk;
    }
  }
  Dev c, idx, out;
  hipError_t e = c.alloc(R);
  if (e == hipSuccess) e = idx.alloc(R * 4);
  if (e == hipSuccess) e = out.alloc(R * 4);
  if (e != hipSuccess) return set_error(kErrHip, "rank buffers: %s", hipGetErrorString(e));
  BWT_HIP(hipMemcpy(c.p, h_c, R, hipMemcpyHostToDevice));
  BWT_HIP(hipMemcpy(idx.p, h_idx, R * 4, hipMemcpyHostToDevice));
  BWT_TRY(bwtgpu_occ_batch(ix, c.as<uint8_t>(), idx.as<uint32_t>(), R, skip_dollar_row,
                           out.as<uint32_t>(), nullptr));
  BWT_HIP(hipMemcpy(h_out, out.p, R * 4, hipMemcpyDeviceToHost));
  return kOk;
}

int bwtgpu_lf_host(const bwtgpu_index *ix, const uint32_t *h_rows, uint64_t R, uint32_t *h_out) {
  if (!ix || !h_rows || !h_out) return set_error(kErrArg, "bad argument");
  if (R == 0) return kOk;
  BWT_HIP(hipSetDevice(ix->device));
  {
    Mailbox &box = thread_mailbox();
    if (R <= kBoxReads && box.open(ix->device)) {
      uint32_t *io = reinterpret_cast<uint32_t *>(box.host);
      memcpy(io, h_rows, R * 4);
      BWT_TRY(bwtgpu_lf_batch(ix, box.on_device<uint32_t>(io), R, box.on_device<uint32_t>(io + kBoxReads), box.stream));
      BWT_HIP(hipStreamSynchronize(box.stream));
      memcpy(h_out, io + kBoxReads, R * 4);
      return kOk;
    }
  }
  Dev rows, out;
  hipError_t e = rows.alloc(R * 4);
  if (e == hipSuccess) e = out.alloc(R * 4);
  if (e != hipSuccess) return set_error(kErrHip, "lf buffers: %s", hipGetErrorString(e));
  BWT_HIP(hipMemcpy(rows.p, h_rows, R * 4, hipMemcpyHostToDevice));
  BWT_TRY(bwtgpu_lf_batch(ix, rows.as<uint32_t>(), R, out.as<uint32_t>(), nullptr));
  BWT_HIP(hipMemcpy(h_out, out.p, R * 4, hipMemcpyDeviceToHost));
  return kOk;
}

}  // extern "C"
