// Host buffers in, host buffers out, at a rate the kernels deserve.
//
// The reference's callers hold their reads in ordinary host memory, one base
// per byte (fmitest.c:150-153, search_reads.c:102-111).  Copying such a batch to
// the device in one blocking call and the results back in another costs 30 times
// what the search takes.  Here a batch is cut into chunks that move through
// three stages at once: host threads pack chunk c+1 to 2 bits per base into
// pinned memory (a quarter of the bytes cross PCIe) and copy the results of
// chunk c-1 out of pinned memory, while the DMA engines and the kernels work on
// chunk c.  Callers that keep their reads packed (bwtgpu_pack_reads_host's
// format) in pinned memory skip the packing.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "common.hpp"

namespace bwtgpu {

// ---- a few host threads for the copying and packing ---------------------------------

class HostPool {
 public:
  static HostPool &get() {
    static HostPool *p = new HostPool();  // never destroyed: its threads may outlive static destruction
    return *p;
  }
  int threads() const { return nthreads_; }
  // fn(begin, end) over [0, n) in pieces of at least `grain`; the caller takes part; returns when all is done
  void run(uint64_t n, uint64_t grain, const std::function<void(uint64_t, uint64_t)> &fn) {
    if (n == 0) return;
    const uint64_t pieces = std::min<uint64_t>((uint64_t)nthreads_ * 4, std::max<uint64_t>(1, n / std::max<uint64_t>(grain, 1)));
    if (pieces <= 1 || nthreads_ <= 1) {
      fn(0, n);
      return;
    }
    std::unique_lock<std::mutex> call(call_mu_);  // one batch at a time
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      n_ = n;
      pieces_ = pieces;
      next_.store(0);
      left_ = (int)workers_.size();
      ++generation_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return left_ == 0; });
    fn_ = nullptr;
  }

 private:
  HostPool() {
    int t = 0;
    if (const char *e = getenv("BWT_MI355X_HOST_THREADS")) t = atoi(e);
    if (t <= 0) {
      const unsigned hw = std::thread::hardware_concurrency();
      t = (int)std::min<unsigned>(8u, hw ? hw : 1u);
    }
    nthreads_ = t > 64 ? 64 : t;
    for (int i = 1; i < nthreads_; ++i) workers_.emplace_back([this] { loop(); });
    for (auto &w : workers_) w.detach();
  }
  void work() {
    for (;;) {
      const uint64_t p = next_.fetch_add(1);
      if (p >= pieces_) return;
      const uint64_t b = n_ * p / pieces_, e = n_ * (p + 1) / pieces_;
      if (e > b) (*fn_)(b, e);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--left_ == 0) done_cv_.notify_all();
      }
    }
  }
  int nthreads_ = 1;
  std::vector<std::thread> workers_;
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(uint64_t, uint64_t)> *fn_ = nullptr;
  uint64_t n_ = 0, pieces_ = 0, generation_ = 0;
  std::atomic<uint64_t> next_{0};
  int left_ = 0;
};

static void parallel_copy(void *dst, const void *src, size_t bytes) {
  HostPool::get().run(bytes, 1u << 20, [&](uint64_t b, uint64_t e) {
    memcpy(static_cast<uint8_t *>(dst) + b, static_cast<const uint8_t *>(src) + b, e - b);
  });
}

// ---- packing -----------------------------------------------------------------------------

// bwt_amd/host/pack_reads.c (plain C, SSSE3 where the CPU has it): reads [b, e) of a
// batch -> their packed words; `limit` is the end of the whole buffer of reads
extern "C" void bwt_pack_read_range(const uint8_t *pats, uint32_t m, uint64_t stride, uint64_t b, uint64_t e,
                                    const uint8_t *limit, uint32_t *out);

static void pack_range(const uint8_t *pats, uint32_t m, uint64_t stride, uint64_t b, uint64_t e,
                       const uint8_t *limit, uint32_t *out) {
  bwt_pack_read_range(pats, m, stride, b, e, limit, out);
}

// ---- pinned staging buffers, kept between calls -----------------------------------------

namespace {
struct Pinned {
  void *p;
  size_t bytes;
};
std::mutex g_pin_mu;
std::vector<Pinned> g_pin_idle;
}  // namespace

// Staging sizes follow the batch (chunk = min(R, ...)): rounded up to a power of two so
// that callers whose batches vary in size find the buffers of the call before.
static size_t pinned_round(size_t bytes) {
  size_t r = 64u << 10;
  while (r < bytes) r <<= 1;
  return r;
}

static void *pinned_get(size_t bytes) {
  bytes = pinned_round(bytes);
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (size_t i = 0; i < g_pin_idle.size(); ++i)
      if (g_pin_idle[i].bytes == bytes) {
        void *p = g_pin_idle[i].p;
        g_pin_idle.erase(g_pin_idle.begin() + (long)i);
        return p;
      }
  }
  void *p = nullptr;
  // portable: the staging of one call may be reused by a call on another GPU
  if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

static void pinned_put(void *p, size_t bytes) {
  if (!p) return;
  bytes = pinned_round(bytes);
  void *drop = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_idle.push_back(Pinned{p, bytes});
    if (g_pin_idle.size() > 16) {
      drop = g_pin_idle.front().p;
      g_pin_idle.erase(g_pin_idle.begin());
    }
  }
  if (drop) (void)hipHostFree(drop);
}

void pinned_release_idle() {
  std::vector<Pinned> drop;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    drop.swap(g_pin_idle);
  }
  for (const Pinned &d : drop) (void)hipHostFree(d.p);
}

// ---- streams and events of the pipeline's slots, kept between calls --------------------

namespace {
struct Lane {
  hipStream_t stream;
  hipEvent_t done;
  int device;
};
std::mutex g_lane_mu;
std::vector<Lane> g_lane_idle;
}  // namespace

static hipError_t lane_get(hipStream_t *stream, hipEvent_t *done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    for (size_t i = 0; i < g_lane_idle.size(); ++i)
      if (g_lane_idle[i].device == dev) {
        *stream = g_lane_idle[i].stream;
        *done = g_lane_idle[i].done;
        g_lane_idle.erase(g_lane_idle.begin() + (long)i);
        return hipSuccess;
      }
  }
  e = hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(done, hipEventDisableTiming);
  return e;
}

static void lane_put(hipStream_t stream, hipEvent_t done, int device) {
  if (!stream || !done) {
    if (done) (void)hipEventDestroy(done);
    if (stream) (void)hipStreamDestroy(stream);
    return;
  }
  std::lock_guard<std::mutex> lk(g_lane_mu);
  g_lane_idle.push_back(Lane{stream, done, device});
}

// ---- the pipeline ------------------------------------------------------------------------------

constexpr int kSlots = 3;                       // chunks in flight
constexpr uint64_t kChunkWordBytes = 8u << 20;  // packed reads per chunk, about

struct Slot {
  uint32_t *h_in = nullptr, *h_out = nullptr;  // pinned
  uint32_t *d_in = nullptr, *d_out = nullptr;  // device: packed reads; sp, ep, locrow, pos
  void *d_lws = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  uint64_t first = 0, count = 0;  // reads of the chunk in flight
  bool busy = false;
};

struct PipeBuffers {
  Slot slot[kSlots];
  size_t in_bytes = 0, out_bytes = 0;
  int device = 0;
  ~PipeBuffers() {
    for (Slot &s : slot) {
      if (s.stream) (void)hipStreamSynchronize(s.stream);
      pinned_put(s.h_in, in_bytes);
      pinned_put(s.h_out, out_bytes);
      pool_free(s.d_in);
      pool_free(s.d_out);
      pool_free(s.d_lws);
      lane_put(s.stream, s.done, device);
    }
  }
  int open(uint64_t chunk, uint32_t words) {
    // device buffers come from the exact-size pool: rounded like the pinned ones, so that
    // batches of varying size find them again
    in_bytes = pinned_round(chunk * words * 4);
    out_bytes = pinned_round(chunk * 16);
    (void)hipGetDevice(&device);
    for (Slot &s : slot) {
      s.h_in = static_cast<uint32_t *>(pinned_get(in_bytes));
      s.h_out = static_cast<uint32_t *>(pinned_get(out_bytes));
      if (!s.h_in || !s.h_out) return set_error(kErrHip, "no pinned host memory for staging (%zu + %zu bytes)", in_bytes, out_bytes);
      hipError_t e = pool_alloc(&s.d_in, in_bytes);
      if (e == hipSuccess) e = pool_alloc(&s.d_out, out_bytes);
      if (e == hipSuccess) e = pool_alloc(&s.d_lws, bwtgpu_locate_workspace_bytes());
      if (e == hipSuccess) e = lane_get(&s.stream, &s.done);
      if (e != hipSuccess) return set_error(kErrHip, "staging buffers: %s", hipGetErrorString(e));
    }
    return kOk;
  }
};

// Search (and locate) of R reads whose packed words either come from h_words (read
// r at h_words + r * words) or are packed here from h_pats (one base per byte, read
// r at h_pats + r * stride).  Results to whichever of the four host arrays are given.
static int search_stream(const bwtgpu_index *ix, const uint8_t *h_pats, uint64_t stride, const uint32_t *h_words,
                         uint32_t m, uint64_t R, uint32_t *h_sp, uint32_t *h_ep, uint32_t *h_locrow,
                         uint32_t *h_pos) {
  const uint32_t words = (m + 15) / 16;
  uint64_t chunk = kChunkWordBytes / ((uint64_t)words * 4);
  if (chunk < 4096) chunk = 4096;
  if (chunk > R) chunk = R;
  BWT_TRY(index_ready_for_search(ix, nullptr));  // before any chunk is timed against another
  PipeBuffers pipe;
  BWT_TRY(pipe.open(chunk, words));
  const bool want_row = h_locrow || h_pos;

  auto finish = [&](Slot &s) -> int {  // results of the slot's chunk to the caller's arrays
    if (!s.busy) return kOk;
    BWT_HIP(hipEventSynchronize(s.done));
    const uint64_t c = s.count;
    uint32_t *o = s.h_out;
    if (h_sp) parallel_copy(h_sp + s.first, o, c * 4);
    if (h_ep) parallel_copy(h_ep + s.first, o + chunk, c * 4);
    if (h_locrow) parallel_copy(h_locrow + s.first, o + 2 * chunk, c * 4);
    if (h_pos) parallel_copy(h_pos + s.first, o + 3 * chunk, c * 4);
    s.busy = false;
    return kOk;
  };

  uint64_t k = 0;
  for (uint64_t r0 = 0; r0 < R; r0 += chunk, ++k) {
    Slot &s = pipe.slot[k % kSlots];
    BWT_TRY(finish(s));  // the chunk that used this slot three chunks ago
    const uint64_t c = R - r0 < chunk ? R - r0 : chunk;
    if (h_words) {
      parallel_copy(s.h_in, h_words + r0 * words, c * words * 4);
    } else {
      uint32_t *dst = s.h_in;
      const uint8_t *first = h_pats + r0 * stride, *limit = h_pats + (R - 1) * stride + m;
      HostPool::get().run(c, 4096, [&](uint64_t b, uint64_t e) { pack_range(first, m, stride, b, e, limit, dst); });
    }
    BWT_HIP(hipMemcpyAsync(s.d_in, s.h_in, c * words * 4, hipMemcpyHostToDevice, s.stream));
    uint32_t *d = s.d_out;
    if (h_pos && ix->d_isa) {  // the search kernel writes the positions itself
      BWT_TRY(search_packed_dev(ix, s.d_in, m, c, d, d + chunk, h_locrow ? d + 2 * chunk : nullptr, d + 3 * chunk,
                                s.stream));
    } else {
      BWT_TRY(bwtgpu_search_packed(ix, s.d_in, m, c, d, d + chunk, want_row ? d + 2 * chunk : nullptr, s.stream));
      if (h_pos)
        BWT_TRY(bwtgpu_locate_rows(ix, d + 2 * chunk, c, d + 3 * chunk, nullptr, s.d_lws,
                                   bwtgpu_locate_workspace_bytes(), s.stream));
    }
    if (h_sp) BWT_HIP(hipMemcpyAsync(s.h_out, d, c * 4, hipMemcpyDeviceToHost, s.stream));
    if (h_ep) BWT_HIP(hipMemcpyAsync(s.h_out + chunk, d + chunk, c * 4, hipMemcpyDeviceToHost, s.stream));
    if (h_locrow) BWT_HIP(hipMemcpyAsync(s.h_out + 2 * chunk, d + 2 * chunk, c * 4, hipMemcpyDeviceToHost, s.stream));
    if (h_pos) BWT_HIP(hipMemcpyAsync(s.h_out + 3 * chunk, d + 3 * chunk, c * 4, hipMemcpyDeviceToHost, s.stream));
    BWT_HIP(hipEventRecord(s.done, s.stream));
    s.first = r0;
    s.count = c;
    s.busy = true;
  }
  for (uint64_t j = 0; j < (uint64_t)kSlots; ++j) BWT_TRY(finish(pipe.slot[(k + j) % kSlots]));  // oldest first
  return kOk;
}

// does a batch of byte reads gain from being packed on the host?  Overlapping
// reads (fmitest's stride 1) are fewer bytes as they are.
bool host_batch_prefers_packing(uint32_t m, uint64_t stride, uint64_t R) {
  const uint64_t as_bytes = (R - 1) * stride + m, as_words = (uint64_t)((m + 15) / 16) * 4 * R;
  return R >= 32768 && as_words <= as_bytes;
}

int search_host_streamed(const bwtgpu_index *ix, const uint8_t *h_pats, uint32_t m, uint64_t stride, uint64_t R,
                         uint32_t *h_sp, uint32_t *h_ep, uint32_t *h_locrow, uint32_t *h_pos) {
  return search_stream(ix, h_pats, stride, nullptr, m, R, h_sp, h_ep, h_locrow, h_pos);
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

int bwtgpu_pack_reads_host(const uint8_t *h_pats, uint32_t m, uint64_t stride, uint64_t R, uint32_t *h_words) {
  if (!h_pats || !h_words || m == 0) return set_error(kErrArg, "bad argument");
  if (R == 0) return kOk;
  const uint8_t *limit = h_pats + (R - 1) * stride + m;
  HostPool::get().run(R, 4096, [&](uint64_t b, uint64_t e) { pack_range(h_pats, m, stride, b, e, limit, h_words); });
  return kOk;
}

int bwtgpu_search_packed_host(const bwtgpu_index *ix, const uint32_t *h_words, uint32_t m, uint64_t R,
                              uint32_t *h_sp, uint32_t *h_ep, uint32_t *h_locrow, uint32_t *h_pos) {
  if (!ix || !h_words || m == 0 || m > 65535) return set_error(kErrArg, "bad argument");
  if (R == 0) return kOk;
  if (h_pos && !ix->d_sa && !ix->has_samples)
    return set_error(kErrArg, "this index was loaded without sampled suffix-array rows: it cannot locate");
  BWT_HIP(hipSetDevice(ix->device));
  return search_stream(ix, nullptr, 0, h_words, m, R, h_sp, h_ep, h_locrow, h_pos);
}

}  // extern "C"
