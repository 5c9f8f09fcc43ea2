// Library plumbing: device selection, error strings, per-launch timing.
#include <cstring>

#include <mutex>
#include <unordered_map>

#include "common.hpp"

namespace bwtgpu {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

// Device buffers of indexes come and go in a few sizes (every index of one
// text length uses the same ones), and hipMalloc of hundreds of megabytes costs
// more than the kernels that fill them.  Freed buffers wait here, keyed by
// device and exact size, for the next request of that size.
namespace {
struct Pooled {
  void *p;
  size_t bytes;
  int device;
};
std::mutex g_pool_mu;
std::unordered_map<void *, Pooled> g_live;  // handed out by pool_alloc
std::vector<Pooled> g_idle;
constexpr size_t kPoolMaxIdle = 48;
// Idle buffers may hold at most a quarter of the device's memory: an index of
// the largest text (39 GB at 3.1 Gbp) still finds its buffers again, and what a
// process no longer uses does not stay out of reach of everything else in it.
size_t g_idle_bytes = 0, g_idle_cap = 0;

size_t idle_cap_locked() {
  if (!g_idle_cap) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || !total_b) {
      (void)hipGetLastError();
      total_b = (size_t)64 << 30;
    }
    g_idle_cap = total_b / 4;
  }
  return g_idle_cap;
}

// with the pool lock held: takes the longest-idle buffers off the list until count and bytes fit
void trim_locked(std::vector<void *> &drop) {
  const size_t cap = idle_cap_locked();
  while (!g_idle.empty() && (g_idle.size() > kPoolMaxIdle || g_idle_bytes > cap)) {
    drop.push_back(g_idle.front().p);
    g_idle_bytes -= g_idle.front().bytes;
    g_idle.erase(g_idle.begin());
  }
}
}  // namespace

hipError_t pool_alloc(void **out, size_t bytes) {
  *out = nullptr;
  if (bytes == 0) bytes = 256;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    for (size_t i = 0; i < g_idle.size(); ++i)
      if (g_idle[i].bytes == bytes && g_idle[i].device == dev) {
        const Pooled hit = g_idle[i];
        g_idle.erase(g_idle.begin() + (long)i);
        g_idle_bytes -= hit.bytes;
        g_live[hit.p] = hit;
        *out = hit.p;
        return hipSuccess;
      }
  }
  void *p = nullptr;
  e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    // idle buffers may be what is in the way: give them back and try once more
    (void)hipGetLastError();
    pool_release_idle();
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lock(g_pool_mu);
  g_live[p] = Pooled{p, bytes, dev};
  *out = p;
  return hipSuccess;
}

// The caller has made sure that no work in flight still uses the buffer.
void pool_free(void *p) {
  if (!p) return;
  std::vector<void *> drop;  // what goes back to the driver
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    auto it = g_live.find(p);
    if (it != g_live.end()) {
      g_idle.push_back(it->second);
      g_idle_bytes += it->second.bytes;
      g_live.erase(it);
      trim_locked(drop);  // the longest-idle buffers make room
    } else {
      drop.push_back(p);
    }
  }
  for (void *d : drop) (void)hipFree(d);
}

// A buffer that was hipMalloc'd outside the pool joins the idle list (a thread
// that ends hands over its build workspace this way).  Makes no HIP call: it may
// run while the process is shutting down.
void pool_adopt_idle(void *p, size_t bytes, int device) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(g_pool_mu);
  g_idle.push_back(Pooled{p, bytes, device});
  g_idle_bytes += bytes;
}

void pool_release_idle() {
  std::vector<Pooled> drop;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    drop.swap(g_idle);
    g_idle_bytes = 0;
  }
  for (const Pooled &d : drop) (void)hipFree(d.p);
}

TimingState &timing() {
  static thread_local TimingState t;
  return t;
}

ScopedKernelTimer::ScopedKernelTimer(const char *name, hipStream_t s)
    : name_(name), s_(s) {
  on_ = timing().enabled;
  if (!on_) return;
  if (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess) {
    on_ = false;
    return;
  }
  (void)hipEventRecord(a_, s_);
}

ScopedKernelTimer::~ScopedKernelTimer() {
  if (!on_) return;
  (void)hipEventRecord(b_, s_);
  timing().pending.push_back({name_, a_, b_});
}

static void timing_resolve() {
  TimingState &t = timing();
  for (auto &p : t.pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess &&
        hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      auto &e = t.totals[p.name];
      e.first += 1;
      e.second += ms;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  t.pending.clear();
}

bool Mailbox::open(int device_ordinal) {
  if (host && device == device_ordinal) return true;
  if (host) {
    (void)hipHostFree(host);
    if (ws) (void)hipFree(ws);
    host = dev = nullptr;
    ws = nullptr;
    ws_bytes = 0;
  }
  void *h = nullptr, *d = nullptr;
  if (hipHostMalloc(&h, kBytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    (void)hipGetLastError();
    if (h) (void)hipHostFree(h);
    return false;
  }
  if (stream) (void)hipStreamDestroy(stream);
  stream = nullptr;
  if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostFree(h);
    return false;
  }
  host = static_cast<uint8_t *>(h);
  dev = static_cast<uint8_t *>(d);
  device = device_ordinal;
  return true;
}

bool Mailbox::scratch(size_t bytes) {
  if (bytes <= ws_bytes) return true;
  if (ws) (void)hipFree(ws);
  ws = nullptr;
  ws_bytes = 0;
  if (hipMalloc(&ws, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  ws_bytes = bytes;
  return true;
}

Mailbox::~Mailbox() {
  // Runs at thread exit, for the main thread after the HIP runtime may have shut
  // down: nothing is freed here (64 KiB of pinned memory and the scratch stay
  // with the process).
}

Mailbox &thread_mailbox() {
  static thread_local Mailbox box;
  return box;
}

}  // namespace bwtgpu

using namespace bwtgpu;

extern "C" {

const char *bwtgpu_last_error(void) { return g_err; }

int bwtgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int bwtgpu_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(kErrDevice, "no HIP device visible (%s)",
                     e == hipSuccess ? "count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= n)
    return set_error(kErrArg, "device %d out of range (have %d)", device, n);
  BWT_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  BWT_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(kErrDevice, "device %d is %s; this library is built for gfx950 only",
                     device, prop.gcnArchName);
  // the one thing bwtgpu_search_static would otherwise set up on its first call
  return static_scratch_setup(device);
}

int bwtgpu_timing_enable(int on) {
  timing().enabled = on != 0;
  return kOk;
}

int bwtgpu_timing_reset(void) {
  timing_resolve();
  timing().totals.clear();
  return kOk;
}

int bwtgpu_timing_get(const char *kernel, uint64_t *launches, double *total_ms) {
  timing_resolve();
  auto it = timing().totals.find(kernel);
  uint64_t l = 0;
  double ms = 0;
  if (it != timing().totals.end()) {
    l = it->second.first;
    ms = it->second.second;
  }
  if (launches) *launches = l;
  if (total_ms) *total_ms = ms;
  return kOk;
}

int bwtgpu_timing_report(char *buf, size_t cap) {
  timing_resolve();
  size_t off = 0;
  for (auto &kv : timing().totals) {
    int w = snprintf(buf + off, off < cap ? cap - off : 0, "%s %llu %.6f\n",
                     kv.first.c_str(), (unsigned long long)kv.second.first,
                     kv.second.second);
    if (w < 0 || off + (size_t)w >= cap) break;
    off += (size_t)w;
  }
  return (int)off;
}

}  // extern "C"
