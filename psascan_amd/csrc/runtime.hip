// runtime.hip -- device selection, memory helpers, error state, small bit utilities.
#include "dev_common.hpp"

#include <dirent.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace psg {
static thread_local std::string g_err;
static hipStream_t g_stream = nullptr;
static bool g_own_stream = false;
static bool g_inited = false;
static int g_device = 0;
static double g_last_ms = 0;

void set_error(const std::string &s) { g_err = s; }
static hipStream_t g_override = nullptr, g_side = nullptr;
hipStream_t stream() { return g_override ? g_override : g_stream; }
hipStream_t side_stream() {
  if (!g_side && hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); g_side = nullptr; }
  return g_side;
}
StreamScope::StreamScope(hipStream_t s) : prev(g_override) { g_override = s; }
StreamScope::~StreamScope() { g_override = prev; }
void note_kernel_ms(double ms) { g_last_ms = ms; }

// ---- device memory: arena for large blocks, size-class cache for small ones -----------------
// A hipMalloc of several GiB costs ~25-30 ms per GiB (0.9 s for a 34 GiB rank structure), and the multi-GiB
// temporaries of a run (rank logs, partition buffers, gap arrays, sorter keys) come in sizes that change
// from pass to pass.  A cache of whole blocks either misses (exact sizes) or lets one purpose take another's
// block (tolerant sizes); both showed up as 100-500 ms stalls in multi-block runs.  So: blocks of >= 1 MiB
// are carved out of a few large segments (best fit, split, coalesce on free; 2 MiB granules) -- after the
// first passes no request reaches the driver.  Reuse is immediate: everything the library enqueues goes to one
// stream (or is waited for before its buffers are released), so a block freed by the host is not in use.
static std::mutex g_pool_mu;
static std::multimap<size_t, void *> g_pool_free;      // small blocks: size -> block
static std::unordered_map<void *, size_t> g_pool_live;  // small blocks: block -> size

struct ArenaSeg {
  char *base = nullptr;
  size_t size = 0, used = 0;
  std::map<size_t, size_t> free;   // offset -> length, disjoint, never adjacent
};
static std::vector<ArenaSeg> g_segs;
static std::unordered_map<void *, std::pair<int, size_t>> g_big_live;   // block -> (segment, length)
static const size_t BIG = (size_t)1 << 20, GRAN = (size_t)2 << 20, SEG_MAX = (size_t)8 << 30;

static size_t pool_round(size_t b) { return (b + 4095) / 4096 * 4096; }
static size_t g_in_use = 0, g_peak = 0, g_small_reserved = 0;   // guarded by g_pool_mu
static double wall_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static double g_drv_ms = 0;      // time spent in hipMalloc for arena segments, their number and bytes (psgx_arena_stats)
static i64 g_drv_calls = 0, g_drv_bytes = 0;
static size_t g_limit = 0;      // psg_set_memory_limit: the library hands out at most this many bytes (0 = what the device has)
static void note_use(size_t add) { g_in_use += add; if (g_in_use > g_peak) g_peak = g_in_use; }

static void *arena_carve(size_t need) {   // best fit over all segments; g_pool_mu held
  int bs = -1;
  size_t boff = 0, blen = ~(size_t)0;
  for (int k = 0; k < (int)g_segs.size(); ++k)
    for (auto &f : g_segs[k].free)
      if (f.second >= need && f.second < blen) { bs = k; boff = f.first; blen = f.second; }
  if (bs < 0) return nullptr;
  ArenaSeg &S = g_segs[(size_t)bs];
  S.free.erase(boff);
  if (blen > need) S.free[boff + need] = blen - need;
  S.used += need;
  note_use(need);
  void *p = S.base + boff;
  g_big_live[p] = {bs, need};
  return p;
}

static void arena_release_empty() {   // give wholly free segments back to the driver; g_pool_mu NOT held
  std::vector<void *> bases;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &S : g_segs)
      if (S.base && S.used == 0) { bases.push_back(S.base); S.base = nullptr; S.size = 0; S.free.clear(); }
  }
  if (!bases.empty()) {
    if (g_stream) (void)hipStreamSynchronize(g_stream);
    if (g_side) (void)hipStreamSynchronize(g_side);
  }
  for (void *b : bases) (void)hipFree(b);
}

hipError_t pool_alloc(void **p, size_t bytes) {
  if (g_limit) {   // a budget below the device's memory (construct_sa --hbm-limit): the spill paths run on any device
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_in_use + bytes > g_limit) { *p = nullptr; return hipErrorOutOfMemory; }
  }
  if (bytes >= BIG) {
    const size_t need = (bytes + GRAN - 1) / GRAN * GRAN;
    size_t total = 0;
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      if ((*p = arena_carve(need)) != nullptr) return hipSuccess;
      for (auto &S : g_segs) total += S.size;
    }
    // new segment: at least the request, otherwise doubling the arena up to 8 GiB steps
    size_t seg = std::max(need, std::min(SEG_MAX, std::max((size_t)64 << 20, total)));
    void *base = nullptr;
    const double t_drv = wall_ms();
    hipError_t e = hipMalloc(&base, seg);
    if (e != hipSuccess && seg > need) { (void)hipGetLastError(); seg = need; e = hipMalloc(&base, seg); }
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_pool_mu); g_drv_ms += wall_ms() - t_drv; g_drv_calls += 1; g_drv_bytes += (i64)seg; }
    if (e != hipSuccess) {   // give everything unused back and retry once
      (void)hipGetLastError();
      pool_trim();
      e = hipMalloc(&base, seg);
    }
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int slot = -1;
    for (int k = 0; k < (int)g_segs.size(); ++k) if (!g_segs[(size_t)k].base) { slot = k; break; }
    if (slot < 0) { g_segs.emplace_back(); slot = (int)g_segs.size() - 1; }
    ArenaSeg &S = g_segs[(size_t)slot];
    S.base = (char *)base; S.size = seg; S.used = 0; S.free.clear(); S.free[0] = seg;
    *p = arena_carve(need);
    return *p ? hipSuccess : hipErrorOutOfMemory;
  }
  size_t need = pool_round(bytes);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_free.lower_bound(need);
    if (it != g_pool_free.end() && it->first == need) {
      *p = it->second;
      g_pool_live[*p] = it->first;
      note_use(it->first);
      g_pool_free.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, need);
  if (e != hipSuccess) {  // give cached blocks back and retry once
    (void)hipGetLastError();
    pool_trim();
    e = hipMalloc(p, need);
  }
  if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_pool_mu); g_pool_live[*p] = need; note_use(need); g_small_reserved += need; }
  return e;
}

void pool_free(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto bg = g_big_live.find(p);
  if (bg != g_big_live.end()) {
    ArenaSeg &S = g_segs[(size_t)bg->second.first];
    size_t off = (size_t)((char *)p - S.base), len = bg->second.second;
    g_big_live.erase(bg);
    S.used -= len;
    g_in_use -= len;
    auto nx = S.free.lower_bound(off);
    if (nx != S.free.end() && off + len == nx->first) { len += nx->second; nx = S.free.erase(nx); }   // merge with the next range
    if (nx != S.free.begin()) {
      auto pv = std::prev(nx);
      if (pv->first + pv->second == off) { off = pv->first; len += pv->second; S.free.erase(pv); }    // and with the previous one
    }
    S.free[off] = len;
    return;
  }
  auto it = g_pool_live.find(p);
  if (it == g_pool_live.end()) { (void)hipFree(p); return; }
  g_pool_free.emplace(it->second, p);
  g_in_use -= it->second;
  g_pool_live.erase(it);
}

size_t pool_cached_bytes();
// bytes a new structure may still take: under a budget what is left of it, otherwise what the driver and the cache have
size_t mem_available() {
  size_t lim, used;
  { std::lock_guard<std::mutex> lk(g_pool_mu); lim = g_limit; used = g_in_use; }
  if (lim) return lim > used ? lim - used : 0;
  size_t f = 0, t = 0;
  (void)hipMemGetInfo(&f, &t);
  return f + pool_cached_bytes();
}

size_t pool_cached_bytes() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  size_t t = 0;
  for (auto &kv : g_pool_free) t += kv.first;
  for (auto &S : g_segs) t += S.size - S.used;
  return t;
}

void pool_trim() {
  std::vector<void *> blocks;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &kv : g_pool_free) { blocks.push_back(kv.second); g_small_reserved -= kv.first; }
    g_pool_free.clear();
  }
  if (!blocks.empty() && g_stream) { (void)hipStreamSynchronize(g_stream); if (g_side) (void)hipStreamSynchronize(g_side); }
  for (void *b : blocks) (void)hipFree(b);
  arena_release_empty();
}

// ---- event recycling -----------------------------------------------------------------------
static std::vector<hipEvent_t> g_events;
hipEvent_t event_acquire() {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_events.empty()) { hipEvent_t e = g_events.back(); g_events.pop_back(); return e; }
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
void event_release(hipEvent_t e) {
  if (!e) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_events.push_back(e);
}

// ---- polling stream sync -----------------------------------------------------------------
hipError_t sync_stream() {
  hipEvent_t e = event_acquire();
  if (!e) return hipStreamSynchronize(stream());
  hipError_t rc = hipEventRecord(e, stream());
  if (rc == hipSuccess) {
    while ((rc = hipEventQuery(e)) == hipErrorNotReady) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
  event_release(e);
  if (rc != hipSuccess) { (void)hipGetLastError(); rc = hipStreamSynchronize(stream()); }
  return rc;
}

// ---- pinned host scratch -----------------------------------------------------------------
static void *g_pin[16] = {nullptr};
static size_t g_pin_bytes[16] = {0};
void *pinned_buf(int slot, size_t bytes) {
  if (slot < 0 || slot >= 16) return nullptr;
  if (bytes < 4096) bytes = 4096;
  if (g_pin_bytes[slot] >= bytes) return g_pin[slot];
  if (g_pin[slot]) { if (g_stream) (void)hipStreamSynchronize(g_stream);  /* rare: keep the blocking wait */ (void)hipHostFree(g_pin[slot]); g_pin[slot] = nullptr; g_pin_bytes[slot] = 0; }
  size_t cap = bytes + bytes / 2;
  if (hipHostMalloc(&g_pin[slot], cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); g_pin[slot] = nullptr; return nullptr; }
  g_pin_bytes[slot] = cap;
  return g_pin[slot];
}

// ---- host <-> device copies ------------------------------------------------------------------
// Pinned host memory (psg_host_alloc, torch pinned tensors) is copied by one DMA.  Pageable memory goes through
// TWO pinned staging buffers: the host fills one while the DMA drains the other, and the only waits are for the
// buffer about to be refilled (was: one buffer, a full stream sync per 32 MiB piece).
static const size_t STAGE_BYTES = (size_t)32 << 20;
static bool is_pinned_host(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}
static hipError_t wait_event_poll(hipEvent_t e) {
  hipError_t rc;
  while ((rc = hipEventQuery(e)) == hipErrorNotReady) {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  return rc;
}
// pageable <-> pinned copies of a piece with a few helper threads: one core moves ~30 GiB/s (and takes page faults on
// fresh memory one at a time), the PCIe link 50 GiB/s
static void host_copy(char *dst, const char *src, size_t n) {
  const size_t MINPAR = (size_t)4 << 20;
  if (n < MINPAR) { memcpy(dst, src, n); return; }
  const int nt = 4;
  std::thread th[nt - 1];
  const size_t per = (n / nt + 4095) & ~(size_t)4095;
  for (int t = 1; t < nt; ++t) {
    const size_t a = std::min(n, per * t), b = std::min(n, per * (t + 1));
    th[t - 1] = std::thread([=] { if (b > a) memcpy(dst + a, src + a, b - a); });
  }
  memcpy(dst, src, std::min(n, per));
  for (int t = 1; t < nt; ++t) th[t - 1].join();
}

int copy_h2d(void *d, const void *h, size_t bytes) {
  if (bytes >= ((size_t)1 << 16) && is_pinned_host(h)) {
    PSG_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream()));
    PSG_HIP(sync_stream());
    return 0;
  }
  const size_t piece = std::min(STAGE_BYTES, std::max<size_t>(bytes, 4096));
  char *pin = (char *)pinned_buf(7, 2 * piece);
  if (!pin) { set_error("pinned staging buffer allocation failed"); return PSG_ENOMEM; }
  hipEvent_t ev[2] = {event_acquire(), event_acquire()};
  bool used[2] = {false, false};
  int rc = 0;
  size_t k = 0;
  for (size_t off = 0; off < bytes; off += piece, ++k) {
    const int s = (int)(k & 1);
    const size_t n = std::min(piece, bytes - off);
    if (used[s] && wait_event_poll(ev[s]) != hipSuccess) { set_error("copy_h2d: device error"); rc = PSG_EDEVICE; break; }
    host_copy(pin + s * piece, (const char *)h + off, n);
    if (hipMemcpyAsync((char *)d + off, pin + s * piece, n, hipMemcpyHostToDevice, stream()) != hipSuccess ||
        hipEventRecord(ev[s], stream()) != hipSuccess) { set_error("copy_h2d: hipMemcpyAsync failed"); rc = PSG_EDEVICE; break; }
    used[s] = true;
  }
  hipError_t e = sync_stream();
  event_release(ev[0]); event_release(ev[1]);
  if (!rc && e != hipSuccess) { set_error(std::string("copy_h2d: ") + hipGetErrorString(e)); rc = PSG_EDEVICE; }
  return rc;
}
int copy_d2h(void *h, const void *d, size_t bytes) {
  if (bytes >= ((size_t)1 << 16) && is_pinned_host(h)) {
    PSG_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream()));
    PSG_HIP(sync_stream());
    return 0;
  }
  const size_t piece = std::min(STAGE_BYTES, std::max<size_t>(bytes, 4096));
  char *pin = (char *)pinned_buf(7, 2 * piece);
  if (!pin) { set_error("pinned staging buffer allocation failed"); return PSG_ENOMEM; }
  hipEvent_t ev[2] = {event_acquire(), event_acquire()};
  int rc = 0;
  const size_t np = (bytes + piece - 1) / piece;
  for (size_t k = 0; k <= np; ++k) {     // piece k is in flight while piece k-1 is copied out of its staging buffer
    if (k < np) {
      const int s = (int)(k & 1);
      const size_t off = k * piece, n = std::min(piece, bytes - off);
      if (hipMemcpyAsync(pin + s * piece, (const char *)d + off, n, hipMemcpyDeviceToHost, stream()) != hipSuccess ||
          hipEventRecord(ev[s], stream()) != hipSuccess) { set_error("copy_d2h: hipMemcpyAsync failed"); rc = PSG_EDEVICE; break; }
    }
    if (k >= 1) {
      const int s = (int)((k - 1) & 1);
      const size_t off = (k - 1) * piece, n = std::min(piece, bytes - off);
      if (wait_event_poll(ev[s]) != hipSuccess) { set_error("copy_d2h: device error"); rc = PSG_EDEVICE; break; }
      host_copy((char *)h + off, pin + s * piece, n);
    }
  }
  hipError_t e = sync_stream();
  event_release(ev[0]); event_release(ev[1]);
  if (!rc && e != hipSuccess) { set_error(std::string("copy_d2h: ") + hipGetErrorString(e)); rc = PSG_EDEVICE; }
  return rc;
}

// ---- single-workgroup scan ------------------------------------------------------------
// 1024 threads x 4 consecutive values per iteration (the tile-sum arrays have ~1 M entries)
__global__ __launch_bounds__(1024) void scan_u64_kernel(u64 *vals, i64 n, u64 *total) {
  __shared__ u64 wsum[16];
  __shared__ u64 carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  int w = threadIdx.x >> 6;
  for (i64 base = 0; base < n; base += 4096) {
    i64 k = base + (i64)threadIdx.x * 4;
    u64 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = k + q < n ? vals[k + q] : 0;
    u64 s = v[0] + v[1] + v[2] + v[3];
    u64 inc = wave_incl_scan(s);
    if (lane_id() == 63) wsum[w] = inc;
    __syncthreads();
    u64 pre = carry_s, tot = 0;
    for (int q = 0; q < 16; ++q) { u64 x = wsum[q]; if (q < w) pre += x; tot += x; }
    u64 run = pre + inc - s;
#pragma unroll
    for (int q = 0; q < 4; ++q) { if (k + q < n) vals[k + q] = run; run += v[q]; }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = carry_s;
}

int scan_u64_inplace(u64 *d_vals, i64 n, u64 *d_total) {
  hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, stream(), d_vals, n, d_total);
  PSG_HIP(hipGetLastError());
  return 0;
}

// ---- bitcopy ---------------------------------------------------------------------------
// one thread per destination word that is touched; edge words use atomics with masks.
__global__ __launch_bounds__(PSG_WG) void bitcopy_kernel(u32 *dst, i64 dst_bit, const u32 *src, i64 src_bit, i64 nbits,
                                                           i64 src_words) {
  i64 first_w = dst_bit >> 5, last_w = (dst_bit + nbits - 1) >> 5;
  i64 w = first_w + (i64)blockIdx.x * PSG_WG + threadIdx.x;
  if (w > last_w) return;
  i64 lo = w << 5;                       // first dst bit of this word
  i64 b0 = lo < dst_bit ? dst_bit : lo;  // valid range [b0, b1)
  i64 b1 = lo + 32 < dst_bit + nbits ? lo + 32 : dst_bit + nbits;
  int cnt = (int)(b1 - b0);
  u32 v = get_bits(src, src_bit + (b0 - dst_bit), cnt, src_words) << (int)(b0 - lo);
  if (cnt == 32) { dst[w] = v; return; }
  u32 mask = (cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u)) << (int)(b0 - lo);
  atomicAnd(&dst[w], ~mask);
  atomicOr(&dst[w], v);
}

__global__ __launch_bounds__(PSG_WG) void popcount_kernel(const u32 *bits, i64 nbits, unsigned long long *out) {
  __shared__ u64 scratch[8];
  i64 nw = (nbits + 31) >> 5;
  u64 acc = 0;
  for (i64 w = (i64)blockIdx.x * PSG_WG + threadIdx.x; w < nw; w += (i64)gridDim.x * PSG_WG) {
    u32 x = bits[w];
    if (w == nw - 1 && (nbits & 31)) x &= (1u << (nbits & 31)) - 1u;
    acc += __popc(x);
  }
  u64 tot = block_sum<u64>(acc, scratch);
  if (threadIdx.x == 0 && tot) atomicAdd(out, (unsigned long long)tot);
}
}  // namespace psg

using namespace psg;

extern "C" {

int psg_init(int device) {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0) {
    set_error(std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
              " -- psascan_amd has no CPU fallback");
    return PSG_EDEVICE;
  }
  PSG_REQUIRE(device >= 0 && device < cnt, "device index out of range");
  PSG_HIP(hipSetDevice(device));
  g_device = device;
  if (!g_stream) {
    PSG_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_own_stream = true;
  }
  g_inited = true;
  return 0;
}

int psg_set_stream(void *s) {
  // the allocator hands freed blocks out again at once because everything runs on ONE stream: work still in
  // flight on the outgoing stream (ours or the caller's) must drain before the next stream may reuse its buffers
  if (g_stream) (void)hipStreamSynchronize(g_stream);
  if (g_own_stream && g_stream) (void)hipStreamDestroy(g_stream);
  g_own_stream = false;
  g_stream = (hipStream_t)s;
  if (!s) {
    PSG_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_own_stream = true;
  }
  g_inited = true;
  return 0;
}

const char *psg_last_error(void) { return g_err.c_str(); }
double psg_last_kernel_ms(void) { return g_last_ms; }

int psg_device_name(char *buf, int cap) {
  hipDeviceProp_t p;
  int dev = 0;
  PSG_HIP(hipGetDevice(&dev));
  PSG_HIP(hipGetDeviceProperties(&p, dev));
  snprintf(buf, cap, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return 0;
}

// The host side of this path is copies between pinned buffers, the page cache and the device: on a two-socket host the
// far socket costs ~20 % of the .sa5 writer's rate (gpurun_out numa_probe: sink 1.38 s near, 1.67 s far, 20 GiB).  Binds
// every thread of the calling process (and so every thread created later) to the CPUs of the device's NUMA node,
// intersected with the mask the process was started with; *node = that node, or -1 when there is nothing to do
// (one node, no sysfs, PSG_NO_NUMA_BIND set).  Never an error: placement is a hint.
int psgx_bind_threads_near_device(int *node) {
  if (node) *node = -1;
  if (getenv("PSG_NO_NUMA_BIND")) return 0;
  int dev = 0;
  char bus[64] = {0};
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) return 0;
  for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
  char path[256];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
  int nd = -1;
  if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &nd) != 1) nd = -1; fclose(f); }
  if (nd < 0) return 0;
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", nd);
  cpu_set_t want, have, both;
  CPU_ZERO(&want);
  int cpus_total = 0;
  if (FILE *f = fopen(path, "r")) {              // "0-63,128-191"
    int a, b;
    while (fscanf(f, "%d", &a) == 1) {
      b = a;
      int ch = fgetc(f);
      if (ch == '-') { if (fscanf(f, "%d", &b) != 1) break; ch = fgetc(f); }
      for (int c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET(c, &want); ++cpus_total; }
      if (ch != ',') break;
    }
    fclose(f);
  }
  if (cpus_total == 0 || sched_getaffinity(0, sizeof have, &have) != 0) return 0;
  CPU_AND(&both, &want, &have);
  if (CPU_COUNT(&both) == 0 || CPU_EQUAL(&both, &have)) return 0;
  if (DIR *d = opendir("/proc/self/task")) {
    while (dirent *e = readdir(d)) {
      const int tid = atoi(e->d_name);
      if (tid > 0) sched_setaffinity(tid, sizeof both, &both);
    }
    closedir(d);
  } else sched_setaffinity(0, sizeof both, &both);
  if (node) *node = nd;
  return 0;
}

int psg_malloc(void **d_ptr, int64_t bytes) {
  PSG_REQUIRE(d_ptr && bytes >= 0, "psg_malloc");
  hipError_t e = psg::pool_alloc(d_ptr, (size_t)(bytes < 16 ? 16 : bytes));
  if (e != hipSuccess) { set_error(std::string("hipMalloc: ") + hipGetErrorString(e)); *d_ptr = nullptr; return e == hipErrorOutOfMemory ? PSG_ENOMEM : PSG_EDEVICE; }
  return 0;
}
int psg_free(void *d_ptr) { if (d_ptr) psg::pool_free(d_ptr); return 0; }
int psg_trim(void) { psg::pool_trim(); return 0; }
int psg_memset(void *d_ptr, int value, int64_t bytes) {
  if (bytes > 0) PSG_HIP(hipMemsetAsync(d_ptr, value, (size_t)bytes, stream()));
  return 0;
}
int psg_h2d(void *d, const void *h, int64_t bytes) {
  if (bytes > 0) return psg::copy_h2d(d, h, (size_t)bytes);
  return 0;
}
int psg_d2h(void *h, const void *d, int64_t bytes) {
  if (bytes > 0) return psg::copy_d2h(h, d, (size_t)bytes);
  return 0;
}
int psg_d2d(void *dd, const void *ds, int64_t bytes) {
  if (bytes > 0) PSG_HIP(hipMemcpyAsync(dd, ds, (size_t)bytes, hipMemcpyDeviceToDevice, stream()));
  return 0;
}
int psg_sync(void) { PSG_HIP(psg::sync_stream()); return 0; }
int psg_host_alloc(void **h_ptr, int64_t bytes) {
  PSG_REQUIRE(h_ptr && bytes >= 0, "psg_host_alloc");
  *h_ptr = nullptr;
  hipError_t e = hipHostMalloc(h_ptr, (size_t)(bytes < 16 ? 16 : bytes), hipHostMallocDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); set_error(std::string("hipHostMalloc: ") + hipGetErrorString(e)); *h_ptr = nullptr; return PSG_ENOMEM; }
  return 0;
}
int psg_host_free(void *h_ptr) {
  if (!h_ptr) return 0;
  (void)psg::sync_stream();
  if (g_side) (void)hipStreamSynchronize(g_side);
  PSG_HIP(hipHostFree(h_ptr));
  return 0;
}
// ---- download in the background: a worker thread with its own stream and its own pair of pinned staging buffers
// drains a device buffer into (pageable) host memory while the library's stream goes on with the next kernels.
// construct_sa uses it for the partial SA of a finished half-block (4-5 bytes per symbol into fresh host memory:
// page faults bound the rate at ~10 GiB/s, which used to be a third of a --device-sort run).
struct psg_copy {
  std::thread th;
  int rc = 0;
  std::string err;
};
// what a background copy needs: kept for the next one (a pinned allocation costs ~10 ms per 64 MiB, a chunked pass starts
// one copy per chunk)
struct CopyLane { hipStream_t st = nullptr; char *pin = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; };
static std::mutex g_lane_mu;
static std::vector<CopyLane> g_lanes;
static bool lane_acquire(CopyLane &L, hipError_t &e, const char *&what) {
  {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    if (!g_lanes.empty()) { L = g_lanes.back(); g_lanes.pop_back(); return true; }
  }
  L = CopyLane();
  what = "hipStreamCreate";
  if ((e = hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking)) != hipSuccess) return false;
  what = "hipHostMalloc";
  if ((e = hipHostMalloc((void **)&L.pin, 2 * psg::STAGE_BYTES, hipHostMallocDefault)) != hipSuccess) { (void)hipStreamDestroy(L.st); return false; }
  what = "hipEventCreate";
  if ((e = hipEventCreateWithFlags(&L.ev[0], hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&L.ev[1], hipEventDisableTiming)) != hipSuccess) {
    if (L.ev[0]) (void)hipEventDestroy(L.ev[0]);
    (void)hipHostFree(L.pin); (void)hipStreamDestroy(L.st);
    return false;
  }
  return true;
}
static void lane_release(const CopyLane &L) { std::lock_guard<std::mutex> lk(g_lane_mu); g_lanes.push_back(L); }
int psg_d2h_begin(void *h_dst, void *d_src, int64_t bytes, int free_src, psg_copy_t **out) {
  PSG_REQUIRE(out && bytes >= 0 && (bytes == 0 || (h_dst && d_src)), "psg_d2h_begin");
  PSG_HIP(psg::sync_stream());          // the source is complete before the worker reads it
  int dev = 0;
  PSG_HIP(hipGetDevice(&dev));
  psg_copy *c = new psg_copy();
  c->th = std::thread([=]() {
    auto fail = [&](const char *what, hipError_t e) { c->rc = PSG_EDEVICE; c->err = std::string("psg_d2h_begin worker: ") + what + ": " + hipGetErrorString(e); };
    hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) { fail("hipSetDevice", e); return; }
    struct Release { void *p; ~Release() { if (p) psg::pool_free(p); } } release{free_src ? d_src : nullptr};   // the source goes back to the allocator the moment it is drained
    if (bytes == 0) return;
    CopyLane L;
    const char *what = "";
    if (!lane_acquire(L, e, what)) { fail(what, e); return; }
    hipStream_t st = L.st;
    char *pin = L.pin;
    hipEvent_t *ev = L.ev;
    const size_t total = (size_t)bytes, piece = psg::STAGE_BYTES;
    const size_t np = (total + piece - 1) / piece;
    for (size_t k = 0; k <= np && !c->rc; ++k) {     // piece k is in flight while piece k-1 leaves its staging buffer
      if (k < np) {
        const int b = (int)(k & 1);
        const size_t off = k * piece, n = std::min(piece, total - off);
        if ((e = hipMemcpyAsync(pin + b * piece, (const char *)d_src + off, n, hipMemcpyDeviceToHost, st)) != hipSuccess ||
            (e = hipEventRecord(ev[b], st)) != hipSuccess) { fail("hipMemcpyAsync", e); break; }
      }
      if (k >= 1) {
        const int b = (int)((k - 1) & 1);
        const size_t off = (k - 1) * piece, n = std::min(piece, total - off);
        if ((e = hipEventSynchronize(ev[b])) != hipSuccess) { fail("hipEventSynchronize", e); break; }
        psg::host_copy((char *)h_dst + off, pin + b * piece, n);
      }
    }
    (void)hipStreamSynchronize(st);
    lane_release(L);
  });
  *out = c;
  return 0;
}
// the other direction: (pageable) host memory into a device buffer in the background -- the next tail chunk of a text
// that stays in host memory goes up while the current one is streamed (stream.hpp:104-106 reads the tail from the
// text file through async_backward_stream_reader: the same double buffering, with PCIe in place of the disk)
int psg_h2d_begin(void *d_dst, const void *h_src, int64_t bytes, psg_copy_t **out) {
  PSG_REQUIRE(out && bytes >= 0 && (bytes == 0 || (d_dst && h_src)), "psg_h2d_begin");
  int dev = 0;
  PSG_HIP(hipGetDevice(&dev));
  psg_copy *c = new psg_copy();
  c->th = std::thread([=]() {
    auto fail = [&](const char *what, hipError_t e) { c->rc = PSG_EDEVICE; c->err = std::string("psg_h2d_begin worker: ") + what + ": " + hipGetErrorString(e); };
    hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) { fail("hipSetDevice", e); return; }
    if (bytes == 0) return;
    CopyLane L;
    const char *what = "";
    if (!lane_acquire(L, e, what)) { fail(what, e); return; }
    hipStream_t st = L.st;
    char *pin = L.pin;
    hipEvent_t *ev = L.ev;
    const size_t total = (size_t)bytes, piece = psg::STAGE_BYTES;
    if (psg::is_pinned_host(h_src)) {     // page-locked source: one DMA, no staging
      // in pieces, one in flight at a time: the copy engine is shared with the library's stream, whose small
      // synchronous copies (tables, counters) would otherwise queue behind a multi-GiB transfer (measured: +450 ms per
      // configs[2] step with one 2 GiB DMA per half-block)
      for (size_t off = 0; off < total && !c->rc; off += piece) {
        const size_t n = std::min(piece, total - off);
        if ((e = hipMemcpyAsync((char *)d_dst + off, (const char *)h_src + off, n, hipMemcpyHostToDevice, st)) != hipSuccess) { fail("hipMemcpyAsync", e); break; }
        if ((e = hipStreamSynchronize(st)) != hipSuccess) { fail("hipStreamSynchronize", e); break; }
      }
      lane_release(L);
      return;
    }
    bool used[2] = {false, false};
    size_t k = 0;
    for (size_t off = 0; off < total && !c->rc; off += piece, ++k) {
      const int b = (int)(k & 1);
      const size_t n = std::min(piece, total - off);
      if (used[b] && (e = hipEventSynchronize(ev[b])) != hipSuccess) { fail("hipEventSynchronize", e); break; }
      psg::host_copy(pin + b * piece, (const char *)h_src + off, n);
      if ((e = hipMemcpyAsync((char *)d_dst + off, pin + b * piece, n, hipMemcpyHostToDevice, st)) != hipSuccess ||
          (e = hipEventRecord(ev[b], st)) != hipSuccess) { fail("hipMemcpyAsync", e); break; }
      used[b] = true;
    }
    if ((e = hipStreamSynchronize(st)) != hipSuccess && !c->rc) fail("hipStreamSynchronize", e);
    lane_release(L);
  });
  *out = c;
  return 0;
}
int psg_copy_wait(psg_copy_t *c) {
  if (!c) return 0;
  if (c->th.joinable()) c->th.join();
  const int rc = c->rc;
  if (rc) set_error(c->err);
  delete c;
  return rc;
}

int psg_set_memory_limit(int64_t bytes) {
  PSG_REQUIRE(bytes >= 0, "psg_set_memory_limit");
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_limit = (size_t)bytes;
  return 0;
}
int psg_device_memory(int64_t *free_bytes, int64_t *total_bytes) {
  size_t f = 0, t = 0;
  PSG_HIP(hipMemGetInfo(&f, &t));
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_limit) { t = std::min(t, g_limit); f = std::min(f, g_limit > g_in_use ? g_limit - g_in_use : 0); }
  }
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return 0;
}
int psgx_arena_stats(double *driver_seconds, int64_t *segments, int64_t *bytes) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (driver_seconds) *driver_seconds = g_drv_ms / 1e3;
  if (segments) *segments = g_drv_calls;
  if (bytes) *bytes = g_drv_bytes;
  return 0;
}
int psg_mem_stats(int64_t *in_use, int64_t *peak_in_use, int64_t *reserved) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  size_t res = g_small_reserved;
  for (auto &S : g_segs) res += S.size;
  if (in_use) *in_use = (int64_t)g_in_use;
  if (peak_in_use) *peak_in_use = (int64_t)g_peak;
  if (reserved) *reserved = (int64_t)res;
  return 0;
}

int psg_bitcopy(uint32_t *d_dst, int64_t dst_bit, const uint32_t *d_src, int64_t src_bit, int64_t nbits) {
  PSG_REQUIRE(dst_bit >= 0 && src_bit >= 0 && nbits >= 0, "psg_bitcopy");
  if (nbits == 0) return 0;
  i64 nwords = ((dst_bit + nbits - 1) >> 5) - (dst_bit >> 5) + 1;
  i64 src_words = (src_bit + nbits + 31) >> 5;
  PSG_REQUIRE(nwords < (1ll << 32), "psg_bitcopy: more than 2^37 bits in one call");   // one thread per word, a launch holds < 2^32 threads
  hipLaunchKernelGGL(bitcopy_kernel, dim3((unsigned)cdiv(nwords, PSG_WG)), dim3(PSG_WG), 0, stream(), d_dst, dst_bit, d_src,
                     src_bit, nbits, src_words);
  PSG_HIP(hipGetLastError());
  return 0;
}

int psg_popcount(const uint32_t *d_bits, int64_t nbits, int64_t *ones) {
  PSG_REQUIRE(nbits >= 0 && ones, "psg_popcount");
  DevBuf acc;
  if (int rc = acc.alloc(8)) return rc;
  PSG_HIP(hipMemsetAsync(acc.p, 0, 8, stream()));
  if (nbits > 0) {
    i64 nw = (nbits + 31) >> 5;
    unsigned grid = (unsigned)std::min<i64>(cdiv(nw, PSG_WG), 4096);
    hipLaunchKernelGGL(popcount_kernel, dim3(grid), dim3(PSG_WG), 0, stream(), d_bits, nbits, acc.as<unsigned long long>());
    PSG_HIP(hipGetLastError());
  }
  u64 v = 0;
  if (int rc_ = psg::copy_d2h(&v, acc.p, 8)) return rc_;
  PSG_HIP(psg::sync_stream());
  *ones = (i64)v;
  return 0;
}

}  // extern "C"
