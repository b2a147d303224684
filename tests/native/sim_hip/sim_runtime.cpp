// TEST INFRASTRUCTURE ONLY -- the simulated HIP runtime behind tests/native/sim_hip/hip/hip_runtime.h, the CPU stand-ins of the respond
// launchers host_respond.hip calls (same contracts as chalametpir_amd/csrc/cpir_internal.hpp states them, incl. the polled fill protocol),
// and the few library functions of other translation units that host_respond.hip links against (tuning accessors, error plumbing,
// SlotMap::reset).  See the header for what is and is not faithful.  Nothing here is part of the product.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "cpir_internal.hpp"
#include "sim_control.hpp"

thread_local sim_idx threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0};
thread_local dim3 blockDim, gridDim;

namespace {

// ---- chaos: every operation of the simulated device may start a little late -------------------------------------------------------------
std::atomic<uint32_t> g_chaos_us{0};
void chaos_pause() {
  const uint32_t c = g_chaos_us.load(std::memory_order_relaxed);
  if (!c) return;
  static thread_local std::minstd_rand rng((unsigned)std::hash<std::thread::id>()(std::this_thread::get_id()));
  const uint32_t us = rng() % (c + 1);
  if (us == 0) return;
  if (us < 5) std::this_thread::yield();
  else std::this_thread::sleep_for(std::chrono::microseconds(us));
}

// ---- memory registry --------------------------------------------------------------------------------------------------------------------
struct Range {
  size_t bytes;
  bool host;
  bool owned;  // allocated here (freed by hipFree / hipHostFree) as against registered by the caller
};
std::mutex g_mem_mu;
std::map<uintptr_t, Range> g_ranges;  // base -> range
std::atomic<int64_t> g_live_blocks{0};

bool find_range(const void* p, uintptr_t* base, Range* r) {
  std::lock_guard<std::mutex> lk(g_mem_mu);
  auto it = g_ranges.upper_bound(reinterpret_cast<uintptr_t>(p));
  if (it == g_ranges.begin()) return false;
  --it;
  if (reinterpret_cast<uintptr_t>(p) >= it->first + it->second.bytes) return false;
  *base = it->first, *r = it->second;
  return true;
}

// ---- streams and events -----------------------------------------------------------------------------------------------------------------
}  // namespace

struct sim_stream {
  std::mutex mu;
  std::condition_variable cv, idle_cv;
  std::deque<std::function<void()>> ops;
  uint64_t enqueued = 0, done = 0;
  bool stop = false;
  std::thread th;
  sim_stream() {
    th = std::thread([this] {
      for (;;) {
        std::function<void()> op;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return stop || !ops.empty(); });
          if (ops.empty()) return;
          op = std::move(ops.front());
          ops.pop_front();
        }
        chaos_pause();
        op();
        {
          std::lock_guard<std::mutex> lk(mu);
          done++;
        }
        idle_cv.notify_all();
      }
    });
  }
  ~sim_stream() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
  void push(std::function<void()> op) {
    {
      std::lock_guard<std::mutex> lk(mu);
      ops.push_back(std::move(op));
      enqueued++;
    }
    cv.notify_one();
  }
  void drain() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t want = enqueued;
    idle_cv.wait(lk, [&] { return done >= want; });
  }
};

struct sim_event {
  std::mutex mu;
  std::condition_variable cv;
  uint64_t recorded = 0, completed = 0;
};

namespace {
std::mutex g_streams_mu;
std::vector<sim_stream*> g_streams;  // every live stream (hipFree and friends wait for the whole device)
sim_stream* default_stream() {
  static sim_stream* s = [] {
    sim_stream* x = new sim_stream;
    std::lock_guard<std::mutex> lk(g_streams_mu);
    g_streams.push_back(x);
    return x;
  }();
  return s;
}
sim_stream* resolve(hipStream_t s) { return s ? s : default_stream(); }
void device_synchronize() {
  std::vector<sim_stream*> all;
  {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    all = g_streams;
  }
  for (sim_stream* s : all) s->drain();
}
thread_local int t_device = 0;
}  // namespace

void sim_enqueue(hipStream_t s, std::function<void()> op) { resolve(s)->push(std::move(op)); }

void sim_enqueue_kernel(hipStream_t s, dim3 grid, dim3 block, std::function<void()> body) {
  resolve(s)->push([grid, block, body = std::move(body)] {
    gridDim = grid, blockDim = block;
    for (unsigned b = 0; b < grid.x; b++) {
      blockIdx = {b, 0, 0};
      for (unsigned t = block.x; t-- > 0;) {  // descending: thread 0 runs last (see the header)
        threadIdx = {t, 0, 0};
        body();
      }
    }
  });
}

// ---- memory -------------------------------------------------------------------------------------------------------------------------------
static hipError_t alloc_block(void** p, size_t bytes, bool host) {
  if (!p) return hipErrorInvalidValue;
  void* q = nullptr;
  if (posix_memalign(&q, 4096, bytes ? bytes : 1) != 0) return hipErrorOutOfMemory;
  memset(q, 0xA5, bytes);  // (neither kind of memory arrives zeroed)
  {
    std::lock_guard<std::mutex> lk(g_mem_mu);
    g_ranges[reinterpret_cast<uintptr_t>(q)] = Range{bytes, host, true};
  }
  g_live_blocks.fetch_add(1);
  *p = q;
  return hipSuccess;
}
static hipError_t free_block(void* p, bool host) {
  if (!p) return hipSuccess;
  device_synchronize();  // (the real calls wait for the device: scripts/probes/free_sync_probe.hip)
  {
    std::lock_guard<std::mutex> lk(g_mem_mu);
    auto it = g_ranges.find(reinterpret_cast<uintptr_t>(p));
    if (it == g_ranges.end() || it->second.host != host || !it->second.owned) return hipErrorInvalidValue;
    g_ranges.erase(it);
  }
  g_live_blocks.fetch_sub(1);
  free(p);
  return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) { return alloc_block(p, bytes, false); }
hipError_t hipFree(void* p) { return free_block(p, false); }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return alloc_block(p, bytes, true); }
hipError_t hipHostFree(void* p) { return free_block(p, true); }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) {
  uintptr_t base;
  Range r;
  if (!find_range(host, &base, &r) || !r.host) return hipErrorInvalidValue;
  *dev = host;
  return hipSuccess;
}
hipError_t hipHostRegister(void* p, size_t bytes, unsigned) {
  std::lock_guard<std::mutex> lk(g_mem_mu);
  g_ranges[reinterpret_cast<uintptr_t>(p)] = Range{bytes, true, false};
  return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* p) {
  uintptr_t base;
  Range r;
  if (!find_range(p, &base, &r)) return hipErrorInvalidValue;  // ordinary pageable memory
  attr->type = r.host ? hipMemoryTypeHost : hipMemoryTypeDevice;
  attr->device = 0;
  attr->devicePointer = const_cast<void*>(p);
  attr->hostPointer = r.host ? const_cast<void*>(p) : nullptr;
  return hipSuccess;
}
hipError_t hipDrvPointerGetAttributes(unsigned n, hipPointer_attribute* which, void** out, hipDeviceptr_t p) {
  uintptr_t base;
  Range r;
  if (!find_range(p, &base, &r)) return hipErrorInvalidValue;
  for (unsigned i = 0; i < n; i++) {
    if (which[i] == HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR) *static_cast<void**>(out[i]) = reinterpret_cast<void*>(base);
    else if (which[i] == HIP_POINTER_ATTRIBUTE_RANGE_SIZE) *static_cast<size_t*>(out[i]) = r.bytes;
    else return hipErrorInvalidValue;
  }
  return hipSuccess;
}

// ---- devices, errors ----------------------------------------------------------------------------------------------------------------------
hipError_t hipGetDevice(int* ordinal) {
  *ordinal = t_device;
  return hipSuccess;
}
hipError_t hipSetDevice(int ordinal) {
  if (ordinal < 0 || ordinal >= 4) return hipErrorInvalidDevice;
  t_device = ordinal;
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipDeviceSynchronize() {
  device_synchronize();
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t) { return "simulated"; }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) {
  *can = 1;
  return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) {
  *least = 0, *greatest = -1;
  return hipSuccess;
}

// ---- streams and events -------------------------------------------------------------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = new sim_stream;
  std::lock_guard<std::mutex> lk(g_streams_mu);
  g_streams.push_back(*s);
  return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
hipError_t hipStreamDestroy(hipStream_t s) {
  if (!s) return hipErrorInvalidValue;
  s->drain();
  {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    for (size_t i = 0; i < g_streams.size(); i++)
      if (g_streams[i] == s) {
        g_streams.erase(g_streams.begin() + i);
        break;
      }
  }
  delete s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  resolve(s)->drain();
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned) {
  *ev = new sim_event;
  g_live_blocks.fetch_add(1);
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t ev) {
  if (!ev) return hipErrorInvalidValue;
  device_synchronize();  // (an operation that is to signal it may still be queued; the real runtime keeps the event alive until then)
  delete ev;
  g_live_blocks.fetch_sub(1);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t s) {
  uint64_t ticket;
  {
    std::lock_guard<std::mutex> lk(ev->mu);
    ticket = ++ev->recorded;
  }
  resolve(s)->push([ev, ticket] {
    {
      std::lock_guard<std::mutex> lk(ev->mu);
      if (ev->completed < ticket) ev->completed = ticket;
    }
    ev->cv.notify_all();
  });
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t ev) {
  std::lock_guard<std::mutex> lk(ev->mu);
  return ev->completed >= ev->recorded ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventSynchronize(hipEvent_t ev) {
  std::unique_lock<std::mutex> lk(ev->mu);
  const uint64_t want = ev->recorded;
  ev->cv.wait(lk, [&] { return ev->completed >= want; });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t ev, unsigned) {
  uint64_t want;
  {
    std::lock_guard<std::mutex> lk(ev->mu);
    want = ev->recorded;
  }
  resolve(s)->push([ev, want] {
    std::unique_lock<std::mutex> lk(ev->mu);
    ev->cv.wait(lk, [&] { return ev->completed >= want; });
  });
  return hipSuccess;
}

// ---- asynchronous copies ------------------------------------------------------------------------------------------------------------------
static std::atomic<int> g_fail_copies{0};  // sim_fail_next_copies: that many hipMemcpyAsync calls fail (error paths of the upload)
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
  int f = g_fail_copies.load(std::memory_order_relaxed);
  while (f > 0 && !g_fail_copies.compare_exchange_weak(f, f - 1)) {
  }
  if (f > 0) return hipErrorUnknown;
  resolve(s)->push([dst, src, bytes] { memcpy(dst, src, bytes); });
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s) {
  resolve(s)->push([=] {
    for (size_t r = 0; r < height; r++) memcpy(static_cast<char*>(dst) + r * dpitch, static_cast<const char*>(src) + r * spitch, width);
  });
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s) {
  resolve(s)->push([dst, value, bytes] { memset(dst, value, bytes); });
  return hipSuccess;
}

// =============================================================================================================================================
// control surface of the test driver (sim_control.hpp)
// =============================================================================================================================================
namespace simctl {
std::atomic<int> inplace_seats{4}, upload_streams{2}, helper_spin_us{0}, fill_timeout_us{2000}, read_once{1}, batch_fusion{1}, takes_slot_map{1};
std::atomic<uint64_t> kernels_launched{0}, polled_kernels{0}, polled_gave_up{0};
void set_chaos_us(uint32_t us) { g_chaos_us.store(us); }
void fail_next_copies(int n) { g_fail_copies.store(n); }
int64_t live_blocks() { return g_live_blocks.load(); }
void device_sync() { device_synchronize(); }
}  // namespace simctl

// =============================================================================================================================================
// the library functions host_respond.hip links against, as stand-ins
// =============================================================================================================================================
namespace cpir {

void set_last_hip_error(hipError_t, const char*, const char*, int) {}
void journal_note(const char*, const void*, size_t, const char*, int) {}
void SlotMap::reset() {
  if (keep_dev) (void)hipFree(keep_dev);
  keep_dev = nullptr;
  keep_host.clear(), keep_bits.clear();
  n_kept = n_pad = n_orig = 0;
}

uint32_t respond_inplace_seats() { return (uint32_t)simctl::inplace_seats.load(); }
uint32_t respond_upload_streams() { return (uint32_t)simctl::upload_streams.load(); }
uint32_t respond_helper_spin_us() { return (uint32_t)simctl::helper_spin_us.load(); }
uint32_t respond_host_fill_timeout_us() { return (uint32_t)simctl::fill_timeout_us.load(); }
bool respond_read_once_applicable(const cpir_dtc_layout& L) { return L.packing == CPIR_PACK_PLANAR && simctl::read_once.load() != 0; }
bool respond_batch_fusion() { return simctl::batch_fusion.load() != 0; }
uint64_t respond_multi_pass_limit_bytes() { return 2560ull << 20; }
uint32_t respond_planar_pass_width(const cpir_dtc_layout&, uint32_t batch) {
  if (batch == 0) return 0;
  const uint32_t passes = (batch + CPIR_PLANAR_WIDE_MAX_QUERIES_PER_PASS - 1) / CPIR_PLANAR_WIDE_MAX_QUERIES_PER_PASS;
  return (batch + passes - 1) / passes;
}
bool respond_batch_takes_slot_map(const cpir_dtc_layout& L, uint32_t batch, bool, uint64_t q_len) {
  return L.packing == CPIR_PACK_PLANAR && simctl::takes_slot_map.load() != 0 && q_len < ((uint64_t)1 << 28) && batch != 0;
}

// The simulated database: `dtc` points at L.num_slots x L.num_cols u32 weights, row-major (slot-major); the response to a query is
//   r[c] = sum_n q[first slot + n] * w[n][c]   (mod 2^32)
// -- every query word is multiplied by a weight of its own slot and column, so a word that is stale, misplaced or missing changes the answer.
// The heavy loop is not instrumented (the sanitizers would spend the run in it); the words at both ends of every 512-slot step are read
// once more through instrumented code, so that a copy racing with the "kernel" is still seen.
__attribute__((no_sanitize("thread"))) __attribute__((no_sanitize("address"))) static void mac_range(const uint32_t* w, uint32_t C, const uint32_t* q,
                                                                                               const uint32_t* keep, uint64_t n0, uint64_t n1, uint32_t* r) {
  for (uint64_t n = n0; n < n1; n++) {
    const uint32_t x = keep ? q[keep[n]] : q[n];
    for (uint32_t c = 0; c < C; c++) r[c] += x * w[n * C + c];
  }
}
static thread_local volatile uint32_t g_touch_sink;  // (per stream thread)
static void touch(const uint32_t* q, const uint32_t* keep, uint64_t n0, uint64_t n1) {  // instrumented reads of the step's first and last word
  if (n1 > n0) g_touch_sink = (keep ? q[keep[n0]] : q[n0]) ^ (keep ? q[keep[n1 - 1]] : q[n1 - 1]);
}

struct FillWatch {  // the polled fill protocol of respond_planar.hip, seen from the kernel's side
  const PlanarHostFill* fill;
  PlanarHostFill copy;
  bool gave_up = false;
  // where each seat's query words are (word 0 of the slots this launch reads) and how many: every count a seat announces is PROBED at once --
  // the last word it claims to be in place is read through instrumented code, long before the stand-in's own walk gets there -- so that a
  // host that announces words it is still writing is caught by the sanitizer even though this "kernel" never outruns a memcpy
  const uint32_t* rows[CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t words = 0;
  uint32_t probed[CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS] = {0, 0, 0, 0};
  explicit FillWatch(const PlanarHostFill* f) : fill(f ? &copy : nullptr) {
    if (f) copy = *f;
  }
  void probe(uint32_t seat, uint32_t steps_in_place) {
    if (!rows[seat] || steps_in_place <= probed[seat]) return;
    probed[seat] = steps_in_place;
    const uint64_t upto = steps_in_place == 0xffffffffu ? words : std::min<uint64_t>(words, (uint64_t)steps_in_place * CPIR_PLANAR_SLOTS_PER_TILE);
    if (upto > 0) touch(rows[seat], nullptr, upto - 1, upto);
  }
  void wait_for_step(uint64_t step) {  // until steps [0, step] are in place for every seat, or the wait is given up FOR GOOD
    if (!fill || gave_up) return;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      uint32_t m = 0xffffffffu;
      for (uint32_t s = 0; s < copy.seats; s++) {
        const uint32_t x = __atomic_load_n(copy.progress + (size_t)s * (CPIR_FILL_LINES * 16), __ATOMIC_ACQUIRE);
        if (x) probe(s, x);
        m = x < m ? x : m;
      }
      if (m == 0xffffffffu || m > step) return;
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > copy.timeout_us) {
        gave_up = true;
        __atomic_store_n(copy.abort_flag, 1u, __ATOMIC_RELAXED);  // (device memory: read by the host only behind the hand-over / the stream)
        simctl::polled_gave_up.fetch_add(1);
        return;
      }
      std::this_thread::yield();
    }
  }
};

int launch_respond(const Device*, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len, uint64_t q_slot_offset, uint32_t batch,
                   uint32_t passes, uint32_t* r, uint32_t*, hipStream_t stream, const uint32_t* keep) {
  if (batch == 0 || passes == 0) return CPIR_ERR_INVALID_ARGUMENT;
  if (keep && L.packing != CPIR_PACK_PLANAR) return CPIR_ERR_INVALID_ARGUMENT;
  const uint64_t N = L.num_slots;
  const uint32_t C = L.num_cols;
  const uint32_t nq = batch * passes;
  simctl::kernels_launched.fetch_add(1);
  sim_enqueue(stream, [=] {
    for (uint32_t i = 0; i < nq; i++) {
      uint32_t* ri = r + (uint64_t)i * C;
      for (uint32_t c = 0; c < C; c++) ri[c] = 0;
      const uint32_t* qi = q + (uint64_t)i * q_len + q_slot_offset;
      touch(qi, keep, 0, N);
      mac_range(dtc, C, qi, keep, 0, N, ri);
    }
  });
  return CPIR_OK;
}

int launch_respond_read_once(const Device*, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len, uint64_t q_slot_offset,
                             uint32_t* r_prezeroed, hipStream_t stream, uint64_t step_lo, uint64_t step_hi, const PlanarHostFill* fill) {
  if (L.packing != CPIR_PACK_PLANAR) return CPIR_ERR_INVALID_ARGUMENT;
  const uint64_t N = L.num_slots, steps = (N + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE;
  if (step_hi == 0) step_hi = steps;
  if (step_lo >= step_hi || step_hi > steps) return CPIR_ERR_INVALID_ARGUMENT;
  if (fill && (!fill->progress || !fill->abort_flag || fill->seats != 1)) return CPIR_ERR_INVALID_ARGUMENT;
  const uint32_t C = L.num_cols;
  FillWatch watch(fill);
  watch.rows[0] = q + q_slot_offset, watch.words = N;
  simctl::kernels_launched.fetch_add(1);
  if (fill) simctl::polled_kernels.fetch_add(1);
  sim_enqueue(stream, [=]() mutable {
    const uint32_t* qs = q + q_slot_offset;
    for (uint64_t s = step_lo; s < step_hi; s++) {
      watch.wait_for_step(s);
      const uint64_t n0 = s * CPIR_PLANAR_SLOTS_PER_TILE, n1 = n0 + CPIR_PLANAR_SLOTS_PER_TILE < N ? n0 + CPIR_PLANAR_SLOTS_PER_TILE : N;
      if (!watch.gave_up) touch(qs, nullptr, n0, n1);  // (a launch that gave up reads whatever is there: its results are void)
      mac_range(dtc, C, qs, nullptr, n0, n1, r_prezeroed);
    }
    (void)q_len;
  });
  return CPIR_OK;
}

int launch_respond_read_rows_in_place(const Device*, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* const* q_rows, uint32_t batch, uint64_t q_len,
                                      uint64_t q_slot_offset, uint32_t* r, hipStream_t stream, const PlanarHostFill* fill, bool r_prezeroed) {
  if (L.packing != CPIR_PACK_PLANAR || batch == 0 || batch > CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS) return CPIR_ERR_INVALID_ARGUMENT;
  if (fill && (!fill->progress || !fill->abort_flag || fill->seats == 0 || fill->seats > batch)) return CPIR_ERR_INVALID_ARGUMENT;
  const uint32_t* rows[CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS] = {nullptr, nullptr, nullptr, nullptr};
  for (uint32_t i = 0; i < batch; i++) {
    if (!q_rows[i]) return CPIR_ERR_INVALID_ARGUMENT;
    rows[i] = q_rows[i];  // (taken at launch time: the table belongs to the caller)
  }
  const uint64_t N = L.num_slots, steps = (N + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE;
  const uint32_t C = L.num_cols;
  FillWatch watch(fill);
  // (only the first fill->seats seats are being copied in; the seats of a round are numbered in the order the callers took them, polled or not)
  for (uint32_t i = 0; i < batch; i++) watch.rows[i] = rows[i] + q_slot_offset;
  watch.words = N;
  simctl::kernels_launched.fetch_add(1);
  if (fill) simctl::polled_kernels.fetch_add(1);
  const uint32_t *r0 = rows[0], *r1 = rows[1], *r2 = rows[2], *r3 = rows[3];
  sim_enqueue(stream, [=]() mutable {
    const uint32_t* rw[4] = {r0, r1, r2, r3};
    if (!r_prezeroed) memset(r, 0, (size_t)batch * C * 4);
    for (uint64_t s = 0; s < steps; s++) {
      watch.wait_for_step(s);
      const uint64_t n0 = s * CPIR_PLANAR_SLOTS_PER_TILE, n1 = n0 + CPIR_PLANAR_SLOTS_PER_TILE < N ? n0 + CPIR_PLANAR_SLOTS_PER_TILE : N;
      for (uint32_t i = 0; i < batch; i++) {
        if (!watch.gave_up) touch(rw[i] + q_slot_offset, nullptr, n0, n1);
        mac_range(dtc, C, rw[i] + q_slot_offset, nullptr, n0, n1, r + (uint64_t)i * C);
      }
    }
    (void)q_len;
  });
  return CPIR_OK;
}

int launch_gather_query(const Device*, const uint32_t* q, uint64_t q_len, uint64_t q_slot_offset, const SlotMap& map, uint32_t batch, uint32_t* out,
                        hipStream_t stream) {
  const uint32_t* keep = map.keep_dev;
  const uint64_t n_kept = map.n_kept, n_pad = map.n_pad;
  simctl::kernels_launched.fetch_add(1);
  sim_enqueue(stream, [=] {
    for (uint32_t i = 0; i < batch; i++)
      for (uint64_t j = 0; j < n_kept; j++) out[(uint64_t)i * n_pad + j] = q[(uint64_t)i * q_len + q_slot_offset + keep[j]];
  });
  return CPIR_OK;
}

}  // namespace cpir
