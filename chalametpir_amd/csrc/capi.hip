// capi.hip -- the extern "C" boundary of libchalamet_hip.so (include/chalamet_hip.h): device context, the device-resident
// Server handle, and the host-side orchestration of Server::setup / Server::respond
// (reference chalametpir_server/src/server.rs:47-78, 103-167, 184-190).
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <list>
#include <memory>
#include <thread>

#include "cpir_internal.hpp"

namespace cpir {

// ---------------------------------------------------------------------------------------------------------------
// error text
// ---------------------------------------------------------------------------------------------------------------
static thread_local char t_last_hip_error[512] = "";

void set_last_hip_error(hipError_t e, const char* what, const char* file, int line) {
  snprintf(t_last_hip_error, sizeof(t_last_hip_error), "%s (%d) from `%s` at %s:%d", hipGetErrorString(e), (int)e, what, file, line);
  (void)hipGetLastError();  // clear the sticky per-thread error so later calls report their own failures
}

// ---------------------------------------------------------------------------------------------------------------
// Server handle
// ---------------------------------------------------------------------------------------------------------------
// Host callers of respond(&self) are COALESCED and PIPELINED (row f3 behind the thread-safe ABI; the reference serves an Arc<Server>
// from many tokio tasks, examples/server.rs:45,55,85).  What a query costs on the host path is its upload (4.7 MB at 2^20 keys) more
// than its kernel, so the front end is built around the host link:
//   * a caller takes a seat in the OPEN arena (opening a free one if need be; the first one in is the arena's leader), copies its query
//     into the arena's pinned block -- unless it already lies in page-locked memory -- and enqueues the upload on ONE upload stream
//     shared by all arenas: queries cross the link one after the other, whole, in the order they were staged, so the first seats of an
//     arena are in HBM early instead of every concurrent upload finishing at the same late moment;
//   * the leader keeps its arena open until the device is free of the previous arena's launch (or the arena is full) and every seat
//     taken so far is staged, then closes it and enqueues ONE batched respond for those seats on the run stream, behind the seats' upload
//     events; callers that arrive later open the next arena and upload while this kernel runs;
//   * a lone caller finds everything idle and is served without an upload at all: the step-major kernel reads every query word exactly
//     once, so it reads them IN PLACE over the host link -- from the caller's buffer when that is page-locked, else from the arena's
//     pinned block, which the caller's thread and the staging helpers fill in two halves, each half's steps launched as soon as it is
//     in place.  (respond.host_zero_copy=0: upload first, as concurrent callers do.)
struct RespondArena {
  uint32_t* q_dev = nullptr;     // kSeats x total_slots u32
  uint32_t* r_dev = nullptr;     // kSeats x C u32
  uint32_t* q_pinned = nullptr;  // kSeats x total_slots u32
  uint32_t* r_pinned = nullptr;  // kSeats x C u32
  std::vector<hipEvent_t> seat_ev;  // upload of seat i has crossed the link
  hipEvent_t done_ev = nullptr;     // the arena's responses are in r_pinned
  const uint32_t* q_pinned_dev = nullptr;  // q_pinned as the device addresses it (a lone query is read in place)
  uint32_t* fill_progress = nullptr;       // in the pinned block: steps of a lone query copied so far (the kernel polls it)
  const uint32_t* fill_progress_dev = nullptr;
  bool r0_zero = false;                    // seat 0 of r_dev holds zeros (guarded by the arena's own leader: one at a time)
  // guarded by Server::mu
  enum State { FREE, OPEN, LAUNCHED, DONE } state = FREE;
  uint32_t joined = 0;  // seats taken
  uint32_t staged = 0;  // seats whose upload is enqueued
  uint32_t left = 0;    // seats whose caller has taken its response
  int status = CPIR_OK; // outcome of the launch (shared by every seat)
};

struct Server {
  std::atomic<int> refs{1};
  Device* dev = nullptr;
  cpir_dtc_layout layout{};
  uint32_t* dtc = nullptr;  // device, layout.total_words u32
  uint64_t slot_offset = 0;
  uint64_t total_slots = 0;
  double setup_timings[CPIR_SETUP_TIMING_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};

  // 3 arenas of 8 seats: a batch of up to 8 rides ONE stream of the database on the matrix cores (a respond kernel takes about as
  // long for 8 queries as for 1, so throughput is batch size over kernel time); one arena is on the device, one is filling, one spare
  static constexpr uint32_t kSeats = 8;
  static constexpr uint32_t kArenas = 4;
  // An arena takes its first kSpread callers freely; further callers prefer to open another arena (so that one arena's uploads overlap
  // another's kernel: 8 concurrent callers split 4 + 4 instead of convoying) and fill seats kSpread.. only once no arena is free.
  static constexpr uint32_t kSpread = 4;
  // CPIR_RESPOND_TRACE=1: per-phase wall time of the host path, printed when the server is destroyed (diagnosis)
  struct Trace {
    std::atomic<uint64_t> calls{0}, solo{0}, ns_solo{0}, batches{0}, ns_seat{0}, ns_stage{0}, ns_gate{0}, ns_enqueue{0}, ns_gpu{0}, ns_follow{0}, ns_out{0};
    std::atomic<uint64_t> batch_hist[9] = {};
  } trace;
  bool trace_on = false;
  std::mutex mu;
  std::condition_variable cv;
  RespondArena arena[kArenas];  // each allocated on first use (a lone caller only ever needs the first)
  bool streams_ready = false;
  hipStream_t up_stream = nullptr;   // every query upload, FIFO
  hipStream_t run_stream = nullptr;  // every batched respond + response download, FIFO
  std::atomic<uint32_t> fill_aborts{0};  // lone queries whose polled launch gave up waiting for the copy (3: stop polling)
  std::atomic<uint64_t> fill_polled{0};  // lone pageable queries answered by one launch polling the copy's progress
  std::mutex upload_mu;              // one query's upload is enqueued at a time (whole queries, not interleaved pieces)
  std::mutex launch_mu;              // one arena's launch sequence is enqueued at a time

  // ---- group handle (cpir_server_setup_multi): the database is split along the filter slots over several devices of this
  // process; `shards` then holds one ordinary server per device and this handle owns no packed database itself.  A host query is
  // SCATTERED: device g receives only q[n_g : n_{g+1}] over its own host link, answers its shard, and the C-word partial
  // responses are summed on the host (u32 wrap-around: order-independent, bit-identical to one device).
  std::vector<Server*> shards;
  struct GroupLane {  // per shard, per call context
    hipStream_t stream = nullptr;
    uint32_t *q_dev = nullptr, *r_dev = nullptr, *q_pinned = nullptr, *r_pinned = nullptr;
  };
  struct GroupCtx {
    bool busy = false;
    std::vector<GroupLane> lanes;
  };
  static constexpr int kGroupCtx = 4;  // concurrent callers served at once; further callers wait
  GroupCtx gctx[kGroupCtx];
  bool gctx_ready = false;
  // one persistent host thread per shard does that shard's staging, enqueues and wait, so the per-device host work of a
  // query (a few tens of microseconds each) runs side by side instead of adding up over the devices
  struct GroupDone {  // on the caller's stack
    std::mutex mu;
    std::condition_variable cv;
    size_t remaining = 0;
    int status = CPIR_OK;
  };
  struct GroupJob {
    const uint32_t* q = nullptr;
    GroupCtx* ctx = nullptr;
    GroupDone* done = nullptr;
  };
  struct GroupWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<GroupJob> jobs;
    bool stop = false;
  };
  std::vector<std::unique_ptr<GroupWorker>> workers;
};

static double now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A lone caller's query is copied into the pinned block by several threads: one core copies ~20 GB/s, the host link takes ~57, and the
// reference's own benchmark is exactly a single caller handing over a pageable buffer (integrations/benches/online_phase.rs:81-97).
// Three helper threads per process, created on first use; a caller that finds them busy (many concurrent callers: their own threads
// already copy side by side) simply copies by itself.
class StagingHelpers {
 public:
  static constexpr int kHelpers = 3;
  struct Job {
    void* dst;
    const void* src;
    size_t bytes;
    std::atomic<int>* done;  // set to 1 when copied
  };
  ~StagingHelpers() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread& t : threads_)
      if (t.joinable()) t.join();
  }
  // exclusive use for one query; false if somebody else holds the helpers
  bool try_acquire() {
    if (!owner_.try_lock()) return false;
    std::lock_guard<std::mutex> lk(mu_);
    if (threads_.empty())
      for (int i = 0; i < kHelpers; i++) threads_.emplace_back([this] { run(); });
    return true;
  }
  void release() { owner_.unlock(); }
  void submit(const Job& j) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      jobs_.push_back(j);
    }
    cv_.notify_one();
  }
  // the submitter helps: run one queued job, if any
  bool help() {
    Job j;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (jobs_.empty()) return false;
      j = jobs_.front();
      jobs_.pop_front();
    }
    memcpy(j.dst, j.src, j.bytes);
    j.done->store(1, std::memory_order_release);
    return true;
  }

 private:
  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
        if (jobs_.empty()) return;
        j = jobs_.front();
        jobs_.pop_front();
      }
      memcpy(j.dst, j.src, j.bytes);
      j.done->store(1, std::memory_order_release);
    }
  }
  std::mutex owner_, mu_;
  std::condition_variable cv_;
  std::deque<Job> jobs_;
  std::vector<std::thread> threads_;
  bool stop_ = false;
};
static StagingHelpers g_staging;

// Wait for an event the device will signal within a few hundred microseconds: poll it for a while (a blocking wait costs tens of
// microseconds of wake-up latency, a tenth of a lone query), then fall back to the blocking wait.
static hipError_t wait_for_event(hipEvent_t ev) {
  const double t0 = now_seconds();
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if (now_seconds() - t0 > 2e-3) break;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  (void)hipGetLastError();  // hipErrorNotReady is sticky-free, but keep the thread's error state clean
  return hipEventSynchronize(ev);
}

static void device_retain(Device* d) { d->refs.fetch_add(1); }
static void device_release(Device* d) {
  if (d && d->refs.fetch_sub(1) == 1) {
    DeviceGuard g(d->ordinal);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
  }
}

// Where the device may read [p, p + bytes) of host memory in place (a DMA straight from the caller's buffer, or a kernel reading a lone
// query where it lies): the device address of p if BOTH ends of the range are page-locked memory the runtime has mapped, at the same distance from each other as on the host (one mapping, or mappings laid end to end); NULL
// otherwise (pageable memory, a registration that covers only part of the buffer).  Only a NO is remembered (per thread, for the last
// buffer asked about: a server loop hands over the same pageable buffer again and again, and a stale no merely stages a buffer that has
// been registered since); a yes is asked again every call, because a stale yes -- the buffer unregistered in between -- would let the
// device read unmapped host pages.
static const void* pinned_range_device_pointer(const void* p, size_t bytes) {
  struct Last {
    const char* lo = nullptr;
    size_t bytes = 0;
    const void* dev = nullptr;
  };
  static thread_local Last last;
  const char* c = static_cast<const char*>(p);
  if (last.lo == c && last.bytes == bytes && last.dev == nullptr) return nullptr;
  const void* dev = nullptr;
  hipPointerAttribute_t lo_attr, hi_attr;
  if (bytes > 0 && hipPointerGetAttributes(&lo_attr, c) == hipSuccess && hipPointerGetAttributes(&hi_attr, c + bytes - 1) == hipSuccess) {
    if (lo_attr.type == hipMemoryTypeHost && hi_attr.type == hipMemoryTypeHost && lo_attr.devicePointer && hi_attr.devicePointer &&
        static_cast<const char*>(hi_attr.devicePointer) - static_cast<const char*>(lo_attr.devicePointer) == (ptrdiff_t)(bytes - 1))
      dev = lo_attr.devicePointer;
  } else {
    (void)hipGetLastError();  // ordinary pageable memory: not an error worth keeping
  }
  last.lo = c, last.bytes = bytes, last.dev = dev;
  return dev;
}

// an arena's query and response seats live in ONE device block and ONE pinned block (pinning is the slow call)
static void arena_free(RespondArena& a) {
  for (hipEvent_t e : a.seat_ev)
    if (e) (void)hipEventDestroy(e);
  if (a.done_ev) (void)hipEventDestroy(a.done_ev);
  if (a.q_dev) (void)hipFree(a.q_dev);
  if (a.q_pinned) (void)hipHostFree(a.q_pinned);
  a = RespondArena{};
}

static void arenas_destroy(Server* srv) {
  for (RespondArena& a : srv->arena) arena_free(a);
  if (srv->up_stream) (void)hipStreamDestroy(srv->up_stream);
  if (srv->run_stream) (void)hipStreamDestroy(srv->run_stream);
  srv->up_stream = srv->run_stream = nullptr;
  srv->streams_ready = false;
}

// spare words behind the response seats of an arena's two blocks: the fill progress of a lone query in CPIR_FILL_LINES copies (pinned
// block; + 16 words so that the copies can start on a 64-byte line), the abort flag (device block, behind seat 0's response)
static constexpr size_t kArenaSpareWords = (size_t)CPIR_FILL_LINES * 16 + 16;
static void publish_fill_progress(uint32_t* lines, uint32_t steps) {
  for (uint32_t i = 0; i < CPIR_FILL_LINES; i++) __atomic_store_n(lines + i * 16, steps, __ATOMIC_RELEASE);
}

// on first use of this arena (caller holds Server::mu).  A shard stages only its own slots of a query, but seats keep the full stride.
static int arena_create(Server* srv, RespondArena& a) {
  // kSeats queries, then kSeats responses (query block first: it stays 16-byte aligned)
  // (+ 16 words behind the responses: the device block's spare words follow seat 0's response when a lone caller has only one seat
  // to fill -- the abort flag of a polled launch; the pinned block's hold the fill progress the kernel polls)
  const size_t qw = (size_t)srv->total_slots * Server::kSeats, rw = ((size_t)srv->layout.num_cols * Server::kSeats + 3) / 4 * 4 + kArenaSpareWords;
  auto fail = [&](hipError_t e, const char* what) {
    set_last_hip_error(e, what, __FILE__, __LINE__);
    arena_free(a);
    return e == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP;
  };
  hipError_t e = hipSuccess;
  if (!srv->streams_ready) {
    // The two streams must not share a hardware queue (uploads would then serialise with kernels): HIP multiplexes the streams of one
    // priority level over a handful of queues in creation order, and a host process (torch, say) has usually created several already.
    // The run stream is created at the highest priority, which has queues of its own.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // (least, greatest): numerically greatest <= least
    e = hipStreamCreateWithFlags(&srv->up_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&srv->run_stream, hipStreamNonBlocking, prio_hi);
    if (e != hipSuccess) {
      if (srv->up_stream) (void)hipStreamDestroy(srv->up_stream);
      srv->up_stream = srv->run_stream = nullptr;
      return fail(e, "hipStreamCreateWithFlags");
    }
    srv->streams_ready = true;
  }
  if ((e = hipMalloc(&a.q_dev, (qw + rw) * 4)) != hipSuccess) return fail(e, "hipMalloc(respond arena)");
  if ((e = hipHostMalloc(&a.q_pinned, (qw + rw) * 4, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc(respond arena)");
  a.r_dev = a.q_dev + qw, a.r_pinned = a.q_pinned + qw;
  {
    void* dp = nullptr;
    if ((e = hipHostGetDevicePointer(&dp, a.q_pinned, 0)) != hipSuccess) return fail(e, "hipHostGetDevicePointer");
    a.q_pinned_dev = static_cast<const uint32_t*>(dp);
    const size_t off = (qw + rw - kArenaSpareWords + 15) / 16 * 16;  // the copies start on a 64-byte line (the block itself is page-aligned)
    a.fill_progress = a.q_pinned + off;
    a.fill_progress_dev = a.q_pinned_dev + off;
  }
  a.r0_zero = false;
  a.seat_ev.assign(Server::kSeats, nullptr);
  for (hipEvent_t& ev : a.seat_ev)
    if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreateWithFlags");
  if ((e = hipEventCreateWithFlags(&a.done_ev, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreateWithFlags");
  return CPIR_OK;
}

static void server_destroy(Server* srv);

static void group_ctx_destroy(Server* srv) {
  for (auto& w : srv->workers) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->stop = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
  }
  srv->workers.clear();
  for (Server::GroupCtx& c : srv->gctx) {
    for (size_t g = 0; g < c.lanes.size(); g++) {
      Server::GroupLane& l = c.lanes[g];
      DeviceGuard dg(srv->shards[g]->dev->ordinal);
      if (l.stream) (void)hipStreamDestroy(l.stream);
      if (l.q_dev) (void)hipFree(l.q_dev);  // q_dev and r_dev are one block
      if (l.q_pinned) (void)hipHostFree(l.q_pinned);  // q_pinned and r_pinned are one block
    }
    c.lanes.clear();
  }
  srv->gctx_ready = false;
}

static void server_destroy(Server* srv) {
  if (!srv) return;
  if (srv->trace_on && srv->trace.calls.load()) {
    const Server::Trace& t = srv->trace;
    const double n = (double)t.calls.load(), nb = (double)(t.batches.load() ? t.batches.load() : 1);
    fprintf(stderr, "[cpir respond trace] %.0f calls in %.0f batches; us per call: seat wait %.1f, staging %.1f, copy out %.1f; followers wait %.1f; "
                    "us per batch (leader): gate %.1f, enqueue %.1f, device %.1f; batch sizes",
            n, nb, t.ns_seat.load() / n / 1e3, t.ns_stage.load() / n / 1e3, t.ns_out.load() / n / 1e3,
            t.ns_follow.load() / (n - nb > 0 ? n - nb : 1) / 1e3, t.ns_gate.load() / nb / 1e3, t.ns_enqueue.load() / nb / 1e3, t.ns_gpu.load() / nb / 1e3);
    for (int i = 1; i <= 8; i++) fprintf(stderr, " %d:%llu", i, (unsigned long long)t.batch_hist[i].load());
    fprintf(stderr, "; served alone (query read in place) %llu, %.1f us each; of those %llu by one launch polling the copy, %u such launches gave up\n",
            (unsigned long long)t.solo.load(), t.solo.load() ? t.ns_solo.load() / (double)t.solo.load() / 1e3 : 0.0,
            (unsigned long long)srv->fill_polled.load(), srv->fill_aborts.load());
  }
  if (!srv->shards.empty()) {
    group_ctx_destroy(srv);
    for (Server* c : srv->shards) server_destroy(c);
    srv->shards.clear();
  }
  {
    DeviceGuard g(srv->dev->ordinal);
    arenas_destroy(srv);
    if (srv->dtc) (void)hipFree(srv->dtc);
  }
  device_release(srv->dev);
  delete srv;
}

// one shard's part of a group respond: stage its slots of the query, upload, answer, download, wait
static int group_shard_respond(const Server* child, Server::GroupLane& l, const uint32_t* q, uint32_t C) {
  const size_t n = (size_t)child->layout.num_slots;
  hipError_t e;
  if (pinned_range_device_pointer(q + child->slot_offset, n * 4) != nullptr) {  // this shard's slots lie in page-locked memory: DMA from there
    e = hipMemcpyAsync(l.q_dev, q + child->slot_offset, n * 4, hipMemcpyHostToDevice, l.stream);
  } else {
    memcpy(l.q_pinned, q + child->slot_offset, n * 4);
    e = hipMemcpyAsync(l.q_dev, l.q_pinned, n * 4, hipMemcpyHostToDevice, l.stream);
  }
  int status = CPIR_OK;
  // a shard answered from ITS slice of the query is an unsharded respond on a database of its own slots
  if (e == hipSuccess) status = launch_respond(child->dev, child->dtc, child->layout, l.q_dev, n, 0, 1, 1, l.r_dev, nullptr, l.stream);
  if (e == hipSuccess && status == CPIR_OK) e = hipMemcpyAsync(l.r_pinned, l.r_dev, (size_t)C * 4, hipMemcpyDeviceToHost, l.stream);
  const hipError_t e2 = hipStreamSynchronize(l.stream);  // drain whatever was enqueued
  if (e == hipSuccess) e = e2;
  if (e != hipSuccess && status == CPIR_OK) {
    set_last_hip_error(e, "group respond (shard)", __FILE__, __LINE__);
    status = CPIR_ERR_HIP;
  }
  return status;
}

static void group_worker_main(Server* srv, size_t g) {
  Server::GroupWorker& w = *srv->workers[g];
  const Server* child = srv->shards[g];
  (void)hipSetDevice(child->dev->ordinal);  // this thread only ever talks to its shard's device
  const uint32_t C = srv->layout.num_cols;
  for (;;) {
    Server::GroupJob job;
    {
      std::unique_lock<std::mutex> lk(w.mu);
      w.cv.wait(lk, [&] { return w.stop || !w.jobs.empty(); });
      if (w.jobs.empty()) return;  // stop requested and nothing left
      job = w.jobs.front();
      w.jobs.pop_front();
    }
    const int st = group_shard_respond(child, job.ctx->lanes[g], job.q, C);
    {
      std::lock_guard<std::mutex> lk(job.done->mu);
      if (st != CPIR_OK && job.done->status == CPIR_OK) job.done->status = st;
      job.done->remaining--;
      job.done->cv.notify_one();  // under the lock: `done` lives on the caller's stack and may go away as soon as it is released
    }
  }
}

// per shard: a stream, a device block (query slice + response) and a pinned block of the same shape, for every call context
static int group_ctx_create(Server* srv) {
  const uint32_t C = srv->layout.num_cols;
  for (Server::GroupCtx& c : srv->gctx) {
    c.lanes.resize(srv->shards.size());
    for (size_t g = 0; g < srv->shards.size(); g++) {
      Server::GroupLane& l = c.lanes[g];
      const Server* child = srv->shards[g];
      DeviceGuard dg(child->dev->ordinal);
      const size_t qw = ((size_t)child->layout.num_slots + 3) / 4 * 4, words = qw + C;
#define TRY_(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_last_hip_error(_e, #e, __FILE__, __LINE__); group_ctx_destroy(srv); \
    return _e == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP; } } while (0)
      TRY_(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
      TRY_(hipMalloc(&l.q_dev, words * 4));
      TRY_(hipHostMalloc(&l.q_pinned, words * 4, hipHostMallocDefault));
#undef TRY_
      l.r_dev = l.q_dev + qw;
      l.r_pinned = l.q_pinned + qw;
    }
  }
  for (size_t g = 0; g < srv->shards.size(); g++) {
    srv->workers.emplace_back(new Server::GroupWorker);
    srv->workers.back()->th = std::thread(group_worker_main, srv, g);
  }
  srv->gctx_ready = true;
  return CPIR_OK;
}

// Server::respond on a group handle: scatter the query slices, one launch per device, sum the partial responses on the host
static int group_respond(Server* srv, const uint32_t* q, uint32_t* r_out) {
  const uint32_t C = srv->layout.num_cols;
  Server::GroupCtx* ctx = nullptr;
  {
    std::unique_lock<std::mutex> lk(srv->mu);
    if (!srv->gctx_ready) CPIR_TRY(group_ctx_create(srv));
    srv->cv.wait(lk, [&] {
      for (Server::GroupCtx& c : srv->gctx)
        if (!c.busy) {
          ctx = &c;
          return true;
        }
      return false;
    });
    ctx->busy = true;
  }
  struct Release {
    Server* srv;
    Server::GroupCtx* c;
    ~Release() {
      {
        std::lock_guard<std::mutex> lk(srv->mu);
        c->busy = false;
      }
      srv->cv.notify_all();
    }
  } rel{srv, ctx};
  Server::GroupDone done;
  done.remaining = srv->shards.size();
  for (auto& w : srv->workers) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->jobs.push_back(Server::GroupJob{q, ctx, &done});
    }
    w->cv.notify_one();
  }
  int status;
  {
    std::unique_lock<std::mutex> lk(done.mu);
    done.cv.wait(lk, [&] { return done.remaining == 0; });
    status = done.status;
  }
  if (status != CPIR_OK) return status;
  memcpy(r_out, ctx->lanes[0].r_pinned, (size_t)C * 4);
  for (size_t g = 1; g < srv->shards.size(); g++) {
    const uint32_t* p = ctx->lanes[g].r_pinned;
    for (uint32_t c = 0; c < C; c++) r_out[c] += p[c];  // u32 wrap-around
  }
  return CPIR_OK;
}

static Server* server_new(Device* dev, const cpir_dtc_layout& L, uint64_t slot_offset, uint64_t total_slots) {
  Server* s = new Server;
  s->dev = dev;
  device_retain(dev);
  s->layout = L;
  s->slot_offset = slot_offset;
  s->total_slots = total_slots;
  const char* tr = getenv("CPIR_RESPOND_TRACE");
  s->trace_on = tr && tr[0] == '1';
  return s;
}

// ---------------------------------------------------------------------------------------------------------------
// setup orchestration
// ---------------------------------------------------------------------------------------------------------------
// Setup's large temporaries (A in HBM: 8.4 GB at 2^20 keys, the unpacked D on host and device: 4.4 GB each, pinned staging) take
// ~0.35 s to release -- longer than the hint matmul.  They are handed to a background thread so that setup returns as soon as the
// server and the hint exist.  Threads still running when the library is unloaded are joined first.
class BackgroundDisposer {
 public:
  ~BackgroundDisposer() {
    std::lock_guard<std::mutex> lk(mu_);
    for (Job& j : jobs_)
      if (j.th.joinable()) j.th.join();
  }
  void run(std::function<void()> f) {
    std::lock_guard<std::mutex> lk(mu_);
    // reap what has finished since the last call: a long-lived process that rebuilds servers must not pile up joinable threads
    for (auto it = jobs_.begin(); it != jobs_.end();) {
      if (it->done->load(std::memory_order_acquire)) {
        it->th.join();
        it = jobs_.erase(it);
      } else {
        ++it;
      }
    }
    auto done = std::make_shared<std::atomic<bool>>(false);
    jobs_.push_back(Job{std::thread([f = std::move(f), done] {
                          f();
                          done->store(true, std::memory_order_release);
                        }),
                        done});
  }

 private:
  struct Job {
    std::thread th;
    std::shared_ptr<std::atomic<bool>> done;
  };
  std::mutex mu_;
  std::list<Job> jobs_;
};
static BackgroundDisposer g_disposer;

// ---------------------------------------------------------------------------------------------------------------
// Expands the public matrix A (1774 x N) on a host thread -- TurboSHAKE128 squeezed row block by row block into two
// pinned staging buffers -- and streams it into HBM on its own copy stream while the caller encodes / uploads / packs D.
// A stays resident (8.4 GB at 2^20 keys, 33 GB at 2^22: sized for 288 GB of HBM) so the hint is ONE matmul launch.
class PublicMatrixUpload {
 public:
  // One target per device: it keeps columns [col_lo, col_lo + col_n) of A (an N-shard; col_n = 0 means all N).  The sponge has
  // to be squeezed for every byte of A whatever is kept; with several targets ONE expansion feeds every device's slab.
  struct Target {
    Device* dev = nullptr;
    uint64_t col_lo = 0, col_n = 0;
    uint32_t* A_dev = nullptr;
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> block_ev;  // one per staging block (XOF mode) or one for the whole upload (caller-supplied A)
  };
  PublicMatrixUpload(Device* dev, uint64_t N, uint64_t col_lo = 0, uint64_t col_n = 0) : N_(N) { add_target(dev, col_lo, col_n); }
  explicit PublicMatrixUpload(uint64_t N) : N_(N) {}
  void add_target(Device* dev, uint64_t col_lo, uint64_t col_n) {
    Target t;
    t.dev = dev, t.col_lo = col_lo, t.col_n = col_n ? col_n : N_;
    device_retain(dev);  // this object may outlive the caller's handle: it is disposed of on a background thread after setup returns
    targets_.push_back(t);
  }
  ~PublicMatrixUpload() {
    cancel_.store(true, std::memory_order_relaxed);  // an early error return must not wait for the rest of the sponge
    join();
    for (Target& t : targets_) {
      {
        DeviceGuard g(t.dev->ordinal);
        for (hipEvent_t e : t.block_ev)
          if (e) (void)hipEventDestroy(e);
        if (t.copy_stream) (void)hipStreamDestroy(t.copy_stream);
        if (t.A_dev) (void)hipFree(t.A_dev);
      }
      device_release(t.dev);
    }
    for (int i = 0; i < 2; i++)
      if (pinned_[i]) (void)hipHostFree(pinned_[i]);
  }

  int start(const uint8_t seed[32], const uint32_t* A_host) {
    const uint64_t rows = CPIR_LWE_DIMENSION;
    // ~64 MiB staging blocks, whole rows (one block = everything when the caller supplies A)
    rows_per_block_ = A_host ? rows : (uint64_t)(64ull << 20) / (N_ * 4);
    if (rows_per_block_ < 1) rows_per_block_ = 1;
    if (rows_per_block_ > rows) rows_per_block_ = rows;
    const uint64_t nblocks = (rows + rows_per_block_ - 1) / rows_per_block_;
    for (Target& t : targets_) {
      DeviceGuard g(t.dev->ordinal);
      CPIR_HIP_TRY(hipMalloc(&t.A_dev, (size_t)rows * t.col_n * 4));
      CPIR_HIP_TRY(hipStreamCreateWithFlags(&t.copy_stream, hipStreamNonBlocking));
      t.block_ev.assign(nblocks, nullptr);
      for (hipEvent_t& e : t.block_ev) CPIR_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      if (A_host) {  // caller supplied A: plain upload, no XOF
        CPIR_HIP_TRY(hipMemcpy2DAsync(t.A_dev, t.col_n * 4, A_host + t.col_lo, N_ * 4, t.col_n * 4, rows, hipMemcpyHostToDevice, t.copy_stream));
        CPIR_HIP_TRY(hipEventRecord(t.block_ev[0], t.copy_stream));
      }
    }
    if (A_host) {
      std::lock_guard<std::mutex> lk(prog_mu_);
      rows_enqueued_ = rows, run_done_ = true;
      return CPIR_OK;
    }
    // portable: the same staging block is the source of copies to every target device
    for (int i = 0; i < 2; i++) CPIR_HIP_TRY(hipHostMalloc(&pinned_[i], (size_t)rows_per_block_ * N_ * 4, hipHostMallocPortable));
    memcpy(seed_, seed, 32);
    worker_ = std::thread([this] {
      const int st = run();
      std::lock_guard<std::mutex> lk(prog_mu_);
      status_ = st, run_done_ = true;
      prog_cv_.notify_all();
    });
    return CPIR_OK;
  }

  double xof_seconds() const { return xof_seconds_; }
  const uint32_t* device_ptr(size_t which = 0) const { return targets_[which].A_dev; }  // valid after start()

  // Block until the upload of rows [0, row_end) of A has been ENQUEUED on target `which`'s copy stream, then make `consumer` (a
  // stream of that device) wait for it: hint rows can be computed while the sponge is still being squeezed for the rows below.
  int wait_rows(uint64_t row_end, hipStream_t consumer, size_t which = 0) {
    {
      std::unique_lock<std::mutex> lk(prog_mu_);
      prog_cv_.wait(lk, [&] { return rows_enqueued_ >= row_end || run_done_; });
      if (rows_enqueued_ < row_end) return status_ != CPIR_OK ? status_ : CPIR_ERR_HIP;
    }
    Target& t = targets_[which];
    DeviceGuard g(t.dev->ordinal);
    CPIR_HIP_TRY(hipStreamWaitEvent(consumer, t.block_ev[(row_end - 1) / rows_per_block_], 0));
    return CPIR_OK;
  }

  // wait until all of A is in HBM (on every target); A_dev receives target `which`'s slab
  int finish(const uint32_t** A_dev, size_t which = 0) {
    join();
    if (status_ != CPIR_OK) return status_;
    for (Target& t : targets_) {
      DeviceGuard g(t.dev->ordinal);
      CPIR_HIP_TRY(hipStreamSynchronize(t.copy_stream));
    }
    *A_dev = targets_[which].A_dev;
    return CPIR_OK;
  }

 private:
  void join() {
    if (worker_.joinable()) worker_.join();
  }
  int run() {
    TurboShake128 xof;  // matrix.rs:542-544
    xof.absorb(seed_, 32);
    xof.finalize(0x1F);
    const uint64_t rows = CPIR_LWE_DIMENSION;
    int buf = 0;
    uint64_t blk = 0;
    for (uint64_t r0 = 0; r0 < rows; r0 += rows_per_block_, buf ^= 1, blk++) {
      const uint64_t rb = (rows - r0 < rows_per_block_) ? rows - r0 : rows_per_block_;
      if (cancel_.load(std::memory_order_relaxed)) return CPIR_ERR_INVALID_ARGUMENT;  // owner is being destroyed; nobody reads this
      if (blk >= 2)  // staging buffer free again on every device? (it was the source of block blk - 2)
        for (Target& t : targets_) {
          DeviceGuard g(t.dev->ordinal);
          CPIR_HIP_TRY(hipEventSynchronize(t.block_ev[blk - 2]));
        }
      const double t0 = now_seconds();
      xof.squeeze(reinterpret_cast<uint8_t*>(pinned_[buf]), (size_t)rb * N_ * 4);  // matrix.rs:546-555: row-major LE u32
      xof_seconds_ += now_seconds() - t0;
      for (Target& t : targets_) {
        DeviceGuard g(t.dev->ordinal);
        CPIR_HIP_TRY(hipMemcpy2DAsync(t.A_dev + r0 * t.col_n, t.col_n * 4, pinned_[buf] + t.col_lo, N_ * 4, t.col_n * 4, rb,
                                      hipMemcpyHostToDevice, t.copy_stream));
        CPIR_HIP_TRY(hipEventRecord(t.block_ev[blk], t.copy_stream));
      }
      {
        std::lock_guard<std::mutex> lk(prog_mu_);
        rows_enqueued_ = r0 + rb;
      }
      prog_cv_.notify_all();
    }
    return CPIR_OK;
  }

  uint64_t N_;
  std::vector<Target> targets_;
  uint32_t* pinned_[2] = {nullptr, nullptr};
  uint64_t rows_per_block_ = 0;
  uint8_t seed_[32];
  std::thread worker_;
  std::atomic<bool> cancel_{false};
  int status_ = CPIR_OK;
  double xof_seconds_ = 0;
  std::mutex prog_mu_;
  std::condition_variable prog_cv_;
  uint64_t rows_enqueued_ = 0;  // rows of A whose upload is on the copy streams
  bool run_done_ = false;
};

static void dispose_async(std::unique_ptr<PublicMatrixUpload> up) {
  PublicMatrixUpload* raw = up.release();
  if (raw) g_disposer.run([raw] { delete raw; });
}

struct DevBuf {  // scoped device allocation
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  // free in the background (on device `ordinal`) instead of at scope exit
  void dispose_async(int ordinal) {
    void* q = p;
    p = nullptr;
    if (q)
      g_disposer.run([q, ordinal] {
        DeviceGuard g(ordinal);
        (void)hipFree(q);
      });
  }
};

// The matrix half of setup once D sits on the host: upload D, pack it, wait for A, one matmul, hint back.
static int setup_from_host_matrix(Device* dev, PublicMatrixUpload& upA, const uint32_t* D, uint64_t N, uint32_t C, uint32_t b,
                                  uint32_t* hint_out, Server** out) {
  cpir_dtc_layout L;
  CPIR_TRY(dtc_layout_for(N, C, b, &L));
  DeviceGuard g(dev->ordinal);
  hipStream_t stream = dev->stream;
  DevBuf D_dev, flag, M_dev;
  CPIR_HIP_TRY(hipMalloc(&D_dev.p, (size_t)N * C * 4));
  CPIR_HIP_TRY(hipMalloc(&flag.p, 4));
  CPIR_HIP_TRY(hipMalloc(&M_dev.p, (size_t)CPIR_LWE_DIMENSION * C * 4));
  Server* srv = server_new(dev, L, 0, N);
  auto fail = [&](int st) { server_destroy(srv); return st; };
#define TRY_(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_last_hip_error(_e, #e, __FILE__, __LINE__); \
    return fail(_e == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP); } } while (0)
  TRY_(hipMalloc(&srv->dtc, (size_t)L.total_words * 4));
  double t0 = now_seconds();
  TRY_(hipMemcpyAsync(D_dev.p, D, (size_t)N * C * 4, hipMemcpyHostToDevice, stream));
  TRY_(hipStreamSynchronize(stream));
  srv->setup_timings[2] = now_seconds() - t0;
  t0 = now_seconds();
  TRY_(hipMemsetAsync(flag.p, 0, 4, stream));
  // The hint matmul takes its right-hand side from the packed image where it can (planar packing with at least one bit plane, entries
  // below 2^b -- checked below): the low-byte operand pieces are the image's own, the pack kernel writes the high-byte pieces next to it
  // in the same pass over D.  (Otherwise D is split into byte planes in a pass of its own, or multiplied on the VALU.)
  const uint32_t* A_dev = upA.device_ptr();
  DevBuf hi_plane, rowsum_ws;
  const uint64_t hi_bytes = planar_hi_plane_bytes(L);
  bool planar_rhs = mfma_matmul_enabled() && hi_bytes && mfma_planar_rhs_applicable(A_dev, N, L);
  if (planar_rhs) {
    TRY_(hipMalloc(&hi_plane.p, (size_t)hi_bytes));
    TRY_(hipMalloc(&rowsum_ws.p, 4 * 128));
  }
  int st = launch_transpose_compress(dev, (const uint32_t*)D_dev.p, C, L, srv->dtc, (uint32_t*)flag.p, stream, hi_plane.p);
  if (st != CPIR_OK) return fail(st);
  uint32_t ored = 0;
  TRY_(hipMemcpyAsync(&ored, flag.p, 4, hipMemcpyDeviceToHost, stream));
  TRY_(hipStreamSynchronize(stream));
  srv->setup_timings[3] = now_seconds() - t0;
  // the hint uses the UNMASKED entries of D (server.rs:61 multiplies before any masking); the packed-16 kernel is
  // exact only if every entry is < 2^16, which holds for every encoded DB (entries < 2^b <= 2^14) and is verified here
  const uint32_t rhs_bits = (ored >> 16) ? 32u : 16u;
  // The hint in row chunks, each launched as soon as its rows of A are on their way to HBM: hint rows [r, r + 128) need only
  // those rows of A, so all but the last chunk's matmul hides behind the sponge.
  const uint64_t chunk = 128;
  // the image holds the fields masked to b bits (matrix.rs:121), the hint wants D as it is (server.rs:61): the same thing only if no entry
  // reaches 2^b
  if (planar_rhs && (ored >> b) != 0) {
    planar_rhs = false;
    hi_plane.dispose_async(dev->ordinal);
  }
  // matrix-core path without a usable image: D is turned into its operand form ONCE and every chunk multiplies against that
  DevBuf rhs;
  const bool mfma = !planar_rhs && mfma_matmul_enabled() && mfma_matmul_applicable(A_dev, N, N, C, rhs_bits);
  if (mfma) {
    TRY_(hipMalloc(&rhs.p, (size_t)mfma_rhs_workspace_bytes(N, C, chunk)));
    st = launch_rhs_split(dev, (const uint32_t*)D_dev.p, C, N, C, rhs.p, stream);
    if (st != CPIR_OK) return fail(st);
  }
  double t_wait = 0, t_last = now_seconds();
  for (uint64_t r0 = 0; r0 < CPIR_LWE_DIMENSION; r0 += chunk) {
    const uint64_t rb = (CPIR_LWE_DIMENSION - r0 < chunk) ? CPIR_LWE_DIMENSION - r0 : chunk;
    t0 = now_seconds();
    st = upA.wait_rows(r0 + rb, stream);
    if (st != CPIR_OK) return fail(st);
    t_last = now_seconds();
    t_wait += t_last - t0;
    if (planar_rhs)
      st = launch_mat_x_mat_mfma_planar(dev, A_dev + r0 * N, N, srv->dtc, L, hi_plane.p, (uint32_t*)rowsum_ws.p, (uint32_t*)M_dev.p + r0 * C, C, rb,
                                        0, stream);
    else if (mfma) st = launch_mat_x_mat_mfma(dev, A_dev + r0 * N, N, rhs.p, N, C, (uint32_t*)M_dev.p + r0 * C, C, rb, chunk, 0, stream);
    else st = launch_mat_x_mat(dev, A_dev + r0 * N, N, (const uint32_t*)D_dev.p, C, (uint32_t*)M_dev.p + r0 * C, C, rb, N, C, rhs_bits, 0, stream);
    if (st != CPIR_OK) return fail(st);
  }
  TRY_(hipStreamSynchronize(stream));
  srv->setup_timings[4] = t_wait;                   // waiting for rows of A (the sponge)
  srv->setup_timings[5] = now_seconds() - t_last;   // what is left of the hint matmul once the last rows of A are there
  srv->setup_timings[1] = upA.xof_seconds();
  t0 = now_seconds();
  TRY_(hipMemcpyAsync(hint_out, M_dev.p, (size_t)CPIR_LWE_DIMENSION * C * 4, hipMemcpyDeviceToHost, stream));
  TRY_(hipStreamSynchronize(stream));
  srv->setup_timings[6] = now_seconds() - t0;
#undef TRY_
  D_dev.dispose_async(dev->ordinal);  // 4*N*C bytes
  rhs.dispose_async(dev->ordinal);
  hi_plane.dispose_async(dev->ordinal);
  *out = srv;
  return CPIR_OK;
}

// slots per shard are multiples of this: no packed word of either layout and no 16-byte query piece straddles two shards
static uint64_t shard_unit(const cpir_dtc_layout& L) {
  uint64_t a = L.slots_per_chunk, b = L.compression_factor, x = a, y = b;
  while (y) {
    const uint64_t t = x % y;
    x = y, y = t;
  }
  return a / x * b;
}

// [lo, hi) of shard g of `shards` (same rule as chalametpir_amd.distributed.shard_range); the last shard takes the ragged tail
static void shard_bounds(uint64_t N, uint64_t unit, size_t g, size_t shards, uint64_t* lo, uint64_t* hi) {
  const uint64_t units = (N + unit - 1) / unit;
  const uint64_t a = units * g / shards * unit, b = units * (g + 1) / shards * unit;
  *lo = a < N ? a : N;
  *hi = b < N ? b : N;
}

// how many of `n_dev` devices get a (non-empty) shard
static size_t group_size(uint64_t N, uint64_t unit, size_t n_dev) {
  const uint64_t units = (N + unit - 1) / unit;
  return units < n_dev ? (size_t)units : n_dev;
}

// The matrix half of setup for a group: every device uploads and packs its rows of D, multiplies its column slab of A (ONE
// host expansion feeds all slabs: upA has one target per shard) by them, and the partial hints are summed on the host.
static int setup_group_from_host_matrix(const std::vector<Device*>& devs, PublicMatrixUpload& upA, const uint32_t* D, uint64_t N,
                                        uint32_t C, uint32_t b, uint32_t* hint_out, Server** out) {
  cpir_dtc_layout Lfull;
  CPIR_TRY(dtc_layout_for(N, C, b, &Lfull));
  const uint64_t unit = shard_unit(Lfull);
  const size_t G = devs.size();
  Server* grp = server_new(devs[0], Lfull, 0, N);
  struct Work {
    DevBuf D_dev, flag, M_dev, hi_plane, rowsum_ws;  // hi_plane: the second operand plane of the hint matmul, written by the pack pass
    uint32_t ored = 0;
  };
  std::vector<Work> work(G);
  auto fail = [&](int st) { server_destroy(grp); return st; };
#define TRY_(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_last_hip_error(_e, #e, __FILE__, __LINE__); \
    return fail(_e == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP); } } while (0)
  double t0 = now_seconds();
  for (size_t g = 0; g < G; g++) {  // enqueue on every device first: uploads and packs of different devices overlap
    uint64_t lo, hi;
    shard_bounds(N, unit, g, G, &lo, &hi);
    cpir_dtc_layout L;
    int st = dtc_layout_for(hi - lo, C, b, &L);
    if (st != CPIR_OK) return fail(st);
    DeviceGuard dg(devs[g]->ordinal);
    Server* child = server_new(devs[g], L, lo, N);
    grp->shards.push_back(child);
    TRY_(hipMalloc(&child->dtc, (size_t)L.total_words * 4));
    TRY_(hipMalloc(&work[g].D_dev.p, (size_t)(hi - lo) * C * 4));
    TRY_(hipMalloc(&work[g].flag.p, 4));
    TRY_(hipMalloc(&work[g].M_dev.p, (size_t)CPIR_LWE_DIMENSION * C * 4));
    hipStream_t stream = devs[g]->stream;
    TRY_(hipMemcpyAsync(work[g].D_dev.p, D + lo * C, (size_t)(hi - lo) * C * 4, hipMemcpyHostToDevice, stream));
    TRY_(hipMemsetAsync(work[g].flag.p, 0, 4, stream));
    // (as setup_from_host_matrix: where the packed image can serve as the matmul's right-hand side, the pack pass prepares it; A's slab
    // for this shard is allocated 16-byte aligned with leading dimension hi - lo)
    const uint64_t hi_bytes = planar_hi_plane_bytes(L);
    if (mfma_matmul_enabled() && hi_bytes && L.packing == CPIR_PACK_PLANAR && (hi - lo) % 4 == 0 && mfma_pipeline() != 0) {
      TRY_(hipMalloc(&work[g].hi_plane.p, (size_t)hi_bytes));
      TRY_(hipMalloc(&work[g].rowsum_ws.p, 4 * ((CPIR_LWE_DIMENSION + 127) / 128 * 128)));
    }
    st = launch_transpose_compress(devs[g], (const uint32_t*)work[g].D_dev.p, C, L, child->dtc, (uint32_t*)work[g].flag.p, stream,
                                   work[g].hi_plane.p);
    if (st != CPIR_OK) return fail(st);
    TRY_(hipMemcpyAsync(&work[g].ored, work[g].flag.p, 4, hipMemcpyDeviceToHost, stream));
  }
  uint32_t ored = 0;
  for (size_t g = 0; g < G; g++) {
    DeviceGuard dg(devs[g]->ordinal);
    TRY_(hipStreamSynchronize(devs[g]->stream));
    ored |= work[g].ored;
  }
  grp->setup_timings[2] = now_seconds() - t0;  // D upload + pack, all devices
  const uint32_t rhs_bits = (ored >> 16) ? 32u : 16u;  // as setup_from_host_matrix
  t0 = now_seconds();
  const uint32_t* A_dev0 = nullptr;
  int st = upA.finish(&A_dev0);
  if (st != CPIR_OK) return fail(st);
  grp->setup_timings[4] = now_seconds() - t0;
  grp->setup_timings[1] = upA.xof_seconds();
  t0 = now_seconds();
  const size_t hint_words = (size_t)CPIR_LWE_DIMENSION * C;
  std::vector<std::vector<uint32_t>> partial(G > 1 ? G - 1 : 0);
  for (size_t g = 0; g < G; g++) {
    const Server* child = grp->shards[g];
    DeviceGuard dg(devs[g]->ordinal);
    const uint32_t* A_dev = nullptr;
    st = upA.finish(&A_dev, g);
    if (st != CPIR_OK) return fail(st);
    const uint64_t n = child->layout.num_slots;
    if (work[g].hi_plane.p && (ored >> b) == 0 && mfma_planar_rhs_applicable(A_dev, n, child->layout))
      st = launch_mat_x_mat_mfma_planar(devs[g], A_dev, n, child->dtc, child->layout, work[g].hi_plane.p, (uint32_t*)work[g].rowsum_ws.p,
                                        (uint32_t*)work[g].M_dev.p, C, CPIR_LWE_DIMENSION, 0, devs[g]->stream);
    else
      st = launch_mat_x_mat(devs[g], A_dev, n, (const uint32_t*)work[g].D_dev.p, C, (uint32_t*)work[g].M_dev.p, C, CPIR_LWE_DIMENSION, n, C,
                            rhs_bits, 0, devs[g]->stream);
    if (st != CPIR_OK) return fail(st);
    uint32_t* dst = hint_out;
    if (g > 0) {
      partial[g - 1].resize(hint_words);
      dst = partial[g - 1].data();
    }
    TRY_(hipMemcpyAsync(dst, work[g].M_dev.p, hint_words * 4, hipMemcpyDeviceToHost, devs[g]->stream));
  }
  for (size_t g = 0; g < G; g++) {
    DeviceGuard dg(devs[g]->ordinal);
    TRY_(hipStreamSynchronize(devs[g]->stream));
  }
  for (size_t g = 1; g < G; g++) {  // hint = sum of the per-shard partial products (u32 wrap-around)
    const uint32_t* p = partial[g - 1].data();
    for (size_t i = 0; i < hint_words; i++) hint_out[i] += p[i];
  }
  grp->setup_timings[5] = now_seconds() - t0;  // partial matmuls + downloads + host sum
#undef TRY_
  for (size_t g = 0; g < G; g++) {
    work[g].D_dev.dispose_async(devs[g]->ordinal);
    work[g].hi_plane.dispose_async(devs[g]->ordinal);
  }
  *out = grp;
  return CPIR_OK;
}

static bool has_device(int* count) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_last_hip_error(e, "hipGetDeviceCount", __FILE__, __LINE__);
    n = 0;
  }
  if (count) *count = n;
  return n > 0;
}

}  // namespace cpir

using namespace cpir;

struct cpir_device : Device {};
struct cpir_server : Server {};
struct cpir_xof : TurboShake128 {};

extern "C" {

// ---------------------------------------------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------------------------------------------
const char* cpir_strerror(int status) {
  switch (status) {  // texts follow the Display impl of ChalametPIRError (reference chalametpir_common/src/error.rs:51-100)
    case CPIR_OK: return "ok";
    case CPIR_ERR_INVALID_MATRIX_DIMENSION: return "The number of rows and columns in the matrix must be non-zero.";
    case CPIR_ERR_INCOMPATIBLE_DIM_MATMUL: return "The matrix dimensions do not allow multiplication.";
    case CPIR_ERR_INVALID_NUMBER_OF_ELEMENTS: return "The matrix must have \"rows * columns\" elements.";
    case CPIR_ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED:
      return "The dimensions are incompatible for multiplication of a row vector and a transposed matrix.";
    case CPIR_ERR_FAILED_TO_DESERIALIZE_MATRIX: return "Matrix deserialization failed";
    case CPIR_ERR_EMPTY_KV_DATABASE: return "Cannot encode empty key-value database.";
    case CPIR_ERR_EXHAUSTED_ATTEMPTS_3WISE: return "Exhausted all attempts to build 3-wise XOR binary fuse filter.";
    case CPIR_ERR_EXHAUSTED_ATTEMPTS_4WISE: return "Exhausted all attempts to build 4-wise XOR binary fuse filter.";
    case CPIR_ERR_KV_DATABASE_SIZE_TOO_LARGE: return "The key-value database is too large; it can have a maximum of 2^42 entries.";
    case CPIR_ERR_UNSUPPORTED_ARITY: return "Binary Fuse Filter supports arity of either 3 or 4.";
    case CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH: return "Encoded database matrix's element bit length mustn't ever exceed 16.";
    case CPIR_ERR_NO_DEVICE: return "No usable HIP device (there is no CPU fallback).";
    case CPIR_ERR_HIP: return "A HIP runtime call failed; see cpir_last_hip_error().";
    case CPIR_ERR_OUT_OF_DEVICE_MEMORY: return "Failed to allocate device or pinned host memory.";
    case CPIR_ERR_BUFFER_TOO_SMALL: return "Caller-provided output buffer is too small.";
    case CPIR_ERR_INVALID_ARGUMENT: return "Invalid argument.";
    case CPIR_ERR_SHARD_RANGE: return "Shard boundaries are not aligned to the packing unit or exceed the query length.";
    default: return "unknown status";
  }
}

const char* cpir_last_hip_error(void) { return t_last_hip_error; }
const char* cpir_version(void) { return "chalamet_hip 0.1.0 (gfx950)"; }
const char* cpir_xof_permutation(void) { return xof_permutation_name(); }

int cpir_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return CPIR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!has_device(nullptr)) return CPIR_ERR_NO_DEVICE;
  CPIR_HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable));
  return CPIR_OK;
}

void cpir_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

// ---------------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------------
int cpir_device_count(int* count) {
  if (!count) return CPIR_ERR_INVALID_ARGUMENT;
  return has_device(count) ? CPIR_OK : CPIR_ERR_NO_DEVICE;
}

int cpir_device_open(int ordinal, cpir_device** out) {
  if (!out) return CPIR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n = 0;
  if (!has_device(&n)) return CPIR_ERR_NO_DEVICE;
  if (ordinal < 0 || ordinal >= n) return CPIR_ERR_NO_DEVICE;
  DeviceGuard g(ordinal);
  if (!g.ok) return CPIR_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  CPIR_HIP_TRY(hipGetDeviceProperties(&prop, ordinal));
  cpir_device* d = new cpir_device;
  d->ordinal = ordinal;
  d->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
  const hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_last_hip_error(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
    delete d;
    return CPIR_ERR_HIP;
  }
  *out = d;
  return CPIR_OK;
}

void cpir_device_close(cpir_device* dev) { device_release(dev); }

int cpir_device_ordinal(const cpir_device* dev, int* ordinal) {
  if (!dev || !ordinal) return CPIR_ERR_INVALID_ARGUMENT;
  *ordinal = dev->ordinal;
  return CPIR_OK;
}

int cpir_device_synchronize(cpir_device* dev) {
  if (!dev) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  CPIR_HIP_TRY(hipDeviceSynchronize());
  return CPIR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// shapes
// ---------------------------------------------------------------------------------------------------------------
uint32_t cpir_compression_factor(uint32_t b) { return compression_factor(b); }
int cpir_find_encoded_db_matrix_element_bit_length(uint64_t n, uint32_t* b) { return find_bit_len(n, b); }
int cpir_filter_shape(uint32_t arity, uint64_t n, uint32_t* sl, uint32_t* scl, uint64_t* nf) { return filter_shape(arity, n, sl, scl, nf); }
uint64_t cpir_encoded_num_cols(uint64_t max_value_byte_len, uint32_t b) { return encoded_num_cols(max_value_byte_len, b); }
int cpir_dtc_layout_for(uint64_t N, uint32_t C, uint32_t b, cpir_dtc_layout* out) { return dtc_layout_for(N, C, b, out); }
int cpir_dtc_layout_for_packing(uint64_t N, uint32_t C, uint32_t b, uint32_t packing, cpir_dtc_layout* out) {
  return dtc_layout_for_packing(N, C, b, packing, out);
}
uint64_t cpir_shard_unit(const cpir_dtc_layout* layout) {
  return (layout && layout->slots_per_chunk && layout->compression_factor) ? shard_unit(*layout) : 0;
}

int cpir_generate_from_seed(uint64_t rows, uint64_t cols, const uint8_t seed[CPIR_SEED_BYTE_LEN], uint32_t* out) {
  if (!seed || !out) return CPIR_ERR_INVALID_ARGUMENT;
  if (rows == 0 || cols == 0) return CPIR_ERR_INVALID_MATRIX_DIMENSION;  // Matrix::from_values, matrix.rs:69-79
  TurboShake128 xof;
  xof.absorb(seed, CPIR_SEED_BYTE_LEN);
  xof.finalize(0x1F);
  xof.squeeze(reinterpret_cast<uint8_t*>(out), (size_t)(rows * cols) * 4);
  return CPIR_OK;
}

int cpir_xof_open(const uint8_t seed[CPIR_SEED_BYTE_LEN], cpir_xof** out) {
  if (!seed || !out) return CPIR_ERR_INVALID_ARGUMENT;
  cpir_xof* x = new cpir_xof;
  x->absorb(seed, CPIR_SEED_BYTE_LEN);  // matrix.rs:542-544
  x->finalize(0x1F);
  *out = x;
  return CPIR_OK;
}

int cpir_xof_squeeze(cpir_xof* xof, void* out, size_t bytes) {
  if (!xof || (!out && bytes)) return CPIR_ERR_INVALID_ARGUMENT;
  xof->squeeze(static_cast<uint8_t*>(out), bytes);
  return CPIR_OK;
}

void cpir_xof_close(cpir_xof* xof) { delete xof; }

// ---------------------------------------------------------------------------------------------------------------
// low-level device operations
// ---------------------------------------------------------------------------------------------------------------
int cpir_op_mat_x_mat(cpir_device* dev, const uint32_t* A, uint64_t lda, const uint32_t* D, uint64_t ldd, uint32_t* M, uint64_t ldm,
                      uint64_t rows, uint64_t inner, uint64_t cols, uint32_t rhs_max_bits, int accumulate, void* stream) {
  if (!dev) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  return launch_mat_x_mat(dev, A, lda, D, ldd, M, ldm, rows, inner, cols, rhs_max_bits, accumulate, pick_stream(dev, stream));
}

const char* cpir_mat_x_mat_kernel_name(uint32_t rhs_max_bits) { return mat_x_mat_kernel_name(rhs_max_bits); }

int cpir_op_transpose_compress(cpir_device* dev, const uint32_t* D, uint64_t ldd, const cpir_dtc_layout* layout, uint32_t* dtc,
                               uint32_t* or_of_entries, void* stream) {
  if (!dev || !layout) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  return launch_transpose_compress(dev, D, ldd, *layout, dtc, or_of_entries, pick_stream(dev, stream));
}

uint64_t cpir_packed_rhs_plane_bytes(const cpir_dtc_layout* layout) {
  if (!layout || check_layout(*layout) != CPIR_OK) return 0;
  return planar_hi_plane_bytes(*layout);
}

int cpir_op_transpose_compress_with_plane(cpir_device* dev, const uint32_t* D, uint64_t ldd, const cpir_dtc_layout* layout, uint32_t* dtc,
                                          uint32_t* or_of_entries, void* hi_plane, void* stream) {
  if (!dev || !layout) return CPIR_ERR_INVALID_ARGUMENT;
  if (hi_plane && planar_hi_plane_bytes(*layout) == 0) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  return launch_transpose_compress(dev, D, ldd, *layout, dtc, or_of_entries, pick_stream(dev, stream), hi_plane);
}

int cpir_op_mat_x_packed(cpir_device* dev, const uint32_t* A, uint64_t lda, const uint32_t* dtc, const cpir_dtc_layout* layout,
                         const void* hi_plane, uint32_t* M, uint64_t ldm, uint64_t rows, int accumulate, void* stream) {
  if (!dev || !layout || !A || !dtc || !hi_plane || !M || rows == 0) return CPIR_ERR_INVALID_ARGUMENT;
  CPIR_TRY(check_layout(*layout));
  if (!mfma_matmul_enabled() || !mfma_planar_rhs_applicable(A, lda, *layout)) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  hipStream_t s = pick_stream(dev, stream);
  void* rowsum = nullptr;  // stream-ordered scratch: the row sums of A (a correction term of the signed-byte split)
  CPIR_HIP_TRY(hipMallocAsync(&rowsum, 4 * ((rows + 127) / 128 * 128), s));
  const int st = launch_mat_x_mat_mfma_planar(dev, A, lda, dtc, *layout, hi_plane, static_cast<uint32_t*>(rowsum), M, ldm, rows, accumulate, s);
  const hipError_t e = hipFreeAsync(rowsum, s);
  if (st != CPIR_OK) return st;
  CPIR_HIP_TRY(e);
  return CPIR_OK;
}

int cpir_op_dtc_import(cpir_device* dev, const uint32_t* compressed, const cpir_dtc_layout* layout, uint32_t* dtc, void* stream) {
  if (!dev || !layout) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  return launch_dtc_import(dev, compressed, *layout, dtc, pick_stream(dev, stream));
}

int cpir_op_dtc_export(cpir_device* dev, const uint32_t* dtc, const cpir_dtc_layout* layout, uint32_t* compressed, void* stream) {
  if (!dev || !layout) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  return launch_dtc_export(dev, dtc, *layout, compressed, pick_stream(dev, stream));
}

uint64_t cpir_respond_scratch_words(const cpir_dtc_layout* layout) { return layout ? respond_scratch_words(*layout, 1) : 0; }
uint64_t cpir_respond_batch_scratch_words(const cpir_dtc_layout* layout, uint32_t batch) {
  return layout ? respond_scratch_words(*layout, batch) : 0;
}

int cpir_op_respond(cpir_device* dev, const uint32_t* dtc, const cpir_dtc_layout* layout, const uint32_t* q, uint64_t q_len,
                    uint64_t q_slot_offset, uint32_t* r, uint32_t* scratch, void* stream) {
  if (!dev || !layout) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  return launch_respond(dev, dtc, *layout, q, q_len, q_slot_offset, 1, 1, r, scratch, pick_stream(dev, stream));
}

// Any batch size.  With batch fusion every pass answers 4 queries from one stream of the database (remainder 2 / 1);
// without it every query is its own pass.  Either way the passes of one kind go into ONE launch.
static int respond_batched(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                           uint64_t q_slot_offset, uint32_t batch, uint32_t* r, uint32_t* scratch, hipStream_t stream) {
  if (!respond_batch_fusion()) {
    // One launch for all passes saves a kernel fill/drain (~10 us) per query, but blocks of a long multi-pass launch drift
    // apart and lose the L2 sharing of q: measured on MI355X it wins up to 1.3 GB per pass (196 vs 204 us) and loses at
    // 5 GB and above (806 vs 770 us), so very large databases get one launch per query.
    // (the matrix-core kernel keeps one launch at every size: 1 440 vs 1 505 us per query at 9.8 GB, 790 vs 799 at 5 GB)
    if (L.packing == CPIR_PACK_PLANAR || L.total_words * 4 <= respond_multi_pass_limit_bytes()) return launch_respond(dev, dtc, L, q, q_len, q_slot_offset, 1, batch, r, scratch, stream);
    for (uint32_t i = 0; i < batch; i++)
      CPIR_TRY(launch_respond(dev, dtc, L, q + (uint64_t)i * q_len, q_len, q_slot_offset, 1, 1, r + (uint64_t)i * L.num_cols, scratch, stream));
    return CPIR_OK;
  }
  uint32_t done = 0;
  if (L.packing == CPIR_PACK_PLANAR) {
    // the matrix-core kernel takes any 1..8 queries per pass: passes of 8, then one pass for the rest
    const uint32_t W8 = CPIR_PLANAR_MAX_QUERIES_PER_PASS;
    if (batch >= W8) {
      CPIR_TRY(launch_respond(dev, dtc, L, q, q_len, q_slot_offset, W8, batch / W8, r, scratch, stream));
      done = batch / W8 * W8;
    }
    if (done < batch)
      CPIR_TRY(launch_respond(dev, dtc, L, q + (uint64_t)done * q_len, q_len, q_slot_offset, batch - done, 1,
                              r + (uint64_t)done * L.num_cols, scratch, stream));
    return CPIR_OK;
  }
  for (uint32_t width : {4u, 2u, 1u}) {
    const uint32_t passes = (batch - done) / width;
    if (passes == 0) continue;
    CPIR_TRY(launch_respond(dev, dtc, L, q + (uint64_t)done * q_len, q_len, q_slot_offset, width, passes,
                            r + (uint64_t)done * L.num_cols, scratch, stream));
    done += passes * width;
  }
  return CPIR_OK;
}

int cpir_op_respond_batch(cpir_device* dev, const uint32_t* dtc, const cpir_dtc_layout* layout, const uint32_t* q, uint64_t q_len,
                          uint64_t q_slot_offset, uint32_t batch, uint32_t* r, uint32_t* scratch, void* stream) {
  if (!dev || !layout || batch == 0) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  return respond_batched(dev, dtc, *layout, q, q_len, q_slot_offset, batch, r, scratch, pick_stream(dev, stream));
}

int cpir_op_synth_fill(cpir_device* dev, uint32_t* out, uint64_t count, uint64_t seed, uint64_t index0, uint32_t mask, void* stream) {
  if (!dev) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  return launch_synth_fill(dev, out, count, seed, index0, mask, pick_stream(dev, stream));
}

const char* cpir_respond_kernel_name(const cpir_dtc_layout* layout) { return layout ? respond_kernel_name(*layout) : ""; }

// ---------------------------------------------------------------------------------------------------------------
// server: construction
// ---------------------------------------------------------------------------------------------------------------
int cpir_server_setup(cpir_device* dev, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN], const uint32_t* pub_mat_a, const uint32_t* D,
                      uint64_t N, uint32_t C, uint32_t b, uint32_t* hint_out, cpir_server** out) {
  if (!dev || !D || !hint_out || !out || (!seed_mu && !pub_mat_a)) return CPIR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (N == 0 || C == 0) return CPIR_ERR_INVALID_MATRIX_DIMENSION;
  if (compression_factor(b) == 0) return CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH;  // matrix.rs:99-101
  const double t_begin = now_seconds();
  auto upA_owner = std::make_unique<PublicMatrixUpload>(dev, N);
  PublicMatrixUpload& upA = *upA_owner;
  static const uint8_t zero_seed[32] = {0};
  CPIR_TRY(upA.start(seed_mu ? seed_mu : zero_seed, pub_mat_a));  // server.rs:59 (runs concurrently with the D work)
  Server* srv = nullptr;
  CPIR_TRY(setup_from_host_matrix(dev, upA, D, N, C, b, hint_out, &srv));
  dispose_async(std::move(upA_owner));  // A leaves HBM in the background
  srv->setup_timings[7] = now_seconds() - t_begin;
  *out = static_cast<cpir_server*>(srv);
  return CPIR_OK;
}

// devices -> the Device list a group really uses (no empty shards), with the upload targets of A registered in shard order
static int group_plan(cpir_device* const* devs, uint32_t n_dev, uint64_t N, uint32_t C, uint32_t b, std::vector<Device*>* use,
                      PublicMatrixUpload* upA) {
  if (!devs || n_dev == 0) return CPIR_ERR_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < n_dev; i++)
    if (!devs[i]) return CPIR_ERR_INVALID_ARGUMENT;
  cpir_dtc_layout L;
  CPIR_TRY(dtc_layout_for(N, C, b, &L));
  const uint64_t unit = shard_unit(L);
  const size_t G = group_size(N, unit, n_dev);
  for (size_t g = 0; g < G; g++) {
    uint64_t lo, hi;
    shard_bounds(N, unit, g, G, &lo, &hi);
    use->push_back(devs[g]);
    upA->add_target(devs[g], lo, hi - lo);
  }
  return CPIR_OK;
}

int cpir_server_setup_multi(cpir_device* const* devs, uint32_t n_dev, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN], const uint32_t* pub_mat_a,
                            const uint32_t* D, uint64_t N, uint32_t C, uint32_t b, uint32_t* hint_out, cpir_server** out) {
  if (!devs || n_dev == 0 || !D || !hint_out || !out || (!seed_mu && !pub_mat_a)) return CPIR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (N == 0 || C == 0) return CPIR_ERR_INVALID_MATRIX_DIMENSION;
  if (compression_factor(b) == 0) return CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH;
  const double t_begin = now_seconds();
  auto upA_owner = std::make_unique<PublicMatrixUpload>(N);
  PublicMatrixUpload& upA = *upA_owner;
  std::vector<Device*> use;
  CPIR_TRY(group_plan(devs, n_dev, N, C, b, &use, &upA));
  static const uint8_t zero_seed[32] = {0};
  CPIR_TRY(upA.start(seed_mu ? seed_mu : zero_seed, pub_mat_a));
  Server* srv = nullptr;
  CPIR_TRY(setup_group_from_host_matrix(use, upA, D, N, C, b, hint_out, &srv));
  dispose_async(std::move(upA_owner));
  srv->setup_timings[7] = now_seconds() - t_begin;
  *out = static_cast<cpir_server*>(srv);
  return CPIR_OK;
}

int cpir_setup_kv_shape(uint32_t arity, const cpir_kv_db* db, uint32_t* b_out, uint64_t* N, uint32_t* C, size_t* hint_bytes_len) {
  if (!db) return CPIR_ERR_INVALID_ARGUMENT;
  if (arity != 3 && arity != 4) return CPIR_ERR_UNSUPPORTED_ARITY;
  if (db->num_pairs == 0) return CPIR_ERR_EMPTY_KV_DATABASE;  // server.rs:48-51
  if (!db->val_off) return CPIR_ERR_INVALID_ARGUMENT;
  uint32_t b = 0;
  CPIR_TRY(find_bit_len(db->num_pairs, &b));  // server.rs:53
  uint64_t nf = 0;
  CPIR_TRY(filter_shape(arity, db->num_pairs, nullptr, nullptr, &nf));
  uint64_t max_len = 0;
  for (uint64_t i = 0; i < db->num_pairs; i++) {
    const uint64_t l = db->val_off[i + 1] - db->val_off[i];
    if (l > max_len) max_len = l;
  }
  const uint64_t cols = encoded_num_cols(max_len, b);
  if (cols == 0 || cols > 0xffffffffull) return CPIR_ERR_INVALID_MATRIX_DIMENSION;
  if (b_out) *b_out = b;
  if (N) *N = nf;
  if (C) *C = (uint32_t)cols;
  if (hint_bytes_len) *hint_bytes_len = 8 + (size_t)CPIR_LWE_DIMENSION * cols * 4;
  return CPIR_OK;
}

int cpir_encode_kv_database(uint32_t arity, const cpir_kv_db* db, uint32_t b, const uint8_t* filter_seed_material, uint32_t max_attempts,
                            uint8_t filter_param_bytes_out[CPIR_FILTER_PARAM_BYTE_LEN], uint32_t* D_out, uint64_t D_cap_words, uint64_t* N,
                            uint32_t* C) {
  if (!db || !filter_param_bytes_out || !D_out || !N || !C) return CPIR_ERR_INVALID_ARGUMENT;
  if (max_attempts == 0) max_attempts = 100;
  Filter filter;
  std::vector<uint32_t> D;
  CPIR_TRY(encode_kv_database(arity, *db, b, filter_seed_material, max_attempts, &filter, &D, N, C));
  if (D_cap_words < D.size()) return CPIR_ERR_BUFFER_TOO_SMALL;
  memcpy(D_out, D.data(), D.size() * 4);
  filter.to_bytes(filter_param_bytes_out);
  return CPIR_OK;
}

// Full Server::setup on one device (n_dev == 1, devs[0]) or on a group of devices
static int setup_kv_common(cpir_device* const* devs, uint32_t n_dev, bool group, uint32_t arity, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN],
                           const cpir_kv_db* db, const uint8_t* filter_seed_material, uint32_t max_attempts, uint8_t* hint_bytes_out,
                           size_t hint_bytes_cap, size_t* hint_bytes_len, uint8_t filter_param_bytes_out[CPIR_FILTER_PARAM_BYTE_LEN],
                           cpir_server** out) {
  if (!devs || n_dev == 0 || !devs[0] || !seed_mu || !db || !hint_bytes_out || !hint_bytes_len || !filter_param_bytes_out || !out)
    return CPIR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  uint32_t b = 0, C = 0;
  uint64_t N = 0;
  size_t need = 0;
  CPIR_TRY(cpir_setup_kv_shape(arity, db, &b, &N, &C, &need));
  if (hint_bytes_cap < need) return CPIR_ERR_BUFFER_TOO_SMALL;
  if (reinterpret_cast<uintptr_t>(hint_bytes_out) % 4 != 0) return CPIR_ERR_INVALID_ARGUMENT;
  if (max_attempts == 0) max_attempts = 100;  // SERVER_SETUP_MAX_ATTEMPT_COUNT, params.rs:10

  // N is known from the key count alone, so the (sequential, seconds-long) XOF expansion of A starts right away and
  // overlaps the (also sequential) filter construction and row encoding below
  const double t_begin = now_seconds();
  auto upA_owner = std::make_unique<PublicMatrixUpload>(N);
  PublicMatrixUpload& upA = *upA_owner;
  std::vector<Device*> use;
  if (group) {
    CPIR_TRY(group_plan(devs, n_dev, N, C, b, &use, &upA));
  } else {
    upA.add_target(devs[0], 0, N);
  }
  CPIR_TRY(upA.start(seed_mu, nullptr));

  Filter filter;
  std::vector<uint32_t> D;
  uint64_t N2 = 0;
  uint32_t C2 = 0;
  CPIR_TRY(encode_kv_database(arity, *db, b, filter_seed_material, max_attempts, &filter, &D, &N2, &C2));  // server.rs:54
  if (N2 != N || C2 != C) return CPIR_ERR_INVALID_ARGUMENT;
  const double t_encode = now_seconds() - t_begin;

  Server* srv = nullptr;
  // hint_bytes = Matrix::to_bytes(hint): [rows][cols][elems] (matrix.rs:947-971, server.rs:62)
  uint32_t* hint = reinterpret_cast<uint32_t*>(hint_bytes_out + 8);
  if (group) CPIR_TRY(setup_group_from_host_matrix(use, upA, D.data(), N, C, b, hint, &srv));
  else CPIR_TRY(setup_from_host_matrix(devs[0], upA, D.data(), N, C, b, hint, &srv));
  const uint32_t hr = CPIR_LWE_DIMENSION, hc = C;
  memcpy(hint_bytes_out, &hr, 4);
  memcpy(hint_bytes_out + 4, &hc, 4);
  *hint_bytes_len = need;
  filter.to_bytes(filter_param_bytes_out);  // server.rs:63
  dispose_async(std::move(upA_owner));  // A leaves HBM, and the unpacked D (4*N*C bytes of host memory) is unmapped, in the background
  {
    auto* dv = new std::vector<uint32_t>(std::move(D));
    g_disposer.run([dv] { delete dv; });
  }
  srv->setup_timings[0] = t_encode;
  srv->setup_timings[7] = now_seconds() - t_begin;
  *out = static_cast<cpir_server*>(srv);
  return CPIR_OK;
}

int cpir_server_setup_kv(cpir_device* dev, uint32_t arity, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN], const cpir_kv_db* db,
                         const uint8_t* filter_seed_material, uint32_t max_attempts, uint8_t* hint_bytes_out, size_t hint_bytes_cap,
                         size_t* hint_bytes_len, uint8_t filter_param_bytes_out[CPIR_FILTER_PARAM_BYTE_LEN], cpir_server** out) {
  if (!dev) return CPIR_ERR_INVALID_ARGUMENT;
  cpir_device* one[1] = {dev};
  return setup_kv_common(one, 1, false, arity, seed_mu, db, filter_seed_material, max_attempts, hint_bytes_out, hint_bytes_cap,
                         hint_bytes_len, filter_param_bytes_out, out);
}

int cpir_server_setup_kv_multi(cpir_device* const* devs, uint32_t n_dev, uint32_t arity, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN],
                               const cpir_kv_db* db, const uint8_t* filter_seed_material, uint32_t max_attempts, uint8_t* hint_bytes_out,
                               size_t hint_bytes_cap, size_t* hint_bytes_len, uint8_t filter_param_bytes_out[CPIR_FILTER_PARAM_BYTE_LEN],
                               cpir_server** out) {
  if (devs)
    for (uint32_t i = 0; i < n_dev; i++)
      if (!devs[i]) return CPIR_ERR_INVALID_ARGUMENT;
  return setup_kv_common(devs, n_dev, true, arity, seed_mu, db, filter_seed_material, max_attempts, hint_bytes_out, hint_bytes_cap,
                         hint_bytes_len, filter_param_bytes_out, out);
}

int cpir_hint_partial_device(cpir_device* dev, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN], const uint32_t* pub_mat_a, const uint32_t* D_dev,
                             uint64_t ldd, uint64_t slot_offset, uint64_t N_shard, uint64_t total_slots, uint32_t C, uint32_t rhs_max_bits,
                             uint32_t* M_dev, void* stream) {
  if (!dev || !D_dev || !M_dev || (!seed_mu && !pub_mat_a)) return CPIR_ERR_INVALID_ARGUMENT;
  if (N_shard == 0 || C == 0 || total_slots == 0) return CPIR_ERR_INVALID_MATRIX_DIMENSION;
  if (slot_offset + N_shard > total_slots) return CPIR_ERR_SHARD_RANGE;
  PublicMatrixUpload upA(dev, total_slots, slot_offset, N_shard);
  static const uint8_t zero_seed[32] = {0};
  CPIR_TRY(upA.start(seed_mu ? seed_mu : zero_seed, pub_mat_a));
  const uint32_t* A_dev = nullptr;
  CPIR_TRY(upA.finish(&A_dev));
  DeviceGuard g(dev->ordinal);
  hipStream_t s = pick_stream(dev, stream);
  CPIR_TRY(launch_mat_x_mat(dev, A_dev, N_shard, D_dev, ldd, M_dev, C, CPIR_LWE_DIMENSION, N_shard, C, rhs_max_bits, 0, s));
  CPIR_HIP_TRY(hipStreamSynchronize(s));  // A_dev dies with upA
  return CPIR_OK;
}

int cpir_server_from_device_matrix(cpir_device* dev, const uint32_t* D_dev, uint64_t ldd, uint64_t N_shard, uint32_t C, uint32_t b,
                                   uint64_t slot_offset, uint64_t total_slots, void* stream, cpir_server** out) {
  if (!dev || !D_dev || !out) return CPIR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  cpir_dtc_layout L;
  CPIR_TRY(dtc_layout_for(N_shard, C, b, &L));
  if (slot_offset + N_shard > total_slots) return CPIR_ERR_SHARD_RANGE;
  DeviceGuard g(dev->ordinal);
  Server* srv = server_new(dev, L, slot_offset, total_slots);
  hipError_t e = hipMalloc(&srv->dtc, (size_t)L.total_words * 4);
  if (e != hipSuccess) {
    set_last_hip_error(e, "hipMalloc(dtc)", __FILE__, __LINE__);
    server_destroy(srv);
    return CPIR_ERR_OUT_OF_DEVICE_MEMORY;
  }
  hipStream_t s = pick_stream(dev, stream);
  int st = launch_transpose_compress(dev, D_dev, ldd, L, srv->dtc, nullptr, s);
  if (st == CPIR_OK) {
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) set_last_hip_error(e, "hipStreamSynchronize", __FILE__, __LINE__), st = CPIR_ERR_HIP;
  }
  if (st != CPIR_OK) {
    server_destroy(srv);
    return st;
  }
  *out = static_cast<cpir_server*>(srv);
  return CPIR_OK;
}

int cpir_server_from_compressed(cpir_device* dev, const uint32_t* compressed, uint32_t C, uint64_t N, uint32_t b, cpir_server** out) {
  if (!dev || !compressed || !out) return CPIR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  cpir_dtc_layout L;
  CPIR_TRY(dtc_layout_for(N, C, b, &L));
  DeviceGuard g(dev->ordinal);
  DevBuf src;
  const size_t src_bytes = (size_t)C * L.words_per_row * 4;
  CPIR_HIP_TRY(hipMalloc(&src.p, src_bytes));
  Server* srv = server_new(dev, L, 0, N);
  auto fail = [&](int st) { server_destroy(srv); return st; };
  hipError_t e = hipMalloc(&srv->dtc, (size_t)L.total_words * 4);
  if (e != hipSuccess) { set_last_hip_error(e, "hipMalloc(dtc)", __FILE__, __LINE__); return fail(CPIR_ERR_OUT_OF_DEVICE_MEMORY); }
  e = hipMemcpyAsync(src.p, compressed, src_bytes, hipMemcpyHostToDevice, dev->stream);
  if (e != hipSuccess) { set_last_hip_error(e, "hipMemcpyAsync", __FILE__, __LINE__); return fail(CPIR_ERR_HIP); }
  int st = launch_dtc_import(dev, (const uint32_t*)src.p, L, srv->dtc, dev->stream);
  if (st != CPIR_OK) return fail(st);
  e = hipStreamSynchronize(dev->stream);
  if (e != hipSuccess) { set_last_hip_error(e, "hipStreamSynchronize", __FILE__, __LINE__); return fail(CPIR_ERR_HIP); }
  *out = static_cast<cpir_server*>(srv);
  return CPIR_OK;
}

int cpir_server_export_compressed(const cpir_server* srv, uint32_t* compressed_out, uint64_t out_words) {
  if (!srv || !compressed_out) return CPIR_ERR_INVALID_ARGUMENT;
  const cpir_dtc_layout& L = srv->layout;
  const uint64_t words = (uint64_t)L.num_cols * L.words_per_row;
  if (out_words < words) return CPIR_ERR_BUFFER_TOO_SMALL;
  if (!srv->shards.empty()) {
    // a group: every shard starts at a multiple of cf slots, so its compressed words are a column range of the whole matrix
    std::vector<uint32_t> part;
    for (const Server* c : srv->shards) {
      const cpir_dtc_layout& Lc = c->layout;
      part.resize((size_t)Lc.num_cols * Lc.words_per_row);
      CPIR_TRY(cpir_server_export_compressed(static_cast<const cpir_server*>(c), part.data(), part.size()));
      const uint64_t w0 = c->slot_offset / L.compression_factor;
      for (uint32_t r = 0; r < L.num_cols; r++)
        memcpy(compressed_out + (size_t)r * L.words_per_row + w0, part.data() + (size_t)r * Lc.words_per_row, (size_t)Lc.words_per_row * 4);
    }
    return CPIR_OK;
  }
  DeviceGuard g(srv->dev->ordinal);
  DevBuf tmp;
  CPIR_HIP_TRY(hipMalloc(&tmp.p, (size_t)words * 4));
  CPIR_TRY(launch_dtc_export(srv->dev, srv->dtc, L, (uint32_t*)tmp.p, srv->dev->stream));
  CPIR_HIP_TRY(hipMemcpyAsync(compressed_out, tmp.p, (size_t)words * 4, hipMemcpyDeviceToHost, srv->dev->stream));
  CPIR_HIP_TRY(hipStreamSynchronize(srv->dev->stream));
  return CPIR_OK;
}

int cpir_server_setup_timings(const cpir_server* srv, double out[CPIR_SETUP_TIMING_COUNT]) {
  if (!srv || !out) return CPIR_ERR_INVALID_ARGUMENT;
  memcpy(out, srv->setup_timings, sizeof(srv->setup_timings));
  return CPIR_OK;
}

cpir_server* cpir_server_retain(cpir_server* srv) {
  if (srv) srv->refs.fetch_add(1);
  return srv;
}

void cpir_server_release(cpir_server* srv) {
  if (srv && srv->refs.fetch_sub(1) == 1) server_destroy(srv);
}

int cpir_server_layout(const cpir_server* srv, cpir_dtc_layout* out) {
  if (!srv || !out) return CPIR_ERR_INVALID_ARGUMENT;
  *out = srv->layout;
  return CPIR_OK;
}

int cpir_server_shard(const cpir_server* srv, uint64_t* slot_offset, uint64_t* total_slots) {
  if (!srv) return CPIR_ERR_INVALID_ARGUMENT;
  if (slot_offset) *slot_offset = srv->slot_offset;
  if (total_slots) *total_slots = srv->total_slots;
  return CPIR_OK;
}

const uint32_t* cpir_server_dtc_device_ptr(const cpir_server* srv) { return srv ? srv->dtc : nullptr; }

int cpir_server_group_size(const cpir_server* srv, uint32_t* shards) {
  if (!srv || !shards) return CPIR_ERR_INVALID_ARGUMENT;
  *shards = (uint32_t)srv->shards.size();
  return CPIR_OK;
}

int cpir_server_group_shard(const cpir_server* srv, uint32_t index, int* device_ordinal, uint64_t* slot_offset, uint64_t* num_slots) {
  if (!srv || index >= srv->shards.size()) return CPIR_ERR_INVALID_ARGUMENT;
  const Server* c = srv->shards[index];
  if (device_ordinal) *device_ordinal = c->dev->ordinal;
  if (slot_offset) *slot_offset = c->slot_offset;
  if (num_slots) *num_slots = c->layout.num_slots;
  return CPIR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// server: respond
// ---------------------------------------------------------------------------------------------------------------
// A caller that found the server idle (it holds arena `a` alone, closed to others): no upload.  The step-major kernel reads each query
// word once, so it reads them where they are: in the caller's buffer if that is page-locked and 16-byte aligned, else in the arena's
// pinned block, filled in two halves by this thread and the staging helpers with each half's steps launched as soon as it is in place
// (the second half is copied while the kernel works on the first).  The launches add up in r_dev, which is kept zeroed between uses.
static int respond_alone(Server* srv, RespondArena* a, const uint32_t* q, uint32_t* r_out) {
  const size_t C = srv->layout.num_cols;
  const size_t q_lo = (size_t)srv->slot_offset, words = (size_t)srv->layout.num_slots;
  hipStream_t st = srv->run_stream;
  std::lock_guard<std::mutex> ll(srv->launch_mu);
  hipError_t e = hipSuccess;
  int rc = CPIR_OK;
  // seat 0's response and the word behind it (the abort flag of a polled launch) are kept zeroed between uses
  if (!a->r0_zero) e = hipMemsetAsync(a->r_dev, 0, (C + 1) * 4, st);
  a->r0_zero = false;
  bool polled = false;
  // the slots this server reads, q[q_lo, q_lo + words), as the device addresses them -- if the whole range is page-locked; the kernel
  // is handed the (possibly virtual) address of q[0] and adds the offset itself
  const uint32_t* in_place = nullptr;
  if (e == hipSuccess && reinterpret_cast<uintptr_t>(q) % 16 == 0) {
    const void* dp = pinned_range_device_pointer(q + q_lo, words * 4);
    if (dp && reinterpret_cast<uintptr_t>(dp) % 16 == (q_lo * 4) % 16) in_place = static_cast<const uint32_t*>(dp) - q_lo;
  }
  if (e == hipSuccess && in_place) {
    rc = launch_respond_read_once(srv->dev, srv->dtc, srv->layout, in_place, srv->total_slots, srv->slot_offset, a->r_dev, st);
  } else if (e == hipSuccess) {
    uint32_t* const qp = a->q_pinned;  // seat 0; same offsets as the caller's buffer
    constexpr size_t kJob = (size_t)1 << 16;  // 256 KiB of u32, a multiple of the kernel's 512-slot step
    constexpr size_t kStepsPerJob = kJob / CPIR_PLANAR_SLOTS_PER_TILE;
    constexpr size_t kMaxJobs = 512;
    const size_t n_jobs = (words + kJob - 1) / kJob;
    const uint32_t fill_timeout_us = respond_host_fill_timeout_us();
    if (words >= ((size_t)1 << 19) && n_jobs <= kMaxJobs && g_staging.try_acquire()) {
      std::atomic<int> done[kMaxJobs];
      // ONE launch, in front of the copy: the kernel takes the steps of q round-robin (front to back over the whole grid) and waits
      // for each step's words to be in place, which this thread announces job by job in *fill_progress; the copy (~55 us for 4.7 MB)
      // runs underneath the stream (~200 us).  A wave that has waited fill_timeout_us gives up and flags the launch as void: the query
      // is then answered again from the (by then complete) pinned block -- a launch that cannot start before this thread moves on
      // (synchronous launches under a debugger or a serialising profiler) costs that timeout once, and after three such launches the
      // server stops polling and launches each half of the query when it is in place.
      polled = fill_timeout_us > 0 && srv->fill_aborts.load(std::memory_order_relaxed) < 3;
      if (polled) {
        publish_fill_progress(a->fill_progress, 0u);
        const PlanarHostFill fill{a->fill_progress_dev, a->r_dev + C, fill_timeout_us};
        rc = launch_respond_read_once(srv->dev, srv->dtc, srv->layout, a->q_pinned_dev, srv->total_slots, srv->slot_offset, a->r_dev, st, 0, 0,
                                      &fill);
        if (rc != CPIR_OK) polled = false;  // nothing was launched
      }
      for (size_t i = 0; i < n_jobs; i++) {
        done[i].store(0, std::memory_order_relaxed);
        const size_t o = q_lo + i * kJob, n = (words - i * kJob < kJob) ? words - i * kJob : kJob;
        g_staging.submit(StagingHelpers::Job{qp + o, q + o, n * 4, &done[i]});
      }
      auto wait_for_job = [&](size_t i) {
        while (!done[i].load(std::memory_order_acquire))
          if (!g_staging.help()) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
          }
      };
      if (polled) {
        for (size_t i = 0; i < n_jobs; i++) {
          wait_for_job(i);
          publish_fill_progress(a->fill_progress, i + 1 == n_jobs ? 0xffffffffu : (uint32_t)((i + 1) * kStepsPerJob));
        }
      } else if (rc == CPIR_OK) {
        const size_t j_half = (n_jobs + 1) / 2;
        size_t next = 0;
        for (int h = 0; h < 2 && rc == CPIR_OK; h++) {
          for (; next < (h ? n_jobs : j_half); next++) wait_for_job(next);
          const uint64_t s_lo = h ? j_half * kStepsPerJob : 0;
          const uint64_t s_hi = h ? (words + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE : j_half * kStepsPerJob;
          if (s_hi > s_lo)
            rc = launch_respond_read_once(srv->dev, srv->dtc, srv->layout, a->q_pinned_dev, srv->total_slots, srv->slot_offset, a->r_dev, st,
                                          s_lo, s_hi);
        }
      }
      for (size_t i = 0; i < n_jobs; i++) wait_for_job(i);  // every job must have run before the stack array goes away, whatever happened
      g_staging.release();
    } else {
      memcpy(qp + q_lo, q + q_lo, words * 4);
      rc = launch_respond_read_once(srv->dev, srv->dtc, srv->layout, a->q_pinned_dev, srv->total_slots, srv->slot_offset, a->r_dev, st);
    }
  }
  for (int attempt = 0; attempt < 2; attempt++) {
    if (e == hipSuccess && rc == CPIR_OK) e = hipMemcpyAsync(a->r_pinned, a->r_dev, (C + 1) * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rc == CPIR_OK) e = hipEventRecord(a->done_ev, st);
    if (e == hipSuccess && rc == CPIR_OK) {
      // zeros for the next lone caller, off this one's critical path
      if (hipMemsetAsync(a->r_dev, 0, (C + 1) * 4, st) == hipSuccess) a->r0_zero = true;
      else (void)hipGetLastError();
      e = wait_for_event(a->done_ev);
    } else {
      (void)hipStreamSynchronize(st);  // whatever was enqueued reads the caller's buffer / the pinned block: drain before returning
    }
    if (!(polled && e == hipSuccess && rc == CPIR_OK && a->r_pinned[C] != 0)) break;
    // the polled launch gave up waiting: its results are void.  The pinned block is complete by now: answer from it, without polling.
    polled = false;
    srv->fill_aborts.fetch_add(1, std::memory_order_relaxed);
    if (!a->r0_zero) e = hipMemsetAsync(a->r_dev, 0, (C + 1) * 4, st);
    a->r0_zero = false;
    if (e == hipSuccess)
      rc = launch_respond_read_once(srv->dev, srv->dtc, srv->layout, a->q_pinned_dev, srv->total_slots, srv->slot_offset, a->r_dev, st);
  }
  if (polled) srv->fill_polled.fetch_add(1, std::memory_order_relaxed);
  if (rc == CPIR_OK && e != hipSuccess) {
    set_last_hip_error(e, "respond (query read in place)", __FILE__, __LINE__);
    rc = CPIR_ERR_HIP;
  }
  if (rc == CPIR_OK) memcpy(r_out, a->r_pinned, C * 4);
  return rc;
}

int cpir_server_respond(const cpir_server* csrv, const uint32_t* q, uint32_t q_rows, uint64_t q_cols, uint32_t* r_out) {
  if (!csrv || !q || !r_out) return CPIR_ERR_INVALID_ARGUMENT;
  Server* srv = const_cast<cpir_server*>(csrv);  // the pool is the only mutable state; it is internally locked
  // matrix.rs:329-331: the query must be a 1 x N row vector
  if (!(q_rows == 1 && q_cols == srv->total_slots)) return CPIR_ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED;
  if (!srv->shards.empty()) return group_respond(srv, q, r_out);
  DeviceGuard g(srv->dev->ordinal);
  const size_t N = (size_t)srv->total_slots, C = srv->layout.num_cols;

  // ---- take a seat ----------------------------------------------------------------------------------------------
  const bool tr = srv->trace_on;
  const double t_enter = tr ? now_seconds() : 0;
  const bool read_once_ok = respond_read_once_applicable(srv->layout);
  std::unique_lock<std::mutex> lk(srv->mu);
  RespondArena* a = nullptr;
  bool solo = false;
  for (;;) {
    for (RespondArena& x : srv->arena)  // 1. an open arena that is still spreading
      if (!a && x.state == RespondArena::OPEN && x.joined < Server::kSpread) a = &x;
    if (!a)
      for (RespondArena& x : srv->arena)  // 2. a free arena
        if (!a && x.state == RespondArena::FREE) {
          if (!x.q_dev) CPIR_TRY(arena_create(srv, x));
          a = &x, x.state = RespondArena::OPEN, x.status = CPIR_OK;
          // nobody else is filling an arena or on the device: this caller is served alone, its query read in place
          solo = read_once_ok;
          for (const RespondArena& y : srv->arena)
            if (&y != a && (y.state == RespondArena::OPEN || y.state == RespondArena::LAUNCHED)) solo = false;
          if (solo) x.state = RespondArena::LAUNCHED;  // closed at once: later callers open the next arena and upload meanwhile
        }
    if (!a)
      for (RespondArena& x : srv->arena)  // 3. no arena free: fill the open one up
        if (!a && x.state == RespondArena::OPEN && x.joined < Server::kSeats) a = &x;
    if (a) break;
    srv->cv.wait(lk);  // every arena is full or in flight
  }
  const uint32_t seat = a->joined++;
  const bool leader = (seat == 0);
  lk.unlock();
  const double t_seated = tr ? now_seconds() : 0;
  if (solo) {
    const int st = respond_alone(srv, a, q, r_out);
    if (tr) {
      srv->trace.calls++, srv->trace.solo++;
      srv->trace.ns_seat += (uint64_t)((t_seated - t_enter) * 1e9), srv->trace.ns_solo += (uint64_t)((now_seconds() - t_seated) * 1e9);
    }
    lk.lock();
    a->state = RespondArena::FREE;
    a->joined = a->staged = a->left = 0;
    srv->cv.notify_all();
    return st;
  }

  // ---- stage the query (the reference copies too: from_bytes .to_vec(), matrix.rs:1001-1007) and enqueue its upload -----------------
  // (a shard reads only its own slots of the query: only those are staged and uploaded)
  hipError_t up = hipSuccess;
  const size_t q_lo = (size_t)srv->slot_offset, q_hi = q_lo + (size_t)srv->layout.num_slots;
  uint32_t* const qd = a->q_dev + seat * N;
  if (pinned_range_device_pointer(q + q_lo, (q_hi - q_lo) * 4) != nullptr) {
    // the caller's buffer is page-locked already (cpir_host_alloc, hipHostMalloc, hipHostRegister): DMA straight from it
    std::lock_guard<std::mutex> ul(srv->upload_mu);
    up = hipMemcpyAsync(qd + q_lo, q + q_lo, (q_hi - q_lo) * 4, hipMemcpyHostToDevice, srv->up_stream);
    if (up == hipSuccess) up = hipEventRecord(a->seat_ev[seat], srv->up_stream);
  } else {
    uint32_t* const qp = a->q_pinned + seat * N;
    const size_t piece = (size_t)1 << 18;  // 1 MiB of u32
    std::unique_lock<std::mutex> ul(srv->upload_mu, std::try_to_lock);
    if (ul.owns_lock()) {
      // nobody else is uploading: in pieces, so that the DMA of one piece runs while the next ones are being copied into the pinned
      // block -- by this thread and, when they are free, by the staging helpers; the pieces are uploaded in order as they complete
      // (each copy costs the copy engine ~15 us whatever its size, so with helpers the query goes up in TWO halves, each copied by all
      // threads in 256 KiB jobs: the second half is copied while the first is on the link)
      const size_t words = q_hi - q_lo;
      if (words >= ((size_t)1 << 19) && g_staging.try_acquire()) {
        constexpr size_t kJob = (size_t)1 << 16;  // 256 KiB of u32
        constexpr size_t kMaxJobs = 512;
        const size_t half = (words / 2 + kJob - 1) / kJob * kJob;
        const size_t n_jobs = (words + kJob - 1) / kJob;
        if (n_jobs <= kMaxJobs) {
          std::atomic<int> done[kMaxJobs];
          for (size_t i = 0; i < n_jobs; i++) {
            done[i].store(0, std::memory_order_relaxed);
            const size_t o = q_lo + i * kJob, n = (q_hi - o < kJob) ? q_hi - o : kJob;
            g_staging.submit(StagingHelpers::Job{qp + o, q + o, n * 4, &done[i]});
          }
          size_t next = 0;
          for (int h = 0; h < 2; h++) {
            const size_t o_lo = q_lo + (h ? half : 0), o_hi = h ? q_hi : q_lo + half;
            const size_t j_hi = (o_hi - q_lo + kJob - 1) / kJob;
            for (; next < j_hi; next++)
              while (!done[next].load(std::memory_order_acquire))
                if (!g_staging.help()) {
#if defined(__x86_64__)
                  __builtin_ia32_pause();
#endif
                }
            if (up == hipSuccess) up = hipMemcpyAsync(qd + o_lo, qp + o_lo, (o_hi - o_lo) * 4, hipMemcpyHostToDevice, srv->up_stream);
          }
        } else {
          memcpy(qp + q_lo, q + q_lo, words * 4);
          up = hipMemcpyAsync(qd + q_lo, qp + q_lo, words * 4, hipMemcpyHostToDevice, srv->up_stream);
        }
        g_staging.release();
      } else {
        for (size_t o = q_lo; o < q_hi && up == hipSuccess; o += piece) {
          const size_t n = (q_hi - o < piece) ? q_hi - o : piece;
          memcpy(qp + o, q + o, n * 4);
          up = hipMemcpyAsync(qd + o, qp + o, n * 4, hipMemcpyHostToDevice, srv->up_stream);
        }
      }
    } else {
      // the link is busy with somebody else's query: copy while waiting, then upload in one piece when it is this query's turn
      memcpy(qp + q_lo, q + q_lo, (q_hi - q_lo) * 4);
      ul.lock();
      up = hipMemcpyAsync(qd + q_lo, qp + q_lo, (q_hi - q_lo) * 4, hipMemcpyHostToDevice, srv->up_stream);
    }
    if (up == hipSuccess) up = hipEventRecord(a->seat_ev[seat], srv->up_stream);
  }
  if (up != hipSuccess) set_last_hip_error(up, "hipMemcpyAsync(query upload)", __FILE__, __LINE__);

  const double t_staged = tr ? now_seconds() : 0;
  lk.lock();
  if (up != hipSuccess) a->status = CPIR_ERR_HIP;
  a->staged++;
  srv->cv.notify_all();
  if (leader) {
    // launch when every seat taken so far is staged AND the device is free of the previous arena's launch (or this one is full);
    // callers keep joining until then
    srv->cv.wait(lk, [&] {
      if (a->staged != a->joined) return false;
      if (a->joined == Server::kSeats) return true;
      for (const RespondArena& x : srv->arena)
        if (x.state == RespondArena::LAUNCHED) return false;
      return true;
    });
    a->state = RespondArena::LAUNCHED;  // closed: later callers open the next arena
    srv->cv.notify_all();
    const uint32_t k = a->joined;
    int st = a->status;
    lk.unlock();
    const double t_gate = tr ? now_seconds() : 0;
    hipError_t e = hipSuccess;
    if (st == CPIR_OK) {
      std::lock_guard<std::mutex> ll(srv->launch_mu);  // the launch sequences of two arenas must not interleave on the run stream
      for (uint32_t i = 0; i < k && e == hipSuccess; i++) e = hipStreamWaitEvent(srv->run_stream, a->seat_ev[i], 0);
      a->r0_zero = false;
      if (e == hipSuccess)
        st = respond_batched(srv->dev, srv->dtc, srv->layout, a->q_dev, srv->total_slots, srv->slot_offset, k, a->r_dev, nullptr, srv->run_stream);
      if (e == hipSuccess && st == CPIR_OK) e = hipMemcpyAsync(a->r_pinned, a->r_dev, (size_t)k * C * 4, hipMemcpyDeviceToHost, srv->run_stream);
      if (e == hipSuccess) e = hipEventRecord(a->done_ev, srv->run_stream);
    }
    const double t_enq = tr ? now_seconds() : 0;
    // always wait for what was enqueued for this arena before it can be reused: the uploads (they may have failed half way) and the launch
    hipError_t e2 = hipSuccess;
    if (st == CPIR_OK && e == hipSuccess) {
      e2 = wait_for_event(a->done_ev);
    } else {
      (void)hipStreamSynchronize(srv->up_stream);
      (void)hipStreamSynchronize(srv->run_stream);
    }
    if (e == hipSuccess) e = e2;
    if (st == CPIR_OK && e != hipSuccess) {
      set_last_hip_error(e, "respond launch / download", __FILE__, __LINE__);
      st = CPIR_ERR_HIP;
    }
    if (tr) {
      const double t_done = now_seconds();
      srv->trace.batches++, srv->trace.batch_hist[k]++;
      srv->trace.ns_gate += (uint64_t)((t_gate - t_staged) * 1e9), srv->trace.ns_enqueue += (uint64_t)((t_enq - t_gate) * 1e9);
      srv->trace.ns_gpu += (uint64_t)((t_done - t_enq) * 1e9);
    }
    lk.lock();
    a->status = st;
    a->state = RespondArena::DONE;
    srv->cv.notify_all();
  } else {
    srv->cv.wait(lk, [&] { return a->state == RespondArena::DONE; });
    if (tr) srv->trace.ns_follow += (uint64_t)((now_seconds() - t_staged) * 1e9);
  }
  const int status = a->status;
  lk.unlock();
  const double t_out0 = tr ? now_seconds() : 0;
  if (status == CPIR_OK) memcpy(r_out, a->r_pinned + seat * C, C * 4);
  if (tr) {
    srv->trace.calls++;
    srv->trace.ns_seat += (uint64_t)((t_seated - t_enter) * 1e9), srv->trace.ns_stage += (uint64_t)((t_staged - t_seated) * 1e9);
    srv->trace.ns_out += (uint64_t)((now_seconds() - t_out0) * 1e9);
  }
  lk.lock();
  if (++a->left == a->joined) {  // last one out frees the arena
    a->state = RespondArena::FREE;
    a->joined = a->staged = a->left = 0;
    srv->cv.notify_all();
  }
  return status;
}

int cpir_server_respond_bytes(const cpir_server* srv, const uint8_t* query, size_t query_len, uint8_t* response, size_t response_cap,
                              size_t* response_len) {
  if (!srv || !query || !response || !response_len) return CPIR_ERR_INVALID_ARGUMENT;
  // Matrix::from_bytes (matrix.rs:973-1010)
  if (query_len <= 8) return CPIR_ERR_FAILED_TO_DESERIALIZE_MATRIX;
  uint32_t rows, cols;
  memcpy(&rows, query, 4);
  memcpy(&cols, query + 4, 4);
  const uint64_t num = (uint64_t)rows * cols;
  if (num == 0 || num * 4 != (uint64_t)(query_len - 8)) return CPIR_ERR_FAILED_TO_DESERIALIZE_MATRIX;
  const uint32_t C = srv->layout.num_cols;
  const size_t need = 8 + (size_t)C * 4;
  if (response_cap < need) return CPIR_ERR_BUFFER_TOO_SMALL;
  // query + 8 may be only byte-aligned; cpir_server_respond memcpy's from it, so no alignment is required here
  std::vector<uint32_t> r(C);
  CPIR_TRY(cpir_server_respond(srv, reinterpret_cast<const uint32_t*>(query + 8), rows, cols, r.data()));
  const uint32_t one = 1;  // Matrix::to_bytes of the 1 x C response (matrix.rs:947-971)
  memcpy(response, &one, 4);
  memcpy(response + 4, &C, 4);
  memcpy(response + 8, r.data(), (size_t)C * 4);
  *response_len = need;
  return CPIR_OK;
}

int cpir_server_respond_device(const cpir_server* srv, const uint32_t* q_dev, uint32_t* r_dev, uint32_t* scratch_dev, void* stream) {
  if (!srv || !q_dev || !r_dev || !srv->shards.empty()) return CPIR_ERR_INVALID_ARGUMENT;  // device pointers belong to ONE device
  DeviceGuard g(srv->dev->ordinal);
  return launch_respond(srv->dev, srv->dtc, srv->layout, q_dev, srv->total_slots, srv->slot_offset, 1, 1, r_dev, scratch_dev,
                        pick_stream(srv->dev, stream));
}

int cpir_server_respond_batch_device(const cpir_server* srv, const uint32_t* q_dev, uint32_t batch, uint32_t* r_dev, uint32_t* scratch_dev,
                                     void* stream) {
  if (!srv || !q_dev || !r_dev || batch == 0 || !srv->shards.empty()) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(srv->dev->ordinal);
  return respond_batched(srv->dev, srv->dtc, srv->layout, q_dev, srv->total_slots, srv->slot_offset, batch, r_dev, scratch_dev,
                         pick_stream(srv->dev, stream));
}

}  // extern "C"
