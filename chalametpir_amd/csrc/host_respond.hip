// host_respond.hip -- Server::respond on HOST buffers (reference chalametpir_server/src/server.rs:184-190): the Server handle's life cycle,
// the arenas that coalesce and pipeline concurrent callers, the lone caller whose query the kernel reads in place, the in-process group of
// shards, and the extern "C" entry points cpir_server_respond*.
#include "server_internal.hpp"

namespace cpir {


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
    // a query compacted onto the slots the server holds (compact.hip) instead of copied: dst[i] = src[idx[i]] for bytes / 4 words --
    // streamed through the bitmap of the kept slots where the CPU has the instructions for it (host_gather.cpp), gathered otherwise
    const uint32_t* idx = nullptr;
    const uint8_t* bits = nullptr;
  };
  // The jobs of ONE query, an array on the submitter's stack: posted with one lock and one wake-up, claimed job by job with an atomic
  // counter (round 4 pushed every job into a deque under the mutex and popped it under the mutex: 72 jobs x 4 threads contending, 7 us
  // before the launch call of a lone pageable query could even be made).
  struct Batch {
    const Job* jobs = nullptr;
    size_t n = 0;
    std::atomic<size_t> next{0};
    std::atomic<int> users{0};  // helpers that hold a pointer to this batch
    uint32_t spin_us = 0;       // how long the helpers keep looking for the NEXT batch behind this one before they go to sleep
  };
  static void copy(const Job& j) {
    const size_t n = j.bytes / 4;
    if (j.idx && j.bits && n) (void)compress_words_streaming(static_cast<uint32_t*>(j.dst), static_cast<const uint32_t*>(j.src), j.bits, j.idx[0], (size_t)j.idx[n - 1] + 1);
    else if (j.idx) gather_words(static_cast<uint32_t*>(j.dst), static_cast<const uint32_t*>(j.src), j.idx, n);
    else memcpy(j.dst, j.src, j.bytes);
  }
  ~StagingHelpers() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_.fetch_add(1, std::memory_order_release);
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
  // hand the batch to the helpers (the caller holds the helpers: try_acquire); it must stay alive until retire() has returned
  void post(Batch* b) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      cur_ = b;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
  }
  // every job of the batch has been run (the caller has seen all `done` flags): no helper may keep a pointer to it
  void retire(Batch* b) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      cur_ = nullptr;
    }
    while (b->users.load(std::memory_order_acquire)) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
  // the submitter helps: run one unclaimed job of its own batch, if any
  static bool help(Batch* b) {
    const size_t i = b->next.fetch_add(1, std::memory_order_relaxed);
    if (i >= b->n) return false;
    copy(b->jobs[i]);
    b->jobs[i].done->store(1, std::memory_order_release);
    return true;
  }

 private:
  void run() {
    (void)pthread_setname_np(pthread_self(), "cpir-stage");
    uint64_t seen = 0;
    double spin_seconds = 0;
    for (;;) {
      // A caller in a loop (the reference's own online bench: one thread, one query after the other) comes back within a few hundred
      // microseconds: the helpers keep looking for that long before they go to sleep -- waking a sleeping thread costs the first
      // kilobytes of the next query's copy, which the kernel launched beside it waits for.
      // (only behind a LONE caller's batch: with concurrent callers the helpers would spin on cores the callers' own copies need)
      const double t0 = now_seconds();
      while (gen_.load(std::memory_order_acquire) == seen && now_seconds() - t0 < spin_seconds) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      Batch* b = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_.load(std::memory_order_relaxed) != seen; });
        if (stop_) return;
        seen = gen_.load(std::memory_order_relaxed);
        b = cur_;
        if (b) b->users.fetch_add(1, std::memory_order_relaxed);
      }
      spin_seconds = 0;
      if (b) {
        while (help(b)) {
        }
        spin_seconds = b->spin_us * 1e-6;
        b->users.fetch_sub(1, std::memory_order_release);
      }
    }
  }
  std::mutex owner_, mu_;
  std::condition_variable cv_;
  Batch* cur_ = nullptr;           // guarded by mu_
  std::atomic<uint64_t> gen_{0};   // bumped under mu_ with every post (and at shutdown): what the helpers watch
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

namespace {
// Device scratch for work enqueued on a stream (see cpir_internal.hpp: never the runtime's stream-ordered allocator).  A small POOL per process:
// a block handed back behind work that is still running (scratch_release_after: an event recorded on the caller's stream) may be handed out
// again only once that event has COMPLETED -- the "never recycle before completion" property round 5 bought with a hipMalloc + hipFree per
// call, without the allocation per call and without every back-to-back call keeping a workspace of its own alive until its event fires
// (the hint product's workspace is 2 bytes per entry of D).  A block nobody has asked for within kIdleSeconds of its event is freed by the
// pool's one background thread (hipFree waits for the device: never on a caller's thread).
class ScratchPool {
 public:
  static constexpr double kIdleSeconds = 0.5;
  ~ScratchPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  int acquire(void** p, size_t bytes, int ordinal) {
    *p = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      Block* best = nullptr;
      for (Block& b : blocks_) {
        if (b.ordinal != ordinal || b.bytes < bytes || b.bytes > bytes + bytes / 2 + (1u << 20)) continue;  // (no 17 GB block for a 4 KiB request)
        if (b.state == PENDING && hipEventQuery(b.done) == hipSuccess) b.state = IDLE, b.t_idle = now_seconds();
        else if (b.state == PENDING) (void)hipGetLastError();  // (hipErrorNotReady: not worth keeping)
        if (b.state == IDLE && (!best || b.bytes < best->bytes)) best = &b;
      }
      if (best) {
        best->state = IN_USE;
        *p = best->p;
        return CPIR_OK;
      }
    }
    void* q = nullptr;
    hipError_t e = CPIR_HIP_MALLOC(&q, bytes);
    if (e == hipErrorOutOfMemory) {  // blocks waiting for their events or idling in the pool may be what is missing: give them back, once
      (void)hipGetLastError();
      drain(ordinal);
      e = CPIR_HIP_MALLOC(&q, bytes);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_last_hip_error(e, "hipMalloc(scratch)", __FILE__, __LINE__);
      return e == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP;
    }
    Block b;
    b.p = q, b.bytes = bytes, b.ordinal = ordinal, b.state = IN_USE;
    if ((e = hipEventCreateWithFlags(&b.done, hipEventDisableTiming)) != hipSuccess) {
      set_last_hip_error(e, "hipEventCreateWithFlags(scratch)", __FILE__, __LINE__);
      (void)CPIR_HIP_FREE(q);
      return CPIR_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(mu_);
    if (!th_.joinable()) th_ = std::thread([this] { run(); });
    blocks_.push_back(b);
    *p = q;
    return CPIR_OK;
  }
  // the block may be reused (or freed) once everything enqueued on `stream` so far has completed
  int release_after(void* p, hipStream_t stream) {
    std::unique_lock<std::mutex> lk(mu_);
    for (Block& b : blocks_)
      if (b.p == p && b.state == IN_USE) {
        const hipError_t e = hipEventRecord(b.done, stream);
        if (e == hipSuccess) {
          b.state = PENDING;
          lk.unlock();
          cv_.notify_all();
          return CPIR_OK;
        }
        // no event to wait on: drain the stream -- the device, if that fails too -- and only then let the block go
        set_last_hip_error(e, "hipEventRecord(scratch)", __FILE__, __LINE__);
        lk.unlock();
        if (hipStreamSynchronize(stream) != hipSuccess) (void)hipDeviceSynchronize();
        lk.lock();
        for (Block& c : blocks_)
          if (c.p == p) c.state = IDLE, c.t_idle = 0;  // (idle since for ever: the background thread frees it on its next round)
        lk.unlock();
        cv_.notify_all();
        return CPIR_ERR_HIP;
      }
    return CPIR_ERR_INVALID_ARGUMENT;  // not a block of this pool
  }
  // until no block of that device waits for its event; the idle ones are freed
  void drain(int ordinal) {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      bool pending = false;
      for (Block& b : blocks_)
        if (b.ordinal == ordinal) {
          if (b.state == IDLE) b.t_idle = 0;
          if (b.state == PENDING || b.state == IDLE || b.state == FREEING) pending = true;
        }
      if (!pending) return;
      cv_.notify_all();
      idle_cv_.wait_for(lk, std::chrono::milliseconds(2));
    }
  }

 private:
  enum State { IN_USE, PENDING, IDLE, FREEING };
  struct Block {
    void* p = nullptr;
    size_t bytes = 0;
    int ordinal = 0;
    hipEvent_t done = nullptr;
    State state = IN_USE;
    double t_idle = 0;
  };
  void run() {
    (void)pthread_setname_np(pthread_self(), "cpir-scratch");
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      if (stop_) return;  // (the process is going down: nothing is given back to a runtime that may be going down too)
      bool any_pending = false, any_idle = false;
      void* victim = nullptr;  // a block to free, or (sync_first) a block whose event cannot be asked: looked up again by address once the lock is back
      bool sync_first = false;
      int victim_ordinal = 0;
      hipEvent_t victim_ev = nullptr;
      const double now = now_seconds();
      for (Block& b : blocks_) {
        if (b.state == PENDING) {
          const hipError_t e = hipEventQuery(b.done);
          if (e == hipSuccess) {
            b.state = IDLE, b.t_idle = now;
          } else {
            (void)hipGetLastError();
            if (e == hipErrorNotReady) {
              any_pending = true;
            } else if (!victim) {
              // the event cannot be asked (a device error): nothing may be assumed about the work that reads the block -- the whole device
              // is waited for before the block can be handed out or freed
              victim = b.p, victim_ordinal = b.ordinal, sync_first = true;
            }
          }
        }
        if (b.state == IDLE) {
          if (now - b.t_idle >= kIdleSeconds) {
            if (!victim) b.state = FREEING, victim = b.p, victim_ordinal = b.ordinal, victim_ev = b.done;
          } else {
            any_idle = true;
          }
        }
      }
      if (victim) {
        lk.unlock();
        {
          DeviceGuard g(victim_ordinal);
          if (sync_first) {
            (void)hipDeviceSynchronize();
          } else {
            (void)hipEventDestroy(victim_ev);
            (void)CPIR_HIP_FREE(victim);
          }
        }
        lk.lock();
        for (size_t i = 0; i < blocks_.size(); i++)
          if (blocks_[i].p == victim) {
            if (sync_first && blocks_[i].state == PENDING) blocks_[i].state = IDLE, blocks_[i].t_idle = now_seconds();
            else if (!sync_first && blocks_[i].state == FREEING) blocks_.erase(blocks_.begin() + (long)i);
            break;
          }
        idle_cv_.notify_all();
        continue;
      }
      idle_cv_.notify_all();
      if (any_pending) cv_.wait_for(lk, std::chrono::microseconds(200));  // events fire within a kernel's time
      else if (any_idle) cv_.wait_for(lk, std::chrono::milliseconds(100));
      else cv_.wait(lk, [&] {
        if (stop_) return true;
        for (const Block& b : blocks_)
          if (b.state == PENDING || b.state == IDLE) return true;
        return false;
      });
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, idle_cv_;
  std::vector<Block> blocks_;
  std::thread th_;
  bool stop_ = false;
};
ScratchPool g_scratch_pool;
}  // namespace

int scratch_acquire(void** p, size_t bytes, hipStream_t stream) {
  if (!p || bytes == 0) return CPIR_ERR_INVALID_ARGUMENT;
  *p = nullptr;
  // The pool's bookkeeping is an event recorded behind the work and asked from the host: a stream that is being CAPTURED into a graph
  // executes nothing now and its events cannot be asked -- and a replay of the graph would read a block that has long been handed to
  // somebody else.  The entry points that need scratch (cpir_op_mat_x_mat, cpir_op_mat_x_packed, respond on a compacted server through
  // kernels that cannot apply the slot map) are therefore not graph-capturable: refused here, before anything is enqueued.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
    set_last_hip_error(hipErrorStreamCaptureUnsupported, "this entry point allocates device scratch and cannot be captured into a graph", __FILE__, __LINE__);
    return CPIR_ERR_INVALID_ARGUMENT;
  }
  (void)hipGetLastError();
  int ordinal = 0;
  if (hipGetDevice(&ordinal) != hipSuccess) return CPIR_ERR_HIP;
  return g_scratch_pool.acquire(p, bytes, ordinal);
}

int scratch_release_after(void* p, hipStream_t stream) {
  if (!p) return CPIR_OK;
  return g_scratch_pool.release_after(p, stream);
}

void scratch_drain(int ordinal) { g_scratch_pool.drain(ordinal); }

void device_retain(Device* d) { d->refs.fetch_add(1); }
void device_release(Device* d) {
  if (d && d->refs.fetch_sub(1) == 1) {
    DeviceGuard g(d->ordinal);
    for (hipStream_t s : d->idle_streams) (void)hipStreamDestroy(s);
    if (d->up_stream) (void)hipStreamDestroy(d->up_stream);
    for (hipStream_t x : d->up_more)
      if (x) (void)hipStreamDestroy(x);
    if (d->run_stream) (void)hipStreamDestroy(d->run_stream);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
  }
}

int device_host_streams(Device* d) {
  std::lock_guard<std::mutex> lk(d->pool_mu);
  if (d->up_stream && d->run_stream) return CPIR_OK;
  // The two streams must not share a hardware queue (uploads would then serialise with kernels): HIP multiplexes the streams of one
  // priority level over a handful of queues in creation order, and a host process (torch, say) has usually created several already.
  // The run stream is created at the highest priority, which has queues of its own.
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // (least, greatest): numerically greatest <= least
  hipStream_t up = nullptr, run = nullptr;
  hipStream_t more[Device::kUpStreams - 1] = {nullptr, nullptr, nullptr};
  hipError_t e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
  for (hipStream_t& x : more)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&run, hipStreamNonBlocking, prio_hi);
  if (e != hipSuccess) {
    set_last_hip_error(e, "hipStreamCreateWithFlags (host path)", __FILE__, __LINE__);
    if (up) (void)hipStreamDestroy(up);
    for (hipStream_t x : more)
      if (x) (void)hipStreamDestroy(x);
    return CPIR_ERR_HIP;
  }
  d->up_stream = up, d->run_stream = run;
  for (int i = 0; i < Device::kUpStreams - 1; i++) d->up_more[i] = more[i];
  return CPIR_OK;
}

hipStream_t device_stream_acquire(Device* d) {
  {
    std::lock_guard<std::mutex> lk(d->pool_mu);
    if (!d->idle_streams.empty()) {
      hipStream_t s = d->idle_streams.back();
      d->idle_streams.pop_back();
      return s;
    }
  }
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_last_hip_error(e, "hipStreamCreateWithFlags (pool)", __FILE__, __LINE__);
    return nullptr;
  }
  return s;
}

void device_stream_release(Device* d, hipStream_t s) {
  if (!s) return;
  (void)hipStreamSynchronize(s);  // nothing may be in flight on a pooled stream
  std::lock_guard<std::mutex> lk(d->pool_mu);
  d->idle_streams.push_back(s);
}

// Where the device may read [p, p + bytes) of host memory in place (a DMA straight from the caller's buffer, or a kernel reading a lone
// query where it lies): the device address of p if the WHOLE range lies inside ONE page-locked allocation / registration the runtime has
// mapped; NULL otherwise (pageable memory, a registration that covers only part of the buffer, two registrations that merely touch).
// The runtime is asked for the extent of the allocation that holds p (range start + size); where it cannot tell, both ends of the range
// are looked up instead and must belong to mappings laid end to end.  Only a NO is remembered (per thread, for the last buffer asked
// about: a server loop hands over the same pageable buffer again and again, and a stale no merely stages a buffer that has been
// registered since); a yes is asked again every call, because a stale yes -- the buffer unregistered in between -- would let the device
// read unmapped host pages.
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
  hipPointerAttribute_t lo_attr;
  if (bytes > 0 && hipPointerGetAttributes(&lo_attr, c) == hipSuccess && lo_attr.type == hipMemoryTypeHost && lo_attr.devicePointer) {
    const char* dp = static_cast<const char*>(lo_attr.devicePointer);
    void* range_start = nullptr;
    size_t range_size = 0;
    hipPointer_attribute which[2] = {HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, HIP_POINTER_ATTRIBUTE_RANGE_SIZE};
    void* out[2] = {&range_start, &range_size};
    if (hipDrvPointerGetAttributes(2, which, out, reinterpret_cast<hipDeviceptr_t>(const_cast<char*>(c))) == hipSuccess && range_start && range_size) {
      // the start is reported in the address space the allocation was made in: the host's for registered / hipHostMalloc'ed memory
      const char* rs = static_cast<const char*>(range_start);
      const bool host_side = rs <= c && c + bytes <= rs + range_size;
      const bool dev_side = rs <= dp && dp + bytes <= rs + range_size;
      if (host_side || dev_side) dev = dp;
    } else {
      (void)hipGetLastError();
      hipPointerAttribute_t hi_attr;
      if (hipPointerGetAttributes(&hi_attr, c + bytes - 1) == hipSuccess && hi_attr.type == hipMemoryTypeHost && hi_attr.devicePointer &&
          static_cast<const char*>(hi_attr.devicePointer) - dp == (ptrdiff_t)(bytes - 1))
        dev = dp;
      else
        (void)hipGetLastError();
    }
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
  if (a.q_dev) (void)CPIR_HIP_FREE(a.q_dev);
  if (a.q_pinned) (void)CPIR_HIP_HOST_FREE(a.q_pinned);
  a = RespondArena{};
}

static void sync_upload_streams(Device* d) {
  if (d->up_stream) (void)hipStreamSynchronize(d->up_stream);
  for (hipStream_t x : d->up_more)
    if (x) (void)hipStreamSynchronize(x);
}

static void arenas_destroy(Server* srv) {
  // whatever this server enqueued on the device's shared host-path streams (a trailing memset of the response seat) must be done before
  // its blocks go (hipFree waits for the device anyway -- scripts/probes/free_sync_probe.hip -- this says it in the code)
  if (srv->run_stream) (void)hipStreamSynchronize(srv->run_stream);
  if (srv->up_stream) sync_upload_streams(srv->dev);
  for (RespondArena& a : srv->arena) arena_free(a);
  srv->up_stream = srv->run_stream = nullptr;  // (owned by the device handle)
}

// spare words behind the response seats of an arena's two blocks: the fill progress of a lone query in CPIR_FILL_LINES copies (pinned
// block; + 16 words so that the copies can start on a 64-byte line), the abort flag (device block, behind seat 0's response)
// (four sets of fill counts -- one per seat of an in-place round, a lone caller uses the first -- + one line behind them: the hand-over flag)
static constexpr uint32_t kRoundSeats = CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS;
static constexpr size_t kArenaSpareWords = (size_t)kRoundSeats * CPIR_FILL_LINES * 16 + 16 + 16;
static void publish_fill_progress(uint32_t* lines, uint32_t steps) {
  for (uint32_t i = 0; i < CPIR_FILL_LINES; i++) __atomic_store_n(lines + i * 16, steps, __ATOMIC_RELEASE);
}

// on first use of this arena (caller holds Server::mu).  A shard stages only its own slots of a query, but seats keep the full stride.
static int arena_create(Server* srv, RespondArena& a) {
  // kSeats queries, then kSeats responses (query block first: it stays 16-byte aligned)
  // (+ 16 words behind the responses: the device block's spare words follow seat 0's response when a lone caller has only one seat
  // to fill -- the abort flag of a polled launch; the pinned block's hold the fill progress the kernel polls)
  const size_t qw = (size_t)srv->total_slots * Server::kSeats, rw = ((size_t)srv->layout.num_cols * Server::kSeats + 3) / 4 * 4 + kArenaSpareWords;
  const size_t qcw = (size_t)srv->map.n_pad * Server::kSeats;  // (slot map: the seats' queries gathered onto the kept slots; device block only)
  auto fail = [&](hipError_t e, const char* what) {
    set_last_hip_error(e, what, __FILE__, __LINE__);
    arena_free(a);
    return e == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP;
  };
  hipError_t e = hipSuccess;
  if (!srv->up_stream) {
    CPIR_TRY(device_host_streams(srv->dev));
    srv->up_stream = srv->dev->up_stream, srv->run_stream = srv->dev->run_stream;
  }
  if ((e = CPIR_HIP_MALLOC(&a.q_dev, (qw + rw + qcw) * 4)) != hipSuccess) return fail(e, "hipMalloc(respond arena)");
  a.q_compact = qcw ? a.q_dev + qw + rw : nullptr;  // behind the responses: q_dev and r_dev keep their places
  // COHERENT (fine-grained) on purpose, whatever HIP_HOST_COHERENT says: a polled launch reads the fill progress and the query words
  // while the host is still writing them (respond_alone), which only works on memory the device does not cache
  if ((e = CPIR_HIP_HOST_MALLOC(&a.q_pinned, (qw + rw) * 4, hipHostMallocCoherent)) != hipSuccess) return fail(e, "hipHostMalloc(respond arena)");
  a.r_dev = a.q_dev + qw, a.r_pinned = a.q_pinned + qw;
  {
    void* dp = nullptr;
    if ((e = hipHostGetDevicePointer(&dp, a.q_pinned, 0)) != hipSuccess) return fail(e, "hipHostGetDevicePointer");
    a.q_pinned_dev = static_cast<const uint32_t*>(dp);
    const size_t off = (qw + rw - kArenaSpareWords + 15) / 16 * 16;  // the copies start on a 64-byte line (the block itself is page-aligned)
    a.fill_progress = a.q_pinned + off;
    a.fill_progress_dev = a.q_pinned_dev + off;
    a.handed = a.fill_progress + (size_t)kRoundSeats * CPIR_FILL_LINES * 16;  // a line of its own behind the fill counts
    a.handed_dev = const_cast<uint32_t*>(a.fill_progress_dev) + (size_t)kRoundSeats * CPIR_FILL_LINES * 16;
    __atomic_store_n(a.handed, 0u, __ATOMIC_RELAXED);
    a.hand_seq = 0;
  }
  a.r0_zero = false, a.r_zero_words = 0;
  a.seat_ev.assign(Server::kSeats, nullptr);
  for (hipEvent_t& ev : a.seat_ev)
    if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreateWithFlags");
  if ((e = hipEventCreateWithFlags(&a.done_ev, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreateWithFlags");
  return CPIR_OK;
}


static void group_dev_destroy(Server* srv) {
  for (Server::GroupDevCtx& c : srv->gdev) {
    for (size_t g = 0; g < c.stream.size(); g++) {
      Device* d = srv->shards[g]->dev;
      DeviceGuard dg(d->ordinal);
      if (c.stream[g]) device_stream_release(d, c.stream[g]);  // (drains it)
      if (g < c.ev.size() && c.ev[g]) (void)hipEventDestroy(c.ev[g]);
      if (g < c.buf.size() && c.buf[g]) (void)CPIR_HIP_FREE(c.buf[g]);
    }
    if (!srv->shards.empty()) {
      DeviceGuard dg(srv->shards[0]->dev->ordinal);
      if (c.in_ev) (void)hipEventDestroy(c.in_ev);
      if (c.done_ev) (void)hipEventDestroy(c.done_ev);
      if (c.partials) (void)CPIR_HIP_FREE(c.partials);
    }
    c = Server::GroupDevCtx{};
  }
}

static void group_ctx_destroy(Server* srv) {
  group_dev_destroy(srv);
  for (auto& w : srv->workers) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->stop = true;
    }
    w->cv.notify_all();
    for (std::thread& t : w->ths)
      if (t.joinable()) t.join();
  }
  srv->workers.clear();
  for (Server::GroupCtx& c : srv->gctx) c.lanes.clear();
  srv->gctx_ready = false;
}

void server_destroy(Server* srv) {
  if (!srv) return;
  if (srv->trace_on && srv->trace.calls.load()) {
    const Server::Trace& t = srv->trace;
    const double n = (double)t.calls.load(), nb = (double)(t.batches.load() ? t.batches.load() : 1);
    fprintf(stderr, "[cpir respond trace] %.0f calls in %.0f batches; us per call: seat wait %.1f, staging %.1f (copy / compaction %.1f, upload lock %.1f, "
                    "copy calls %.1f where split), copy out %.1f; followers wait %.1f; us per batch (leader): gate %.1f, enqueue %.1f, device %.1f; batch sizes",
            n, nb, t.ns_seat.load() / n / 1e3, t.ns_stage.load() / n / 1e3, t.ns_stage_copy.load() / n / 1e3, t.ns_stage_lock.load() / n / 1e3,
            t.ns_stage_enq.load() / n / 1e3, t.ns_out.load() / n / 1e3,
            t.ns_follow.load() / (n - nb > 0 ? n - nb : 1) / 1e3, t.ns_gate.load() / nb / 1e3, t.ns_enqueue.load() / nb / 1e3, t.ns_gpu.load() / nb / 1e3);
    for (int i = 1; i <= 8; i++) fprintf(stderr, " %d:%llu", i, (unsigned long long)t.batch_hist[i].load());
    fprintf(stderr, "; served alone (query read in place) %llu, %.1f us each; of those %llu by one launch polling the copy, %u such launches gave up\n",
            (unsigned long long)t.solo.load(), t.solo.load() ? t.ns_solo.load() / (double)t.solo.load() / 1e3 : 0.0,
            (unsigned long long)srv->fill_polled.load(), srv->fill_aborts.load());
    if (t.rounds.load()) {
      const double nr = (double)t.rounds.load();
      fprintf(stderr, "[cpir respond trace] %.0f in-place rounds, us per round (leader): seat to closed %.1f, launch calls %.1f, own copy behind them %.1f, "
                      "until handed over %.1f\n", nr, t.ns_r_close.load() / nr / 1e3, t.ns_r_launch.load() / nr / 1e3, t.ns_r_copy.load() / nr / 1e3,
              t.ns_r_done.load() / nr / 1e3);
    }
    if (t.polled.load()) {
      const double np = (double)t.polled.load();
      fprintf(stderr, "[cpir respond trace] polled launches, us from entry: jobs with the helpers %.1f, launch call back %.1f, last job copied %.1f\n",
              t.ns_p_submit.load() / np / 1e3, t.ns_p_launch.load() / np / 1e3, t.ns_p_copied.load() / np / 1e3);
    }
  }
  if (!srv->shards.empty()) {
    group_ctx_destroy(srv);
    for (Server* c : srv->shards) server_destroy(c);
    srv->shards.clear();
  }
  {
    DeviceGuard g(srv->dev->ordinal);
    arenas_destroy(srv);
    if (srv->dtc) (void)CPIR_HIP_FREE(srv->dtc);
    srv->map.reset();
  }
  device_release(srv->dev);
  delete srv;
}

// (the worker is handed its own GroupWorker: srv->workers is still growing while the first threads start, and reading the vector from
// here raced with its reallocation -- found by MALLOC_PERTURB_, which fills the freed storage)
static void group_worker_main(Server* srv, Server::GroupWorker* worker, size_t g) {
  Server::GroupWorker& w = *worker;
  const Server* child = srv->shards[g];
  (void)pthread_setname_np(pthread_self(), "cpir-group");
  (void)hipSetDevice(child->dev->ordinal);  // this thread only ever talks to its shard's device
  for (;;) {
    Server::GroupJob job;
    {
      std::unique_lock<std::mutex> lk(w.mu);
      w.cv.wait(lk, [&] { return w.stop || !w.jobs.empty(); });
      if (w.jobs.empty()) return;  // stop requested and nothing left
      job = w.jobs.front();
      w.jobs.pop_front();
    }
    // (a shard answered from ITS slots of the query: an ordinary host call on a server whose slots start at slot_offset)
    const int st = cpir_server_respond(static_cast<const cpir_server*>(child), job.q, 1, srv->total_slots, job.ctx->lanes[g].r.data());
    {
      std::lock_guard<std::mutex> lk(job.done->mu);
      if (st != CPIR_OK && job.done->status == CPIR_OK) job.done->status = st;
      job.done->remaining--;
      job.done->cv.notify_one();  // under the lock: `done` lives on the caller's stack and may go away as soon as it is released
    }
  }
}

// per call context and shard: room for the shard's partial response; per shard: its queue and the threads that serve it
static int group_ctx_create(Server* srv) {
  const uint32_t C = srv->layout.num_cols;
  for (Server::GroupCtx& c : srv->gctx) {
    c.lanes.resize(srv->shards.size());
    for (Server::GroupLane& l : c.lanes) l.r.assign(C, 0u);
  }
  srv->workers.reserve(srv->shards.size());
  for (size_t g = 0; g < srv->shards.size(); g++) srv->workers.emplace_back(new Server::GroupWorker);
  for (size_t g = 0; g < srv->shards.size(); g++)
    for (int t = 0; t < Server::kGroupCtx; t++) srv->workers[g]->ths.emplace_back(group_worker_main, srv, srv->workers[g].get(), g);
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
  memcpy(r_out, ctx->lanes[0].r.data(), (size_t)C * 4);
  for (size_t g = 1; g < srv->shards.size(); g++) {
    const uint32_t* p = ctx->lanes[g].r.data();
    for (uint32_t c = 0; c < C; c++) r_out[c] += p[c];  // u32 wrap-around
  }
  return CPIR_OK;
}


// A lone caller's response leaves the device by a one-block kernel behind the respond kernel on the same stream instead of a copy-engine
// download + an event: it stores the C response words (and the abort flag behind them) into the arena's page-locked block, zeroes the
// device copy for the next caller (no memset either), fences, and stores this call's sequence number into a flag line of that block,
// which the caller polls.  The respond kernel's own results are complete at the kernel boundary, so no fence is needed inside the big grid
// (round 3 tried handing over from the respond kernel's last block: every block then needed a release fence, +40 us).
__global__ void __launch_bounds__(256) respond_hand_over_kernel(uint32_t* __restrict__ r_dev, uint32_t words, uint32_t* __restrict__ r_host,
                                                                uint32_t* __restrict__ flag_host, uint32_t seq) {
  for (uint32_t i = threadIdx.x; i < words; i += 256) {
    __hip_atomic_store(r_host + i, r_dev[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    r_dev[i] = 0;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// partial responses of the shards, [shards][stride] words on the root device -> r[i] = sum over shards (u32 wrap-around)
__global__ void __launch_bounds__(256) group_sum_kernel(const uint32_t* __restrict__ partials, uint32_t shards, uint64_t stride, uint64_t count,
                                                        uint32_t* __restrict__ r) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
    uint32_t v = 0;
    for (uint32_t g = 0; g < shards; g++) v += partials[(uint64_t)g * stride + i];
    r[i] = v;
  }
}

// the CURRENT device `from` may address the memory of device `to` (both directions are asked for by the caller, each from its own device)
static int enable_peer_access(int from, int to) {
  int can = 0;
  hipError_t e = hipDeviceCanAccessPeer(&can, from, to);
  if (e == hipSuccess && can) {
    e = hipDeviceEnablePeerAccess(to, 0);
    if (e == hipSuccess) return CPIR_OK;
    if (e == hipErrorPeerAccessAlreadyEnabled) {
      (void)hipGetLastError();  // (sticky otherwise: the next launch check would report it)
      return CPIR_OK;
    }
  } else if (e == hipSuccess) {
    e = hipErrorPeerAccessUnsupported;
  }
  (void)hipGetLastError();
  char what[160];
  snprintf(what, sizeof what, "peer access from device %d to device %d (device-resident queries on a group need it; host queries do not)", from, to);
  set_last_hip_error(e, what, __FILE__, __LINE__);
  return CPIR_ERR_HIP;
}

static int group_dev_create(Server* srv, Server::GroupDevCtx& c) {
  const size_t G = srv->shards.size();
  const uint32_t C = srv->layout.num_cols;
  Device* root = srv->shards[0]->dev;
  c.stream.assign(G, nullptr), c.ev.assign(G, nullptr), c.buf.assign(G, nullptr);
  auto fail = [&](hipError_t e, const char* what) {
    set_last_hip_error(e, what, __FILE__, __LINE__);
    return e == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP;
  };
  hipError_t e;
  for (size_t g = 0; g < G; g++) {
    const Server* child = srv->shards[g];
    DeviceGuard dg(child->dev->ordinal);
    // peer access both ways: the shard pulls its slots of the queries out of the root's memory and pushes its partial responses into the
    // root's table.  Without it a device-resident query cannot be exchanged (the host entry points do not need it: they scatter and sum
    // on the host): a clear error instead of copies that fail later, at enqueue time or behind it.
    if (child->dev->ordinal != root->ordinal) CPIR_TRY(enable_peer_access(child->dev->ordinal, root->ordinal));
    if (!(c.stream[g] = device_stream_acquire(child->dev))) return CPIR_ERR_HIP;
    if ((e = hipEventCreateWithFlags(&c.ev[g], hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreateWithFlags(group)");
    const size_t words = ((size_t)child->layout.num_slots + 3) / 4 * 4 * Server::kBatchCap + (size_t)(C + 3) / 4 * 4 * Server::kBatchCap +
                         (size_t)child->map.n_pad * Server::kBatchCap;
    if ((e = CPIR_HIP_MALLOC(&c.buf[g], words * 4)) != hipSuccess) return fail(e, "hipMalloc(group lane)");
  }
  DeviceGuard dg(root->ordinal);
  for (size_t g = 1; g < G; g++) {
    const int peer = srv->shards[g]->dev->ordinal;
    if (peer != root->ordinal) CPIR_TRY(enable_peer_access(root->ordinal, peer));
  }
  if ((e = hipEventCreateWithFlags(&c.in_ev, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreateWithFlags(group)");
  if ((e = hipEventCreateWithFlags(&c.done_ev, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreateWithFlags(group)");
  if ((e = CPIR_HIP_MALLOC(&c.partials, G * (size_t)Server::kBatchCap * C * 4)) != hipSuccess) return fail(e, "hipMalloc(group partials)");
  c.ready = true, c.used = false;
  return CPIR_OK;
}

// cpir_server_respond_device / _batch_device on a GROUP handle: q_dev (batch x N words) and r_dev (batch x C words) on the root device
// (the device of shard 0), `stream` a stream of that device.  Everything is enqueued; nothing waits on the host.
static int group_respond_device(Server* srv, const uint32_t* q_dev, uint32_t batch, uint32_t* r_dev, hipStream_t stream) {
  const size_t G = srv->shards.size();
  const uint32_t C = srv->layout.num_cols;
  const uint64_t N = srv->total_slots;
  Device* root = srv->shards[0]->dev;
  std::lock_guard<std::mutex> lk(srv->gdev_mu);
  Server::GroupDevCtx& c = srv->gdev[srv->gdev_next++ % Server::kGroupCtx];
  if (!c.ready) {
    const int st = group_dev_create(srv, c);
    if (st != CPIR_OK) {
      group_dev_destroy(srv);
      return st;
    }
  }
#define TRY_(expr) do { const hipError_t _e = (expr); if (_e != hipSuccess) { set_last_hip_error(_e, #expr, __FILE__, __LINE__); return CPIR_ERR_HIP; } } while (0)
  for (uint32_t done = 0; done < batch; done += Server::kBatchCap) {
    const uint32_t nb = batch - done < Server::kBatchCap ? batch - done : Server::kBatchCap;
    {
      DeviceGuard dg(root->ordinal);
      TRY_(hipEventRecord(c.in_ev, stream));
    }
    for (size_t g = 0; g < G; g++) {
      const Server* child = srv->shards[g];
      DeviceGuard dg(child->dev->ordinal);
      hipStream_t s = c.stream[g];
      const size_t n = (size_t)child->layout.num_slots, qw = (n + 3) / 4 * 4, rw = (size_t)(C + 3) / 4 * 4;
      uint32_t* q_loc = c.buf[g];
      uint32_t* r_loc = q_loc + qw * Server::kBatchCap;
      uint32_t* qc_loc = child->map.active() ? r_loc + rw * Server::kBatchCap : nullptr;
      TRY_(hipStreamWaitEvent(s, c.in_ev, 0));
      // this shard's slots of the nb queries: rows of n words out of rows of N words, over the peer link where the devices differ
      TRY_(hipMemcpy2DAsync(q_loc, qw * 4, q_dev + (uint64_t)done * N + child->slot_offset, N * 4, n * 4, nb, hipMemcpyDeviceToDevice, s));
      CPIR_TRY(server_respond_on_device(child, q_loc, qw, 0, nb, nb == 1, r_loc, nullptr, qc_loc, s));
      // the previous round's table has been summed before this one writes into it -- waited for HERE, behind this round's respond, so that
      // round k + 1 streams the database while the root still adds up round k
      if (c.used) TRY_(hipStreamWaitEvent(s, c.done_ev, 0));
      TRY_(hipMemcpyAsync(c.partials + (g * Server::kBatchCap) * (size_t)C, r_loc, (size_t)nb * C * 4, hipMemcpyDeviceToDevice, s));
      TRY_(hipEventRecord(c.ev[g], s));
    }
    DeviceGuard dg(root->ordinal);
    for (size_t g = 0; g < G; g++) TRY_(hipStreamWaitEvent(stream, c.ev[g], 0));
    const uint64_t count = (uint64_t)nb * C;
    const unsigned blocks = (unsigned)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    hipLaunchKernelGGL(group_sum_kernel, dim3(blocks), dim3(256), 0, stream, c.partials, (uint32_t)G, (uint64_t)Server::kBatchCap * C, count,
                       r_dev + (uint64_t)done * C);
    TRY_(hipGetLastError());
    TRY_(hipEventRecord(c.done_ev, stream));
    c.used = true;
  }
#undef TRY_
  return CPIR_OK;
}

Server* server_new(Device* dev, const cpir_dtc_layout& L, uint64_t slot_offset, uint64_t total_slots) {
  Server* s = new Server;
  s->dev = dev;
  device_retain(dev);
  s->layout = L;
  s->phys = L;
  s->slot_offset = slot_offset;
  s->total_slots = total_slots;
  const char* tr = getenv("CPIR_RESPOND_TRACE");
  s->trace_on = tr && tr[0] == '1';
  return s;
}


void server_set_physical(Server* srv, const cpir_dtc_layout& phys, SlotMap* map) {
  srv->phys = phys;
  srv->map.reset();
  if (map && map->active()) {
    srv->map.keep_dev = map->keep_dev, map->keep_dev = nullptr;
    srv->map.keep_host = std::move(map->keep_host);
    srv->map.keep_bits = std::move(map->keep_bits);
    srv->map.n_kept = map->n_kept, srv->map.n_pad = map->n_pad, srv->map.n_orig = map->n_orig;
    map->reset();
  }
}

int server_respond_on_device(const Server* srv, const uint32_t* q, uint64_t q_len, uint64_t q_slot_offset, uint32_t batch, bool lone, uint32_t* r,
                             uint32_t* scratch, uint32_t* qc, hipStream_t stream) {
  if (!srv->map.active()) {
    if (lone) return launch_respond(srv->dev, srv->dtc, srv->phys, q, q_len, q_slot_offset, 1, 1, r, scratch, stream);
    return respond_batched(srv->dev, srv->dtc, srv->phys, q, q_len, q_slot_offset, batch, r, scratch, stream);
  }
  // only the slots with a non-zero row are resident: gather the queries onto them, then an ordinary respond on the compact database
  const SlotMap& m = srv->map;
  if (q_slot_offset + m.n_orig > q_len) return CPIR_ERR_SHARD_RANGE;
  // Where every launch of the batch is a wide pass, the kernel applies the map itself while it gathers the query words (no pass over the
  // queries in front of it: 12.0 -> 10.7 us per query for fused batches on the 2^20-key database Server::setup builds).  The decision and
  // the launches read the tuning separately: a launch that finds itself on another kernel after all refuses the map, and the batch is
  // answered again the long way.
  if (respond_batch_takes_slot_map(srv->phys, batch, lone, q_len)) {
    const int st = lone ? launch_respond(srv->dev, srv->dtc, srv->phys, q, q_len, q_slot_offset, 1, 1, r, scratch, stream, m.keep_dev)
                        : respond_batched(srv->dev, srv->dtc, srv->phys, q, q_len, q_slot_offset, batch, r, scratch, stream, m.keep_dev);
    if (st != CPIR_ERR_INVALID_ARGUMENT) return st;
  }
  uint32_t* own = nullptr;
  if (!qc) {
    CPIR_TRY(scratch_acquire(reinterpret_cast<void**>(&own), (size_t)batch * m.n_pad * 4, stream));
    qc = own;
  }
  int st = launch_gather_query(srv->dev, q, q_len, q_slot_offset, m, batch, qc, stream);
  if (st == CPIR_OK)
    st = lone ? launch_respond(srv->dev, srv->dtc, srv->phys, qc, m.n_pad, 0, 1, 1, r, scratch, stream)
              : respond_batched(srv->dev, srv->dtc, srv->phys, qc, m.n_pad, 0, batch, r, scratch, stream);
  if (own) {
    const int st2 = scratch_release_after(own, stream);
    if (st == CPIR_OK) st = st2;
  }
  return st;
}

// Any batch size.  With batch fusion every pass answers 4 queries from one stream of the database (remainder 2 / 1);
// without it every query is its own pass.  Either way the passes of one kind go into ONE launch.
int respond_batched(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                           uint64_t q_slot_offset, uint32_t batch, uint32_t* r, uint32_t* scratch, hipStream_t stream, const uint32_t* keep) {
  if (keep && L.packing != CPIR_PACK_PLANAR) return CPIR_ERR_INVALID_ARGUMENT;  // (see server_respond_on_device)
  if (!respond_batch_fusion()) {
    // One launch for all passes saves a kernel fill/drain (~10 us) per query, but blocks of a long multi-pass launch drift
    // apart and lose the L2 sharing of q: measured on MI355X it wins up to 1.3 GB per pass (196 vs 204 us) and loses at
    // 5 GB and above (806 vs 770 us), so very large databases get one launch per query.
    // (the matrix-core kernel keeps one launch at every size: 1 440 vs 1 505 us per query at 9.8 GB, 790 vs 799 at 5 GB)
    if (L.packing == CPIR_PACK_PLANAR || L.total_words * 4 <= respond_multi_pass_limit_bytes()) return launch_respond(dev, dtc, L, q, q_len, q_slot_offset, 1, batch, r, scratch, stream, keep);
    for (uint32_t i = 0; i < batch; i++)
      CPIR_TRY(launch_respond(dev, dtc, L, q + (uint64_t)i * q_len, q_len, q_slot_offset, 1, 1, r + (uint64_t)i * L.num_cols, scratch, stream));
    return CPIR_OK;
  }
  uint32_t done = 0;
  if (L.packing == CPIR_PACK_PLANAR) {
    // the matrix-core kernels take any 1..W queries per pass (W up to 24 on the wide kernel -- as few passes as that allows, all of about
    // the same width; 4 where the tuning sends fused passes to the step-major kernel): passes of W in one launch, then the rest as one pass
    const uint32_t W = respond_planar_pass_width(L, batch);
    if (batch >= W) {
      CPIR_TRY(launch_respond(dev, dtc, L, q, q_len, q_slot_offset, W, batch / W, r, scratch, stream, keep));
      done = batch / W * W;
    }
    if (done < batch)
      CPIR_TRY(launch_respond(dev, dtc, L, q + (uint64_t)done * q_len, q_len, q_slot_offset, batch - done, 1, r + (uint64_t)done * L.num_cols, scratch, stream, keep));
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


}  // namespace cpir

using namespace cpir;

extern "C" {

// ---------------------------------------------------------------------------------------------------------------
// server: respond
// ---------------------------------------------------------------------------------------------------------------
// A caller that found the server idle (it holds arena `a` alone, closed to others): no upload.  The step-major kernel reads each query
// word once, so it reads them where they are: in the caller's buffer if that is page-locked and 16-byte aligned, else in the arena's
// pinned block, which this thread and the staging helpers fill front to back while the kernel -- launched FIRST -- waits for each step's
// words (see below; two plain launches, each when its half is in place, if polling is off).  r_dev is kept zeroed between uses.
static int respond_alone(Server* srv, RespondArena* a, const uint32_t* q, uint32_t* r_out) {
  const size_t C = srv->layout.num_cols;
  // With a slot map (compact.hip) the kernel reads a COMPACT query -- the words of the kept slots only, q_len = map.n_pad, no offset --
  // which exists nowhere yet: it is produced in the pinned block by the same copy jobs that otherwise just copy (dst[i] = q[q_lo +
  // keep[i]]), so the polled launch applies unchanged and a page-locked caller buffer is no shortcut any more.
  const bool mapped = srv->map.active();
  const size_t q_lo = (size_t)srv->slot_offset;
  const size_t words = mapped ? (size_t)srv->map.n_kept : (size_t)srv->layout.num_slots;  // words of q the kernel reads
  const uint64_t kq_len = mapped ? srv->map.n_pad : srv->total_slots, kq_off = mapped ? 0 : srv->slot_offset;  // as the kernel addresses them
  const cpir_dtc_layout& L = srv->phys;
  hipStream_t st = srv->run_stream;
  std::lock_guard<std::mutex> ll(srv->dev->launch_mu);
  hipError_t e = hipSuccess;
  int rc = CPIR_OK;
  // seat 0's response and the word behind it (the abort flag of a polled launch) are kept zeroed between uses
  if (!a->r0_zero) e = hipMemsetAsync(a->r_dev, 0, (C + 1) * 4, st);
  a->r0_zero = false, a->r_zero_words = 0;
  bool polled = false;
  // the slots this server reads, q[q_lo, q_lo + words), as the device addresses them -- if the whole range is page-locked; the kernel
  // is handed the (possibly virtual) address of q[0] and adds the offset itself
  const uint32_t* in_place = nullptr;
  if (e == hipSuccess && !mapped && reinterpret_cast<uintptr_t>(q) % 16 == 0) {
    const void* dp = pinned_range_device_pointer(q + q_lo, words * 4);
    if (dp && reinterpret_cast<uintptr_t>(dp) % 16 == (q_lo * 4) % 16) in_place = static_cast<const uint32_t*>(dp) - q_lo;
  }
  if (e == hipSuccess && in_place) {
    journal_note("respond: q read in place", q + q_lo, words * 4, __FILE__, __LINE__);
    rc = launch_respond_read_once(srv->dev, srv->dtc, L, in_place, kq_len, kq_off, a->r_dev, st);
  } else if (e == hipSuccess) {
    // seat 0 of the pinned block: the same offsets as the caller's buffer, or (slot map) the compact query from word 0
    uint32_t* const qp = mapped ? a->q_pinned : a->q_pinned + q_lo;
    const uint32_t* const src = q + q_lo;
    const uint32_t* const idx = mapped ? srv->map.keep_host.data() : nullptr;
    const uint8_t* const bits = (mapped && compress_words_vectorised()) ? srv->map.keep_bits.data() : nullptr;
    // copy jobs of 64 KiB (32 steps of the kernel): the grid starts on the first 256 steps at once, and those 512 KiB are in place when
    // FOUR threads have copied two jobs each, not when two threads have copied 256 KiB each; queries too long for the job table take
    // 256 KiB jobs.  Measured, one caller, 16 pageable buffers taken in turn (scripts/host_path_cold.py): 231 us per query against 235-244
    // with 256 KiB jobs; more helper threads (7 instead of 3) and launching before handing out the jobs change nothing.
    constexpr size_t kJobSmall = (size_t)1 << 14, kJobBig = (size_t)1 << 16, kMaxJobs = 512;
    static_assert(kJobSmall % CPIR_PLANAR_SLOTS_PER_TILE == 0 && kJobBig % CPIR_PLANAR_SLOTS_PER_TILE == 0,
                  "a copy job must end on a step boundary: the kernel is told whole steps");
    const size_t kJob = (words + kJobSmall - 1) / kJobSmall <= kMaxJobs ? kJobSmall : kJobBig;
    const size_t kStepsPerJob = kJob / CPIR_PLANAR_SLOTS_PER_TILE;
    const size_t n_jobs = (words + kJob - 1) / kJob;
    // job i: words [i * kJob, ...) of what the kernel reads
    auto job = [&](size_t i, std::atomic<int>* done) {
      const size_t o = i * kJob, n = (words - o < kJob) ? words - o : kJob;
      return mapped ? StagingHelpers::Job{qp + o, src, n * 4, done, idx + o, bits} : StagingHelpers::Job{qp + o, src + o, n * 4, done, nullptr, nullptr};
    };
    // how long a wave waits for the words of a step (the tuning value, default 2 ms), but never less than the whole copy would take at
    // 5 GB/s -- a quarter of what ONE core copies: the last steps of a long query are legitimately waited for that long
    uint32_t fill_timeout_us = respond_host_fill_timeout_us();
    // (values below 100 us are taken as they are: how the tests make a launch give up)
    if (fill_timeout_us >= 100 && words * 4 / 5000 > fill_timeout_us) fill_timeout_us = (uint32_t)(words * 4 / 5000);
    if (words >= ((size_t)1 << 19) && n_jobs <= kMaxJobs && g_staging.try_acquire()) {
      std::atomic<int> done[kMaxJobs];
      // ONE launch that overlaps the copy: the kernel takes the steps of q round-robin (front to back over the whole grid) and waits
      // for each step's words to be in place, which this thread announces job by job in *fill_progress; the copy (~55 us for 4.7 MB)
      // runs underneath the stream (~200 us).  The jobs are handed to the helpers FIRST -- their wake-up and the first kilobytes overlap
      // the ~10 us this thread spends in the launch call -- and the count starts at 0, so the kernel cannot run ahead of them.
      // A wave that has waited fill_timeout_us gives up and flags the launch as void: the query
      // is then answered again from the (by then complete) pinned block -- a launch that cannot start before this thread moves on
      // (synchronous launches under a debugger or a serialising profiler) costs that timeout once, and after three such launches the
      // server stops polling and launches each half of the query when it is in place.
      // (no 128-byte line of the pinned block may straddle two copy jobs -- a wave that has seen job i's count could otherwise fetch a
      // line whose tail belongs to job i + 1: the words the kernel reads must start on a line boundary, which every shard_unit() multiple
      // does, and a compact query does by starting at word 0)
      polled = fill_timeout_us > 0 && srv->fill_aborts.load(std::memory_order_relaxed) < 3 && (mapped || (q_lo * 4) % 128 == 0);
      const bool ptr = srv->trace_on && polled;
      const double tp0 = ptr ? now_seconds() : 0;
      double tp1 = 0, tp2 = 0;
      if (polled) publish_fill_progress(a->fill_progress, 0u);
      StagingHelpers::Job jobs[kMaxJobs];
      for (size_t i = 0; i < n_jobs; i++) done[i].store(0, std::memory_order_relaxed), jobs[i] = job(i, &done[i]);
      StagingHelpers::Batch batch;
      batch.jobs = jobs, batch.n = n_jobs, batch.spin_us = respond_helper_spin_us();
      g_staging.post(&batch);
      if (ptr) tp1 = now_seconds();
      if (polled) {
        journal_note("respond: polled launch", a->q_pinned, words * 4, __FILE__, __LINE__);
        const PlanarHostFill fill{a->fill_progress_dev, a->r_dev + C, fill_timeout_us};
        rc = launch_respond_read_once(srv->dev, srv->dtc, L, a->q_pinned_dev, kq_len, kq_off, a->r_dev, st, 0, 0, &fill);
        if (rc != CPIR_OK) polled = false;  // nothing was launched
      }
      if (ptr) tp2 = now_seconds();
      auto wait_for_job = [&](size_t i) {
        while (!done[i].load(std::memory_order_acquire))
          if (!StagingHelpers::help(&batch)) {
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
          if (s_hi > s_lo) rc = launch_respond_read_once(srv->dev, srv->dtc, L, a->q_pinned_dev, kq_len, kq_off, a->r_dev, st, s_lo, s_hi);
        }
      }
      for (size_t i = 0; i < n_jobs; i++) wait_for_job(i);  // every job must have run before the stack arrays go away, whatever happened
      g_staging.retire(&batch);
      g_staging.release();
      if (ptr) {
        const double tp3 = now_seconds();
        srv->trace.polled++;
        srv->trace.ns_p_submit += (uint64_t)((tp1 - tp0) * 1e9), srv->trace.ns_p_launch += (uint64_t)((tp2 - tp0) * 1e9);
        srv->trace.ns_p_copied += (uint64_t)((tp3 - tp0) * 1e9);
      }
    } else if (words >= 2 * kJobSmall && n_jobs <= kMaxJobs && fill_timeout_us > 0 && srv->fill_aborts.load(std::memory_order_relaxed) < 3 &&
               (mapped || (q_lo * 4) % 128 == 0)) {
      // A shorter query (the slice of a shard of a group, a small database), or the helpers are taken: this thread copies alone -- but
      // still UNDER the launch, job by job, the kernel polling the copy's progress as above.
      polled = true;
      publish_fill_progress(a->fill_progress, 0u);
      journal_note("respond: polled launch (one copier)", a->q_pinned, words * 4, __FILE__, __LINE__);
      const PlanarHostFill fill{a->fill_progress_dev, a->r_dev + C, fill_timeout_us};
      rc = launch_respond_read_once(srv->dev, srv->dtc, L, a->q_pinned_dev, kq_len, kq_off, a->r_dev, st, 0, 0, &fill);
      if (rc != CPIR_OK) polled = false;  // nothing was launched
      for (size_t i = 0; i < n_jobs; i++) {
        StagingHelpers::copy(job(i, nullptr));
        if (polled) publish_fill_progress(a->fill_progress, i + 1 == n_jobs ? 0xffffffffu : (uint32_t)((i + 1) * kStepsPerJob));
      }
      if (!polled) rc = launch_respond_read_once(srv->dev, srv->dtc, L, a->q_pinned_dev, kq_len, kq_off, a->r_dev, st);
    } else {
      // (a very short query, or polling is off: this thread copies / compacts alone, then launches)
      StagingHelpers::copy(StagingHelpers::Job{qp, src, words * 4, nullptr, idx, mapped ? bits : nullptr});
      rc = launch_respond_read_once(srv->dev, srv->dtc, L, a->q_pinned_dev, kq_len, kq_off, a->r_dev, st);
    }
  }
  // (measured, one caller at 2^20 keys, scripts/host_path_cold.py: page-locked 214.5 -> 211.1 us, pageable 235 -> 232 us against the
  // copy-engine download + event of round 3, which stays as the path of last resort)
  constexpr bool no_hand_over = false;
  for (int attempt = 0; attempt < 2; attempt++) {
    if (e == hipSuccess && rc == CPIR_OK && !no_hand_over) {
      // the response is handed over by a one-block kernel (which also leaves seat 0 of r_dev zeroed): poll its flag
      const uint32_t seq = ++a->hand_seq ? a->hand_seq : ++a->hand_seq;  // never 0
      uint32_t* const r_host_dev = const_cast<uint32_t*>(a->q_pinned_dev) + (a->r_pinned - a->q_pinned);
      hipLaunchKernelGGL(respond_hand_over_kernel, dim3(1), dim3(256), 0, st, a->r_dev, (uint32_t)(C + 1), r_host_dev, a->handed_dev, seq);
      e = hipGetLastError();
      if (e == hipSuccess) {
        a->r0_zero = true, a->r_zero_words = (uint32_t)(C + 1);
        const double t0 = now_seconds();
        bool got = false;
        while (!(got = __atomic_load_n(a->handed, __ATOMIC_ACQUIRE) == seq)) {
          if (now_seconds() - t0 > 5e-3) break;  // something is very slow or wrong: fall back to waiting on the stream
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
        if (!got) {
          e = hipStreamSynchronize(st);
          if (e == hipSuccess && __atomic_load_n(a->handed, __ATOMIC_ACQUIRE) != seq) e = hipErrorUnknown;
        }
      }
    } else if (e == hipSuccess && rc == CPIR_OK) {
      e = hipMemcpyAsync(a->r_pinned, a->r_dev, (C + 1) * 4, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipEventRecord(a->done_ev, st);
      if (e == hipSuccess) {
        // zeros for the next lone caller, off this one's critical path
        if (hipMemsetAsync(a->r_dev, 0, (C + 1) * 4, st) == hipSuccess) a->r0_zero = true, a->r_zero_words = (uint32_t)(C + 1);
        else (void)hipGetLastError();
        e = wait_for_event(a->done_ev);
      }
    }
    if (!(e == hipSuccess && rc == CPIR_OK)) (void)hipStreamSynchronize(st);  // whatever was enqueued reads the caller's buffer / the pinned block: drain before returning
    if (!(polled && e == hipSuccess && rc == CPIR_OK && a->r_pinned[C] != 0)) break;
    // the polled launch gave up waiting: its results are void.  The pinned block is complete by now: answer from it, without polling.
    polled = false;
    journal_note("respond: polled launch VOID", a->q_pinned, words * 4, __FILE__, __LINE__);
    srv->served.polled_void.fetch_add(1, std::memory_order_relaxed);
    srv->fill_aborts.fetch_add(1, std::memory_order_relaxed);
    if (!a->r0_zero) e = hipMemsetAsync(a->r_dev, 0, (C + 1) * 4, st);
    a->r0_zero = false, a->r_zero_words = 0;
    if (e == hipSuccess) rc = launch_respond_read_once(srv->dev, srv->dtc, L, a->q_pinned_dev, kq_len, kq_off, a->r_dev, st);
  }
  if (polled) {
    srv->fill_polled.fetch_add(1, std::memory_order_relaxed);
    // a launch that kept up pays back one that did not: three void launches IN A ROW (a host that cannot feed the kernel: synchronous
    // launches, a throttled CPU quota) switch polling off, a stall now and then does not
    uint32_t n = srv->fill_aborts.load(std::memory_order_relaxed);
    while (n > 0 && n < 3 && !srv->fill_aborts.compare_exchange_weak(n, n - 1, std::memory_order_relaxed)) {
    }
  }
  if (rc == CPIR_OK && e != hipSuccess) {
    set_last_hip_error(e, "respond (query read in place)", __FILE__, __LINE__);
    rc = CPIR_ERR_HIP;
  }
  if (rc == CPIR_OK) memcpy(r_out, a->r_pinned, C * 4);
  return rc;
}

// An IN-PLACE ROUND (RespondArena::in_place): two to four concurrent callers answered by ONE pass of the step-major kernel that reads every
// query over the host link where it lies -- a page-locked caller buffer as it is; a pageable query from the seat's part of the arena's pinned
// block, into which its caller's own thread copies it front to back WHILE the pass, launched in front of the copies, polls each seat's
// progress (what respond_alone does for a lone caller, once per seat).  Against the upload path -- stage, upload the queries one after the
// other (83 us each at 2^20 keys x 1 kB), then the kernel -- a round of two costs one pass (~230 us), not an upload and a pass each.
// The caller holds seat `seat` of the open arena `a`; q0_in_place: the device-visible address of q[0] where the query is page-locked.
static int respond_in_round(Server* srv, RespondArena* a, uint32_t seat, const uint32_t* q, const uint32_t* q0_in_place, uint32_t inplace_cap,
                            uint32_t round_at_entry, uint32_t* r_out) {
  const size_t C = srv->layout.num_cols;
  // With a slot map (compact.hip) the pass reads COMPACT queries -- the words of the kept slots only, from word 0 of the seat's block,
  // q_len = map.n_pad -- which the copy jobs produce (as respond_alone's do): every caller is then a copying one, page-locked or not.
  const bool mapped = srv->map.active();
  const size_t q_lo = (size_t)srv->slot_offset;
  const size_t words = mapped ? (size_t)srv->map.n_kept : (size_t)srv->layout.num_slots;  // words of a query the pass reads
  const uint64_t kq_len = mapped ? srv->map.n_pad : srv->total_slots, kq_off = mapped ? 0 : srv->slot_offset;  // as the kernel addresses them
  const bool leader = seat == 0, pageable = q0_in_place == nullptr;
  const size_t stride = ((size_t)srv->total_slots + 31) / 32 * 32;  // of the seats' blocks: each starts on a 128-byte line (see respond_alone)
  uint32_t* const my_lines = a->fill_progress + (size_t)seat * CPIR_FILL_LINES * 16;
  uint32_t* const my_block = a->q_pinned + seat * stride + (mapped ? 0 : q_lo);
  const uint32_t* const idx = mapped ? srv->map.keep_host.data() : nullptr;
  const uint8_t* const bits = (mapped && compress_words_vectorised()) ? srv->map.keep_bits.data() : nullptr;
  // copy jobs as a lone caller's: 64 KiB (32 steps of the kernel), 256 KiB where the query is long
  constexpr size_t kJobSmall = (size_t)1 << 14, kJobBig = (size_t)1 << 16;
  const size_t kJob = (words + kJobSmall - 1) / kJobSmall <= 512 ? kJobSmall : kJobBig;
  const size_t n_jobs = pageable ? (words + kJob - 1) / kJob : 0;
  size_t next_job = 0;
  auto copy_a_job = [&] {
    const size_t o = next_job * kJob, n = (words - o < kJob) ? words - o : kJob;
    StagingHelpers::copy(mapped ? StagingHelpers::Job{my_block + o, q + q_lo, n * 4, nullptr, idx + o, bits}
                                : StagingHelpers::Job{my_block + o, q + q_lo + o, n * 4, nullptr, nullptr, nullptr});
    next_job++;
    publish_fill_progress(my_lines, next_job == n_jobs ? 0xffffffffu : (uint32_t)(next_job * (kJob / CPIR_PLANAR_SLOTS_PER_TILE)));
  };
  std::unique_lock<std::mutex> lk(srv->mu, std::defer_lock);
  int status = CPIR_OK;
  if (!leader) {
    while (next_job < n_jobs) copy_a_job();
    lk.lock();
    a->staged++;
    srv->cv.notify_all();
    lk.unlock();
    // the pass is a couple of hundred microseconds away and a sleeping thread takes tens of them to wake: look for its end for a while --
    // three times what one pass over the image takes (a round of four costs about two) + 100 us, 2 ms at most; a pass that is later than
    // that (a void pass answered again, a device busy with somebody else's work) is slept for, so that a follower burns a core for no longer
    // than the round can reasonably take (what the host path costs in cores)
    const double t0 = now_seconds();
    const double pass_seconds = (double)srv->phys.total_words * 4 / 6.8e12;
    const double spin = 3 * pass_seconds + 100e-6 < 2e-3 ? 3 * pass_seconds + 100e-6 : 2e-3;
    while (__atomic_load_n(&a->rounds_done, __ATOMIC_ACQUIRE) == round_at_entry && now_seconds() - t0 < spin) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    lk.lock();
    srv->cv.wait(lk, [&] { return a->state == RespondArena::DONE; });
  } else {
    // The leader closes the round when the device is free of the launch before AND the callers recently seen beside it have joined, or a
    // moment has passed (they were answered by the same pass and come back within tens of microseconds of each other; a second query costs
    // the pass a tenth of its time, a second pass all of it) -- and copies its own query meanwhile, job by job.
    // (the moment: a third of what one pass over the image takes, 5 .. 60 us -- a small database must not wait longer for company that may
    // have gone another way than answering alone would take)
    const double pass_s = (double)srv->phys.total_words * 4 / 6.8e12;
    const double window = pass_s * 0.3 < 5e-6 ? 5e-6 : (pass_s * 0.3 > 60e-6 ? 60e-6 : pass_s * 0.3);
    double t_ready = -1;
    uint32_t k = 0;
    const bool tr = srv->trace_on;
    const double tr0 = tr ? now_seconds() : 0;
    double tr1 = 0, tr2 = 0, tr3 = 0;
    for (bool closed = false; !closed;) {
      if (next_job < n_jobs) copy_a_job();
      lk.lock();
      bool device_free = true;
      for (const RespondArena& x : srv->arena)
        if (x.state == RespondArena::LAUNCHED) device_free = false;
      if (device_free) {
        const double now = now_seconds();
        if (t_ready < 0) t_ready = now;
        const uint32_t expected = srv->peak_inside < inplace_cap ? srv->peak_inside : inplace_cap;
        if (a->joined >= expected || now - t_ready >= window) {
          closed = true, k = a->joined;
          a->state = RespondArena::LAUNCHED;  // closed: later callers open the next arena
          srv->cv.notify_all();
        } else if (next_job >= n_jobs) {
          srv->cv.wait_for(lk, std::chrono::duration<double>(window - (now - t_ready)));
        }
      } else if (next_job >= n_jobs) {
        srv->cv.wait(lk);  // nothing left to copy: until the launch before is done
      }
      lk.unlock();
    }
    if (tr) tr1 = now_seconds();
    srv->served.in_place_rounds.fetch_add(1, std::memory_order_relaxed);
    srv->served.in_place_calls.fetch_add(k, std::memory_order_relaxed);
    bool polled = false;
    for (uint32_t i = 0; i < k; i++) polled = polled || a->seat_polled[i];
    uint32_t fill_timeout_us = respond_host_fill_timeout_us();
    if (fill_timeout_us >= 100 && words * 4 / 5000 > fill_timeout_us) fill_timeout_us = (uint32_t)(words * 4 / 5000);  // (as for a lone caller)
    hipStream_t st = srv->run_stream;
    hipError_t e = hipSuccess;
    int rc = CPIR_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
      uint32_t seq = 0;
      {
        std::lock_guard<std::mutex> ll(srv->dev->launch_mu);
        // (the k responses and the word behind them -- the abort flag of a polled pass -- start from zero: left so by the hand-over of the
        // round before where that was as wide, else cleared here)
        const bool clean = a->r_zero_words >= k * C + 1;
        a->r0_zero = false, a->r_zero_words = 0;
        if (!clean) e = hipMemsetAsync(a->r_dev, 0, ((size_t)k * C + 1) * 4, st);
        const PlanarHostFill fill{a->fill_progress_dev, a->r_dev + (size_t)k * C, fill_timeout_us, k};
        if (e == hipSuccess)
          rc = launch_respond_read_rows_in_place(srv->dev, srv->dtc, srv->phys, a->seat_q, k, kq_len, kq_off, a->r_dev, st, polled ? &fill : nullptr, true);
        if (e == hipSuccess && rc == CPIR_OK) {
          // the responses (and the flag word) are handed over by the one-block kernel that also leaves them zeroed on the device
          seq = ++a->hand_seq ? a->hand_seq : ++a->hand_seq;  // never 0
          uint32_t* const r_host_dev = const_cast<uint32_t*>(a->q_pinned_dev) + (a->r_pinned - a->q_pinned);
          hipLaunchKernelGGL(respond_hand_over_kernel, dim3(1), dim3(256), 0, st, a->r_dev, (uint32_t)(k * C + 1), r_host_dev, a->handed_dev, seq);
          e = hipGetLastError();
          if (e == hipSuccess) a->r0_zero = true, a->r_zero_words = (uint32_t)(k * C + 1);
        }
      }
      if (attempt == 0) {  // the leader's own query, under the pass that is already waiting for it
        if (tr) tr2 = now_seconds();
        while (next_job < n_jobs) copy_a_job();
        if (tr) tr3 = now_seconds();
        lk.lock();
        a->staged++;
        lk.unlock();
      }
      if (e == hipSuccess && rc == CPIR_OK) {
        const double t0 = now_seconds();
        bool got = false;
        while (!(got = __atomic_load_n(a->handed, __ATOMIC_ACQUIRE) == seq)) {
          if (now_seconds() - t0 > 5e-3) break;  // something is very slow or wrong: wait on the stream
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
        if (!got) {
          e = hipStreamSynchronize(st);
          if (e == hipSuccess && __atomic_load_n(a->handed, __ATOMIC_ACQUIRE) != seq) e = hipErrorUnknown;
        }
      } else {
        (void)hipStreamSynchronize(st);  // whatever was enqueued reads the callers' buffers / the pinned block: drain before anybody returns
      }
      if (!(polled && e == hipSuccess && rc == CPIR_OK && a->r_pinned[(size_t)k * C] != 0)) break;
      // a wave gave up waiting for a copy: the pass is void.  Answer again, without polling, once every seat's copy is complete.
      polled = false;
      journal_note("respond: polled round VOID", a->q_pinned, words * 4, __FILE__, __LINE__);
      srv->served.polled_void.fetch_add(1, std::memory_order_relaxed);
      srv->fill_aborts.fetch_add(1, std::memory_order_relaxed);
      lk.lock();
      srv->cv.wait(lk, [&] { return a->staged == k; });
      lk.unlock();
    }
    if (polled) {  // (a pass that kept up pays back one that did not, as for a lone caller)
      uint32_t n = srv->fill_aborts.load(std::memory_order_relaxed);
      while (n > 0 && n < 3 && !srv->fill_aborts.compare_exchange_weak(n, n - 1, std::memory_order_relaxed)) {
      }
    }
    if (tr) {
      Server::Trace& t = srv->trace;
      t.rounds++, t.batch_hist[k]++;
      t.ns_r_close += (uint64_t)((tr1 - tr0) * 1e9), t.ns_r_launch += (uint64_t)((tr2 - tr1) * 1e9), t.ns_r_copy += (uint64_t)((tr3 - tr2) * 1e9);
      t.ns_r_done += (uint64_t)((now_seconds() - tr3) * 1e9);
    }
    status = rc;
    if (rc == CPIR_OK && e != hipSuccess) {
      set_last_hip_error(e, "respond (round read in place)", __FILE__, __LINE__);
      status = CPIR_ERR_HIP;
    }
    lk.lock();
    // (a caller whose copy is still running -- a void pass that was not answered again because of an error -- must not find its block reused)
    srv->cv.wait(lk, [&] { return a->staged == k; });
    a->status = status;
    a->state = RespondArena::DONE;
    __atomic_fetch_add(&a->rounds_done, 1u, __ATOMIC_RELEASE);
    srv->cv.notify_all();
  }
  status = a->status;
  lk.unlock();
  if (status == CPIR_OK) memcpy(r_out, a->r_pinned + seat * C, C * 4);
  if (srv->trace_on) srv->trace.calls++;
  lk.lock();
  srv->inside--;
  if (++a->left == a->joined) {  // last one out frees the arena
    a->state = RespondArena::FREE;
    a->joined = a->staged = a->left = 0;
    srv->cv.notify_all();
  }
  return status;
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
  const bool read_once_ok = respond_read_once_applicable(srv->phys);
  const size_t q_lo = (size_t)srv->slot_offset, q_hi = q_lo + (size_t)srv->layout.num_slots;
  const void* const q_dev_visible = pinned_range_device_pointer(q + q_lo, (q_hi - q_lo) * 4);
  const bool caller_pinned = q_dev_visible != nullptr;
  // A query that a pass may read in place beside those of other callers (RespondArena::in_place): page-locked, 16-byte aligned, no slot map
  // (as many callers per pass as the tuning allows and as the step-major kernel's LDS accumulators hold responses: 48 KiB, one u32 per query
  // and padded column -- 8 kB values: one query, no such rounds)
  uint32_t inplace_cap = respond_inplace_seats();
  {
    const uint64_t per_query = (uint64_t)(C + 63) / 64 * 256;
    if (per_query * inplace_cap > (48u << 10)) inplace_cap = (uint32_t)((48u << 10) / per_query);
    // (short queries -- below 2^19 words, 2 MB -- at most three at a time: their uploads are cheap, and FOUR callers do 5-10 % better with two
    // arenas that alternate between uploading and answering -- 2^18 keys x 1 kB, page-locked: 22.4 against 21.2 k queries/s; three: 15.5 -> 19.5 k)
    if (q_hi - q_lo < ((size_t)1 << 19) && inplace_cap > 3) inplace_cap = 3;
  }
  const uint32_t* q0_in_place = nullptr;  // the device-visible address of q[0]
  const bool mapped_srv = srv->map.active();
  if (caller_pinned && inplace_cap >= 2 && read_once_ok && !mapped_srv) {
    const uint32_t* const p = static_cast<const uint32_t*>(q_dev_visible) - q_lo;
    if (reinterpret_cast<uintptr_t>(p) % 16 == 0) q0_in_place = p;
  }
  // ... or pageable and long enough for a pass to poll its copy (as a lone caller's: 2^15 words, respond.host_fill_timeout_us > 0, fewer than
  // three passes in a row that gave up; the shard's slots start on a 128-byte line of the query)
  // (a server with a slot map reads compact queries, which exist nowhere until they are staged: a pageable query is compacted while it is
  // copied, at the cost of the copy; a page-locked one stays on the upload path there -- DMA of the whole query, the kernel applies the
  // map -- which is as fast for two such callers and faster for four: 9.0 against 7.0 k queries/s, profiles/r5_inplace_rounds_ab.txt)
  const bool stage_in_place = !q0_in_place && !(mapped_srv && caller_pinned) && inplace_cap >= 2 && read_once_ok &&
                              (mapped_srv ? (size_t)srv->map.n_kept : q_hi - q_lo) >= ((size_t)1 << 15) && (mapped_srv || (q_lo * 4) % 128 == 0) &&
                              respond_host_fill_timeout_us() > 0 && srv->fill_aborts.load(std::memory_order_relaxed) < 3;
  std::unique_lock<std::mutex> lk(srv->mu);
  RespondArena* a = nullptr;
  bool solo = false;
  // ... and only while the callers recently seen inside at the same time are few enough for one such pass: more of them are link-bound
  // either way, and the upload path overlaps their copies with the kernel of the arena before
  const uint32_t company = srv->inside + 1 > srv->peak_inside ? srv->inside + 1 : srv->peak_inside;
  const bool want_in_place = (q0_in_place || stage_in_place) && company >= 2 && company <= inplace_cap;
  for (;;) {
    if (want_in_place) {
      for (RespondArena& x : srv->arena)  // 1. an open in-place arena with a seat left
        if (!a && x.state == RespondArena::OPEN && x.in_place && x.joined < inplace_cap) a = &x;
      if (!a)
        for (RespondArena& x : srv->arena)  // 2. a free arena
          if (!a && x.state == RespondArena::FREE) {
            if (!x.q_dev) CPIR_TRY(arena_create(srv, x));
            a = &x, x.state = RespondArena::OPEN, x.status = CPIR_OK;
            x.compact_seats = false, x.in_place = true;
          }
    } else {
      for (RespondArena& x : srv->arena)  // 1. an open arena that is still spreading
        if (!a && x.state == RespondArena::OPEN && !x.in_place && x.joined < srv->spread()) a = &x;
      if (!a)
        for (RespondArena& x : srv->arena)  // 2. a free arena
          if (!a && x.state == RespondArena::FREE) {
            if (!x.q_dev) CPIR_TRY(arena_create(srv, x));
            a = &x, x.state = RespondArena::OPEN, x.status = CPIR_OK;
            x.compact_seats = srv->map.active() && !caller_pinned;  // (see RespondArena)
            x.in_place = false;
            // nobody else is filling an arena or on the device, and no company expected (with recent concurrent callers a lone launch
            // would only split the batch they are about to form): this caller is served alone, its query read in place
            solo = read_once_ok && srv->spread() == 1;
            for (const RespondArena& y : srv->arena)
              if (&y != a && (y.state == RespondArena::OPEN || y.state == RespondArena::LAUNCHED)) solo = false;
            if (solo) x.state = RespondArena::LAUNCHED;  // closed at once: later callers open the next arena and upload meanwhile
          }
      if (!a)
        for (RespondArena& x : srv->arena)  // 3. no arena free: fill the open one up
          if (!a && x.state == RespondArena::OPEN && !x.in_place && x.joined < Server::kSeats) a = &x;
    }
    if (a) break;
    srv->cv.wait(lk);  // every arena is full or in flight
  }
  const uint32_t seat = a->joined++;
  const bool leader = (seat == 0);
  const bool compact = a->compact_seats;
  const bool in_place_round = a->in_place;
  const uint32_t round_at_entry = __atomic_load_n(&a->rounds_done, __ATOMIC_RELAXED);
  if (in_place_round) {
    // (under the lock, before the round can be closed and launched: where the pass finds this seat's query, and how much of it is in place)
    const size_t stride = ((size_t)srv->total_slots + 31) / 32 * 32;
    a->seat_q[seat] = q0_in_place ? q0_in_place : a->q_pinned_dev + seat * stride;
    a->seat_polled[seat] = q0_in_place == nullptr;
    publish_fill_progress(a->fill_progress + (size_t)seat * CPIR_FILL_LINES * 16, q0_in_place ? 0xffffffffu : 0u);
    if (!leader) srv->cv.notify_all();  // (a leader with nothing to copy sleeps on its window: the company it waits for is here)
  }
  srv->caller_enters();
  lk.unlock();
  if (in_place_round) {
    srv->served.calls.fetch_add(1, std::memory_order_relaxed);
    return respond_in_round(srv, a, seat, q, q0_in_place, inplace_cap, round_at_entry, r_out);
  }
  const double t_seated = tr ? now_seconds() : 0;
  srv->served.calls.fetch_add(1, std::memory_order_relaxed);
  if (solo) {
    srv->served.alone.fetch_add(1, std::memory_order_relaxed);
    const int st = respond_alone(srv, a, q, r_out);
    if (tr) {
      srv->trace.calls++, srv->trace.solo++;
      srv->trace.ns_seat += (uint64_t)((t_seated - t_enter) * 1e9), srv->trace.ns_solo += (uint64_t)((now_seconds() - t_seated) * 1e9);
    }
    lk.lock();
    srv->inside--;
    a->state = RespondArena::FREE;
    a->joined = a->staged = a->left = 0;
    srv->cv.notify_all();
    return st;
  }

  // ---- stage the query (the reference copies too: from_bytes .to_vec(), matrix.rs:1001-1007) and enqueue its upload -----------------
  // (a shard reads only its own slots of the query: only those are staged and uploaded)
  // (two upload streams taken in turn, query by query: a copy costs the copy engine ~15 us of set-up whatever its size, which one stream
  // pays between two queries' 83 us on the link and two streams overlap)
  hipError_t up = hipSuccess;
  hipStream_t ups = nullptr;
  const uint32_t n_ups = respond_upload_streams();
  auto take_upload_stream = [&] {  // under upload_mu
    const uint32_t t = srv->dev->up_turn++ % n_ups;
    ups = t ? srv->dev->up_more[t - 1] : srv->up_stream;
  };
  uint32_t* const qd = a->q_dev + seat * N;
  if (compact) {
    // A server that holds only the slots with a non-zero row: the query is COMPACTED onto them while it is staged (host_gather.cpp: one
    // sequential pass over the source through the bitmap of the kept slots, about the cost of the memcpy it replaces), so the link carries
    // n_kept of the N words -- 8/9 for a real encoded database -- and the launch needs no map.  In source pieces of 1 MiB where nobody
    // else is uploading (the DMA of one piece runs while the next is compacted), else whole and uploaded when the link is this query's.
    const SlotMap& m = srv->map;
    uint32_t* const qp = a->q_pinned + seat * N;              // (room for N words; n_kept are written)
    uint32_t* const qcd = a->q_compact + seat * m.n_pad;      // the seat's compact query on the device
    const uint32_t* const src = q + q_lo;
    const size_t n_orig = (size_t)m.n_orig;
    // (every upload starts on a 256-byte boundary of the seat and is a whole number of 64 words long -- the last one runs to n_pad, whose
    // words behind n_kept nobody reads: copies with ragged ends take the runtime's slow path)
    const size_t n_pad = (size_t)m.n_pad;
    std::unique_lock<std::mutex> ul(srv->dev->upload_mu, std::try_to_lock);
    double tc = 0, tl = 0, te = 0, t0 = tr ? now_seconds() : 0;  // (trace: compaction / waiting for the link's turn / the runtime's copy calls)
    auto lap = [&](double& acc) {
      if (tr) {
        const double t1 = now_seconds();
        acc += t1 - t0, t0 = t1;
      }
    };
    if (!compress_words_vectorised()) {
      gather_words(qp, src, m.keep_host.data(), (size_t)m.n_kept);
      lap(tc);
      if (!ul.owns_lock()) ul.lock();
      lap(tl);
      take_upload_stream();
      up = hipMemcpyAsync(qcd, qp, n_pad * 4, hipMemcpyHostToDevice, ups);
      lap(te);
    } else if (ul.owns_lock()) {
      take_upload_stream();
      const size_t piece = (size_t)1 << 18;
      size_t out = 0, sent = 0;
      for (size_t o = 0; o < n_orig && up == hipSuccess; o += piece) {
        const bool last = o + piece >= n_orig;
        out += compress_words_streaming(qp + out, src, m.keep_bits.data(), o, last ? n_orig : o + piece);
        lap(tc);
        const size_t upto = last ? n_pad : (out & ~(size_t)63);
        if (upto > sent) up = hipMemcpyAsync(qcd + sent, qp + sent, (upto - sent) * 4, hipMemcpyHostToDevice, ups), sent = upto;
        lap(te);
      }
    } else {
      (void)compress_words_streaming(qp, src, m.keep_bits.data(), 0, n_orig);
      lap(tc);
      ul.lock();
      lap(tl);
      take_upload_stream();
      up = hipMemcpyAsync(qcd, qp, n_pad * 4, hipMemcpyHostToDevice, ups);
      lap(te);
    }
    if (up == hipSuccess) up = hipEventRecord(a->seat_ev[seat], ups);
    if (tr) srv->trace.ns_stage_copy += (uint64_t)(tc * 1e9), srv->trace.ns_stage_lock += (uint64_t)(tl * 1e9), srv->trace.ns_stage_enq += (uint64_t)(te * 1e9);
  } else if (caller_pinned) {
    // the caller's buffer is page-locked already (cpir_host_alloc, hipHostMalloc, hipHostRegister): DMA straight from it
    std::lock_guard<std::mutex> ul(srv->dev->upload_mu);
    take_upload_stream();
    up = hipMemcpyAsync(qd + q_lo, q + q_lo, (q_hi - q_lo) * 4, hipMemcpyHostToDevice, ups);
    if (up == hipSuccess) up = hipEventRecord(a->seat_ev[seat], ups);
  } else {
    uint32_t* const qp = a->q_pinned + seat * N;
    const size_t piece = (size_t)1 << 18;  // 1 MiB of u32
    std::unique_lock<std::mutex> ul(srv->dev->upload_mu, std::try_to_lock);
    if (ul.owns_lock()) {
      take_upload_stream();
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
          StagingHelpers::Job jobs[kMaxJobs];
          for (size_t i = 0; i < n_jobs; i++) {
            done[i].store(0, std::memory_order_relaxed);
            const size_t o = q_lo + i * kJob, n = (q_hi - o < kJob) ? q_hi - o : kJob;
            jobs[i] = StagingHelpers::Job{qp + o, q + o, n * 4, &done[i]};
          }
          StagingHelpers::Batch batch;
          batch.jobs = jobs, batch.n = n_jobs;
          g_staging.post(&batch);
          size_t next = 0;
          for (int h = 0; h < 2; h++) {
            const size_t o_lo = q_lo + (h ? half : 0), o_hi = h ? q_hi : q_lo + half;
            const size_t j_hi = (o_hi - q_lo + kJob - 1) / kJob;
            for (; next < j_hi; next++)
              while (!done[next].load(std::memory_order_acquire))
                if (!StagingHelpers::help(&batch)) {
#if defined(__x86_64__)
                  __builtin_ia32_pause();
#endif
                }
            if (up == hipSuccess) up = hipMemcpyAsync(qd + o_lo, qp + o_lo, (o_hi - o_lo) * 4, hipMemcpyHostToDevice, ups);
          }
          g_staging.retire(&batch);  // (every job has run: no helper may keep a pointer to the stack arrays)
        } else {
          memcpy(qp + q_lo, q + q_lo, words * 4);
          up = hipMemcpyAsync(qd + q_lo, qp + q_lo, words * 4, hipMemcpyHostToDevice, ups);
        }
        g_staging.release();
      } else {
        for (size_t o = q_lo; o < q_hi && up == hipSuccess; o += piece) {
          const size_t n = (q_hi - o < piece) ? q_hi - o : piece;
          memcpy(qp + o, q + o, n * 4);
          up = hipMemcpyAsync(qd + o, qp + o, n * 4, hipMemcpyHostToDevice, ups);
        }
      }
    } else {
      // the link is busy with somebody else's query: copy while waiting, then upload in one piece when it is this query's turn
      const double t0 = tr ? now_seconds() : 0;
      memcpy(qp + q_lo, q + q_lo, (q_hi - q_lo) * 4);
      const double t1 = tr ? now_seconds() : 0;
      ul.lock();
      const double t2 = tr ? now_seconds() : 0;
      take_upload_stream();
      up = hipMemcpyAsync(qd + q_lo, qp + q_lo, (q_hi - q_lo) * 4, hipMemcpyHostToDevice, ups);
      if (tr) srv->trace.ns_stage_copy += (uint64_t)((t1 - t0) * 1e9), srv->trace.ns_stage_lock += (uint64_t)((t2 - t1) * 1e9), srv->trace.ns_stage_enq += (uint64_t)((now_seconds() - t2) * 1e9);
    }
    if (up == hipSuccess) up = hipEventRecord(a->seat_ev[seat], ups);
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
    // ... and, when the device is free right away, for a moment longer if company is expected (spread() callers make a batch and fewer
    // have joined): callers that were answered together come back within tens of microseconds of each other, and the first one back
    // would otherwise launch alone and leave the rest to the next pass (8 kB values, 8 callers: passes of 1 and 7 alternating).  The
    // moment is a tenth of the kernel's time, at most 100 us.
    const auto gate_ready = [&] {
      if (a->staged != a->joined) return false;
      if (a->joined == Server::kSeats) return true;
      for (const RespondArena& x : srv->arena)
        if (x.state == RespondArena::LAUNCHED) return false;
      return true;
    };
    const double window = srv->batching_window_seconds();
    double t_ready = -1;  // when the gate was first found open (the moment counts from there, not from when this caller arrived:
                          // the callers of the pass that has just finished are the company to wait for)
    for (;;) {
      if (!gate_ready()) {
        srv->cv.wait(lk);
        continue;
      }
      const double now = now_seconds();
      if (t_ready < 0) t_ready = now;
      if (a->joined >= srv->spread() || now - t_ready >= window) break;
      srv->cv.wait_for(lk, std::chrono::duration<double>(window - (now - t_ready)));
    }
    a->state = RespondArena::LAUNCHED;  // closed: later callers open the next arena
    srv->cv.notify_all();
    const uint32_t k = a->joined;
    int st = a->status;
    lk.unlock();
    srv->served.uploaded_rounds.fetch_add(1, std::memory_order_relaxed);
    srv->served.in_uploaded_rounds.fetch_add(k, std::memory_order_relaxed);
    const double t_gate = tr ? now_seconds() : 0;
    hipError_t e = hipSuccess;
    if (st == CPIR_OK) {
      std::lock_guard<std::mutex> ll(srv->dev->launch_mu);  // the launch sequences of two arenas must not interleave on the run stream
      for (uint32_t i = 0; i < k && e == hipSuccess; i++) e = hipStreamWaitEvent(srv->run_stream, a->seat_ev[i], 0);
      a->r0_zero = false, a->r_zero_words = 0;
      if (e == hipSuccess)
        st = compact ? respond_batched(srv->dev, srv->dtc, srv->phys, a->q_compact, srv->map.n_pad, 0, k, a->r_dev, nullptr, srv->run_stream)
                     : server_respond_on_device(srv, a->q_dev, srv->total_slots, srv->slot_offset, k, false, a->r_dev, nullptr, a->q_compact, srv->run_stream);
      if (e == hipSuccess && st == CPIR_OK) e = hipMemcpyAsync(a->r_pinned, a->r_dev, (size_t)k * C * 4, hipMemcpyDeviceToHost, srv->run_stream);
      if (e == hipSuccess) e = hipEventRecord(a->done_ev, srv->run_stream);
    }
    const double t_enq = tr ? now_seconds() : 0;
    // always wait for what was enqueued for this arena before it can be reused: the uploads (they may have failed half way) and the launch
    hipError_t e2 = hipSuccess;
    if (st == CPIR_OK && e == hipSuccess) {
      e2 = wait_for_event(a->done_ev);
    } else {
      sync_upload_streams(srv->dev);
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
  srv->inside--;
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
  // query + 8 may be only byte-aligned: every host-side read of the query words is a byte copy or an unaligned vector load (the staging
  // memcpy's; host_gather.cpp's compaction of a lone query of a server with a slot map: load_word / loadu / gathers), so no alignment is
  // required here (tests/test_gpu_compact.py::test_wire_buffer_at_odd_addresses)
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
  if (!srv || !q_dev || !r_dev) return CPIR_ERR_INVALID_ARGUMENT;
  if (!srv->shards.empty())  // a group: q and r on the device of shard 0; the exchange is peer copies + a sum kernel there
    return group_respond_device(const_cast<cpir_server*>(srv), q_dev, 1, r_dev, pick_stream(srv->shards[0]->dev, stream));
  DeviceGuard g(srv->dev->ordinal);
  return server_respond_on_device(srv, q_dev, srv->total_slots, srv->slot_offset, 1, true, r_dev, scratch_dev, nullptr, pick_stream(srv->dev, stream));
}

int cpir_server_respond_batch_device(const cpir_server* srv, const uint32_t* q_dev, uint32_t batch, uint32_t* r_dev, uint32_t* scratch_dev,
                                     void* stream) {
  if (!srv || !q_dev || !r_dev || batch == 0) return CPIR_ERR_INVALID_ARGUMENT;
  if (!srv->shards.empty()) return group_respond_device(const_cast<cpir_server*>(srv), q_dev, batch, r_dev, pick_stream(srv->shards[0]->dev, stream));
  DeviceGuard g(srv->dev->ordinal);
  return server_respond_on_device(srv, q_dev, srv->total_slots, srv->slot_offset, batch, false, r_dev, scratch_dev, nullptr,
                                  pick_stream(srv->dev, stream));
}


}  // extern "C"
