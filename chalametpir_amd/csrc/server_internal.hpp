// server_internal.hpp -- what the three host-side translation units of libchalamet_hip.so share: the Server handle (capi.hip: the extern "C"
// boundary, devices, shapes, low-level operations; host_setup.hip: Server::setup; host_respond.hip: Server::respond on host buffers).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <list>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "cpir_internal.hpp"

namespace cpir {

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
//     pinned block, which the caller's thread and the staging helpers fill front to back WHILE the kernel, launched first, polls the
//     copy's progress (respond_alone in host_respond.hip).  (respond.host_zero_copy=0: upload first, as concurrent callers do.)
struct RespondArena {
  uint32_t* q_dev = nullptr;     // kSeats x total_slots u32
  uint32_t* q_compact = nullptr; // kSeats x map.n_pad u32 (servers with a slot map only): the seats' queries gathered onto the kept slots
  uint32_t* r_dev = nullptr;     // kSeats x C u32
  uint32_t* q_pinned = nullptr;  // kSeats x total_slots u32
  uint32_t* r_pinned = nullptr;  // kSeats x C u32
  std::vector<hipEvent_t> seat_ev;  // upload of seat i has crossed the link
  hipEvent_t done_ev = nullptr;     // the arena's responses are in r_pinned
  const uint32_t* q_pinned_dev = nullptr;  // q_pinned as the device addresses it (a lone query is read in place)
  uint32_t* fill_progress = nullptr;       // in the pinned block: steps of a lone query copied so far (the kernel polls it)
  const uint32_t* fill_progress_dev = nullptr;
  uint32_t* handed = nullptr;              // in the pinned block: sequence number of the lone response last handed over (respond_hand_over_kernel)
  uint32_t* handed_dev = nullptr;
  uint32_t hand_seq = 0;                   // (guarded like r0_zero: one lone caller per arena at a time)
  bool r0_zero = false;                    // seat 0 of r_dev holds zeros (guarded by the arena's own leader: one at a time)
  uint32_t r_zero_words = 0;               // ... and so do this many words from r_dev[0] on (what the last hand-over kernel left behind)
  // (servers with a slot map) the seats of this round hold COMPACT queries -- every caller compacts its query onto the kept slots while it
  // stages it, uploads 8/9 of the words into q_compact, and the launch needs no map -- or whole ones (DMA straight from page-locked caller
  // buffers, the kernel applies the map).  Decided by the caller that opens the arena, the same for all its seats; guarded by Server::mu.
  bool compact_seats = false;
  // (a few concurrent callers) the seats of this round are not uploaded at all: ONE pass of the
  // step-major kernel reads every seat's query IN PLACE over the host link, each from its caller's own buffer -- the upload (83 us per query
  // at 2^20 keys x 1 kB, one after the other) disappears behind the stream of the database, as it does for a lone caller -- or, a pageable
  // query, from the seat's part of the pinned block while its caller copies it in (respond_in_round).  seat_q[i] is the device-visible
  // address of word 0 of seat i's query.  Decided by the caller that opens the arena; guarded by Server::mu.
  bool in_place = false;
  const uint32_t* seat_q[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool seat_polled[8] = {false, false, false, false, false, false, false, false};  // seat i's query is pageable: copied into the pinned block under the pass
  uint32_t rounds_done = 0;  // (atomic accesses) bumped when a round's responses are in r_pinned: what the followers of an in-place round spin on
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
  cpir_dtc_layout layout{};  // the LOGICAL database of this server / shard: num_slots = the slots of the query it answers for (what the C ABI
                             // reports, what import / export speak, what the staging of host queries is sized by)
  cpir_dtc_layout phys{};    // the image in `dtc`.  Equal to `layout` unless only the slots with a non-zero row are served (compact.hip):
                             // then its num_slots is map.n_kept and queries are compacted through `map` in front of every launch
  SlotMap map;               // active(): the kept slots of this shard, relative to slot_offset
  uint32_t* dtc = nullptr;  // device, phys.total_words u32
  uint64_t slot_offset = 0;
  uint64_t total_slots = 0;
  double setup_timings[CPIR_SETUP_TIMING_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};

  // 3 arenas of 8 seats: a batch of up to 8 rides ONE stream of the database on the matrix cores (a respond kernel takes about as
  // long for 8 queries as for 1, so throughput is batch size over kernel time); one arena is on the device, one is filling, one spare
  static constexpr uint32_t kSeats = 8;
  static constexpr uint32_t kArenas = 4;
  // An arena takes its first spread() callers freely; further callers prefer to open another arena (so that one arena's uploads overlap
  // another's kernel) and fill the seats beyond only once no arena is free.  With T callers recently seen inside respond at the same
  // time, a kernel of K us (about the same for 1..4 queries) and u us of upload per query: two arenas of T/2 that alternate between
  // uploading and computing cost max(2K/T, u) per query, one arena of T costs u + K/T -- so spread() is T/2 where K < T*u (2^20 keys x
  // 1 kB: 8 callers 4 + 4, 4 callers 2 + 2 -- a fixed spread of 4 left four callers in 3 + 1: 7.0 instead of 8.0 k queries/s) and T where
  // the kernel outweighs the uploads (8 kB values: 8 callers in ONE pass).  K and u are estimated from the layout: resident bytes at
  // 6.8 TB/s, query bytes at 55 GB/s + 15 us.
  uint32_t inside = 0, peak_inside = 1, calls_below_peak = 0;  // guarded by mu
  void caller_enters() {
    inside++;
    if (inside >= peak_inside) {
      peak_inside = inside, calls_below_peak = 0;
    } else if (++calls_below_peak >= 8) {  // the peak decays when the callers have gone: one step per 8 calls below it
      peak_inside = inside > peak_inside - 1 ? inside : peak_inside - 1, calls_below_peak = 0;
    }
  }
  bool kernel_outweighs_uploads() const {  // K >= T * u: one arena for all recent callers
    const double k_us = (double)phys.total_words * 4 / 6.8e6, u_us = (double)layout.num_slots * 4 / 55e3 + 15;
    return k_us >= peak_inside * u_us;
  }
  // how long a leader whose gate is open waits for the company spread() promises: only where one pass is to answer ALL recent callers
  // (they come back within tens of microseconds of each other, and the first one back would otherwise launch alone: 8 kB values, 4
  // callers: passes of 1 and 3 alternating, 1 270 queries/s; with the moment's wait passes of 4, 1 960).  A tenth of the kernel's time, at
  // most 100 us.  Where two arenas alternate the device is busy when a leader is ready, which is wait enough (and waiting on top of it
  // cost 3 callers at 2^20 keys x 1 kB a fifth of their throughput).
  double batching_window_seconds() const {
    if (!kernel_outweighs_uploads()) return 0;
    const double w = (double)phys.total_words * 4 / 6.8e12 * 0.1;
    return w < 100e-6 ? w : 100e-6;
  }
  uint32_t spread() const {
    const uint32_t s = kernel_outweighs_uploads() ? peak_inside : (peak_inside + 1) / 2;
    return s < 1 ? 1 : (s > kSeats ? kSeats : s);
  }
  // CPIR_RESPOND_TRACE=1: per-phase wall time of the host path, printed when the server is destroyed (diagnosis)
  struct Trace {
    std::atomic<uint64_t> calls{0}, solo{0}, ns_solo{0}, batches{0}, ns_seat{0}, ns_stage{0}, ns_gate{0}, ns_enqueue{0}, ns_gpu{0}, ns_follow{0}, ns_out{0};
    std::atomic<uint64_t> ns_stage_copy{0}, ns_stage_lock{0}, ns_stage_enq{0};  // staging split: host copy / compaction, waiting for the upload lock, the runtime's copy calls
    std::atomic<uint64_t> batch_hist[9] = {};
    // a lone caller whose query is staged under a polled launch: until the jobs are with the helpers / until the launch call is back /
    // until the last job is published / until the response is there
    std::atomic<uint64_t> polled{0}, ns_p_submit{0}, ns_p_launch{0}, ns_p_copied{0}, ns_p_done{0};
    // the leader of an in-place round: from its seat until the round is closed / until the launch calls are back / until its own copy is
    // complete / until the responses are handed over
    std::atomic<uint64_t> rounds{0}, ns_r_close{0}, ns_r_launch{0}, ns_r_copy{0}, ns_r_done{0};
  } trace;
  bool trace_on = false;
  // how the host callers have been served (cpir_server_host_path_counts): always counted, a relaxed add each
  struct Served {
    std::atomic<uint64_t> calls{0}, alone{0}, polled_void{0}, in_uploaded_rounds{0}, uploaded_rounds{0}, in_place_calls{0}, in_place_rounds{0};
  } served;
  std::mutex mu;
  std::condition_variable cv;
  RespondArena arena[kArenas];  // each allocated on first use (a lone caller only ever needs the first)
  hipStream_t up_stream = nullptr;   // the device's host-path streams (Device::up_stream / run_stream), looked up when the first arena is
  hipStream_t run_stream = nullptr;  // created; owned by the device handle, not by this server
  std::atomic<uint32_t> fill_aborts{0};  // lone queries whose polled launch gave up waiting for the copy (3: stop polling)
  std::atomic<uint64_t> fill_polled{0};  // lone pageable queries answered by one launch polling the copy's progress

  // ---- group handle (cpir_server_setup_multi): the database is split along the filter slots over several devices of this
  // process; `shards` then holds one ordinary server per device and this handle owns no packed database itself.  A host query is
  // SCATTERED: device g receives only q[n_g : n_{g+1}] over its own host link, answers its shard, and the C-word partial
  // responses are summed on the host (u32 wrap-around: order-independent, bit-identical to one device).
  std::vector<Server*> shards;
  struct GroupLane {  // per shard, per call context
    std::vector<uint32_t> r;  // the shard's C-word partial response
  };
  struct GroupCtx {
    bool busy = false;
    std::vector<GroupLane> lanes;
  };
  // The same handle asked on DEVICE pointers (cpir_server_respond_device / _batch_device on a group): q and r live on the device of shard 0
  // (the root); every shard's stream copies its slots of the queries over the peer link, answers them, and copies its C-word partial
  // responses into the root's table; a kernel on the caller's stream, behind one event per shard, adds the table up.  Nothing touches the
  // host and no collective library is involved -- xGMI peer copies and stream-ordered events only.
  struct GroupDevCtx {
    bool ready = false, used = false;
    std::vector<hipStream_t> stream;   // per shard, on its device
    std::vector<hipEvent_t> ev;        // per shard: its partial responses are in the root's table
    std::vector<uint32_t*> buf;        // per shard, one device block: [kBatchCap x slots] queries, [kBatchCap x C] responses, (slot map) compact queries
    uint32_t* partials = nullptr;      // root: shards x kBatchCap x C
    hipEvent_t in_ev = nullptr, done_ev = nullptr;  // root: the queries are ready / the table has been summed (it may be overwritten)
  };
  static constexpr uint32_t kBatchCap = 48;  // queries per round of a device-resident group call: two wide passes of 24 per shard (larger batches go round by round)
  static constexpr int kGroupCtx = 4;  // concurrent callers served at once; further callers wait
  GroupDevCtx gdev[kGroupCtx];
  uint32_t gdev_next = 0;
  std::mutex gdev_mu;  // one enqueue sequence at a time
  GroupCtx gctx[kGroupCtx];
  bool gctx_ready = false;
  // Every shard is an ordinary server (its slots of the query start at slot_offset) and is asked through its own cpir_server_respond: its
  // slots of the query are read in place over ITS device's host link, or copied in under a polled launch, and concurrent callers of the
  // group meet again inside every shard, in its rounds and arenas.  kGroupCtx persistent host threads per shard make those calls, so the
  // shards of one query work side by side and the queries of several callers reach a shard together.
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
  struct GroupWorker {  // one queue per shard, served by kGroupCtx threads
    std::vector<std::thread> ths;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<GroupJob> jobs;
    bool stop = false;
  };
  std::vector<std::unique_ptr<GroupWorker>> workers;
};


inline double now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// host_respond.hip
void device_retain(Device* d);
void device_release(Device* d);
Server* server_new(Device* dev, const cpir_dtc_layout& L, uint64_t slot_offset, uint64_t total_slots);
void server_destroy(Server* srv);
// any batch size on device pointers: with batch fusion every pass answers up to 24 queries (the wide pass; 12 / 8 with it switched off) from one stream of the database, without it
// every query is its own pass; either way the passes of one kind go into ONE launch
int respond_batched(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len, uint64_t q_slot_offset,
                    uint32_t batch, uint32_t* r, uint32_t* scratch, hipStream_t stream, const uint32_t* keep = nullptr);
// The same for a SERVER: `batch` queries of q_len entries on the device, this server's slots starting at q_slot_offset of each.  With a
// slot map the queries are first gathered onto the kept slots (into qc, batch x map.n_pad words, or a stream-ordered allocation when qc
// is NULL).  lone: batch == 1 answered as cpir_server_respond_device does (one launch, no batching logic).
int server_respond_on_device(const Server* srv, const uint32_t* q, uint64_t q_len, uint64_t q_slot_offset, uint32_t batch, bool lone, uint32_t* r,
                             uint32_t* scratch, uint32_t* qc, hipStream_t stream);
// after the image is packed: adopt `map` (moved from) and the physical layout it implies
void server_set_physical(Server* srv, const cpir_dtc_layout& phys, SlotMap* map);

// host_setup.hip
// slots per shard are multiples of this: no packed word of either layout and no 16-byte query piece straddles two shards
uint64_t shard_unit(const cpir_dtc_layout& L);
struct DevBuf {  // scoped device allocation
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)CPIR_HIP_FREE(p);
  }
  void dispose_async(int ordinal);  // free in the background (on device `ordinal`) instead of at scope exit
};

}  // namespace cpir

struct cpir_device : cpir::Device {};
struct cpir_server : cpir::Server {};
struct cpir_xof : cpir::TurboShake128 {};
