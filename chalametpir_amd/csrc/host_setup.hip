// host_setup.hip -- Server::setup (reference chalametpir_server/src/server.rs:47-78, 103-167): the expansion of the public matrix A on a
// host thread streamed into HBM, upload + packing of D, the hint matmul in row chunks as A arrives, the N-sharded group of devices behind
// one handle, the key-value front end (filter construction + row encoding on the host), and the extern "C" constructors of a server.
#include "server_internal.hpp"

namespace cpir {

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
                          (void)pthread_setname_np(pthread_self(), "cpir-dispose");
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
        if (t.copy_stream) device_stream_release(t.dev, t.copy_stream);
        if (t.A_dev) (void)CPIR_HIP_FREE(t.A_dev);
      }
      device_release(t.dev);
    }
    for (int i = 0; i < 2; i++)
      if (pinned_[i]) (void)CPIR_HIP_HOST_FREE(pinned_[i]);
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
      CPIR_HIP_TRY(CPIR_HIP_MALLOC(&t.A_dev, (size_t)rows * t.col_n * 4));
      if (!(t.copy_stream = device_stream_acquire(t.dev))) return CPIR_ERR_HIP;
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
    for (int i = 0; i < 2; i++) CPIR_HIP_TRY(CPIR_HIP_HOST_MALLOC(&pinned_[i], (size_t)rows_per_block_ * N_ * 4, hipHostMallocPortable));
    memcpy(seed_, seed, 32);
    worker_ = std::thread([this] {
      (void)pthread_setname_np(pthread_self(), "cpir-xof");
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

void DevBuf::dispose_async(int ordinal) {
  void* q = p;
  p = nullptr;
  if (q)
    g_disposer.run([q, ordinal] {
      DeviceGuard g(ordinal);
      (void)CPIR_HIP_FREE(q);
    });
}

// Packs D (L.num_slots x C on the device, leading dimension ldd) into srv->dtc (allocated here).  With an active `map` (the rows of D that
// have a non-zero field: build_slot_map, compact.hip) only those rows are packed -- by the pack kernel itself through the map (planar packing), else gathered into a stream-ordered temporary first -- and the
// server adopts the map and the compact layout; hi_plane (the hint matmul's second operand plane, written by the pack pass) only goes with an
// uncompacted image.  Enqueues on `stream`; the caller synchronises.
static int pack_into_server(Device* dev, Server* srv, const uint32_t* D_dev, uint64_t ldd, const cpir_dtc_layout& L, SlotMap* map, uint32_t* flag_dev,
                            void* hi_plane, hipStream_t stream) {
  cpir_dtc_layout P = L;
  uint32_t* Dc = nullptr;
  bool in_kernel = false;  // the pack kernel applies the map itself
  if (map->active()) {
    if (hi_plane) return CPIR_ERR_INVALID_ARGUMENT;
    CPIR_TRY(dtc_layout_for_packing(map->n_kept, L.num_cols, L.mat_elem_bit_len, L.packing, &P));
    in_kernel = transpose_compress_takes_slot_map(P);
    if (!in_kernel) {
      // (the other two packings) the kept rows are gathered into a temporary of (almost) D's size while D is still resident: where the
      // device has no room for that -- a database that fits beside its image but not twice -- the map is dropped and the whole matrix
      // packed as it is.  Compaction is an optimisation, never a reason for setup to fail.
      const int as = scratch_acquire(reinterpret_cast<void**>(&Dc), (size_t)map->n_kept * L.num_cols * 4, stream);
      if (as != CPIR_OK) {
        if (as != CPIR_ERR_OUT_OF_DEVICE_MEMORY) return as;
        Dc = nullptr;
        map->reset();
        P = L;
      }
    }
  }
  const hipError_t de = CPIR_HIP_MALLOC(&srv->dtc, (size_t)P.total_words * 4);
  if (de != hipSuccess) {
    set_last_hip_error(de, "hipMalloc(dtc)", __FILE__, __LINE__);
    if (Dc) (void)scratch_release_after(Dc, stream);
    return de == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP;
  }
  int st;
  if (map->active() && in_kernel) {
    // planar packing: the pack kernel reads row keep[n] of D for slot n of the image -- no gathered copy of D, no second D's worth of memory
    st = launch_transpose_compress(dev, D_dev, ldd, P, srv->dtc, flag_dev, stream, nullptr, map->keep_dev);
  } else if (map->active()) {
    st = launch_gather_rows(dev, D_dev, ldd, *map, L.num_cols, Dc, stream);
    if (st == CPIR_OK) st = launch_transpose_compress(dev, Dc, L.num_cols, P, srv->dtc, flag_dev, stream, nullptr);
    const int st2 = scratch_release_after(Dc, stream);
    if (st == CPIR_OK) st = st2;
  } else {
    st = launch_transpose_compress(dev, D_dev, ldd, L, srv->dtc, flag_dev, stream, hi_plane);
  }
  if (st == CPIR_OK) server_set_physical(srv, P, map);
  return st;
}

// The matrix half of setup once D sits on the host: upload D, pack it, wait for A, one matmul, hint back.
static int setup_from_host_matrix(Device* dev, PublicMatrixUpload& upA, const uint32_t* D, uint64_t N, uint32_t C, uint32_t b,
                                  uint32_t* hint_out, Server** out) {
  cpir_dtc_layout L;
  CPIR_TRY(dtc_layout_for(N, C, b, &L));
  DeviceGuard g(dev->ordinal);
  hipStream_t stream = dev->stream;
  DevBuf D_dev, flag, M_dev;
  CPIR_HIP_TRY(CPIR_HIP_MALLOC(&D_dev.p, (size_t)N * C * 4));
  CPIR_HIP_TRY(CPIR_HIP_MALLOC(&flag.p, 4));
  CPIR_HIP_TRY(CPIR_HIP_MALLOC(&M_dev.p, (size_t)CPIR_LWE_DIMENSION * C * 4));
  Server* srv = server_new(dev, L, 0, N);
  auto fail = [&](int st) { server_destroy(srv); return st; };
#define TRY_(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_last_hip_error(_e, #e, __FILE__, __LINE__); \
    return fail(_e == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP); } } while (0)
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
  const uint64_t hi_bytes = planar_hi_plane_bytes(L);  // 0 with one bit plane (b = 9): the matmul expands it from the image itself
  // Rows of D without a non-zero field (the slots of a real encoded database that no key owns: 11 % of them at arity 3) are left out of
  // the resident image where that pays (compact.hip); the hint is still computed from D as it is, all rows and unmasked (server.rs:61),
  // through the byte-plane split of the whole matrix -- an image that lacks rows cannot serve as the matmul's right-hand side.
  SlotMap map;
  uint32_t ored_rows = 0;
  int st = CPIR_OK;
  if (compact_slots_mode() != 0) {
    st = build_slot_map(dev, (const uint32_t*)D_dev.p, C, N, C, b, stream, &map, &ored_rows);
    if (st != CPIR_OK) return fail(st);
  }
  bool planar_rhs = !map.active() && mfma_matmul_enabled() && mfma_planar_rhs_applicable(A_dev, N, L);
  if (planar_rhs) {
    if (hi_bytes) TRY_(CPIR_HIP_MALLOC(&hi_plane.p, (size_t)hi_bytes));
    TRY_(CPIR_HIP_MALLOC(&rowsum_ws.p, 4 * 128));
  }
  st = pack_into_server(dev, srv, (const uint32_t*)D_dev.p, C, L, &map, (uint32_t*)flag.p, hi_plane.p, stream);
  if (st != CPIR_OK) return fail(st);
  uint32_t ored = 0;
  TRY_(hipMemcpyAsync(&ored, flag.p, 4, hipMemcpyDeviceToHost, stream));
  TRY_(hipStreamSynchronize(stream));
  ored |= ored_rows;  // (the pack pass of a compacted image saw the kept rows only)
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
    TRY_(CPIR_HIP_MALLOC(&rhs.p, (size_t)mfma_rhs_workspace_bytes(N, C, chunk)));
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
uint64_t shard_unit(const cpir_dtc_layout& L) {
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
    uint32_t ored = 0, ored_rows = 0;
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
    TRY_(CPIR_HIP_MALLOC(&work[g].D_dev.p, (size_t)(hi - lo) * C * 4));
    TRY_(CPIR_HIP_MALLOC(&work[g].flag.p, 4));
    TRY_(CPIR_HIP_MALLOC(&work[g].M_dev.p, (size_t)CPIR_LWE_DIMENSION * C * 4));
    hipStream_t stream = devs[g]->stream;
    TRY_(hipMemcpyAsync(work[g].D_dev.p, D + lo * C, (size_t)(hi - lo) * C * 4, hipMemcpyHostToDevice, stream));
    TRY_(hipMemsetAsync(work[g].flag.p, 0, 4, stream));
    // (as setup_from_host_matrix: where the packed image can serve as the matmul's right-hand side, the pack pass prepares it; A's slab
    // for this shard is allocated 16-byte aligned with leading dimension hi - lo)
    const uint64_t hi_bytes = planar_hi_plane_bytes(L);
    // (rows without a non-zero field are left out of a shard's image where that pays, as in setup_from_host_matrix; such a shard multiplies
    // its slab of A by the unpacked rows.  Finding them synchronises this device's stream: the shards' uploads no longer overlap there)
    SlotMap map;
    if (compact_slots_mode() != 0) {
      st = build_slot_map(devs[g], (const uint32_t*)work[g].D_dev.p, C, hi - lo, C, b, stream, &map, &work[g].ored_rows);
      if (st != CPIR_OK) return fail(st);
    }
    if (!map.active() && mfma_matmul_enabled() && L.packing == CPIR_PACK_PLANAR && L.mat_elem_bit_len >= 9 && (hi - lo) % 4 == 0 && mfma_pipeline() != 0) {
      if (hi_bytes) TRY_(CPIR_HIP_MALLOC(&work[g].hi_plane.p, (size_t)hi_bytes));  // (one bit plane: none, the matmul expands it from the image)
      TRY_(CPIR_HIP_MALLOC(&work[g].rowsum_ws.p, 4 * ((CPIR_LWE_DIMENSION + 127) / 128 * 128)));
    }
    st = pack_into_server(devs[g], child, (const uint32_t*)work[g].D_dev.p, C, L, &map, (uint32_t*)work[g].flag.p, work[g].hi_plane.p, stream);
    if (st != CPIR_OK) return fail(st);
    TRY_(hipMemcpyAsync(&work[g].ored, work[g].flag.p, 4, hipMemcpyDeviceToHost, stream));
  }
  uint32_t ored = 0;
  for (size_t g = 0; g < G; g++) {
    DeviceGuard dg(devs[g]->ordinal);
    TRY_(hipStreamSynchronize(devs[g]->stream));
    ored |= work[g].ored | work[g].ored_rows;
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
    if (work[g].rowsum_ws.p && !child->map.active() && (ored >> b) == 0 && mfma_planar_rhs_applicable(A_dev, n, child->layout))
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


}  // namespace cpir

using namespace cpir;

extern "C" {

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
  hipStream_t s = pick_stream(dev, stream);
  SlotMap map;  // rows without a non-zero field are left out of the image where that pays (compact.hip)
  int st = compact_slots_mode() != 0 ? build_slot_map(dev, D_dev, ldd, N_shard, C, b, s, &map, nullptr) : CPIR_OK;
  if (st == CPIR_OK) st = pack_into_server(dev, srv, D_dev, ldd, L, &map, nullptr, nullptr, s);
  if (st == CPIR_OK) {
    const hipError_t e = hipStreamSynchronize(s);
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
  CPIR_HIP_TRY(CPIR_HIP_MALLOC(&src.p, src_bytes));
  Server* srv = server_new(dev, L, 0, N);
  auto fail = [&](int st) { server_destroy(srv); return st; };
  hipError_t e = hipMemcpyAsync(src.p, compressed, src_bytes, hipMemcpyHostToDevice, dev->stream);
  if (e != hipSuccess) { set_last_hip_error(e, "hipMemcpyAsync", __FILE__, __LINE__); return fail(CPIR_ERR_HIP); }
  // an imported database is served without its empty rows like one that was set up here (compact.hip): the OR over the compressed rows
  // says which slots hold something, the kept slots' fields are gathered into a compressed matrix of their own, and THAT is imported
  SlotMap map;
  int st = build_slot_map_from_compressed(dev, (const uint32_t*)src.p, L.words_per_row, N, C, b, L.compression_factor, dev->stream, &map);
  if (st != CPIR_OK) return fail(st);
  cpir_dtc_layout P = L;
  DevBuf compact;
  if (map.active()) {
    st = dtc_layout_for_packing(map.n_kept, C, b, L.packing, &P);
    if (st != CPIR_OK) return fail(st);
    e = CPIR_HIP_MALLOC(&compact.p, (size_t)C * P.words_per_row * 4);
    if (e == hipErrorOutOfMemory) {  // no room for a second (compact) copy beside the source: import the whole matrix, serve every slot
      (void)hipGetLastError();
      compact.p = nullptr;
      map.reset();
      P = L;
    } else if (e != hipSuccess) {
      set_last_hip_error(e, "hipMalloc(compact compressed matrix)", __FILE__, __LINE__);
      return fail(CPIR_ERR_HIP);
    } else {
      st = launch_gather_compressed(dev, (const uint32_t*)src.p, L.words_per_row, map, C, L.compression_factor, P.words_per_row, (uint32_t*)compact.p, dev->stream);
      if (st != CPIR_OK) return fail(st);
    }
  }
  e = CPIR_HIP_MALLOC(&srv->dtc, (size_t)P.total_words * 4);
  if (e != hipSuccess) { set_last_hip_error(e, "hipMalloc(dtc)", __FILE__, __LINE__); return fail(CPIR_ERR_OUT_OF_DEVICE_MEMORY); }
  st = launch_dtc_import(dev, map.active() ? (const uint32_t*)compact.p : (const uint32_t*)src.p, P, srv->dtc, dev->stream);
  if (st != CPIR_OK) return fail(st);
  server_set_physical(srv, P, &map);
  e = hipStreamSynchronize(dev->stream);
  if (e != hipSuccess) { set_last_hip_error(e, "hipStreamSynchronize", __FILE__, __LINE__); return fail(CPIR_ERR_HIP); }
  *out = static_cast<cpir_server*>(srv);
  return CPIR_OK;
}


}  // extern "C"
