// Internal declarations shared by the translation units of libchalamet_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/chalamet_hip.h"

namespace cpir {

// ---- error plumbing --------------------------------------------------------------------------
void set_last_hip_error(hipError_t e, const char* what, const char* file, int line);

#define CPIR_HIP_TRY(expr)                                                    \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      ::cpir::set_last_hip_error(_e, #expr, __FILE__, __LINE__);              \
      return (_e == hipErrorOutOfMemory) ? CPIR_ERR_OUT_OF_DEVICE_MEMORY      \
             : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? CPIR_ERR_NO_DEVICE \
                                                                       : CPIR_ERR_HIP;      \
    }                                                                         \
  } while (0)

#define CPIR_TRY(expr)            \
  do {                            \
    int _s = (expr);              \
    if (_s != CPIR_OK) return _s; \
  } while (0)

// ---- allocation journal (diagnosis) ------------------------------------------------------------
// Every device / page-locked allocation and release of the library goes through CPIR_HIP_MALLOC / _FREE / _HOST_MALLOC / _HOST_FREE and is noted in a small ring (what, address, bytes, where, thread); the
// fatal-signal handler (capi.hip, CPIR_ABORT_BACKTRACE=1) prints the tail of it, so that the address in a "Memory access fault by GPU"
// line of the runtime can be matched to the block it fell into -- or used to fall into.  A few nanoseconds per call.
void journal_note(const char* what, const void* p, size_t bytes, const char* file, int line);
void journal_dump(int fd);  // async-signal-safe enough for a handler that is about to die

inline hipError_t traced_hipMalloc(void** p, size_t n, const char* f, int l) {
  const hipError_t e = hipMalloc(p, n);
  journal_note(e == hipSuccess ? "hipMalloc" : "hipMalloc FAILED", p ? *p : nullptr, n, f, l);
  return e;
}
inline hipError_t traced_hipHostMalloc(void** p, size_t n, unsigned flags, const char* f, int l) {
  const hipError_t e = hipHostMalloc(p, n, flags);
  journal_note(e == hipSuccess ? "hipHostMalloc" : "hipHostMalloc FAILED", p ? *p : nullptr, n, f, l);
  return e;
}
inline hipError_t traced_hipFree(void* p, const char* f, int l) {
  journal_note("hipFree", p, 0, f, l);
  return hipFree(p);
}
inline hipError_t traced_hipHostFree(void* p, const char* f, int l) {
  journal_note("hipHostFree", p, 0, f, l);
  return hipHostFree(p);
}
#define CPIR_HIP_MALLOC(p, n) ::cpir::traced_hipMalloc(reinterpret_cast<void**>(p), (n), __FILE__, __LINE__)
#define CPIR_HIP_HOST_MALLOC(p, n, flags) ::cpir::traced_hipHostMalloc(reinterpret_cast<void**>(p), (n), (flags), __FILE__, __LINE__)
#define CPIR_HIP_FREE(p) ::cpir::traced_hipFree((p), __FILE__, __LINE__)
#define CPIR_HIP_HOST_FREE(p) ::cpir::traced_hipHostFree((p), __FILE__, __LINE__)

// ---- device context ---------------------------------------------------------------------------
struct Device {
  std::atomic<int> refs{1};
  int ordinal = 0;
  int num_cus = 0;
  hipStream_t stream = nullptr;  // the handle's own stream: used by the synchronous host-pointer entry points
  // Streams live as long as the device handle, not as long as a server: servers come and go (a test suite creates hundreds), and the
  // runtime's stream life cycle -- queues, completion handlers on its own thread -- is not something to exercise once per server.
  //   * up_stream / run_stream: the two streams of the host path of Server::respond (every query upload, FIFO / every batched respond and
  //     response download, FIFO, highest priority), shared by all servers on this device, created on first use; upload_mu / launch_mu
  //     keep one query's upload pieces / one arena's launch sequence together on them;
  //   * idle_streams: a pool of plain non-blocking streams for the transient users (the upload of A in setup, the lanes of a group).
  std::mutex pool_mu;
  std::vector<hipStream_t> idle_streams;
  static constexpr int kUpStreams = 4;
  hipStream_t up_stream = nullptr, run_stream = nullptr;
  hipStream_t up_more[kUpStreams - 1] = {nullptr, nullptr, nullptr};  // further upload streams (tuning "respond.upload_streams")
  uint32_t up_turn = 0;  // (guarded by upload_mu) queries take the upload streams in turn: one's copy is set up while another's crosses the link
  std::mutex upload_mu, launch_mu;
};
// host_respond.hip
int device_host_streams(Device* d);             // creates up_stream / run_stream on first use (caller has made the device current)
hipStream_t device_stream_acquire(Device* d);   // a drained non-blocking stream from the pool (created if the pool is empty); nullptr + last error
void device_stream_release(Device* d, hipStream_t s);  // waits for the stream to drain and returns it to the pool

// RAII "make this device current for the calling thread"
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int ordinal) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == ordinal) || (hipSetDevice(ordinal) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// `stream` arguments of the C ABI are hipStream_t values; NULL is HIP's legacy default (null) stream -- which is also
// what torch.cuda.current_stream().cuda_stream is when no stream context is active -- never "some other stream".
inline hipStream_t pick_stream(const Device*, void* stream) { return reinterpret_cast<hipStream_t>(stream); }

// Diagnosis code (per-block timing traces, switches that skip part of a kernel's work to time the rest) exists only in a build with
// -DCPIR_DIAG (`make diag`: lib/diag/libchalamet_hip.so, for scripts/wide_ablate.py and the pmc_*.sh passes).  The release library has no
// code path that skips work or that a variable of the environment could switch on: Server::respond has no mode in which it lies.
#ifdef CPIR_DIAG
#define CPIR_DIAG_ONLY(...) __VA_ARGS__
#else
#define CPIR_DIAG_ONLY(...)
#endif

// ---- kernel launchers (defined in the .hip files) ---------------------------------------------
// respond.hip
uint64_t respond_scratch_words(const cpir_dtc_layout& L, uint32_t batch);
// One launch = `passes` independent passes over the database, each answering `batch` (1, 2 or 4) queries that share the
// stream of that pass: q holds passes*batch queries of q_len entries, r passes*batch responses of num_cols entries.
int launch_respond(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                   uint64_t q_slot_offset, uint32_t batch, uint32_t passes, uint32_t* r, uint32_t* scratch, hipStream_t stream,
                   const uint32_t* keep = nullptr);
// keep: the slot map of a compacted image, applied by the kernel itself -- the wide kernel only (respond_batch_takes_slot_map says whether the
// launches of a batch are); with any other kernel the call fails with CPIR_ERR_INVALID_ARGUMENT and the caller gathers the queries first
bool respond_batch_takes_slot_map(const cpir_dtc_layout& L, uint32_t batch, bool lone, uint64_t q_len);
// A launch whose query is still being copied into page-locked host memory while the kernel runs (respond_planar.hip): the host counts
// the 512-slot steps of q in place so far in *progress (host memory, device-visible address), front to back; the kernel waits for each
// step it needs, at most timeout_us per wave, and sets *abort_flag (device memory, zeroed by the caller) if a wave gave up.
constexpr uint32_t CPIR_FILL_LINES = 64;  // the count is kept in this many copies, one per 64-byte line (16 words apart): block b polls copy b % 64
struct PlanarHostFill {
  const uint32_t* progress;
  uint32_t* abort_flag;
  uint32_t timeout_us;
  uint32_t seats = 1;  // queries of the pass that are being copied in, each counted in CPIR_FILL_LINES copies of its own, one set behind the other
                       // (the queries of a round of concurrent host callers, each copied by its caller's thread)
};
// one query whose words are read exactly once (q may therefore live in page-locked host memory: device-visible pointer), r already zero;
// planar packing only, CPIR_ERR_INVALID_ARGUMENT where the step-major kernel does not apply
int launch_respond_read_once(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                             uint64_t q_slot_offset, uint32_t* r_prezeroed, hipStream_t stream, uint64_t step_lo = 0, uint64_t step_hi = 0,
                             const PlanarHostFill* fill = nullptr);
// steps [step_lo, step_hi) of 512 slots only (0, 0 = all): a query may be answered by several launches, each over the steps whose
// query words are in place by then; they add up in r
// up to CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS queries answered by ONE pass that reads each of them in place (q_rows: device-visible address of
// word 0 of every query, every one q_len words long); r (batch x C) is zeroed by the call unless the caller says it is zero already
int launch_respond_read_rows_in_place(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* const* q_rows, uint32_t batch,
                                      uint64_t q_len, uint64_t q_slot_offset, uint32_t* r, hipStream_t stream, const PlanarHostFill* fill = nullptr,
                                      bool r_prezeroed = false);
uint32_t respond_inplace_seats();   // tuning "respond.inplace_seats" (0 = off, 2..4)
uint32_t respond_upload_streams();  // tuning "respond.upload_streams" (1..4)
uint32_t respond_helper_spin_us();  // tuning "respond.helper_spin_us"
uint32_t respond_host_fill_timeout_us();  // tuning "respond.host_fill_timeout_us"; 0 = never launch in front of the copy
bool respond_read_once_applicable(const cpir_dtc_layout& L);  // planar packing, LDS room for one response, respond.host_zero_copy on
const char* respond_kernel_name(const cpir_dtc_layout& L);
// respond_planar.hip (CPIR_PACK_PLANAR: the MFMA path); same contract as launch_respond
// the step-major kernel: one A row set (up to 4 queries per pass), slice order; reads every query word once per column window.  in_place:
// launched as the in-place host path needs it (whole steps round-robin over the blocks, far-mode fragment schedule; one column window or
// CPIR_ERR_INVALID_ARGUMENT); r_prezeroed: the caller has zeroed r already; [step_lo, step_hi): see launch_respond_read_once.
constexpr uint32_t CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS = 4;
int launch_respond_planar_ks(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                             uint64_t q_slot_offset, uint32_t batch, uint32_t passes, uint32_t* r, hipStream_t stream, int blocks_per_cu,
                             bool nontemporal, bool xcd_split, bool in_place, bool r_prezeroed, uint64_t step_lo, uint64_t step_hi,
                             const PlanarHostFill* fill, const uint32_t* const* q_rows = nullptr);
// q_rows (then q == NULL, one pass): the address of word 0 of each of the `batch` queries, wherever each lies -- the queries of concurrent
// host callers read in place, each from its caller's page-locked buffer (device-visible addresses)
uint32_t respond_planar_pass_width(const cpir_dtc_layout& L, uint32_t batch);  // respond.hip: queries per pass of a fused batch under the current tuning
// the wide kernel: one 8-wave block per CU, up to 24 queries (six A row sets, looped) per stream of the database, any number of passes
constexpr uint32_t CPIR_PLANAR_WIDE_MAX_QUERIES_PER_PASS = 24;
int launch_respond_planar_wide(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                               uint64_t q_slot_offset, uint32_t batch, uint32_t passes, uint32_t* r, hipStream_t stream, bool nontemporal,
                               bool xcd_split, int interleave, const uint32_t* keep = nullptr);
// interleave: order of the passes of one launch -- 0 slice order, 1 interleaved, -1 by the image's size (planar_passes_interleaved)
// keep: the slot map of a compacted image (SlotMap::keep_dev) -- the kernel then gathers the query words itself; L is the PHYSICAL layout
bool planar_passes_interleaved(const cpir_dtc_layout& L, uint32_t passes, int interleave);
bool respond_batch_fusion();
uint64_t respond_multi_pass_limit_bytes();

// pack.hip
int launch_transpose_compress(const Device* dev, const uint32_t* D, uint64_t ldd, const cpir_dtc_layout& L, uint32_t* dtc,
                              uint32_t* or_of_entries, hipStream_t stream, void* hi_plane = nullptr, const uint32_t* keep = nullptr);
// keep: a slot map applied by the pack kernel itself (L = the COMPACT layout, slot n of the image = row keep[n] of D; device memory, at
// least L.num_slots entries) -- where transpose_compress_takes_slot_map(L); else the caller gathers the rows first
bool transpose_compress_takes_slot_map(const cpir_dtc_layout& L);
// planar packing: hi_plane (planar_hi_plane_bytes(L) bytes, 16-byte aligned; 0 bytes for b <= 8) also receives the byte (field >> 8) XOR
// 0x80 of every field as 1 KiB MFMA operand pieces [column tile of 16][k-block of 64 slots] -- with the low-byte pieces of the image
// itself the right-hand side of the hint matmul, so that Server::setup reads D once (launch_mat_x_mat_mfma_planar)
uint64_t planar_hi_plane_bytes(const cpir_dtc_layout& L);
const char* pack_kernel_name(const cpir_dtc_layout& L);
int pack_rows_mode();
void set_pack_rows_mode(int m);
int launch_dtc_import(const Device* dev, const uint32_t* compressed, const cpir_dtc_layout& L, uint32_t* dtc, hipStream_t stream);
int launch_dtc_export(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, uint32_t* compressed, hipStream_t stream);

// compact.hip: serving only the slots whose row of D has a non-zero field (a real encoded database leaves N - n rows all zero)
struct SlotMap {
  uint32_t* keep_dev = nullptr;     // device: compact index -> slot (relative to the first slot of this database / shard), increasing;
                                    // n_pad entries, the padding holds 0xFFFFFFFF
  std::vector<uint32_t> keep_host;  // the same, n_kept entries (a lone host query is compacted on the host while it is staged)
  std::vector<uint8_t> keep_bits;   // the same as a bitmap over the n_orig slots (+ 8 bytes of padding): the streaming form of that compaction
  uint64_t n_kept = 0;              // 0: no map, every slot is served
  uint64_t n_pad = 0;               // stride and length of a compact query: n_kept rounded up to 128 words
  uint64_t n_orig = 0;              // slots the map selects from
  bool active() const { return n_kept != 0; }
  void reset();
  // owns keep_dev: a map that is dropped on an error path (between build_slot_map and the server adopting it) frees its device memory
  SlotMap() = default;
  ~SlotMap() { reset(); }
  SlotMap(const SlotMap&) = delete;
  SlotMap& operator=(const SlotMap&) = delete;
};
int compact_slots_mode();  // tuning "layout.compact_slots": 0 never, 1 (default) where at least 1/32 of the rows are zero, 2 whenever a row is
void set_compact_slots_mode(int m);
int build_slot_map(const Device* dev, const uint32_t* D_dev, uint64_t ldd, uint64_t N, uint32_t C, uint32_t b, hipStream_t stream, SlotMap* map,
                   uint32_t* or_of_entries_host);
int slot_map_from_flags(const std::vector<uint8_t>& flags, uint64_t N, hipStream_t stream, SlotMap* map);
int build_slot_map_from_compressed(const Device* dev, const uint32_t* src_dev, uint64_t W, uint64_t N, uint32_t C, uint32_t b, uint32_t cf,
                                   hipStream_t stream, SlotMap* map);  // (cpir_server_from_compressed: the database arrives in the reference's form)
int launch_gather_compressed(const Device* dev, const uint32_t* src_dev, uint64_t W, const SlotMap& map, uint32_t C, uint32_t cf, uint64_t Wc,
                             uint32_t* out, hipStream_t stream);
int launch_gather_rows(const Device* dev, const uint32_t* D_dev, uint64_t ldd, const SlotMap& map, uint32_t C, uint32_t* out, hipStream_t stream);
int launch_gather_query(const Device* dev, const uint32_t* q, uint64_t q_len, uint64_t q_slot_offset, const SlotMap& map, uint32_t batch,
                        uint32_t* out, hipStream_t stream);
int launch_expand_ref(const Device* dev, const uint32_t* compact_ref, uint64_t Wc, const SlotMap& map, uint64_t W, uint32_t C, uint32_t cf,
                      uint32_t* out, hipStream_t stream);
// host_gather.cpp: dst[i] = src[idx[i]] on the host (AVX-512 / AVX2 gathers where the CPU has them)
void gather_words(uint32_t* dst, const uint32_t* src, const uint32_t* idx, size_t count);
const char* gather_words_variant();
// the same from a bitmap of the kept slots (bit s of bits[]: slot s is kept; padded by 8 readable bytes): the words src[s], s in
// [s_lo, s_hi), whose bit is set, in order; returns how many.  The AVX-512 form streams; without it callers use gather_words.
bool compress_words_vectorised();
size_t compress_words(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo, size_t s_hi);
// the same for a destination that is cold in the caches and read by the device next (non-temporal stores where the CPU and dst's alignment allow)
size_t compress_words_streaming(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo, size_t s_hi);

// matmul.hip
int launch_mat_x_mat(const Device* dev, const uint32_t* A, uint64_t lda, const uint32_t* D, uint64_t ldd, uint32_t* M,
                     uint64_t ldm, uint64_t rows, uint64_t inner, uint64_t cols, uint32_t rhs_max_bits, int accumulate,
                     hipStream_t stream);

const char* mat_x_mat_kernel_name(uint32_t rhs_max_bits);
// matmul_mfma.hip: the same product on the i8 matrix cores, for right-hand sides below 2^16 (every encoded database).  The right-hand
// side is prepared ONCE (launch_rhs_split: byte planes in MFMA operand order + column sums, into a caller-owned workspace of
// mfma_rhs_workspace_bytes) and then multiplied by any number of row blocks of A of at most `max_rows` rows each.
bool mfma_matmul_applicable(const uint32_t* A, uint64_t lda, uint64_t inner, uint64_t cols, uint32_t rhs_max_bits);
// The same product with the right-hand side taken from a PLANAR respond image of D (entries < 2^b, b >= 9) and the high-byte plane
// launch_transpose_compress wrote next to it: no separate pass over D.  rowsum_ws: 4 * round_up(rows, 128) bytes of device scratch.
bool mfma_planar_rhs_applicable(const uint32_t* A, uint64_t lda, const cpir_dtc_layout& L);
int launch_mat_x_mat_mfma_planar(const Device* dev, const uint32_t* A, uint64_t lda, const uint32_t* dtc, const cpir_dtc_layout& L,
                                 const void* hi_plane, uint32_t* rowsum_ws, uint32_t* M, uint64_t ldm, uint64_t rows, int accumulate,
                                 hipStream_t stream);
uint64_t mfma_rhs_workspace_bytes(uint64_t inner, uint64_t cols, uint64_t max_rows);
// Device scratch for work enqueued on a stream -- NOT the runtime's stream-ordered allocator.  On this runtime (ROCm 7.2, gfx950) a kernel
// was seen reading a block of hipMallocAsync that had just been recycled on the same stream before the kernel in front of it had written it
// (scripts/probes/mallocasync_order_probe.hip: fill + check kernels alone, 4 in 20 000 iterations; the hint products of a group's shards,
// back to back on one stream: 1 in 100 wrong in every entry -- scripts/probes/matmul_backtoback_repro.cpp, setup_hint_repro.cpp).
// scratch_acquire: a block of the current device for work about to be enqueued on `stream` (refused with CPIR_ERR_INVALID_ARGUMENT while that
// stream is being captured into a graph: the entry points that need scratch are not graph-capturable) -- from a small pool, where a block
// handed back earlier is reused only once the event recorded behind its last user has COMPLETED, else hipMalloc; scratch_release_after: the
// block may be reused or freed once everything enqueued on `stream` so far has completed (an event; a block nobody asks for within half
// a second of it is freed by one background thread) -- the caller may return without synchronising; scratch_drain(ordinal): until no
// released block of that device waits for its event, the idle ones freed.
int scratch_acquire(void** p, size_t bytes, hipStream_t stream);
int scratch_release_after(void* p, hipStream_t stream);
void scratch_drain(int ordinal);

// M[r][c] = 0 for r < rows, c < cols (leading dimension ldm) as a KERNEL on `stream` (one launch whatever ldm is)
int launch_zero_matrix(uint32_t* M, uint64_t ldm, uint64_t rows, uint64_t cols, hipStream_t stream);
// `words` u32 zeroed by that kernel.  Everything a kernel of this library ACCUMULATES into (responses, column / row sums) is zeroed this
// way in front of it, never by hipMemsetAsync: captured into a hipGraph, a memset node in front of the respond kernel was not ordered before
// it on the second and later launches of the first graph a process instantiated (ROCm 7.2, gfx950; scripts/probes/graph_capture_dbg.py:
// the responses lacked the contributions of the blocks that ran before the late memset) -- two kernel nodes are ordered every time
// (tests/test_gpu_stress.py::test_plain_respond_entry_points_can_be_captured_into_a_graph_and_replayed).
inline int zero_words(uint32_t* p, uint64_t words, hipStream_t stream) { return words ? launch_zero_matrix(p, words, 1, words, stream) : CPIR_OK; }
int launch_rhs_split(const Device* dev, const uint32_t* D, uint64_t ldd, uint64_t inner, uint64_t cols, void* workspace, hipStream_t stream);
int launch_mat_x_mat_mfma(const Device* dev, const uint32_t* A, uint64_t lda, const void* workspace, uint64_t inner, uint64_t cols,
                          uint32_t* M, uint64_t ldm, uint64_t rows, uint64_t ws_max_rows, int accumulate, hipStream_t stream);
bool mfma_matmul_enabled();
void set_mfma_matmul(bool on);
int mfma_pipeline();
void set_mfma_pipeline(int on);
#ifdef CPIR_DIAG
int mfma_ablate();
void set_mfma_ablate(int bits);
#endif

// synth.hip
int launch_synth_fill(const Device* dev, uint32_t* out, uint64_t count, uint64_t seed, uint64_t index0, uint32_t mask,
                      hipStream_t stream);

// ---- host-side pieces (plain C++) -------------------------------------------------------------
// host_xof.cpp : TurboSHAKE128 (RFC 9861)
struct TurboShake128 {
  uint64_t s[25];
  unsigned pos;
  TurboShake128();
  void absorb(const uint8_t* in, size_t len);
  void finalize(uint8_t domain_sep = 0x1F);
  void squeeze(uint8_t* out, size_t len);
};
void turboshake128(const uint8_t* msg, size_t len, uint8_t* out, size_t out_len);
const char* xof_permutation_name();

// host_shapes.cpp
unsigned usable_cpus();
uint32_t compression_factor(uint32_t b);
int find_bit_len(uint64_t n, uint32_t* b);
int filter_shape(uint32_t arity, uint64_t n, uint32_t* seg_len, uint32_t* seg_count_len, uint64_t* num_fp);
uint64_t encoded_num_cols(uint64_t max_value_byte_len, uint32_t b);
int dtc_layout_for(uint64_t N, uint32_t C, uint32_t b, cpir_dtc_layout* out);
int dtc_layout_for_packing(uint64_t N, uint32_t C, uint32_t b, uint32_t packing, cpir_dtc_layout* out);
uint32_t dense_fields_per_word64(uint32_t b);
uint32_t planar_hi_planes(uint32_t b);
bool planar_offered(uint32_t b);
void set_default_dense(bool on);
void set_default_planar(bool on);
int check_layout(const cpir_dtc_layout& L);

// host_encoder.cpp : binary fuse filter + row codec (Matrix::from_kv_database)
struct Filter {
  uint8_t seed[32];
  uint32_t arity;
  uint32_t segment_length;
  uint32_t segment_count_length;
  uint64_t num_fingerprints;
  uint64_t filter_size;
  uint64_t mat_elem_bit_len;
  void to_bytes(uint8_t out[CPIR_FILTER_PARAM_BYTE_LEN]) const;
};
int encode_kv_database(uint32_t arity, const cpir_kv_db& db, uint32_t b, const uint8_t* filter_seeds, uint32_t max_attempts,
                       Filter* filter, std::vector<uint32_t>* D, uint64_t* N, uint32_t* C);

}  // namespace cpir
