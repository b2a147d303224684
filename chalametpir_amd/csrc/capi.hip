// capi.hip -- the extern "C" boundary of libchalamet_hip.so (include/chalamet_hip.h): errors, page-locked host memory, devices, shapes, the
// streaming XOF, the low-level device operations and the accessors of a server handle.  Server::setup lives in host_setup.hip,
// Server::respond on host buffers in host_respond.hip (reference chalametpir_server/src/server.rs:47-78, 103-167, 184-190).
#include "server_internal.hpp"

#include <execinfo.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <unistd.h>

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
// CPIR_ABORT_BACKTRACE=1 (read when the library is loaded): on SIGABRT / SIGSEGV / SIGBUS print which thread aborted and its native stack to fd 2, then
// hand over to the handler that was installed before (Python's faulthandler, or the default action).  An abort() raised by the
// runtime's own threads (a GPU memory fault reported by ROCr, a failed runtime assertion, glibc's heap checks) otherwise leaves no
// trace of WHERE it came from.  Diagnosis only; the test suite switches it on (tests/conftest.py).
// ---------------------------------------------------------------------------------------------------------------
struct JournalEntry {
  const char* what;
  const void* p;
  size_t bytes;
  const char* file;
  int line;
  long tid;
};
static constexpr unsigned kJournal = 512;
static JournalEntry g_journal[kJournal];
static std::atomic<uint64_t> g_journal_n{0};

// Only kept while the fatal-signal backtrace is switched on (CPIR_ABORT_BACKTRACE=1, read once when the library is loaded): the ring's
// slots are plain structs written without synchronisation -- good enough for a diagnosis aid that a dying process prints, not something
// every allocation of a production server should race on.
static std::atomic<bool> g_journal_on{false};

void journal_note(const char* what, const void* p, size_t bytes, const char* file, int line) {
  if (!g_journal_on.load(std::memory_order_relaxed)) return;
  const uint64_t i = g_journal_n.fetch_add(1, std::memory_order_relaxed);
  JournalEntry& e = g_journal[i % kJournal];
  const char* slash = strrchr(file, '/');
  e = JournalEntry{what, p, bytes, slash ? slash + 1 : file, line, (long)syscall(SYS_gettid)};
}

void journal_dump(int fd) {
  const uint64_t n = g_journal_n.load(std::memory_order_relaxed);
  const uint64_t from = n > 96 ? n - 96 : 0;
  char line[200];
  int k = snprintf(line, sizeof(line), "[cpir] allocation journal, entries %llu..%llu (oldest first):\n", (unsigned long long)from, (unsigned long long)n);
  if (k > 0) (void)!write(fd, line, (size_t)k);
  for (uint64_t i = from; i < n; i++) {
    const JournalEntry& e = g_journal[i % kJournal];
    k = snprintf(line, sizeof(line), "[cpir]   #%llu %-20s %p %zu bytes  %s:%d  tid %ld\n", (unsigned long long)i, e.what ? e.what : "?", e.p, e.bytes,
                 e.file ? e.file : "?", e.line, e.tid);
    if (k > 0) (void)!write(fd, line, (size_t)k);
  }
}

static struct sigaction g_prev_fatal[NSIG];
static void abort_backtrace_handler(int sig, siginfo_t* info, void* uctx) {
  const struct sigaction g_prev_abort = g_prev_fatal[sig];
  char name[32] = "?";
  (void)prctl(PR_GET_NAME, name, 0, 0, 0);
  char line[160];
  const int n = snprintf(line, sizeof(line), "\n[cpir] signal %d (%s) on thread '%s' (tid %ld, pid %ld), fault address %p; native stack:\n", sig,
                         sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : "fatal", name,
                         (long)syscall(SYS_gettid), (long)getpid(), info ? info->si_addr : nullptr);
  if (n > 0) (void)!write(2, line, (size_t)n);
  void* frames[64];
  const int depth = backtrace(frames, 64);
  backtrace_symbols_fd(frames, depth, 2);
  (void)!write(2, "[cpir] end of native stack\n", 27);
  journal_dump(2);
  if ((g_prev_abort.sa_flags & SA_SIGINFO) && g_prev_abort.sa_sigaction) {
    g_prev_abort.sa_sigaction(sig, info, uctx);
  } else if (g_prev_abort.sa_handler != SIG_DFL && g_prev_abort.sa_handler != SIG_IGN && g_prev_abort.sa_handler) {
    g_prev_abort.sa_handler(sig);
  }
  signal(sig, SIG_DFL);  // (the previous handler normally re-raises by itself)
  raise(sig);
}

__attribute__((constructor)) static void install_abort_backtrace() {
  const char* on = getenv("CPIR_ABORT_BACKTRACE");
  if (!on || on[0] != '1') return;
  g_journal_on.store(true, std::memory_order_relaxed);
  void* warm[4];
  (void)backtrace(warm, 4);  // loads libgcc's unwinder now, not inside the handler
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = abort_backtrace_handler;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  for (int sig : {SIGABRT, SIGSEGV, SIGBUS}) (void)sigaction(sig, &sa, &g_prev_fatal[sig]);
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
  CPIR_HIP_TRY(CPIR_HIP_HOST_MALLOC(out, bytes, hipHostMallocPortable));
  return CPIR_OK;
}

void cpir_host_free(void* p) {
  if (p) (void)CPIR_HIP_HOST_FREE(p);
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

void cpir_device_close(cpir_device* dev) {
  if (dev) scratch_drain(dev->ordinal);
  device_release(dev);
}

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

int cpir_packed_rhs_offered(const cpir_dtc_layout* layout) {
  return layout && check_layout(*layout) == CPIR_OK && layout->packing == CPIR_PACK_PLANAR && layout->mat_elem_bit_len >= 9;
}

uint64_t cpir_packed_rhs_plane_bytes(const cpir_dtc_layout* layout) {
  if (!layout || check_layout(*layout) != CPIR_OK) return 0;
  return planar_hi_plane_bytes(*layout);
}

int cpir_op_transpose_compress_with_plane(cpir_device* dev, const uint32_t* D, uint64_t ldd, const cpir_dtc_layout* layout, uint32_t* dtc,
                                          uint32_t* or_of_entries, void* hi_plane, void* stream) {
  if (!dev || !layout) return CPIR_ERR_INVALID_ARGUMENT;
  if (!cpir_packed_rhs_offered(layout) || (hi_plane != nullptr) != (planar_hi_plane_bytes(*layout) != 0)) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  return launch_transpose_compress(dev, D, ldd, *layout, dtc, or_of_entries, pick_stream(dev, stream), hi_plane);
}

int cpir_op_mat_x_packed(cpir_device* dev, const uint32_t* A, uint64_t lda, const uint32_t* dtc, const cpir_dtc_layout* layout,
                         const void* hi_plane, uint32_t* M, uint64_t ldm, uint64_t rows, int accumulate, void* stream) {
  if (!dev || !layout || !A || !dtc || !M || rows == 0) return CPIR_ERR_INVALID_ARGUMENT;  // (hi_plane: NULL exactly where the plane has 0 bytes)
  CPIR_TRY(check_layout(*layout));
  if (!mfma_matmul_enabled() || !mfma_planar_rhs_applicable(A, lda, *layout)) return CPIR_ERR_INVALID_ARGUMENT;
  DeviceGuard g(dev->ordinal);
  hipStream_t s = pick_stream(dev, stream);
  void* rowsum = nullptr;  // scratch (released behind the launch): the row sums of A (a correction term of the signed-byte split)
  CPIR_TRY(scratch_acquire(&rowsum, 4 * ((rows + 127) / 128 * 128), s));
  const int st = launch_mat_x_mat_mfma_planar(dev, A, lda, dtc, *layout, hi_plane, static_cast<uint32_t*>(rowsum), M, ldm, rows, accumulate, s);
  const int st2 = scratch_release_after(rowsum, s);
  return st != CPIR_OK ? st : st2;
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
const char* cpir_pack_kernel_name(const cpir_dtc_layout* layout) { return layout ? pack_kernel_name(*layout) : ""; }
uint32_t cpir_respond_batch_pass_width(const cpir_dtc_layout* layout, uint32_t batch) {
  if (!layout || batch == 0) return 0;
  if (layout->packing != CPIR_PACK_PLANAR) return batch >= 4 ? 4 : (batch >= 2 ? 2 : 1);  // (passes of 4, then 2, then 1)
  const uint32_t w = respond_planar_pass_width(*layout, batch);
  return w < batch ? w : batch;
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
  CPIR_HIP_TRY(CPIR_HIP_MALLOC(&tmp.p, (size_t)words * 4));
  if (srv->map.active()) {
    // only the slots with a non-zero row are resident: export those in the reference's representation, then spread them over all N slots
    // (a dropped slot's fields are zero -- that is why it was dropped)
    const cpir_dtc_layout& P = srv->phys;
    DevBuf part;
    CPIR_HIP_TRY(CPIR_HIP_MALLOC(&part.p, (size_t)P.num_cols * P.words_per_row * 4));
    CPIR_TRY(launch_dtc_export(srv->dev, srv->dtc, P, (uint32_t*)part.p, srv->dev->stream));
    CPIR_TRY(launch_expand_ref(srv->dev, (const uint32_t*)part.p, P.words_per_row, srv->map, L.words_per_row, L.num_cols, L.compression_factor,
                               (uint32_t*)tmp.p, srv->dev->stream));
    CPIR_HIP_TRY(hipMemcpyAsync(compressed_out, tmp.p, (size_t)words * 4, hipMemcpyDeviceToHost, srv->dev->stream));
    CPIR_HIP_TRY(hipStreamSynchronize(srv->dev->stream));
    return CPIR_OK;
  }
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

int cpir_server_host_path_counts(const cpir_server* srv, uint64_t out[CPIR_HOST_PATH_COUNT]) {
  if (!srv || !out) return CPIR_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < CPIR_HOST_PATH_COUNT; i++) out[i] = 0;
  auto add = [&](const Server* one) {
    const Server::Served& s = one->served;
    out[0] += s.calls.load(std::memory_order_relaxed), out[1] += s.alone.load(std::memory_order_relaxed);
    out[2] += one->fill_polled.load(std::memory_order_relaxed), out[3] += s.polled_void.load(std::memory_order_relaxed);
    out[4] += s.in_uploaded_rounds.load(std::memory_order_relaxed), out[5] += s.uploaded_rounds.load(std::memory_order_relaxed);
    out[6] += s.in_place_calls.load(std::memory_order_relaxed), out[7] += s.in_place_rounds.load(std::memory_order_relaxed);
  };
  if (srv->shards.empty()) add(srv);
  else
    for (const Server* c : srv->shards) add(c);  // (a group: every query is answered by every shard, and counted there)
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

int cpir_server_physical_layout(const cpir_server* srv, cpir_dtc_layout* out) {
  if (!srv || !out) return CPIR_ERR_INVALID_ARGUMENT;
  *out = srv->phys;
  return CPIR_OK;
}

int cpir_server_slots_served(const cpir_server* srv, uint64_t* served, uint64_t* of_slots) {
  if (!srv) return CPIR_ERR_INVALID_ARGUMENT;
  uint64_t kept = 0;
  if (!srv->shards.empty()) {
    for (const Server* c : srv->shards) kept += c->map.active() ? c->map.n_kept : c->layout.num_slots;
  } else {
    kept = srv->map.active() ? srv->map.n_kept : srv->layout.num_slots;
  }
  if (served) *served = kept;
  if (of_slots) *of_slots = srv->layout.num_slots;
  return CPIR_OK;
}

int cpir_server_kept_slots(const cpir_server* srv, uint32_t* out, uint64_t cap) {
  if (!srv || !out || !srv->shards.empty()) return CPIR_ERR_INVALID_ARGUMENT;
  if (!srv->map.active()) return CPIR_ERR_INVALID_ARGUMENT;
  if (cap < srv->map.n_kept) return CPIR_ERR_BUFFER_TOO_SMALL;
  memcpy(out, srv->map.keep_host.data(), (size_t)srv->map.n_kept * 4);
  return CPIR_OK;
}

const char* cpir_host_gather_variant(void) { return gather_words_variant(); }

int cpir_host_gather_words(uint32_t* dst, const uint32_t* src, const uint32_t* idx, uint64_t count) {
  if ((!dst || !src || !idx) && count) return CPIR_ERR_INVALID_ARGUMENT;
  gather_words(dst, src, idx, (size_t)count);
  return CPIR_OK;
}

int cpir_host_compress_words(uint32_t* dst, const uint32_t* src, const uint8_t* bits, uint64_t s_lo, uint64_t s_hi, uint64_t* count) {
  if (!dst || !src || !bits || !count || s_lo > s_hi) return CPIR_ERR_INVALID_ARGUMENT;
  // (a 64-byte aligned destination takes the non-temporal form the arenas' staging uses, any other the plain one: both are reachable from tests)
  *count = compress_words_streaming(dst, src, bits, (size_t)s_lo, (size_t)s_hi);
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

}  // extern "C"
