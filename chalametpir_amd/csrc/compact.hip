// compact.hip -- serving only the filter slots that some key owns.
//
// A REAL encoded database (Matrix::from_kv_database, reference chalametpir_common/src/matrix.rs:702-746) has N = num_fingerprints rows
// but only n = number of keys of them are ever written: the binary fuse filter gives every key exactly one slot of its own (the peel
// order, matrix.rs:727-740) and the other N - n rows stay all zero -- 11.1 % of the rows at arity 3, 7.2 % at arity 4 for 2^20 keys.
// A zero row contributes 0 to every column of q * D whatever q[n] is, so the respond stream can skip it without changing a bit of any
// response: the packed image holds only the rows with a non-zero field ("kept" slots), and the query is compacted through the same
// map in front of the kernel,  q'[i] = q[keep[i]].  Nothing here knows about filters: the zero rows are FOUND on the device (after
// masking to b bits, exactly what row_wise_compress keeps, matrix.rs:121), so a matrix from any source is treated alike, and a matrix
// without such rows (the synthetic benchmark matrix) is left alone.
//
//   row_any_kernel        which rows have a non-zero field; also ORs the UNMASKED entries (the hint multiplies D as it is, server.rs:61)
//   gather_rows_kernel    D'[i][:] = D[keep[i]][:]                       (once per setup, in front of the pack kernel)
//   gather_query_kernel   q'[b][i] = q[b][offset + keep[i]], 0 beyond    (per respond launch: 4.7 MB read + 4.2 MB written at 2^20 keys)
//   expand_ref_kernel     the reference's C x ceil(N/cf) words from the words of the compact matrix (export only)
#include "cpir_internal.hpp"

namespace cpir {
namespace {

constexpr int kThreads = 256;

__global__ void __launch_bounds__(kThreads) row_any_kernel(const uint32_t* __restrict__ D, uint64_t ldd, uint64_t N, uint32_t C, uint32_t mask,
                                                            uint8_t* __restrict__ flags, uint32_t* __restrict__ or_of_entries) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t ored = 0;
  for (uint64_t row = (uint64_t)blockIdx.x * 4 + wave; row < N; row += (uint64_t)gridDim.x * 4) {
    const uint32_t* src = D + row * ldd;
    uint32_t v = 0;
    for (uint32_t c = lane; c < C; c += 64) v |= src[c];
    ored |= v;
    const bool any = __ballot((v & mask) != 0) != 0;
    if (lane == 0) flags[row] = any ? 1 : 0;
  }
  if (or_of_entries) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ored |= __shfl_xor(ored, off, 64);
    if (lane == 0 && ored) atomicOr(or_of_entries, ored);
  }
}

__global__ void __launch_bounds__(kThreads) gather_rows_kernel(const uint32_t* __restrict__ D, uint64_t ldd, const uint32_t* __restrict__ keep,
                                                                uint64_t n_kept, uint32_t C, uint32_t* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t i = (uint64_t)blockIdx.x * 4 + wave; i < n_kept; i += (uint64_t)gridDim.x * 4) {
    const uint32_t* src = D + (uint64_t)keep[i] * ldd;
    uint32_t* dst = out + i * C;
    for (uint32_t c = lane; c < C; c += 64) dst[c] = src[c];
  }
}

// one thread = 4 consecutive compact slots of one query: one 16-byte load of the map, four word loads of q (the map is increasing, so a
// wave reads a nearly contiguous run of q), one 16-byte store.  keep[] is padded to n_pad with 0xFFFFFFFF: those words are written as 0.
__global__ void __launch_bounds__(kThreads) gather_query_kernel(const uint32_t* __restrict__ q, uint64_t q_len, uint64_t q_slot_offset,
                                                                 const uint32_t* __restrict__ keep, uint64_t n_pad, uint32_t batch,
                                                                 uint32_t* __restrict__ out) {
  const uint64_t quads = n_pad / 4;
  const uint64_t total = quads * batch;
  for (uint64_t t = (uint64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += (uint64_t)gridDim.x * kThreads) {
    const uint64_t b = t / quads, i = (t % quads) * 4;
    const uint4 k = *reinterpret_cast<const uint4*>(keep + i);
    const uint32_t* src = q + b * q_len + q_slot_offset;
    uint4 v;
    v.x = k.x != 0xFFFFFFFFu ? src[k.x] : 0u;
    v.y = k.y != 0xFFFFFFFFu ? src[k.y] : 0u;
    v.z = k.z != 0xFFFFFFFFu ? src[k.z] : 0u;
    v.w = k.w != 0xFFFFFFFFu ? src[k.w] : 0u;
    *reinterpret_cast<uint4*>(out + b * n_pad + i) = v;
  }
}

// The same two questions asked of a database that arrives in the REFERENCE's compressed form (C rows of W words, cf fields of 32 / cf bits
// a word; cpir_server_from_compressed): word w of the OR over all rows tells which of its cf slots hold something ...
__global__ void __launch_bounds__(kThreads) or_rows_kernel(const uint32_t* __restrict__ src, uint64_t W, uint32_t C, uint32_t* __restrict__ ored) {
  for (uint64_t w = (uint64_t)blockIdx.x * kThreads + threadIdx.x; w < W; w += (uint64_t)gridDim.x * kThreads) {
    uint32_t v = 0;
    for (uint32_t c = 0; c < C; c++) v |= src[(uint64_t)c * W + w];
    ored[w] = v;
  }
}

// ... and the compressed matrix of the kept slots alone: field j of word wc of row c is the field of slot keep[cf * wc + j] (0 beyond the kept slots)
__global__ void __launch_bounds__(kThreads) gather_compressed_kernel(const uint32_t* __restrict__ src, uint64_t W, const uint32_t* __restrict__ keep,
                                                                      uint64_t n_kept, uint32_t C, uint32_t cf, uint64_t Wc, uint32_t* __restrict__ out) {
  const uint32_t S = 32 / cf;
  const uint32_t slot_mask = S == 32 ? 0xFFFFFFFFu : ((1u << S) - 1u);
  const uint64_t total = (uint64_t)C * Wc;
  for (uint64_t t = (uint64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += (uint64_t)gridDim.x * kThreads) {
    const uint64_t c = t / Wc, wc = t % Wc;
    uint32_t v = 0;
    for (uint32_t j = 0; j < cf; j++) {
      const uint64_t i = wc * cf + j;
      if (i >= n_kept) break;
      const uint32_t n = keep[i];
      v |= ((src[c * W + n / cf] >> ((n % cf) * S)) & slot_mask) << (j * S);
    }
    out[t] = v;
  }
}

__global__ void __launch_bounds__(kThreads) rank_fill_kernel(const uint32_t* __restrict__ keep, uint64_t n_kept, uint32_t* __restrict__ rank) {
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n_kept; i += (uint64_t)gridDim.x * kThreads) rank[keep[i]] = (uint32_t)i;
}

// word (c, w) of the reference's compressed matrix over ALL N slots: field j is slot n = cf * w + j; a slot that was dropped holds 0, a
// kept one is field rank[n] % cf of word rank[n] / cf of the compact matrix's row c (slot width 32 / cf bits, matrix.rs:103-167)
__global__ void __launch_bounds__(kThreads) expand_ref_kernel(const uint32_t* __restrict__ compact, uint64_t Wc, const uint32_t* __restrict__ rank,
                                                               uint64_t N, uint64_t W, uint32_t C, uint32_t cf, uint32_t* __restrict__ dst) {
  const uint32_t S = 32 / cf;
  const uint32_t slot_mask = S == 32 ? 0xFFFFFFFFu : ((1u << S) - 1u);
  const uint64_t total = (uint64_t)C * W;
  for (uint64_t t = (uint64_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += (uint64_t)gridDim.x * kThreads) {
    const uint64_t c = t / W, w = t % W;
    uint32_t out = 0;
    for (uint32_t j = 0; j < cf; j++) {
      const uint64_t n = w * cf + j;
      if (n >= N) break;
      const uint32_t i = rank[n];
      if (i == 0xFFFFFFFFu) continue;
      out |= ((compact[c * Wc + i / cf] >> ((i % cf) * S)) & slot_mask) << (j * S);
    }
    dst[t] = out;
  }
}

uint32_t grid_for_items(const Device* dev, uint64_t items) {
  uint64_t g = (items + kThreads - 1) / kThreads;
  const uint64_t cap = (uint64_t)dev->num_cus * 16;
  if (g > cap) g = cap;
  return g < 1 ? 1u : (uint32_t)g;
}

std::atomic<int> g_compact_mode{1};  // tuning "layout.compact_slots": 0 never, 1 where at least 1/32 of the rows are zero, 2 whenever a row is zero

}  // namespace

int compact_slots_mode() { return g_compact_mode.load(std::memory_order_relaxed); }
void set_compact_slots_mode(int m) { g_compact_mode.store(m, std::memory_order_relaxed); }

void SlotMap::reset() {
  if (keep_dev) (void)CPIR_HIP_FREE(keep_dev);
  keep_dev = nullptr;
  keep_host.clear();
  keep_host.shrink_to_fit();
  keep_bits.clear();
  keep_bits.shrink_to_fit();
  n_kept = n_pad = n_orig = 0;
}

// Finds the rows of D (N x C on the device, leading dimension ldd) with a non-zero field after masking to b bits and, if the tuning mode
// says the zero rows are worth skipping, fills `map` (device + host copies of the kept slots, increasing).  *or_of_entries_host (optional)
// receives the OR of all UNMASKED entries.  Synchronises `stream`.  map->n_kept == 0 afterwards means "serve every slot".
int build_slot_map(const Device* dev, const uint32_t* D_dev, uint64_t ldd, uint64_t N, uint32_t C, uint32_t b, hipStream_t stream, SlotMap* map,
                   uint32_t* or_of_entries_host) {
  map->reset();
  const int mode = compact_slots_mode();
  if (or_of_entries_host) *or_of_entries_host = 0;
  if ((mode == 0 && !or_of_entries_host) || N == 0 || N >= 0xFFFFFFF0ull) return CPIR_OK;
  uint8_t* flags_dev = nullptr;
  const size_t flag_bytes = (size_t)(N + 3) / 4 * 4;  // the OR word sits behind the flags, 4-byte aligned
  CPIR_TRY(scratch_acquire(reinterpret_cast<void**>(&flags_dev), flag_bytes + 4, stream));
  uint32_t* const or_dev = reinterpret_cast<uint32_t*>(flags_dev + flag_bytes);
  hipError_t e = hipMemsetAsync(or_dev, 0, 4, stream);
  std::vector<uint8_t> flags((size_t)N);
  uint32_t ored = 0;
  if (e == hipSuccess) {
    const uint32_t mask = (b >= 32) ? 0xFFFFFFFFu : ((1u << b) - 1u);
    const uint64_t blocks = (N + 3) / 4;
    const uint64_t cap = (uint64_t)dev->num_cus * 16;
    hipLaunchKernelGGL(row_any_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(kThreads), 0, stream, D_dev, ldd, N, C, mask, flags_dev, or_dev);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(flags.data(), flags_dev, (size_t)N, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&ored, or_dev, 4, hipMemcpyDeviceToHost, stream);
  const hipError_t e2 = hipStreamSynchronize(stream);
  (void)scratch_release_after(flags_dev, stream);
  if (e == hipSuccess) e = e2;
  CPIR_HIP_TRY(e);
  if (or_of_entries_host) *or_of_entries_host = ored;
  return slot_map_from_flags(flags, N, stream, map);
}

// flags[n] != 0: slot n holds something.  Fills `map` if the tuning mode says the empty slots are worth skipping; synchronises `stream`.
int slot_map_from_flags(const std::vector<uint8_t>& flags, uint64_t N, hipStream_t stream, SlotMap* map) {
  const int mode = compact_slots_mode();
  hipError_t e = hipSuccess;
  uint64_t kept = 0;
  for (uint64_t n = 0; n < N; n++) kept += flags[n];
  const uint64_t zeros = N - kept;
  const bool compact = kept > 0 && ((mode == 2 && zeros > 0) || (mode == 1 && zeros * 32 >= N));
  if (!compact) return CPIR_OK;
  map->n_orig = N;
  map->n_kept = kept;
  map->n_pad = (kept + 127) / 128 * 128;  // whole 512-byte runs per query row: every 16-byte piece of a compact query is aligned
  map->keep_host.resize((size_t)map->n_pad, 0xFFFFFFFFu);
  uint64_t i = 0;
  for (uint64_t n = 0; n < N; n++)
    if (flags[n]) map->keep_host[(size_t)i++] = (uint32_t)n;
  e = CPIR_HIP_MALLOC(&map->keep_dev, (size_t)map->n_pad * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(map->keep_dev, map->keep_host.data(), (size_t)map->n_pad * 4, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) {
    set_last_hip_error(e, "slot map upload", __FILE__, __LINE__);
    map->reset();
    return e == hipErrorOutOfMemory ? CPIR_ERR_OUT_OF_DEVICE_MEMORY : CPIR_ERR_HIP;
  }
  map->keep_host.resize((size_t)map->n_kept);  // the host copy holds the kept slots only
  map->keep_bits.assign((size_t)(N + 7) / 8 + 8, 0);
  for (uint64_t n = 0; n < N; n++)
    if (flags[n]) map->keep_bits[(size_t)(n >> 3)] |= (uint8_t)(1u << (n & 7));
  return CPIR_OK;
}

// The slot map of a database in the reference's compressed form on the device (C x W words, cf fields of 32 / cf bits a word, b of them
// significant).  Synchronises `stream`; map->n_kept == 0 afterwards means "serve every slot".
int build_slot_map_from_compressed(const Device* dev, const uint32_t* src_dev, uint64_t W, uint64_t N, uint32_t C, uint32_t b, uint32_t cf,
                                   hipStream_t stream, SlotMap* map) {
  map->reset();
  if (compact_slots_mode() == 0 || N == 0 || N >= 0xFFFFFFF0ull || cf == 0 || cf > 32) return CPIR_OK;
  uint32_t* ored_dev = nullptr;
  CPIR_TRY(scratch_acquire(reinterpret_cast<void**>(&ored_dev), (size_t)W * 4, stream));
  hipLaunchKernelGGL(or_rows_kernel, dim3(grid_for_items(dev, W)), dim3(kThreads), 0, stream, src_dev, W, C, ored_dev);
  hipError_t e = hipGetLastError();
  std::vector<uint32_t> ored((size_t)W);
  if (e == hipSuccess) e = hipMemcpyAsync(ored.data(), ored_dev, (size_t)W * 4, hipMemcpyDeviceToHost, stream);
  const hipError_t e2 = hipStreamSynchronize(stream);
  (void)scratch_release_after(ored_dev, stream);
  if (e == hipSuccess) e = e2;
  CPIR_HIP_TRY(e);
  const uint32_t S = 32 / cf, mask = (b >= 32) ? 0xFFFFFFFFu : ((1u << b) - 1u);
  std::vector<uint8_t> flags((size_t)N);
  for (uint64_t n = 0; n < N; n++) flags[(size_t)n] = ((ored[(size_t)(n / cf)] >> ((n % cf) * S)) & mask) != 0 ? 1 : 0;
  return slot_map_from_flags(flags, N, stream, map);
}

int launch_gather_compressed(const Device* dev, const uint32_t* src_dev, uint64_t W, const SlotMap& map, uint32_t C, uint32_t cf, uint64_t Wc,
                             uint32_t* out, hipStream_t stream) {
  if (!src_dev || !out || !map.keep_dev || cf == 0) return CPIR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(gather_compressed_kernel, dim3(grid_for_items(dev, (uint64_t)C * Wc)), dim3(kThreads), 0, stream, src_dev, W, map.keep_dev,
                     map.n_kept, C, cf, Wc, out);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

int launch_gather_rows(const Device* dev, const uint32_t* D_dev, uint64_t ldd, const SlotMap& map, uint32_t C, uint32_t* out, hipStream_t stream) {
  if (!D_dev || !out || !map.keep_dev) return CPIR_ERR_INVALID_ARGUMENT;
  const uint64_t blocks = (map.n_kept + 3) / 4, cap = (uint64_t)dev->num_cus * 16;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(kThreads), 0, stream, D_dev, ldd, map.keep_dev, map.n_kept,
                     C, out);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

int launch_gather_query(const Device* dev, const uint32_t* q, uint64_t q_len, uint64_t q_slot_offset, const SlotMap& map, uint32_t batch,
                        uint32_t* out, hipStream_t stream) {
  if (!q || !out || !map.keep_dev || batch == 0) return CPIR_ERR_INVALID_ARGUMENT;
  if (q_slot_offset + map.n_orig > q_len) return CPIR_ERR_SHARD_RANGE;
  hipLaunchKernelGGL(gather_query_kernel, dim3(grid_for_items(dev, map.n_pad / 4 * batch)), dim3(kThreads), 0, stream, q, q_len, q_slot_offset,
                     map.keep_dev, map.n_pad, batch, out);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

// compact_ref: C x Wc words (the reference's representation of the COMPACT matrix, on the device) -> out: C x W words over all N slots
int launch_expand_ref(const Device* dev, const uint32_t* compact_ref, uint64_t Wc, const SlotMap& map, uint64_t W, uint32_t C, uint32_t cf,
                      uint32_t* out, hipStream_t stream) {
  if (!compact_ref || !out || !map.keep_dev) return CPIR_ERR_INVALID_ARGUMENT;
  uint32_t* rank = nullptr;
  CPIR_TRY(scratch_acquire(reinterpret_cast<void**>(&rank), (size_t)map.n_orig * 4, stream));
  hipError_t e = hipMemsetAsync(rank, 0xFF, (size_t)map.n_orig * 4, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(rank_fill_kernel, dim3(grid_for_items(dev, map.n_kept)), dim3(kThreads), 0, stream, map.keep_dev, map.n_kept, rank);
    hipLaunchKernelGGL(expand_ref_kernel, dim3(grid_for_items(dev, (uint64_t)C * W)), dim3(kThreads), 0, stream, compact_ref, Wc, rank, map.n_orig, W, C,
                       cf, out);
    e = hipGetLastError();
  }
  const int st2 = scratch_release_after(rank, stream);
  CPIR_HIP_TRY(e);
  return st2;
}

}  // namespace cpir
