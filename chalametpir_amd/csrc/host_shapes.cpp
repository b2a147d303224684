// host_shapes.cpp -- host-only shape arithmetic of the server path (no device needed).
#include <atomic>
#include <cmath>
#include <cstdlib>

#include "cpir_internal.hpp"

#include <sched.h>

#include <cstdio>

namespace cpir {

// CPUs this process may really use: min(affinity mask, cgroup v2 CPU quota).  A GPU box shows a container all 256 hardware threads
// but gives it a quota of 16: std::thread::hardware_concurrency() then oversubscribes 16x and the quota throttles EVERY thread of
// the process -- including the one squeezing the sponge that bounds setup.
unsigned usable_cpus() {
  unsigned n = 0;
#if defined(__linux__)
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = (unsigned)CPU_COUNT(&set);
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32] = {0};
    unsigned long long period = 0;
    if (fscanf(f, "%31s %llu", quota, &period) == 2 && quota[0] != 'm' && period > 0) {
      const unsigned long long q = strtoull(quota, nullptr, 10);
      const unsigned lim = (unsigned)((q + period / 2) / period);
      if (lim >= 1 && (n == 0 || lim < n)) n = lim;
    }
    fclose(f);
  }
#endif
  return n ? n : 4;
}

// reference chalametpir_common/src/matrix.rs:103-167: packing factor by element bit length
uint32_t compression_factor(uint32_t b) {
  if (b < CPIR_MIN_ELEM_BIT_LEN || b > CPIR_MAX_ELEM_BIT_LEN) return 0;
  return b >= 11 ? 2u : (b >= 9 ? 3u : 4u);
}

// reference chalametpir_server/src/server.rs:193-218: largest b with 2^32 >= 8 * (2^b)^2 * floor(sqrt(n))
int find_bit_len(uint64_t n, uint32_t* out) {
  if (!out) return CPIR_ERR_INVALID_ARGUMENT;
  if (n == 0) return CPIR_ERR_EMPTY_KV_DATABASE;  // Server::setup rejects an empty DB first (server.rs:49-51)
  uint64_t root = (uint64_t)std::sqrt((long double)n);
  while (root * root > n) root--;
  while ((root + 1) * (root + 1) <= n) root++;
  const unsigned __int128 q = (unsigned __int128)1 << 32;
  uint32_t b = 0;
  while (b < 32 && q >= ((unsigned __int128)8 << (2 * b)) * root) b++;
  b = b ? b - 1 : 0;
  if (b < CPIR_MIN_ELEM_BIT_LEN) return CPIR_ERR_KV_DATABASE_SIZE_TOO_LARGE;
  *out = b;
  return CPIR_OK;
}

// reference chalametpir_common/src/binary_fuse_filter.rs:519-538 (segment_length, size_factor) and :52-67 / :261-276
int filter_shape(uint32_t arity, uint64_t n, uint32_t* seg_len, uint32_t* seg_count_len, uint64_t* num_fp) {
  if (arity != 3 && arity != 4) return CPIR_ERR_UNSUPPORTED_ARITY;
  if (n == 0) return CPIR_ERR_EMPTY_KV_DATABASE;
  const double size = (double)(uint32_t)n;
  double exponent, factor;
  if (arity == 3) {
    exponent = std::floor(std::log(size) / std::log(3.33) + 2.25);
    factor = std::fmax(1.125, 0.875 + 0.25 * std::log(1e6) / std::log(size));
  } else {
    exponent = std::floor(std::log(size) / std::log(2.91) - 0.5);
    factor = std::fmax(1.075, 0.77 + 0.305 * std::log(6e5) / std::log(size));
  }
  // the reference casts the f64 exponent with `as usize`, which saturates negatives to 0 (4-wise filter, one key)
  uint32_t seg = 1u << (unsigned)(exponent < 0 ? 0.0 : exponent);
  if (seg > (1u << 18)) seg = 1u << 18;
  const uint32_t capacity = n > 1 ? (uint32_t)std::round(size * factor) : 0;
  const uint32_t init_segments = (capacity + seg - 1) / seg;
  const uint32_t proposed = (init_segments * seg + seg - 1) / seg;
  const uint32_t segments = proposed < arity ? 1 : proposed - (arity - 1);
  if (seg_len) *seg_len = seg;
  if (seg_count_len) *seg_count_len = segments * seg;
  if (num_fp) *num_fp = (uint64_t)(segments + arity - 1) * seg;
  return CPIR_OK;
}

// reference chalametpir_common/src/matrix.rs:694-700: 256-bit key digest + value + 1 boundary byte, b bits per element
uint64_t encoded_num_cols(uint64_t max_value_byte_len, uint32_t b) {
  if (b == 0) return 0;
  return (256 + 8 * max_value_byte_len + 8 + b - 1) / b;
}

// fields per u64 of the dense packing, or 0 where dense64 is not offered: it must be denser than the reference packing
// (floor(64/b) > 2*cf: b in {4,5,6,7,9,11,12}), and b in {4,5,6} is left out for now -- 10..16 fields per u64 need a
// narrower per-lane share to stay within the register file (respond.hip), and those bit lengths only occur above 2^26 keys
uint32_t dense_fields_per_word64(uint32_t b) {
  const uint32_t cf = compression_factor(b);
  if (cf == 0 || b < 7) return 0;
  const uint32_t k = 64 / b;
  return k > 2 * cf ? k : 0;
}

// bit planes above the low byte in the planar packing.  planar stores a field as its low byte plus (b - 8) one-bit planes: exactly
// b bits per field for b >= 9; for b <= 8 the byte alone (8 bits per field, what the reference packing spends there too), so that
// every bit length the reference allows runs on the matrix cores
uint32_t planar_hi_planes(uint32_t b) { return (compression_factor(b) != 0 && b >= 9) ? b - 8 : 0; }
bool planar_offered(uint32_t b) { return compression_factor(b) != 0; }

static std::atomic<int> g_default_dense{1};
static std::atomic<int> g_default_planar{1};
void set_default_dense(bool on) { g_default_dense.store(on ? 1 : 0); }
void set_default_planar(bool on) { g_default_planar.store(on ? 1 : 0); }

int dtc_layout_for_packing(uint64_t N, uint32_t C, uint32_t b, uint32_t packing, cpir_dtc_layout* out) {
  if (!out) return CPIR_ERR_INVALID_ARGUMENT;
  const uint32_t cf = compression_factor(b);
  if (cf == 0) return CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH;  // matrix.rs:99-101
  if (N == 0 || C == 0) return CPIR_ERR_INVALID_MATRIX_DIMENSION;
  memset(out, 0, sizeof(*out));
  out->num_slots = N;
  out->num_cols = C;
  out->mat_elem_bit_len = b;
  out->compression_factor = cf;
  out->words_per_row = (N + cf - 1) / cf;
  out->packing = packing;
  if (packing == CPIR_PACK_REFERENCE) {
    out->fields_per_word = cf;
    out->chunk_words = 1024;
  } else if (packing == CPIR_PACK_DENSE64) {
    const uint32_t k = dense_fields_per_word64(b);
    if (k == 0) return CPIR_ERR_INVALID_ARGUMENT;
    out->fields_per_word = k;
    out->chunk_words = 2048;
  } else if (packing == CPIR_PACK_PLANAR) {
    // super-tiles of 16 columns x 512 slots: 8 KiB of low bytes + 1 KiB per high bit plane, stored tile after tile
    // (column tile major, then along the slots); after the tiles one u32 per padded column: the wrap-around sum of its fields
    const uint32_t hb = planar_hi_planes(b);
    const uint64_t rp = ((uint64_t)C + CPIR_DTC_ROW_ALIGN - 1) / CPIR_DTC_ROW_ALIGN * CPIR_DTC_ROW_ALIGN;
    if (rp > 0xffffffffull) return CPIR_ERR_INVALID_MATRIX_DIMENSION;
    out->fields_per_word = 0;
    out->chunk_words = (8 + hb) * 256;  // u32 words of one super-tile
    out->slots_per_chunk = CPIR_PLANAR_SLOTS_PER_TILE;
    const uint64_t ks = (N + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE;
    out->words_per_row_padded = ks * (out->chunk_words / 16);  // per column: each super-tile spends chunk_words / 16 on it
    out->rows_padded = (uint32_t)rp;
    out->total_words = rp * out->words_per_row_padded + rp;
    return CPIR_OK;
  } else {
    return CPIR_ERR_INVALID_ARGUMENT;
  }
  out->slots_per_chunk = (uint64_t)out->fields_per_word * 1024;
  const uint64_t chunks = (N + out->slots_per_chunk - 1) / out->slots_per_chunk;
  out->words_per_row_padded = chunks * out->chunk_words;
  const uint64_t rp = ((uint64_t)C + CPIR_DTC_ROW_ALIGN - 1) / CPIR_DTC_ROW_ALIGN * CPIR_DTC_ROW_ALIGN;
  if (rp > 0xffffffffull) return CPIR_ERR_INVALID_MATRIX_DIMENSION;
  out->rows_padded = (uint32_t)rp;
  out->total_words = rp * out->words_per_row_padded;
  return CPIR_OK;
}

int dtc_layout_for(uint64_t N, uint32_t C, uint32_t b, cpir_dtc_layout* out) {
  if (g_default_planar.load() && planar_offered(b)) return dtc_layout_for_packing(N, C, b, CPIR_PACK_PLANAR, out);
  const bool dense = g_default_dense.load() && dense_fields_per_word64(b) != 0;
  return dtc_layout_for_packing(N, C, b, dense ? CPIR_PACK_DENSE64 : CPIR_PACK_REFERENCE, out);
}

// a layout handed in through the C ABI must be exactly what the library would have produced for its shape
int check_layout(const cpir_dtc_layout& L) {
  cpir_dtc_layout want;
  CPIR_TRY(dtc_layout_for_packing(L.num_slots, L.num_cols, L.mat_elem_bit_len, L.packing, &want));
  const bool same = want.compression_factor == L.compression_factor && want.words_per_row == L.words_per_row &&
                    want.words_per_row_padded == L.words_per_row_padded && want.rows_padded == L.rows_padded &&
                    want.total_words == L.total_words && want.fields_per_word == L.fields_per_word &&
                    want.chunk_words == L.chunk_words && want.slots_per_chunk == L.slots_per_chunk;
  return same ? CPIR_OK : CPIR_ERR_INVALID_ARGUMENT;
}

}  // namespace cpir
