/*
 * chalamet_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See chalamet_oracle.h for
 * scope, pinning status and the meaning of the reference short names (matrix.rs, bff.rs, ...).
 */
#include "chalamet_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HOT __attribute__((target_clones("avx512f", "avx2", "default")))

/* ============================================================================================
 * TurboSHAKE128 (RFC 9861): Keccak-p[1600, n_r = 12] sponge, rate 168 B, capacity 256 bit.
 * The reference calls it through the `turboshake` crate (=0.4.1): TurboShake128::default(),
 * absorb, finalize::<DEFAULT_DOMAIN_SEPARATOR = 0x1F>, squeeze (matrix.rs:542-554, bff.rs:569-574,
 * serialization.rs:24-29).  Keccak-p[1600,12] = the LAST 12 rounds of Keccak-f[1600] (FIPS 202
 * round indices 12..23).
 * ============================================================================================ */
#define TS128_RATE 168u

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

static inline uint64_t rol64(uint64_t x, unsigned n) { return (x << n) | (x >> (64 - n)); }

static void keccak_p1600_12(uint64_t s[25]) {
  static const unsigned rho[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const unsigned pil[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (unsigned round = 12; round < 24; round++) {
    uint64_t bc[5], t;
    /* theta */
    for (unsigned i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
    for (unsigned i = 0; i < 5; i++) {
      t = bc[(i + 4) % 5] ^ rol64(bc[(i + 1) % 5], 1);
      for (unsigned j = 0; j < 25; j += 5) s[j + i] ^= t;
    }
    /* rho + pi */
    t = s[1];
    for (unsigned i = 0; i < 24; i++) {
      unsigned j = pil[i];
      uint64_t tmp = s[j];
      s[j] = rol64(t, rho[i]);
      t = tmp;
    }
    /* chi */
    for (unsigned j = 0; j < 25; j += 5) {
      for (unsigned i = 0; i < 5; i++) bc[i] = s[j + i];
      for (unsigned i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    /* iota */
    s[0] ^= KECCAK_RC[round];
  }
}

void or_ts128_init(or_turboshake128* st) { memset(st, 0, sizeof(*st)); }

void or_ts128_absorb(or_turboshake128* st, const uint8_t* in, size_t len) {
  uint8_t* sb = (uint8_t*)st->s; /* little-endian host assumed, as the reference's raw casts do (matrix.rs:549-555) */
  while (len > 0) {
    size_t take = TS128_RATE - st->pos;
    if (take > len) take = len;
    for (size_t i = 0; i < take; i++) sb[st->pos + i] ^= in[i];
    st->pos += (unsigned)take;
    in += take;
    len -= take;
    if (st->pos == TS128_RATE) {
      keccak_p1600_12(st->s);
      st->pos = 0;
    }
  }
}

void or_ts128_finalize(or_turboshake128* st, uint8_t domain_sep) {
  uint8_t* sb = (uint8_t*)st->s;
  sb[st->pos] ^= domain_sep;
  sb[TS128_RATE - 1] ^= 0x80;
  keccak_p1600_12(st->s);
  st->pos = 0;
  st->squeezing = 1;
}

void or_ts128_squeeze(or_turboshake128* st, uint8_t* out, size_t len) {
  const uint8_t* sb = (const uint8_t*)st->s;
  while (len > 0) {
    if (st->pos == TS128_RATE) {
      keccak_p1600_12(st->s);
      st->pos = 0;
    }
    size_t take = TS128_RATE - st->pos;
    if (take > len) take = len;
    memcpy(out, sb + st->pos, take);
    st->pos += (unsigned)take;
    out += take;
    len -= take;
  }
}

void or_turboshake128_hash(const uint8_t* msg, size_t mlen, uint8_t domain_sep, uint8_t* out, size_t olen) {
  or_turboshake128 st;
  or_ts128_init(&st);
  or_ts128_absorb(&st, msg, mlen);
  or_ts128_finalize(&st, domain_sep);
  or_ts128_squeeze(&st, out, olen);
}

/* ============================================================================================
 * Matrix
 * ============================================================================================ */

unsigned or_compression_factor(unsigned b) {
  /* matrix.rs:103-167: 11..=14 -> 2, 9..=10 -> 3, 4..=8 -> 4 */
  if (b < OR_MIN_CIPHER_TEXT_BIT_LEN || b > OR_MAX_CIPHER_TEXT_BIT_LEN) return 0;
  if (b >= 11) return 2;
  if (b >= 9) return 3;
  return 4;
}

int or_generate_from_seed(uint64_t rows, uint64_t cols, const uint8_t seed[32], uint32_t* out) {
  /* matrix.rs:541-558 */
  if (rows == 0 || cols == 0) return OR_ERR_INVALID_MATRIX_DIMENSION;
  or_turboshake128 st;
  or_ts128_init(&st);
  or_ts128_absorb(&st, seed, OR_SEED_BYTE_LEN);
  or_ts128_finalize(&st, 0x1F);
  or_ts128_squeeze(&st, (uint8_t*)out, (size_t)(rows * cols) * sizeof(uint32_t));
  return OR_OK;
}

HOT static void mul_rows(const uint32_t* lhs, uint64_t lrows, uint64_t lcols, const uint32_t* rhs, uint64_t rcols,
                         uint32_t* out) {
  /* matrix.rs:1050-1055: every output element is an independent wrapping fold over k.  The loop nest is
   * re-ordered (k outer, c inner) so the CPU baseline streams rhs rows; the value of each output is the
   * same wrapping sum (u32 addition is associative and commutative mod 2^32). */
#pragma omp parallel for schedule(static)
  for (uint64_t r = 0; r < lrows; r++) {
    uint32_t* o = out + r * rcols;
    for (uint64_t c = 0; c < rcols; c++) o[c] = 0;
    for (uint64_t k = 0; k < lcols; k++) {
      const uint32_t a = lhs[r * lcols + k];
      const uint32_t* b = rhs + k * rcols;
      for (uint64_t c = 0; c < rcols; c++) o[c] += a * b[c];
    }
  }
}

int or_mul(const uint32_t* lhs, uint64_t lrows, uint64_t lcols, const uint32_t* rhs, uint64_t rrows, uint64_t rcols,
           uint32_t* out) {
  /* matrix.rs:1040-1059 */
  if (lrows == 0 || lcols == 0 || rrows == 0 || rcols == 0) return OR_ERR_INVALID_MATRIX_DIMENSION;
  if (lcols != rrows) return OR_ERR_INCOMPATIBLE_DIM_MATMUL; /* matrix.rs:1044-1046 */
  mul_rows(lhs, lrows, lcols, rhs, rcols, out);
  return OR_OK;
}

int or_add(const uint32_t* lhs, uint64_t lrows, uint64_t lcols, const uint32_t* rhs, uint64_t rrows, uint64_t rcols,
           uint32_t* out) {
  /* matrix.rs:1070-1086 */
  if (!(lrows == rrows && lcols == rcols)) return OR_ERR_INCOMPATIBLE_DIM_MATADD;
  uint64_t n = lrows * lcols;
  for (uint64_t i = 0; i < n; i++) out[i] = lhs[i] + rhs[i];
  return OR_OK;
}

int or_transpose(const uint32_t* in, uint64_t rows, uint64_t cols, uint32_t* out) {
  /* matrix.rs:517-527: res[(ridx, cidx)] = self[(cidx, ridx)], res is cols x rows */
  if (rows == 0 || cols == 0) return OR_ERR_INVALID_MATRIX_DIMENSION;
#pragma omp parallel for schedule(static)
  for (uint64_t ridx = 0; ridx < cols; ridx++)
    for (uint64_t cidx = 0; cidx < rows; cidx++) out[ridx * rows + cidx] = in[cidx * cols + ridx];
  return OR_OK;
}

int or_identity(uint64_t rows, uint32_t* out) {
  /* matrix.rs:498-510 */
  if (rows == 0) return OR_ERR_INVALID_MATRIX_DIMENSION;
  memset(out, 0, (size_t)(rows * rows) * sizeof(uint32_t));
  for (uint64_t i = 0; i < rows; i++) out[i * rows + i] = 1;
  return OR_OK;
}

int or_row_wise_compress(const uint32_t* in, uint64_t rows, uint64_t cols, unsigned b, uint32_t* out) {
  /* matrix.rs:98-205.  One arm per compression factor in the reference; the arms differ only in
   * COMPRESSION_FACTOR and BITS_PER_UNCOMPRESSED_ELEMENT = 32 / COMPRESSION_FACTOR, so they are folded here. */
  const unsigned cf = or_compression_factor(b);
  if (cf == 0) return OR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH; /* matrix.rs:99-101 */
  if (rows == 0 || cols == 0) return OR_ERR_INVALID_MATRIX_DIMENSION;
  const unsigned slot = 32u / cf;
  const uint32_t mask = (1u << b) - 1u;
  const uint64_t ocols = (cols + cf - 1) / cf; /* div_ceil, matrix.rs:112,140,172 */
#pragma omp parallel for schedule(static)
  for (uint64_t r = 0; r < rows; r++) {
    for (uint64_t c = 0; c < ocols; c++) {
      const uint64_t d0 = c * cf;
      uint32_t w = in[r * cols + d0] & mask; /* matrix.rs:121,149,181 */
      for (unsigned j = 1; j < cf; j++)
        if (d0 + j < cols) w |= (in[r * cols + d0 + j] & mask) << (j * slot); /* matrix.rs:123-125,151-157,183-193 */
      out[r * ocols + c] = w;
    }
  }
  return OR_OK;
}

int or_row_wise_decompress(const uint32_t* in, uint64_t rows, uint64_t in_cols, unsigned b, uint64_t num_cols,
                           uint32_t* out) {
  /* matrix.rs:207-316 */
  const unsigned cf = or_compression_factor(b);
  if (cf == 0) return OR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH;
  if ((num_cols + cf - 1) / cf != in_cols) return OR_ERR_INVALID_NUMBER_OF_ELEMENTS; /* assert_eq! at :221,250,282 */
  const unsigned slot = 32u / cf;
  const uint32_t mask = (1u << b) - 1u;
  for (uint64_t r = 0; r < rows; r++)
    for (uint64_t c = 0; c < in_cols; c++) {
      const uint32_t w = in[r * in_cols + c];
      for (unsigned j = 0; j < cf; j++)
        if (c * cf + j < num_cols) out[r * num_cols + c * cf + j] = (w >> (j * slot)) & mask;
    }
  return OR_OK;
}

HOT static void respond_rows(const uint32_t* q, uint64_t q_cols, const uint32_t* rhs, uint64_t rhs_rows,
                             uint64_t rhs_cols, unsigned cf, unsigned b, uint32_t* out) {
  const unsigned slot = 32u / cf;
  const uint32_t mask = (1u << b) - 1u;
  /* matrix.rs:345 / 383 / 429: par_iter_mut over the output elements, one sequential fold per output.  Static schedule:
   * every row costs the same, and together with or_first_touch_copy() it keeps a thread on the rows it first touched
   * (NUMA-local on a multi-socket host), which is the best case for this CPU baseline. */
#pragma omp parallel for schedule(static)
  for (uint64_t c_idx = 0; c_idx < rhs_rows; c_idx++) {
    const uint32_t* row = rhs + c_idx * rhs_cols;
    uint32_t acc = 0;
    /* first (rhs.cols - 1) compressed elements: matrix.rs:350-358, 388-397, 434-444 */
    if (cf == 3) {
      for (uint64_t w = 0; w + 1 < rhs_cols; w++) {
        const uint32_t e = row[w];
        const uint32_t* qq = q + w * 3;
        acc += qq[0] * (e & mask) + qq[1] * ((e >> 10) & mask) + qq[2] * ((e >> 20) & mask);
      }
    } else if (cf == 2) {
      for (uint64_t w = 0; w + 1 < rhs_cols; w++) {
        const uint32_t e = row[w];
        const uint32_t* qq = q + w * 2;
        acc += qq[0] * (e & mask) + qq[1] * ((e >> 16) & mask);
      }
    } else {
      for (uint64_t w = 0; w + 1 < rhs_cols; w++) {
        const uint32_t e = row[w];
        const uint32_t* qq = q + w * 4;
        acc += qq[0] * (e & mask) + qq[1] * ((e >> 8) & mask) + qq[2] * ((e >> 16) & mask) + qq[3] * ((e >> 24) & mask);
      }
    }
    /* last compressed element with bounds checks on the query index: matrix.rs:360-375, 399-421, 446-475 */
    {
      const uint64_t w = rhs_cols - 1;
      const uint32_t e = row[w];
      uint64_t d = w * cf;
      acc += q[d] * (e & mask);
      for (unsigned j = 1; j < cf; j++) {
        d += 1;
        if (d < q_cols) acc += q[d] * ((e >> (j * slot)) & mask);
      }
    }
    out[c_idx] = acc;
  }
}

int or_row_vector_x_compressed_transposed_matrix(const uint32_t* q, uint64_t q_rows, uint64_t q_cols,
                                                 const uint32_t* rhs, uint64_t rhs_rows, uint64_t rhs_cols,
                                                 uint64_t decompressed_num_cols, unsigned b, uint32_t* out) {
  /* matrix.rs:328-485 */
  if (!(q_rows == 1 && q_cols == decompressed_num_cols))
    return OR_ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED; /* matrix.rs:329-331 */
  const unsigned cf = or_compression_factor(b);
  if (cf == 0) return OR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH; /* the reference panics here (matrix.rs:478-481) */
  if (rhs_rows == 0 || rhs_cols == 0) return OR_ERR_INVALID_MATRIX_DIMENSION;
  /* The reference indexes q unchecked (get_unchecked, matrix.rs:1019); it is only sound when the compressed
   * width matches the decompressed width, which Server::setup guarantees (server.rs:66-67). */
  if ((decompressed_num_cols + cf - 1) / cf != rhs_cols) return OR_ERR_INVALID_NUMBER_OF_ELEMENTS;
  respond_rows(q, q_cols, rhs, rhs_rows, rhs_cols, cf, b, out);
  return OR_OK;
}

size_t or_matrix_num_bytes(uint64_t rows, uint64_t cols) { return 8u + (size_t)(rows * cols) * 4u; /* matrix.rs:94-96 */ }

int or_matrix_to_bytes(const uint32_t* elems, uint32_t rows, uint32_t cols, uint8_t* out, size_t out_cap) {
  /* matrix.rs:947-971 */
  const size_t n = or_matrix_num_bytes(rows, cols);
  if (out_cap < n) return OR_ERR_BUFFER_TOO_SMALL;
  out[0] = (uint8_t)rows; out[1] = (uint8_t)(rows >> 8); out[2] = (uint8_t)(rows >> 16); out[3] = (uint8_t)(rows >> 24);
  out[4] = (uint8_t)cols; out[5] = (uint8_t)(cols >> 8); out[6] = (uint8_t)(cols >> 16); out[7] = (uint8_t)(cols >> 24);
  memcpy(out + 8, elems, n - 8);
  return OR_OK;
}

int or_matrix_from_bytes(const uint8_t* bytes, size_t len, uint32_t* rows, uint32_t* cols) {
  /* matrix.rs:973-1010 */
  if (len <= 8) return OR_ERR_FAILED_TO_DESERIALIZE_MATRIX; /* :978-980 */
  const uint32_t r = (uint32_t)bytes[0] | ((uint32_t)bytes[1] << 8) | ((uint32_t)bytes[2] << 16) | ((uint32_t)bytes[3] << 24);
  const uint32_t c = (uint32_t)bytes[4] | ((uint32_t)bytes[5] << 8) | ((uint32_t)bytes[6] << 16) | ((uint32_t)bytes[7] << 24);
  /* the reference computes rows*cols in u32 (:988); a wire matrix that overflows that is not representable there.
   * 64-bit here; rows*cols == 0 is rejected exactly as :990-992. */
  const uint64_t num = (uint64_t)r * (uint64_t)c;
  if (num == 0) return OR_ERR_FAILED_TO_DESERIALIZE_MATRIX;
  if (num * 4u != (uint64_t)(len - 8)) return OR_ERR_FAILED_TO_DESERIALIZE_MATRIX; /* :994-999 */
  *rows = r;
  *cols = c;
  return OR_OK;
}

/* ============================================================================================
 * Server
 * ============================================================================================ */

static uint64_t isqrt_u64(uint64_t n) {
  /* usize::isqrt (server.rs:197): floor(sqrt(n)) */
  if (n == 0) return 0;
  uint64_t x = (uint64_t)sqrt((double)n);
  while (x * x > n) x--;
  while ((x + 1) * (x + 1) <= n) x++;
  return x;
}

int or_find_encoded_db_matrix_element_bit_length(uint64_t db_entry_count, unsigned* bit_len) {
  /* server.rs:193-218 */
  const unsigned __int128 Q = ((unsigned __int128)1) << 32; /* u32::MAX + 1 */
  const uint64_t sq = isqrt_u64(db_entry_count);
  uint64_t b = 0;
  unsigned __int128 rho = 1;
  while (Q >= (8 * rho * rho) * (unsigned __int128)sq) { /* :202 (usize math in the reference; cannot overflow before exit for sq >= 1) */
    b += 1;
    rho = ((unsigned __int128)1) << b;
    if (b > 64) break; /* sq == 0 would loop forever in the reference too; callers reject empty DBs first (server.rs:49-51) */
  }
  b = (b == 0) ? 0 : b - 1; /* overflowing_sub, :207-210 */
  if (b >= 4) { /* MIN_MAT_ELEM_BIT_LEN, :213 */
    *bit_len = (unsigned)b;
    return OR_OK;
  }
  return OR_ERR_KV_DATABASE_SIZE_TOO_LARGE;
}

int or_server_respond(const uint32_t* dtc, uint64_t dtc_rows, uint64_t dtc_cols, uint64_t decompressed_num_cols,
                      unsigned b, const uint8_t* query, size_t query_len, uint8_t* response, size_t response_cap,
                      size_t* response_len) {
  /* server.rs:184-190 */
  uint32_t qr, qc;
  int rc = or_matrix_from_bytes(query, query_len, &qr, &qc);
  if (rc != OR_OK) return rc;
  const size_t need = or_matrix_num_bytes(1, dtc_rows);
  if (response_cap < need) return OR_ERR_BUFFER_TOO_SMALL;
  uint32_t* q = (uint32_t*)malloc((size_t)qr * qc * 4u); /* from_bytes copies (matrix.rs:1007); also fixes alignment */
  uint32_t* r = (uint32_t*)malloc((size_t)dtc_rows * 4u);
  memcpy(q, query + 8, (size_t)qr * qc * 4u);
  rc = or_row_vector_x_compressed_transposed_matrix(q, qr, qc, dtc, dtc_rows, dtc_cols, decompressed_num_cols, b, r);
  if (rc == OR_OK) {
    rc = or_matrix_to_bytes(r, 1, (uint32_t)dtc_rows, response, response_cap);
    *response_len = need;
  }
  free(q);
  free(r);
  return rc;
}

int or_server_setup_from_matrix(const uint8_t seed_mu[32], const uint32_t* D, uint64_t N, uint64_t C, unsigned b,
                                uint32_t* hint, uint32_t* dtc) {
  /* server.rs:59-67 with D given */
  if (or_compression_factor(b) == 0) return OR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH;
  uint32_t* A = (uint32_t*)malloc((size_t)OR_LWE_DIMENSION * N * 4u);
  if (!A) return OR_ERR_BUFFER_TOO_SMALL;
  int rc = or_generate_from_seed(OR_LWE_DIMENSION, N, seed_mu, A); /* :59 */
  if (rc == OR_OK) rc = or_mul(A, OR_LWE_DIMENSION, N, D, N, C, hint); /* :61 */
  free(A);
  if (rc != OR_OK) return rc;
  uint32_t* Dt = (uint32_t*)malloc((size_t)N * C * 4u);
  if (!Dt) return OR_ERR_BUFFER_TOO_SMALL;
  rc = or_transpose(D, N, C, Dt); /* :64 */
  if (rc == OR_OK) rc = or_row_wise_compress(Dt, C, N, b, dtc); /* :67 */
  free(Dt);
  return rc;
}

/* ============================================================================================
 * Binary fuse filter
 * ============================================================================================ */

uint32_t or_bff_segment_length(uint32_t arity, uint32_t size) {
  /* bff.rs:519-529 */
  if (size == 0) return 4;
  double e;
  if (arity == 3) e = floor(log((double)size) / log(3.33) + 2.25);
  else if (arity == 4) e = floor(log((double)size) / log(2.91) - 0.5);
  else return 65536;
  /* Rust's `f64 as usize` saturates: a negative exponent (4-wise, size == 1: floor(-0.5) = -1) becomes 0 */
  if (e < 0) e = 0;
  return 1u << (unsigned)e;
}

double or_bff_size_factor(uint32_t arity, uint32_t size) {
  /* bff.rs:532-538 */
  if (arity == 3) return fmax(1.125, 0.875 + 0.25 * log(1e6) / log((double)size));
  if (arity == 4) return fmax(1.075, 0.77 + 0.305 * log(6e5) / log((double)size));
  return 2.0;
}

int or_bff_shape(uint32_t arity, uint64_t db_size, uint32_t* segment_length, uint32_t* segment_count_length,
                 uint64_t* num_fingerprints) {
  /* bff.rs:47-67 (3-wise), 256-276 (4-wise) */
  if (arity != 3 && arity != 4) return OR_ERR_UNSUPPORTED_ARITY;
  if (db_size == 0) return OR_ERR_EMPTY_KV_DATABASE;
  uint32_t seg = or_bff_segment_length(arity, (uint32_t)db_size);
  if (seg > (1u << 18)) seg = 1u << 18; /* .min(1 << 18) */
  const double sf = or_bff_size_factor(arity, (uint32_t)db_size);
  const uint32_t capacity = db_size > 1 ? (uint32_t)round((double)db_size * sf) : 0;
  const uint32_t init_segment_count = (capacity + seg - 1) / seg;
  uint32_t array_len = init_segment_count * seg;
  const uint32_t proposed = (array_len + seg - 1) / seg;
  const uint32_t segment_count = proposed < arity ? 1 : proposed - (arity - 1);
  array_len = (segment_count + arity - 1) * seg;
  *segment_length = seg;
  *segment_count_length = segment_count * seg;
  *num_fingerprints = array_len;
  return OR_OK;
}

uint64_t or_murmur64(uint64_t h) {
  /* bff.rs:553-560 */
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return h;
}

uint64_t or_mix(uint64_t key, uint64_t seed) { return or_murmur64(key + seed); /* bff.rs:563-565 */ }

static uint64_t le64(const uint8_t* p) {
  uint64_t v = 0;
  for (unsigned i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
  return v;
}

void or_hash_of_key(const uint8_t* key, size_t key_len, uint64_t out[4]) {
  /* bff.rs:568-584 */
  uint8_t d[32];
  or_turboshake128_hash(key, key_len, 0x1F, d, 32);
  for (unsigned i = 0; i < 4; i++) out[i] = le64(d + 8 * i);
}

uint64_t or_mix256(const uint64_t key[4], const uint8_t seed[32]) {
  /* bff.rs:588-601 */
  uint64_t sw[4];
  for (unsigned i = 0; i < 4; i++) sw[i] = le64(seed + 8 * i);
  uint64_t total = 0;
  for (unsigned k = 0; k < 4; k++) {
    uint64_t acc = 0;
    for (unsigned s = 0; s < 4; s++) acc = or_murmur64(acc + or_mix(key[k], sw[s]));
    total += acc;
  }
  return total;
}

void or_hash_batch_3(uint64_t hash, uint32_t segment_length, uint32_t segment_count_length, uint32_t h[3]) {
  /* bff.rs:605-617 */
  const uint32_t m = segment_length - 1;
  const uint64_t hi = (uint64_t)(((unsigned __int128)hash * (unsigned __int128)segment_count_length) >> 64);
  h[0] = (uint32_t)hi;
  h[1] = h[0] + segment_length;
  h[2] = h[1] + segment_length;
  h[1] ^= ((uint32_t)(hash >> 18)) & m;
  h[2] ^= ((uint32_t)hash) & m;
}

void or_hash_batch_4(uint64_t hash, uint32_t segment_length, uint32_t segment_count_length, uint32_t h[4]) {
  /* bff.rs:621-635 */
  const uint32_t m = segment_length - 1;
  const uint64_t hi = (uint64_t)(((unsigned __int128)hash * (unsigned __int128)segment_count_length) >> 64);
  h[0] = (uint32_t)hi;
  h[1] = h[0] + segment_length;
  h[2] = h[1] + segment_length;
  h[3] = h[2] + segment_length;
  h[1] ^= ((uint32_t)hash) & m;
  h[2] ^= ((uint32_t)(hash >> 16)) & m;
  h[3] ^= ((uint32_t)(hash >> 32)) & m;
}

static void put_le32(uint8_t* p, uint32_t v) { for (unsigned i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static void put_le64(uint8_t* p, uint64_t v) { for (unsigned i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

void or_bff_to_bytes(const or_bff* f, uint8_t out[OR_BFF_BYTE_LEN]) {
  /* bff.rs:462-486: seed(32) arity(4) segment_length(4) segment_count_length(4) num_fingerprints(usize=8)
   * filter_size(8) mat_elem_bit_len(8) = 68 bytes on a 64-bit target */
  memcpy(out, f->seed, 32);
  put_le32(out + 32, f->arity);
  put_le32(out + 36, f->segment_length);
  put_le32(out + 40, f->segment_count_length);
  put_le64(out + 44, f->num_fingerprints);
  put_le64(out + 52, f->filter_size);
  put_le64(out + 60, f->mat_elem_bit_len);
}

int or_bff_from_bytes(const uint8_t* bytes, size_t len, or_bff* f) {
  /* bff.rs:488-513 */
  if (len != OR_BFF_BYTE_LEN) return OR_ERR_FAILED_TO_DESERIALIZE_FILTER;
  memcpy(f->seed, bytes, 32);
  f->arity = le32(bytes + 32);
  f->segment_length = le32(bytes + 36);
  f->segment_count_length = le32(bytes + 40);
  f->num_fingerprints = le64(bytes + 44);
  f->filter_size = le64(bytes + 52);
  f->mat_elem_bit_len = le64(bytes + 60);
  return OR_OK;
}

/* ---- tiny open-addressing map hash(u64) -> key index, standing in for HashMap<u64,&[u8]> (bff.rs:76,125) ---- */
typedef struct {
  uint64_t* keys;
  uint64_t* vals;
  uint8_t* used;
  uint64_t cap;
} u64map;

static int u64map_init(u64map* m, uint64_t n) {
  uint64_t cap = 16;
  while (cap < n * 2 + 2) cap <<= 1;
  m->cap = cap;
  m->keys = (uint64_t*)malloc(cap * 8);
  m->vals = (uint64_t*)malloc(cap * 8);
  m->used = (uint8_t*)calloc(cap, 1);
  return (m->keys && m->vals && m->used) ? 0 : -1;
}
static void u64map_clear(u64map* m) { memset(m->used, 0, m->cap); }
static void u64map_free(u64map* m) { free(m->keys); free(m->vals); free(m->used); }
static void u64map_put(u64map* m, uint64_t k, uint64_t v) {
  uint64_t i = or_murmur64(k) & (m->cap - 1);
  while (m->used[i] && m->keys[i] != k) i = (i + 1) & (m->cap - 1);
  m->used[i] = 1; m->keys[i] = k; m->vals[i] = v; /* HashMap::insert overwrites on equal key (bff.rs:125) */
}
static uint64_t u64map_get(const u64map* m, uint64_t k) {
  uint64_t i = or_murmur64(k) & (m->cap - 1);
  while (m->used[i]) {
    if (m->keys[i] == k) return m->vals[i];
    i = (i + 1) & (m->cap - 1);
  }
  return UINT64_MAX;
}

static uint8_t mod3(uint8_t x) { return x > 2 ? (uint8_t)(x - 3) : x; } /* bff.rs:541-543 */
static uint8_t mod4(uint8_t x) { return x > 3 ? (uint8_t)(x - 4) : x; } /* bff.rs:546-548 */

/* bff.rs:40-235 (arity 3) and 249-456 (arity 4), folded on `arity` where the two are textually parallel. */
static int bff_construct(uint32_t arity, const or_kv_db* db, unsigned b, const uint8_t* filter_seeds,
                         uint32_t max_attempts, or_bff* out, uint64_t** out_reverse_order, uint8_t** out_reverse_h,
                         u64map* hash_to_key, uint32_t* attempts_used) {
  const uint64_t db_size = db->num_pairs;
  if (db_size == 0) return OR_ERR_EMPTY_KV_DATABASE;
  uint32_t segment_length, segment_count_length;
  uint64_t num_fingerprints;
  int rc = or_bff_shape(arity, db_size, &segment_length, &segment_count_length, &num_fingerprints);
  if (rc != OR_OK) return rc;
  const uint32_t segment_count = segment_count_length / segment_length;

  uint32_t* alone = (uint32_t*)calloc(num_fingerprints, 4);
  uint8_t* t2count = (uint8_t*)calloc(num_fingerprints, 1);
  uint64_t* t2hash = (uint64_t*)calloc(num_fingerprints, 8);
  uint8_t* reverse_h = (uint8_t*)calloc(db_size, 1);
  uint64_t* reverse_order = (uint64_t*)calloc(db_size + 1, 8);
  reverse_order[db_size] = 1; /* bff.rs:74 */

  unsigned block_bits = 1; /* bff.rs:78-84 */
  while ((1u << block_bits) < segment_count) block_bits++;
  const uint64_t block_bits_mask = (1ull << block_bits) - 1;
  const uint64_t start_pos_len = 1ull << block_bits;
  uint64_t* start_pos = (uint64_t*)calloc(start_pos_len, 8);

  /* the reference hashes every key with TurboSHAKE128 once per attempt (bff.rs:113); the digest does not depend on
   * the attempt, so it is computed once here. */
  uint64_t* hashed = (uint64_t*)malloc(db_size * 4 * 8);
  for (uint64_t i = 0; i < db_size; i++)
    or_hash_of_key(db->keys + db->key_off[i], (size_t)(db->key_off[i + 1] - db->key_off[i]), hashed + 4 * i);

  int done = 0;
  uint64_t ultimate_size = 0;
  uint8_t seed[32];
  memset(seed, 0, 32);
  uint32_t attempt = 0;
  for (; attempt < max_attempts; attempt++) {
    memcpy(seed, filter_seeds + 32 * (size_t)attempt, 32); /* bff.rs:103-106: rng.fill_bytes(&mut seed) */
    for (uint64_t idx = 0; idx < start_pos_len; idx++)
      start_pos[idx] = (uint64_t)(((unsigned __int128)idx * db_size) >> block_bits); /* bff.rs:108-110 (u64 mul in ref) */
    u64map_clear(hash_to_key);

    for (uint64_t ki = 0; ki < db_size; ki++) { /* bff.rs:112-126 */
      const uint64_t hash = or_mix256(hashed + 4 * ki, seed);
      uint64_t segment_index = hash >> (64 - block_bits);
      while (reverse_order[start_pos[segment_index]] != 0) {
        segment_index += 1;
        segment_index &= block_bits_mask;
      }
      reverse_order[start_pos[segment_index]] = hash;
      start_pos[segment_index] += 1;
      u64map_put(hash_to_key, hash, ki);
    }

    int error = 0;
    if (arity == 3) { /* bff.rs:128-153 */
      for (uint64_t i = 0; i < db_size; i++) {
        const uint64_t hash = reverse_order[i];
        uint32_t h[3];
        or_hash_batch_3(hash, segment_length, segment_count_length, h);
        t2count[h[0]] += 4; t2hash[h[0]] ^= hash;
        t2count[h[1]] += 4; t2count[h[1]] ^= 1; t2hash[h[1]] ^= hash;
        t2count[h[2]] += 4; t2count[h[2]] ^= 2; t2hash[h[2]] ^= hash;
        error = t2count[h[0]] < 4 || t2count[h[1]] < 4 || t2count[h[2]] < 4; /* NB: only the last key's flag survives, as in the reference (:144) */
      }
    } else { /* bff.rs:337-368 */
      uint8_t count_mask = 0;
      for (uint64_t i = 0; i < db_size; i++) {
        const uint64_t hash = reverse_order[i];
        uint32_t h[4];
        or_hash_batch_4(hash, segment_length, segment_count_length, h);
        t2count[h[0]] += 4; t2hash[h[0]] ^= hash; count_mask |= t2count[h[0]];
        t2count[h[1]] += 4; t2count[h[1]] ^= 1; t2hash[h[1]] ^= hash; count_mask |= t2count[h[1]];
        t2count[h[2]] += 4; t2count[h[2]] ^= 2; t2hash[h[2]] ^= hash; count_mask |= t2count[h[2]];
        t2count[h[3]] += 4; t2count[h[3]] ^= 3; t2hash[h[3]] ^= hash; count_mask |= t2count[h[3]];
      }
      error = count_mask >= 0x80;
    }
    if (error) {
      memset(reverse_order, 0, db_size * 8);
      memset(t2count, 0, num_fingerprints);
      memset(t2hash, 0, num_fingerprints * 8);
      continue;
    }

    uint64_t qsize = 0; /* bff.rs:155-161 / 370-376 */
    for (uint64_t idx = 0; idx < num_fingerprints; idx++) {
      alone[qsize] = (uint32_t)idx;
      if ((t2count[idx] >> 2) == 1) qsize++;
    }

    uint64_t stack_size = 0; /* bff.rs:163-203 / 378-424 */
    while (qsize > 0) {
      qsize--;
      const uint64_t index = alone[qsize];
      if ((t2count[index] >> 2) == 1) {
        const uint64_t hash = t2hash[index];
        const uint8_t found = t2count[index] & 3;
        reverse_h[stack_size] = found;
        reverse_order[stack_size] = hash;
        stack_size++;
        if (arity == 3) {
          uint32_t h[3], h012[5];
          or_hash_batch_3(hash, segment_length, segment_count_length, h);
          h012[1] = h[1]; h012[2] = h[2]; h012[3] = h[0]; h012[4] = h012[1];
          for (unsigned step = 1; step <= 2; step++) {
            const uint64_t other = h012[found + step];
            alone[qsize] = (uint32_t)other;
            if ((t2count[other] >> 2) == 2) qsize++;
            t2count[other] -= 4;
            t2count[other] ^= mod3((uint8_t)(found + step));
            t2hash[other] ^= hash;
          }
        } else {
          uint32_t h[4], h0123[7];
          or_hash_batch_4(hash, segment_length, segment_count_length, h);
          h0123[1] = h[1]; h0123[2] = h[2]; h0123[3] = h[3]; h0123[4] = h[0]; h0123[5] = h0123[1]; h0123[6] = h0123[2];
          for (unsigned step = 1; step <= 3; step++) {
            const uint64_t other = h0123[found + step];
            alone[qsize] = (uint32_t)other;
            qsize += ((t2count[other] >> 2) == 2) ? 1 : 0;
            t2count[other] -= 4;
            t2count[other] ^= mod4((uint8_t)(found + step));
            t2hash[other] ^= hash;
          }
        }
      }
    }

    if (stack_size == db_size) { /* bff.rs:205-210 / 426-431 */
      ultimate_size = stack_size;
      done = 1;
      break;
    }
    memset(reverse_order, 0, db_size * 8);
    memset(t2count, 0, num_fingerprints);
    memset(t2hash, 0, num_fingerprints * 8);
  }

  free(alone); free(t2count); free(t2hash); free(start_pos); free(hashed);
  if (!done) {
    free(reverse_h); free(reverse_order);
    return arity == 3 ? OR_ERR_EXHAUSTED_ATTEMPTS_3WISE : OR_ERR_EXHAUSTED_ATTEMPTS_4WISE;
  }
  memcpy(out->seed, seed, 32);
  out->arity = arity;
  out->segment_length = segment_length;
  out->segment_count_length = segment_count_length;
  out->num_fingerprints = num_fingerprints;
  out->filter_size = ultimate_size;
  out->mat_elem_bit_len = b;
  *out_reverse_order = reverse_order;
  *out_reverse_h = reverse_h;
  if (attempts_used) *attempts_used = attempt + 1;
  return OR_OK;
}

uint64_t or_encoded_num_cols(uint64_t max_value_byte_len, unsigned b) {
  /* matrix.rs:694-700: (HASHED_KEY_BIT_LEN + max_value_bit_len + 8).div_ceil(mat_elem_bit_len) */
  return (256u + max_value_byte_len * 8u + 8u + b - 1) / b;
}

int or_from_kv_database(uint32_t arity, const or_kv_db* db, unsigned b, const uint8_t* filter_seeds,
                        uint32_t max_attempts, or_bff* out_filter, uint32_t* mat, uint64_t mat_rows, uint64_t mat_cols,
                        uint32_t* attempts_used) {
  /* matrix.rs:633-648 -> 687-755 (3-wise) / 819-894 (4-wise) */
  if (arity != 3 && arity != 4) return OR_ERR_UNSUPPORTED_ARITY;
  if (db->num_pairs == 0) return OR_ERR_EMPTY_KV_DATABASE;
  u64map h2k;
  if (u64map_init(&h2k, db->num_pairs) != 0) return OR_ERR_BUFFER_TOO_SMALL;
  uint64_t* reverse_order = NULL;
  uint8_t* reverse_h = NULL;
  int rc = bff_construct(arity, db, b, filter_seeds, max_attempts, out_filter, &reverse_order, &reverse_h, &h2k, attempts_used);
  if (rc != OR_OK) { u64map_free(&h2k); return rc; }

  uint64_t max_value_byte_len = 0; /* matrix.rs:696 */
  for (uint64_t i = 0; i < db->num_pairs; i++) {
    const uint64_t l = db->val_off[i + 1] - db->val_off[i];
    if (l > max_value_byte_len) max_value_byte_len = l;
  }
  const uint64_t rows = out_filter->num_fingerprints;
  const uint64_t cols = or_encoded_num_cols(max_value_byte_len, b);
  if (rows != mat_rows || cols != mat_cols) { free(reverse_order); free(reverse_h); u64map_free(&h2k); return OR_ERR_BUFFER_TOO_SMALL; }
  memset(mat, 0, (size_t)(rows * cols) * 4u); /* Matrix::new zero-initialises, matrix.rs:702 */
  const uint32_t mask = (1u << b) - 1u;
  uint32_t* row = (uint32_t*)malloc((size_t)cols * 4u);

  for (uint64_t ii = out_filter->filter_size; ii-- > 0;) { /* (0..filter_size).rev(), matrix.rs:707 / 839 */
    const uint64_t hash = reverse_order[ii];
    const uint64_t ki = u64map_get(&h2k, hash);
    const uint8_t* key = db->keys + db->key_off[ki];
    const size_t key_len = (size_t)(db->key_off[ki + 1] - db->key_off[ki]);
    const uint8_t* val = db->values + db->val_off[ki];
    const size_t val_len = (size_t)(db->val_off[ki + 1] - db->val_off[ki]);
    const unsigned found = reverse_h[ii];
    uint32_t hs[7];
    uint64_t idx[4];
    if (arity == 3) {
      uint32_t h[3];
      or_hash_batch_3(hash, out_filter->segment_length, out_filter->segment_count_length, h);
      hs[0] = h[0]; hs[1] = h[1]; hs[2] = h[2]; hs[3] = hs[0]; hs[4] = hs[1]; /* matrix.rs:715-719 */
    } else {
      uint32_t h[4];
      or_hash_batch_4(hash, out_filter->segment_length, out_filter->segment_count_length, h);
      hs[0] = h[0]; hs[1] = h[1]; hs[2] = h[2]; hs[3] = h[3]; hs[4] = hs[0]; hs[5] = hs[1]; hs[6] = hs[2]; /* matrix.rs:847-853 */
    }
    for (unsigned j = 0; j < arity; j++) idx[j] = hs[found + j];
    or_encode_kv_as_row(key, key_len, val, val_len, b, cols, row); /* matrix.rs:721 / 855 */
    for (uint64_t e = 0; e < cols; e++) {
      /* matrix.rs:727-740 / 862-879: subtract the other slots' fingerprints (masking from the 2nd on), then the hash mask */
      uint32_t v = row[e] - mat[idx[1] * cols + e];
      for (unsigned j = 2; j < arity; j++) v = (v - mat[idx[j] * cols + e]) & mask;
      const uint32_t m = ((uint32_t)or_mix(hash, e)) & mask;
      row[e] = (v - m) & mask;
    }
    memcpy(mat + idx[0] * cols, row, (size_t)cols * 4u); /* matrix.rs:742-745 / 881-884 */
  }
  free(row); free(reverse_order); free(reverse_h); u64map_free(&h2k);
  return OR_OK;
}

int or_recover_value(const uint32_t* mat, uint64_t mat_rows, uint64_t mat_cols, const or_bff* filter, const uint8_t* key,
                     size_t key_len, uint8_t* value, size_t value_cap, size_t* value_len) {
  /* matrix.rs:768-805 (3-wise) / 907-945 (4-wise) */
  (void)mat_rows;
  const unsigned b = (unsigned)filter->mat_elem_bit_len;
  const uint32_t mask = (1u << b) - 1u;
  uint64_t hk[4];
  or_hash_of_key(key, key_len, hk);
  const uint64_t hash = or_mix256(hk, filter->seed);
  uint32_t h[4];
  if (filter->arity == 3) or_hash_batch_3(hash, filter->segment_length, filter->segment_count_length, h);
  else or_hash_batch_4(hash, filter->segment_length, filter->segment_count_length, h);
  uint32_t* row = (uint32_t*)malloc((size_t)mat_cols * 4u);
  for (uint64_t e = 0; e < mat_cols; e++) {
    uint32_t v = 0;
    for (unsigned j = 0; j < filter->arity; j++) v += mat[(uint64_t)h[j] * mat_cols + e];
    row[e] = (v + (((uint32_t)or_mix(hash, e)) & mask)) & mask;
  }
  const size_t cap = (size_t)(mat_cols * b / 8);
  uint8_t* kv = (uint8_t*)malloc(cap + 8);
  size_t kv_len = 0;
  int rc = or_decode_kv_from_row(row, mat_cols, b, kv, cap + 8, &kv_len);
  free(row);
  if (rc == OR_OK) {
    uint8_t hkb[32];
    for (unsigned i = 0; i < 4; i++) put_le64(hkb + 8 * i, hk[i]);
    uint8_t acc = 0;
    for (unsigned i = 0; i < 32; i++) acc ^= (uint8_t)(kv[i] ^ hkb[i]); /* client.rs:252: an XOR fold, as the reference has it */
    if (acc != 0) rc = OR_ERR_DECODED_ROW_NOT_PREPENDED_WITH_DIGEST;
    else if (kv_len - 32 > value_cap) rc = OR_ERR_BUFFER_TOO_SMALL;
    else { memcpy(value, kv + 32, kv_len - 32); *value_len = kv_len - 32; }
  }
  free(kv);
  return rc;
}

/* ============================================================================================
 * Row codec
 * ============================================================================================ */

static uint64_t u64_from_le_bytes(const uint8_t* p, size_t n) {
  /* serialization.rs:199-208 */
  uint64_t w = 0;
  if (n > 8) n = 8;
  for (size_t i = 0; i < n; i++) w |= (uint64_t)p[i] << (8 * i);
  return w;
}

static void pack_bytes_into_row(const uint8_t* src, size_t src_len, unsigned b, uint64_t elem_mask, uint64_t* buffer,
                                size_t* buf_num_bits, uint32_t* row, uint64_t* row_offset) {
  /* serialization.rs:42-69 (hashed key) and 71-98 (value): the same loop twice in the reference */
  size_t byte_offset = 0;
  while (byte_offset < src_len) {
    const size_t remaining = src_len - byte_offset;
    const size_t unset_bits = 64 - *buf_num_bits;
    const size_t fillable_bits = unset_bits & ~(size_t)7;
    size_t fillable_bytes = fillable_bits / 8;
    if (fillable_bytes > remaining) fillable_bytes = remaining;
    const size_t read_bits = fillable_bytes * 8;
    const uint64_t word = u64_from_le_bytes(src + byte_offset, fillable_bytes);
    byte_offset += fillable_bytes;
    /* `read_word << buf_num_bits`: buf_num_bits < 64 whenever fillable_bytes > 0; a 0-byte read contributes 0 */
    if (fillable_bytes > 0) *buffer |= word << *buf_num_bits;
    *buf_num_bits += read_bits;
    const size_t n_elems = *buf_num_bits / b;
    for (size_t e = 0; e < n_elems; e++) {
      row[*row_offset + e] = (uint32_t)(*buffer & elem_mask);
      *buffer >>= b;
      *buf_num_bits -= b;
    }
    *row_offset += n_elems;
  }
}

void or_encode_kv_as_row(const uint8_t* key, size_t key_len, const uint8_t* value, size_t value_len, unsigned b,
                         uint64_t num_cols, uint32_t* row) {
  /* serialization.rs:22-116 */
  uint8_t hashed_key[32];
  or_turboshake128_hash(key, key_len, 0x1F, hashed_key, 32);
  memset(row, 0, (size_t)num_cols * 4u);
  uint64_t row_offset = 0;
  const uint64_t elem_mask = (1ull << b) - 1;
  uint64_t buffer = 0;
  size_t buf_num_bits = 0;
  pack_bytes_into_row(hashed_key, 32, b, elem_mask, &buffer, &buf_num_bits, row, &row_offset);
  pack_bytes_into_row(value, value_len, b, elem_mask, &buffer, &buf_num_bits, row, &row_offset);
  buffer |= (uint64_t)0x81 << buf_num_bits; /* boundary mark, serialization.rs:100-102 */
  buf_num_bits += 8;
  while (buf_num_bits > 0) { /* serialization.rs:104-113 */
    const size_t readable = buf_num_bits < b ? buf_num_bits : b;
    row[row_offset] = (uint32_t)(buffer & elem_mask);
    buffer >>= readable;
    buf_num_bits -= readable;
    row_offset++;
  }
}

int or_decode_kv_from_row(const uint32_t* row, uint64_t num_cols, unsigned b, uint8_t* out, size_t out_cap, size_t* out_len) {
  /* serialization.rs:132-184 */
  const size_t num_extractable_bits = ((size_t)num_cols * b) & ~(size_t)7;
  const size_t num_bytes = num_extractable_bits / 8;
  if (out_cap < num_bytes) return OR_ERR_BUFFER_TOO_SMALL;
  memset(out, 0, num_bytes);
  const uint32_t mask = (1u << b) - 1u;
  uint64_t buffer = 0;
  size_t buf_num_bits = 0, byte_offset = 0;
  for (uint64_t row_offset = 0; row_offset < num_cols; row_offset++) {
    const size_t remaining_bits = num_extractable_bits - (byte_offset * 8 + buf_num_bits);
    const uint32_t sel = row[row_offset] & mask;
    buffer |= (uint64_t)sel << buf_num_bits;
    buf_num_bits += (b < remaining_bits) ? b : remaining_bits;
    const size_t decodable_bits = buf_num_bits & ~(size_t)7;
    const size_t decodable_bytes = decodable_bits / 8;
    for (size_t i = 0; i < decodable_bytes && i < 8; i++) out[byte_offset + i] = (uint8_t)(buffer >> (8 * i)); /* u64_to_le_bytes :220-226 */
    buffer = decodable_bits >= 64 ? 0 : buffer >> decodable_bits;
    buf_num_bits -= decodable_bits;
    byte_offset += decodable_bytes;
  }
  /* boundary search from the back: serialization.rs:164-183 */
  size_t pos = num_bytes;
  while (pos > 0 && out[pos - 1] != 0x81) pos--;
  if (pos == 0) return OR_ERR_ROW_NOT_DECODABLE;
  const size_t boundary = pos - 1;
  for (size_t i = boundary + 1; i < num_bytes; i++)
    if (out[i] != 0) return OR_ERR_ROW_NOT_DECODABLE;
  if (!(boundary > 32)) return OR_ERR_ROW_NOT_DECODABLE; /* `boundary_idx_from_front > 32`, :172 */
  *out_len = boundary; /* kv.truncate(boundary_idx_from_front) */
  return OR_OK;
}

/* ============================================================================================
 * Client (end-to-end check only)
 * ============================================================================================ */

int or_ternary_from_u32(uint32_t val, uint32_t* out) {
  /* matrix.rs:577-612 */
  const uint32_t INTERVAL = (UINT32_MAX - 2u) / 3u;
  const uint32_t REJ_MAX = INTERVAL * 3u;
  if (val > REJ_MAX) return 0;
  if (val <= INTERVAL) *out = 0;
  else if (val <= 2u * INTERVAL) *out = 1;
  else *out = UINT32_MAX;
  return 1;
}

static uint32_t query_indicator(unsigned b) { return (uint32_t)((1ull << 32) / (1ull << b)); /* client.rs:277-282 */ }

int or_client_query(const uint32_t* A, const uint32_t* hint, uint64_t N, uint64_t C, const or_bff* filter,
                    const uint8_t* key, size_t key_len, const uint32_t* secret_s, const uint32_t* error_e,
                    uint32_t* query_b, uint32_t* secret_c) {
  /* client.rs:95-140 (3-wise) / 142-194 (4-wise) */
  if (filter->arity != 3 && filter->arity != 4) return OR_ERR_UNSUPPORTED_ARITY;
  int rc = or_mul(secret_s, 1, OR_LWE_DIMENSION, A, OR_LWE_DIMENSION, N, query_b); /* s * A, :106 */
  if (rc != OR_OK) return rc;
  for (uint64_t i = 0; i < N; i++) query_b[i] += error_e[i]; /* + e */
  rc = or_mul(secret_s, 1, OR_LWE_DIMENSION, hint, OR_LWE_DIMENSION, C, secret_c); /* c = s * M, :107 */
  if (rc != OR_OK) return rc;
  uint64_t hk[4];
  or_hash_of_key(key, key_len, hk);
  const uint64_t hash = or_mix256(hk, filter->seed);
  uint32_t h[4];
  if (filter->arity == 3) or_hash_batch_3(hash, filter->segment_length, filter->segment_count_length, h);
  else or_hash_batch_4(hash, filter->segment_length, filter->segment_count_length, h);
  const uint32_t ind = query_indicator((unsigned)filter->mat_elem_bit_len);
  for (unsigned j = 0; j < filter->arity; j++) { /* overflowing_add checks, :115-134 */
    const uint32_t old = query_b[h[j]];
    const uint32_t sum = old + ind;
    if (sum < old) return OR_ERR_ARITHMETIC_OVERFLOW_ADDING_QUERY_INDICATOR;
    query_b[h[j]] = sum;
  }
  return OR_OK;
}

int or_client_process_response(const or_bff* filter, const uint8_t* key, size_t key_len, const uint32_t* secret_c,
                               const uint32_t* response, uint64_t C, uint8_t* value, size_t value_cap, size_t* value_len) {
  /* client.rs:209-275 */
  const unsigned b = (unsigned)filter->mat_elem_bit_len;
  const uint32_t rounding_factor = query_indicator(b);
  const uint32_t rounding_floor = rounding_factor / 2;
  const uint32_t mask = (1u << b) - 1u;
  uint64_t hk[4];
  or_hash_of_key(key, key_len, hk);
  const uint64_t hash = or_mix256(hk, filter->seed);
  uint32_t* row = (uint32_t*)malloc((size_t)C * 4u);
  for (uint64_t i = 0; i < C; i++) { /* :226-241 */
    const uint32_t unscaled = response[i] - secret_c[i];
    uint32_t rounded = unscaled / rounding_factor;
    if (unscaled % rounding_factor > rounding_floor) rounded += 1;
    row[i] = ((rounded & mask) + (uint32_t)or_mix(hash, i)) & mask;
  }
  const size_t cap = (size_t)(C * b / 8);
  uint8_t* kv = (uint8_t*)malloc(cap + 8);
  size_t kv_len = 0;
  int rc = or_decode_kv_from_row(row, C, b, kv, cap + 8, &kv_len);
  free(row);
  if (rc == OR_OK) {
    uint8_t hkb[32];
    for (unsigned i = 0; i < 4; i++) put_le64(hkb + 8 * i, hk[i]);
    uint8_t acc = 0;
    for (unsigned i = 0; i < 32; i++) acc ^= (uint8_t)(kv[i] ^ hkb[i]); /* client.rs:252: an XOR fold, as the reference has it */
    if (acc != 0) rc = OR_ERR_DECODED_ROW_NOT_PREPENDED_WITH_DIGEST;
    else if (kv_len - 32 > value_cap) rc = OR_ERR_BUFFER_TOO_SMALL;
    else { memcpy(value, kv + 32, kv_len - 32); *value_len = kv_len - 32; }
  }
  free(kv);
  return rc;
}

/* ============================================================================================
 * Synthetic inputs (shared definition with the device generator)
 * ============================================================================================ */

uint64_t or_synth_u64(uint64_t seed, uint64_t index) {
  /* splitmix64 finaliser over a Weyl sequence keyed by seed */
  uint64_t z = seed * 0xD1342543DE82EF95ULL + (index + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

void or_synth_fill_u32(uint32_t* out, uint64_t count, uint64_t seed, uint64_t index0, uint32_t mask) {
#pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < count; i++) out[i] = (uint32_t)(or_synth_u64(seed, index0 + i) >> 32) & mask;
}

void or_first_touch_copy(uint32_t* dst, const uint32_t* src, uint64_t rows, uint64_t cols) {
  /* copy a row-major matrix so that each row's pages are first touched by the thread that respond_rows() gives that row to */
#pragma omp parallel for schedule(static)
  for (uint64_t r = 0; r < rows; r++) memcpy(dst + r * cols, src + r * cols, (size_t)cols * 4u);
}

void or_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int or_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
