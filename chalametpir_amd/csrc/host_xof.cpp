// host_xof.cpp -- TurboSHAKE128 (RFC 9861) on the host: the XOF behind Matrix::generate_from_seed (reference
// chalametpir_common/src/matrix.rs:541-558), hash_of_key (binary_fuse_filter.rs:568-584) and encode_kv_as_row
// (serialization.rs:23-32).  The reference gets it from the `turboshake` crate (=0.4.1, not vendored);
// this is an independent implementation of the published algorithm: Keccak-p[1600, 12 rounds] sponge, rate 168 bytes,
// domain separation byte (0x1F in every reference call site) followed by pad10*1.
//
// Expanding the public matrix A is one sponge squeezed for 4*1774*N bytes (7.8 GiB at 2^20 keys): sequential by
// construction, so it stays on a host core and is overlapped with H2D + the device matmul (capi.hip).
// Two permutations, picked once at load time: on hosts with AVX-512VL every lane lives in the low qword of its own xmm register
// (32 of them against 16 general registers), so the 25-lane state spills far less, and xor3 / chi are ONE vpternlogq each and a
// rotate one vprolq: ~90 vector ops per round against ~130 scalar ones; whole squeezed blocks are written straight from the
// registers (keccak_squeeze_blocks_avx512vl).  Measured expanding A for 2^20 keys (8.37 GB) on the GPU box's EPYC 9575F:
// 4.27 s against 4.89 s for the scalar permutation (1.96 vs 1.71 GB/s).  Elsewhere the scalar permutation runs.
// (A plane-per-zmm variant -- vprolvq / vpermq / vpternlogq with a 5x5 qword transpose per round for pi -- was written earlier,
// verified and measured SLOWER than scalar on the GPU box's EPYC 9575F: its round is a chain of dependent cross-lane permutes.)
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "cpir_internal.hpp"

namespace cpir {
namespace {

constexpr unsigned kRate = 168;

constexpr uint64_t kRoundConstants[12] = {  // iota constants of Keccak-f[1600] rounds 12..23
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

inline uint64_t rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

// One permutation = 12 rounds, each written out on 25 named lanes (theta, rho+pi into B, chi+iota back into A).
// Multi-versioned: on x86-64-v3 hosts (every EPYC/Xeon a GPU box has) chi's ~b & c becomes one ANDN and the rotates RORX;
// the baseline clone keeps the library loadable anywhere.  The dispatch is resolved once at load time (ifunc).
#if defined(__clang__) && defined(__x86_64__)  // hipcc builds the library; gcc (sanitizer builds) lacks this clone syntax
__attribute__((target_clones("default", "arch=x86-64-v3")))
#endif
void keccak_p1600_12_scalar(uint64_t* A) {
  uint64_t a00 = A[0], a01 = A[1], a02 = A[2], a03 = A[3], a04 = A[4];
  uint64_t a05 = A[5], a06 = A[6], a07 = A[7], a08 = A[8], a09 = A[9];
  uint64_t a10 = A[10], a11 = A[11], a12 = A[12], a13 = A[13], a14 = A[14];
  uint64_t a15 = A[15], a16 = A[16], a17 = A[17], a18 = A[18], a19 = A[19];
  uint64_t a20 = A[20], a21 = A[21], a22 = A[22], a23 = A[23], a24 = A[24];
  for (int round = 0; round < 12; round++) {
    // theta
    const uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20;
    const uint64_t c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21;
    const uint64_t c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22;
    const uint64_t c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23;
    const uint64_t c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
    const uint64_t d0 = c4 ^ rotl(c1, 1);
    const uint64_t d1 = c0 ^ rotl(c2, 1);
    const uint64_t d2 = c1 ^ rotl(c3, 1);
    const uint64_t d3 = c2 ^ rotl(c4, 1);
    const uint64_t d4 = c3 ^ rotl(c0, 1);
    // rho + pi: B[y][2x+3y] = rot(A[x][y]); lanes indexed x + 5y
    const uint64_t b00 = a00 ^ d0;
    const uint64_t b01 = rotl(a06 ^ d1, 44);
    const uint64_t b02 = rotl(a12 ^ d2, 43);
    const uint64_t b03 = rotl(a18 ^ d3, 21);
    const uint64_t b04 = rotl(a24 ^ d4, 14);
    const uint64_t b05 = rotl(a03 ^ d3, 28);
    const uint64_t b06 = rotl(a09 ^ d4, 20);
    const uint64_t b07 = rotl(a10 ^ d0, 3);
    const uint64_t b08 = rotl(a16 ^ d1, 45);
    const uint64_t b09 = rotl(a22 ^ d2, 61);
    const uint64_t b10 = rotl(a01 ^ d1, 1);
    const uint64_t b11 = rotl(a07 ^ d2, 6);
    const uint64_t b12 = rotl(a13 ^ d3, 25);
    const uint64_t b13 = rotl(a19 ^ d4, 8);
    const uint64_t b14 = rotl(a20 ^ d0, 18);
    const uint64_t b15 = rotl(a04 ^ d4, 27);
    const uint64_t b16 = rotl(a05 ^ d0, 36);
    const uint64_t b17 = rotl(a11 ^ d1, 10);
    const uint64_t b18 = rotl(a17 ^ d2, 15);
    const uint64_t b19 = rotl(a23 ^ d3, 56);
    const uint64_t b20 = rotl(a02 ^ d2, 62);
    const uint64_t b21 = rotl(a08 ^ d3, 55);
    const uint64_t b22 = rotl(a14 ^ d4, 39);
    const uint64_t b23 = rotl(a15 ^ d0, 41);
    const uint64_t b24 = rotl(a21 ^ d1, 2);
    // chi (+ iota on lane 0)
    a00 = b00 ^ (~b01 & b02) ^ kRoundConstants[round];
    a01 = b01 ^ (~b02 & b03);
    a02 = b02 ^ (~b03 & b04);
    a03 = b03 ^ (~b04 & b00);
    a04 = b04 ^ (~b00 & b01);
    a05 = b05 ^ (~b06 & b07);
    a06 = b06 ^ (~b07 & b08);
    a07 = b07 ^ (~b08 & b09);
    a08 = b08 ^ (~b09 & b05);
    a09 = b09 ^ (~b05 & b06);
    a10 = b10 ^ (~b11 & b12);
    a11 = b11 ^ (~b12 & b13);
    a12 = b12 ^ (~b13 & b14);
    a13 = b13 ^ (~b14 & b10);
    a14 = b14 ^ (~b10 & b11);
    a15 = b15 ^ (~b16 & b17);
    a16 = b16 ^ (~b17 & b18);
    a17 = b17 ^ (~b18 & b19);
    a18 = b18 ^ (~b19 & b15);
    a19 = b19 ^ (~b15 & b16);
    a20 = b20 ^ (~b21 & b22);
    a21 = b21 ^ (~b22 & b23);
    a22 = b22 ^ (~b23 & b24);
    a23 = b23 ^ (~b24 & b20);
    a24 = b24 ^ (~b20 & b21);
  }
  A[0] = a00, A[1] = a01, A[2] = a02, A[3] = a03, A[4] = a04;
  A[5] = a05, A[6] = a06, A[7] = a07, A[8] = a08, A[9] = a09;
  A[10] = a10, A[11] = a11, A[12] = a12, A[13] = a13, A[14] = a14;
  A[15] = a15, A[16] = a16, A[17] = a17, A[18] = a18, A[19] = a19;
  A[20] = a20, A[21] = a21, A[22] = a22, A[23] = a23, A[24] = a24;
}

#if defined(__x86_64__)
// one lane per xmm register (low qword), AVX-512VL
#define CPIR_X3(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0x96)   /* a ^ b ^ c    */
#define CPIR_CHI(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0xD2)  /* a ^ (~b & c) */
#define CPIR_ROL(a, n) _mm_rol_epi64(a, n)
#define CPIR_VL_LOAD_STATE(A)                                                                                                      \
  __m128i a00 = LD(A, 0), a01 = LD(A, 1), a02 = LD(A, 2), a03 = LD(A, 3), a04 = LD(A, 4), a05 = LD(A, 5), a06 = LD(A, 6),               \
          a07 = LD(A, 7), a08 = LD(A, 8), a09 = LD(A, 9), a10 = LD(A, 10), a11 = LD(A, 11), a12 = LD(A, 12), a13 = LD(A, 13),            \
          a14 = LD(A, 14), a15 = LD(A, 15), a16 = LD(A, 16), a17 = LD(A, 17), a18 = LD(A, 18), a19 = LD(A, 19), a20 = LD(A, 20),         \
          a21 = LD(A, 21), a22 = LD(A, 22), a23 = LD(A, 23), a24 = LD(A, 24)
#define CPIR_VL_STORE_STATE(A)                                                                                                     \
  ST(A, 0, a00), ST(A, 1, a01), ST(A, 2, a02), ST(A, 3, a03), ST(A, 4, a04), ST(A, 5, a05), ST(A, 6, a06), ST(A, 7, a07),              \
      ST(A, 8, a08), ST(A, 9, a09), ST(A, 10, a10), ST(A, 11, a11), ST(A, 12, a12), ST(A, 13, a13), ST(A, 14, a14), ST(A, 15, a15),     \
      ST(A, 16, a16), ST(A, 17, a17), ST(A, 18, a18), ST(A, 19, a19), ST(A, 20, a20), ST(A, 21, a21), ST(A, 22, a22), ST(A, 23, a23),   \
      ST(A, 24, a24)
#define LD(A, i) _mm_loadl_epi64(reinterpret_cast<const __m128i*>((A) + (i)))
#define ST(A, i, v) _mm_storel_epi64(reinterpret_cast<__m128i*>((A) + (i)), v)
// theta: column parities and their rotations, d_x = c_{x-1} ^ rol(c_{x+1}, 1) folded into the lane xor (xor3);
// rho + pi with the same lane map as the scalar permutation above; chi (+ iota on lane 0)
#define CPIR_KECCAK_12_ROUNDS_VL \
  for (int round = 0; round < 12; round++) { \
    const __m128i c0 = CPIR_X3(CPIR_X3(a00, a05, a10), a15, a20), c1 = CPIR_X3(CPIR_X3(a01, a06, a11), a16, a21), \
                  c2 = CPIR_X3(CPIR_X3(a02, a07, a12), a17, a22), c3 = CPIR_X3(CPIR_X3(a03, a08, a13), a18, a23), \
                  c4 = CPIR_X3(CPIR_X3(a04, a09, a14), a19, a24); \
    const __m128i r0 = CPIR_ROL(c0, 1), r1 = CPIR_ROL(c1, 1), r2 = CPIR_ROL(c2, 1), r3 = CPIR_ROL(c3, 1), r4 = CPIR_ROL(c4, 1); \
    const __m128i b00 = CPIR_X3(a00, c4, r1), b01 = CPIR_ROL(CPIR_X3(a06, c0, r2), 44), b02 = CPIR_ROL(CPIR_X3(a12, c1, r3), 43), \
                  b03 = CPIR_ROL(CPIR_X3(a18, c2, r4), 21), b04 = CPIR_ROL(CPIR_X3(a24, c3, r0), 14); \
    const __m128i b05 = CPIR_ROL(CPIR_X3(a03, c2, r4), 28), b06 = CPIR_ROL(CPIR_X3(a09, c3, r0), 20), b07 = CPIR_ROL(CPIR_X3(a10, c4, r1), 3), \
                  b08 = CPIR_ROL(CPIR_X3(a16, c0, r2), 45), b09 = CPIR_ROL(CPIR_X3(a22, c1, r3), 61); \
    const __m128i b10 = CPIR_ROL(CPIR_X3(a01, c0, r2), 1), b11 = CPIR_ROL(CPIR_X3(a07, c1, r3), 6), b12 = CPIR_ROL(CPIR_X3(a13, c2, r4), 25), \
                  b13 = CPIR_ROL(CPIR_X3(a19, c3, r0), 8), b14 = CPIR_ROL(CPIR_X3(a20, c4, r1), 18); \
    const __m128i b15 = CPIR_ROL(CPIR_X3(a04, c3, r0), 27), b16 = CPIR_ROL(CPIR_X3(a05, c4, r1), 36), b17 = CPIR_ROL(CPIR_X3(a11, c0, r2), 10), \
                  b18 = CPIR_ROL(CPIR_X3(a17, c1, r3), 15), b19 = CPIR_ROL(CPIR_X3(a23, c2, r4), 56); \
    const __m128i b20 = CPIR_ROL(CPIR_X3(a02, c1, r3), 62), b21 = CPIR_ROL(CPIR_X3(a08, c2, r4), 55), b22 = CPIR_ROL(CPIR_X3(a14, c3, r0), 39), \
                  b23 = CPIR_ROL(CPIR_X3(a15, c4, r1), 41), b24 = CPIR_ROL(CPIR_X3(a21, c0, r2), 2); \
    a00 = _mm_xor_si128(CPIR_CHI(b00, b01, b02), _mm_cvtsi64_si128((long long)kRoundConstants[round])); \
    a01 = CPIR_CHI(b01, b02, b03), a02 = CPIR_CHI(b02, b03, b04), a03 = CPIR_CHI(b03, b04, b00), a04 = CPIR_CHI(b04, b00, b01); \
    a05 = CPIR_CHI(b05, b06, b07), a06 = CPIR_CHI(b06, b07, b08), a07 = CPIR_CHI(b07, b08, b09), a08 = CPIR_CHI(b08, b09, b05), a09 = CPIR_CHI(b09, b05, b06); \
    a10 = CPIR_CHI(b10, b11, b12), a11 = CPIR_CHI(b11, b12, b13), a12 = CPIR_CHI(b12, b13, b14), a13 = CPIR_CHI(b13, b14, b10), a14 = CPIR_CHI(b14, b10, b11); \
    a15 = CPIR_CHI(b15, b16, b17), a16 = CPIR_CHI(b16, b17, b18), a17 = CPIR_CHI(b17, b18, b19), a18 = CPIR_CHI(b18, b19, b15), a19 = CPIR_CHI(b19, b15, b16); \
    a20 = CPIR_CHI(b20, b21, b22), a21 = CPIR_CHI(b21, b22, b23), a22 = CPIR_CHI(b22, b23, b24), a23 = CPIR_CHI(b23, b24, b20), a24 = CPIR_CHI(b24, b20, b21); \
  }

__attribute__((target("avx512f,avx512vl"))) void keccak_p1600_12_avx512vl(uint64_t* A) {
  CPIR_VL_LOAD_STATE(A);
  CPIR_KECCAK_12_ROUNDS_VL
  CPIR_VL_STORE_STATE(A);
}

// `nblocks` squeezed blocks in one go: permute, write the 168 rate bytes straight from the registers, repeat -- the state never
// goes through memory between blocks (the block-at-a-time path stores 25 lanes, copies 168 bytes with wider loads that cannot be
// forwarded from those stores, and reloads)
__attribute__((target("avx512f,avx512vl"))) void keccak_squeeze_blocks_avx512vl(uint64_t* A, uint8_t* out, size_t nblocks) {
  CPIR_VL_LOAD_STATE(A);
  for (size_t blk = 0; blk < nblocks; blk++, out += kRate) {
    CPIR_KECCAK_12_ROUNDS_VL
    __m128i* o = reinterpret_cast<__m128i*>(out);
    _mm_storeu_si128(o + 0, _mm_unpacklo_epi64(a00, a01)), _mm_storeu_si128(o + 1, _mm_unpacklo_epi64(a02, a03));
    _mm_storeu_si128(o + 2, _mm_unpacklo_epi64(a04, a05)), _mm_storeu_si128(o + 3, _mm_unpacklo_epi64(a06, a07));
    _mm_storeu_si128(o + 4, _mm_unpacklo_epi64(a08, a09)), _mm_storeu_si128(o + 5, _mm_unpacklo_epi64(a10, a11));
    _mm_storeu_si128(o + 6, _mm_unpacklo_epi64(a12, a13)), _mm_storeu_si128(o + 7, _mm_unpacklo_epi64(a14, a15));
    _mm_storeu_si128(o + 8, _mm_unpacklo_epi64(a16, a17)), _mm_storeu_si128(o + 9, _mm_unpacklo_epi64(a18, a19));
    _mm_storel_epi64(o + 10, a20);
  }
  CPIR_VL_STORE_STATE(A);
}
#undef LD
#undef ST
#undef CPIR_VL_LOAD_STATE
#undef CPIR_VL_STORE_STATE
#undef CPIR_KECCAK_12_ROUNDS_VL
#undef CPIR_X3
#undef CPIR_CHI
#undef CPIR_ROL
#endif

using PermFn = void (*)(uint64_t*);
using SqueezeFn = void (*)(uint64_t*, uint8_t*, size_t);

void keccak_squeeze_blocks_generic(uint64_t* A, uint8_t* out, size_t nblocks);

// CPIR_XOF_SCALAR=1 in the environment forces the scalar permutation (A/B timing, tests of both paths)
PermFn pick_permutation() {
#if defined(__x86_64__)
  __builtin_cpu_init();
  const char* force = getenv("CPIR_XOF_SCALAR");
  if (!(force && force[0] == '1') && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl")) return keccak_p1600_12_avx512vl;
#endif
  return keccak_p1600_12_scalar;
}

const PermFn g_perm = pick_permutation();

inline void keccak_p1600_12(uint64_t* A) { g_perm(A); }

void keccak_squeeze_blocks_generic(uint64_t* A, uint8_t* out, size_t nblocks) {
  for (size_t blk = 0; blk < nblocks; blk++, out += kRate) {
    g_perm(A);
    memcpy(out, A, kRate);  // little-endian lanes (x86-64 host)
  }
}

SqueezeFn pick_squeeze() {
#if defined(__x86_64__)
  if (g_perm == keccak_p1600_12_avx512vl) return keccak_squeeze_blocks_avx512vl;
#endif
  return keccak_squeeze_blocks_generic;
}

const SqueezeFn g_squeeze_blocks = pick_squeeze();

}  // namespace

const char* xof_permutation_name() { return g_perm == keccak_p1600_12_scalar ? "scalar" : "avx512vl (lane per xmm)"; }

TurboShake128::TurboShake128() : pos(0) { memset(s, 0, sizeof(s)); }

void TurboShake128::absorb(const uint8_t* in, size_t len) {
  uint8_t* bytes = reinterpret_cast<uint8_t*>(s);  // little-endian lanes (x86-64 host)
  while (len) {
    size_t n = kRate - pos;
    if (n > len) n = len;
    for (size_t i = 0; i < n; i++) bytes[pos + i] ^= in[i];
    pos += (unsigned)n, in += n, len -= n;
    if (pos == kRate) keccak_p1600_12(s), pos = 0;
  }
}

void TurboShake128::finalize(uint8_t domain_sep) {
  uint8_t* bytes = reinterpret_cast<uint8_t*>(s);
  bytes[pos] ^= domain_sep;
  bytes[kRate - 1] ^= 0x80;
  keccak_p1600_12(s);
  pos = 0;
}

void TurboShake128::squeeze(uint8_t* out, size_t len) {
  const uint8_t* bytes = reinterpret_cast<const uint8_t*>(s);
  while (len) {
    if (pos == kRate && len >= kRate) {  // whole blocks: the state stays in registers from block to block
      const size_t nblocks = len / kRate;
      g_squeeze_blocks(s, out, nblocks);
      out += nblocks * kRate, len -= nblocks * kRate;  // the state now holds the last block written: pos stays kRate
      continue;
    }
    if (pos == kRate) keccak_p1600_12(s), pos = 0;
    size_t n = kRate - pos;
    if (n > len) n = len;
    memcpy(out, bytes + pos, n);
    pos += (unsigned)n, out += n, len -= n;
  }
}

void turboshake128(const uint8_t* msg, size_t len, uint8_t* out, size_t out_len) {
  TurboShake128 x;
  x.absorb(msg, len);
  x.finalize(0x1F);
  x.squeeze(out, out_len);
}

}  // namespace cpir
