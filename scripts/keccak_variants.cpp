// Tuning aid for host_xof.cpp: variants of the lane-per-xmm AVX-512VL Keccak-p[1600,12], timed on the host CPU of the box.
//   g++ -O3 -std=c++17 scripts/keccak_variants.cpp -o /tmp/kv && /tmp/kv        (or clang++)
// Prints ns per permutation for: one state (the product's loop), the same with the 12 rounds fully unrolled, and TWO independent
// states interleaved (if two cost much less than twice one, a single state is latency-bound, not port-bound).
#include <immintrin.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>

static constexpr uint64_t RC[12] = {0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#define X3(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0x96)
#define CHI(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0xD2)
#define ROL(a, n) _mm_rol_epi64(a, n)
#define XX(a, b) _mm_xor_si128(a, b)
#define ROLF(a, n) _mm_shldi_epi64(a, a, n)
#define DECL(P) __m128i P##00, P##01, P##02, P##03, P##04, P##05, P##06, P##07, P##08, P##09, P##10, P##11, P##12, P##13, P##14, P##15, P##16, P##17, P##18, P##19, P##20, P##21, P##22, P##23, P##24
#define LOADS(P, A)                                                                                                               \
  P##00 = L(A, 0), P##01 = L(A, 1), P##02 = L(A, 2), P##03 = L(A, 3), P##04 = L(A, 4), P##05 = L(A, 5), P##06 = L(A, 6), P##07 = L(A, 7), \
  P##08 = L(A, 8), P##09 = L(A, 9), P##10 = L(A, 10), P##11 = L(A, 11), P##12 = L(A, 12), P##13 = L(A, 13), P##14 = L(A, 14),         \
  P##15 = L(A, 15), P##16 = L(A, 16), P##17 = L(A, 17), P##18 = L(A, 18), P##19 = L(A, 19), P##20 = L(A, 20), P##21 = L(A, 21),       \
  P##22 = L(A, 22), P##23 = L(A, 23), P##24 = L(A, 24)
#define STORES(P, A)                                                                                                              \
  S(A, 0, P##00), S(A, 1, P##01), S(A, 2, P##02), S(A, 3, P##03), S(A, 4, P##04), S(A, 5, P##05), S(A, 6, P##06), S(A, 7, P##07),     \
  S(A, 8, P##08), S(A, 9, P##09), S(A, 10, P##10), S(A, 11, P##11), S(A, 12, P##12), S(A, 13, P##13), S(A, 14, P##14), S(A, 15, P##15), \
  S(A, 16, P##16), S(A, 17, P##17), S(A, 18, P##18), S(A, 19, P##19), S(A, 20, P##20), S(A, 21, P##21), S(A, 22, P##22), S(A, 23, P##23), \
  S(A, 24, P##24)
#define L(A, i) _mm_loadl_epi64((const __m128i*)((A) + (i)))
#define S(A, i, v) _mm_storel_epi64((__m128i*)((A) + (i)), v)
#define ROUND(a, rc)                                                                                                             \
  {                                                                                                                               \
    const __m128i c0 = X3(X3(a##00, a##05, a##10), a##15, a##20), c1 = X3(X3(a##01, a##06, a##11), a##16, a##21),                 \
                  c2 = X3(X3(a##02, a##07, a##12), a##17, a##22), c3 = X3(X3(a##03, a##08, a##13), a##18, a##23),                 \
                  c4 = X3(X3(a##04, a##09, a##14), a##19, a##24);                                                                 \
    const __m128i r0 = ROL(c0, 1), r1 = ROL(c1, 1), r2 = ROL(c2, 1), r3 = ROL(c3, 1), r4 = ROL(c4, 1);                            \
    const __m128i b00 = X3(a##00, c4, r1), b01 = ROL(X3(a##06, c0, r2), 44), b02 = ROL(X3(a##12, c1, r3), 43),                    \
                  b03 = ROL(X3(a##18, c2, r4), 21), b04 = ROL(X3(a##24, c3, r0), 14);                                             \
    const __m128i b05 = ROL(X3(a##03, c2, r4), 28), b06 = ROL(X3(a##09, c3, r0), 20), b07 = ROL(X3(a##10, c4, r1), 3),            \
                  b08 = ROL(X3(a##16, c0, r2), 45), b09 = ROL(X3(a##22, c1, r3), 61);                                             \
    const __m128i b10 = ROL(X3(a##01, c0, r2), 1), b11 = ROL(X3(a##07, c1, r3), 6), b12 = ROL(X3(a##13, c2, r4), 25),             \
                  b13 = ROL(X3(a##19, c3, r0), 8), b14 = ROL(X3(a##20, c4, r1), 18);                                              \
    const __m128i b15 = ROL(X3(a##04, c3, r0), 27), b16 = ROL(X3(a##05, c4, r1), 36), b17 = ROL(X3(a##11, c0, r2), 10),           \
                  b18 = ROL(X3(a##17, c1, r3), 15), b19 = ROL(X3(a##23, c2, r4), 56);                                             \
    const __m128i b20 = ROL(X3(a##02, c1, r3), 62), b21 = ROL(X3(a##08, c2, r4), 55), b22 = ROL(X3(a##14, c3, r0), 39),           \
                  b23 = ROL(X3(a##15, c4, r1), 41), b24 = ROL(X3(a##21, c0, r2), 2);                                              \
    a##00 = _mm_xor_si128(CHI(b00, b01, b02), _mm_cvtsi64_si128((long long)(rc)));                                                \
    a##01 = CHI(b01, b02, b03), a##02 = CHI(b02, b03, b04), a##03 = CHI(b03, b04, b00), a##04 = CHI(b04, b00, b01);               \
    a##05 = CHI(b05, b06, b07), a##06 = CHI(b06, b07, b08), a##07 = CHI(b07, b08, b09), a##08 = CHI(b08, b09, b05), a##09 = CHI(b09, b05, b06); \
    a##10 = CHI(b10, b11, b12), a##11 = CHI(b11, b12, b13), a##12 = CHI(b12, b13, b14), a##13 = CHI(b13, b14, b10), a##14 = CHI(b14, b10, b11); \
    a##15 = CHI(b15, b16, b17), a##16 = CHI(b16, b17, b18), a##17 = CHI(b17, b18, b19), a##18 = CHI(b18, b19, b15), a##19 = CHI(b19, b15, b16); \
    a##20 = CHI(b20, b21, b22), a##21 = CHI(b21, b22, b23), a##22 = CHI(b22, b23, b24), a##23 = CHI(b23, b24, b20), a##24 = CHI(b24, b20, b21); \
  }


// ---- the same round on general-purpose registers (for the question below: can ONE core run a vector and a scalar permutation side by side?) ----
#define SX3(a, b, c) ((a) ^ (b) ^ (c))
#define SCHI(a, b, c) ((a) ^ (~(b) & (c)))
#define SROL(a, n) (((a) << (n)) | ((a) >> (64 - (n))))
#define SDECL(P) uint64_t P##00, P##01, P##02, P##03, P##04, P##05, P##06, P##07, P##08, P##09, P##10, P##11, P##12, P##13, P##14, P##15, P##16, P##17, P##18, P##19, P##20, P##21, P##22, P##23, P##24
#define SLOADS(P, A) P##00 = A[0], P##01 = A[1], P##02 = A[2], P##03 = A[3], P##04 = A[4], P##05 = A[5], P##06 = A[6], P##07 = A[7], P##08 = A[8], P##09 = A[9], P##10 = A[10], P##11 = A[11], P##12 = A[12], P##13 = A[13], P##14 = A[14], P##15 = A[15], P##16 = A[16], P##17 = A[17], P##18 = A[18], P##19 = A[19], P##20 = A[20], P##21 = A[21], P##22 = A[22], P##23 = A[23], P##24 = A[24]
#define SSTORES(P, A) A[0] = P##00, A[1] = P##01, A[2] = P##02, A[3] = P##03, A[4] = P##04, A[5] = P##05, A[6] = P##06, A[7] = P##07, A[8] = P##08, A[9] = P##09, A[10] = P##10, A[11] = P##11, A[12] = P##12, A[13] = P##13, A[14] = P##14, A[15] = P##15, A[16] = P##16, A[17] = P##17, A[18] = P##18, A[19] = P##19, A[20] = P##20, A[21] = P##21, A[22] = P##22, A[23] = P##23, A[24] = P##24
#define SROUND(a, rc)                                                                                                             \
  {                                                                                                                               \
    const uint64_t c0 = SX3(SX3(a##00, a##05, a##10), a##15, a##20), c1 = SX3(SX3(a##01, a##06, a##11), a##16, a##21),                 \
                  c2 = SX3(SX3(a##02, a##07, a##12), a##17, a##22), c3 = SX3(SX3(a##03, a##08, a##13), a##18, a##23),                 \
                  c4 = SX3(SX3(a##04, a##09, a##14), a##19, a##24);                                                                 \
    const uint64_t r0 = SROL(c0, 1), r1 = SROL(c1, 1), r2 = SROL(c2, 1), r3 = SROL(c3, 1), r4 = SROL(c4, 1);                            \
    const uint64_t b00 = SX3(a##00, c4, r1), b01 = SROL(SX3(a##06, c0, r2), 44), b02 = SROL(SX3(a##12, c1, r3), 43),                    \
                  b03 = SROL(SX3(a##18, c2, r4), 21), b04 = SROL(SX3(a##24, c3, r0), 14);                                             \
    const uint64_t b05 = SROL(SX3(a##03, c2, r4), 28), b06 = SROL(SX3(a##09, c3, r0), 20), b07 = SROL(SX3(a##10, c4, r1), 3),            \
                  b08 = SROL(SX3(a##16, c0, r2), 45), b09 = SROL(SX3(a##22, c1, r3), 61);                                             \
    const uint64_t b10 = SROL(SX3(a##01, c0, r2), 1), b11 = SROL(SX3(a##07, c1, r3), 6), b12 = SROL(SX3(a##13, c2, r4), 25),             \
                  b13 = SROL(SX3(a##19, c3, r0), 8), b14 = SROL(SX3(a##20, c4, r1), 18);                                              \
    const uint64_t b15 = SROL(SX3(a##04, c3, r0), 27), b16 = SROL(SX3(a##05, c4, r1), 36), b17 = SROL(SX3(a##11, c0, r2), 10),           \
                  b18 = SROL(SX3(a##17, c1, r3), 15), b19 = SROL(SX3(a##23, c2, r4), 56);                                             \
    const uint64_t b20 = SROL(SX3(a##02, c1, r3), 62), b21 = SROL(SX3(a##08, c2, r4), 55), b22 = SROL(SX3(a##14, c3, r0), 39),           \
                  b23 = SROL(SX3(a##15, c4, r1), 41), b24 = SROL(SX3(a##21, c0, r2), 2);                                              \
    a##00 = (SCHI(b00, b01, b02) ^ (rc));                                                \
    a##01 = SCHI(b01, b02, b03), a##02 = SCHI(b02, b03, b04), a##03 = SCHI(b03, b04, b00), a##04 = SCHI(b04, b00, b01);               \
    a##05 = SCHI(b05, b06, b07), a##06 = SCHI(b06, b07, b08), a##07 = SCHI(b07, b08, b09), a##08 = SCHI(b08, b09, b05), a##09 = SCHI(b09, b05, b06); \
    a##10 = SCHI(b10, b11, b12), a##11 = SCHI(b11, b12, b13), a##12 = SCHI(b12, b13, b14), a##13 = SCHI(b13, b14, b10), a##14 = SCHI(b14, b10, b11); \
    a##15 = SCHI(b15, b16, b17), a##16 = SCHI(b16, b17, b18), a##17 = SCHI(b17, b18, b19), a##18 = SCHI(b18, b19, b15), a##19 = SCHI(b19, b15, b16); \
    a##20 = SCHI(b20, b21, b22), a##21 = SCHI(b21, b22, b23), a##22 = SCHI(b22, b23, b24), a##23 = SCHI(b23, b24, b20), a##24 = SCHI(b24, b20, b21); \
  }


#define ROUND_SHLD(a, rc)                                                                                                             \
  {                                                                                                                               \
    const __m128i c0 = X3(X3(a##00, a##05, a##10), a##15, a##20), c1 = X3(X3(a##01, a##06, a##11), a##16, a##21),                 \
                  c2 = X3(X3(a##02, a##07, a##12), a##17, a##22), c3 = X3(X3(a##03, a##08, a##13), a##18, a##23),                 \
                  c4 = X3(X3(a##04, a##09, a##14), a##19, a##24);                                                                 \
    const __m128i r0 = ROLF(c0, 1), r1 = ROLF(c1, 1), r2 = ROLF(c2, 1), r3 = ROLF(c3, 1), r4 = ROLF(c4, 1);                            \
    const __m128i b00 = X3(a##00, c4, r1), b01 = ROLF(X3(a##06, c0, r2), 44), b02 = ROLF(X3(a##12, c1, r3), 43),                    \
                  b03 = ROLF(X3(a##18, c2, r4), 21), b04 = ROLF(X3(a##24, c3, r0), 14);                                             \
    const __m128i b05 = ROLF(X3(a##03, c2, r4), 28), b06 = ROLF(X3(a##09, c3, r0), 20), b07 = ROLF(X3(a##10, c4, r1), 3),            \
                  b08 = ROLF(X3(a##16, c0, r2), 45), b09 = ROLF(X3(a##22, c1, r3), 61);                                             \
    const __m128i b10 = ROLF(X3(a##01, c0, r2), 1), b11 = ROLF(X3(a##07, c1, r3), 6), b12 = ROLF(X3(a##13, c2, r4), 25),             \
                  b13 = ROLF(X3(a##19, c3, r0), 8), b14 = ROLF(X3(a##20, c4, r1), 18);                                              \
    const __m128i b15 = ROLF(X3(a##04, c3, r0), 27), b16 = ROLF(X3(a##05, c4, r1), 36), b17 = ROLF(X3(a##11, c0, r2), 10),           \
                  b18 = ROLF(X3(a##17, c1, r3), 15), b19 = ROLF(X3(a##23, c2, r4), 56);                                             \
    const __m128i b20 = ROLF(X3(a##02, c1, r3), 62), b21 = ROLF(X3(a##08, c2, r4), 55), b22 = ROLF(X3(a##14, c3, r0), 39),           \
                  b23 = ROLF(X3(a##15, c4, r1), 41), b24 = ROLF(X3(a##21, c0, r2), 2);                                              \
    a##00 = _mm_xor_si128(CHI(b00, b01, b02), _mm_cvtsi64_si128((long long)(rc)));                                                \
    a##01 = CHI(b01, b02, b03), a##02 = CHI(b02, b03, b04), a##03 = CHI(b03, b04, b00), a##04 = CHI(b04, b00, b01);               \
    a##05 = CHI(b05, b06, b07), a##06 = CHI(b06, b07, b08), a##07 = CHI(b07, b08, b09), a##08 = CHI(b08, b09, b05), a##09 = CHI(b09, b05, b06); \
    a##10 = CHI(b10, b11, b12), a##11 = CHI(b11, b12, b13), a##12 = CHI(b12, b13, b14), a##13 = CHI(b13, b14, b10), a##14 = CHI(b14, b10, b11); \
    a##15 = CHI(b15, b16, b17), a##16 = CHI(b16, b17, b18), a##17 = CHI(b17, b18, b19), a##18 = CHI(b18, b19, b15), a##19 = CHI(b19, b15, b16); \
    a##20 = CHI(b20, b21, b22), a##21 = CHI(b21, b22, b23), a##22 = CHI(b22, b23, b24), a##23 = CHI(b23, b24, b20), a##24 = CHI(b24, b20, b21); \
  }


#define ROUND_MIX(a, rc)                                                                                                             \
  {                                                                                                                               \
    const __m128i c0 = X3(X3(a##00, a##05, a##10), a##15, a##20), c1 = X3(X3(a##01, a##06, a##11), a##16, a##21),                 \
                  c2 = X3(X3(a##02, a##07, a##12), a##17, a##22), c3 = X3(X3(a##03, a##08, a##13), a##18, a##23),                 \
                  c4 = X3(X3(a##04, a##09, a##14), a##19, a##24);                                                                 \
    const __m128i r0 = ROL(c0, 1), r1 = ROL(c1, 1), r2 = ROL(c2, 1), r3 = ROL(c3, 1), r4 = ROL(c4, 1);                            \
    const __m128i b00 = X3(a##00, c4, r1), b01 = ROL(X3(a##06, c0, r2), 44), b02 = ROL(X3(a##12, c1, r3), 43),                    \
                  b03 = ROL(X3(a##18, c2, r4), 21), b04 = ROL(X3(a##24, c3, r0), 14);                                             \
    const __m128i b05 = ROLF(X3(a##03, c2, r4), 28), b06 = ROLF(X3(a##09, c3, r0), 20), b07 = ROLF(X3(a##10, c4, r1), 3),            \
                  b08 = ROLF(X3(a##16, c0, r2), 45), b09 = ROLF(X3(a##22, c1, r3), 61);                                             \
    const __m128i b10 = ROLF(X3(a##01, c0, r2), 1), b11 = ROLF(X3(a##07, c1, r3), 6), b12 = ROLF(X3(a##13, c2, r4), 25),             \
                  b13 = ROLF(X3(a##19, c3, r0), 8), b14 = ROLF(X3(a##20, c4, r1), 18);                                              \
    const __m128i b15 = ROLF(X3(a##04, c3, r0), 27), b16 = ROLF(X3(a##05, c4, r1), 36), b17 = ROLF(X3(a##11, c0, r2), 10),           \
                  b18 = ROLF(X3(a##17, c1, r3), 15), b19 = ROLF(X3(a##23, c2, r4), 56);                                             \
    const __m128i b20 = ROL(X3(a##02, c1, r3), 62), b21 = ROL(X3(a##08, c2, r4), 55), b22 = ROL(X3(a##14, c3, r0), 39),           \
                  b23 = ROL(X3(a##15, c4, r1), 41), b24 = ROL(X3(a##21, c0, r2), 2);                                              \
    a##00 = _mm_xor_si128(CHI(b00, b01, b02), _mm_cvtsi64_si128((long long)(rc)));                                                \
    a##01 = CHI(b01, b02, b03), a##02 = CHI(b02, b03, b04), a##03 = CHI(b03, b04, b00), a##04 = CHI(b04, b00, b01);               \
    a##05 = CHI(b05, b06, b07), a##06 = CHI(b06, b07, b08), a##07 = CHI(b07, b08, b09), a##08 = CHI(b08, b09, b05), a##09 = CHI(b09, b05, b06); \
    a##10 = CHI(b10, b11, b12), a##11 = CHI(b11, b12, b13), a##12 = CHI(b12, b13, b14), a##13 = CHI(b13, b14, b10), a##14 = CHI(b14, b10, b11); \
    a##15 = CHI(b15, b16, b17), a##16 = CHI(b16, b17, b18), a##17 = CHI(b17, b18, b19), a##18 = CHI(b18, b19, b15), a##19 = CHI(b19, b15, b16); \
    a##20 = CHI(b20, b21, b22), a##21 = CHI(b21, b22, b23), a##22 = CHI(b22, b23, b24), a##23 = CHI(b23, b24, b20), a##24 = CHI(b24, b20, b21); \
  }


#define ROUND_XORC(a, rc)                                                                                                             \
  {                                                                                                                               \
    const __m128i c0 = XX(XX(a##00, a##05), XX(XX(a##10, a##15), a##20)), c1 = XX(XX(a##01, a##06), XX(XX(a##11, a##16), a##21)),                 \
                  c2 = XX(XX(a##02, a##07), XX(XX(a##12, a##17), a##22)), c3 = XX(XX(a##03, a##08), XX(XX(a##13, a##18), a##23)),                 \
                  c4 = XX(XX(a##04, a##09), XX(XX(a##14, a##19), a##24));                                                                 \
    const __m128i r0 = ROL(c0, 1), r1 = ROL(c1, 1), r2 = ROL(c2, 1), r3 = ROL(c3, 1), r4 = ROL(c4, 1);                            \
    const __m128i b00 = X3(a##00, c4, r1), b01 = ROL(X3(a##06, c0, r2), 44), b02 = ROL(X3(a##12, c1, r3), 43),                    \
                  b03 = ROL(X3(a##18, c2, r4), 21), b04 = ROL(X3(a##24, c3, r0), 14);                                             \
    const __m128i b05 = ROL(X3(a##03, c2, r4), 28), b06 = ROL(X3(a##09, c3, r0), 20), b07 = ROL(X3(a##10, c4, r1), 3),            \
                  b08 = ROL(X3(a##16, c0, r2), 45), b09 = ROL(X3(a##22, c1, r3), 61);                                             \
    const __m128i b10 = ROL(X3(a##01, c0, r2), 1), b11 = ROL(X3(a##07, c1, r3), 6), b12 = ROL(X3(a##13, c2, r4), 25),             \
                  b13 = ROL(X3(a##19, c3, r0), 8), b14 = ROL(X3(a##20, c4, r1), 18);                                              \
    const __m128i b15 = ROL(X3(a##04, c3, r0), 27), b16 = ROL(X3(a##05, c4, r1), 36), b17 = ROL(X3(a##11, c0, r2), 10),           \
                  b18 = ROL(X3(a##17, c1, r3), 15), b19 = ROL(X3(a##23, c2, r4), 56);                                             \
    const __m128i b20 = ROL(X3(a##02, c1, r3), 62), b21 = ROL(X3(a##08, c2, r4), 55), b22 = ROL(X3(a##14, c3, r0), 39),           \
                  b23 = ROL(X3(a##15, c4, r1), 41), b24 = ROL(X3(a##21, c0, r2), 2);                                              \
    a##00 = _mm_xor_si128(CHI(b00, b01, b02), _mm_cvtsi64_si128((long long)(rc)));                                                \
    a##01 = CHI(b01, b02, b03), a##02 = CHI(b02, b03, b04), a##03 = CHI(b03, b04, b00), a##04 = CHI(b04, b00, b01);               \
    a##05 = CHI(b05, b06, b07), a##06 = CHI(b06, b07, b08), a##07 = CHI(b07, b08, b09), a##08 = CHI(b08, b09, b05), a##09 = CHI(b09, b05, b06); \
    a##10 = CHI(b10, b11, b12), a##11 = CHI(b11, b12, b13), a##12 = CHI(b12, b13, b14), a##13 = CHI(b13, b14, b10), a##14 = CHI(b14, b10, b11); \
    a##15 = CHI(b15, b16, b17), a##16 = CHI(b16, b17, b18), a##17 = CHI(b17, b18, b19), a##18 = CHI(b18, b19, b15), a##19 = CHI(b19, b15, b16); \
    a##20 = CHI(b20, b21, b22), a##21 = CHI(b21, b22, b23), a##22 = CHI(b22, b23, b24), a##23 = CHI(b23, b24, b20), a##24 = CHI(b24, b20, b21); \
  }


#define ROUND_XORC_CHI2(a, rc)                                                                                                             \
  {                                                                                                                               \
    const __m128i c0 = XX(XX(a##00, a##05), XX(XX(a##10, a##15), a##20)), c1 = XX(XX(a##01, a##06), XX(XX(a##11, a##16), a##21)),                 \
                  c2 = XX(XX(a##02, a##07), XX(XX(a##12, a##17), a##22)), c3 = XX(XX(a##03, a##08), XX(XX(a##13, a##18), a##23)),                 \
                  c4 = XX(XX(a##04, a##09), XX(XX(a##14, a##19), a##24));                                                                 \
    const __m128i r0 = ROL(c0, 1), r1 = ROL(c1, 1), r2 = ROL(c2, 1), r3 = ROL(c3, 1), r4 = ROL(c4, 1);                            \
    const __m128i b00 = X3(a##00, c4, r1), b01 = ROL(X3(a##06, c0, r2), 44), b02 = ROL(X3(a##12, c1, r3), 43),                    \
                  b03 = ROL(X3(a##18, c2, r4), 21), b04 = ROL(X3(a##24, c3, r0), 14);                                             \
    const __m128i b05 = ROL(X3(a##03, c2, r4), 28), b06 = ROL(X3(a##09, c3, r0), 20), b07 = ROL(X3(a##10, c4, r1), 3),            \
                  b08 = ROL(X3(a##16, c0, r2), 45), b09 = ROL(X3(a##22, c1, r3), 61);                                             \
    const __m128i b10 = ROL(X3(a##01, c0, r2), 1), b11 = ROL(X3(a##07, c1, r3), 6), b12 = ROL(X3(a##13, c2, r4), 25),             \
                  b13 = ROL(X3(a##19, c3, r0), 8), b14 = ROL(X3(a##20, c4, r1), 18);                                              \
    const __m128i b15 = ROL(X3(a##04, c3, r0), 27), b16 = ROL(X3(a##05, c4, r1), 36), b17 = ROL(X3(a##11, c0, r2), 10),           \
                  b18 = ROL(X3(a##17, c1, r3), 15), b19 = ROL(X3(a##23, c2, r4), 56);                                             \
    const __m128i b20 = ROL(X3(a##02, c1, r3), 62), b21 = ROL(X3(a##08, c2, r4), 55), b22 = ROL(X3(a##14, c3, r0), 39),           \
                  b23 = ROL(X3(a##15, c4, r1), 41), b24 = ROL(X3(a##21, c0, r2), 2);                                              \
    a##00 = _mm_xor_si128(CHI(b00, b01, b02), _mm_cvtsi64_si128((long long)(rc)));                                                \
    a##01 = CHI(b01, b02, b03), a##02 = CHI(b02, b03, b04), a##03 = CHI(b03, b04, b00), a##04 = CHI(b04, b00, b01);               \
    a##05 = CHI(b05, b06, b07), a##06 = CHI(b06, b07, b08), a##07 = CHI(b07, b08, b09), a##08 = CHI(b08, b09, b05), a##09 = CHI(b09, b05, b06); \
    a##10 = CHI(b10, b11, b12), a##11 = CHI(b11, b12, b13), a##12 = CHI(b12, b13, b14), a##13 = CHI(b13, b14, b10), a##14 = CHI(b14, b10, b11); \
    a##15 = XX(b15, _mm_andnot_si128(b16, b17)), a##16 = XX(b16, _mm_andnot_si128(b17, b18)), a##17 = XX(b17, _mm_andnot_si128(b18, b19)), a##18 = XX(b18, _mm_andnot_si128(b19, b15)), a##19 = XX(b19, _mm_andnot_si128(b15, b16)); \
    a##20 = XX(b20, _mm_andnot_si128(b21, b22)), a##21 = XX(b21, _mm_andnot_si128(b22, b23)), a##22 = XX(b22, _mm_andnot_si128(b23, b24)), a##23 = XX(b23, _mm_andnot_si128(b24, b20)), a##24 = XX(b24, _mm_andnot_si128(b20, b21)); \
  }


__attribute__((target("avx512f,avx512vl"), noinline)) void one_loop(uint64_t* A, long n) {
  DECL(a);
  LOADS(a, A);
  for (long k = 0; k < n; k++)
    for (int r = 0; r < 12; r++) ROUND(a, RC[r]);
  STORES(a, A);
}
__attribute__((target("avx512f,avx512vl"), noinline)) void one_unrolled(uint64_t* A, long n) {
  DECL(a);
  LOADS(a, A);
  for (long k = 0; k < n; k++) {
    ROUND(a, RC[0]) ROUND(a, RC[1]) ROUND(a, RC[2]) ROUND(a, RC[3]) ROUND(a, RC[4]) ROUND(a, RC[5])
    ROUND(a, RC[6]) ROUND(a, RC[7]) ROUND(a, RC[8]) ROUND(a, RC[9]) ROUND(a, RC[10]) ROUND(a, RC[11])
  }
  STORES(a, A);
}
__attribute__((target("avx512f,avx512vl,avx512vbmi2"), noinline)) void one_shld(uint64_t* A, long n) {
  DECL(a);
  LOADS(a, A);
  for (long k = 0; k < n; k++)
    for (int r = 0; r < 12; r++) ROUND_SHLD(a, RC[r]);
  STORES(a, A);
}
__attribute__((target("avx512f,avx512vl,avx512vbmi2"), noinline)) void one_mix(uint64_t* A, long n) {
  DECL(a);
  LOADS(a, A);
  for (long k = 0; k < n; k++)
    for (int r = 0; r < 12; r++) ROUND_MIX(a, RC[r]);
  STORES(a, A);
}
__attribute__((target("avx512f,avx512vl"), noinline)) void one_xorc(uint64_t* A, long n) {
  DECL(a);
  LOADS(a, A);
  for (long k = 0; k < n; k++)
    for (int r = 0; r < 12; r++) ROUND_XORC(a, RC[r]);
  STORES(a, A);
}
__attribute__((target("avx512f,avx512vl"), noinline)) void one_xorc_chi2(uint64_t* A, long n) {
  DECL(a);
  LOADS(a, A);
  for (long k = 0; k < n; k++)
    for (int r = 0; r < 12; r++) ROUND_XORC_CHI2(a, RC[r]);
  STORES(a, A);
}
__attribute__((target("avx512f,avx512vl"), noinline)) void two_interleaved(uint64_t* A, uint64_t* B, long n) {
  DECL(a);
  DECL(z);
  LOADS(a, A);
  LOADS(z, B);
  for (long k = 0; k < n; k++)
    for (int r = 0; r < 12; r++) {
      ROUND(a, RC[r]);
      ROUND(z, RC[r]);
    }
  STORES(a, A);
  STORES(z, B);
}


__attribute__((target("bmi,bmi2"), noinline)) void one_scalar(uint64_t* A, long n) {
  SDECL(a);
  SLOADS(a, A);
  for (long k = 0; k < n; k++)
    for (int r = 0; r < 12; r++) SROUND(a, RC[r]);
  SSTORES(a, A);
}
// ONE vector state and ONE scalar state, independent, round by round in the same loop: if this takes about as long as the slower of the two
// alone, the core runs both pipes side by side and a HYBRID single state (some lanes in xmm registers, some in general-purpose ones) has
// something to gain; if it takes their sum, it has not
__attribute__((target("avx512f,avx512vl,bmi,bmi2"), noinline)) void vector_and_scalar(uint64_t* A, uint64_t* B, long n) {
  DECL(a);
  SDECL(z);
  LOADS(a, A);
  SLOADS(z, B);
  for (long k = 0; k < n; k++)
    for (int r = 0; r < 12; r++) {
      ROUND(a, RC[r]);
      SROUND(z, RC[r]);
    }
  STORES(a, A);
  SSTORES(z, B);
}

int main() {
  uint64_t s1[25], s2[25], s3[25];
  for (int i = 0; i < 25; i++) s1[i] = s2[i] = s3[i] = 0x0123456789abcdefULL * (i + 1);
  one_loop(s1, 1000);
  one_unrolled(s2, 1000);
  printf("unrolled matches loop: %d\n", memcmp(s1, s2, 200) == 0);
  const long n = 20000000;
  auto time = [&](const char* name, auto f, int states) {
    auto t0 = std::chrono::steady_clock::now();
    f();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%-24s %.1f ns per permutation (%.2f GB/s per 168-byte block stream)\n", name, dt / n / states * 1e9, 168.0 * n * states / dt / 1e9);
  };
  time("one state, round loop", [&] { one_loop(s1, n); }, 1);
  time("one state, unrolled", [&] { one_unrolled(s2, n); }, 1);
  time("two states interleaved", [&] { two_interleaved(s1, s3, n); }, 2);
  uint64_t s4[25], s5[25], s6[25];
  for (int i = 0; i < 25; i++) s4[i] = s5[i] = s6[i] = 0x0123456789abcdefULL * (i + 1);
  one_loop(s4, 1000), one_xorc(s5, 1000), one_xorc_chi2(s6, 1000);
  printf("xor-parity variants match: %d %d\n", memcmp(s4, s5, 200) == 0, memcmp(s4, s6, 200) == 0);
  uint64_t s7[25], s8[25];
  for (int i = 0; i < 25; i++) s7[i] = s8[i] = 0x0123456789abcdefULL * (i + 1);
  one_shld(s7, 1000), one_mix(s8, 1000);
  printf("funnel-shift variants match: %d %d\n", memcmp(s4, s7, 200) == 0, memcmp(s4, s8, 200) == 0);
  time("rotates by vpshldq", [&] { one_shld(s7, n); }, 1);
  time("rotates half vpshldq", [&] { one_mix(s8, n); }, 1);
  time("parity by xor chains", [&] { one_xorc(s5, n); }, 1);
  time("+ chi of 2 rows andn/xor", [&] { one_xorc_chi2(s6, n); }, 1);
  uint64_t s9[25], s10[25], s11[25];
  for (int i = 0; i < 25; i++) s9[i] = s10[i] = s11[i] = 0x0123456789abcdefULL * (i + 1);
  one_scalar(s9, 1000);
  printf("scalar round matches: %d\n", memcmp(s4, s9, 200) == 0);
  time("one state, scalar", [&] { one_scalar(s9, n); }, 1);
  time("vector + scalar states", [&] { vector_and_scalar(s10, s11, n); }, 2);
  printf("%llx %llx %llx\n", (unsigned long long)s1[0], (unsigned long long)s2[0], (unsigned long long)s3[0]);
}
