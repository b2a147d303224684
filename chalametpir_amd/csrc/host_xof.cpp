// host_xof.cpp -- TurboSHAKE128 (RFC 9861) on the host: the XOF behind Matrix::generate_from_seed (reference
// chalametpir_common/src/matrix.rs:541-558), hash_of_key (binary_fuse_filter.rs:568-584) and encode_kv_as_row
// (serialization.rs:23-32).  The reference gets it from the `turboshake` crate (=0.4.1, not vendored);
// this is an independent implementation of the published algorithm: Keccak-p[1600, 12 rounds] sponge, rate 168 bytes,
// domain separation byte (0x1F in every reference call site) followed by pad10*1.
//
// Expanding the public matrix A is one sponge squeezed for 4*1774*N bytes (7.8 GiB at 2^20 keys): sequential by
// construction, so it stays on a host core and is overlapped with H2D + the device matmul (capi.hip).
// Measured on the GPU box's EPYC 9575F: this scalar permutation squeezes 1.45-1.7 GB/s.  An AVX-512 single-state variant
// (five planes in zmm registers, vprolvq / vpermq / vpternlogq, a 5x5 qword transpose per round for pi) was written,
// verified and measured at 1.31 GB/s -- its round is a chain of ~5 dependent cross-lane permutes -- so it was dropped.
#include <cstddef>
#include <cstdint>
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
void keccak_p1600_12(uint64_t* A) {
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

}  // namespace

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
