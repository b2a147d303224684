// host_gather.cpp -- compacting a query on the host: how a lone host query is brought onto the slots a server really holds (compact.hip)
// WHILE it is copied into the page-locked block the kernel reads in place (host_respond.hip, respond_alone).  The kept slots are
// increasing and skip about one slot in nine (the rows of a binary-fuse-encoded database that no key owns), so the source is read almost
// sequentially; what decides the speed is how that is put to the CPU.  The kernel consumes a compact query at ~25 GB/s (4.2 MB under a
// 170 us stream) and four threads copy side by side: each has to sustain ~6 GB/s of output.
//   compress_words   the kept slots as a BITMAP over the source: 16 source words per step -- one sequential 64-byte load, vpcompressd in
//                    registers (compressing straight to memory is microcoded on Zen 4), one masked store of popcount words.  A few cycles
//                    per 16 source words, and the loads are plain sequential ones the prefetchers follow.  AVX-512 hosts (every GPU box).
//   gather_words     dst[i] = src[idx[i]] from an index list: 16 / 8 indices per vpgatherdd (AVX-512 / AVX2) or one load per word;
//                    measured ~1.3 words per nanosecond and thread on the GPU box's EPYC 9575F with the source hot in cache, i.e. below what the
//                    kernel wants: a lone query took 231 us page-locked / 315 us from a cold pageable buffer against 214 / 247 without a
//                    slot map -- which is why the bitmap form exists.  Kept as the fallback for hosts without AVX-512.
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

// The SOURCE may be only byte-aligned: cpir_server_respond_bytes hands over `query + 8` of a wire buffer (matrix.rs:1001-1007 reads it with
// a byte copy too).  Every scalar read of it goes through this; the vector paths use unaligned loads / gathers by construction.
inline uint32_t load_word(const uint32_t* src, size_t i) {
  uint32_t v;
  memcpy(&v, reinterpret_cast<const unsigned char*>(src) + 4 * i, 4);
  return v;
}

void gather_scalar(uint32_t* dst, const uint32_t* src, const uint32_t* idx, size_t count) {
  for (size_t i = 0; i < count; i++) dst[i] = load_word(src, idx[i]);
}

size_t compress_scalar(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo, size_t s_hi) {
  uint32_t* d = dst;
  for (size_t s = s_lo; s < s_hi; s++)
    if ((bits[s >> 3] >> (s & 7)) & 1) *d++ = load_word(src, s);
  return (size_t)(d - dst);
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) void gather_avx2(uint32_t* dst, const uint32_t* src, const uint32_t* idx, size_t count) {
  size_t i = 0;
  for (; i + 8 <= count; i += 8) {
    const __m256i k = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(idx + i));
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_i32gather_epi32(reinterpret_cast<const int*>(src), k, 4));
  }
  for (; i < count; i++) dst[i] = load_word(src, idx[i]);
}

__attribute__((target("avx512f"))) void gather_avx512(uint32_t* dst, const uint32_t* src, const uint32_t* idx, size_t count) {
  size_t i = 0;
  for (; i + 16 <= count; i += 16) {
    const __m512i k = _mm512_loadu_si512(idx + i);
    _mm512_storeu_si512(dst + i, _mm512_i32gather_epi32(k, src, 4));
  }
  for (; i < count; i++) dst[i] = load_word(src, idx[i]);
}

// set bits of the bitmap in [s_lo, s_hi) (the bitmap is padded by 8 bytes)
__attribute__((target("popcnt"))) size_t count_bits(const uint8_t* bits, size_t s_lo, size_t s_hi) {
  size_t n = 0, s = s_lo;
  for (; s < s_hi && (s & 7); s++) n += (bits[s >> 3] >> (s & 7)) & 1;
  for (; s + 64 <= s_hi; s += 64) {
    uint64_t w;
    memcpy(&w, bits + (s >> 3), 8);
    n += (size_t)_mm_popcnt_u64(w);
  }
  for (; s < s_hi; s++) n += (bits[s >> 3] >> (s & 7)) & 1;
  return n;
}

// MASKED: every store writes exactly its k words (round 4's form; kept for A/B timing, CPIR_GATHER=avx512-masked).  Otherwise the job's
// output count is taken from the bitmap first and every store but the last few is a FULL 64-byte one -- the words behind the k valid ones
// are overwritten by this job's own next outputs; only where fewer than 16 words of the job's range remain is the store masked, so nothing
// is ever written outside [dst, dst + count) (the words behind belong to another thread's job).  Masked stores of a varying number of words
// cost this CPU generation several times a plain store: with 8 callers compacting side by side a query took 535 us against 204 us for the
// memcpy it replaces (round 5, EPYC 9575F).
template <bool MASKED>
__attribute__((target("avx512f,avx512bw,popcnt"))) size_t compress_avx512_impl(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo,
                                                                                 size_t s_hi) {
  uint32_t* d = dst;
  uint32_t* const d_full_end = MASKED ? dst : dst + count_bits(bits, s_lo, s_hi) - 0;  // full stores may start at d with d + 16 <= d_full_end
  size_t s = s_lo;
  for (; s + 16 <= s_hi; s += 16) {
    uint32_t w;
    memcpy(&w, bits + (s >> 3), 4);  // bits [s, s + 16) start at bit s % 8 of these 4 bytes (the bitmap is padded by 8 bytes)
    const __mmask16 m = (__mmask16)(w >> (s & 7));
    const __m512i v = _mm512_maskz_compress_epi32(m, _mm512_loadu_si512(src + s));
    const unsigned k = (unsigned)_mm_popcnt_u32(m);
    if (!MASKED && d + 16 <= d_full_end) _mm512_storeu_si512(d, v);
    else _mm512_mask_storeu_epi32(d, (__mmask16)((1u << k) - 1u), v);  // exactly k words
    d += k;
  }
  if (s < s_hi) {  // fewer than 16 source words left: nothing at or beyond s_hi is read
    const unsigned r = (unsigned)(s_hi - s);
    uint32_t w;
    memcpy(&w, bits + (s >> 3), 4);
    const __mmask16 in = (__mmask16)((1u << r) - 1u);
    const __mmask16 m = (__mmask16)((w >> (s & 7)) & in);
    const __m512i v = _mm512_maskz_compress_epi32(m, _mm512_maskz_loadu_epi32(in, src + s));
    const unsigned k = (unsigned)_mm_popcnt_u32(m);
    _mm512_mask_storeu_epi32(d, (__mmask16)((1u << k) - 1u), v);
    d += k;
  }
  return (size_t)(d - dst);
}
__attribute__((target("avx512f,avx512bw,popcnt"))) size_t compress_avx512(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo, size_t s_hi) {
  return compress_avx512_impl<false>(dst, src, bits, s_lo, s_hi);
}

// The same into a destination that is COLD in this core's caches and will be read by the device next (an arena's page-locked staging block,
// written by one caller after the other and read by the DMA engine in between): plain stores to such lines first fetch them for ownership --
// in situ a compaction took 370 (8 callers) to 550 us (16) where the memcpy it replaces, which streams, took 170 to 230.  Here the
// compressed words are collected in a 16 KiB buffer that stays in L1 and leave it as aligned non-temporal 64-byte stores.  dst 64-byte
// aligned; writes exactly the returned number of words.
__attribute__((target("avx512f,avx512bw,popcnt"))) size_t compress_stream_avx512(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo,
                                                                                   size_t s_hi) {
  constexpr size_t kChunk = 4096;  // words
  alignas(64) uint32_t buf[kChunk + 32];
  size_t fill = 0;
  uint32_t* d = dst;
  // (a macro, not a lambda: a lambda would not inherit this function's target attribute)
#define CPIR_FLUSH_CHUNK()                                                                                                      \
  do {                                                                                                                          \
    for (size_t i_ = 0; i_ < kChunk; i_ += 16) _mm512_stream_si512(reinterpret_cast<__m512i*>(d + i_), _mm512_load_si512(buf + i_)); \
    d += kChunk;                                                                                                                \
    _mm512_store_si512(buf, _mm512_loadu_si512(buf + kChunk)); /* at most 15 words beyond the chunk */                          \
    fill -= kChunk;                                                                                                             \
  } while (0)
  size_t s = s_lo;
  for (; s + 16 <= s_hi; s += 16) {
    uint32_t w;
    memcpy(&w, bits + (s >> 3), 4);
    const __mmask16 m = (__mmask16)(w >> (s & 7));
    _mm512_storeu_si512(buf + fill, _mm512_maskz_compress_epi32(m, _mm512_loadu_si512(src + s)));  // (a full store: the buffer has the slack)
    fill += (unsigned)_mm_popcnt_u32(m);
    if (fill >= kChunk) CPIR_FLUSH_CHUNK();
  }
  if (s < s_hi) {
    const unsigned r = (unsigned)(s_hi - s);
    uint32_t w;
    memcpy(&w, bits + (s >> 3), 4);
    const __mmask16 in = (__mmask16)((1u << r) - 1u);
    const __mmask16 m = (__mmask16)((w >> (s & 7)) & in);
    _mm512_storeu_si512(buf + fill, _mm512_maskz_compress_epi32(m, _mm512_maskz_loadu_epi32(in, src + s)));
    fill += (unsigned)_mm_popcnt_u32(m);
    if (fill >= kChunk) CPIR_FLUSH_CHUNK();
  }
  size_t i = 0;
  for (; i + 16 <= fill; i += 16) _mm512_stream_si512(reinterpret_cast<__m512i*>(d + i), _mm512_load_si512(buf + i));
  if (i < fill) _mm512_mask_storeu_epi32(d + i, (__mmask16)((1u << (fill - i)) - 1u), _mm512_load_si512(buf + i));
  d += fill;
  _mm_sfence();
#undef CPIR_FLUSH_CHUNK
  return (size_t)(d - dst);
}
__attribute__((target("avx512f,avx512bw,popcnt"))) size_t compress_avx512_masked(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo,
                                                                                   size_t s_hi) {
  return compress_avx512_impl<true>(dst, src, bits, s_lo, s_hi);
}
#endif

using GatherFn = void (*)(uint32_t*, const uint32_t*, const uint32_t*, size_t);
using CompressFn = size_t (*)(uint32_t*, const uint32_t*, const uint8_t*, size_t, size_t);

struct Picked {
  GatherFn gather;
  const char* gather_name;
  CompressFn compress;  // the vector one, or NULL (the caller then gathers from the index list)
};

Picked pick() {
  const char* force = getenv("CPIR_GATHER");  // "scalar" / "avx2" / "avx512-gather" / "avx512-masked": forces a lesser variant (tests, A/B timing)
#if defined(__x86_64__)
  __builtin_cpu_init();
  const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("popcnt");
  const bool has2 = __builtin_cpu_supports("avx2");
  if (force && !strcmp(force, "scalar")) return {gather_scalar, "scalar", nullptr};
  if (force && !strcmp(force, "avx2") && has2) return {gather_avx2, "avx2", nullptr};
  if (force && !strcmp(force, "avx512-gather") && has512) return {gather_avx512, "avx512", nullptr};
  if (force && !strcmp(force, "avx512-masked") && has512) return {gather_avx512, "avx512", compress_avx512_masked};
  if (has512) return {gather_avx512, "avx512", compress_avx512};
  if (has2) return {gather_avx2, "avx2", nullptr};
#else
  (void)force;
#endif
  return {gather_scalar, "scalar", nullptr};
}

const Picked& picked() {
  static const Picked p = pick();
  return p;
}

}  // namespace

void gather_words(uint32_t* dst, const uint32_t* src, const uint32_t* idx, size_t count) {
  // the vector gathers sign-extend their 32-bit indices: keep them for index lists that stay below 2^31 (the list is increasing: look at its end)
  if (count && idx[count - 1] >= 0x80000000u) return gather_scalar(dst, src, idx, count);
  picked().gather(dst, src, idx, count);
}

const char* gather_words_variant() { return picked().compress ? "avx512-compress" : picked().gather_name; }

bool compress_words_vectorised() { return picked().compress != nullptr; }

size_t compress_words(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo, size_t s_hi) {
  return picked().compress ? picked().compress(dst, src, bits, s_lo, s_hi) : compress_scalar(dst, src, bits, s_lo, s_hi);
}

size_t compress_words_streaming(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo, size_t s_hi) {
#if defined(__x86_64__)
  if (picked().compress == compress_avx512 && reinterpret_cast<uintptr_t>(dst) % 64 == 0) return compress_stream_avx512(dst, src, bits, s_lo, s_hi);
#endif
  return compress_words(dst, src, bits, s_lo, s_hi);
}

}  // namespace cpir
