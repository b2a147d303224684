// host_gather.cpp -- dst[i] = src[idx[i]] on the host: how a lone host query is compacted onto the slots a server really holds
// (compact.hip) WHILE it is copied into the page-locked block the kernel reads in place (host_respond.hip, respond_alone).  The index
// list is increasing and skips about one slot in nine (the rows of a binary-fuse-encoded database that no key owns), so the source is
// read almost sequentially; what decides the speed is how the indices are turned into loads.  Three variants, picked once per process:
//   avx512: 16 indices per vpgatherdd, one 64-byte store;   avx2: 8 per vpgatherdd;   scalar: one load per word.
// The kernel consumes a compact query at ~25 GB/s (4.2 MB under a 170 us stream), four threads copy side by side: each has to sustain
// ~6 GB/s, i.e. 1.5 words per nanosecond -- which the scalar loop (about one word per nanosecond) does not reach and the gathers do.
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

void gather_scalar(uint32_t* dst, const uint32_t* src, const uint32_t* idx, size_t count) {
  for (size_t i = 0; i < count; i++) dst[i] = src[idx[i]];
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) void gather_avx2(uint32_t* dst, const uint32_t* src, const uint32_t* idx, size_t count) {
  size_t i = 0;
  for (; i + 8 <= count; i += 8) {
    const __m256i k = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(idx + i));
    // (indices are below 2^32 - 16 and are used as UNSIGNED offsets: the gather sign-extends 32-bit indices, so anything at or above
    // 2^31 words goes through the scalar tail below -- a 8 GiB query, far beyond any shard)
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_i32gather_epi32(reinterpret_cast<const int*>(src), k, 4));
  }
  for (; i < count; i++) dst[i] = src[idx[i]];
}

__attribute__((target("avx512f"))) void gather_avx512(uint32_t* dst, const uint32_t* src, const uint32_t* idx, size_t count) {
  size_t i = 0;
  for (; i + 16 <= count; i += 16) {
    const __m512i k = _mm512_loadu_si512(idx + i);
    _mm512_storeu_si512(dst + i, _mm512_i32gather_epi32(k, src, 4));
  }
  for (; i < count; i++) dst[i] = src[idx[i]];
}
#endif

using GatherFn = void (*)(uint32_t*, const uint32_t*, const uint32_t*, size_t);

struct Picked {
  GatherFn fn;
  const char* name;
};

Picked pick() {
  const char* force = getenv("CPIR_GATHER");  // "scalar" / "avx2" / "avx512": tests compare the variants; anything else = best available
#if defined(__x86_64__)
  __builtin_cpu_init();
  const bool has512 = __builtin_cpu_supports("avx512f"), has2 = __builtin_cpu_supports("avx2");
  if (force && !strcmp(force, "scalar")) return {gather_scalar, "scalar"};
  if (force && !strcmp(force, "avx2") && has2) return {gather_avx2, "avx2"};
  if (has512) return {gather_avx512, "avx512"};
  if (has2) return {gather_avx2, "avx2"};
#else
  (void)force;
#endif
  return {gather_scalar, "scalar"};
}

const Picked& picked() {
  static const Picked p = pick();
  return p;
}

}  // namespace

void gather_words(uint32_t* dst, const uint32_t* src, const uint32_t* idx, size_t count) {
  // the vector gathers sign-extend their 32-bit indices: keep them for index lists that stay below 2^31 (the list is increasing: look at its end)
  if (count && idx[count - 1] >= 0x80000000u) return gather_scalar(dst, src, idx, count);
  picked().fn(dst, src, idx, count);
}

const char* gather_words_variant() { return picked().name; }

}  // namespace cpir
