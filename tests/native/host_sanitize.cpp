// AddressSanitizer / UBSan run of the product's HOST code (XOF, shape arithmetic, KV encoder) -- GPU sanitizers are not
// available on the pool, so the CPU-side C++ is what gets sanitised.  Built and run by tests/test_host_sanitizers.py.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../chalametpir_amd/csrc/cpir_internal.hpp"

using namespace cpir;

static int fail(const char* what) {
  fprintf(stderr, "FAIL: %s\n", what);
  return 1;
}

int main() {
  // ---- XOF: RFC 9861 known answer, and chunked squeezing equals one-shot squeezing --------------------------------------
  {
    uint8_t out[64];
    turboshake128(nullptr, 0, out, 32);
    static const uint8_t want[4] = {0x1e, 0x41, 0x5f, 0x1c};
    if (memcmp(out, want, 4) != 0) return fail("TurboSHAKE128 KAT");
    std::vector<uint8_t> a(5000), b(5000);
    TurboShake128 x, y;
    uint8_t seed[32] = {7};
    x.absorb(seed, 32), x.finalize();
    y.absorb(seed, 16), y.absorb(seed + 16, 16), y.finalize();
    x.squeeze(a.data(), a.size());
    for (size_t off = 0, step = 1; off < b.size(); off += step, step = step * 3 + 1) {
      if (off + step > b.size()) step = b.size() - off;
      y.squeeze(b.data() + off, step);
    }
    if (a != b) return fail("chunked squeeze");
  }
  // ---- shapes -----------------------------------------------------------------------------------------------------------
  for (uint64_t n : {1ull, 2ull, 3ull, 1000ull, 1ull << 20, 1ull << 22, 1ull << 40}) {
    uint32_t b = 0, sl, scl;
    uint64_t nf;
    if (find_bit_len(n, &b) != CPIR_OK) return fail("find_bit_len");
    for (uint32_t arity : {3u, 4u})
      if (n < (1ull << 32) && filter_shape(arity, n, &sl, &scl, &nf) != CPIR_OK) return fail("filter_shape");
    cpir_dtc_layout L;
    if (dtc_layout_for(n < (1ull << 30) ? n : (1ull << 30), 940, b, &L) != CPIR_OK || check_layout(L) != CPIR_OK) return fail("layout");
  }
  // ---- encoder: both arities, tiny to mid-size, ragged key/value lengths, forced retries ---------------------------------
  std::mt19937_64 rng(42);
  for (uint32_t arity : {3u, 4u}) {
    for (uint64_t n : {1ull, 2ull, 5ull, 300ull, 5000ull}) {
      std::vector<uint8_t> keys, vals, seeds(32 * 100);
      std::vector<uint64_t> koff{0}, voff{0};
      for (uint64_t i = 0; i < n; i++) {
        const size_t kl = 1 + rng() % 32, vl = 1 + rng() % 97;
        for (size_t j = 0; j < kl; j++) keys.push_back((uint8_t)rng());
        keys[koff.back()] = (uint8_t)i, keys.push_back((uint8_t)(i >> 8)), keys.push_back((uint8_t)(i >> 16));  // distinct
        for (size_t j = 0; j < vl; j++) vals.push_back((uint8_t)rng());
        koff.push_back(keys.size()), voff.push_back(vals.size());
      }
      for (auto& s : seeds) s = (uint8_t)rng();
      cpir_kv_db db{n, keys.data(), koff.data(), vals.data(), voff.data()};
      for (uint32_t b : {4u, 9u, 14u}) {
        Filter f;
        std::vector<uint32_t> D;
        uint64_t N = 0;
        uint32_t C = 0;
        const int st = encode_kv_database(arity, db, b, seeds.data(), 100, &f, &D, &N, &C);
        if (st != CPIR_OK) return fail("encode_kv_database");
        if (D.size() != N * C || f.filter_size != n) return fail("encoder shape");
        for (uint32_t v : D)
          if (v >> b) return fail("entry wider than b bits");
        uint8_t fb[CPIR_FILTER_PARAM_BYTE_LEN];
        f.to_bytes(fb);
      }
    }
  }
  {  // error paths
    cpir_kv_db empty{0, nullptr, nullptr, nullptr, nullptr};
    Filter f;
    std::vector<uint32_t> D;
    uint64_t N;
    uint32_t C;
    if (encode_kv_database(3, empty, 9, nullptr, 100, &f, &D, &N, &C) != CPIR_ERR_EMPTY_KV_DATABASE) return fail("empty db");
    if (encode_kv_database(5, empty, 9, nullptr, 100, &f, &D, &N, &C) != CPIR_ERR_UNSUPPORTED_ARITY) return fail("arity");
  }
  puts("host sanitizer run ok");
  return 0;
}
