// Focused repro for a hint that differs between cpir_server_setup and cpir_server_setup_multi (seen under the ThreadSanitizer build's timing):
// both set up the same small database with empty rows again and again, and every hint is compared with a plain CPU product of
// cpir_generate_from_seed's A and D.   Built like the TSan driver (against lib/tsan) or against the release library.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "chalamet_hip.h"

#define CK(x)                                                                                          \
  do {                                                                                                 \
    int s_ = (x);                                                                                      \
    if (s_ != 0) {                                                                                     \
      fprintf(stderr, "%s failed: %s %s\n", #x, cpir_strerror(s_), cpir_last_hip_error());             \
      exit(1);                                                                                         \
    }                                                                                                  \
  } while (0)

static uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const int holes = argc > 2 ? atoi(argv[2]) : 1;
  cpir_device* dev = nullptr;
  CK(cpir_device_open(0, &dev));
  const uint64_t N = 3 * 1536 * 2 + 512;
  const uint32_t C = 20, b = 9, R = CPIR_LWE_DIMENSION;
  std::vector<uint32_t> D((size_t)N * C), A((size_t)R * N), want((size_t)R * C), h1(want.size()), h2(want.size());
  for (size_t i = 0; i < D.size(); i++) D[i] = (uint32_t)mix(i + 100) & ((1u << b) - 1);
  if (holes)
    for (uint64_t n = 0; n < N; n++)
      if (mix(n + 17) % 5 == 0) memset(&D[(size_t)n * C], 0, (size_t)C * 4);
  uint8_t seed[CPIR_SEED_BYTE_LEN];
  for (int i = 0; i < CPIR_SEED_BYTE_LEN; i++) seed[i] = (uint8_t)(i * 7 + 1);
  CK(cpir_generate_from_seed(R, N, seed, A.data()));
  for (uint32_t r = 0; r < R; r++)
    for (uint32_t c = 0; c < C; c++) {
      uint32_t s = 0;
      for (uint64_t n = 0; n < N; n++) s += A[(size_t)r * N + n] * D[(size_t)n * C + c];
      want[(size_t)r * C + c] = s;
    }
  // per-shard partial products, to name what a wrong group hint is made of (the shard bounds are read from the first group built)
  std::vector<std::vector<uint32_t>> part;
  std::vector<uint64_t> lo_of, n_of;
  int bad1 = 0, bad2 = 0;
  for (int it = 0; it < reps; it++) {
    cpir_server *one = nullptr, *grp = nullptr;
    memset(h1.data(), 0xAB, h1.size() * 4), memset(h2.data(), 0xAB, h2.size() * 4);
    CK(cpir_server_setup(dev, seed, nullptr, D.data(), N, C, b, h1.data(), &one));
    cpir_device* devs[3] = {dev, dev, dev};
    CK(cpir_server_setup_multi(devs, 3, seed, nullptr, D.data(), N, C, b, h2.data(), &grp));
    if (part.empty()) {
      uint32_t G = 0;
      CK(cpir_server_group_size(grp, &G));
      for (uint32_t g = 0; g < G; g++) {
        int ord = 0;
        uint64_t lo = 0, n = 0;
        CK(cpir_server_group_shard(grp, g, &ord, &lo, &n));
        lo_of.push_back(lo), n_of.push_back(n);
        part.emplace_back((size_t)R * C);
        for (uint32_t r = 0; r < R; r++)
          for (uint32_t c = 0; c < C; c++) {
            uint32_t sum = 0;
            for (uint64_t k = lo; k < lo + n; k++) sum += A[(size_t)r * N + k] * D[(size_t)k * C + c];
            part.back()[(size_t)r * C + c] = sum;
          }
        fprintf(stderr, "shard %u: slots [%llu, %llu)\n", g, (unsigned long long)lo, (unsigned long long)(lo + n));
      }
    }
    for (int which = 0; which < 2; which++) {
      const std::vector<uint32_t>& h = which ? h2 : h1;
      size_t wrong = 0, first = 0;
      uint32_t rmin = R, rmax = 0;
      for (size_t i = 0; i < h.size(); i++)
        if (h[i] != want[i]) {
          if (!wrong) first = i;
          wrong++;
          const uint32_t r = (uint32_t)(i / C);
          rmin = r < rmin ? r : rmin, rmax = r > rmax ? r : rmax;
        }
      if (wrong) {
        (which ? bad2 : bad1)++;
        // what is it?  want + sum over shards of coefficient * partial, coefficients in {-1, 0, +1}: -1 = the shard's part missing, +1 = counted twice
        for (int c0 = -1; c0 <= 1 && which; c0++)
          for (int c1 = -1; c1 <= 1; c1++)
            for (int c2 = -1; c2 <= 1; c2++) {
              if (part.size() != 3 || (!c0 && !c1 && !c2)) continue;
              bool all = true;
              for (size_t i = 0; i < h.size() && all; i++)
                all = h[i] == want[i] + (uint32_t)c0 * part[0][i] + (uint32_t)c1 * part[1][i] + (uint32_t)c2 * part[2][i];
              if (all) fprintf(stderr, "   = the right hint %+d x shard 0's part %+d x shard 1's %+d x shard 2's\n", c0, c1, c2);
            }
        fprintf(stderr, "iteration %d: %s hint WRONG in %zu of %zu entries, rows %u..%u (first: row %zu col %zu got %08x want %08x)\n", it,
                which ? "cpir_server_setup_multi" : "cpir_server_setup", wrong, h.size(), rmin, rmax, first / C, first % C, h[first], want[first]);
      }
    }
    cpir_server_release(one);
    cpir_server_release(grp);
  }
  printf("%d iterations: single-server hint wrong %d times, group hint wrong %d times\n", reps, bad1, bad2);
  cpir_device_close(dev);
  return bad1 + bad2 ? 1 : 0;
}
