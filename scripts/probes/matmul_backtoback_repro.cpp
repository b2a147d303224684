// Isolates the hint matmul of the group setup path: cpir_op_mat_x_mat three times back to back on ONE stream, on the three column slabs of A
// and row blocks of D a group of three would use (inner 3072, 3072, 3584; 20 columns; rows with nothing in them), each product compared with a
// plain CPU product.   usage: matmul_backtoback_repro [iterations]
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "chalamet_hip.h"

#define CK(x) do { int s_ = (x); if (s_ != 0) { fprintf(stderr, "%s failed: %s %s\n", #x, cpir_strerror(s_), cpir_last_hip_error()); exit(1); } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 100;
  cpir_device* dev = nullptr;
  CK(cpir_device_open(0, &dev));
  const uint32_t C = 20, b = 9, R = CPIR_LWE_DIMENSION;
  const uint64_t lo[3] = {0, 3072, 6144}, n[3] = {3072, 3072, 3584}, N = 9728;
  std::vector<uint32_t> D((size_t)N * C), A((size_t)R * N);
  for (size_t i = 0; i < D.size(); i++) D[i] = (uint32_t)mix(i + 100) & ((1u << b) - 1);
  for (uint64_t k = 0; k < N; k++)
    if (mix(k + 17) % 5 == 0) memset(&D[(size_t)k * C], 0, (size_t)C * 4);
  for (size_t i = 0; i < A.size(); i++) A[i] = (uint32_t)mix(i * 7 + 3);
  std::vector<uint32_t> want[3], slab[3], got((size_t)R * C);
  uint32_t *A_dev[3], *D_dev[3], *M_dev[3];
  hipStream_t st;
  HK(hipStreamCreate(&st));
  for (int g = 0; g < 3; g++) {
    want[g].assign((size_t)R * C, 0);
    slab[g].resize((size_t)R * n[g]);
    for (uint32_t r = 0; r < R; r++) {
      memcpy(&slab[g][(size_t)r * n[g]], &A[(size_t)r * N + lo[g]], n[g] * 4);
      for (uint32_t c = 0; c < C; c++) {
        uint32_t s = 0;
        for (uint64_t k = 0; k < n[g]; k++) s += A[(size_t)r * N + lo[g] + k] * D[(size_t)(lo[g] + k) * C + c];
        want[g][(size_t)r * C + c] = s;
      }
    }
    HK(hipMalloc((void**)&A_dev[g], slab[g].size() * 4));
    HK(hipMalloc((void**)&D_dev[g], n[g] * C * 4));
    HK(hipMalloc((void**)&M_dev[g], (size_t)R * C * 4));
    HK(hipMemcpy(A_dev[g], slab[g].data(), slab[g].size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(D_dev[g], &D[(size_t)lo[g] * C], n[g] * C * 4, hipMemcpyHostToDevice));
  }
  int bad = 0;
  for (int it = 0; it < reps; it++) {
    for (int g = 0; g < 3; g++) HK(hipMemsetAsync(M_dev[g], 0xCD, (size_t)R * C * 4, st));
    for (int g = 0; g < 3; g++) CK(cpir_op_mat_x_mat(dev, A_dev[g], n[g], D_dev[g], C, M_dev[g], C, R, n[g], C, 16, 0, st));
    HK(hipStreamSynchronize(st));
    for (int g = 0; g < 3; g++) {
      HK(hipMemcpy(got.data(), M_dev[g], got.size() * 4, hipMemcpyDeviceToHost));
      size_t wrong = 0, first = 0;
      for (size_t i = 0; i < got.size(); i++)
        if (got[i] != want[g][i]) {
          if (!wrong) first = i;
          wrong++;
        }
      if (wrong) {
        bad++;
        fprintf(stderr, "iteration %d: product %d (inner %llu) WRONG in %zu of %zu entries (first: row %zu col %zu got %08x want %08x)\n", it, g,
                (unsigned long long)n[g], wrong, got.size(), first / C, first % C, got[first], want[g][first]);
      }
    }
  }
  printf("%d iterations x 3 products: %d wrong\n", reps, bad);
  cpir_device_close(dev);
  return bad ? 1 : 0;
}
