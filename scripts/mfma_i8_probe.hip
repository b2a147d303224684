// Probe of v_mfma_i32_16x16x64_i8 on the MI355X, the instruction the planar respond kernel is built on.  Run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -w scripts/mfma_i8_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Checks, with exact integer data against a host triple loop:
//   1. C/D map: lane l holds column l&15, rows 4*(l>>4) + i in register i;
//   2. A and B use the SAME (lane group, byte) -> k map (so a kernel that puts slot s in byte j of lane group g of both
//      operands never needs to know the map), and which k that is under the natural hypothesis k = 16*(l>>4) + j;
//   3. the i32 accumulator wraps (two's complement) instead of saturating;
//   4. issue cost of the instruction (one wave per SIMD, dependent and independent chains).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void one_mfma(const int8_t* A /*16 x 64*/, const int8_t* B /*64 x 16*/, const int* Cin, int* Cout, int reps) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  v4i a, b, c;
  int8_t* ab = reinterpret_cast<int8_t*>(&a);
  int8_t* bb = reinterpret_cast<int8_t*>(&b);
  for (int j = 0; j < 16; j++) {
    ab[j] = A[r * 64 + 16 * g + j];
    bb[j] = B[(16 * g + j) * 16 + r];
  }
  for (int i = 0; i < 4; i++) c[i] = Cin[(4 * g + i) * 16 + r];
  for (int k = 0; k < reps; k++) c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; i++) Cout[(4 * g + i) * 16 + r] = c[i];
}

template <int CHAINS>
__global__ void __launch_bounds__(256) rate(int* out, int iters) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  v4i c[CHAINS];
  for (int i = 0; i < CHAINS; i++) c[i] = v4i{0, 0, 0, 0};
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < CHAINS; i++) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
  int s = 0;
  for (int i = 0; i < CHAINS; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  std::vector<int8_t> A(16 * 64), B(64 * 16);
  std::vector<int> Cin(256), Cout(256), want(256);
  srand(1);
  for (auto& x : A) x = (int8_t)(rand() % 256 - 128);
  for (auto& x : B) x = (int8_t)(rand() % 256 - 128);
  for (auto& x : Cin) x = rand() % 1000 - 500;
  int8_t *dA, *dB;
  int *dCi, *dCo;
  hipMalloc(&dA, A.size());
  hipMalloc(&dB, B.size());
  hipMalloc(&dCi, 1024);
  hipMalloc(&dCo, 1024);
  auto run = [&](int reps) {
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dCi, Cin.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dCi, dCo, reps);
    hipMemcpy(Cout.data(), dCo, 1024, hipMemcpyDeviceToHost);
  };
  run(1);
  int bad = 0;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      long s = Cin[i * 16 + j];
      for (int k = 0; k < 64; k++) s += (long)A[i * 64 + k] * B[k * 16 + j];
      want[i * 16 + j] = (int)s;
      bad += (want[i * 16 + j] != Cout[i * 16 + j]);
    }
  printf("1+2. natural maps (A[row l&15][k=16(l>>4)+j], B[k][col l&15], C col l&15 row 4(l>>4)+i): %s (%d wrong of 256)\n",
         bad ? "MISMATCH" : "exact", bad);
  // wrap: A = B = -128 everywhere: every product 16384, 64 per MFMA => +2^20 per rep; 4096 reps => 2^32 = wraps to Cin
  for (auto& x : A) x = -128;
  for (auto& x : B) x = -128;
  run(2048);
  long expect_nowrap = (long)Cin[0] + 2048L * (1 << 20);
  printf("3. after 2048 reps of +2^20: C[0][0] = %d; two's complement wrap would give %d, saturation %d\n", Cout[0],
         (int)(uint32_t)(expect_nowrap & 0xffffffff), 2147483647);
  run(4096);
  printf("   after 4096 reps (+2^32): C[0][0] = %d, Cin was %d => %s\n", Cout[0], Cin[0], Cout[0] == Cin[0] ? "WRAPS" : "does not wrap");

  int* out;
  hipMalloc(&out, 256 * 4 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  auto time_it = [&](auto kern, int chains, const char* name) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e-3 / ((double)iters * chains);
    printf("4. %s: %.2f ns per MFMA per wave (one wave per SIMD) = %.1f cycles at 2.4 GHz; chip-wide %.1f TMAC/s\n", name, per * 1e9,
           per * 2.4e9, 16.0 * 16 * 64 * 1024 / per / 1e12);
  };
  time_it(rate<1>, 1, "dependent chain   ");
  time_it(rate<4>, 4, "4 independent accs");
  return 0;
}
