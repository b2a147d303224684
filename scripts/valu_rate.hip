// Micro-benchmark: issue rate of the integer VALU instructions the kernels are built on.  Run on the MI355X box:
//   hipcc --offload-arch=gfx950 -O3 -w scripts/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Every wave runs 8 independent dependency chains of one instruction; 256 CUs x 8 blocks x 4 waves resident.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int WHICH>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t* out, int iters) {
  uint32_t x = threadIdx.x * 2654435761u + 12345u, y = blockIdx.x * 40503u + 77u;
  uint32_t a[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  uint64_t w[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        if (WHICH == 0) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y));
        if (WHICH == 1) asm volatile("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(a[c]) : "v"(x), "v"(y));
        if (WHICH == 2) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y));
        if (WHICH == 3) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y));
        if (WHICH == 4) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[c]) : "v"(x));
        if (WHICH == 5) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[c]) : "v"(x), "v"(y) : "vcc");
        if (WHICH == 6) asm volatile("v_bfe_u32 %0, %0, 9, 9" : "+v"(a[c]));
        if (WHICH == 7) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[c]) : "v"(x));
        if (WHICH == 8) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(x), "v"(y));
        if (WHICH == 9) asm volatile("v_alignbit_b32 %0, %0, %1, 27" : "+v"(a[c]) : "v"(x));
        if (WHICH == 10) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(x), "v"(y));
        if (WHICH == 11) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y));
        if (WHICH == 12) asm volatile("v_lshrrev_b32 %0, 9, %0" : "+v"(a[c]));
        if (WHICH == 13) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(x), "v"(y));
      }
    }
  }
  uint32_t r = 0;
  for (int c = 0; c < 8; c++) r ^= a[c] ^ (uint32_t)w[c] ^ (uint32_t)(w[c] >> 32);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int WHICH>
double run(uint32_t* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<WHICH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<WHICH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, blocks = cus * 8, iters = 1000;
  uint32_t* out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  const double ops = (double)blocks * 256 * iters * 16 * 8;  // lane-instructions
  const char* names[] = {"v_mad_u32_u24", "v_mad_u32_u16", "v_dot2_u32_u16", "v_dot4_u32_u8", "v_mul_lo_u32", "v_mad_u64_u32", "v_bfe_u32",
                         "v_add_u32", "v_and_or_b32", "v_alignbit_b32", "v_perm_b32", "v_fma_f32", "v_lshrrev_b32", "v_add3_u32"};
  double ms[14];
  ms[0] = run<0>(out, blocks, iters); ms[1] = run<1>(out, blocks, iters); ms[2] = run<2>(out, blocks, iters);
  ms[3] = run<3>(out, blocks, iters); ms[4] = run<4>(out, blocks, iters); ms[5] = run<5>(out, blocks, iters);
  ms[6] = run<6>(out, blocks, iters); ms[7] = run<7>(out, blocks, iters); ms[8] = run<8>(out, blocks, iters);
  ms[9] = run<9>(out, blocks, iters); ms[10] = run<10>(out, blocks, iters); ms[11] = run<11>(out, blocks, iters);
  ms[12] = run<12>(out, blocks, iters); ms[13] = run<13>(out, blocks, iters);
  printf("%d CUs, nominal clock %d MHz\n", cus, p.clockRate / 1000);
  for (int i = 0; i < 14; i++)
    printf("%-16s %8.3f ms  %7.2f T lane-ops/s  %6.1f lanes/clk/CU @2.4GHz\n", names[i], ms[i], ops / ms[i] / 1e9, ops / (ms[i] * 1e-3) / 2.4e9 / cus);
  return 0;
}
