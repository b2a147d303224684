// Probe: how fast can 256-thread blocks read an N x C u32 row-major matrix (C = 940: rows of 3760 bytes, not a multiple of anything nice)
// when a wave instruction covers  SEG  rows x (1024 / SEG) contiguous bytes of each -- the read side of pack.hip's streaming kernel
// (SEG = 4: four rows 16 apart x 256 B) against the same bytes fetched as 2 x 512 B and 1 x 1024 B.  Same structure otherwise: 16 loads
// of 16 bytes per lane in flight, then a short dependent reduction, 3 waves per SIMD worth of registers; every byte of D read once.
//   hipcc --offload-arch=gfx950 -O3 -w scripts/probes/row_segment_read_probe.hip -o /tmp/rsp && /tmp/rsp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int SEG>  // rows per wave instruction
__global__ void __launch_bounds__(256) read_kernel(const uint32_t* __restrict__ D, uint64_t N, uint32_t C, uint32_t stripes, uint32_t* out) {
  // a wave = (1024 / SEG) bytes = COLS columns x 64*SEG/4... keep the pack kernel's unit: one wave covers COLS = 256 / SEG * ... columns
  constexpr uint32_t LPR = 64 / SEG;         // lanes per row segment
  constexpr uint32_t COLS = LPR * 4;         // columns per wave (64 for SEG 4, 128 for SEG 2, 256 for SEG 1)
  constexpr uint32_t ROWS_PER_PASS = 16 * SEG;  // 16 loads per lane, SEG rows per instruction
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t g = lane / LPR, lc = lane % LPR;
  const uint32_t stripe = (blockIdx.x % stripes) * 4 + wave;
  const uint64_t step = blockIdx.x / stripes;   // 512 rows per step
  const uint32_t c0 = stripe * COLS + 4 * lc;
  const uint32_t csafe = c0 + 3 < C ? c0 : 0;
  uint32_t acc = 0;
  for (uint32_t pass = 0; pass < 512 / ROWS_PER_PASS; pass++) {
    uint4 v[16];
    const uint64_t r0 = step * 512 + (uint64_t)pass * ROWS_PER_PASS + 16 * g;   // SEG row groups 16 apart, 16 consecutive rows each
#pragma unroll
    for (int j = 0; j < 16; j++) {
      uint64_t r = r0 + j;
      if (r >= N) r = N - 1;
      v[j] = *reinterpret_cast<const uint4*>(D + r * C + csafe);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) acc += (v[j].x ^ v[j].y) + (v[j].z ^ v[j].w);
    asm volatile("" : "+v"(acc));
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int SEG>
double run(const uint32_t* D, uint64_t N, uint32_t C, uint32_t* out) {
  constexpr uint32_t COLS = (64 / SEG) * 4;
  const uint32_t stripes = ((C + COLS - 1) / COLS + 3) / 4;
  const uint32_t steps = (uint32_t)((N + 511) / 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(read_kernel<SEG>, dim3(steps * stripes), dim3(256), 0, 0, D, N, C, stripes, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(read_kernel<SEG>, dim3(steps * stripes), dim3(256), 0, 0, D, N, C, stripes, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const uint64_t N = 1179648;
  const uint32_t C = 940;
  uint32_t *D, *out;
  hipMalloc(&D, N * C * 4 + 4096);
  hipMalloc(&out, 64);
  hipMemset(D, 1, N * C * 4);
  const double gb = N * C * 4 / 1e9;
  const double t4 = run<4>(D, N, C, out), t2 = run<2>(D, N, C, out), t1 = run<1>(D, N, C, out);
  printf("rows x bytes per wave instruction:  4 x 256 B: %.3f ms = %.0f GB/s;  2 x 512 B: %.3f ms = %.0f GB/s;  1 x 1024 B: %.3f ms = %.0f GB/s\n", t4,
         gb / t4 * 1e3, t2, gb / t2 * 1e3, t1, gb / t1 * 1e3);
  return 0;
}
