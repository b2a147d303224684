// atomic_flush_probe.hip -- what does the end of a respond pass cost when G blocks all add their K partial sums into the SAME K words?
// The step-major respond kernel (respond_planar.hip) keeps a pass's responses in LDS and leaves through one u32 atomicAdd per (block,
// query, padded column): 512 blocks x 7 680 words for a fused pass of 8 queries at 2^20 keys x 1 kB, all at about the same moment.
// Variants: same words from every block / same words, every block starting at a rotated offset / words of its own per block (no
// contention: the price of the traffic alone) / plain stores into a per-block slab + a second kernel that sums the slabs.
//   hipcc --offload-arch=gfx950 -O3 atomic_flush_probe.hip -o atomic_flush_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) flush_same(unsigned* r, unsigned K, int rotate) {
  const unsigned start = rotate ? (unsigned)(((unsigned long long)blockIdx.x * K) / gridDim.x) / 64 * 64 : 0;
  for (unsigned i = threadIdx.x; i < K; i += 256) {
    unsigned j = i + start;
    if (j >= K) j -= K;
    atomicAdd(r + j, blockIdx.x + i);
  }
}
__global__ void __launch_bounds__(256) flush_own(unsigned* r, unsigned K) {
  for (unsigned i = threadIdx.x; i < K; i += 256) atomicAdd(r + (size_t)blockIdx.x * K + i, blockIdx.x + i);
}
__global__ void __launch_bounds__(256) store_own(unsigned* slab, unsigned K) {
  for (unsigned i = threadIdx.x; i < K; i += 256) slab[(size_t)blockIdx.x * K + i] = blockIdx.x + i;
}
__global__ void __launch_bounds__(256) sum_slabs(const unsigned* slab, unsigned K, unsigned G, unsigned* r) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K) return;
  unsigned s = 0;
  for (unsigned g = 0; g < G; g++) s += slab[(size_t)g * K + i];
  r[i] += s;
}
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("FAILED %s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

int main() {
  for (unsigned G : {256u, 512u}) {
    for (unsigned K : {960u, 7680u}) {
      unsigned *r, *slab;
      CK(hipMalloc(&r, (size_t)G * K * 4));
      CK(hipMalloc(&slab, (size_t)G * K * 4));
      CK(hipMemset(r, 0, (size_t)G * K * 4));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      const int reps = 20;
      for (int v = 0; v < 4; v++) {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
          CK(hipEventRecord(e0, 0));
          for (int k = 0; k < reps; k++) {
            if (v == 0) hipLaunchKernelGGL(flush_same, dim3(G), dim3(256), 0, 0, r, K, 0);
            if (v == 1) hipLaunchKernelGGL(flush_same, dim3(G), dim3(256), 0, 0, r, K, 1);
            if (v == 2) hipLaunchKernelGGL(flush_own, dim3(G), dim3(256), 0, 0, r, K);
            if (v == 3) {
              hipLaunchKernelGGL(store_own, dim3(G), dim3(256), 0, 0, slab, K);
              hipLaunchKernelGGL(sum_slabs, dim3((K + 255) / 256), dim3(256), 0, 0, slab, K, G, r);
            }
          }
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        const char* names[] = {"same words", "same words, rotated start", "own words (no contention)", "plain stores + sum kernel"};
        printf("G = %3u blocks, K = %4u words: %-28s %7.2f us per flush\n", G, K, names[v], best * 1e3f / reps);
      }
      CK(hipFree(r));
      CK(hipFree(slab));
    }
  }
  return 0;
}
