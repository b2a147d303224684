// Does work on memory from hipMallocAsync stay in stream order when blocks are recycled at once on the same stream?  Per iteration three
// rounds (sizes as the three hint products of a small group): hipMallocAsync, a kernel that fills the block with a value, a kernel that checks
// every word against that value, hipFreeAsync.  A mismatch means the check ran before / without its fill.
//   hipcc --offload-arch=gfx950 scripts/probes/mallocasync_order_probe.hip -o /tmp/mao && /tmp/mao [iterations]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill(unsigned* p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void check(const unsigned* p, size_t n, unsigned v, unsigned* bad) {
  unsigned mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) mine += p[i] != v;
  if (mine) atomicAdd(bad, mine);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 2000;
  hipStream_t s;
  HK(hipStreamCreate(&s));
  unsigned* bad;
  HK(hipMalloc((void**)&bad, 12));
  const size_t words[3] = {30000, 30000, 35000};  // (~ the right-hand-side workspaces of three products with 3072 / 3072 / 3584 x 20 entries)
  unsigned long long total_bad[3] = {0, 0, 0};
  int bad_iters = 0;
  for (int it = 0; it < reps; it++) {
    HK(hipMemsetAsync(bad, 0, 12, s));
    for (int g = 0; g < 3; g++) {
      unsigned* ws = nullptr;
      HK(hipMallocAsync((void**)&ws, words[g] * 4, s));
      const unsigned v = (unsigned)(it * 3 + g + 1);
      HK(hipMemsetAsync(ws, 0, 256, s));  // (a small memset in front, as the library has)
      fill<<<64, 256, 0, s>>>(ws, words[g], v);
      check<<<256, 256, 0, s>>>(ws, words[g], v, bad + g);
      HK(hipFreeAsync(ws, s));
    }
    unsigned h[3];
    HK(hipMemcpyAsync(h, bad, 12, hipMemcpyDeviceToHost, s));
    HK(hipStreamSynchronize(s));
    if (h[0] | h[1] | h[2]) {
      bad_iters++;
      if (bad_iters <= 5) fprintf(stderr, "iteration %d: words that did not hold their round's value: %u / %u / %u\n", it, h[0], h[1], h[2]);
    }
    for (int g = 0; g < 3; g++) total_bad[g] += h[g];
  }
  printf("%d iterations: %d with a mismatch (words: %llu / %llu / %llu)\n", reps, bad_iters, total_bad[0], total_bad[1], total_bad[2]);
  return bad_iters ? 1 : 0;
}
