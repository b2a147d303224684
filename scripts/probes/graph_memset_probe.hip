// graph_memset_probe.hip -- round 6: does a captured hipMemsetAsync of a few bytes run on EVERY launch of the instantiated graph?
// Found through tests/test_gpu_stress.py (a lone respond captured into a graph: memset of the 132-byte response + the kernel that adds into
// it): the first launch of the graph was right, the later ones were off by the buffer's previous contents.  Nothing of the library here:
// capture { hipMemsetAsync(r, 0, bytes); add_one<<<>>>(r, words) } on a stream, instantiate, then per launch: fill r with 0xFFFFFFFF from
// the host, launch the graph, read r back -- every word must be 1.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/graph_memset_probe.hip -o /tmp/graph_memset_probe && /tmp/graph_memset_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void add_one(uint32_t* r, uint32_t words) {
  for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) atomicAdd(r + i, 1u);
}
__global__ void zero_words(uint32_t* r, uint32_t words) {
  for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) r[i] = 0;
}

// second question: does a kernel node see what the HOST has copied into its input between two launches of the graph?  r[i] = q[i] summed
// over `rows` rows of q, q read with 16-byte cached loads by a persistent grid of 256 blocks, the way the respond kernel reads its queries
__global__ void __launch_bounds__(512) sum_rows(uint32_t* r, const uint32_t* q, uint32_t words, uint32_t rows) {
  for (uint32_t i = (blockIdx.x * 512 + threadIdx.x) * 4; i + 3 < words; i += gridDim.x * 512 * 4) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < rows; k++) {
      const uint4 v = *reinterpret_cast<const uint4*>(q + (size_t)k * words + i);
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    atomicAdd(r + i, acc.x), atomicAdd(r + i + 1, acc.y), atomicAdd(r + i + 2, acc.z), atomicAdd(r + i + 3, acc.w);
  }
}

// third question: the same with a kernel shaped like the respond kernel's launch: one 200-byte struct passed by value, dynamic LDS raised
// with hipFuncSetAttribute, 512 threads, an eager launch first -- TWO graphs instantiated one after the other, each launched three times
struct Big {
  const uint32_t* q;
  uint32_t* r;
  uint64_t pad[21];
  uint32_t words, rows;
};
__global__ void __launch_bounds__(512) big_kernel(const Big a) {
  extern __shared__ uint32_t lds[];
  lds[threadIdx.x] = (uint32_t)a.pad[threadIdx.x % 21];
  __syncthreads();
  for (uint32_t i = blockIdx.x * 512 + threadIdx.x; i < a.words; i += gridDim.x * 512) {
    uint32_t v = lds[(threadIdx.x + 1) % 512] - (uint32_t)a.pad[(threadIdx.x + 1) % 512 % 21];
    for (uint32_t k = 0; k < a.rows; k++) v += a.q[(size_t)k * a.words + i];
    atomicAdd(a.r + i, v);
  }
}
static void big_probe(hipStream_t s) {
  const uint32_t words = 4685, rows = 1;
  uint32_t *r, *q;
  CK(hipMalloc(&r, words * 4));
  CK(hipMalloc(&q, (size_t)rows * words * 4));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 << 10);
  Big a{};
  a.q = q, a.r = r, a.words = words, a.rows = rows;
  for (int i = 0; i < 21; i++) a.pad[i] = 77 + i;
  hipLaunchKernelGGL(big_kernel, dim3(256), dim3(512), 40 << 10, s, a);  // eager once
  CK(hipStreamSynchronize(s));
  std::vector<uint32_t> hq((size_t)rows * words), hr(words);
  for (int graph = 0; graph < 2; graph++) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    CK(hipMemsetAsync(r, 0, words * 4, s));
    hipLaunchKernelGGL(big_kernel, dim3(256), dim3(512), 40 << 10, s, a);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    int wrong = 0;
    for (int launch = 0; launch < 3; launch++) {
      for (size_t i = 0; i < hq.size(); i++) hq[i] = (uint32_t)(i * 2654435761u) + 1000003u * (uint32_t)(launch + 3 * graph);
      CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(hr.data(), r, words * 4, hipMemcpyDeviceToHost));
      uint32_t bad = 0;
      for (uint32_t i = 0; i < words; i++) bad += hr[i] != hq[i];
      wrong += bad != 0;
    }
    printf("struct-by-value kernel with dynamic LDS, graph %d of the process: %d of 3 launches wrong\n", graph + 1, wrong);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  CK(hipFree(r));
  CK(hipFree(q));
}

static void input_probe(hipStream_t s) {
  for (uint32_t words : {4096u, 1u << 16, 1u << 20}) {
    const uint32_t rows = 4;
    uint32_t *r, *q;
    CK(hipMalloc(&r, words * 4));
    CK(hipMalloc(&q, (size_t)rows * words * 4));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    CK(hipMemsetAsync(r, 0, words * 4, s));
    hipLaunchKernelGGL(sum_rows, dim3(256), dim3(512), 0, s, r, q, words, rows);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<uint32_t> hq((size_t)rows * words), hr(words);
    int wrong_launches = 0, first_wrong = -1;
    for (int launch = 0; launch < 6; launch++) {
      for (size_t i = 0; i < hq.size(); i++) hq[i] = (uint32_t)(i * 2654435761u) + 1000003u * (uint32_t)launch;
      CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(hr.data(), r, words * 4, hipMemcpyDeviceToHost));
      uint32_t bad = 0;
      for (uint32_t i = 0; i < words; i++) {
        uint32_t want = 0;
        for (uint32_t k = 0; k < rows; k++) want += hq[(size_t)k * words + i];
        bad += hr[i] != want;
      }
      if (bad) {
        wrong_launches++;
        if (first_wrong < 0) first_wrong = launch;
      }
      // an eager launch of the same kernel between two launches of the graph (as a caller that mixes both would do)
      hipLaunchKernelGGL(sum_rows, dim3(256), dim3(512), 0, s, r, q, words, rows);
      CK(hipStreamSynchronize(s));
    }
    printf("input rewritten by the host between launches, %8u words x %u rows: %d of 6 launches wrong%s\n", words, rows, wrong_launches,
           wrong_launches ? (first_wrong == 0 ? " (from the first)" : " (the first launch was right)") : "");
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipFree(r));
    CK(hipFree(q));
  }
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  big_probe(s);
  input_probe(s);
  for (int zero_by_kernel = 0; zero_by_kernel < 2; zero_by_kernel++)
    for (uint32_t words : {33u, 165u, 940u, 4096u, 30080u}) {
      uint32_t* r;
      CK(hipMalloc(&r, words * 4));
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
      if (zero_by_kernel) hipLaunchKernelGGL(zero_words, dim3(1), dim3(256), 0, s, r, words);
      else CK(hipMemsetAsync(r, 0, words * 4, s));
      hipLaunchKernelGGL(add_one, dim3(1), dim3(256), 0, s, r, words);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      std::vector<uint32_t> h(words);
      int wrong_launches = 0, first_wrong = -1;
      for (int launch = 0; launch < 6; launch++) {
        for (uint32_t& x : h) x = 0xFFFFFFFFu;
        CK(hipMemcpy(r, h.data(), words * 4, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), r, words * 4, hipMemcpyDeviceToHost));
        uint32_t bad = 0;
        for (uint32_t x : h) bad += x != 1u;
        if (bad) {
          wrong_launches++;
          if (first_wrong < 0) first_wrong = launch;
        }
      }
      printf("%-22s %6u words (%7u bytes): %d of 6 launches wrong%s\n", zero_by_kernel ? "zeroed by a kernel" : "hipMemsetAsync", words, words * 4, wrong_launches,
             wrong_launches ? (first_wrong == 0 ? " (from the first)" : " (the first launch was right)") : "");
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
      CK(hipFree(r));
    }
  return 0;
}
