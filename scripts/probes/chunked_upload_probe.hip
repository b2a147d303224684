// chunked_upload_probe.hip -- round 6, the lone host caller: how early can a kernel that is launched TOGETHER with a chunked
// copy-engine upload of its query start consuming it?
//
// The experiment asked for: instead of the step-major kernel reading a page-locked query in place over the link (204 us against the wide
// kernel's 187 on a resident query), upload the query with the copy engine in chunks on a stream of its own, publish a progress word per
// chunk in DEVICE memory in stream order, and let the wide kernel -- launched at the same moment on another stream -- poll it.  What
// that buys is bounded from below by   (first chunk visible to the waiting kernel) + 187 us + hand-over,   so the first term is measured
// here, on the bare runtime, with nothing of the library in the way:
//   * a 4.7 MB page-locked "query" (N = 1 179 648 words), CHUNKS 2-D copies (one row per XCD slice: chunk j of all 8 slices in ONE call,
//     which is how the wide kernel's eight slices would be fed round-robin) on an upload stream, each followed by a 4-byte progress
//     write in stream order (hipStreamWriteValue32, and, as a second variant, a one-thread kernel);
//   * a one-wave kernel launched FIRST on another stream records the device wall clock when it starts and when it sees each progress
//     value; the host records when its calls return and when the kernel's last timestamp lands.
// Prints, per variant and chunk count: host time spent in the enqueue calls, kernel start -> chunk 1 / chunk k / last chunk visible,
// first call -> everything visible as the host sees it.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/chunked_upload_probe.hip -o /tmp/chunked_upload_probe && /tmp/chunked_upload_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// one wave: t[0] = wall clock at start, t[i] = wall clock when progress >= i (i = 1 .. chunks), then a flag in host memory
__global__ void watcher(const uint32_t* progress, uint32_t chunks, uint64_t* t, uint32_t* done_host, uint32_t seq, uint64_t give_up_ticks) {
  if (threadIdx.x != 0) return;
  const uint64_t t0 = wall_clock64();
  t[0] = t0;
  uint32_t seen = 0;
  while (seen < chunks) {
    const uint32_t p = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t now = wall_clock64();
    for (; seen < p && seen < chunks; seen++) t[seen + 1] = now;
    if (now - t0 > give_up_ticks) break;  // (every wave reaches the end: the grid always drains)
  }
  __threadfence_system();
  __hip_atomic_store(done_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void publish(uint32_t* progress, uint32_t v) { __hip_atomic_store(progress, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

int main(int argc, char** argv) {
  const size_t N = 1179648, slices = 8, slice_words = N / slices;
  uint32_t *q_host, *q_dev, *progress, *done_host;
  uint64_t* t_dev;
  CK(hipHostMalloc(&q_host, N * 4, hipHostMallocDefault));
  CK(hipHostMalloc(&done_host, 64, hipHostMallocCoherent));
  CK(hipMalloc(&q_dev, N * 4));
  CK(hipMalloc(&progress, 64));
  CK(hipMalloc(&t_dev, 72 * 8));
  for (size_t i = 0; i < N; i++) q_host[i] = (uint32_t)i * 2654435761u;
  hipStream_t up, run;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&run, hipStreamNonBlocking, hi));
  uint32_t seq = 0;
  std::vector<uint64_t> t(72);
  for (int variant = 0; variant < 2; variant++)
    for (uint32_t chunks : {1u, 2u, 4u, 8u, 16u, 32u}) {
      const size_t cw = slice_words / chunks;  // words of one slice per chunk
      std::vector<double> first, kth, last, host_total, enq;
      for (int rep = 0; rep < 30; rep++) {
        CK(hipMemsetAsync(progress, 0, 4, up));
        CK(hipStreamSynchronize(up));
        CK(hipStreamSynchronize(run));
        __atomic_store_n(done_host, 0u, __ATOMIC_RELAXED);
        ++seq;
        const double h0 = now_us();
        hipLaunchKernelGGL(watcher, dim3(1), dim3(64), 0, run, progress, chunks, t_dev, done_host, seq, (uint64_t)2000000);  // 20 ms
        for (uint32_t j = 0; j < chunks; j++) {
          // chunk j of every slice: 8 rows of cw words, row pitch = one slice
          CK(hipMemcpy2DAsync(q_dev + j * cw, slice_words * 4, q_host + j * cw, slice_words * 4, cw * 4, slices, hipMemcpyHostToDevice, up));
          if (variant == 0) CK(hipStreamWriteValue32(up, progress, j + 1, 0));
          else hipLaunchKernelGGL(publish, dim3(1), dim3(1), 0, up, progress, j + 1);
        }
        const double h1 = now_us();
        while (__atomic_load_n(done_host, __ATOMIC_ACQUIRE) != seq) {
        }
        const double h2 = now_us();
        CK(hipMemcpy(t.data(), t_dev, (chunks + 1) * 8, hipMemcpyDeviceToHost));
        if (rep < 5) continue;  // warm-up
        first.push_back((t[1] - t[0]) / 100.0);  // wall clock: 100 MHz
        kth.push_back((t[(chunks + 1) / 2] - t[0]) / 100.0);
        last.push_back((t[chunks] - t[0]) / 100.0);
        host_total.push_back(h2 - h0);
        enq.push_back(h1 - h0);
      }
      auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      printf("%-22s %2u chunk(s) of %7zu B: enqueue calls %6.1f us | kernel start -> chunk 1 visible %6.1f us, half %6.1f us, all %6.1f us | first call -> all visible (host) %6.1f us\n",
             variant == 0 ? "hipStreamWriteValue32" : "one-thread kernel", chunks, cw * 4 * slices, med(enq), med(first), med(kth), med(last), med(host_total));
    }
  // the same bytes as ONE plain copy, for scale
  {
    std::vector<double> one;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 20; rep++) {
      const double h0 = now_us();
      CK(hipMemcpyAsync(q_dev, q_host, N * 4, hipMemcpyHostToDevice, up));
      CK(hipStreamSynchronize(up));
      if (rep >= 5) one.push_back(now_us() - h0);
    }
    std::sort(one.begin(), one.end());
    printf("one hipMemcpyAsync of %zu B + hipStreamSynchronize: %.1f us (%.1f GB/s)\n", N * 4, one[one.size() / 2], N * 4 / one[one.size() / 2] / 1e3);
  }
  return 0;
}
