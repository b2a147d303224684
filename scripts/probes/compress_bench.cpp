// Probe: what compacting a query onto the kept slots costs on the host (host_gather.cpp compress_words) next to the memcpy it replaces, for T
// threads side by side -- the staging step of concurrent callers of Server::respond on a server with a slot map.
//   hipcc -O2 -pthread -x c++ -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ scripts/probes/compress_bench.cpp -o /tmp/compress_bench -Lchalametpir_amd/lib -lchalamet_hip -Wl,-rpath,$PWD/chalametpir_amd/lib
// Destination: heap memory, and page-locked COHERENT host memory (what an arena's staging block is); sources cycled so that they are cold.
//   CPIR_GATHER=avx512-masked /tmp/compress_bench      (round 4's masked stores)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
namespace cpir {
size_t compress_words(uint32_t* dst, const uint32_t* src, const uint8_t* bits, size_t s_lo, size_t s_hi);
const char* gather_words_variant();
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t N = 1179648;
  std::vector<uint8_t> bits(N / 8 + 16, 0);
  uint64_t x = 88172645463325252ull;
  for (size_t n = 0; n < N; n++) {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    if (x % 9 != 0) bits[n >> 3] |= (uint8_t)(1u << (n & 7));
  }
  printf("variant %s\n", cpir::gather_words_variant());
  for (int T : {1, 4, 8, 16}) {
    const int S = 8;  // sources per thread, cycled: 8 x 4.7 MB x T threads do not stay in the caches
    std::vector<std::vector<uint32_t>> src(T * S, std::vector<uint32_t>(N, 7)), heap(T, std::vector<uint32_t>(N + 64));
    std::vector<uint32_t*> pinned(T, nullptr), dst(T, nullptr);
    for (int t = 0; t < T; t++)
      if (hipHostMalloc(reinterpret_cast<void**>(&pinned[t]), (N + 64) * 4, hipHostMallocCoherent) != hipSuccess) return 1;
    for (int where = 0; where < 2; where++)
    for (int mode = 0; mode < 2; mode++) {
      for (int t = 0; t < T; t++) dst[t] = where ? pinned[t] : heap[t].data();
      std::vector<double> us(T);
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
          const int reps = 40;
          const double t0 = now();
          size_t sink = 0;
          for (int r = 0; r < reps; r++) {
            const uint32_t* s = src[S * t + (r % S)].data();
            if (mode == 0) sink += cpir::compress_words(dst[t], s, bits.data(), 0, N);
            else memcpy(dst[t], s, N * 4), sink += dst[t][r];
          }
          us[t] = (now() - t0) / reps * 1e6 + (sink == 1 ? 1e-9 : 0);
        });
      for (auto& t : th) t.join();
      double sum = 0;
      for (double u : us) sum += u;
      printf("%2d threads  %s into %s: %.0f us per query\n", T, mode == 0 ? "compress" : "memcpy  ", where ? "page-locked coherent memory" : "heap memory", sum / T);
    }
    for (int t = 0; t < T; t++) (void)hipHostFree(pinned[t]);
  }
  return 0;
}
