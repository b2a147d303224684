// synth.hip -- counter-based synthetic data for benchmarks and spot-checkable tests.
// Not part of the reference: the reference benches build a random KV database with an OS-seeded ChaCha8
// (integrations/benches/online_phase.rs:14-31), which is neither reproducible nor feasible at 30+ GB on a host.
// out[i] = hi32(splitmix64-finaliser(seed, index0 + i)) & mask; the same function exists on the host in the
// test oracle so any tile can be regenerated and checked without materialising the whole matrix.
#include "cpir_internal.hpp"

namespace cpir {
namespace {

__host__ __device__ inline uint64_t synth_u64(uint64_t seed, uint64_t index) {
  uint64_t z = seed * 0xD1342543DE82EF95ULL + (index + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) synth_fill_kernel(uint32_t* __restrict__ out, uint64_t count, uint64_t seed,
                                                         uint64_t index0, uint32_t mask) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride)
    out[i] = (uint32_t)(synth_u64(seed, index0 + i) >> 32) & mask;
}

}  // namespace

int launch_synth_fill(const Device* dev, uint32_t* out, uint64_t count, uint64_t seed, uint64_t index0, uint32_t mask,
                      hipStream_t stream) {
  if (!out && count) return CPIR_ERR_INVALID_ARGUMENT;
  if (count == 0) return CPIR_OK;
  uint64_t blocks = (count + 255) / 256;
  const uint64_t cap = (uint64_t)dev->num_cus * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, count, seed, index0, mask);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace cpir
