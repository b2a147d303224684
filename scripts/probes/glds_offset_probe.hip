// Probe: does the immediate offset of global_load_lds_dwordx4 move the LDS destination too, or only the global source?
//   hipcc --offload-arch=gfx950 -O3 -w scripts/probes/glds_offset_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(const uint32_t* src, uint32_t* out) {
  __shared__ uint32_t buf[1024];  // 4 KiB
  const uint32_t lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) buf[i] = 0xDEAD0000u + i;
  __syncthreads();
  const uint32_t off = lane * 16u;
  const uint32_t ldsaddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&buf[0]);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(off), "s"(ldsaddr), "s"(src) : "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = buf[i];
}
int main() {
  uint32_t *src, *out, h[1024], hs[1024];
  hipMalloc(&src, 8192); hipMalloc(&out, 4096);
  for (int i = 0; i < 1024; i++) hs[i] = i;  // word i of the source = i  (bytes 0..4095)
  hipMemcpy(src, hs, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
  hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
  // expected if the offset moves only the SOURCE: buf[0..255] = 256..511; if it moves BOTH: buf[256..511] = 256..511
  printf("buf[0]=%08x buf[255]=%08x buf[256]=%08x buf[511]=%08x buf[512]=%08x\n", h[0], h[255], h[256], h[511], h[512]);
  printf("%s\n", h[0] == 256 ? "offset moves the global source only" : (h[256] == 256 ? "offset moves source AND LDS destination" : "unexpected"));
  return 0;
}
