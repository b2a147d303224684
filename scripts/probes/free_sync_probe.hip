// free_sync_probe.hip -- which runtime calls wait for work in flight on OTHER (non-blocking, prioritised) streams of the device?
// A kernel spins for ~200 ms on a non-blocking high-priority stream; each candidate call is timed while it runs.  The server's
// destroy path (host_respond.hip: server_destroy) must not free memory a kernel may still touch, and does not want to rely on an
// undocumented implicit wait.   hipcc --offload-arch=gfx950 -O2 free_sync_probe.hip -o free_sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (out) atomicAdd(out, 1u);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("FAILED %s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

int main() {
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  const unsigned long long ticks = 20000000ull;  // 200 ms at 100 MHz
  for (int what = 0; what < 6; what++) {
    hipStream_t s;
    CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    unsigned* flag;
    CK(hipMalloc(&flag, 4));
    CK(hipMemset(flag, 0, 4));
    void *dbuf = nullptr, *hbuf = nullptr;
    CK(hipMalloc(&dbuf, 1 << 20));
    CK(hipHostMalloc(&hbuf, 1 << 20, hipHostMallocDefault));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, flag);
    CK(hipEventRecord(ev, s));
    const double t0 = now();
    const char* name = "";
    switch (what) {
      case 0: name = "hipFree(unrelated device buffer)"; CK(hipFree(dbuf)); dbuf = nullptr; break;
      case 1: name = "hipHostFree(unrelated pinned buffer)"; CK(hipHostFree(hbuf)); hbuf = nullptr; break;
      case 2: name = "hipEventDestroy(pending event)"; CK(hipEventDestroy(ev)); ev = nullptr; break;
      case 3: name = "hipStreamDestroy(busy stream)"; CK(hipStreamDestroy(s)); s = nullptr; break;
      case 4: name = "hipMalloc"; { void* p; CK(hipMalloc(&p, 1 << 20)); CK(hipFree(p)); name = "hipMalloc+hipFree(fresh buffer)"; } break;
      case 5: name = "hipHostMalloc"; { void* p; CK(hipHostMalloc(&p, 1 << 20, 0)); const double t1 = now(); printf("  hipHostMalloc alone: %.1f ms\n", (t1 - t0) * 1e3); CK(hipHostFree(p)); name = "hipHostMalloc+hipHostFree(fresh buffer)"; } break;
    }
    const double dt = now() - t0;
    printf("%-45s returned after %7.1f ms -> %s\n", name, dt * 1e3, dt > 0.15 ? "WAITS for the other stream's kernel" : "does NOT wait");
    CK(hipDeviceSynchronize());
    if (dbuf) CK(hipFree(dbuf));
    if (hbuf) CK(hipHostFree(hbuf));
    if (ev) CK(hipEventDestroy(ev));
    if (s) CK(hipStreamDestroy(s));
    CK(hipFree(flag));
  }
  return 0;
}
