// TEST INFRASTRUCTURE ONLY -- a SIMULATED HIP runtime, just enough of the API for chalametpir_amd/csrc/host_respond.hip (the host state
// machine of Server::respond: seats, arenas, in-place rounds, polled launches, the group workers) to be compiled as plain C++ and driven on
// the CPU under ThreadSanitizer / AddressSanitizer (tests/test_host_state_machine.py, tests/native/host_state_machine_driver.cpp).
// Nothing under chalametpir_amd/ includes this file: the product is built against /opt/rocm/include; this directory is put in FRONT of the
// include path by the test's own compile command and nowhere else.
//
// What is simulated (tests/native/sim_hip/sim_runtime.cpp): streams are FIFO queues, each drained by a thread of its own; events, stream
// waits, asynchronous copies and memsets are operations in those queues; "device memory" is host memory; page-locked host memory is host
// memory listed in a registry that hipPointerGetAttributes / hipDrvPointerGetAttributes answer from; a kernel launch
// (hipLaunchKernelGGL) runs the __global__ function once per (block, thread) on the stream's thread with threadIdx / blockIdx set --
// threads of a block in DESCENDING order and __syncthreads() a no-op, which is faithful for the two small kernels of host_respond.hip
// (their only work behind a barrier is thread 0's).  The respond kernels themselves are replaced by CPU stand-ins with the same
// contract, including the polled fill protocol (sim_runtime.cpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>

enum hipError_t {
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorOutOfMemory = 2,
  hipErrorNoDevice = 100,
  hipErrorInvalidDevice = 101,
  hipErrorPeerAccessUnsupported = 217,
  hipErrorNotReady = 600,
  hipErrorPeerAccessAlreadyEnabled = 704,
  hipErrorStreamCaptureUnsupported = 900,
  hipErrorUnknown = 999,
};

struct sim_stream;
struct sim_event;
typedef sim_stream* hipStream_t;
typedef sim_event* hipEvent_t;
typedef void* hipDeviceptr_t;

enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t {
  hipMemoryType type;
  int device;
  void* devicePointer;
  void* hostPointer;
};
enum hipPointer_attribute { HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR = 11, HIP_POINTER_ATTRIBUTE_RANGE_SIZE = 12 };

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct sim_idx {
  unsigned x, y, z;
};
extern thread_local sim_idx threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __launch_bounds__(...)
#define __forceinline__ inline

constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocCoherent = 0x40000000u;
constexpr int __HIP_MEMORY_SCOPE_SYSTEM = 5;

// ---- memory ---------------------------------------------------------------------------------------------------------------------------
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipHostRegister(void* p, size_t bytes, unsigned flags);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* p);
hipError_t hipDrvPointerGetAttributes(unsigned n, hipPointer_attribute* which, void** out, hipDeviceptr_t p);
template <class T>
inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
template <class T>
inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags) { return hipHostMalloc(reinterpret_cast<void**>(p), bytes, flags); }

// ---- devices, errors ------------------------------------------------------------------------------------------------------------------
hipError_t hipGetDevice(int* ordinal);
hipError_t hipSetDevice(int ordinal);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipDeviceCanAccessPeer(int* can, int from, int to);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);

// ---- streams and events ---------------------------------------------------------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int priority);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t ev, unsigned flags);
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) {
  *st = hipStreamCaptureStatusNone;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t ev);
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t s);
hipError_t hipEventQuery(hipEvent_t ev);
hipError_t hipEventSynchronize(hipEvent_t ev);

// ---- asynchronous copies --------------------------------------------------------------------------------------------------------------
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);

// ---- kernels --------------------------------------------------------------------------------------------------------------------------
// enqueue `body` on `s`: run once per (block, thread), threads of a block in descending order (see the head of this file)
void sim_enqueue_kernel(hipStream_t s, dim3 grid, dim3 block, std::function<void()> body);
// any work of the simulated device, as one operation of a stream (the stand-ins of the respond kernels)
void sim_enqueue(hipStream_t s, std::function<void()> op);
template <class K, class... A>
inline void sim_launch_kernel(hipStream_t s, dim3 grid, dim3 block, K kernel, A... args) {  // (arguments are taken BY VALUE at launch time, as a real launch does)
  sim_enqueue_kernel(s, grid, block, [=]() { kernel(args...); });
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) sim_launch_kernel((stream), (grid), (block), (kernel), __VA_ARGS__)

#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __syncthreads() {}
