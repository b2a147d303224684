// TEST INFRASTRUCTURE ONLY -- what the driver of the host state machine test may turn on the simulated device (sim_runtime.cpp)
#pragma once
#include <atomic>
#include <cstdint>

namespace simctl {
// the tuning values host_respond.hip asks the library for (respond.hip in the product)
extern std::atomic<int> inplace_seats, upload_streams, helper_spin_us, fill_timeout_us, read_once, batch_fusion, takes_slot_map;
// what the simulated device has done
extern std::atomic<uint64_t> kernels_launched, polled_kernels, polled_gave_up;
void set_chaos_us(uint32_t us);  // every operation of a simulated stream starts up to this many microseconds late (uniform)
void fail_next_copies(int n);    // the next n hipMemcpyAsync calls fail
int64_t live_blocks();           // device / page-locked blocks and events currently allocated
void device_sync();
}  // namespace simctl
