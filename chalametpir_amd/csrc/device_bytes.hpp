// Device-side byte shuffling shared by pack.hip and matmul_mfma.hip (HIP translation units only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace cpir {

// byte k of four words -> one dword (byte t of the result = byte k of the t-th argument); sel01 = k | (4 + k) << 8
__device__ __forceinline__ uint32_t gather_byte4(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t sel01) {
  const uint32_t p01 = __builtin_amdgcn_perm(y, x, sel01);  // byte 0 = x.byte[k], byte 1 = y.byte[k]   (sel01 = k | (4 + k) << 8)
  const uint32_t p23 = __builtin_amdgcn_perm(w, z, sel01);
  return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

// position of 16-byte piece p = 64*T + 16*g + 4*a + i inside a wave's staging window: the low two bits are XOR-ed with (a >> 1) | (T & 1) << 1
// so that the 8 lanes a ds_write_b128 serves per cycle (a = 0..3, two values of T; i and g fixed) hit 8 different 16-byte bank groups;
// a permutation inside aligned groups of 4 pieces, so the linear read-back stays conflict-free
__device__ __forceinline__ uint32_t stage_swz(uint32_t p) { return p ^ (((p >> 3) & 1u) | (((p >> 6) & 1u) << 1)); }


}  // namespace cpir
