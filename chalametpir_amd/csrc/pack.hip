// pack.hip -- build the HBM-resident packed database from the encoded DB matrix D.
//
// Replaces, fused into one pass over D:
//   * gpu_utils::mat_transpose + shaders/mat_transpose.glsl   (reference chalametpir_server/src/gpu/gpu_utils.rs:222-281,
//     chalametpir_server/shaders/mat_transpose.glsl:21-36) == Matrix::transpose (chalametpir_common/src/matrix.rs:517-527)
//   * Matrix::row_wise_compress, which the reference runs single-threaded on the CPU after reading the transpose back
//     (chalametpir_server/src/server.rs:151-156, matrix.rs:98-205).
//
//   word(c, w) = sum_{j<cf} (D[cf*w + j][c] & (2^b - 1)) << (j * 32/cf)        missing tail fields = 0
//
// Roofline: HBM, 4*N*C bytes read + 4*C*ceil(N/cf) bytes written; no readback of the transpose to the host.
// D tiles are read with rows of D along the lanes' contiguous axis (coalesced), transposed through LDS, and written
// as 256-byte runs along the packed-word axis.  All padding of the device layout (cpir_dtc_layout) is written as zero.
#include "cpir_internal.hpp"

namespace cpir {
namespace {

constexpr int kTileCols = 64;   // columns of D (= rows of DtC) per block
constexpr int kTileWords = 64;  // packed words per block along n
constexpr int kThreads = 256;

template <int CF, bool VEC>
__global__ void __launch_bounds__(kThreads) transpose_compress_kernel(const uint32_t* __restrict__ D, uint64_t ldd, uint64_t N,
                                                                       uint32_t C, uint32_t b, uint32_t* __restrict__ dtc,
                                                                       uint64_t row_stride, uint32_t rows_padded,
                                                                       uint32_t* __restrict__ or_of_entries) {
  constexpr int S = 32 / CF;
  constexpr int kRows = CF * kTileWords;  // slots (rows of D) per block
  __shared__ uint16_t tile[kRows][kTileCols + 2];  // fields are < 2^14 after masking

  const int tid = threadIdx.x;
  const uint64_t n0 = (uint64_t)blockIdx.x * kRows;
  const uint32_t c0 = blockIdx.y * kTileCols;
  const uint32_t mask = (1u << b) - 1u;
  uint32_t seen = 0;

  // ---- load: 16 lanes x 4 columns cover the 64 tile columns of one D row; 16 rows of D per pass -----------------
  const int lc = (tid & 15) * 4;
  const int lr = tid >> 4;
#pragma unroll 4
  for (int rr = lr; rr < kRows; rr += kThreads / 16) {
    const uint64_t n = n0 + rr;
    uint32_t v[4] = {0, 0, 0, 0};
    if (n < N) {
      const uint32_t* src = D + n * ldd + c0 + lc;
      if (VEC && c0 + lc + 3 < C) {
        const uint4 t = *reinterpret_cast<const uint4*>(src);
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (c0 + lc + i < C) v[i] = src[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      seen |= v[i];
      tile[rr][lc + i] = (uint16_t)(v[i] & mask);  // matrix.rs:121,149,181: every field is masked to b bits
    }
  }
  __syncthreads();

  // ---- store: each wave owns 16 of the 64 output rows; lanes run along the packed-word axis (256 B per row) -----
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint64_t w = (uint64_t)blockIdx.x * kTileWords + lane;
  if (w < row_stride) {
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
      const int cl = wave * 16 + i;
      const uint32_t c = c0 + cl;
      if (c < rows_padded) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < CF; j++) word |= (uint32_t)tile[CF * lane + j][cl] << (S * j);
        dtc[(uint64_t)c * row_stride + w] = word;  // rows >= C and slots >= N were loaded as zero
      }
    }
  }

  if (or_of_entries) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) seen |= __shfl_xor(seen, off, 64);
    if (lane == 0 && seen) atomicOr(or_of_entries, seen);
  }
}

// Normalise one packed word of the reference's compressed matrix: keep b bits per slot and drop the fields that lie
// beyond N in the last word (the reference ignores them by bounds-checking the query index, matrix.rs:360-375,399-421,446-475).
template <int CF>
__device__ __forceinline__ uint32_t normalise_word(uint32_t word, uint64_t w, uint64_t N, uint32_t mask) {
  constexpr int S = 32 / CF;
  uint32_t out = 0;
#pragma unroll
  for (int j = 0; j < CF; j++) {
    const uint32_t f = (word >> (S * j)) & mask;
    if (w * CF + j < N) out |= f << (S * j);
  }
  return out;
}

template <int CF>
__global__ void __launch_bounds__(kThreads) dtc_import_kernel(const uint32_t* __restrict__ src, uint64_t W, uint64_t N, uint32_t C,
                                                               uint32_t b, uint32_t* __restrict__ dtc, uint64_t row_stride,
                                                               uint32_t rows_padded) {
  const uint32_t mask = (1u << b) - 1u;
  const uint64_t total = (uint64_t)rows_padded * row_stride;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kThreads) {
    const uint64_t c = i / row_stride, w = i % row_stride;
    uint32_t v = 0;
    if (c < C && w < W) v = normalise_word<CF>(src[c * W + w], w, N, mask);
    dtc[i] = v;
  }
}

__global__ void __launch_bounds__(kThreads) dtc_export_kernel(const uint32_t* __restrict__ dtc, uint64_t row_stride, uint64_t W,
                                                               uint32_t C, uint32_t* __restrict__ dst) {
  const uint64_t total = (uint64_t)C * W;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kThreads) {
    const uint64_t c = i / W, w = i % W;
    dst[i] = dtc[c * row_stride + w];
  }
}

int check_layout(const cpir_dtc_layout& L) {
  const uint32_t cf = compression_factor(L.mat_elem_bit_len);
  if (cf == 0) return CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH;
  cpir_dtc_layout want;
  CPIR_TRY(dtc_layout_for(L.num_slots, L.num_cols, L.mat_elem_bit_len, &want));
  if (memcmp(&want, &L, sizeof(L)) != 0) return CPIR_ERR_INVALID_ARGUMENT;
  return CPIR_OK;
}

uint32_t grid_for(const Device* dev, uint64_t total) {
  uint64_t blocks = (total + kThreads - 1) / kThreads;
  const uint64_t cap = (uint64_t)dev->num_cus * 8;
  if (blocks > cap) blocks = cap;
  return (uint32_t)(blocks ? blocks : 1);
}

}  // namespace

int launch_transpose_compress(const Device* dev, const uint32_t* D, uint64_t ldd, const cpir_dtc_layout& L, uint32_t* dtc,
                              uint32_t* or_of_entries, hipStream_t stream) {
  (void)dev;
  if (!D || !dtc || ldd < L.num_cols) return CPIR_ERR_INVALID_ARGUMENT;
  CPIR_TRY(check_layout(L));
  const uint32_t cf = L.compression_factor;
  const dim3 grid((unsigned)(L.words_per_row_padded / kTileWords), (L.rows_padded + kTileCols - 1) / kTileCols);
  if (L.words_per_row_padded / kTileWords > 0x7fffffffull) return CPIR_ERR_INVALID_ARGUMENT;
  const bool vec = (ldd % 4 == 0) && (reinterpret_cast<uintptr_t>(D) % 16 == 0);
#define LAUNCH(CF_, VEC_)                                                                                            \
  hipLaunchKernelGGL((transpose_compress_kernel<CF_, VEC_>), grid, dim3(kThreads), 0, stream, D, ldd, L.num_slots,   \
                     L.num_cols, L.mat_elem_bit_len, dtc, L.words_per_row_padded, L.rows_padded, or_of_entries)
  if (cf == 2) { if (vec) LAUNCH(2, true); else LAUNCH(2, false); }
  else if (cf == 3) { if (vec) LAUNCH(3, true); else LAUNCH(3, false); }
  else { if (vec) LAUNCH(4, true); else LAUNCH(4, false); }
#undef LAUNCH
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

int launch_dtc_import(const Device* dev, const uint32_t* compressed, const cpir_dtc_layout& L, uint32_t* dtc, hipStream_t stream) {
  if (!compressed || !dtc) return CPIR_ERR_INVALID_ARGUMENT;
  CPIR_TRY(check_layout(L));
  const uint32_t grid = grid_for(dev, L.total_words);
#define LAUNCH(CF_)                                                                                                     \
  hipLaunchKernelGGL((dtc_import_kernel<CF_>), dim3(grid), dim3(kThreads), 0, stream, compressed, L.words_per_row,      \
                     L.num_slots, L.num_cols, L.mat_elem_bit_len, dtc, L.words_per_row_padded, L.rows_padded)
  if (L.compression_factor == 2) LAUNCH(2);
  else if (L.compression_factor == 3) LAUNCH(3);
  else LAUNCH(4);
#undef LAUNCH
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

int launch_dtc_export(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, uint32_t* compressed, hipStream_t stream) {
  if (!compressed || !dtc) return CPIR_ERR_INVALID_ARGUMENT;
  CPIR_TRY(check_layout(L));
  const uint32_t grid = grid_for(dev, (uint64_t)L.num_cols * L.words_per_row);
  hipLaunchKernelGGL(dtc_export_kernel, dim3(grid), dim3(kThreads), 0, stream, dtc, L.words_per_row_padded, L.words_per_row,
                     L.num_cols, compressed);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace cpir
