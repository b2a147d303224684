// matmul.hip -- the offline hint product  M = A . D  (u32, wrap-around), LDS-tiled on the integer VALU.
//
// Replaces gpu_utils::mat_x_mat + shaders/mat_x_mat.glsl (reference chalametpir_server/src/gpu/gpu_utils.rs:156-220,
// chalametpir_server/shaders/mat_x_mat.glsl:26-46: one thread per output element, full-K loop from global memory, B read
// with stride C) == impl Mul for &Matrix (chalametpir_common/src/matrix.rs:1040-1059).
//
//   M[r][c] = sum_k A[r][k] *wrap D[k][c]
//
// u32 wrap-around is not an MFMA input type, so this runs on the VALU and is compute-bound (1774*N*C MACs against
// 4*(1774*N + N*C + 1774*C) bytes: ~150 MAC/B at the 2^20-key config).  Two kernels:
//
//  * packed16 (the setup path: every D entry < 2^16 -- entries of the encoded DB are < 2^b <= 2^14):
//      A is split into 16-bit halves and TWO consecutive k are packed per dword while staging into LDS:
//        a_lo2 = (A[r][k] & 0xffff) | (A[r][k+1] << 16),  a_hi2 = (A[r][k] >> 16) | (A[r][k+1] & 0xffff0000),
//        d2    =  D[k][c]           | (D[k+1][c] << 16)
//      and each thread runs  acc_lo = v_dot2_u32_u16(a_lo2, d2, acc_lo),  acc_hi = v_dot2_u32_u16(a_hi2, d2, acc_hi);
//      M = acc_lo + (acc_hi << 16)  is identical mod 2^32.  One full-rate VALU op per MAC instead of
//      v_mul_lo_u32 (quarter rate) + v_add.
//  * general (any u32 D; keeps the C ABI a true mat_x_mat, e.g. the reference's A*I = A = I*A identity test,
//      matrix.rs:1275-1317): plain v_mul_lo_u32 + add on the same tiling.
//
// Tiling: 64 x 128 outputs per 256-thread block, 4 x 8 per thread, K staged 32 (packed) / 16 (general) deep through
// LDS with the next stage's global loads in flight during the current stage's math (register staging); split-K over
// blockIdx.z with u32 atomicAdd (exact, order-independent) so the 1774 x 940 hint still fills 256 CUs.
#include "cpir_internal.hpp"

namespace cpir {
namespace {

constexpr int BM = 64;
constexpr int BN = 128;
constexpr int kThreads = 256;
constexpr int TM = 4;
constexpr int TN = 8;
constexpr int PAD_A = 4;

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}

struct MatArgs {
  const uint32_t* A;
  const uint32_t* D;
  uint32_t* M;
  uint64_t lda, ldd, ldm;
  uint64_t rows, inner, cols;
  uint64_t k_per_split;  // multiple of the stage depth
  int atomic;            // accumulate with atomicAdd (split-K or accumulate mode)
};

// ---------------------------------------------------------------------------------------------------------------
// packed16 kernel
// ---------------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ void __launch_bounds__(kThreads) mat_x_mat_packed16_kernel(const MatArgs a) {
  constexpr int KS = 32;   // k per stage
  constexpr int KP = 16;   // k-pairs per stage
  __shared__ __attribute__((aligned(16))) uint32_t s_alo[KP][BM + PAD_A];
  __shared__ __attribute__((aligned(16))) uint32_t s_ahi[KP][BM + PAD_A];
  __shared__ __attribute__((aligned(16))) uint32_t s_d[KP][BN];

  const int tid = threadIdx.x;
  const int tx = tid & 15;  // along n
  const int ty = tid >> 4;  // along m
  const uint64_t m0 = (uint64_t)blockIdx.y * BM;
  const uint64_t n0 = (uint64_t)blockIdx.x * BN;
  const uint64_t k_begin = (uint64_t)blockIdx.z * a.k_per_split;
  const uint64_t k_end = (k_begin + a.k_per_split < a.inner) ? k_begin + a.k_per_split : a.inner;
  if (k_begin >= k_end) return;

  // staging roles
  const int a_row = tid >> 2;        // 0..63
  const int a_seg = tid & 3;         // 8 consecutive k each
  const int d_kp = tid >> 4;         // 0..15
  const int d_seg = tid & 15;        // 8 consecutive columns each

  uint32_t ra[8];       // A: 8 consecutive k of one row
  uint32_t rd0[8], rd1[8];  // D: rows 2*kp and 2*kp+1, 8 columns

  auto load_stage = [&](uint64_t k0) {
    {  // A
      const uint64_t r = m0 + a_row;
      const uint64_t k = k0 + (uint64_t)a_seg * 8;
      if (r < a.rows && VEC && k + 7 < k_end) {
        const uint4* p = reinterpret_cast<const uint4*>(a.A + r * a.lda + k);
        const uint4 u = p[0], v = p[1];
        ra[0] = u.x, ra[1] = u.y, ra[2] = u.z, ra[3] = u.w, ra[4] = v.x, ra[5] = v.y, ra[6] = v.z, ra[7] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) ra[i] = (r < a.rows && k + i < k_end) ? a.A[r * a.lda + k + i] : 0u;
      }
    }
    {  // D
      const uint64_t k = k0 + 2 * (uint64_t)d_kp;
      const uint64_t c = n0 + (uint64_t)d_seg * 8;
      const bool vec = VEC && c + 7 < a.cols;
      if (k < k_end && vec) {
        const uint4* p = reinterpret_cast<const uint4*>(a.D + k * a.ldd + c);
        const uint4 u = p[0], v = p[1];
        rd0[0] = u.x, rd0[1] = u.y, rd0[2] = u.z, rd0[3] = u.w, rd0[4] = v.x, rd0[5] = v.y, rd0[6] = v.z, rd0[7] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) rd0[i] = (k < k_end && c + i < a.cols) ? a.D[k * a.ldd + c + i] : 0u;
      }
      if (k + 1 < k_end && vec) {
        const uint4* p = reinterpret_cast<const uint4*>(a.D + (k + 1) * a.ldd + c);
        const uint4 u = p[0], v = p[1];
        rd1[0] = u.x, rd1[1] = u.y, rd1[2] = u.z, rd1[3] = u.w, rd1[4] = v.x, rd1[5] = v.y, rd1[6] = v.z, rd1[7] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) rd1[i] = (k + 1 < k_end && c + i < a.cols) ? a.D[(k + 1) * a.ldd + c + i] : 0u;
      }
    }
  };

  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t e = ra[2 * i], o = ra[2 * i + 1];
      s_alo[a_seg * 4 + i][a_row] = (e & 0xffffu) | (o << 16);
      s_ahi[a_seg * 4 + i][a_row] = (e >> 16) | (o & 0xffff0000u);
    }
    uint4 lo, hi;
    lo.x = (rd0[0] & 0xffffu) | (rd1[0] << 16);
    lo.y = (rd0[1] & 0xffffu) | (rd1[1] << 16);
    lo.z = (rd0[2] & 0xffffu) | (rd1[2] << 16);
    lo.w = (rd0[3] & 0xffffu) | (rd1[3] << 16);
    hi.x = (rd0[4] & 0xffffu) | (rd1[4] << 16);
    hi.y = (rd0[5] & 0xffffu) | (rd1[5] << 16);
    hi.z = (rd0[6] & 0xffffu) | (rd1[6] << 16);
    hi.w = (rd0[7] & 0xffffu) | (rd1[7] << 16);
    *reinterpret_cast<uint4*>(&s_d[d_kp][d_seg * 8]) = lo;
    *reinterpret_cast<uint4*>(&s_d[d_kp][d_seg * 8 + 4]) = hi;
  };

  uint32_t acc_lo[TM][TN], acc_hi[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) acc_lo[i][j] = acc_hi[i][j] = 0;

  load_stage(k_begin);
  for (uint64_t k0 = k_begin; k0 < k_end; k0 += KS) {
    store_stage();
    __syncthreads();
    if (k0 + KS < k_end) load_stage(k0 + KS);
#pragma unroll 4
    for (int kp = 0; kp < KP; kp++) {
      const uint4 al = *reinterpret_cast<const uint4*>(&s_alo[kp][ty * TM]);
      const uint4 ah = *reinterpret_cast<const uint4*>(&s_ahi[kp][ty * TM]);
      // this thread's 8 columns are two 16-byte runs (tx*4 and 64 + tx*4): conflict-free ds_read_b128
      const uint4 d0 = *reinterpret_cast<const uint4*>(&s_d[kp][tx * 4]);
      const uint4 d1 = *reinterpret_cast<const uint4*>(&s_d[kp][64 + tx * 4]);
      const uint32_t av_lo[TM] = {al.x, al.y, al.z, al.w};
      const uint32_t av_hi[TM] = {ah.x, ah.y, ah.z, ah.w};
      const uint32_t dv[TN] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
          acc_lo[i][j] = dot2(av_lo[i], dv[j], acc_lo[i][j]);
          acc_hi[i][j] = dot2(av_hi[i], dv[j], acc_hi[i][j]);
        }
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TM; i++) {
    const uint64_t r = m0 + (uint64_t)ty * TM + i;
    if (r >= a.rows) continue;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const uint64_t c = n0 + (uint64_t)(j < 4 ? tx * 4 + j : 64 + tx * 4 + (j - 4));
      if (c >= a.cols) continue;
      const uint32_t v = acc_lo[i][j] + (acc_hi[i][j] << 16);
      if (a.atomic) atomicAdd(a.M + r * a.ldm + c, v);
      else a.M[r * a.ldm + c] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// general u32 kernel
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) mat_x_mat_u32_kernel(const MatArgs a) {
  constexpr int KS = 16;
  __shared__ __attribute__((aligned(16))) uint32_t s_a[KS][BM + PAD_A];
  __shared__ __attribute__((aligned(16))) uint32_t s_d[KS][BN];

  const int tid = threadIdx.x;
  const int tx = tid & 15;
  const int ty = tid >> 4;
  const uint64_t m0 = (uint64_t)blockIdx.y * BM;
  const uint64_t n0 = (uint64_t)blockIdx.x * BN;
  const uint64_t k_begin = (uint64_t)blockIdx.z * a.k_per_split;
  const uint64_t k_end = (k_begin + a.k_per_split < a.inner) ? k_begin + a.k_per_split : a.inner;
  if (k_begin >= k_end) return;

  const int a_row = tid >> 2;   // 0..63
  const int a_seg = tid & 3;    // 4 consecutive k each
  const int d_k = tid >> 4;     // 0..15
  const int d_seg = tid & 15;   // 8 consecutive columns each

  uint32_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = 0;

  for (uint64_t k0 = k_begin; k0 < k_end; k0 += KS) {
    {
      const uint64_t r = m0 + a_row;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint64_t k = k0 + (uint64_t)a_seg * 4 + i;
        s_a[a_seg * 4 + i][a_row] = (r < a.rows && k < k_end) ? a.A[r * a.lda + k] : 0u;
      }
      const uint64_t k = k0 + d_k;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint64_t c = n0 + (uint64_t)d_seg * 8 + i;
        s_d[d_k][d_seg * 8 + i] = (k < k_end && c < a.cols) ? a.D[k * a.ldd + c] : 0u;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < KS; kk++) {
      const uint4 av4 = *reinterpret_cast<const uint4*>(&s_a[kk][ty * TM]);
      const uint4 d0 = *reinterpret_cast<const uint4*>(&s_d[kk][tx * 4]);
      const uint4 d1 = *reinterpret_cast<const uint4*>(&s_d[kk][64 + tx * 4]);
      const uint32_t av[TM] = {av4.x, av4.y, av4.z, av4.w};
      const uint32_t dv[TN] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] += av[i] * dv[j];
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TM; i++) {
    const uint64_t r = m0 + (uint64_t)ty * TM + i;
    if (r >= a.rows) continue;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const uint64_t c = n0 + (uint64_t)(j < 4 ? tx * 4 + j : 64 + tx * 4 + (j - 4));
      if (c >= a.cols) continue;
      if (a.atomic) atomicAdd(a.M + r * a.ldm + c, acc[i][j]);
      else a.M[r * a.ldm + c] = acc[i][j];
    }
  }
}

}  // namespace

static std::atomic<int> g_use_mfma{1};  // cpir_tuning_set("matmul.mfma", 0) keeps the VALU kernels (A/B timing, tests of both paths)
bool mfma_matmul_enabled() { return g_use_mfma.load() != 0; }
void set_mfma_matmul(bool on) { g_use_mfma.store(on ? 1 : 0); }
static std::atomic<int> g_mfma_pipeline{1};
int mfma_pipeline() { return g_mfma_pipeline.load(); }
void set_mfma_pipeline(int on) { g_mfma_pipeline.store(on); }
#ifdef CPIR_DIAG
static std::atomic<int> g_mfma_ablate{0};
int mfma_ablate() { return g_mfma_ablate.load(); }
void set_mfma_ablate(int bits) { g_mfma_ablate.store(bits); }
#endif

__global__ void __launch_bounds__(256) zero_matrix_kernel(uint32_t* __restrict__ M, uint64_t ldm, uint64_t rows, uint64_t cols) {
  const uint64_t total = rows * cols;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) M[(i / cols) * ldm + i % cols] = 0u;
}

int launch_zero_matrix(uint32_t* M, uint64_t ldm, uint64_t rows, uint64_t cols, hipStream_t stream) {
  if (!M || rows == 0 || cols == 0 || ldm < cols) return CPIR_ERR_INVALID_ARGUMENT;
  uint64_t blocks = (rows * cols + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_matrix_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, M, ldm, rows, cols);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

const char* mat_x_mat_kernel_name(uint32_t rhs_max_bits) {
  if (rhs_max_bits > 16) return "mat_x_mat_u32_kernel";
  if (!mfma_matmul_enabled()) return "mat_x_mat_packed16_kernel";
  return mfma_pipeline() ? "mat_x_mat_mfma_pipe_kernel" : "mat_x_mat_mfma_kernel";  // (the pipelined kernel is a template: <false> here, <true> behind cpir_op_mat_x_packed)
}

int launch_mat_x_mat(const Device* dev, const uint32_t* A, uint64_t lda, const uint32_t* D, uint64_t ldd, uint32_t* M,
                     uint64_t ldm, uint64_t rows, uint64_t inner, uint64_t cols, uint32_t rhs_max_bits, int accumulate,
                     hipStream_t stream) {
  if (!A || !D || !M) return CPIR_ERR_INVALID_ARGUMENT;
  if (rows == 0 || inner == 0 || cols == 0) return CPIR_ERR_INVALID_MATRIX_DIMENSION;  // Matrix::new, matrix.rs:45-55
  if (lda < inner || ldd < cols || ldm < cols) return CPIR_ERR_INVALID_ARGUMENT;
  if (rhs_max_bits == 0 || rhs_max_bits > 32) return CPIR_ERR_INVALID_ARGUMENT;
  if (mfma_matmul_enabled() && mfma_matmul_applicable(A, lda, inner, cols, rhs_max_bits)) {
    // the matrix-core path (matmul_mfma.hip); its prepared right-hand side lives in a scratch block released behind the launch
    void* ws = nullptr;
    CPIR_TRY(scratch_acquire(&ws, mfma_rhs_workspace_bytes(inner, cols, rows), stream));
    int st = launch_rhs_split(dev, D, ldd, inner, cols, ws, stream);
    if (st == CPIR_OK) st = launch_mat_x_mat_mfma(dev, A, lda, ws, inner, cols, M, ldm, rows, rows, accumulate, stream);
    const int st2 = scratch_release_after(ws, stream);
    return st != CPIR_OK ? st : st2;
  }
  const bool packed = rhs_max_bits <= 16;
  const uint64_t ks = packed ? 32 : 16;

  const uint64_t tiles_m = (rows + BM - 1) / BM;
  const uint64_t tiles_n = (cols + BN - 1) / BN;
  if (tiles_m > 65535 || tiles_n > 0x7fffffffull) return CPIR_ERR_INVALID_ARGUMENT;
  // split-K (u32 atomics keep the sum exact) so that the grid is a whole number of "rounds" of resident blocks: with 224
  // output tiles (the 1774 x 940 hint) and 3 resident blocks per CU, 4 splits = 896 blocks run as one full round of 768 plus
  // a round that keeps a sixth of the chip busy (58 % efficiency, measured 21 TMAC/s); 24 splits = 5376 blocks = exactly 7
  // rounds.  Pick the split count with the best round efficiency, fewest splits on ties.
  int occ = 0;
  const void* fn = packed ? reinterpret_cast<const void*>(mat_x_mat_packed16_kernel<true>) : reinterpret_cast<const void*>(mat_x_mat_u32_kernel);
  CPIR_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, kThreads, 0));
  if (occ < 1) occ = 1;
  const uint64_t resident = (uint64_t)dev->num_cus * (uint64_t)occ;
  const uint64_t tiles = tiles_m * tiles_n;
  uint64_t max_splits = (inner + 2047) / 2048;  // at least 2048 k per split: the epilogue's atomics stay negligible
  if (max_splits > 64) max_splits = 64;
  if (max_splits < 1) max_splits = 1;
  uint64_t splits = 1;
  double best = 0.0;
  for (uint64_t sct = 1; sct <= max_splits; sct++) {
    const uint64_t blocks = tiles * sct;
    const uint64_t rounds = (blocks + resident - 1) / resident;
    const double eff = (double)blocks / (double)(rounds * resident);
    if (eff > best + 0.02) best = eff, splits = sct;
  }
  uint64_t k_per_split = (inner + splits - 1) / splits;
  k_per_split = (k_per_split + ks - 1) / ks * ks;
  splits = (inner + k_per_split - 1) / k_per_split;

  MatArgs a;
  a.A = A, a.D = D, a.M = M, a.lda = lda, a.ldd = ldd, a.ldm = ldm;
  a.rows = rows, a.inner = inner, a.cols = cols;
  a.k_per_split = k_per_split;
  a.atomic = (splits > 1 || accumulate) ? 1 : 0;
  if (a.atomic && !accumulate)
    CPIR_TRY(launch_zero_matrix(M, ldm, rows, cols, stream));

  const dim3 grid((unsigned)tiles_n, (unsigned)tiles_m, (unsigned)splits);
  if (packed) {
    const bool vec = (lda % 4 == 0) && (ldd % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(D) % 16 == 0);
    if (vec) hipLaunchKernelGGL(mat_x_mat_packed16_kernel<true>, grid, dim3(kThreads), 0, stream, a);
    else hipLaunchKernelGGL(mat_x_mat_packed16_kernel<false>, grid, dim3(kThreads), 0, stream, a);
  } else {
    hipLaunchKernelGGL(mat_x_mat_u32_kernel, grid, dim3(kThreads), 0, stream, a);
  }
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace cpir
