// matmul_mfma.hip -- the offline hint product  M = A . D  (u32, wrap-around) on the i8 matrix cores.
//
// Replaces gpu_utils::mat_x_mat + shaders/mat_x_mat.glsl (reference chalametpir_server/src/gpu/gpu_utils.rs:156-220,
// chalametpir_server/shaders/mat_x_mat.glsl:26-46) == impl Mul for &Matrix (chalametpir_common/src/matrix.rs:1040-1059) whenever every entry
// of D is below 2^16 (any encoded database: entries < 2^b <= 2^14); matmul.hip keeps the VALU kernels for general u32 right-hand sides and
// for operands that are not 16-byte loadable.
//
// u32 wrap-around products are not an MFMA type, but the product splits exactly into signed-byte products (the same split as
// respond_planar.hip, with BOTH bytes of D offset):
//
//   A[r][k] = sum_{i<4} 2^(8i) (as_i + 128)          as_i = byte i of A[r][k] XOR 0x80, read as a signed byte
//   D[k][c] = (d_0 + 128) + 256 (d_1 + 128)          d_j  = byte j of D[k][c] XOR 0x80
//
//   sum_k A D = sum_{s<4} 2^(8s) * [ sum_{i+j=s} sum_k as_i d_j ]          <- 7 MFMAs per 16 rows x 16 columns x 64 k (i + j = 4 vanishes mod 2^32)
//             + 0x8080 * sum_k A[r][k]  +  0x80808080 * sum_k D[k][c]  -  K * 0x80808080 * 0x8080            (mod 2^32)
//
// The i32 accumulators of v_mfma_i32_16x16x64_i8 wrap (scripts/mfma_i8_probe.hip), so everything is exact mod 2^32.  Products that share
// the shift s = i + j share an accumulator, so a 16 x 16 output tile costs 4 accumulators, not 8.
//
// Three launches:
//   1. rhs_split_kernel: D (inner x cols, u32 row-major) -> two byte planes in MFMA B-operand order ([column tile of 16][k-step of 64]
//      [byte j][1 KiB]: lane 16*g + c holds bytes d_j of column c, k = 64*ks + 16*g + 0..15), zero bytes as padding, plus the column sums.
//      Once per right-hand side: Server::setup multiplies many row blocks of A by the same D.
//   2. mat_x_mat_mfma_kernel: 128 x 128 output tile per 512-thread block (8 waves as 2 x 4, 64 x 32 per wave, 128 accumulator registers),
//      K in steps of 64 through double-buffered LDS (96 KiB): the D planes arrive by LDS-DMA (global_load_lds, 1 KiB pieces as stored),
//      A is loaded 128 B per row segment, split into its four byte limbs in registers (v_perm_b32) and written as MFMA A-operand pieces;
//      blocks that own column tile 0 also sum the rows of A.  Persistent grid: the K axis is split 8 ways by blockIdx % 8 (blocks that
//      share an XCD then work on the same K range and share their A and D tiles in that XCD's L2), inside an XCD the 32 blocks take
//      (row tile, column tile) pairs of one K sub-range at a time.  Results leave through u32 atomics (order-independent, exact).
//   3. hint_fixup_kernel: adds the three correction terms.
#include "cpir_internal.hpp"
#include "device_bytes.hpp"

namespace cpir {
namespace {

constexpr int kThreads = 256;   // rhs_split_kernel
constexpr int kMT = 512;        // mat_x_mat_mfma_kernel: 8 waves
constexpr uint32_t kBM = 128, kBN = 128, kBK = 64;
constexpr uint32_t kPiecesA = 8 * 4, kPiecesB = 8 * 2;  // 1 KiB pieces per LDS buffer: [8 row tiles of 16][4 limbs], [8 column tiles of 16][2 bytes]

typedef int v4i __attribute__((ext_vector_type(4)));

// Where the 16 bytes of A-operand lane l = 16 * g + r (k-group g, row r of the 16-row tile) sit inside a 1 KiB A piece in LDS: at slot l with
// bits 2 and 4 swapped (r's bit 2 <-> g's bit 0).  The conversion of A writes ONE dword per lane and limb (ds_write_b32: two groups of 32
// lanes, bank = dword address mod 32), and a wave-instruction covers 4 rows x both g of a pair: in lane order the two g land 64 dwords
// apart, i.e. on the same banks -- every one of the 16 writes of a wave and k-step paid a 2-way conflict (SQ_LDS_BANK_CONFLICT: exactly 32
// extra cycles per wave and k-step, 5.3e8 per hint at 2^20 keys).  With the swap the low three slot bits of a write group are (g0, r1, r0):
// 8 distinct values x 4 dwords = all 32 banks once.  The fragment reads (ds_read_b128: four groups of 16 lanes, {0-3,12-15,20-27} ..., bank =
// dword address mod 64) stay conflict-free: each group's slots still cover the 16 residues mod 16 once.  The D pieces arrive by LDS-DMA in
// stored order and are not affected.
__device__ __forceinline__ uint32_t a_slot(uint32_t l) { return (l & 0x2Bu) | ((l >> 2) & 1u) << 4 | ((l >> 4) & 1u) << 2; }

// ---------------------------------------------------------------------------------------------------------------
// 1. right-hand side -> byte planes + column sums
// ---------------------------------------------------------------------------------------------------------------
struct SplitArgs {
  const uint32_t* D;
  uint64_t ld, inner;
  uint32_t cols, ct16, stripe_groups, ks_total;  // ct16: column tiles of 16 in the planes (padded to whole 128-column block tiles)
  uint4* planes;
  uint32_t* colsum;
};

// One wave = 64 columns (4 column tiles) x 8 consecutive k-steps; structure and staging window as planar_pack_stream_kernel (pack.hip):
// lane (g, lc) loads 16 rows x 16 bytes per k-step and holds, for each of its 4 columns, the 16 consecutive k of one operand piece.
template <bool VEC>
__global__ void __launch_bounds__(kThreads) rhs_split_kernel(const SplitArgs a) {
  __shared__ uint4 stage[kThreads / 64][256];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t g = lane >> 4, lc = lane & 15;
  const uint32_t sg = blockIdx.x % a.stripe_groups;
  const uint32_t ks0 = (blockIdx.x / a.stripe_groups) * 8;
  const uint32_t stripe = sg * 4 + wave;
  const uint32_t c0 = stripe * 64 + 4 * lc;
  bool cvalid[4];
#pragma unroll
  for (int i = 0; i < 4; i++) cvalid[i] = c0 + i < a.cols;
  const uint32_t csafe = cvalid[0] ? c0 : 0;  // VEC: c0 < cols <= ld with c0 and ld multiples of 4, so 16 bytes at c0 are inside the row
  uint4* const my_stage = stage[wave];
  const uint32_t piece0 = 64 * (lc >> 2) + 16 * g + 4 * (lc & 3);
  uint32_t sum[4] = {0, 0, 0, 0};

#pragma unroll
  for (int kk = 0; kk < 8; kk++) {
    const uint32_t ks = ks0 + kk;
    if (ks >= a.ks_total) break;  // wave-uniform
    uint32_t v[16][4];
    const uint64_t nb = (uint64_t)ks * kBK + 16 * g;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t n = nb + j;
      const uint64_t nr = n < a.inner ? n : a.inner - 1;
      if constexpr (VEC) {
        const uint4 t = *reinterpret_cast<const uint4*>(a.D + nr * a.ld + csafe);
        v[j][0] = t.x, v[j][1] = t.y, v[j][2] = t.z, v[j][3] = t.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) v[j][i] = a.D[nr * a.ld + (cvalid[i] ? c0 + i : 0)];
      }
    }
    // padding (k >= inner or column >= cols) must contribute nothing: its plane bytes are 0, i.e. the entry reads 0x8080 before the XOR
    uint32_t W[4][2][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int d = 0; d < 4; d++) {
        uint32_t e[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const bool ok = cvalid[i] && nb + 4 * d + t < a.inner;
          const uint32_t x = v[4 * d + t][i];
          sum[i] += ok ? x : 0u;
          e[t] = ok ? x : 0x8080u;
        }
        W[i][0][d] = gather_byte4(e[0], e[1], e[2], e[3], 0x0400u) ^ 0x80808080u;
        W[i][1][d] = gather_byte4(e[0], e[1], e[2], e[3], 0x0501u) ^ 0x80808080u;
      }
#pragma unroll
    for (int i = 0; i < 4; i++) asm volatile("" : "+v"(sum[i]));  // keep this k-step's values from living on (see pack.hip)
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
      for (int i = 0; i < 4; i++) my_stage[stage_swz(piece0 + i)] = make_uint4(W[i][j][0], W[i][j][1], W[i][j][2], W[i][j][3]);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 x = my_stage[stage_swz(64 * k + lane)];
        const uint32_t T = stripe * 4 + k;
        if (T < a.ct16) a.planes[(((uint64_t)T * a.ks_total + ks) * 2 + j) * 64 + lane] = x;
      }
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t s = sum[i];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0 && s) atomicAdd(a.colsum + c0 + i, s);  // s != 0 implies c0 + i < cols
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 2. the product
// ---------------------------------------------------------------------------------------------------------------
struct MfmaArgs {
  const uint32_t* A;
  uint64_t lda;
  const uint4* planes;
  uint32_t* M;
  uint64_t ldm;
  uint32_t* rowsum;
  uint64_t rows, inner, cols;
  uint32_t RT, CT, KS;  // row tiles of 128, column tiles of 128, k-steps of 64
  uint32_t S;           // K sub-ranges per XCD range
  uint32_t nx;          // K axis split by blockIdx % nx (8, or 1 for tiny grids)
  CPIR_DIAG_ONLY(uint32_t ablate;)  // a diagnosis build only (results are garbage): 1 no MFMA, 2 no A conversion / LDS writes, 4 no A loads, 8 no D DMA
  // Right-hand side straight from a planar respond image (pipelined kernel only; `planes` is NULL then): the low-byte operand pieces ARE
  // the first 8 KiB of every super-tile of the image ([column tile][step of 512 slots][(8 + HB) KiB]), the high-byte pieces come from the
  // plane pack.hip writes next to it ([column tile][k-block of 64][1 KiB]).  Slots and columns past the end hold field 0 there (bytes
  // 0x80, not 0x00 as in `planes`), so the kernel feeds A = 0x80808080 (all limbs zero) for k >= inner.
  // With ONE bit plane (b = 9) there is no high-byte plane at all (hi_plane NULL): the high byte of a field is that one bit, and every wave
  // expands its dword of the image's bit plane into the 1 KiB operand piece in registers (8 VALU operations per k-step) -- the pack pass
  // then writes nothing but the image, 1.11 GB less at 2^20 keys.
  const uint4* lo_tiles;
  const uint4* hi_plane;
  uint32_t lo_st16;      // uint4 per super-tile: (8 + HB) * 64
  uint32_t lo_ks512;     // super-tile steps per column tile
  uint32_t b_col_tiles;  // column tiles of 16 present in the image and the plane
  uint32_t kb_total;     // k-blocks per column tile in hi_plane
};

__global__ void __launch_bounds__(kMT) __attribute__((amdgpu_waves_per_eu(2, 2))) mat_x_mat_mfma_kernel(const MfmaArgs a) {
  __shared__ uint4 lds[2][(kPiecesA + kPiecesB) * 64];  // 2 x 48 KiB

  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wm = wave >> 2, wn = wave & 3;  // this wave's 64 x 32 part of the block tile
  const uint32_t xcd = blockIdx.x % a.nx, slot = blockIdx.x / a.nx, slots = gridDim.x / a.nx;  // host: gridDim.x % nx == 0
  const uint32_t ksx0 = (uint32_t)(((uint64_t)a.KS * xcd) / a.nx), ksx1 = (uint32_t)(((uint64_t)a.KS * (xcd + 1)) / a.nx);
  const uint32_t pairs = a.RT * a.CT;
  const uint64_t units = (uint64_t)pairs * a.S;

  // ---- A staging roles: wave-instruction x = 8*j + wave (j < 4) covers 8 rows x 128 bytes: half h = x & 1 of the 64 k, rows 8*(x >> 1) .. + 7
  const uint32_t r8 = lane >> 3, q8 = lane & 7;
  uint32_t row_l[4], wdw[4];
  uint32_t kq[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t x = 8 * j + wave, h = x & 1, rb = x >> 1;
    row_l[j] = 8 * rb + r8;
    const uint32_t q = 8 * h + q8;  // 16-byte quad of the row's 256 bytes: k = 4q .. 4q + 3
    kq[j] = 4 * q;
    // dword index (inside the A region of a buffer) of limb 0: piece (row tile of 16, limb), slot 16*(q >> 2) + row % 16, dword q & 3
    wdw[j] = (((row_l[j] >> 4) * 4) * 64 + a_slot(16 * (q >> 2) + (row_l[j] & 15))) * 4 + (q & 3);
  }

  for (uint64_t u = slot; u < units; u += slots) {
    const uint32_t sub = (uint32_t)(u / pairs), pair = (uint32_t)(u % pairs), rt = pair / a.CT, ct = pair % a.CT;
    const uint32_t k0 = ksx0 + (uint32_t)(((uint64_t)(ksx1 - ksx0) * sub) / a.S);
    const uint32_t k1 = ksx0 + (uint32_t)(((uint64_t)(ksx1 - ksx0) * (sub + 1)) / a.S);
    if (k0 >= k1) continue;  // block-uniform

    const uint32_t* arow[4];
    bool rvalid[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint64_t r = (uint64_t)rt * kBM + row_l[j];
      rvalid[j] = r < a.rows;
      arow[j] = a.A + (rvalid[j] ? r : a.rows - 1) * a.lda;  // rows past the end re-read the last row; their outputs are dropped
    }
    const bool sum_rows = (ct == 0);  // exactly one block per (row tile, K sub-range) adds up the rows of A
    uint32_t rs[4] = {0, 0, 0, 0};

    v4i acc[4][2][4];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int s = 0; s < 4; s++) acc[m][n][s] = v4i{0, 0, 0, 0};

    uint4 ra[4];
    auto load_a = [&](uint32_t ks) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint64_t k = (uint64_t)ks * kBK + kq[j];
        // inner % 4 == 0 (host): a quad lies wholly inside or wholly outside; outside it meets zero plane bytes, so any readable address does
        ra[j] = *reinterpret_cast<const uint4*>(arow[j] + (k < a.inner ? k : 0));
      }
    };
    auto dma_b = [&](uint32_t ks, uint32_t buf) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const uint4* src = a.planes + (((uint64_t)(ct * 8 + wave) * a.KS + ks) * 2 + e) * 64 + lane;
        uint4* dst = &lds[buf][(kPiecesA + wave * 2 + e) * 64];  // wave-uniform base; the DMA adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    };
    auto store_a = [&](uint32_t ks, uint32_t buf) {
      uint32_t* base = reinterpret_cast<uint32_t*>(&lds[buf][0]);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint4 t = ra[j];
        if (sum_rows) {
          const bool in = rvalid[j] && (uint64_t)ks * kBK + kq[j] < a.inner;
          rs[j] += in ? t.x + t.y + t.z + t.w : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
          base[wdw[j] + i * 256] = gather_byte4(t.x, t.y, t.z, t.w, (uint32_t)i | ((4u + i) << 8)) ^ 0x80808080u;
      }
    };
    auto mfma_step = [&](uint32_t buf) {
      const uint4* A_ = &lds[buf][0];
      const uint4* B_ = &lds[buf][kPiecesA * 64];
      v4i bf[2][2];
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const uint4 t = B_[((wn * 2 + n) * 2 + j) * 64 + lane];
          bf[n][j] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
        }
#pragma unroll
      for (int m = 0; m < 4; m++) {
        v4i af[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint4 t = A_[((wm * 4 + m) * 4 + i) * 64 + a_slot(lane)];
          af[i] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
          acc[m][n][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0], bf[n][0], acc[m][n][0], 0, 0, 0);
          acc[m][n][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[1], bf[n][0], acc[m][n][1], 0, 0, 0);
          acc[m][n][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[2], bf[n][0], acc[m][n][2], 0, 0, 0);
          acc[m][n][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[3], bf[n][0], acc[m][n][3], 0, 0, 0);
          acc[m][n][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0], bf[n][1], acc[m][n][1], 0, 0, 0);
          acc[m][n][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[1], bf[n][1], acc[m][n][2], 0, 0, 0);
          acc[m][n][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[2], bf[n][1], acc[m][n][3], 0, 0, 0);
        }
      }
    };

    // prologue: first k-step into buffer 0
    load_a(k0);
    dma_b(k0, 0);
    store_a(k0, 0);
    __syncthreads();  // also drains the DMA (vmcnt(0)) before anybody reads
    for (uint32_t ks = k0; ks < k1; ks++) {
      const uint32_t buf = (ks - k0) & 1;
      const bool more = ks + 1 < k1;
      if (more) {
        CPIR_DIAG_ONLY(if (!(a.ablate & 4))) load_a(ks + 1);
        CPIR_DIAG_ONLY(if (!(a.ablate & 8))) dma_b(ks + 1, buf ^ 1);  // that buffer was last read in the previous iteration, which ended with a barrier
      }
      CPIR_DIAG_ONLY(if (!(a.ablate & 1))) mfma_step(buf);
      if (more CPIR_DIAG_ONLY(&& !(a.ablate & 2))) store_a(ks + 1, buf ^ 1);
      __syncthreads();
    }

    // ---- this unit's part of the output tile: sum_s acc_s << 8s, one u32 atomic per element ----
    const uint32_t fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int n = 0; n < 2; n++) {
        const uint64_t c = (uint64_t)ct * kBN + wn * 32 + n * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint64_t r = (uint64_t)rt * kBM + wm * 64 + m * 16 + fq * 4 + i;
          const uint32_t v = (uint32_t)acc[m][n][0][i] + ((uint32_t)acc[m][n][1][i] << 8) + ((uint32_t)acc[m][n][2][i] << 16) +
                             ((uint32_t)acc[m][n][3][i] << 24);
          if (r < a.rows && c < a.cols) atomicAdd(a.M + r * a.ldm + c, v);
        }
      }
    if (sum_rows) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t s = rs[j];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (q8 == 0 && rvalid[j] && s) atomicAdd(a.rowsum + (uint64_t)rt * kBM + row_l[j], s);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 2b. the same product, software-pipelined by hand (the default)
// ---------------------------------------------------------------------------------------------------------------
// What the kernel above leaves on the table (measured at 2^20 keys: 16.9 ms against 8.4 ms for its MFMA + LDS-read loop alone): hipcc puts
// `s_waitcnt vmcnt(0)` in front of the first LDS read that follows a global_load_lds (the DMA may alias it) and in front of every
// __syncthreads(), so each k-step waits for the loads it has just issued, and the A conversion (VALU + LDS writes) runs after the MFMAs
// instead of beside them.  Here every vector-memory operation of the loop is inline asm, which the compiler does not track, and the
// waits are counted by hand:
//   * A tiles are requested TWO k-steps ahead into two register sets (a quad of a set is re-requested as soon as it has been converted),
//     the D planes two k-steps ahead into a ring of three LDS stages; a wave counts the vector-memory operations it has issued, so
//     "wait until operation X has completed" is `s_waitcnt vmcnt(n)` with n = operations issued after X;
//   * inside a k-step every wave interleaves its 56 MFMAs of k-step t (four groups of 14, one per 16-row tile) with the conversion of
//     its four quads of A(t + 1) (one quad behind each group): the two waves of a SIMD then always have both matrix and vector work to
//     issue, and the conversion hides in the shadow of the MFMAs (a 16-pass MFMA leaves the SIMD's vector issue free half the time);
//   * ONE raw s_barrier per k-step: LDS writes are drained (lgkmcnt(0)) before it, and a wave has waited for its own DMA pieces of
//     D(t + 1) before it.  A(t + 1) goes to the A buffer (t + 1) & 1, last read during k-step t - 1; D(t + 2) to stage (t + 2) % 3, ditto.
constexpr uint32_t kStagesB = 3;
constexpr uint32_t kPipePieces = 2 * kPiecesA + kStagesB * kPiecesB;  // 112 KiB

// Wait until at most `n` of this wave's vector-memory operations are outstanding.  In steady state n is always the same small number
// (8 in front of the conversions, 10 behind the MFMAs); anything else (the first and last k-steps of a unit) simply drains.
template <int STEADY>
__device__ __forceinline__ void wait_vm_at_most(uint32_t n) {
  if (n >= STEADY) {
    if constexpr (STEADY == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (STEADY == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// a pointer the compiler must keep in scalar registers (it is wave-uniform by construction; readfirstlane makes that provable)
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct PipeLane {  // per-lane constants of the A staging roles (as in the kernel above: wave-instruction x = 8*j + wave, j < 4)
  uint32_t row_l[4], kq[4], wdw[4];
  uint32_t q8;
};

struct PipeUnit {  // per-unit state of one wave
  const uint32_t* a_base;   // first row of the block's row tile (wave-uniform)
  const uint4* b_base;      // this wave's column tile of 16 in the planes, k-step 0 (wave-uniform); planar: its low-byte pieces in the image
  const uint4* h_base;      // planar: its pieces in the high-byte plane
  uint32_t a_off[4];        // byte offset of this lane's row inside the row tile (rows past the end re-read the last row)
  bool rvalid[4];
  uint32_t rs[4];           // row sums of A (only kept by blocks with column tile 0)
  bool sum_rows;
  uint32_t issued;          // vector-memory operations this wave has issued in this unit
};

// request quad j of A(ks) (16 bytes of this lane's row) into `reg`
__device__ __forceinline__ void pipe_load_quad(const MfmaArgs& a, const PipeLane& pl, PipeUnit& u, v4i& reg, int j, uint32_t ks) {
  // inner % 4 == 0 (host): a quad lies wholly inside or wholly outside the row; outside (only possible in the last k-step of the K axis) it
  // meets zero plane bytes, so any readable address does: the start of the row
  const uint32_t* const base = uniform_ptr(u.a_base);
  uint32_t off = u.a_off[j] + ks * (kBK * 4u) + pl.kq[j] * 4u;
  if ((uint64_t)(ks + 1) * kBK > a.inner && (uint64_t)ks * kBK + pl.kq[j] >= a.inner) off = u.a_off[j];
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(reg) : "v"(off), "s"(base) : "memory");
  u.issued += 1;
}

// kinds of right-hand side of the pipelined kernel
constexpr int kRhsPlanes = 0;    // byte planes of rhs_split_kernel
constexpr int kRhsImage = 1;     // planar respond image (low-byte pieces) + the high-byte plane of the pack pass
constexpr int kRhsImageBit = 2;  // planar respond image with ONE bit plane (b = 9): the high-byte pieces are expanded from it in registers

// request this wave's two 1 KiB pieces of D(ks) into stage `st` of the LDS ring (LDS-DMA: the destination is M0 + lane * 16); returns the
// value of `issued` after the request.  kRhsImageBit: only the low-byte piece; the high-byte piece is expanded from the bit plane
// (pipe_load_bits / pipe_store_hi_from_bit).
template <int RHS>
__device__ __forceinline__ uint32_t pipe_dma_b(const MfmaArgs& a, PipeUnit& u, uint32_t lds_b0, uint32_t lane, uint32_t wave, uint32_t ks, uint32_t st) {
  constexpr bool planar = RHS != kRhsPlanes;
  constexpr int kPieces = RHS == kRhsImageBit ? 1 : 2;
#pragma unroll
  for (int e = 0; e < kPieces; e++) {
    const uint4* const base = uniform_ptr(planar && e ? u.h_base : u.b_base);
    const uint32_t piece = planar ? (e ? ks * 64u : (ks >> 3) * a.lo_st16 + (ks & 7u) * 64u) : (ks * 2u + (uint32_t)e) * 64u;  // in uint4
    const uint32_t off = piece * 16u + lane * 16u;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_b0 + ((st * kPiecesB + wave * 2 + (uint32_t)e) * 1024u));
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(dst), "s"(base)
                 : "memory");
  }
  u.issued += kPieces;
  return u.issued;
}

// kRhsImageBit: request this lane's dword of the bit plane of k-step ks's super-tile into `h`: k-blocks 2w and 2w + 1 of a super-tile share
// dword w of the lane's 16 bytes of the plane (which follows the 8 KiB of low-byte pieces)
__device__ __forceinline__ void pipe_load_bits(const MfmaArgs& a, PipeUnit& u, uint32_t lane, uint32_t ks, uint32_t& h) {
  const uint32_t* const base = reinterpret_cast<const uint32_t*>(uniform_ptr(u.b_base));
  const uint32_t off = ((ks >> 3) * a.lo_st16 + 512u) * 16u + ((ks & 7u) >> 1) * 4u + lane * 16u;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(h) : "v"(off), "s"(base) : "memory");
  u.issued += 1;
}

// kRhsImageBit: this wave's high-byte piece of k-step `ks` (k-block ks & 7 of its super-tile) from the bit-plane dword `h` into stage `st`:
// bit 8*jj + 4*(kb & 1) + d of the dword is the high bit of slot 16*g + 4*d + jj of the k-block (pack.hip), i.e. byte jj of dword d of the
// piece; the operand byte is that bit XOR 0x80.  The caller has waited for the load of `h` and runs lds_barrier() behind this.
__device__ __forceinline__ void pipe_store_hi_from_bit(uint4* lds, uint32_t lane, uint32_t wave, uint32_t st, uint32_t sh /* 4 * (ks & 1) */, uint32_t h) {
  uint4 v;
  v.x = ((h >> (sh + 0u)) & 0x01010101u) | 0x80808080u;
  v.y = ((h >> (sh + 1u)) & 0x01010101u) | 0x80808080u;
  v.z = ((h >> (sh + 2u)) & 0x01010101u) | 0x80808080u;
  v.w = ((h >> (sh + 3u)) & 0x01010101u) | 0x80808080u;
  lds[(2 * kPiecesA + st * kPiecesB + wave * 2 + 1) * 64 + lane] = v;
}

// the four byte limbs of four consecutive words: limb i = bytes i of (x, y, z, w), eight v_perm_b32 (a 4 x 4 byte transpose in two stages)
__device__ __forceinline__ void byte_limbs(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t (&limb)[4]) {
  const uint32_t a = __builtin_amdgcn_perm(y, x, 0x05010400u);  // x0 y0 x1 y1
  const uint32_t b = __builtin_amdgcn_perm(y, x, 0x07030602u);  // x2 y2 x3 y3
  const uint32_t c = __builtin_amdgcn_perm(w, z, 0x05010400u);  // z0 w0 z1 w1
  const uint32_t d = __builtin_amdgcn_perm(w, z, 0x07030602u);  // z2 w2 z3 w3
  limb[0] = __builtin_amdgcn_perm(c, a, 0x05040100u);
  limb[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
  limb[2] = __builtin_amdgcn_perm(d, b, 0x05040100u);
  limb[3] = __builtin_amdgcn_perm(d, b, 0x07060302u);
}

// split quad j of A (k-step ks_abs = k0 + k) into its four byte limbs and write them into the A buffer k & 1
__device__ __forceinline__ void pipe_store_quad(const MfmaArgs& a, const PipeLane& pl, PipeUnit& u, uint4* lds, const v4i& reg, int j, uint32_t k,
                                                uint32_t ks_abs) {
  uint32_t* base = reinterpret_cast<uint32_t*>(lds + (k & 1) * kPiecesA * 64);
  const bool k_in = (uint64_t)ks_abs * kBK + pl.kq[j] < a.inner;  // past the end of the rows: all limbs zero, whatever was read
  const uint32_t x = k_in ? (uint32_t)reg[0] : 0x80808080u, y = k_in ? (uint32_t)reg[1] : 0x80808080u;
  const uint32_t z = k_in ? (uint32_t)reg[2] : 0x80808080u, w = k_in ? (uint32_t)reg[3] : 0x80808080u;
  if (u.sum_rows) {
    const bool in = u.rvalid[j] && k_in;
    u.rs[j] += in ? x + y + z + w : 0u;
  }
  uint32_t limb[4];
  byte_limbs(x ^ 0x80808080u, y ^ 0x80808080u, z ^ 0x80808080u, w ^ 0x80808080u, limb);
#pragma unroll
  for (int i = 0; i < 4; i++) base[pl.wdw[j] + i * 256] = limb[i];
}

// One MFMA and two micro-operations of the conversion of a quad, alternating, the order pinned by sched_barriers: a 16-pass MFMA keeps the
// SIMD's vector issue busy for only half of its cycles, so the conversion hides behind the MFMAs of the SAME wave (the two waves of a
// SIMD run in lockstep between barriers: left to itself the compiler issues the 14 MFMAs of a group back to back and both waves then
// convert at the same time, with the matrix pipe idle).  LDS reads may still move across the pins (the next group's A fragments).
// (x, y, z, w) = the quad; wbase = where limb 0 of this lane's dword goes; limbs i are 256 dwords apart.
// A wave's 128 accumulator registers: [row tile of 16][column tile of 16][shift s] v4i for v_mfma_i32_16x16x64_i8.  EMU32 (a -DCPIR_DIAG build
// only, hints WRONG): the same 128 registers as [pair of row tiles][shift s] v16i, every pair of 16x16x64 MFMAs replaced by ONE
// v_mfma_i32_32x32x32_i8 -- the same byte products, operand fetches and register count per k-step, half as many matrix instructions of twice
// the length: what the instruction SHAPE alone changes (cpir_tuning_set("matmul.ablate", 16), scripts/setup_kernels_timing.py).
typedef int v16i __attribute__((ext_vector_type(16)));
template <bool EMU32>
struct MfmaAcc {
  v4i a[4][2][4];
};
template <>
struct MfmaAcc<true> {
  v16i a[2][4];
};

template <bool SUM, bool EMU32>
__device__ __forceinline__ void mfma_group_with_conversion(MfmaAcc<EMU32>& accs, const int m, const v4i (&af)[4], const v4i (&bf)[2][2], uint32_t x,
                                                           uint32_t y, uint32_t z, uint32_t w, uint32_t* wbase, bool store, bool rin, uint32_t& rs) {
  uint32_t pa = 0, pb = 0, pc = 0, pd = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0, r1 = 0, r2 = 0;
  auto micro = [&](int k) {
    switch (k) {
      case 0: if constexpr (SUM) r1 = x + y; break;
      case 1: if constexpr (SUM) r2 = z + w; break;
      case 2: x ^= 0x80808080u; break;
      case 3: y ^= 0x80808080u; break;
      case 4: z ^= 0x80808080u; break;
      case 5: w ^= 0x80808080u; break;
      case 6: pa = __builtin_amdgcn_perm(y, x, 0x05010400u); break;  // x0 y0 x1 y1
      case 7: pb = __builtin_amdgcn_perm(y, x, 0x07030602u); break;  // x2 y2 x3 y3
      case 8: pc = __builtin_amdgcn_perm(w, z, 0x05010400u); break;
      case 9: pd = __builtin_amdgcn_perm(w, z, 0x07030602u); break;
      case 10: l0 = __builtin_amdgcn_perm(pc, pa, 0x05040100u); break;
      case 11: l1 = __builtin_amdgcn_perm(pc, pa, 0x07060302u); break;
      case 12: l2 = __builtin_amdgcn_perm(pd, pb, 0x05040100u); break;
      case 13: l3 = __builtin_amdgcn_perm(pd, pb, 0x07060302u); break;
      case 14: if constexpr (SUM) r1 += r2; break;
      case 15: if constexpr (SUM) rs += rin ? r1 : 0u; break;
      case 16:
        if (store) wbase[0] = l0, wbase[256] = l1;
        break;
      case 17:
        if (store) wbase[512] = l2, wbase[768] = l3;
        break;
      default: break;
    }
  };
  if constexpr (!EMU32) {
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int q = 0; q < 7; q++) {
        const int sidx = q < 4 ? q : q - 3, ai = q < 4 ? q : q - 4, bj = q < 4 ? 0 : 1, step = n * 7 + q;
        accs.a[m][n][sidx] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[ai], bf[n][bj], accs.a[m][n][sidx], 0, 0, 0);
        if (step < 9) {
          __builtin_amdgcn_sched_barrier(0x100);
          micro(2 * step);
          micro(2 * step + 1);
          __builtin_amdgcn_sched_barrier(0x100);
        }
      }
  } else {
#pragma unroll
    for (int q = 0; q < 7; q++) {  // seven 32-pass MFMAs, the 18 micro-operations spread 3, 3, 3, 3, 2, 2, 2 behind them
      const int sidx = q < 4 ? q : q - 3, ai = q < 4 ? q : q - 4, bj = q < 4 ? 0 : 1;
      accs.a[m >> 1][sidx] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ai], bf[m & 1][bj], accs.a[m >> 1][sidx], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0x100);
      const int first = q < 4 ? 3 * q : 12 + 2 * (q - 4), count = q < 4 ? 3 : 2;
#pragma unroll
      for (int t = 0; t < count; t++) micro(first + t);
      __builtin_amdgcn_sched_barrier(0x100);
    }
  }
}

// One (row tile, column tile, K sub-range) unit for one wave.  Register set S0 carries the even k-steps (relative to k0), S1 the odd ones.
template <bool SUM, int RHS, bool EMU32>
__device__ __forceinline__ void mfma_pipe_unit(const MfmaArgs& a, const PipeLane& pl, PipeUnit& u, uint4* lds, uint32_t lds_b0, uint32_t lane,
                                               uint32_t wave, uint32_t k0, uint32_t T, MfmaAcc<EMU32>& acc) {
  constexpr bool PLANAR = RHS != kRhsPlanes;
  constexpr bool BIT = RHS == kRhsImageBit;
  const uint32_t wm = wave >> 2, wn = wave & 3;
  const uint32_t alane = a_slot(lane);  // this lane's slot inside an A piece (see a_slot)
  v4i S0[4], S1[4];
  // kRhsImageBit: ONE register carries the bit-plane dword of the next k-step.  At the end of k-step t (behind the wait that covers its
  // load, in front of the barrier) the piece of k-step t + 1 is expanded from it into stage (t + 1) % 3, and the dword of k-step t + 2 is
  // requested into the same register -- the LAST request of the k-step, so the steady-state waits keep their shape (see `steady`).
  uint32_t H = 0;
  const uint32_t sh_even = 4u * (k0 & 1u), sh_odd = 4u * ((k0 + 1u) & 1u);  // 4 * (k-block & 1) of the even / odd k-steps (relative to k0)
  // ---- prologue: the first two k-steps are requested, A(0) is converted, A(2) requested behind it ----
#pragma unroll
  for (int j = 0; j < 4; j++) pipe_load_quad(a, pl, u, S0[j], j, k0);
  const uint32_t bm0 = pipe_dma_b<RHS>(a, u, lds_b0, lane, wave, k0, 0);
  if constexpr (BIT) pipe_load_bits(a, u, lane, k0, H);
  uint32_t bm_next = 0;  // `issued` after the request for the D stage of the NEXT k-step
  if (T > 1) {
#pragma unroll
    for (int j = 0; j < 4; j++) pipe_load_quad(a, pl, u, S1[j], j, k0 + 1);
    bm_next = pipe_dma_b<RHS>(a, u, lds_b0, lane, wave, k0 + 1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(S0[0]), "+v"(S0[1]), "+v"(S0[2]), "+v"(S0[3]), "+v"(H)::"memory");
  (void)bm0;
  if constexpr (BIT) pipe_store_hi_from_bit(lds, lane, wave, 0, sh_even, H);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    pipe_store_quad(a, pl, u, lds, S0[j], j, 0, k0);
    if (T > 2) pipe_load_quad(a, pl, u, S0[j], j, k0 + 2);
  }
  if constexpr (BIT) {  // (behind the quads, as in every k-step: the steady-state waits count on that order)
    if (T > 1) pipe_load_bits(a, u, lane, k0 + 1, H);
  }
  uint32_t set_mark = u.issued;  // `issued` after the last request into the set that is converted NEXT ... (tracked per parity below)
  uint32_t mark0 = u.issued, mark1 = (T > 1) ? bm_next - 2 : 0;  // after the loads of S0 (A(2)) / S1 (A(1))
  (void)set_mark;
  lds_barrier();

  // k-step t: MFMAs of t from A buffer t & 1 and D stage t % 3, with the conversion of A(t + 1) (register set of parity (t + 1) & 1, re-requested
  // quad by quad with A(t + 3)) interleaved behind the four MFMA groups
  auto step = [&](uint32_t t, v4i(&set)[4], uint32_t& mark, uint32_t sh_next) {
    uint32_t bm_new = 0;
    if (t + 2 < T) bm_new = pipe_dma_b<RHS>(a, u, lds_b0, lane, wave, k0 + t + 2, (t + 2) % kStagesB);
    const bool conv = t + 1 < T;
    if (conv) {
      wait_vm_at_most<8>(u.issued - mark);
      asm volatile("" : "+v"(set[0]), "+v"(set[1]), "+v"(set[2]), "+v"(set[3])::"memory");
      if ((uint64_t)(k0 + t + 2) * kBK > a.inner) {  // A(t + 1) reaches past the end of the rows: those quads become all-zero limbs
#pragma unroll
        for (int m = 0; m < 4; m++)
          if ((uint64_t)(k0 + t + 1) * kBK + pl.kq[m] >= a.inner) set[m] = v4i{(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
      }
    }
    const uint4* A_ = lds + (t & 1) * kPiecesA * 64;
    const uint4* B_ = lds + (2 * kPiecesA + (t % kStagesB) * kPiecesB) * 64;
    v4i bf[2][2];
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const uint4 q = B_[((wn * 2 + n) * 2 + j) * 64 + lane];
        bf[n][j] = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
      }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      v4i af[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint4 q = A_[((wm * 4 + m) * 4 + i) * 64 + alane];
        af[i] = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
      }
      uint32_t* const wbase = reinterpret_cast<uint32_t*>(lds + ((t + 1) & 1) * kPiecesA * 64) + pl.wdw[m];
      const bool rin = conv && u.rvalid[m] && (uint64_t)(k0 + t + 1) * kBK + pl.kq[m] < a.inner;
      mfma_group_with_conversion<SUM, EMU32>(acc, m, af, bf, (uint32_t)set[m][0], (uint32_t)set[m][1], (uint32_t)set[m][2], (uint32_t)set[m][3], wbase,
                                      conv, rin, u.rs[m]);
      if (conv && t + 3 < T) pipe_load_quad(a, pl, u, set[m], m, k0 + t + 3);
    }
    if (conv) mark = u.issued;
    if (t + 1 < T) {
      if constexpr (BIT) {
        // (these few k-steps at the edges of a unit simply drain: the dword of k-step t + 1 and its low-byte piece have landed)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(H)::"memory");
        pipe_store_hi_from_bit(lds, lane, wave, (t + 1) % kStagesB, sh_next, H);
        if (t + 2 < T) pipe_load_bits(a, u, lane, k0 + t + 2, H);
      } else {
        wait_vm_at_most<10>(u.issued - bm_next);  // this wave's pieces of D(t + 1) have landed
      }
    }
    bm_next = bm_new;
    lds_barrier();
  };
  // ---- steady state: k-steps t with t + 3 inside the unit and no ragged K tail in reach; nothing conditional, every address advanced
  // by adding a constant to a scalar base, the waits are the literal steady-state counts (see wait_vm_at_most) ----
  const bool ragged_tail = (uint64_t)(k0 + T) * kBK > a.inner;   // the last k-step of this unit reaches past the end of the rows of A
  const uint32_t T_safe = ragged_tail ? T - 1 : T;
  const uint32_t Ts = T_safe > 3 ? ((T_safe - 3) & ~1u) : 0;      // steady steps (an even number: two per loop trip)
  if (Ts) {
    const uint32_t* a_run = uniform_ptr(u.a_base + (uint64_t)(k0 + 3) * kBK);      // A(t + 3), advanced by one k-step per step
    // D(t + 2): the two pieces of that k-step.  Byte planes: 2 KiB per k-step, side by side.  Planar image: the low-byte piece advances
    // 1 KiB per k-step inside a super-tile and skips the bit planes when it crosses into the next one; the high-byte piece 1 KiB per k-step.
    // (planar: both addresses are recomputed from the k-step every time -- a handful of scalar operations -- rather than carried along:
    // the kernel has no scalar registers to spare)
    uint32_t ks_dma = k0 + 2;
    const uint4* b_run = PLANAR ? nullptr : uniform_ptr(u.b_base + (uint64_t)ks_dma * 128);
    const uint32_t lo_st16 = PLANAR ? (uint32_t)__builtin_amdgcn_readfirstlane(a.lo_st16) : 0u;
    uint32_t aoffq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) aoffq[j] = u.a_off[j] + pl.kq[j] * 4u;
    const uint32_t dma_voff = lane * 16u;
    const uint32_t dma_dst0 = lds_b0 + wave * 2048u;  // this wave's two pieces inside a stage
    uint32_t st_read = 0, st_dma = 2;                // stage of k-step t, stage that receives D(t + 2)
    auto steady = [&](uint32_t t, v4i(&set)[4], const uint32_t a_par /* (t + 1) & 1 */, const uint32_t r_par /* t & 1 */, const uint32_t sh_next) {
      // the first fragments are requested from LDS before anything else, so that their latency runs while the DMA is being issued
      const uint4* A_ = lds + r_par * kPiecesA * 64;
      const uint4* B_ = lds + (2 * kPiecesA + st_read * kPiecesB) * 64;
      v4i bf[2][2], af0[4];
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const uint4 q = B_[((wn * 2 + n) * 2 + j) * 64 + lane];
          bf[n][j] = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
        }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint4 q = A_[((wm * 4 + 0) * 4 + i) * 64 + alane];
        af0[i] = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
      }
      {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(dma_dst0 + st_dma * (kPiecesB * 1024u));
        uint32_t keep;
        // (the second piece lands 1 KiB behind the first: M0 + 1024, and the immediate offset moves source and destination alike, so its
        // base is passed 1 KiB early)
        if constexpr (BIT) {
          // the low-byte piece by LDS-DMA; the bit-plane dword of the same k-step is requested at the END of this k-step (below)
          const uint4* const lo = uniform_ptr(u.b_base + ((ks_dma >> 3) * lo_st16 + (ks_dma & 7u) * 64u));
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep)
                       : "v"(dma_voff), "s"(dst), "s"(lo)
                       : "memory");
        } else if constexpr (PLANAR) {
          const uint4* const lo = uniform_ptr(u.b_base + ((ks_dma >> 3) * lo_st16 + (ks_dma & 7u) * 64u));
          const uint4* const hi_early = uniform_ptr(u.h_base + ks_dma * 64u - 64);
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep)
                       : "v"(dma_voff), "s"(dst), "s"(lo), "s"(hi_early)
                       : "memory");
          ks_dma++;
        } else {
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep)
                       : "v"(dma_voff), "s"(dst), "s"(b_run)
                       : "memory");
          b_run += 128;
        }
      }
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(set[0]), "+v"(set[1]), "+v"(set[2]), "+v"(set[3])::"memory");
#pragma unroll
      for (int m = 0; m < 4; m++) {
        v4i af[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (m == 0) {
            af[i] = af0[i];
          } else {
            const uint4 q = A_[((wm * 4 + m) * 4 + i) * 64 + alane];
            af[i] = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
          }
        }
        uint32_t* const wbase = reinterpret_cast<uint32_t*>(lds + a_par * kPiecesA * 64) + pl.wdw[m];
        mfma_group_with_conversion<SUM, EMU32>(acc, m, af, bf, (uint32_t)set[m][0], (uint32_t)set[m][1], (uint32_t)set[m][2], (uint32_t)set[m][3],
                                        wbase, true, u.rvalid[m], u.rs[m]);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(set[m]) : "v"(aoffq[m]), "s"(a_run) : "memory");
      }
      a_run += kBK;
      if constexpr (BIT) {
        // Requests of a k-step s, in order: low-byte DMA of D(s + 2), four quads of A(s + 3), bit dword of D(s + 2).  Outstanding here, newest
        // first: quads of this k-step (4), DMA of D(t + 2) -- five that may stay in flight; behind them the bit dword of D(t + 1), the quads of
        // the previous k-step and the DMA of D(t + 1), which must all have landed (the steady wait of the byte-plane kernels, vmcnt(10),
        // covers the same requests: theirs come in pairs at the head of a k-step).
        asm volatile("s_waitcnt vmcnt(5)" : "+v"(H)::"memory");
        pipe_store_hi_from_bit(lds, lane, wave, st_read == 2 ? 0 : st_read + 1, sh_next, H);
        const uint32_t* const bits =
            uniform_ptr(reinterpret_cast<const uint32_t*>(u.b_base + ((ks_dma >> 3) * lo_st16 + 512u)) + ((ks_dma & 7u) >> 1));
        asm volatile("global_load_dword %0, %1, %2" : "=v"(H) : "v"(dma_voff), "s"(bits) : "memory");
        ks_dma++;
      } else {
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      }
      lds_barrier();
      st_read = st_read == 2 ? 0 : st_read + 1;
      st_dma = st_dma == 2 ? 0 : st_dma + 1;
      (void)t;
    };
    for (uint32_t t = 0; t < Ts; t += 2) {
      steady(t, S1, 1, 0, sh_odd);
      steady(t + 1, S0, 0, 1, sh_even);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain once; the remaining steps run on the generic path with exact bookkeeping
    u.issued = 0, mark0 = 0, mark1 = 0, bm_next = 0;
  }
  for (uint32_t t = Ts; t < T; t += 2) {  // (unrolled by two so that the register set of each step is a compile-time choice; Ts is even)
    step(t, S1, mark1, sh_odd);
    if (t + 1 < T) step(t + 1, S0, mark0, sh_even);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifdef CPIR_DIAG
template <int RHS, bool EMU32 = false>
#else
template <int RHS>  // (the release kernel has no such parameter: its name in a trace is mat_x_mat_mfma_pipe_kernel<RHS>)
#endif
__global__ void __launch_bounds__(kMT) __attribute__((amdgpu_waves_per_eu(2, 2))) mat_x_mat_mfma_pipe_kernel(const MfmaArgs a) {
#ifndef CPIR_DIAG
  constexpr bool EMU32 = false;
#endif
  constexpr bool PLANAR = RHS != kRhsPlanes;
  __shared__ uint4 lds[kPipePieces * 64];

  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wm = wave >> 2, wn = wave & 3;
  const uint32_t xcd = blockIdx.x % a.nx, slot = blockIdx.x / a.nx, slots = gridDim.x / a.nx;  // host: gridDim.x % nx == 0
  const uint32_t ksx0 = (uint32_t)(((uint64_t)a.KS * xcd) / a.nx), ksx1 = (uint32_t)(((uint64_t)a.KS * (xcd + 1)) / a.nx);
  const uint32_t pairs = a.RT * a.CT;
  const uint64_t units = (uint64_t)pairs * a.S;
  const uint32_t lds_b0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&lds[2 * kPiecesA * 64]);

  PipeLane pl;
  const uint32_t r8 = lane >> 3;
  pl.q8 = lane & 7;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t x = 8 * j + wave, h = x & 1, rb = x >> 1;
    pl.row_l[j] = 8 * rb + r8;
    const uint32_t q = 8 * h + pl.q8;
    pl.kq[j] = 4 * q;
    pl.wdw[j] = (((pl.row_l[j] >> 4) * 4) * 64 + a_slot(16 * (q >> 2) + (pl.row_l[j] & 15))) * 4 + (q & 3);
  }

  for (uint64_t un = slot; un < units; un += slots) {
    const uint32_t sub = (uint32_t)(un / pairs), pair = (uint32_t)(un % pairs), rt = pair / a.CT, ct = pair % a.CT;
    const uint32_t k0 = ksx0 + (uint32_t)(((uint64_t)(ksx1 - ksx0) * sub) / a.S);
    const uint32_t k1 = ksx0 + (uint32_t)(((uint64_t)(ksx1 - ksx0) * (sub + 1)) / a.S);
    if (k0 >= k1) continue;  // block-uniform

    PipeUnit u;
    u.a_base = uniform_ptr(a.A + (uint64_t)rt * kBM * a.lda);
    if constexpr (PLANAR) {  // (column tiles past the image's last one are computed from its last tile and never stored)
      const uint32_t T16 = ct * 8 + wave < a.b_col_tiles ? ct * 8 + wave : a.b_col_tiles - 1;
      u.b_base = uniform_ptr(a.lo_tiles + (uint64_t)T16 * a.lo_ks512 * a.lo_st16);
      u.h_base = RHS == kRhsImage ? uniform_ptr(a.hi_plane + (uint64_t)T16 * a.kb_total * 64) : u.b_base;
    } else {
      u.b_base = uniform_ptr(a.planes + (uint64_t)(ct * 8 + wave) * a.KS * 128);  // 2 pieces of 64 uint4 per k-step
      u.h_base = u.b_base;
    }
    u.sum_rows = (ct == 0);
    u.issued = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint64_t r = (uint64_t)rt * kBM + pl.row_l[j];
      u.rvalid[j] = r < a.rows;
      const uint64_t rr = u.rvalid[j] ? pl.row_l[j] : (a.rows - 1 - (uint64_t)rt * kBM);  // rows past the end re-read the last row
      u.a_off[j] = (uint32_t)(rr * a.lda * 4);                                            // host: (kBM - 1) * lda * 4 + inner * 4 < 2^32
      u.rs[j] = 0;
    }

    MfmaAcc<EMU32> acc;
    if constexpr (!EMU32) {
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
          for (int s2 = 0; s2 < 4; s2++) acc.a[m][n][s2] = v4i{0, 0, 0, 0};
    }
#ifdef CPIR_DIAG
    else {
#pragma unroll
      for (int mp = 0; mp < 2; mp++)
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) acc.a[mp][s2] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    }
#endif

    if (u.sum_rows) mfma_pipe_unit<true, RHS, EMU32>(a, pl, u, lds, lds_b0, lane, wave, k0, k1 - k0, acc);
    else mfma_pipe_unit<false, RHS, EMU32>(a, pl, u, lds, lds_b0, lane, wave, k0, k1 - k0, acc);

    // ---- this unit's part of the output tile: sum_s acc_s << 8s, one u32 atomic per element ----
    const uint32_t fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int n = 0; n < 2; n++) {
        const uint64_t c = (uint64_t)ct * kBN + wn * 32 + n * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint64_t r = (uint64_t)rt * kBM + wm * 64 + m * 16 + fq * 4 + i;
          uint32_t v = 0;
          if constexpr (!EMU32) {
            v = (uint32_t)acc.a[m][n][0][i] + ((uint32_t)acc.a[m][n][1][i] << 8) + ((uint32_t)acc.a[m][n][2][i] << 16) + ((uint32_t)acc.a[m][n][3][i] << 24);
          }
#ifdef CPIR_DIAG
          else {  // (the emulation's values mean nothing: every register still leaves through one atomic)
            const int e = (m & 1) * 8 + n * 4 + i;
            v = (uint32_t)acc.a[m >> 1][0][e] + ((uint32_t)acc.a[m >> 1][1][e] << 8) + ((uint32_t)acc.a[m >> 1][2][e] << 16) + ((uint32_t)acc.a[m >> 1][3][e] << 24);
          }
#endif
          if (r < a.rows && c < a.cols) atomicAdd(a.M + r * a.ldm + c, v);
        }
      }
    if (u.sum_rows) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t s2 = u.rs[j];
        s2 += __shfl_xor(s2, 1, 64);
        s2 += __shfl_xor(s2, 2, 64);
        s2 += __shfl_xor(s2, 4, 64);
        if (pl.q8 == 0 && u.rvalid[j] && s2) atomicAdd(a.rowsum + (uint64_t)rt * kBM + pl.row_l[j], s2);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 3. correction terms
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) hint_fixup_kernel(uint32_t* __restrict__ M, uint64_t ldm, uint64_t rows, uint64_t cols,
                                                               const uint32_t* __restrict__ rowsum, const uint32_t* __restrict__ colsum,
                                                               uint32_t k_term) {
  const uint64_t total = rows * cols;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kThreads) {
    const uint64_t r = i / cols, c = i % cols;
    M[r * ldm + c] += 0x8080u * rowsum[r] + 0x80808080u * colsum[c] - k_term;  // after the product kernel on the same stream
  }
}

uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

}  // namespace

// ---- workspace: [planes][colsum][rowsum] ------------------------------------------------------------------------------------------
static uint64_t rhs_planes_bytes(uint64_t inner, uint64_t cols) {
  const uint64_t ct16 = (cols + kBN - 1) / kBN * 8, ks = (inner + kBK - 1) / kBK;
  return ct16 * ks * 2048;
}
static uint64_t rhs_colsum_words(uint64_t cols) { return (cols + kBN - 1) / kBN * kBN; }

uint64_t mfma_rhs_workspace_bytes(uint64_t inner, uint64_t cols, uint64_t max_rows) {
  return rhs_planes_bytes(inner, cols) + 4 * rhs_colsum_words(cols) + 4 * ((max_rows + kBM - 1) / kBM * kBM);
}

bool mfma_matmul_applicable(const uint32_t* A, uint64_t lda, uint64_t inner, uint64_t cols, uint32_t rhs_max_bits) {
  if (rhs_max_bits > 16 || inner % 4 != 0 || lda % 4 != 0 || reinterpret_cast<uintptr_t>(A) % 16 != 0) return false;
  const uint64_t ks = (inner + kBK - 1) / kBK, ct = (cols + kBN - 1) / kBN;
  return ks <= 0x7fffffffull && ct * 8 <= 0xffffffu;
}

int launch_rhs_split(const Device* dev, const uint32_t* D, uint64_t ldd, uint64_t inner, uint64_t cols, void* workspace, hipStream_t stream) {
  (void)dev;
  if (!D || !workspace || inner == 0 || cols == 0 || ldd < cols) return CPIR_ERR_INVALID_ARGUMENT;
  SplitArgs sa;
  sa.D = D, sa.ld = ldd, sa.inner = inner, sa.cols = (uint32_t)cols;
  sa.ct16 = (uint32_t)((cols + kBN - 1) / kBN * 8);
  sa.stripe_groups = (sa.ct16 + 15) / 16;
  sa.ks_total = (uint32_t)((inner + kBK - 1) / kBK);
  sa.planes = reinterpret_cast<uint4*>(workspace);
  sa.colsum = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(workspace) + rhs_planes_bytes(inner, cols));
  CPIR_TRY(zero_words(sa.colsum, rhs_colsum_words(cols), stream));
  const uint64_t blocks = (uint64_t)((sa.ks_total + 7) / 8) * sa.stripe_groups;
  if (blocks > 0x7fffffffull) return CPIR_ERR_INVALID_ARGUMENT;
  const bool vec = (ldd % 4 == 0) && (reinterpret_cast<uintptr_t>(D) % 16 == 0);
  if (vec) hipLaunchKernelGGL(rhs_split_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, sa);
  else hipLaunchKernelGGL(rhs_split_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, sa);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

// the product + the correction terms, the right-hand side described by a.planes or a.lo_tiles / a.hi_plane (already filled in)
static int launch_product(const Device* dev, MfmaArgs& a, const uint32_t* colsum, int accumulate, hipStream_t stream);

int launch_mat_x_mat_mfma(const Device* dev, const uint32_t* A, uint64_t lda, const void* workspace, uint64_t inner, uint64_t cols,
                          uint32_t* M, uint64_t ldm, uint64_t rows, uint64_t ws_max_rows, int accumulate, hipStream_t stream) {
  if (!A || !workspace || !M || rows == 0 || rows > ws_max_rows || lda < inner || ldm < cols) return CPIR_ERR_INVALID_ARGUMENT;
  const uint8_t* ws = reinterpret_cast<const uint8_t*>(workspace);
  MfmaArgs a;
  a.A = A, a.lda = lda, a.M = M, a.ldm = ldm, a.rows = rows, a.inner = inner, a.cols = cols;
  a.planes = reinterpret_cast<const uint4*>(ws);
  a.lo_tiles = nullptr, a.hi_plane = nullptr, a.lo_st16 = 0, a.lo_ks512 = 0, a.b_col_tiles = 0, a.kb_total = 0;
  const uint32_t* colsum = reinterpret_cast<const uint32_t*>(ws + rhs_planes_bytes(inner, cols));
  a.rowsum = const_cast<uint32_t*>(colsum) + rhs_colsum_words(cols);
  return launch_product(dev, a, colsum, accumulate, stream);
}

// the hand-pipelined kernel is on and can address a row tile of A (and the k-steps of the right-hand side) with 32-bit byte offsets:
// with lda = N that is N < 2^23 slots.  Asked BEFORE a path is chosen (setup falls back to the split right-hand side or the VALU
// matmul), and again by launch_product.
static bool mfma_pipe_addressable(uint64_t lda, uint64_t inner) {
  const uint64_t ks = (inner + kBK - 1) / kBK;
  return mfma_pipeline() != 0 && ((uint64_t)(kBM - 1) * lda + inner) * 4 < (1ull << 32) && ks * 2048 + 2048 < (1ull << 32);
}

bool mfma_planar_rhs_applicable(const uint32_t* A, uint64_t lda, const cpir_dtc_layout& L) {
  // the image must hold at least one bit plane (b >= 9; below that the high-byte plane does not exist), the hand-pipelined kernel must
  // be on and able to address everything with 32-bit byte offsets
  const uint64_t ks512 = (L.num_slots + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE;
  return L.packing == CPIR_PACK_PLANAR && L.mat_elem_bit_len >= 9 && mfma_pipe_addressable(lda, L.num_slots) &&
         mfma_matmul_applicable(A, lda, L.num_slots, L.num_cols, 16) && ks512 * (L.chunk_words / 4) * 16 + (1u << 20) < (1ull << 32);
}

int launch_mat_x_mat_mfma_planar(const Device* dev, const uint32_t* A, uint64_t lda, const uint32_t* dtc, const cpir_dtc_layout& L,
                                 const void* hi_plane, uint32_t* rowsum_ws, uint32_t* M, uint64_t ldm, uint64_t rows, int accumulate,
                                 hipStream_t stream) {
  if (!A || !dtc || !rowsum_ws || !M || rows == 0 || lda < L.num_slots || ldm < L.num_cols) return CPIR_ERR_INVALID_ARGUMENT;
  if (!mfma_planar_rhs_applicable(A, lda, L)) return CPIR_ERR_INVALID_ARGUMENT;
  // one bit plane (b = 9): the high-byte pieces come out of the image itself and no plane is taken; more planes: the pack pass's plane
  if ((planar_hi_plane_bytes(L) != 0) != (hi_plane != nullptr)) return CPIR_ERR_INVALID_ARGUMENT;
  MfmaArgs a;
  a.A = A, a.lda = lda, a.M = M, a.ldm = ldm, a.rows = rows, a.inner = L.num_slots, a.cols = L.num_cols;
  a.planes = nullptr;
  a.lo_tiles = reinterpret_cast<const uint4*>(dtc);
  a.hi_plane = reinterpret_cast<const uint4*>(hi_plane);
  a.lo_st16 = L.chunk_words / 4;  // chunk_words = (8 + HB) * 256 u32 per super-tile
  a.lo_ks512 = (uint32_t)((L.num_slots + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE);
  a.b_col_tiles = L.rows_padded / 16;
  a.kb_total = (uint32_t)((L.num_slots + 63) / 64);
  a.rowsum = rowsum_ws;
  const uint32_t* colsum = dtc + (uint64_t)L.rows_padded * L.words_per_row_padded;  // per-column field sums behind the tiles
  return launch_product(dev, a, colsum, accumulate, stream);
}

static int launch_product(const Device* dev, MfmaArgs& a, const uint32_t* colsum, int accumulate, hipStream_t stream) {
  const uint64_t rows = a.rows, cols = a.cols, inner = a.inner, lda = a.lda, ldm = a.ldm;
  uint32_t* const M = a.M;
  a.RT = (uint32_t)((rows + kBM - 1) / kBM), a.CT = (uint32_t)((cols + kBN - 1) / kBN), a.KS = (uint32_t)((inner + kBK - 1) / kBK);
  // persistent grid, one block per CU (96 KiB of LDS and 2 waves per SIMD each)
  uint32_t grid = (uint32_t)dev->num_cus;
  a.nx = (grid % 8 == 0 && a.KS >= 64) ? 8u : 1u;
  const uint32_t slots = grid / a.nx;
  // K sub-ranges per XCD range: as few as make the (pair, sub-range) units a whole number of rounds of the XCD's blocks
  const uint64_t pairs = (uint64_t)a.RT * a.CT;
  uint64_t g = pairs, h = slots;
  while (h) {
    const uint64_t t = g % h;
    g = h, h = t;
  }
  uint64_t S = slots / g;
  const uint64_t ks_per_x = (a.KS + a.nx - 1) / a.nx;
  while (S > 1 && ks_per_x / S < 8) S /= 2;  // a unit shorter than 8 k-steps is all prologue and flush
  a.S = (uint32_t)(S ? S : 1);
  CPIR_DIAG_ONLY(a.ablate = (uint32_t)mfma_ablate();)

  // the hand-pipelined kernel addresses a row tile of A with 32-bit byte offsets
  const bool pipe = mfma_pipe_addressable(lda, inner);
  if (!pipe && a.lo_tiles) return CPIR_ERR_INVALID_ARGUMENT;  // only the pipelined kernel reads the planar image (callers ask mfma_planar_rhs_applicable first)
  CPIR_TRY(zero_words(a.rowsum, round_up((uint32_t)rows, kBM), stream));
  if (!accumulate) CPIR_TRY(launch_zero_matrix(M, ldm, rows, cols, stream));
#ifdef CPIR_DIAG
  if (pipe && (a.ablate & 16u)) {  // the 32x32x32 emulation (see MfmaAcc)
    if (a.lo_tiles && !a.hi_plane) hipLaunchKernelGGL((mat_x_mat_mfma_pipe_kernel<kRhsImageBit, true>), dim3(grid), dim3(kMT), 0, stream, a);
    else if (a.lo_tiles) hipLaunchKernelGGL((mat_x_mat_mfma_pipe_kernel<kRhsImage, true>), dim3(grid), dim3(kMT), 0, stream, a);
    else hipLaunchKernelGGL((mat_x_mat_mfma_pipe_kernel<kRhsPlanes, true>), dim3(grid), dim3(kMT), 0, stream, a);
  } else
#endif
  if (pipe && a.lo_tiles && !a.hi_plane) hipLaunchKernelGGL(mat_x_mat_mfma_pipe_kernel<kRhsImageBit>, dim3(grid), dim3(kMT), 0, stream, a);
  else if (pipe && a.lo_tiles) hipLaunchKernelGGL(mat_x_mat_mfma_pipe_kernel<kRhsImage>, dim3(grid), dim3(kMT), 0, stream, a);
  else if (pipe) hipLaunchKernelGGL(mat_x_mat_mfma_pipe_kernel<kRhsPlanes>, dim3(grid), dim3(kMT), 0, stream, a);
  else hipLaunchKernelGGL(mat_x_mat_mfma_kernel, dim3(grid), dim3(kMT), 0, stream, a);
  const uint32_t k_term = (uint32_t)inner * (0x80808080u * 0x8080u);
  uint64_t fb = (rows * cols + kThreads - 1) / kThreads;
  if (fb > (uint64_t)dev->num_cus * 8) fb = (uint64_t)dev->num_cus * 8;
  hipLaunchKernelGGL(hint_fixup_kernel, dim3((unsigned)fb), dim3(kThreads), 0, stream, M, ldm, rows, cols, a.rowsum, colsum, k_term);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace cpir
