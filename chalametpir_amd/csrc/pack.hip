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
#include "device_bytes.hpp"

namespace cpir {

static std::atomic<int> g_pack_rows{-1};  // tuning "pack.rows": -1 by width, 0 the 64-column waves, 1 whole rows per block
int pack_rows_mode() { return g_pack_rows.load(); }
void set_pack_rows_mode(int m) { g_pack_rows.store(m); }

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

// ---- dense64 packing -----------------------------------------------------------------------------------------------------
// slot (local to this DtC) -> (u64 word of the row, bit offset) -- the private permutation documented at cpir_dtc_layout
__device__ __forceinline__ void dense_locate(uint64_t slot, uint32_t K, uint32_t B, uint64_t* word64, uint32_t* bit) {
  const uint64_t per_chunk = (uint64_t)K * 1024;
  const uint64_t chunk = slot / per_chunk;
  const uint32_t within = (uint32_t)(slot % per_chunk);
  const uint32_t j = within >> 10, p = within & 1023;
  const uint32_t m = ((p >> 1) & 1) * 512 + 2 * (p >> 2) + (p & 1);
  *word64 = chunk * 1024 + m;
  *bit = j * B;
}

// inverse: (chunk, u64 word m of the chunk, field j) -> local slot
__device__ __forceinline__ uint64_t dense_slot(uint64_t chunk, uint32_t m, uint32_t j, uint32_t K) {
  const uint32_t L = m >> 9, r = m & 511;
  const uint32_t p = 4 * (r >> 1) + 2 * L + (r & 1);
  return chunk * K * 1024 + (uint64_t)j * 1024 + p;
}

// One block = 64 positions p of one chunk (all K field planes) x 32 columns of D.
template <int B, bool VEC>
__global__ void __launch_bounds__(kThreads) transpose_compress_dense_kernel(const uint32_t* __restrict__ D, uint64_t ldd, uint64_t N,
                                                                             uint32_t C, uint32_t* __restrict__ dtc,
                                                                             uint64_t row_stride, uint32_t rows_padded,
                                                                             uint32_t* __restrict__ or_of_entries) {
  constexpr int K = 64 / B;
  constexpr int kCols = 32;
  __shared__ uint16_t tile[K * 64][kCols + 2];

  const int tid = threadIdx.x;
  const uint64_t chunk = blockIdx.x >> 4;
  const uint32_t p0 = (blockIdx.x & 15) * 64;
  const uint32_t c0 = blockIdx.y * kCols;
  constexpr uint32_t mask = (1u << B) - 1u;
  uint32_t seen = 0;

  // ---- load: 8 lanes x 4 columns cover the 32 tile columns of one D row; 32 rows of D per pass ----------------------
  const int lc = (tid & 7) * 4;
  const int lr = tid >> 3;
#pragma unroll 2
  for (int rr = lr; rr < K * 64; rr += kThreads / 8) {
    const uint32_t j = rr >> 6, pp = rr & 63;
    const uint64_t n = chunk * K * 1024 + (uint64_t)j * 1024 + p0 + pp;
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
      tile[rr][lc + i] = (uint16_t)(v[i] & mask);
    }
  }
  __syncthreads();

  // ---- store: a wave writes, per column, two 256-byte runs of u64 words (L = 0 and L = 1 halves of the chunk) --------
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t L = lane >> 5, mm = lane & 31;
  const uint32_t t = (p0 >> 2) + (mm >> 1), e = mm & 1;
  const uint32_t pp = 4 * t + 2 * L + e - p0;
  const uint32_t m = L * 512 + 2 * t + e;
  uint64_t* out64 = reinterpret_cast<uint64_t*>(dtc);
  const uint64_t stride64 = row_stride / 2;
#pragma unroll
  for (int i = 0; i < kCols / 4; i++) {
    const int cl = wave * (kCols / 4) + i;
    const uint32_t c = c0 + cl;
    if (c < rows_padded) {
      uint64_t word = 0;
#pragma unroll
      for (int j = 0; j < K; j++) word |= (uint64_t)tile[j * 64 + pp][cl] << (j * B);
      out64[(uint64_t)c * stride64 + chunk * 1024 + m] = word;
    }
  }

  if (or_of_entries) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) seen |= __shfl_xor(seen, off, 64);
    if (lane == 0 && seen) atomicOr(or_of_entries, seen);
  }
}

// reference compressed matrix (C x W) -> dense64 device layout; fields >= N and bits >= b are dropped as in normalise_word
__global__ void __launch_bounds__(kThreads) dtc_import_dense_kernel(const uint32_t* __restrict__ src, uint64_t W, uint64_t N, uint32_t C,
                                                                     uint32_t b, uint32_t cf, uint32_t K, uint64_t* __restrict__ dtc64,
                                                                     uint64_t stride64, uint32_t rows_padded) {
  const uint32_t mask = (1u << b) - 1u, S = 32 / cf;
  const uint64_t total = (uint64_t)rows_padded * stride64;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kThreads) {
    const uint64_t c = i / stride64, w64 = i % stride64;
    uint64_t word = 0;
    if (c < C) {
      for (uint32_t j = 0; j < K; j++) {
        const uint64_t slot = dense_slot(w64 >> 10, (uint32_t)(w64 & 1023), j, K);
        if (slot < N) {
          const uint32_t f = (src[c * W + slot / cf] >> ((uint32_t)(slot % cf) * S)) & mask;
          word |= (uint64_t)f << (j * b);
        }
      }
    }
    dtc64[i] = word;
  }
}

__global__ void __launch_bounds__(kThreads) dtc_export_dense_kernel(const uint64_t* __restrict__ dtc64, uint64_t stride64, uint64_t W,
                                                                     uint64_t N, uint32_t C, uint32_t b, uint32_t cf, uint32_t K,
                                                                     uint32_t* __restrict__ dst) {
  const uint32_t mask = (1u << b) - 1u, S = 32 / cf;
  const uint64_t total = (uint64_t)C * W;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kThreads) {
    const uint64_t c = i / W, w = i % W;
    uint32_t out = 0;
    for (uint32_t jj = 0; jj < cf; jj++) {
      const uint64_t slot = w * cf + jj;
      if (slot < N) {
        uint64_t w64;
        uint32_t bit;
        dense_locate(slot, K, b, &w64, &bit);
        out |= ((uint32_t)(dtc64[c * stride64 + w64] >> bit) & mask) << (jj * S);
      }
    }
    dst[i] = out;
  }
}

// ---- planar packing (CPIR_PACK_PLANAR; layout documented at cpir_dtc_layout, consumed by respond_planar.hip) -----------------------
// One block = one super-tile: 16 columns x 512 slots.  FROM_REF = false reads the encoded DB matrix D (N x C, leading dim ld);
// FROM_REF = true reads the reference's compressed transposed matrix (C x W, ld = W), dropping fields >= N and bits >= b as
// normalise_word does.  Also accumulates, per column, the wrap-around sum of its fields (the correction term of the signed-byte
// arithmetic) into colsum (zeroed by the launcher).
template <bool FROM_REF>
__global__ void __launch_bounds__(kThreads) planar_pack_kernel(const uint32_t* __restrict__ src, uint64_t ld, uint64_t N, uint32_t C,
                                                                uint32_t b, uint32_t cf, uint32_t hb, uint32_t ks_total,
                                                                uint4* __restrict__ tiles, uint32_t* __restrict__ colsum,
                                                                uint32_t* __restrict__ or_of_entries) {
  __shared__ uint16_t tile[CPIR_PLANAR_SLOTS_PER_TILE][18];
  const int tid = threadIdx.x;
  const uint32_t ks = blockIdx.x, T = blockIdx.y, c0 = T * 16;
  const uint64_t n0 = (uint64_t)ks * CPIR_PLANAR_SLOTS_PER_TILE;
  const uint32_t mask = (1u << b) - 1u;
  uint32_t seen = 0;
  if constexpr (!FROM_REF) {
    for (int i = tid; i < (int)CPIR_PLANAR_SLOTS_PER_TILE * 16; i += kThreads) {  // 16 lanes cover one 64-byte piece of a D row
      const uint32_t c = i & 15, sl = i >> 4;
      const uint64_t n = n0 + sl;
      const uint32_t v = (n < N && c0 + c < C) ? src[n * ld + c0 + c] : 0u;
      seen |= v;
      tile[sl][c] = (uint16_t)(v & mask);  // matrix.rs:121,149,181: every field is masked to b bits
    }
  } else {
    const uint32_t S = 32 / cf;
    for (int i = tid; i < (int)CPIR_PLANAR_SLOTS_PER_TILE * 16; i += kThreads) {  // lanes run along the packed words of one row
      const uint32_t sl = i & (CPIR_PLANAR_SLOTS_PER_TILE - 1), c = i >> 9;
      const uint64_t n = n0 + sl;
      uint32_t v = 0;
      if (n < N && c0 + c < C) v = (src[(uint64_t)(c0 + c) * ld + n / cf] >> ((uint32_t)(n % cf) * S)) & mask;
      tile[sl][c] = (uint16_t)v;
    }
  }
  __syncthreads();

  const uint32_t st16 = (8 + hb) * 64;
  uint4* base = tiles + ((uint64_t)T * ks_total + ks) * st16;
  // low bytes, XOR 0x80 (signed-byte operand): k-block kb, lane l = 16*g + cl holds slots 64*kb + 16*g + 0..15 of column cl
  for (int piece = tid; piece < 512; piece += kThreads) {
    const uint32_t kb = piece >> 6, l = piece & 63, cl = l & 15, g = l >> 4;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int jx = 0; jx < 16; jx++) {
      const uint32_t f = tile[kb * 64 + g * 16 + jx][cl];
      w[jx >> 2] |= ((f & 0xFFu) ^ 0x80u) << (8 * (jx & 3));
    }
    base[kb * 64 + l] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // bit planes: plane p, lane l: dword w, bit 8*jj + 4*s + d  <-  bit 8+p of slot 64*(2w+s) + 16*g + 4*d + jj
  for (uint32_t p = tid >> 6; p < hb; p += kThreads / 64) {
    const uint32_t l = tid & 63, cl = l & 15, g = l >> 4;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ww = 0; ww < 4; ww++)
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int d = 0; d < 4; d++)
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            const uint32_t f = tile[(2 * ww + s2) * 64 + g * 16 + 4 * d + jj][cl];
            w[ww] |= ((f >> (8 + p)) & 1u) << (8 * jj + 4 * s2 + d);
          }
    base[512 + p * 64 + l] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (tid < 16) {
    uint32_t sum = 0;
    for (uint32_t sl = 0; sl < CPIR_PLANAR_SLOTS_PER_TILE; sl++) sum += tile[sl][tid];
    if (sum) atomicAdd(colsum + c0 + tid, sum);
  }
  if (or_of_entries) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) seen |= __shfl_xor(seen, off, 64);
    if ((tid & 63) == 0 && seen) atomicOr(or_of_entries, seen);
  }
}

// ---- planar packing from D, the setup path: no LDS transpose, no block barrier -----------------------------------------------------
// One WAVE builds one unit = 64 columns (4 column tiles) x one super-tile step (512 slots); the 4 waves of a block take 4 adjacent
// 64-column stripes of the same rows, so a block reads 1 KiB contiguous of every D row it touches.  Lane (g, lc) owns columns
// 4*lc .. 4*lc+3 of the stripe and the 16-slot group g: per k-block it loads 16 rows x 16 bytes (the wave: 4 rows x 256 B per
// instruction, 16 KiB in flight), and then HOLDS, for each of its 4 columns, the 16 consecutive slots that make one 16-byte piece of the
// MFMA operand image -- the transpose happens in registers (v_perm_b32 byte gathers).  The bit planes of a (column, group) span all 8
// k-blocks of the step and are accumulated in registers across them.  Pieces leave through a wave-private, swizzled 4 KiB LDS window that
// only re-orders them so that every global store instruction writes 1 KiB contiguous.
// Per-column field sums (correction term of the signed-byte arithmetic): v_sad_u8 over the packed low bytes + popcounts of the planes.
// the packed image is written once and not read again by the pass that writes it: non-temporal stores keep it from displacing D's lines
typedef uint32_t pack_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_nt(uint4* p, const uint4& x) {
  __builtin_nontemporal_store(pack_u32x4{x.x, x.y, x.z, x.w}, reinterpret_cast<pack_u32x4*>(p));
}

struct PackStreamArgs {
  const uint32_t* D;
  uint64_t ld, N;
  uint32_t C, col_tiles, stripe_groups, ks_total;
  uint32_t low_mask;  // (2^b - 1) in every byte for b < 8 (the fields are masked to b bits: matrix.rs:121,149,181), all ones otherwise
  uint4* tiles;
  uint32_t* colsum;
  uint32_t* or_of_entries;
  // optional: the byte (field >> 8) XOR 0x80 of every field as a plane of 1 KiB MFMA operand pieces [column tile][k-block of 64 slots],
  // laid out like the low-byte pieces inside the tiles -- the second operand plane of the hint matmul (matmul_mfma.hip), whose first
  // plane ARE the low-byte pieces of the image; written in the same pass so that D is read once for both products
  uint4* hi_plane;
  uint32_t kb_total;  // k-blocks of 64 slots in the plane: ceil(N / 64)
  // optional slot map (compact.hip): slot n of the image is row keep[n] of D -- the rows that hold something are packed straight out of
  // the whole matrix, no gathered copy of it in between (which would need room for D twice); NULL: slot n is row n
  const uint32_t* keep;
};

// GUARD: this step reaches past the last slot (only the last step of a database whose N is not a multiple of 512)
template <int HB, bool VEC, bool GUARD>
__device__ __forceinline__ void pack_stream_unit(const PackStreamArgs& a, uint4* my_stage, uint32_t lane, uint32_t stripe, uint32_t ks) {
  constexpr uint32_t ST16 = (8 + HB) * 64;  // uint4 per super-tile
  constexpr uint32_t HMASK = ((1u << HB) - 1u) * 0x01010101u;
  constexpr int HP = HB ? HB : 1;  // (no zero-length arrays)
  const uint32_t g = lane >> 4, lc = lane & 15;
  const uint32_t c0 = stripe * 64 + 4 * lc;  // this lane's first column
  const uint64_t n0 = (uint64_t)ks * CPIR_PLANAR_SLOTS_PER_TILE;
  bool cvalid[4];
#pragma unroll
  for (int i = 0; i < 4; i++) cvalid[i] = c0 + i < a.C;
  // VEC (ld and c0 multiples of 4, D 16-byte aligned): c0 < C <= ld implies c0 + 3 < ld, so a lane either loads 16 valid bytes of its
  // rows or lies wholly past C, reads column 0.. of the same rows instead and zeroes them
  const uint32_t csafe = cvalid[0] ? c0 : 0;
  const uint32_t piece0 = 64 * (lc >> 2) + 16 * g + 4 * (lc & 3);

  uint32_t plane[4][HP][4];
  uint32_t lowsum[4] = {0, 0, 0, 0}, seen = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int p = 0; p < HP; p++)
#pragma unroll
      for (int w = 0; w < 4; w++) plane[i][p][w] = 0;

#pragma unroll
  for (int kb = 0; kb < 8; kb++) {
    uint32_t v[16][4];
    const uint64_t nb = n0 + 64 * kb + 16 * g;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t n = nb + j;
      const uint64_t ns = (!GUARD || n < a.N) ? n : a.N - 1;  // clamped rows are zeroed below
      const uint64_t nr = a.keep ? (uint64_t)a.keep[ns] : ns;
      if constexpr (VEC) {
        const uint4 t = *reinterpret_cast<const uint4*>(a.D + nr * a.ld + csafe);
        v[j][0] = t.x, v[j][1] = t.y, v[j][2] = t.z, v[j][3] = t.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) v[j][i] = a.D[nr * a.ld + (cvalid[i] ? c0 + i : 0)];
      }
    }
    uint32_t seen_kb = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const bool rv = !GUARD || nb + j < a.N;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        v[j][i] = (rv && cvalid[i]) ? v[j][i] : 0u;
        seen_kb |= v[j][i];
      }
    }
    seen |= seen_kb;
    asm volatile("" : "+v"(seen));  // materialise here: otherwise the OR tree over all 8 k-blocks is built at the end and every value lives until then
    uint32_t W[4][4], WH[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int d = 0; d < 4; d++) {
        WH[i][d] = 0x80808080u;
        const uint32_t lo = gather_byte4(v[4 * d][i], v[4 * d + 1][i], v[4 * d + 2][i], v[4 * d + 3][i], 0x0400u) & a.low_mask;
        lowsum[i] = __builtin_amdgcn_sad_u8(lo, 0u, lowsum[i]);
        W[i][d] = lo ^ 0x80808080u;  // signed-byte operand
        // bits 8.. of the four fields, one per byte (fields are masked to b bits: matrix.rs:121,149,181), spread over the planes:
        // plane p, dword kb >> 1, bit 8*jj + 4*(kb & 1) + d  <-  bit 8+p of slot 64*kb + 16*g + 4*d + jj
        if constexpr (HB > 0) {
          const uint32_t hi = gather_byte4(v[4 * d][i], v[4 * d + 1][i], v[4 * d + 2][i], v[4 * d + 3][i], 0x0501u) & HMASK;
#pragma unroll
          for (int p = 0; p < HB; p++) plane[i][p][kb >> 1] |= ((hi >> p) & 0x01010101u) << (4 * (kb & 1) + d);
          WH[i][d] = hi ^ 0x80808080u;
        }
      }
    // materialise the accumulators now (as `seen` above): the compiler would otherwise sink this k-block's plane arithmetic to where the
    // planes are stored, after the last k-block, and keep all 64 loaded values of all 8 k-blocks alive (512 registers + scratch)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      asm volatile("" : "+v"(lowsum[i]));
#pragma unroll
      for (int p = 0; p < HB; p++) asm volatile("" : "+v"(plane[i][p][kb >> 1]));
    }
    // the four pieces of this lane -> their places in the four tiles' k-block kb, via the wave's staging window
#pragma unroll
    for (int i = 0; i < 4; i++) my_stage[stage_swz(piece0 + i)] = make_uint4(W[i][0], W[i][1], W[i][2], W[i][3]);
    __builtin_amdgcn_wave_barrier();  // LDS serves a wave's accesses in order; this only pins the compiler's schedule
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint4 x = my_stage[stage_swz(64 * k + lane)];
      const uint32_t T = stripe * 4 + k;
      if (T < a.col_tiles) store16_nt(a.tiles + ((uint64_t)T * a.ks_total + ks) * ST16 + kb * 64 + lane, x);
    }
    __builtin_amdgcn_wave_barrier();
    if (HB > 0 && a.hi_plane) {  // wave-uniform: the same four pieces of the high-byte plane, through the same window
      const uint32_t kbg = ks * 8 + kb;
#pragma unroll
      for (int i = 0; i < 4; i++) my_stage[stage_swz(piece0 + i)] = make_uint4(WH[i][0], WH[i][1], WH[i][2], WH[i][3]);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 x = my_stage[stage_swz(64 * k + lane)];
        const uint32_t T = stripe * 4 + k;
        if (T < a.col_tiles && kbg < a.kb_total) store16_nt(a.hi_plane + ((uint64_t)T * a.kb_total + kbg) * 64 + lane, x);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // one k-block's 16 loads per lane in flight at a time (16 KiB per wave); the other waves of the SIMD hide the latency
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int p = 0; p < HB; p++) {
#pragma unroll
    for (int i = 0; i < 4; i++) my_stage[stage_swz(piece0 + i)] = make_uint4(plane[i][p][0], plane[i][p][1], plane[i][p][2], plane[i][p][3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint4 x = my_stage[stage_swz(64 * k + lane)];
      const uint32_t T = stripe * 4 + k;
      if (T < a.col_tiles) store16_nt(a.tiles + ((uint64_t)T * a.ks_total + ks) * ST16 + 512 + p * 64 + lane, x);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // column sums of the fields of this step: low bytes + 2^(8+p) * (ones in plane p), summed over the four slot groups
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t sum = lowsum[i];
#pragma unroll
    for (int p = 0; p < HB; p++)
      sum += (uint32_t)(__builtin_popcount(plane[i][p][0]) + __builtin_popcount(plane[i][p][1]) + __builtin_popcount(plane[i][p][2]) +
                        __builtin_popcount(plane[i][p][3])) << (8 + p);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (g == 0 && sum) atomicAdd(a.colsum + c0 + i, sum);  // sum != 0 implies c0 + i < C
  }
  if (a.or_of_entries) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) seen |= __shfl_xor(seen, off, 64);
    if (lane == 0 && seen) atomicOr(a.or_of_entries, seen);
  }
}

// GUARD = false: steps [0, ks_count) all lie inside the database; GUARD = true: the one last step of a database whose slot count is not
// a multiple of 512 (its own launch, so that its clamps and selects do not cost the main kernel registers)
template <int HB, bool VEC, bool GUARD>
__global__ void __launch_bounds__(kThreads) planar_pack_stream_kernel(const PackStreamArgs a, uint32_t ks_first) {
  __shared__ uint4 stage[kThreads / 64][256];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t sg = blockIdx.x % a.stripe_groups;  // stripe group fastest: co-resident blocks read whole rows of D
  const uint32_t ks = ks_first + blockIdx.x / a.stripe_groups;
  pack_stream_unit<HB, VEC, GUARD>(a, stage[wave], lane, sg * 4 + wave, ks);
}

// ---- the same pass with WHOLE ROWS per block (wide databases) ---------------------------------------------------------------------
// pack_stream_unit gives a wave 64 columns, so one load instruction fetches 4 x 256 bytes from 4 rows 16 apart and a block reads 1 KiB
// pieces of its rows: the memory system sees short segments from many rows at once.  Here a wave owns 256 adjacent columns (16 column
// tiles): lane l holds columns 4l .. 4l + 3 of the stripe, one load instruction reads 1 KiB CONTIGUOUS of one row, and the four waves of
// a block cover 1024 columns -- with C <= 1024 (2^20 keys x 1 kB values: 940) a block streams whole rows, 16 consecutive rows = one
// contiguous run of D per batch.  A batch is 16 rows (one slot group g of a k-block); the 512 slots of a step are walked slot group by
// slot group (g outer, the 8 k-blocks inner), so that the bit planes of a (column, slot group) -- which span all 8 k-blocks -- still
// accumulate in four registers per column and plane.  What a batch produces for a column tile is the QUARTER of an operand piece that
// belongs to slot group g: 16 lanes x 16 bytes = 256 contiguous bytes, so every store instruction writes four 256-byte runs (the
// streaming kernel writes 1 KiB runs; reads outnumber writes 3.5 to 1).  The stores are non-temporal: the image is not read again by this
// pass and should not displace D's lines on their way through the caches (about 2 % at 2^20 keys and with 8 kB values).
// position of fragment p = 4 * lane + i (written) / 64 * k + lane (read back) in the wave's staging window: the low two bits are XOR-ed with
// bits 4..5, so that the 8 lanes a ds_write_b128 serves together -- 64 bytes apart -- fall into 8 different 16-byte bank groups; a
// permutation inside aligned groups of 4 fragments, so the linear read-back stays conflict-free
__device__ __forceinline__ uint32_t rows_swz(uint32_t p) { return p ^ ((p >> 4) & 3u); }


template <int HB, bool VEC, bool GUARD>
__device__ __forceinline__ void pack_rows_unit(const PackStreamArgs& a, uint4* my_stage, uint32_t lane, uint32_t wstripe, uint32_t ks) {
  constexpr uint32_t ST16 = (8 + HB) * 64;  // uint4 per super-tile
  constexpr uint32_t HMASK = ((1u << HB) - 1u) * 0x01010101u;
  constexpr int HP = HB ? HB : 1;
  const uint32_t c0 = wstripe * 256 + 4 * lane;  // this lane's first column
  const uint32_t T0 = wstripe * 16;              // the wave's first column tile
  const uint64_t n0 = (uint64_t)ks * CPIR_PLANAR_SLOTS_PER_TILE;
  bool cvalid[4];
#pragma unroll
  for (int i = 0; i < 4; i++) cvalid[i] = c0 + i < a.C;
  const uint32_t csafe = cvalid[0] ? c0 : 0;  // (as pack_stream_unit: a lane wholly past C reads column 0.. of the same rows and zeroes them)
  uint32_t colsum_acc[4] = {0, 0, 0, 0}, seen = 0;
  uint32_t plane[4][HP][4];
  uint32_t lowsum[4] = {0, 0, 0, 0};

  // the 16 rows of batch (g_, kb_): row pointers are wave-uniform (scalar registers), the lane adds its column offset
  auto load_batch = [&](uint4(&buf)[16], uint32_t g_, uint32_t kb_) {
    const uint64_t nb = n0 + 64 * kb_ + 16 * g_;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t n = nb + j;
      const uint64_t ns = (!GUARD || n < a.N) ? n : a.N - 1;  // clamped rows are zeroed in process_batch
      const uint64_t nr = a.keep ? (uint64_t)a.keep[ns] : ns;   // (wave-uniform: scalar loads)
      if constexpr (VEC) {
        buf[j] = *reinterpret_cast<const uint4*>(a.D + nr * a.ld + csafe);
      } else {
        const uint32_t* row = a.D + nr * a.ld;
        buf[j] = make_uint4(row[cvalid[0] ? c0 : 0], row[cvalid[1] ? c0 + 1 : 0], row[cvalid[2] ? c0 + 2 : 0], row[cvalid[3] ? c0 + 3 : 0]);
      }
    }
  };

  auto process_batch = [&](const uint4(&buf)[16], uint32_t g, int kb) {
    const uint64_t nb = n0 + 64 * kb + 16 * g;
    // columns past C and (GUARD) rows past N contribute zero fields.  The row condition is wave-uniform and only exists in the guarded
    // kernel; the column condition is applied to the 16 gathered words of a column, not to its 64 loaded ones
    uint32_t e[16][4];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const bool rv = !GUARD || nb + j < a.N;
      e[j][0] = rv ? buf[j].x : 0u, e[j][1] = rv ? buf[j].y : 0u, e[j][2] = rv ? buf[j].z : 0u, e[j][3] = rv ? buf[j].w : 0u;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t col_or = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) col_or |= e[j][i];
      seen |= cvalid[i] ? col_or : 0u;
    }
    asm volatile("" : "+v"(seen));
    uint32_t W[4][4], WH[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int d = 0; d < 4; d++) {
        WH[i][d] = 0x80808080u;
        uint32_t lo = gather_byte4(e[4 * d][i], e[4 * d + 1][i], e[4 * d + 2][i], e[4 * d + 3][i], 0x0400u) & a.low_mask;
        lo = cvalid[i] ? lo : 0u;
        lowsum[i] = __builtin_amdgcn_sad_u8(lo, 0u, lowsum[i]);
        W[i][d] = lo ^ 0x80808080u;
        if constexpr (HB > 0) {
          uint32_t hi = gather_byte4(e[4 * d][i], e[4 * d + 1][i], e[4 * d + 2][i], e[4 * d + 3][i], 0x0501u) & HMASK;
          hi = cvalid[i] ? hi : 0u;
#pragma unroll
          for (int p = 0; p < HB; p++) plane[i][p][kb >> 1] |= ((hi >> p) & 0x01010101u) << (4 * (kb & 1) + d);
          WH[i][d] = hi ^ 0x80808080u;
        }
      }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      asm volatile("" : "+v"(lowsum[i]));
#pragma unroll
      for (int p = 0; p < HB; p++) asm volatile("" : "+v"(plane[i][p][kb >> 1]));
    }
    // this lane's four fragments (slot group g of k-block kb, columns c0 .. c0 + 3) -> their quarter pieces, via the staging window
#pragma unroll
    for (int i = 0; i < 4; i++) my_stage[rows_swz(4 * lane + i)] = make_uint4(W[i][0], W[i][1], W[i][2], W[i][3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint4 x = my_stage[rows_swz(64 * k + lane)];
      const uint32_t T = T0 + 4 * k + (lane >> 4);
      if (T < a.col_tiles) store16_nt(a.tiles + ((uint64_t)T * a.ks_total + ks) * ST16 + kb * 64 + 16 * g + (lane & 15), x);
    }
    __builtin_amdgcn_wave_barrier();
    if (HB > 1 && a.hi_plane) {  // wave-uniform: the same fragments of the high-byte plane, through the same window (one bit plane: no such plane)
      const uint32_t kbg = ks * 8 + kb;
#pragma unroll
      for (int i = 0; i < 4; i++) my_stage[rows_swz(4 * lane + i)] = make_uint4(WH[i][0], WH[i][1], WH[i][2], WH[i][3]);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 x = my_stage[rows_swz(64 * k + lane)];
        const uint32_t T = T0 + 4 * k + (lane >> 4);
        if (T < a.col_tiles && kbg < a.kb_total) store16_nt(a.hi_plane + ((uint64_t)T * a.kb_total + kbg) * 64 + 16 * g + (lane & 15), x);
      }
      __builtin_amdgcn_wave_barrier();
    }
  };

  // (Requesting batch t + 1 before batch t is processed -- two register sets -- changes nothing: 1.12 ms either way at 2^20 keys; the pass is
  // bound by what the memory system makes of the mix of this read stream and the scattered 256-byte writes, and where the image happens to
  // lie relative to D moves the time by up to 10 %.)
  uint4 buf[16];
#pragma unroll 1
  for (uint32_t g = 0; g < 4; g++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      lowsum[i] = 0;
#pragma unroll
      for (int p = 0; p < HP; p++)
#pragma unroll
        for (int w = 0; w < 4; w++) plane[i][p][w] = 0;
    }
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
      load_batch(buf, g, kb);
      process_batch(buf, g, kb);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the bit planes of slot group g: one fragment per column and plane
#pragma unroll
    for (int p = 0; p < HB; p++) {
#pragma unroll
      for (int i = 0; i < 4; i++) my_stage[rows_swz(4 * lane + i)] = make_uint4(plane[i][p][0], plane[i][p][1], plane[i][p][2], plane[i][p][3]);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 x = my_stage[rows_swz(64 * k + lane)];
        const uint32_t T = T0 + 4 * k + (lane >> 4);
        if (T < a.col_tiles) store16_nt(a.tiles + ((uint64_t)T * a.ks_total + ks) * ST16 + 512 + p * 64 + 16 * g + (lane & 15), x);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // field sums of this slot group: low bytes + 2^(8+p) * (ones in plane p)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t sum = lowsum[i];
#pragma unroll
      for (int p = 0; p < HB; p++)
        sum += (uint32_t)(__builtin_popcount(plane[i][p][0]) + __builtin_popcount(plane[i][p][1]) + __builtin_popcount(plane[i][p][2]) +
                          __builtin_popcount(plane[i][p][3])) << (8 + p);
      colsum_acc[i] += sum;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (colsum_acc[i]) atomicAdd(a.colsum + c0 + i, colsum_acc[i]);  // != 0 implies c0 + i < C
  if (a.or_of_entries) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) seen |= __shfl_xor(seen, off, 64);
    if (lane == 0 && seen) atomicOr(a.or_of_entries, seen);
  }
}

template <int HB, bool VEC, bool GUARD>
__global__ void __launch_bounds__(kThreads) planar_pack_rows_kernel(const PackStreamArgs a, uint32_t ks_first) {
  __shared__ uint4 stage[kThreads / 64][256];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t sg = blockIdx.x % a.stripe_groups;  // groups of 1024 columns; fastest, so that co-resident blocks read whole rows
  const uint32_t ks = ks_first + blockIdx.x / a.stripe_groups;
  const uint32_t wstripe = sg * 4 + wave;
  if (wstripe * 16 >= a.col_tiles) return;  // wave-uniform: nothing of this stripe exists (no block-wide barrier anywhere)
  pack_rows_unit<HB, VEC, GUARD>(a, stage[wave], lane, wstripe, ks);
}

// field (n, c) of a planar image
__device__ __forceinline__ uint32_t planar_field(const uint8_t* bytes, uint64_t n, uint32_t c, uint32_t hb, uint32_t ks_total) {
  const uint32_t T = c >> 4, cl = c & 15;
  const uint64_t ks = n / CPIR_PLANAR_SLOTS_PER_TILE;
  const uint32_t sl = (uint32_t)(n % CPIR_PLANAR_SLOTS_PER_TILE), kb = sl >> 6, g = (sl >> 4) & 3, jx = sl & 15, l = g * 16 + cl;
  const uint8_t* t = bytes + ((uint64_t)T * ks_total + ks) * ((8 + hb) * 1024ull);
  uint32_t f = (uint32_t)t[kb * 1024 + l * 16 + jx] ^ 0x80u;
  for (uint32_t p = 0; p < hb; p++) {
    const uint32_t word = reinterpret_cast<const uint32_t*>(t + 8192 + p * 1024 + l * 16)[kb >> 1];
    f |= ((word >> (8 * (jx & 3) + 4 * (kb & 1) + (jx >> 2))) & 1u) << (8 + p);
  }
  return f;
}

__global__ void __launch_bounds__(kThreads) planar_export_kernel(const uint8_t* __restrict__ bytes, uint64_t W, uint64_t N, uint32_t C,
                                                                  uint32_t cf, uint32_t hb, uint32_t ks_total, uint32_t* __restrict__ dst) {
  const uint32_t S = 32 / cf;
  const uint64_t total = (uint64_t)C * W;
  for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kThreads) {
    const uint64_t c = i / W, w = i % W;
    uint32_t out = 0;
    for (uint32_t jj = 0; jj < cf; jj++) {
      const uint64_t n = w * cf + jj;
      if (n < N) out |= planar_field(bytes, n, (uint32_t)c, hb, ks_total) << (jj * S);
    }
    dst[i] = out;
  }
}

int launch_planar_pack(const uint32_t* src, uint64_t ld, bool from_ref, const cpir_dtc_layout& L, uint32_t* dtc, uint32_t* or_of_entries,
                       hipStream_t stream, uint4* hi_plane = nullptr, const uint32_t* keep = nullptr) {
  const uint32_t hb = planar_hi_planes(L.mat_elem_bit_len);
  const uint64_t ks_total = (L.num_slots + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE;
  const uint32_t col_tiles = L.rows_padded / 16;
  if (!planar_offered(L.mat_elem_bit_len) || ks_total > 0x7fffffffull || col_tiles > 65535u) return CPIR_ERR_INVALID_ARGUMENT;
  uint32_t* colsum = dtc + (uint64_t)L.rows_padded * L.words_per_row_padded;
  CPIR_TRY(zero_words(colsum, L.rows_padded, stream));
  if (from_ref && (hi_plane || keep)) return CPIR_ERR_INVALID_ARGUMENT;
  if (from_ref) {
    const dim3 grid((unsigned)ks_total, col_tiles);
    hipLaunchKernelGGL((planar_pack_kernel<true>), grid, dim3(kThreads), 0, stream, src, ld, L.num_slots, L.num_cols, L.mat_elem_bit_len,
                       L.compression_factor, hb, (uint32_t)ks_total, reinterpret_cast<uint4*>(dtc), colsum, or_of_entries);
    CPIR_HIP_TRY(hipGetLastError());
    return CPIR_OK;
  }
  // wide databases: whole rows per block (planar_pack_rows_kernel); narrow ones keep the 64-column waves, which waste fewer lanes there
  const int rows_mode = pack_rows_mode();
  // (with two or more bit planes the whole-rows kernel needs more registers than two waves per SIMD leave it: the 64-column waves stay,
  // whatever "pack.rows" says)
  const bool rows = hb <= 1 && (rows_mode > 0 || (rows_mode < 0 && col_tiles >= 32));
  const uint32_t stripe_groups = rows ? (col_tiles + 63) / 64 : (col_tiles + 15) / 16;  // 4 waves x 16 (4) tiles per block
  if (ks_total * stripe_groups > 0x7fffffffull) return CPIR_ERR_INVALID_ARGUMENT;
  const uint64_t full_steps = L.num_slots / CPIR_PLANAR_SLOTS_PER_TILE;  // steps that lie wholly inside the database
  const bool vec = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0);
  PackStreamArgs pa;
  pa.D = src, pa.ld = ld, pa.N = L.num_slots, pa.C = L.num_cols, pa.col_tiles = col_tiles, pa.stripe_groups = stripe_groups;
  pa.ks_total = (uint32_t)ks_total, pa.tiles = reinterpret_cast<uint4*>(dtc), pa.colsum = colsum, pa.or_of_entries = or_of_entries;
  pa.low_mask = L.mat_elem_bit_len < 8 ? ((1u << L.mat_elem_bit_len) - 1u) * 0x01010101u : 0xFFFFFFFFu;
  pa.hi_plane = hb ? hi_plane : nullptr;
  pa.kb_total = (uint32_t)((L.num_slots + 63) / 64);
  pa.keep = keep;
#define LAUNCH_PP3(KERNEL_, HB_, VEC_)                                                                                            \
  do {                                                                                                                              \
    if (full_steps)                                                                                                                 \
      hipLaunchKernelGGL((KERNEL_<HB_, VEC_, false>), dim3((unsigned)(full_steps * stripe_groups)), dim3(kThreads), 0, stream, pa, 0u); \
    if (full_steps < ks_total)                                                                                                      \
      hipLaunchKernelGGL((KERNEL_<HB_, VEC_, true>), dim3(stripe_groups), dim3(kThreads), 0, stream, pa, (uint32_t)full_steps);     \
  } while (0)
  // (the whole-rows kernel exists for at most one bit plane -- every BASELINE configuration --: with more it needs the registers of a whole SIMD)
#define LAUNCH_PP2(HB_, VEC_)                                                      \
  do {                                                                             \
    if constexpr ((HB_) <= 1) {                                                    \
      if (rows) LAUNCH_PP3(planar_pack_rows_kernel, ((HB_) <= 1 ? (HB_) : 1), VEC_); \
      else LAUNCH_PP3(planar_pack_stream_kernel, HB_, VEC_);                       \
    } else {                                                                       \
      LAUNCH_PP3(planar_pack_stream_kernel, HB_, VEC_);                            \
    }                                                                              \
  } while (0)
#define LAUNCH_PP(HB_)               \
  do {                               \
    if (vec) LAUNCH_PP2(HB_, true);  \
    else LAUNCH_PP2(HB_, false);     \
  } while (0)
  switch (hb) {
    case 0: LAUNCH_PP(0); break;
    case 1: LAUNCH_PP(1); break;
    case 2: LAUNCH_PP(2); break;
    case 3: LAUNCH_PP(3); break;
    case 4: LAUNCH_PP(4); break;
    case 5: LAUNCH_PP(5); break;
    case 6: LAUNCH_PP(6); break;
    default: return CPIR_ERR_INVALID_ARGUMENT;
  }
#undef LAUNCH_PP2
#undef LAUNCH_PP3
#undef LAUNCH_PP
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace (reopened below)

// name of the kernel launch_transpose_compress runs for a database of this layout (packed from D, 16-byte loadable), as a kernel trace
// shows it -- so that a benchmark line and a rocprof summary can be matched
const char* pack_kernel_name(const cpir_dtc_layout& L) {
  if (L.packing == CPIR_PACK_DENSE64) return "transpose_compress_dense_kernel";
  if (L.packing != CPIR_PACK_PLANAR) return "transpose_compress_kernel";
  const uint32_t hb = planar_hi_planes(L.mat_elem_bit_len), col_tiles = L.rows_padded / 16;
  const int rows_mode = pack_rows_mode();
  const bool rows = hb <= 1 && (rows_mode > 0 || (rows_mode < 0 && col_tiles >= 32));  // (the same rule as launch_planar_pack)
  return rows ? "planar_pack_rows_kernel" : "planar_pack_stream_kernel";
}

namespace {

uint32_t grid_for(const Device* dev, uint64_t total) {
  uint64_t blocks = (total + kThreads - 1) / kThreads;
  const uint64_t cap = (uint64_t)dev->num_cus * 8;
  if (blocks > cap) blocks = cap;
  return (uint32_t)(blocks ? blocks : 1);
}

}  // namespace

uint64_t planar_hi_plane_bytes(const cpir_dtc_layout& L) {
  // (one bit plane, b = 9: the hint matmul expands its high-byte operand from the image's own bit plane -- no plane to write)
  if (L.packing != CPIR_PACK_PLANAR || planar_hi_planes(L.mat_elem_bit_len) <= 1) return 0;
  return (uint64_t)(L.rows_padded / 16) * ((L.num_slots + 63) / 64) * 1024;
}

// keep (planar packing only -- transpose_compress_takes_slot_map): L describes the COMPACT image and slot n of it is row keep[n] of D
bool transpose_compress_takes_slot_map(const cpir_dtc_layout& L) { return L.packing == CPIR_PACK_PLANAR; }

int launch_transpose_compress(const Device* dev, const uint32_t* D, uint64_t ldd, const cpir_dtc_layout& L, uint32_t* dtc,
                              uint32_t* or_of_entries, hipStream_t stream, void* hi_plane, const uint32_t* keep) {
  (void)dev;
  if (!D || !dtc || ldd < L.num_cols) return CPIR_ERR_INVALID_ARGUMENT;
  CPIR_TRY(check_layout(L));
  if (reinterpret_cast<uintptr_t>(dtc) % 16 != 0) return CPIR_ERR_INVALID_ARGUMENT;
  if (hi_plane && (L.packing != CPIR_PACK_PLANAR || reinterpret_cast<uintptr_t>(hi_plane) % 16 != 0)) return CPIR_ERR_INVALID_ARGUMENT;
  if (keep && !transpose_compress_takes_slot_map(L)) return CPIR_ERR_INVALID_ARGUMENT;
  if (L.packing == CPIR_PACK_PLANAR) return launch_planar_pack(D, ldd, false, L, dtc, or_of_entries, stream, reinterpret_cast<uint4*>(hi_plane), keep);
  if (L.packing == CPIR_PACK_DENSE64) {
    const uint64_t chunks = L.words_per_row_padded / L.chunk_words;
    if (chunks * 16 > 0x7fffffffull) return CPIR_ERR_INVALID_ARGUMENT;
    const dim3 grid((unsigned)(chunks * 16), (L.rows_padded + 31) / 32);
    const bool vec = (ldd % 4 == 0) && (reinterpret_cast<uintptr_t>(D) % 16 == 0);
#define LAUNCH_DENSE(B_)                                                                                                        \
  do {                                                                                                                          \
    if (vec)                                                                                                                    \
      hipLaunchKernelGGL((transpose_compress_dense_kernel<B_, true>), grid, dim3(kThreads), 0, stream, D, ldd, L.num_slots,     \
                         L.num_cols, dtc, L.words_per_row_padded, L.rows_padded, or_of_entries);                                 \
    else                                                                                                                        \
      hipLaunchKernelGGL((transpose_compress_dense_kernel<B_, false>), grid, dim3(kThreads), 0, stream, D, ldd, L.num_slots,    \
                         L.num_cols, dtc, L.words_per_row_padded, L.rows_padded, or_of_entries);                                 \
  } while (0)
    switch (L.mat_elem_bit_len) {
      case 7: LAUNCH_DENSE(7); break;
      case 9: LAUNCH_DENSE(9); break;
      case 11: LAUNCH_DENSE(11); break;
      case 12: LAUNCH_DENSE(12); break;
      default: return CPIR_ERR_INVALID_ARGUMENT;
    }
#undef LAUNCH_DENSE
    CPIR_HIP_TRY(hipGetLastError());
    return CPIR_OK;
  }
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
  if (L.packing == CPIR_PACK_PLANAR) {
    if (reinterpret_cast<uintptr_t>(dtc) % 16 != 0) return CPIR_ERR_INVALID_ARGUMENT;
    return launch_planar_pack(compressed, L.words_per_row, true, L, dtc, nullptr, stream);
  }
  if (L.packing == CPIR_PACK_DENSE64) {
    const uint64_t stride64 = L.words_per_row_padded / 2;
    hipLaunchKernelGGL(dtc_import_dense_kernel, dim3(grid_for(dev, (uint64_t)L.rows_padded * stride64)), dim3(kThreads), 0, stream,
                       compressed, L.words_per_row, L.num_slots, L.num_cols, L.mat_elem_bit_len, L.compression_factor,
                       L.fields_per_word, reinterpret_cast<uint64_t*>(dtc), stride64, L.rows_padded);
    CPIR_HIP_TRY(hipGetLastError());
    return CPIR_OK;
  }
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
  if (L.packing == CPIR_PACK_PLANAR) {
    const uint64_t ks_total = (L.num_slots + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE;
    hipLaunchKernelGGL(planar_export_kernel, dim3(grid), dim3(kThreads), 0, stream, reinterpret_cast<const uint8_t*>(dtc), L.words_per_row,
                       L.num_slots, L.num_cols, L.compression_factor, planar_hi_planes(L.mat_elem_bit_len), (uint32_t)ks_total, compressed);
    CPIR_HIP_TRY(hipGetLastError());
    return CPIR_OK;
  }
  if (L.packing == CPIR_PACK_DENSE64) {
    hipLaunchKernelGGL(dtc_export_dense_kernel, dim3(grid), dim3(kThreads), 0, stream, reinterpret_cast<const uint64_t*>(dtc),
                       L.words_per_row_padded / 2, L.words_per_row, L.num_slots, L.num_cols, L.mat_elem_bit_len,
                       L.compression_factor, L.fields_per_word, compressed);
    CPIR_HIP_TRY(hipGetLastError());
    return CPIR_OK;
  }
  hipLaunchKernelGGL(dtc_export_kernel, dim3(grid), dim3(kThreads), 0, stream, dtc, L.words_per_row_padded, L.words_per_row,
                     L.num_cols, compressed);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace cpir
