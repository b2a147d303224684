// respond.hip -- the online hot loop: r = q . D, streamed over the packed, transposed database resident in HBM.
//
// Replaces Matrix::row_vector_x_compressed_transposed_matrix (reference chalametpir_common/src/matrix.rs:328-485), which
// the reference runs on the CPU with rayon over the C outputs (matrix.rs:345/383/429); the reference has no GPU respond.
//
//   r[c] = sum_{n < N} q[n] *wrap field_{n mod cf}( DtC[c][n / cf] )            (u32, wrap-around)
//
// Roofline: HBM read.  Algorithmic bytes per query = 4*C*ceil(N/cf) + 4*N + 4*C.  MFMA does not apply
// (u32 wrap-around is not an MFMA type) and the VALU work per streamed dword is small, so the design is all about the
// memory system of an MI355X:
//
//   * work unit = R database rows x one 1024-word chunk (256 lanes x one 16-byte load per row): every wave-instruction
//     reads 1 KiB contiguous, every lane keeps R independent 16-byte loads in flight;
//   * the slice of q a lane needs (4*cf consecutive entries, 48 B for cf = 3) is loaded ONCE per unit into registers and
//     reused for the R rows, so q traffic (served by L2 / Infinity Cache) is cf/R of the HBM stream;
//   * exact u32 products without v_mul_lo_u32 (quarter rate): q is split once per unit into 16-bit halves and each
//     field (< 2^16) goes through two full-rate v_mad_u32_u24:  acc_lo += q_lo*d, acc_hi += q_hi*d,
//     r = acc_lo + (acc_hi << 16)  -- identical mod 2^32;
//   * persistent grid (CUs x resident blocks), units split evenly so there is no tail wave; the K (chunk) axis is first
//     split 8 ways by blockIdx % 8 -- blocks that share an XCD (observed round-robin placement; speed only, never
//     correctness) then share one eighth of q in that XCD's 4 MiB L2;
//   * per-row partial sums leave the block through a wave shuffle + LDS reduce and one u32 atomicAdd per (block, row);
//     integer atomics make the result independent of arrival order, i.e. still bit-exact.
//
// The database rows are padded to 16 rows / 1024 words with zeros (cpir_dtc_layout), so the only ragged edge left is
// the END OF q in the last chunk, handled by a guarded (wave-uniform) path.
#include "cpir_internal.hpp"

namespace cpir {

namespace {

constexpr int kThreads = 256;
constexpr int kWordsPerLane = 4;
constexpr int kChunkWords = kThreads * kWordsPerLane;
static_assert(kChunkWords == CPIR_DTC_WORD_ALIGN, "chunk must equal the layout's word alignment");

struct RespondArgs {
  const uint32_t* dtc;
  const uint32_t* q;
  uint32_t* r;
  uint64_t row_stride;     // words
  uint64_t q_len;          // entries in (each) q
  uint64_t q_slot_offset;  // first global slot held by this DtC
  uint32_t num_cols;       // C: rows of DtC that produce output
  uint32_t groups;         // rows_padded / R
  uint32_t chunks_total;   // words_per_row_padded / 1024
  uint32_t nx;             // K-axis split by blockIdx % nx (8 or 1)
  uint32_t q_scalar;       // q not 16-byte loadable -> guarded scalar loads everywhere
  uint32_t* zero_next;     // output of the NEXT launch on this stream, zeroed here so a run of launches needs one memset
  uint32_t zero_count;
};

template <int CF>
__device__ __forceinline__ uint32_t field(uint32_t w, int j) {
  constexpr int S = 32 / CF;
  // unused high bits (cf = 3: bits 30,31) and bits >= b inside a slot are zero by construction of the layout
  if (j == CF - 1) return w >> (S * (CF - 1));
  return __builtin_amdgcn_ubfe(w, S * j, S);
}

__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ uint4 load16(const uint4* p) {
  if constexpr (NT) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int CF, int R, int Q, bool NT>
__global__ void __launch_bounds__(kThreads) respond_kernel(const RespondArgs a) {
  constexpr int NQ = kWordsPerLane * CF;  // q entries per lane per chunk
  __shared__ uint32_t sm[Q][kThreads / 64][R];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  if (blockIdx.x == 0 && a.zero_next)
    for (uint32_t i = tid; i < a.zero_count; i += kThreads) a.zero_next[i] = 0;

  // ---- static partition of the (row group, chunk) units over the persistent grid --------------------------------
  const uint32_t nx = a.nx;
  const uint32_t xcd = blockIdx.x % nx;
  const uint32_t j = blockIdx.x / nx;
  const uint32_t nb = gridDim.x / nx;  // host guarantees gridDim.x % nx == 0
  const uint32_t kb = (uint32_t)(((uint64_t)a.chunks_total * xcd) / nx);
  const uint32_t ke = (uint32_t)(((uint64_t)a.chunks_total * (xcd + 1)) / nx);
  const uint32_t span = ke - kb;
  if (span == 0) return;
  const uint64_t units = (uint64_t)a.groups * span;
  const uint64_t u_begin = units * j / nb;
  const uint64_t u_end = units * (j + 1) / nb;
  if (u_begin >= u_end) return;

  uint32_t g = (uint32_t)(u_begin / span);
  uint32_t kc = kb + (uint32_t)(u_begin % span);

  uint32_t acc_lo[Q][R], acc_hi[Q][R];
#pragma unroll
  for (int b = 0; b < Q; b++)
#pragma unroll
    for (int r = 0; r < R; r++) acc_lo[b][r] = acc_hi[b][r] = 0;

  auto flush = [&](uint32_t grp) {
#pragma unroll
    for (int b = 0; b < Q; b++)
#pragma unroll
      for (int r = 0; r < R; r++) {
        const uint32_t v = wave_sum(acc_lo[b][r] + (acc_hi[b][r] << 16));
        if (lane == 0) sm[b][wave][r] = v;
        acc_lo[b][r] = acc_hi[b][r] = 0;
      }
    __syncthreads();
    if (tid < Q * R) {
      const int b = tid / R, r = tid % R;
      uint32_t s = 0;
#pragma unroll
      for (int w = 0; w < kThreads / 64; w++) s += sm[b][w][r];
      const uint32_t row = grp * R + r;
      if (row < a.num_cols) atomicAdd(a.r + (uint64_t)b * a.num_cols + row, s);
    }
    __syncthreads();
  };

  for (uint64_t u = u_begin; u < u_end; u++) {
    const uint64_t w0 = (uint64_t)kc * kChunkWords + (uint64_t)tid * kWordsPerLane;
    const uint64_t qbase = a.q_slot_offset + (uint64_t)CF * w0;
    // wave-uniform: does this chunk reach past the end of q?
    const bool guarded = a.q_scalar || (a.q_slot_offset + (uint64_t)CF * ((uint64_t)(kc + 1) * kChunkWords) > a.q_len);

    // ---- the R x 16-byte database loads of this unit (issued first: they are the long-latency HBM stream) -------
    const uint4* p = reinterpret_cast<const uint4*>(a.dtc + (uint64_t)g * R * a.row_stride + w0);
    const uint64_t stride16 = a.row_stride / 4;
    uint4 d[R];
#pragma unroll
    for (int r = 0; r < R; r++) d[r] = load16<NT>(p + (uint64_t)r * stride16);

    // ---- this lane's slice of q, split into 16-bit halves ---------------------------------------------------------
    uint32_t qlo[Q][NQ], qhi[Q][NQ];
#pragma unroll
    for (int b = 0; b < Q; b++) {
      const uint32_t* qb = a.q + (uint64_t)b * a.q_len;
      uint32_t qv[NQ];
      if (!guarded) {
        const uint4* q4 = reinterpret_cast<const uint4*>(qb + qbase);
#pragma unroll
        for (int i = 0; i < CF; i++) {
          const uint4 t = q4[i];
          qv[4 * i + 0] = t.x;
          qv[4 * i + 1] = t.y;
          qv[4 * i + 2] = t.z;
          qv[4 * i + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NQ; i++) qv[i] = (qbase + i < a.q_len) ? qb[qbase + i] : 0u;
      }
#pragma unroll
      for (int i = 0; i < NQ; i++) {
        qlo[b][i] = qv[i] & 0xffffu;
        qhi[b][i] = qv[i] >> 16;
      }
    }

    // ---- multiply-accumulate ---------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint32_t wd[4] = {d[r].x, d[r].y, d[r].z, d[r].w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int jf = 0; jf < CF; jf++) {
          const uint32_t f = field<CF>(wd[k], jf);
#pragma unroll
          for (int b = 0; b < Q; b++) {
            acc_lo[b][r] = mad24(qlo[b][k * CF + jf], f, acc_lo[b][r]);
            acc_hi[b][r] = mad24(qhi[b][k * CF + jf], f, acc_hi[b][r]);
          }
        }
      }
    }

    // ---- next unit -------------------------------------------------------------------------------------------------
    kc++;
    if (kc == ke) {
      flush(g);
      g++;
      kc = kb;
    }
  }
  if (kc != kb) flush(g);  // a partially covered row group is still pending
}

// ---- tuning state (benchmark harness can override; defaults chosen from measurements) ----------------
struct Tuning {
  int rows_per_unit = 8;   // R in {4, 8, 16}
  int nontemporal = 1;     // streamed DB loads with the nt cache policy (read once per query; keeps q in L2)
  int blocks_per_cu = 2;   // resident 256-thread blocks per CU; 0 = ask the occupancy API (7 for R = 8)
  int xcd_split = 1;       // split the K axis by blockIdx % 8
  int batch_fusion = 1;    // respond_batch: 1 = passes of 4/2/1 queries share one DB stream, 0 = one pass per query
};
Tuning g_tuning;
std::mutex g_tuning_mu;

using KernelFn = void (*)(const RespondArgs);

template <int CF, int Q>
KernelFn pick_kernel(int R, bool nt) {
  if constexpr (Q == 1) {
    if (R == 4) return nt ? respond_kernel<CF, 4, Q, true> : respond_kernel<CF, 4, Q, false>;
    if (R == 16) return nt ? respond_kernel<CF, 16, Q, true> : respond_kernel<CF, 16, Q, false>;
    return nt ? respond_kernel<CF, 8, Q, true> : respond_kernel<CF, 8, Q, false>;
  } else {
    // batched variants keep Q*R accumulator pairs in registers: R is fixed at 4
    return nt ? respond_kernel<CF, 4, Q, true> : respond_kernel<CF, 4, Q, false>;
  }
}

template <int Q>
KernelFn pick_kernel_cf(uint32_t cf, int R, bool nt) {
  switch (cf) {
    case 2: return pick_kernel<2, Q>(R, nt);
    case 3: return pick_kernel<3, Q>(R, nt);
    default: return pick_kernel<4, Q>(R, nt);
  }
}

}  // namespace

extern "C" int cpir_tuning_set(const char* key, int value) {
  if (!key) return CPIR_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  if (!strcmp(key, "respond.rows_per_unit")) {
    if (value != 4 && value != 8 && value != 16) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.rows_per_unit = value;
  } else if (!strcmp(key, "respond.nontemporal")) {
    g_tuning.nontemporal = value ? 1 : 0;
  } else if (!strcmp(key, "respond.blocks_per_cu")) {
    if (value < 0 || value > 8) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.blocks_per_cu = value;
  } else if (!strcmp(key, "respond.xcd_split")) {
    g_tuning.xcd_split = value ? 1 : 0;
  } else if (!strcmp(key, "respond.batch_fusion")) {
    g_tuning.batch_fusion = value ? 1 : 0;
  } else {
    return CPIR_ERR_INVALID_ARGUMENT;
  }
  return CPIR_OK;
}

bool respond_batch_fusion() {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  return g_tuning.batch_fusion != 0;
}

uint64_t respond_scratch_words(const cpir_dtc_layout&, uint32_t) {
  // the current kernel needs no scratch (partial sums leave through integer atomics); the parameter stays in the ABI so
  // a partial-buffer variant can be swapped in without changing callers
  return 0;
}

const char* respond_kernel_name(const cpir_dtc_layout&) { return "respond_kernel"; }

int launch_respond(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                   uint64_t q_slot_offset, uint32_t batch, uint32_t* r, uint32_t* /*scratch*/, hipStream_t stream,
                   bool r_is_zero, uint32_t* zero_next, uint32_t zero_count) {
  if (!dtc || !q || !r || batch == 0) return CPIR_ERR_INVALID_ARGUMENT;
  const uint32_t cf = L.compression_factor;
  if (cf != compression_factor(L.mat_elem_bit_len) || cf == 0) return CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH;
  // shape invariants every kernel variant relies on (checked on the host before any launch)
  if (L.words_per_row_padded == 0 || L.words_per_row_padded % kChunkWords != 0 || L.rows_padded % CPIR_DTC_ROW_ALIGN != 0 ||
      L.rows_padded < L.num_cols || L.words_per_row_padded < L.words_per_row ||
      L.words_per_row != (L.num_slots + cf - 1) / cf)
    return CPIR_ERR_INVALID_ARGUMENT;
  if (q_slot_offset % cf != 0 || q_slot_offset + L.num_slots > q_len) return CPIR_ERR_SHARD_RANGE;
  if (reinterpret_cast<uintptr_t>(dtc) % 16 != 0) return CPIR_ERR_INVALID_ARGUMENT;
  if (L.words_per_row_padded / kChunkWords > 0xffffffffull) return CPIR_ERR_INVALID_ARGUMENT;

  Tuning t;
  {
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    t = g_tuning;
  }
  int R = t.rows_per_unit;
  KernelFn fn = nullptr;
  if (batch == 1) {
    fn = pick_kernel_cf<1>(cf, R, t.nontemporal);
  } else if (batch == 2) {
    fn = pick_kernel_cf<2>(cf, R, t.nontemporal), R = 4;
  } else if (batch == 4) {
    fn = pick_kernel_cf<4>(cf, R, t.nontemporal), R = 4;
  } else {
    return CPIR_ERR_INVALID_ARGUMENT;  // callers split other batch sizes into 4 / 2 / 1
  }

  RespondArgs a;
  a.dtc = dtc;
  a.q = q;
  a.r = r;
  a.row_stride = L.words_per_row_padded;
  a.q_len = q_len;
  a.q_slot_offset = q_slot_offset;
  a.num_cols = L.num_cols;
  a.groups = L.rows_padded / (uint32_t)R;
  a.chunks_total = (uint32_t)(L.words_per_row_padded / kChunkWords);
  a.q_scalar = (reinterpret_cast<uintptr_t>(q) % 16 != 0 || q_slot_offset % 4 != 0 || (batch > 1 && q_len % 4 != 0)) ? 1u : 0u;

  int bpc = t.blocks_per_cu;
  if (bpc == 0) {
    int occ = 0;
    CPIR_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(fn), kThreads, 0));
    bpc = occ < 1 ? 1 : (occ > 8 ? 8 : occ);
  }
  const uint64_t units = (uint64_t)a.groups * a.chunks_total;
  uint64_t grid = (uint64_t)dev->num_cus * (uint64_t)bpc;
  a.nx = (t.xcd_split && a.chunks_total >= 8 && grid % 8 == 0) ? 8u : 1u;
  if (grid > units) {
    grid = units;
    if (a.nx == 8) grid = (grid / 8) * 8;
    if (grid == 0) grid = 1, a.nx = 1;
  }

  a.zero_next = zero_next;
  a.zero_count = zero_next ? zero_count : 0;
  if (!r_is_zero) CPIR_HIP_TRY(hipMemsetAsync(r, 0, (size_t)batch * L.num_cols * sizeof(uint32_t), stream));
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), 0, stream, a);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace cpir
