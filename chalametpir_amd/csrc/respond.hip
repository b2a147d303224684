// respond.hip -- the online hot loop: r = q . D, streamed over the packed, transposed database resident in HBM.
//
// Replaces Matrix::row_vector_x_compressed_transposed_matrix (reference chalametpir_common/src/matrix.rs:328-485), which
// the reference runs on the CPU with rayon over the C outputs (matrix.rs:345/383/429); the reference has no GPU respond.
//
//   r[c] = sum_{n < N} q[n] *wrap f(c, n),     f(c, n) = D[n][c] & (2^b - 1)        (u32, wrap-around)
//
// Roofline: HBM read.  Algorithmic bytes per query = 4*C*ceil(N/cf) + 4*N + 4*C (the REFERENCE packing).
// MFMA does not apply (u32 wrap-around is not an MFMA type) and the VALU work per streamed dword is small, so the design
// is all about the memory system of an MI355X:
//
//   * work unit = R database rows x one chunk of a row; every lane issues R x (1 or 2) independent 16-byte loads and every
//     wave-instruction reads 1 KiB contiguous (`nt` policy: the database is read once per query);
//   * the slice of q a lane needs is loaded ONCE per unit into registers and reused for the R rows, so q traffic (served by
//     L2 / Infinity Cache) is a fraction 1/R of what a row-at-a-time kernel would pull;
//   * exact u32 products in ONE VALU op each: acc64 += q*d with v_mad_u64_u32 (the low dword of the 64-bit running sum
//     is the wrap-around u32 result).  On gfx950 every VOP3 integer op issues at half rate, so this beats both
//     v_mul_lo_u32 + add and the "split q into 16-bit halves, two 16x16+32 MADs" trick (measured: scripts/valu_rate.hip);
//   * persistent grid (CUs x resident blocks), units split evenly so there is no tail wave; the chunk axis is first split
//     8 ways by blockIdx % 8 -- blocks that share an XCD (observed round-robin placement; speed only, never correctness)
//     then share one eighth of q in that XCD's 4 MiB L2;
//   * per-row partial sums leave the block through a wave shuffle + LDS reduce and one u32 atomicAdd per (block, row);
//     integer atomics make the result independent of arrival order, i.e. still bit-exact.
//
// Two packings of the fields (cpir_dtc_layout): the reference's cf fields per u32, and dense64 (K = floor(64/b) fields of b
// bits per u64, e.g. 7 instead of 6 per 8 bytes at b = 9 => 14 % fewer bytes streamed).  Rows are padded to 16 rows / whole
// chunks with zeros, so the only ragged edge left is the END OF q in the last chunk (guarded, wave-uniform path).
#include "cpir_internal.hpp"

namespace cpir {

namespace {

constexpr int kThreads = 256;

struct RespondArgs {
  const uint32_t* dtc;
  const uint32_t* q;
  uint32_t* r;
  uint64_t row_stride;     // u32 words
  uint64_t q_len;          // entries in (each) q
  uint64_t q_slot_offset;  // first global slot held by this DtC
  uint32_t num_cols;       // C: rows of DtC that produce output
  uint32_t groups;         // rows_padded / R
  uint32_t chunks_total;   // row_stride / chunk_words
  uint32_t nx;             // chunk-axis split by blockIdx % nx (8 or 1)
  uint32_t q_scalar;       // q not 16-byte loadable -> guarded scalar loads everywhere
  uint32_t passes;         // independent passes over the database in this launch; pass i uses queries [i*Q, (i+1)*Q)
  uint32_t interleave;     // 0: every block walks its own slice of the database once per pass (one query at a time
                           //    grid-wide: q stays in L2); 1: passes laid end to end in the unit space, so different blocks
                           //    stream the same rows for different queries at the same time (pays when the shard fits the
                           //    256 MiB Infinity Cache)
};

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

// acc += q * f with a 64-bit accumulator: ONE v_mad_u64_u32 per product; the low 32 bits of the running sum are the exact
// wrap-around u32 result.  Measured on MI355X (scripts/valu_rate.hip): v_mad_u64_u32 issues at 52 lanes/clk/CU, the 16x16+32
// and 24x24+32 multipliers (v_mad_u32_u16 / _u24), v_mul_lo_u32, v_dot2_u32_u16 and v_bfe_u32 all at ~60 -- every VOP3
// integer op runs at half the 128 lanes/clk/CU of the simple VOP2 ops.  Splitting q into 16-bit halves (two MADs per
// product) therefore costs 2 x 1/60 against 1/52 for the 64-bit MAD: the wide MAD is the cheaper exact product.
// (An asm statement because hipcc, seeing that only the low dword is used in the end, rewrites the C expression into
// v_mul_lo_u32 plus a 64-bit add -- two half-rate ops.  Plain VALU: no memory effects; VCC receives the unused carry.)
__device__ __forceinline__ void mac64(uint64_t& acc, uint32_t q, uint32_t f) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(q), "v"(f) : "vcc");
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- packing policies ------------------------------------------------------------------------------------------------------
// A policy says how one lane's share of a (row, chunk) unit is laid out: kLoads 16-byte pieces of the row, kNQ entries of q,
// and which q entry multiplies which field.

// The reference's words: cf fields per u32 (matrix.rs:103-167).  Lane t owns words [4t, 4t+4) of the 1024-word chunk and the
// 4*cf consecutive q entries they cover.
template <int CF>
struct RefPack {
  static constexpr int kLoads = 1;
  static constexpr int kChunkWords = 1024;
  static constexpr int kSlotsPerChunk = CF * 1024;
  static constexpr int kNQ = 4 * CF;

  // index (within this lane's kNQ entries) -> offset from the chunk's first slot
  __device__ static __forceinline__ uint32_t q_offset(int tid, int i) { return (uint32_t)(CF * 4 * tid + i); }

  __device__ static __forceinline__ uint32_t field(uint32_t w, int j) {
    constexpr int S = 32 / CF;
    // unused high bits (cf = 3: bits 30,31) and bits >= b inside a slot are zero by construction of the layout;
    // plain shift / and are VOP2 (full rate), v_bfe_u32 is VOP3 (half rate)
    if (j == CF - 1) return w >> (S * (CF - 1));
    if (j == 0) return w & ((1u << S) - 1u);
    return __builtin_amdgcn_ubfe(w, S * j, S);
  }

  template <int Q>
  __device__ static __forceinline__ void mac(const uint4 (&d)[kLoads], const uint32_t (&qv)[Q][kNQ], uint64_t (&acc)[Q]) {
    const uint32_t wd[4] = {d[0].x, d[0].y, d[0].z, d[0].w};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int j = 0; j < CF; j++) {
        const uint32_t f = field(wd[k], j);
#pragma unroll
        for (int b = 0; b < Q; b++) mac64(acc[b], qv[b][k * CF + j], f);
      }
  }
};

// dense64: K fields of exactly B bits per u64.  A chunk is 1024 u64 words; lane t owns u64 words {2t, 2t+1} of each 512-word
// half (two fully coalesced 16-byte loads) and, for every field plane j, the 4 consecutive q entries j*1024 + 4t .. +3
// (one aligned 16-byte load per plane): field j of word (L, e) multiplies q entry j*4 + 2L + e.
template <int B>
struct DensePack {
  static constexpr int K = 64 / B;
  static constexpr int kLoads = 2;
  static constexpr int kChunkWords = 2048;
  static constexpr int kSlotsPerChunk = K * 1024;
  static constexpr int kNQ = 4 * K;

  __device__ static __forceinline__ uint32_t q_offset(int tid, int i) { return (uint32_t)((i >> 2) * 1024 + 4 * tid + (i & 3)); }

  __device__ static __forceinline__ uint32_t field(uint32_t lo, uint32_t hi, int j) {
    const int o = j * B;
    if (o == 0) return lo & ((1u << B) - 1u);
    if (o + B <= 32) return (o + B == 32) ? (lo >> o) : __builtin_amdgcn_ubfe(lo, o, B);
    if (o >= 32) return (j == K - 1) ? (hi >> (o - 32)) : __builtin_amdgcn_ubfe(hi, o - 32, B);  // bits above K*B are zero
    return __builtin_amdgcn_alignbit(hi, lo, o) & ((1u << B) - 1u);                               // straddles the dword boundary
  }

  template <int Q>
  __device__ static __forceinline__ void mac(const uint4 (&d)[kLoads], const uint32_t (&qv)[Q][kNQ], uint64_t (&acc)[Q]) {
#pragma unroll
    for (int L = 0; L < 2; L++) {
      const uint32_t lo[2] = {d[L].x, d[L].z};
      const uint32_t hi[2] = {d[L].y, d[L].w};
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int j = 0; j < K; j++) {
          const uint32_t f = field(lo[e], hi[e], j);
#pragma unroll
          for (int b = 0; b < Q; b++) mac64(acc[b], qv[b][j * 4 + 2 * L + e], f);
        }
    }
  }
};

template <class P, int R, int Q, bool NT>
__global__ void __launch_bounds__(kThreads) respond_kernel(const RespondArgs a) {
  constexpr int NQ = P::kNQ;
  __shared__ uint32_t sm[Q][kThreads / 64][R];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- static partition of the (pass, row group, chunk) units over the persistent grid -------------------------------
  // One launch can carry several PASSES: independent queries that each stream the whole database, so P passes cost one
  // kernel fill/drain instead of P (worth ~10 us per query: 5 % of a 1.3 GB stream, 30 % of an eighth of it on an 8-GPU
  // shard).  Two orders, chosen by the host (a.interleave), same arithmetic:
  //   slice order:      a block keeps its slice of the units and walks it once per pass;
  //   interleaved order: the passes are laid end to end and the whole (pass, unit) space is split evenly.
  const uint32_t nx = a.nx;
  const uint32_t xcd = blockIdx.x % nx;
  const uint32_t j = blockIdx.x / nx;
  const uint32_t nb = gridDim.x / nx;  // host guarantees gridDim.x % nx == 0
  const uint32_t kb = (uint32_t)(((uint64_t)a.chunks_total * xcd) / nx);
  const uint32_t ke = (uint32_t)(((uint64_t)a.chunks_total * (xcd + 1)) / nx);
  const uint32_t span = ke - kb;
  if (span == 0) return;
  const uint64_t units = (uint64_t)a.groups * span;  // units of ONE pass in this XCD's slice of the chunk axis
  uint64_t ub, ue, u, count;                          // a pass covers units [ub, ue) for this block; u = current unit
  uint32_t pass;
  if (a.interleave) {
    const uint64_t total = units * a.passes;
    const uint64_t ib = total * j / nb, ie = total * (j + 1) / nb;
    ub = 0, ue = units, count = ie - ib;
    pass = (uint32_t)(ib / units), u = ib % units;
  } else {
    ub = units * j / nb, ue = units * (j + 1) / nb;
    count = (ue - ub) * a.passes;
    pass = 0, u = ub;
  }
  if (count == 0) return;
  const uint32_t g0 = (uint32_t)(ub / span), kc0 = kb + (uint32_t)(ub % span);  // where a pass starts for this block
  uint32_t g = (uint32_t)(u / span), kc = kb + (uint32_t)(u % span);
  const uint32_t* qpass = a.q + (uint64_t)pass * Q * a.q_len;
  uint32_t* rpass = a.r + (uint64_t)pass * Q * a.num_cols;

  uint64_t acc[R][Q];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int b = 0; b < Q; b++) acc[r][b] = 0;

  auto flush = [&](uint32_t grp) {
#pragma unroll
    for (int b = 0; b < Q; b++)
#pragma unroll
      for (int r = 0; r < R; r++) {
        const uint32_t v = wave_sum((uint32_t)acc[r][b]);  // low dword = the wrap-around u32 sum
        if (lane == 0) sm[b][wave][r] = v;
        acc[r][b] = 0;
      }
    __syncthreads();
    if (tid < Q * R) {
      const int b = tid / R, r = tid % R;
      uint32_t s = 0;
#pragma unroll
      for (int w = 0; w < kThreads / 64; w++) s += sm[b][w][r];
      const uint32_t row = grp * R + r;
      if (row < a.num_cols) atomicAdd(rpass + (uint64_t)b * a.num_cols + row, s);
    }
    __syncthreads();
  };

  const uint64_t stride16 = a.row_stride / 4;
  for (uint64_t i = 0; i < count; i++) {
    const uint64_t slot0 = a.q_slot_offset + (uint64_t)kc * P::kSlotsPerChunk;  // first slot of this chunk
    // wave-uniform: does this chunk reach past the end of q?
    const bool guarded = a.q_scalar || (slot0 + P::kSlotsPerChunk > a.q_len);

    // ---- this lane's slice of q --------------------------------------------------------------------------------------
    uint32_t qv[Q][NQ];
#pragma unroll
    for (int b = 0; b < Q; b++) {
      const uint32_t* qb = qpass + (uint64_t)b * a.q_len;
      if (!guarded) {
#pragma unroll
        for (int i = 0; i < NQ / 4; i++) {  // q_offset(tid, 4i) is a multiple of 4 for both packings
          const uint4 t = *reinterpret_cast<const uint4*>(qb + slot0 + P::q_offset(tid, 4 * i));
          qv[b][4 * i + 0] = t.x;
          qv[b][4 * i + 1] = t.y;
          qv[b][4 * i + 2] = t.z;
          qv[b][4 * i + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NQ; i++) {
          const uint64_t n = slot0 + P::q_offset(tid, i);
          qv[b][i] = (n < a.q_len) ? qb[n] : 0u;
        }
      }
    }

    // ---- the database loads of this unit: the long-latency HBM stream.  Issued AFTER the q loads on purpose: vmcnt retires
    // in issue order, so the (L2-hit) q slice is complete as soon as the first row arrives and row r's math can start
    // while rows r+1.. are still in flight --------------------------------------------------------------------------
    const uint4* p = reinterpret_cast<const uint4*>(a.dtc + (uint64_t)g * R * a.row_stride + (uint64_t)kc * P::kChunkWords) + tid;
    uint4 d[R][P::kLoads];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int l = 0; l < P::kLoads; l++) d[r][l] = load16<NT>(p + (uint64_t)r * stride16 + l * kThreads);

    // ---- multiply-accumulate ---------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < R; r++) P::template mac<Q>(d[r], qv, acc[r]);

    // ---- next unit -------------------------------------------------------------------------------------------------
    kc++, u++;
    const bool row_done = (kc == ke), pass_done = (u == ue), last = (i + 1 == count);
    if (row_done || pass_done || last) flush(g);  // the accumulators belong to (pass, row group g)
    if (row_done) g++, kc = kb;
    if (pass_done) {  // same slice of the database, next query
      u = ub, g = g0, kc = kc0;
      qpass += (uint64_t)Q * a.q_len;
      rpass += (uint64_t)Q * a.num_cols;
    }
  }
}

// ---- tuning state (benchmark harness can override; defaults chosen from measurements) ----------------
struct Tuning {
  int nontemporal = 1;     // streamed DB loads with the nt cache policy (read once per query; keeps q in L2)
  int blocks_per_cu = 2;   // resident 256-thread blocks per CU; 0 = ask the occupancy API
  int xcd_split = 1;       // split the chunk axis by blockIdx % 8
  int batch_fusion = 1;    // respond_batch: 1 = passes of 4/2/1 queries share one DB stream, 0 = one pass per query
  int interleave_passes = -1;  // order of the passes of one launch: 0 slice order, 1 interleaved, -1 by shard size
  int planar_blocks_per_cu = 0;  // the step-major matrix-core kernel: 0 = one block per CU (measured best), 1 / 2 to force
  int multi_pass_limit_mb = 2560;  // unfused batches: databases above this size get one launch per query
  int host_fill_timeout_us = 2000;  // a lone pageable host query: ONE launch polling the copy's progress, each wave for at most this long (0: off)
  int host_zero_copy = 1;          // a lone host query is read by the kernel in place (page-locked memory), not uploaded first
  int upload_streams = 2;          // concurrent host callers: their uploads take this many streams in turn (1..4)
  int inplace_seats = 4;           // concurrent page-locked host callers, up to this many at a time: ONE pass reads their queries in place (0: off)
  int helper_spin_us = 300;        // a lone pageable caller: the copy helpers keep looking for its next query this long before they sleep
  // planar packing, device-resident queries: 1 = the wide kernel takes every launch (the step-major kernel serves the in-place host path
  // only); 2 = the step-major kernel wherever it applies (passes of up to 4 queries in slice order: tests and A/B runs); 3 = as 2, launched
  // as the in-place host path launches it (strided steps, far-mode fragment schedule: diagnosis)
  int ks_major = 1;
};
Tuning g_tuning;
std::mutex g_tuning_mu;

using KernelFn = void (*)(const RespondArgs);

struct Picked {
  KernelFn fn = nullptr;
  int R = 0;
};

template <class P, int Q>
Picked pick_rows(bool nt) {
  if constexpr (Q == 1) {
    // R database rows per unit: 8 (reference packing: one load per row and lane; dense64: two, i.e. 16 loads in flight per lane) -- rounds
    // 1-2 also built R = 4 and 16 as tuning variants; 8 was the measured best at every configuration and is all that is left
    return {nt ? respond_kernel<P, 8, 1, true> : respond_kernel<P, 8, 1, false>, 8};
  } else {
    // fused batches keep Q*R accumulator pairs in registers
    constexpr int RB = (P::kLoads == 1) ? 4 : 2;
    return {nt ? respond_kernel<P, RB, Q, true> : respond_kernel<P, RB, Q, false>, RB};
  }
}

template <int Q>
Picked pick_kernel(const cpir_dtc_layout& L, bool nt) {
  if (L.packing == CPIR_PACK_REFERENCE) {
    switch (L.compression_factor) {
      case 2: return pick_rows<RefPack<2>, Q>(nt);
      case 3: return pick_rows<RefPack<3>, Q>(nt);
      default: return pick_rows<RefPack<4>, Q>(nt);
    }
  }
  switch (L.mat_elem_bit_len) {  // the bit lengths dense_fields_per_word64() offers
    case 7: return pick_rows<DensePack<7>, Q>(nt);
    case 9: return pick_rows<DensePack<9>, Q>(nt);
    case 11: return pick_rows<DensePack<11>, Q>(nt);
    case 12: return pick_rows<DensePack<12>, Q>(nt);
    default: return {};
  }
}

}  // namespace

extern "C" int cpir_tuning_set(const char* key, int value) {
  if (!key) return CPIR_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  if (!strcmp(key, "respond.nontemporal")) {
    g_tuning.nontemporal = value ? 1 : 0;
  } else if (!strcmp(key, "respond.blocks_per_cu")) {
    if (value < 0 || value > 8) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.blocks_per_cu = value;
  } else if (!strcmp(key, "respond.planar_blocks_per_cu")) {
    if (value < 0 || value > 8) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.planar_blocks_per_cu = value;
  } else if (!strcmp(key, "respond.multi_pass_limit_mb")) {
    if (value < 0) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.multi_pass_limit_mb = value;
  } else if (!strcmp(key, "respond.xcd_split")) {
    g_tuning.xcd_split = value ? 1 : 0;
  } else if (!strcmp(key, "respond.batch_fusion")) {
    g_tuning.batch_fusion = value ? 1 : 0;
  } else if (!strcmp(key, "respond.interleave_passes")) {
    if (value < -1 || value > 1) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.interleave_passes = value;
  } else if (!strcmp(key, "respond.host_fill_timeout_us")) {
    if (value < 0 || value > 1000000) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.host_fill_timeout_us = value;
  } else if (!strcmp(key, "respond.host_zero_copy")) {
    g_tuning.host_zero_copy = value ? 1 : 0;
  } else if (!strcmp(key, "respond.helper_spin_us")) {
    if (value < 0 || value > 10000) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.helper_spin_us = value;
  } else if (!strcmp(key, "respond.inplace_seats")) {
    if (value < 0 || value == 1 || value > (int)CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.inplace_seats = value;
  } else if (!strcmp(key, "respond.upload_streams")) {
    if (value < 1 || value > 4) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.upload_streams = value;
  } else if (!strcmp(key, "respond.ks_major")) {
    if (value < 1 || value > 3) return CPIR_ERR_INVALID_ARGUMENT;
    g_tuning.ks_major = value;
  } else if (!strcmp(key, "layout.dense")) {
    set_default_dense(value != 0);
  } else if (!strcmp(key, "layout.planar")) {
    set_default_planar(value != 0);
  } else if (!strcmp(key, "layout.compact_slots")) {
    if (value < 0 || value > 2) return CPIR_ERR_INVALID_ARGUMENT;
    set_compact_slots_mode(value);
  } else if (!strcmp(key, "pack.rows")) {
    if (value < -1 || value > 1) return CPIR_ERR_INVALID_ARGUMENT;
    set_pack_rows_mode(value);
  } else if (!strcmp(key, "matmul.mfma")) {
    set_mfma_matmul(value != 0);
  } else if (!strcmp(key, "matmul.pipeline")) {
    set_mfma_pipeline(value != 0);
#ifdef CPIR_DIAG
  } else if (!strcmp(key, "matmul.ablate")) {  // (a diagnosis build only: the release library has no switch that skips work)
    set_mfma_ablate(value);
#endif
  } else {
    return CPIR_ERR_INVALID_ARGUMENT;
  }
  return CPIR_OK;
}

// every key back to the default it had when the library was loaded (a test suite that flips keys must not leak them into later tests)
extern "C" void cpir_tuning_reset(void) {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  g_tuning = Tuning{};
  set_default_dense(true);
  set_default_planar(true);
  set_mfma_matmul(true);
  set_mfma_pipeline(1);
  CPIR_DIAG_ONLY(set_mfma_ablate(0);)
  set_pack_rows_mode(-1);
  set_compact_slots_mode(1);
}

uint64_t respond_multi_pass_limit_bytes() {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  return (uint64_t)g_tuning.multi_pass_limit_mb << 20;
}

bool respond_batch_fusion() {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  return g_tuning.batch_fusion != 0;
}

// queries per pass for a fused batch of `batch` queries on a planar image: as few passes as 24 queries each allow (the wide kernel), all of
// (almost) the same width -- 32 queries are two passes of 16, not 24 + 8; passes of 4 where the tuning sends fused passes to the
// step-major kernel
uint32_t respond_planar_pass_width(const cpir_dtc_layout& L, uint32_t batch) {
  Tuning t;
  {
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    t = g_tuning;
  }
  if (batch == 0) return 0;
  if (t.ks_major >= 2) return batch < CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS ? batch : CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS;
  const uint32_t passes = (batch + CPIR_PLANAR_WIDE_MAX_QUERIES_PER_PASS - 1) / CPIR_PLANAR_WIDE_MAX_QUERIES_PER_PASS;
  return (batch + passes - 1) / passes;
}

// Whether the launches respond_batched (or a lone launch_respond) makes for a batch under the current tuning can apply a slot map
// themselves: the wide kernel can (32-bit word offsets inside a row set: queries below 2^28 words), the step-major kernel and the VALU
// kernels cannot -- the caller then gathers the queries first.
bool respond_batch_takes_slot_map(const cpir_dtc_layout& L, uint32_t batch, bool lone, uint64_t q_len) {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  return L.packing == CPIR_PACK_PLANAR && g_tuning.ks_major == 1 && q_len < ((uint64_t)1 << 28) && batch != 0;
}

uint64_t respond_scratch_words(const cpir_dtc_layout&, uint32_t) {
  // the current kernel needs no scratch (partial sums leave through integer atomics); the parameter stays in the ABI so
  // a partial-buffer variant can be swapped in without changing callers
  return 0;
}

const char* respond_kernel_name(const cpir_dtc_layout& L) {
  return L.packing == CPIR_PACK_PLANAR ? "respond_planar_wide_kernel" : "respond_kernel";
}

uint32_t respond_helper_spin_us() {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  return (uint32_t)g_tuning.helper_spin_us;
}

uint32_t respond_inplace_seats() {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  return (uint32_t)g_tuning.inplace_seats;
}

uint32_t respond_upload_streams() {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  return (uint32_t)g_tuning.upload_streams;
}

uint32_t respond_host_fill_timeout_us() {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  return (uint32_t)g_tuning.host_fill_timeout_us;
}

bool respond_read_once_applicable(const cpir_dtc_layout& L) {
  std::lock_guard<std::mutex> lk(g_tuning_mu);
  // one query's responses must fit the step-major kernel's LDS accumulators (48 KiB: 12288 padded columns)
  return g_tuning.host_zero_copy != 0 && L.packing == CPIR_PACK_PLANAR && (uint64_t)(L.num_cols + 63) / 64 * 256 <= (48u << 10);
}

int launch_respond_read_once(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                             uint64_t q_slot_offset, uint32_t* r_prezeroed, hipStream_t stream, uint64_t step_lo, uint64_t step_hi,
                             const PlanarHostFill* fill) {
  if (!dtc || !q || !r_prezeroed || L.packing != CPIR_PACK_PLANAR) return CPIR_ERR_INVALID_ARGUMENT;
  CPIR_TRY(check_layout(L));
  if (q_slot_offset + L.num_slots > q_len) return CPIR_ERR_SHARD_RANGE;
  if (reinterpret_cast<uintptr_t>(dtc) % 16 != 0) return CPIR_ERR_INVALID_ARGUMENT;
  Tuning t;
  {
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    t = g_tuning;
  }
  return launch_respond_planar_ks(dev, dtc, L, q, q_len, q_slot_offset, 1, 1, r_prezeroed, stream, t.planar_blocks_per_cu, t.nontemporal != 0,
                                  t.xcd_split != 0, true, true, step_lo, step_hi, fill);
}

int launch_respond_read_rows_in_place(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* const* q_rows, uint32_t batch,
                                      uint64_t q_len, uint64_t q_slot_offset, uint32_t* r, hipStream_t stream, const PlanarHostFill* fill,
                                      bool r_prezeroed) {
  if (!dtc || !q_rows || !r || L.packing != CPIR_PACK_PLANAR) return CPIR_ERR_INVALID_ARGUMENT;
  if (fill && fill->seats != batch) return CPIR_ERR_INVALID_ARGUMENT;  // (every query of the pass has a count of its own, complete from the start or not)
  CPIR_TRY(check_layout(L));
  if (q_slot_offset + L.num_slots > q_len) return CPIR_ERR_SHARD_RANGE;
  if (reinterpret_cast<uintptr_t>(dtc) % 16 != 0) return CPIR_ERR_INVALID_ARGUMENT;
  Tuning t;
  {
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    t = g_tuning;
  }
  // (launched as a lone host query is: whole steps round-robin over the blocks, so that the queries are consumed front to back -- they may
  // still be arriving -- and the far-mode fragment schedule; measured the same as contiguous units where they are complete, 228 vs 231 us
  // for two queries at 2^20 keys x 1 kB, scripts/probes/inplace_batch.py)
  return launch_respond_planar_ks(dev, dtc, L, nullptr, q_len, q_slot_offset, batch, 1, r, stream, t.planar_blocks_per_cu, t.nontemporal != 0,
                                  t.xcd_split != 0, true, r_prezeroed, 0, 0, fill, q_rows);
}

int launch_respond(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                   uint64_t q_slot_offset, uint32_t batch, uint32_t passes, uint32_t* r, uint32_t* /*scratch*/, hipStream_t stream,
                   const uint32_t* keep) {
  if (!dtc || !q || !r || batch == 0 || passes == 0) return CPIR_ERR_INVALID_ARGUMENT;
  // shape invariants every kernel variant relies on (checked on the host before any launch)
  CPIR_TRY(check_layout(L));
  if (q_slot_offset + L.num_slots > q_len) return CPIR_ERR_SHARD_RANGE;  // the shard's slots must lie inside the query
  if (reinterpret_cast<uintptr_t>(dtc) % 16 != 0) return CPIR_ERR_INVALID_ARGUMENT;

  Tuning t;
  {
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    t = g_tuning;
  }
  if (L.packing == CPIR_PACK_PLANAR) {  // the matrix-core path (respond_planar.hip)
    // the step-major kernel where the tuning asks for it and it applies (one row set, slice order, no slot map); else the wide kernel:
    // 1 .. 24 queries per pass, any number of passes in either order.  Measured at 2^20 keys x 1 kB, 32 passes of one query a launch
    // (scripts/families_ab.py): tile-major kernel of rounds 1-4 / wide, slice order 184.8-185.1 / 181.2-184.1 us per query; interleaved
    // on 1/8 of the slots 11.1-11.4 / 8.2, on 1/2 50.0 / 34.3
    const bool ks = t.ks_major >= 2 && batch <= CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS && !keep && !planar_passes_interleaved(L, passes, t.interleave_passes);
    if (ks) {
      const int st = launch_respond_planar_ks(dev, dtc, L, q, q_len, q_slot_offset, batch, passes, r, stream, t.planar_blocks_per_cu,
                                              t.nontemporal != 0, t.xcd_split != 0, t.ks_major == 3, false, 0, 0, nullptr);
      if (st != CPIR_ERR_INVALID_ARGUMENT || t.ks_major != 3) return st;  // (in-place order: one column window or nothing -- the wide kernel then)
    }
    return launch_respond_planar_wide(dev, dtc, L, q, q_len, q_slot_offset, batch, passes, r, stream, t.nontemporal != 0, t.xcd_split != 0,
                                      t.interleave_passes, keep);
  }
  if (keep) return CPIR_ERR_INVALID_ARGUMENT;  // only the wide kernel applies a slot map itself: the caller gathers the queries first
  if (L.words_per_row_padded / L.chunk_words > 0xffffffffull) return CPIR_ERR_INVALID_ARGUMENT;
  Picked k;
  if (batch == 1) k = pick_kernel<1>(L, t.nontemporal);
  else if (batch == 2) k = pick_kernel<2>(L, t.nontemporal);
  else if (batch == 4) k = pick_kernel<4>(L, t.nontemporal);
  else return CPIR_ERR_INVALID_ARGUMENT;  // callers split other batch sizes into 4 / 2 / 1
  if (!k.fn) return CPIR_ERR_INVALID_ARGUMENT;

  RespondArgs a;
  a.dtc = dtc;
  a.q = q;
  a.r = r;
  a.row_stride = L.words_per_row_padded;
  a.q_len = q_len;
  a.q_slot_offset = q_slot_offset;
  a.num_cols = L.num_cols;
  a.groups = L.rows_padded / (uint32_t)k.R;
  a.chunks_total = (uint32_t)(L.words_per_row_padded / L.chunk_words);
  a.q_scalar = (reinterpret_cast<uintptr_t>(q) % 16 != 0 || q_slot_offset % 4 != 0 || (batch * passes > 1 && q_len % 4 != 0)) ? 1u : 0u;
  a.passes = passes;
  // Order of the passes, from measurements on MI355X: below ~1 GB per pass the interleaved order wins
  // (concurrent passes share database bytes on die: 23 vs 31 us per query on a 160 MB shard, 92 vs 104 us at 640 MB);
  // at 1.3 GB and above the slice order wins (196 vs 198 us)
  a.interleave = (passes > 1 && (t.interleave_passes == 1 || (t.interleave_passes < 0 && L.total_words * 4 <= (960ull << 20)))) ? 1u : 0u;

  int bpc = t.blocks_per_cu;
  if (bpc == 0) {
    int occ = 0;
    CPIR_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(k.fn), kThreads, 0));
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

  CPIR_TRY(zero_words(r, (uint64_t)batch * passes * L.num_cols, stream));
  hipLaunchKernelGGL(k.fn, dim3((unsigned)grid), dim3(kThreads), 0, stream, a);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace cpir
