// respond_planar.hip -- the online hot loop on the i8 matrix cores, for the planar packing (CPIR_PACK_PLANAR).
//
// Same contraction as respond.hip (reference chalametpir_common/src/matrix.rs:328-485):
//
//   r[c] = sum_{n < N} q[n] *wrap f(c, n)      (u32, wrap-around)
//
// u32 wrap-around products are not an MFMA type, but the sum splits exactly into signed-byte products, which are
// (v_mfma_i32_16x16x64_i8; its i32 accumulators wrap, scripts/mfma_i8_probe.hip):
//
//   q[n]   = sum_{i<4} 2^(8i) * (qs_i[n] + 128)          qs_i = byte i of q[n], XOR 0x80, read as a signed byte
//   f(c,n) = ls + 128 + 256 * h                          ls = low byte of f, XOR 0x80, signed;  h = f >> 8  (0 .. 2^(b-8) - 1)
//
//   sum_n q f = sum_i 2^(8i) * [ sum_n qs_i * ls  +  256 * sum_n qs_i * h ]      <- two MFMAs per 64 slots x 16 columns
//             + 128 * sum_n (q[n] - 0x80808080)                                  <- per query, same for every column
//             + 0x80808080 * sum_n f(c,n)                                        <- per column, stored behind the tiles
//
// all mod 2^32.  The matrix cores do the 1.1 * 10^9 products per query that cost the VALU kernel ~65 us of issue time; what is
// left for the VALU is turning the high bit planes into bytes (two VOP2 ops per 4 fields) and gathering the query bytes.
// The A operand rows are (query, byte): 16 rows = up to 4 queries answered by ONE stream of the database.
//
// Roofline: HBM read, exactly b bits per field (1.6 % fewer bytes than dense64 at b = 9, 15.6 % fewer than the reference packing).
//
//   * work unit of a block = 4 column tiles (64 columns, one per wave) x one super-tile step (512 slots); the step's A fragments
//     (8 KiB) are built once per block from q (L2 / Infinity Cache hits; each wave gathers two of the eight k-blocks), shared
//     through double-buffered LDS and read back just in time, one ds_read_b128 per MFMA pair;
//   * a wave's tile step is 8 + HB fully coalesced 1 KiB wave-loads (`nt`), issued one step ahead of the MFMAs that consume them;
//   * persistent grid, units split evenly over the blocks, one barrier per step; the slot axis is first split 8 ways by
//     blockIdx % 8 as in respond.hip so that an XCD's L2 holds one eighth of q;
//   * the accumulator tile has the column on the lane and (query, byte) in the register index, so the recombination
//     sum_i 2^(8i) (lo_i + 256 hi_i) is lane-local; one u32 atomicAdd per (wave, query, column) and tile group.
#include "cpir_internal.hpp"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace cpir {
namespace {

constexpr int kThreads = 256;
constexpr int kM = 4;  // column tiles per work unit = waves per block

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct PlanarArgs {
  const uint32_t* dtc;
  const uint32_t* q;
  uint32_t* r;
  uint64_t q_len;          // entries in each query
  uint64_t q_slot_offset;  // first global slot held by this DtC
  uint64_t num_slots;      // slots held by this DtC (N of the shard)
  uint32_t num_cols;       // C
  uint32_t col_tiles;      // rows_padded / 16
  uint32_t tile_groups;    // ceil(col_tiles / kM)
  uint32_t tg_lo, tg_n;    // step-major kernel: this launch covers tile groups [tg_lo, tg_lo + tg_n) only (a column window: its LDS accumulators
                           // hold tg_n * 64 columns per query; wide databases answer a fused batch window by window)
  uint32_t ks_total;       // super-tile steps along the slots: ceil(N / 512)
  uint32_t ks_lo, ks_hi;   // this launch covers steps [ks_lo, ks_hi) only (the whole axis unless a host query is being pipelined)
  uint32_t nx;             // slot-axis split by blockIdx % nx (8 or 1)
  uint32_t q_per_pass;     // queries answered per pass (1..8): rows 4*i .. 4*i+3 of A row set i / 4 belong to query i
  uint32_t passes;         // independent passes over the database in this launch
  uint32_t q_scalar;       // q not 16-byte loadable -> guarded scalar loads everywhere
  uint32_t interleave;     // order of the passes of one launch (see the kernel)
  uint32_t q_far;          // step-major kernel: q sits behind the host link -> a whole step of units between requesting and using it
  const uint32_t* colsum;  // step-major kernel: per-column field sums behind the tiles (NULL: this launch does not cover step 0)
  // step-major kernel, q read in place from host memory that is still being FILLED while the kernel runs (a lone pageable host query):
  uint32_t strided;          // blocks take whole steps round-robin (step s -> block s % blocks), so the grid consumes q front to back
  const uint32_t* progress;  // host memory: number of 512-slot steps of q in place so far, CPIR_FILL_LINES copies 64 bytes apart (NULL: all of q is in place)
  uint32_t* abort_flag;      // device memory: set when a wave has given up waiting (the launch's results are then void)
  uint64_t poll_ticks;       // ... after this many ticks of the 100 MHz wall clock
  const uint32_t* keep;      // wide pass: the database holds only the slots keep[0 .. num_slots) of the query (increasing, relative to q_slot_offset;
                             // compact.hip); NULL: slot n of the database is word q_slot_offset + n of the query
  uint64_t* trace;           // step-major kernel, diagnosis only (CPIR_KS_TRACE): per block 4 words -- wall clock at entry, after the first fragments, at the end; visits
  uint32_t ablate;           // wide pass, diagnosis only (CPIR_WIDE_ABLATE; results are WRONG while non-zero): 1 no rebuild of the fragments, 2 no flush, 4 no MFMAs
};

template <bool NT>
__device__ __forceinline__ uint4 load16(const uint4* p) {
  if constexpr (NT) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}

__device__ __forceinline__ v4i as_v4i(const uint4& u) { return v4i{(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }

// byte `limb` of four consecutive query words -> one dword (k order = word order); sel01 = limb | (4 + limb) << 8
__device__ __forceinline__ uint32_t gather_limb(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t sel01) {
  const uint32_t p01 = __builtin_amdgcn_perm(y, x, sel01);  // byte 0 = x.byte[limb], byte 1 = y.byte[limb]
  const uint32_t p23 = __builtin_amdgcn_perm(w, z, sel01);
  return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

__device__ __forceinline__ uint32_t comp(const uint4& u, int i) { return i == 0 ? u.x : (i == 1 ? u.y : (i == 2 ? u.z : u.w)); }

// One step of a block: wave w multiplies column tile 4*tg + w by the step's A fragments (shared through LDS), with the loads of
// its NEXT tile and the block's next A fragments issued first.  P = parity of the step (register / LDS double buffering).
// NS = sets of 16 A rows: one set answers up to 4 queries per pass, two sets up to 8 (twice the MFMAs on the same stream); the step-major
// kernel below also runs with three (up to 12 queries per pass).
template <int HB, int NS, bool NT>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(NS == 1 ? 3 : 2, NS == 1 ? 3 : 2)))
respond_planar_kernel(const PlanarArgs a) {
  constexpr int NL = 8 + HB;     // 16-byte loads per lane and tile step
  constexpr int ST16 = NL * 64;  // uint4 per super-tile
  __shared__ uint4 abuf[2][NS][8][64];  // A fragments of a step: [parity][row set][k-block][lane]

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint32_t cl = lane & 15;   // B / C operand: column inside the tile;  A operand: row = 4 * query + byte
  const uint32_t grp = lane >> 4;  // A / B operand: 16-slot group inside a k-block;  C operand: query (rows 4*grp .. 4*grp+3)

  // ---- static partition of the (tile group, step) units over the BLOCKS of the persistent grid (block-uniform) -------
  const uint32_t nx = a.nx;
  const uint32_t xcd = blockIdx.x % nx;
  const uint32_t j = blockIdx.x / nx;
  const uint32_t nb = gridDim.x / nx;  // host guarantees gridDim.x % nx == 0
  const uint32_t ks_len = a.ks_hi - a.ks_lo;
  const uint32_t kb0 = a.ks_lo + (uint32_t)(((uint64_t)ks_len * xcd) / nx);
  const uint32_t ke0 = a.ks_lo + (uint32_t)(((uint64_t)ks_len * (xcd + 1)) / nx);
  const uint32_t span = ke0 - kb0;
  if (span == 0) return;
  const uint64_t units = (uint64_t)a.tile_groups * span;
  // Two orders of the passes of one launch (same arithmetic; as respond.hip):
  //   slice order:       a block keeps its slice [sb, se) of the units and walks it once per pass;
  //   interleaved order: the passes are laid end to end and the whole (pass, unit) space is split evenly, so different blocks
  //                      stream the same tiles for different queries at about the same time (pays when the shard fits the
  //                      256 MiB Infinity Cache).
  uint64_t sb, se, u0, total;  // a pass covers units [sb, se) for this block; it starts at unit u0 of pass pass0 and does `total` steps
  uint32_t pass0;
  if (a.interleave) {
    const uint64_t all = units * a.passes;
    const uint64_t ib = all * j / nb, ie = all * (j + 1) / nb;
    sb = 0, se = units, total = ie - ib;
    pass0 = (uint32_t)(ib / units), u0 = ib % units;
  } else {
    sb = units * j / nb, se = units * (j + 1) / nb;
    total = (se - sb) * a.passes;
    pass0 = 0, u0 = sb;
  }
  if (total == 0) return;
  const uint32_t tgs = (uint32_t)(sb / span), kss = kb0 + (uint32_t)(sb % span);  // where a pass starts for this block
  const uint32_t tg0 = (uint32_t)(u0 / span), ks0 = kb0 + (uint32_t)(u0 % span);  // where this block starts

  const uint32_t nq = a.q_per_pass;
  bool arow[NS];     // does this lane's A row of set s belong to a query?
  uint32_t qi[NS];   // which one (query 0 for unused rows: they load the same cache lines and store zeros)
#pragma unroll
  for (int s = 0; s < NS; s++) {
    arow[s] = 4 * s + (cl >> 2) < nq;
    qi[s] = arow[s] ? 4 * s + (cl >> 2) : 0;
  }
  const uint32_t limb = cl & 3;
  const uint32_t sel01 = limb | ((4 + limb) << 8);
  const uint4* const tiles = reinterpret_cast<const uint4*>(a.dtc);

  // (tile group, step, pass) of the current step and of the next one
  uint32_t tg = tg0, ks = ks0, pass = pass0;
  uint64_t u = u0;  // unit inside the pass

  // ---- A fragments of a step -> abuf[par]: wave w builds k-blocks 2w and 2w+1 ---------------------------------------------
  // Split in two so that the (L2-hit) query loads are ISSUED before the step's database loads and CONSUMED after the MFMAs of the
  // previous step: vmcnt retires in issue order, so waiting for the query words never waits for the HBM stream behind them.
  // Every lane loads (lanes whose A row is unused read query 0's words again -- same cache lines -- and store zeros): no
  // divergent branch around the loads.
  auto guarded_step = [&](uint32_t ks_) {  // block-uniform: does this step reach past the end of the shard / of q, or is q unaligned?
    const uint64_t slot0 = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE;
    return a.q_scalar || slot0 + CPIR_PLANAR_SLOTS_PER_TILE > a.num_slots ||
           a.q_slot_offset + slot0 + CPIR_PLANAR_SLOTS_PER_TILE > a.q_len;
  };
  auto a_issue = [&](uint4(&raw)[NS][2][4], uint32_t ks_, uint32_t pass_) {
    const uint64_t base = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE + (2 * wave) * 64 + grp * 16;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const uint32_t* qrow = a.q + ((uint64_t)pass_ * nq + qi[s]) * a.q_len + a.q_slot_offset;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint4* src = reinterpret_cast<const uint4*>(qrow + base + h * 64);
#pragma unroll
        for (int d = 0; d < 4; d++) raw[s][h][d] = src[d];
      }
    }
  };
  auto a_finish = [&](const uint4(&raw)[NS][2][4], int par) {
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        uint32_t o[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint4& t = raw[s][h][d];
          const uint32_t v = gather_limb(t.x, t.y, t.z, t.w, sel01) ^ 0x80808080u;
          o[d] = arow[s] ? v : 0u;
        }
        abuf[par][s][2 * wave + h][lane] = make_uint4(o[0], o[1], o[2], o[3]);
      }
  };
  // the rare guarded step: scalar, bounds-checked loads.  Slots past the end of the shard or of the query take part with
  // qs = 0 (byte 0x00), exactly as planar_init_kernel counts them.
  auto a_guarded = [&](uint32_t ks_, uint32_t pass_, int par) {
    const uint64_t slot0 = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE;
    for (int s = 0; s < NS; s++) {
      const uint32_t* qrow = a.q + ((uint64_t)pass_ * nq + qi[s]) * a.q_len + a.q_slot_offset;
      for (int h = 0; h < 2; h++) {
        const int kb = 2 * wave + h;
        const uint64_t base = slot0 + kb * 64 + grp * 16;
        uint32_t o[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          uint32_t w[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const uint64_t n = base + d * 4 + e;
            const bool ok = n < a.num_slots && a.q_slot_offset + n < a.q_len;
            w[e] = ok ? (qrow[n] ^ 0x80808080u) : 0u;
          }
          o[d] = arow[s] ? gather_limb(w[0], w[1], w[2], w[3], sel01) : 0u;
        }
        abuf[par][s][kb][lane] = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
  };

  v4i acc_lo[NS], acc_hi[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) acc_lo[s] = v4i{0, 0, 0, 0}, acc_hi[s] = v4i{0, 0, 0, 0};
  uint4 b0[NL], b1[NL];

  auto load_tile = [&](uint4(&dst)[NL], uint32_t tg_, uint32_t ks_) {
    // the idle wave of a ragged last tile group re-reads the first super-tile of the image (L2-hot) instead of branching
    // around the loads; its MFMAs and its flush are skipped
    const uint32_t T = tg_ * kM + wave;
    const uint4* p = tiles + (T < a.col_tiles ? ((uint64_t)T * a.ks_total + ks_) * ST16 : 0) + lane;
#pragma unroll
    for (int i = 0; i < NL; i++) dst[i] = load16<NT>(p + i * 64);
  };

  auto flush = [&](uint32_t tg_, uint32_t pass_) {
    const uint32_t T = tg_ * kM + wave;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      if (T < a.col_tiles) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) v += ((uint32_t)acc_lo[s][i] + ((uint32_t)acc_hi[s][i] << 8)) << (8 * i);
        const uint32_t col = T * 16 + cl, query = 4 * s + grp;
        if (query < nq && col < a.num_cols) atomicAdd(a.r + ((uint64_t)pass_ * nq + query) * a.num_cols + col, v);
      }
      acc_lo[s] = v4i{0, 0, 0, 0}, acc_hi[s] = v4i{0, 0, 0, 0};
    }
  };

  auto step = [&](uint4(&cur)[NL], uint4(&nxt)[NL], int par, bool last) {
    // where the next step is
    uint32_t tg_n = tg, ks_n = ks + 1, pass_n = pass;
    const bool pass_end = (u + 1 == se);
    if (pass_end) tg_n = tgs, ks_n = kss, pass_n = pass + 1;
    else if (ks_n == ke0) ks_n = kb0, tg_n = tg + 1;
    uint4 raw[NS][2][4];
    const bool g_n = guarded_step(ks_n);
    if (!last) {
      if (!g_n) a_issue(raw, ks_n, pass_n);
      load_tile(nxt, tg_n, ks_n);
    }
    if (tg * kM + wave < a.col_tiles) {
#pragma unroll
      for (int kb = 0; kb < 8; kb++) {
        v4i hb;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          uint32_t x = 0;
#pragma unroll
          for (int p = 0; p < HB; p++) x += ((comp(cur[8 + p], kb >> 1) >> (4 * (kb & 1) + d)) & 0x01010101u) << p;
          hb[d] = (int)x;
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
          const uint4 au = abuf[par][s][kb][lane];
          const v4i af = as_v4i(au);
          acc_lo[s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, as_v4i(cur[kb]), acc_lo[s], 0, 0, 0);
          if constexpr (HB > 0) acc_hi[s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, hb, acc_hi[s], 0, 0, 0);
        }
      }
    }
    if (!last) {
      if (!g_n) a_finish(raw, par ^ 1);
      else a_guarded(ks_n, pass_n, par ^ 1);
    }
    if (pass_end || tg_n != tg || last) flush(tg, pass);
    __syncthreads();  // A fragments of the next step are in LDS; everybody is done with this step's
    tg = tg_n, ks = ks_n, pass = pass_n;
    u = pass_end ? sb : u + 1;
  };

  // prologue: first tile and first A fragments
  load_tile(b0, tg0, ks0);
  a_guarded(ks0, pass0, 0);
  __syncthreads();
  uint64_t i = 0;
  for (; i + 2 <= total; i += 2) {
    step(b0, b1, 0, false);
    step(b1, b0, 1, i + 2 == total);
  }
  if (i < total) step(b0, b1, 0, true);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same contraction walked STEP-MAJOR: a block owns a contiguous range of (step, tile group) units ordered by step first, so
//   * the A fragments of a 512-slot step are built ONCE per block and step and serve every tile group of that step (the kernel above
//     rebuilds them for every unit: at 2^20 keys each word of q is gathered 15 times, with 8 kB values 115 times) -- every word of q is
//     read by (almost) exactly one block, which is what lets a lone host query be read straight from page-locked HOST memory (zero-copy:
//     the 83 us upload of a 4.7 MB query disappears behind the 190 us stream instead of preceding it);
//   * one barrier per STEP instead of one per unit (the waves of a block run free between the steps);
//   * the responses accumulate in LDS (one u32 per query and padded column) and leave through one pass of u32 atomics per block;
//   * the two correction terms of the signed-byte split are added here, not by planar_init_kernel (which would read q a second time):
//     while a wave gathers its share of a step's query words it also sums them, and every (step, tile group) unit adds
//     128 * (sum of the step's valid query words) - 0x40404000 * (valid slots of the step) to each of its 64 columns -- every unit
//     is visited exactly once, so every column receives the whole per-query term; the units of step 0 add 0x80808080 * colsum[column].
// Same arithmetic, same packed image, same results bit for bit.  Used for the slice pass order (every pass its own stream from HBM);
// the interleaved order of the multi-GPU shards stays on the kernel above.
template <int HB, int NS, bool NT>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))  // two blocks a CU is the grid; at 3 it spills
respond_planar_ks_kernel(const PlanarArgs a) {
  constexpr int NL = 8 + HB;     // 16-byte loads per lane and tile step
  constexpr int ST16 = NL * 64;  // uint4 per super-tile
  __shared__ uint4 abuf[2][NS][8][64];     // A fragments of a step: [parity][row set][k-block][lane]
  __shared__ uint32_t ksum[2][kThreads / 64][4 * NS];  // per step parity, wave and query: the sum of the query words that wave gathered
  extern __shared__ uint32_t racc[];       // [query of the pass][padded column]: this block's part of the responses

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint32_t cl = lane & 15;
  const uint32_t grp = lane >> 4;
  const uint32_t cpad = a.tg_n * (kM * 16);  // columns of this launch's window (padded to whole tile groups)

  const uint32_t nx = a.nx;
  const uint32_t xcd = blockIdx.x % nx;
  const uint32_t j = blockIdx.x / nx;
  const uint32_t nb = gridDim.x / nx;  // host guarantees gridDim.x % nx == 0
  const uint32_t ks_len = a.ks_hi - a.ks_lo;
  const uint32_t kb0 = a.ks_lo + (uint32_t)(((uint64_t)ks_len * xcd) / nx);
  const uint32_t ke0 = a.ks_lo + (uint32_t)(((uint64_t)ks_len * (xcd + 1)) / nx);
  const uint32_t span = ke0 - kb0;
  const uint32_t TG = a.tg_n;  // tile groups of the window, numbered from 0 here; tile group tg of the window is a.tg_lo + tg of the image
  // A block's work is a sequence of VISITS: (step, first tile group, one past the last tile group).
  //   contiguous order: units u = (step kb0 + u / TG, tile group u % TG) of [0, TG * span) split evenly over the blocks;
  //   strided order (nx == 1): `rounds` whole steps per block, step kb0 + r * nb + j in round r -- at any moment the grid works on nb
  //   neighbouring steps, so q is consumed front to back -- and the units of the span % nb steps left over split evenly as above.
  const uint32_t rounds = a.strided ? span / nb : 0;
  const uint32_t tail0 = kb0 + rounds * nb;  // first step of the evenly split part
  const uint64_t units = (uint64_t)TG * (ke0 - tail0);
  const uint64_t sb = units * j / nb, se = units * (j + 1) / nb;
  const uint32_t tail_visits = se > sb ? (uint32_t)((se - 1) / TG - sb / TG + 1) : 0;
  const uint32_t n_visits = rounds + tail_visits;
  if (n_visits == 0) return;  // block-uniform: an idle block takes part in nothing
  if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 0] = wall_clock64(), a.trace[blockIdx.x * 4 + 3] = n_visits;
  // (the 64-bit divisions once, not per unit; plain scalars, no structs: selecting between structs captured by reference sends them
  // through scratch memory)
  const uint32_t tail_ks = tail0 + (uint32_t)(sb / TG), tail_tg0 = (uint32_t)(sb % TG);
  const uint32_t tail_tg1 = se > sb ? (uint32_t)((se - 1) % TG) + 1 : TG;
  auto visit_ks = [=](uint32_t v) { return v < rounds ? kb0 + v * nb + j : tail_ks + (v - rounds); };
  auto visit_tg0 = [=](uint32_t v) { return v == rounds ? tail_tg0 : 0u; };
  auto visit_tg1 = [=](uint32_t v) { return v + 1 == n_visits && tail_visits ? tail_tg1 : TG; };

  const uint32_t nq = a.q_per_pass;
  bool arow[NS];
  uint32_t qi[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    arow[s] = 4 * s + (cl >> 2) < nq;
    qi[s] = arow[s] ? 4 * s + (cl >> 2) : 0;
  }
  const uint32_t limb = cl & 3;
  const uint32_t sel01 = limb | ((4 + limb) << 8);
  const uint4* const tiles = reinterpret_cast<const uint4*>(a.dtc);

  auto guarded_step = [&](uint32_t ks_) __attribute__((always_inline)) {
    const uint64_t slot0 = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE;
    return a.q_scalar || slot0 + CPIR_PLANAR_SLOTS_PER_TILE > a.num_slots || a.q_slot_offset + slot0 + CPIR_PLANAR_SLOTS_PER_TILE > a.q_len;
  };
  // q still being filled by the host (a.progress): wait until step ks_ is in place.  Every wave polls for itself (one request per wave and
  // poll, a couple of microseconds each over the link); a wave that has waited poll_ticks gives up FOR GOOD, raises the abort flag and
  // carries on with whatever it reads -- no wave ever leaves the common control flow, so the grid always drains; the host discards the
  // results of an aborted launch.
  // The count is published in CPIR_FILL_LINES copies, one per 64-byte line, and a block polls line (block % lines): uncached reads of ONE
  // host line are served one after the other (measured: ~170 ns each, i.e. 1024 polling waves get an answer every 174 us and the kernel
  // took 1.6 ms with everything in place); 16 waves per line are answered within 3 us.  The last count seen is kept: it only grows.
  bool gave_up = false;
  uint32_t seen = 0;
  const uint32_t* const my_progress = a.progress ? a.progress + (blockIdx.x % CPIR_FILL_LINES) * 16 : nullptr;
  auto wait_for_step = [&](uint32_t ks_) __attribute__((always_inline)) {
    if (!my_progress || gave_up || seen > ks_) return;
    const uint64_t t0 = wall_clock64();
    // RELAXED system-scope loads (they bypass the caches by themselves): an acquire would invalidate the L2 under the whole grid's feet
    // at every poll.  Nothing needs it: the words waited for are fetched only after the loop has seen the count (control dependency),
    // from host memory that this kernel has not touched before, and the host publishes the count with a release store after the copy.
    while ((seen = __hip_atomic_load(my_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) <= ks_) {
      if (wall_clock64() - t0 > a.poll_ticks) {
        gave_up = true;
        if (lane == 0) atomicOr(a.abort_flag, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(64);  // ~2 us between polls of a wave: a poll is a 64-byte read over the host link
    }
    // the query words of the step are fetched AFTER this point, in program order: nothing may be hoisted above the loop (the control
    // dependency already orders the hardware's requests; this pins the compiler).  Wavefront scope: no cache maintenance is emitted.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const uint32_t half = lane >> 5, l32 = lane & 31;
  auto a_issue = [&](uint4(&raw)[2 * NS], uint32_t ks_, uint32_t pass_) __attribute__((always_inline)) {
    const uint64_t base = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE + wave * 128 + l32 * 4;
#pragma unroll
    for (int i = 0; i < 2 * NS; i++) {
      const uint32_t row = 2 * i + half;
      raw[i] = make_uint4(0, 0, 0, 0);
      if (row < nq) raw[i] = *reinterpret_cast<const uint4*>(a.q + ((uint64_t)pass_ * nq + row) * a.q_len + a.q_slot_offset + base);
    }
  };
  // this wave's sum of the query words of a step, per query: the four lanes that share a query word (one per byte limb) count it once
  // (limb 0), the four 16-slot groups are added up with two shuffles, lane (group 0, limb 0) of every query writes
  auto store_ksum = [&](const uint32_t(&part)[NS], int par) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      uint32_t v = part[s];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (grp == 0 && limb == 0) ksum[par][wave][4 * s + (cl >> 2)] = arow[s] ? v : 0u;
    }
  };
  // A wave loads its 128 slots of every query row of a step with ONE fully coalesced 16-byte load per two rows (512 contiguous bytes a
  // row: whole lines, each requested once -- which matters when q sits in host memory behind the link).  To reach the fragment order it
  // parks the four rows of a row set in the 2 KiB of abuf it is about to fill (its own two k-blocks of that row set), reads them back
  // lane by lane, and then overwrites them with the fragments: LDS operations of one wave complete in order, nobody else touches the
  // slice before the step's barrier, and the fences keep the compiler from reordering the three phases.
  auto wave_lds_fence = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto a_finish = [&](const uint4(&raw)[2 * NS], int par) __attribute__((always_inline)) {
    uint32_t part[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
      uint32_t* const stage = reinterpret_cast<uint32_t*>(&abuf[par][s][2 * wave][0]);  // 4 rows x 128 words
      *reinterpret_cast<uint4*>(stage + (0 + half) * 128 + l32 * 4) = raw[2 * s];
      *reinterpret_cast<uint4*>(stage + (2 + half) * 128 + l32 * 4) = raw[2 * s + 1];
      wave_lds_fence();
      uint4 back[2][4];
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int d = 0; d < 4; d++) back[h][d] = *reinterpret_cast<const uint4*>(stage + (cl >> 2) * 128 + h * 64 + grp * 16 + d * 4);
      wave_lds_fence();
      part[s] = 0;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        uint32_t o[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint4 t = back[h][d];
          part[s] += (t.x + t.y) + (t.z + t.w);
          const uint32_t v = gather_limb(t.x, t.y, t.z, t.w, sel01) ^ 0x80808080u;
          o[d] = arow[s] ? v : 0u;
        }
        abuf[par][s][2 * wave + h][lane] = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
    store_ksum(part, par);
  };
  auto a_guarded = [&](uint32_t ks_, uint32_t pass_, int par) __attribute__((always_inline)) {
    const uint64_t slot0 = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE;
    uint32_t part[NS];
    for (int s = 0; s < NS; s++) {
      part[s] = 0;
      const uint32_t* qrow = a.q + ((uint64_t)pass_ * nq + qi[s]) * a.q_len + a.q_slot_offset;
      for (int h = 0; h < 2; h++) {
        const int kb = 2 * wave + h;
        const uint64_t base = slot0 + kb * 64 + grp * 16;
        uint32_t o[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          uint32_t w[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const uint64_t n = base + d * 4 + e;
            const bool ok = n < a.num_slots && a.q_slot_offset + n < a.q_len;
            const uint32_t x = ok ? qrow[n] : 0u;
            part[s] += x;
            w[e] = ok ? (x ^ 0x80808080u) : 0u;
          }
          o[d] = arow[s] ? gather_limb(w[0], w[1], w[2], w[3], sel01) : 0u;
        }
        abuf[par][s][kb][lane] = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
    store_ksum(part, par);
  };
  auto load_tile = [&](uint4(&dst)[NL], uint32_t tg_, uint32_t ks_) __attribute__((always_inline)) {
    const uint32_t T = (a.tg_lo + tg_) * kM + wave;
    const uint4* p = tiles + (T < a.col_tiles ? ((uint64_t)T * a.ks_total + ks_) * ST16 : 0) + lane;
#pragma unroll
    for (int i = 0; i < NL; i++) dst[i] = load16<NT>(p + i * 64);
  };
  // valid slots of a step: inside the shard and inside the query
  const uint64_t room = a.q_len > a.q_slot_offset ? a.q_len - a.q_slot_offset : 0;
  const uint64_t nvalid_total = a.num_slots < room ? a.num_slots : room;
  auto valid_slots = [&](uint32_t ks_) __attribute__((always_inline)) -> uint32_t {
    const uint64_t lo = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE;
    if (lo >= nvalid_total) return 0u;
    const uint64_t left = nvalid_total - lo;
    return left < CPIR_PLANAR_SLOTS_PER_TILE ? (uint32_t)left : CPIR_PLANAR_SLOTS_PER_TILE;
  };

  // ---- the passes of the launch as ONE pipeline --------------------------------------------------------------------------------------
  // A pass ends where the next begins: the last unit of a pass prefetches the next pass's first tile (the same tile -- every pass walks the
  // same visits) and its last visit builds the next pass's first fragments, exactly as for the next visit inside a pass; the pass's
  // responses leave at the boundary, between two barriers, while those requests are in flight (round 3 drained the pipeline, flushed and
  // ran a prologue for every pass: 26.4 us per query for passes of 8 at 2^20 keys, 25.3 now).  With THREE row sets a pass answers 12
  // queries: 18.8 us per query at 2^20 keys (two column windows of 8 / 7 tile groups, so that two blocks still share a CU's LDS), 134 with
  // 8 kB values (194 with 8 per pass), 70.5 at 2^22 keys (97.5); 233 VGPR, no scratch.
  // What the flush itself costs: 512 blocks x 7 680 words -- a fused pass of 8 -- leave as 3.9 M u32 atomics in one burst: 10 us even with
  // words of their own per block, 19.7 us when every block walks the words in the same order, 11.5 us when each starts at an offset of its
  // own (scripts/probes/atomic_flush_probe.hip); in the kernel the pass takes 1-2 us per query less without it.  Tried and dropped: a
  // double-sized accumulator whose finished half is trickled out during the NEXT pass, two atomics per thread and unit -- 27.0-27.4 us per
  // query: the halves need 93 KiB of LDS, i.e. one block per CU (28.0 us on its own against 25.2 with two), and the trickled atomics sit
  // in the wave's vector-memory queue between the tile prefetch and its wait, which, counted in issue order, then covers them too.
  const uint32_t rtotal = nq * cpad;
  for (uint32_t i = threadIdx.x; i < rtotal; i += kThreads) racc[i] = 0;
  uint32_t pass = 0;
  // every block starts its round over the words at an offset of its own (a multiple of 64), so that the blocks are spread over them
  const uint32_t fstart = (uint32_t)(((uint64_t)blockIdx.x * rtotal) / gridDim.x) & ~63u;
  auto flush_pass = [&](uint32_t of_pass, bool zero) __attribute__((always_inline)) {
    for (uint32_t cursor = threadIdx.x; cursor < rtotal; cursor += kThreads) {
      uint32_t i2 = cursor + fstart;
      if (i2 >= rtotal) i2 -= rtotal;
      const uint32_t query = i2 / cpad, col = a.tg_lo * (kM * 16) + i2 % cpad, val = racc[i2];
      if (zero) racc[i2] = 0;
      if (col < a.num_cols && val) atomicAdd(a.r + ((uint64_t)of_pass * nq + query) * a.num_cols + col, val);
    }
  };

  // prologue: A fragments of the first visit's step of pass 0, the first tile
  uint32_t v = 0;
  uint32_t cks = visit_ks(0), ctg1 = visit_tg1(0);  // the current visit
  uint32_t tg = visit_tg0(0);
  uint4 b0[NL], b1[NL];
  load_tile(b0, tg, cks);
  wait_for_step(cks);
  if (guarded_step(cks)) {
    a_guarded(cks, 0, 0);
  } else {
    uint4 raw0[2 * NS];
    a_issue(raw0, cks, 0);
    a_finish(raw0, 0);
  }
  __syncthreads();
  if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 1] = wall_clock64();

  int par = 0;
  // The NEXT visit's query words are requested in the first unit of the current visit; they are turned into fragments in the same
  // unit when q is near (L2), in the visit's last unit when q is far (host memory: a whole visit's worth of units, tens of
  // microseconds, covers the latency of the link).
  bool first_of_visit = true;
  bool done = false;
  uint4 raw[2 * NS];
  auto unit = [&](uint4(&cur)[NL], uint4(&nxt)[NL]) __attribute__((always_inline)) {
    const uint32_t ks = cks;
    const bool last_of_visit = tg + 1 == ctg1;
    const bool pass_ends = v + 1 == n_visits;             // this visit is the pass's last
    const bool more_visits = !pass_ends || pass + 1 < a.passes;  // something follows: the next visit, or visit 0 of the next pass
    const uint32_t nv = pass_ends ? 0u : v + 1, npass = pass_ends ? pass + 1 : pass;
    const uint32_t nks = visit_ks(nv);  // (unused values when nothing follows)
    const bool last = last_of_visit && !more_visits;
    const uint32_t tg_n = last_of_visit ? visit_tg0(nv) : tg + 1, ks_n = last_of_visit ? nks : ks;
    const bool build = (a.q_far ? last_of_visit : first_of_visit) && more_visits;  // block-uniform
    const bool g_n = more_visits && guarded_step(nks);
    // The next visit's query words.  q complete where it lies: requested now, a whole visit ahead of their use.  q still being FILLED by
    // the host: the fill count is only REQUESTED now (one 64-byte read over the link, ~2 us) and looked at after this unit's MFMAs, and
    // the words are requested then -- a wave that waited for the count here, in front of its tile prefetch, left its SIMD and its share
    // of the HBM stream idle for those 2 us in every visit (one block per CU: nobody else to fill in): ~18 of 200 us per query.
    const bool want_next = first_of_visit && more_visits;
    const bool ask = want_next && my_progress && !gave_up && seen <= nks;
    uint32_t early = 0;
    if (ask) early = __hip_atomic_load(my_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (want_next && !my_progress && !g_n) a_issue(raw, nks, npass);
    // ALWAYS issued (the very last unit asks for its own tile again: 9 KiB per block and launch): with a conditional prefetch the compiler
    // cannot count the loads in flight and waits for all of them, this tile's successor included, before the first MFMA
    load_tile(nxt, last ? tg : tg_n, last ? ks : ks_n);
    v4i acc_lo[NS], acc_hi[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) acc_lo[s] = v4i{0, 0, 0, 0}, acc_hi[s] = v4i{0, 0, 0, 0};
    const uint32_t Tw = tg * kM + wave;          // tile of the window (indexes the LDS accumulators)
    const uint32_t T = a.tg_lo * kM + Tw;        // tile of the image
    if (T < a.col_tiles) {
#pragma unroll
      for (int kb = 0; kb < 8; kb++) {
        v4i hb;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          uint32_t x = 0;
#pragma unroll
          for (int p = 0; p < HB; p++) x += ((comp(cur[8 + p], kb >> 1) >> (4 * (kb & 1) + d)) & 0x01010101u) << p;
          hb[d] = (int)x;
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
          const uint4 au = abuf[par][s][kb][lane];
          const v4i af = as_v4i(au);
          acc_lo[s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, as_v4i(cur[kb]), acc_lo[s], 0, 0, 0);
          if constexpr (HB > 0) acc_hi[s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, hb, acc_hi[s], 0, 0, 0);
        }
      }
      const uint32_t nvs = valid_slots(ks);
      const uint32_t col_term = (ks == 0 && a.colsum) ? 0x80808080u * a.colsum[T * 16 + cl] : 0u;
#pragma unroll
      for (int s = 0; s < NS; s++) {
        uint32_t val = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) val += ((uint32_t)acc_lo[s][i] + ((uint32_t)acc_hi[s][i] << 8)) << (8 * i);
        const uint32_t query = 4 * s + grp;
        if (query < nq) {
          const uint32_t qsum = (ksum[par][0][query] + ksum[par][1][query]) + (ksum[par][2][query] + ksum[par][3][query]);
          val += 128u * qsum - 0x40404000u * nvs + col_term;  // 128 * 0x80808080 = 0x40404000 mod 2^32
          atomicAdd(&racc[query * cpad + Tw * 16 + cl], val);  // LDS; this wave owns tile T of every step
        }
      }
    }
    if (want_next && my_progress) {
      if (ask) {
        seen = early;
        if (seen <= nks) wait_for_step(nks);  // not there yet: poll as before (with the timeout that lets the grid drain)
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // the words are requested after the count was seen, in program order
      }
      if (!g_n) a_issue(raw, nks, npass);
    }
    if (build) {
      if (!g_n) a_finish(raw, par ^ 1);
      else a_guarded(nks, npass, par ^ 1);
    }
    first_of_visit = false;
    if (last_of_visit) {
      if (more_visits) {
        __syncthreads();  // everybody is done with this step's fragments (and, at a pass boundary, with accumulating this pass); the next step's are complete
        par ^= 1;
        first_of_visit = true;
        if (pass_ends) {  // the pass's responses leave now (the next pass's first tile and fragments are on their way meanwhile)
          flush_pass(pass, true);
          __syncthreads();  // clean before anybody accumulates for the next pass
          pass = npass;
        }
        v = nv, cks = nks, ctg1 = visit_tg1(v);
      } else {
        done = true;
      }
    }
    tg = tg_n;
  };
  while (!done) {
    unit(b0, b1);
    if (done) break;
    unit(b1, b0);
  }

  // ---- the last pass ----
  __syncthreads();
  flush_pass(pass, false);
  if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 2] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The WIDE pass (round 4): up to 24 queries answered by ONE stream of the database.
//
// What bounds a fused pass is the LDS, not the matrix cores (three row sets keep them 27 % busy): a query costs 2 KiB of A fragments per
// step and 4 bytes per column of accumulators, and the step-major kernel above runs TWO 4-wave blocks per CU, each with its own
// double-buffered fragments -- 12 queries per pass, and at 2^20 keys x 1 kB only in two column windows, i.e. the 12 queries are gathered
// twice.  Here ONE 8-wave block owns the CU's whole LDS: single-buffered fragments (6 row sets x 8 KiB) + the accumulators of up to 24
// queries x 1 024 columns (96 KiB).  Same walk (512-slot steps, contiguous units split evenly over the blocks, slot axis split over the
// XCDs), same arithmetic, same packed image, same responses bit for bit; differences to the kernel above:
//   * a unit is a step x 8 column tiles (one per wave); PlanarArgs::tg_lo / tg_n count groups of EIGHT tiles here;
//   * every wave gathers ONE k-block (64 slots) of all row sets: one 16-byte load per lane and row set (a row set's four query rows x
//     256 contiguous bytes), parked in the fragment slab it is about to fill and read back in fragment order, as above;
//   * the row sets are a LOOP (their number is a run-time value, the tile stays in registers and is multiplied by one set after the
//     other, two accumulators live at a time), not an unrolled dimension: 6 sets unrolled would need ~400 VGPRs;
//   * fragments are single-buffered: the step's last unit ends with barrier / build the next step's fragments from registers (their
//     loads were issued in the step's first unit) / barrier; the next tile's loads are in flight across both.
constexpr int kWThreads = 512;
constexpr int kWM = 8;        // column tiles per work unit = waves per block
constexpr int kWMaxSets = 6;  // row sets of 4 queries

template <int HB, bool NT, bool MAP>
__global__ void __launch_bounds__(kWThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
respond_planar_wide_kernel(const PlanarArgs a) {
  constexpr int NL = 8 + HB;
  constexpr int ST16 = NL * 64;
  extern __shared__ uint4 wsm[];
  const uint32_t nq = a.q_per_pass;
  const uint32_t ns = (nq + 3) >> 2;                                    // row sets in use (1..6)
  uint4* const abuf = wsm;                                             // [ns][8 k-blocks][64 lanes]: A fragments of the current step
  uint32_t* const qs = reinterpret_cast<uint32_t*>(wsm + ns * 512);    // [parity of the step][32]: per query, the sum of the step's query words
  uint32_t* const racc = qs + 64;                                      // [query of the pass][padded column of the window]

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint32_t cl = lane & 15;
  const uint32_t grp = lane >> 4;
  const uint32_t cpad = a.tg_n * (kWM * 16);

  const uint32_t nx = a.nx;
  const uint32_t xcd = blockIdx.x % nx;
  const uint32_t j = blockIdx.x / nx;
  const uint32_t nb = gridDim.x / nx;  // host guarantees gridDim.x % nx == 0
  const uint32_t ks_len = a.ks_hi - a.ks_lo;
  const uint32_t kb0 = a.ks_lo + (uint32_t)(((uint64_t)ks_len * xcd) / nx);
  const uint32_t ke0 = a.ks_lo + (uint32_t)(((uint64_t)ks_len * (xcd + 1)) / nx);
  const uint32_t TG = a.tg_n;
  // visits (step, first tile group, one past the last): units u = (step kb0 + u / TG, tile group u % TG) split evenly over the blocks
  const uint64_t units = (uint64_t)TG * (ke0 - kb0);
  const uint64_t sb = units * j / nb, se = units * (j + 1) / nb;
  const uint32_t n_visits = se > sb ? (uint32_t)((se - 1) / TG - sb / TG + 1) : 0;
  if (n_visits == 0) return;  // block-uniform: an idle block takes part in nothing
  if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 0] = wall_clock64(), a.trace[blockIdx.x * 4 + 3] = n_visits;
  const uint32_t first_ks = kb0 + (uint32_t)(sb / TG), first_tg0 = (uint32_t)(sb % TG), last_tg1 = (uint32_t)((se - 1) % TG) + 1;
  auto visit_ks = [=](uint32_t v) { return first_ks + v; };
  auto visit_tg0 = [=](uint32_t v) { return v == 0 ? first_tg0 : 0u; };
  auto visit_tg1 = [=](uint32_t v) { return v + 1 == n_visits ? last_tg1 : TG; };

  const uint32_t limb = cl & 3;
  const uint32_t sel01 = limb | ((4 + limb) << 8);
  const uint4* const tiles = reinterpret_cast<const uint4*>(a.dtc);
  const uint32_t rr = lane >> 4, l16 = lane & 15;  // as a LOADER of query words: row of the set, 16-byte piece of the wave's 64 slots

  // A slot map (a.keep: only the rows of D that hold something are in the image, compact.hip) is applied HERE, where the query words are
  // gathered anyway: slot n of the image multiplies word keep[n] of the query.  (As a pass of its own in front of the launch the gather
  // read and wrote every query once more: 1.8 us per query next to the 10.5 of a pass of 24.)  The four indices of a lane are one 16-byte
  // load, requested in the visit's first unit; the four word loads per row set follow in the next unit, when the indices have arrived.
  // (MAP is a template parameter: the index registers and the word loads' addresses cost the plain kernel 24 VGPRs it does not have)
  constexpr bool mapped = MAP;
  auto guarded_step = [&](uint32_t ks_) __attribute__((always_inline)) {
    const uint64_t slot0 = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE;
    if (mapped) return slot0 + CPIR_PLANAR_SLOTS_PER_TILE > a.num_slots;  // (word loads: no alignment to ask for)
    return a.q_scalar || slot0 + CPIR_PLANAR_SLOTS_PER_TILE > a.num_slots || a.q_slot_offset + slot0 + CPIR_PLANAR_SLOTS_PER_TILE > a.q_len;
  };
  auto a_issue = [&](uint4(&raw)[kWMaxSets], uint32_t ks_, uint32_t pass_) __attribute__((always_inline)) {
    const uint64_t base = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE + wave * 64 + l16 * 4;
#pragma unroll
    for (int s = 0; s < kWMaxSets; s++) {
      raw[s] = make_uint4(0, 0, 0, 0);
      const uint32_t row = 4 * s + rr;
      if (row < nq) raw[s] = *reinterpret_cast<const uint4*>(a.q + ((uint64_t)pass_ * nq + row) * a.q_len + a.q_slot_offset + base);
    }
  };
  auto idx_issue = [&](uint32_t ks_) __attribute__((always_inline)) {
    return *reinterpret_cast<const uint4*>(a.keep + (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE + wave * 64 + l16 * 4);
  };
  // (a row set's base address is uniform, the lane adds a 32-bit word offset -- its row of the set and its slot: the host sends queries
  // beyond 2^28 words through the gather pass; the rows of a partly filled last set read row 0 of the set and are zeroed)
  const uint32_t roff = rr * (uint32_t)a.q_len;
  auto a_issue_mapped = [&](uint4(&raw)[kWMaxSets], const uint4& idx, uint32_t pass_) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < kWMaxSets; s++) {
      raw[s] = make_uint4(0, 0, 0, 0);
      if ((uint32_t)(4 * s) < nq) {  // block-uniform
        const uint32_t* const bs = a.q + ((uint64_t)pass_ * nq + 4 * s) * a.q_len + a.q_slot_offset;
        const bool valid = 4 * s + rr < nq;
        const uint32_t o = valid ? roff : 0u;
        const uint4 v = make_uint4(bs[o + idx.x], bs[o + idx.y], bs[o + idx.z], bs[o + idx.w]);
        raw[s] = valid ? v : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto wave_lds_fence = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // this wave's sum of the query words of its k-block, per query: counted once by the limb-0 lane of every 16-slot group, the four groups
  // added up with two shuffles, added to the step's sum by lane (group 0, limb 0).  The sums of a step live in qs[parity of the step]:
  // zeroed while the step BEFORE is being multiplied (after the barrier that ended its build, when the last reader of that half is
  // long gone), added to between the two barriers of the build, read -- one word per row set and tile -- until the next build.
  auto store_ksum = [&](uint32_t part, int s, int par) __attribute__((always_inline)) {
    uint32_t v = part;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const uint32_t query = 4 * s + (cl >> 2);
    if (grp == 0 && limb == 0 && query < nq) atomicAdd(&qs[par * 32 + query], v);
  };
  auto a_finish = [&](const uint4(&raw)[kWMaxSets], int par) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < kWMaxSets; s++) {
      if ((uint32_t)s < ns) {  // block-uniform
        uint32_t* const stage = reinterpret_cast<uint32_t*>(abuf + (s * 8 + wave) * 64);  // 4 rows x 64 words: this wave's slab of the set
        *reinterpret_cast<uint4*>(stage + rr * 64 + l16 * 4) = raw[s];
        wave_lds_fence();
        uint4 back[4];
#pragma unroll
        for (int d = 0; d < 4; d++) back[d] = *reinterpret_cast<const uint4*>(stage + (cl >> 2) * 64 + grp * 16 + d * 4);
        wave_lds_fence();
        const bool arow = 4 * s + (cl >> 2) < nq;
        uint32_t part = 0, o[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint4 t = back[d];
          part += (t.x + t.y) + (t.z + t.w);
          const uint32_t v = gather_limb(t.x, t.y, t.z, t.w, sel01) ^ 0x80808080u;
          o[d] = arow ? v : 0u;
        }
        abuf[(s * 8 + wave) * 64 + lane] = make_uint4(o[0], o[1], o[2], o[3]);
        store_ksum(part, s, par);
      }
    }
  };
  auto a_guarded = [&](uint32_t ks_, uint32_t pass_, int par) __attribute__((always_inline)) {
    const uint64_t base = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE + wave * 64 + grp * 16;
    for (uint32_t s = 0; s < ns; s++) {
      const uint32_t query = 4 * s + (cl >> 2);
      const bool arow = query < nq;
      const uint32_t* qrow = a.q + ((uint64_t)pass_ * nq + (arow ? query : 0)) * a.q_len + a.q_slot_offset;
      uint32_t part = 0, o[4];
#pragma unroll
      for (int d = 0; d < 4; d++) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint64_t n = base + d * 4 + e;
          const bool ok = n < a.num_slots && (mapped || a.q_slot_offset + n < a.q_len);
          const uint32_t x = ok ? qrow[mapped ? (uint64_t)a.keep[n] : n] : 0u;
          part += x;
          w[e] = ok ? (x ^ 0x80808080u) : 0u;
        }
        o[d] = arow ? gather_limb(w[0], w[1], w[2], w[3], sel01) : 0u;
      }
      abuf[(s * 8 + wave) * 64 + lane] = make_uint4(o[0], o[1], o[2], o[3]);
      store_ksum(part, (int)s, par);
    }
  };
  auto load_tile = [&](uint4(&dst)[NL], uint32_t tg_, uint32_t ks_) __attribute__((always_inline)) {
    const uint32_t T = (a.tg_lo + tg_) * kWM + wave;
    const uint4* p = tiles + (T < a.col_tiles ? ((uint64_t)T * a.ks_total + ks_) * ST16 : 0) + lane;
#pragma unroll
    for (int i = 0; i < NL; i++) dst[i] = load16<NT>(p + i * 64);
  };
  const uint64_t room = a.q_len > a.q_slot_offset ? a.q_len - a.q_slot_offset : 0;
  const uint64_t nvalid_total = (mapped || a.num_slots < room) ? a.num_slots : room;
  auto valid_slots = [&](uint32_t ks_) __attribute__((always_inline)) -> uint32_t {
    const uint64_t lo = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE;
    if (lo >= nvalid_total) return 0u;
    const uint64_t left = nvalid_total - lo;
    return left < CPIR_PLANAR_SLOTS_PER_TILE ? (uint32_t)left : CPIR_PLANAR_SLOTS_PER_TILE;
  };

  const uint32_t rtotal = nq * cpad;
  for (uint32_t i = threadIdx.x; i < rtotal; i += kWThreads) racc[i] = 0;
  if (threadIdx.x < 64) qs[threadIdx.x] = 0;
  __syncthreads();  // (the first build adds to the sums)
  const uint32_t fstart = (uint32_t)(((uint64_t)blockIdx.x * rtotal) / gridDim.x) & ~63u;  // every block starts its round at an offset of its own
  auto flush_pass = [&](uint32_t of_pass, bool zero) __attribute__((always_inline)) {
    for (uint32_t cursor = threadIdx.x; cursor < rtotal; cursor += kWThreads) {
      uint32_t i2 = cursor + fstart;
      if (i2 >= rtotal) i2 -= rtotal;
      const uint32_t query = i2 / cpad, col = a.tg_lo * (kWM * 16) + i2 % cpad, val = racc[i2];
      if (zero) racc[i2] = 0;
      if (col < a.num_cols && val) atomicAdd(a.r + ((uint64_t)of_pass * nq + query) * a.num_cols + col, val);
    }
  };

  // prologue: the first tile, the fragments of the first visit's step of pass 0
  uint32_t pass = 0, v = 0;
  uint32_t cks = visit_ks(0), ctg1 = visit_tg1(0);
  uint32_t tg = visit_tg0(0);
  uint4 b0[NL], b1[NL];
  uint4 raw[kWMaxSets];
  load_tile(b0, tg, cks);
  uint4 idx = make_uint4(0, 0, 0, 0);
  bool idx_pending = false;  // the next visit's indices are requested, its query words are not yet
  if (guarded_step(cks)) {
    a_guarded(cks, 0, 0);
  } else {
    if (mapped) {
      idx = idx_issue(cks);
      a_issue_mapped(raw, idx, 0);
    } else {
      a_issue(raw, cks, 0);
    }
    a_finish(raw, 0);
  }
  __syncthreads();

  int par = 0;  // parity of the current step's sums
  bool first_of_visit = true;
  bool done = false;
  auto unit = [&](uint4(&cur)[NL], uint4(&nxt)[NL]) __attribute__((always_inline)) {
    const uint32_t ks = cks;
    const bool last_of_visit = tg + 1 == ctg1;
    const bool pass_ends = v + 1 == n_visits;
    const bool more_visits = !pass_ends || pass + 1 < a.passes;
    const uint32_t nv = pass_ends ? 0u : v + 1, npass = pass_ends ? pass + 1 : pass;
    const uint32_t nks = visit_ks(nv);
    const bool last = last_of_visit && !more_visits;
    const uint32_t tg_n = last_of_visit ? visit_tg0(nv) : tg + 1, ks_n = last_of_visit ? nks : ks;
    const bool g_n = more_visits && guarded_step(nks);
    // the next visit's query words: requested now, a whole visit ahead of their use (L2 / HBM)
    if (more_visits && !g_n) {
      if (!mapped) {
        if (first_of_visit) a_issue(raw, nks, npass);
      } else if (first_of_visit) {
        idx = idx_issue(nks);
        idx_pending = true;
        if (last_of_visit) a_issue_mapped(raw, idx, npass), idx_pending = false;  // (a visit of one unit: no later unit to do it in)
      } else if (idx_pending) {
        a_issue_mapped(raw, idx, npass);
        idx_pending = false;
      }
    }
    if (first_of_visit && threadIdx.x < 32) qs[(par ^ 1) * 32 + threadIdx.x] = 0;  // the next step's sums start from zero
    load_tile(nxt, last ? tg : tg_n, last ? ks : ks_n);  // always issued (the very last unit asks for its own tile again), see above
    const uint32_t Tw = tg * kWM + wave;      // tile of the window (indexes the LDS accumulators)
    const uint32_t T = a.tg_lo * kWM + Tw;    // tile of the image
    if (T < a.col_tiles && !(a.ablate & 4u)) {
      v4i hbv[HB > 0 ? 8 : 1];
      constexpr bool kPeel = HB <= 4;  // (five and six planes: the peeled set's extra live values spill 12-44 bytes per lane; expanded up front there)
      if constexpr (HB > 0 && !kPeel) {
#pragma unroll
        for (int kb = 0; kb < 8; kb++)
#pragma unroll
          for (int d = 0; d < 4; d++) {
            uint32_t x = 0;
#pragma unroll
            for (int p = 0; p < HB; p++) x += ((comp(cur[8 + p], kb >> 1) >> (4 * (kb & 1) + d)) & 0x01010101u) << p;
            hbv[kb][d] = (int)x;
          }
      }
      const uint32_t nvs = valid_slots(ks);
      const uint32_t base_term = ((ks == 0 && a.colsum) ? 0x80808080u * a.colsum[T * 16 + cl] : 0u) - 0x40404000u * nvs;
      uint32_t* const rcol = racc + Tw * 16 + cl;
      // The row sets one after the other: 16 MFMAs each on the fragments in f[], every fragment register refilled with the NEXT set's
      // fragment of the same k-block right behind the two MFMAs that read it -- a read is issued eight MFMA pairs (256 cycles of
      // matrix-core issue) before its use.  (With the reads in front of their MFMAs, two by two, a set cost four LDS round trips: 32 us
      // per set and pass at 2^20 keys against 16 us of matrix-core issue.  A second full set of registers -- all eight reads in front
      // of the set -- does not fit: 18 to 39 VGPRs spilled.)  The last set re-reads itself: no condition in the loop.
      // The FIRST set is peeled off the loop: it expands the tile's bit planes into the high-byte operands k-block by k-block between
      // its MFMAs (some 80 VALU instructions per tile that otherwise run in front of the loop with the matrix cores idle).
      uint4 f[8];
#pragma unroll
      for (int kb = 0; kb < 8; kb++) f[kb] = abuf[kb * 64 + lane];
      auto row_set = [&](uint32_t s, auto first) __attribute__((always_inline)) {
        const uint4* const ap = abuf + (s + 1 < ns ? s + 1 : s) * 512 + lane;
        // no branch around the rows of a partly filled last set: they add 0 to the last query's word (their fragments are 0)
        const uint32_t query = 4 * s + grp, qq = query < nq ? query : nq - 1;
        const uint32_t qsum = qs[par * 32 + qq];  // (requested in front of the MFMAs, used behind them)
        v4i acc_lo = v4i{0, 0, 0, 0}, acc_hi = v4i{0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
          if constexpr (HB > 0 && decltype(first)::value) {
#pragma unroll
            for (int d = 0; d < 4; d++) {
              uint32_t x = 0;
#pragma unroll
              for (int p = 0; p < HB; p++) x += ((comp(cur[8 + p], kb >> 1) >> (4 * (kb & 1) + d)) & 0x01010101u) << p;
              hbv[kb][d] = (int)x;
            }
          }
          acc_lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(as_v4i(f[kb]), as_v4i(cur[kb]), acc_lo, 0, 0, 0);
          if constexpr (HB > 0) acc_hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(as_v4i(f[kb]), hbv[kb], acc_hi, 0, 0, 0);
          f[kb] = ap[kb * 64];
        }
        uint32_t val = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) val += ((uint32_t)acc_lo[i] + ((uint32_t)acc_hi[i] << 8)) << (8 * i);
        val += 128u * qsum + base_term;  // 128 * 0x80808080 = 0x40404000 mod 2^32
        atomicAdd(rcol + qq * cpad, query < nq ? val : 0u);  // LDS; this wave owns tile T of the step
      };
      row_set(0u, std::integral_constant<bool, kPeel>{});
#pragma unroll 1
      for (uint32_t s = 1; s < ns; s++) row_set(s, std::false_type{});
    }
    first_of_visit = false;
    if (last_of_visit) {
      if (more_visits) {
        __syncthreads();  // everybody is done with this step's fragments (and, at a pass boundary, with accumulating this pass)
        if (pass_ends) {
          if (!(a.ablate & 2u)) flush_pass(pass, true);  // (nobody accumulates for the next pass before the barrier below)
          pass = npass;
        }
        if (!(a.ablate & 1u)) {
          if (!g_n) a_finish(raw, par ^ 1);
          else a_guarded(nks, npass, par ^ 1);
        }
        __syncthreads();  // the next step's fragments are complete
        par ^= 1;
        first_of_visit = true;
        v = nv, cks = nks, ctg1 = visit_tg1(v);
      } else {
        done = true;
      }
    }
    tg = tg_n;
  };
  while (!done) {
    unit(b0, b1);
    if (done) break;
    unit(b1, b0);
  }

  __syncthreads();
  if (!(a.ablate & 2u)) flush_pass(pass, false);
  if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 2] = wall_clock64();
}

// r[q][c] += 128 * sum_n (q[n] - 0x80808080) over this block's slice of the valid slots, and (slice 0 only) += 0x80808080 * colsum[c]:
// the two correction terms of the signed-byte split (top of the file).  r was zeroed on the stream before; all updates are u32
// atomic adds, so their order against the main kernel's does not matter.
__global__ void __launch_bounds__(kThreads) planar_init_kernel(const uint32_t* __restrict__ q, uint64_t q_len, uint64_t q_slot_offset,
                                                                uint64_t num_slots, const uint32_t* __restrict__ colsum,
                                                                uint32_t num_cols, uint32_t* __restrict__ r, uint32_t split,
                                                                uint64_t range_lo, uint64_t range_hi) {
  __shared__ uint32_t sm[kThreads / 64];
  const uint32_t qi = blockIdx.x / split, s = blockIdx.x % split;
  const uint64_t room = q_len > q_slot_offset ? q_len - q_slot_offset : 0;
  const uint64_t nvalid = num_slots < room ? num_slots : room;
  // the valid slots of [range_lo, range_hi): this launch's part of the slot axis (colsum == NULL: its column term was added before)
  const uint64_t a0 = range_lo < nvalid ? range_lo : nvalid, a1 = range_hi < nvalid ? range_hi : nvalid;
  const uint64_t lo = a0 + (a1 - a0) * s / split, hi = a0 + (a1 - a0) * (s + 1) / split;
  const uint32_t* src = q + (uint64_t)qi * q_len + q_slot_offset;
  uint32_t sum = 0;
#pragma unroll 8
  for (uint64_t n = lo + threadIdx.x; n < hi; n += kThreads) sum += src[n];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = sum;
  __syncthreads();
  sum = sm[0] + sm[1] + sm[2] + sm[3];
  const uint32_t k = 128u * sum - 0x40404000u * (uint32_t)(hi - lo);  // 128 * 0x80808080 = 0x40404000 mod 2^32
  for (uint32_t c = threadIdx.x; c < num_cols; c += kThreads) {
    const uint32_t v = k + ((s == 0 && colsum) ? 0x80808080u * colsum[c] : 0u);
    atomicAdd(r + (uint64_t)qi * num_cols + c, v);
  }
}

using KernelFn = void (*)(const PlanarArgs);

template <int NS>
KernelFn pick_hb(uint32_t hb, bool nt) {
  switch (hb) {
    case 0: return nt ? respond_planar_kernel<0, NS, true> : respond_planar_kernel<0, NS, false>;  // b <= 8: the byte alone
    case 1: return nt ? respond_planar_kernel<1, NS, true> : respond_planar_kernel<1, NS, false>;
    case 2: return nt ? respond_planar_kernel<2, NS, true> : respond_planar_kernel<2, NS, false>;
    case 3: return nt ? respond_planar_kernel<3, NS, true> : respond_planar_kernel<3, NS, false>;
    case 4: return nt ? respond_planar_kernel<4, NS, true> : respond_planar_kernel<4, NS, false>;
    case 5: return nt ? respond_planar_kernel<5, NS, true> : respond_planar_kernel<5, NS, false>;
    case 6: return nt ? respond_planar_kernel<6, NS, true> : respond_planar_kernel<6, NS, false>;
    default: return nullptr;
  }
}

KernelFn pick(uint32_t hb, uint32_t batch, bool nt) { return batch <= 4 ? pick_hb<1>(hb, nt) : pick_hb<2>(hb, nt); }

template <int NS>
KernelFn pick_ks_hb(uint32_t hb, bool nt) {
  switch (hb) {
    case 0: return nt ? respond_planar_ks_kernel<0, NS, true> : respond_planar_ks_kernel<0, NS, false>;
    case 1: return nt ? respond_planar_ks_kernel<1, NS, true> : respond_planar_ks_kernel<1, NS, false>;
    case 2: return nt ? respond_planar_ks_kernel<2, NS, true> : respond_planar_ks_kernel<2, NS, false>;
    case 3: return nt ? respond_planar_ks_kernel<3, NS, true> : respond_planar_ks_kernel<3, NS, false>;
    case 4: return nt ? respond_planar_ks_kernel<4, NS, true> : respond_planar_ks_kernel<4, NS, false>;
    case 5: return nt ? respond_planar_ks_kernel<5, NS, true> : respond_planar_ks_kernel<5, NS, false>;
    case 6: return nt ? respond_planar_ks_kernel<6, NS, true> : respond_planar_ks_kernel<6, NS, false>;
    default: return nullptr;
  }
}

KernelFn pick_ks(uint32_t hb, uint32_t batch, bool nt) {
  return batch <= 4 ? pick_ks_hb<1>(hb, nt) : (batch <= 8 ? pick_ks_hb<2>(hb, nt) : pick_ks_hb<3>(hb, nt));
}

template <int HB>
KernelFn pick_wide_hb(bool nt, bool map) {
  if (map) return nt ? respond_planar_wide_kernel<HB, true, true> : nullptr;  // (a slot map with cached loads: not built; the caller gathers first)
  return nt ? respond_planar_wide_kernel<HB, true, false> : respond_planar_wide_kernel<HB, false, false>;
}
KernelFn pick_wide(uint32_t hb, bool nt, bool map) {
  switch (hb) {
    case 0: return pick_wide_hb<0>(nt, map);
    case 1: return pick_wide_hb<1>(nt, map);
    case 2: return pick_wide_hb<2>(nt, map);
    case 3: return pick_wide_hb<3>(nt, map);
    case 4: return pick_wide_hb<4>(nt, map);
    case 5: return pick_wide_hb<5>(nt, map);
    case 6: return pick_wide_hb<6>(nt, map);
    default: return nullptr;
  }
}

// LDS of a wide block: everything a CU has but a margin (the kernel has no static LDS)
constexpr uint32_t kWideLdsBudget = 156u << 10;

}  // namespace

// How many queries one pass over the image can answer in a launch of `passes` passes under the current dispatch rules: 12 where the
// step-major kernel takes the launch (three row sets of 4 queries), 8 where the tile-major kernel does (interleaved passes of small shards,
// or the step-major kernel switched off).
uint32_t planar_max_queries_per_pass(const cpir_dtc_layout& L, uint32_t passes, int interleave, int ks_mode) {
  const bool inter = passes > 1 && (interleave == 1 || (interleave < 0 && L.total_words * 4 <= (960ull << 20)));
  return (ks_mode >= 1 && !inter) ? CPIR_PLANAR_MAX_QUERIES_PER_PASS : 8u;
}

int launch_respond_planar(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                          uint64_t q_slot_offset, uint32_t batch, uint32_t passes, uint32_t* r, hipStream_t stream, int blocks_per_cu,
                          bool nontemporal, bool xcd_split, int interleave, int ks_mode, bool r_prezeroed, uint64_t step_lo, uint64_t step_hi,
                          const PlanarHostFill* fill) {
  // shape invariants the kernel relies on (layout already checked by the caller)
  if (L.packing != CPIR_PACK_PLANAR || batch == 0 || batch > CPIR_PLANAR_MAX_QUERIES_PER_PASS || passes == 0) return CPIR_ERR_INVALID_ARGUMENT;
  const uint32_t hb = planar_hi_planes(L.mat_elem_bit_len);
  // order of the passes: interleaved where the shard is small enough for concurrent passes to share its bytes on die (same rule
  // as respond.hip), slice order for HBM-sized streams.  Measured on MI355X, us per query, one query per pass, 32 passes:
  //   1/8 of the 2^20-key DB (157 MB): slice+nt 24.6, interleaved+nt 16.9, interleaved+cached loads 12.6
  //   1/2 (629 MB): 95.7 / 74.1 / 51.7;  whole (1.26 GB): 186.5 / 154.6 / 107.2 -- above the HBM roof, i.e. on-die reuse
  const bool inter = passes > 1 && (interleave == 1 || (interleave < 0 && L.total_words * 4 <= (960ull << 20)));
  // `nt` loads keep a once-per-query stream out of the caches; passes that are meant to share bytes on die use plain loads
  const bool nt = nontemporal && !inter;
  KernelFn fn = pick(hb, batch > 8 ? 8 : batch, nt);  // (batch > 8 never reaches it: checked below)
  if (!fn || L.chunk_words != (8 + hb) * 256 || L.rows_padded % 16 != 0) return CPIR_ERR_INVALID_ARGUMENT;
  const uint64_t ks_total = (L.num_slots + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE;
  if (ks_total > 0xffffffffull || ks_total * (L.chunk_words / 16) != L.words_per_row_padded) return CPIR_ERR_INVALID_ARGUMENT;

  PlanarArgs a;
  a.dtc = dtc;
  a.q = q;
  a.r = r;
  a.q_len = q_len;
  a.q_slot_offset = q_slot_offset;
  a.num_slots = L.num_slots;
  a.num_cols = L.num_cols;
  a.col_tiles = L.rows_padded / 16;
  a.tile_groups = (a.col_tiles + kM - 1) / kM;
  a.ks_total = (uint32_t)ks_total;
  // [step_lo, step_hi) in super-tile steps; step_hi = 0 means the whole slot axis.  A partial launch ADDS its part to r: the first
  // part (step_lo == 0) zeroes r and adds the per-column term, every part adds the per-query term of its own slots.
  if (step_hi == 0) step_hi = ks_total;
  if (step_lo >= step_hi || step_hi > ks_total) return CPIR_ERR_INVALID_ARGUMENT;
  a.ks_lo = (uint32_t)step_lo, a.ks_hi = (uint32_t)step_hi;
  a.q_per_pass = batch;
  a.passes = passes;
  a.interleave = inter ? 1u : 0u;
  a.q_far = ks_mode == 3 ? 1u : 0u;
  // q in host memory: whole steps round-robin over the blocks, so that q is consumed front to back (it may still be arriving) and
  // every word crosses the link exactly once
  a.strided = ks_mode == 3 ? 1u : 0u;
  a.progress = fill ? fill->progress : nullptr;
  a.abort_flag = fill ? fill->abort_flag : nullptr;
  a.poll_ticks = fill ? (uint64_t)fill->timeout_us * 100 : 0;
  a.ablate = 0;
  a.keep = nullptr;
  a.trace = nullptr;
  if (fill && (ks_mode != 3 || !fill->progress || !fill->abort_flag)) return CPIR_ERR_INVALID_ARGUMENT;
  a.q_scalar = (reinterpret_cast<uintptr_t>(q) % 16 != 0 || q_slot_offset % 4 != 0 || (batch * passes > 1 && q_len % 4 != 0)) ? 1u : 0u;

  // resident blocks per CU, measured on MI355X at 2^20 keys: streaming (slice order) 2 blocks 188.7 us per query, 3 blocks 192.9,
  // 1 block 273; sharing passes (interleaved) 3 blocks 107 vs 2 blocks 118.  The two-row-set kernel fits 2 blocks per CU.
  // The step-major kernel where it applies: slice order (every pass its own stream), LDS room for the pass's responses.  It adds the
  // correction terms itself (and reads every query word once), so no init kernel in front of it.  Measured at 2^20 keys, us per query,
  // tile-major / step-major: one query, one launch 200.6 / 191.9 (2^22 keys 757 / 723, 8 kB values 1453 / 1394); passes of 8 queries
  // 32.0 / 25.6, of 4 51.6 / 48.3; one query per pass, 32 passes a launch 185.3 / 185.0 (within half a per cent either way at every
  // config but 2^22 keys) -- so mode 1 keeps the tile-major kernel for that streaming case.  One-row-set step-major blocks run ONE per CU
  // (191.9 against 202.3 with two: half as many prologues and flushes, and 4 waves x 2 tiles in flight already cover the latency).
  // The step-major kernel keeps a pass's responses in LDS: 64 columns x batch u32 per tile group, at most 48 KiB.  Where all tile groups
  // do not fit (8 kB values: 7 312 columns x 8 queries = 234 KB) the launch is repeated over column WINDOWS of as many tile groups as fit,
  // each a launch of its own over all steps -- the query words are gathered once per window (a few MB against the GBs of the stream).
  // (blocks of the two-row-set kernel run two per CU and share its 160 KiB: 48 KiB each beside the 32 KiB of A fragments.  One block per CU
  // with 112 KiB -- 3 windows instead of 5 at 8 kB values -- measured the same: 191.9 against 193.5 us per query)
  // (three row sets -- 9 to 12 queries per pass, step-major kernel only -- keep 48 KiB of A fragments, so their accumulators get 31 KiB:
  // two blocks still share a CU, and a database wider than 10 tile groups is answered in column windows: 2 at 2^20 keys x 1 kB)
  const uint32_t racc_budget = batch > 8 ? (31u << 10) : (48u << 10);
  const uint32_t max_tg = racc_budget / (batch * kM * 16 * (uint32_t)sizeof(uint32_t));  // 24 tile groups for 8 queries, 192 for one, 10 for 12
  const uint32_t windows = (a.tile_groups + max_tg - 1) / max_tg;
  const uint32_t tg_per_window = (a.tile_groups + windows - 1) / windows;
  // (a query beyond 8 MiB no longer stays in the XCDs' L2 next to the stream, and the tile-major kernel gathers every word of it once
  // per tile group: at 2^22 keys, 18.9 MB, the step-major kernel streams 727 against 743 us per query, so it takes that case too)
  const bool long_query = (uint64_t)(a.ks_hi - a.ks_lo) * CPIR_PLANAR_SLOTS_PER_TILE * 4 > (8ull << 20);
  const bool want_ks = ks_mode >= 2 || (ks_mode == 1 && (batch >= 2 || passes == 1 || long_query));
  // (a query read in place over the host link must be read ONCE: one window or nothing)
  KernelFn fn_ks = (want_ks && !inter && (windows == 1 || ks_mode != 3)) ? pick_ks(hb, batch, nt) : nullptr;
  if (ks_mode == 3 && !fn_ks) return CPIR_ERR_INVALID_ARGUMENT;  // the caller relies on q being read once
  if (batch > 8 && !fn_ks) return CPIR_ERR_INVALID_ARGUMENT;      // the tile-major kernel has two row sets at most (planar_max_queries_per_pass)
  int bpc = blocks_per_cu > 0 ? blocks_per_cu : (fn_ks ? (batch <= 4 ? 1 : 2) : (inter ? 3 : 2));
  if (batch > 4 && bpc > 2) bpc = 2;
  // grid of a launch over `tgs` tile groups
  auto grid_for_units = [&](uint32_t tgs, uint32_t* nx_out) {
    const uint64_t units = (uint64_t)tgs * (a.ks_hi - a.ks_lo);
    uint64_t grid = (uint64_t)dev->num_cus * (uint64_t)bpc;
    uint32_t nx = (xcd_split && !a.strided && (a.ks_hi - a.ks_lo) >= 8 && grid % 8 == 0) ? 8u : 1u;
    const uint64_t blocks_needed = (units + kThreads / 64 - 1) / (kThreads / 64);
    if (grid > blocks_needed) {
      grid = blocks_needed;
      if (nx == 8) grid = (grid / 8) * 8;
      if (grid == 0) grid = 1, nx = 1;
    }
    *nx_out = nx;
    return grid;
  };

  const uint32_t nq = batch * passes;
  const bool first = (step_lo == 0);
  const uint32_t* colsum = first ? dtc + (uint64_t)L.rows_padded * L.words_per_row_padded : nullptr;
  if (first && !r_prezeroed) CPIR_HIP_TRY(hipMemsetAsync(r, 0, (size_t)nq * L.num_cols * sizeof(uint32_t), stream));
  if (fn_ks) {
    a.colsum = colsum;
    for (uint32_t w = 0; w < windows; w++) {
      a.tg_lo = w * tg_per_window;
      if (a.tg_lo >= a.tile_groups) break;
      a.tg_n = a.tile_groups - a.tg_lo < tg_per_window ? a.tile_groups - a.tg_lo : tg_per_window;
      const uint64_t grid = grid_for_units(a.tg_n, &a.nx);
      const size_t racc_bytes = (size_t)batch * a.tg_n * (kM * 16) * sizeof(uint32_t);
      // diagnosis (CPIR_KS_TRACE=1; launches in the in-place order only, i.e. the lone host caller's -- NOT the polled one, whose host side
      // must keep running while the kernel does -- or respond.ks_major = 3): when each block started, had its first fragments, ended;
      // printed per launch, which is synchronised for it
      static const bool trace_env = getenv("CPIR_KS_TRACE") != nullptr;
      static uint64_t* trace_dev = nullptr;
      const bool tracing = trace_env && ks_mode == 3 && !fill && windows == 1 && grid <= 4096;
      if (tracing) {
        if (!trace_dev && hipMalloc(reinterpret_cast<void**>(&trace_dev), 4096 * 4 * sizeof(uint64_t)) != hipSuccess) trace_dev = nullptr;
        if (trace_dev) (void)hipMemsetAsync(trace_dev, 0, 4096 * 4 * sizeof(uint64_t), stream);
        a.trace = trace_dev;
      }
      hipLaunchKernelGGL(fn_ks, dim3((unsigned)grid), dim3(kThreads), racc_bytes, stream, a);
      if (tracing && trace_dev) {
        std::vector<uint64_t> t((size_t)grid * 4);
        if (hipMemcpyAsync(t.data(), trace_dev, t.size() * 8, hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess) {
          uint64_t t0 = ~0ull;
          for (uint64_t b2 = 0; b2 < grid; b2++)
            if (t[b2 * 4] && t[b2 * 4] < t0) t0 = t[b2 * 4];
          std::vector<double> st, ff, en;
          for (uint64_t b2 = 0; b2 < grid; b2++)
            if (t[b2 * 4]) st.push_back((t[b2 * 4] - t0) * 0.01), ff.push_back((t[b2 * 4 + 1] - t0) * 0.01), en.push_back((t[b2 * 4 + 2] - t0) * 0.01);
          auto q3 = [](std::vector<double> v2, double* lo, double* med, double* hi) {
            std::sort(v2.begin(), v2.end());
            *lo = v2.front(), *med = v2[v2.size() / 2], *hi = v2.back();
          };
          if (!st.empty()) {
            double a0, a1, a2, b0, b1, b2_, c0, c1, c2;
            q3(st, &a0, &a1, &a2), q3(ff, &b0, &b1, &b2_), q3(en, &c0, &c1, &c2);
            fprintf(stderr, "[ks trace] blocks %zu  start %.1f/%.1f/%.1f  first fragments %.1f/%.1f/%.1f  end %.1f/%.1f/%.1f us (min/median/max after the first block's start)\n",
                    st.size(), a0, a1, a2, b0, b1, b2_, c0, c1, c2);
          }
        }
        a.trace = nullptr;
      }
    }
    CPIR_HIP_TRY(hipGetLastError());
    return CPIR_OK;
  }
  a.colsum = nullptr;
  a.tg_lo = 0, a.tg_n = a.tile_groups;
  const uint64_t grid = grid_for_units(a.tile_groups, &a.nx);
  const uint64_t range_lo = step_lo * CPIR_PLANAR_SLOTS_PER_TILE, range_hi = step_hi * CPIR_PLANAR_SLOTS_PER_TILE;
  // slices of the query per init block: at most 16 Ki slots each (a lone query must not leave a handful of blocks reading hundreds of
  // KB each in front of the main kernel: 12.6 us with 64 Ki-slot slices at 2^20 keys)
  uint32_t split = (uint32_t)((range_hi - range_lo + 16383) / 16384);
  split = split < 4 ? 4 : (split > 256 ? 256 : split);
  hipLaunchKernelGGL(planar_init_kernel, dim3(nq * split), dim3(kThreads), 0, stream, q, q_len, q_slot_offset, L.num_slots, colsum,
                     L.num_cols, r, split, range_lo, range_hi);
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), 0, stream, a);
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

// The wide pass (respond_planar_wide_kernel): `passes` passes of `batch` (1..CPIR_PLANAR_WIDE_MAX_QUERIES_PER_PASS) queries each, one 8-wave
// block per CU; a database whose accumulators do not fit beside the fragments is answered in column windows of whole groups of 8 tiles.
int launch_respond_planar_wide(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                               uint64_t q_slot_offset, uint32_t batch, uint32_t passes, uint32_t* r, hipStream_t stream, bool nontemporal,
                               bool xcd_split, const uint32_t* keep) {
  if (L.packing != CPIR_PACK_PLANAR || batch == 0 || batch > CPIR_PLANAR_WIDE_MAX_QUERIES_PER_PASS || passes == 0) return CPIR_ERR_INVALID_ARGUMENT;
  const uint32_t hb = planar_hi_planes(L.mat_elem_bit_len);
  KernelFn fn = pick_wide(hb, nontemporal, keep != nullptr);
  if (keep && q_len >= ((uint64_t)1 << 28)) return CPIR_ERR_INVALID_ARGUMENT;  // (32-bit word offsets inside a row set, see the kernel)
  if (!fn || L.chunk_words != (8 + hb) * 256 || L.rows_padded % 16 != 0) return CPIR_ERR_INVALID_ARGUMENT;
  const uint64_t ks_total = (L.num_slots + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE;
  if (ks_total > 0xffffffffull || ks_total * (L.chunk_words / 16) != L.words_per_row_padded) return CPIR_ERR_INVALID_ARGUMENT;

  PlanarArgs a;
  a.dtc = dtc;
  a.q = q;
  a.r = r;
  a.q_len = q_len;
  a.q_slot_offset = q_slot_offset;
  a.num_slots = L.num_slots;
  a.num_cols = L.num_cols;
  a.col_tiles = L.rows_padded / 16;
  a.tile_groups = (a.col_tiles + kWM - 1) / kWM;  // groups of EIGHT tiles here
  a.ks_total = (uint32_t)ks_total;
  a.ks_lo = 0, a.ks_hi = (uint32_t)ks_total;
  a.q_per_pass = batch;
  a.passes = passes;
  a.interleave = 0;
  a.q_far = 0;
  a.strided = 0;
  a.progress = nullptr;
  a.abort_flag = nullptr;
  a.poll_ticks = 0;
  a.q_scalar = (reinterpret_cast<uintptr_t>(q) % 16 != 0 || q_slot_offset % 4 != 0 || (batch * passes > 1 && q_len % 4 != 0)) ? 1u : 0u;
  a.colsum = dtc + (uint64_t)L.rows_padded * L.words_per_row_padded;
  static const uint32_t ablate_env = [] {
    const char* e = getenv("CPIR_WIDE_ABLATE");
    return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
  }();
  a.ablate = ablate_env;
  a.trace = nullptr;
  a.keep = keep;  // (device memory, at least L.num_slots entries, 16-byte aligned; the caller has checked that the slots it names lie inside q)
  if (keep && reinterpret_cast<uintptr_t>(keep) % 16 != 0) return CPIR_ERR_INVALID_ARGUMENT;

  const uint32_t ns = (batch + 3) / 4;
  const uint32_t fixed = ns * (8u << 10) + 256u;                                        // fragments + per-step query sums (two parities x 32)
  const uint32_t per_tg = batch * (kWM * 16) * (uint32_t)sizeof(uint32_t);              // accumulators of one group of 8 tiles
  const uint32_t max_tg = (kWideLdsBudget - fixed) / per_tg;                            // 8 groups (1 024 columns) for 24 queries
  if (max_tg == 0) return CPIR_ERR_INVALID_ARGUMENT;
  const uint32_t windows = (a.tile_groups + max_tg - 1) / max_tg;
  const uint32_t tg_per_window = (a.tile_groups + windows - 1) / windows;

  // (up to 156 KiB of dynamic LDS: say so once per instantiation; a runtime that does not know the attribute is not an error)
  static std::atomic<bool> lds_raised[7][2][2];
  if (!lds_raised[hb][nontemporal ? 1 : 0][keep ? 1 : 0].exchange(true, std::memory_order_relaxed)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLdsBudget) != hipSuccess) (void)hipGetLastError();
  }
  CPIR_HIP_TRY(hipMemsetAsync(r, 0, (size_t)batch * passes * L.num_cols * sizeof(uint32_t), stream));
  for (uint32_t w = 0; w < windows; w++) {
    a.tg_lo = w * tg_per_window;
    if (a.tg_lo >= a.tile_groups) break;
    a.tg_n = a.tile_groups - a.tg_lo < tg_per_window ? a.tile_groups - a.tg_lo : tg_per_window;
    const uint64_t units = (uint64_t)a.tg_n * ks_total;
    uint64_t grid = (uint64_t)dev->num_cus;  // one block per CU
    a.nx = (xcd_split && ks_total >= 8 && grid % 8 == 0) ? 8u : 1u;
    if (grid > units) {
      grid = units;
      if (a.nx == 8) grid = (grid / 8) * 8;
      if (grid == 0) grid = 1, a.nx = 1;
    }
    const size_t lds = (size_t)fixed + (size_t)a.tg_n * per_tg;
    static const bool trace_env = getenv("CPIR_KS_TRACE") != nullptr;  // (diagnosis, as for the step-major kernel: per XCD here)
    static uint64_t* trace_dev = nullptr;
    const bool tracing = trace_env && windows == 1 && grid <= 4096;
    if (tracing) {
      if (!trace_dev && hipMalloc(reinterpret_cast<void**>(&trace_dev), 4096 * 4 * sizeof(uint64_t)) != hipSuccess) trace_dev = nullptr;
      if (trace_dev) (void)hipMemsetAsync(trace_dev, 0, 4096 * 4 * sizeof(uint64_t), stream);
      a.trace = trace_dev;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kWThreads), lds, stream, a);
    if (tracing && trace_dev) {
      std::vector<uint64_t> t((size_t)grid * 4);
      if (hipMemcpyAsync(t.data(), trace_dev, t.size() * 8, hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess) {
        uint64_t t0 = ~0ull;
        for (uint64_t b2 = 0; b2 < grid; b2++)
          if (t[b2 * 4] && t[b2 * 4] < t0) t0 = t[b2 * 4];
        double sum[8] = {0}, mx[8] = {0}, mn[8];
        int cnt[8] = {0};
        for (int x = 0; x < 8; x++) mn[x] = 1e30;
        for (uint64_t b2 = 0; b2 < grid; b2++)
          if (t[b2 * 4]) {
            const double e = (t[b2 * 4 + 2] - t0) * 0.01;
            const int x = (int)(b2 % 8);
            sum[x] += e, cnt[x]++;
            if (e > mx[x]) mx[x] = e;
            if (e < mn[x]) mn[x] = e;
          }
        fprintf(stderr, "[wide trace] batch %u passes %u, block ends per XCD (min/mean/max us):", batch, passes);
        for (int x = 0; x < 8; x++)
          if (cnt[x]) fprintf(stderr, "  %d: %.0f/%.0f/%.0f", x, mn[x], sum[x] / cnt[x], mx[x]);
        fprintf(stderr, "\n");
      }
      a.trace = nullptr;
    }
  }
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace cpir
