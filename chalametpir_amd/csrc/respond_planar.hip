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
// Two kernels walk the same packed image with the same arithmetic and give the same responses bit for bit:
//   * respond_planar_wide_kernel -- every device-resident launch: 1 .. 24 queries per pass (one to six A row sets walked in a loop), any
//     number of passes per launch in slice or interleaved order, a slot map applied while the query words are gathered.  One 8-wave block
//     per CU; a work unit is a 512-slot step x 8 column tiles (one per wave); the step's A fragments are built once per block and step;
//     the responses accumulate in LDS and leave through one pass of u32 atomics per pass.  It adds the two correction terms itself.
//   * respond_planar_ks_kernel -- the lone HOST query: up to 4 queries per pass (one row set), every query word read exactly once, in
//     strided order, optionally while the host is still copying the query in (polled fill count).  4-wave blocks, one per CU.
// (Rounds 1-4 also had a tile-major kernel with an init kernel in front and two / three row sets of the step-major one; the wide kernel is
// faster than each of them wherever they were dispatched -- scripts/families_ab.py, profiles/r5_families_ab.txt -- and they are gone.)
//   * a wave's tile step is 8 + HB fully coalesced 1 KiB wave-loads (`nt` when streaming), issued one unit ahead of the MFMAs that consume them;
//   * persistent grid, units split evenly over the blocks; the slot axis is first split 8 ways by blockIdx % 8 so that an XCD's L2 holds
//     one eighth of q;
//   * the accumulator tile has the column on the lane and (query, byte) in the register index, so the recombination
//     sum_i 2^(8i) (lo_i + 256 hi_i) is lane-local.
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
  uint32_t q_per_pass;     // queries answered per pass: rows 4*i .. 4*i+3 of A row set i / 4 belong to query i
  uint32_t passes;         // independent passes over the database in this launch
  uint32_t q_scalar;       // q not 16-byte loadable -> guarded scalar loads everywhere
  uint32_t interleave;     // wide pass: order of the passes of one launch (see the kernel)
  uint32_t q_far;          // step-major kernel: q sits behind the host link -> a whole step of units between requesting and using it
  const uint32_t* colsum;  // step-major kernel: per-column field sums behind the tiles (NULL: this launch does not cover step 0)
  // step-major kernel, q read in place from host memory that is still being FILLED while the kernel runs (a lone pageable host query):
  uint32_t strided;          // blocks take whole steps round-robin (step s -> block s % blocks), so the grid consumes q front to back
  const uint32_t* progress;  // host memory: number of 512-slot steps of q in place so far, CPIR_FILL_LINES copies 64 bytes apart (NULL: all of q is in place)
  uint32_t progress_seats;   // ... one such set of copies per query of the pass, one behind the other (1: a lone query)
  uint32_t* abort_flag;      // device memory: set when a wave has given up waiting (the launch's results are then void)
  uint64_t poll_ticks;       // ... after this many ticks of the 100 MHz wall clock
  const uint32_t* keep;      // wide pass: the database holds only the slots keep[0 .. num_slots) of the query (increasing, relative to q_slot_offset;
                             // compact.hip); NULL: slot n of the database is word q_slot_offset + n of the query
  const uint32_t *q_row0, *q_row1, *q_row2, *q_row3;  // (four plain fields: an array inside the argument struct is sent through scratch memory)
                             // step-major kernel, one pass: where word 0 of each query of the pass lies -- the queries of concurrent host callers,
                             // read in place from wherever each caller's page-locked buffer is (q_row0 == NULL: query i is q + i * q_len)
#ifdef CPIR_DIAG  // (a diagnosis build only: the release library has neither field nor any code that reads them)
  uint64_t* trace;           // CPIR_KS_TRACE: per block 4 words -- wall clock at entry, after the first fragments, at the end; visits
  uint32_t ablate;           // wide pass, CPIR_WIDE_ABLATE (results are WRONG while non-zero): 1 no rebuild of the fragments, 2 no flush, 4 no MFMAs
#endif
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

// ---------------------------------------------------------------------------------------------------------------------------------
// The STEP-MAJOR kernel of the lone host query: a block owns (512-slot step, group of 4 column tiles) units ordered by step first, so
//   * the A fragments of a step are built ONCE per block and step and serve every tile group of that step -- every word of q is read by
//     (almost) exactly one block, which is what lets a lone host query be read straight from page-locked HOST memory (zero-copy: the
//     83 us upload of a 4.7 MB query disappears behind the 190 us stream instead of preceding it);
//   * one barrier per STEP (the waves of a block run free between the steps);
//   * the responses accumulate in LDS (one u32 per query and padded column) and leave through one pass of u32 atomics per block;
//   * the two correction terms of the signed-byte split are added here: while a wave gathers its share of a step's query words it also
//     sums them, and every (step, tile group) unit adds 128 * (sum of the step's valid query words) - 0x40404000 * (valid slots of the
//     step) to each of its 64 columns -- every unit is visited exactly once, so every column receives the whole per-query term; the
//     units of step 0 add 0x80808080 * colsum[column].
// One row set (up to 4 queries per pass), slice order only; everything device-resident runs on the wide kernel below.
template <int HB, bool NT>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
respond_planar_ks_kernel(const PlanarArgs a) {
  constexpr int NS = 1;          // sets of 16 A rows: one set answers up to 4 queries per pass (wider passes: the wide kernel)
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
  CPIR_DIAG_ONLY(if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 0] = wall_clock64(), a.trace[blockIdx.x * 4 + 3] = n_visits;)
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
  // (the queries of a round of concurrent callers are copied in by their callers' threads, each at its own pace, each counted in a set of
  // copies of its own: a step is in place when it is in place for every one of them)
  auto in_place_so_far = [&]() __attribute__((always_inline)) {
    uint32_t m = __hip_atomic_load(my_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (uint32_t s = 1; s < a.progress_seats; s++) {
      const uint32_t x = __hip_atomic_load(my_progress + s * (CPIR_FILL_LINES * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      m = x < m ? x : m;
    }
    return m;
  };
  auto wait_for_step = [&](uint32_t ks_) __attribute__((always_inline)) {
    if (!my_progress || gave_up || seen > ks_) return;
    const uint64_t t0 = wall_clock64();
    // RELAXED system-scope loads (they bypass the caches by themselves): an acquire would invalidate the L2 under the whole grid's feet
    // at every poll.  Nothing needs it: the words waited for are fetched only after the loop has seen the count (control dependency),
    // from host memory that this kernel has not touched before, and the host publishes the count with a release store after the copy.
    while ((seen = in_place_so_far()) <= ks_) {
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
  const uint32_t* const qr0 = a.q_row0;
  const uint32_t* const qr1 = a.q_row1;
  const uint32_t* const qr2 = a.q_row2;
  const uint32_t* const qr3 = a.q_row3;
  auto a_issue = [&](uint4(&raw)[2 * NS], uint32_t ks_, uint32_t pass_) __attribute__((always_inline)) {
    const uint64_t base = (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE + wave * 128 + l32 * 4;
#pragma unroll
    for (int i = 0; i < 2 * NS; i++) {
      const uint32_t row = 2 * i + half;
      raw[i] = make_uint4(0, 0, 0, 0);
      if (row < nq) {
        // (a table of row addresses: uniform pointers, the lane picks the one of its row -- rows 2 i and 2 i + 1 by its half of the wave)
        const uint32_t* const rp = qr0 ? (i == 0 ? (half ? qr1 : qr0) : (half ? qr3 : qr2)) : a.q + ((uint64_t)pass_ * nq + row) * a.q_len;
        raw[i] = *reinterpret_cast<const uint4*>(rp + a.q_slot_offset + base);
      }
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
      // (two selects by bit, not a chain of comparisons: the compiler turns the chain into a table in scratch memory)
      const uint32_t* const t_lo = (qi[s] & 1u) ? qr1 : qr0;
      const uint32_t* const t_hi = (qi[s] & 1u) ? qr3 : qr2;
      const uint32_t* const tabled = (qi[s] & 2u) ? t_hi : t_lo;
      const uint32_t* qrow = (qr0 ? tabled : a.q + ((uint64_t)pass_ * nq + qi[s]) * a.q_len) + a.q_slot_offset;
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
  // responses leave at the boundary, between two barriers, while those requests are in flight.
  // What the flush costs: the blocks' words leave as u32 atomics in one burst; every block starts its round over the words at an offset of
  // its own (19.7 us for a pass of 8 when every block walks them in the same order, 11.5 us with the offsets: scripts/probes/atomic_flush_probe.hip).
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
  CPIR_DIAG_ONLY(if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 1] = wall_clock64();)

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
    if (ask) early = in_place_so_far();
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
  CPIR_DIAG_ONLY(if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 2] = wall_clock64();)
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The WIDE kernel: every device-resident launch -- 1 .. 24 queries per pass answered by ONE stream of the database, any number of passes.
//
// ONE 8-wave block owns the CU's whole LDS: single-buffered A fragments (1 .. 6 row sets x 8 KiB) + the accumulators of up to 24 queries x
// 1 024 columns (96 KiB).  (Round 4's step-major kernel with two / three row sets ran TWO 4-wave blocks per CU, each with its own
// double-buffered fragments: 12 queries per pass at most, and at 2^20 keys x 1 kB only in two column windows.)  Same walk as the step-major
// kernel above (512-slot steps, contiguous units split evenly over the blocks, slot axis split over the XCDs), same arithmetic, same
// packed image, same responses bit for bit; what differs:
//   * a unit is a step x 8 column tiles (one per wave); PlanarArgs::tg_lo / tg_n count groups of EIGHT tiles here;
//   * every wave gathers ONE k-block (64 slots) of all row sets: one 16-byte load per lane and row set (a row set's four query rows x
//     256 contiguous bytes), parked in the fragment slab it is about to fill and read back in fragment order, as above;
//   * the row sets are a LOOP (their number is a run-time value, the tile stays in registers and is multiplied by one set after the
//     other, two accumulators live at a time), not an unrolled dimension: 6 sets unrolled would need ~400 VGPRs;
//   * fragments are single-buffered: the step's last unit ends with barrier / build the next step's fragments from registers (their
//     loads were issued in the step's first unit) / barrier; the next tile's loads are in flight across both;
//   * the passes of a launch go in slice or in interleaved order (see the kernel), and a slot map is applied while the query words are
//     gathered (MAP).
constexpr int kWThreads = 512;
constexpr int kWM = 8;        // column tiles per work unit = waves per block
constexpr int kWMaxSets = 6;  // row sets of 4 queries

// EMU32 (a -DCPIR_DIAG build only, responses WRONG): the row sets are taken two at a time and each pair's two 16x16x64 MFMAs replaced by ONE
// v_mfma_i32_32x32x32_i8 on the first set's fragment -- the same number of byte products per tile, half the fragment reads and half the
// operand bytes per product, 16 accumulator registers per product instead of 4: what a 32-column image layout would make of the matrix
// cores' and the LDS's share of a fused pass, measured without building that layout (scripts/wide_ablate.py, CPIR_WIDE_ABLATE bit 8).
#ifdef CPIR_DIAG
template <int HB, bool NT, bool MAP, int DV = 0>  // DV (diagnosis variants, responses WRONG): 1 EMU32, 2 / 3 the same B / A operand for all k-blocks of a row set, 4 eleven of twelve MFMAs
#else
template <int HB, bool NT, bool MAP>  // (the release kernel has no such parameter: its name in a trace is respond_planar_wide_kernel<HB, NT, MAP>)
#endif
__global__ void __launch_bounds__(kWThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
respond_planar_wide_kernel(const PlanarArgs a) {
#ifndef CPIR_DIAG
  constexpr bool EMU32 = false;
  constexpr int DV = 0;
#else
  constexpr bool EMU32 = DV == 1;
#endif
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
  // A pass is a sequence of VISITS (step, first tile group, one past the last): units u = (step kb0 + u / TG, tile group u % TG), u in [s0, s1).
  // Two orders of the passes of one launch (same arithmetic):
  //   slice order:       the units are split evenly over the blocks; a block walks ITS slice [s0, s1) once per pass -- every pass is a stream
  //                      of the database from HBM of its own;
  //   interleaved order: the passes are laid end to end and the whole (pass, unit) space is split evenly, so that different blocks walk
  //                      the same tiles for different passes at about the same time and share them on die (pays where the image is small
  //                      enough to stay there: the shards of a multi-GPU server).  A block then takes part in the passes [p_first, p_last]:
  //                      the tail of the first, whole ones, the head of the last.
  const uint64_t units = (uint64_t)TG * (ke0 - kb0);
  uint64_t r0, r1;  // this block's range: of the units (slice order) or of the (pass, unit) space (interleaved)
  uint32_t p_first = 0, p_last = a.passes - 1;
  if (a.interleave) {
    const uint64_t all = units * a.passes;
    r0 = all * j / nb, r1 = all * (j + 1) / nb;
    if (r1 > r0) p_first = (uint32_t)(r0 / units), p_last = (uint32_t)((r1 - 1) / units);
  } else {
    r0 = units * j / nb, r1 = units * (j + 1) / nb;
  }
  if (r1 <= r0) return;  // block-uniform: an idle block takes part in nothing
  CPIR_DIAG_ONLY(if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 0] = wall_clock64(), a.trace[blockIdx.x * 4 + 3] = (uint32_t)(r1 - r0);)
  // The visits of a pass for this block, as plain block-uniform scalars.  There are only three kinds of pass: the block's first (in the
  // interleaved order possibly only its tail), a whole one, the block's last (possibly only its head) -- computed HERE, once: a 64-bit
  // division inside the loop costs nine vector registers at its most crowded point.  In slice order all three are the same.
  uint32_t f_nv, f_ks0, f_tg0, f_tg1;  // number of visits, step of visit 0, first tile group of visit 0, end of the last visit
  uint32_t m_nv, m_ks0, m_tg0, m_tg1;
  uint32_t l_nv, l_ks0, l_tg0, l_tg1;
  auto visits_of = [&](uint64_t s0, uint64_t s1, uint32_t& nv_, uint32_t& ks0_, uint32_t& tg0_, uint32_t& tg1_) __attribute__((always_inline)) {
    nv_ = (uint32_t)((s1 - 1) / TG - s0 / TG + 1);
    ks0_ = kb0 + (uint32_t)(s0 / TG), tg0_ = (uint32_t)(s0 % TG), tg1_ = (uint32_t)((s1 - 1) % TG) + 1;
  };
  if (a.interleave) {
    const uint64_t first0 = r0 - (uint64_t)p_first * units, last1 = r1 - (uint64_t)p_last * units;
    visits_of(first0, p_first == p_last ? last1 : units, f_nv, f_ks0, f_tg0, f_tg1);
    visits_of(0, units, m_nv, m_ks0, m_tg0, m_tg1);
    visits_of(0, last1, l_nv, l_ks0, l_tg0, l_tg1);
  } else {
    visits_of(r0, r1, f_nv, f_ks0, f_tg0, f_tg1);
    m_nv = l_nv = f_nv, m_ks0 = l_ks0 = f_ks0, m_tg0 = l_tg0 = f_tg0, m_tg1 = l_tg1 = f_tg1;
  }
  uint32_t c_nv = f_nv, c_ks0 = f_ks0, c_tg0 = f_tg0, c_tg1 = f_tg1;  // the current pass
  uint32_t n_nv, n_ks0, n_tg0, n_tg1;                                  // the pass after it (what the last visit of a pass prefetches and builds for)
  auto pass_after = [&](uint32_t p) __attribute__((always_inline)) {   // p + 1 <= p_last
    const bool is_last = p + 1 == p_last;
    n_nv = is_last ? l_nv : m_nv, n_ks0 = is_last ? l_ks0 : m_ks0, n_tg0 = is_last ? l_tg0 : m_tg0, n_tg1 = is_last ? l_tg1 : m_tg1;
  };
  n_nv = c_nv, n_ks0 = c_ks0, n_tg0 = c_tg0, n_tg1 = c_tg1;
  if (p_first < p_last) pass_after(p_first);

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
  // (a row set's base address is uniform -- scalar registers --, the lane adds ITS row of the set and its 16-byte piece: one 64-bit
  // offset per lane, computed once)
  const uint64_t q_lane_off = (uint64_t)rr * a.q_len + (uint32_t)(wave * 64 + l16 * 4);
  auto a_issue = [&](uint4(&raw)[kWMaxSets], uint32_t ks_, uint32_t pass_) __attribute__((always_inline)) {
    const uint32_t* const bs0 = a.q + (uint64_t)pass_ * nq * a.q_len + a.q_slot_offset + (uint64_t)ks_ * CPIR_PLANAR_SLOTS_PER_TILE;
#pragma unroll
    for (int s = 0; s < kWMaxSets; s++) {
      raw[s] = make_uint4(0, 0, 0, 0);
      const uint32_t row = 4 * s + rr;
      if (row < nq) raw[s] = *reinterpret_cast<const uint4*>(bs0 + (uint64_t)(4 * s) * a.q_len + q_lane_off);
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
        const bool arow = 4 * s + (cl >> 2) < nq CPIR_DIAG_ONLY(&& !(a.ablate & 16u));  // (diagnosis, bit 16: all-zero A fragments -- the same
                                                                                      // matrix instructions on operands that draw less power)
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

  // prologue: the first tile, the fragments of the first visit's step of the block's first pass
  uint32_t pass = p_first, v = 0;
  uint32_t cks = c_ks0, ctg1 = c_nv == 1 ? c_tg1 : TG;
  uint32_t tg = c_tg0;
  uint4 b0[NL], b1[NL];
  uint4 raw[kWMaxSets];
  load_tile(b0, tg, cks);
  uint4 idx = make_uint4(0, 0, 0, 0);
  bool idx_pending = false;  // the next visit's indices are requested, its query words are not yet
  if (guarded_step(cks)) {
    a_guarded(cks, pass, 0);
  } else {
    if (mapped) {
      idx = idx_issue(cks);
      a_issue_mapped(raw, idx, pass);
    } else {
      a_issue(raw, cks, pass);
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
    const bool pass_ends = v + 1 == c_nv;
    const bool more_visits = !pass_ends || pass < p_last;
    const uint32_t nv = pass_ends ? 0u : v + 1, npass = pass_ends ? pass + 1 : pass;
    const uint32_t nks = pass_ends ? n_ks0 : c_ks0 + nv;  // (unused values when nothing follows)
    const bool last = last_of_visit && !more_visits;
    const uint32_t tg_n = last_of_visit ? (pass_ends ? n_tg0 : 0u) : tg + 1, ks_n = last_of_visit ? nks : ks;
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
    if (T < a.col_tiles CPIR_DIAG_ONLY(&& !(a.ablate & 4u))) {
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
      auto row_set = [&](uint32_t s, auto first, auto with_hi) __attribute__((always_inline)) {
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
          // (diagnosis variants 2 / 3: the SAME B operands / A fragment for all eight k-blocks of a row set -- what the matrix cores draw when
          // only one of their two operands changes from instruction to instruction; the indices fold at compile time)
          const int kbB = DV == 2 ? 0 : kb, kbA = DV == 3 ? 0 : kb;
          // (... and the operands that are not multiplied are still loaded and waited for: an empty statement that claims to read them)
          if constexpr (DV == 2) asm volatile("" ::"v"(cur[kb].x), "v"(cur[kb].y), "v"(cur[kb].z), "v"(cur[kb].w));
          if constexpr (DV == 3) asm volatile("" ::"v"(f[kb].x), "v"(f[kb].y), "v"(f[kb].z), "v"(f[kb].w));
          acc_lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(as_v4i(f[kbA]), as_v4i(cur[kbB]), acc_lo, 0, 0, 0);
          if constexpr (HB > 0 && decltype(with_hi)::value) acc_hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(as_v4i(f[kbA]), hbv[kbB], acc_hi, 0, 0, 0);
          f[kb] = ap[kb * 64];
        }
        uint32_t val = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) val += ((uint32_t)acc_lo[i] + ((uint32_t)acc_hi[i] << 8)) << (8 * i);
        val += 128u * qsum + base_term;  // 128 * 0x80808080 = 0x40404000 mod 2^32
        atomicAdd(rcol + qq * cpad, query < nq ? val : 0u);  // LDS; this wave owns tile T of the step
      };
      if constexpr (!EMU32) {
        row_set(0u, std::integral_constant<bool, kPeel>{}, std::true_type{});
        if constexpr (DV == 4) {
          // (diagnosis variant 4, responses WRONG: the SIXTH row set of a pass of 24 without its high-plane MFMAs -- 11 instead of 12 matrix
          // instructions per k-block, i.e. exactly what high-plane fragments of 3 rows per query (limb 3 x 2^8 = 0 mod 2^32) would save, with
          // none of what they would cost: their LDS, their fragment reads, their second atomic -- the upper bound of that lever, in situ)
#pragma unroll 1
          for (uint32_t s = 1; s < ns && s < 5; s++) row_set(s, std::false_type{}, std::true_type{});
          if (ns == 6) row_set(5u, std::false_type{}, std::false_type{});
        } else {
#pragma unroll 1
          for (uint32_t s = 1; s < ns; s++) row_set(s, std::false_type{}, std::true_type{});
        }
      } else {
        typedef int v16i __attribute__((ext_vector_type(16)));
        if constexpr (HB > 0 && kPeel) {
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
#pragma unroll 1
        for (uint32_t s = 0; s < ns; s += 2) {
          const uint4* const ap = abuf + (s + 2 < ns ? s + 2 : s) * 512 + lane;
          const uint32_t query = 4 * s + grp, qq = query < nq ? query : nq - 1;
          const uint32_t qsum = qs[par * 32 + qq];
          v16i acc_lo = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc_hi = acc_lo;
#pragma unroll
          for (int kb = 0; kb < 8; kb++) {
            acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(as_v4i(f[kb]), as_v4i(cur[kb]), acc_lo, 0, 0, 0);
            if constexpr (HB > 0) acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(as_v4i(f[kb]), hbv[kb], acc_hi, 0, 0, 0);
            f[kb] = ap[kb * 64];
          }
          uint32_t val = 0, val2 = 0;
#pragma unroll
          for (int i = 0; i < 8; i++) val += ((uint32_t)acc_lo[i] + ((uint32_t)acc_hi[i] << 8)) << (8 * (i & 3));
#pragma unroll
          for (int i = 8; i < 16; i++) val2 += ((uint32_t)acc_lo[i] + ((uint32_t)acc_hi[i] << 8)) << (8 * (i & 3));
          val += 128u * qsum + base_term;
          atomicAdd(rcol + qq * cpad, query < nq ? val : 0u);
          const uint32_t query2 = query + 4, qq2 = query2 < nq ? query2 : nq - 1;
          atomicAdd(rcol + qq2 * cpad, query2 < nq ? val2 + base_term : 0u);  // (the pair's second set: as many LDS atomics as the real kernel)
        }
      }
    }
    first_of_visit = false;
    if (last_of_visit) {
      if (more_visits) {
        __syncthreads();  // everybody is done with this step's fragments (and, at a pass boundary, with accumulating this pass)
        if (pass_ends) {
          CPIR_DIAG_ONLY(if (!(a.ablate & 2u)))
          flush_pass(pass, true);  // (nobody accumulates for the next pass before the barrier below)
          pass = npass;
          c_nv = n_nv, c_ks0 = n_ks0, c_tg0 = n_tg0, c_tg1 = n_tg1;
          if (pass < p_last) pass_after(pass);
        }
        CPIR_DIAG_ONLY(if (!(a.ablate & 1u))) {
          if (!g_n) a_finish(raw, par ^ 1);
          else a_guarded(nks, npass, par ^ 1);
        }
        __syncthreads();  // the next step's fragments are complete
        par ^= 1;
        first_of_visit = true;
        v = nv, cks = nks, ctg1 = v + 1 == c_nv ? c_tg1 : TG;
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
  CPIR_DIAG_ONLY(if (!(a.ablate & 2u)))
  flush_pass(pass, false);
  CPIR_DIAG_ONLY(if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 4 + 2] = wall_clock64();)
}

#ifdef CPIR_DIAG
// per-block timing trace of a launch (CPIR_KS_TRACE=1, diagnosis build only): 4 words per block -- wall clock at entry, behind the first
// fragments, at the end; visits.  The launch is synchronised for it.
uint64_t* diag_trace_begin(hipStream_t stream) {
  static uint64_t* trace_dev = nullptr;
  if (!trace_dev && hipMalloc(reinterpret_cast<void**>(&trace_dev), 4096 * 4 * sizeof(uint64_t)) != hipSuccess) trace_dev = nullptr;
  if (trace_dev) (void)hipMemsetAsync(trace_dev, 0, 4096 * 4 * sizeof(uint64_t), stream);
  return trace_dev;
}
bool diag_trace_fetch(const uint64_t* trace_dev, uint64_t grid, hipStream_t stream, std::vector<uint64_t>& t, uint64_t* t0) {
  t.assign((size_t)grid * 4, 0);
  if (hipMemcpyAsync(t.data(), trace_dev, t.size() * 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return false;
  *t0 = ~0ull;
  for (uint64_t b = 0; b < grid; b++)
    if (t[b * 4] && t[b * 4] < *t0) *t0 = t[b * 4];
  return true;
}
void diag_trace_print_blocks(const uint64_t* trace_dev, uint64_t grid, hipStream_t stream) {
  std::vector<uint64_t> t;
  uint64_t t0;
  if (!diag_trace_fetch(trace_dev, grid, stream, t, &t0)) return;
  std::vector<double> st, ff, en;
  for (uint64_t b = 0; b < grid; b++)
    if (t[b * 4]) st.push_back((t[b * 4] - t0) * 0.01), ff.push_back((t[b * 4 + 1] - t0) * 0.01), en.push_back((t[b * 4 + 2] - t0) * 0.01);
  if (st.empty()) return;
  auto q3 = [](std::vector<double> v, double* lo, double* med, double* hi) {
    std::sort(v.begin(), v.end());
    *lo = v.front(), *med = v[v.size() / 2], *hi = v.back();
  };
  double a0, a1, a2, b0, b1, b2, c0, c1, c2;
  q3(st, &a0, &a1, &a2), q3(ff, &b0, &b1, &b2), q3(en, &c0, &c1, &c2);
  fprintf(stderr, "[ks trace] blocks %zu  start %.1f/%.1f/%.1f  first fragments %.1f/%.1f/%.1f  end %.1f/%.1f/%.1f us (min/median/max after the first block's start)\n",
          st.size(), a0, a1, a2, b0, b1, b2, c0, c1, c2);
}
void diag_trace_print_xcds(const uint64_t* trace_dev, uint64_t grid, uint32_t batch, uint32_t passes, hipStream_t stream) {
  std::vector<uint64_t> t;
  uint64_t t0;
  if (!diag_trace_fetch(trace_dev, grid, stream, t, &t0)) return;
  double sum[8] = {0}, mx[8] = {0}, mn[8];
  int cnt[8] = {0};
  for (int x = 0; x < 8; x++) mn[x] = 1e30;
  for (uint64_t b = 0; b < grid; b++)
    if (t[b * 4]) {
      const double e = (t[b * 4 + 2] - t0) * 0.01;
      const int x = (int)(b % 8);
      sum[x] += e, cnt[x]++;
      if (e > mx[x]) mx[x] = e;
      if (e < mn[x]) mn[x] = e;
    }
  fprintf(stderr, "[wide trace] batch %u passes %u, block ends per XCD (min/mean/max us):", batch, passes);
  for (int x = 0; x < 8; x++)
    if (cnt[x]) fprintf(stderr, "  %d: %.0f/%.0f/%.0f", x, mn[x], sum[x] / cnt[x], mx[x]);
  fprintf(stderr, "\n");
}
#endif

using KernelFn = void (*)(const PlanarArgs);

KernelFn pick_ks(uint32_t hb, bool nt) {
  switch (hb) {
    case 0: return nt ? respond_planar_ks_kernel<0, true> : respond_planar_ks_kernel<0, false>;  // b <= 8: the byte alone
    case 1: return nt ? respond_planar_ks_kernel<1, true> : respond_planar_ks_kernel<1, false>;
    case 2: return nt ? respond_planar_ks_kernel<2, true> : respond_planar_ks_kernel<2, false>;
    case 3: return nt ? respond_planar_ks_kernel<3, true> : respond_planar_ks_kernel<3, false>;
    case 4: return nt ? respond_planar_ks_kernel<4, true> : respond_planar_ks_kernel<4, false>;
    case 5: return nt ? respond_planar_ks_kernel<5, true> : respond_planar_ks_kernel<5, false>;
    case 6: return nt ? respond_planar_ks_kernel<6, true> : respond_planar_ks_kernel<6, false>;
    default: return nullptr;
  }
}

template <int HB>
KernelFn pick_wide_hb(bool nt, bool map) {
  if (map) return nt ? respond_planar_wide_kernel<HB, true, true> : respond_planar_wide_kernel<HB, false, true>;
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

// Order of the passes of one launch: interleaved where the image is small enough for concurrent passes to share its bytes on die, slice
// order for HBM-sized streams.  Measured on MI355X, us per query, one query per pass, 32 passes a launch (scripts/families_ab.py, round 5):
//   1/8 of the 2^20-key DB (157 MB): slice + nt 23.8, interleaved + cached loads 8.2;  1/4 (313 MB): 47.5 / 17.2;
//   1/2 (626 MB): 95.3 / 34.3;  whole (1.25 GB): 184.1 / 68.6 -- above the HBM roof, i.e. on-die reuse, never the headline
bool planar_passes_interleaved(const cpir_dtc_layout& L, uint32_t passes, int interleave) {
  return passes > 1 && (interleave == 1 || (interleave < 0 && L.total_words * 4 <= (960ull << 20)));
}

// The step-major kernel: `passes` passes of `batch` (1..CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS) queries each, slice order.  It reads every query
// word exactly once per column window, which is what the in-place host path needs (in_place: whole steps round-robin over the blocks, so
// that the grid consumes q front to back, and the far-mode fragment schedule; `fill`: q is still being copied in while the kernel runs).
int launch_respond_planar_ks(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                             uint64_t q_slot_offset, uint32_t batch, uint32_t passes, uint32_t* r, hipStream_t stream, int blocks_per_cu,
                             bool nontemporal, bool xcd_split, bool in_place, bool r_prezeroed, uint64_t step_lo, uint64_t step_hi,
                             const PlanarHostFill* fill, const uint32_t* const* q_rows) {
  // shape invariants the kernel relies on (layout already checked by the caller)
  if (L.packing != CPIR_PACK_PLANAR || batch == 0 || batch > CPIR_PLANAR_KS_MAX_QUERIES_PER_PASS || passes == 0) return CPIR_ERR_INVALID_ARGUMENT;
  if (q_rows && (passes != 1 || q)) return CPIR_ERR_INVALID_ARGUMENT;  // (a table of row addresses: one pass, and no base beside it)
  const uint32_t hb = planar_hi_planes(L.mat_elem_bit_len);
  KernelFn fn = pick_ks(hb, nontemporal);
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
  a.interleave = 0;
  a.q_far = in_place ? 1u : 0u;
  // q in host memory: whole steps round-robin over the blocks, so that q is consumed front to back (it may still be arriving) and
  // every word crosses the link exactly once
  a.strided = in_place ? 1u : 0u;
  a.progress = fill ? fill->progress : nullptr;
  a.progress_seats = fill ? fill->seats : 1u;
  if (fill && (fill->seats == 0 || fill->seats > batch)) return CPIR_ERR_INVALID_ARGUMENT;
  a.abort_flag = fill ? fill->abort_flag : nullptr;
  a.poll_ticks = fill ? (uint64_t)fill->timeout_us * 100 : 0;
  a.keep = nullptr;
  CPIR_DIAG_ONLY(a.ablate = 0; a.trace = nullptr;)
  if (fill && (!in_place || !fill->progress || !fill->abort_flag)) return CPIR_ERR_INVALID_ARGUMENT;
  a.q_scalar = (reinterpret_cast<uintptr_t>(q) % 16 != 0 || q_slot_offset % 4 != 0 || (batch * passes > 1 && q_len % 4 != 0)) ? 1u : 0u;
  a.q_row0 = a.q_row1 = a.q_row2 = a.q_row3 = nullptr;
  if (q_rows) {
    a.q_scalar = q_slot_offset % 4 != 0 ? 1u : 0u;
    const uint32_t* rows[4];
    for (uint32_t i = 0; i < 4; i++) {
      rows[i] = q_rows[i < batch ? i : 0];  // (the rows beyond the batch are never read; a valid address all the same)
      if (!rows[i]) return CPIR_ERR_INVALID_ARGUMENT;
      if (reinterpret_cast<uintptr_t>(rows[i]) % 16 != 0) a.q_scalar = 1u;
    }
    a.q_row0 = rows[0], a.q_row1 = rows[1], a.q_row2 = rows[2], a.q_row3 = rows[3];
  }

  // The kernel keeps a pass's responses in LDS: 64 columns x batch u32 per tile group, at most 48 KiB.  Where all tile groups do not fit
  // (one query beyond 12 288 columns) the launch is repeated over column WINDOWS of as many tile groups as fit, each a launch of its own
  // over all steps -- the query words are gathered once per window (a few MB against the GBs of the stream).
  const uint32_t max_tg = (48u << 10) / (batch * kM * 16 * (uint32_t)sizeof(uint32_t));  // 192 tile groups for one query, 48 for four
  const uint32_t windows = (a.tile_groups + max_tg - 1) / max_tg;
  const uint32_t tg_per_window = (a.tile_groups + windows - 1) / windows;
  if (in_place && windows != 1) return CPIR_ERR_INVALID_ARGUMENT;  // (a query read in place over the host link must be read ONCE: one window or nothing)
  // One block per CU (measured at 2^20 keys, one query: 191.9 us against 202.3 with two: half as many prologues and flushes, and 4 waves x
  // 2 tiles in flight already cover the latency)
  const int bpc = blocks_per_cu > 0 ? (blocks_per_cu > 2 ? 2 : blocks_per_cu) : 1;
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
  a.colsum = first ? dtc + (uint64_t)L.rows_padded * L.words_per_row_padded : nullptr;
  if (first && !r_prezeroed) CPIR_TRY(zero_words(r, (uint64_t)nq * L.num_cols, stream));
  for (uint32_t w = 0; w < windows; w++) {
    a.tg_lo = w * tg_per_window;
    if (a.tg_lo >= a.tile_groups) break;
    a.tg_n = a.tile_groups - a.tg_lo < tg_per_window ? a.tile_groups - a.tg_lo : tg_per_window;
    const uint64_t grid = grid_for_units(a.tg_n, &a.nx);
    const size_t racc_bytes = (size_t)batch * a.tg_n * (kM * 16) * sizeof(uint32_t);
#ifdef CPIR_DIAG
    // diagnosis (CPIR_KS_TRACE=1; launches in the in-place order only, i.e. the lone host caller's -- NOT the polled one, whose host side
    // must keep running while the kernel does): when each block started, had its first fragments, ended; printed per launch, which is
    // synchronised for it
    static const bool trace_env = getenv("CPIR_KS_TRACE") != nullptr;
    const bool tracing = trace_env && in_place && !fill && windows == 1 && grid <= 4096;
    if (tracing) a.trace = diag_trace_begin(stream);
#endif
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), racc_bytes, stream, a);
#ifdef CPIR_DIAG
    if (tracing && a.trace) diag_trace_print_blocks(a.trace, grid, stream);
    a.trace = nullptr;
#endif
  }
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}


int launch_respond_planar_wide(const Device* dev, const uint32_t* dtc, const cpir_dtc_layout& L, const uint32_t* q, uint64_t q_len,
                               uint64_t q_slot_offset, uint32_t batch, uint32_t passes, uint32_t* r, hipStream_t stream, bool nontemporal,
                               bool xcd_split, int interleave, const uint32_t* keep) {
  if (L.packing != CPIR_PACK_PLANAR || batch == 0 || batch > CPIR_PLANAR_WIDE_MAX_QUERIES_PER_PASS || passes == 0) return CPIR_ERR_INVALID_ARGUMENT;
  const uint32_t hb = planar_hi_planes(L.mat_elem_bit_len);
  // order of the passes: interleaved where the image is small enough for concurrent passes to share its bytes on die, slice order for
  // HBM-sized streams; `nt` loads keep a once-per-pass stream out of the caches, passes that are meant to share bytes use plain loads
  // (the interleaved order numbers the (pass, unit) space in 64 bits and multiplies it by a block index: far beyond any real launch, but the
  // slice order is always there)
  const uint64_t units_all = (uint64_t)((L.rows_padded / 16 + kWM - 1) / kWM) * ((L.num_slots + CPIR_PLANAR_SLOTS_PER_TILE - 1) / CPIR_PLANAR_SLOTS_PER_TILE);
  const bool inter = planar_passes_interleaved(L, passes, interleave) && units_all * passes < ((uint64_t)1 << 50);
  const bool nt = nontemporal && !inter;
  KernelFn fn = pick_wide(hb, nt, keep != nullptr);
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
  a.interleave = inter ? 1u : 0u;
  a.q_far = 0;
  a.strided = 0;
  a.progress = nullptr;
  a.progress_seats = 1;
  a.abort_flag = nullptr;
  a.poll_ticks = 0;
  a.q_scalar = (reinterpret_cast<uintptr_t>(q) % 16 != 0 || q_slot_offset % 4 != 0 || (batch * passes > 1 && q_len % 4 != 0)) ? 1u : 0u;
  a.colsum = dtc + (uint64_t)L.rows_padded * L.words_per_row_padded;
#ifdef CPIR_DIAG
  static const uint32_t ablate_env = [] {
    const char* e = getenv("CPIR_WIDE_ABLATE");
    return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
  }();
  a.ablate = ablate_env;
  a.trace = nullptr;
  if ((ablate_env & 8u) && hb == 1 && nt && !keep) fn = respond_planar_wide_kernel<1, true, false, 1>;  // (the 32x32x32 emulation: see the kernel)
  if ((ablate_env & 32u) && hb == 1 && nt && !keep) fn = respond_planar_wide_kernel<1, true, false, 2>;  // (the same B operands for a whole row set)
  if ((ablate_env & 64u) && hb == 1 && nt && !keep) fn = respond_planar_wide_kernel<1, true, false, 3>;  // (the same A fragment for a whole row set)
  if ((ablate_env & 128u) && hb == 1 && nt && !keep) fn = respond_planar_wide_kernel<1, true, false, 4>;  // (11 of 12 matrix instructions per k-block)
#endif
  a.q_row0 = a.q_row1 = a.q_row2 = a.q_row3 = nullptr;
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
  if (!lds_raised[hb][nt ? 1 : 0][keep ? 1 : 0].exchange(true, std::memory_order_relaxed)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLdsBudget) != hipSuccess) (void)hipGetLastError();
  }
  CPIR_TRY(zero_words(r, (uint64_t)batch * passes * L.num_cols, stream));
  for (uint32_t w = 0; w < windows; w++) {
    a.tg_lo = w * tg_per_window;
    if (a.tg_lo >= a.tile_groups) break;
    a.tg_n = a.tile_groups - a.tg_lo < tg_per_window ? a.tile_groups - a.tg_lo : tg_per_window;
    const uint64_t units = (uint64_t)a.tg_n * ks_total * (inter ? passes : 1u);  // what the blocks share out: the units, or the (pass, unit) space
    uint64_t grid = (uint64_t)dev->num_cus;  // one block per CU
    a.nx = (xcd_split && ks_total >= 8 && grid % 8 == 0) ? 8u : 1u;
    if (grid > units) {
      grid = units;
      if (a.nx == 8) grid = (grid / 8) * 8;
      if (grid == 0) grid = 1, a.nx = 1;
    }
    const size_t lds = (size_t)fixed + (size_t)a.tg_n * per_tg;
#ifdef CPIR_DIAG
    static const bool trace_env = getenv("CPIR_KS_TRACE") != nullptr;  // (as for the step-major kernel; the block ends per XCD here)
    const bool tracing = trace_env && windows == 1 && grid <= 4096;
    if (tracing) a.trace = diag_trace_begin(stream);
#endif
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kWThreads), lds, stream, a);
#ifdef CPIR_DIAG
    if (tracing && a.trace) diag_trace_print_xcds(a.trace, grid, batch, passes, stream);
    a.trace = nullptr;
#endif
  }
  CPIR_HIP_TRY(hipGetLastError());
  return CPIR_OK;
}

}  // namespace cpir