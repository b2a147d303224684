// TEST INFRASTRUCTURE ONLY -- drives the REAL host state machine of Server::respond (chalametpir_amd/csrc/host_respond.hip, compiled as
// plain C++ against the simulated HIP runtime of tests/native/sim_hip/) through randomized caller interleavings, under ThreadSanitizer
// and AddressSanitizer, without a GPU (tests/test_host_state_machine.py).
//
// What runs for real: cpir_server_respond / _respond_bytes / _respond_batch_device and everything under them in host_respond.hip -- seat and
// arena admission, the leaders' gates and windows, in-place rounds (page-locked and polled pageable seats), the lone caller (in place,
// polled with and without the staging helpers, void launches answered again), the upload path (compacting seats on a server with a slot
// map, split uploads with the helpers, the upload streams taken in turn), the hand-over kernel, the group handle's workers and its device
// exchange, host_gather.cpp's compaction.  What is simulated: the runtime (streams as threads, events, copies) and the respond kernels
// (CPU stand-ins with the same contract: tests/native/sim_hip/sim_runtime.cpp).
//
// Every call's response is compared with the exact answer for ITS query (every word of a query carries a weight of its own slot and
// column), so a caller that is answered with somebody else's seat, from a half-copied block or twice shows up as a wrong response.
// Reference behaviour held: an Arc<Server> shared by many tasks, one respond per task (chalametpir_server/examples/server.rs:45-93).
//
//   host_state_machine_driver [calls [seed [runners]]]      -> "host state machine run ok: ..." and exit code 0
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>

#include "server_internal.hpp"
#include "sim_control.hpp"

using namespace cpir;

namespace {

uint32_t mix(uint64_t x) {
  x ^= x >> 33, x *= 0xff51afd7ed558ccdull, x ^= x >> 33, x *= 0xc4ceb9fe1a85ec53ull, x ^= x >> 33;
  return (uint32_t)x;
}

struct Bed {           // one server under test with everything needed to check its answers
  const char* name;
  cpir_server* handle = nullptr;
  uint64_t N = 0;      // words of a query
  uint32_t C = 0;
  bool group = false, mapped = false;
  std::vector<uint32_t*> q_pageable, q_pinned;  // the same queries in pageable (malloc; some at odd word offsets) and page-locked memory
  std::vector<std::vector<uint8_t>> q_wire;    // ... and as wire bytes (8-byte header + words: the words are only 8-byte aligned)
  std::vector<std::vector<uint32_t>> want;      // exact responses
  std::vector<void*> to_free;
};

cpir_dtc_layout sim_layout(uint64_t slots, uint32_t C, double pretend_pass_us) {
  cpir_dtc_layout L{};
  L.num_slots = slots, L.num_cols = C, L.mat_elem_bit_len = 9, L.compression_factor = 3;
  L.words_per_row = (slots + 2) / 3, L.words_per_row_padded = L.words_per_row, L.rows_padded = 16;
  L.total_words = (uint64_t)(pretend_pass_us * 1e-6 * 6.8e12 / 4);  // (only the host's timing heuristics read it: the weights are slots x C words)
  L.packing = CPIR_PACK_PLANAR, L.fields_per_word = 0, L.chunk_words = 9 * 256, L.slots_per_chunk = CPIR_PLANAR_SLOTS_PER_TILE;
  return L;
}

// an ordinary server (or one shard of a group) over slots [off, off + n) of queries of N words; returns the weights and the kept slots
Server* make_server(Device* dev, uint64_t N, uint64_t off, uint64_t n, uint32_t C, bool mapped, uint64_t seed, double pass_us, std::vector<uint32_t>* keep_out,
                    std::vector<uint32_t>* w_out) {
  const cpir_dtc_layout L = sim_layout(n, C, pass_us);
  Server* s = server_new(dev, L, off, N);
  std::vector<uint32_t> keep;
  if (mapped) {
    for (uint64_t i = 0; i < n; i++)
      if (mix(seed * 77 + i) % 9 != 0) keep.push_back((uint32_t)i);
    SlotMap m;
    m.n_kept = keep.size(), m.n_pad = (keep.size() + 127) / 128 * 128, m.n_orig = n;
    m.keep_host = keep;
    m.keep_bits.assign((n + 7) / 8 + 8, 0);
    for (uint32_t k : keep) m.keep_bits[k >> 3] |= (uint8_t)(1u << (k & 7));
    if (hipMalloc(reinterpret_cast<void**>(&m.keep_dev), m.n_pad * 4) != hipSuccess) abort();
    for (uint64_t i = 0; i < m.n_pad; i++) m.keep_dev[i] = i < keep.size() ? keep[i] : 0xffffffffu;
    cpir_dtc_layout phys = sim_layout(keep.size(), C, pass_us * 8 / 9);
    server_set_physical(s, phys, &m);
  } else {
    for (uint64_t i = 0; i < n; i++) keep.push_back((uint32_t)i);
  }
  uint32_t* w = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&w), keep.size() * C * 4) != hipSuccess) abort();
  for (uint64_t i = 0; i < keep.size() * C; i++) w[i] = mix(seed * 1000003 + i) | 1u;
  s->dtc = w;
  *keep_out = keep;
  w_out->assign(w, w + keep.size() * C);
  return s;
}

void add_exact(const uint32_t* q, uint64_t off, const std::vector<uint32_t>& keep, const std::vector<uint32_t>& w, uint32_t C, std::vector<uint32_t>& r) {
  for (uint64_t i = 0; i < keep.size(); i++)
    for (uint32_t c = 0; c < C; c++) r[c] += q[off + keep[i]] * w[i * C + c];
}

Bed make_bed(const char* name, Device* dev, uint64_t N, uint32_t C, bool mapped, int shards, uint64_t seed, double pass_us, int n_queries) {
  Bed b;
  b.name = name, b.N = N, b.C = C, b.mapped = mapped, b.group = shards > 1;
  struct Part {
    uint64_t off;
    std::vector<uint32_t> keep, w;
  };
  std::vector<Part> parts;
  if (shards <= 1) {
    parts.emplace_back();
    parts[0].off = 0;
    b.handle = static_cast<cpir_server*>(make_server(dev, N, 0, N, C, mapped, seed, pass_us, &parts[0].keep, &parts[0].w));
  } else {
    Server* parent = server_new(dev, sim_layout(N, C, pass_us), 0, N);
    const uint64_t unit = 1536;  // (lcm of the planar step's 512 slots and cf = 3, as cpir_shard_unit gives for b = 9)
    for (int g = 0; g < shards; g++) {
      const uint64_t units = (N + unit - 1) / unit, lo = std::min(N, units * g / shards * unit), hi = std::min(N, units * (g + 1) / shards * unit);
      parts.emplace_back();
      parts.back().off = lo;
      parent->shards.push_back(make_server(dev, N, lo, hi - lo, C, mapped, seed + 31 * g, pass_us / shards, &parts.back().keep, &parts.back().w));
    }
    b.handle = static_cast<cpir_server*>(parent);
  }
  for (int i = 0; i < n_queries; i++) {
    // pageable: every other one at an odd WORD offset (4 bytes past a 16-byte boundary: not 16-byte loadable)
    uint32_t* raw = static_cast<uint32_t*>(malloc((N + 8) * 4));
    b.to_free.push_back(raw);
    uint32_t* q = raw + (i & 1);
    for (uint64_t n = 0; n < N; n++) q[n] = mix(seed * 9176 + (uint64_t)i * N + n);
    b.q_pageable.push_back(q);
    uint32_t* p = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&p), N * 4, hipHostMallocDefault) != hipSuccess) abort();
    memcpy(p, q, N * 4);
    b.q_pinned.push_back(p);
    std::vector<uint8_t> wire(8 + N * 4);
    const uint32_t one = 1, cols = (uint32_t)N;
    memcpy(wire.data(), &one, 4), memcpy(wire.data() + 4, &cols, 4), memcpy(wire.data() + 8, q, N * 4);
    b.q_wire.push_back(std::move(wire));
    std::vector<uint32_t> r(C, 0);
    for (const Part& p2 : parts) add_exact(q, p2.off, p2.keep, p2.w, C, r);
    b.want.push_back(std::move(r));
  }
  return b;
}

struct Tally {
  std::atomic<uint64_t> calls{0}, wrong{0}, errors{0}, errors_allowed{0};
};
// copy failures are injected process-wide (the upload streams belong to the device, which every bed shares): an error status is expected
// for calls that began before the last injection window closed (+ a second: an arena that took the failure may still be open)
std::atomic<double> g_inject_until{0};

// one caller: `n` calls on the bed, each with a query drawn at random, from pageable / page-locked memory or as wire bytes
void caller(Bed* b, int n, uint32_t seed, int pinned_pct, int wire_pct, Tally* t) {
  std::minstd_rand rng(seed);
  std::vector<uint32_t> r(b->C);
  std::vector<uint8_t> resp(8 + (size_t)b->C * 4);
  for (int k = 0; k < n; k++) {
    const size_t qi = rng() % b->want.size();
    const int kind = (int)(rng() % 100);
    const double t_begin = now_seconds();
    int st;
    if (kind < wire_pct) {
      size_t len = 0;
      st = cpir_server_respond_bytes(b->handle, b->q_wire[qi].data(), b->q_wire[qi].size(), resp.data(), resp.size(), &len);
      if (st == CPIR_OK) memcpy(r.data(), resp.data() + 8, (size_t)b->C * 4);
    } else {
      const uint32_t* q = kind < wire_pct + pinned_pct ? b->q_pinned[qi] : b->q_pageable[qi];
      st = cpir_server_respond(b->handle, q, 1, b->N, r.data());
    }
    t->calls++;
    if (st != CPIR_OK) {
      (t_begin < g_inject_until.load() ? t->errors_allowed : t->errors)++;
      continue;
    }
    if (memcmp(r.data(), b->want[qi].data(), (size_t)b->C * 4) != 0) {
      if (t->wrong++ < 5) fprintf(stderr, "WRONG response: bed %s query %zu kind %d\n", b->name, qi, kind);
    }
  }
}

}  // namespace

int main(int argc, char** argv) {
  const uint64_t target_calls = argc > 1 ? strtoull(argv[1], nullptr, 0) : 100000;
  const uint32_t seed = argc > 2 ? (uint32_t)strtoul(argv[2], nullptr, 0) : 1;
  Device* dev = new Device;
  dev->ordinal = 0, dev->num_cus = 8;
  (void)hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking);

  // small beds carry the bulk of the interleavings (queries of 2^15 + 4096 words: long enough for polled copies and in-place rounds);
  // the big ones reach the paths that need 2^19 words (the staging helpers under a polled launch, uploads in two halves)
  const uint64_t Ns = (1u << 15) + 4096, Nb = (1u << 19) + 8192;
  std::vector<Bed> beds;
  beds.push_back(make_bed("plain", dev, Ns, 2, false, 1, 11, 90, 6));
  beds.push_back(make_bed("slot-map", dev, Ns, 2, true, 1, 12, 90, 6));
  beds.push_back(make_bed("wide-kernel-outweighs-uploads", dev, Ns, 3, false, 1, 13, 2000, 4));  // (one arena for all recent callers + batching window)
  beds.push_back(make_bed("group-of-3", dev, Ns * 3, 2, false, 3, 14, 90, 4));
  beds.push_back(make_bed("group-of-2-slot-map", dev, Ns * 2, 2, true, 2, 15, 90, 4));
  beds.push_back(make_bed("big", dev, Nb, 1, false, 1, 16, 190, 3));
  beds.push_back(make_bed("big-slot-map", dev, Nb, 1, true, 1, 17, 190, 3));

  for (size_t i = 5; i < beds.size(); i++) static_cast<Server*>(beds[i].handle)->trace_on = true;  // (counts the polled launches that had the staging helpers)
  Tally t;
  std::atomic<uint64_t> phases{0}, device_batches{0}, wrong_dim{0};
  std::vector<std::unique_ptr<std::mutex>> bed_busy;
  for (size_t i = 0; i < beds.size(); i++) bed_busy.emplace_back(new std::mutex);
  // several RUNNERS side by side, each a sequence of phases on a bed nobody else is using (the beds share the device's upload and run streams
  // and its launch / upload locks, as the servers of one process do); a phase = a crew of callers with the library's tuning drawn afresh
  auto runner = [&](uint32_t rseed) {
    std::minstd_rand rng(rseed);
    hipStream_t my_stream = nullptr;
    (void)hipStreamCreateWithFlags(&my_stream, hipStreamNonBlocking);
    const int crews[] = {1, 1, 2, 2, 2, 3, 3, 4, 4, 4, 6, 9, 16, 40};
    while (t.calls.load() < target_calls) {
      const size_t bi = rng() % 100 < 8 ? 5 + rng() % 2 : rng() % 5;
      std::unique_lock<std::mutex> mine(*bed_busy[bi], std::try_to_lock);
      if (!mine.owns_lock()) continue;
      phases++;
      Bed& b = beds[bi];
      const bool big = b.N > (1u << 19);
      const int big_crews[] = {1, 1, 1, 2, 3, 4, 6};
      const int T = big ? big_crews[rng() % 7] : crews[rng() % (sizeof crews / sizeof crews[0])];
      // (about as many calls per phase whatever the crew; a lone caller of a big bed stays long enough for the server's memory of earlier
      // company -- peak_inside, one step down per 8 calls -- to fade: only then is it served alone, with the staging helpers under a polled launch)
      const int per = big ? (T == 1 ? 20 : 2 + (int)(rng() % 3)) : std::max(3, (int)(40 + rng() % 40) / T);
      // the library's tuning, drawn per phase (process-wide, like the library's own: concurrent phases see each other's draws)
      const int seats[] = {0, 2, 3, 4, 4, 4};
      simctl::inplace_seats = seats[rng() % 6];
      simctl::upload_streams = 1 + (int)(rng() % 4);
      simctl::helper_spin_us = (rng() % 3 == 0) ? 50 : 0;
      simctl::read_once = rng() % 8 != 0;
      simctl::batch_fusion = rng() % 4 != 0;
      simctl::takes_slot_map = rng() % 3 != 0;
      const int give_up = (int)(rng() % 10);  // one phase in ten: polled launches give up at once (void passes answered again, polling switched off after three)
      simctl::fill_timeout_us = give_up == 0 ? 1 : (give_up == 1 ? 0 : 2000);
      simctl::set_chaos_us(rng() % 2 == 0 ? 0 : (rng() % 8 == 0 ? 40 : (uint32_t)(rng() % 8)));
      const bool inject = rng() % 25 == 0;
      if (inject) {
        g_inject_until.store(now_seconds() + 3600);
        simctl::fail_next_copies(1 + (int)(rng() % 3));
      }
      if (give_up == 0) {  // (a server that has stopped polling after three void launches in a row starts again)
        static_cast<Server*>(b.handle)->fill_aborts.store(0);
        for (Server* c : static_cast<Server*>(b.handle)->shards) c->fill_aborts.store(0);
      }
      const int pcts[] = {0, 30, 60, 100};
      const int pinned_pct = pcts[rng() % 4], wire_pct = pinned_pct == 100 ? 0 : (int)(rng() % 30);
      std::vector<std::thread> ths;
      for (int i = 0; i < T; i++) ths.emplace_back(caller, &b, per, (uint32_t)(rng()), pinned_pct, wire_pct, &t);
      // now and then, beside the host callers: a batch on DEVICE pointers (on a group: peer copies + the sum kernel), and a wrong dimension
      if (rng() % 6 == 0) {
        const uint32_t nb = 1 + rng() % 5;
        uint32_t *qd = nullptr, *rd = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&qd), (size_t)nb * b.N * 4) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&rd), (size_t)nb * b.C * 4) != hipSuccess) abort();
        std::vector<size_t> qis;
        for (uint32_t i = 0; i < nb; i++) qis.push_back(rng() % b.want.size()), memcpy(qd + (size_t)i * b.N, b.q_pinned[qis.back()], b.N * 4);
        const double t_begin = now_seconds();
        const int st = nb == 1 ? cpir_server_respond_device(b.handle, qd, rd, nullptr, my_stream) : cpir_server_respond_batch_device(b.handle, qd, nb, rd, nullptr, my_stream);
        (void)hipStreamSynchronize(my_stream);
        t.calls++, device_batches++;
        if (st != CPIR_OK) (t_begin < g_inject_until.load() ? t.errors_allowed : t.errors)++;
        else
          for (uint32_t i = 0; i < nb; i++)
            if (memcmp(rd + (size_t)i * b.C, b.want[qis[i]].data(), (size_t)b.C * 4) != 0 && !(t_begin < g_inject_until.load())) t.wrong++;
        (void)hipFree(qd), (void)hipFree(rd);
      }
      if (rng() % 10 == 0) {
        std::vector<uint32_t> r(b.C);
        if (cpir_server_respond(b.handle, b.q_pageable[0], 1, b.N - 1, r.data()) != CPIR_ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED) t.errors++;
        if (cpir_server_respond(b.handle, b.q_pageable[0], 2, b.N, r.data()) != CPIR_ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED) t.errors++;
        wrong_dim += 2;
      }
      for (std::thread& th : ths) th.join();
      if (inject) {
        simctl::fail_next_copies(0);
        g_inject_until.store(now_seconds() + 1.0);
      }
    }
    (void)hipStreamDestroy(my_stream);
  };
  {
    const int n_runners = argc > 3 ? atoi(argv[3]) : 5;
    std::vector<std::thread> rs;
    for (int i = 0; i < n_runners; i++) rs.emplace_back(runner, seed * 7919u + 104729u * (uint32_t)i);
    for (std::thread& th : rs) th.join();
  }

  // how the callers were served, summed over the beds (and their shards): every call of a plain server is counted in exactly one way
  uint64_t calls = 0, alone = 0, uploaded = 0, uploaded_rounds = 0, in_place = 0, in_place_rounds = 0, polled = 0, polled_void = 0, polled_with_helpers = 0;
  for (size_t i = 5; i < beds.size(); i++) polled_with_helpers += static_cast<Server*>(beds[i].handle)->trace.polled.load();
  bool counts_ok = true;
  for (Bed& b : beds) {
    std::vector<Server*> all;
    Server* s = static_cast<Server*>(b.handle);
    if (s->shards.empty()) all.push_back(s);
    else all = s->shards;
    for (Server* x : all) {
      const uint64_t c = x->served.calls.load(), a = x->served.alone.load(), u = x->served.in_uploaded_rounds.load(), ip = x->served.in_place_calls.load();
      counts_ok = counts_ok && c == a + u + ip;
      calls += c, alone += a, uploaded += u, in_place += ip, uploaded_rounds += x->served.uploaded_rounds.load();
      in_place_rounds += x->served.in_place_rounds.load(), polled += x->fill_polled.load(), polled_void += x->served.polled_void.load();
      // every arena is back where it started: nobody inside, no seat taken
      std::lock_guard<std::mutex> lk(x->mu);
      counts_ok = counts_ok && x->inside == 0;
      for (const RespondArena& a2 : x->arena) counts_ok = counts_ok && a2.state == RespondArena::FREE && a2.joined == 0 && a2.staged == 0 && a2.left == 0;
    }
  }
  for (Bed& b : beds) {
    server_destroy(static_cast<Server*>(b.handle));
    for (uint32_t* p : b.q_pinned) (void)hipHostFree(p);
    for (void* p : b.to_free) free(p);
  }
  scratch_drain(0);
  device_release(dev);
  const int64_t leaked = simctl::live_blocks();
  printf("calls %llu (device batches %llu, wrong-dimension calls %llu) in %llu phases; wrong %llu, errors %llu, errors under injected copy failures %llu\n",
         (unsigned long long)t.calls.load(), (unsigned long long)device_batches.load(), (unsigned long long)wrong_dim.load(), (unsigned long long)phases.load(),
         (unsigned long long)t.wrong.load(), (unsigned long long)t.errors.load(), (unsigned long long)t.errors_allowed.load());
  printf("lone polled launches with the staging helpers (queries of 2^19 words and more): %llu\n", (unsigned long long)polled_with_helpers);
  printf("served: calls %llu = alone %llu (polled %llu) + in uploaded rounds %llu (%llu rounds) + in in-place rounds %llu (%llu rounds); void polled passes %llu; "
         "simulated kernels %llu (polled %llu, gave up %llu); blocks still allocated %lld\n",
         (unsigned long long)calls, (unsigned long long)alone, (unsigned long long)polled, (unsigned long long)uploaded, (unsigned long long)uploaded_rounds,
         (unsigned long long)in_place, (unsigned long long)in_place_rounds, (unsigned long long)polled_void, (unsigned long long)simctl::kernels_launched.load(),
         (unsigned long long)simctl::polled_kernels.load(), (unsigned long long)simctl::polled_gave_up.load(), (long long)leaked);
  const bool ok = t.wrong.load() == 0 && t.errors.load() == 0 && counts_ok && leaked == 0 && alone > 0 && uploaded > 0 && in_place > 0 && polled > 0 && polled_void > 0 && polled_with_helpers > 0;
  if (!counts_ok) printf("COUNTS / ARENA STATE INCONSISTENT\n");
  puts(ok ? "host state machine run ok" : "host state machine run FAILED");
  return ok ? 0 : 1;
}
