// tsan_driver.cpp -- the library's HOST code under ThreadSanitizer (device code is untouched: `make -C chalametpir_amd/csrc tsan` compiles the
// host side of every translation unit with -fsanitize=thread into chalametpir_amd/lib/tsan/).  Drives, through the C ABI, what the
// multi-threaded parts do: servers created and destroyed next to each other, lone pageable / page-locked callers (polled launch, in-place
// read), bursts of concurrent callers (arenas, staging helpers), clones, the in-process group (worker threads), Server::setup (XOF
// thread, background disposal); since round 4 also databases with empty rows (the slot map: built on import, applied by the staging helpers'
// compress jobs for a lone caller and by the kernel for the arenas) and device-resident batches on a group from two threads (peer
// exchange contexts).  Answers are compared with the first answer to the same query.  Run on the GPU box:
//   TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" chalametpir_amd/lib/tsan/tsan_driver
// Reports whose stacks lie wholly inside the HIP runtime are the runtime's business (it is not instrumented); the ones to read name
// cpir:: frames.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "chalamet_hip.h"

#define CK(x)                                                                                   \
  do {                                                                                          \
    int _s = (x);                                                                               \
    if (_s != 0) {                                                                              \
      fprintf(stderr, "%s failed: %s %s\n", #x, cpir_strerror(_s), cpir_last_hip_error());     \
      exit(1);                                                                                  \
    }                                                                                           \
  } while (0)

static uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static void fill(uint32_t* p, size_t n, uint64_t seed, uint32_t mask) {
  for (size_t i = 0; i < n; i++) p[i] = (uint32_t)(mix(seed * 0x100000001B3ull + i) >> 32) & mask;
}

struct Shape {
  uint64_t N;
  uint32_t C, b;
};

static std::atomic<int> g_bad{0};

static void exercise(cpir_device* dev, const Shape& sh, int round) {
  const uint32_t cf = cpir_compression_factor(sh.b);
  const uint64_t W = (sh.N + cf - 1) / cf;
  std::vector<uint32_t> dtc((size_t)sh.C * W);
  fill(dtc.data(), dtc.size(), 7 + round, 0xFFFFFFFFu);
  if (round % 2 == 1) {  // every sixth slot holds nothing: the server leaves those out of its image and every path goes through the slot map
    const uint32_t S = 32 / cf, slot_mask = S == 32 ? 0xFFFFFFFFu : ((1u << S) - 1u);
    for (uint64_t n = 0; n < sh.N; n++)
      if (mix(n * 31 + round) % 6 == 0)
        for (uint32_t c = 0; c < sh.C; c++) dtc[(size_t)c * W + n / cf] &= ~(slot_mask << ((n % cf) * S));
  }
  cpir_server* srv = nullptr;
  CK(cpir_server_from_compressed(dev, dtc.data(), sh.C, sh.N, sh.b, &srv));
  const int kQ = 4;
  std::vector<std::vector<uint32_t>> q(kQ, std::vector<uint32_t>(sh.N)), want(kQ, std::vector<uint32_t>(sh.C));
  std::vector<uint32_t*> qp(kQ);
  for (int i = 0; i < kQ; i++) {
    fill(q[i].data(), sh.N, 1000 + i + 16 * round, 0xFFFFFFFFu);
    CK(cpir_host_alloc(4 * sh.N, (void**)&qp[i]));
    memcpy(qp[i], q[i].data(), 4 * sh.N);
    CK(cpir_server_respond(srv, q[i].data(), 1, sh.N, want[i].data()));  // lone, pageable
  }
  std::atomic<const char*> phase{"lone"};
  auto ask = [&](cpir_server* s, int i, bool pinned) {
    std::vector<uint32_t> r(sh.C);
    CK(cpir_server_respond(s, pinned ? qp[i] : q[i].data(), 1, sh.N, r.data()));
    if (memcmp(r.data(), want[i].data(), 4 * sh.C) != 0) {
      g_bad++;
      uint32_t wrong = 0;
      for (uint32_t c = 0; c < sh.C; c++) wrong += r[c] != want[i][c];
      fprintf(stderr, "MISMATCH: N %llu round %d, phase \"%s\", query %d %s, %u of %u columns differ (first: got %08x want %08x)\n", (unsigned long long)sh.N,
              round, phase.load(), i, pinned ? "page-locked" : "pageable", wrong, sh.C, r[0], want[i][0]);
    }
  };
  for (int i = 0; i < kQ; i++) ask(srv, i, true);  // lone, page-locked
  for (int i = 0; i < kQ; i++) ask(srv, i, false);
  {  // a burst of concurrent callers
    phase = "burst of 8";
    std::vector<std::thread> ts;
    for (int t = 0; t < 8; t++)
      ts.emplace_back([&, t] {
        for (int k = 0; k < 3; k++) ask(srv, (t + k) % kQ, (t & 1) != 0);
      });
    for (auto& t : ts) t.join();
  }
  // a few callers at a time: in-place rounds (respond.inplace_seats) -- page-locked queries read where they lie, pageable ones copied into
  // their seats by their callers' threads while the pass polls every seat's progress; mixed crews; started together
  for (int crew = 2; crew <= 4; crew++) {
    phase = crew == 2 ? "crew of 2 (page-locked)" : crew == 3 ? "crew of 3 (mixed)" : "crew of 4 (pageable)";
    std::atomic<int> ready{0};
    std::vector<std::thread> ts;
    for (int t = 0; t < crew; t++)
      ts.emplace_back([&, t] {
        ready++;
        while (ready.load() < crew) std::this_thread::yield();
        for (int k = 0; k < 4; k++) ask(srv, (t + k) % kQ, crew == 3 ? (t & 1) != 0 : crew == 2);
      });
    for (auto& t : ts) t.join();
  }
  {
    uint64_t counts[CPIR_HOST_PATH_COUNT];
    CK(cpir_server_host_path_counts(srv, counts));
    fprintf(stderr, "  N %llu round %d: calls %llu, alone %llu, in uploaded rounds %llu (%llu), in in-place rounds %llu (%llu), polled passes given up %llu\n",
            (unsigned long long)sh.N, round, (unsigned long long)counts[0], (unsigned long long)counts[1], (unsigned long long)counts[4],
            (unsigned long long)counts[5], (unsigned long long)counts[6], (unsigned long long)counts[7], (unsigned long long)counts[3]);
  }
  phase = "clone / second server";
  cpir_server* clone = cpir_server_retain(srv);
  cpir_server_release(srv);
  ask(clone, 0, false);
  ask(clone, 1, true);
  {  // a second server comes and goes while the clone is being asked from another thread
    std::thread other([&] {
      for (int k = 0; k < 4; k++) ask(clone, k % kQ, k & 1);
    });
    cpir_server* s2 = nullptr;
    CK(cpir_server_from_compressed(dev, dtc.data(), sh.C, sh.N, sh.b, &s2));
    ask(s2, 2, false);
    cpir_server_release(s2);
    other.join();
  }
  cpir_server_release(clone);
  for (int i = 0; i < kQ; i++) cpir_host_free(qp[i]);
}

static void exercise_setup_and_group(cpir_device* dev, int round) {
  const uint64_t N = 3 * 1536 * 2 + 512;
  const uint32_t C = 20, b = 9;
  std::vector<uint32_t> D((size_t)N * C), hint1((size_t)CPIR_LWE_DIMENSION * C), hint2(hint1.size());
  fill(D.data(), D.size(), 99 + round, (1u << b) - 1);
  if (round % 2 == 1)  // rows that hold nothing: both servers compact
    for (uint64_t n = 0; n < N; n++)
      if (mix(n + 17) % 5 == 0) memset(&D[(size_t)n * C], 0, (size_t)C * 4);
  uint8_t seed[CPIR_SEED_BYTE_LEN];
  for (int i = 0; i < CPIR_SEED_BYTE_LEN; i++) seed[i] = (uint8_t)(i * 7 + round);
  cpir_server *one = nullptr, *grp = nullptr;
  CK(cpir_server_setup(dev, seed, nullptr, D.data(), N, C, b, hint1.data(), &one));
  cpir_device* devs[3] = {dev, dev, dev};
  CK(cpir_server_setup_multi(devs, 3, seed, nullptr, D.data(), N, C, b, hint2.data(), &grp));
  if (memcmp(hint1.data(), hint2.data(), hint1.size() * 4) != 0) g_bad++, fprintf(stderr, "MISMATCH: round %d, the group's hint\n", round);
  std::vector<uint32_t> q(N), r1(C);
  fill(q.data(), N, 5 + round, 0xFFFFFFFFu);
  CK(cpir_server_respond(one, q.data(), 1, N, r1.data()));
  std::vector<std::thread> ts;
  for (int t = 0; t < 4; t++)
    ts.emplace_back([&] {
      std::vector<uint32_t> r(C);
      for (int k = 0; k < 3; k++) {
        CK(cpir_server_respond(grp, q.data(), 1, N, r.data()));
        if (memcmp(r.data(), r1.data(), 4 * C) != 0) g_bad++, fprintf(stderr, "MISMATCH: round %d, group respond on host pointers (call %d of its thread)\n", round, k);
      }
    });
  for (auto& t : ts) t.join();
  {  // device-resident batches on the group from two threads at once (each call takes the next exchange context)
    const uint32_t kB = 5;
    uint32_t *q_dev = nullptr, *r_dev = nullptr;
    if (hipMalloc((void**)&q_dev, (size_t)kB * N * 4) != hipSuccess || hipMalloc((void**)&r_dev, (size_t)2 * kB * C * 4) != hipSuccess) exit(1);
    for (uint32_t i = 0; i < kB; i++)
      if (hipMemcpy(q_dev + (size_t)i * N, q.data(), N * 4, hipMemcpyHostToDevice) != hipSuccess) exit(1);
    std::vector<std::thread> ds;
    for (int t = 0; t < 2; t++)
      ds.emplace_back([&, t] {
        hipStream_t st;
        if (hipStreamCreate(&st) != hipSuccess) exit(1);
        std::vector<uint32_t> r((size_t)kB * C);
        for (int k = 0; k < 3; k++) {
          CK(cpir_server_respond_batch_device(grp, q_dev, kB, r_dev + (size_t)t * kB * C, nullptr, st));
          if (hipMemcpyAsync(r.data(), r_dev + (size_t)t * kB * C, r.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) exit(1);
          for (uint32_t i = 0; i < kB; i++)
            if (memcmp(r.data() + (size_t)i * C, r1.data(), 4 * C) != 0) g_bad++, fprintf(stderr, "MISMATCH: round %d, group respond on device pointers (query %u)\n", round, (unsigned)i);
        }
        (void)hipStreamDestroy(st);
      });
    for (auto& t : ds) t.join();
    (void)hipFree(q_dev);
    (void)hipFree(r_dev);
  }
  cpir_server_release(grp);
  cpir_server_release(one);
}

int main() {
  cpir_device* dev = nullptr;
  CK(cpir_device_open(0, &dev));
  const Shape shapes[] = {{(1u << 19) + 4096 * 3 + 5, 24, 9}, {77824, 130, 10}, {3 * 1536 + 1, 19, 6}};
  for (int round = 0; round < 2; round++) {
    for (const Shape& sh : shapes) exercise(dev, sh, round);
    exercise_setup_and_group(dev, round);
    fprintf(stderr, "round %d done, mismatches so far %d\n", round, g_bad.load());
  }
  cpir_device_close(dev);
  printf("tsan driver finished, mismatches %d\n", g_bad.load());
  return g_bad.load() ? 1 : 0;
}
