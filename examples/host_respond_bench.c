/* host_respond_bench.c -- Server::respond through the C ABI from plain C threads (what a Rust caller sees: no interpreter, no GIL).
 *
 * The reference's own online bench is one caller in a loop handing `server.respond(&query_bytes)` a pageable Vec<u8>
 * (integrations/benches/online_phase.rs:81-97); its example server answers one tokio task per connection on an Arc<Server>
 * (chalametpir_server/examples/server.rs:59-93).  This program builds a server of a BASELINE shape from a synthetic packed matrix
 * (cpir_server_from_compressed: no XOF, no hint -- only respond is timed), then measures cpir_server_respond on host buffers:
 * latency of a lone caller (pageable / page-locked query) and throughput of T closed-loop callers, T = 1, 2, 4, 8, 16.
 * Every response is compared with the first answer to the same query (the queries repeat), so a wrong answer cannot go unnoticed.
 *
 *   build:  make -C chalametpir_amd/csrc host_respond_bench      run:  chalametpir_amd/lib/host_respond_bench [n_keys_log2=20] [value_bytes=1024] [arity=3] [lone]
 * prints one JSON object. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "chalamet_hip.h"

#define CHECK(x)                                                                                             \
  do {                                                                                                       \
    int _s = (x);                                                                                            \
    if (_s != 0) {                                                                                           \
      fprintf(stderr, "%s failed: %s %s\n", #x, cpir_strerror(_s), cpir_last_hip_error());                  \
      exit(1);                                                                                               \
    }                                                                                                        \
  } while (0)

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static uint64_t mix(uint64_t x) {  /* splitmix64 */
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

enum { kQueries = 16 };
static cpir_server* g_srv;
static uint64_t g_N;
static uint32_t g_C;
static uint32_t* g_q[kQueries];        /* pageable */
static uint32_t* g_qp[kQueries];       /* page-locked (cpir_host_alloc) */
static uint32_t* g_want[kQueries];
static int g_bad;

typedef struct {
  int id, calls, pinned;
} job_t;

static void* caller(void* arg) {
  const job_t* j = (const job_t*)arg;
  uint32_t* r = (uint32_t*)malloc(4 * (size_t)g_C);
  for (int i = 0; i < j->calls; i++) {
    const int k = (j->id + i) % kQueries;
    CHECK(cpir_server_respond(g_srv, j->pinned ? g_qp[k] : g_q[k], 1, g_N, r));
    if (memcmp(r, g_want[k], 4 * (size_t)g_C) != 0) __atomic_fetch_add(&g_bad, 1, __ATOMIC_RELAXED);
  }
  free(r);
  return NULL;
}

static double run(int threads, int calls, int pinned) {  /* queries per second */
  pthread_t th[64];
  job_t jobs[64];
  const double t0 = now();
  for (int t = 0; t < threads; t++) {
    jobs[t].id = t * 5, jobs[t].calls = calls, jobs[t].pinned = pinned == 2 ? (t & 1) == 0 : pinned; /* 2: every other caller page-locked */
    pthread_create(&th[t], NULL, caller, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  return threads * (double)calls / (now() - t0);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 20;
  const uint64_t value_bytes = argc > 2 ? strtoull(argv[2], NULL, 10) : 1024;
  const uint32_t arity = argc > 3 ? (uint32_t)atoi(argv[3]) : 3;
  const uint64_t n_keys = 1ull << lg;
  uint32_t b = 0, sl = 0, scl = 0;
  CHECK(cpir_find_encoded_db_matrix_element_bit_length(n_keys, &b));
  CHECK(cpir_filter_shape(arity, n_keys, &sl, &scl, &g_N));
  g_C = (uint32_t)cpir_encoded_num_cols(value_bytes, b);
  const uint32_t cf = cpir_compression_factor(b), S = 32 / cf;
  const uint64_t W = (g_N + cf - 1) / cf;
  cpir_device* dev = NULL;
  CHECK(cpir_device_open(0, &dev));
  /* the reference's compressed matrix (C x W words, cf fields of b bits per word), synthetic uniform fields */
  uint32_t* dtc = (uint32_t*)malloc(4 * (size_t)g_C * W);
  if (!dtc) return 1;
  const uint32_t fmask = (1u << b) - 1u;
  for (uint64_t i = 0; i < (uint64_t)g_C * W; i++) {
    const uint64_t x = mix(i);
    uint32_t w = 0;
    for (uint32_t f = 0; f < cf; f++) w |= ((uint32_t)(x >> (16 * f)) & fmask) << (f * S);
    dtc[i] = w;
  }
  /* CPIR_BENCH_HOLES=K: every K-th slot (pseudo-randomly) holds nothing in any column, like the rows of a real encoded database that no
   * key owns (one in nine at arity 3): the server then leaves them out of its image and compacts every query onto the others */
  if (getenv("CPIR_BENCH_HOLES")) {
    const uint64_t K = strtoull(getenv("CPIR_BENCH_HOLES"), NULL, 10);
    for (uint64_t n = 0; K && n < g_N; n++)
      if (mix(n ^ 0xabcdefull) % K == 0)
        for (uint32_t c = 0; c < g_C; c++) dtc[(uint64_t)c * W + n / cf] &= ~(((1u << S) - 1u) << ((n % cf) * S));
  }
  if (g_N % cf) /* fields past the last slot are zero */
    for (uint32_t c = 0; c < g_C; c++) dtc[(uint64_t)c * W + W - 1] &= (1u << ((g_N % cf) * S)) - 1u;
  CHECK(cpir_server_from_compressed(dev, dtc, g_C, g_N, b, &g_srv));
  free(dtc);
  for (int k = 0; k < kQueries; k++) {
    /* (64-byte aligned like the buffers of most allocators for sizes like this; CPIR_BENCH_QUERY_SKEW=8 shifts it like the word array
     * behind the 8-byte header of the wire format) */
    {
      void* raw = NULL;
      const char* skew = getenv("CPIR_BENCH_QUERY_SKEW");
      if (posix_memalign(&raw, 4096, 4 * g_N + 4096) != 0) return 1;
      g_q[k] = (uint32_t*)((char*)raw + (skew ? atoi(skew) : 0));
    }
    void* p = NULL;
    CHECK(cpir_host_alloc(4 * g_N, &p));
    g_qp[k] = (uint32_t*)p;
    for (uint64_t i = 0; i < g_N; i++) g_q[k][i] = (uint32_t)mix(((uint64_t)(k + 1) << 40) + i);
    memcpy(g_qp[k], g_q[k], 4 * g_N);
    g_want[k] = (uint32_t*)malloc(4 * (size_t)g_C);
    CHECK(cpir_server_respond(g_srv, g_q[k], 1, g_N, g_want[k]));
  }
  if (getenv("CPIR_BENCH_INPLACE_SEATS")) CHECK(cpir_tuning_set("respond.inplace_seats", atoi(getenv("CPIR_BENCH_INPLACE_SEATS")))); /* A/B */
  /* 4th argument "tT" (e.g. t16) + 5th 0 / 1 / 2: ONLY T closed-loop callers with pageable / page-locked / mixed queries, three rounds (with
   * CPIR_RESPOND_TRACE=1 the library prints where their time went when the server is released) */
  if (argc > 4 && argv[4][0] == 't') {
    const int T = atoi(argv[4] + 1), pinned = argc > 5 ? atoi(argv[5]) : 0;
    if (argc > 6) CHECK(cpir_tuning_set("respond.upload_streams", atoi(argv[6])));
    (void)run(T, 4, pinned);
    printf("{\"callers\": %d, \"pinned\": %d, \"queries_per_sec\": [", T, pinned);
    for (int r = 0; r < 3; r++) printf("%s%.0f", r ? ", " : "", run(T, 960 / T, pinned));
    uint64_t served[CPIR_HOST_PATH_COUNT];
    CHECK(cpir_server_host_path_counts(g_srv, served));
    printf("], \"mismatches\": %d, \"served\": {\"calls\": %llu, \"alone\": %llu, \"in_uploaded_rounds\": %llu, \"uploaded_rounds\": %llu, "
           "\"in_in_place_rounds\": %llu, \"in_place_rounds\": %llu, \"polled_passes_given_up\": %llu}}\n",
           g_bad, (unsigned long long)served[0], (unsigned long long)served[1], (unsigned long long)served[4], (unsigned long long)served[5],
           (unsigned long long)served[6], (unsigned long long)served[7], (unsigned long long)served[3]);
    cpir_server_release(g_srv);
    cpir_device_close(dev);
    return g_bad ? 2 : 0;
  }
  const int lone_only = argc > 4 && !strcmp(argv[4], "lone"); /* 4th argument "lone": the single-caller latencies only */
  if (lone_only && argc > 5) CHECK(cpir_tuning_set("respond.helper_spin_us", atoi(argv[5]))); /* ... 5th: respond.helper_spin_us */
  const int lone = 200;
  /* the lone caller is the thread that allocated the buffers and built the server (memory placed where it runs) */
  double lone_pageable, lone_pinned;
  {
    job_t j = {0, lone, 0};
    double t0 = now();
    caller(&j);
    lone_pageable = (now() - t0) * 1e6 / lone;
    j.pinned = 1;
    t0 = now();
    caller(&j);
    lone_pinned = (now() - t0) * 1e6 / lone;
  }
  /* (the lone caller comes FIRST: after a burst of concurrent callers the server expects company for a while -- its estimate of the
   * recent concurrency decays one step per 8 calls -- and serves a lone caller through the upload path meanwhile: 270 instead of 230 us) */
  if (!lone_only) (void)run(8, 4, 0); /* first use of every arena */
  printf("{\"n_keys_log2\": %d, \"value_bytes\": %llu, \"arity\": %u, \"N\": %llu, \"C\": %u, \"b\": %u, \"query_bytes\": %llu, "
         "\"one_caller_us_per_query\": %.1f, \"one_caller_pinned_query_us_per_query\": %.1f",
         lg, (unsigned long long)value_bytes, arity, (unsigned long long)g_N, g_C, b, (unsigned long long)(4 * g_N), lone_pageable, lone_pinned);
  const int threads[] = {2, 4, 8, 16};
  for (int i = 0; i < 4 && !lone_only; i++) {
    const int T = threads[i], calls = 1920 / T;
    const double qp = run(T, calls, 0), qn = run(T, calls, 1);
    printf(", \"callers_%d_queries_per_sec\": %.0f, \"callers_%d_pinned_queries_per_sec\": %.0f", T, qp, T, qn);
  }
  printf(", \"mismatches\": %d, \"note\": \"cpir_server_respond from C threads on one handle (closed loop), synthetic packed matrix, query pool of %d\"}\n",
         g_bad, kQueries);
  cpir_server_release(g_srv);
  cpir_device_close(dev);
  return g_bad ? 2 : 0;
}
