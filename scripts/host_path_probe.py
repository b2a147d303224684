#!/usr/bin/env python3
"""Closed-loop callers on ONE server handle through the host-pointer C ABI (cpir_server_respond), at a BASELINE config.
usage: host_path_probe.py <cfg> <threads> <per-thread calls> <pinned 0|1> [zero-copy 0|1]   (CPIR_RESPOND_TRACE=1 prints the phase split at exit)
The expected responses are always computed with respond.host_zero_copy=0 (upload + tile-major kernel)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402
from bench import CONFIGS  # noqa: E402

cfg, threads, per, pinned = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
zero_copy = int(sys.argv[5]) if len(sys.argv) > 5 else 1
n_keys, arity, value_bytes = CONFIGS[cfg]
b = cp.find_encoded_db_matrix_element_bit_length(n_keys)
_, _, N = cp.filter_shape(arity, n_keys)
C = cp.encoded_num_cols(value_bytes, b)
dev = cp.Device(0)
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
torch.cuda.synchronize()
del D
rng = np.random.default_rng(1)
qs = [rng.integers(0, 1 << 32, size=N, dtype=np.uint64).astype(np.uint32) for _ in range(threads)]
pins = []
if pinned:
    for q in qs:
        p = cp.PinnedArray(N)
        p.array[:] = q
        pins.append(p)
bufs = [p.array for p in pins] if pinned else qs
for kv in filter(None, os.environ.get("CPIR_TUNE", "").split(",")):  # e.g. CPIR_TUNE=respond.host_hand_over=0
    k, v = kv.split("=")
    cp.tuning_set(k, int(v))
cp.tuning_set("respond.host_zero_copy", 0)
want = [srv.respond_array(q) for q in qs]
cp.tuning_set("respond.host_zero_copy", zero_copy)
bad = [0]


def work(k):
    for _ in range(per):
        r = srv.respond_array(bufs[k])
        if not np.array_equal(r, want[k]):
            bad[0] += 1


ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
t0 = time.perf_counter()
[t.start() for t in ts]
[t.join() for t in ts]
dt = time.perf_counter() - t0
print(f"{cfg} threads={threads} pinned={pinned} zero_copy={zero_copy}: {threads * per / dt:.0f} queries/s, {dt / per * 1e6:.0f} us per call per thread, mismatches {bad[0]}", flush=True)
srv.close()
