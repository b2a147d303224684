#!/usr/bin/env python3
"""Host-pointer respond on a GROUP handle (cpir_server_setup_multi, here G shards on ONE device) against a single server: us per query for one
caller, pageable and page-locked, and queries/s for 4 concurrent callers.   python scripts/probes/group_host_latency.py [G]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, C, b = 1179648, 940, 9
dev = cp.Device(0)
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
torch.cuda.synchronize()
one = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
torch.cuda.synchronize()
grp, _ = cp.Server.setup_from_matrix(bytes(32), D.cpu().numpy().view(np.uint32), b, devices=[dev] * G)
del D
rng = np.random.default_rng(5)
qs = [rng.integers(0, 1 << 32, size=N, dtype=np.uint32) for _ in range(4)]
pins = [cp.PinnedArray(N) for _ in range(4)]
for pa, q in zip(pins, qs):
    pa.array[:] = q
want = [one.respond_array(q) for q in qs]
for name, srv in (("single server", one), (f"group of {G}", grp)):
    for kind, bufs in (("pageable", qs), ("page-locked", [pa.array for pa in pins])):
        for i in range(8):
            assert np.array_equal(srv.respond_array(bufs[i % 4]), want[i % 4])
        t0 = time.perf_counter()
        reps = 200
        for i in range(reps):
            srv.respond_array(bufs[i % 4])
        us = (time.perf_counter() - t0) / reps * 1e6

        def loop(t, n=100):
            for i in range(n):
                srv.respond_array(bufs[(t + i) % 4])

        ts = [threading.Thread(target=loop, args=(t,)) for t in range(4)]
        t0 = time.perf_counter()
        [t.start() for t in ts]
        [t.join() for t in ts]
        qps = 400 / (time.perf_counter() - t0)
        print(f"{name:14s} {kind:11s}: one caller {us:7.1f} us per query   4 callers {qps:7.0f} queries/s", flush=True)
