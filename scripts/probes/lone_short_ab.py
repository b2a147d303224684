#!/usr/bin/env python3
"""One pageable host caller on SHORT queries (below 2^19 words: small databases, the slices of a group's shards): the copy under a polled
launch (one copier) against copy-then-launch (respond.host_fill_timeout_us = 0), us per query.   python scripts/probes/lone_short_ab.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402

dev = cp.Device(0)
stream = torch.cuda.current_stream()
rng = np.random.default_rng(3)
for N, C, b in ((77824, 846, 10), (147456, 940, 9), (303104, 846, 10), (393216, 940, 9)):
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
    srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
    torch.cuda.synchronize()
    del D
    qs = [rng.integers(0, 1 << 32, size=N, dtype=np.uint32) for _ in range(8)]
    want = None
    row = [f"N {N:7d} C {C} b {b}:"]
    for timeout_us in (2000, 0, 2000, 0):
        cp.tuning_set("respond.host_fill_timeout_us", timeout_us)
        for i in range(16):
            r = srv.respond_array(qs[i % 8])
        if want is None:
            want = r.copy()
        assert np.array_equal(srv.respond_array(qs[7]), want)
        t0 = time.perf_counter()
        for i in range(400):
            srv.respond_array(qs[i % 8])
        row.append(f"{'polled' if timeout_us else 'copy first'} {(time.perf_counter() - t0) / 400 * 1e6:6.1f} us")
    print("   ".join(row), flush=True)
    srv.close()
cp.tuning_set("respond.host_fill_timeout_us", 2000)
