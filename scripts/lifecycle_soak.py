#!/usr/bin/env python3
"""Lifecycle soak (GPU box): servers of a few shapes are set up, asked through the host entry point by 1 lone caller (query read in
place: pageable -> polled launch, page-locked -> straight from the buffer) and by bursts of concurrent callers with fresh random queries,
cloned, released and destroyed, over and over for --seconds.  Every response is compared with exact 64-bit sums from the unpacked matrix.
Looks for what a long-lived serving process would hit: arenas reused across phases, first use after idle, release while nothing is in
flight, device memory returned (the free-memory figure at the end is compared with the one at the start)."""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chalametpir_amd as cp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
args = ap.parse_args()
dev = cp.Device(0)
stream = torch.cuda.current_stream()
rng = np.random.default_rng(5)
shapes = [((1 << 19) + 4096 * 3, 24, 9), ((1 << 20) + 512 * 7 + 128, 12, 10), (77824, 846, 10), (600_000, 40, 12)]
free0 = torch.cuda.mem_get_info()[0]
t_end = time.time() + args.seconds
rounds = checked = 0
while time.time() < t_end:
    N, C, b = shapes[rounds % len(shapes)]
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    dev.synth_fill(D, N * C, 0xD00 + rounds, mask=(1 << b) - 1, stream=stream)
    srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
    torch.cuda.synchronize()
    D64 = D.to(torch.int64)

    def exact(q):
        qq = torch.from_numpy(q.astype(np.int64)).cuda()
        return ((qq[:, None] * D64).sum(dim=0) & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)

    pin = cp.PinnedArray(N)
    bad = []

    def caller(k, n_calls, use_pin):
        r = np.random.default_rng(1000 * rounds + k)
        for _ in range(n_calls):
            q = r.integers(0, 1 << 32, size=N, dtype=np.uint64).astype(np.uint32)
            if use_pin:
                pin.array[:] = q
                got = srv.respond_array(pin.array)
            else:
                got = srv.respond_array(q)
            if not np.array_equal(got, exact(q)):
                bad.append((rounds, k))

    for phase in range(3):
        caller(0, 3, False)  # lone, pageable
        caller(1, 3, True)   # lone, page-locked
        ts = [threading.Thread(target=caller, args=(10 + k, 2, False)) for k in range(6)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        checked += 6 + 12
    clone = srv.clone() if hasattr(srv, "clone") else None
    srv.close()
    if clone is not None:
        q = rng.integers(0, 1 << 32, size=N, dtype=np.uint64).astype(np.uint32)
        if not np.array_equal(clone.respond_array(q), exact(q)):
            bad.append((rounds, "clone"))
        checked += 1
        clone.close()
    pin.close()
    if bad:
        print("MISMATCH", bad, flush=True)
        sys.exit(1)
    del D, D64
    rounds += 1
    if rounds % 8 == 0:
        print(f"{rounds} servers, {checked} responses ok", flush=True)
torch.cuda.synchronize()
torch.cuda.empty_cache()
time.sleep(1.0)  # background disposal
free1 = torch.cuda.mem_get_info()[0]
print(f"lifecycle soak ok: {rounds} servers, {checked} responses; device memory free {free0 / 1e9:.2f} GB before, {free1 / 1e9:.2f} GB after", flush=True)
