#!/usr/bin/env python3
"""Fused batches at a BASELINE shape: passes of 4 through the step-major kernel (respond.ks_major = 2) against the wide kernel (up to
24 queries per stream of the database), microseconds per query from events, responses compared with each other.
   python scripts/wide_ab.py [N C b [HOLES]]   (default: 2^20 keys x 1 kB = 1179648 x 940, b = 9; HOLES: fraction of the rows set to zero --
                                                a real encoded database has 0.111 -- which the server then leaves out of its image)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402

N, C, b = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1179648, 940, 9)
dev = cp.Device(0)
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
if len(sys.argv) >= 5 and float(sys.argv[4]) > 0:
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    D[torch.rand(N, device="cuda", generator=g) < float(sys.argv[4])] = 0
srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
torch.cuda.synchronize()
print("slots served:", srv.slots_served(), flush=True)
del D
NQ = 96
q = torch.empty((NQ, N), dtype=torch.int32, device="cuda")
for i in range(NQ):
    dev.synth_fill(q, N, 0x1000 + i, offset_words=i * N, stream=stream)
ref = torch.empty((NQ, C), dtype=torch.int32, device="cuda")
cp.tuning_set("respond.ks_major", 2)
srv.respond_batch_device(q, NQ, ref, stream=stream)
torch.cuda.synchronize()


def timed(k, reps):
    r = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
    for _ in range(2):
        srv.respond_batch_device(q[:k], k, r, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        srv.respond_batch_device(q[:k], k, r, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    same = bool(torch.equal(r, ref[:k]))
    return e0.elapsed_time(e1) * 1e3 / reps, same


reps = max(3, int(20 * 1.2e9 / (N * C * b / 8)))
for k in (1, 2, 4, 5, 6, 8, 12, 13, 16, 20, 24, 28, 32, 36, 48, 64, 72, 96):
    row = [f"batch {k:3d}:"]
    for ks_major in (2, 1):
        cp.tuning_set("respond.ks_major", ks_major)
        us, same = timed(k, reps)
        row.append(f"{'step-major' if ks_major == 2 else 'wide'}: {us:8.1f} us = {us / k:6.2f} us/query{'' if same else '  RESPONSES DIFFER'}")
    print("   ".join(row), flush=True)
cp.tuning_set("respond.ks_major", 1)
