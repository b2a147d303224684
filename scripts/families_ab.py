#!/usr/bin/env python3
"""A launch of many independent passes (one query each -- the headline's loop and every multi-GPU shard): 32 passes a launch on the whole
database and on 1/2, 1/4, 1/8 of its slots, in slice and in interleaved order, on the wide and on the step-major kernel; microseconds per
query from events, responses compared with each other.
   python scripts/families_ab.py [N C b [passes]]   (default: 2^20 keys x 1 kB = 1179648 x 940, b = 9, 32 passes)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402

N0, C, b = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1179648, 940, 9)
P = int(sys.argv[4]) if len(sys.argv) >= 5 else 32
dev = cp.Device(0)
stream = torch.cuda.current_stream()
cp.tuning_set("respond.batch_fusion", 0)

# (round 5, before the tile-major kernel of rounds 1-4 was deleted, this script also timed it: profiles/r5_families_ab.txt)
MODES = [
    ("wide, order by size (as dispatched)", {}),
    ("wide, slice + nt", {"respond.interleave_passes": 0}),
    ("wide, interleaved + cached", {"respond.interleave_passes": 1}),
    ("step-major, slice + nt", {"respond.ks_major": 2, "respond.interleave_passes": 0}),
]


def apply(tune):
    cp.tuning_reset()
    cp.tuning_set("respond.batch_fusion", 0)
    for k, v in tune.items():
        try:
            cp.tuning_set(k, v)
        except Exception:  # noqa: BLE001 -- a key this build does not have
            return False
    return True


for frac in (1, 2, 4, 8):
    N = (N0 // frac) // 1536 * 1536
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
    srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
    torch.cuda.synchronize()
    del D
    q = torch.empty((2 * P, N), dtype=torch.int32, device="cuda")
    for i in range(2 * P):
        dev.synth_fill(q, N, 0x1000 + i, offset_words=i * N, stream=stream)
    ref = None
    reps = max(4, int(6 * frac))
    print(f"--- 1/{frac} of the slots: N = {N}, image {srv.layout.total_words * 4 / 1e6:.0f} MB, {P} passes a launch", flush=True)
    for name, tune in MODES:
        if not apply(tune):
            continue
        r = torch.full((P, C), -1, dtype=torch.int32, device="cuda")
        for k in range(3):
            srv.respond_batch_device(q[(k % 2) * P:(k % 2 + 1) * P], P, r, stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(reps):
            srv.respond_batch_device(q[(k % 2) * P:(k % 2 + 1) * P], P, r, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        srv.respond_batch_device(q[:P], P, r, stream=stream)
        torch.cuda.synchronize()
        if ref is None:
            ref = r.clone()
        same = bool(torch.equal(r, ref))
        us = e0.elapsed_time(e1) * 1e3 / (reps * P)
        print(f"   {name:45s} {us:8.2f} us per query{'' if same else '   RESPONSES DIFFER'}", flush=True)
    del srv, q
    torch.cuda.empty_cache()
cp.tuning_reset()
