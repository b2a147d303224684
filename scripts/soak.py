#!/usr/bin/env python3
"""Soak test (GPU box): for --seconds, random full-size queries through every respond path of the planar server -- device single,
device batches (fused and unfused, both pass orders), host bytes (pageable and page-locked; 1..8 and 20 concurrent callers, crews of 2-4), a 5-shard
in-process group -- each compared with exact 64-bit sums computed by torch from the unpacked matrix.  Prints a count and exits 1 on
the first mismatch."""
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
ap.add_argument("--holes", type=float, default=0.0, help="fraction of the rows of D set to zero (a real encoded database has 0.11): the servers "
                                                         "then keep only the other rows resident and every path goes through the slot map")
args = ap.parse_args()
dev = cp.Device(0)
N, C, b = 1179648, 940, 9
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
if args.holes > 0:
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    D[torch.rand(N, device="cuda", generator=g) < args.holes] = 0
srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
print("slots served:", srv.slots_served(), flush=True)
# (a handle of its own for the crews of two to four: a handle remembers the 20 callers of the burst for a hundred calls and would send the
# crews through the upload path meanwhile)
srv_few = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
grp, _ = cp.Server.setup_from_matrix(bytes(32), D.cpu().numpy().view(np.uint32), b, devices=[dev] * 5)
torch.cuda.synchronize()


def exact(qs):
    out = []
    for q in qs:
        acc = torch.zeros(C, dtype=torch.int64, device="cuda")
        for lo in range(0, N, 1 << 17):
            qq = q[lo:lo + (1 << 17)].to(torch.int64) & 0xFFFFFFFF
            acc += (qq[:, None] * D[lo:lo + (1 << 17)].to(torch.int64)).sum(dim=0)
        out.append((acc & 0xFFFFFFFF).cpu().numpy().astype(np.uint32))
    return out


t_end = time.time() + args.seconds
rounds = checks = 0
seed = 0x9000
while time.time() < t_end:
    k = 8 if rounds % 2 == 0 else 13  # (13: a pass of 12 on three row sets + one more)
    Q = torch.empty((k, N), dtype=torch.int32, device="cuda")
    for i in range(k):
        dev.synth_fill(Q, N, seed + i, offset_words=i * N, stream=stream)
    seed += k
    want = exact([Q[i] for i in range(k)])
    got = []
    R = torch.empty((k, C), dtype=torch.int32, device="cuda")
    for fusion in (1, 0):
        for order in (0, 1):
            cp.tuning_set("respond.batch_fusion", fusion)
            cp.tuning_set("respond.interleave_passes", order)
            R.fill_(-1)
            srv.respond_batch_device(Q, k, R, stream=stream)
            torch.cuda.synchronize()
            got.append(("batch", fusion, order, R.cpu().numpy().view(np.uint32).copy()))
    cp.tuning_set("respond.batch_fusion", 1)
    cp.tuning_set("respond.interleave_passes", -1)
    # the in-process group asked on DEVICE pointers (peer exchange + sum kernel on the root)
    R.fill_(-1)
    grp.respond_batch_device(Q, k, R, stream=stream)
    torch.cuda.synchronize()
    got.append(("group-device", 1, -1, R.cpu().numpy().view(np.uint32).copy()))
    for name, fusion, order, r in got:
        for i in range(k):
            if not np.array_equal(r[i], want[i]):
                print("MISMATCH", name, fusion, order, i, seed)
                sys.exit(1)
            checks += 1
    hq = [Q[i].cpu().numpy().view(np.uint32) for i in range(k)]
    pin = cp.PinnedArray(N)
    res = [None] * k

    def call(i):
        if i % 2:
            res[i] = srv.respond_array(hq[i])
        else:
            res[i] = grp.respond_array(hq[i])

    ts = [threading.Thread(target=call, args=(i,)) for i in range(k)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    pin.array[:] = hq[0]
    res.append(srv.respond_array(pin.array))
    res.append(grp.respond_array(pin.array))
    pin.close()
    for i, r in enumerate(res):
        w = want[i] if i < k else want[0]
        if not np.array_equal(r, w):
            print("MISMATCH host", i, seed)
            sys.exit(1)
        checks += 1
    # 20 concurrent callers on the ONE handle, pageable and page-locked buffers mixed, three calls each: every arena of the coalescing
    # front end fills up (spread admission, full arenas, lone stragglers at the end)
    pins = [cp.PinnedArray(N) for _ in range(4)]
    for j, pa in enumerate(pins):
        pa.array[:] = hq[j]
    bad = []

    def hammer(t):
        for rep in range(3):
            i = (t + rep) % k
            buf = pins[i].array if (i < 4 and t % 2 == 0) else hq[i]
            if not np.array_equal(srv.respond_array(buf), want[i]):
                bad.append((t, rep, i))

    ts = [threading.Thread(target=hammer, args=(t,)) for t in range(20)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    # crews of two to four callers, started together, four calls each: the in-place rounds (respond.inplace_seats) -- page-locked queries
    # read where they lie, pageable ones copied in under a pass that polls every seat; all page-locked, all pageable, mixed
    for crew in (2, 3, 4):
        start = threading.Barrier(crew)

        def few(t):
            start.wait()
            for rep in range(4):
                i = (t + rep) % min(k, 4)
                kind = (rounds + crew) % 3  # 0 page-locked, 1 pageable, 2 mixed
                buf = pins[i].array if (kind == 0 or (kind == 2 and t % 2 == 0)) else hq[i]
                if not np.array_equal(srv_few.respond_array(buf), want[i]):
                    bad.append((crew, t, rep, i))

        ts = [threading.Thread(target=few, args=(t,)) for t in range(crew)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    for pa in pins:
        pa.close()
    if bad:
        print("MISMATCH concurrent", bad[:5], seed)
        sys.exit(1)
    checks += 60 + 4 * (2 + 3 + 4)
    rounds += 1
print(f"soak ok: {rounds} rounds, {checks} responses checked against exact 64-bit sums")
print("how the host callers of the server were served:", srv.host_path_counts())
print("... and the crews of two to four, on a handle of their own:", srv_few.host_path_counts())
