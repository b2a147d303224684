#!/usr/bin/env python3
"""How fast a fused pass reads its queries IN PLACE from page-locked host memory (no upload): cpir_server_respond_batch_device handed the
device-visible address of a pinned host block, 1 .. 8 queries, through the step-major kernel (<= 4 per pass) and the wide kernel; us per
launch from events, responses compared with the same queries resident in HBM.   python scripts/probes/inplace_batch.py [N C b]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402

N, C, b = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1179648, 940, 9)
dev = cp.Device(0)
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
torch.cuda.synchronize()
del D
NQ = 8
q = torch.empty((NQ, N), dtype=torch.int32, device="cuda")
for i in range(NQ):
    dev.synth_fill(q, N, 0x1000 + i, offset_words=i * N, stream=stream)
torch.cuda.synchronize()
q_host = torch.empty((NQ, N), dtype=torch.int32, pin_memory=True)
q_host.copy_(q)
ref = torch.empty((NQ, C), dtype=torch.int32, device="cuda")
srv.respond_batch_device(q, NQ, ref, stream=stream)
torch.cuda.synchronize()


def timed(src, k, reps=20):
    r = torch.full((k, C), -1, dtype=torch.int32, device="cuda")
    for _ in range(3):
        srv.respond_batch_device(src, k, r, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        srv.respond_batch_device(src, k, r, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, bool(torch.equal(r, ref[:k]))


for ks_major, name in ((2, "step-major"), (3, "step-major, strided + far"), (1, "wide")):
    cp.tuning_set("respond.ks_major", ks_major)
    for k in (1, 2, 3, 4, 6, 8):
        if ks_major >= 2 and k > 4:
            continue
        t_dev, ok_d = timed(q, k)
        t_host, ok_h = timed(q_host, k)
        gbps = k * N * 4 / t_host / 1e3
        print(f"{name:26s} batch {k}: resident {t_dev:7.1f} us   in place {t_host:7.1f} us = {gbps:5.1f} GB/s over the link"
              f"{'' if ok_d and ok_h else '   RESPONSES DIFFER'}", flush=True)
cp.tuning_set("respond.ks_major", 1)
