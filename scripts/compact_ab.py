#!/usr/bin/env python3
"""A database with the zero rows of a real encoded one (2^20 keys x 1 kB: 11.1 % of the rows, at random) served from device queries:
the headline's loop (32 queries a launch, one pass each) with and without the slot map, per-kernel times from events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402

N, C, b = 1179648, 940, 9
dev = cp.Device(0)
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
g = torch.Generator(device="cuda")
g.manual_seed(1)
perm = torch.randperm(N, device="cuda", generator=g)
targets = [int(x) for x in sys.argv[1:]] or [1048576]
q = torch.empty((64, N), dtype=torch.int32, device="cuda")
for i in range(64):
    dev.synth_fill(q, N, 0x1000 + i, offset_words=i * N, stream=stream)
r = torch.empty((32, C), dtype=torch.int32, device="cuda")
cp.tuning_set("respond.batch_fusion", 0)
res = {}
Dfull = D
for mode, kept in [(0, N)] + [(1, t) for t in targets] + [(0, N)]:
    D = Dfull.clone()
    D[perm[kept:]] = 0
    cp.tuning_set("layout.compact_slots", mode)
    srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
    torch.cuda.synchronize()
    for k in range(3):
        srv.respond_batch_device(q[32 * (k % 2):32 * (k % 2) + 32], 32, r, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for k in range(20):
        srv.respond_batch_device(q[32 * (k % 2):32 * (k % 2) + 32], 32, r, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (20 * 32)
    print(f"compact_slots={mode}: served {srv.slots_served()} = {-(-srv.slots_served()[0] // 512)} steps, {us:.2f} us per query ({us / srv.slots_served()[0] * 1e6:.2f} ps per slot)", flush=True)
    srv.close()
    del D
