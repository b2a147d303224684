#!/usr/bin/env python3
"""Tuning aid (run on the GPU box): respond time per query against the number of slots N, to see whether the distance between
the column-tile streams of the planar layout (ceil(N/512) super-tiles) matters to the HBM channels.
Usage: python scripts/stride_probe.py N [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chalametpir_amd as cp  # noqa: E402

C, b, passes = 940, 9, 16
dev = cp.Device(0)
cp.tuning_set("respond.batch_fusion", 0)
stream = torch.cuda.current_stream()
for N in [int(x) for x in sys.argv[1:]]:
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
    srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
    del D
    q = torch.empty((passes, N), dtype=torch.int32, device="cuda")
    for i in range(passes):
        dev.synth_fill(q, N, 0x1000 + i, offset_words=i * N, stream=stream)
    r = torch.empty((passes, C), dtype=torch.int32, device="cuda")
    for _ in range(2):
        srv.respond_batch_device(q, passes, r, stream=stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    reps = 4
    for _ in range(reps):
        srv.respond_batch_device(q, passes, r, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * passes)
    gb = srv.layout.total_words * 4 / 1e9
    print(f"N={N} steps={-(-N // 512)} image {gb:.3f} GB: {us:.1f} us per query = {gb / us * 1e3:.0f} GB/s", flush=True)
    srv.close()
    del q, r
    torch.cuda.empty_cache()
