#!/usr/bin/env python3
"""The fused-batch loop alone, for counter passes (scripts/pmc_wide.sh): launches of `batch` queries (default 48 = two wide passes of 24) on
the 2^20-key x 1 kB shape, nothing else on the device.   python3 scripts/wide_loop.py [batch [launches]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 48
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
N, C, b = 1179648, 940, 9
dev = cp.Device(0)
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
torch.cuda.synchronize()
del D
q = torch.empty((batch, N), dtype=torch.int32, device="cuda")
for i in range(batch):
    dev.synth_fill(q, N, 0x1000 + i, offset_words=i * N, stream=stream)
r = torch.empty((batch, C), dtype=torch.int32, device="cuda")
for _ in range(launches):
    srv.respond_batch_device(q, batch, r, stream=stream)
torch.cuda.synchronize()
print("done", batch, launches, cp.respond_batch_pass_width(srv.physical_layout, batch))
