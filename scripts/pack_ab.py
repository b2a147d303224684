#!/usr/bin/env python3
"""A/B of the two planar pack kernels (tuning "pack.rows": 0 = 64-column waves, 1 = whole rows per block) at a BASELINE config: time per
launch (HIP events on the launch stream, D resident in HBM) and bit-identity of the two images.   usage: pack_ab.py [cfg2] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402
from bench import CONFIGS  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_keys, arity, value_bytes = CONFIGS[cfg]
b = cp.find_encoded_db_matrix_element_bit_length(n_keys)
_, _, N = cp.filter_shape(arity, n_keys)
C = cp.encoded_num_cols(value_bytes, b)
cf = 2 if b >= 11 else (3 if b >= 9 else 4)
dev = cp.Device(0)
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
L = cp.dtc_layout_for(N, C, b)
imgs = {}
alg = 4 * N * C + 4 * C * -(-N // cf)
for mode in (0, 1):
    cp.tuning_set("pack.rows", mode)
    dtc = torch.zeros(int(L.total_words), dtype=torch.int32, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    dev.transpose_compress(D, L, dtc, or_of_entries=flag, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        dev.transpose_compress(D, L, dtc, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    imgs[mode] = dtc
    moved = 4 * N * C + 4 * int(L.total_words)
    print(f"{cfg} N={N} C={C} b={b} pack.rows={mode}: {ms:.3f} ms  algorithmic {alg / ms / 1e6:.0f} GB/s = {alg / ms / 1e6 / 8000:.3f} of 8 TB/s; "
          f"moved {moved / ms / 1e6:.0f} GB/s; OR of entries {int(flag.item()):#x}", flush=True)
print("images identical:", bool(torch.equal(imgs[0], imgs[1])), flush=True)
