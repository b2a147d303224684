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
moved = 4 * N * C + 4 * int(L.total_words)
modes = [int(x) for x in os.environ.get("CPIR_PACK_MODES", "0,1").split(",")]
# the same destination buffers for every mode (where a buffer lies in HBM relative to D shifts the time by several per cent), two of them
bufs = [torch.zeros(int(L.total_words), dtype=torch.int32, device="cuda") for _ in range(2)]
pad = torch.zeros(3 << 20, dtype=torch.int32, device="cuda")  # (keeps the second buffer from sitting at the same offset modulo large powers of two)
bufs.append(torch.zeros(int(L.total_words), dtype=torch.int32, device="cuda"))
for rnd in range(2):
    for mode in modes:
        cp.tuning_set("pack.rows", mode)
        times = []
        for dtc in bufs:
            dev.transpose_compress(D, L, dtc, stream=stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                dev.transpose_compress(D, L, dtc, stream=stream)
            e1.record(stream)
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / reps)
        imgs[mode] = bufs[0].clone()
        ms = min(times)
        print(f"{cfg} N={N} C={C} b={b} pack.rows={mode}: per buffer {' '.join(f'{t:.3f}' for t in times)} ms; best: algorithmic {alg / ms / 1e6:.0f} GB/s = "
              f"{alg / ms / 1e6 / 8000:.3f} of 8 TB/s, moved {moved / ms / 1e6:.0f} GB/s", flush=True)
print("images identical:", all(bool(torch.equal(imgs[modes[0]], imgs[m])) for m in modes), flush=True)
