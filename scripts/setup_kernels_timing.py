#!/usr/bin/env python3
"""Times the two offline kernels alone at a BASELINE config (synthetic A and D resident in HBM): the hint matmul through
cpir_op_mat_x_mat and transpose+compress through cpir_op_transpose_compress.  Run under `rocprofv3 --kernel-trace --stats` for the
per-kernel split (rhs_split_kernel / mat_x_mat_mfma_kernel / hint_fixup_kernel).   usage: setup_kernels_timing.py [cfg2] [reps] [mfma]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

if os.environ.get("CPIR_MM_ABLATE"):  # the ablation switch exists only in the diagnosis build of the library (`make diag`)
    from chalametpir_amd import _native  # noqa: E402

    _native.use_diag_build()
import chalametpir_amd as cp  # noqa: E402
from bench import CONFIGS  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_keys, arity, value_bytes = CONFIGS[cfg]
b = cp.find_encoded_db_matrix_element_bit_length(n_keys)
_, _, N = cp.filter_shape(arity, n_keys)
C = cp.encoded_num_cols(value_bytes, b)
dev = cp.Device(0)
cp.tuning_set("matmul.mfma", mfma)
stream = torch.cuda.current_stream()
R = 1774
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
A = torch.empty((R, N), dtype=torch.int32, device="cuda")
dev.synth_fill(A, R * N, 0xA, stream=stream)
M = torch.empty((R, C), dtype=torch.int32, device="cuda")
L = cp.dtc_layout_for(N, C, b)
dtc = torch.empty(int(L.total_words) + int(L.rows_padded) + 64, dtype=torch.int32, device="cuda")


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


mm = timed(lambda: dev.mat_x_mat(A, D, M, R, N, C, rhs_max_bits=16, stream=stream))
for bits in [int(x) for x in os.environ.get("CPIR_MM_ABLATE", "").split(",") if x]:  # (the diagnosis build only: see the imports)
    cp.tuning_set("matmul.ablate", bits)
    t = timed(lambda: dev.mat_x_mat(A, D, M, R, N, C, rhs_max_bits=16, stream=stream))
    print(f"  ablate {bits:2d} (1 no MFMA, 2 no A conversion, 4 no A loads, 8 no D DMA): {t:.3f} ms", flush=True)
    cp.tuning_set("matmul.ablate", 0)
pk = timed(lambda: dev.transpose_compress(D, L, dtc, stream=stream))
print(f"{cfg}: N={N} C={C} b={b}  {cp.mat_x_mat_kernel_name(16)}: {mm:.3f} ms = {R * N * C / mm / 1e9:.1f} TMAC/s;  pack: {pk:.3f} ms", flush=True)
