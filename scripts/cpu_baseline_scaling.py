"""Thread scaling of the CPU baseline (oracle respond) on the GPU box's host: OMP_NUM_THREADS=<n> python scripts/cpu_baseline_scaling.py"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import oracle as orc
N, C, b = 1179648, 940, 9
W = N // 3
dtc = orc.synth_fill_u32(C * W, 5, 0, 0x3FFFFFFF).reshape(C, W)
dtc = orc.first_touch_copy(dtc)
q = orc.synth_fill_u32(N, 6)
for i in range(3):
    t = time.perf_counter(); orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b); dt = time.perf_counter() - t
print("usable cpus", orc.usable_cpus(), "threads", orc.num_threads(), "ms", round(dt * 1e3, 2), "GB/s", round(dtc.nbytes / dt / 1e9, 1), flush=True)
