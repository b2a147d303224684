#!/usr/bin/env python3
"""Diagnosis: the wide pass with parts switched off (CPIR_WIDE_ABLATE bit mask, results wrong): 1 no rebuild of the A fragments per step,
2 no flush of the responses, 4 no MFMAs (the stream alone), 8 the row sets two at a time on v_mfma_i32_32x32x32_i8 (same byte products, half
the fragment reads and operand bytes: what a 32-column image layout would do to the matrix cores' and the LDS's share), 16 all-zero A
fragments (the same instructions, operands that toggle nothing: is the pass bound by the power the matrix cores draw?), 32 / 64 the SAME B
operands / A fragment for all eight k-blocks of a row set (everything still loaded and waited for: does it help when only one operand changes
from instruction to instruction?), 128 the sixth row set of a pass of 24 without its high-plane MFMAs (11 instead of 12 matrix instructions per
k-block: the upper bound of what high-plane fragments with 3 rows per query could save).  One process per setting (the mask is read once).
   python scripts/wide_ablate.py            -> runs itself once per mask and prints microseconds per launch for batches of 16 and 24"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch

    from chalametpir_amd import _native

    _native.use_diag_build()  # CPIR_WIDE_ABLATE exists only in the diagnosis build of the library (`make diag`)
    import chalametpir_amd as cp

    N, C, b = 1179648, 940, 9
    dev = cp.Device(0)
    stream = torch.cuda.current_stream()
    D = torch.empty((N, C), dtype=torch.int32, device="cuda")
    dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
    srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
    torch.cuda.synchronize()
    del D
    q = torch.empty((48, N), dtype=torch.int32, device="cuda")
    for i in range(48):
        dev.synth_fill(q, N, 0x1000 + i, offset_words=i * N, stream=stream)
    out = []
    for k in (16, 24, 48):
        r = torch.empty((k, C), dtype=torch.int32, device="cuda")
        for _ in range(24):  # (the clocks of a device that has just idled take a few milliseconds to settle)
            srv.respond_batch_device(q[:k], k, r, stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(20):
            srv.respond_batch_device(q[:k], k, r, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        out.append(f"batch {k}: {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us")
    print(f"ablate {os.environ.get('CPIR_WIDE_ABLATE', '0'):>2}:  " + "   ".join(out), flush=True)
else:
    for mask in [int(x) for x in os.environ.get('CPIR_ABLATE_MASKS', '0,1,2,4,3,7,8,16,32,64,0,8,16,32,64').split(',')]:  # (8: the 32x32x32 emulation -- same byte products, half the fragment reads)
        env = dict(os.environ, CPIR_WIDE_ABLATE=str(mask))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
