#!/usr/bin/env python3
"""Lone caller through cpir_server_respond at cfg2, as bench.py's respond_host_path measures it: 16 pageable query arrays taken in turn
(each call reads a buffer the caches no longer hold) and one page-locked buffer.  CPIR_RESPOND_TRACE=1 adds the library's phase split."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402

N, C, b = 1179648, 940, 9
dev = cp.Device(0)
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
torch.cuda.synchronize()
del D
rng = np.random.default_rng(1)
qs = [rng.integers(0, 1 << 32, size=N, dtype=np.uint64).astype(np.uint32) for _ in range(16)]
pin = cp.PinnedArray(N)
pin.array[:] = qs[0]
for kv in filter(None, os.environ.get("CPIR_TUNE", "").split(",")):  # e.g. CPIR_TUNE=respond.ks_major=3
    k, v = kv.split("=")
    cp.tuning_set(k, int(v))
only = os.environ.get("CPIR_ONLY", "")  # "cold" / "pinned" / "device": one kind of call only (a kernel trace then shows that kernel alone)
if only:
    r_dev = torch.empty(C, dtype=torch.int32, device="cuda")
    q_dev = torch.from_numpy(qs[0].view(np.int32)).cuda()
    import time as _t
    t0 = _t.perf_counter()
    for i in range(200):
        if only == "cold":
            srv.respond_array(qs[i % 16])
        elif only == "pinned":
            srv.respond_array(pin.array)
        else:
            srv.respond_device(q_dev, r_dev, stream=stream)
            torch.cuda.synchronize()
    print(f"{only}: {(_t.perf_counter() - t0) / 200 * 1e6:.1f} us per call", flush=True)
    pin.close()
    srv.close()
    sys.exit(0)
for rep in range(3):
    for q in qs[:4]:
        srv.respond_array(q)
    t0 = time.perf_counter()
    for i in range(128):
        srv.respond_array(qs[i % 16])
    cold = (time.perf_counter() - t0) / 128
    t0 = time.perf_counter()
    for i in range(128):
        srv.respond_array(qs[0])
    hot = (time.perf_counter() - t0) / 128
    t0 = time.perf_counter()
    for i in range(128):
        srv.respond_array(pin.array)
    pinned = (time.perf_counter() - t0) / 128
    print(f"lone caller, us per query: pageable cold {cold * 1e6:.1f}  pageable hot {hot * 1e6:.1f}  page-locked {pinned * 1e6:.1f}", flush=True)
pin.close()
srv.close()
