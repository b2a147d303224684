#!/usr/bin/env python3
"""A/B timing of respond kernel variants in ONE process, interleaved rounds (the only comparison that means anything: boxes and runs
differ by a few per cent).   usage: respond_ab.py <cfg> "<key=value,...>" "<key=value,...>" ... [--batch=N] [--rounds=R] [--fusion=F] [--hostq=1]
Each variant is a comma list of cpir_tuning_set settings applied on top of the defaults; prints min / median us per query."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chalametpir_amd as cp  # noqa: E402
from bench import CONFIGS  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
cfg, variants = args[0], args[1:]
batch, rounds = int(opts.get("batch", 32)), int(opts.get("rounds", 7))
fusion = int(opts.get("fusion", 0))
n_keys, arity, value_bytes = CONFIGS[cfg]
b = cp.find_encoded_db_matrix_element_bit_length(n_keys)
_, _, N = cp.filter_shape(arity, n_keys)
C = cp.encoded_num_cols(value_bytes, b)
dev = cp.Device(0)
stream = torch.cuda.current_stream()
D = torch.empty((N, C), dtype=torch.int32, device="cuda")
dev.synth_fill(D, N * C, 0xD, mask=(1 << b) - 1, stream=stream)
srv = cp.Server.from_device_matrix(D, N, C, b, device=dev, stream=stream)
torch.cuda.synchronize()
del D
Q = torch.empty((batch, N), dtype=torch.int32, device="cuda")
for i in range(batch):
    dev.synth_fill(Q, N, 0x1000 + i, offset_words=i * N, stream=stream)
if int(opts.get("hostq", 0)):  # the queries stay in page-locked HOST memory: the kernels read them over the link (respond.ks_major=2 reads each word once)
    Qh = torch.empty((batch, N), dtype=torch.int32, pin_memory=True)
    Qh.copy_(Q)
    torch.cuda.synchronize()
    Q = Qh
R = torch.empty((batch, C), dtype=torch.int32, device="cuda")
cp.tuning_set("respond.batch_fusion", fusion)
defaults = {"respond.ks_major": 1, "respond.nontemporal": 1, "respond.planar_blocks_per_cu": 0, "respond.xcd_split": 1, "respond.interleave_passes": -1}


def apply(v):
    for k, x in defaults.items():
        cp.tuning_set(k, x)
    for kv in filter(None, v.split(",")):
        k, x = kv.split("=")
        cp.tuning_set(k, int(x))


res = {v: [] for v in variants}
ref = None
for rnd in range(rounds + 1):
    for v in variants:
        apply(v)
        srv.respond_batch_device(Q, batch, R, stream=stream)
        torch.cuda.synchronize()
        if ref is None:
            ref = R.clone()
        assert torch.equal(R, ref), v
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(4):
            srv.respond_batch_device(Q, batch, R, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        if rnd:
            res[v].append(e0.elapsed_time(e1) * 1e3 / (4 * batch))
for v in variants:
    print(f"{cfg} batch={batch} fusion={fusion} [{v or 'defaults'}]: min {min(res[v]):.2f}  median {statistics.median(res[v]):.2f} us per query", flush=True)
