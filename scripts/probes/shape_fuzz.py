#!/usr/bin/env python3
"""Shape fuzz (GPU box): random small databases -- every bit length 4..14, N from 1 slot up (dense around 1, 63-65, 511-513, multiples of 1536),
C from 1 column up, with and without all-zero rows -- through Server::setup from the matrix (hint AND packed image against the oracle),
one respond, a fused batch of 1..30 queries and the wire-bytes entry point, all against the oracle.   python3 scripts/probes/shape_fuzz.py [cases [seed]]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chalametpir_amd as cp  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker only)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = cp.Device(0)
stream = torch.cuda.current_stream()
seed = bytes(range(32))
bad = 0
for case in range(cases):
    b = int(rng.integers(4, 15))
    kind = int(rng.integers(0, 6))
    N = int([rng.integers(1, 8), rng.integers(60, 70), rng.integers(505, 520), rng.integers(1, 3) * 1536 + rng.integers(-3, 4),
             rng.integers(1, 5000), rng.integers(1, 300)][kind])
    C = int([1, rng.integers(1, 20), rng.integers(14, 19), rng.integers(1, 300), 16, rng.integers(120, 140)][int(rng.integers(0, 6))])
    D = rng.integers(0, 1 << b, size=(N, C), dtype=np.uint64).astype(np.uint32)
    if rng.integers(0, 3) == 0 and N > 8:
        D[rng.random(N) < 0.3] = 0
    try:
        srv, hint = cp.Server.setup_from_matrix(seed, D, b, device=dev)
        want_hint, want_dtc = orc.server_setup_from_matrix(seed, D, b)
        ok = np.array_equal(hint, want_hint) and np.array_equal(srv.export_compressed(), want_dtc)
        nb = int(rng.integers(1, 31))
        qs = rng.integers(0, 1 << 32, size=(nb, N), dtype=np.uint64).astype(np.uint32)
        want = np.stack([orc.row_vector_x_compressed_transposed_matrix(qs[i], want_dtc, N, b)[0] for i in range(nb)])
        q_dev = torch.from_numpy(qs.view(np.int32)).cuda()
        r_dev = torch.full((nb, C), -1, dtype=torch.int32, device="cuda")
        srv.respond_batch_device(q_dev, nb, r_dev, stream=stream)
        r1 = torch.full((C,), -1, dtype=torch.int32, device="cuda")
        srv.respond_device(q_dev[nb - 1], r1, stream=stream)
        torch.cuda.synchronize()
        ok = ok and np.array_equal(r_dev.cpu().numpy().view(np.uint32), want) and np.array_equal(r1.cpu().numpy().view(np.uint32), want[nb - 1])
        wire = np.array([1, N], dtype="<u4").tobytes() + qs[0].tobytes()
        ok = ok and srv.respond(wire) == orc.server_respond(want_dtc, N, b, wire)
        srv.close()
    except Exception as exc:  # noqa: BLE001
        ok = False
        print(f"case {case}: b={b} N={N} C={C}: {exc!r}", flush=True)
    if not ok:
        bad += 1
        print(f"case {case}: MISMATCH b={b} N={N} C={C}", flush=True)
    if case % 50 == 49:
        print(f"{case + 1} cases, {bad} bad", flush=True)
print(f"shape fuzz: {cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
