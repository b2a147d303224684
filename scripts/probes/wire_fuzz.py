#!/usr/bin/env python3
"""Wire fuzz (GPU box): Server::respond on byte strings a client could send -- well-formed queries, truncated and over-long ones, headers with
rows / cols of 0, 1, N - 1, N, N + 1, 2^31, 2^32 - 1, products that overflow, lengths that do not match -- against the oracle's restatement of
Matrix::from_bytes + the dimension check (matrix.rs:973-1010, 329-331): the same status code, or the same response bytes.
   python3 scripts/probes/wire_fuzz.py [cases [seed]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chalametpir_amd as cp  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker only)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = cp.Device(0)
bad = 0
kinds = {}
for b, N, C in ((9, 1536 + 7, 19), (6, 515, 3), (12, 64, 130)):
    D = rng.integers(0, 1 << b, size=(N, C), dtype=np.uint64).astype(np.uint32)
    srv, _ = cp.Server.setup_from_matrix(bytes(range(32)), D, b, device=dev)
    dtc = orc.row_wise_compress(orc.transpose(D), b)
    interesting = [0, 1, 2, N - 1, N, N + 1, 2 * N, 1 << 16, 1 << 31, (1 << 32) - 1]
    for case in range(cases // 3):
        rows = int(rng.choice(interesting)) if rng.integers(0, 4) else int(rng.integers(0, 1 << 32))
        cols = int(rng.choice(interesting)) if rng.integers(0, 4) else int(rng.integers(0, 1 << 32))
        style = int(rng.integers(0, 5))
        if style == 0:
            rows, cols, nbytes = 1, N, 4 * N  # well-formed
        elif style == 1:
            nbytes = 4 * ((rows * cols) % (1 << 20)) if rows * cols < (1 << 20) else int(rng.integers(0, 4 * N + 16))  # length matches where it can
        elif style == 2:
            nbytes = int(rng.integers(0, 16))  # short
        else:
            nbytes = int(rng.integers(0, 4 * N + 64))
        wire = np.array([rows, cols], dtype="<u4").tobytes() + rng.bytes(nbytes)
        if style == 4:
            wire = wire[: int(rng.integers(0, 12))]  # cut inside the header
        try:
            want, want_err = orc.server_respond(dtc, N, b, wire), None
        except orc.OracleError as e:
            want, want_err = None, e.code
        try:
            got, got_err = srv.respond(wire), None
        except cp.ChalametPIRError as e:
            got, got_err = None, e.code
        kinds[want_err] = kinds.get(want_err, 0) + 1
        if got != want or got_err != want_err:
            bad += 1
            if bad <= 10:
                print(f"MISMATCH rows={rows} cols={cols} len={len(wire)}: product {got_err if got is None else 'bytes'} oracle {want_err if want is None else 'bytes'}", flush=True)
    srv.close()
print(f"wire fuzz: {cases // 3 * 3} cases, {bad} bad; oracle outcomes (status -> count, None = answered): {kinds}")
sys.exit(1 if bad else 0)
