#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ from the CPU oracle.

The reference (Rust) holds no fixed vectors and cannot be run in the build image (no cargo/rustc), so these vectors are
produced by oracle/chalamet_oracle.c -- which tests/test_oracle_properties.py pins against the reference's property
tests, RFC 9861 and the README byte sizes -- and then frozen, so that (a) the oracle cannot drift silently and (b) the
HIP path is checked against data that does not depend on the oracle being importable.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

Everything is seeded; re-running reproduces the committed files byte for byte (np.savez_compressed aside from zip
timestamps, which is why the test compares array contents, not file hashes)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as orc  # noqa: E402
from _cases import cf_of, random_db_matrix, random_query, wire  # noqa: E402


def respond_cases():
    """every element bit length x every N mod cf, ragged row counts; small enough for pure inspection"""
    rng = np.random.default_rng(20260101)
    out = {}
    for b in range(4, 15):
        cf = cf_of(b)
        for tail in range(cf):
            N = cf * int(rng.integers(3, 40)) + tail
            C = int(rng.integers(1, 24))
            D = random_db_matrix(rng, N, C, b)
            q = random_query(rng, N)
            dtc = orc.row_wise_compress(orc.transpose(D), b)
            r = orc.row_vector_x_compressed_transposed_matrix(q, dtc, N, b)[0]
            k = f"b{b}_t{tail}"
            out[k + "_D"], out[k + "_q"], out[k + "_dtc"], out[k + "_r"] = D, q, dtc, r
    return out


def setup_cases():
    """seeded Server::setup from a matrix: hint = A(seed)*D and the packed DB, plus one wire-level respond"""
    rng = np.random.default_rng(20260102)
    out = {}
    for i, (b, N, C) in enumerate(((9, 301, 12), (10, 96, 7), (13, 65, 5), (6, 130, 9))):
        seed = rng.bytes(32)
        D = random_db_matrix(rng, N, C, b)
        hint, dtc = orc.server_setup_from_matrix(seed, D, b)
        q = random_query(rng, N)
        resp = orc.server_respond(dtc, N, b, wire(q))
        k = f"s{i}"
        out[k + "_seed"] = np.frombuffer(seed, dtype=np.uint8)
        out[k + "_b"] = np.array([b], dtype=np.uint32)
        out[k + "_D"], out[k + "_hint"], out[k + "_dtc"], out[k + "_q"] = D, hint, dtc, q
        out[k + "_resp"] = np.frombuffer(resp, dtype=np.uint8)
    return out


def kv_cases():
    """seeded Server::setup from a KV database (explicit key order + filter seeds): D, filter bytes"""
    rng = np.random.default_rng(20260103)
    out = {}
    for arity in (3, 4):
        n = 40
        keys = [rng.bytes(int(rng.integers(16, 33))) for _ in range(n)]
        vals = [rng.bytes(int(rng.integers(1, 20))) for _ in range(n)]
        fseeds = rng.bytes(3200)
        b = orc.find_encoded_db_matrix_element_bit_length(n)
        D, filt, used = orc.from_kv_database(arity, keys, vals, b, fseeds)
        k = f"kv{arity}"
        out[k + "_keys"] = np.frombuffer(b"".join(keys), dtype=np.uint8)
        out[k + "_klen"] = np.array([len(x) for x in keys], dtype=np.uint32)
        out[k + "_vals"] = np.frombuffer(b"".join(vals), dtype=np.uint8)
        out[k + "_vlen"] = np.array([len(x) for x in vals], dtype=np.uint32)
        out[k + "_fseeds"] = np.frombuffer(fseeds, dtype=np.uint8)
        out[k + "_b"] = np.array([b], dtype=np.uint32)
        out[k + "_D"] = D
        out[k + "_filter"] = np.frombuffer(filt.to_bytes(), dtype=np.uint8)
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "respond_cases.npz"), **respond_cases())
    np.savez_compressed(os.path.join(HERE, "setup_cases.npz"), **setup_cases())
    np.savez_compressed(os.path.join(HERE, "kv_cases.npz"), **kv_cases())
    for f in ("respond_cases.npz", "setup_cases.npz", "kv_cases.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
