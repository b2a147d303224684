"""Shared case generators for the parity tests (seeded; no OS randomness, unlike the reference's tests)."""
import numpy as np

ALL_BITS = list(range(4, 15))  # MIN..MAX_CIPHER_TEXT_BIT_LEN, params.rs:14-17


def cf_of(b):
    return 2 if b >= 11 else (3 if b >= 9 else 4)


def random_db_matrix(rng, N, C, b):
    return rng.integers(0, 1 << b, size=(N, C), dtype=np.uint64).astype(np.uint32)


def random_query(rng, N):
    return rng.integers(0, 1 << 32, size=N, dtype=np.uint64).astype(np.uint32)


def wire(mat):
    """Matrix::to_bytes (matrix.rs:947-971) written independently of both oracle and product"""
    mat = np.ascontiguousarray(mat, dtype="<u4")
    if mat.ndim == 1:
        mat = mat.reshape(1, -1)
    return np.array(mat.shape, dtype="<u4").tobytes() + mat.tobytes()


def unwire(b):
    rows, cols = np.frombuffer(b[:8], dtype="<u4")
    return np.frombuffer(b[8:], dtype="<u4").reshape(rows, cols)
