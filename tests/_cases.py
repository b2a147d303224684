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


def synth_u32_at(indices, seed, mask=0xFFFFFFFF):
    """the counter-based synthetic generator (csrc/synth.hip, oracle or_synth_u64) at ARBITRARY indices, vectorised in numpy:
    hi32(splitmix64 finaliser of (seed, index)) & mask.  A third, independent statement of the same function, so that tests can
    rebuild single columns of a 30 GB matrix on the host."""
    with np.errstate(over="ignore"):
        idx = np.asarray(indices, dtype=np.uint64)
        z = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(32)).astype(np.uint32)) & np.uint32(mask)


class OwnMapping:
    """A u32 buffer in a private anonymous mapping of its own (mmap), page-aligned, for tests that hipHostRegister caller memory.
    NEVER register glibc heap memory (a numpy array) in this suite: on the GPU boxes a hipHostRegister / hipHostUnregister cycle over heap
    pages is followed, some allocations later, by "Memory access fault by GPU node-N on address <heap address>" -- with the HIP runtime and
    torch alone, no code of this repository involved (scripts/probes/register_then_copy_probe.py; profiles/HISTORY_design_r3.md 4.6).  Dedicated mappings, unmapped
    after unregistering, do not show it (scripts/probes/register_mmap_probe.py)."""

    def __init__(self, count):
        import mmap

        self.nbytes = (4 * count + 4095) // 4096 * 4096
        self._mm = mmap.mmap(-1, self.nbytes)
        self.array = np.frombuffer(self._mm, dtype=np.uint32, count=count)
        self.address = self.array.ctypes.data
        assert self.address % 4096 == 0

    def close(self):
        self.array = None
        try:
            self._mm.close()
        except BufferError:  # a view is still alive somewhere: leave the mapping to the garbage collector rather than fail the test
            pass
