"""ctypes binding of the CPU ORACLE (oracle/chalamet_oracle.c).

TEST INFRASTRUCTURE ONLY.  Import this from tests/, from __graft_entry__.smoke() and from bench.py's
`cpu_baseline` leg -- never from the product package `chalametpir_amd`.  The product path has no
CPU fallback and fails loudly if its HIP library is missing.

Every wrapper takes/returns numpy uint32 arrays and raises OracleError(code) on a non-zero status,
where `code` is the OR_ERR_* value mirroring the reference's ChalametPIRError variant.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libchalamet_oracle.so")

LWE_DIMENSION = 1774
SEED_BYTE_LEN = 32

# error codes (chalamet_oracle.h)
OK = 0
ERR_INVALID_MATRIX_DIMENSION = 1
ERR_INCOMPATIBLE_DIM_MATMUL = 2
ERR_INCOMPATIBLE_DIM_MATADD = 3
ERR_INVALID_NUMBER_OF_ELEMENTS = 4
ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED = 5
ERR_INVALID_DIMENSION_FOR_VECTOR = 6
ERR_FAILED_TO_DESERIALIZE_MATRIX = 7
ERR_EMPTY_KV_DATABASE = 8
ERR_EXHAUSTED_ATTEMPTS_3WISE = 9
ERR_EXHAUSTED_ATTEMPTS_4WISE = 10
ERR_ROW_NOT_DECODABLE = 11
ERR_DECODED_ROW_NOT_PREPENDED_WITH_DIGEST = 12
ERR_FAILED_TO_DESERIALIZE_FILTER = 13
ERR_KV_DATABASE_SIZE_TOO_LARGE = 14
ERR_INVALID_HINT_MATRIX = 15
ERR_ARITHMETIC_OVERFLOW_ADDING_QUERY_INDICATOR = 16
ERR_UNSUPPORTED_ARITY = 17
ERR_INVALID_RESPONSE_VECTOR = 18
ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH = 19
ERR_BUFFER_TOO_SMALL = 100


class OracleError(Exception):
    def __init__(self, code: int):
        super().__init__(f"oracle status {code}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds). Returns the .so path."""
    src = os.path.join(_HERE, "chalamet_oracle.c")
    hdr = os.path.join(_HERE, "chalamet_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


class _Bff(C.Structure):
    _fields_ = [
        ("seed", C.c_uint8 * 32),
        ("arity", C.c_uint32),
        ("segment_length", C.c_uint32),
        ("segment_count_length", C.c_uint32),
        ("num_fingerprints", C.c_uint64),
        ("filter_size", C.c_uint64),
        ("mat_elem_bit_len", C.c_uint64),
    ]


class _KvDb(C.Structure):
    _fields_ = [
        ("num_pairs", C.c_uint64),
        ("keys", C.c_void_p),
        ("key_off", C.c_void_p),
        ("values", C.c_void_p),
        ("val_off", C.c_void_p),
    ]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.or_synth_u64.restype = C.c_uint64
        _lib.or_synth_u64.argtypes = [C.c_uint64, C.c_uint64]
        _lib.or_murmur64.restype = C.c_uint64
        _lib.or_murmur64.argtypes = [C.c_uint64]
        _lib.or_mix.restype = C.c_uint64
        _lib.or_mix.argtypes = [C.c_uint64, C.c_uint64]
        _lib.or_mix256.restype = C.c_uint64
        _lib.or_bff_size_factor.restype = C.c_double
        _lib.or_bff_size_factor.argtypes = [C.c_uint32, C.c_uint32]
        _lib.or_bff_segment_length.restype = C.c_uint32
        _lib.or_bff_segment_length.argtypes = [C.c_uint32, C.c_uint32]
        _lib.or_encoded_num_cols.restype = C.c_uint64
        _lib.or_encoded_num_cols.argtypes = [C.c_uint64, C.c_uint]
        _lib.or_matrix_num_bytes.restype = C.c_size_t
        _lib.or_matrix_num_bytes.argtypes = [C.c_uint64, C.c_uint64]
        # OpenMP defaults to one thread per VISIBLE CPU; a GPU box shows 256 of them to a container with a 16-CPU quota, and
        # 256 spinning threads make every small oracle call cost 0.1 s: start from the CPUs this process may really use
        _lib.or_set_num_threads(C.c_int(usable_cpus()))
    return _lib


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc: int):
    if rc != OK:
        raise OracleError(rc)


def num_threads() -> int:
    return int(lib().or_num_threads())


def set_num_threads(n: int) -> None:
    lib().or_set_num_threads(C.c_int(n))


def usable_cpus() -> int:
    """CPUs this process may actually use: min(affinity mask, cgroup v2 CPU quota) -- a container with a 16-CPU quota on a
    256-thread host runs 128 OpenMP threads slower than 16"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


# ------------------------------------------------------------------ XOF
def turboshake128(msg: bytes, out_len: int, domain_sep: int = 0x1F) -> bytes:
    out = (C.c_uint8 * out_len)()
    m = (C.c_uint8 * len(msg)).from_buffer_copy(msg) if msg else None
    lib().or_turboshake128_hash(m, C.c_size_t(len(msg)), C.c_uint8(domain_sep), out, C.c_size_t(out_len))
    return bytes(out)


# ------------------------------------------------------------------ Matrix
def compression_factor(b: int) -> int:
    return int(lib().or_compression_factor(C.c_uint(b)))


def generate_from_seed(rows: int, cols: int, seed: bytes) -> np.ndarray:
    assert len(seed) == 32
    out = np.empty((rows, cols), dtype=np.uint32)
    s = (C.c_uint8 * 32).from_buffer_copy(seed)
    _chk(lib().or_generate_from_seed(C.c_uint64(rows), C.c_uint64(cols), s, _p(out)))
    return out


def mul(lhs: np.ndarray, rhs: np.ndarray) -> np.ndarray:
    lhs, rhs = _u32(lhs), _u32(rhs)
    out = np.empty((lhs.shape[0], rhs.shape[1]), dtype=np.uint32)
    _chk(lib().or_mul(_p(lhs), C.c_uint64(lhs.shape[0]), C.c_uint64(lhs.shape[1]), _p(rhs), C.c_uint64(rhs.shape[0]),
                      C.c_uint64(rhs.shape[1]), _p(out)))
    return out


def add(lhs: np.ndarray, rhs: np.ndarray) -> np.ndarray:
    lhs, rhs = _u32(lhs), _u32(rhs)
    out = np.empty(lhs.shape, dtype=np.uint32)
    _chk(lib().or_add(_p(lhs), C.c_uint64(lhs.shape[0]), C.c_uint64(lhs.shape[1]), _p(rhs), C.c_uint64(rhs.shape[0]),
                      C.c_uint64(rhs.shape[1]), _p(out)))
    return out


def transpose(m: np.ndarray) -> np.ndarray:
    m = _u32(m)
    out = np.empty((m.shape[1], m.shape[0]), dtype=np.uint32)
    _chk(lib().or_transpose(_p(m), C.c_uint64(m.shape[0]), C.c_uint64(m.shape[1]), _p(out)))
    return out


def identity(n: int) -> np.ndarray:
    out = np.empty((n, n), dtype=np.uint32)
    _chk(lib().or_identity(C.c_uint64(n), _p(out)))
    return out


def row_wise_compress(m: np.ndarray, b: int) -> np.ndarray:
    m = _u32(m)
    cf = compression_factor(b)
    if cf == 0:
        raise OracleError(ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH)
    out = np.empty((m.shape[0], -(-m.shape[1] // cf)), dtype=np.uint32)
    _chk(lib().or_row_wise_compress(_p(m), C.c_uint64(m.shape[0]), C.c_uint64(m.shape[1]), C.c_uint(b), _p(out)))
    return out


def row_wise_decompress(m: np.ndarray, b: int, num_cols: int) -> np.ndarray:
    m = _u32(m)
    out = np.empty((m.shape[0], num_cols), dtype=np.uint32)
    _chk(lib().or_row_wise_decompress(_p(m), C.c_uint64(m.shape[0]), C.c_uint64(m.shape[1]), C.c_uint(b),
                                      C.c_uint64(num_cols), _p(out)))
    return out


def row_vector_x_compressed_transposed_matrix(q: np.ndarray, rhs: np.ndarray, decompressed_num_cols: int, b: int) -> np.ndarray:
    q, rhs = _u32(q), _u32(rhs)
    if q.ndim == 1:
        q = q.reshape(1, -1)
    out = np.empty((1, rhs.shape[0]), dtype=np.uint32)
    _chk(lib().or_row_vector_x_compressed_transposed_matrix(
        _p(q), C.c_uint64(q.shape[0]), C.c_uint64(q.shape[1]), _p(rhs), C.c_uint64(rhs.shape[0]), C.c_uint64(rhs.shape[1]),
        C.c_uint64(decompressed_num_cols), C.c_uint(b), _p(out)))
    return out


def matrix_to_bytes(m: np.ndarray) -> bytes:
    m = _u32(m)
    n = 8 + m.size * 4
    out = (C.c_uint8 * n)()
    _chk(lib().or_matrix_to_bytes(_p(m), C.c_uint32(m.shape[0]), C.c_uint32(m.shape[1]), out, C.c_size_t(n)))
    return bytes(out)


def matrix_from_bytes(b: bytes) -> np.ndarray:
    rows, cols = C.c_uint32(), C.c_uint32()
    buf = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")
    _chk(lib().or_matrix_from_bytes(buf, C.c_size_t(len(b)), C.byref(rows), C.byref(cols)))
    return np.frombuffer(b, dtype="<u4", offset=8).reshape(rows.value, cols.value).copy()


# ------------------------------------------------------------------ Server
def find_encoded_db_matrix_element_bit_length(n: int) -> int:
    b = C.c_uint()
    _chk(lib().or_find_encoded_db_matrix_element_bit_length(C.c_uint64(n), C.byref(b)))
    return b.value


def server_respond(dtc: np.ndarray, decompressed_num_cols: int, b: int, query: bytes) -> bytes:
    dtc = _u32(dtc)
    cap = 8 + 4 * dtc.shape[0]
    out = (C.c_uint8 * cap)()
    out_len = C.c_size_t()
    qb = (C.c_uint8 * max(len(query), 1)).from_buffer_copy(query if query else b"\0")
    _chk(lib().or_server_respond(_p(dtc), C.c_uint64(dtc.shape[0]), C.c_uint64(dtc.shape[1]), C.c_uint64(decompressed_num_cols),
                                 C.c_uint(b), qb, C.c_size_t(len(query)), out, C.c_size_t(cap), C.byref(out_len)))
    return bytes(out[: out_len.value])


def server_setup_from_matrix(seed: bytes, D: np.ndarray, b: int):
    """-> (hint 1774xC, dtc CxW) following server.rs:59-67."""
    D = _u32(D)
    N, Cc = D.shape
    cf = compression_factor(b)
    if cf == 0:
        raise OracleError(ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH)
    hint = np.empty((LWE_DIMENSION, Cc), dtype=np.uint32)
    dtc = np.empty((Cc, -(-N // cf)), dtype=np.uint32)
    s = (C.c_uint8 * 32).from_buffer_copy(seed)
    _chk(lib().or_server_setup_from_matrix(s, _p(D), C.c_uint64(N), C.c_uint64(Cc), C.c_uint(b), _p(hint), _p(dtc)))
    return hint, dtc


# ------------------------------------------------------------------ BFF
@dataclass
class Filter:
    seed: bytes
    arity: int
    segment_length: int
    segment_count_length: int
    num_fingerprints: int
    filter_size: int
    mat_elem_bit_len: int

    def _c(self) -> _Bff:
        f = _Bff()
        C.memmove(f.seed, self.seed, 32)
        f.arity, f.segment_length, f.segment_count_length = self.arity, self.segment_length, self.segment_count_length
        f.num_fingerprints, f.filter_size, f.mat_elem_bit_len = self.num_fingerprints, self.filter_size, self.mat_elem_bit_len
        return f

    @staticmethod
    def _from_c(f: _Bff) -> "Filter":
        return Filter(bytes(f.seed), f.arity, f.segment_length, f.segment_count_length, f.num_fingerprints, f.filter_size,
                      f.mat_elem_bit_len)

    def to_bytes(self) -> bytes:
        out = (C.c_uint8 * 68)()
        c = self._c()
        lib().or_bff_to_bytes(C.byref(c), out)
        return bytes(out)

    @staticmethod
    def from_bytes(b: bytes) -> "Filter":
        f = _Bff()
        buf = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")
        _chk(lib().or_bff_from_bytes(buf, C.c_size_t(len(b)), C.byref(f)))
        return Filter._from_c(f)

    def bits_per_entry(self) -> float:
        return self.num_fingerprints * self.mat_elem_bit_len / self.filter_size


def bff_shape(arity: int, db_size: int):
    sl, scl, nf = C.c_uint32(), C.c_uint32(), C.c_uint64()
    _chk(lib().or_bff_shape(C.c_uint32(arity), C.c_uint64(db_size), C.byref(sl), C.byref(scl), C.byref(nf)))
    return sl.value, scl.value, nf.value


def encoded_num_cols(max_value_byte_len: int, b: int) -> int:
    return int(lib().or_encoded_num_cols(C.c_uint64(max_value_byte_len), C.c_uint(b)))


class _FlatDb:
    def __init__(self, keys, values):
        self.n = len(keys)
        self.kbuf = np.frombuffer(b"".join(keys) or b"\0", dtype=np.uint8).copy()
        self.vbuf = np.frombuffer(b"".join(values) or b"\0", dtype=np.uint8).copy()
        self.koff = np.zeros(self.n + 1, dtype=np.uint64)
        self.voff = np.zeros(self.n + 1, dtype=np.uint64)
        np.cumsum([len(k) for k in keys], out=self.koff[1:])
        np.cumsum([len(v) for v in values], out=self.voff[1:])
        self.c = _KvDb(self.n, self.kbuf.ctypes.data, self.koff.ctypes.data, self.vbuf.ctypes.data, self.voff.ctypes.data)


def from_kv_database(arity: int, keys, values, b: int, filter_seeds: bytes, max_attempts: int = 100):
    """matrix.rs:633-648. keys/values: lists of bytes (explicit order). filter_seeds: >= 32*max_attempts bytes.
    -> (D ndarray num_fingerprints x cols, Filter, attempts_used)"""
    if len(keys) == 0:
        raise OracleError(ERR_EMPTY_KV_DATABASE)
    assert len(filter_seeds) >= 32 * max_attempts
    db = _FlatDb(keys, values)
    _, _, nf = bff_shape(arity, len(keys))
    cols = encoded_num_cols(max(len(v) for v in values), b)
    mat = np.empty((nf, cols), dtype=np.uint32)
    f = _Bff()
    used = C.c_uint32()
    seeds = (C.c_uint8 * len(filter_seeds)).from_buffer_copy(filter_seeds)
    _chk(lib().or_from_kv_database(C.c_uint32(arity), C.byref(db.c), C.c_uint(b), seeds, C.c_uint32(max_attempts), C.byref(f),
                                   _p(mat), C.c_uint64(nf), C.c_uint64(cols), C.byref(used)))
    return mat, Filter._from_c(f), used.value


def recover_value(mat: np.ndarray, filt: Filter, key: bytes) -> bytes:
    mat = _u32(mat)
    cap = mat.shape[1] * 2 + 16
    out = (C.c_uint8 * cap)()
    n = C.c_size_t()
    k = (C.c_uint8 * len(key)).from_buffer_copy(key)
    c = filt._c()
    _chk(lib().or_recover_value(_p(mat), C.c_uint64(mat.shape[0]), C.c_uint64(mat.shape[1]), C.byref(c), k, C.c_size_t(len(key)),
                                out, C.c_size_t(cap), C.byref(n)))
    return bytes(out[: n.value])


# ------------------------------------------------------------------ row codec
def encode_kv_as_row(key: bytes, value: bytes, b: int, num_cols: int) -> np.ndarray:
    row = np.empty(num_cols, dtype=np.uint32)
    k = (C.c_uint8 * max(len(key), 1)).from_buffer_copy(key or b"\0")
    v = (C.c_uint8 * max(len(value), 1)).from_buffer_copy(value or b"\0")
    lib().or_encode_kv_as_row(k, C.c_size_t(len(key)), v, C.c_size_t(len(value)), C.c_uint(b), C.c_uint64(num_cols), _p(row))
    return row


def decode_kv_from_row(row: np.ndarray, b: int) -> bytes:
    row = _u32(row)
    cap = row.size * b // 8 + 8
    out = (C.c_uint8 * cap)()
    n = C.c_size_t()
    _chk(lib().or_decode_kv_from_row(_p(row), C.c_uint64(row.size), C.c_uint(b), out, C.c_size_t(cap), C.byref(n)))
    return bytes(out[: n.value])


# ------------------------------------------------------------------ client (e2e check)
def ternary_vector(n: int, rng: np.random.Generator) -> np.ndarray:
    """matrix.rs:572-619 with the draws taken from `rng` instead of an OS-seeded ChaCha8."""
    out = np.empty(n, dtype=np.uint32)
    i = 0
    t = C.c_uint32()
    while i < n:
        draws = rng.integers(0, 1 << 32, size=n - i, dtype=np.uint64).astype(np.uint32)
        for d in draws:
            if lib().or_ternary_from_u32(C.c_uint32(int(d)), C.byref(t)):
                out[i] = t.value
                i += 1
    return out


def ternary_vector_np(n: int, rng: np.random.Generator) -> np.ndarray:
    """ternary_vector for long vectors (the error vector e has N entries): the same rejection sampling and interval map as
    matrix.rs:577-612, stated in numpy; tests/test_oracle_properties.py pins it to or_ternary_from_u32 draw by draw"""
    interval = (0xFFFFFFFF - 2) // 3
    out = np.empty(n, dtype=np.uint32)
    i = 0
    while i < n:
        d = rng.integers(0, 1 << 32, size=n - i, dtype=np.uint64)
        d = d[d <= 3 * interval]
        t = np.where(d <= interval, 0, np.where(d <= 2 * interval, 1, 0xFFFFFFFF)).astype(np.uint32)
        out[i:i + t.size] = t
        i += t.size
    return out


def client_query(A: np.ndarray, hint: np.ndarray, filt: Filter, key: bytes, secret_s: np.ndarray, error_e: np.ndarray):
    A, hint, secret_s, error_e = _u32(A), _u32(hint), _u32(secret_s), _u32(error_e)
    N, Cc = A.shape[1], hint.shape[1]
    qb = np.empty(N, dtype=np.uint32)
    sc = np.empty(Cc, dtype=np.uint32)
    k = (C.c_uint8 * len(key)).from_buffer_copy(key)
    c = filt._c()
    _chk(lib().or_client_query(_p(A), _p(hint), C.c_uint64(N), C.c_uint64(Cc), C.byref(c), k, C.c_size_t(len(key)), _p(secret_s),
                               _p(error_e), _p(qb), _p(sc)))
    return qb, sc


def client_process_response(filt: Filter, key: bytes, secret_c: np.ndarray, response: np.ndarray) -> bytes:
    secret_c, response = _u32(secret_c).reshape(-1), _u32(response).reshape(-1)
    cap = response.size * 2 + 16
    out = (C.c_uint8 * cap)()
    n = C.c_size_t()
    k = (C.c_uint8 * len(key)).from_buffer_copy(key)
    c = filt._c()
    _chk(lib().or_client_process_response(C.byref(c), k, C.c_size_t(len(key)), _p(secret_c), _p(response), C.c_uint64(response.size),
                                          out, C.c_size_t(cap), C.byref(n)))
    return bytes(out[: n.value])


def filter_slots(filt: Filter, key: bytes):
    """the `arity` filter slots a key hashes to, as the client derives them (client.rs:109-113 / 158-163: hash_of_key -> mix256 with
    the filter's seed -> hash_batch_for_{3,4}_wise_xor_filter): the rows of D whose masked sum carries the key's value"""
    L = lib()
    hk = (C.c_uint64 * 4)()
    k = (C.c_uint8 * max(len(key), 1)).from_buffer_copy(key if key else b"\0")
    L.or_hash_of_key(k, C.c_size_t(len(key)), hk)
    seed = (C.c_uint8 * 32).from_buffer_copy(filt.seed)
    L.or_mix256.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)]
    h64 = L.or_mix256(hk, seed)
    h = (C.c_uint32 * 4)()
    fn = L.or_hash_batch_3 if filt.arity == 3 else L.or_hash_batch_4
    fn(C.c_uint64(h64), C.c_uint32(filt.segment_length), C.c_uint32(filt.segment_count_length), h)
    return [int(h[j]) for j in range(filt.arity)]


def query_indicator(b: int) -> int:
    """client.rs:277-282: 2^32 / 2^b"""
    return (1 << 32) >> b


def first_touch_copy(m: np.ndarray) -> np.ndarray:
    """copy of a row-major matrix whose pages are first touched by the OpenMP threads that will later stream those rows"""
    m = _u32(m)
    out = np.empty_like(m)
    lib().or_first_touch_copy(_p(out), _p(m), C.c_uint64(m.shape[0]), C.c_uint64(m.shape[1]))
    return out


# ------------------------------------------------------------------ synthetic inputs
def synth_fill_u32(count: int, seed: int, index0: int = 0, mask: int = 0xFFFFFFFF) -> np.ndarray:
    out = np.empty(count, dtype=np.uint32)
    lib().or_synth_fill_u32(_p(out), C.c_uint64(count), C.c_uint64(seed), C.c_uint64(index0), C.c_uint32(mask))
    return out
