"""Host-side mirror of `chalametpir_server::Server` (reference chalametpir_server/src/server.rs:15-219) over the C ABI.

    server, hint_bytes, filter_param_bytes = Server.setup(seed_mu, db, arity=3)     # server.rs:47 / 103
    response_bytes = server.respond(query_bytes)                                      # server.rs:184

Same names, argument meaning and error behaviour as the reference (errors are raised as ChalametPIRError carrying the
reference's variant name).  Everything that computes goes through libchalamet_hip.so; there is no CPU path here.
PyTorch is optional plumbing: tensors are only used to hand device memory / streams to the *_device entry points.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional, Sequence, Tuple

import numpy as np

from . import _native
from ._native import DtcLayout, KvDb
from .errors import ChalametPIRError
from .params import LWE_DIMENSION, SEED_BYTE_LEN, SERVER_SETUP_MAX_ATTEMPT_COUNT


def _check(status: int) -> None:
    if status != 0:
        lib = _native.load()
        msg = lib.cpir_strerror(status).decode()
        detail = lib.cpir_last_hip_error().decode() if status in (64, 65, 66) else ""
        raise ChalametPIRError(status, msg, detail)


def _seed_arg(seed: bytes):
    if len(seed) != SEED_BYTE_LEN:
        raise ValueError(f"seed must be {SEED_BYTE_LEN} bytes")  # &[u8; SEED_BYTE_LEN] in the reference: a type error there
    return (C.c_uint8 * SEED_BYTE_LEN).from_buffer_copy(seed)


def _u32_host(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _tensor_ptr(t) -> int:
    """device address of a torch tensor holding 32-bit elements (int32 / uint32 views are both fine)"""
    if t.element_size() != 4 or not t.is_contiguous():
        raise ValueError("expected a contiguous tensor of 4-byte elements")
    return t.data_ptr()


def _device_array(devices: Sequence["Device"]):
    """cpir_device* const* for the *_multi constructors"""
    arr = (C.c_void_p * len(devices))(*[d._h for d in devices])
    return arr


def _stream_ptr(stream) -> Optional[int]:
    if stream is None:
        return None
    return int(getattr(stream, "cuda_stream", stream))


class Device:
    """cpir_device: replaces gpu_utils::setup_gpu() (reference gpu_utils.rs:25-79)."""

    def __init__(self, ordinal: int = 0):
        self._lib = _native.load()
        h = C.c_void_p()
        _check(self._lib.cpir_device_open(ordinal, C.byref(h)))
        self._h = h
        self.ordinal = ordinal

    @staticmethod
    def count() -> int:
        n = C.c_int()
        st = _native.load().cpir_device_count(C.byref(n))
        return n.value if st == 0 else 0

    def synchronize(self) -> None:
        _check(self._lib.cpir_device_synchronize(self._h))

    def close(self) -> None:
        if self._h:
            self._lib.cpir_device_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- low-level device ops on caller-owned device memory (torch tensors or raw addresses) -----------------------
    def mat_x_mat(self, A, D, M, rows: int, inner: int, cols: int, *, lda=None, ldd=None, ldm=None, rhs_max_bits: int = 32,
                  accumulate: bool = False, stream=None) -> None:
        """gpu_utils::mat_x_mat (reference gpu_utils.rs:156-220)"""
        _check(self._lib.cpir_op_mat_x_mat(self._h, _tensor_ptr(A), lda or inner, _tensor_ptr(D), ldd or cols, _tensor_ptr(M),
                                           ldm or cols, rows, inner, cols, rhs_max_bits, int(accumulate), _stream_ptr(stream)))

    def transpose_compress(self, D, layout: DtcLayout, dtc, *, ldd=None, or_of_entries=None, stream=None) -> None:
        """gpu_utils::mat_transpose + Matrix::row_wise_compress (reference gpu_utils.rs:222-281, matrix.rs:98-205)"""
        _check(self._lib.cpir_op_transpose_compress(self._h, _tensor_ptr(D), ldd or layout.num_cols, C.byref(layout), _tensor_ptr(dtc),
                                                    _tensor_ptr(or_of_entries) if or_of_entries is not None else None,
                                                    _stream_ptr(stream)))

    def transpose_compress_with_plane(self, D, layout: DtcLayout, dtc, hi_plane, *, ldd=None, or_of_entries=None, stream=None) -> None:
        """transpose_compress that also writes the second operand plane of the hint matmul (planar packing, b >= 10; with b = 9 there is
        no plane -- hi_plane None -- and this is plain transpose_compress): D is read once for the packed image and for hint = A * D"""
        _check(self._lib.cpir_op_transpose_compress_with_plane(self._h, _tensor_ptr(D), ldd or layout.num_cols, C.byref(layout), _tensor_ptr(dtc),
                                                               _tensor_ptr(or_of_entries) if or_of_entries is not None else None,
                                                               _tensor_ptr(hi_plane) if hi_plane is not None else None, _stream_ptr(stream)))

    def mat_x_packed(self, A, dtc, layout: DtcLayout, hi_plane, M, rows: int, *, lda=None, ldm=None, accumulate: bool = False, stream=None) -> None:
        """M (+)= A * D with D given as its packed image + the plane of transpose_compress_with_plane (entries of D below 2^b)"""
        _check(self._lib.cpir_op_mat_x_packed(self._h, _tensor_ptr(A), lda or layout.num_slots, _tensor_ptr(dtc), C.byref(layout),
                                              _tensor_ptr(hi_plane) if hi_plane is not None else None, _tensor_ptr(M), ldm or layout.num_cols, rows,
                                              int(accumulate),
                                              _stream_ptr(stream)))

    def respond(self, dtc, layout: DtcLayout, q, r, *, q_len=None, q_slot_offset: int = 0, stream=None) -> None:
        """Matrix::row_vector_x_compressed_transposed_matrix (reference matrix.rs:328-485) on device tensors"""
        _check(self._lib.cpir_op_respond(self._h, _tensor_ptr(dtc), C.byref(layout), _tensor_ptr(q),
                                         q_len if q_len is not None else q.numel(), q_slot_offset, _tensor_ptr(r), None,
                                         _stream_ptr(stream)))

    def hint_partial(self, seed_mu: bytes, D_shard, slot_offset: int, num_slots: int, total_slots: int, num_cols: int, M_out, *,
                     ldd: Optional[int] = None, rhs_max_bits: int = 16, stream=None) -> None:
        """this shard's share of hint = A(seed) * D (reference server.rs:59-61): M_out (1774 x C device tensor) is overwritten with
        A[:, slot_offset:slot_offset+num_slots] * D_shard; sum the partials over shards (distributed.reduce_u32_)"""
        _check(self._lib.cpir_hint_partial_device(self._h, _seed_arg(seed_mu), None, _tensor_ptr(D_shard), ldd or num_cols, slot_offset,
                                                  num_slots, total_slots, num_cols, rhs_max_bits, _tensor_ptr(M_out), _stream_ptr(stream)))

    def synth_fill(self, out, count: int, seed: int, index0: int = 0, mask: int = 0xFFFFFFFF, *, offset_words: int = 0, stream=None) -> None:
        _check(self._lib.cpir_op_synth_fill(self._h, _tensor_ptr(out) + 4 * offset_words, count, seed, index0, mask, _stream_ptr(stream)))


def dtc_layout_for(num_slots: int, num_cols: int, mat_elem_bit_len: int, packing: Optional[int] = None) -> DtcLayout:
    """cpir_dtc_layout of the device-resident packed DB; packing None = the library default (dense64 where offered),
    0 = reference packing, 1 = dense64"""
    L = DtcLayout()
    if packing is None:
        _check(_native.load().cpir_dtc_layout_for(num_slots, num_cols, mat_elem_bit_len, C.byref(L)))
    else:
        _check(_native.load().cpir_dtc_layout_for_packing(num_slots, num_cols, mat_elem_bit_len, packing, C.byref(L)))
    return L


def packed_rhs_offered(layout: DtcLayout) -> bool:
    """whether Device.mat_x_packed takes an image of this layout as its right-hand side (planar packing, b >= 9)"""
    return bool(_native.load().cpir_packed_rhs_offered(C.byref(layout)))


def packed_rhs_plane_bytes(layout: DtcLayout) -> int:
    """bytes of the second operand plane Device.transpose_compress_with_plane writes: 0 with b = 9 (the matmul expands the image's one
    bit plane itself) and where the pairing is not offered at all (packed_rhs_offered)"""
    return int(_native.load().cpir_packed_rhs_plane_bytes(C.byref(layout)))


def find_encoded_db_matrix_element_bit_length(db_entry_count: int) -> int:
    """Server::find_encoded_db_matrix_element_bit_length (reference server.rs:193-218)"""
    b = C.c_uint32()
    _check(_native.load().cpir_find_encoded_db_matrix_element_bit_length(db_entry_count, C.byref(b)))
    return b.value


def filter_shape(arity: int, db_entry_count: int) -> Tuple[int, int, int]:
    """(segment_length, segment_count_length, num_fingerprints) (reference binary_fuse_filter.rs:52-67, 519-538)"""
    sl, scl, nf = C.c_uint32(), C.c_uint32(), C.c_uint64()
    _check(_native.load().cpir_filter_shape(arity, db_entry_count, C.byref(sl), C.byref(scl), C.byref(nf)))
    return sl.value, scl.value, nf.value


def encoded_num_cols(max_value_byte_len: int, mat_elem_bit_len: int) -> int:
    return int(_native.load().cpir_encoded_num_cols(max_value_byte_len, mat_elem_bit_len))


def generate_from_seed(rows: int, cols: int, seed: bytes) -> np.ndarray:
    """Matrix::generate_from_seed (reference matrix.rs:541-558); host-side XOF of the product library"""
    out = np.empty((rows, cols), dtype=np.uint32)
    _check(_native.load().cpir_generate_from_seed(rows, cols, _seed_arg(seed), _ptr(out)))
    return out


class SeedExpander:
    """The XOF stream behind Matrix::generate_from_seed (reference matrix.rs:541-558), squeezed piecemeal (cpir_xof_*): `squeeze_into(a)`
    fills the u32 array `a` with the next a.size words of the row-major matrix."""

    def __init__(self, seed: bytes):
        self._lib = _native.load()
        h = C.c_void_p()
        _check(self._lib.cpir_xof_open(_seed_arg(seed), C.byref(h)))
        self._h = h

    def squeeze_into(self, out: np.ndarray) -> None:
        if out.dtype != np.uint32 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("expected a contiguous uint32 array")
        _check(self._lib.cpir_xof_squeeze(self._h, _ptr(out), out.nbytes))

    def close(self) -> None:
        if self._h:
            self._lib.cpir_xof_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode_kv_database(db: Mapping[bytes, bytes], arity: int, mat_elem_bit_len: int, filter_seed_material: Optional[bytes] = None,
                       max_attempts: int = SERVER_SETUP_MAX_ATTEMPT_COUNT) -> Tuple[np.ndarray, bytes]:
    """Matrix::from_kv_database::<ARITY> (reference matrix.rs:633-648) on the host -> (D as N x C u32, filter_param_bytes)"""
    lib = _native.load()
    if len(db) == 0:
        raise ChalametPIRError(8, lib.cpir_strerror(8).decode())
    flat = _FlatKvDb(db)
    _, _, nf = filter_shape(arity, len(db))
    cols = encoded_num_cols(max(len(v) for v in db.values()), mat_elem_bit_len)
    D = np.empty((nf, cols), dtype=np.uint32)
    fbytes = (C.c_uint8 * _native.FILTER_PARAM_BYTE_LEN)()
    seeds = None
    if filter_seed_material is not None:
        if len(filter_seed_material) < 32 * max_attempts:
            raise ValueError("filter_seed_material must hold 32 bytes per attempt")
        seeds = (C.c_uint8 * len(filter_seed_material)).from_buffer_copy(filter_seed_material)
    N, Cc = C.c_uint64(), C.c_uint32()
    _check(lib.cpir_encode_kv_database(arity, C.byref(flat.c), mat_elem_bit_len, seeds, max_attempts, fbytes, _ptr(D), D.size, C.byref(N),
                                       C.byref(Cc)))
    assert (N.value, Cc.value) == D.shape
    return D, bytes(fbytes)


class PinnedArray:
    """A u32 numpy array in page-locked host memory (cpir_host_alloc): queries kept in one are uploaded by DMA straight from it.
    `.array` is the numpy view; the memory lives as long as this object."""

    def __init__(self, count: int):
        self._lib = _native.load()
        p = C.c_void_p()
        _check(self._lib.cpir_host_alloc(4 * count, C.byref(p)))
        self._p = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(count,))

    def close(self) -> None:
        if self._p:
            self.array = None
            self._lib.cpir_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def respond_batch_pass_width(layout: DtcLayout, batch: int) -> int:
    """queries per pass a fused batch of `batch` queries is cut into on this layout under the current tuning"""
    return int(_native.load().cpir_respond_batch_pass_width(C.byref(layout), int(batch)))


def pack_kernel_name(layout: DtcLayout) -> str:
    """the kernel Device.transpose_compress runs for this layout, as a kernel trace names it"""
    return _native.load().cpir_pack_kernel_name(C.byref(layout)).decode()


def mat_x_mat_kernel_name(rhs_max_bits: int = 16) -> str:
    return _native.load().cpir_mat_x_mat_kernel_name(rhs_max_bits).decode()


def tuning_set(key: str, value: int) -> None:
    _check(_native.load().cpir_tuning_set(key.encode(), int(value)))


def host_gather(src: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """src[idx] through the library's host gather (the routine that compacts a lone host query; cpir_host_gather_words)"""
    src, idx = np.ascontiguousarray(src, dtype=np.uint32), np.ascontiguousarray(idx, dtype=np.uint32)
    out = np.empty(idx.size, dtype=np.uint32)
    _check(_native.load().cpir_host_gather_words(_ptr(out), _ptr(src), _ptr(idx), idx.size))
    return out


def host_compress(src: np.ndarray, keep_mask: np.ndarray, s_lo: int = 0, s_hi: Optional[int] = None, *, dst_misalign_words: int = 0) -> np.ndarray:
    """src[s] for s in [s_lo, s_hi) with keep_mask[s] set, through the library's streaming compaction (cpir_host_compress_words).  The
    destination starts `dst_misalign_words` words past a 64-byte boundary: 0 takes the non-temporal form (what the arenas' staging blocks
    get), anything else the plain one.  The words around the output are checked for not having been touched."""
    src = np.ascontiguousarray(src, dtype=np.uint32)
    s_hi = src.size if s_hi is None else s_hi
    bits = np.concatenate([np.packbits(np.asarray(keep_mask, dtype=bool), bitorder="little"), np.zeros(8, dtype=np.uint8)])
    cap = max(1, s_hi - s_lo)
    raw = np.full(cap + 64, 0xA5A5A5A5, dtype=np.uint32)
    off = (-(raw.ctypes.data // 4)) % 16 + 16 + dst_misalign_words  # a 64-byte boundary of the buffer, a guard of 16 words in front
    n = C.c_uint64()
    _check(_native.load().cpir_host_compress_words(raw.ctypes.data + 4 * off, _ptr(src), bits.ctypes.data, s_lo, s_hi, C.byref(n)))
    if not (np.all(raw[:off] == 0xA5A5A5A5) and np.all(raw[off + n.value:] == 0xA5A5A5A5)):
        raise AssertionError("host compaction wrote outside its output range")
    return raw[off: off + n.value].copy()


def host_gather_variant() -> str:
    return _native.load().cpir_host_gather_variant().decode()


def tuning_reset() -> None:
    """every tuning key back to its default (cpir_tuning_reset)"""
    _native.load().cpir_tuning_reset()


class _FlatKvDb:
    """HashMap<&[u8], &[u8]> flattened into the cpir_kv_db arrays (iteration order of the mapping = key order)."""

    def __init__(self, db: Optional[Mapping[bytes, bytes]] = None, *, arrays=None):
        if arrays is not None:  # (keys u8, key_off u64[n+1], values u8, val_off u64[n+1]) already flat
            kbuf, koff, vbuf, voff = arrays
            self.kbuf = np.ascontiguousarray(kbuf, dtype=np.uint8)
            self.vbuf = np.ascontiguousarray(vbuf, dtype=np.uint8)
            self.koff = np.ascontiguousarray(koff, dtype=np.uint64)
            self.voff = np.ascontiguousarray(voff, dtype=np.uint64)
            self.n = len(self.koff) - 1
            if len(self.voff) != self.n + 1 or int(self.koff[-1]) > self.kbuf.size or int(self.voff[-1]) > self.vbuf.size:
                raise ValueError("inconsistent flat key-value arrays")
        else:
            keys = list(db.keys())
            vals = [db[k] for k in keys]
            self.n = len(keys)
            self.kbuf = np.frombuffer(b"".join(keys) or b"\0", dtype=np.uint8).copy()
            self.vbuf = np.frombuffer(b"".join(vals) or b"\0", dtype=np.uint8).copy()
            self.koff = np.zeros(self.n + 1, dtype=np.uint64)
            self.voff = np.zeros(self.n + 1, dtype=np.uint64)
            if self.n:
                np.cumsum([len(k) for k in keys], out=self.koff[1:])
                np.cumsum([len(v) for v in vals], out=self.voff[1:])
        self.c = KvDb(self.n, _ptr(self.kbuf), _ptr(self.koff), _ptr(self.vbuf), _ptr(self.voff))


class Server:
    """Device-resident replacement of `struct Server` (reference server.rs:15-21): the packed, transposed DB lives in HBM."""

    def __init__(self, handle: C.c_void_p, device: Device):
        self._lib = _native.load()
        self._h = handle
        self.device = device
        L = DtcLayout()
        _check(self._lib.cpir_server_layout(self._h, C.byref(L)))
        self.layout = L  # the LOGICAL database (num_slots = the query slots this server answers for)
        P = DtcLayout()
        _check(self._lib.cpir_server_physical_layout(self._h, C.byref(P)))
        self.physical_layout = P  # the resident image: fewer slots when only the rows with a non-zero field are served (slots_served)
        off, tot = C.c_uint64(), C.c_uint64()
        _check(self._lib.cpir_server_shard(self._h, C.byref(off), C.byref(tot)))
        self.slot_offset, self.total_slots = off.value, tot.value

    # ---- construction -------------------------------------------------------------------------------------------------
    @staticmethod
    def setup(seed_mu: bytes, db: Mapping[bytes, bytes], arity: int = 3, *, device: Optional[Device] = None,
              devices: Optional[Sequence[Device]] = None, filter_seed_material: Optional[bytes] = None,
              max_attempts: int = SERVER_SETUP_MAX_ATTEMPT_COUNT) -> Tuple["Server", bytes, bytes]:
        """Server::setup::<ARITY>(seed_mu, db) -> (Server, hint_bytes, filter_param_bytes)   (reference server.rs:47-78 / 103-167)
        `devices`: split the database over several devices of this process (group handle, cpir_server_setup_kv_multi)"""
        lib = _native.load()
        if arity not in (3, 4):
            raise ChalametPIRError(17, lib.cpir_strerror(17).decode())  # const { assert!(ARITY == 3 || ARITY == 4) }, matrix.rs:638
        if len(db) == 0:
            raise ChalametPIRError(8, lib.cpir_strerror(8).decode())  # EmptyKVDatabase, server.rs:48-51
        return Server._setup_flat(seed_mu, _FlatKvDb(db), arity, device, filter_seed_material, max_attempts, devices)

    @staticmethod
    def setup_flat(seed_mu: bytes, keys, key_off, values, val_off, arity: int = 3, *, device: Optional[Device] = None,
                   devices: Optional[Sequence[Device]] = None, filter_seed_material: Optional[bytes] = None,
                   max_attempts: int = SERVER_SETUP_MAX_ATTEMPT_COUNT) -> Tuple["Server", bytes, bytes]:
        """Server::setup on a database already flattened into the cpir_kv_db arrays (what the Rust shim hands over):
        keys / values are u8 buffers, key_off / val_off hold num_pairs + 1 offsets.  Keys must be distinct.
        `devices`: the group handle (cpir_server_setup_kv_multi: what rust/server_hip.rs calls when CHALAMET_HIP_DEVICES names several)."""
        lib = _native.load()
        if arity not in (3, 4):
            raise ChalametPIRError(17, lib.cpir_strerror(17).decode())
        flat = _FlatKvDb(arrays=(keys, key_off, values, val_off))
        if flat.n == 0:
            raise ChalametPIRError(8, lib.cpir_strerror(8).decode())
        return Server._setup_flat(seed_mu, flat, arity, device, filter_seed_material, max_attempts, devices)

    @staticmethod
    def _setup_flat(seed_mu, flat, arity, device, filter_seed_material, max_attempts, devices=None):
        lib = _native.load()
        b, N, Cc, need = C.c_uint32(), C.c_uint64(), C.c_uint32(), C.c_size_t()
        _check(lib.cpir_setup_kv_shape(arity, C.byref(flat.c), C.byref(b), C.byref(N), C.byref(Cc), C.byref(need)))
        device = device or Device(0)
        hint = np.empty(need.value // 4, dtype=np.uint32)  # 4-byte aligned wire image
        hint_len = C.c_size_t()
        fbytes = (C.c_uint8 * _native.FILTER_PARAM_BYTE_LEN)()
        seeds = None
        if filter_seed_material is not None:
            if len(filter_seed_material) < 32 * max_attempts:
                raise ValueError("filter_seed_material must hold 32 bytes per attempt")
            seeds = (C.c_uint8 * len(filter_seed_material)).from_buffer_copy(filter_seed_material)
        h = C.c_void_p()
        if devices:
            _check(lib.cpir_server_setup_kv_multi(_device_array(devices), len(devices), arity, _seed_arg(seed_mu), C.byref(flat.c), seeds,
                                                  max_attempts, _ptr(hint), need.value, C.byref(hint_len), fbytes, C.byref(h)))
            return Server(h, devices[0]), hint.tobytes()[: hint_len.value], bytes(fbytes)
        _check(lib.cpir_server_setup_kv(device._h, arity, _seed_arg(seed_mu), C.byref(flat.c), seeds, max_attempts, _ptr(hint), need.value,
                                        C.byref(hint_len), fbytes, C.byref(h)))
        return Server(h, device), hint.tobytes()[: hint_len.value], bytes(fbytes)

    @staticmethod
    def setup_from_matrix(seed_mu: bytes, D: np.ndarray, mat_elem_bit_len: int, *, pub_mat_a: Optional[np.ndarray] = None,
                          device: Optional[Device] = None, devices: Optional[Sequence[Device]] = None) -> Tuple["Server", np.ndarray]:
        """The matrix half of Server::setup (reference server.rs:59-67) from an already encoded DB matrix D (N x C).
        Returns (Server, hint) with hint the 1774 x C matrix whose to_bytes image is `hint_bytes`.
        `devices`: split the database over several devices of this process (group handle, cpir_server_setup_multi)."""
        lib = _native.load()
        D = _u32_host(D)
        if D.ndim != 2:
            raise ValueError("D must be N x C")
        N, Cc = D.shape
        device = device or Device(0)
        hint = np.empty((LWE_DIMENSION, Cc), dtype=np.uint32)
        a_ptr = None
        if pub_mat_a is not None:
            pub_mat_a = _u32_host(pub_mat_a)
            if pub_mat_a.shape != (LWE_DIMENSION, N):
                raise ChalametPIRError(2, lib.cpir_strerror(2).decode())  # IncompatibleDimensionForMatrixMultiplication
            a_ptr = _ptr(pub_mat_a)
        h = C.c_void_p()
        if devices:
            _check(lib.cpir_server_setup_multi(_device_array(devices), len(devices), _seed_arg(seed_mu), a_ptr, _ptr(D), N, Cc,
                                               mat_elem_bit_len, _ptr(hint), C.byref(h)))
            return Server(h, devices[0]), hint
        _check(lib.cpir_server_setup(device._h, _seed_arg(seed_mu), a_ptr, _ptr(D), N, Cc, mat_elem_bit_len, _ptr(hint), C.byref(h)))
        return Server(h, device), hint

    @staticmethod
    def from_compressed(compressed: np.ndarray, decompressed_num_cols: int, mat_elem_bit_len: int, *,
                        device: Optional[Device] = None) -> "Server":
        """From the reference's own Server fields (server.rs:16-21): compressed transposed matrix C x ceil(N/cf), N, b."""
        lib = _native.load()
        compressed = _u32_host(compressed)
        device = device or Device(0)
        cf = lib.cpir_compression_factor(mat_elem_bit_len)
        if cf == 0:
            raise ChalametPIRError(19, lib.cpir_strerror(19).decode())
        if compressed.ndim != 2 or compressed.shape[1] != -(-decompressed_num_cols // cf):
            raise ChalametPIRError(4, lib.cpir_strerror(4).decode())
        h = C.c_void_p()
        _check(lib.cpir_server_from_compressed(device._h, _ptr(compressed), compressed.shape[0], decompressed_num_cols, mat_elem_bit_len,
                                               C.byref(h)))
        return Server(h, device)

    @staticmethod
    def from_device_matrix(D_dev, num_slots: int, num_cols: int, mat_elem_bit_len: int, *, device: Device, ldd: Optional[int] = None,
                           slot_offset: int = 0, total_slots: Optional[int] = None, stream=None) -> "Server":
        """Pack a (shard of the) encoded DB that already sits in HBM (torch tensor, num_slots x num_cols)."""
        lib = _native.load()
        h = C.c_void_p()
        _check(lib.cpir_server_from_device_matrix(device._h, _tensor_ptr(D_dev), ldd or num_cols, num_slots, num_cols, mat_elem_bit_len,
                                                  slot_offset, total_slots if total_slots is not None else num_slots,
                                                  _stream_ptr(stream), C.byref(h)))
        return Server(h, device)

    def clone(self) -> "Server":
        """#[derive(Clone)] (reference server.rs:15): shares the immutable device-resident DB."""
        return Server(C.c_void_p(self._lib.cpir_server_retain(self._h)), self.device)

    def close(self) -> None:
        if self._h:
            self._lib.cpir_server_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- properties ---------------------------------------------------------------------------------------------------
    @property
    def decompressed_num_cols(self) -> int:  # server.rs:19
        return int(self.layout.num_slots)

    @property
    def mat_elem_bit_len(self) -> int:  # server.rs:20
        return int(self.layout.mat_elem_bit_len)

    @property
    def response_len(self) -> int:
        return int(self.layout.num_cols)

    def slots_served(self) -> Tuple[int, int]:
        """(slots resident in the image, slots of the logical database): they differ when rows of D without a non-zero field were left
        out (a real encoded database: the slots no key owns, reference matrix.rs:702-746); a group handle sums over its shards"""
        a, b = C.c_uint64(), C.c_uint64()
        _check(self._lib.cpir_server_slots_served(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kept_slots(self) -> Optional[np.ndarray]:
        """the slots the image holds (increasing, relative to this shard's first slot), or None when every slot is served"""
        served, of = self.slots_served()
        if served == of or self.group_shards():
            return None
        out = np.empty(served, dtype=np.uint32)
        _check(self._lib.cpir_server_kept_slots(self._h, _ptr(out), out.size))
        return out

    def group_shards(self):
        """[(device ordinal, first slot, slots)] of a group handle (Server.setup(..., devices=[...])); [] for an ordinary server"""
        n = C.c_uint32()
        _check(self._lib.cpir_server_group_size(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            dev, lo, cnt = C.c_int(), C.c_uint64(), C.c_uint64()
            _check(self._lib.cpir_server_group_shard(self._h, i, C.byref(dev), C.byref(lo), C.byref(cnt)))
            out.append((dev.value, lo.value, cnt.value))
        return out

    SETUP_PHASES = ("encode", "xof_expand_A", "D_h2d", "transpose_compress", "wait_for_A", "hint_matmul", "hint_d2h", "total")

    def setup_timings(self) -> dict:
        """wall-clock split (seconds) of the setup call that built this server"""
        out = (C.c_double * 8)()
        _check(self._lib.cpir_server_setup_timings(self._h, out))
        return dict(zip(self.SETUP_PHASES, [float(x) for x in out]))

    # (cpir_server_host_path_counts: "alone_polled" is the subset of "alone" answered by a launch that polled the copy of a pageable query;
    # "polled_passes_given_up" counts void passes of lone launches AND of in-place rounds; calls == alone + in_uploaded_rounds + in_in_place_rounds)
    HOST_PATH_COUNTS = ("calls", "alone", "alone_polled", "polled_passes_given_up", "in_uploaded_rounds", "uploaded_rounds",
                        "in_in_place_rounds", "in_place_rounds")

    def host_path_counts(self) -> dict:
        """how the host callers of respond() have been served by this handle so far (cpir_server_host_path_counts)"""
        out = (C.c_uint64 * 8)()
        _check(self._lib.cpir_server_host_path_counts(self._h, out))
        return dict(zip(self.HOST_PATH_COUNTS, [int(x) for x in out]))

    def export_compressed(self) -> np.ndarray:
        """compressed_transposed_parsed_db_mat_d in the reference's layout (C x ceil(N/cf)) (server.rs:18)"""
        out = np.empty((self.layout.num_cols, self.layout.words_per_row), dtype=np.uint32)
        _check(self._lib.cpir_server_export_compressed(self._h, _ptr(out), out.size))
        return out

    # ---- respond ------------------------------------------------------------------------------------------------------
    def respond(self, query: bytes) -> bytes:
        """Server::respond(&self, query: &[u8]) -> Result<Vec<u8>, ChalametPIRError>   (reference server.rs:184-190)"""
        cap = 8 + 4 * self.layout.num_cols
        out = (C.c_uint8 * cap)()
        n = C.c_size_t()
        qbuf = (C.c_uint8 * max(len(query), 1)).from_buffer_copy(query if len(query) else b"\0")
        _check(self._lib.cpir_server_respond_bytes(self._h, C.addressof(qbuf), len(query), C.addressof(out), cap, C.byref(n)))
        return bytes(out[: n.value])

    def respond_from_address(self, address: int, length: int) -> bytes:
        """Server::respond on `length` wire bytes at a raw host address (no copy on this side; any alignment)"""
        cap = 8 + 4 * self.layout.num_cols
        out = (C.c_uint8 * cap)()
        n = C.c_size_t()
        _check(self._lib.cpir_server_respond_bytes(self._h, C.c_void_p(address), length, C.addressof(out), cap, C.byref(n)))
        return bytes(out[: n.value])

    def respond_array(self, q: np.ndarray) -> np.ndarray:
        """respond on the element array of a 1 x N query (query[8..] of the wire image)"""
        q = _u32_host(q)
        rows, cols = (1, q.shape[0]) if q.ndim == 1 else q.shape
        r = np.empty(self.layout.num_cols, dtype=np.uint32)
        _check(self._lib.cpir_server_respond(self._h, _ptr(q), rows, cols, _ptr(r)))
        return r

    def respond_device(self, q_dev, r_dev, stream=None) -> None:
        """enqueue respond on device tensors: q_dev total_slots x u32, r_dev C x u32 (partial response for a shard)"""
        _check(self._lib.cpir_server_respond_device(self._h, _tensor_ptr(q_dev), _tensor_ptr(r_dev), None, _stream_ptr(stream)))

    def respond_batch_device(self, q_dev, batch: int, r_dev, stream=None) -> None:
        _check(self._lib.cpir_server_respond_batch_device(self._h, _tensor_ptr(q_dev), batch, _tensor_ptr(r_dev), None, _stream_ptr(stream)))
