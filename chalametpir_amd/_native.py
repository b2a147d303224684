"""ctypes binding of libchalamet_hip.so (the C ABI declared in include/chalamet_hip.h).

The library is built in-tree by `make -C chalametpir_amd/csrc` (see __graft_entry__.build()).  There is no CPU
fallback anywhere in this package: if the shared library is missing, `load()` raises, and if no HIP device is
usable every compute entry point returns CPIR_ERR_NO_DEVICE / CPIR_ERR_HIP, surfaced as ChalametPIRError.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libchalamet_hip.so")
CSRC_DIR = os.path.join(_PKG, "csrc")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "chalamet_hip.h")

LWE_DIMENSION = 1774
SEED_BYTE_LEN = 32
FILTER_PARAM_BYTE_LEN = 68
PACK_REFERENCE = 0
PACK_DENSE64 = 1

u8p = C.POINTER(C.c_uint8)
u32p = C.c_void_p  # device or host u32 pointers are passed as raw addresses
vp = C.c_void_p


class DtcLayout(C.Structure):
    """cpir_dtc_layout"""

    _fields_ = [
        ("num_slots", C.c_uint64),
        ("num_cols", C.c_uint32),
        ("mat_elem_bit_len", C.c_uint32),
        ("compression_factor", C.c_uint32),
        ("words_per_row", C.c_uint64),
        ("words_per_row_padded", C.c_uint64),
        ("rows_padded", C.c_uint32),
        ("total_words", C.c_uint64),
        ("packing", C.c_uint32),
        ("fields_per_word", C.c_uint32),
        ("chunk_words", C.c_uint32),
        ("slots_per_chunk", C.c_uint64),
    ]


class KvDb(C.Structure):
    """cpir_kv_db"""

    _fields_ = [
        ("num_pairs", C.c_uint64),
        ("keys", C.c_void_p),
        ("key_off", C.c_void_p),
        ("values", C.c_void_p),
        ("val_off", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/chalamet_hip.h declares appears here
SIGNATURES = {
    "cpir_strerror": (C.c_char_p, [C.c_int]),
    "cpir_last_hip_error": (C.c_char_p, []),
    "cpir_version": (C.c_char_p, []),
    "cpir_xof_permutation": (C.c_char_p, []),
    "cpir_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(vp)]),
    "cpir_host_free": (None, [vp]),
    "cpir_tuning_set": (C.c_int, [C.c_char_p, C.c_int]),
    "cpir_tuning_reset": (None, []),
    "cpir_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "cpir_device_open": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "cpir_device_close": (None, [vp]),
    "cpir_device_ordinal": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "cpir_device_synchronize": (C.c_int, [vp]),
    "cpir_compression_factor": (C.c_uint32, [C.c_uint32]),
    "cpir_find_encoded_db_matrix_element_bit_length": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint32)]),
    "cpir_filter_shape": (C.c_int, [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "cpir_encoded_num_cols": (C.c_uint64, [C.c_uint64, C.c_uint32]),
    "cpir_generate_from_seed": (C.c_int, [C.c_uint64, C.c_uint64, u8p, vp]),
    "cpir_xof_open": (C.c_int, [u8p, C.POINTER(vp)]),
    "cpir_xof_squeeze": (C.c_int, [vp, vp, C.c_size_t]),
    "cpir_xof_close": (None, [vp]),
    "cpir_op_mat_x_mat": (C.c_int, [vp, u32p, C.c_uint64, u32p, C.c_uint64, u32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                    C.c_uint32, C.c_int, vp]),
    "cpir_mat_x_mat_kernel_name": (C.c_char_p, [C.c_uint32]),
    "cpir_dtc_layout_for": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(DtcLayout)]),
    "cpir_dtc_layout_for_packing": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(DtcLayout)]),
    "cpir_shard_unit": (C.c_uint64, [C.POINTER(DtcLayout)]),
    "cpir_op_transpose_compress": (C.c_int, [vp, u32p, C.c_uint64, C.POINTER(DtcLayout), u32p, u32p, vp]),
    "cpir_packed_rhs_plane_bytes": (C.c_uint64, [C.POINTER(DtcLayout)]),
    "cpir_packed_rhs_offered": (C.c_int, [C.POINTER(DtcLayout)]),
    "cpir_op_transpose_compress_with_plane": (C.c_int, [vp, u32p, C.c_uint64, C.POINTER(DtcLayout), u32p, u32p, vp, vp]),
    "cpir_op_mat_x_packed": (C.c_int, [vp, u32p, C.c_uint64, u32p, C.POINTER(DtcLayout), vp, u32p, C.c_uint64, C.c_uint64, C.c_int, vp]),
    "cpir_op_dtc_import": (C.c_int, [vp, u32p, C.POINTER(DtcLayout), u32p, vp]),
    "cpir_op_dtc_export": (C.c_int, [vp, u32p, C.POINTER(DtcLayout), u32p, vp]),
    "cpir_respond_scratch_words": (C.c_uint64, [C.POINTER(DtcLayout)]),
    "cpir_op_respond": (C.c_int, [vp, u32p, C.POINTER(DtcLayout), u32p, C.c_uint64, C.c_uint64, u32p, u32p, vp]),
    "cpir_respond_batch_scratch_words": (C.c_uint64, [C.POINTER(DtcLayout), C.c_uint32]),
    "cpir_op_respond_batch": (C.c_int, [vp, u32p, C.POINTER(DtcLayout), u32p, C.c_uint64, C.c_uint64, C.c_uint32, u32p, u32p, vp]),
    "cpir_op_synth_fill": (C.c_int, [vp, u32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, vp]),
    "cpir_respond_kernel_name": (C.c_char_p, [C.POINTER(DtcLayout)]),
    "cpir_pack_kernel_name": (C.c_char_p, [C.POINTER(DtcLayout)]),
    "cpir_respond_batch_pass_width": (C.c_uint32, [C.POINTER(DtcLayout), C.c_uint32]),
    "cpir_server_setup": (C.c_int, [vp, u8p, u32p, u32p, C.c_uint64, C.c_uint32, C.c_uint32, u32p, C.POINTER(vp)]),
    "cpir_server_setup_kv": (C.c_int, [vp, C.c_uint32, u8p, C.POINTER(KvDb), u8p, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t),
                                       u8p, C.POINTER(vp)]),
    "cpir_server_setup_multi": (C.c_int, [C.POINTER(vp), C.c_uint32, u8p, u32p, u32p, C.c_uint64, C.c_uint32, C.c_uint32, u32p,
                                          C.POINTER(vp)]),
    "cpir_server_setup_kv_multi": (C.c_int, [C.POINTER(vp), C.c_uint32, C.c_uint32, u8p, C.POINTER(KvDb), u8p, C.c_uint32, vp,
                                             C.c_size_t, C.POINTER(C.c_size_t), u8p, C.POINTER(vp)]),
    "cpir_server_group_size": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
    "cpir_server_group_shard": (C.c_int, [vp, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cpir_encode_kv_database": (C.c_int, [C.c_uint32, C.POINTER(KvDb), C.c_uint32, u8p, C.c_uint32, u8p, vp, C.c_uint64,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "cpir_setup_kv_shape": (C.c_int, [C.c_uint32, C.POINTER(KvDb), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]),
    "cpir_hint_partial_device": (C.c_int, [vp, u8p, u32p, u32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u32p,
                                           vp]),
    "cpir_server_from_device_matrix": (C.c_int, [vp, u32p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp,
                                                 C.POINTER(vp)]),
    "cpir_server_from_compressed": (C.c_int, [vp, u32p, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
    "cpir_server_export_compressed": (C.c_int, [vp, u32p, C.c_uint64]),
    "cpir_server_setup_timings": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "cpir_server_host_path_counts": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "cpir_server_retain": (vp, [vp]),
    "cpir_server_release": (None, [vp]),
    "cpir_server_layout": (C.c_int, [vp, C.POINTER(DtcLayout)]),
    "cpir_server_shard": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cpir_server_physical_layout": (C.c_int, [vp, C.POINTER(DtcLayout)]),
    "cpir_server_slots_served": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cpir_server_kept_slots": (C.c_int, [vp, u32p, C.c_uint64]),
    "cpir_host_gather_variant": (C.c_char_p, []),
    "cpir_host_gather_words": (C.c_int, [u32p, u32p, u32p, C.c_uint64]),
    "cpir_host_compress_words": (C.c_int, [u32p, u32p, vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cpir_server_dtc_device_ptr": (vp, [vp]),
    "cpir_server_respond_bytes": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "cpir_server_respond": (C.c_int, [vp, u32p, C.c_uint32, C.c_uint64, u32p]),
    "cpir_server_respond_device": (C.c_int, [vp, u32p, u32p, u32p, vp]),
    "cpir_server_respond_batch_device": (C.c_int, [vp, u32p, C.c_uint32, u32p, u32p, vp]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC_DIR, "-j8", "-s"], check=True)
    # bench-only tools (hbm_read_ceiling, host_respond_bench): their failure must not fail the library build
    subprocess.run(["make", "-C", CSRC_DIR, "-j8", "-s", "-k", "tools"], check=False)
    return LIB_PATH


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same SONAME
    as the system ROCm ones).  If this library pulled in the system runtime first and torch then loaded its bundled HSA
    runtime, the process would hold two runtimes and torch would report "No HIP GPUs are available"; device pointers and
    streams could not be shared either.  So when torch is installed its bundled runtime is loaded first and
    libchalamet_hip.so binds to it by SONAME.  Without torch (a C / Rust host) the system ROCm runtime is used.
    CPIR_HIP_RUNTIME=system skips the preload."""
    if os.environ.get("CPIR_HIP_RUNTIME", "") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def use_library(path: str) -> None:
    """Tuning scripts only: bind another build of the library (the -DCPIR_DIAG one, `make -C chalametpir_amd/csrc diag`) instead of the
    release library.  Must be called before anything has loaded the library; the package, the tests and bench.py never call it."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("the library is already loaded")
    LIB_PATH = os.path.abspath(path)


def use_diag_build() -> str:
    """Tuning scripts only: build (if need be) and bind the diagnosis build of the library; returns its path."""
    path = os.path.join(_PKG, "lib", "diag", "libchalamet_hip.so")
    # (the library travels to the GPU box, its objects do not: make would rebuild everything there -- two minutes -- although nothing changed)
    sources = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".hip", ".cpp", ".hpp"))] + [HEADER_PATH]
    if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in sources):
        subprocess.run(["make", "-C", CSRC_DIR, "-s", "-j8", "diag"], check=True)
    use_library(path)
    return path


def load():
    """dlopen libchalamet_hip.so and type every entry point.  Raises if the library is not built: fail loudly."""
    global _lib
    if _lib is None:
        _preload_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC_DIR}` (or __graft_entry__.build()). "
                "chalametpir_amd has no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
