"""chalametpir_amd -- MI355X-native server hot path for ChalametPIR (reference itzmeanjan/ChalametPIR v0.7.0).

Public surface mirrors the reference's server crate (chalametpir_server/src/lib.rs:80-81): `Server`, `ChalametPIRError`,
`SEED_BYTE_LEN`.  All compute runs in libchalamet_hip.so (hand-written HIP kernels for gfx950, include/chalamet_hip.h);
importing this package never falls back to a CPU implementation.
"""
from .errors import ChalametPIRError
from .params import LWE_DIMENSION, SEED_BYTE_LEN
from .server import (Device, PinnedArray, SeedExpander, Server, dtc_layout_for, encode_kv_database, encoded_num_cols, filter_shape,
                     find_encoded_db_matrix_element_bit_length, generate_from_seed, host_compress, host_gather, host_gather_variant, mat_x_mat_kernel_name, pack_kernel_name, packed_rhs_offered, packed_rhs_plane_bytes, respond_batch_pass_width, tuning_reset, tuning_set)

__all__ = ["Server", "Device", "PinnedArray", "SeedExpander", "ChalametPIRError", "SEED_BYTE_LEN", "LWE_DIMENSION", "dtc_layout_for", "encode_kv_database", "encoded_num_cols",
           "filter_shape", "find_encoded_db_matrix_element_bit_length", "generate_from_seed", "host_compress", "host_gather", "host_gather_variant", "mat_x_mat_kernel_name", "pack_kernel_name", "packed_rhs_offered", "packed_rhs_plane_bytes", "respond_batch_pass_width", "tuning_reset", "tuning_set"]
