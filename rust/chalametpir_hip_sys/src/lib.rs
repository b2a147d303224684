//! SOURCE ONLY -- not compile-tested (no Rust toolchain in the build image).
//! Raw declarations of the C ABI in `include/chalamet_hip.h`; keep the two in sync.
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_int, c_void};

pub const CPIR_LWE_DIMENSION: u32 = 1774;
pub const CPIR_SEED_BYTE_LEN: usize = 32;
pub const CPIR_FILTER_PARAM_BYTE_LEN: usize = 68;
pub const CPIR_SETUP_TIMING_COUNT: usize = 8;
pub const CPIR_HOST_PATH_COUNT: usize = 8;

pub const CPIR_OK: c_int = 0;
pub const CPIR_ERR_INVALID_MATRIX_DIMENSION: c_int = 1;
pub const CPIR_ERR_INCOMPATIBLE_DIM_MATMUL: c_int = 2;
pub const CPIR_ERR_INVALID_NUMBER_OF_ELEMENTS: c_int = 4;
pub const CPIR_ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED: c_int = 5;
pub const CPIR_ERR_FAILED_TO_DESERIALIZE_MATRIX: c_int = 7;
pub const CPIR_ERR_EMPTY_KV_DATABASE: c_int = 8;
pub const CPIR_ERR_EXHAUSTED_ATTEMPTS_3WISE: c_int = 9;
pub const CPIR_ERR_EXHAUSTED_ATTEMPTS_4WISE: c_int = 10;
pub const CPIR_ERR_KV_DATABASE_SIZE_TOO_LARGE: c_int = 14;
pub const CPIR_ERR_UNSUPPORTED_ARITY: c_int = 17;
pub const CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH: c_int = 19;
pub const CPIR_ERR_NO_DEVICE: c_int = 64;
pub const CPIR_ERR_HIP: c_int = 65;
pub const CPIR_ERR_OUT_OF_DEVICE_MEMORY: c_int = 66;
pub const CPIR_ERR_BUFFER_TOO_SMALL: c_int = 67;
pub const CPIR_ERR_INVALID_ARGUMENT: c_int = 68;
pub const CPIR_ERR_SHARD_RANGE: c_int = 69;

#[repr(C)]
pub struct cpir_device {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cpir_server {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cpir_xof {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct cpir_dtc_layout {
    pub num_slots: u64,
    pub num_cols: u32,
    pub mat_elem_bit_len: u32,
    pub compression_factor: u32,
    pub words_per_row: u64,
    pub words_per_row_padded: u64,
    pub rows_padded: u32,
    pub total_words: u64,
    pub packing: u32,
    pub fields_per_word: u32,
    pub chunk_words: u32,
    pub slots_per_chunk: u64,
}

#[repr(C)]
pub struct cpir_kv_db {
    pub num_pairs: u64,
    pub keys: *const u8,
    pub key_off: *const u64,
    pub values: *const u8,
    pub val_off: *const u64,
}

unsafe extern "C" {
    pub fn cpir_strerror(status: c_int) -> *const c_char;
    pub fn cpir_last_hip_error() -> *const c_char;
    pub fn cpir_version() -> *const c_char;

    pub fn cpir_device_count(count: *mut c_int) -> c_int;
    pub fn cpir_device_open(ordinal: c_int, out: *mut *mut cpir_device) -> c_int;
    pub fn cpir_device_close(dev: *mut cpir_device);
    pub fn cpir_device_ordinal(dev: *const cpir_device, ordinal: *mut c_int) -> c_int;
    pub fn cpir_device_synchronize(dev: *mut cpir_device) -> c_int;

    pub fn cpir_compression_factor(mat_elem_bit_len: u32) -> u32;
    pub fn cpir_find_encoded_db_matrix_element_bit_length(db_entry_count: u64, mat_elem_bit_len: *mut u32) -> c_int;
    pub fn cpir_filter_shape(arity: u32, db_entry_count: u64, segment_length: *mut u32, segment_count_length: *mut u32,
                             num_fingerprints: *mut u64) -> c_int;
    pub fn cpir_encoded_num_cols(max_value_byte_len: u64, mat_elem_bit_len: u32) -> u64;
    pub fn cpir_generate_from_seed(rows: u64, cols: u64, seed: *const u8, out: *mut u32) -> c_int;
    pub fn cpir_xof_open(seed: *const u8, out: *mut *mut cpir_xof) -> c_int;
    pub fn cpir_xof_squeeze(xof: *mut cpir_xof, out: *mut c_void, bytes: usize) -> c_int;
    pub fn cpir_xof_close(xof: *mut cpir_xof);
    pub fn cpir_xof_permutation() -> *const c_char;
    pub fn cpir_host_alloc(bytes: usize, out: *mut *mut core::ffi::c_void) -> c_int;
    pub fn cpir_host_free(p: *mut core::ffi::c_void);
    pub fn cpir_dtc_layout_for(num_slots: u64, num_cols: u32, mat_elem_bit_len: u32, out: *mut cpir_dtc_layout) -> c_int;
    pub fn cpir_dtc_layout_for_packing(num_slots: u64, num_cols: u32, mat_elem_bit_len: u32, packing: u32, out: *mut cpir_dtc_layout) -> c_int;
    pub fn cpir_shard_unit(layout: *const cpir_dtc_layout) -> u64;

    pub fn cpir_op_mat_x_mat(dev: *mut cpir_device, a: *const u32, lda: u64, d: *const u32, ldd: u64, m: *mut u32, ldm: u64,
                             rows: u64, inner: u64, cols: u64, rhs_max_bits: u32, accumulate: c_int, stream: *mut c_void) -> c_int;
    pub fn cpir_mat_x_mat_kernel_name(rhs_max_bits: u32) -> *const c_char;
    pub fn cpir_op_transpose_compress(dev: *mut cpir_device, d: *const u32, ldd: u64, layout: *const cpir_dtc_layout,
                                      dtc: *mut u32, or_of_entries: *mut u32, stream: *mut c_void) -> c_int;
    pub fn cpir_packed_rhs_plane_bytes(layout: *const cpir_dtc_layout) -> u64;
    pub fn cpir_packed_rhs_offered(layout: *const cpir_dtc_layout) -> c_int;
    pub fn cpir_op_transpose_compress_with_plane(dev: *mut cpir_device, d: *const u32, ldd: u64, layout: *const cpir_dtc_layout,
                                                 dtc: *mut u32, or_of_entries: *mut u32, hi_plane: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn cpir_op_mat_x_packed(dev: *mut cpir_device, a: *const u32, lda: u64, dtc: *const u32, layout: *const cpir_dtc_layout,
                                hi_plane: *const c_void, m: *mut u32, ldm: u64, rows: u64, accumulate: c_int, stream: *mut c_void) -> c_int;
    pub fn cpir_op_dtc_import(dev: *mut cpir_device, compressed: *const u32, layout: *const cpir_dtc_layout, dtc: *mut u32,
                              stream: *mut c_void) -> c_int;
    pub fn cpir_op_dtc_export(dev: *mut cpir_device, dtc: *const u32, layout: *const cpir_dtc_layout, compressed: *mut u32,
                              stream: *mut c_void) -> c_int;
    pub fn cpir_respond_scratch_words(layout: *const cpir_dtc_layout) -> u64;
    pub fn cpir_op_respond(dev: *mut cpir_device, dtc: *const u32, layout: *const cpir_dtc_layout, q: *const u32, q_len: u64,
                           q_slot_offset: u64, r: *mut u32, scratch: *mut u32, stream: *mut c_void) -> c_int;
    pub fn cpir_respond_batch_scratch_words(layout: *const cpir_dtc_layout, batch: u32) -> u64;
    pub fn cpir_op_respond_batch(dev: *mut cpir_device, dtc: *const u32, layout: *const cpir_dtc_layout, q: *const u32, q_len: u64,
                                 q_slot_offset: u64, batch: u32, r: *mut u32, scratch: *mut u32, stream: *mut c_void) -> c_int;
    pub fn cpir_op_synth_fill(dev: *mut cpir_device, out: *mut u32, count: u64, seed: u64, index0: u64, mask: u32,
                              stream: *mut c_void) -> c_int;
    pub fn cpir_tuning_set(key: *const c_char, value: c_int) -> c_int;
    pub fn cpir_tuning_reset();
    pub fn cpir_pack_kernel_name(layout: *const cpir_dtc_layout) -> *const c_char;
    pub fn cpir_respond_batch_pass_width(layout: *const cpir_dtc_layout, batch: u32) -> u32;
    pub fn cpir_respond_kernel_name(layout: *const cpir_dtc_layout) -> *const c_char;

    pub fn cpir_server_setup(dev: *mut cpir_device, seed_mu: *const u8, pub_mat_a: *const u32, d: *const u32, n: u64, c: u32,
                             mat_elem_bit_len: u32, hint_out: *mut u32, out: *mut *mut cpir_server) -> c_int;
    pub fn cpir_server_setup_kv(dev: *mut cpir_device, arity: u32, seed_mu: *const u8, db: *const cpir_kv_db,
                                filter_seed_material: *const u8, max_attempts: u32, hint_bytes_out: *mut u8, hint_bytes_cap: usize,
                                hint_bytes_len: *mut usize, filter_param_bytes_out: *mut u8, out: *mut *mut cpir_server) -> c_int;
    pub fn cpir_server_setup_multi(devs: *const *mut cpir_device, n_dev: u32, seed_mu: *const u8, pub_mat_a: *const u32, d: *const u32,
                                   n: u64, c: u32, mat_elem_bit_len: u32, hint_out: *mut u32, out: *mut *mut cpir_server) -> c_int;
    pub fn cpir_server_setup_kv_multi(devs: *const *mut cpir_device, n_dev: u32, arity: u32, seed_mu: *const u8, db: *const cpir_kv_db,
                                      filter_seed_material: *const u8, max_attempts: u32, hint_bytes_out: *mut u8, hint_bytes_cap: usize,
                                      hint_bytes_len: *mut usize, filter_param_bytes_out: *mut u8, out: *mut *mut cpir_server) -> c_int;
    pub fn cpir_server_group_size(srv: *const cpir_server, shards: *mut u32) -> c_int;
    pub fn cpir_server_group_shard(srv: *const cpir_server, index: u32, device_ordinal: *mut c_int, slot_offset: *mut u64,
                                   num_slots: *mut u64) -> c_int;
    pub fn cpir_encode_kv_database(arity: u32, db: *const cpir_kv_db, mat_elem_bit_len: u32, filter_seed_material: *const u8,
                                   max_attempts: u32, filter_param_bytes_out: *mut u8, d_out: *mut u32, d_cap_words: u64,
                                   n: *mut u64, c: *mut u32) -> c_int;
    pub fn cpir_setup_kv_shape(arity: u32, db: *const cpir_kv_db, mat_elem_bit_len: *mut u32, n: *mut u64, c: *mut u32,
                               hint_bytes_len: *mut usize) -> c_int;
    pub fn cpir_hint_partial_device(dev: *mut cpir_device, seed_mu: *const u8, pub_mat_a: *const u32, d_dev: *const u32, ldd: u64,
                                    slot_offset: u64, n_shard: u64, total_slots: u64, c: u32, rhs_max_bits: u32, m_dev: *mut u32,
                                    stream: *mut c_void) -> c_int;
    pub fn cpir_server_from_device_matrix(dev: *mut cpir_device, d_dev: *const u32, ldd: u64, n_shard: u64, c: u32,
                                          mat_elem_bit_len: u32, slot_offset: u64, total_slots: u64, stream: *mut c_void,
                                          out: *mut *mut cpir_server) -> c_int;
    pub fn cpir_server_from_compressed(dev: *mut cpir_device, compressed: *const u32, c: u32, n: u64, mat_elem_bit_len: u32,
                                       out: *mut *mut cpir_server) -> c_int;
    pub fn cpir_server_export_compressed(srv: *const cpir_server, compressed_out: *mut u32, out_words: u64) -> c_int;
    pub fn cpir_server_setup_timings(srv: *const cpir_server, out: *mut f64) -> c_int;
    pub fn cpir_server_host_path_counts(srv: *const cpir_server, out: *mut u64) -> c_int;
    pub fn cpir_server_retain(srv: *mut cpir_server) -> *mut cpir_server;
    pub fn cpir_server_release(srv: *mut cpir_server);
    pub fn cpir_server_layout(srv: *const cpir_server, out: *mut cpir_dtc_layout) -> c_int;
    pub fn cpir_server_shard(srv: *const cpir_server, slot_offset: *mut u64, total_slots: *mut u64) -> c_int;
    pub fn cpir_server_dtc_device_ptr(srv: *const cpir_server) -> *const u32;
    pub fn cpir_server_physical_layout(srv: *const cpir_server, out: *mut cpir_dtc_layout) -> c_int;
    pub fn cpir_server_slots_served(srv: *const cpir_server, served: *mut u64, of_slots: *mut u64) -> c_int;
    pub fn cpir_server_kept_slots(srv: *const cpir_server, out: *mut u32, cap: u64) -> c_int;
    pub fn cpir_host_gather_variant() -> *const c_char;
    pub fn cpir_host_gather_words(dst: *mut u32, src: *const u32, idx: *const u32, count: u64) -> c_int;
    pub fn cpir_host_compress_words(dst: *mut u32, src: *const u32, bits: *const u8, s_lo: u64, s_hi: u64, count: *mut u64) -> c_int;
    pub fn cpir_server_respond_bytes(srv: *const cpir_server, query: *const u8, query_len: usize, response: *mut u8,
                                     response_cap: usize, response_len: *mut usize) -> c_int;
    pub fn cpir_server_respond(srv: *const cpir_server, q: *const u32, q_rows: u32, q_cols: u64, r_out: *mut u32) -> c_int;
    pub fn cpir_server_respond_device(srv: *const cpir_server, q_dev: *const u32, r_dev: *mut u32, scratch_dev: *mut u32,
                                      stream: *mut c_void) -> c_int;
    pub fn cpir_server_respond_batch_device(srv: *const cpir_server, q_dev: *const u32, batch: u32, r_dev: *mut u32,
                                            scratch_dev: *mut u32, stream: *mut c_void) -> c_int;
}
