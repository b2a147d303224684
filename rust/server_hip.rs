//! SOURCE ONLY -- not compile-tested (no Rust toolchain in the build image).
//!
//! The `#[cfg(feature = "hip")]` variant of `chalametpir_server::Server` (reference chalametpir_server/src/server.rs:15-219).
//! Same public surface as the reference: `Server::setup::<ARITY>(&seed, db) -> Result<(Server, Vec<u8>, Vec<u8>), ChalametPIRError>`,
//! `Server::respond(&self, &[u8]) -> Result<Vec<u8>, ChalametPIRError>`, `Server: Clone + Send + Sync`.
//! Unlike the reference's `gpu` feature (setup only, server.rs:103-167) the packed database stays resident in HBM and
//! `respond` runs on the GPU as well.

use chalametpir_common::{error::ChalametPIRError, params::{SEED_BYTE_LEN, SERVER_SETUP_MAX_ATTEMPT_COUNT}};
use chalametpir_hip_sys as sys;
use std::collections::HashMap;

/// Owns one reference on a `cpir_server`; `Clone` retains, `Drop` releases (server.rs:15 `#[derive(Clone)]`).
pub struct Server {
    handle: *mut sys::cpir_server,
    response_cols: u32,
}

// The handle is immutable after setup and `cpir_server_respond*` is thread-safe and re-entrant (include/chalamet_hip.h).
unsafe impl Send for Server {}
unsafe impl Sync for Server {}

impl Clone for Server {
    fn clone(&self) -> Self {
        Server { handle: unsafe { sys::cpir_server_retain(self.handle) }, response_cols: self.response_cols }
    }
}

impl Drop for Server {
    fn drop(&mut self) {
        unsafe { sys::cpir_server_release(self.handle) }
    }
}

/// cpir_status -> ChalametPIRError.  1..19 are the reference's own variants (error.rs:24-49); 64.. need the new `Hip*`
/// variants below added to the enum (or, if the enum must stay frozen, the closest `Vulkan*` ones, error.rs:10-22).
fn map_status(status: i32, max_attempts: usize) -> ChalametPIRError {
    match status {
        sys::CPIR_ERR_INVALID_MATRIX_DIMENSION => ChalametPIRError::InvalidMatrixDimension,
        sys::CPIR_ERR_INCOMPATIBLE_DIM_MATMUL => ChalametPIRError::IncompatibleDimensionForMatrixMultiplication,
        sys::CPIR_ERR_INVALID_NUMBER_OF_ELEMENTS => ChalametPIRError::InvalidNumberOfElementsInMatrix,
        sys::CPIR_ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED => ChalametPIRError::IncompatibleDimensionForRowVectorTransposedMatrixMultiplication,
        sys::CPIR_ERR_FAILED_TO_DESERIALIZE_MATRIX => ChalametPIRError::FailedToDeserializeMatrixFromBytes,
        sys::CPIR_ERR_EMPTY_KV_DATABASE => ChalametPIRError::EmptyKVDatabase,
        sys::CPIR_ERR_EXHAUSTED_ATTEMPTS_3WISE => ChalametPIRError::ExhaustedAllAttemptsToBuild3WiseXorFilter(max_attempts),
        sys::CPIR_ERR_EXHAUSTED_ATTEMPTS_4WISE => ChalametPIRError::ExhaustedAllAttemptsToBuild4WiseXorFilter(max_attempts),
        sys::CPIR_ERR_KV_DATABASE_SIZE_TOO_LARGE => ChalametPIRError::KVDatabaseSizeTooLarge,
        sys::CPIR_ERR_UNSUPPORTED_ARITY => ChalametPIRError::UnsupportedArityForBinaryFuseFilter,
        sys::CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH => ChalametPIRError::ImpossibleEncodedDBMatrixElementBitLength,
        sys::CPIR_ERR_NO_DEVICE => ChalametPIRError::HipDeviceNotFound,          // ~ VulkanPhysicalDeviceNotFound
        sys::CPIR_ERR_OUT_OF_DEVICE_MEMORY => ChalametPIRError::HipOutOfMemory,  // ~ VulkanBufferCreationFailed
        _ => ChalametPIRError::HipRuntimeCallFailed,                             // ~ VulkanCommandBufferExecutionFailed
    }
}

impl Server {
    pub fn setup<const ARITY: u32>(seed_μ: &[u8; SEED_BYTE_LEN], db: HashMap<&[u8], &[u8]>) -> Result<(Server, Vec<u8>, Vec<u8>), ChalametPIRError> {
        const { assert!(ARITY == 3 || ARITY == 4) }
        if db.is_empty() {
            return Err(ChalametPIRError::EmptyKVDatabase); // server.rs:48-51
        }

        // flatten the HashMap into the cpir_kv_db arrays (iteration order = key order handed to the encoder)
        let mut keys = Vec::new();
        let mut vals = Vec::new();
        let mut key_off = vec![0u64];
        let mut val_off = vec![0u64];
        for (k, v) in db.iter() {
            keys.extend_from_slice(k);
            vals.extend_from_slice(v);
            key_off.push(keys.len() as u64);
            val_off.push(vals.len() as u64);
        }
        let flat = sys::cpir_kv_db { num_pairs: db.len() as u64, keys: keys.as_ptr(), key_off: key_off.as_ptr(), values: vals.as_ptr(), val_off: val_off.as_ptr() };

        let (mut b, mut n, mut c, mut hint_len) = (0u32, 0u64, 0u32, 0usize);
        let st = unsafe { sys::cpir_setup_kv_shape(ARITY, &flat, &mut b, &mut n, &mut c, &mut hint_len) };
        if st != sys::CPIR_OK {
            return Err(map_status(st, SERVER_SETUP_MAX_ATTEMPT_COUNT));
        }

        // replaces gpu_utils::setup_gpu(), gpu_utils.rs:25.  CHALAMET_HIP_DEVICES=0,1,2,3 splits the database over several GPUs of
        // this process behind the one handle (cpir_server_setup_kv_multi); default: device 0.
        let ordinals: Vec<i32> = std::env::var("CHALAMET_HIP_DEVICES")
            .ok()
            .map(|s| s.split(',').filter_map(|x| x.trim().parse().ok()).collect())
            .filter(|v: &Vec<i32>| !v.is_empty())
            .unwrap_or_else(|| vec![0]);
        let mut devs: Vec<*mut sys::cpir_device> = Vec::with_capacity(ordinals.len());
        let close_all = |devs: &Vec<*mut sys::cpir_device>| devs.iter().for_each(|d| unsafe { sys::cpir_device_close(*d) });
        for o in &ordinals {
            let mut dev = core::ptr::null_mut();
            let st = unsafe { sys::cpir_device_open(*o, &mut dev) };
            if st != sys::CPIR_OK {
                close_all(&devs);
                return Err(map_status(st, SERVER_SETUP_MAX_ATTEMPT_COUNT));
            }
            devs.push(dev);
        }

        let mut hint_words = vec![0u32; hint_len.div_ceil(4)]; // 4-byte aligned backing store for the wire image
        let mut filter_param_bytes = vec![0u8; sys::CPIR_FILTER_PARAM_BYTE_LEN];
        let mut handle = core::ptr::null_mut();
        let mut written = 0usize;
        let st = unsafe {
            if devs.len() == 1 {
                sys::cpir_server_setup_kv(devs[0], ARITY, seed_μ.as_ptr(), &flat, core::ptr::null(), SERVER_SETUP_MAX_ATTEMPT_COUNT as u32,
                                          hint_words.as_mut_ptr().cast(), hint_len, &mut written, filter_param_bytes.as_mut_ptr(), &mut handle)
            } else {
                sys::cpir_server_setup_kv_multi(devs.as_ptr(), devs.len() as u32, ARITY, seed_μ.as_ptr(), &flat, core::ptr::null(),
                                                SERVER_SETUP_MAX_ATTEMPT_COUNT as u32, hint_words.as_mut_ptr().cast(), hint_len, &mut written,
                                                filter_param_bytes.as_mut_ptr(), &mut handle)
            }
        };
        close_all(&devs); // the server keeps its own references on the devices
        if st != sys::CPIR_OK {
            return Err(map_status(st, SERVER_SETUP_MAX_ATTEMPT_COUNT));
        }
        let hint_bytes = unsafe { core::slice::from_raw_parts(hint_words.as_ptr().cast::<u8>(), written) }.to_vec();
        Ok((Server { handle, response_cols: c }, hint_bytes, filter_param_bytes))
    }

    pub fn respond(&self, query: &[u8]) -> Result<Vec<u8>, ChalametPIRError> {
        let mut response = vec![0u8; 8 + 4 * self.response_cols as usize];
        let mut len = 0usize;
        let st = unsafe { sys::cpir_server_respond_bytes(self.handle, query.as_ptr(), query.len(), response.as_mut_ptr(), response.len(), &mut len) };
        if st != sys::CPIR_OK {
            return Err(map_status(st, 0));
        }
        response.truncate(len);
        Ok(response)
    }
}

/// A query buffer in page-locked host memory (`cpir_host_alloc`): bytes read from the network straight into it are read by the respond
/// kernel IN PLACE, without the staging copy a pageable `Vec<u8>` goes through (one caller, 2^20 keys x 1 kB: 214 us per query against
/// 231 us).  Optional: `Server::respond` takes any `&[u8]`; this is what `examples/server.rs` would read its query into
/// (`stream.read_exact(buf.as_mut_slice())`, chalametpir_server/examples/server.rs:75-84).  The wire image starts 8 bytes into the
/// allocation so that the u32 words behind the 8-byte header (matrix.rs:947-971) are 16-byte aligned, which the in-place read needs.
pub struct PinnedQuery {
    base: *mut u8,
    len: usize,
}

unsafe impl Send for PinnedQuery {}

impl PinnedQuery {
    /// `wire_len` = 8 + 4 * num_fingerprints (the length `Client::query` produces)
    pub fn new(wire_len: usize) -> Result<PinnedQuery, ChalametPIRError> {
        let mut p = core::ptr::null_mut();
        let st = unsafe { sys::cpir_host_alloc(wire_len + 8, &mut p) };
        if st != sys::CPIR_OK {
            return Err(map_status(st, 0));
        }
        Ok(PinnedQuery { base: p.cast(), len: wire_len })
    }
    pub fn as_mut_slice(&mut self) -> &mut [u8] {
        unsafe { core::slice::from_raw_parts_mut(self.base.add(8), self.len) }
    }
    pub fn as_slice(&self) -> &[u8] {
        unsafe { core::slice::from_raw_parts(self.base.add(8), self.len) }
    }
}

impl Drop for PinnedQuery {
    fn drop(&mut self) {
        unsafe { sys::cpir_host_free(self.base.cast()) }
    }
}
