/*
 * chalamet_hip.h -- C ABI of libchalamet_hip.so: the MI355X (gfx950) replacement for the `gpu` feature of
 * chalametpir_server (reference: itzmeanjan/ChalametPIR v0.7.0).
 *
 * This is the drop-in boundary.  Everything a Rust FFI shim for `chalametpir_server::Server::{setup,respond}`
 * would bind is declared here with plain pointers and sizes; no C++/torch types cross it.  Each entry point cites
 * the reference interface it replaces (paths relative to the reference checkout):
 *   server.rs    = chalametpir_server/src/server.rs
 *   gpu_utils.rs = chalametpir_server/src/gpu/gpu_utils.rs
 *   matrix.rs    = chalametpir_common/src/matrix.rs
 *   error.rs     = chalametpir_common/src/error.rs
 *
 * Conventions
 *   - every function returns a cpir_status (0 = ok); outputs go through pointers;
 *   - matrices are row-major u32, (r,c) -> elems[r*cols + c]                         (matrix.rs:26-31,1013-1029)
 *   - "wire bytes" are Matrix::to_bytes images: [rows u32 LE][cols u32 LE][elems LE]  (matrix.rs:947-1010)
 *   - all element counts are 64-bit (the reference sizes buffers in u32: matrix.rs:50,71,546,988,1048)
 *   - `stream` arguments are hipStream_t passed as void* (NULL = HIP's default (null) stream, as in any HIP call)
 *   - host-pointer entry points are synchronous (return when outputs are written), like the reference's
 *     fence-waited Vulkan calls (gpu_utils.rs:129-135,213-219); *_device entry points only enqueue.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point returns
 *     CPIR_ERR_NO_DEVICE / CPIR_ERR_HIP.
 */
#ifndef CHALAMET_HIP_H
#define CHALAMET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPIR_LWE_DIMENSION 1774u  /* params.rs:1  */
#define CPIR_SEED_BYTE_LEN 32u    /* params.rs:5  */
#define CPIR_MIN_ELEM_BIT_LEN 4u  /* params.rs:14 */
#define CPIR_MAX_ELEM_BIT_LEN 14u /* params.rs:17 */

/* Status codes.  1..19 map one-to-one onto the ChalametPIRError variants this path can raise (error.rs:24-49);
 * 64.. replace the thirteen Vulkan* variants (error.rs:10-22) that the reference's GPU plugin raises. */
typedef enum cpir_status {
  CPIR_OK = 0,
  CPIR_ERR_INVALID_MATRIX_DIMENSION = 1,             /* InvalidMatrixDimension                       error.rs:25 */
  CPIR_ERR_INCOMPATIBLE_DIM_MATMUL = 2,              /* IncompatibleDimensionForMatrixMultiplication error.rs:26 */
  CPIR_ERR_INVALID_NUMBER_OF_ELEMENTS = 4,           /* InvalidNumberOfElementsInMatrix              error.rs:28 */
  CPIR_ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED = 5, /* IncompatibleDimensionForRowVectorTransposedMatrixMultiplication error.rs:29 */
  CPIR_ERR_FAILED_TO_DESERIALIZE_MATRIX = 7,         /* FailedToDeserializeMatrixFromBytes           error.rs:31 */
  CPIR_ERR_EMPTY_KV_DATABASE = 8,                    /* EmptyKVDatabase                              error.rs:34 */
  CPIR_ERR_EXHAUSTED_ATTEMPTS_3WISE = 9,             /* ExhaustedAllAttemptsToBuild3WiseXorFilter    error.rs:35 */
  CPIR_ERR_EXHAUSTED_ATTEMPTS_4WISE = 10,            /* ExhaustedAllAttemptsToBuild4WiseXorFilter    error.rs:36 */
  CPIR_ERR_KV_DATABASE_SIZE_TOO_LARGE = 14,          /* KVDatabaseSizeTooLarge                       error.rs:42 */
  CPIR_ERR_UNSUPPORTED_ARITY = 17,                   /* UnsupportedArityForBinaryFuseFilter          error.rs:47 */
  CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH = 19,       /* ImpossibleEncodedDBMatrixElementBitLength    error.rs:49 */
  CPIR_ERR_NO_DEVICE = 64,        /* ~ VulkanLibraryNotFound / VulkanPhysicalDeviceNotFound   (gpu_utils.rs:26,43,59) */
  CPIR_ERR_HIP = 65,              /* any failing HIP runtime call; ~ VulkanCommandBufferExecutionFailed (gpu_utils.rs:129-135) */
  CPIR_ERR_OUT_OF_DEVICE_MEMORY = 66, /* ~ VulkanBufferCreationFailed (gpu_utils.rs:102,116,153) */
  CPIR_ERR_BUFFER_TOO_SMALL = 67, /* caller-provided output buffer too small (no reference analogue: Rust returns Vec) */
  CPIR_ERR_INVALID_ARGUMENT = 68, /* NULL pointer / nonsensical size on the C side (unrepresentable in the Rust API) */
  CPIR_ERR_SHARD_RANGE = 69       /* shard boundaries not aligned to the packing unit */
} cpir_status;

/* Static description of a status; never NULL. (Display impl, error.rs:51-100) */
const char* cpir_strerror(int status);
/* Text of the last failing HIP call on the calling thread ("" if none). */
const char* cpir_last_hip_error(void);
/* Library version string, e.g. "chalamet_hip 0.1.0 (gfx950)". */
const char* cpir_version(void);

/* ------------------------------------------------------------------------------------------------
 * Device context: replaces gpu_utils::setup_gpu() -> (Device, Queue, MemoryAllocator, CmdBufAllocator)
 * (gpu_utils.rs:25-79).  Ref-counted; servers keep their device alive.
 * ------------------------------------------------------------------------------------------------ */
typedef struct cpir_device cpir_device;

int cpir_device_count(int* count);
int cpir_device_open(int ordinal, cpir_device** out);
void cpir_device_close(cpir_device* dev);
int cpir_device_ordinal(const cpir_device* dev, int* ordinal);
int cpir_device_synchronize(cpir_device* dev);

/* Page-locked host memory for query buffers (optional): cpir_server_respond* detects a query that lies in page-locked memory
 * (from here, hipHostMalloc or hipHostRegister) and uploads it by DMA straight from the caller's buffer instead of staging it
 * through the library's own pinned block -- about 80 us less per 4.7 MB query.  Any other pointer works as before. */
int cpir_host_alloc(size_t bytes, void** out);
void cpir_host_free(void* p);

/* ------------------------------------------------------------------------------------------------
 * Shape helpers (host-only arithmetic, no device needed).
 * ------------------------------------------------------------------------------------------------ */
/* compression factor of the packed DB for an element bit length: 2 (11..14), 3 (9..10), 4 (4..8); 0 = invalid
 * (matrix.rs:103-167) */
uint32_t cpir_compression_factor(uint32_t mat_elem_bit_len);
/* Server::find_encoded_db_matrix_element_bit_length (server.rs:193-218) */
int cpir_find_encoded_db_matrix_element_bit_length(uint64_t db_entry_count, uint32_t* mat_elem_bit_len);
/* Filter shape for n keys: N = num_fingerprints (binary_fuse_filter.rs:52-67,261-276,519-538) */
int cpir_filter_shape(uint32_t arity, uint64_t db_entry_count, uint32_t* segment_length,
                      uint32_t* segment_count_length, uint64_t* num_fingerprints);
/* C = cols of D for the longest value (matrix.rs:694-700) */
uint64_t cpir_encoded_num_cols(uint64_t max_value_byte_len, uint32_t mat_elem_bit_len);
/* Matrix::generate_from_seed (matrix.rs:541-558): TurboSHAKE128(seed || 0x1F) squeezed into rows*cols LE u32.
 * Host-side, sequential by construction of the sponge. */
int cpir_generate_from_seed(uint64_t rows, uint64_t cols, const uint8_t seed[CPIR_SEED_BYTE_LEN], uint32_t* out);
/* The same stream, squeezed piecemeal: open = TurboSHAKE128 absorbed seed || 0x1F, every squeeze continues where the last one stopped
 * (so rows * cols * 4 bytes squeezed in any pieces are generate_from_seed's matrix).  Lets ONE process of a node expand A block by block
 * and hand column slabs to the ranks that hold the shards (chalametpir_amd.distributed.scatter_public_matrix) instead of every rank
 * squeezing the whole sponge.  Not thread-safe per handle. */
typedef struct cpir_xof cpir_xof;
int cpir_xof_open(const uint8_t seed[CPIR_SEED_BYTE_LEN], cpir_xof** out);
int cpir_xof_squeeze(cpir_xof* xof, void* out, size_t bytes);
void cpir_xof_close(cpir_xof* xof);
/* Which Keccak-p[1600,12] implementation this host runs for the XOF ("scalar" or "avx512vl (lane per xmm)"); picked once at
 * load time from the CPU's features, CPIR_XOF_SCALAR=1 in the environment forces the scalar one. */
const char* cpir_xof_permutation(void);

/* ------------------------------------------------------------------------------------------------
 * Low-level device operations on caller-owned DEVICE pointers (what gpu_utils::mat_x_mat / mat_transpose and
 * the two GLSL kernels provide, plus the respond mat-vec the reference only has on the CPU).
 * The caller (e.g. a torch process) owns memory and stream; these only enqueue work.
 * ------------------------------------------------------------------------------------------------ */

/* gpu_utils::mat_x_mat + shaders/mat_x_mat.glsl (gpu_utils.rs:156-220) == impl Mul for &Matrix (matrix.rs:1040-1059):
 *   M[r][c] (+)= sum_k A[r][k] *wrap D[k][c],  A rows x inner (leading dim lda), D inner x cols (ldd), M rows x cols (ldm).
 * rhs_max_bits: an upper bound on the bit width of every D entry: <= 16 selects the matrix-core kernel (exact signed-byte split,
 * csrc/matmul_mfma.hip; it needs A 16-byte aligned with lda and inner multiples of 4, else the packed 16-bit dot-product kernel on
 * the VALU runs), 32 the general u32 kernel; results are identical whenever the bound is true.  The matrix-core path keeps its
 * prepared right-hand side (2 bytes per entry of D) in scratch memory of its own (hipMalloc now, freed by a background thread once the
 * work enqueued here has completed -- not hipMallocAsync: see scratch_acquire in csrc/cpir_internal.hpp).
 * accumulate != 0 adds into M (used for K-sharded / row-block pipelined hints), else M is overwritten. */
int cpir_op_mat_x_mat(cpir_device* dev, const uint32_t* A, uint64_t lda, const uint32_t* D, uint64_t ldd, uint32_t* M,
                      uint64_t ldm, uint64_t rows, uint64_t inner, uint64_t cols, uint32_t rhs_max_bits, int accumulate,
                      void* stream);

/* Name of the kernel cpir_op_mat_x_mat runs for this rhs_max_bits (for matching rocprof traces and labelling benchmarks). */
const char* cpir_mat_x_mat_kernel_name(uint32_t rhs_max_bits);

/* Layout of the device-resident packed database ("DtC").  The logical content is exactly
 * Matrix::transpose (matrix.rs:517-527) followed by Matrix::row_wise_compress (matrix.rs:98-205): for every column c of D
 * (= row c of DtC) the N fields  f(c, n) = D[n][c] & (2^b - 1).  Two physical packings of those fields exist:
 *
 *  CPIR_PACK_REFERENCE  the reference's own words:  word(c, w) = sum_{j<cf} f(c, cf*w + j) << (j * 32/cf), cf = 2/3/4
 *                       fields per u32 (matrix.rs:103-167); a chunk is 1024 u32 words of one row (cf*1024 slots).
 *  CPIR_PACK_DENSE64    K = floor(64 / b) fields of exactly b bits per u64 (b = 9: 7 per 8 bytes instead of 6), used when
 *                       it is denser than the reference packing.  A chunk is 1024 u64 words of one row (K*1024 slots);
 *                       inside a chunk field j of u64 word m holds slot  chunk_base + j*1024 + p(m),  where
 *                       m = L*512 + 2t + e  <->  p = 4t + 2L + e  (t < 256, L,e in {0,1}): lane t of a workgroup then reads
 *                       two fully coalesced 16-byte pieces per row and, per field plane j, ONE aligned 16-byte piece of q.
 *
 *  CPIR_PACK_PLANAR     laid out as operands of the i8 matrix cores (v_mfma_i32_16x16x64_i8): a field is split into its low byte and, for
 *                       b >= 9, (b - 8) one-bit planes -- exactly b bits per field; for b <= 8 the byte alone (8 bits per field, what
 *                       the reference packing spends there too).  The unit is a super-tile of
 *                       16 columns x 512 slots = 8 k-blocks of 64 slots: 8 x 1 KiB of low bytes (k-block kb, lane l = 16*g + c holds,
 *                       in byte j of its 16 bytes, (f XOR 0x80) & 0xFF of column c, slot 64*kb + 16*g + j -- the B operand of one
 *                       MFMA as it stands) followed by 1 KiB per bit plane p (lane l holds 4 dwords; dword w, bit 8*jj + 4*s + d is
 *                       bit 8+p of the field at slot 64*(2w+s) + 16*g + 4*d + jj).  Super-tiles are stored column-tile major, then
 *                       along the slots; after them one u32 per padded column holds the wrap-around sum of that column's fields
 *                       (a correction term of the signed-byte arithmetic, see respond_planar.hip).  chunk_words = u32 words of one
 *                       super-tile, slots_per_chunk = 512, words_per_row_padded = u32 words spent per column, fields_per_word = 0.
 *
 * For the first two packings rows are stored row-major with the row stride padded to whole chunks (zero words) and the row count padded to
 * CPIR_DTC_ROW_ALIGN (zero rows), so every 16-byte load is aligned and no kernel has a ragged tail.  The packing is private to
 * the device: cpir_op_dtc_import / _export and cpir_server_from_compressed / _export_compressed convert from / to the
 * reference's C x ceil(N/cf) matrix bit-exactly. */
#define CPIR_PACK_REFERENCE 0u
#define CPIR_PACK_DENSE64 1u
#define CPIR_PACK_PLANAR 2u
#define CPIR_PLANAR_SLOTS_PER_TILE 512u
typedef struct cpir_dtc_layout {
  uint64_t num_slots;            /* N: filter slots = rows of D = decompressed columns of D^T (server.rs:66) */
  uint32_t num_cols;             /* C: columns of D = rows of D^T = response length */
  uint32_t mat_elem_bit_len;     /* b */
  uint32_t compression_factor;   /* cf of the REFERENCE packing (2/3/4), whatever the device packing is */
  uint64_t words_per_row;        /* W = ceil(N / cf): the reference's compressed width (import / export shape) */
  uint64_t words_per_row_padded; /* device row stride in u32 words, multiple of chunk_words */
  uint32_t rows_padded;          /* >= C, multiple of CPIR_DTC_ROW_ALIGN */
  uint64_t total_words;          /* rows_padded * words_per_row_padded (u32 words of device memory) */
  uint32_t packing;              /* CPIR_PACK_REFERENCE, CPIR_PACK_DENSE64 or CPIR_PACK_PLANAR */
  uint32_t fields_per_word;      /* cf (per u32) for the reference packing, K (per u64) for dense64, 0 for planar */
  uint32_t chunk_words;          /* u32 words of one row per chunk: 1024 (reference) or 2048 (dense64); planar: u32 words of one
                                    super-tile = max(b, 8) * 256 */
  uint64_t slots_per_chunk;      /* cf*1024 (reference), K*1024 (dense64), 512 (planar); see cpir_shard_unit */
} cpir_dtc_layout;
#define CPIR_DTC_WORD_ALIGN 1024u
#define CPIR_DTC_ROW_ALIGN 16u

/* Default layout for a database shape: planar (the matrix-core respond) for every bit length.  cpir_tuning_set("layout.planar", 0)
 * switches it off process-wide: then dense64 where it is offered (b in {7, 9, 11, 12}), else the reference packing;
 * ("layout.dense", 0) switches dense64 off as well. */
int cpir_dtc_layout_for(uint64_t num_slots, uint32_t num_cols, uint32_t mat_elem_bit_len, cpir_dtc_layout* out);
/* Granularity of shard boundaries along the filter slots for this layout: lcm(slots_per_chunk, compression_factor).  A multi-GPU
 * partition whose boundaries are multiples of it splits no chunk / super-tile of the device packing, no packed word of the
 * reference's representation (import / export) and no 16-byte query load.  cpir_server_setup_multi and
 * chalametpir_amd.distributed.shard_range both use it.  0 for a NULL / invalid layout. */
uint64_t cpir_shard_unit(const cpir_dtc_layout* layout);
/* Explicit packing choice (CPIR_ERR_INVALID_ARGUMENT if that packing is not offered for b). */
int cpir_dtc_layout_for_packing(uint64_t num_slots, uint32_t num_cols, uint32_t mat_elem_bit_len, uint32_t packing,
                                cpir_dtc_layout* out);

/* gpu_utils::mat_transpose + shaders/mat_transpose.glsl (gpu_utils.rs:222-281) FUSED with the CPU
 * Matrix::row_wise_compress the reference runs after reading the transpose back (server.rs:151-156):
 * D (N x C, leading dim ldd, device) -> packed DtC in `layout->packing` (device, `layout->total_words` u32, fully written
 * incl. padding).
 * If `or_of_entries` (device u32) is non-NULL the bitwise OR of all D entries is OR-ed into it (lets the caller prove
 * the rhs_max_bits bound it passes to cpir_op_mat_x_mat). */
int cpir_op_transpose_compress(cpir_device* dev, const uint32_t* D, uint64_t ldd, const cpir_dtc_layout* layout,
                               uint32_t* dtc, uint32_t* or_of_entries, void* stream);

/* The packed image as the right-hand side of the hint product (server.rs:61 + 64-67 with ONE pass over D): for the planar packing with
 * b >= 9 (cpir_packed_rhs_offered) the image's low-byte operand pieces are also the first operand plane of the matrix-core matmul
 * (csrc/matmul_mfma.hip).  The second plane is the byte (field >> 8) XOR 0x80 of every field:
 *   b = 9  : that byte is the image's ONE bit plane; the matmul expands it in registers and no plane exists (plane bytes 0, hi_plane NULL);
 *   b >= 10: cpir_packed_rhs_plane_bytes(layout) bytes, written by the same pass into `hi_plane` (device, 16-byte aligned).
 * Otherwise as cpir_op_transpose_compress.  CPIR_ERR_INVALID_ARGUMENT where the pairing is not offered or hi_plane does not match. */
int cpir_packed_rhs_offered(const cpir_dtc_layout* layout);
uint64_t cpir_packed_rhs_plane_bytes(const cpir_dtc_layout* layout);
int cpir_op_transpose_compress_with_plane(cpir_device* dev, const uint32_t* D, uint64_t ldd, const cpir_dtc_layout* layout,
                                          uint32_t* dtc, uint32_t* or_of_entries, void* hi_plane, void* stream);
/* M (rows x layout->num_cols, leading dim ldm) (+)= A (rows x layout->num_slots, leading dim lda, 16-byte aligned, lda and num_slots
 * multiples of 4) * D, with D given as its packed image `dtc` and the plane written by cpir_op_transpose_compress_with_plane (NULL where that
 * plane has 0 bytes: b = 9).  Equal to
 * cpir_op_mat_x_mat on the unpacked D whenever every entry of D is below 2^b (the image holds the entries masked to b bits, matrix.rs:121;
 * the OR of the entries from the packing pass proves it).  CPIR_ERR_INVALID_ARGUMENT where the pairing is not offered. */
int cpir_op_mat_x_packed(cpir_device* dev, const uint32_t* A, uint64_t lda, const uint32_t* dtc, const cpir_dtc_layout* layout,
                         const void* hi_plane, uint32_t* M, uint64_t ldm, uint64_t rows, int accumulate, void* stream);

/* Inverse direction for import/export of the reference's own compressed matrix (C x W, row-major, as held in
 * Server.compressed_transposed_parsed_db_mat_d, server.rs:18): pad/normalise into the device layout, and strip. */
int cpir_op_dtc_import(cpir_device* dev, const uint32_t* compressed_rowmajor, const cpir_dtc_layout* layout, uint32_t* dtc,
                       void* stream);
int cpir_op_dtc_export(cpir_device* dev, const uint32_t* dtc, const cpir_dtc_layout* layout, uint32_t* compressed_rowmajor,
                       void* stream);

/* Matrix::row_vector_x_compressed_transposed_matrix (matrix.rs:328-485), the respond hot loop:
 *   r[c] = sum_{n<N} q[n] *wrap field_{n mod cf}(word(c, n / cf)),  c < C.
 * q: N u32 (device), r: C u32 (device), scratch: cpir_respond_scratch_words(layout) u32 (device).
 * q_slot_offset/q_len describe a shard: this DtC holds slots [q_slot_offset, q_slot_offset + layout->num_slots) of a
 * larger database and `q` points at the FULL query (q_len entries); pass 0 / layout->num_slots when unsharded. */
uint64_t cpir_respond_scratch_words(const cpir_dtc_layout* layout);
int cpir_op_respond(cpir_device* dev, const uint32_t* dtc, const cpir_dtc_layout* layout, const uint32_t* q,
                    uint64_t q_len, uint64_t q_slot_offset, uint32_t* r, uint32_t* scratch, void* stream);
/* Same contraction for `batch` queries (q: batch x q_len, r: batch x C, row-major): with "respond.batch_fusion" on (the default) the
 * queries share streams of the database -- on the planar packing up to 24 per pass (six sets of 16 A rows on the i8 matrix cores,
 * 4 queries each: the wide pass; cpir_respond_batch_pass_width says how a batch is cut), any batch size in as few passes as that
 * allows; off: one pass each, all in one launch.  Semantically `batch` independent cpir_op_respond calls, bit for bit. */
uint64_t cpir_respond_batch_scratch_words(const cpir_dtc_layout* layout, uint32_t batch);
int cpir_op_respond_batch(cpir_device* dev, const uint32_t* dtc, const cpir_dtc_layout* layout, const uint32_t* q,
                          uint64_t q_len, uint64_t q_slot_offset, uint32_t batch, uint32_t* r, uint32_t* scratch,
                          void* stream);

/* Counter-based synthetic data (benchmarks / spot-checkable tests): out[i] = hi32(mix(seed, index0+i)) & mask. */
int cpir_op_synth_fill(cpir_device* dev, uint32_t* out, uint64_t count, uint64_t seed, uint64_t index0, uint32_t mask,
                       void* stream);
/* Tuning knobs of the respond kernel (benchmark harness only; defaults are the measured best):
 *   "respond.nontemporal" {0,1}, "respond.blocks_per_cu" 0..8 (0 = occupancy API),
 *   "respond.xcd_split" {0,1}, "respond.batch_fusion" {0,1} (0: every query of a batch call streams the database on its
 *   own, i.e. a batch call is only a cheaper way to enqueue independent responds), "respond.interleave_passes" {-1,0,1}
 *   (order in which one launch walks its passes; -1 = by shard size), "respond.planar_blocks_per_cu" 0..8, "respond.multi_pass_limit_mb"
 *   (unfused batches on databases above this size get one launch per query; VALU packings), "respond.ks_major" 1..3 (planar packing,
 *   device-resident queries: 1 -- the default -- every launch runs on the wide kernel (1 .. 24 queries per pass, one 8-wave block per CU,
 *   passes in slice or interleaved order) and the step-major kernel serves the in-place host path only; 2 the step-major kernel wherever
 *   it applies, i.e. passes of up to 4 queries in slice order -- tests and A/B runs; 3 as 2, launched as the in-place host path launches it),
 *   "respond.helper_spin_us" 0..10000 (a lone pageable caller's query is copied into page-locked memory by this thread and three helper
 *   threads: behind such a query the helpers keep looking for the next one this long -- a caller in a loop comes back within that time and
 *   does not pay for waking them -- before they go to sleep; default 300, 0: sleep at once),
 *   "respond.upload_streams" 1..4 (concurrent host callers: their query uploads take this many HIP streams in turn, so that one copy is
 *   set up while another crosses the link; default 2),
 *   "respond.inplace_seats" 0, 2..4 (a few concurrent host callers are answered by ONE pass that reads every query in place over the host
 *   link instead of uploading them one after the other -- a page-locked query from its caller's own buffer, a pageable one of 2^15+ words
 *   from the server's pinned block while its caller's thread copies it in: up to this many callers per pass, and only while no more than
 *   that were recently seen inside at the same time; default 4, 0: off),
 *   "respond.host_zero_copy" {0,1}
 *   (1, the default: cpir_server_respond serves a caller that finds the server idle without an upload, the kernel reading the
 *   query in place from page-locked host memory; 0: always stage + upload first), "respond.host_fill_timeout_us" 0..1000000
 *   (a lone PAGEABLE query of 2^15+ words: one launch in front of the copy into pinned memory, every wave waiting at most this long
 *   for the words of a step -- default 2000, raised to what copying the whole query takes at 5 GB/s (values below 100 are taken as they
 *   are: tests); 0: two launches, each when its half of the query is in place.  The pageable queries of an in-place round of concurrent
 *   callers, respond.inplace_seats, are polled the same way, seat by seat; 0 sends them through the upload path), "matmul.mfma" {0,1} (1, the default:
 *   cpir_op_mat_x_mat and the setup paths run right-hand sides below 2^16 on the i8 matrix cores; 0: on the integer VALU),
 *   "matmul.pipeline" {0,1} (1, the default: the software-pipelined matrix-core kernel; 0: its first cut),
 *   "pack.rows" {-1,0,1} (the planar pack pass with at most one bit plane: 1 = a block streams whole rows of D, 0 = 64-column waves, -1 --
 *   the default -- by width; with two or more bit planes always the 64-column waves),
 *   "layout.dense" {0,1} and "layout.planar" {0,1}
 *   (default packing chosen by cpir_dtc_layout_for and therefore by every cpir_server_* constructor: planar where it is
 *   offered and enabled, else dense64 where offered and enabled, else the reference packing),
 *   "layout.compact_slots" {0,1,2} (whether a constructor leaves the rows of D without a non-zero field out of the resident image:
 *   0 never, 1 -- the default -- where they are at least 1/32 of the rows, 2 whenever there is one; see cpir_server_slots_served).
 * Process-wide; results are bit-identical for every setting: the library has no key and reads no environment variable that makes it skip
 * work or answer wrongly (the ablation switches of the tuning scripts exist only in a -DCPIR_DIAG build, `make diag`). */
int cpir_tuning_set(const char* key, int value);
/* Every key back to its default (what a test harness calls between tests: the knobs are process-wide state). */
void cpir_tuning_reset(void);
/* Name of the dominant kernel last launched by cpir_op_respond for this layout (for matching rocprof traces). */
const char* cpir_respond_kernel_name(const cpir_dtc_layout* layout);
/* Name of the kernel cpir_op_transpose_compress runs for this layout (under the current "pack.rows" setting), as a kernel trace shows it. */
const char* cpir_pack_kernel_name(const cpir_dtc_layout* layout);
/* Queries per pass a FUSED batch of `batch` queries is cut into on this layout under the current tuning (the queries of a pass share one
 * stream of the database): planar 24 at most -- as few passes as that allows, all of about the same width (4 where "respond.ks_major"
 * sends fused passes to the step-major kernel); other packings 4 (then 2, then 1).  0 for a NULL layout or an empty batch. */
uint32_t cpir_respond_batch_pass_width(const cpir_dtc_layout* layout, uint32_t batch);

/* ------------------------------------------------------------------------------------------------
 * Server handle: the device-resident replacement of `struct Server` (server.rs:15-21).
 * Immutable after setup, ref-counted: Clone = cpir_server_retain, Drop = cpir_server_release.
 * cpir_server_respond* are thread-safe and re-entrant on one handle (the reference's respond(&self) is called
 * concurrently from many tokio tasks on an Arc<Server>: chalametpir_server/examples/server.rs:45,55,85).
 * ------------------------------------------------------------------------------------------------ */
typedef struct cpir_server cpir_server;

/* The matrix half of Server::setup (server.rs:59-67 CPU build, server.rs:117-156 gpu build), from the encoded DB:
 *   A = generate_from_seed(1774, N, seed_mu)   (pass pub_mat_a = NULL), or caller-supplied A (1774 x N, host)
 *   hint = A * D            -> hint_out: 1774 x C u32 (host), i.e. hint_bytes without the 8-byte header
 *   server = compress(transpose(D)) resident in HBM.
 * D: N x C row-major u32 on the host.  mat_elem_bit_len as chosen by cpir_find_encoded_db_matrix_element_bit_length.
 * The XOF expansion of A is pipelined in row blocks with H2D and the device matmul. */
int cpir_server_setup(cpir_device* dev, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN], const uint32_t* pub_mat_a,
                      const uint32_t* D, uint64_t N, uint32_t C, uint32_t mat_elem_bit_len, uint32_t* hint_out,
                      cpir_server** out);

/* Full Server::setup::<ARITY>(seed_mu, db) (server.rs:47-78 / 103-167) from a key-value database handed over as flat
 * arrays (the Rust shim flattens its HashMap<&[u8], &[u8]>): binary-fuse-filter construction, row encoding, hint,
 * resident packed DB.  filter_seed_material: 32*max_attempts bytes of candidate filter seeds (the reference draws them
 * from an OS-seeded ChaCha20, binary_fuse_filter.rs:100-106; NULL = draw from the OS here too).
 * hint_bytes_out / filter_param_bytes_out receive the exact wire images the reference returns
 * (hint: 8 + 4*1774*C bytes; filter params: 68 bytes, binary_fuse_filter.rs:462-486). */
typedef struct cpir_kv_db {
  uint64_t num_pairs;
  const uint8_t* keys;      /* concatenated key bytes */
  const uint64_t* key_off;  /* num_pairs + 1 offsets  */
  const uint8_t* values;    /* concatenated value bytes */
  const uint64_t* val_off;  /* num_pairs + 1 offsets  */
} cpir_kv_db;
#define CPIR_FILTER_PARAM_BYTE_LEN 68u
int cpir_server_setup_kv(cpir_device* dev, uint32_t arity, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN], const cpir_kv_db* db,
                         const uint8_t* filter_seed_material, uint32_t max_attempts, uint8_t* hint_bytes_out,
                         size_t hint_bytes_cap, size_t* hint_bytes_len, uint8_t filter_param_bytes_out[CPIR_FILTER_PARAM_BYTE_LEN],
                         cpir_server** out);
/* The same two constructors on a GROUP of devices of this process (new: the reference has no multi-device code).
 * The database is split along the filter slots over devs[0 .. n_dev) (fewer if it has fewer packing units than devices; a device
 * may be listed more than once); ONE host expansion of A feeds every device's column slab, every device packs its rows of D and
 * multiplies its slab by them, and the per-shard partial hints are summed on the host.  The handle behaves like any other in
 * cpir_server_respond / _respond_bytes / _export_compressed / _retain / _release: a host query is SCATTERED -- device g reads only
 * its slots of q over its own host link (every shard is an ordinary server asked through cpir_server_respond: in place, in rounds with the
 * other callers of the group), answers its shard, and the C-word partial responses are summed on the host (u32 wrap-around,
 * bit-identical to one device).  THE EXCHANGE OF THE HOST ENTRY POINTS IS A HOST SUM, not a collective: 3.7 kB per shard and query.
 * The *_device entry points take a group handle too: q_dev and r_dev then live on the device of shard 0 (the root), every shard's own
 * stream pulls its slots of the queries over the peer link (xGMI between GPUs of one node; peer access is enabled where the hardware
 * offers it), answers them, and pushes its C-word partial responses into a table on the root, where a kernel on the caller's stream --
 * behind one event per shard -- adds them up.  Stream-ordered end to end: nothing waits on the host, no collective library is involved
 * (multi-PROCESS callers use one ordinary shard server per rank, cpir_server_from_device_matrix, and their own collective: RCCL
 * through torch.distributed in chalametpir_amd/distributed.py). */
int cpir_server_setup_multi(cpir_device* const* devs, uint32_t n_dev, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN], const uint32_t* pub_mat_a,
                            const uint32_t* D, uint64_t N, uint32_t C, uint32_t mat_elem_bit_len, uint32_t* hint_out, cpir_server** out);
int cpir_server_setup_kv_multi(cpir_device* const* devs, uint32_t n_dev, uint32_t arity, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN],
                               const cpir_kv_db* db, const uint8_t* filter_seed_material, uint32_t max_attempts, uint8_t* hint_bytes_out,
                               size_t hint_bytes_cap, size_t* hint_bytes_len,
                               uint8_t filter_param_bytes_out[CPIR_FILTER_PARAM_BYTE_LEN], cpir_server** out);
/* Number of shards of a group handle (0 for an ordinary server), and where shard `index` lives. */
int cpir_server_group_size(const cpir_server* srv, uint32_t* shards);
int cpir_server_group_shard(const cpir_server* srv, uint32_t index, int* device_ordinal, uint64_t* slot_offset, uint64_t* num_slots);

/* Host-only first half of Server::setup: Matrix::from_kv_database::<ARITY> (matrix.rs:633-648, binary_fuse_filter.rs:40-456,
 * serialization.rs:22-116).  Writes D (N x C row-major, D_cap_words >= N*C from cpir_setup_kv_shape) and the 68-byte
 * filter parameters.  No device involved. */
int cpir_encode_kv_database(uint32_t arity, const cpir_kv_db* db, uint32_t mat_elem_bit_len, const uint8_t* filter_seed_material,
                            uint32_t max_attempts, uint8_t filter_param_bytes_out[CPIR_FILTER_PARAM_BYTE_LEN], uint32_t* D_out,
                            uint64_t D_cap_words, uint64_t* N, uint32_t* C);
/* Sizes needed before calling cpir_server_setup_kv. */
int cpir_setup_kv_shape(uint32_t arity, const cpir_kv_db* db, uint32_t* mat_elem_bit_len, uint64_t* N, uint32_t* C,
                        size_t* hint_bytes_len);

/* One shard's share of the hint (multi-GPU setup): with the database split along the filter-slot axis,
 *   hint = sum over shards of  A[:, slot_offset : slot_offset + N_shard] * D[slot_offset : slot_offset + N_shard, :].
 * Expands A from seed_mu on the host (the whole sponge has to be squeezed; only this shard's columns are uploaded), or takes
 * those columns from a caller-supplied full A (pub_mat_a, 1774 x total_slots, host), multiplies by the shard of D that already
 * sits on the device (N_shard x C, leading dim ldd) and OVERWRITES M_dev (1774 x C, device) with the partial product.  The caller
 * sum-reduces the partials across shards (u32 wrap-around, e.g. chalametpir_amd.distributed.reduce_u32_).  Synchronous. */
int cpir_hint_partial_device(cpir_device* dev, const uint8_t seed_mu[CPIR_SEED_BYTE_LEN], const uint32_t* pub_mat_a,
                             const uint32_t* D_dev, uint64_t ldd, uint64_t slot_offset, uint64_t N_shard, uint64_t total_slots,
                             uint32_t C, uint32_t rhs_max_bits, uint32_t* M_dev, void* stream);

/* Build a server from matrices that already live on the device (multi-GPU shards, benchmarks):
 * D_dev is N_shard x C (ldd) on `dev`; the shard holds global slots [slot_offset, slot_offset + N_shard) of a
 * database with `total_slots` slots.  Shards are packed independently, so any slot_offset gives correct partial responses (the
 * kernels fall back to guarded scalar query loads when slot_offset is not a multiple of 4); multiples of cpir_shard_unit() -- what
 * chalametpir_amd.distributed.shard_range produces -- keep the 16-byte query loads aligned and let the shards' exported compressed
 * matrices tile the whole one.  CPIR_ERR_SHARD_RANGE: slot_offset + N_shard > total_slots. */
int cpir_server_from_device_matrix(cpir_device* dev, const uint32_t* D_dev, uint64_t ldd, uint64_t N_shard, uint32_t C,
                                   uint32_t mat_elem_bit_len, uint64_t slot_offset, uint64_t total_slots, void* stream,
                                   cpir_server** out);
/* Build a server from the reference's own in-memory representation: compressed transposed matrix (C x ceil(N/cf), host),
 * decompressed_num_cols = N, mat_elem_bit_len (the three fields of struct Server, server.rs:16-21). */
int cpir_server_from_compressed(cpir_device* dev, const uint32_t* compressed, uint32_t C, uint64_t N, uint32_t mat_elem_bit_len,
                                cpir_server** out);
/* Copy the packed DB back in the reference's representation (C x ceil(N/cf) u32, host). */
int cpir_server_export_compressed(const cpir_server* srv, uint32_t* compressed_out, uint64_t out_words);

/* Wall-clock split of the setup call that built this server, in seconds (0 for phases that did not run):
 *   [0] encode (filter construction + row encoding, host)   [1] XOF expansion of A (host thread, overlapped)
 *   [2] D host->device                                       [3] transpose+compress kernel
 *   [4] wait for A after D was ready (un-overlapped XOF/H2D)  [5] hint matmul kernel
 *   [6] hint device->host                                     [7] total wall time of the call */
#define CPIR_SETUP_TIMING_COUNT 8
int cpir_server_setup_timings(const cpir_server* srv, double out[CPIR_SETUP_TIMING_COUNT]);

/* How the host callers of cpir_server_respond have been served by this handle so far (a group handle: summed over its shards, each of
 * which answers -- and counts -- every query):
 *   [0] calls answered   [1] ... alone (no upload: the query read in place over the host link -- from the caller's page-locked buffer,
 *       or from the server's pinned block while it is copied in; [2] is the subset of [1] answered that second way)
 *   [2] of [1]: by one launch polling the copy of a pageable query
 *   [3] polled passes that gave up waiting for a copy and were answered again -- lone launches of [2] AND in-place rounds of [7] with a
 *       pageable seat (so [3] may exceed what [2] alone would allow)
 *   [4] calls answered in uploaded rounds (concurrent callers: staged, uploaded, one fused pass per round)   [5] uploaded rounds
 *   [6] calls answered in in-place rounds (a few concurrent callers: one pass reads their queries over the link, respond.inplace_seats)
 *   [7] in-place rounds */
#define CPIR_HOST_PATH_COUNT 8
int cpir_server_host_path_counts(const cpir_server* srv, uint64_t out[CPIR_HOST_PATH_COUNT]);

cpir_server* cpir_server_retain(cpir_server* srv);  /* #[derive(Clone)] (server.rs:15) */
void cpir_server_release(cpir_server* srv);         /* Drop */
int cpir_server_layout(const cpir_server* srv, cpir_dtc_layout* out);  /* the LOGICAL database: num_slots = the query slots it answers for */
int cpir_server_shard(const cpir_server* srv, uint64_t* slot_offset, uint64_t* total_slots);
const uint32_t* cpir_server_dtc_device_ptr(const cpir_server* srv);     /* the resident image: cpir_server_physical_layout describes it */

/* Serving only the slots that hold something.  A real encoded database (Matrix::from_kv_database, matrix.rs:702-746) has
 * N = num_fingerprints rows of which only n = number of keys are ever written (the slot the filter's peel order assigns to each key,
 * matrix.rs:727-740): N - n rows -- 11 % at arity 3, 7 % at arity 4 -- are all zero and contribute 0 to every response whatever the
 * query holds there.  Setup finds such rows on the device (after masking to b bits, what row_wise_compress keeps, matrix.rs:121) and,
 * where they make up at least 1/32 of the database, packs only the others; every respond entry point then gathers the query onto the
 * kept slots in front of the kernel (device queries: a 4.7 MB gather kernel; a lone host query: while it is copied into the page-locked
 * block the kernel reads).  Responses are bit-identical by construction; export/import still speak the reference's C x ceil(N/cf) matrix.
 * cpir_tuning_set("layout.compact_slots", 0 | 1 | 2): never / when >= 1/32 of the rows are zero (default) / whenever a row is zero.
 *   cpir_server_physical_layout: the layout of the resident image (num_slots = kept slots; equal to cpir_server_layout without a map)
 *   cpir_server_slots_served:    kept slots and the slots they were chosen from (a group handle: summed over its shards)
 *   cpir_server_kept_slots:      the kept slots, increasing, relative to the shard's first slot (CPIR_ERR_INVALID_ARGUMENT without a map) */
int cpir_server_physical_layout(const cpir_server* srv, cpir_dtc_layout* out);
int cpir_server_slots_served(const cpir_server* srv, uint64_t* served, uint64_t* of_slots);
int cpir_server_kept_slots(const cpir_server* srv, uint32_t* out, uint64_t cap);
/* dst[i] = src[idx[i]] on the host, as the lone-caller path compacts a query (AVX-512 / AVX2 gathers where the CPU has them, chosen
 * once per process; CPIR_GATHER=scalar|avx2|avx512-gather in the environment forces a lesser variant): exposed for tests */
const char* cpir_host_gather_variant(void);
int cpir_host_gather_words(uint32_t* dst, const uint32_t* src, const uint32_t* idx, uint64_t count);
/* the streaming form (AVX-512 hosts: "avx512-compress"): the words src[s], s in [s_lo, s_hi), whose bit s is set in `bits` (padded by 8
 * readable bytes), in order; *count = how many were written */
int cpir_host_compress_words(uint32_t* dst, const uint32_t* src, const uint8_t* bits, uint64_t s_lo, uint64_t s_hi, uint64_t* count);

/* Server::respond(&self, query: &[u8]) -> Result<Vec<u8>, _> on wire bytes (server.rs:184-190):
 * from_bytes validation (matrix.rs:973-1010) -> mat-vec -> to_bytes. response_cap >= 8 + 4*C. */
int cpir_server_respond_bytes(const cpir_server* srv, const uint8_t* query, size_t query_len, uint8_t* response,
                              size_t response_cap, size_t* response_len);
/* Same on raw element arrays: q = query[8..] (q_rows x q_cols u32, host), r_out: C u32 (host).
 * q_rows/q_cols are the header fields; anything but 1 x N is rejected as matrix.rs:329-331 does. */
int cpir_server_respond(const cpir_server* srv, const uint32_t* q, uint32_t q_rows, uint64_t q_cols, uint32_t* r_out);
/* Device-resident variant: q_dev (total_slots u32) and r_dev (C u32) on the server's device; enqueues on `stream`
 * (NULL = HIP's default stream) and returns without synchronising.  For a shard, r_dev receives the shard's
 * PARTIAL response; the caller sum-reduces partials across shards (u32 wrap-around add). `scratch_dev` must hold
 * cpir_respond_scratch_words() u32, or NULL to use a per-stream scratch owned by the handle.
 * A GROUP handle: q_dev, r_dev and `stream` belong to the device of shard 0; r_dev receives the COMPLETE response (see
 * cpir_server_setup_multi); scratch_dev is ignored. */
int cpir_server_respond_device(const cpir_server* srv, const uint32_t* q_dev, uint32_t* r_dev, uint32_t* scratch_dev,
                               void* stream);
int cpir_server_respond_batch_device(const cpir_server* srv, const uint32_t* q_dev, uint32_t batch, uint32_t* r_dev,
                                     uint32_t* scratch_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CHALAMET_HIP_H */
