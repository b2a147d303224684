// chalamet_hip.hpp -- the reference's `chalametpir_server::Server` API surface in C++17, header-only, on top of the C ABI of
// chalamet_hip.h (libchalamet_hip.so).
//
// Why this file exists: the reference is Rust and the build image has no Rust toolchain, so rust/server_hip.rs -- the `#[cfg(feature =
// "hip")]` variant of `Server` a maintainer would add -- is source only.  This header is the SAME shim, line for
// line, in a language the image compiles: the same public surface with the same meaning and the same errors
//
//   Server::setup::<ARITY>(&seed, HashMap<&[u8], &[u8]>) -> Result<(Server, Vec<u8>, Vec<u8>), ChalametPIRError>     server.rs:47-78 / 103-167
//   Server::respond(&self, &[u8]) -> Result<Vec<u8>, ChalametPIRError>                                               server.rs:184-190
//   Server: Clone (retain) + Drop (release), shareable between threads (examples/server.rs:45,55,85: Arc<Server>)     server.rs:15
//
// so that the parity test tests/native/test_pir.cpp reads like the reference's own integrations/src/test_pir.rs, and is compiled and run
// (tests/test_gpu_cpp_api.py).  No compute happens here: every call is one or two calls into the C ABI.
#pragma once

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>
#include <utility>
#include <variant>
#include <vector>

#include "chalamet_hip.h"

namespace chalametpir {

constexpr size_t SEED_BYTE_LEN = CPIR_SEED_BYTE_LEN;               // params.rs:5
constexpr uint32_t SERVER_SETUP_MAX_ATTEMPT_COUNT = 100;           // params.rs:8

// ChalametPIRError (chalametpir_common/src/error.rs:7-49): the variants this path can raise, + the Hip* ones that stand where the
// reference's thirteen Vulkan* variants stand (error.rs:10-22).  `#[derive(Debug, PartialEq)]`: comparable, printable.
struct ChalametPIRError {
  enum Kind {
    InvalidMatrixDimension,
    IncompatibleDimensionForMatrixMultiplication,
    InvalidNumberOfElementsInMatrix,
    IncompatibleDimensionForRowVectorTransposedMatrixMultiplication,
    FailedToDeserializeMatrixFromBytes,
    EmptyKVDatabase,
    ExhaustedAllAttemptsToBuild3WiseXorFilter,
    ExhaustedAllAttemptsToBuild4WiseXorFilter,
    KVDatabaseSizeTooLarge,
    UnsupportedArityForBinaryFuseFilter,
    ImpossibleEncodedDBMatrixElementBitLength,
    HipDeviceNotFound,     // ~ VulkanLibraryNotFound / VulkanPhysicalDeviceNotFound
    HipOutOfMemory,        // ~ VulkanBufferCreationFailed
    HipRuntimeCallFailed,  // ~ VulkanCommandBufferExecutionFailed
  } kind;
  size_t attempts = 0;  // the payload of ExhaustedAllAttemptsToBuild{3,4}WiseXorFilter(usize)
  int status = 0;       // the cpir_status it was made from
  bool operator==(const ChalametPIRError& o) const { return kind == o.kind && attempts == o.attempts; }
  bool operator!=(const ChalametPIRError& o) const { return !(*this == o); }
  std::string to_string() const { return std::string(cpir_strerror(status)); }  // Display (error.rs:51-100)
};

// cpir_status -> ChalametPIRError, as rust/server_hip.rs::map_status
inline ChalametPIRError map_status(int status, size_t max_attempts) {
  using K = ChalametPIRError;
  switch (status) {
    case CPIR_ERR_INVALID_MATRIX_DIMENSION: return {K::InvalidMatrixDimension, 0, status};
    case CPIR_ERR_INCOMPATIBLE_DIM_MATMUL: return {K::IncompatibleDimensionForMatrixMultiplication, 0, status};
    case CPIR_ERR_INVALID_NUMBER_OF_ELEMENTS: return {K::InvalidNumberOfElementsInMatrix, 0, status};
    case CPIR_ERR_INCOMPATIBLE_DIM_ROWVEC_X_TRANSPOSED: return {K::IncompatibleDimensionForRowVectorTransposedMatrixMultiplication, 0, status};
    case CPIR_ERR_FAILED_TO_DESERIALIZE_MATRIX: return {K::FailedToDeserializeMatrixFromBytes, 0, status};
    case CPIR_ERR_EMPTY_KV_DATABASE: return {K::EmptyKVDatabase, 0, status};
    case CPIR_ERR_EXHAUSTED_ATTEMPTS_3WISE: return {K::ExhaustedAllAttemptsToBuild3WiseXorFilter, max_attempts, status};
    case CPIR_ERR_EXHAUSTED_ATTEMPTS_4WISE: return {K::ExhaustedAllAttemptsToBuild4WiseXorFilter, max_attempts, status};
    case CPIR_ERR_KV_DATABASE_SIZE_TOO_LARGE: return {K::KVDatabaseSizeTooLarge, 0, status};
    case CPIR_ERR_UNSUPPORTED_ARITY: return {K::UnsupportedArityForBinaryFuseFilter, 0, status};
    case CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH: return {K::ImpossibleEncodedDBMatrixElementBitLength, 0, status};
    case CPIR_ERR_NO_DEVICE: return {K::HipDeviceNotFound, 0, status};
    case CPIR_ERR_OUT_OF_DEVICE_MEMORY: return {K::HipOutOfMemory, 0, status};
    default: return {K::HipRuntimeCallFailed, 0, status};
  }
}

// Result<T, ChalametPIRError>: just enough of it for callers to read like the reference's (`.expect("...")`, `match`)
template <class T>
class Result {
 public:
  Result(T v) : v_(std::move(v)) {}
  Result(ChalametPIRError e) : v_(e) {}
  bool is_ok() const { return v_.index() == 0; }
  bool is_err() const { return !is_ok(); }
  const ChalametPIRError& unwrap_err() const { return std::get<1>(v_); }
  T expect(const char* what) && {
    if (is_err()) {
      std::fprintf(stderr, "%s: %s\n", what, unwrap_err().to_string().c_str());
      std::abort();
    }
    return std::move(std::get<0>(v_));
  }
  T unwrap() && { return std::move(*this).expect("called `Result::unwrap()` on an `Err` value"); }

 private:
  std::variant<T, ChalametPIRError> v_;
};

// &[u8]: a borrowed run of bytes, hashable and comparable by content (the key and value type of the reference's HashMap<&[u8], &[u8]>)
struct Bytes {
  const uint8_t* ptr = nullptr;
  size_t len = 0;
  Bytes() = default;
  Bytes(const uint8_t* p, size_t n) : ptr(p), len(n) {}
  Bytes(const std::vector<uint8_t>& v) : ptr(v.data()), len(v.size()) {}  // NOLINT: &v[..]
  const uint8_t* data() const { return ptr; }
  size_t size() const { return len; }
  bool operator==(const Bytes& o) const { return len == o.len && (len == 0 || std::memcmp(ptr, o.ptr, len) == 0); }
};
struct BytesHash {
  size_t operator()(const Bytes& b) const {  // FNV-1a
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < b.len; i++) h = (h ^ b.ptr[i]) * 1099511628211ull;
    return (size_t)h;
  }
};
inline Bytes bytes_of(const std::vector<uint8_t>& v) { return Bytes(v); }

// Owns one reference on a cpir_server; copying retains (Clone), destruction releases (Drop).  The handle is immutable after setup and
// cpir_server_respond* is thread-safe and re-entrant, so a Server may be shared between threads (the reference's Arc<Server>).
class Server {
 public:
  Server(const Server& o) : handle_(cpir_server_retain(o.handle_)), response_cols_(o.response_cols_) {}
  Server(Server&& o) noexcept : handle_(o.handle_), response_cols_(o.response_cols_) { o.handle_ = nullptr; }
  Server& operator=(Server o) noexcept {
    std::swap(handle_, o.handle_);
    std::swap(response_cols_, o.response_cols_);
    return *this;
  }
  ~Server() {
    if (handle_) cpir_server_release(handle_);
  }

  // Server::setup::<ARITY>(&seed_mu, db) -> (Server, hint_bytes, filter_param_bytes).  `db`: any iterable of (key, value) pairs whose
  // members have data() / size() over bytes -- a std::unordered_map<Bytes, Bytes> is the HashMap<&[u8], &[u8]> of the reference; its
  // iteration order is the key order handed to the encoder, as in the reference (bff.rs:112).
  template <uint32_t ARITY, class KvDb>
  static Result<std::tuple<Server, std::vector<uint8_t>, std::vector<uint8_t>>> setup(const std::array<uint8_t, SEED_BYTE_LEN>& seed_mu, const KvDb& db) {
    static_assert(ARITY == 3 || ARITY == 4, "const { assert!(ARITY == 3 || ARITY == 4) }");  // matrix.rs:638
    using Out = Result<std::tuple<Server, std::vector<uint8_t>, std::vector<uint8_t>>>;
    if (db.empty()) return Out(map_status(CPIR_ERR_EMPTY_KV_DATABASE, 0));  // server.rs:48-51

    // flatten the map into the cpir_kv_db arrays
    std::vector<uint8_t> keys, vals;
    std::vector<uint64_t> key_off{0}, val_off{0};
    for (const auto& kv : db) {
      const auto* k = reinterpret_cast<const uint8_t*>(kv.first.data());
      const auto* v = reinterpret_cast<const uint8_t*>(kv.second.data());
      keys.insert(keys.end(), k, k + kv.first.size());
      vals.insert(vals.end(), v, v + kv.second.size());
      key_off.push_back(keys.size());
      val_off.push_back(vals.size());
    }
    const cpir_kv_db flat{(uint64_t)(key_off.size() - 1), keys.data(), key_off.data(), vals.data(), val_off.data()};

    uint32_t b = 0, c = 0;
    uint64_t n = 0;
    size_t hint_len = 0;
    int st = cpir_setup_kv_shape(ARITY, &flat, &b, &n, &c, &hint_len);
    if (st != CPIR_OK) return Out(map_status(st, SERVER_SETUP_MAX_ATTEMPT_COUNT));

    // replaces gpu_utils::setup_gpu() (gpu_utils.rs:25).  CHALAMET_HIP_DEVICES=0,1,2,3 splits the database over several GPUs of this
    // process behind the one handle (cpir_server_setup_kv_multi); default: device 0.
    std::vector<int> ordinals;
    if (const char* env = std::getenv("CHALAMET_HIP_DEVICES")) {
      for (const char* p = env; *p;) {
        char* end = nullptr;
        const long o = std::strtol(p, &end, 10);
        if (end == p) break;
        ordinals.push_back((int)o);
        p = (*end == ',') ? end + 1 : end;
      }
    }
    if (ordinals.empty()) ordinals.push_back(0);
    std::vector<cpir_device*> devs;
    auto close_all = [&] {
      for (cpir_device* d : devs) cpir_device_close(d);
    };
    for (int o : ordinals) {
      cpir_device* dev = nullptr;
      st = cpir_device_open(o, &dev);
      if (st != CPIR_OK) {
        close_all();
        return Out(map_status(st, SERVER_SETUP_MAX_ATTEMPT_COUNT));
      }
      devs.push_back(dev);
    }

    std::vector<uint32_t> hint_words((hint_len + 3) / 4);  // 4-byte aligned backing store for the wire image
    std::vector<uint8_t> filter_param_bytes(CPIR_FILTER_PARAM_BYTE_LEN);
    cpir_server* handle = nullptr;
    size_t written = 0;
    if (devs.size() == 1)
      st = cpir_server_setup_kv(devs[0], ARITY, seed_mu.data(), &flat, nullptr, SERVER_SETUP_MAX_ATTEMPT_COUNT, reinterpret_cast<uint8_t*>(hint_words.data()),
                                hint_len, &written, filter_param_bytes.data(), &handle);
    else
      st = cpir_server_setup_kv_multi(devs.data(), (uint32_t)devs.size(), ARITY, seed_mu.data(), &flat, nullptr, SERVER_SETUP_MAX_ATTEMPT_COUNT,
                                      reinterpret_cast<uint8_t*>(hint_words.data()), hint_len, &written, filter_param_bytes.data(), &handle);
    close_all();  // the server keeps its own references on the devices
    if (st != CPIR_OK) return Out(map_status(st, SERVER_SETUP_MAX_ATTEMPT_COUNT));
    const auto* hb = reinterpret_cast<const uint8_t*>(hint_words.data());
    return Out(std::make_tuple(Server(handle, c), std::vector<uint8_t>(hb, hb + written), std::move(filter_param_bytes)));
  }

  // Server::respond(&self, query: &[u8]) -> Result<Vec<u8>, ChalametPIRError>: wire bytes in (Matrix::to_bytes of the 1 x N query), wire
  // bytes out; a malformed query is FailedToDeserializeMatrixFromBytes, a wrong length
  // IncompatibleDimensionForRowVectorTransposedMatrixMultiplication (matrix.rs:973-1010, 329-331)
  Result<std::vector<uint8_t>> respond(Bytes query) const {
    std::vector<uint8_t> response(8 + 4 * (size_t)response_cols_);
    size_t len = 0;
    const int st = cpir_server_respond_bytes(handle_, query.data(), query.size(), response.data(), response.size(), &len);
    if (st != CPIR_OK) return Result<std::vector<uint8_t>>(map_status(st, 0));
    response.resize(len);
    return Result<std::vector<uint8_t>>(std::move(response));
  }
  Result<std::vector<uint8_t>> respond(const std::vector<uint8_t>& query) const { return respond(bytes_of(query)); }

  const cpir_server* handle() const { return handle_; }

 private:
  Server(cpir_server* h, uint32_t c) : handle_(h), response_cols_(c) {}
  cpir_server* handle_;
  uint32_t response_cols_;
};

// A query buffer in page-locked host memory (cpir_host_alloc): bytes read from the network straight into it are read by the respond
// kernel IN PLACE, without the staging copy a pageable buffer goes through (rust/server_hip.rs::PinnedQuery).  The wire image starts 8
// bytes into the allocation so that the u32 words behind the 8-byte header (matrix.rs:947-971) are 16-byte aligned.
class PinnedQuery {
 public:
  explicit PinnedQuery(size_t wire_len) : len_(wire_len) {
    if (cpir_host_alloc(wire_len + 8, &base_) != CPIR_OK) base_ = nullptr;
  }
  PinnedQuery(const PinnedQuery&) = delete;
  PinnedQuery& operator=(const PinnedQuery&) = delete;
  ~PinnedQuery() {
    if (base_) cpir_host_free(base_);
  }
  bool ok() const { return base_ != nullptr; }
  uint8_t* data() { return static_cast<uint8_t*>(base_) + 8; }
  size_t size() const { return len_; }
  Bytes as_slice() const { return Bytes(static_cast<const uint8_t*>(base_) + 8, len_); }

 private:
  void* base_ = nullptr;
  size_t len_;
};

}  // namespace chalametpir
