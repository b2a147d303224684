// Keyword PIR end to end through the C++ mirror of the reference's Server API (include/chalamet_hip.hpp) -- written to read like the
// reference's own integration test, integrations/src/test_pir.rs:12-142: random key-value databases of 2^8 .. 2^16 pairs (keys 16-32 bytes,
// values 1-512 bytes: chalametpir_common/src/utils.rs:22-45), Server::setup::<ARITY>, Client::setup, ten keys queried per database,
// the value recovered must be the stored one; ArithmeticOverflowAddingQueryIndicator from the client means "ask again" (test_pir.rs:66-70).
//
// The SERVER is the product: chalametpir::Server on libchalamet_hip.so (an MI355X does the setup's hint product and every respond).
// The CLIENT is test infrastructure: the oracle's restatement of chalametpir_client (oracle/chalamet_oracle.h: or_client_query,
// or_client_process_response), wrapped here in a small Client class with the reference's method names.
//
//   test_pir [iterations [max log2 pairs]]      -> "test_keyword_pir ... ok" lines, exit code 0
#include <algorithm>
#include <cstdio>
#include <map>
#include <random>
#include <thread>
#include <unordered_map>

#include "chalamet_hip.hpp"
extern "C" {
#include "chalamet_oracle.h"
}

using chalametpir::Bytes;
using chalametpir::BytesHash;
using chalametpir::ChalametPIRError;
using chalametpir::Server;

namespace {

// chalametpir_client::Client (client.rs:39-282), restated on the oracle -- the checker, not the product
class Client {
 public:
  static Client setup(const std::array<uint8_t, 32>& seed_mu, const std::vector<uint8_t>& hint_bytes, const std::vector<uint8_t>& filter_param_bytes, std::mt19937_64* rng) {
    Client c;
    c.rng_ = rng;
    if (or_bff_from_bytes(filter_param_bytes.data(), filter_param_bytes.size(), &c.filter_) != OR_OK) std::abort();
    uint32_t rows, cols;
    std::memcpy(&rows, hint_bytes.data(), 4), std::memcpy(&cols, hint_bytes.data() + 4, 4);
    if (rows != OR_LWE_DIMENSION || hint_bytes.size() != 8 + 4 * (size_t)rows * cols) std::abort();  // InvalidHintMatrix, client.rs:48-52
    c.C_ = cols, c.N_ = c.filter_.num_fingerprints;
    c.hint_.resize((size_t)rows * cols);
    std::memcpy(c.hint_.data(), hint_bytes.data() + 8, c.hint_.size() * 4);
    c.A_.resize((size_t)OR_LWE_DIMENSION * c.N_);
    if (or_generate_from_seed(OR_LWE_DIMENSION, c.N_, seed_mu.data(), c.A_.data()) != OR_OK) std::abort();
    return c;
  }
  // client.rs:95-194: Ok(query_bytes) or Err(status) -- OR_ERR_ARITHMETIC_OVERFLOW_ADDING_QUERY_INDICATOR asks the caller to try again
  int query(Bytes key, std::vector<uint8_t>* query_bytes) {
    std::vector<uint32_t> s(OR_LWE_DIMENSION), e(N_), b(N_), c(C_);
    auto ternary = [&](std::vector<uint32_t>& v) {
      for (uint32_t& x : v)
        while (!or_ternary_from_u32((uint32_t)(*rng_)(), &x)) {
        }
    };
    ternary(s), ternary(e);
    const int st = or_client_query(A_.data(), hint_.data(), N_, C_, &filter_, key.data(), key.size(), s.data(), e.data(), b.data(), c.data());
    if (st != OR_OK) return st;
    pending_[std::string(reinterpret_cast<const char*>(key.data()), key.size())] = c;
    query_bytes->resize(8 + 4 * N_);
    const uint32_t one = 1, n32 = (uint32_t)N_;
    std::memcpy(query_bytes->data(), &one, 4), std::memcpy(query_bytes->data() + 4, &n32, 4), std::memcpy(query_bytes->data() + 8, b.data(), 4 * N_);
    return OR_OK;
  }
  // client.rs:209-275
  int process_response(Bytes key, const std::vector<uint8_t>& response_bytes, std::vector<uint8_t>* value) {
    const auto it = pending_.find(std::string(reinterpret_cast<const char*>(key.data()), key.size()));
    if (it == pending_.end() || response_bytes.size() != 8 + 4 * C_) return OR_ERR_INVALID_RESPONSE_VECTOR;
    std::vector<uint32_t> r(C_);
    std::memcpy(r.data(), response_bytes.data() + 8, 4 * C_);
    value->assign(4 * C_, 0);
    size_t len = 0;
    const int st = or_client_process_response(&filter_, key.data(), key.size(), it->second.data(), r.data(), C_, value->data(), value->size(), &len);
    pending_.erase(it);
    if (st == OR_OK) value->resize(len);
    return st;
  }

 private:
  or_bff filter_{};
  uint64_t N_ = 0, C_ = 0;
  std::vector<uint32_t> A_, hint_;
  std::map<std::string, std::vector<uint32_t>> pending_;
  std::mt19937_64* rng_ = nullptr;
};

// chalametpir_common::utils::generate_random_kv_database (utils.rs:22-45)
std::unordered_map<std::string, std::vector<uint8_t>> generate_random_kv_database(size_t num_kv_pairs, std::mt19937_64& rng) {
  std::unordered_map<std::string, std::vector<uint8_t>> kv;
  kv.reserve(num_kv_pairs);
  while (kv.size() < num_kv_pairs) {
    std::string key(16 + rng() % 17, '\0');
    std::vector<uint8_t> value(1 + rng() % 512);
    for (char& ch : key) ch = (char)rng();
    for (uint8_t& x : value) x = (uint8_t)rng();
    kv.emplace(std::move(key), std::move(value));
  }
  return kv;
}

template <uint32_t ARITY>
void test_keyword_pir(size_t iterations, unsigned max_lg, uint64_t seed) {
  constexpr size_t MIN_NUM_KV_PAIRS = (size_t)1 << 8;
  const size_t MAX_NUM_KV_PAIRS = (size_t)1 << max_lg;
  constexpr size_t NUMBER_OF_PIR_QUERIES = 10;
  std::mt19937_64 rng(seed);

  for (size_t test_iter = 0; test_iter < iterations; test_iter++) {
    const size_t num_kv_pairs_in_db = MIN_NUM_KV_PAIRS + rng() % (MAX_NUM_KV_PAIRS - MIN_NUM_KV_PAIRS + 1);
    const auto kv_db = generate_random_kv_database(num_kv_pairs_in_db, rng);
    std::unordered_map<Bytes, Bytes, BytesHash> kv_db_as_ref;  // HashMap<&[u8], &[u8]>
    for (const auto& kv : kv_db) kv_db_as_ref.emplace(Bytes(reinterpret_cast<const uint8_t*>(kv.first.data()), kv.first.size()), Bytes(kv.second));

    std::array<uint8_t, 32> seed_mu;
    for (uint8_t& x : seed_mu) x = (uint8_t)rng();

    auto [server, hint_bytes, filter_param_bytes] = Server::setup<ARITY>(seed_mu, kv_db_as_ref).expect("Server setup failed");
    Client client = Client::setup(seed_mu, hint_bytes, filter_param_bytes, &rng);
    const Server shared = server;  // #[derive(Clone)]: a second owner answers as well as the first

    std::vector<Bytes> all_keys;
    for (const auto& kv : kv_db_as_ref) all_keys.push_back(kv.first);
    std::shuffle(all_keys.begin(), all_keys.end(), rng);
    all_keys.resize(std::min(all_keys.size(), NUMBER_OF_PIR_QUERIES));

    for (size_t i = 0; i < all_keys.size();) {
      const Bytes key = all_keys[i], value = kv_db_as_ref.at(key);
      std::vector<uint8_t> query_bytes, received_value;
      const int q = client.query(key, &query_bytes);
      if (q != OR_OK) {
        if (q != OR_ERR_ARITHMETIC_OVERFLOW_ADDING_QUERY_INDICATOR) std::abort();  // assert_eq!(e, ChalametPIRError::ArithmeticOverflowAddingQueryIndicator)
        continue;                                                                 // is_current_key_processed = false
      }
      const auto response_bytes = (i & 1 ? shared : server).respond(query_bytes).expect("Server can't respond");
      if (client.process_response(key, response_bytes, &received_value) != OR_OK) {
        std::fprintf(stderr, "Client can't extract value from response\n");
        std::abort();
      }
      if (!(Bytes(received_value) == value)) {  // assert_eq!(value, received_value)
        std::fprintf(stderr, "arity %u, %zu pairs: value mismatch for key %zu\n", ARITY, num_kv_pairs_in_db, i);
        std::abort();
      }
      i++;
    }

    // one Arc<Server> answered from many tasks at once (chalametpir_server/examples/server.rs:45-93): the queries of four keys prepared one
    // after the other, answered by four threads side by side -- each through a clone of its own --, decoded one after the other
    {
      std::vector<Bytes> keys4(all_keys.begin(), all_keys.begin() + std::min<size_t>(4, all_keys.size()));
      std::vector<std::vector<uint8_t>> queries(keys4.size()), responses(keys4.size());
      for (size_t i = 0; i < keys4.size();)
        if (client.query(keys4[i], &queries[i]) == OR_OK) i++;
      std::vector<std::thread> tasks;
      for (size_t i = 0; i < keys4.size(); i++)
        tasks.emplace_back([&, i, mine = server] { responses[i] = mine.respond(queries[i]).expect("Server can't respond"); });
      for (std::thread& t : tasks) t.join();
      for (size_t i = 0; i < keys4.size(); i++) {
        std::vector<uint8_t> received_value;
        if (client.process_response(keys4[i], responses[i], &received_value) != OR_OK || !(Bytes(received_value) == kv_db_as_ref.at(keys4[i]))) {
          std::fprintf(stderr, "arity %u: concurrent task %zu got a wrong value\n", ARITY, i);
          std::abort();
        }
      }
    }

    // what the reference's Server::respond rejects, this one rejects the same way (matrix.rs:973-1010, 329-331)
    std::vector<uint8_t> bad(8, 0);
    if (!(server.respond(bad).unwrap_err() == chalametpir::map_status(CPIR_ERR_FAILED_TO_DESERIALIZE_MATRIX, 0))) std::abort();
    std::vector<uint8_t> short_query(8 + 4 * 7, 0);
    const uint32_t one = 1, seven = 7;
    std::memcpy(short_query.data(), &one, 4), std::memcpy(short_query.data() + 4, &seven, 4);
    if (server.respond(short_query).unwrap_err().kind != ChalametPIRError::IncompatibleDimensionForRowVectorTransposedMatrixMultiplication) std::abort();
    std::printf("test_keyword_pir_with_%u_wise_xor_filter: iteration %zu, %zu pairs, %zu values recovered ... ok\n", ARITY, test_iter, num_kv_pairs_in_db, all_keys.size());
  }
}

}  // namespace

int main(int argc, char** argv) {
  const size_t iterations = argc > 1 ? std::strtoul(argv[1], nullptr, 0) : 10;  // NUM_TEST_ITERATIONS
  const unsigned max_lg = argc > 2 ? (unsigned)std::strtoul(argv[2], nullptr, 0) : 16;
  // Server::setup of an empty database is EmptyKVDatabase before any device is asked for (server.rs:48-51)
  const std::unordered_map<Bytes, Bytes, BytesHash> empty;
  if (!(Server::setup<3>(std::array<uint8_t, 32>{}, empty).unwrap_err() == chalametpir::map_status(CPIR_ERR_EMPTY_KV_DATABASE, 0))) return 2;
  test_keyword_pir<3>(iterations, max_lg, 0x3333);
  test_keyword_pir<4>(iterations, max_lg, 0x4444);
  std::puts("test_pir ok");
  return 0;
}
