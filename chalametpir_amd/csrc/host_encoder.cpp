// host_encoder.cpp -- key-value database -> encoded DB matrix D (the part of Server::setup in front of the matrix work).
//
// Reference behaviour restated (paths relative to the reference checkout):
//   Matrix::from_kv_database::<ARITY>            chalametpir_common/src/matrix.rs:633-648, 687-755 (3-wise), 819-894 (4-wise)
//   BinaryFuseFilter::construct_{3,4}_wise_...   chalametpir_common/src/binary_fuse_filter.rs:40-235, 249-456
//   hash_of_key / mix256 / mix / murmur64        binary_fuse_filter.rs:553-601
//   hash_batch_for_{3,4}_wise_xor_filter         binary_fuse_filter.rs:605-635
//   encode_kv_as_row                             chalametpir_common/src/serialization.rs:22-116
//   BinaryFuseFilter::to_bytes (68 bytes)        binary_fuse_filter.rs:462-486
//
// This is sequential pointer-chasing work (peeling a hypergraph, then filling D in reverse peel order, every row
// depending on rows written before it), so it stays on the host; only the per-key TurboSHAKE128 digests, which are
// independent, are spread over host threads.  The reference draws the filter seed from an OS-seeded ChaCha20 and
// walks a HashMap in unspecified order (binary_fuse_filter.rs:100-112), so D is not reproducible there; here the key
// order is the caller's array order and the candidate seeds can be injected, which makes D a pure function of its
// inputs (needed for parity tests) without changing the construction.
#include <algorithm>
#include <functional>
#include <cmath>
#include <random>
#include <thread>

#include "cpir_internal.hpp"

namespace cpir {
namespace {

inline uint64_t load_le64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);  // little-endian host
  return v;
}

inline uint64_t murmur64(uint64_t h) {  // binary_fuse_filter.rs:553-560
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return h;
}
inline uint64_t mix(uint64_t key, uint64_t seed) { return murmur64(key + seed); }  // :563-565

struct Digest {
  uint64_t w[4];
};

uint64_t mix256(const Digest& d, const uint64_t seed_words[4]) {  // :588-601
  uint64_t sum = 0;
  for (int k = 0; k < 4; k++) {
    uint64_t acc = 0;
    for (int s = 0; s < 4; s++) acc = murmur64(acc + mix(d.w[k], seed_words[s]));
    sum += acc;
  }
  return sum;
}

struct Slots {
  uint32_t h[4];
};

inline Slots slots_of(uint64_t hash, uint32_t arity, uint32_t seg_len, uint32_t seg_count_len) {  // :605-635
  Slots s;
  const uint32_t m = seg_len - 1;
  const uint32_t base = (uint32_t)(((unsigned __int128)hash * seg_count_len) >> 64);
  s.h[0] = base;
  s.h[1] = base + seg_len;
  s.h[2] = base + 2 * seg_len;
  s.h[3] = base + 3 * seg_len;
  if (arity == 3) {
    s.h[1] ^= (uint32_t)(hash >> 18) & m;
    s.h[2] ^= (uint32_t)hash & m;
  } else {
    s.h[1] ^= (uint32_t)hash & m;
    s.h[2] ^= (uint32_t)(hash >> 16) & m;
    s.h[3] ^= (uint32_t)(hash >> 32) & m;
  }
  return s;
}

// hash -> key index; linear probing, power-of-two capacity.  Stands in for the reference's HashMap<u64, &[u8]>.
class HashIndex {
 public:
  explicit HashIndex(uint64_t n) {
    uint64_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    mask_ = cap - 1;
    keys_.assign(cap, 0);
    vals_.assign(cap, kEmpty);
  }
  void clear() { std::fill(vals_.begin(), vals_.end(), kEmpty); }
  void put(uint64_t hash, uint64_t idx) {
    uint64_t i = murmur64(hash) & mask_;
    while (vals_[i] != kEmpty && keys_[i] != hash) i = (i + 1) & mask_;
    keys_[i] = hash, vals_[i] = idx;  // a later equal hash replaces the earlier one, as HashMap::insert does
  }
  uint64_t get(uint64_t hash) const {
    uint64_t i = murmur64(hash) & mask_;
    while (vals_[i] != kEmpty) {
      if (keys_[i] == hash) return vals_[i];
      i = (i + 1) & mask_;
    }
    return kEmpty;
  }
  static constexpr uint64_t kEmpty = ~0ull;

 private:
  uint64_t mask_;
  std::vector<uint64_t> keys_, vals_;
};

// serialization.rs:22-116 with the key digest already computed: bit-pack digest || value || 0x81 into b-bit elements.
void pack_row(const uint8_t digest[32], const uint8_t* value, size_t value_len, uint32_t b, uint32_t* row, uint64_t cols) {
  std::fill(row, row + cols, 0u);
  const uint64_t elem_mask = (1ull << b) - 1;
  uint64_t buf = 0;
  unsigned bits = 0;
  uint64_t at = 0;
  auto feed = [&](const uint8_t* src, size_t len) {
    size_t off = 0;
    while (off < len) {
      size_t take = ((64 - bits) & ~7u) / 8;  // whole bytes that still fit the 64-bit window
      if (take > len - off) take = len - off;
      uint64_t word = 0;
      for (size_t i = 0; i < take; i++) word |= (uint64_t)src[off + i] << (8 * i);
      off += take;
      buf |= word << bits;
      bits += 8 * (unsigned)take;
      for (unsigned e = bits / b; e > 0; e--) {
        row[at++] = (uint32_t)(buf & elem_mask);
        buf >>= b;
        bits -= b;
      }
    }
  };
  feed(digest, 32);
  feed(value, value_len);
  buf |= (uint64_t)0x81 << bits;  // boundary mark, serialization.rs:100-102
  bits += 8;
  while (bits > 0) {
    const unsigned take = bits < b ? bits : b;
    row[at++] = (uint32_t)(buf & elem_mask);
    buf >>= take;
    bits -= take;
  }
}

// worker threads of the encoder: the CPUs this process may use minus one -- during Server::setup another host thread is squeezing
// the sponge for A, which is the critical path (capi.hip), and must not be throttled by an oversubscribed CPU quota
unsigned encoder_threads() {
  const unsigned cpus = usable_cpus();
  const unsigned nt = cpus > 1 ? cpus - 1 : 1;
  return nt < 32 ? nt : 32;
}

void parallel_for(uint64_t n, const std::function<void(uint64_t, uint64_t)>& body) {
  unsigned nt = encoder_threads();
  if (n < 4096) nt = 1;
  if (nt <= 1) return body(0, n);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nt; t++) pool.emplace_back([&, t] { body(n * t / nt, n * (t + 1) / nt); });
  for (auto& th : pool) th.join();
}

}  // namespace

void Filter::to_bytes(uint8_t out[CPIR_FILTER_PARAM_BYTE_LEN]) const {
  // binary_fuse_filter.rs:462-486: seed | arity u32 | segment_length u32 | segment_count_length u32 | three usize (u64) fields
  memcpy(out, seed, 32);
  memcpy(out + 32, &arity, 4);
  memcpy(out + 36, &segment_length, 4);
  memcpy(out + 40, &segment_count_length, 4);
  memcpy(out + 44, &num_fingerprints, 8);
  memcpy(out + 52, &filter_size, 8);
  memcpy(out + 60, &mat_elem_bit_len, 8);
}

int encode_kv_database(uint32_t arity, const cpir_kv_db& db, uint32_t b, const uint8_t* filter_seeds, uint32_t max_attempts,
                       Filter* filter, std::vector<uint32_t>* D, uint64_t* N_out, uint32_t* C_out) {
  if (arity != 3 && arity != 4) return CPIR_ERR_UNSUPPORTED_ARITY;
  const uint64_t n = db.num_pairs;
  if (n == 0) return CPIR_ERR_EMPTY_KV_DATABASE;  // binary_fuse_filter.rs:47-50
  if (!db.keys || !db.key_off || !db.values || !db.val_off) return CPIR_ERR_INVALID_ARGUMENT;
  if (compression_factor(b) == 0) return CPIR_ERR_IMPOSSIBLE_ELEMENT_BIT_LENGTH;

  uint32_t seg_len, seg_count_len;
  uint64_t num_fp;
  CPIR_TRY(filter_shape(arity, n, &seg_len, &seg_count_len, &num_fp));
  const uint32_t segments = seg_count_len / seg_len;

  // independent per-key digests (binary_fuse_filter.rs:113 recomputes them per attempt; they do not depend on it)
  std::vector<Digest> digests(n);
  std::vector<uint8_t> digest_bytes(32 * n);
  parallel_for(n, [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      uint8_t* d = &digest_bytes[32 * i];
      turboshake128(db.keys + db.key_off[i], (size_t)(db.key_off[i + 1] - db.key_off[i]), d, 32);
      for (int k = 0; k < 4; k++) digests[i].w[k] = load_le64(d + 8 * k);
    }
  });

  unsigned block_bits = 1;  // binary_fuse_filter.rs:78-84
  while ((1u << block_bits) < segments) block_bits++;
  const uint64_t n_blocks = 1ull << block_bits;

  std::vector<uint64_t> order(n + 1), bucket_pos(n_blocks), xor_hash(num_fp);
  std::vector<uint8_t> count(num_fp), found_slot(n);
  std::vector<uint32_t> stack(num_fp);
  HashIndex index(n);
  std::random_device os_entropy;

  uint8_t seed[32];
  bool built = false;
  for (uint32_t attempt = 0; attempt < max_attempts && !built; attempt++) {
    if (filter_seeds) {
      memcpy(seed, filter_seeds + 32 * (size_t)attempt, 32);
    } else {
      for (int i = 0; i < 8; i++) {
        const uint32_t r = os_entropy();
        memcpy(seed + 4 * i, &r, 4);
      }
    }
    uint64_t seed_words[4];
    for (int i = 0; i < 4; i++) seed_words[i] = load_le64(seed + 8 * i);

    std::fill(order.begin(), order.end(), 0);
    order[n] = 1;  // sentinel so the probing below stops at the end (:74)
    std::fill(count.begin(), count.end(), 0);
    std::fill(xor_hash.begin(), xor_hash.end(), 0);
    index.clear();
    for (uint64_t i = 0; i < n_blocks; i++) bucket_pos[i] = (uint64_t)(((unsigned __int128)i * n) >> block_bits);  // :108-110

    // bucket the keys by the top bits of their hash (:112-126)
    for (uint64_t i = 0; i < n; i++) {
      const uint64_t hash = mix256(digests[i], seed_words);
      uint64_t blk = hash >> (64 - block_bits);
      while (order[bucket_pos[blk]] != 0) blk = (blk + 1) & (n_blocks - 1);
      order[bucket_pos[blk]++] = hash;
      index.put(hash, i);
    }

    // degree counting; the low two bits of `count` accumulate the XOR of the slot positions (:128-145, :337-360)
    bool bad = false;
    uint8_t seen_bits = 0;
    for (uint64_t i = 0; i < n; i++) {
      const uint64_t hash = order[i];
      const Slots s = slots_of(hash, arity, seg_len, seg_count_len);
      for (uint32_t j = 0; j < arity; j++) {
        count[s.h[j]] = (uint8_t)((count[s.h[j]] + 4) ^ j);
        xor_hash[s.h[j]] ^= hash;
        seen_bits |= count[s.h[j]];
      }
      if (arity == 3) bad = count[s.h[0]] < 4 || count[s.h[1]] < 4 || count[s.h[2]] < 4;  // last key decides, as in :144
    }
    if (arity == 4) bad = seen_bits >= 0x80;  // :362
    if (bad) continue;

    // peel: repeatedly take a slot of degree 1 (:155-203, :370-424)
    uint64_t top = 0;
    for (uint64_t i = 0; i < num_fp; i++) {
      stack[top] = (uint32_t)i;
      if ((count[i] >> 2) == 1) top++;
    }
    uint64_t peeled = 0;
    while (top > 0) {
      const uint32_t slot = stack[--top];
      if ((count[slot] >> 2) != 1) continue;
      const uint64_t hash = xor_hash[slot];
      const uint8_t which = count[slot] & 3;
      found_slot[peeled] = which;
      order[peeled] = hash;
      peeled++;
      const Slots s = slots_of(hash, arity, seg_len, seg_count_len);
      for (uint32_t step = 1; step < arity; step++) {
        const uint32_t pos = (which + step) % arity;
        const uint32_t other = s.h[pos];
        stack[top] = other;
        if ((count[other] >> 2) == 2) top++;
        count[other] = (uint8_t)((count[other] - 4) ^ pos);
        xor_hash[other] ^= hash;
      }
    }
    built = (peeled == n);  // :205-210
  }
  if (!built) return arity == 3 ? CPIR_ERR_EXHAUSTED_ATTEMPTS_3WISE : CPIR_ERR_EXHAUSTED_ATTEMPTS_4WISE;

  uint64_t max_value_len = 0;  // matrix.rs:696
  for (uint64_t i = 0; i < n; i++) max_value_len = std::max<uint64_t>(max_value_len, db.val_off[i + 1] - db.val_off[i]);
  const uint64_t cols = encoded_num_cols(max_value_len, b);
  if (cols > 0xffffffffull) return CPIR_ERR_INVALID_MATRIX_DIMENSION;

  D->assign(num_fp * cols, 0u);  // Matrix::new zero-fills: unused slots stay zero (matrix.rs:702)
  const uint32_t mask = (1u << b) - 1u;
  uint32_t* mat = D->data();

  // The reference fills D in one sequential pass in reverse peel order (matrix.rs:707-746 / :839-885):
  //     D[slot_i] = (row_i - D[o1] - D[o2] (- D[o3]) - mix(hash_i, e)) & mask.
  // Only the subtraction of the OTHER slots' rows carries a dependency, and it is per column.  So it is split in two:
  //
  //  pass 1 (parallel over keys): D[slot_i] = (row_i - mix(hash_i, e)) & mask.  Every key owns a distinct slot, the bit
  //          packing of the 1 kB values and the 940 murmur mixes per key are the expensive part and are independent.
  //  pass 2 (parallel over COLUMN blocks, sequential in reverse peel order inside a block):
  //          D[slot_i][e] = (D[slot_i][e] - D[o1][e] - D[o2][e] (- D[o3][e])) & mask.
  //          When key i is processed every other slot it touches is either unowned (zero) or owned by a key peeled
  //          later, i.e. already final: a key peeled EARLIER owned a slot of degree 1 at that time, which key i (still
  //          present then) cannot touch.  Columns never interact, so the blocks need no synchronisation.
  //  Same values mod 2^b as the reference's single pass.
  struct Placement {
    uint32_t dst, o1, o2, o3;
  };
  std::vector<Placement> place(n);
  parallel_for(n, [&](uint64_t lo, uint64_t hi) {
    std::vector<uint32_t> row(cols);
    for (uint64_t i = lo; i < hi; i++) {
      const uint64_t hash = order[i];
      const uint64_t ki = index.get(hash);
      const Slots s = slots_of(hash, arity, seg_len, seg_count_len);
      const uint32_t which = found_slot[i];
      place[i] = {s.h[which], s.h[(which + 1) % arity], s.h[(which + 2) % arity], arity == 4 ? s.h[(which + 3) % arity] : 0u};
      pack_row(&digest_bytes[32 * ki], db.values + db.val_off[ki], (size_t)(db.val_off[ki + 1] - db.val_off[ki]), b, row.data(), cols);
      uint32_t* dst = mat + (uint64_t)s.h[which] * cols;
      for (uint64_t e = 0; e < cols; e++) dst[e] = (row[e] - (uint32_t)mix(hash, e)) & mask;
    }
  });
  {
    uint64_t blocks = encoder_threads();
    if (n < 4096) blocks = 1;
    if (blocks > cols) blocks = cols;
    auto column_block = [&](uint64_t c0, uint64_t c1) {
      for (uint64_t i = n; i-- > 0;) {  // reverse peel order, matrix.rs:707 / :839
        const Placement& p = place[i];
        uint32_t* dst = mat + (uint64_t)p.dst * cols;
        const uint32_t* o1 = mat + (uint64_t)p.o1 * cols;
        const uint32_t* o2 = mat + (uint64_t)p.o2 * cols;
        if (arity == 4) {
          const uint32_t* o3 = mat + (uint64_t)p.o3 * cols;
          for (uint64_t e = c0; e < c1; e++) dst[e] = (dst[e] - o1[e] - o2[e] - o3[e]) & mask;
        } else {
          for (uint64_t e = c0; e < c1; e++) dst[e] = (dst[e] - o1[e] - o2[e]) & mask;
        }
      }
    };
    if (blocks <= 1) {
      column_block(0, cols);
    } else {
      std::vector<std::thread> pool;
      for (uint64_t t = 0; t < blocks; t++) pool.emplace_back(column_block, cols * t / blocks, cols * (t + 1) / blocks);
      for (auto& th : pool) th.join();
    }
  }

  memcpy(filter->seed, seed, 32);
  filter->arity = arity;
  filter->segment_length = seg_len;
  filter->segment_count_length = seg_count_len;
  filter->num_fingerprints = num_fp;
  filter->filter_size = n;
  filter->mat_elem_bit_len = b;
  *N_out = num_fp;
  *C_out = (uint32_t)cols;
  return CPIR_OK;
}

}  // namespace cpir
