// Random values of a ZK configuration (p3r_config.zk; HidingFriPcs's `rng`, recursion/examples/common/mod.rs:536-542).
// The reference draws them sequentially from SmallRng::seed_from_u64(rng_seed); a device cannot follow a sequential
// generator cell by cell and - the proofs being randomised - no byte of them is pinned by it, so
// the values come from a counter-based generator instead: cell `idx` of stream `stream` of the `nonce`-th proof under
// `seed` is
//     mix64(key + idx * GOLD) mod p,   key = mix64(seed ^ mix64(nonce * GOLD + stream + 1)),
// mix64 = splitmix64's finaliser.  Streams: (round << 20) | matrix, rounds 0 random, 1 main, 2 quotient, 4 permutation,
// 5 quotient masks (the preprocessed round is padded with zeros).  The CPU oracle (oracle/stark.hpp: zk_rand) restates
// it, so HIP == oracle stays a byte comparison under ZK.  Like SmallRng it is NOT a cryptographic generator.
#pragma once
#include <cstdint>

#include "field.h"

namespace p3r {

constexpr uint64_t kZkGold = 0x9E3779B97F4A7C15ull;
P3R_HD uint64_t zk_mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
enum { ZK_ROUND_RANDOM = 0, ZK_ROUND_MAIN = 1, ZK_ROUND_QUOTIENT = 2, ZK_ROUND_PREP = 3, ZK_ROUND_PERM = 4, ZK_ROUND_QMASK = 5 };
inline uint64_t zk_stream_key(uint64_t seed, uint64_t nonce, int round, size_t mat) {
  const uint32_t stream = ((uint32_t)round << 20) | (uint32_t)mat;
  return zk_mix64(seed ^ zk_mix64(nonce * kZkGold + (uint64_t)stream + 1));
}
// the value as a Montgomery word
template <class PP>
P3R_HD uint32_t zk_rand_mont(uint64_t key, uint64_t idx) {
  return Fp<PP>::from_canonical((uint32_t)(zk_mix64(key + idx * kZkGold) % PP::P)).v;
}

}  // namespace p3r
