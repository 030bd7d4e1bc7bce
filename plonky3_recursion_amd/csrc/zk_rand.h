// Random values of a ZK configuration (p3r_config.zk; HidingFriPcs's `rng: R`, recursion/examples/common/mod.rs:536-542 -
// the generator is the CALLER's there, and a caller that wants hiding passes a cryptographic one).
// The reference draws sequentially from its RNG; a device cannot follow a sequential generator cell by cell and - the
// proofs being randomised - no byte of them is pinned by it.  So the values come from a KEYED
// counter-based generator: ChaCha with 8 rounds (Bernstein's ChaCha, the 8-round variant: RFC 8439's block function
// with 4 double rounds instead of 10) under the context's 256-bit key.  Block input words 12..15 are
//     [ counter, stream, nonce_lo, nonce_hi ]      nonce = proofs made so far by the context,
//     stream = (round << 20) | matrix              rounds: 0 random, 1 main, 2 quotient, 4 permutation, 5 quotient masks
//                                                  (the preprocessed round is padded with zeros)
// and cell `idx` of a stream is a field element drawn by REJECTION from 31-bit words (no modulo bias): candidates 2j and
// 2j + 1 (j = idx mod 8) of block idx / 8, the first one below p; if both are refused (probability < 2^-8), the words of
// up to seven fallback blocks [ idx mod 2^32, stream | f << 24 | (idx >> 32) << 27 ], f = 1..7, in order.
// The CPU oracle restates this (oracle/stark.hpp: ZkStream), so HIP == oracle stays a byte comparison under a fixed key.
//
// The key: p3r_config.zk_key (eight words).  Unless the caller asks for reproducible proofs (P3R_EXT_ZK_DETERMINISTIC),
// p3r_create mixes 128 bits from the operating system into it (zk_context_key), so two contexts - or two runs - never
// mask different witnesses with the same values even when the caller's key repeats or is zero.
#pragma once
#include <cstdint>

#include "field.h"

namespace p3r {

enum { ZK_ROUND_RANDOM = 0, ZK_ROUND_MAIN = 1, ZK_ROUND_QUOTIENT = 2, ZK_ROUND_PREP = 3, ZK_ROUND_PERM = 4, ZK_ROUND_QMASK = 5 };
// salts of a hiding MMCS (p3r_config.mmcs_salt_elems): stream round kSaltRound + the round of the committed batch, matrix =
// its position in the batch; kSaltRoundFri for the FRI commit-phase trees (matrix = phase)
constexpr int kSaltRound = 8, kSaltRoundFri = 14;

// what a kernel needs besides the stream id: the context's key and the proof's nonce (passed by value)
struct ZkKey {
  uint32_t k[8];
  uint32_t nonce_lo, nonce_hi;
};
inline uint32_t zk_stream_id(int round, size_t mat) { return ((uint32_t)round << 20) | (uint32_t)mat; }

P3R_HD uint32_t zk_rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define P3R_ZK_QR(a, b, c, d) \
  a += b; d ^= a; d = zk_rotl(d, 16); c += d; b ^= c; b = zk_rotl(b, 12); \
  a += b; d ^= a; d = zk_rotl(d, 8);  c += d; b ^= c; b = zk_rotl(b, 7);
// one ChaCha8 block: out[16] = core(in) + in
P3R_HD void zk_chacha8_block(const uint32_t key[8], uint32_t w12, uint32_t w13, uint32_t w14, uint32_t w15, uint32_t out[16]) {
  const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                           key[4], key[5], key[6], key[7], w12, w13, w14, w15};
  uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7];
  uint32_t x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    P3R_ZK_QR(x0, x4, x8, x12) P3R_ZK_QR(x1, x5, x9, x13) P3R_ZK_QR(x2, x6, x10, x14) P3R_ZK_QR(x3, x7, x11, x15)
    P3R_ZK_QR(x0, x5, x10, x15) P3R_ZK_QR(x1, x6, x11, x12) P3R_ZK_QR(x2, x7, x8, x13) P3R_ZK_QR(x3, x4, x9, x14)
  }
  out[0] = x0 + in[0]; out[1] = x1 + in[1]; out[2] = x2 + in[2]; out[3] = x3 + in[3];
  out[4] = x4 + in[4]; out[5] = x5 + in[5]; out[6] = x6 + in[6]; out[7] = x7 + in[7];
  out[8] = x8 + in[8]; out[9] = x9 + in[9]; out[10] = x10 + in[10]; out[11] = x11 + in[11];
  out[12] = x12 + in[12]; out[13] = x13 + in[13]; out[14] = x14 + in[14]; out[15] = x15 + in[15];
}
#undef P3R_ZK_QR

// canonical value of cell `idx` of `stream` (idx < 2^35)
template <class PP>
P3R_HD uint32_t zk_rand_canonical(const ZkKey& key, uint32_t stream, uint64_t idx) {
  uint32_t blk[16];
  zk_chacha8_block(key.k, (uint32_t)(idx >> 3), stream, key.nonce_lo, key.nonce_hi, blk);
  const int j = (int)(idx & 7);
  // words 2j, 2j + 1 by compare-and-select (a dynamic index would put the block in scratch memory on the device)
  uint32_t c0 = blk[0], c1 = blk[1];
#pragma unroll
  for (int t = 1; t < 8; ++t) { c0 = j == t ? blk[2 * t] : c0; c1 = j == t ? blk[2 * t + 1] : c1; }
  uint32_t v = c0 & 0x7FFFFFFFu;
  if (v < PP::P) return v;
  v = c1 & 0x7FFFFFFFu;
  if (v < PP::P) return v;
  for (uint32_t f = 1; f <= 7; ++f) {
    zk_chacha8_block(key.k, (uint32_t)idx, stream | (f << 24) | ((uint32_t)(idx >> 32) << 27), key.nonce_lo, key.nonce_hi, blk);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v = blk[i] & 0x7FFFFFFFu;
      if (v < PP::P) return v;
    }
  }
  return v % PP::P;   // 114 refusals in a row: probability below 2^-450
}
// the value as a Montgomery word
template <class PP>
P3R_HD uint32_t zk_rand_mont(const ZkKey& key, uint32_t stream, uint64_t idx) {
  return Fp<PP>::from_canonical(zk_rand_canonical<PP>(key, stream, idx)).v;
}

// the key a context draws with: the caller's key as it is (deterministic: tests, replay), or the first eight words of the
// ChaCha8 block of (caller's key, 128 bits of operating-system entropy)
inline void zk_context_key(const uint32_t caller_key[8], const uint32_t entropy[4], bool deterministic, uint32_t out[8]) {
  if (deterministic) {
    for (int i = 0; i < 8; ++i) out[i] = caller_key[i];
    return;
  }
  uint32_t blk[16];
  zk_chacha8_block(caller_key, entropy[0], entropy[1], entropy[2], entropy[3], blk);
  for (int i = 0; i < 8; ++i) out[i] = blk[i];
}

}  // namespace p3r
