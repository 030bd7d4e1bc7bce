// Field order of the postcard-serialised proof structs (p3r_config.proof_layout): shared by the
// prover's writer (prove_impl.hip.h), the native verifier's reader (verify_impl.h) and the ctx.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace p3r {

// Order of the fields of the postcard-serialised structs (serde writes fields in declaration order; the
// declarations are upstream's).  Default = identity = the order read off the destructuring patterns at
// recursion/src/types/proof.rs:403-409,452-457,527-534,585-589 and pcs/fri/targets.rs:104-110.
//   batch[5]   BatchProof:   0 commitments, 1 opened_values, 2 opening_proof, 3 global_lookup_data, 4 degree_bits
//   fri[5]     FriProof:     0 commit_phase_commits, 1 commit_pow_witnesses, 2 query_proofs, 3 final_poly, 4 query_pow_witness
//   opened[8]  OpenedValues: 0 trace_local, 1 trace_next, 2 preprocessed_local, 3 preprocessed_next, 4 quotient_chunks,
//                            5 random, 6 permutation_local, 7 permutation_next
// p3r_config.proof_layout is the 18 bytes batch | fri | opened (each a permutation).
struct ProofLayout {
  uint8_t batch[5] = {0, 1, 2, 3, 4};
  uint8_t fri[5] = {0, 1, 2, 3, 4};
  uint8_t opened[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  static bool is_perm(const uint8_t* v, int n) {
    uint32_t seen = 0;
    for (int i = 0; i < n; ++i) { if (v[i] >= n || (seen >> v[i] & 1)) return false; seen |= 1u << v[i]; }
    return true;
  }
  // false when `bytes` is not three permutations
  bool set(const uint8_t* bytes, size_t n) {
    if (!bytes) return true;
    if (n != 18 || !is_perm(bytes, 5) || !is_perm(bytes + 5, 5) || !is_perm(bytes + 10, 8)) return false;
    std::memcpy(batch, bytes, 5); std::memcpy(fri, bytes + 5, 5); std::memcpy(opened, bytes + 10, 8);
    return true;
  }
};

}  // namespace p3r
