// Arity-4 MMCS (MerkleTreeMmcs<.., 4, 8> over the width-32 permutation: `MyMmcsArity4`,
// recursion/examples/recursive_aggregation.rs:1024-1046): the level schedule, host only.  Shared by the prover's
// tree builder (p3r_core.hip), its query phase (prove_impl.hip.h) and the native verifier (verify_impl.h).
//
// What such a tree is, as far as the reference states it, is its in-circuit verifier (recursion/src/pcs/mmcs.rs:
// 866-960 `padded_len`, `arity4_path_schedule` - "matching native arity_schedule"): a level compresses `step`
// children, 4, or 2 (a bridge, zero-padded to four chunks) when a matrix that is still to be injected is taller
// than the next quaternary layer; a logical layer of 2 or 3 nodes is padded to 4 with zero digests; the matrices
// whose height equals the new layer's are injected after the level as one more compression
// (node, their row digest, 0, 0).  The native tree builder (p3-merkle-tree) is not in the tree.
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

namespace p3r {

struct Mmcs4Level {
  int step;             // 2 or 4
  size_t logical_next;  // nodes of the layer this level produces
  size_t padded_next;   // its allocated width (zero digests above logical_next)
  size_t inject_h;      // height of the matrices injected after the level, 0: none
  int bits;             // index bits consumed before this level
};

inline size_t mmcs4_npt(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
inline size_t mmcs4_padded_len(size_t raw, size_t n = 4) { return raw <= 1 ? raw : (raw >= n ? (raw + n - 1) / n * n : n); }

// `heights`: the heights of the committed matrices, any order.  One root (cap_height 0).
inline std::vector<Mmcs4Level> mmcs4_schedule(std::vector<size_t> heights) {
  std::stable_sort(heights.begin(), heights.end(), [](size_t a, size_t b) { return a > b; });
  const size_t max_height = heights.at(0), leaf_npt = mmcs4_npt(max_height);
  size_t at = 0;
  while (at < heights.size() && mmcs4_npt(heights[at]) == leaf_npt) ++at;
  std::vector<Mmcs4Level> levels;
  size_t curr = mmcs4_padded_len(max_height);
  int bits = 0;
  while (curr > 1) {
    int step = 2;
    if (curr >= 4) {
      const size_t target = mmcs4_npt(curr / 4);
      bool intermediate = false;
      for (size_t k = at; k < heights.size(); ++k) intermediate |= mmcs4_npt(heights[k]) > target;
      step = intermediate ? 2 : 4;
    }
    const size_t logical_next = curr / step;
    curr = mmcs4_padded_len(logical_next);
    size_t inject = 0;
    if (at < heights.size() && mmcs4_npt(heights[at]) == mmcs4_npt(logical_next)) {
      inject = heights[at];
      while (at < heights.size() && heights[at] == inject) ++at;
    }
    levels.push_back({step, logical_next, curr, inject, bits});
    bits += step == 4 ? 2 : 1;
  }
  return levels;
}
inline size_t mmcs4_proof_len(const std::vector<Mmcs4Level>& levels) {
  size_t n = 0;
  for (auto& l : levels) n += l.step - 1;
  return n;
}

}  // namespace p3r
