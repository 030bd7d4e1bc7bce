// ORACLE (test infrastructure): trace -> matrix restatements for the recursion tables.
// PARITY UNPINNED (see field.hpp).
#pragma once
#include "hash.hpp"

namespace orc {

// One Poseidon2CircuitRow, main-trace fields only
// (circuit/src/ops/poseidon2_perm/trace.rs:94-125).
template <class FP>
struct P2Row {
  bool new_start = false, merkle_path = false, mmcs_bit = false;
  Fe<FP> mmcs_index_sum;
  std::array<Fe<FP>, WIDTH> input{};
};

// Poseidon2CircuitAir::generate_trace_rows (poseidon2-circuit-air/src/air.rs:280-520),
// arity-2 layout: [Poseidon2Cols | mmcs_bit | mmcs_index_sum].
//   pass 1 (:372-435) sequential accumulator: continuation Merkle rows use 2*acc + bit,
//          every other row resets to the row's own mmcs_index_sum (:401-412);
//   pass 2 (:454-506) one permutation per row, every round's cells recorded.
template <class FP>
Matrix<FP> p2_generate_trace_rows(const Poseidon2<FP>& p2, const std::vector<P2Row<FP>>& rows) {
  using F = Fe<FP>;
  const size_t n = rows.size();
  log2_strict(n);  // callers pad to a power of two (:287-290)
  const size_t ncols = Poseidon2<FP>::perm_cols() + 2;
  Matrix<FP> m(n, ncols);
  F prev = F::zero();
  for (size_t i = 0; i < n; ++i) {
    const auto& op = rows[i];
    if (i > 0 && op.merkle_path && !op.new_start)
      prev = prev + prev + (op.mmcs_bit ? F::one() : F::zero());
    else
      prev = op.mmcs_index_sum;
    std::vector<F> cells;
    auto s = op.input;
    p2.permute(s, &cells);
    if (cells.size() != ncols - 2) throw std::runtime_error("Poseidon2Cols width mismatch");
    for (size_t c = 0; c < cells.size(); ++c) m.at(i, c) = cells[c];
    m.at(i, ncols - 2) = op.mmcs_bit ? F::one() : F::zero();
    m.at(i, ncols - 1) = prev;
  }
  return m;
}

}  // namespace orc
