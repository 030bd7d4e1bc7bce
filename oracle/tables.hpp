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

// The width-32 table (arity-4 compression shape, 4 * CAPACITY_EXT == WIDTH_EXT): same two passes, the arity-4 layout
// [Poseidon2Cols | mmcs_bit | mmcs_bit2 | mmcs_bit * mmcs_bit2 | mmcs_index_sum] and a base-four accumulator
// (air.rs:370-432: continuation Merkle rows use 4 * acc + bit + 2 * bit2).
template <class FP>
struct P2WRow {
  bool new_start = false, merkle_path = false, mmcs_bit = false, mmcs_bit2 = false;
  Fe<FP> mmcs_index_sum;
  std::array<Fe<FP>, WIDTH32> input{};
};
template <class FP>
Matrix<FP> p2w_generate_trace_rows(const Poseidon2W32<FP>& p2, const std::vector<P2WRow<FP>>& rows) {
  using F = Fe<FP>;
  const size_t n = rows.size();
  log2_strict(n);
  const size_t pc = Poseidon2W32<FP>::perm_cols(), ncols = pc + 4;
  Matrix<FP> m(n, ncols);
  F prev = F::zero();
  for (size_t i = 0; i < n; ++i) {
    const auto& op = rows[i];
    if (i > 0 && op.merkle_path && !op.new_start)
      prev = prev * F(4) + (op.mmcs_bit ? F::one() : F::zero()) + (op.mmcs_bit2 ? F(2) : F::zero());
    else
      prev = op.mmcs_index_sum;
    std::vector<F> cells;
    auto s = op.input;
    p2.permute(s, &cells);
    if (cells.size() != pc) throw std::runtime_error("Poseidon2Cols<32> width mismatch");
    for (size_t c = 0; c < cells.size(); ++c) m.at(i, c) = cells[c];
    m.at(i, pc) = op.mmcs_bit ? F::one() : F::zero();
    m.at(i, pc + 1) = op.mmcs_bit2 ? F::one() : F::zero();
    m.at(i, pc + 2) = (op.mmcs_bit && op.mmcs_bit2) ? F::one() : F::zero();
    m.at(i, pc + 3) = prev;
  }
  return m;
}

}  // namespace orc

// ===========================================================================================
// Trace -> matrix and preprocessed-trace restatements for the primitive tables.
// ===========================================================================================
#include "air.hpp"

namespace orc {

template <class FP>
void pad_rows(Matrix<FP>& m, size_t min_height) {
  size_t h = 1;
  while (h < std::max<size_t>(m.h, 1)) h <<= 1;
  size_t mh = 1;
  while (mh < min_height) mh <<= 1;
  h = std::max(h, mh);
  m.v.resize(h * m.w);  // zero padding (pad_to_min_power_of_two_height(.., F::ZERO))
  m.h = h;
}

// ConstAir::trace_to_matrix (const_air.rs:88-127) / WitnessSendAir::trace_to_matrix
// (public_air.rs:127-169) / RecomposeAir::trace_to_matrix (recompose_air.rs:96-119):
// values: n_ops x D, laid out `lanes` ops per row, zero padded.
template <class FP>
Matrix<FP> lanes_trace_to_matrix(const std::vector<Fe<FP>>& values, int lanes, size_t min_height, int D = 4) {
  size_t n_ops = values.size() / D;
  size_t rows = std::max<size_t>((n_ops + lanes - 1) / lanes, 1);
  Matrix<FP> m(rows, (size_t)lanes * D);
  std::copy(values.begin(), values.end(), m.v.begin());
  pad_rows(m, min_height);
  return m;
}
// preprocessed_trace() of the same AIRs: flat per-op columns, `lanes` ops per row
// (RowMajorMatrix::from_flat_padded + pad_to_min_power_of_two_height).
template <class FP>
Matrix<FP> lanes_prep_to_matrix(const std::vector<Fe<FP>>& prep, int per_op, int lanes, size_t min_height) {
  size_t n_ops = prep.size() / per_op;
  size_t rows = std::max<size_t>((n_ops + lanes - 1) / lanes, 1);
  Matrix<FP> m(rows, (size_t)lanes * per_op);
  std::copy(prep.begin(), prep.end(), m.v.begin());
  pad_rows(m, min_height);
  return m;
}

// ---- ALU: schedule (alu_air.rs:349-463) ----
struct AluEntry {
  int kind;  // 0 = Op(first), 1 = PackedHorner(first, k), 2 = Separator
  size_t first = 0;
  int k = 1;
};
template <class FP>
bool alu_compute_schedule(const std::vector<Fe<FP>>& prep13, int lanes, int pack_k, std::vector<AluEntry>& sched) {
  const size_t plw = 13, n = prep13.size() / plw;
  sched.clear();
  if (n == 0) return false;
  std::vector<bool> is_h(n);
  bool any = false;
  for (size_t i = 0; i < n; ++i) { is_h[i] = prep13[i * plw + 4].v == 1; any = any || is_h[i]; }
  if (!any) return false;
  std::vector<std::vector<size_t>> chains;
  std::vector<size_t> cur, non_chain;
  for (size_t i = 0; i < n; ++i) {
    if (is_h[i]) cur.push_back(i);
    else { if (!cur.empty()) { chains.push_back(cur); cur.clear(); } non_chain.push_back(i); }
  }
  if (!cur.empty()) chains.push_back(cur);
  size_t nc = 0;
  auto fill_row = [&]() {
    while (sched.size() % lanes != 0) {
      if (nc < non_chain.size()) sched.push_back({0, non_chain[nc++], 1});
      else sched.push_back({2, 0, 1});
    }
  };
  sched.push_back({2, 0, 1});
  fill_row();
  for (size_t ci = 0; ci < chains.size(); ++ci) {
    const auto& chain = chains[ci];
    if (ci > 0) { fill_row(); sched.push_back({2, 0, 1}); fill_row(); }
    size_t i = 0;
    while (i < chain.size()) {
      size_t k_try = std::min<size_t>(chain.size() - i, pack_k);
      size_t best = 1;
      for (size_t k = k_try; k >= 2; --k) {
        bool ok = true;
        for (size_t j = 1; j < k && ok; ++j) ok = chain[i + j] == chain[i] + j;
        for (size_t j = 0; j < k && ok; ++j) ok = prep13[chain[i + j] * plw + 6] == prep13[chain[i] * plw + 6];
        if (ok) { best = k; break; }
      }
      if (best >= 2) { sched.push_back({1, chain[i], (int)best}); i += best; }
      else { sched.push_back({0, chain[i], 1}); i += 1; }
      fill_row();
    }
  }
  fill_row();
  while (nc < non_chain.size()) sched.push_back({0, non_chain[nc++], 1});
  fill_row();
  return true;
}

template <class FP>
FeX<FP> e4_at(const std::vector<Fe<FP>>& values, size_t op, int operand, int D, uint32_t W = 0) {
  FeX<FP> e(D, W);
  for (int d = 0; d < D; ++d) e.c[d] = values[(op * 4 + operand) * D + d];
  return e;
}

// AluAir::trace_to_matrix (alu_air.rs:497-608). values: n_ops x 4 operands x D.
template <class FP>
Matrix<FP> alu_trace_to_matrix(const AirDesc& a, const std::vector<Fe<FP>>& values,
                               const std::vector<Fe<FP>>& prep13, size_t min_height) {
  using EF = FeX<FP>;
  const int D = a.D;
  const int lanes = a.lanes, LW = 4 * D, k_max = a.horner_k;
  const int width = air_width<FP>(a), num_int = alu_num_int(k_max);
  std::vector<AluEntry> sched;
  bool has = alu_compute_schedule<FP>(prep13, lanes, k_max, sched);
  const size_t n_ops = values.size() / (4 * D);
  size_t entries = has ? sched.size() : n_ops;
  size_t rows = std::max<size_t>((entries + lanes - 1) / lanes, 1);
  Matrix<FP> m(rows, width);
  auto put = [&](size_t off, const EF& e) { for (int d = 0; d < D; ++d) m.v[off + d] = e.c[d]; };
  if (has) {
    EF prev = EF::zero(D);
    for (size_t pos = 0; pos < sched.size(); ++pos) {
      size_t row = pos / lanes, lane = pos % lanes;
      const auto& en = sched[pos];
      size_t base = row * width + lane * LW;
      if (en.kind == 0) {
        for (int o = 0; o < 4; ++o) put(base + o * D, e4_at<FP>(values, en.first, o, D, a.W));
        if (lane == 0) prev = e4_at<FP>(values, en.first, 3, D, a.W);
      } else if (en.kind == 1) {
        int k = en.k;
        for (int o = 0; o < 3; ++o) put(base + o * D, e4_at<FP>(values, en.first, o, D, a.W));
        put(base + 3 * D, e4_at<FP>(values, en.first + k - 1, 3, D, a.W));
        if (lane == 0) {
          size_t extra = row * width + (size_t)lanes * LW;
          EF b = e4_at<FP>(values, en.first, 1, D, a.W), acc = prev;
          int step = 0;
          for (int s = 0; s < num_int; ++s) {
            size_t i0 = en.first + step, i1 = i0 + 1;
            if (i1 < en.first + k) {
              EF o0 = acc * b + e4_at<FP>(values, i0, 2, D, a.W) - e4_at<FP>(values, i0, 0, D, a.W);
              acc = o0 * b + e4_at<FP>(values, i1, 2, D, a.W) - e4_at<FP>(values, i1, 0, D, a.W);
              step += 2;
            } else {
              acc = acc * b + e4_at<FP>(values, i0, 2, D, a.W) - e4_at<FP>(values, i0, 0, D, a.W);
              step += 1;
            }
            put(extra + s * D, acc);
          }
          size_t ac_base = extra + num_int * D;
          for (int t = 1; t < k; ++t) {
            put(ac_base + 2 * (t - 1) * D, e4_at<FP>(values, en.first + t, 0, D, a.W));
            put(ac_base + 2 * (t - 1) * D + D, e4_at<FP>(values, en.first + t, 2, D, a.W));
          }
          put(ac_base + 2 * (k_max - 1) * D, b * b);
          prev = e4_at<FP>(values, en.first + k - 1, 3, D, a.W);
        }
      } else if (lane == 0) {
        prev = EF::zero(D);
      }
    }
  } else {
    for (size_t op = 0; op < n_ops; ++op)
      for (int o = 0; o < 4; ++o) put((op / lanes) * width + (op % lanes) * LW + o * D, e4_at<FP>(values, op, o, D, a.W));
  }
  pad_rows(m, min_height);
  return m;
}

// AluAir::preprocessed_trace (alu_air.rs:613-706)
template <class FP>
Matrix<FP> alu_preprocessed_trace(const AirDesc& a, const std::vector<Fe<FP>>& prep13, size_t min_height) {
  using F = Fe<FP>;
  const int lanes = a.lanes, plw = 13, k_max = a.horner_k;
  const int pw = air_prep_width(a);
  std::vector<AluEntry> sched;
  bool has = alu_compute_schedule<FP>(prep13, lanes, k_max, sched);
  const size_t n_ops = prep13.size() / plw;
  size_t entries = has ? sched.size() : n_ops;
  size_t rows = std::max<size_t>((entries + lanes - 1) / lanes, 1);
  Matrix<FP> m(rows, pw);
  if (!has) {
    for (size_t op = 0; op < n_ops; ++op)
      for (int j = 0; j < plw; ++j) m.v[(op / lanes) * pw + (op % lanes) * plw + j] = prep13[op * plw + j];
  } else {
    for (size_t pos = 0; pos < sched.size(); ++pos) {
      size_t row = pos / lanes, lane = pos % lanes;
      const auto& en = sched[pos];
      size_t base = row * pw + lane * plw;
      if (en.kind == 0) {
        for (int j = 0; j < plw; ++j) m.v[base + j] = prep13[en.first * plw + j];
      } else if (en.kind == 1 && lane == 0) {
        int k = en.k;
        size_t last = en.first + k - 1;
        for (int j = 0; j < plw; ++j) m.v[base + j] = prep13[en.first * plw + j];
        m.v[base + 8] = prep13[last * plw + 8];    // out_idx of the last op
        m.v[base + 10] = prep13[last * plw + 10];  // mult_out of the last op
        m.v[base + 9] = m.v[base + 9] * F((uint64_t)k);  // mult_b *= k
        F mult_a_lane = m.v[base + 0];
        size_t extra = row * pw + (size_t)lanes * plw;
        m.v[extra + (k - 2)] = F::one();           // sel_k
        for (int t = 1; t < k; ++t) {
          const F* src = &prep13[(en.first + t) * plw];
          size_t p = extra + (k_max - 1) + 6 * (t - 1);
          m.v[p + 0] = src[5]; m.v[p + 1] = src[7]; m.v[p + 2] = src[11]; m.v[p + 3] = src[12];
          m.v[p + 4] = mult_a_lane * src[11];
          m.v[p + 5] = mult_a_lane * src[12];
        }
      }
    }
  }
  pad_rows(m, min_height);
  return m;
}

// Poseidon2 preprocessed trace: extract_preprocessed_from_operations (air.rs:697-794, D=4
// non-compact layout) followed by BaseAir::preprocessed_trace padding (air.rs:613-649: the
// first padding row carries new_start = 1 at width-2).
template <class FP>
struct P2CtlRow {
  bool new_start, merkle_path, mmcs_ctl_enabled;
  uint32_t in_ctl[4], input_indices[4], output_indices[2], mmcs_index_sum_idx;
  Fe<FP> out_ctl[2];
};
template <class FP>
Matrix<FP> p2_preprocessed_trace(const std::vector<P2CtlRow<FP>>& rows, size_t min_height) {
  using F = Fe<FP>;
  constexpr int D = 4;  // the non-compact layout is the D = 4 one
  const size_t w = 24, n = rows.size();
  Matrix<FP> m(std::max<size_t>(n, 1), w);
  for (size_t r = 0; r < n; ++r) {
    const auto& op = rows[r];
    F* o = &m.v[r * w];
    for (int l = 0; l < 4; ++l) {
      bool ctl = op.in_ctl[l];
      o[l * 4 + 0] = F((uint64_t)op.input_indices[l] * D);
      o[l * 4 + 1] = F(ctl ? 1 : 0);
      o[l * 4 + 2] = F((!op.new_start && !op.merkle_path && !ctl) ? 1 : 0);
      o[l * 4 + 3] = F((!op.new_start && op.merkle_path && !ctl) ? 1 : 0);
    }
    for (int l = 0; l < 2; ++l) {
      o[16 + l * 2] = F((uint64_t)op.output_indices[l] * D);
      o[16 + l * 2 + 1] = op.out_ctl[l];
    }
    o[20] = F((uint64_t)op.mmcs_index_sum_idx * D);
    o[21] = F((op.mmcs_ctl_enabled && op.merkle_path) ? 1 : 0);
    o[22] = F(op.new_start ? 1 : 0);
    o[23] = F(op.merkle_path ? 1 : 0);
  }
  size_t natural = n;
  if (n == 0) m.h = 0;
  pad_rows(m, min_height);
  if (m.h > natural) m.v[natural * w + w - 2] = F::one();
  return m;
}

// Compact D1 layout of the same table (IL = 16, OL = 8: one witness per state element; air.rs:730-763 for the
// CTL fields, circuit/src/ops/poseidon_perm/executor.rs:720-741 for the header incl. the length tag in slot 8),
// witness indices scaled by the circuit's extension degree d; padding as above (air.rs:613-649).
template <class FP>
struct P2CtlRowD1 {
  bool new_start, merkle_path, mmcs_ctl_enabled;
  uint32_t in_ctl[16], input_indices[16], output_indices[8], mmcs_index_sum_idx, absorb_len;
  Fe<FP> out_ctl[8];
};
template <class FP>
Matrix<FP> p2_preprocessed_trace_d1(const std::vector<P2CtlRowD1<FP>>& rows, size_t min_height, int d) {
  using F = Fe<FP>;
  const size_t w = 62, n = rows.size();
  Matrix<FP> m(std::max<size_t>(n, 1), w);
  for (size_t r = 0; r < n; ++r) {
    const auto& op = rows[r];
    F* o = &m.v[r * w];
    if (!op.merkle_path)
      for (int l = 8; l < 16; ++l)
        if (op.in_ctl[l]) throw std::runtime_error("compact D=1 Poseidon2: capacity must not be witness-fed on sponge rows");
    for (int l = 0; l < 8; ++l) o[l] = F(op.in_ctl[l] ? 1 : 0);
    o[8] = F(op.absorb_len);
    o[9] = F(op.new_start ? 0 : 1);
    for (int l = 0; l < 8; ++l) o[10 + l] = F((!op.new_start && !op.merkle_path && !op.in_ctl[l]) ? 1 : 0);
    for (int l = 0; l < 8; ++l) o[18 + l] = F((!op.new_start && op.merkle_path && !op.in_ctl[l]) ? 1 : 0);
    for (int l = 0; l < 16; ++l) o[26 + l] = F((uint64_t)op.input_indices[l] * d);
    for (int l = 0; l < 8; ++l) o[42 + l] = F((uint64_t)op.output_indices[l] * d);
    for (int l = 0; l < 8; ++l) o[50 + l] = op.out_ctl[l];
    o[58] = F((uint64_t)op.mmcs_index_sum_idx * d);
    o[59] = F((op.mmcs_ctl_enabled && op.merkle_path) ? 1 : 0);
    o[60] = F(op.new_start ? 1 : 0);
    o[61] = F(op.merkle_path ? 1 : 0);
  }
  size_t natural = n;
  if (n == 0) m.h = 0;
  pad_rows(m, min_height);
  if (m.h > natural) m.v[natural * w + w - 2] = F::one();
  return m;
}

}  // namespace orc
