// ORACLE (test infrastructure, never linked into the product): CPU restatement of the caller side
// of prove_next_layer - the flattened Circuit, its preprocessing and its runner.
//
//   generate_preprocessed_columns   circuit/src/circuit.rs:237-510
//   NPO executors' preprocess()     circuit/src/ops/poseidon_perm/executor.rs:690-920 (non-compact D>1 layout),
//                                   circuit/src/ops/recompose.rs:172-192
//   poseidon_preprocess_for_prover  circuit-prover/src/batch_stark_prover.rs:97-246
//   recompose_preprocess_for_op     circuit-prover/src/batch_stark_prover/recompose.rs:294-358
//   get_airs_and_degrees_with_prep  circuit-prover/src/common.rs:127-390 (primitive-table part)
//   CircuitRunner::{execute_all, run} circuit/src/tables/runner.rs:195-510
//   PoseidonPermExecutor::execute   circuit/src/ops/poseidon_perm/executor.rs:921-972 (+ helpers :103-520); the arity-4
//                                   shape of the width-32 table (COP_P2W): is_arity4 :92-95, place_arity4_running_hash
//                                   :141-160, fill_sibling_data :166-201, resolve_mmcs_bit2 :290-303, update_chain_state
//                                   :462-491, validate_ext_inputs :493-545, preprocess_* :770-893
//   RecomposeExecutor::execute      circuit/src/ops/recompose.rs:115-170
//   Ext/BinaryDecompositionHint     circuit/src/builder/circuit_builder.rs:1659-1810
//
// PARITY: generate_preprocessed_columns is pinned by the literal expectations of the reference's
// own unit tests (circuit.rs:560-760, transcribed in tests/golden/reference_unit_tests.json); the
// runner by runner.rs:556-790 and tables/{alu,constant,public}.rs tests.  Everything that needs the
// un-vendored p3-* crates (the permutation itself) stays UNPINNED, see field.hpp.
#pragma once
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "field.hpp"
#include "hash.hpp"

namespace orc {

constexpr uint32_t NO_W = 0xFFFFFFFFu;
enum : uint32_t {
  COP_CONST = 0, COP_PUBLIC = 1, COP_ADD = 2, COP_MUL = 3, COP_BOOL = 4, COP_MULADD = 5, COP_HORNER = 6,
  COP_HINT_EXT = 7, COP_HINT_BIN = 8, COP_P2 = 9, COP_RECOMPOSE = 10,
  // a permutation of the width-32 table (D4_W32: width_ext 8, rate_ext 6, capacity_ext 2 - the arity-4 compression shape):
  // ext = [in0..in7, mmcs_index_sum, mmcs_bit, mmcs_bit2, n_out, out0..]
  COP_P2W = 11
};
inline bool is_alu(uint32_t k) { return k >= COP_ADD && k <= COP_HORNER; }

struct COp {
  uint32_t kind, a, b, c, out, aux, ext_off, ext_len;
};

struct CircuitDesc {
  uint32_t witness_count = 0;
  std::vector<COp> ops;
  std::vector<uint32_t> ext;
  std::vector<uint32_t> public_rows, private_rows;
  std::vector<std::pair<uint32_t, uint32_t>> rewrite;  // (duplicate, canonical)
  const uint32_t* ext_of(const COp& op) const {
    if ((size_t)op.ext_off + op.ext_len > ext.size()) throw std::runtime_error("op ext slice out of range");
    return ext.data() + op.ext_off;
  }
};

// PreprocessedColumns<F, D> (circuit.rs:26-65) for the three-table backend.
struct Preprocessed {
  uint32_t P = 0;
  int D = 4;
  std::vector<uint32_t> prim_const, prim_public, prim_alu12;  // primitive[Const|Public|Alu]
  std::vector<uint32_t> p2_rows;     // non_primitive[poseidon2_perm/..]: 24 values per op
  std::vector<uint32_t> recompose;   // non_primitive[recompose]: [output idx, 1] per op
  // non_primitive[recompose/coeff] (ops with aux = 1): [output idx, 1] then [coefficient idx, 1] per coefficient
  // (ops/recompose.rs:174-192) - a separate op type, hence a separate table and a separate duplicate map
  std::vector<uint32_t> recompose_coeff;
  std::vector<uint32_t> ext_reads;
  std::vector<uint32_t> p2w_rows;    // non_primitive[poseidon2_perm/.._d4_w32]: 48 values per op
  std::vector<bool> dup_p2, dup_recompose, dup_recompose_coeff, dup_p2w;  // dup_npo_outputs[op_type][wid]
  std::set<uint32_t> hint_output_wids;
  uint32_t idx(uint32_t wid) const { return (uint32_t)(((uint64_t)wid * (uint64_t)D) % P); }  // base_field_index
  void read(uint32_t wid) {  // increment_ext_reads
    if (wid >= ext_reads.size()) ext_reads.resize((size_t)wid + 1, 0);
    ext_reads[wid]++;
  }
};

// circuit.rs:237-510
inline Preprocessed generate_preprocessed_columns(const CircuitDesc& c, uint32_t P, int D) {
  Preprocessed pp;
  pp.P = P; pp.D = D;
  std::vector<bool> defined(c.witness_count, false);
  auto is_def = [&](uint32_t w) { return w < defined.size() && defined[w]; };
  auto define = [&](uint32_t w) {
    if (w >= defined.size()) defined.resize((size_t)w + 1, false);
    defined[w] = true;
  };
  std::set<uint32_t> private_wids(c.private_rows.begin(), c.private_rows.end());
  std::set<uint32_t> const_public;
  for (auto& op : c.ops)
    if (op.kind == COP_CONST || op.kind == COP_PUBLIC) const_public.insert(op.out);
  for (auto& op : c.ops)
    if (op.kind == COP_HINT_EXT || op.kind == COP_HINT_BIN) {
      const uint32_t* outs = c.ext_of(op);
      for (uint32_t i = 0; i < op.ext_len; ++i)
        if (!const_public.count(outs[i])) pp.hint_output_wids.insert(outs[i]);
    }
  const auto& hints = pp.hint_output_wids;

  for (auto& op : c.ops) {
    switch (op.kind) {
      case COP_CONST: pp.prim_const.push_back(pp.idx(op.out)); define(op.out); break;
      case COP_PUBLIC: pp.prim_public.push_back(pp.idx(op.out)); define(op.out); break;
      case COP_ADD: case COP_MUL: case COP_BOOL: case COP_MULADD: case COP_HORNER: {
        const uint32_t sel[4] = {op.kind == COP_ADD, op.kind == COP_BOOL, op.kind == COP_MULADD, op.kind == COP_HORNER};
        const bool out_def = is_def(op.out), b_def = is_def(op.b);
        auto state_of = [&](uint32_t w) -> uint32_t {
          if (is_def(w)) return 1;
          const bool aliased = !out_def && w == op.out;
          if ((private_wids.count(w) || hints.count(w)) && !aliased) return 2;
          return 0;
        };
        const uint32_t a_state = state_of(op.a);
        const bool has_c = op.c != NO_W;
        const uint32_t c_wid = has_c ? op.c : 0;
        const uint32_t c_state = has_c ? state_of(op.c) : 0;
        const bool b_private_creator = !b_def && private_wids.count(op.b);
        const bool out_backward = out_def || hints.count(op.out);
        const bool out_creator = !out_def;
        const bool b_creator = b_private_creator || (out_backward && !b_def);
        const uint32_t row[12] = {sel[0], sel[1], sel[2], sel[3], pp.idx(op.a), pp.idx(op.b), pp.idx(c_wid),
                                  pp.idx(op.out), a_state, b_creator, c_state, out_creator};
        pp.prim_alu12.insert(pp.prim_alu12.end(), row, row + 12);
        if (!b_creator) pp.read(op.b);
        if (!out_creator) pp.read(op.out);
        if (a_state == 1) pp.read(op.a);
        if (c_state == 1) pp.read(c_wid);
        if (out_creator) define(op.out);
        if (b_creator) define(op.b);
        if (a_state == 2) define(op.a);
        if (c_state == 2) define(c_wid);
        break;
      }
      case COP_P2: {
        // executor.rs:770-920, D > 1 layout: 4 x [idx, in_ctl, normal_chain_sel, merkle_chain_sel],
        // 2 x [out idx, out_ctl], [mmcs idx, mmcs_merkle_flag, new_start, merkle_path]
        const uint32_t* e = c.ext_of(op);
        if (op.ext_len < 7) throw std::runtime_error("poseidon2 op: ext too short");
        const uint32_t n_out = e[6];
        if ((n_out != 2 && n_out != 4) || op.ext_len != 7 + n_out) throw std::runtime_error("poseidon2 op: bad output count");
        const bool new_start = op.aux & 1, merkle = (op.aux >> 1) & 1;
        for (int l = 0; l < 4; ++l) {
          const bool empty = e[l] == NO_W;
          if (empty) { pp.p2_rows.push_back(0); pp.p2_rows.push_back(0); }
          else if (merkle) { pp.p2_rows.push_back(pp.idx(e[l])); pp.p2_rows.push_back(1); }
          else { pp.p2_rows.push_back(pp.idx(e[l])); pp.read(e[l]); pp.p2_rows.push_back(1); }
          pp.p2_rows.push_back(!new_start && !merkle && empty);
          pp.p2_rows.push_back(!new_start && merkle && empty);
        }
        for (int l = 0; l < 2; ++l) {
          const uint32_t w = e[7 + l];
          if (w == NO_W) { pp.p2_rows.push_back(0); pp.p2_rows.push_back(0); }
          else { pp.p2_rows.push_back(pp.idx(w)); pp.p2_rows.push_back(1); }
        }
        const bool mmcs_en = e[4] != NO_W;
        pp.p2_rows.push_back(mmcs_en ? pp.idx(e[4]) : 0);
        pp.p2_rows.push_back(mmcs_en && merkle);
        pp.p2_rows.push_back(new_start);
        pp.p2_rows.push_back(merkle);
        // duplicate-output bookkeeping over the exposed (rate) outputs, circuit.rs:464-491
        for (int l = 0; l < 2; ++l) {
          const uint32_t w = e[7 + l];
          if (w == NO_W) continue;
          if (is_def(w)) {
            if (w >= pp.dup_p2.size()) pp.dup_p2.resize((size_t)w + 1, false);
            pp.dup_p2[w] = true;
            pp.read(w);
          } else {
            define(w);
          }
        }
        break;
      }
      case COP_P2W: {
        // executor.rs:770-893 for is_arity4_shape(): 8 x [idx, in_ctl, normal_chain_sel, merkle_chain_sel] where EVERY named
        // limb is a witness read (Merkle rows included: their AIR sends a bare in_ctl), 6 x [out idx, out_ctl], then on a
        // Merkle row the witness indices of the two direction bits (both read) in the accumulator / flag slots
        const uint32_t* e = c.ext_of(op);
        if (D != 4) throw std::runtime_error("width-32 poseidon2 op: D = 4 circuits only");
        if (op.ext_len < 12) throw std::runtime_error("width-32 poseidon2 op: ext too short");
        const uint32_t n_out = e[11];
        if ((n_out != 6 && n_out != 8) || op.ext_len != 12 + n_out) throw std::runtime_error("width-32 poseidon2 op: bad output count");
        const bool new_start = op.aux & 1, merkle = (op.aux >> 1) & 1;
        if (e[8] != NO_W) throw std::runtime_error("width-32 poseidon2 op: mmcs_index_sum is not supported on this table");
        if (merkle && (e[9] == NO_W || e[10] == NO_W)) throw std::runtime_error("width-32 poseidon2 op: a Merkle row needs both direction bits");
        if (!merkle && (e[9] != NO_W || e[10] != NO_W)) throw std::runtime_error("width-32 poseidon2 op: a direction bit on a sponge row");
        for (int l = 0; l < 8; ++l) {
          const bool empty = e[l] == NO_W;
          if (empty) { pp.p2w_rows.push_back(0); pp.p2w_rows.push_back(0); }
          else { pp.p2w_rows.push_back(pp.idx(e[l])); pp.read(e[l]); pp.p2w_rows.push_back(1); }
          pp.p2w_rows.push_back(!new_start && !merkle && empty);
          pp.p2w_rows.push_back(!new_start && merkle && empty);
        }
        for (int l = 0; l < 6; ++l) {
          const uint32_t w = e[12 + l];
          if (w == NO_W) { pp.p2w_rows.push_back(0); pp.p2w_rows.push_back(0); }
          else { pp.p2w_rows.push_back(pp.idx(w)); pp.p2w_rows.push_back(1); }
        }
        if (merkle) {
          pp.p2w_rows.push_back(pp.idx(e[9])); pp.read(e[9]);
          pp.p2w_rows.push_back(pp.idx(e[10])); pp.read(e[10]);
        } else {
          pp.p2w_rows.push_back(0);   // no mmcs_index_sum witness
          pp.p2w_rows.push_back(0);   // mmcs_merkle_flag = ctl && merkle
        }
        pp.p2w_rows.push_back(new_start);
        pp.p2w_rows.push_back(merkle);
        for (int l = 0; l < 6; ++l) {   // circuit.rs:464-491 over the num_exposed_outputs() = rate_ext outputs
          const uint32_t w = e[12 + l];
          if (w == NO_W) continue;
          if (is_def(w)) {
            if (w >= pp.dup_p2w.size()) pp.dup_p2w.resize((size_t)w + 1, false);
            pp.dup_p2w[w] = true;
            pp.read(w);
          } else {
            define(w);
          }
        }
        break;
      }
      case COP_RECOMPOSE: {
        if (op.ext_len != 4) throw std::runtime_error("recompose op: needs 4 coefficient witnesses");
        const bool coeff = op.aux == 1;
        auto& rows = coeff ? pp.recompose_coeff : pp.recompose;
        auto& dups = coeff ? pp.dup_recompose_coeff : pp.dup_recompose;
        rows.push_back(pp.idx(op.out));
        rows.push_back(1);
        if (coeff)   // register_non_primitive_output_index per coefficient: named, not marked defined, no read counted
          for (uint32_t k = 0; k < op.ext_len; ++k) {
            rows.push_back(pp.idx(c.ext[op.ext_off + k]));
            rows.push_back(1);
          }
        if (is_def(op.out)) {
          if (op.out >= dups.size()) dups.resize((size_t)op.out + 1, false);
          dups[op.out] = true;
          pp.read(op.out);
        } else {
          define(op.out);
        }
        break;
      }
      case COP_HINT_EXT: case COP_HINT_BIN: break;
      default: throw std::runtime_error("unknown op kind");
    }
  }
  if (pp.ext_reads.size() < c.witness_count) pp.ext_reads.resize(c.witness_count, 0);
  for (uint32_t w : c.private_rows)
    if (!is_def(w)) throw std::runtime_error("UnclaimedPrivateInput: witness " + std::to_string(w));
  return pp;
}

// The per-table preprocessed data prove_all_tables consumes (the orc_workload / p3r_layer_desc
// conventions), from the generic columns.
struct CircuitPrep {
  std::vector<uint32_t> const_prep, public_prep, alu_prep13, recompose_prep;
  // the `recompose/coeff` rows: the layer's second Recompose table - or, when the circuit has no plain Recompose op,
  // its only one (a table without rows is not proved): then they are in recompose_prep and this flag is set
  std::vector<uint32_t> recompose_coeff_prep;
  bool recompose_coeff_only = false;
  std::vector<uint32_t> p2_rows;  // 24 per row, out_ctl replaced by the multiplicity
  std::vector<uint32_t> p2w_rows; // 48 per row (the width-32 table), out_ctl replaced by the multiplicity
  std::vector<uint32_t> ext_reads;
};

inline CircuitPrep get_airs_and_degrees_with_prep(Preprocessed pp) {
  const uint32_t P = pp.P, D = (uint32_t)pp.D, neg1 = P - 1;
  CircuitPrep out;
  auto reads = [&](uint32_t wid) { return wid < pp.ext_reads.size() ? pp.ext_reads[wid] % P : 0u; };
  // ---- poseidon_preprocess_for_prover, phase 1: conditional mmcs_index_sum reads (:110-175)
  const size_t n_p2 = pp.p2_rows.size() / 24;
  {
    size_t h = 1;
    while (h < n_p2) h <<= 1;
    const bool has_padding = h > n_p2;
    for (size_t r = 0; r < n_p2; ++r) {
      const uint32_t flag = pp.p2_rows[r * 24 + 21];
      uint32_t next_ns;
      if (r + 1 < n_p2) next_ns = pp.p2_rows[(r + 1) * 24 + 22];
      else if (has_padding) next_ns = 1;
      else next_ns = pp.p2_rows[22];
      if (flag && next_ns) pp.read(pp.p2_rows[r * 24 + 20] / D);
    }
  }
  // ---- phase 2: out_ctl <- -1 (duplicate) or +ext_reads (:177-243)
  out.p2_rows = pp.p2_rows;
  for (size_t r = 0; r < n_p2; ++r)
    for (int j = 0; j < 2; ++j) {
      uint32_t& ctl = out.p2_rows[r * 24 + 17 + 2 * j];
      if (!ctl) continue;
      const uint32_t wid = out.p2_rows[r * 24 + 16 + 2 * j] / D;
      const bool dup = wid < pp.dup_p2.size() && pp.dup_p2[wid];
      ctl = dup ? neg1 : reads(wid);
    }
  // the width-32 table: phase 1 skips the arity-4 op types (batch_stark_prover.rs:121-127), phase 2 is the same
  out.p2w_rows = pp.p2w_rows;
  for (size_t r = 0; r < out.p2w_rows.size() / 48; ++r)
    for (int j = 0; j < 6; ++j) {
      uint32_t& ctl = out.p2w_rows[r * 48 + 33 + 2 * j];
      if (!ctl) continue;
      const uint32_t wid = out.p2w_rows[r * 48 + 32 + 2 * j] / D;
      const bool dup = wid < pp.dup_p2w.size() && pp.dup_p2w[wid];
      ctl = dup ? neg1 : reads(wid);
    }
  // ---- recompose_preprocess_for_op (recompose.rs:294-358)
  auto recompose_for_op = [&](std::vector<uint32_t> rows, const std::vector<bool>& dups, size_t rec_w) {
    for (size_t r = 0; r < rows.size() / rec_w; ++r) {
      uint32_t* row = &rows[rec_w * r];
      const uint32_t wid = row[0] / D;
      const bool dup = wid < dups.size() && dups[wid];
      row[1] = dup ? neg1 : reads(wid);
      // coefficient tuples: a hint output is created here with its read count, anything else is named with 0 (:341-352)
      for (size_t k = 2; k < rec_w; k += 2) {
        const uint32_t cw = row[k] / D;
        row[k + 1] = pp.hint_output_wids.count(cw) ? reads(cw) : 0u;
      }
    }
    return rows;
  };
  out.recompose_prep = recompose_for_op(pp.recompose, pp.dup_recompose, 2);
  out.recompose_coeff_prep = recompose_for_op(pp.recompose_coeff, pp.dup_recompose_coeff, 2 + 2 * D);
  if (out.recompose_prep.empty() && !out.recompose_coeff_prep.empty()) {
    out.recompose_prep.swap(out.recompose_coeff_prep);
    out.recompose_coeff_only = true;
  }
  // ---- primitive tables (common.rs:186-368)
  for (uint32_t idx : pp.prim_const) { out.const_prep.push_back(reads(idx / D)); out.const_prep.push_back(idx); }
  for (uint32_t idx : pp.prim_public) { out.public_prep.push_back(reads(idx / D)); out.public_prep.push_back(idx); }
  for (size_t i = 0; i < pp.prim_alu12.size() / 12; ++i) {
    const uint32_t* ch = &pp.prim_alu12[i * 12];
    auto reader_col = [&](uint32_t state, uint32_t idx) -> uint32_t {
      if (state == 1) return 1;
      if (state == 2) return (P - reads(idx / D)) % P;
      return 0;
    };
    const uint32_t mult_b = ch[9] ? reads(ch[5] / D) : neg1;
    const uint32_t mult_out = ch[11] ? reads(ch[7] / D) : neg1;
    const uint32_t row[13] = {neg1, ch[0], ch[1], ch[2], ch[3], ch[4], ch[5], ch[6], ch[7], mult_b, mult_out,
                              reader_col(ch[8], ch[4]), reader_col(ch[10], ch[6])};
    out.alu_prep13.insert(out.alu_prep13.end(), row, row + 13);
  }
  if (pp.prim_alu12.empty()) out.alu_prep13.assign(13, 0);  // dummy row (:283-286)
  out.ext_reads = pp.ext_reads;
  return out;
}

// ------------------------------------------------------------------------------------ runner
template <class FP>
struct RunInputs {
  std::vector<Fe4<FP>> public_values, private_values;
  std::map<uint32_t, std::array<Fe4<FP>, 2>> private_data;  // NonPrimitiveOpId -> sibling limbs
  std::map<uint32_t, std::array<Fe4<FP>, 6>> private_data_w32;  // width-32 Merkle rows: three sibling digests
};

template <class FP>
struct RunTraces {
  using F = Fe<FP>;
  using E = Fe4<FP>;
  std::vector<E> witness;
  std::vector<E> const_values, public_values;
  std::vector<std::array<E, 4>> alu_values;  // a, b, c, out
  struct P2Row { bool new_start, merkle_path, mmcs_bit, mmcs_ctl_enabled; F mmcs_index_sum; std::array<F, 16> input; };
  std::vector<P2Row> p2_rows;
  struct P2WRow { bool new_start, merkle_path, mmcs_bit, mmcs_bit2; F mmcs_index_sum; std::array<F, 32> input; };
  std::vector<P2WRow> p2w_rows;
  std::vector<std::array<F, 4>> recompose_values;
  std::vector<std::array<F, 4>> recompose_coeff_values;   // rows of the ops with aux = 1 when the circuit has both kinds
};

// CircuitRunner::run (runner.rs:195-253) for D = 4.
// p2w: the width-32 permutation (needed only by a circuit that holds COP_P2W ops)
template <class FP>
RunTraces<FP> run_circuit(const CircuitDesc& c, const Poseidon2<FP>& p2, const RunInputs<FP>& in, const Poseidon2W32<FP>* p2w = nullptr) {
  using F = Fe<FP>;
  using E = Fe4<FP>;
  RunTraces<FP> T;
  bool has_plain_recompose = false;   // the rows of `recompose/coeff` ops form a second table only next to a first one
  for (auto& op : c.ops) has_plain_recompose = has_plain_recompose || (op.kind == COP_RECOMPOSE && op.aux != 1);
  std::vector<E> w(c.witness_count);
  std::vector<bool> set(c.witness_count, false);
  auto wid_str = [](uint32_t x) { return "WitnessId(" + std::to_string(x) + ")"; };
  auto get = [&](uint32_t x) -> E {
    if (x >= w.size() || !set[x]) throw std::runtime_error("WitnessNotSet: " + wid_str(x));
    return w[x];
  };
  auto put = [&](uint32_t x, const E& v) {  // set_witness (:473-510)
    if (x >= w.size()) throw std::runtime_error("WitnessIdOutOfBounds: " + wid_str(x));
    if (set[x]) {
      if (w[x] != v) throw std::runtime_error("WitnessConflict: " + wid_str(x));
      return;
    }
    w[x] = v; set[x] = true;
  };
  if (in.public_values.size() != c.public_rows.size()) throw std::runtime_error("PublicInputLengthMismatch");
  if (in.private_values.size() != c.private_rows.size()) throw std::runtime_error("PrivateInputLengthMismatch");
  for (size_t i = 0; i < c.public_rows.size(); ++i) put(c.public_rows[i], in.public_values[i]);
  for (size_t i = 0; i < c.private_rows.size(); ++i) put(c.private_rows[i], in.private_values[i]);

  bool have_normal = false, have_merkle = false;
  std::array<E, 4> last_normal{}, last_merkle{};
  // PoseidonExecutionState is per op type: the width-32 table has its own chain state
  bool w_have_normal = false, w_have_merkle = false;
  std::array<E, 8> w_last_normal{}, w_last_merkle{};
  for (auto& op : c.ops) {
    switch (op.kind) {
      case COP_CONST: {
        const uint32_t* v = c.ext_of(op);
        if (op.ext_len != 4) throw std::runtime_error("const op: needs 4 coefficients");
        put(op.out, E(F(v[0]), F(v[1]), F(v[2]), F(v[3])));
        break;
      }
      case COP_PUBLIC:
        if (op.out >= set.size() || !set[op.out]) throw std::runtime_error("PublicInputNotSet: " + wid_str(op.out));
        break;
      case COP_ADD: case COP_MUL: {
        const E a = get(op.a);
        E b, o;
        if (op.b < set.size() && set[op.b]) {
          b = w[op.b];
          o = op.kind == COP_ADD ? a + b : a * b;
          put(op.out, o);
        } else {
          o = get(op.out);
          if (op.kind == COP_ADD) b = o - a;
          else {
            if (a == E::zero()) throw std::runtime_error("DivisionByZero");
            b = o * a.inv();
          }
          put(op.b, b);
        }
        T.alu_values.push_back({a, b, E::zero(), o});
        break;
      }
      case COP_BOOL: {
        const E a = get(op.a);
        put(op.out, a);
        T.alu_values.push_back({a, E::zero(), a, a});
        break;
      }
      case COP_MULADD: {
        const E a = get(op.a), b = get(op.b), ab = a * b;
        if (op.aux != NO_W) put(op.aux, ab);
        const E cv = op.c != NO_W ? get(op.c) : E::zero();
        const E o = ab + cv;
        put(op.out, o);
        T.alu_values.push_back({a, b, cv, o});
        break;
      }
      case COP_HORNER: {
        if (op.aux == NO_W || op.c == NO_W) throw std::runtime_error("HornerAcc requires acc and c");
        const E acc = get(op.aux), a = get(op.a), b = get(op.b), cv = get(op.c);
        const E o = acc * b + cv - a;
        put(op.out, o);
        T.alu_values.push_back({a, b, cv, o});
        break;
      }
      case COP_HINT_EXT: {
        if (op.ext_len != 4) throw std::runtime_error("ExtDecompositionHint: needs 4 outputs");
        const E v = get(op.a);
        const uint32_t* outs = c.ext_of(op);
        for (int i = 0; i < 4; ++i) put(outs[i], E(v.c[i]));
        break;
      }
      case COP_HINT_BIN: {
        if (op.ext_len > 31 * 4) throw std::runtime_error("BinaryDecompositionTooManyBits");
        const E v = get(op.a);
        const uint32_t* outs = c.ext_of(op);
        uint32_t o = 0;
        for (int k = 0; k < 4 && o < op.ext_len; ++k)
          for (int i = 0; i < 31 && o < op.ext_len; ++i) put(outs[o++], E(F((v.c[k].v >> i) & 1)));
        break;
      }
      case COP_P2: {
        const uint32_t* e = c.ext_of(op);
        const uint32_t n_out = e[6];
        const bool new_start = op.aux & 1, merkle = (op.aux >> 1) & 1;
        // private data only in Merkle mode (executor.rs:253-272)
        auto pd = in.private_data.find(op.a);
        if (pd != in.private_data.end() && !merkle)
          throw std::runtime_error("IncorrectNonPrimitiveOpPrivateData: private data provided for non-Merkle operation");
        // mmcs_bit (:283-338)
        bool bit = false;
        if (e[5] != NO_W) {
          const E v = get(e[5]);
          if (v == E::zero()) bit = false;
          else if (v == E::one()) bit = true;
          else throw std::runtime_error("IncorrectNonPrimitiveOpPrivateData: boolean mmcs_bit (0 or 1)");
        } else if (merkle) {
          throw std::runtime_error("IncorrectNonPrimitiveOpPrivateData: mmcs_bit must be provided when merkle_path=true");
        }
        // init_chain_state (:103-139)
        std::array<E, 4> st{};
        if (!new_start) {
          if (merkle) {
            if (!have_merkle) throw std::runtime_error("Poseidon2ChainMissingPreviousState");
            st[0] = last_merkle[0]; st[1] = last_merkle[1];
          } else {
            if (!have_normal) throw std::runtime_error("Poseidon2ChainMissingPreviousState");
            st = last_normal;
          }
        }
        // fill_sibling_data (:166-201), arity 2: capacity limbs
        if (merkle && pd != in.private_data.end()) { st[2] = pd->second[0]; st[3] = pd->second[1]; }
        // apply_witness_values (:207-219)
        for (int l = 0; l < 4; ++l)
          if (e[l] != NO_W) st[l] = get(e[l]);
        // apply_merkle_swap (:227-234)
        if (merkle && bit) { std::swap(st[0], st[2]); std::swap(st[1], st[3]); }
        std::array<F, 16> state;
        for (int l = 0; l < 4; ++l)
          for (int k = 0; k < 4; ++k) state[l * 4 + k] = st[l].c[k];
        typename RunTraces<FP>::P2Row row{};
        row.new_start = new_start; row.merkle_path = merkle; row.mmcs_bit = bit;
        row.input = state;
        if (e[4] != NO_W) {  // build_trace_row (:389-397) + trace.rs:209-217
          const E v = get(e[4]);
          if (v.c[1].v || v.c[2].v || v.c[3].v)
            throw std::runtime_error("IncorrectNonPrimitiveOpPrivateData: base field mmcs_index_sum");
          row.mmcs_index_sum = v.c[0];
          row.mmcs_ctl_enabled = true;
        }
        p2.permute(state);
        std::array<E, 4> outv;
        for (int l = 0; l < 4; ++l) outv[l] = E(state[l * 4], state[l * 4 + 1], state[l * 4 + 2], state[l * 4 + 3]);
        for (uint32_t l = 0; l < n_out; ++l)
          if (e[7 + l] != NO_W) put(e[7 + l], outv[l]);
        if (merkle) { last_merkle = outv; have_merkle = true; }
        else { last_normal = outv; have_normal = true; }
        T.p2_rows.push_back(row);
        break;
      }
      case COP_P2W: {
        if (!p2w) throw std::runtime_error("the circuit holds width-32 Poseidon2 ops: the width-32 permutation is needed");
        const uint32_t* e = c.ext_of(op);
        const uint32_t n_out = e[11];
        const bool new_start = op.aux & 1, merkle = (op.aux >> 1) & 1;
        auto pd = in.private_data_w32.find(op.a);
        if (pd != in.private_data_w32.end() && !merkle)
          throw std::runtime_error("IncorrectNonPrimitiveOpPrivateData: private data provided for non-Merkle operation");
        auto boolean = [&](uint32_t wid, const char* label) {   // resolve_boolean_witness (:305-338)
          if (wid == NO_W) {
            if (merkle) throw std::runtime_error(std::string("IncorrectNonPrimitiveOpPrivateData: ") + label + " must be provided when merkle_path=true");
            return false;
          }
          const E v = get(wid);
          if (v == E::zero()) return false;
          if (v == E::one()) return true;
          throw std::runtime_error(std::string("IncorrectNonPrimitiveOpPrivateData: boolean ") + label + " (0 or 1)");
        };
        const bool bit = boolean(e[9], "mmcs_bit"), bit2 = boolean(e[10], "mmcs_bit2");
        const int pos = (int)bit + 2 * (int)bit2;
        // init_chain_state (:103-139): a Merkle row of the arity-4 shape starts from zeros ...
        std::array<E, 8> st{};
        const bool have_prev = merkle ? w_have_merkle : w_have_normal;
        if (!new_start) {
          if (!have_prev) throw std::runtime_error("Poseidon2ChainMissingPreviousState");
          if (!merkle) st = w_last_normal;
        }
        // ... and place_arity4_running_hash (:141-160) writes the previous digest (capacity_ext = 2 limbs) into chunk pos
        if (merkle && !new_start) { st[2 * pos] = w_last_merkle[0]; st[2 * pos + 1] = w_last_merkle[1]; }
        // fill_sibling_data (:166-201): the chunks other than pos, ascending
        if (merkle && pd != in.private_data_w32.end()) {
          int written = 0;
          for (int chunk = 0; chunk < 4; ++chunk) {
            if (chunk == pos) continue;
            st[2 * chunk] = pd->second[written]; st[2 * chunk + 1] = pd->second[written + 1];
            written += 2;
          }
        }
        // apply_witness_values (:207-219); no swap on this shape (:227-234)
        for (int l = 0; l < 8; ++l)
          if (e[l] != NO_W) st[l] = get(e[l]);
        typename RunTraces<FP>::P2WRow row{};
        row.new_start = new_start; row.merkle_path = merkle; row.mmcs_bit = bit; row.mmcs_bit2 = bit2;
        row.mmcs_index_sum = F(0);   // inputs[width_ext] is empty (build_trace_row :389-397)
        for (int l = 0; l < 8; ++l)
          for (int k = 0; k < 4; ++k) row.input[l * 4 + k] = st[l].c[k];
        std::array<F, 32> state = row.input;
        p2w->permute(state);
        std::array<E, 8> outv;
        for (int l = 0; l < 8; ++l) outv[l] = E(state[l * 4], state[l * 4 + 1], state[l * 4 + 2], state[l * 4 + 3]);
        for (uint32_t l = 0; l < n_out; ++l)
          if (e[12 + l] != NO_W) put(e[12 + l], outv[l]);
        // update_chain_state (:462-491): a sponge row of the arity-4 shape seeds the Merkle chain too
        if (merkle) { w_last_merkle = outv; w_have_merkle = true; }
        else { w_last_normal = outv; w_have_normal = true; w_last_merkle = outv; w_have_merkle = true; }
        T.p2w_rows.push_back(row);
        break;
      }
      case COP_RECOMPOSE: {
        const uint32_t* ins = c.ext_of(op);
        std::array<F, 4> co;
        for (int i = 0; i < 4; ++i) co[i] = get(ins[i]).c[0];
        put(op.out, E(co[0], co[1], co[2], co[3]));
        (op.aux == 1 && has_plain_recompose ? T.recompose_coeff_values : T.recompose_values).push_back(co);
        break;
      }
      default: throw std::runtime_error("unknown op kind");
    }
  }
  // ALU-dedup rewrite (:199-216)
  {
    std::map<uint32_t, uint32_t> rw(c.rewrite.begin(), c.rewrite.end());
    for (auto& [dup, canon] : c.rewrite) {
      uint32_t cur = canon;
      for (auto it = rw.find(cur); it != rw.end(); it = rw.find(cur)) cur = it->second;
      if (cur < set.size() && set[cur]) put(dup, w[cur]);
    }
  }
  for (size_t i = 0; i < w.size(); ++i)
    if (!set[i]) throw std::runtime_error("WitnessNotSetForIndex: " + std::to_string(i));
  for (auto& op : c.ops) {
    if (op.kind == COP_CONST) {
      const uint32_t* v = c.ext_of(op);
      T.const_values.push_back(E(F(v[0]), F(v[1]), F(v[2]), F(v[3])));
    } else if (op.kind == COP_PUBLIC) {
      T.public_values.push_back(w[op.out]);
    }
  }
  if (T.alu_values.empty()) T.alu_values.push_back({E::zero(), E::zero(), E::zero(), E::zero()});  // tables/alu.rs:69-73
  T.witness = std::move(w);
  return T;
}

}  // namespace orc
