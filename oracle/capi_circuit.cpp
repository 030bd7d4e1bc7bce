// ORACLE (test infrastructure): extern "C" entry points for the circuit-side restatement
// (circuit.hpp): preprocessing of a flattened Circuit and the sequential CircuitRunner.
// Results are exposed as named flat arrays of canonical u32.
#include <cstring>
#include <map>
#include <memory>
#include <string>

#include "circuit.hpp"

using namespace orc;

extern "C" {
typedef struct orc_circuit_desc {
  uint32_t witness_count;
  size_t n_ops;     const COp* ops;
  size_t n_ext;     const uint32_t* ext;
  size_t n_public;  const uint32_t* public_rows;
  size_t n_private; const uint32_t* private_input_rows;
  size_t n_rewrite; const uint32_t* witness_rewrite;
} orc_circuit_desc;
void orc_set_error(const char* s);
}

namespace {
struct CircuitHandle {
  CircuitDesc c;
  std::map<std::string, std::vector<uint32_t>> arr;
};

template <class Fn>
int guard3(Fn&& fn) {
  try { fn(); return 0; } catch (const std::exception& e) { orc_set_error(e.what()); return -1; }
}

template <class FP>
void do_run(CircuitHandle& H, const uint32_t* rc, const uint32_t* pub, const uint32_t* priv, size_t n_pd,
            const uint32_t* pd_ids, const uint32_t* pd_sib, const uint32_t* w32_rc = nullptr, const uint32_t* w32_diag = nullptr,
            size_t n_pdw = 0, const uint32_t* pdw_ids = nullptr, const uint32_t* pdw_sib = nullptr) {
  using F = Fe<FP>;
  using E = Fe4<FP>;
  auto ef = [](const uint32_t* p) {
    for (int i = 0; i < 4; ++i)
      if (p[i] >= FP::P) throw std::runtime_error("non-canonical input");
    return E(F(p[0]), F(p[1]), F(p[2]), F(p[3]));
  };
  Poseidon2<FP> p2(rc);
  RunInputs<FP> in;
  for (size_t i = 0; i < H.c.public_rows.size(); ++i) in.public_values.push_back(ef(pub + 4 * i));
  for (size_t i = 0; i < H.c.private_rows.size(); ++i) in.private_values.push_back(ef(priv + 4 * i));
  for (size_t i = 0; i < n_pd; ++i) in.private_data[pd_ids[i]] = {ef(pd_sib + 8 * i), ef(pd_sib + 8 * i + 4)};
  for (size_t i = 0; i < n_pdw; ++i) {
    std::array<E, 6> sib;
    for (int l = 0; l < 6; ++l) sib[l] = ef(pdw_sib + 24 * i + 4 * l);
    in.private_data_w32[pdw_ids[i]] = sib;
  }
  std::unique_ptr<Poseidon2W32<FP>> p2w;
  if (w32_rc && w32_diag) p2w = std::make_unique<Poseidon2W32<FP>>(w32_rc, w32_diag);
  auto T = run_circuit<FP>(H.c, p2, in, p2w.get());
  auto put_e = [](std::vector<uint32_t>& d, const E& e) { for (int i = 0; i < 4; ++i) d.push_back(e.c[i].v); };
  auto& A = H.arr;
  for (const char* k : {"witness", "const_values", "public_values", "alu_values", "p2_inputs", "p2_flags",
                        "p2_mmcs_index_sum", "recompose_values", "recompose_coeff_values", "p2w_inputs", "p2w_flags",
                        "p2w_mmcs_index_sum"}) A[k].clear();
  for (auto& e : T.witness) put_e(A["witness"], e);
  for (auto& e : T.const_values) put_e(A["const_values"], e);
  for (auto& e : T.public_values) put_e(A["public_values"], e);
  for (auto& r : T.alu_values) for (auto& e : r) put_e(A["alu_values"], e);
  for (auto& r : T.p2_rows) {
    for (auto x : r.input) A["p2_inputs"].push_back(x.v);
    A["p2_flags"].insert(A["p2_flags"].end(), {r.new_start, r.merkle_path, r.mmcs_bit, r.mmcs_ctl_enabled});
    A["p2_mmcs_index_sum"].push_back(r.mmcs_index_sum.v);
  }
  for (auto& r : T.p2w_rows) {
    for (auto x : r.input) A["p2w_inputs"].push_back(x.v);
    A["p2w_flags"].insert(A["p2w_flags"].end(), {r.new_start, r.merkle_path, r.mmcs_bit, r.mmcs_bit2});
    A["p2w_mmcs_index_sum"].push_back(r.mmcs_index_sum.v);
  }
  for (auto& r : T.recompose_values) for (auto x : r) A["recompose_values"].push_back(x.v);
  for (auto& r : T.recompose_coeff_values) for (auto x : r) A["recompose_coeff_values"].push_back(x.v);
}
}  // namespace

extern "C" {

void* orc_circuit_new(const orc_circuit_desc* d) {
  auto* H = new CircuitHandle();
  H->c.witness_count = d->witness_count;
  H->c.ops.assign(d->ops, d->ops + d->n_ops);
  H->c.ext.assign(d->ext, d->ext + d->n_ext);
  H->c.public_rows.assign(d->public_rows, d->public_rows + d->n_public);
  H->c.private_rows.assign(d->private_input_rows, d->private_input_rows + d->n_private);
  for (size_t i = 0; i < d->n_rewrite; ++i)
    H->c.rewrite.push_back({d->witness_rewrite[2 * i], d->witness_rewrite[2 * i + 1]});
  return H;
}
void orc_circuit_free(void* h) { delete static_cast<CircuitHandle*>(h); }

// generate_preprocessed_columns::<D> + get_airs_and_degrees_with_prep; named arrays:
//   raw: prim_const, prim_public, prim_alu12, p2_rows_raw, recompose_raw, ext_reads_raw, hint_output_wids
//   final: const_prep, public_prep, alu_prep13, p2_prep_rows, recompose_prep, ext_reads,
//          p2_in_ctl, p2_input_indices, p2_out_ctl, p2_output_indices, p2_mmcs_index_sum_idx, p2_prep_flags
int orc_circuit_preprocess(void* h, uint32_t modulus, int D) {
  return guard3([&] {
    auto& H = *static_cast<CircuitHandle*>(h);
    Preprocessed pp = generate_preprocessed_columns(H.c, modulus, D);
    auto& A = H.arr;
    A["prim_const"] = pp.prim_const; A["prim_public"] = pp.prim_public; A["prim_alu12"] = pp.prim_alu12;
    A["p2_rows_raw"] = pp.p2_rows; A["recompose_raw"] = pp.recompose; A["ext_reads_raw"] = pp.ext_reads;
    A["hint_output_wids"].assign(pp.hint_output_wids.begin(), pp.hint_output_wids.end());
    CircuitPrep cp = get_airs_and_degrees_with_prep(pp);
    A["const_prep"] = cp.const_prep; A["public_prep"] = cp.public_prep; A["alu_prep13"] = cp.alu_prep13;
    A["p2_prep_rows"] = cp.p2_rows; A["recompose_prep"] = cp.recompose_prep; A["ext_reads"] = cp.ext_reads;
    A["recompose_coeff_prep"] = cp.recompose_coeff_prep;
    A["p2w_rows_raw"] = pp.p2w_rows; A["p2w_prep"] = cp.p2w_rows;
    for (const char* k : {"p2_in_ctl", "p2_input_indices", "p2_out_ctl", "p2_output_indices",
                          "p2_mmcs_index_sum_idx", "p2_prep_flags"}) A[k].clear();
    const uint32_t d = (uint32_t)D;
    for (size_t r = 0; r < cp.p2_rows.size() / 24; ++r) {
      const uint32_t* row = &cp.p2_rows[r * 24];
      for (int l = 0; l < 4; ++l) { A["p2_input_indices"].push_back(row[4 * l] / d); A["p2_in_ctl"].push_back(row[4 * l + 1]); }
      for (int l = 0; l < 2; ++l) { A["p2_output_indices"].push_back(row[16 + 2 * l] / d); A["p2_out_ctl"].push_back(row[17 + 2 * l]); }
      A["p2_mmcs_index_sum_idx"].push_back(row[20] / d);
      A["p2_prep_flags"].insert(A["p2_prep_flags"].end(), {row[22], row[23], row[21]});
    }
  });
}

int orc_circuit_run(void* h, int field, const uint32_t* rc, const uint32_t* public_values,
                    const uint32_t* private_values, size_t n_private_data, const uint32_t* pd_op_ids,
                    const uint32_t* pd_siblings) {
  return guard3([&] {
    auto& H = *static_cast<CircuitHandle*>(h);
    if (field == 0) do_run<KoalaBear>(H, rc, public_values, private_values, n_private_data, pd_op_ids, pd_siblings);
    else if (field == 1) do_run<BabyBear>(H, rc, public_values, private_values, n_private_data, pd_op_ids, pd_siblings);
    else throw std::runtime_error("unknown field id");
  });
}

// the same for a circuit that holds width-32 Poseidon2 ops (COP_P2W): the constants of the width-32 permutation and the
// private data of its Merkle rows (24 values = three sibling digests per op id)
int orc_circuit_run_w32(void* h, int field, const uint32_t* rc, const uint32_t* w32_rc, const uint32_t* w32_diag,
                        const uint32_t* public_values, const uint32_t* private_values, size_t n_private_data,
                        const uint32_t* pd_op_ids, const uint32_t* pd_siblings, size_t n_private_data_w32,
                        const uint32_t* pdw_op_ids, const uint32_t* pdw_siblings) {
  return guard3([&] {
    auto& H = *static_cast<CircuitHandle*>(h);
    if (field == 0) do_run<KoalaBear>(H, rc, public_values, private_values, n_private_data, pd_op_ids, pd_siblings, w32_rc, w32_diag,
                                      n_private_data_w32, pdw_op_ids, pdw_siblings);
    else if (field == 1) do_run<BabyBear>(H, rc, public_values, private_values, n_private_data, pd_op_ids, pd_siblings, w32_rc, w32_diag,
                                          n_private_data_w32, pdw_op_ids, pdw_siblings);
    else throw std::runtime_error("unknown field id");
  });
}

int orc_circuit_get(void* h, const char* name, const uint32_t** ptr, size_t* len) {
  auto& H = *static_cast<CircuitHandle*>(h);
  auto it = H.arr.find(name);
  if (it == H.arr.end()) return -1;
  *ptr = it->second.data();
  *len = it->second.size();
  return 0;
}

}  // extern "C"
