// Host-side C++ mirror of the reference's interface for the prove_next_layer path, over the C ABI
// of p3r.h.  Header-only, C++17, no HIP or torch types: link against libp3r_hip.so.
//
// The reference is compiled Rust; a Rust caller binds p3r.h directly.  This header
// is the same surface for C++ callers, with the reference's names, argument meaning and error
// behaviour (errors are exceptions carrying the reference's error text):
//   TablePacking                 circuit-prover/src/batch_stark_prover/packing.rs:10-161
//   Traces<EF> (flattened)       circuit/src/tables/mod.rs:49-62
//   CircuitProverData            circuit-prover/src/batch_stark_prover.rs:314-341
//   NonPrimitiveTableEntry       :272-290        BatchStarkProof  :610-636 (+ validate :666-681)
//   BatchStarkProver::{prove_all_tables, verify_all_tables}   :1203-1268
//   Circuit<EF> / CircuitRunner  circuit/src/circuit.rs:152-181, circuit/src/tables/runner.rs:22-253
//   FriRecursionBackend          recursion/src/backend/fri.rs:113-128
//   ProveNextLayerParams, RecursionInput, RecursionOutput, NextLayerPrepCache,
//   build_next_layer_prep, prove_next_layer                   recursion/src/recursion.rs:96-139,221-234,295-298,342-502
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "p3r.h"

namespace p3r {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

enum class Field : uint32_t { KoalaBear = P3R_FIELD_KOALA_BEAR, BabyBear = P3R_FIELD_BABY_BEAR };
inline uint32_t modulus(Field f) { return f == Field::KoalaBear ? 0x7f000001u : 0x78000001u; }
inline uint32_t binomial_w(Field f) { return f == Field::KoalaBear ? 3u : 11u; }

// FRI / PCS parameters (recursion/examples/common/mod.rs:464-486)
struct FriParams {
  uint32_t log_blowup = 2, max_log_arity = 2, cap_height = 0, log_final_poly_len = 5, commit_pow_bits = 0,
           query_pow_bits = 15, num_queries = 54;
  // arity of the PCS's MMCS: 2 = MyMmcs (width-16 permutation), 4 = MyMmcsArity4 (width-32 permutation, W16 challenger:
  // recursion/examples/recursive_aggregation.rs:902-1046; cap_height must be 0)
  uint32_t mmcs_arity = 2;
  // ZK: the PCS is HidingFriPcs with `num_random_codewords` random codewords (create_config_zk,
  // recursion/examples/common/mod.rs:511-553: two codewords); `zk_key` is the 256-bit key of its generator (the library
  // mixes operating-system entropy into it unless `zk_deterministic` asks for reproducible proofs: tests, replay)
  // the library's built-in width-32 constants are self-generated (P3R_EXT_UNPINNED_W32_DEFAULTS, p3r.h): using the
  // arity-4 MMCS / the width-32 table without the caller's constants must be asked for
  bool allow_unpinned_w32_defaults = false;
  bool zk = false;
  uint32_t num_random_codewords = 2;
  std::array<uint32_t, 8> zk_key{};
  bool zk_deterministic = false;
  // MerkleTreeHidingMmcs for the input and FRI commit-phase MMCSs (recursion/tests/zk_hiding_mmcs.rs: 4); 0 = the plain MMCS
  uint32_t mmcs_salt_elems = 0;
};

// ext_degree: the circuit extension degree of the traces - 4, or 5 for KoalaBear circuits over the quintic trinomial
// extension (primitive tables, at the prove_all_tables boundary; include/p3r.h)
inline p3r_config make_config(Field field, const FriParams& p, int device = 0, const std::vector<uint32_t>* rc = nullptr,
                              uint32_t ext_degree = 4, uint32_t ext_w = 0 /* W of x^D = W for ext_degree 2 / 6 / 8 */,
                              uint32_t challenge_degree = 4 /* 5: KoalaBear's quintic challenge field */) {
  p3r_config c{};
  c.abi_version = P3R_ABI_VERSION;
  c.field = (uint32_t)field;
  c.ext_degree = ext_degree;
  c.ext_w = ext_w;
  c.challenge_degree = challenge_degree;
  c.log_blowup = p.log_blowup; c.max_log_arity = p.max_log_arity; c.cap_height = p.cap_height;
  c.log_final_poly_len = p.log_final_poly_len; c.commit_pow_bits = p.commit_pow_bits;
  c.query_pow_bits = p.query_pow_bits; c.num_queries = p.num_queries;
  c.mmcs_arity = p.mmcs_arity;
  if (p.allow_unpinned_w32_defaults) c.ext_choices |= P3R_EXT_UNPINNED_W32_DEFAULTS;
  c.zk = p.zk ? 1u : 0u;
  c.num_random_codewords = p.zk ? p.num_random_codewords : 0u;
  for (int i = 0; i < 8; ++i) c.zk_key[i] = p.zk_key[i];
  if (p.zk_deterministic) c.ext_choices |= P3R_EXT_ZK_DETERMINISTIC;
  c.mmcs_salt_elems = p.mmcs_salt_elems;
  c.device = device;
  if (rc) { c.poseidon2_rc = rc->data(); c.poseidon2_rc_len = (uint32_t)rc->size(); }
  return c;
}

// One per GPU, not thread-safe, one call in flight (the reference's RecursionOutput is !Send).
class Context {
 public:
  Context(Field field, const FriParams& fri, int device = 0, std::vector<uint32_t> poseidon2_rc = {}, uint32_t ext_degree = 4,
          uint32_t ext_w = 0, uint32_t challenge_degree = 4)
      : field_(field), fri_(fri), rc_(std::move(poseidon2_rc)) {
    cfg_ = make_config(field, fri, device, rc_.empty() ? nullptr : &rc_, ext_degree, ext_w, challenge_degree);
    h_ = p3r_create(&cfg_);
    if (!h_) throw Error(P3R_ENODEV, p3r_last_error(nullptr));
  }
  ~Context() { if (h_) p3r_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  p3r_ctx* raw() const { return h_; }
  const p3r_config& config() const { return cfg_; }
  Field field() const { return field_; }
  uint32_t ext_degree() const { return cfg_.ext_degree; }
  const FriParams& fri() const { return fri_; }
  void check(int rc) const { if (rc != P3R_OK) throw Error(rc, p3r_last_error(h_)); }
  template <class T> T* ptr(T* p) const { if (!p) throw Error(P3R_EINVAL, p3r_last_error(h_)); return p; }
  void sync() const { check(p3r_sync(h_)); }

 private:
  Field field_;
  FriParams fri_;
  std::vector<uint32_t> rc_;
  p3r_config cfg_{};
  p3r_ctx* h_ = nullptr;
};

struct TablePacking {
  uint32_t public_lanes = 1, alu_lanes = 3, horner_packed_steps = 4, recompose_lanes = 1, min_trace_height = 1;
  static TablePacking create(uint32_t public_lanes, uint32_t alu_lanes) {  // TablePacking::new
    TablePacking t;
    t.public_lanes = public_lanes; t.alu_lanes = alu_lanes;
    return t;
  }
  // FRI needs log_height > log_final_poly_len + log_blowup (packing.rs:100-106)
  TablePacking& with_fri_params(uint32_t log_final_poly_len, uint32_t log_blowup) {
    min_trace_height = 1u << (log_final_poly_len + log_blowup + 1);
    return *this;
  }
  void validate() const {  // packing.rs:140-161
    if (!public_lanes) throw Error(P3R_EINVAL, "ZeroLanes(\"public_lanes\")");
    if (!alu_lanes) throw Error(P3R_EINVAL, "ZeroLanes(\"alu_lanes\")");
    if (!recompose_lanes) throw Error(P3R_EINVAL, "ZeroNpoLanes(recompose)");
    if (!min_trace_height || (min_trace_height & (min_trace_height - 1)))
      throw Error(P3R_EINVAL, "BadMinTraceHeight(" + std::to_string(min_trace_height) + ")");
    if (horner_packed_steps < 2) throw Error(P3R_EINVAL, "BadHornerPackedSteps(" + std::to_string(horner_packed_steps) + ")");
  }
};

// Flattened Traces<EF>, D = 4, canonical u32.
struct Traces {
  std::vector<uint32_t> const_values, public_values;  // n x D (D = the context's ext_degree)
  std::vector<uint32_t> alu_values;                   // n x 4D: [a, b, c, out]
  std::vector<uint32_t> p2_input_values;              // n x 16
  std::vector<uint8_t> p2_new_start, p2_merkle_path, p2_mmcs_bit;
  std::vector<uint32_t> p2_mmcs_index_sum;
  std::vector<uint32_t> recompose_values;             // n x D
  std::vector<uint32_t> recompose_coeff_values;       // n x D: rows of the second Recompose table (`recompose/coeff`)
};

// Per-op preprocessed data as get_airs_and_degrees_with_prep leaves it (see p3r_layer_desc).
struct CircuitPrep {
  std::vector<uint32_t> const_prep, public_prep, alu_prep13, recompose_prep;
  std::vector<uint8_t> p2_new_start, p2_merkle_path, p2_mmcs_ctl_enabled, p2_in_ctl;
  std::vector<uint32_t> p2_input_indices, p2_out_ctl, p2_output_indices, p2_mmcs_index_sum_idx;
  // IL x OL = 4 x 2 limbs per row under ext_degree 4; 16 x 8 elements (the compact-D1 table) under ext_degree 5,
  // where p2_absorb_len holds the sponge length tags (empty: zeros); include/p3r.h
  std::vector<uint8_t> p2_absorb_len;
  // the "recompose/coeff" table (per-coefficient bus tuples): recompose_prep is n x (2 + 2 D)
  bool recompose_coeff_lookups = false;
  // a layer holding BOTH Recompose tables (recompose_table_provers(lanes, true)): recompose_prep is the plain kind,
  // this the `recompose/coeff` table, n x (2 + 2 D)
  std::vector<uint32_t> recompose_coeff_prep;
};

struct Circuit {  // flattened Circuit<EF>
  uint32_t witness_count = 0;
  std::vector<p3r_op> ops;
  std::vector<uint32_t> ext, public_rows, private_input_rows, witness_rewrite /* pairs */;
};

struct CircuitInputs {
  std::vector<uint32_t> public_values, private_values;                  // x D each (the context's ext_degree)
  std::vector<uint32_t> private_data_op_ids, private_data_siblings;     // siblings x 8
  // private data of width-32 Merkle rows (P3R_OP_POSEIDON2_W32_PERM): three sibling digests, siblings x 24
  std::vector<uint32_t> private_data_w32_op_ids, private_data_w32_siblings;
};

class CircuitProverData {
 public:
  CircuitProverData(const Context& ctx, const CircuitPrep& prep, const TablePacking& packing) : ctx_(&ctx), packing_(packing) {
    packing.validate();
    p3r_layer_desc d{};
    d.counts.n_const = prep.const_prep.size() / 2; d.counts.n_public = prep.public_prep.size() / 2;
    d.counts.n_alu = prep.alu_prep13.size() / 13; d.counts.n_p2 = prep.p2_new_start.size();
    const size_t rec_w = 2 + (prep.recompose_coeff_lookups ? 2 * ctx.ext_degree() : 0);
    if (prep.recompose_prep.size() % rec_w) throw Error(P3R_EINVAL, "recompose_prep must be n x " + std::to_string(rec_w));
    d.counts.n_recompose = prep.recompose_prep.size() / rec_w;
    d.recompose_coeff_lookups = prep.recompose_coeff_lookups;
    const size_t rec2_w = 2 + 2 * ctx.ext_degree();
    if (prep.recompose_coeff_prep.size() % rec2_w) throw Error(P3R_EINVAL, "recompose_coeff_prep must be n x " + std::to_string(rec2_w));
    d.counts.n_recompose_coeff = prep.recompose_coeff_prep.size() / rec2_w;
    d.recompose_coeff_prep = prep.recompose_coeff_prep.data();
    d.public_lanes = packing.public_lanes; d.alu_lanes = packing.alu_lanes;
    d.horner_packed_steps = packing.horner_packed_steps; d.recompose_lanes = packing.recompose_lanes;
    d.min_trace_height = packing.min_trace_height;
    d.const_prep = prep.const_prep.data(); d.public_prep = prep.public_prep.data();
    d.alu_prep13 = prep.alu_prep13.data(); d.recompose_prep = prep.recompose_prep.data();
    d.p2_new_start = prep.p2_new_start.data(); d.p2_merkle_path = prep.p2_merkle_path.data();
    d.p2_mmcs_ctl_enabled = prep.p2_mmcs_ctl_enabled.data(); d.p2_in_ctl = prep.p2_in_ctl.data();
    d.p2_input_indices = prep.p2_input_indices.data(); d.p2_out_ctl = prep.p2_out_ctl.data();
    d.p2_output_indices = prep.p2_output_indices.data(); d.p2_mmcs_index_sum_idx = prep.p2_mmcs_index_sum_idx.data();
    const size_t il = ctx.ext_degree() == 4 ? 4 : 16, ol = il / 2;
    if (prep.p2_in_ctl.size() != d.counts.n_p2 * il || prep.p2_input_indices.size() != d.counts.n_p2 * il ||
        prep.p2_out_ctl.size() != d.counts.n_p2 * ol || prep.p2_output_indices.size() != d.counts.n_p2 * ol ||
        (!prep.p2_absorb_len.empty() && prep.p2_absorb_len.size() != d.counts.n_p2))
      throw Error(P3R_EINVAL, "Poseidon2 CTL arrays do not match the row count for ext_degree " + std::to_string(ctx.ext_degree()));
    if (!prep.p2_absorb_len.empty()) d.p2_absorb_len = prep.p2_absorb_len.data();
    rows_ = d.counts;
    preprocessed_commitment.resize(size_t(8) << ctx.fri().cap_height);
    owned_ = ctx.ptr(p3r_layer_create(ctx.raw(), &d, preprocessed_commitment.data()));
    layer_ = owned_;
    read_shape();
  }
  // view of the data a prepared circuit owns
  CircuitProverData(const Context& ctx, const p3r_layer* borrowed, const TablePacking& packing, p3r_layer_desc_counts rows,
                    std::vector<uint32_t> commitment)
      : preprocessed_commitment(std::move(commitment)), ctx_(&ctx), packing_(packing), rows_(rows), layer_(borrowed) {
    read_shape();
  }
  ~CircuitProverData() { if (owned_) p3r_layer_free(ctx_->raw(), owned_); }
  CircuitProverData(const CircuitProverData&) = delete;
  CircuitProverData& operator=(const CircuitProverData&) = delete;
  const p3r_layer* raw() const { return layer_; }
  bool recompose_coeff_lookups = false;
  const TablePacking& packing() const { return packing_; }
  const TablePacking& effective_packing() const { return effective_; }  // reduce_lanes_if_dummy applied
  const p3r_layer_desc_counts& rows() const { return rows_; }
  std::vector<uint32_t> preprocessed_commitment;  // (1 << cap_height) x 8, canonical
  std::array<size_t, 5> table_heights{};          // 0 = table absent from the batch
  size_t recompose_coeff_height = 0;              // the second Recompose table (`recompose/coeff` next to `recompose`)

 private:
  void read_shape() {
    ctx_->check(p3r_layer_table_heights(layer_, table_heights.data()));
    ctx_->check(p3r_layer_recompose_coeff_height(layer_, &recompose_coeff_height));
    uint32_t kind = 0;
    ctx_->check(p3r_layer_recompose_kind(layer_, &kind));
    recompose_coeff_lookups = kind != 0;   // the table at position 4 is `recompose/coeff`
    effective_ = packing_;
    ctx_->check(p3r_layer_effective_lanes(layer_, &effective_.public_lanes, &effective_.alu_lanes));
  }
  const Context* ctx_;
  TablePacking packing_, effective_;
  p3r_layer_desc_counts rows_{};
  p3r_layer* owned_ = nullptr;
  const p3r_layer* layer_ = nullptr;
};

struct NonPrimitiveTableEntry {
  std::string op_type;
  size_t rows = 0, lanes = 1;
  std::vector<uint32_t> public_values;
  uint32_t air_variant = 0;  // AirVariant::Baseline
};

struct BatchStarkProof {
  std::vector<uint8_t> proof;  // postcard bytes of the inner BatchProof<SC>
  TablePacking table_packing;  // the EFFECTIVE packing (batch_stark_prover.rs:1617-1622)
  std::array<size_t, 3> rows{};
  uint32_t alu_variant = 1;    // AirVariant::Optimized (:1106-1114)
  uint32_t ext_degree = 4;
  std::optional<uint32_t> w_binomial;
  bool alu_quintic_trinomial = false;
  std::vector<NonPrimitiveTableEntry> non_primitives;
  std::vector<uint32_t> preprocessed_commitment;  // stark_common
  std::vector<uint32_t> preprocessed_widths, degree_bits;
  bool montgomery_field_encoding = true;
  uint32_t modulus = 0;

  void validate() const {  // batch_stark_prover.rs:666-681
    switch (ext_degree) { case 1: case 2: case 4: case 5: case 6: case 8: break;
      default: throw Error(P3R_EINVAL, "UnsupportedExtDegree(" + std::to_string(ext_degree) + ")"); }
    table_packing.validate();
    for (auto& e : non_primitives) if (!e.lanes) throw Error(P3R_EINVAL, "ZeroNpoLanes(" + e.op_type + ")");
  }
  // the proved tables in instance order
  std::vector<p3r_air_desc> airs() const {
    std::vector<p3r_air_desc> a = {{P3R_AIR_CONST, 1, 2, 0}, {P3R_AIR_PUBLIC, table_packing.public_lanes, 2, 0},
                                   {P3R_AIR_ALU, table_packing.alu_lanes, table_packing.horner_packed_steps, 0}};
    for (auto& e : non_primitives) {
      const bool p2 = e.op_type.rfind("poseidon2_perm/", 0) == 0;
      const bool d1 = e.op_type.size() >= 7 && e.op_type.compare(e.op_type.size() - 7, 7, "_d1_w16") == 0;
      const bool w32 = e.op_type.size() >= 7 && e.op_type.compare(e.op_type.size() - 7, 7, "_d4_w32") == 0;
      if (p2 && w32 && ext_degree == 4) a.push_back({P3R_AIR_POSEIDON2_W32, 1, 2, 0});   // the table of the arity-4 MMCS rows
      else if (p2 && (ext_degree == 4 || d1)) a.push_back({P3R_AIR_POSEIDON2, 1, 2, 0});
      else if (e.op_type == "recompose") a.push_back({P3R_AIR_RECOMPOSE, (uint32_t)e.lanes, 2, 0});
      else if (e.op_type == "recompose/coeff") a.push_back({P3R_AIR_RECOMPOSE, (uint32_t)e.lanes, 2, 1});
      else throw Error(P3R_EUNSUPPORTED, "MissingTableProver(" + e.op_type + ")");
    }
    return a;
  }
  // Inverse of to_postcard: one pass of the native parser (p3r_batch_stark_proof_parse: framing of the inner
  // BatchProof, every field element in range, the metadata fields and the rules of validate()) - what a node of the
  // aggregation tree runs on a child that arrived from another process.
  // challenge_degree 5: a proof over KoalaBear's quintic challenge field (five words per extension element).
  // zk: the proof is a hiding PCS's (FriParams::zk; P3R_PROOF_ZK): its opening proof is the tuple (random opened values, FriProof).
  static BatchStarkProof from_postcard(const std::vector<uint8_t>& data, Field field, bool montgomery_field_encoding = true,
                                       uint32_t challenge_degree = 4, bool zk = false, bool salted = false) {
    p3r_batch_stark_meta m;
    char err[256] = {0};
    const int rc = p3r_batch_stark_proof_parse((uint32_t)field, data.data(), data.size(),
                                               (montgomery_field_encoding ? 0 : P3R_PROVE_CANONICAL_FIELD_ENCODING) |
                                                   (challenge_degree == 5 ? P3R_PROOF_QUINTIC_CHALLENGE : 0) | (zk ? P3R_PROOF_ZK : 0) |
                                                   (salted ? P3R_PROOF_SALTED : 0),
                                               nullptr, &m, err, sizeof err);
    if (rc != P3R_OK) throw Error(rc, err);
    BatchStarkProof p;
    p.proof.assign(data.begin(), data.begin() + m.proof_len);
    p.table_packing.public_lanes = m.public_lanes; p.table_packing.alu_lanes = m.alu_lanes;
    p.table_packing.min_trace_height = m.min_trace_height; p.table_packing.horner_packed_steps = m.horner_packed_steps;
    for (int i = 0; i < 3; ++i) p.rows[i] = (size_t)m.rows[i];
    p.alu_variant = m.alu_variant; p.ext_degree = m.ext_degree;
    if (m.has_w_binomial) p.w_binomial = m.w_binomial;
    p.alu_quintic_trinomial = m.alu_quintic_trinomial != 0;
    for (uint32_t i = 0; i < m.n_non_primitives; ++i) {
      const p3r_npo_table_entry& e = m.non_primitives[i];
      NonPrimitiveTableEntry n;
      n.op_type = e.op_type; n.rows = (size_t)e.rows; n.lanes = e.lanes; n.air_variant = e.air_variant;
      n.public_values.assign(e.public_values, e.public_values + e.n_public_values);
      p.non_primitives.push_back(std::move(n));
    }
    // recompose lanes: the first Recompose entry of either kind, else TablePacking.npo_lanes (the rule of prover.py)
    bool have_lanes = false;
    for (const auto& n : p.non_primitives)
      if (!have_lanes && (n.op_type == "recompose" || n.op_type == "recompose/coeff")) {
        p.table_packing.recompose_lanes = n.lanes;
        have_lanes = true;
      }
    for (const char* name : {"recompose", "recompose/coeff"})
      for (uint32_t i = 0; i < m.n_npo_lanes && !have_lanes; ++i)
        if (std::string(m.npo_lanes[i].op_type) == name) {
          p.table_packing.recompose_lanes = m.npo_lanes[i].lanes;
          have_lanes = true;
        }
    if (m.has_stark_common) {
      p.preprocessed_commitment.assign(m.commitment, m.commitment + 8 * m.cap_len);
      p.preprocessed_widths.assign(m.preprocessed_widths, m.preprocessed_widths + m.n_instances);
      p.degree_bits.assign(m.degree_bits, m.degree_bits + m.n_instances);
    }
    p.montgomery_field_encoding = montgomery_field_encoding;
    p.modulus = field == Field::KoalaBear ? 0x7f000001u : 0x78000001u;
    return p;
  }
  // postcard bytes of the whole BatchStarkProof<SC> (serde derives of :610-636, packing.rs:9-27,
  // RowCounts :459-460, NonPrimitiveTableEntry :272-290, SerializedStarkCommon :495-511)
  std::vector<uint8_t> to_postcard() const {
    std::vector<uint8_t> out = proof;
    auto varint = [&](uint64_t v) { while (v >= 0x80) { out.push_back((uint8_t)(v | 0x80)); v >>= 7; } out.push_back((uint8_t)v); };
    auto str = [&](const std::string& s) { varint(s.size()); out.insert(out.end(), s.begin(), s.end()); };
    auto fe = [&](uint32_t x) { varint(montgomery_field_encoding ? (uint32_t)(((uint64_t)x << 32) % modulus) : x); };
    varint(table_packing.public_lanes); varint(table_packing.alu_lanes);
    size_t n_npo = 0;
    for (auto& e : non_primitives) n_npo += e.lanes != 1;
    varint(n_npo);
    for (auto& e : non_primitives) if (e.lanes != 1) { str(e.op_type); varint(e.lanes); }
    varint(table_packing.min_trace_height); varint(table_packing.horner_packed_steps);
    for (size_t r : rows) varint(r ? r : 1);
    varint(alu_variant); varint(ext_degree);
    if (w_binomial) { out.push_back(1); fe(*w_binomial); } else out.push_back(0);
    out.push_back(alu_quintic_trinomial);
    varint(non_primitives.size());
    for (auto& e : non_primitives) {
      str(e.op_type); varint(e.rows); varint(e.lanes);
      varint(e.public_values.size());
      for (uint32_t v : e.public_values) fe(v);
      varint(e.air_variant);
    }
    if (preprocessed_commitment.empty()) { out.push_back(0); return out; }
    out.push_back(1);
    varint(preprocessed_commitment.size() / 8);
    for (uint32_t v : preprocessed_commitment) fe(v);
    varint(preprocessed_widths.size());
    for (size_t i = 0; i < preprocessed_widths.size(); ++i) { out.push_back(1); varint(i); varint(preprocessed_widths[i]); varint(degree_bits[i]); }
    varint(preprocessed_widths.size());
    for (size_t i = 0; i < preprocessed_widths.size(); ++i) varint(i);
    return out;
  }
};

// verify_batch behind verify_all_tables: host code, no device needed.
inline void verify_all_tables(const p3r_config& cfg, const BatchStarkProof& proof) {
  proof.validate();
  if (proof.preprocessed_commitment.empty()) throw Error(P3R_EINVAL, "proof carries no preprocessed commitment (stark_common)");
  // the proof's extension metadata must be the verifier's (batch_stark_prover.rs:1245-1263)
  if (proof.ext_degree != cfg.ext_degree)
    throw Error(P3R_EINVAL, "ExtDegreeMismatch: proof has ext_degree " + std::to_string(proof.ext_degree) +
                                ", the verifier expects " + std::to_string(cfg.ext_degree));
  // EF = BinomialExtensionField<F, 4>: the field's W; EF = QuinticTrinomialExtensionField<F>: no W, the trinomial flag
  const bool quintic = cfg.ext_degree == 5;
  const uint32_t want_w = cfg.field == P3R_FIELD_KOALA_BEAR ? 3u : 11u;
  // D = 1 (the base field) and the quintic trinomial extension have no W; D = 2 / 6 / 8 carry the verifier's ext_w
  const bool generic = cfg.ext_degree == 2 || cfg.ext_degree == 6 || cfg.ext_degree == 8;
  const bool has_w = cfg.ext_degree == 4 || generic;
  if (has_w ? (!proof.w_binomial || *proof.w_binomial != (generic ? cfg.ext_w : want_w)) : proof.w_binomial.has_value())
    throw Error(P3R_EINVAL, "BinomialWMismatch");
  if (proof.alu_quintic_trinomial != quintic) throw Error(P3R_EINVAL, "QuinticReductionMismatch");
  auto airs = proof.airs();
  if (proof.degree_bits.size() != airs.size() || proof.preprocessed_widths.size() != airs.size())
    throw Error(P3R_EINVAL, "InvalidProofShape: AIR list and stark_common metadata differ in length");
  std::vector<uint32_t> degree_bits(proof.degree_bits.begin(), proof.degree_bits.end());
  char err[512] = {0};
  int rc = p3r_verify_batch(&cfg, airs.data(), airs.size(), proof.preprocessed_commitment.data(), degree_bits.data(),
                            proof.proof.data(),
                            proof.proof.size(), proof.montgomery_field_encoding ? 0 : P3R_PROVE_CANONICAL_FIELD_ENCODING, err,
                            sizeof err);
  if (rc != P3R_OK) throw Error(rc, std::string("Verify(") + err + ")");
}

namespace detail {
template <class Fn>
std::vector<uint8_t> proof_call(const Context& ctx, Fn&& fn) {
  std::vector<uint8_t> buf(size_t(1) << 20);
  size_t n = 0;
  int rc = fn(buf.data(), buf.size(), &n);
  // the library keeps a proof that did not fit (it is not recomputed: under zk a second call makes ANOTHER proof)
  if (rc == P3R_EBUFFER && n > buf.size()) { buf.resize(n); rc = p3r_take_proof(ctx.raw(), buf.data(), buf.size(), &n); }
  ctx.check(rc);
  buf.resize(n);
  return buf;
}
inline p3r_traces traces_struct(const Traces& t, uint32_t d = 4) {
  p3r_traces s{};
  if (t.const_values.size() % d || t.public_values.size() % d || t.alu_values.size() % (4 * d) || t.recompose_values.size() % d)
    throw Error(P3R_EINVAL, "Traces values are not n x D / n x 4D for ext_degree " + std::to_string(d));
  s.n_const = t.const_values.size() / d; s.const_values = t.const_values.data();
  s.n_public = t.public_values.size() / d; s.public_values = t.public_values.data();
  s.n_alu = t.alu_values.size() / (4 * d); s.alu_values = t.alu_values.data();
  s.p2.n = t.p2_input_values.size() / 16; s.p2.input_values = t.p2_input_values.data();
  s.p2.new_start = t.p2_new_start.data(); s.p2.merkle_path = t.p2_merkle_path.data(); s.p2.mmcs_bit = t.p2_mmcs_bit.data();
  s.p2.mmcs_index_sum = t.p2_mmcs_index_sum.data();
  s.n_recompose = t.recompose_values.size() / d; s.recompose_values = t.recompose_values.data();
  if (t.recompose_coeff_values.size() % d) throw Error(P3R_EINVAL, "recompose_coeff_values must hold n x D values");
  s.n_recompose_coeff = t.recompose_coeff_values.size() / d; s.recompose_coeff_values = t.recompose_coeff_values.data();
  return s;
}
}  // namespace detail

// Traces kept in HBM (uploaded, or produced there by the device CircuitRunner).
class ResidentTraces {
 public:
  ResidentTraces(const Context& ctx, p3r_dtraces* h) : ctx_(&ctx), h_(h) {}
  ResidentTraces(const Context& ctx, const CircuitProverData& cpd, const Traces& t) : ctx_(&ctx) {
    p3r_traces s = detail::traces_struct(t, ctx.ext_degree());
    h_ = ctx.ptr(p3r_traces_upload(ctx.raw(), cpd.raw(), &s));
  }
  ~ResidentTraces() { if (h_) p3r_traces_free(ctx_->raw(), h_); }
  ResidentTraces(ResidentTraces&& o) noexcept : ctx_(o.ctx_), h_(o.h_) { o.h_ = nullptr; }
  ResidentTraces(const ResidentTraces&) = delete;
  const p3r_dtraces* raw() const { return h_; }
  std::vector<uint32_t> download(const CircuitProverData& cpd, p3r_traces_array which, size_t len) const {
    std::vector<uint32_t> out(len);
    ctx_->check(p3r_dtraces_get(ctx_->raw(), cpd.raw(), h_, which, out.data(), len));
    return out;
  }

 private:
  const Context* ctx_;
  p3r_dtraces* h_ = nullptr;
};

// A circuit prepared once: preprocessed columns + commitment + the levelised execution schedule.
class PreparedCircuit {
 public:
  PreparedCircuit(const Context& ctx, Circuit circuit, const TablePacking& packing) : ctx_(&ctx), circuit_(std::move(circuit)) {
    packing.validate();
    p3r_circuit_desc d{};
    d.witness_count = circuit_.witness_count;
    d.n_ops = circuit_.ops.size(); d.ops = circuit_.ops.data();
    d.n_ext = circuit_.ext.size(); d.ext = circuit_.ext.data();
    d.n_public = circuit_.public_rows.size(); d.public_rows = circuit_.public_rows.data();
    d.n_private = circuit_.private_input_rows.size(); d.private_input_rows = circuit_.private_input_rows.data();
    d.n_rewrite = circuit_.witness_rewrite.size() / 2; d.witness_rewrite = circuit_.witness_rewrite.data();
    d.public_lanes = packing.public_lanes; d.alu_lanes = packing.alu_lanes; d.horner_packed_steps = packing.horner_packed_steps;
    d.recompose_lanes = packing.recompose_lanes; d.min_trace_height = packing.min_trace_height;
    std::vector<uint32_t> commit(size_t(8) << ctx.fri().cap_height);
    h_ = ctx.ptr(p3r_circuit_create(ctx.raw(), &d, commit.data()));
    p3r_layer_desc_counts counts{};
    ctx.check(p3r_circuit_counts(h_, &counts));
    ctx.check(p3r_circuit_levels(h_, &levels_));
    cpd_ = std::make_unique<CircuitProverData>(ctx, p3r_circuit_layer(h_), packing, counts, std::move(commit));
  }
  ~PreparedCircuit() { cpd_.reset(); if (h_) p3r_circuit_free(ctx_->raw(), h_); }
  PreparedCircuit(const PreparedCircuit&) = delete;
  const Circuit& circuit() const { return circuit_; }
  const CircuitProverData& circuit_prover_data() const { return *cpd_; }
  size_t levels() const { return levels_; }
  // CircuitRunner::{set_public_inputs, set_private_inputs, set_private_data, run} on the device
  ResidentTraces run(const CircuitInputs& in) const {
    p3r_circuit_inputs s = inputs_struct(in);
    return ResidentTraces(*ctx_, ctx_->ptr(p3r_circuit_run(ctx_->raw(), h_, &s)));
  }
  std::vector<uint8_t> prove(const CircuitInputs& in, bool canonical_field_encoding = false) const {
    p3r_circuit_inputs s = inputs_struct(in);
    const uint32_t flags = canonical_field_encoding ? P3R_PROVE_CANONICAL_FIELD_ENCODING : 0;
    return detail::proof_call(*ctx_, [&](uint8_t* b, size_t cap, size_t* n) { return p3r_prove_next_layer(ctx_->raw(), h_, &s, flags, b, cap, n); });
  }

 private:
  p3r_circuit_inputs inputs_struct(const CircuitInputs& in) const {
    // set_public_inputs / set_private_inputs length checks (runner.rs:84-90,107-113)
    const size_t d = ctx_->ext_degree();   // D coefficients per input
    if (in.public_values.size() != d * circuit_.public_rows.size())
      throw Error(P3R_EINVAL, "PublicInputLengthMismatch { expected: " + std::to_string(circuit_.public_rows.size()) + ", got: " +
                                  std::to_string(in.public_values.size() / d) + " }");
    if (in.private_values.size() != d * circuit_.private_input_rows.size())
      throw Error(P3R_EINVAL, "PrivateInputLengthMismatch { expected: " + std::to_string(circuit_.private_input_rows.size()) +
                                  ", got: " + std::to_string(in.private_values.size() / d) + " }");
    if (in.private_data_siblings.size() != 8 * in.private_data_op_ids.size())
      throw Error(P3R_EINVAL, "private_data_siblings must hold two extension limbs per op id");
    p3r_circuit_inputs s{};
    s.public_values = in.public_values.data(); s.private_values = in.private_values.data();
    s.n_private_data = in.private_data_op_ids.size(); s.private_data_op_ids = in.private_data_op_ids.data();
    s.private_data_siblings = in.private_data_siblings.data();
    if (in.private_data_w32_siblings.size() != 24 * in.private_data_w32_op_ids.size())
      throw Error(P3R_EINVAL, "private_data_w32_siblings must hold three digests of two extension limbs per op id");
    s.n_private_data_w32 = in.private_data_w32_op_ids.size(); s.private_data_w32_op_ids = in.private_data_w32_op_ids.data();
    s.private_data_w32_siblings = in.private_data_w32_siblings.data();
    return s;
  }
  const Context* ctx_;
  Circuit circuit_;
  p3r_circuit* h_ = nullptr;
  size_t levels_ = 0;
  std::unique_ptr<CircuitProverData> cpd_;
};

class BatchStarkProver {
 public:
  BatchStarkProver(const Context& ctx, const TablePacking& packing) : ctx_(&ctx), table_packing_(packing) {}
  BatchStarkProof prove_all_tables(const Traces& traces, const CircuitProverData& cpd, bool canonical_field_encoding = false) const {
    p3r_traces s = detail::traces_struct(traces, ctx_->ext_degree());
    const uint32_t flags = canonical_field_encoding ? P3R_PROVE_CANONICAL_FIELD_ENCODING : 0;
    auto bytes = detail::proof_call(*ctx_, [&](uint8_t* b, size_t cap, size_t* n) { return p3r_prove_all_tables(ctx_->raw(), cpd.raw(), &s, flags, b, cap, n); });
    return wrap(std::move(bytes), cpd, canonical_field_encoding);
  }
  BatchStarkProof prove_all_tables(const ResidentTraces& traces, const CircuitProverData& cpd, bool canonical_field_encoding = false) const {
    const uint32_t flags = canonical_field_encoding ? P3R_PROVE_CANONICAL_FIELD_ENCODING : 0;
    auto bytes = detail::proof_call(*ctx_, [&](uint8_t* b, size_t cap, size_t* n) { return p3r_prove_all_tables_resident(ctx_->raw(), cpd.raw(), traces.raw(), flags, b, cap, n); });
    return wrap(std::move(bytes), cpd, canonical_field_encoding);
  }
  void verify_all_tables(const BatchStarkProof& proof) const { p3r::verify_all_tables(ctx_->config(), proof); }

 private:
  // metadata as batch_stark_prover.rs:1598-1641 assembles it
  BatchStarkProof wrap(std::vector<uint8_t> bytes, const CircuitProverData& cpd, bool canonical) const {
    BatchStarkProof p;
    p.proof = std::move(bytes);
    const TablePacking& tp = cpd.effective_packing();
    p.table_packing = tp;
    p.rows = {cpd.rows().n_const, cpd.rows().n_public, cpd.rows().n_alu};
    p.ext_degree = ctx_->ext_degree();
    if (p.ext_degree == 4) p.w_binomial = binomial_w(ctx_->field());
    else if (p.ext_degree == 2 || p.ext_degree == 6 || p.ext_degree == 8) p.w_binomial = ctx_->config().ext_w;
    p.alu_quintic_trinomial = p.ext_degree == 5;
    const uint32_t k = tp.horner_packed_steps;
    const bool d4 = p.ext_degree == 4;   // D = 5 circuits carry the compact-D1 Poseidon2 table (62 preprocessed columns)
    const bool coeff = cpd.recompose_coeff_lookups;
    const uint32_t widths[6] = {2, 2 * tp.public_lanes, 13 * tp.alu_lanes + 7 * (k - 1), d4 ? 24u : 62u,
                                (2 + (coeff ? 2 * p.ext_degree : 0)) * tp.recompose_lanes,
                                (2 + 2 * p.ext_degree) * tp.recompose_lanes};
    const size_t heights[6] = {cpd.table_heights[0], cpd.table_heights[1], cpd.table_heights[2], cpd.table_heights[3],
                               cpd.table_heights[4], cpd.recompose_coeff_height};
    for (int i = 0; i < 6; ++i) {
      if (!heights[i]) continue;
      p.preprocessed_widths.push_back(widths[i]);
      uint32_t db = 0;
      while ((size_t(1) << db) < heights[i]) ++db;
      p.degree_bits.push_back(db + (ctx_->config().zk ? 1u : 0u));   // ZK: the extended degree bits (recursion.rs:374)
    }
    if (cpd.table_heights[3])  // Poseidon2Prover reports the PADDED row count (poseidon2.rs:1449)
      p.non_primitives.push_back({!d4 ? "poseidon2_perm/koala_bear_d1_w16"
                                  : ctx_->field() == Field::KoalaBear ? "poseidon2_perm/koala_bear_d4_w16" : "poseidon2_perm/baby_bear_d4_w16",
                                  cpd.table_heights[3], 1, {}, 0});
    if (cpd.table_heights[4])  // RecomposeProver reports the op count (recompose.rs:125)
      p.non_primitives.push_back({coeff ? "recompose/coeff" : "recompose", cpd.rows().n_recompose, tp.recompose_lanes, {}, 0});
    if (cpd.recompose_coeff_height)  // both table provers registered: `recompose`, then `recompose/coeff`
      p.non_primitives.push_back({"recompose/coeff", cpd.rows().n_recompose_coeff, tp.recompose_lanes, {}, 0});
    p.preprocessed_commitment = cpd.preprocessed_commitment;
    p.montgomery_field_encoding = !canonical;
    p.modulus = modulus(ctx_->field());
    return p;
  }
  const Context* ctx_;
  TablePacking table_packing_;
};

// ---- recursion API (recursion/src/recursion.rs)
struct FriRecursionBackend {  // registers the Poseidon2 + Recompose table provers for D = 4 (backend/fri.rs:693-721)
  // and none for any other degree (`else { Vec::new() }`): a D = 5 layer is proved from its primitive tables
  size_t non_primitive_provers(size_t ext_degree) const { return ext_degree == 4 ? 2 : 0; }
};
// recursion/src/backend/fri.rs:741-852: the backend of KoalaBear quintic circuits.  With the D1 permutation (`cl = true`,
// :816-835) it registers the compact-D1 Poseidon2 table and BOTH Recompose tables for D = 5 circuits, nothing otherwise;
// all three are built in, so a Context with ext_degree = 5 proves whatever tables the circuit fills (five or six).
struct FriRecursionBackendD5 : FriRecursionBackend {
  size_t non_primitive_provers(size_t ext_degree) const { return ext_degree == 5 ? 3 : 0; }
};
struct ProveNextLayerParams { TablePacking table_packing; };

struct NextLayerPrepCache {
  std::unique_ptr<BatchStarkProver> prover;
  std::unique_ptr<PreparedCircuit> prepared_circuit;          // when built from a Circuit
  std::unique_ptr<CircuitProverData> owned_prover_data;       // when built from flattened preprocessed columns
  const CircuitProverData& circuit_prover_data() const { return prepared_circuit ? prepared_circuit->circuit_prover_data() : *owned_prover_data; }
};

inline NextLayerPrepCache build_next_layer_prep(const Context& ctx, Circuit circuit, const FriRecursionBackend& backend,
                                                const ProveNextLayerParams& params) {
  backend.non_primitive_provers(ctx.ext_degree());
  NextLayerPrepCache c;
  c.prover = std::make_unique<BatchStarkProver>(ctx, params.table_packing);
  c.prepared_circuit = std::make_unique<PreparedCircuit>(ctx, std::move(circuit), params.table_packing);
  return c;
}
inline NextLayerPrepCache build_next_layer_prep(const Context& ctx, const CircuitPrep& prep, const FriRecursionBackend& backend,
                                                const ProveNextLayerParams& params) {
  backend.non_primitive_provers(ctx.ext_degree());
  NextLayerPrepCache c;
  c.prover = std::make_unique<BatchStarkProver>(ctx, params.table_packing);
  c.owned_prover_data = std::make_unique<CircuitProverData>(ctx, prep, params.table_packing);
  return c;
}

// What prove_next_layer proves: the inputs of the verifier circuit (run on the device), or Traces.
struct RecursionInput {
  const CircuitInputs* circuit_inputs = nullptr;
  const Traces* traces = nullptr;
};
struct RecursionOutput {
  BatchStarkProof proof;
};

inline RecursionOutput prove_next_layer(const RecursionInput& input, const Context& ctx, const FriRecursionBackend& backend,
                                        const ProveNextLayerParams&, const NextLayerPrepCache& prep) {
  backend.non_primitive_provers(ctx.ext_degree());
  (void)ctx;
  if (input.traces) return {prep.prover->prove_all_tables(*input.traces, prep.circuit_prover_data())};
  if (!input.circuit_inputs || !prep.prepared_circuit)
    throw Error(P3R_EINVAL, "without Traces, prove_next_layer needs a prepared Circuit and its inputs");
  ResidentTraces t = prep.prepared_circuit->run(*input.circuit_inputs);  // runner.run() (recursion.rs:478)
  return {prep.prover->prove_all_tables(t, prep.circuit_prover_data())};
}

// ---- 2-to-1 aggregation (recursion/src/recursion.rs:72-99, 656-762)
struct AggregationCircuitFingerprint {
  uint32_t witness_count = 0;
  size_t public_flat_len = 0, private_flat_len = 0, ops_len = 0;
  // The reference keys the slot by the four lengths and runs the NEW circuit against the cached prover data.
  // A hit here also reuses the cached execution schedule, so the key additionally binds the circuit's content.
  uint64_t content_digest[2] = {0, 0};
  bool operator==(const AggregationCircuitFingerprint& o) const {
    return witness_count == o.witness_count && public_flat_len == o.public_flat_len && private_flat_len == o.private_flat_len &&
           ops_len == o.ops_len && content_digest[0] == o.content_digest[0] && content_digest[1] == o.content_digest[1];
  }
};
namespace detail {
// two independent 64-bit multiply-xorshift streams over the words of the circuit arrays
inline void digest_words(uint64_t h[2], const uint32_t* w, size_t n) {
  uint64_t a = h[0] ^ (0x9E3779B97F4A7C15ull * (n + 1)), b = h[1] + 0xC2B2AE3D27D4EB4Full * (n + 1);
  for (size_t i = 0; i < n; ++i) {
    a = (a ^ w[i]) * 0xFF51AFD7ED558CCDull; a ^= a >> 32;
    b = (b + w[i]) * 0xC4CEB9FE1A85EC53ull; b ^= b >> 29;
  }
  h[0] = a; h[1] = b;
}
}  // namespace detail
inline AggregationCircuitFingerprint aggregation_circuit_fingerprint(const Circuit& c) {
  AggregationCircuitFingerprint fp{c.witness_count, c.public_rows.size(), c.private_input_rows.size(), c.ops.size(), {0, 0}};
  static_assert(sizeof(p3r_op) == 8 * sizeof(uint32_t), "p3r_op is eight words");
  detail::digest_words(fp.content_digest, reinterpret_cast<const uint32_t*>(c.ops.data()), c.ops.size() * 8);
  detail::digest_words(fp.content_digest, c.ext.data(), c.ext.size());
  detail::digest_words(fp.content_digest, c.public_rows.data(), c.public_rows.size());
  detail::digest_words(fp.content_digest, c.private_input_rows.data(), c.private_input_rows.size());
  detail::digest_words(fp.content_digest, c.witness_rewrite.data(), c.witness_rewrite.size());
  return fp;
}
struct AggregationPrepCache {
  AggregationCircuitFingerprint circuit_fingerprint;
  std::unique_ptr<BatchStarkProver> prover;
  std::unique_ptr<PreparedCircuit> prepared_circuit;  // owns the CircuitProverData
};
// The inputs of the aggregation circuit from the two halves it verifies: left first, the right proof's
// non-primitive op ids offset by the left verifier's op count (recursion.rs:596-640).
inline CircuitInputs pack_aggregation_inputs(const CircuitInputs& l, const CircuitInputs& r, uint32_t left_non_primitive_ops) {
  CircuitInputs o = l;
  o.public_values.insert(o.public_values.end(), r.public_values.begin(), r.public_values.end());
  o.private_values.insert(o.private_values.end(), r.private_values.begin(), r.private_values.end());
  for (uint32_t id : r.private_data_op_ids) o.private_data_op_ids.push_back(id + left_non_primitive_ops);
  o.private_data_siblings.insert(o.private_data_siblings.end(), r.private_data_siblings.begin(), r.private_data_siblings.end());
  for (uint32_t id : r.private_data_w32_op_ids) o.private_data_w32_op_ids.push_back(id + left_non_primitive_ops);
  o.private_data_w32_siblings.insert(o.private_data_w32_siblings.end(), r.private_data_w32_siblings.begin(), r.private_data_w32_siblings.end());
  return o;
}
// `prep_cache` is the reference's Option<&mut Option<AggregationPrepCache>>: null = no caching; an
// empty slot is filled; a filled slot is used while the circuit fingerprint matches and replaced
// when it does not.
inline RecursionOutput prove_aggregation_layer(const RecursionInput& left, const RecursionInput& right, const Circuit& verification_circuit,
                                               const Context& ctx, const FriRecursionBackend& backend, const ProveNextLayerParams& params,
                                               std::unique_ptr<AggregationPrepCache>* prep_cache = nullptr,
                                               uint32_t left_non_primitive_ops = 0) {
  if (!left.circuit_inputs || !right.circuit_inputs)
    throw Error(P3R_EINVAL, "prove_aggregation_layer needs the circuit inputs of both sides");
  backend.non_primitive_provers(ctx.ext_degree());
  const AggregationCircuitFingerprint fp = aggregation_circuit_fingerprint(verification_circuit);
  const CircuitInputs inputs = pack_aggregation_inputs(*left.circuit_inputs, *right.circuit_inputs, left_non_primitive_ops);
  if (prep_cache && *prep_cache && (*prep_cache)->circuit_fingerprint == fp) {
    AggregationPrepCache& c = **prep_cache;
    ResidentTraces t = c.prepared_circuit->run(inputs);
    return {c.prover->prove_all_tables(t, c.prepared_circuit->circuit_prover_data())};
  }
  auto fresh = std::make_unique<AggregationPrepCache>();
  fresh->circuit_fingerprint = fp;
  fresh->prover = std::make_unique<BatchStarkProver>(ctx, params.table_packing);
  fresh->prepared_circuit = std::make_unique<PreparedCircuit>(ctx, verification_circuit, params.table_packing);
  ResidentTraces t = fresh->prepared_circuit->run(inputs);
  RecursionOutput out{fresh->prover->prove_all_tables(t, fresh->prepared_circuit->circuit_prover_data())};
  if (prep_cache) *prep_cache = std::move(fresh);
  return out;
}

}  // namespace p3r
