// ORACLE (test infrastructure): extern "C" entry points for the full batch-STARK restatement:
// build the five table instances of a recursion layer from flattened Traces, prove, verify,
// (de)serialise.  PARITY UNPINNED (see field.hpp).
#include <cstring>
#include <memory>
#include <string>

#include "proof_io.hpp"
#include "stark.hpp"
#include "tables.hpp"

using namespace orc;

extern "C" {

// Flattened `Traces<EF>` + per-op preprocessed data of one recursion layer, as the
// reference's prove_all_tables consumes them (circuit/src/tables/mod.rs:49-62;
// circuit-prover/src/common.rs:198-368 for the preprocessed conventions). All canonical u32.
typedef struct orc_workload {
  size_t n_const;  const uint32_t* const_values;  /* n x 4 */ const uint32_t* const_prep;  /* n x 2: mult, idx */
  size_t n_public; const uint32_t* public_values; /* n x 4 */ const uint32_t* public_prep; /* n x 2 */
  size_t n_alu;    const uint32_t* alu_values;    /* n x 16 */ const uint32_t* alu_prep13;  /* n x 13 */
  size_t n_p2;     const uint32_t* p2_inputs;     /* n x 16 */
  const uint32_t* p2_flags;          /* n x 4: new_start, merkle_path, mmcs_bit, mmcs_ctl_enabled */
  const uint32_t* p2_mmcs_index_sum; /* n */
  const uint32_t* p2_in_ctl;         /* n x 4 */
  const uint32_t* p2_input_indices;  /* n x 4 (witness ids, unscaled) */
  const uint32_t* p2_out_ctl;        /* n x 2 (multiplicity) */
  const uint32_t* p2_output_indices; /* n x 2 */
  const uint32_t* p2_mmcs_index_sum_idx; /* n */
  size_t n_recompose; const uint32_t* recompose_values; /* n x 4 */ const uint32_t* recompose_prep; /* n x 2: idx, mult */
  /* TablePacking (circuit-prover/src/batch_stark_prover/packing.rs:10-27) */
  uint32_t public_lanes, alu_lanes, horner_packed_steps, recompose_lanes, min_trace_height;
  /* extension degree D of the circuit's element field: 0 or 4 = binomial x^4 = W (every "x 4" above), 5 = KoalaBear
   * quintic trinomial: values are n x 5 / n x 20, n_recompose = 0, and the Poseidon2 table is the compact-D1 one -
   * p2_in_ctl / p2_input_indices are n x 16, p2_out_ctl / p2_output_indices n x 8, p2_absorb_len n (nullable: zeros) */
  uint32_t ext_degree;
  const uint32_t* p2_absorb_len;
  /* 1: the Recompose table is the "recompose/coeff" variant (recompose_air.rs:196-226) - recompose_prep is
   * n x (2 + 2 D): [D*output_idx, out_mult, (D*coeff_idx_i, coeff_mult_i) x D]; recompose_values are n x D either way */
  uint32_t recompose_coeff_lookups;
  /* W of the binomial extension x^D = W for ext_degree 2, 6, 8 (0 otherwise) */
  uint32_t ext_w;
  /* the SECOND Recompose table of a layer holding both kinds (recompose_table_provers(lanes, true),
   * batch_stark_prover.rs:1914-1932: `recompose`, then `recompose/coeff`): n x D values, n x (2 + 2 D) preprocessed;
   * recompose_prep is then the plain table and recompose_coeff_lookups is 0 */
  size_t n_recompose_coeff; const uint32_t* recompose_coeff_values; const uint32_t* recompose_coeff_prep;
  /* the width-32 Poseidon2 table of the arity-4 MMCS (Poseidon2Config::*_D4_W32; D = 4 circuits), proved right after the
   * width-16 one: n x 32 inputs, n x 4 flags (new_start, merkle_path, mmcs_bit, mmcs_bit2), n index sums, and the
   * ASSEMBLED preprocessed rows n x 48 (Poseidon2PreprocessedRow<8, 6>: 8 x {idx, in_ctl, normal_chain_sel,
   * merkle_chain_sel} | 6 x {idx, out_ctl} | bit-0 witness idx | bit-1 witness idx | new_start | merkle_path);
   * w32_rc / w32_diag: the permutation's constants (canonical; 8 * 32 + partial rounds, 32) */
  size_t n_p2w; const uint32_t* p2w_inputs; const uint32_t* p2w_flags; const uint32_t* p2w_mmcs_index_sum; const uint32_t* p2w_prep;
  const uint32_t* w32_rc; const uint32_t* w32_diag;
} orc_workload;

typedef struct orc_params {
  uint32_t log_blowup, max_log_arity, cap_height, log_final_poly_len, commit_pow_bits, query_pow_bits,
      num_queries;
  // twins of p3r_config.ext_choices (bit 0: unpacked lookups) and p3r_config.fri_log_arities
  uint32_t ext_choices;
  uint32_t n_fri_log_arities;
  uint8_t fri_log_arities[32];
  uint8_t proof_layout[18];   // twin of p3r_config.proof_layout; all zero = identity
  uint32_t challenge_degree;  // 0 / 4: the quartic challenge field; 5: KoalaBear's quintic trinomial extension
  uint32_t mmcs_arity;        // 0 / 2: binary trees over the width-16 permutation; 4: the arity-4 MMCS (width 32)
  // ZK: HidingFriPcs (create_config_zk, recursion/examples/common/mod.rs:511-553); twins of p3r_config.zk*
  uint32_t zk, num_random_codewords;
  uint32_t zk_key[8];   // p3r_config.zk_key, taken as it is (the P3R_EXT_ZK_DETERMINISTIC reading)
  uint64_t zk_nonce;
  // proof-of-work witnesses to use instead of the smallest ones, in grind order (commit phases, then queries); canonical
  uint32_t n_forced_pow;
  uint32_t forced_pow[40];
  uint32_t mmcs_salt_elems;   // twin of p3r_config.mmcs_salt_elems: MerkleTreeHidingMmcs with that many salt elements per row (0: plain)
} orc_params;

const char* orc_last_error();
}

namespace {
thread_local std::string g_err2;
void set_err(const std::string& e);

struct LayerBase {
  virtual ~LayerBase() = default;
  virtual size_t num_tables() const = 0;
  virtual void info(size_t i, uint32_t* out6) const = 0;
  virtual void get(size_t i, int which, uint32_t* out) const = 0;
  virtual void prep_commit(const orc_params& p, uint32_t* cap_out) = 0;
  virtual std::vector<uint8_t> prove(const orc_params& p, int enc) = 0;
  virtual void verify(const orc_params& p, const uint32_t* prep_cap, const uint8_t* bytes, size_t n, int enc) const = 0;
};

// the challenge field of the calls that follow (field.hpp: a process-wide setting of this test oracle)
void set_challenge_degree(const orc_params& p) { challenge_degree() = p.challenge_degree == 5 ? 5 : 4; }

Layout to_layout(const orc_params& p) {
  set_challenge_degree(p);
  Layout L;
  bool any = false;
  for (int i = 0; i < 18; ++i) any |= p.proof_layout[i] != 0;
  if (!any) return L;
  std::memcpy(L.batch, p.proof_layout, 5);
  std::memcpy(L.fri, p.proof_layout + 5, 5);
  std::memcpy(L.opened, p.proof_layout + 10, 8);
  return L;
}
StarkParams to_sp(const orc_params& p) {
  set_challenge_degree(p);
  StarkParams s;
  s.log_blowup = p.log_blowup; s.max_log_arity = p.max_log_arity; s.cap_height = p.cap_height;
  s.log_final_poly_len = p.log_final_poly_len; s.commit_pow_bits = p.commit_pow_bits;
  s.query_pow_bits = p.query_pow_bits; s.num_queries = p.num_queries;
  s.lookup_unpacked = (p.ext_choices & 1u) != 0;
  s.mmcs_arity = p.mmcs_arity == 4 ? 4 : 2;
  s.zk = p.zk != 0;
  s.num_random_codewords = p.zk ? (p.num_random_codewords ? (int)p.num_random_codewords : 2) : 0;
  if (s.zk && (s.num_random_codewords < 1 || s.num_random_codewords > 8)) throw std::runtime_error("num_random_codewords must be in 1..8");
  for (int i = 0; i < 8; ++i) s.zk_key[i] = p.zk_key[i];
  s.zk_nonce = p.zk_nonce;
  s.mmcs_salt_elems = (int)p.mmcs_salt_elems;
  if (s.mmcs_salt_elems < 0 || s.mmcs_salt_elems > 16) throw std::runtime_error("mmcs_salt_elems must be in 0..16");
  for (uint32_t i = 0; i < p.n_forced_pow && i < 40; ++i) s.forced_pow.push_back(p.forced_pow[i]);
  for (uint32_t i = 0; i < p.n_fri_log_arities && i < 32; ++i) s.fri_log_arities.push_back(p.fri_log_arities[i]);
  return s;
}

template <class FP>
struct Layer : LayerBase {
  using F = Fe<FP>;
  Poseidon2<FP> p2;
  std::vector<Instance<FP>> insts;
  std::unique_ptr<ProverData<FP>> pd;
  explicit Layer(const uint32_t* rc) : p2(rc) {}

  static std::vector<F> vec(const uint32_t* p, size_t n) {
    std::vector<F> v(n);
    for (size_t i = 0; i < n; ++i) {
      if (p[i] >= FP::P) throw std::runtime_error("non-canonical workload element");
      v[i] = F(p[i]);
    }
    return v;
  }

  // order [Const, Public, Alu, Poseidon2, Recompose]
  // (circuit-prover/src/batch_stark_prover.rs:1493-1519; backend/fri.rs:693-721)
  // Lanes of a Public / ALU table holding at most the dummy op fall back to 1
  // (reduce_lanes_if_dummy, batch_stark_prover.rs:1305-1318); a non-primitive table with no
  // rows is not part of the batch (batch_stark_prover/poseidon2.rs:1089-1092, recompose.rs:77-80).
  void build(const orc_workload& w) {
    const size_t mh = w.min_trace_height;
    const int D = w.ext_degree ? (int)w.ext_degree : 4;
    // 1 = base-field circuits (the base proof of recursive_fibonacci: CircuitBuilder<F>, tests.rs:433), 4, 5
    const bool generic = D == 2 || D == 6 || D == 8;   // binomial x^D = ext_w: primitive tables + Recompose (tests.rs:486)
    if (D != 1 && D != 4 && D != 5 && !generic) throw std::runtime_error("UnsupportedExtDegree");
    if (generic && (w.ext_w == 0 || w.n_p2)) throw std::runtime_error("MissingWForExtension / no Poseidon2 table for this degree");
    const uint32_t W = generic ? w.ext_w : 0;
    if (D == 5 && FP::P != KoalaBear::P) throw std::runtime_error("D = 5 is KoalaBear's quintic extension");
    const int public_lanes = w.n_public <= 1 ? 1 : (int)w.public_lanes;
    const int alu_lanes = w.n_alu <= 1 ? 1 : (int)w.alu_lanes;
    // the width-32 permutation: of the width-32 table below, and of the arity-4 MMCS (orc_params.mmcs_arity)
    if (w.w32_rc && w.w32_diag) p2.w32 = std::make_shared<Poseidon2W32<FP>>(w.w32_rc, w.w32_diag);
    {
      Instance<FP> in;
      in.air.kind = AIR_CONST; in.air.lanes = 1; in.air.D = D; in.air.W = W;
      in.main = lanes_trace_to_matrix<FP>(vec(w.const_values, w.n_const * D), 1, mh, D);
      in.prep = lanes_prep_to_matrix<FP>(vec(w.const_prep, w.n_const * 2), 2, 1, mh);
      insts.push_back(std::move(in));
    }
    {
      Instance<FP> in;
      in.air.kind = AIR_PUBLIC; in.air.lanes = public_lanes; in.air.D = D; in.air.W = W;
      in.main = lanes_trace_to_matrix<FP>(vec(w.public_values, w.n_public * D), in.air.lanes, mh, D);
      in.prep = lanes_prep_to_matrix<FP>(vec(w.public_prep, w.n_public * 2), 2, in.air.lanes, mh);
      insts.push_back(std::move(in));
    }
    {
      Instance<FP> in;
      in.air.kind = AIR_ALU; in.air.lanes = alu_lanes; in.air.horner_k = (int)w.horner_packed_steps; in.air.D = D; in.air.W = W;
      auto values = vec(w.alu_values, w.n_alu * 4 * D);
      auto prep = vec(w.alu_prep13, w.n_alu * 13);
      in.main = alu_trace_to_matrix<FP>(in.air, values, prep, mh);
      in.prep = alu_preprocessed_trace<FP>(in.air, prep, mh);
      if (in.main.h != in.prep.h) throw std::runtime_error("ALU main/prep height mismatch");
      insts.push_back(std::move(in));
    }
    if (w.n_p2 > 0) {
      Instance<FP> in;
      in.air.kind = AIR_POSEIDON2; in.air.D = D;
      // pad the op list to a power of two >= min height with filler rows
      // (new_start = true, zero state: batch_stark_prover/poseidon2.rs:1125-1140)
      size_t n = 1;
      while (n < std::max(w.n_p2, mh)) n <<= 1;
      std::vector<P2Row<FP>> rows(n);
      std::vector<P2CtlRow<FP>> ctl(D == 4 ? w.n_p2 : 0);
      std::vector<P2CtlRowD1<FP>> ctl1(D == 4 ? 0 : w.n_p2);
      for (size_t r = 0; r < n; ++r) {
        if (r < w.n_p2) {
          rows[r].new_start = w.p2_flags[r * 4 + 0]; rows[r].merkle_path = w.p2_flags[r * 4 + 1];
          rows[r].mmcs_bit = w.p2_flags[r * 4 + 2];
          rows[r].mmcs_index_sum = F(w.p2_mmcs_index_sum[r]);
          for (int k = 0; k < 16; ++k) rows[r].input[k] = F(w.p2_inputs[r * 16 + k]);
          if (D != 4) {
            auto& c = ctl1[r];
            c.new_start = rows[r].new_start; c.merkle_path = rows[r].merkle_path;
            c.mmcs_ctl_enabled = w.p2_flags[r * 4 + 3];
            for (int l = 0; l < 16; ++l) { c.in_ctl[l] = w.p2_in_ctl[r * 16 + l]; c.input_indices[l] = w.p2_input_indices[r * 16 + l]; }
            for (int l = 0; l < 8; ++l) { c.out_ctl[l] = F(w.p2_out_ctl[r * 8 + l]); c.output_indices[l] = w.p2_output_indices[r * 8 + l]; }
            c.mmcs_index_sum_idx = w.p2_mmcs_index_sum_idx[r];
            c.absorb_len = w.p2_absorb_len ? w.p2_absorb_len[r] : 0;
            continue;
          }
          auto& c = ctl[r];
          c.new_start = rows[r].new_start; c.merkle_path = rows[r].merkle_path;
          c.mmcs_ctl_enabled = w.p2_flags[r * 4 + 3];
          for (int l = 0; l < 4; ++l) { c.in_ctl[l] = w.p2_in_ctl[r * 4 + l]; c.input_indices[l] = w.p2_input_indices[r * 4 + l]; }
          for (int l = 0; l < 2; ++l) { c.out_ctl[l] = F(w.p2_out_ctl[r * 2 + l]); c.output_indices[l] = w.p2_output_indices[r * 2 + l]; }
          c.mmcs_index_sum_idx = w.p2_mmcs_index_sum_idx[r];
        } else {
          rows[r].new_start = true;
        }
      }
      in.main = p2_generate_trace_rows<FP>(p2, rows);
      in.prep = D == 4 ? p2_preprocessed_trace<FP>(ctl, n) : p2_preprocessed_trace_d1<FP>(ctl1, n, D);
      insts.push_back(std::move(in));
    }
    if (w.n_p2w > 0) {
      if (D != 4) throw std::runtime_error("the width-32 Poseidon2 table belongs to D = 4 circuits");
      if (!w.w32_rc || !w.w32_diag) throw std::runtime_error("the width-32 Poseidon2 table needs its constants");
      p2.w32 = std::make_shared<Poseidon2W32<FP>>(w.w32_rc, w.w32_diag);
      Instance<FP> in;
      in.air.kind = AIR_POSEIDON2_W32; in.air.D = 4;
      size_t n = 1;
      while (n < std::max(w.n_p2w, mh)) n <<= 1;
      std::vector<P2WRow<FP>> rows(n);
      for (size_t r = 0; r < n; ++r) {
        if (r >= w.n_p2w) { rows[r].new_start = true; continue; }   // filler rows: new_start, zero state
        rows[r].new_start = w.p2w_flags[r * 4 + 0]; rows[r].merkle_path = w.p2w_flags[r * 4 + 1];
        rows[r].mmcs_bit = w.p2w_flags[r * 4 + 2]; rows[r].mmcs_bit2 = w.p2w_flags[r * 4 + 3];
        rows[r].mmcs_index_sum = F(w.p2w_mmcs_index_sum[r]);
        for (int k = 0; k < 32; ++k) rows[r].input[k] = F(w.p2w_inputs[r * 32 + k]);
      }
      in.main = p2w_generate_trace_rows<FP>(*p2.w32, rows);
      // BaseAir::preprocessed_trace padding (air.rs:613-649): zero rows, the first one with new_start = 1 at width - 2
      const size_t pw = 48;
      in.prep = Matrix<FP>(n, pw);
      auto cells = vec(w.p2w_prep, w.n_p2w * pw);
      std::copy(cells.begin(), cells.end(), in.prep.v.begin());
      if (n > w.n_p2w) in.prep.v[w.n_p2w * pw + pw - 2] = F::one();
      insts.push_back(std::move(in));
    }
    if (w.n_recompose > 0) {
      Instance<FP> in;
      in.air.kind = AIR_RECOMPOSE; in.air.lanes = (int)w.recompose_lanes; in.air.D = D; in.air.W = W;
      in.air.coeff_lookups = w.recompose_coeff_lookups ? 1 : 0;
      const int plw = 2 + (in.air.coeff_lookups ? 2 * D : 0);
      in.main = lanes_trace_to_matrix<FP>(vec(w.recompose_values, w.n_recompose * D), in.air.lanes, mh, D);
      in.prep = lanes_prep_to_matrix<FP>(vec(w.recompose_prep, w.n_recompose * plw), plw, in.air.lanes, mh);
      // RecomposeAir::trace_to_matrix pads only to a power of two; the prover then pads dynamic
      // tables to min_height (batch_stark_prover.rs:1515): same result as padding here.
      insts.push_back(std::move(in));
    }
    if (w.n_recompose_coeff > 0) {
      if (w.recompose_coeff_lookups) throw std::runtime_error("two Recompose tables: the first is the plain kind");
      Instance<FP> in;
      in.air.kind = AIR_RECOMPOSE; in.air.lanes = (int)w.recompose_lanes; in.air.D = D; in.air.W = W;
      in.air.coeff_lookups = 1;
      const int plw = 2 + 2 * D;
      in.main = lanes_trace_to_matrix<FP>(vec(w.recompose_coeff_values, w.n_recompose_coeff * D), in.air.lanes, mh, D);
      in.prep = lanes_prep_to_matrix<FP>(vec(w.recompose_coeff_prep, w.n_recompose_coeff * plw), plw, in.air.lanes, mh);
      insts.push_back(std::move(in));
    }
    for (auto& in : insts)
      if (in.main.h != in.prep.h) throw std::runtime_error("main/preprocessed height mismatch");
  }

  size_t num_tables() const override { return insts.size(); }
  void info(size_t i, uint32_t* o) const override {
    const auto& in = insts.at(i);
    o[0] = in.air.kind; o[1] = in.air.lanes; o[2] = in.air.horner_k; o[3] = (uint32_t)in.main.h;
    o[4] = (uint32_t)in.main.w; o[5] = (uint32_t)in.prep.w;
  }
  void get(size_t i, int which, uint32_t* out) const override {
    const auto& m = which == 0 ? insts.at(i).main : insts.at(i).prep;
    for (size_t k = 0; k < m.v.size(); ++k) out[k] = m.v[k].v;
  }
  int pd_zk = -1;
  void ensure_pd(const orc_params& p) {
    const int z = p.zk ? (int)to_sp(p).num_random_codewords : 0;
    if (!pd || pd_zk != z) pd = std::make_unique<ProverData<FP>>(make_prover_data<FP>(p2, to_sp(p), insts));
    pd_zk = z;
  }
  void prep_commit(const orc_params& p, uint32_t* cap_out) override {
    ensure_pd(p);
    for (auto& d : pd->prep.cap) for (auto x : d) *cap_out++ = x.v;
  }
  std::vector<uint8_t> prove(const orc_params& p, int enc) override {
    ensure_pd(p);
    auto proof = prove_batch<FP>(p2, to_sp(p), insts, *pd);
    return serialize_proof<FP>(proof, enc, to_layout(p), p.mmcs_salt_elems != 0);
  }
  void verify(const orc_params& p, const uint32_t* prep_cap, const uint8_t* bytes, size_t n, int enc) const override {
    auto proof = deserialize_proof<FP>(bytes, n, enc, to_layout(p), p.zk != 0, p.mmcs_salt_elems != 0);
    std::vector<InstanceShape> shapes;
    for (auto& in : insts) shapes.push_back({in.air});
    typename BatchProof<FP>::Cap cap(size_t(1) << p.cap_height);
    for (auto& d : cap) for (auto& x : d) x = F(*prep_cap++);
    verify_batch<FP>(p2, to_sp(p), shapes, cap, proof);
  }
};

template <class Fn>
int guard2(Fn&& fn) {
  try { fn(); return 0; } catch (const std::exception& e) { set_err(e.what()); return -1; }
}
}  // namespace

// error string shared with capi.cpp through a tiny setter
extern "C" void orc_set_error(const char* s);
namespace { void set_err(const std::string& e) { orc_set_error(e.c_str()); } }

extern "C" {

void* orc_layer_build(int field, const uint32_t* rc, const orc_workload* w) {
  LayerBase* out = nullptr;
  guard2([&] {
    if (field == 0) { auto l = std::make_unique<Layer<KoalaBear>>(rc); l->build(*w); out = l.release(); }
    else if (field == 1) { auto l = std::make_unique<Layer<BabyBear>>(rc); l->build(*w); out = l.release(); }
    else throw std::runtime_error("unknown field id");
  });
  return out;
}
void orc_layer_free(void* h) { delete static_cast<LayerBase*>(h); }
size_t orc_layer_num_tables(const void* h) { return static_cast<const LayerBase*>(h)->num_tables(); }
int orc_layer_table_info(const void* h, size_t i, uint32_t* out6) {
  return guard2([&] { static_cast<const LayerBase*>(h)->info(i, out6); });
}
int orc_layer_get_matrix(const void* h, size_t i, int which, uint32_t* out) {
  return guard2([&] { static_cast<const LayerBase*>(h)->get(i, which, out); });
}
int orc_layer_prep_commit(void* h, const orc_params* p, uint32_t* cap_out) {
  return guard2([&] { static_cast<LayerBase*>(h)->prep_commit(*p, cap_out); });
}
// proves; *bytes_out is malloc'd (free with orc_bytes_free)
int orc_layer_prove(void* h, const orc_params* p, int field_encoding, uint8_t** bytes_out, size_t* len_out) {
  return guard2([&] {
    auto v = static_cast<LayerBase*>(h)->prove(*p, field_encoding);
    *bytes_out = (uint8_t*)malloc(v.size());
    memcpy(*bytes_out, v.data(), v.size());
    *len_out = v.size();
  });
}
void orc_bytes_free(uint8_t* b) { free(b); }
// verify_batch against this layer's AIR shapes; returns 0 iff the proof is accepted
// verify_batch from the statement alone (AIR descriptors, no tables): what a verifier holds.
// airs4[i] = {kind, lanes, horner_packed_steps, coeff_lookups}.  Used for layers whose tables the
// CPU would take minutes to rebuild (2^20 / 2^22 rows).
int orc_verify_batch(int field, const uint32_t* rc, const orc_params* p, size_t n_airs, const uint32_t* airs4,
                     const uint32_t* prep_cap, const uint8_t* bytes, size_t len, int field_encoding) {
  return guard2([&] {
    auto run = [&](auto tag) {
      using FP = decltype(tag);
      using F = Fe<FP>;
      Poseidon2<FP> p2(rc);
      auto proof = deserialize_proof<FP>(bytes, len, field_encoding, to_layout(*p), p->zk != 0, p->mmcs_salt_elems != 0);
      std::vector<InstanceShape> shapes;
      for (size_t i = 0; i < n_airs; ++i) {
        AirDesc a;
        a.kind = (int)airs4[4 * i]; a.lanes = (int)airs4[4 * i + 1]; a.horner_k = (int)airs4[4 * i + 2];
        a.coeff_lookups = (int)(airs4[4 * i + 3] & 0xFF);
        a.D = ((airs4[4 * i + 3] >> 8) & 0xFF) ? (int)((airs4[4 * i + 3] >> 8) & 0xFF) : 4;   // bits 8..15: extension degree of the circuit (0 = 4)
        a.W = a.D == 2 || a.D == 6 || a.D == 8 ? airs4[4 * i + 3] >> 16 : 0;                   // bits 16..: W of a generic binomial (small W only)
        shapes.push_back({a});
      }
      typename BatchProof<FP>::Cap cap(size_t(1) << p->cap_height);
      const uint32_t* c = prep_cap;
      for (auto& d : cap) for (auto& x : d) x = F(*c++);
      verify_batch<FP>(p2, to_sp(*p), shapes, cap, proof);
    };
    if (field == 0) run(KoalaBear{});
    else if (field == 1) run(BabyBear{});
    else throw std::runtime_error("unknown field id");
  });
}

// the same with the constants of the width-32 permutation, for statements that hold an AIR_POSEIDON2_W32 instance
int orc_verify_batch_w32(int field, const uint32_t* rc, const uint32_t* w32_rc, const uint32_t* w32_diag, const orc_params* p,
                         size_t n_airs, const uint32_t* airs4, const uint32_t* prep_cap, const uint8_t* bytes, size_t len,
                         int field_encoding) {
  return guard2([&] {
    auto run = [&](auto tag) {
      using FP = decltype(tag);
      using F = Fe<FP>;
      Poseidon2<FP> p2(rc);
      p2.w32 = std::make_shared<Poseidon2W32<FP>>(w32_rc, w32_diag);
      auto proof = deserialize_proof<FP>(bytes, len, field_encoding, to_layout(*p), p->zk != 0, p->mmcs_salt_elems != 0);
      std::vector<InstanceShape> shapes;
      for (size_t i = 0; i < n_airs; ++i) {
        AirDesc a;
        a.kind = (int)airs4[4 * i]; a.lanes = (int)airs4[4 * i + 1]; a.horner_k = (int)airs4[4 * i + 2];
        a.coeff_lookups = (int)(airs4[4 * i + 3] & 0xFF);
        a.D = ((airs4[4 * i + 3] >> 8) & 0xFF) ? (int)((airs4[4 * i + 3] >> 8) & 0xFF) : 4;
        a.W = a.D == 2 || a.D == 6 || a.D == 8 ? airs4[4 * i + 3] >> 16 : 0;
        shapes.push_back({a});
      }
      typename BatchProof<FP>::Cap cap(size_t(1) << p->cap_height);
      const uint32_t* c = prep_cap;
      for (auto& d : cap) for (auto& x : d) x = F(*c++);
      verify_batch<FP>(p2, to_sp(*p), shapes, cap, proof);
    };
    if (field == 0) run(KoalaBear{});
    else if (field == 1) run(BabyBear{});
    else throw std::runtime_error("unknown field id");
  });
}

int orc_layer_verify(const void* h, const orc_params* p, const uint32_t* prep_cap, const uint8_t* bytes,
                     size_t len, int field_encoding) {
  return guard2([&] { static_cast<const LayerBase*>(h)->verify(*p, prep_cap, bytes, len, field_encoding); });
}

}  // extern "C"
