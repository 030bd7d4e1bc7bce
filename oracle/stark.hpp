// ORACLE (test infrastructure): CPU restatement of p3_batch_stark::prove_batch /
// verify_batch over TwoAdicFriPcs<MerkleTreeMmcs<Poseidon2>> with the LogUp lookup argument,
// for the five recursion tables.  PARITY UNPINNED (see field.hpp): the prover body lives in
// un-vendored p3-* 0.6 crates; the statement, transcript order and every check below are
// pinned by the in-tree circuit verifier, cited per step:
//
//   transcript order            recursion/src/verifier/batch_stark.rs:521-627
//   domains, rounds, points     recursion/src/verifier/batch_stark.rs:629-852
//   opened-value observation    recursion/src/verifier/batch_stark.rs:1114-1276
//   per-AIR quotient identity   recursion/src/verifier/batch_stark.rs:886-1021
//   LogUp challenge layout      recursion/src/verifier/batch_stark.rs:1026-1112
//   quotient recomposition      recursion/src/verifier/quotient.rs:60-140
//   selectors                   recursion/src/pcs/fri/targets.rs:868-908
//   constraint folding          recursion/src/traits/air.rs:162-182
//   FRI                         recursion/src/pcs/fri/targets.rs:748-866, verifier.rs:424-1838
//   ZK (HidingFriPcs)           recursion/src/verifier/batch_stark.rs:424-428 (presence), :487-490,536 (degrees),
//                               :623-661 (random commitment / round), :701-735 (quotient domains), :855-864 and
//                               :1116-1260 (FRI random opened values), pcs/fri/targets.rs:1076-1130 (merge)
//
// Choices that upstream leaves un-pinned and that DESIGN.md section "EXT choices" lists:
// LogUp per-row constraint polynomials and same-bus packing rule, FRI arity schedule rule,
// smallest-witness grinding.
#pragma once
#include <map>

#include "air.hpp"
#include "dft.hpp"

namespace orc {

struct StarkParams {
  int log_blowup = 2, max_log_arity = 2, cap_height = 0, log_final_poly_len = 5;
  int mmcs_arity = 2;   // 4: the arity-4 MMCS over the width-32 permutation (MyMmcsArity4, recursive_aggregation.rs:1024-1046)
  int commit_pow_bits = 0, query_pow_bits = 15, num_queries = 54;
  // selectable details the in-tree reference does not pin (twins of p3r_config.ext_choices /
  // fri_log_arities, include/p3r.h)
  bool lookup_unpacked = false;
  std::vector<int> fri_log_arities;
  // ZK: the configuration of create_config_zk (recursion/examples/common/mod.rs:511-553): HidingFriPcs with
  // `num_random_codewords` = 2 over the SAME (non-hiding) MMCS.  The prover side of HidingFriPcs lives in the
  // un-vendored p3-fri crate and its proofs are randomised, so byte parity with upstream is undefined: the
  // construction below is written from the acceptance conditions of the in-tree verifier (cited at each step) and
  // shared with the HIP prover, random values included (zk_rand below), so that HIP == oracle stays a byte check.
  std::vector<uint32_t> forced_pow;   // proof-of-work witnesses to use instead of the smallest ones (Challenger::forced)
  bool zk = false;
  int num_random_codewords = 2;
  std::array<uint32_t, 8> zk_key{};   // the key of the hiding PCS's generator (`rng: R`'s counterpart; p3r_config.zk_key taken as it is)
  uint64_t zk_nonce = 0;   // proofs made so far under this configuration (the PCS's RNG state advances per commit)
  // MerkleTreeHidingMmcs for the input AND the FRI commit-phase MMCS (recursion/tests/zk_hiding_mmcs.rs: "the
  // upstream-recommended ZK setup", SALT_ELEMS = 4 there): every committed matrix gets mmcs_salt_elems random elements
  // per row appended to its leaf preimage; an opening proof is (per-matrix salts, sibling digests).  0: MerkleTreeMmcs.
  int mmcs_salt_elems = 0;
};
// salts are drawn from the configuration's keyed generator: stream round = kSaltRound + the round of the committed batch
// (ZK_ROUND_* below), kSaltRoundFri for the commit-phase trees (matrix = phase)
constexpr int kSaltRound = 8, kSaltRoundFri = 14;
template <class FP>
typename MerkleTree<FP>::SaltSpec salt_spec(const StarkParams& sp, int round, size_t first_mat = 0, bool prep = false) {
  typename MerkleTree<FP>::SaltSpec s;
  s.elems = sp.mmcs_salt_elems;
  s.base.key = sp.zk_key;
  s.base.nonce = prep ? 0 : sp.zk_nonce;   // the preprocessed commitment is made once per circuit, before any proof
  s.round = round;
  s.first_mat = first_mat;
  return s;
}

// (the keyed generator of the ZK configuration - ZkStream, zk_rand - lives in hash.hpp: the hiding MMCS draws its salts from it)
enum { ZK_ROUND_RANDOM = 0, ZK_ROUND_MAIN = 1, ZK_ROUND_QUOTIENT = 2, ZK_ROUND_PREP = 3, ZK_ROUND_PERM = 4, ZK_ROUND_QMASK = 5 };
inline uint32_t zk_stream(int round, size_t mat) { return ((uint32_t)round << 20) | (uint32_t)mat; }

template <class FP>
struct Instance {
  AirDesc air;
  Matrix<FP> main;  // height = padded trace height
  Matrix<FP> prep;  // BaseAir::preprocessed_trace(), same height
};

// ------------------------------------------------------------------ proof containers
template <class FP>
struct BatchOpening {
  std::vector<std::vector<Fe<FP>>> opened_values;          // per matrix of the batch
  std::vector<std::array<Fe<FP>, DIGEST>> opening_proof;   // sibling digests bottom-up
  std::vector<std::vector<Fe<FP>>> salts;                  // hiding MMCS: per matrix; Proof = (salts, siblings)
};
template <class FP>
struct CommitPhaseStep {
  uint8_t log_arity = 1;
  std::vector<Fe4<FP>> sibling_values;
  std::vector<std::array<Fe<FP>, DIGEST>> opening_proof;
  std::vector<std::vector<Fe<FP>>> salts;                  // hiding MMCS: one entry (the phase's single matrix)
};
template <class FP>
struct QueryProof {
  std::vector<BatchOpening<FP>> input_proof;               // one per round
  std::vector<CommitPhaseStep<FP>> commit_phase_openings;
};
template <class FP>
struct FriProof {
  using Cap = std::vector<std::array<Fe<FP>, DIGEST>>;
  std::vector<Cap> commit_phase_commits;
  std::vector<Fe<FP>> commit_pow_witnesses;
  std::vector<QueryProof<FP>> query_proofs;
  std::vector<Fe4<FP>> final_poly;
  Fe<FP> query_pow_witness;
};
template <class FP>
struct OpenedValues {
  using EF = Fe4<FP>;
  std::vector<EF> trace_local;
  bool has_trace_next = false;
  std::vector<EF> trace_next;
  std::vector<EF> preprocessed_local, preprocessed_next;   // always present for circuit tables
  std::vector<std::vector<EF>> quotient_chunks;
  std::vector<EF> permutation_local, permutation_next;     // flattened aux columns (aux_width * 4)
  bool has_random = false;
  std::vector<EF> random;                                  // ZK: the random polynomial at zeta, Challenge::DIMENSION values
};
template <class FP>
struct BatchProof {
  using Cap = std::vector<std::array<Fe<FP>, DIGEST>>;
  Cap main_commit, permutation_commit, quotient_commit;
  bool has_permutation = false;
  std::vector<OpenedValues<FP>> opened;                    // per instance
  FriProof<FP> fri;
  std::vector<bool> has_terminal;
  std::vector<Fe4<FP>> lookup_terminals;                   // per instance (valid iff has_terminal)
  std::vector<size_t> degree_bits;                         // ZK: EXTENDED degree bits (base + 1)
  // ZK (HidingFriPcs): commitment of the random round; opening_proof = (OpenedValues, FriProof) whose first item are
  // the values of the random codeword columns, rounds -> matrices -> points -> num_random_codewords values
  bool has_random = false;
  Cap random_commit;
  std::vector<std::vector<std::vector<std::vector<Fe4<FP>>>>> fri_random;
};

// ------------------------------------------------------------------ LogUp
// Multiplicity degrees per interaction, in push order (for the packing degree rule).
inline std::vector<int> interaction_mult_degrees(const AirDesc& a) {
  std::vector<int> d;
  switch (a.kind) {
    case AIR_CONST: d = {1}; break;
    case AIR_PUBLIC: d.assign(a.lanes, 1); break;
    case AIR_RECOMPOSE: d.assign(a.lanes * (1 + (a.coeff_lookups ? a.D : 0)), 1); break;
    case AIR_ALU:
      for (int l = 0; l < a.lanes; ++l) { d.push_back(2); d.push_back(1); d.push_back(2); d.push_back(1); }
      for (int t = 1; t < a.horner_k; ++t) { d.push_back(1); d.push_back(1); }
      break;
    case AIR_POSEIDON2:
      if (a.D == 4) { d = {2, 2, 2, 2, 1, 1, 2}; break; }
      // compact D1: 8 rate sends (in_ctl * not_merkle), 8 output receives, the accumulator send (air.rs:1721-1785)
      d.assign(8, 2); d.insert(d.end(), 8, 1); d.push_back(2);
      break;
    case AIR_POSEIDON2_W32: d.assign(8 + 6 + 2, 1); break;   // in_ctl sends, out_ctl receives, the two direction-bit reads (-merkle_path)
  }
  return d;
}
// Maximum degree of the AIR's own (base) constraints under upstream's accounting
// (preprocessed/main columns 1, is_transition 0, is_first/is_last 1).
template <class FP>
int air_base_constraint_degree(const AirDesc& a) {
  switch (a.kind) {
    case AIR_ALU: return 3;
    case AIR_POSEIDON2: return 3;
    case AIR_POSEIDON2_W32: return 3;   // eval_arity4's degree notes, air.rs:1168-1176
    default: return 0;
  }
}
inline int group_degree(const std::vector<int>& mult_deg, const std::vector<int>& members) {
  int K = (int)members.size();
  int deg = 1 + K;  // fraction column times every denominator
  for (int m : members) deg = std::max(deg, mult_deg[m] + K - 1);
  return deg;
}
inline int log2_ceil(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

struct LookupLayout {
  std::vector<std::vector<int>> groups;  // each group -> interaction indices sharing one aux column
  int log_quotient_chunks = 0;
  int aux_width() const { return groups.empty() ? 0 : (int)groups.size() + 1; }
};
// Lookups::from_air + get_log_num_quotient_chunks + pack_same_bus(budget = 2^log_chunks + 1)
// (circuit-prover/src/batch_stark_prover.rs:925-941).  Packing rule: greedy in declaration
// order while the packed constraint degree stays within the budget.
// is_zk: get_log_num_quotient_chunks(.., is_zk) = log2_ceil(max(degree + is_zk, 2) - 1) and
// budget = 2^log_chunks + 1 - is_zk (batch_stark_prover.rs:931-939).
template <class FP>
LookupLayout lookup_layout(const AirDesc& a, bool unpacked = false, int is_zk = 0) {
  LookupLayout L;
  auto md = interaction_mult_degrees(a);
  int max_deg = air_base_constraint_degree<FP>(a);
  for (size_t i = 0; i < md.size(); ++i) max_deg = std::max(max_deg, group_degree(md, {(int)i}));
  max_deg = std::max(max_deg + is_zk, 2);
  L.log_quotient_chunks = log2_ceil(max_deg - 1);
  const int budget = (1 << L.log_quotient_chunks) + 1 - is_zk;
  std::vector<int> cur;
  for (size_t i = 0; i < md.size(); ++i) {
    auto trial = cur;
    trial.push_back((int)i);
    if (!cur.empty() && (unpacked || group_degree(md, trial) > budget)) {
      L.groups.push_back(cur);
      cur = {(int)i};
    } else {
      cur = trial;
    }
  }
  if (!cur.empty()) L.groups.push_back(cur);
  return L;
}

template <class FP>
struct LookupChallenges {
  Fe4<FP> prefix;  // alpha + gamma for bus 0 ("WitnessChecks"), gamma = beta^W
  Fe4<FP> beta;
};
// get_perm_challenges (verifier/batch_stark.rs:1026-1112): one (alpha, beta) pair; every
// lookup is on the single global bus "WitnessChecks" (bus id 0); W = widest tuple = 1 + D.
template <class FP>
LookupChallenges<FP> sample_lookup_challenges(Challenger<FP>& ch, int D) {
  Fe4<FP> alpha = ch.sample_ext(), beta = ch.sample_ext();
  Fe4<FP> gamma = beta;
  for (int i = 1; i < 1 + D; ++i) gamma *= beta;
  return {alpha + gamma, beta};
}

// denominator prefix + sum_j beta^j * field_j, generic over value type
template <class FP, class V>
Fe4<FP> lookup_denominator(const LookupChallenges<FP>& c, const std::vector<V>& fields);
template <class FP>
Fe4<FP> lookup_denom_f(const LookupChallenges<FP>& c, const std::vector<Fe<FP>>& fields) {
  Fe4<FP> d = c.prefix, bp = Fe4<FP>::one();
  for (auto& f : fields) { d += bp * f; bp *= c.beta; }
  return d;
}
template <class FP>
Fe4<FP> lookup_denom_e(const LookupChallenges<FP>& c, const std::vector<Fe4<FP>>& fields) {
  Fe4<FP> d = c.prefix, bp = Fe4<FP>::one();
  for (auto& f : fields) { d += bp * f; bp *= c.beta; }
  return d;
}

// LogUp constraints for one row, appended after the AIR's base constraints:
//   per group g:   f_g * prod_k d_k - sum_k m_k * prod_{l != k} d_l
//   is_first * s ;  is_transition * (s' - s - sum_g f_g) ;  is_last * (s + sum_g f_g - T)
template <class FP>
void logup_constraints(const LookupLayout& L, const std::vector<Fe4<FP>>& denoms,
                       const std::vector<Fe4<FP>>& mults, const std::vector<Fe4<FP>>& aux_local,
                       const std::vector<Fe4<FP>>& aux_next, Fe4<FP> is_first, Fe4<FP> is_last,
                       Fe4<FP> is_transition, Fe4<FP> terminal, std::vector<Fe4<FP>>& out) {
  using EF = Fe4<FP>;
  EF sum_f = EF::zero();
  for (size_t g = 0; g < L.groups.size(); ++g) {
    const auto& mem = L.groups[g];
    EF f = aux_local[g + 1];
    EF prod = EF::one();
    for (int k : mem) prod *= denoms[k];
    EF rhs = EF::zero();
    for (size_t a = 0; a < mem.size(); ++a) {
      EF t = mults[mem[a]];
      for (size_t bb = 0; bb < mem.size(); ++bb)
        if (bb != a) t *= denoms[mem[bb]];
      rhs += t;
    }
    out.push_back(f * prod - rhs);
    sum_f += f;
  }
  out.push_back(is_first * aux_local[0]);
  out.push_back(is_transition * (aux_next[0] - aux_local[0] - sum_f));
  out.push_back(is_last * (aux_local[0] + sum_f - terminal));
}

// ------------------------------------------------------------------ helpers
template <class FP>
Matrix<FP> rows_bitrev(const Matrix<FP>& m) {
  Matrix<FP> o(m.h, m.w);
  int l = log2_strict(m.h);
  for (size_t i = 0; i < m.h; ++i)
    std::copy(m.v.begin() + i * m.w, m.v.begin() + (i + 1) * m.w, o.v.begin() + bitrev((uint32_t)i, l) * m.w);
  return o;
}
template <class FP>
Fe4<FP> horner_ext(const std::vector<Fe<FP>>& coeffs, Fe4<FP> x) {
  Fe4<FP> acc = Fe4<FP>::zero();
  for (size_t i = coeffs.size(); i-- > 0;) acc = acc * x + Fe4<FP>(coeffs[i]);
  return acc;
}
// Evaluate every column of `evals` (given over the coset `dshift * <w_h>`, natural order) at z.
template <class FP>
std::vector<Fe4<FP>> open_matrix(const Matrix<FP>& evals, Fe<FP> dshift, Fe4<FP> z) {
  std::vector<Fe4<FP>> out(evals.w);
  Fe4<FP> zz = z * dshift.inv();
#pragma omp parallel for schedule(dynamic) if (evals.h >= 1024)
  for (size_t c = 0; c < evals.w; ++c) {
    std::vector<Fe<FP>> col(evals.h);
    for (size_t r = 0; r < evals.h; ++r) col[r] = evals.at(r, c);
    out[c] = horner_ext<FP>(idft<FP>(col), zz);
  }
  return out;
}

template <class FP>
struct Selectors {
  Fe4<FP> is_first, is_last, is_transition, inv_vanishing;
};
// selectors of the trace domain <w_n> at an arbitrary point (fri/targets.rs:868-908)
template <class FP>
Selectors<FP> selectors_at(int log_n, Fe4<FP> x) {
  using EF = Fe4<FP>;
  Fe<FP> ginv = Fe<FP>::two_adic_generator(log_n).inv();
  EF zh = x.pow(uint64_t(1) << log_n) - EF::one();
  Selectors<FP> s;
  s.is_first = zh * (x - EF::one()).inv();
  s.is_last = zh * (x - EF(ginv)).inv();
  s.is_transition = x - EF(ginv);
  s.inv_vanishing = zh.inv();
  return s;
}

// One committed batch on the prover side.
template <class FP>
struct Committed {
  std::vector<Matrix<FP>> ldes;  // bit-reversed LDEs
  MerkleTree<FP> tree;
  typename BatchProof<FP>::Cap cap;
};
template <class FP>
Committed<FP> commit_ldes(const Poseidon2<FP>& p2, std::vector<Matrix<FP>> ldes, int cap_height, int arity = 2,
                          const typename MerkleTree<FP>::SaltSpec* salt = nullptr) {
  Committed<FP> c;
  c.ldes = std::move(ldes);
  std::vector<const Matrix<FP>*> ptrs;
  for (auto& m : c.ldes) ptrs.push_back(&m);
  c.tree = MerkleTree<FP>::commit(p2, ptrs, cap_height, arity, salt);
  c.cap = c.tree.cap();
  return c;
}

// HidingFriPcs::commit on one matrix (p3-fri, un-vendored; the acceptance side is batch_stark.rs:629-661: the matrix
// is opened over the EXTENDED trace domain of size 2h, and its first w columns at zeta / zeta * g_h must be the trace
// polynomial's): the h x w evaluations become 2h x (w + R) - row 2i = [row i | R random values], row 2i + 1 all
// random - i.e. the interpolant over <g_2h> agrees with the trace on <g_h> = even powers of g_2h, and the last R
// columns are the random codewords FRI batches in.  zero_fill: the preprocessed round (public data, committed once
// per circuit shape: its padding is zeros so that the commitment does not depend on the seed).
template <class FP>
Matrix<FP> zk_randomize(const Matrix<FP>& m, int R, const ZkStream& key, bool zero_fill) {
  using F = Fe<FP>;
  const size_t w2 = m.w + (size_t)R;
  Matrix<FP> o(2 * m.h, w2);
  for (size_t r = 0; r < 2 * m.h; ++r)
    for (size_t c = 0; c < w2; ++c) {
      if (!(r & 1) && c < m.w) o.at(r, c) = m.at(r / 2, c);
      else o.at(r, c) = zero_fill ? F::zero() : zk_rand<FP>(key, r * w2 + c);
    }
  return o;
}

// Preprocessed commitment shared by every proof of a circuit shape
// (ProverData::from_airs_and_degrees; recursion/src/recursion.rs:376).
template <class FP>
struct ProverData {
  Committed<FP> prep;
  std::vector<Matrix<FP>> evals;   // the committed evaluations over the (ZK: extended) trace domain
};
template <class FP>
ProverData<FP> make_prover_data(const Poseidon2<FP>& p2, const StarkParams& sp,
                                const std::vector<Instance<FP>>& insts) {
  ProverData<FP> pd;
  std::vector<Matrix<FP>> ldes;
  for (auto& in : insts) {
    pd.evals.push_back(sp.zk ? zk_randomize<FP>(in.prep, sp.num_random_codewords, ZkStream{}, true) : in.prep);
    ldes.push_back(coset_lde_bitrev<FP>(pd.evals.back(), sp.log_blowup, Fe<FP>::generator()));
  }
  const auto prep_salt = salt_spec<FP>(sp, kSaltRound + ZK_ROUND_PREP, 0, true);
  pd.prep = commit_ldes<FP>(p2, std::move(ldes), sp.cap_height, sp.mmcs_arity, &prep_salt);
  return pd;
}

// ------------------------------------------------------------------ FRI
template <class FP>
struct FriInput {
  int log_height;
  std::vector<Fe4<FP>> ro;  // reduced openings in committed (bit-reversed) order
};

template <class FP>
Fe4<FP> fold2(Fe4<FP> e0, Fe4<FP> e1, Fe4<FP> beta, Fe<FP> x0) {
  // e0 + (beta - x0)(e1 - e0) * (-1/2)/x0   (fri/verifier.rs:562-577)
  Fe<FP> inv = (-(Fe<FP>(2).inv())) * x0.inv();
  return e0 + (beta - Fe4<FP>(x0)) * (e1 - e0) * inv;
}
// Fold one row of 2^la sibling evaluations (fri/verifier.rs:587-781): la sequential arity-2
// folds with beta, beta^2, ...; `row` is the index of the folded element at the NEW height.
template <class FP>
Fe4<FP> fold_row(std::vector<Fe4<FP>> e, size_t row, int log_new_height, int la, Fe4<FP> beta) {
  using F = Fe<FP>;
  F ss = F::two_adic_generator(log_new_height + la).pow(bitrev((uint32_t)row, log_new_height));
  F omega = F::two_adic_generator(la);
  Fe4<FP> b = beta;
  for (int step = 0; step < la; ++step) {
    int log_dom = la - step;
    F om_s = omega.pow(uint64_t(1) << step);
    size_t pairs = e.size() / 2;
    for (size_t j = 0; j < pairs; ++j) {
      F x0 = ss * om_s.pow(bitrev((uint32_t)(2 * j), log_dom));
      e[j] = fold2<FP>(e[2 * j], e[2 * j + 1], b, x0);
    }
    e.resize(pairs);
    ss = ss * ss;
    b = b * b;
  }
  return e[0];
}
// Arity schedule: fold as far as max_log_arity allows without skipping the next roll-in
// height or the final height (upstream rule un-pinned).
inline int choose_log_arity(int log_cur, int log_final, int max_log_arity, int log_next_input) {
  int la = std::min(max_log_arity, log_cur - log_final);
  if (log_next_input >= 0 && log_next_input < log_cur) la = std::min(la, log_cur - log_next_input);
  return std::max(la, 1);
}

template <class FP>
struct FriProverState {
  std::vector<Matrix<FP>> leaves;  // per phase: rows x (arity*4)
  std::vector<MerkleTree<FP>> trees;
  std::vector<int> log_arities;
};

template <class FP>
void fri_commit_phase(const Poseidon2<FP>& p2, const StarkParams& sp, std::vector<FriInput<FP>> inputs,
                      Challenger<FP>& ch, FriProof<FP>& proof, FriProverState<FP>& st) {
  using EF = Fe4<FP>;
  using F = Fe<FP>;
  std::sort(inputs.begin(), inputs.end(), [](auto& a, auto& b) { return a.log_height > b.log_height; });
  const int log_final = sp.log_final_poly_len + sp.log_blowup;
  std::vector<EF> folded = inputs[0].ro;
  size_t next = 1;
  int log_cur = inputs[0].log_height;
  while (log_cur > log_final) {
    int log_next = next < inputs.size() ? inputs[next].log_height : -1;
    int la = choose_log_arity(log_cur, log_final, sp.max_log_arity, log_next);
    if (!sp.fri_log_arities.empty()) {  // explicit schedule: must be legal (reach every roll-in and the final height)
      const size_t ph = st.log_arities.size();
      int limit = log_cur - log_final;
      if (log_next >= 0 && log_next < log_cur) limit = std::min(limit, log_cur - log_next);
      if (ph >= sp.fri_log_arities.size() || sp.fri_log_arities[ph] < 1 || sp.fri_log_arities[ph] > limit ||
          sp.fri_log_arities[ph] > sp.max_log_arity)
        throw std::runtime_error("fri_log_arities does not fit the proof");
      la = sp.fri_log_arities[ph];
    }
    size_t arity = size_t(1) << la, rows = folded.size() >> la;
    const int DC = EF::deg();   // ExtensionMmcs: a leaf is `arity` extension elements flattened to base words
    Matrix<FP> leaves(rows, arity * DC);
    for (size_t r = 0; r < rows; ++r)
      for (size_t j = 0; j < arity; ++j)
        for (int k = 0; k < DC; ++k) leaves.at(r, j * DC + k) = folded[r * arity + j].c[k];
    st.leaves.push_back(leaves);
    std::vector<const Matrix<FP>*> ptr{&st.leaves.back()};
    const auto fri_salt = salt_spec<FP>(sp, kSaltRoundFri, st.trees.size());   // ExtensionMmcs over the hiding MMCS: the flattened row, then its salt
    st.trees.push_back(MerkleTree<FP>::commit(p2, ptr, sp.cap_height, sp.mmcs_arity, &fri_salt));
    // the tree keeps a pointer to the matrix: re-point it at the stored copy after push_back moves
    st.log_arities.push_back(la);
    auto cap = st.trees.back().cap();
    proof.commit_phase_commits.push_back(cap);
    for (auto& d : cap) ch.observe_arr(d);
    proof.commit_pow_witnesses.push_back(ch.grind(sp.commit_pow_bits));
    EF beta = ch.sample_ext();
    std::vector<EF> nf(rows);
#pragma omp parallel for schedule(static) if (rows >= 1024)
    for (size_t r = 0; r < rows; ++r) {
      std::vector<EF> e(folded.begin() + r * arity, folded.begin() + (r + 1) * arity);
      nf[r] = fold_row<FP>(e, r, log_cur - la, la, beta);
    }
    log_cur -= la;
    if (next < inputs.size() && inputs[next].log_height == log_cur) {
      EF bp = beta.pow(arity);
      for (size_t i = 0; i < rows; ++i) nf[i] += bp * inputs[next].ro[i];
      ++next;
    }
    folded = nf;
  }
  if (next != inputs.size()) throw std::runtime_error("FRI: an input height was never rolled in");
  // final polynomial: natural-order evaluations over the (unshifted) subgroup -> coefficients
  size_t m = folded.size();
  int lm = log2_strict(m);
  std::vector<EF> nat(m);
  for (size_t i = 0; i < m; ++i) nat[bitrev((uint32_t)i, lm)] = folded[i];
  // inverse DFT coefficient-wise on the 4 base coordinates
  std::vector<EF> coeffs(m);
  for (int k = 0; k < EF::deg(); ++k) {
    std::vector<F> col(m);
    for (size_t i = 0; i < m; ++i) col[i] = nat[i].c[k];
    auto cc = idft<FP>(col);
    for (size_t i = 0; i < m; ++i) coeffs[i].c[k] = cc[i];
  }
  size_t flen = size_t(1) << sp.log_final_poly_len;
  for (size_t i = flen; i < m; ++i)
    if (!coeffs[i].is_zero()) throw std::runtime_error("FRI: final polynomial degree too high (constraints unsatisfied?)");
  coeffs.resize(flen);
  proof.final_poly = coeffs;
  for (auto& c : coeffs) ch.observe_ext(c);
  for (int la : st.log_arities) ch.observe(F((uint64_t)la));
  proof.query_pow_witness = ch.grind(sp.query_pow_bits);
}

// ------------------------------------------------------------------ prove_batch
template <class FP>
struct AuxTrace {
  Matrix<FP> flat;  // n x (aux_width*4)
  Fe4<FP> terminal;
};

// Evaluations of a polynomial of degree < h, given over a * <w_h> (natural order), over b * <w_h>.
template <class FP>
std::vector<Fe<FP>> coset_move(const std::vector<Fe<FP>>& evals, Fe<FP> a, Fe<FP> b) {
  return coset_dft<FP>(idft<FP>(evals), 0, b * a.inv());
}

template <class FP>
BatchProof<FP> prove_batch(const Poseidon2<FP>& p2, const StarkParams& sp,
                           const std::vector<Instance<FP>>& insts, const ProverData<FP>& pd) {
  using F = Fe<FP>;
  using EF = Fe4<FP>;
  const size_t ni = insts.size();
  const F gen = F::generator();
  const int zk = sp.zk ? 1 : 0, R = zk ? sp.num_random_codewords : 0;
  const int DC = EF::deg();
  auto key = [&](int round, size_t mat) { return ZkStream{sp.zk_key, sp.zk_nonce, zk_stream(round, mat)}; };
  BatchProof<FP> proof;
  Challenger<FP> ch(&p2);
  if (!sp.forced_pow.empty()) ch.forced = &sp.forced_pow;
  std::vector<int> log_n(ni), log_e(ni);   // base / extended (committed) trace degree bits
  std::vector<LookupLayout> layouts(ni);
  for (size_t i = 0; i < ni; ++i) {
    log_n[i] = log2_strict(insts[i].main.h);
    log_e[i] = log_n[i] + zk;
    if ((int)insts[i].main.w != air_width<FP>(insts[i].air)) throw std::runtime_error("main width mismatch");
    if ((int)insts[i].prep.w != air_prep_width(insts[i].air) || insts[i].prep.h != insts[i].main.h)
      throw std::runtime_error("preprocessed shape mismatch");
    layouts[i] = lookup_layout<FP>(insts[i].air, sp.lookup_unpacked, zk);
    proof.degree_bits.push_back(log_e[i]);
  }
  if (pd.evals.size() != ni) throw std::runtime_error("prover data of another batch");
  // 1. commit main traces (one MMCS over all instances); ZK: randomised over the extended domain
  std::vector<Matrix<FP>> main_ev(ni), main_ldes;
  for (size_t i = 0; i < ni; ++i) {
    main_ev[i] = zk ? zk_randomize<FP>(insts[i].main, R, key(ZK_ROUND_MAIN, i), false) : insts[i].main;
    main_ldes.push_back(coset_lde_bitrev<FP>(main_ev[i], sp.log_blowup, gen));
  }
  const auto main_salt = salt_spec<FP>(sp, kSaltRound + ZK_ROUND_MAIN);
  auto main_c = commit_ldes<FP>(p2, std::move(main_ldes), sp.cap_height, sp.mmcs_arity, &main_salt);
  proof.main_commit = main_c.cap;
  // 2. transcript head (batch_stark.rs:521-578): extended degree bits, base degree bits, width, chunk count
  ch.observe_base_as_ext(F((uint64_t)ni));
  for (size_t i = 0; i < ni; ++i) {
    ch.observe_base_as_ext(F((uint64_t)log_e[i]));
    ch.observe_base_as_ext(F((uint64_t)log_n[i]));
    ch.observe_base_as_ext(F((uint64_t)insts[i].main.w));
    ch.observe_base_as_ext(F(uint64_t(1) << (layouts[i].log_quotient_chunks + zk)));   // :490
  }
  for (auto& d : proof.main_commit) ch.observe_arr(d);
  // (public values: the circuit tables have none)
  for (size_t i = 0; i < ni; ++i) ch.observe_base_as_ext(F((uint64_t)insts[i].prep.w));
  for (auto& d : pd.prep.cap) ch.observe_arr(d);
  // 3. LogUp challenges, permutation traces
  bool any_lookup = false;
  for (auto& L : layouts) any_lookup |= !L.groups.empty();
  LookupChallenges<FP> lc{};
  if (any_lookup) lc = sample_lookup_challenges<FP>(ch, insts[0].air.D);
  std::vector<AuxTrace<FP>> aux(ni);
  std::vector<Matrix<FP>> aux_ev(ni);
  proof.has_terminal.assign(ni, false);
  proof.lookup_terminals.assign(ni, EF::zero());
  std::vector<int> perm_insts;
  for (size_t i = 0; i < ni; ++i) {
    const auto& L = layouts[i];
    if (L.groups.empty()) continue;
    const auto& in = insts[i];
    const size_t n = in.main.h;
    const int aw = L.aux_width();
    aux[i].flat = Matrix<FP>(n, aw * DC);
    // the fractions of a row depend on that row only (parallel); the running sum is a serial prefix
#pragma omp parallel for schedule(static) if (n >= 1024)
    for (size_t r = 0; r < n; ++r) {
      EvalCtx<FP, F> b;
      size_t rn = (r + 1) % n;
      b.local = &in.main.v[r * in.main.w]; b.next = &in.main.v[rn * in.main.w];
      b.prep_local = &in.prep.v[r * in.prep.w]; b.prep_next = &in.prep.v[rn * in.prep.w];
      b.record_constraints = false;
      b.is_first = b.is_last = b.is_transition = F::zero();
      eval_air<FP, F>(in.air, p2, b);
      for (size_t g = 0; g < L.groups.size(); ++g) {
        EF f = EF::zero();
        for (int m : L.groups[g]) {
          const auto& it = b.interactions[m];
          if (it.mult.v != 0) f += lookup_denom_f<FP>(lc, it.fields).inv() * it.mult;
        }
        for (int k = 0; k < DC; ++k) aux[i].flat.at(r, (g + 1) * DC + k) = f.c[k];
      }
    }
    EF run = EF::zero();
    for (size_t r = 0; r < n; ++r) {
      for (int k = 0; k < DC; ++k) aux[i].flat.at(r, k) = run.c[k];
      for (size_t g = 0; g < L.groups.size(); ++g) {
        EF f;
        for (int k = 0; k < DC; ++k) f.c[k] = aux[i].flat.at(r, (g + 1) * DC + k);
        run += f;
      }
    }
    aux[i].terminal = run;
    proof.has_terminal[i] = true;
    proof.lookup_terminals[i] = run;
    perm_insts.push_back((int)i);
  }
  Committed<FP> perm_c;
  if (any_lookup) {
    std::vector<Matrix<FP>> ldes;
    for (size_t k = 0; k < perm_insts.size(); ++k) {
      const int i = perm_insts[k];
      aux_ev[i] = zk ? zk_randomize<FP>(aux[i].flat, R, key(ZK_ROUND_PERM, k), false) : aux[i].flat;
      ldes.push_back(coset_lde_bitrev<FP>(aux_ev[i], sp.log_blowup, gen));
    }
    const auto perm_salt = salt_spec<FP>(sp, kSaltRound + ZK_ROUND_PERM);
    perm_c = commit_ldes<FP>(p2, std::move(ldes), sp.cap_height, sp.mmcs_arity, &perm_salt);
    proof.has_permutation = true;
    proof.permutation_commit = perm_c.cap;
    for (auto& d : perm_c.cap) ch.observe_arr(d);
    for (size_t i = 0; i < ni; ++i)
      if (proof.has_terminal[i]) ch.observe_ext(proof.lookup_terminals[i]);
  }
  // 4. constraint-folding challenge, quotients
  EF alpha = ch.sample_ext();
  std::vector<Matrix<FP>> q_chunk_evals;   // per (instance, chunk): the committed evaluations over its chunk coset
  std::vector<F> q_chunk_shift;
  std::vector<std::pair<int, int>> q_chunk_owner;
  for (size_t i = 0; i < ni; ++i) {
    const auto& in = insts[i];
    const auto& L = layouts[i];
    // quotient domain: ext_dom.create_disjoint_domain(1 << (base_db + log_qd + is_zk)), 2^(log_qd + is_zk) chunks
    // (batch_stark.rs:701-717)
    const int lq = L.log_quotient_chunks + zk, C = 1 << lq;
    const size_t n = in.main.h, qn = n << lq;
    if (L.log_quotient_chunks > sp.log_blowup) throw std::runtime_error("quotient domain larger than the LDE");
    // evaluations on the quotient coset gen*<w_qn>, natural order = first qn rows of the
    // bit-reversed LDE, un-reversed (Pcs::get_evaluations_on_domain)
    auto on_q = [&](const Matrix<FP>& lde) {
      Matrix<FP> head(qn, lde.w);
      std::copy(lde.v.begin(), lde.v.begin() + qn * lde.w, head.v.begin());
      return rows_bitrev<FP>(head);
    };
    Matrix<FP> mq = on_q(main_c.ldes[i]), pq = on_q(pd.prep.ldes[i]), aq;
    int perm_pos = -1;
    for (size_t k = 0; k < perm_insts.size(); ++k) if (perm_insts[k] == (int)i) perm_pos = (int)k;
    if (perm_pos >= 0) aq = on_q(perm_c.ldes[perm_pos]);
    Matrix<FP> qflat(qn, DC);
    const F wq = F::two_adic_generator(log_n[i] + lq);
    const int aw = L.aux_width();
    std::vector<F> xs(qn);
    {
      F x = gen;
      for (size_t r = 0; r < qn; ++r, x *= wq) xs[r] = x;
    }
#pragma omp parallel for schedule(static) if (qn >= 1024)
    for (size_t r = 0; r < qn; ++r) {
      const F x = xs[r];
      size_t rn = (r + C) % qn;   // x * g_n: the next row of the BASE trace domain
      EvalCtx<FP, F> b;
      b.local = &mq.v[r * mq.w]; b.next = &mq.v[rn * mq.w];
      b.prep_local = &pq.v[r * pq.w]; b.prep_next = &pq.v[rn * pq.w];
      auto sel = selectors_at<FP>(log_n[i], EF(x));
      b.is_first = sel.is_first.c[0]; b.is_last = sel.is_last.c[0]; b.is_transition = sel.is_transition.c[0];
      eval_air<FP, F>(in.air, p2, b);
      std::vector<EF> ext_cons;
      if (aw) {
        std::vector<EF> al(aw), an(aw), den, mul;
        for (int c = 0; c < aw; ++c)
          for (int k = 0; k < DC; ++k) { al[c].c[k] = aq.at(r, c * DC + k); an[c].c[k] = aq.at(rn, c * DC + k); }
        for (auto& it : b.interactions) { den.push_back(lookup_denom_f<FP>(lc, it.fields)); mul.push_back(EF(it.mult)); }
        logup_constraints<FP>(L, den, mul, al, an, sel.is_first, sel.is_last, sel.is_transition,
                              proof.lookup_terminals[i], ext_cons);
      }
      EF acc = EF::zero();
      for (auto& c : b.constraints) acc = acc * alpha + EF(c);
      for (auto& c : ext_cons) acc = acc * alpha + c;
      EF q = acc * sel.inv_vanishing;
      for (int k = 0; k < DC; ++k) qflat.at(r, k) = q.c[k];
    }
    // split_evals: chunk c = rows c, c+C, ... ; split_domains: shift gen * wq^c
    std::vector<Matrix<FP>> ce(C, Matrix<FP>(n, DC));
    std::vector<F> sh(C);
    for (int c = 0; c < C; ++c) {
      for (size_t r = 0; r < n; ++r)
        for (int k = 0; k < DC; ++k) ce[c].at(r, k) = qflat.at(r * C + c, k);
      sh[c] = gen * wq.pow(c);
    }
    if (zk) {
      // HidingFriPcs::commit_quotient, from the acceptance side: the verifier opens chunk c over
      // natural_domain_for_degree(2n) (batch_stark.rs:719-727) and recomposes quotient(zeta) = sum_c zp_c(zeta) q'_c(zeta),
      // zp_c(x) = prod_{j != c} Z_j(x) / Z_j(s_c), Z_j(x) = (x / s_j)^n - 1 (verifier/quotient.rs).  q'_c = q_c + Z_c t_c
      // leaves the sum unchanged iff sum_c k_c t_c = 0 with k_c = prod_{j != c} 1 / Z_j(s_c): C - 1 independent random
      // t_c of degree < n and t_{C-1} = -(1 / k_{C-1}) sum_{c < C-1} k_c t_c (eprint 2024/1037's chunk masking).
      // Each t_c (c < C - 1) is given by n random evaluations over U = u * <g_n>, u = s_{C-1} * g_2n; the committed matrix
      // of chunk c is q'_c over s_c * <g_2n>: even rows q_c (the chunk evaluations), odd rows q_c - 2 t_c on
      // s_c g_2n <g_n> (Z_c = g_2n^n - 1 = -2 there), plus R fully random codeword columns.
      const F g2 = F::two_adic_generator(log_n[i] + 1);
      const F u = sh[C - 1] * g2;
      std::vector<F> kc(C);
      for (int c = 0; c < C; ++c) {
        F d = F::one();
        for (int j = 0; j < C; ++j)
          if (j != c) d *= (sh[c] * sh[j].inv()).pow(uint64_t(1) << log_n[i]) - F::one();
        kc[c] = d.inv();
      }
      std::vector<Matrix<FP>> tU(C, Matrix<FP>(n, DC));   // t_c over U
      for (int c = 0; c + 1 < C; ++c) {
        const ZkStream tk = key(ZK_ROUND_QMASK, q_chunk_evals.size() + c);
        for (size_t r = 0; r < n; ++r)
          for (int k = 0; k < DC; ++k) tU[c].at(r, k) = zk_rand<FP>(tk, r * DC + k);
      }
      const F neg_inv_last = -(kc[C - 1].inv());
      for (size_t r = 0; r < n; ++r)
        for (int k = 0; k < DC; ++k) {
          F acc = F::zero();
          for (int c = 0; c + 1 < C; ++c) acc += kc[c] * tU[c].at(r, k);
          tU[C - 1].at(r, k) = acc * neg_inv_last;
        }
      for (int c = 0; c < C; ++c) {
        const ZkStream ck = key(ZK_ROUND_QUOTIENT, q_chunk_evals.size() + c);
        const size_t w2 = (size_t)DC + R;
        Matrix<FP> m(2 * n, w2);
        for (int k = 0; k < DC; ++k) {
          std::vector<F> qc(n), tc(n);
          for (size_t r = 0; r < n; ++r) { qc[r] = ce[c].at(r, k); tc[r] = tU[c].at(r, k); }
          auto q_odd = coset_move<FP>(qc, sh[c], sh[c] * g2), t_odd = coset_move<FP>(tc, u, sh[c] * g2);
          for (size_t r = 0; r < n; ++r) {
            m.at(2 * r, k) = qc[r];
            m.at(2 * r + 1, k) = q_odd[r] - t_odd[r] - t_odd[r];
          }
        }
        for (size_t r = 0; r < 2 * n; ++r)
          for (size_t k = DC; k < w2; ++k) m.at(r, k) = zk_rand<FP>(ck, r * w2 + k);
        ce[c] = std::move(m);
      }
    }
    for (int c = 0; c < C; ++c) {
      q_chunk_evals.push_back(std::move(ce[c]));
      q_chunk_shift.push_back(sh[c]);
      q_chunk_owner.emplace_back((int)i, c);
    }
  }
  std::vector<Matrix<FP>> q_ldes;
  for (size_t k = 0; k < q_chunk_evals.size(); ++k)
    q_ldes.push_back(coset_lde_bitrev<FP>(q_chunk_evals[k], sp.log_blowup, gen * q_chunk_shift[k].inv()));
  const auto quot_salt = salt_spec<FP>(sp, kSaltRound + ZK_ROUND_QUOTIENT);
  auto quot_c = commit_ldes<FP>(p2, std::move(q_ldes), sp.cap_height, sp.mmcs_arity, &quot_salt);
  proof.quotient_commit = quot_c.cap;
  for (auto& d : quot_c.cap) ch.observe_arr(d);
  // ZK: the random round - per instance a fully random matrix of Challenge::DIMENSION (+ R) columns over the extended
  // trace domain; its commitment is observed after the quotient's (batch_stark.rs:623-625), its round comes first (:645-661)
  Committed<FP> rand_c;
  std::vector<Matrix<FP>> rand_ev(ni);
  if (zk) {
    std::vector<Matrix<FP>> ldes;
    for (size_t i = 0; i < ni; ++i) {
      const ZkStream rk = key(ZK_ROUND_RANDOM, i);
      const size_t w2 = (size_t)DC + R, h2 = insts[i].main.h * 2;
      rand_ev[i] = Matrix<FP>(h2, w2);
      for (size_t r = 0; r < h2; ++r)
        for (size_t c = 0; c < w2; ++c) rand_ev[i].at(r, c) = zk_rand<FP>(rk, r * w2 + c);
      ldes.push_back(coset_lde_bitrev<FP>(rand_ev[i], sp.log_blowup, gen));
    }
    const auto rand_salt = salt_spec<FP>(sp, kSaltRound + ZK_ROUND_RANDOM);
    rand_c = commit_ldes<FP>(p2, std::move(ldes), sp.cap_height, sp.mmcs_arity, &rand_salt);
    proof.has_random = true;
    proof.random_commit = rand_c.cap;
    for (auto& d : rand_c.cap) ch.observe_arr(d);
  }
  EF zeta = ch.sample_ext();

  // 5. open (rounds: [random,] main, quotient, preprocessed, permutation); observe in round/matrix/point order.  Every
  // committed matrix is opened in full (ZK: its R random codeword columns included - the values HidingFriPcs splits off
  // into the opening proof's first item and the verifier merges back, batch_stark.rs:855-864, :1116-1260)
  proof.opened.resize(ni);
  struct OpenItem { int round, mat, log_h; EF z; std::vector<EF> vals; };
  std::vector<OpenItem> items;  // in reduced-opening order
  const int r0 = zk;            // index of the main round
  if (zk)
    for (size_t i = 0; i < ni; ++i)
      items.push_back({0, (int)i, log_e[i], zeta, open_matrix<FP>(rand_ev[i], F::one(), zeta)});
  for (size_t i = 0; i < ni; ++i) {
    EF zn = zeta * F::two_adic_generator(log_n[i]);   // zeta * g of the BASE trace domain (:663-700)
    items.push_back({r0, (int)i, log_e[i], zeta, open_matrix<FP>(main_ev[i], F::one(), zeta)});
    if (air_uses_next(insts[i].air)) items.push_back({r0, (int)i, log_e[i], zn, open_matrix<FP>(main_ev[i], F::one(), zn)});
  }
  for (size_t k = 0; k < q_chunk_evals.size(); ++k) {
    int i = q_chunk_owner[k].first;
    items.push_back({r0 + 1, (int)k, log_e[i], zeta, open_matrix<FP>(q_chunk_evals[k], q_chunk_shift[k], zeta)});
  }
  for (size_t i = 0; i < ni; ++i) {
    EF zn = zeta * F::two_adic_generator(log_n[i]);
    items.push_back({r0 + 2, (int)i, log_e[i], zeta, open_matrix<FP>(pd.evals[i], F::one(), zeta)});
    items.push_back({r0 + 2, (int)i, log_e[i], zn, open_matrix<FP>(pd.evals[i], F::one(), zn)});
  }
  for (size_t k = 0; k < perm_insts.size(); ++k) {
    int i = perm_insts[k];
    EF zn = zeta * F::two_adic_generator(log_n[i]);
    items.push_back({r0 + 3, (int)k, log_e[i], zeta, open_matrix<FP>(aux_ev[i], F::one(), zeta)});
    items.push_back({r0 + 3, (int)k, log_e[i], zn, open_matrix<FP>(aux_ev[i], F::one(), zn)});
  }
  for (auto& it : items) for (auto& v : it.vals) ch.observe_ext(v);
  // the proof's fields: the first w values of each opening; ZK: the trailing R per (round, matrix, point)
  {
    const int n_rounds = (zk ? 1 : 0) + 3 + (any_lookup ? 1 : 0);
    if (zk) proof.fri_random.resize(n_rounds);
    for (auto& it : items) {
      std::vector<EF> head(it.vals.begin(), it.vals.end() - R), tail(it.vals.end() - R, it.vals.end());
      if (zk) {
        auto& rd = proof.fri_random[it.round];
        if ((int)rd.size() <= it.mat) rd.resize(it.mat + 1);
        rd[it.mat].push_back(tail);
      }
      const int rr = it.round - r0;
      if (rr == -1) { proof.opened[it.mat].has_random = true; proof.opened[it.mat].random = head; }
      else if (rr == 0) {
        auto& ov = proof.opened[it.mat];
        if (ov.trace_local.empty()) ov.trace_local = head;
        else { ov.has_trace_next = true; ov.trace_next = head; }
      } else if (rr == 1) proof.opened[q_chunk_owner[it.mat].first].quotient_chunks.push_back(head);
      else if (rr == 2) {
        auto& ov = proof.opened[it.mat];
        if (ov.preprocessed_local.empty()) ov.preprocessed_local = head; else ov.preprocessed_next = head;
      } else {
        auto& ov = proof.opened[perm_insts[it.mat]];
        if (ov.permutation_local.empty()) ov.permutation_local = head; else ov.permutation_next = head;
      }
    }
  }

  // 6. FRI: batching challenge, per-height reduced openings
  EF fri_alpha = ch.sample_ext();
  std::vector<const Committed<FP>*> rounds;
  if (zk) rounds.push_back(&rand_c);
  rounds.push_back(&main_c); rounds.push_back(&quot_c); rounds.push_back(&pd.prep);
  if (any_lookup) rounds.push_back(&perm_c);
  std::map<int, std::pair<EF, std::vector<EF>>> ros;  // log_height -> (alpha_pow, ro)
  for (auto& it : items) {
    const Matrix<FP>& lde = rounds[it.round]->ldes[it.mat];
    int lh = it.log_h + sp.log_blowup;
    if (lde.h != (size_t(1) << lh) || lde.w != it.vals.size()) throw std::runtime_error("internal: opening shape");
    auto& e = ros[lh];
    if (e.second.empty()) { e.first = EF::one(); e.second.assign(lde.h, EF::zero()); }
    const F wl = F::two_adic_generator(lh);
#pragma omp parallel for schedule(static) if (lde.h >= 1024)
    for (size_t r = 0; r < lde.h; ++r) {
      F x = gen * wl.pow(bitrev((uint32_t)r, lh));
      EF inv = (it.z - EF(x)).inv();
      EF ap = e.first, acc = EF::zero();
      for (size_t c = 0; c < lde.w; ++c) {
        acc += ap * (it.vals[c] - EF(lde.at(r, c)));
        ap *= fri_alpha;
      }
      e.second[r] += acc * inv;
    }
    e.first *= fri_alpha.pow(lde.w);
  }
  std::vector<FriInput<FP>> inputs;
  for (auto& kv : ros) inputs.push_back({kv.first, kv.second.second});
  FriProverState<FP> st;
  st.leaves.reserve(64); st.trees.reserve(64);
  fri_commit_phase<FP>(p2, sp, inputs, ch, proof.fri, st);
  // the Merkle trees hold pointers into st.leaves (reserved above, so no reallocation)
  int log_max = 0;
  for (auto& in : inputs) log_max = std::max(log_max, in.log_height);
  for (int q = 0; q < sp.num_queries; ++q) {
    size_t index = ch.sample_bits(log_max);
    QueryProof<FP> qp;
    for (const Committed<FP>* cmp : rounds) {
      const auto& cm = *cmp;
      BatchOpening<FP> bo;
      size_t ridx = index >> (log_max - cm.tree.log_max_h);
      cm.tree.open(ridx, bo.opened_values, bo.opening_proof, sp.mmcs_salt_elems ? &bo.salts : nullptr);
      qp.input_proof.push_back(bo);
    }
    size_t idx = index;
    for (size_t p = 0; p < st.trees.size(); ++p) {
      int la = st.log_arities[p];
      size_t row = idx >> la, pos = idx & ((size_t(1) << la) - 1);
      CommitPhaseStep<FP> step;
      step.log_arity = (uint8_t)la;
      for (size_t j = 0; j < (size_t(1) << la); ++j) {
        if (j == pos) continue;
        EF e;
        for (int k = 0; k < EF::deg(); ++k) e.c[k] = st.leaves[p].at(row, j * EF::deg() + k);
        step.sibling_values.push_back(e);
      }
      std::vector<std::vector<F>> ov;
      st.trees[p].open(row, ov, step.opening_proof, sp.mmcs_salt_elems ? &step.salts : nullptr);
      qp.commit_phase_openings.push_back(step);
      idx = row;
    }
    proof.fri.query_proofs.push_back(qp);
  }
  return proof;
}

// ------------------------------------------------------------------ verify_batch
// Shape of one instance as the verifier knows it.
struct InstanceShape {
  AirDesc air;
};

template <class FP>
void verify_batch(const Poseidon2<FP>& p2, const StarkParams& sp, const std::vector<InstanceShape>& shapes,
                  const typename BatchProof<FP>::Cap& prep_commit, const BatchProof<FP>& proof) {
  using F = Fe<FP>;
  using EF = Fe4<FP>;
  auto fail = [](const std::string& m) { throw std::runtime_error("verify: " + m); };
  const size_t ni = shapes.size();
  if (proof.opened.size() != ni || proof.degree_bits.size() != ni) fail("instance count mismatch");
  if (proof.has_terminal.size() != ni || proof.lookup_terminals.size() != ni) fail("terminal count mismatch");
  const F gen = F::generator();
  const int zk = sp.zk ? 1 : 0, R = zk ? sp.num_random_codewords : 0;
  const int DC = EF::deg();
  // randomisation must match the PCS's ZK setting (batch_stark.rs:424-428: RandomizationError)
  if (proof.has_random != (zk != 0)) fail("RandomizationError: random commitment presence does not match the ZK setting");
  for (auto& ov : proof.opened)
    if (ov.has_random != (zk != 0)) fail("RandomizationError: random opened values presence does not match the ZK setting");
  std::vector<LookupLayout> layouts(ni);
  std::vector<int> log_n(ni), log_e(ni);
  bool any_lookup = false;
  for (size_t i = 0; i < ni; ++i) {
    layouts[i] = lookup_layout<FP>(shapes[i].air, sp.lookup_unpacked, zk);
    log_e[i] = (int)proof.degree_bits[i];
    // base_db = ext_db - is_zk (:536): "Extended degree bits smaller than ZK adjustment"
    if (log_e[i] < zk) fail("extended degree bits smaller than the ZK adjustment");
    log_n[i] = log_e[i] - zk;
    if (log_e[i] + sp.log_blowup > FP::TWO_ADICITY) fail("degree too large");
    any_lookup |= !layouts[i].groups.empty();
    const auto& ov = proof.opened[i];
    const int w = air_width<FP>(shapes[i].air), pw = air_prep_width(shapes[i].air);
    if ((int)ov.trace_local.size() != w) fail("trace width");
    if (ov.has_trace_next != air_uses_next(shapes[i].air)) fail("trace_next presence");
    if (ov.has_trace_next && (int)ov.trace_next.size() != w) fail("trace_next width");
    if ((int)ov.preprocessed_local.size() != pw || (int)ov.preprocessed_next.size() != pw) fail("prep width");
    // quotient_degree = 1 << (log_qd + is_zk) (:487-496)
    if ((int)ov.quotient_chunks.size() != (1 << (layouts[i].log_quotient_chunks + zk))) fail("chunk count");
    for (auto& c : ov.quotient_chunks) if (c.size() != (size_t)DC) fail("chunk width");
    if (zk && ov.random.size() != (size_t)DC) fail("RandomizationError: random opened values length");   // :506-511
    const size_t aflat = (size_t)layouts[i].aux_width() * DC;
    if (ov.permutation_local.size() != aflat || ov.permutation_next.size() != aflat) fail("permutation width");
    if (proof.has_terminal[i] != !layouts[i].groups.empty()) fail("terminal presence");
  }
  if (proof.has_permutation != any_lookup) fail("permutation commitment presence");
  Challenger<FP> ch(&p2);
  ch.observe_base_as_ext(F((uint64_t)ni));
  for (size_t i = 0; i < ni; ++i) {
    ch.observe_base_as_ext(F((uint64_t)log_e[i]));
    ch.observe_base_as_ext(F((uint64_t)log_n[i]));
    ch.observe_base_as_ext(F((uint64_t)air_width<FP>(shapes[i].air)));
    ch.observe_base_as_ext(F(uint64_t(1) << (layouts[i].log_quotient_chunks + zk)));
  }
  for (auto& d : proof.main_commit) ch.observe_arr(d);
  for (size_t i = 0; i < ni; ++i) ch.observe_base_as_ext(F((uint64_t)air_prep_width(shapes[i].air)));
  for (auto& d : prep_commit) ch.observe_arr(d);
  LookupChallenges<FP> lc{};
  if (any_lookup) {
    lc = sample_lookup_challenges<FP>(ch, shapes[0].air.D);
    for (auto& d : proof.permutation_commit) ch.observe_arr(d);
    for (size_t i = 0; i < ni; ++i) if (proof.has_terminal[i]) ch.observe_ext(proof.lookup_terminals[i]);
  }
  EF alpha = ch.sample_ext();
  for (auto& d : proof.quotient_commit) ch.observe_arr(d);
  if (zk) for (auto& d : proof.random_commit) ch.observe_arr(d);   // :623-625
  EF zeta = ch.sample_ext();

  // rounds: (commitment cap, matrices (log_height, width, points(z, values)))
  // ZK: every matrix carries R more columns, whose values at each point come from the opening proof's first item
  // (merge_hiding_random_openings, pcs/fri/targets.rs:1076-1130: round / matrix / point counts must match)
  struct MatOpen { int log_h; std::vector<std::pair<EF, std::vector<EF>>> pts; };
  struct Round { const typename BatchProof<FP>::Cap* cap; std::vector<MatOpen> mats; };
  std::vector<Round> rounds;
  if (zk) {
    Round r{&proof.random_commit, {}};
    for (size_t i = 0; i < ni; ++i) r.mats.push_back({log_e[i], {{zeta, proof.opened[i].random}}});   // :645-661
    rounds.push_back(r);
  }
  {
    Round r{&proof.main_commit, {}};
    for (size_t i = 0; i < ni; ++i) {
      MatOpen m{log_e[i], {{zeta, proof.opened[i].trace_local}}};
      if (proof.opened[i].has_trace_next)
        m.pts.push_back({zeta * F::two_adic_generator(log_n[i]), proof.opened[i].trace_next});   // base-domain generator
      r.mats.push_back(m);
    }
    rounds.push_back(r);
  }
  {
    // randomized_quotient_domains: natural_domain_for_degree(size << is_zk) (:719-727)
    Round r{&proof.quotient_commit, {}};
    for (size_t i = 0; i < ni; ++i)
      for (auto& c : proof.opened[i].quotient_chunks) r.mats.push_back({log_e[i], {{zeta, c}}});
    rounds.push_back(r);
  }
  {
    Round r{&prep_commit, {}};
    for (size_t i = 0; i < ni; ++i) {
      EF zn = zeta * F::two_adic_generator(log_n[i]);
      r.mats.push_back({log_e[i], {{zeta, proof.opened[i].preprocessed_local}, {zn, proof.opened[i].preprocessed_next}}});
    }
    rounds.push_back(r);
  }
  if (any_lookup) {
    Round r{&proof.permutation_commit, {}};
    for (size_t i = 0; i < ni; ++i) {
      if (proof.opened[i].permutation_local.empty()) continue;
      EF zn = zeta * F::two_adic_generator(log_n[i]);
      r.mats.push_back({log_e[i], {{zeta, proof.opened[i].permutation_local}, {zn, proof.opened[i].permutation_next}}});
    }
    rounds.push_back(r);
  }
  if (zk) {
    if (proof.fri_random.size() != rounds.size()) fail("hiding FRI proof shape: random rounds count does not match commitments");
    for (size_t r = 0; r < rounds.size(); ++r) {
      if (proof.fri_random[r].size() != rounds[r].mats.size()) fail("hiding FRI proof shape: random matrices count does not match");
      for (size_t m = 0; m < rounds[r].mats.size(); ++m) {
        auto& pts = rounds[r].mats[m].pts;
        if (proof.fri_random[r][m].size() != pts.size()) fail("hiding FRI proof shape: random points count does not match");
        for (size_t p = 0; p < pts.size(); ++p) {
          const auto& extra = proof.fri_random[r][m][p];
          if ((int)extra.size() != R) fail("hiding FRI proof shape: random codeword count");
          pts[p].second.insert(pts[p].second.end(), extra.begin(), extra.end());
        }
      }
    }
  } else if (!proof.fri_random.empty()) fail("random opened values in a non-ZK proof");
  for (auto& r : rounds) for (auto& m : r.mats) for (auto& pt : m.pts) for (auto& v : pt.second) ch.observe_ext(v);

  // FRI challenges (fri/targets.rs:748-814)
  const auto& fp = proof.fri;
  EF fri_alpha = ch.sample_ext();
  std::vector<int> log_arities;
  if (!fp.query_proofs.empty())
    for (auto& s : fp.query_proofs[0].commit_phase_openings) log_arities.push_back(s.log_arity);
  if (fp.commit_phase_commits.size() != log_arities.size() || fp.commit_pow_witnesses.size() != log_arities.size())
    fail("FRI phase count");
  std::vector<EF> betas;
  for (size_t p = 0; p < fp.commit_phase_commits.size(); ++p) {
    for (auto& d : fp.commit_phase_commits[p]) ch.observe_arr(d);
    if (!ch.check_witness(sp.commit_pow_bits, fp.commit_pow_witnesses[p])) fail("commit PoW");
    betas.push_back(ch.sample_ext());
  }
  if (fp.final_poly.size() != (size_t(1) << sp.log_final_poly_len)) fail("final poly length");
  for (auto& c : fp.final_poly) ch.observe_ext(c);
  for (int la : log_arities) ch.observe(F((uint64_t)la));
  if (!ch.check_witness(sp.query_pow_bits, fp.query_pow_witness)) fail("query PoW");
  int total_red = 0;
  for (int la : log_arities) total_red += la;
  const int log_max = total_red + sp.log_final_poly_len + sp.log_blowup;
  if ((int)fp.query_proofs.size() != sp.num_queries) fail("query count");

  for (auto& qp : fp.query_proofs) {
    size_t index = ch.sample_bits(log_max);
    if (qp.input_proof.size() != rounds.size()) fail("input proof round count");
    std::map<int, std::pair<EF, EF>> ro;  // log_height -> (alpha_pow, ro)
    for (size_t r = 0; r < rounds.size(); ++r) {
      const auto& rd = rounds[r];
      const auto& bo = qp.input_proof[r];
      if (bo.opened_values.size() != rd.mats.size()) fail("opened matrix count");
      int batch_max = 0;
      std::vector<std::pair<size_t, size_t>> dims;
      for (size_t m = 0; m < rd.mats.size(); ++m) {
        batch_max = std::max(batch_max, rd.mats[m].log_h + sp.log_blowup);
        dims.emplace_back(size_t(1) << (rd.mats[m].log_h + sp.log_blowup), bo.opened_values[m].size());
      }
      if (batch_max > log_max) fail("a committed matrix taller than the FRI domain");
      size_t ridx = index >> (log_max - batch_max);
      if (sp.mmcs_salt_elems) {   // `(salts, siblings)`: one salt of SALT_ELEMS per matrix of the batch (mmcs.rs:339-347)
        if (bo.salts.size() != dims.size()) fail("input MMCS opening: salt count");
        for (auto& sl : bo.salts) if ((int)sl.size() != sp.mmcs_salt_elems) fail("input MMCS opening: salt length");
      } else if (!bo.salts.empty()) fail("input MMCS opening: salts under a non-hiding MMCS");
      if (!MerkleTree<FP>::verify(p2, *rd.cap, sp.cap_height, dims, ridx, bo.opened_values, bo.opening_proof, sp.mmcs_arity,
                                  sp.mmcs_salt_elems ? &bo.salts : nullptr))
        fail("input MMCS opening");
      for (size_t m = 0; m < rd.mats.size(); ++m) {
        int lh = rd.mats[m].log_h + sp.log_blowup;
        // x = GENERATOR * g_lh^{rev(index >> (log_max - lh))}   (fri/verifier.rs:921-981)
        size_t ih = index >> (log_max - lh);
        F x = gen * F::two_adic_generator(lh).pow(bitrev((uint32_t)ih, lh));
        auto it = ro.find(lh);
        if (it == ro.end()) it = ro.emplace(lh, std::make_pair(EF::one(), EF::zero())).first;
        for (auto& pt : rd.mats[m].pts) {
          if (pt.second.size() != bo.opened_values[m].size()) fail("opened width vs point values");
          EF inv = (pt.first - EF(x)).inv();
          for (size_t c = 0; c < pt.second.size(); ++c) {
            it->second.second += it->second.first * (pt.second[c] - EF(bo.opened_values[m][c])) * inv;
            it->second.first *= fri_alpha;
          }
        }
      }
    }
    if (ro.count(sp.log_blowup) && !ro[sp.log_blowup].second.is_zero()) fail("height-1 reduced opening");
    if (!ro.count(log_max)) fail("no reduced opening at the maximum height");
    EF folded = ro[log_max].second;
    size_t idx = index;
    int log_cur = log_max;
    if (qp.commit_phase_openings.size() != log_arities.size()) fail("commit phase opening count");
    for (size_t p = 0; p < log_arities.size(); ++p) {
      const auto& step = qp.commit_phase_openings[p];
      int la = step.log_arity;
      if (la != log_arities[p] || la < 1 || la > sp.max_log_arity) fail("log_arity");
      size_t arity = size_t(1) << la, pos = idx & (arity - 1), row = idx >> la;
      if (step.sibling_values.size() != arity - 1) fail("sibling count");
      std::vector<EF> evals(arity);
      size_t s = 0;
      for (size_t j = 0; j < arity; ++j) evals[j] = (j == pos) ? folded : step.sibling_values[s++];
      std::vector<F> flat;
      for (auto& e : evals) for (int k = 0; k < EF::deg(); ++k) flat.push_back(e.c[k]);
      std::vector<std::pair<size_t, size_t>> dims{{size_t(1) << (log_cur - la), arity * (size_t)EF::deg()}};
      if (sp.mmcs_salt_elems) {
        if (step.salts.size() != 1 || (int)step.salts[0].size() != sp.mmcs_salt_elems) fail("commit-phase MMCS opening: salt shape");
      } else if (!step.salts.empty()) fail("commit-phase MMCS opening: salts under a non-hiding MMCS");
      if (!MerkleTree<FP>::verify(p2, fp.commit_phase_commits[p], sp.cap_height, dims, row, {flat}, step.opening_proof, sp.mmcs_arity,
                                  sp.mmcs_salt_elems ? &step.salts : nullptr))
        fail("commit-phase MMCS opening");
      folded = fold_row<FP>(evals, row, log_cur - la, la, betas[p]);
      log_cur -= la;
      idx = row;
      if (log_cur < log_max && ro.count(log_cur)) folded += betas[p].pow(arity) * ro[log_cur].second;
    }
    for (auto& kv : ro)
      if (kv.first < log_cur && kv.first != sp.log_blowup) fail("reduced opening below the final height");
    // final polynomial at g^{rev(idx)} of the final domain (fri/verifier.rs:887-915)
    F xf = F::two_adic_generator(log_cur).pow(bitrev((uint32_t)idx, log_cur));
    EF ev = EF::zero();
    for (size_t i = fp.final_poly.size(); i-- > 0;) ev = ev * EF(xf) + fp.final_poly[i];
    if (ev != folded) fail("final polynomial mismatch");
  }

  // per-AIR quotient identity (batch_stark.rs:886-1017) and terminal sum (:1019-1021)
  EF tsum = EF::zero();
  for (size_t i = 0; i < ni; ++i) {
    const auto& L = layouts[i];
    const auto& ov = proof.opened[i];
    // quotient domain of size 2^(base_db + log_qd + is_zk) split into 2^(log_qd + is_zk) chunk domains of the BASE
    // trace size (:701-717): the chunks are recomposed over those, not over the randomised opening domains
    const int lq = L.log_quotient_chunks + zk, C = 1 << lq;
    // recompose quotient(zeta) from chunks (verifier/quotient.rs:60-140)
    const F wq = F::two_adic_generator(log_n[i] + lq);
    std::vector<F> shifts(C);
    for (int c = 0; c < C; ++c) shifts[c] = gen * wq.pow(c);
    auto zh_coset = [&](F shift, EF x) { return (x * shift.inv()).pow(uint64_t(1) << log_n[i]) - EF::one(); };
    EF quotient = EF::zero();
    for (int c = 0; c < C; ++c) {
      EF zp = EF::one();
      for (int j = 0; j < C; ++j) {
        if (j == c) continue;
        zp *= zh_coset(shifts[j], zeta) * zh_coset(shifts[j], EF(shifts[c])).inv();
      }
      EF val = EF::zero();
      for (int e = 0; e < EF::deg(); ++e) {
        EF basis = EF::zero();
        basis.c[e] = F::one();
        val += basis * ov.quotient_chunks[c][e];
      }
      quotient += zp * val;
    }
    auto sel = selectors_at<FP>(log_n[i], zeta);
    EvalCtx<FP, EF> b;
    std::vector<EF> zeros(ov.trace_local.size(), EF::zero());
    b.local = ov.trace_local.data();
    b.next = ov.has_trace_next ? ov.trace_next.data() : zeros.data();
    b.prep_local = ov.preprocessed_local.data();
    b.prep_next = ov.preprocessed_next.data();
    b.is_first = sel.is_first; b.is_last = sel.is_last; b.is_transition = sel.is_transition;
    eval_air<FP, EF>(shapes[i].air, p2, b);
    std::vector<EF> ext_cons;
    const int aw = L.aux_width();
    if (aw) {
      auto recompose = [&](const std::vector<EF>& flat) {
        std::vector<EF> out(aw, EF::zero());
        for (int c = 0; c < aw; ++c)
          for (int e = 0; e < EF::deg(); ++e) {
            EF basis = EF::zero();
            basis.c[e] = F::one();
            out[c] += basis * flat[c * EF::deg() + e];
          }
        return out;
      };
      auto al = recompose(ov.permutation_local), an = recompose(ov.permutation_next);
      std::vector<EF> den, mul;
      for (auto& it : b.interactions) { den.push_back(lookup_denom_e<FP>(lc, it.fields)); mul.push_back(it.mult); }
      logup_constraints<FP>(L, den, mul, al, an, sel.is_first, sel.is_last, sel.is_transition,
                            proof.lookup_terminals[i], ext_cons);
      tsum += proof.lookup_terminals[i];
    }
    EF acc = EF::zero();
    for (auto& c : b.constraints) acc = acc * alpha + c;
    for (auto& c : ext_cons) acc = acc * alpha + c;
    if (acc * sel.inv_vanishing != quotient) fail("constraints do not match the quotient for instance " + std::to_string(i));
  }
  if (any_lookup && !tsum.is_zero()) fail("lookup terminals do not sum to zero");
}

}  // namespace orc
