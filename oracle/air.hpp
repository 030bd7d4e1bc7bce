// ORACLE (test infrastructure): the five recursion-table AIRs restated from the reference's
// `Air::eval` implementations, generic over the value type so the same statement is used by
// the CPU prover (V = base field, one quotient-domain row) and the CPU verifier
// (V = extension field, openings at zeta).  PARITY UNPINNED (see field.hpp).
//
// Constraint ORDER is significant (alpha folding, recursion/src/traits/air.rs:162-182) and
// follows the order of `assert_*` calls in the cited Rust functions.  Interactions are
// recorded in `push_interaction` order; the LogUp constraints built from them are appended
// after all base constraints (logup.hpp).
#pragma once
#include <string>

#include "hash.hpp"

namespace orc {

// AIR_POSEIDON2_W32: the width-32 D = 4 table of the arity-4 MMCS (Poseidon2CircuitAir{Koala,Baby}BearD4Width32)
enum AirKind { AIR_CONST = 0, AIR_PUBLIC = 1, AIR_ALU = 2, AIR_POSEIDON2 = 3, AIR_RECOMPOSE = 4, AIR_POSEIDON2_W32 = 5 };
// Circuit extension degree: 1 (base-field circuits), 4 (binomial x^4 = W) or 5 (KoalaBear quintic trinomial x^5 + x^2 - 1,
// alu_air.rs:115-134; its Poseidon2 table is the compact-D1 width-16 one, eval_poseidon2_d1 below; Recompose is D = 4 only).
// The STARK's own challenge field stays the degree-4 binomial extension, as in the reference's D = 5 unit tests
// (batch_stark_prover/tests.rs:844-1029: QuinticTrinomialExtensionField traces under config::koala_bear()).
constexpr int kMaxD = 8;

struct AirDesc {
  int kind = AIR_CONST;
  int lanes = 1;
  int horner_k = 2;          // ALU: TablePacking::horner_packed_steps (packing.rs:10-27)
  int coeff_lookups = 0;     // Recompose: challenger.d() != D (backend/fri.rs:693-721)
  int D = 4;                 // extension degree of the circuit's element field
  uint32_t W = 0;            // W of a binomial extension x^D = W; 0: the field's W of its degree-4 extension
};

// ---- widths (shape_golden.rs:32-68 pins the ALU formula) ----
inline int alu_num_int(int k) { return (k - 1) / 2; }                        // alu_columns.rs
inline int alu_extra_prep_width(int k) { return (k - 1) + 6 * (k - 1); }     // alu_columns.rs
template <class FP>
int air_width(const AirDesc& a) {
  const int D = a.D;
  switch (a.kind) {
    case AIR_CONST: return D;                                                // const_air.rs:88-127
    case AIR_PUBLIC: return a.lanes * D;                                     // public_air.rs:127-169
    case AIR_ALU: return a.lanes * 4 * D + (alu_num_int(a.horner_k) + 2 * (a.horner_k - 1) + 1) * D;  // alu_air.rs:320-325
    case AIR_POSEIDON2: return Poseidon2<FP>::perm_cols() + 2;               // air.rs:561-585
    case AIR_RECOMPOSE: return a.lanes * D;                                  // recompose_air.rs:96-119
    case AIR_POSEIDON2_W32: return Poseidon2W32<FP>::perm_cols() + 4;        // num_cols_arity4 (cols.rs:122-130)
  }
  throw std::runtime_error("bad air kind");
}
inline int air_prep_width(const AirDesc& a) {
  const int D = a.D;
  switch (a.kind) {
    case AIR_CONST: return 2;
    case AIR_PUBLIC: return a.lanes * 2;
    case AIR_ALU: return a.lanes * 13 + alu_extra_prep_width(a.horner_k);    // alu_air.rs:333-335
    case AIR_POSEIDON2: return D == 4 ? 4 * 4 + 2 * 2 + 4 : 26 + 16 + 8 + 8 + 4;  // preprocessed.rs:104-108, :127-145 (compact D1)
    case AIR_RECOMPOSE: return a.lanes * (2 + (a.coeff_lookups ? 2 * D : 0));
    case AIR_POSEIDON2_W32: return 8 * 4 + 6 * 2 + 4;   // poseidon_preprocessed_row_width(WIDTH_EXT = 8, RATE_EXT = 6), preprocessed.rs:104-108
  }
  throw std::runtime_error("bad air kind");
}
// BaseAir::main_next_row_columns non-empty? (traits/air.rs:106-112: const/public/recompose omit it)
inline bool air_uses_next(const AirDesc& a) { return a.kind == AIR_ALU || a.kind == AIR_POSEIDON2 || a.kind == AIR_POSEIDON2_W32; }

// ---- evaluation context ----
template <class FP, class V>
struct EvalCtx {
  using F = Fe<FP>;
  const V* local = nullptr;
  const V* next = nullptr;
  const V* prep_local = nullptr;
  const V* prep_next = nullptr;
  V is_first, is_last, is_transition;
  struct Interaction {
    std::vector<V> fields;
    V mult;
  };
  std::vector<V> constraints;
  std::vector<Interaction> interactions;
  bool record_constraints = true;

  static V K(uint64_t x) { return V(F(x)); }
  static V KF(F f) { return V(f); }
  void assert_zero(const V& e) { if (record_constraints) constraints.push_back(e); }
  void push_interaction(std::vector<V> fields, V mult) { interactions.push_back({std::move(fields), mult}); }
};

// Fe(F) "copy" so V(F(x)) works uniformly for V = Fe.
template <class FP> inline Fe<FP> lift_to(const Fe<FP>& f, Fe<FP>*) { return f; }

// ---- WitnessSendAir (Const / Public): public_air.rs:209-239 ----
template <class FP, class V>
void eval_witness_send(const AirDesc& a, EvalCtx<FP, V>& b) {
  const int D = a.D;
  for (int lane = 0; lane < a.lanes; ++lane) {
    V mult = b.prep_local[lane * 2 + 0];
    V idx = b.prep_local[lane * 2 + 1];
    std::vector<V> f{idx};
    for (int j = 0; j < D; ++j) f.push_back(b.local[lane * D + j]);
    b.push_interaction(std::move(f), mult);
  }
}

// ---- RecomposeAir: recompose_air.rs:141-198 ----
template <class FP, class V>
void eval_recompose(const AirDesc& a, EvalCtx<FP, V>& b) {
  const int D = a.D;
  const int plw = 2 + (a.coeff_lookups ? 2 * D : 0);
  for (int lane = 0; lane < a.lanes; ++lane) {
    const V* p = b.prep_local + lane * plw;
    std::vector<V> f{p[0]};
    for (int j = 0; j < D; ++j) f.push_back(b.local[lane * D + j]);
    b.push_interaction(std::move(f), p[1]);
    if (a.coeff_lookups) {
      for (int i = 0; i < D; ++i) {
        std::vector<V> cf{p[2 + 2 * i], b.local[lane * D + i]};
        for (int j = 1; j < D; ++j) cf.push_back(b.K(0));
        b.push_interaction(std::move(cf), p[2 + 2 * i + 1]);
      }
    }
  }
}

// x*y on D-coefficient slices: F[x]/(x^4 - W) (ext_mul_binomial, alu_air.rs:715-733) or, for D = 5,
// F[x]/(x^5 + x^2 - 1) (ext_mul_quintic_trinomial, alu_air.rs:737-765: x^5 = 1 - x^2, x^6 = x - x^3,
// x^7 = x^2 - x^4, x^8 = x^3 + x^2 - 1)
template <class FP, class V>
std::array<V, kMaxD> ext_mul(int D, const V* x, const V* y, uint32_t W = 0) {
  std::array<V, kMaxD> acc;
  for (auto& e : acc) e = EvalCtx<FP, V>::K(0);
  if (D == 5 && W == 0) {
    V c[9];
    for (auto& e : c) e = EvalCtx<FP, V>::K(0);
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) c[i + j] = c[i + j] + x[i] * y[j];
    V c5_minus_c8 = c[5] - c[8];
    acc[0] = c[0] + c5_minus_c8;
    acc[1] = c[1] + c[6];
    acc[2] = c[2] - c5_minus_c8 + c[7];
    acc[3] = c[3] - c[6] + c[8];
    acc[4] = c[4] - c[7];
    return acc;
  }
  const V w = EvalCtx<FP, V>::K(W ? W : FP::W);
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) {
      V term = x[i] * y[j];
      int k = i + j;
      if (k < D) acc[k] = acc[k] + term;
      else acc[k - D] = acc[k - D] + w * term;
    }
  return acc;
}

// ---- AluAir: interactions alu_air.rs:1000-1085, constraints alu_air.rs:764-996 ----
template <class FP, class V>
void eval_alu(const AirDesc& a, EvalCtx<FP, V>& b) {
  const int D = a.D;
  const int lanes = a.lanes, LW = 4 * D, PW = 13, k_max = a.horner_k;
  const int extra_main = lanes * LW, extra_prep = lanes * PW;
  const int num_int = alu_num_int(k_max);
  const int ac_base = extra_main + num_int * D;
  auto sel_k_idx = [](int k) { return k - 2; };
  auto step_prep = [&](int t) { return (k_max - 1) + 6 * (t - 1); };
  const V* L = b.local; const V* N = b.next; const V* PL = b.prep_local; const V* PN = b.prep_next;

  // interactions: 4 per lane (a,b,c,out), then (a_t,c_t) for t = 1..K-1
  for (int lane = 0; lane < lanes; ++lane) {
    const V* m = L + lane * LW;
    const V* p = PL + lane * PW;
    V mult_a = p[0], mult_b = p[9], mult_out = p[10], a_rd = p[11], c_rd = p[12];
    V mults[4] = {mult_a * a_rd, mult_b, mult_a * c_rd, mult_out};
    for (int i = 0; i < 4; ++i) {
      std::vector<V> f{p[5 + i]};
      for (int j = 0; j < D; ++j) f.push_back(m[i * D + j]);
      b.push_interaction(std::move(f), mults[i]);
    }
  }
  for (int t = 1; t < k_max; ++t) {
    const V* sp = PL + extra_prep + step_prep(t);
    int off = ac_base + 2 * (t - 1) * D;
    std::vector<V> fa{sp[0]}, fc{sp[1]};
    for (int j = 0; j < D; ++j) fa.push_back(L[off + j]);
    for (int j = 0; j < D; ++j) fc.push_back(L[off + D + j]);
    b.push_interaction(std::move(fa), sp[4]);
    b.push_interaction(std::move(fc), sp[5]);
  }

  const V zero = b.K(0), one = b.K(1);
  for (int lane = 0; lane < lanes; ++lane) {
    const V* ll = L + lane * LW; const V* ln = N + lane * LW;
    const V* pc = PL + lane * PW; const V* pn = PN + lane * PW;
    const V *A = ll, *B = ll + D, *C = ll + 2 * D, *O = ll + 3 * D;
    V mult_a = pc[0], sel_add = pc[1], sel_bool = pc[2], sel_muladd = pc[3], sel_horner = pc[4];
    V active = zero - mult_a;
    V sel_mul = active - sel_bool - sel_muladd - sel_horner - sel_add;
    for (int i = 0; i < D; ++i) b.assert_zero(sel_add * (A[i] + B[i] - O[i]));                 // ADD
    auto ab = ext_mul<FP, V>(D, A, B, a.W);
    for (int i = 0; i < D; ++i) b.assert_zero(sel_mul * (ab[i] - O[i]));                        // MUL
    b.assert_zero(sel_bool * A[0] * (A[0] - one));                                              // BOOL
    for (int i = 1; i < D; ++i) b.assert_zero(sel_bool * A[i]);
    for (int i = 0; i < D; ++i) b.assert_zero(sel_muladd * (ab[i] + C[i] - O[i]));              // MUL_ADD
    // HORNER_ACC
    V next_sel_horner = pn[4];
    const V *NA = ln, *NB = ln + D, *NC = ln + 2 * D, *NO = ln + 3 * D;
    auto out_next_b = ext_mul<FP, V>(D, O, NB, a.W);
    if (lane == 0) {
      const V* next_int0 = N + extra_main;
      V any_cur = zero, any_next = zero, sel_ge3_next = zero;
      for (int kk = 2; kk <= k_max; ++kk) any_cur = any_cur + PL[extra_prep + sel_k_idx(kk)];
      for (int kk = 2; kk <= k_max; ++kk) any_next = any_next + PN[extra_prep + sel_k_idx(kk)];
      V next_sel_k2 = PN[extra_prep + sel_k_idx(2)];
      for (int kk = 3; kk <= k_max; ++kk) sel_ge3_next = sel_ge3_next + PN[extra_prep + sel_k_idx(kk)];
      const int b_sq_base = ac_base + 2 * (k_max - 1) * D;
      const V* b_sq = L + b_sq_base; const V* b_sq_next = N + b_sq_base;
      auto bb = ext_mul<FP, V>(D, B, B, a.W);
      for (int i = 0; i < D; ++i) b.assert_zero(any_cur * (b_sq[i] - bb[i]));
      auto out_b_sq = ext_mul<FP, V>(D, O, b_sq_next, a.W);
      auto c0_b_next = ext_mul<FP, V>(D, NC, NB, a.W);
      auto a0_b_next = ext_mul<FP, V>(D, NA, NB, a.W);
      const V* a1_next = N + ac_base; const V* c1_next = N + ac_base + D;
      for (int i = 0; i < D; ++i) {                                                              // 1) packed inter-row
        V poly = out_b_sq[i] + c0_b_next[i] - a0_b_next[i] + c1_next[i] - a1_next[i];
        b.assert_zero(next_sel_k2 * (poly - NO[i]));
        b.assert_zero(sel_ge3_next * (poly - next_int0[i]));
      }
      V next_sel_single = next_sel_horner - any_next;                                            // 2) single-step
      for (int i = 0; i < D; ++i) b.assert_zero(next_sel_single * (out_next_b[i] + NC[i] - NA[i] - NO[i]));
      for (int kk = 3; kk <= k_max; ++kk) {                                                      // 3) intra-row legs
        V sel_kk = PL[extra_prep + sel_k_idx(kk)];
        int s = 2, slot = 0;
        while (s < kk) {
          const V* int_curr = L + extra_main + slot * D;
          int off_s = ac_base + 2 * (s - 1) * D;
          const V* a_s = L + off_s; const V* c_s = L + off_s + D;
          if (s + 1 < kk) {
            int off_sp1 = ac_base + 2 * s * D;
            const V* a_sp1 = L + off_sp1; const V* c_sp1 = L + off_sp1 + D;
            auto int_b_sq = ext_mul<FP, V>(D, int_curr, b_sq, a.W);
            auto c_s_b = ext_mul<FP, V>(D, c_s, B, a.W);
            auto a_s_b = ext_mul<FP, V>(D, a_s, B, a.W);
            const V* target = (s + 2 >= kk) ? O : (L + extra_main + (slot + 1) * D);
            for (int i = 0; i < D; ++i) {
              V prod = int_b_sq[i] + c_s_b[i] - a_s_b[i] + c_sp1[i] - a_sp1[i];
              b.assert_zero(sel_kk * (prod - target[i]));
            }
            if (!(s + 2 >= kk)) slot += 1;
            s += 2;
          } else {
            auto int_b = ext_mul<FP, V>(D, int_curr, B, a.W);
            for (int i = 0; i < D; ++i) b.assert_zero(sel_kk * (int_b[i] + c_s[i] - a_s[i] - O[i]));
            s += 1;
          }
        }
      }
    } else {
      for (int i = 0; i < D; ++i) b.assert_zero(next_sel_horner * (out_next_b[i] + NC[i] - NA[i] - NO[i]));
    }
  }
}

// p3_poseidon2_air::eval through the SubAirBuilder over the permutation columns (air.rs:1137-1158), restated
// from the Poseidon2Cols semantics (SURVEY.md appendix A "Inner perm-AIR constraints").
template <class FP, class V>
void eval_poseidon2_perm(const Poseidon2<FP>& p2, EvalCtx<FP, V>& b) {
  constexpr int R = FP::SBOX_REGS;
  const V* L = b.local;
  const V zero = b.K(0);
  auto full_off = [&](int r) { return WIDTH + r * (WIDTH * R + WIDTH); };
  const int partial_off = full_off(HALF_FULL);
  const int ending_off = partial_off + FP::PARTIAL * (R + 1);
  auto end_off = [&](int r) { return ending_off + r * (WIDTH * R + WIDTH); };
  std::array<V, WIDTH> s;
  for (int i = 0; i < WIDTH; ++i) s[i] = L[i];
  auto external = [&](std::array<V, WIDTH>& st) {
    std::array<V, WIDTH> o;
    for (int i = 0; i < WIDTH; ++i) {
      V acc = zero;
      for (int j = 0; j < WIDTH; ++j) acc = acc + st[j] * b.KF(p2.ext[i][j]);
      o[i] = acc;
    }
    st = o;
  };
  auto internal = [&](std::array<V, WIDTH>& st) {
    V sum = zero;
    for (auto& x : st) sum = sum + x;
    for (int i = 0; i < WIDTH; ++i) st[i] = st[i] * b.KF(p2.diag[i]) + sum;
  };
  auto sbox = [&](V x, const V* reg) -> V {
    if (FP::SBOX_DEGREE == 3) return x * x * x;
    // (7,1): committed x^3 register, then x^7 = (x^3)^2 * x
    V c3 = reg[0];
    b.assert_zero(c3 - x * x * x);
    return c3 * c3 * x;
  };
  external(s);
  int k = 0;
  auto full_round = [&](int col) {
    for (int i = 0; i < WIDTH; ++i) {
      V x = s[i] + b.KF(p2.rc[k + i]);
      s[i] = sbox(x, L + col + i * R);
    }
    k += WIDTH;
    external(s);
    const V* post = L + col + WIDTH * R;
    for (int i = 0; i < WIDTH; ++i) {
      b.assert_zero(s[i] - post[i]);
      s[i] = post[i];
    }
  };
  for (int r = 0; r < HALF_FULL; ++r) full_round(full_off(r));
  for (int r = 0; r < FP::PARTIAL; ++r) {
    int col = partial_off + r * (R + 1);
    V x = s[0] + b.KF(p2.rc[k++]);
    s[0] = sbox(x, L + col);
    b.assert_zero(s[0] - L[col + R]);
    s[0] = L[col + R];
    internal(s);
  }
  for (int r = 0; r < HALF_FULL; ++r) full_round(end_off(r));
}

// ---- Poseidon2CircuitAir (D=4, width 16, arity-2 shape) ----
// interactions poseidon2-circuit-air/src/air.rs:1790-1893; circuit constraints :937,:1049-1122;
// inner permutation AIR p3_poseidon2_air::eval via SubAirBuilder (:1137-1158) restated from
// the Poseidon2Cols semantics (SURVEY.md appendix A "Inner perm-AIR constraints").
template <class FP, class V>
void eval_poseidon2(const Poseidon2<FP>& p2, EvalCtx<FP, V>& b) {
  constexpr int D = 4;  // the D = 4 width-16 arity-2 table (the D = 5 backend's compact-D1 table is not restated)
  constexpr int WE = 4, RE = 2, R = FP::SBOX_REGS;
  const int pc = Poseidon2<FP>::perm_cols();
  const V* L = b.local; const V* N = b.next; const V* PL = b.prep_local; const V* PN = b.prep_next;
  // column offsets inside Poseidon2Cols
  auto full_off = [&](int r) { return WIDTH + r * (WIDTH * R + WIDTH); };                       // beginning rounds
  const int partial_off = full_off(HALF_FULL);
  const int ending_off = partial_off + FP::PARTIAL * (R + 1);
  auto end_off = [&](int r) { return ending_off + r * (WIDTH * R + WIDTH); };
  const V* local_out = L + end_off(HALF_FULL - 1) + WIDTH * R;  // ending_full_rounds[3].post
  const V* next_in = N;                                          // next.perm.inputs
  const V mmcs_bit = L[pc], mmcs_index_sum = L[pc + 1];
  const V next_bit = N[pc], next_index_sum = N[pc + 1];
  // prep row: input_limbs[4]{idx,in_ctl,normal_chain_sel,merkle_chain_sel} | output_limbs[2]{idx,out_ctl}
  //           | mmcs_index_sum_ctl_idx | mmcs_merkle_flag | new_start | merkle_path
  auto in_limb = [&](const V* P, int l, int f) { return P[l * 4 + f]; };
  auto out_limb = [&](const V* P, int l, int f) { return P[16 + l * 2 + f]; };
  const V one = b.K(1), zero = b.K(0);

  // --- interactions ---
  {
    V not_merkle = one - PL[23];
    for (int l = 0; l < WE; ++l) {
      std::vector<V> f{in_limb(PL, l, 0)};
      for (int d = 0; d < D; ++d) f.push_back(L[l * D + d]);
      b.push_interaction(std::move(f), zero - in_limb(PL, l, 1) * not_merkle);
    }
    for (int l = 0; l < RE; ++l) {
      std::vector<V> f{out_limb(PL, l, 0)};
      for (int d = 0; d < D; ++d) f.push_back(local_out[l * D + d]);
      b.push_interaction(std::move(f), out_limb(PL, l, 1));
    }
    std::vector<V> f{PL[20], mmcs_index_sum};
    for (int d = 1; d < D; ++d) f.push_back(zero);
    b.push_interaction(std::move(f), zero - PL[21] * PN[22]);
  }

  // --- circuit constraints ---
  b.assert_zero(mmcs_bit * (one - mmcs_bit));  // assert_bool: x * (1 - x)? see note below
  for (int l = 0; l < WE; ++l)
    for (int d = 0; d < D; ++d)
      b.assert_zero(b.is_transition * in_limb(PN, l, 2) * (next_in[l * D + d] - local_out[l * D + d]));
  V is_left = one - next_bit;
  for (int i = 0; i < RE; ++i) {
    V gate = in_limb(PN, i, 3) * is_left;
    for (int d = 0; d < D; ++d)
      b.assert_zero(b.is_transition * gate * (next_in[i * D + d] - local_out[i * D + d]));
  }
  for (int i = 0; i < RE; ++i) {
    V gate = in_limb(PN, i, 3) * next_bit;
    for (int d = 0; d < D; ++d)
      b.assert_zero(b.is_transition * gate * (next_in[(RE + i) * D + d] - local_out[i * D + d]));
  }
  {
    V not_next_new_start = one - PN[22];
    b.assert_zero(b.is_transition * not_next_new_start * PN[23] *
                  (next_index_sum - (mmcs_index_sum * b.K(2) + next_bit)));
  }

  eval_poseidon2_perm<FP, V>(p2, b);
}

// ---- Poseidon2CircuitAir, compact D1 layout (width 16, rate 8, one witness per state element) on a witness bus of
// WB = D value slots (KoalaBearD1Width16WitnessBus5 for D = 5 circuits: batch_stark_prover/poseidon2.rs:1244-1285) ----
// preprocessed row (air.rs:730-763, poseidon-circuit-cols preprocessed.rs:121-145), 62 columns:
//   [0..8) in_ctl | 8 cap_tag | 9 cap_chain_enable | [10..18) rate sponge-chain sel | [18..26) rate Merkle-chain sel
//   | [26..42) input idx | [42..50) output idx | [50..58) out_ctl | 58 mmcs idx | 59 mmcs_merkle_flag | 60 new_start
//   | 61 merkle_path
// constraints air.rs:937-1031, interactions :1721-1785.
template <class FP, class V>
void eval_poseidon2_d1(const AirDesc& a, const Poseidon2<FP>& p2, EvalCtx<FP, V>& b) {
  constexpr int WE = 16, RE = 8, R = FP::SBOX_REGS, HDR = 26, TAIL = HDR + WE + RE + RE;
  const int WB = a.D;
  const int pc = Poseidon2<FP>::perm_cols();
  const V* L = b.local; const V* N = b.next; const V* PL = b.prep_local; const V* PN = b.prep_next;
  const int ending_off = WIDTH + HALF_FULL * (WIDTH * R + WIDTH) + FP::PARTIAL * (R + 1);
  const V* local_out = L + ending_off + (HALF_FULL - 1) * (WIDTH * R + WIDTH) + WIDTH * R;
  const V* next_in = N;
  const V mmcs_bit = L[pc], mmcs_index_sum = L[pc + 1];
  const V next_bit = N[pc], next_index_sum = N[pc + 1];
  const V one = b.K(1), zero = b.K(0);

  // --- interactions ---
  {
    const V not_merkle = one - PL[TAIL + 3];
    for (int l = 0; l < RE; ++l) {
      std::vector<V> f{PL[HDR + l], L[l]};
      for (int d = 1; d < WB; ++d) f.push_back(zero);
      b.push_interaction(std::move(f), zero - PL[l] * not_merkle);
    }
    for (int l = 0; l < RE; ++l) {
      std::vector<V> f{PL[HDR + WE + l], local_out[l]};
      for (int d = 1; d < WB; ++d) f.push_back(zero);
      b.push_interaction(std::move(f), PL[HDR + WE + RE + l]);
    }
    std::vector<V> f{PL[TAIL], mmcs_index_sum};
    for (int d = 1; d < WB; ++d) f.push_back(zero);
    b.push_interaction(std::move(f), zero - PL[TAIL + 1] * PN[TAIL + 2]);
  }

  // --- circuit constraints ---
  b.assert_zero(mmcs_bit * (one - mmcs_bit));
  const V cap_chain_enable = PN[RE + 1], cap_tag = PN[RE];
  const V next_new_start = PN[TAIL + 2], next_merkle_path = PN[TAIL + 3];
  const V not_merkle = one - next_merkle_path;
  for (int l = 0; l < RE; ++l)
    b.assert_zero(b.is_transition * PN[RE + 2 + l] * (next_in[l] - local_out[l]));
  for (int l = RE; l < WE; ++l) {
    const V tag = l == RE ? cap_tag : zero;
    b.assert_zero(b.is_transition * (cap_chain_enable * not_merkle) * (next_in[l] - local_out[l] - tag));
  }
  const V is_left = one - next_bit;
  for (int i = 0; i < RE; ++i) {
    const V sel = PN[RE + 2 + RE + i];
    b.assert_zero(b.is_transition * (sel * is_left) * (next_in[i] - local_out[i]));
    b.assert_zero(b.is_transition * (sel * next_bit) * (next_in[RE + i] - local_out[i]));
  }
  for (int l = RE; l < WE; ++l) {
    const V tag = l == RE ? cap_tag : zero;
    b.assert_zero(b.is_transition * next_new_start * not_merkle * (next_in[l] - tag));
  }
  b.assert_zero(b.is_transition * (one - next_new_start) * next_merkle_path *
                (next_index_sum - (mmcs_index_sum * b.K(2) + next_bit)));
  eval_poseidon2_perm<FP, V>(p2, b);
}

// ---- Poseidon2CircuitAir, arity-4 compression shape (D = 4, width 32: WIDTH_EXT = 8, RATE_EXT = 6, CAPACITY_EXT = 2) ----
// interactions poseidon2-circuit-air/src/air.rs:1790-1870 (non-compact branch, is_arity4); circuit constraints
// eval_arity4 :1178-1342; the inner permutation AIR over the first perm_cols columns as for width 16.
template <class FP, class V>
void eval_poseidon2_w32_perm(const Poseidon2W32<FP>& p2, EvalCtx<FP, V>& b) {
  constexpr int R = FP::SBOX_REGS, W = WIDTH32;
  const V* L = b.local;
  const V zero = b.K(0);
  auto full_off = [&](int r) { return W + r * (W * R + W); };
  const int partial_off = full_off(HALF_FULL);
  const int ending_off = partial_off + FP::PARTIAL_W32 * (R + 1);
  auto end_off = [&](int r) { return ending_off + r * (W * R + W); };
  std::array<V, W> s;
  for (int i = 0; i < W; ++i) s[i] = L[i];
  auto external = [&](std::array<V, W>& st) {
    std::array<V, W> o;
    for (int i = 0; i < W; ++i) {
      V acc = zero;
      for (int j = 0; j < W; ++j) acc = acc + st[j] * b.KF(Poseidon2W32<FP>::ext_entry(i, j));
      o[i] = acc;
    }
    st = o;
  };
  auto internal = [&](std::array<V, W>& st) {
    V sum = zero;
    for (auto& x : st) sum = sum + x;
    for (int i = 0; i < W; ++i) st[i] = st[i] * b.KF(p2.diag[i]) + sum;
  };
  auto sbox = [&](V x, const V* reg) -> V {
    if (FP::SBOX_DEGREE == 3) return x * x * x;
    V c3 = reg[0];
    b.assert_zero(c3 - x * x * x);
    return c3 * c3 * x;
  };
  external(s);
  int k = 0;
  auto full_round = [&](int col) {
    for (int i = 0; i < W; ++i) {
      V x = s[i] + b.KF(p2.rc[k + i]);
      s[i] = sbox(x, L + col + i * R);
    }
    k += W;
    external(s);
    const V* post = L + col + W * R;
    for (int i = 0; i < W; ++i) {
      b.assert_zero(s[i] - post[i]);
      s[i] = post[i];
    }
  };
  for (int r = 0; r < HALF_FULL; ++r) full_round(full_off(r));
  for (int r = 0; r < FP::PARTIAL_W32; ++r) {
    int col = partial_off + r * (R + 1);
    V x = s[0] + b.KF(p2.rc[k++]);
    s[0] = sbox(x, L + col);
    b.assert_zero(s[0] - L[col + R]);
    s[0] = L[col + R];
    internal(s);
  }
  for (int r = 0; r < HALF_FULL; ++r) full_round(end_off(r));
}

template <class FP, class V>
void eval_poseidon2_w32(const Poseidon2W32<FP>& p2, EvalCtx<FP, V>& b) {
  constexpr int D = 4, WE = 8, RE = 6, CE = 2, R = FP::SBOX_REGS, W = WIDTH32;
  const int pc = Poseidon2W32<FP>::perm_cols();
  const V* L = b.local; const V* N = b.next; const V* PL = b.prep_local; const V* PN = b.prep_next;
  const int out_off = pc - W;   // ending_full_rounds[3].post: the last W columns of Poseidon2Cols
  const V* local_out = L + out_off;
  const V* next_in = N;
  const V bit = L[pc], bit2 = L[pc + 1], bit_x_bit2 = L[pc + 2], index_sum = L[pc + 3];
  const V next_bit = N[pc], next_bit2 = N[pc + 1], next_bit_x_bit2 = N[pc + 2], next_index_sum = N[pc + 3];
  (void)R;
  // prep row: input_limbs[8]{idx, in_ctl, normal_chain_sel, merkle_chain_sel} | output_limbs[6]{idx, out_ctl}
  //           | mmcs_index_sum_ctl_idx (here: witness of bit 0) | mmcs_merkle_flag (here: witness of bit 1) | new_start | merkle_path
  auto in_limb = [&](const V* P, int l, int f) { return P[l * 4 + f]; };
  auto out_limb = [&](const V* P, int l, int f) { return P[WE * 4 + l * 2 + f]; };
  const int tail = WE * 4 + RE * 2;
  const V one = b.K(1), zero = b.K(0);

  // --- interactions (air.rs:1790-1870) ---
  for (int l = 0; l < WE; ++l) {
    std::vector<V> f{in_limb(PL, l, 0)};
    for (int d = 0; d < D; ++d) f.push_back(L[l * D + d]);
    b.push_interaction(std::move(f), zero - in_limb(PL, l, 1));   // arity-4: the bare in_ctl (pads and injected slots only)
  }
  for (int l = 0; l < RE; ++l) {
    std::vector<V> f{out_limb(PL, l, 0)};
    for (int d = 0; d < D; ++d) f.push_back(local_out[l * D + d]);
    b.push_interaction(std::move(f), out_limb(PL, l, 1));
  }
  {
    const V merkle = PL[tail + 3];
    std::vector<V> f0{PL[tail], bit};
    for (int d = 1; d < D; ++d) f0.push_back(zero);
    b.push_interaction(std::move(f0), zero - merkle);
    std::vector<V> f1{PL[tail + 1], bit2};
    for (int d = 1; d < D; ++d) f1.push_back(zero);
    b.push_interaction(std::move(f1), zero - merkle);
  }

  // --- circuit constraints (eval_arity4) ---
  b.assert_zero(bit * (one - bit));
  b.assert_zero(bit2 * (one - bit2));
  b.assert_zero(bit_x_bit2 - bit * bit2);
  for (int l = 0; l < WE; ++l)
    for (int d = 0; d < D; ++d)
      b.assert_zero(b.is_transition * in_limb(PN, l, 2) * (next_in[l * D + d] - local_out[l * D + d]));
  const V h[4] = {one - next_bit - next_bit2 + next_bit_x_bit2, next_bit - next_bit_x_bit2, next_bit2 - next_bit_x_bit2, next_bit_x_bit2};
  for (int chunk = 0; chunk < 4; ++chunk)
    for (int slot = 0; slot < CE; ++slot) {
      const int g = chunk * CE + slot;
      const V gate = in_limb(PN, g, 3) * h[chunk];
      for (int d = 0; d < D; ++d)
        b.assert_zero(b.is_transition * gate * (next_in[g * D + d] - local_out[slot * D + d]));
    }
  b.assert_zero(b.is_transition * (one - PN[tail + 2]) * PN[tail + 3] *
                (next_index_sum - (index_sum * b.K(4) + next_bit + b.K(2) * next_bit2)));

  eval_poseidon2_w32_perm<FP, V>(p2, b);
}

template <class FP, class V>
void eval_air(const AirDesc& a, const Poseidon2<FP>& p2, EvalCtx<FP, V>& b) {
  switch (a.kind) {
    case AIR_CONST:
    case AIR_PUBLIC: eval_witness_send<FP, V>(a, b); break;
    case AIR_ALU: eval_alu<FP, V>(a, b); break;
    case AIR_POSEIDON2:
      if (a.D == 4) eval_poseidon2<FP, V>(p2, b);
      else eval_poseidon2_d1<FP, V>(a, p2, b);
      break;
    case AIR_RECOMPOSE: eval_recompose<FP, V>(a, b); break;
    case AIR_POSEIDON2_W32:
      if (a.D != 4 || !p2.w32) throw std::runtime_error("the width-32 Poseidon2 table is the D = 4 one and needs its constants");
      eval_poseidon2_w32<FP, V>(*p2.w32, b);
      break;
    default: throw std::runtime_error("bad air kind");
  }
}

}  // namespace orc
