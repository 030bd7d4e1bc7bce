// Statements of the five recursion-table AIRs: bus interactions (LogUp K7) and constraint
// folding (quotient K8).  They are generic in the VIEW they read: on the device a view is one
// row of the column-major matrices (one lane = one row, every access a coalesced load) and
// values are base-field elements; the verifier (verify_impl.h) evaluates the same statements
// at zeta, where a view is the opened values and a value is an extension element.
//
// Reference anchors (constraint ORDER follows the `assert_*` order of these functions):
//   WitnessSendAir (Const/Public)  circuit-prover/src/air/public_air.rs:209-239
//   RecomposeAir                   circuit-prover/src/air/recompose_air.rs:141-198
//   AluAir                         circuit-prover/src/air/alu_air.rs:764-996 (constraints),
//                                  :1000-1085 (interactions), columns alu_columns.rs:9-46
//   Poseidon2CircuitAir            poseidon2-circuit-air/src/air.rs:937,1049-1122 (circuit
//                                  constraints), :1137-1158 (inner permutation AIR),
//                                  :1790-1893 (interactions); prep row preprocessed.rs:160-
#pragma once
#include "run_schedule.h"
#include "field.h"
#include "poseidon2.h"

namespace p3r {

// AIR_POSEIDON2_W32: the width-32 D = 4 table of the arity-4 MMCS (Poseidon2CircuitAir{Koala,Baby}BearD4Width32,
// poseidon2-circuit-air/src/public_types.rs:179-187,396-404)
enum AirKind { AIR_CONST = 0, AIR_PUBLIC = 1, AIR_ALU = 2, AIR_POSEIDON2 = 3, AIR_RECOMPOSE = 4, AIR_POSEIDON2_W32 = 5 };
// its preprocessed row (Poseidon2PreprocessedRow<WIDTH_EXT = 8, RATE_EXT = 6>, poseidon-circuit-cols/src/preprocessed.rs):
//   8 x {idx, in_ctl, normal_chain_sel, merkle_chain_sel} | 6 x {idx, out_ctl} | bit-0 witness | bit-1 witness | new_start | merkle_path
constexpr int kP2WPrepWidth = 48, kP2WOutLimbs = 32, kP2WTail = 44;

constexpr int kMaxExtD = 8;  // widest circuit extension: bus tuples hold at most 1 + kMaxExtD fields
// compact-D1 Poseidon2 preprocessed row (poseidon-circuit-cols/src/preprocessed.rs:121-145), 62 columns:
//   [0..8) in_ctl | 8 length tag | 9 cap_chain_enable | [10..18) rate sponge-chain sel | [18..26) rate Merkle-chain sel
//   | [26..42) input idx | [42..50) output idx | [50..58) out_ctl | 58 mmcs idx | 59 mmcs_merkle_flag | 60 new_start
//   | 61 merkle_path
// (kP2D1Hdr, kP2D1Tail, kP2D1PrepWidth: run_schedule.h - the device-side preparation writes the same rows)

struct AirParams {
  int kind;
  int lanes;
  int horner_k;
  int coeff_lookups;
  int lookup_unpacked = 0;  // p3r_config.ext_choices & P3R_EXT_LOOKUP_UNPACKED
  // Circuit extension degree D of the table's witness values (p3r_config.ext_degree): 1 = base-field circuits,
  // 4 = binomial x^4 = W, 5 = KoalaBear quintic trinomial x^5 + x^2 - 1.  Bus tuples are (idx, v_0..v_{D-1}); the
  // Poseidon2 table is the D4 width-16 one for D = 4 and the compact-D1 one (on a D-slot bus) for D = 1 / 5.
  // D = 2, 6, 8: the binomial extension x^D = W with W = ext_w_mont (Montgomery word; p3r_config.ext_w), the
  // primitive tables and Recompose.
  int ext_d = 4;
  uint32_t ext_w_mont = 0;
};

// Row window over column-major main / preprocessed matrices of a common height.
template <class PP>
struct RowView {
  using F = Fp<PP>;
  gptr<const uint32_t> main;  // (global address space: see as_global, field.h)
  gptr<const uint32_t> prep;
  size_t h;         // matrix height (trace height for K7, LDE height for K8)
  size_t row, nxt;  // local row and "next" row (already wrapped / bit-reverse mapped)
  using V = F;
  __device__ __forceinline__ F L(int c) const { return F::raw(main[(size_t)c * h + row]); }
  __device__ __forceinline__ F N(int c) const { return F::raw(main[(size_t)c * h + nxt]); }
  __device__ __forceinline__ F PL(int c) const { return F::raw(prep[(size_t)c * h + row]); }
  __device__ __forceinline__ F PN(int c) const { return F::raw(prep[(size_t)c * h + nxt]); }
};

template <class F, int D>
struct VD {
  F c[D];
};
template <class F>
using V4 = VD<F, 4>;
template <int D, class V, class G>
P3R_HD VD<V, D> loadD(G&& get, int col) {
  VD<V, D> r;
#pragma unroll
  for (int i = 0; i < D; ++i) r.c[i] = get(col + i);
  return r;
}
template <class V, class G>
P3R_HD V4<V> load4(G&& get, int col) { return loadD<4, V>(get, col); }
// x*y in the circuit's extension: F[x]/(x^4 - W) (alu_air.rs:715-733) or, for D = 5, F[x]/(x^5 + x^2 - 1)
// (ext_mul_quintic_trinomial, alu_air.rs:735-762: x^5 = 1 - x^2, x^6 = x - x^3, x^7 = x^2 - x^4,
// x^8 = x^3 + x^2 - 1)
template <class PP, int D, class V>
P3R_HD VD<V, D> mulD(const VD<V, D>& a, const VD<V, D>& b, uint32_t w_mont = 0) {
  VD<V, D> r;
  if constexpr (D == 2 || D == 6 || D == 8) {
    // generic binomial x^D = W (ext_mul_binomial, alu_air.rs:715-733), W a run-time value
    const V W = Lift<V>::of(Fp<PP>::raw(w_mont));
    V lo[D], hi[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { lo[k] = V::zero(); hi[k] = V::zero(); }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        if (i + j < D) lo[i + j] = lo[i + j] + a.c[i] * b.c[j];
        else hi[i + j - D] = hi[i + j - D] + a.c[i] * b.c[j];
      }
#pragma unroll
    for (int k = 0; k < D; ++k) r.c[k] = lo[k] + W * hi[k];
  } else if constexpr (D == 1) {
    r.c[0] = a.c[0] * b.c[0];   // base-field circuits
  } else if constexpr (D == 4) {
    const V W = Lift<V>::of(Fp<PP>::from_canonical(PP::EXT_W));
    r.c[0] = a.c[0] * b.c[0] + W * (a.c[1] * b.c[3] + a.c[2] * b.c[2] + a.c[3] * b.c[1]);
    r.c[1] = a.c[0] * b.c[1] + a.c[1] * b.c[0] + W * (a.c[2] * b.c[3] + a.c[3] * b.c[2]);
    r.c[2] = a.c[0] * b.c[2] + a.c[1] * b.c[1] + a.c[2] * b.c[0] + W * (a.c[3] * b.c[3]);
    r.c[3] = a.c[0] * b.c[3] + a.c[1] * b.c[2] + a.c[2] * b.c[1] + a.c[3] * b.c[0];
  } else {
    static_assert(D == 5, "circuit extension degree must be 1, 4 or 5");
    const V c5 = a.c[1] * b.c[4] + a.c[2] * b.c[3] + a.c[3] * b.c[2] + a.c[4] * b.c[1];
    const V c6 = a.c[2] * b.c[4] + a.c[3] * b.c[3] + a.c[4] * b.c[2];
    const V c7 = a.c[3] * b.c[4] + a.c[4] * b.c[3];
    const V c8 = a.c[4] * b.c[4];
    const V c58 = c5 - c8;
    r.c[0] = a.c[0] * b.c[0] + c58;
    r.c[1] = a.c[0] * b.c[1] + a.c[1] * b.c[0] + c6;
    r.c[2] = a.c[0] * b.c[2] + a.c[1] * b.c[1] + a.c[2] * b.c[0] - c58 + c7;
    r.c[3] = a.c[0] * b.c[3] + a.c[1] * b.c[2] + a.c[2] * b.c[1] + a.c[3] * b.c[0] - c6 + c8;
    r.c[4] = a.c[0] * b.c[4] + a.c[1] * b.c[3] + a.c[2] * b.c[2] + a.c[3] * b.c[1] + a.c[4] * b.c[0] - c7;
  }
  return r;
}
template <class PP, class V>
P3R_HD V4<V> mul4(const V4<V>& a, const V4<V>& b) { return mulD<PP, 4, V>(a, b); }

// ---------------------------------------------------------------- interactions (push order)
// sink.add(idx, v, mult): one bus tuple (idx, v_0..v_{D-1}) with signed multiplicity.  D is the circuit
// extension degree of the primitive tables (a.ext_d); the Poseidon2 and Recompose tables exist for D = 4 only.
template <class PP, int D = 4, class View, class Sink>
P3R_HD void air_interactions(const AirParams& a, const View& v, Sink& sink) {
  using F = typename View::V;  // value type of the view
  auto L = [&](int c) { return v.L(c); };
  switch (a.kind) {
    case AIR_CONST:
    case AIR_PUBLIC:
      for (int lane = 0; lane < a.lanes; ++lane)
        sink.add(v.PL(lane * 2 + 1), loadD<D, F>(L, lane * D), v.PL(lane * 2));
      break;
    case AIR_RECOMPOSE: {
      // output tuple (idx, v_0..v_{D-1}); the "recompose/coeff" variant adds one tuple (idx_i, v_i, 0, ..) per
      // coefficient (recompose_air.rs:196-226)
      const int plw = 2 + (a.coeff_lookups ? 2 * D : 0);
      for (int lane = 0; lane < a.lanes; ++lane) {
        sink.add(v.PL(lane * plw), loadD<D, F>(L, lane * D), v.PL(lane * plw + 1));
        if (a.coeff_lookups)
          for (int i = 0; i < D; ++i) {
            VD<F, D> t;
            t.c[0] = v.L(lane * D + i);
#pragma unroll
            for (int j = 1; j < D; ++j) t.c[j] = F::zero();
            sink.add(v.PL(lane * plw + 2 + 2 * i), t, v.PL(lane * plw + 3 + 2 * i));
          }
      }
    } break;
    case AIR_ALU: {
      const int lanes = a.lanes, k_max = a.horner_k;
      for (int lane = 0; lane < lanes; ++lane) {
        const int m = lane * 4 * D, p = lane * 13;
        F mult_a = v.PL(p), a_rd = v.PL(p + 11), c_rd = v.PL(p + 12);
        sink.add(v.PL(p + 5), loadD<D, F>(L, m), mult_a * a_rd);
        sink.add(v.PL(p + 6), loadD<D, F>(L, m + D), v.PL(p + 9));
        sink.add(v.PL(p + 7), loadD<D, F>(L, m + 2 * D), mult_a * c_rd);
        sink.add(v.PL(p + 8), loadD<D, F>(L, m + 3 * D), v.PL(p + 10));
      }
      const int extra_main = lanes * 4 * D, extra_prep = lanes * 13;
      const int ac_base = extra_main + ((k_max - 1) / 2) * D;
      for (int t = 1; t < k_max; ++t) {
        const int sp = extra_prep + (k_max - 1) + 6 * (t - 1);
        const int off = ac_base + 2 * D * (t - 1);
        sink.add(v.PL(sp), loadD<D, F>(L, off), v.PL(sp + 4));
        sink.add(v.PL(sp + 1), loadD<D, F>(L, off + D), v.PL(sp + 5));
      }
    } break;
    case AIR_POSEIDON2:
      if constexpr (D == 4) {
        constexpr int pc = p2_perm_cols<PP>();
        constexpr int out_col = pc - P2_WIDTH;  // ending_full_rounds[3].post
        F not_merkle = F::one() - v.PL(23);
        for (int l = 0; l < 4; ++l)
          sink.add(v.PL(l * 4), load4<F>(L, l * 4), -(v.PL(l * 4 + 1) * not_merkle));
        for (int l = 0; l < 2; ++l)
          sink.add(v.PL(16 + l * 2), load4<F>(L, out_col + l * 4), v.PL(16 + l * 2 + 1));
        V4<F> t;
        t.c[0] = v.L(pc + 1);
        t.c[1] = t.c[2] = t.c[3] = F::zero();
        sink.add(v.PL(20), t, -(v.PL(21) * v.PN(22)));
      } else if constexpr (D == 1 || D == 5) {
        // compact-D1 table on the D-slot witness bus (KoalaBearD1Width16WitnessBus5; air.rs:1721-1785): 8 rate
        // sends, 8 output receives, the accumulator send; tuples (idx, v, 0, ..)
        constexpr int pc = p2_perm_cols<PP>();
        constexpr int out_col = pc - P2_WIDTH;
        const F not_merkle = F::one() - v.PL(kP2D1Tail + 3);
        VD<F, D> t;
#pragma unroll
        for (int i = 1; i < D; ++i) t.c[i] = F::zero();
        for (int l = 0; l < 8; ++l) {
          t.c[0] = v.L(l);
          sink.add(v.PL(kP2D1Hdr + l), t, -(v.PL(l) * not_merkle));
        }
        for (int l = 0; l < 8; ++l) {
          t.c[0] = v.L(out_col + l);
          sink.add(v.PL(kP2D1Hdr + 16 + l), t, v.PL(kP2D1Hdr + 24 + l));
        }
        t.c[0] = v.L(pc + 1);
        sink.add(v.PL(kP2D1Tail), t, -(v.PL(kP2D1Tail + 1) * v.PN(kP2D1Tail + 2)));
      }
      break;
    case AIR_POSEIDON2_W32:
      // non-compact branch with is_arity4 (air.rs:1790-1870): 8 input sends with the bare in_ctl (pads and injected
      // slots only), 6 output receives, and the two direction bits read from their witnesses on Merkle rows
      if constexpr (D == 4) {
        constexpr int pc = p2w_perm_cols<PP>();
        constexpr int out_col = pc - P2W_WIDTH;
        for (int l = 0; l < 8; ++l) sink.add(v.PL(l * 4), load4<F>(L, l * 4), -v.PL(l * 4 + 1));
        for (int l = 0; l < 6; ++l) sink.add(v.PL(kP2WOutLimbs + l * 2), load4<F>(L, out_col + l * 4), v.PL(kP2WOutLimbs + l * 2 + 1));
        const F neg_merkle = -v.PL(kP2WTail + 3);
        V4<F> t;
        t.c[1] = t.c[2] = t.c[3] = F::zero();
        t.c[0] = v.L(pc);
        sink.add(v.PL(kP2WTail), t, neg_merkle);
        t.c[0] = v.L(pc + 1);
        sink.add(v.PL(kP2WTail + 1), t, neg_merkle);
      }
      break;
  }
}

// ---------------------------------------------------------------- constraints
// fold.base(c): the next base-field constraint value, in declaration order.
template <class PP, int D = 4, class View, class Fold>
P3R_HD void alu_constraints(const AirParams& a, const View& v, Fold& fold) {
  using F = typename View::V;
  auto L = [&](int c) { return v.L(c); };
  auto N = [&](int c) { return v.N(c); };
  const int lanes = a.lanes, k_max = a.horner_k;
  const int extra_main = lanes * 4 * D, extra_prep = lanes * 13;
  const int num_int = (k_max - 1) / 2;
  const int ac_base = extra_main + num_int * D;
  const F one = F::one();
  for (int lane = 0; lane < lanes; ++lane) {
    const int m = lane * 4 * D, p = lane * 13;
    VD<F, D> A = loadD<D, F>(L, m), B = loadD<D, F>(L, m + D), C = loadD<D, F>(L, m + 2 * D), O = loadD<D, F>(L, m + 3 * D);
    F mult_a = v.PL(p), sel_add = v.PL(p + 1), sel_bool = v.PL(p + 2), sel_muladd = v.PL(p + 3),
      sel_horner = v.PL(p + 4);
    F sel_mul = -mult_a - sel_bool - sel_muladd - sel_horner - sel_add;
#pragma unroll
    for (int i = 0; i < D; ++i) fold.base(sel_add * (A.c[i] + B.c[i] - O.c[i]));
    VD<F, D> ab = mulD<PP, D, F>(A, B, a.ext_w_mont);
#pragma unroll
    for (int i = 0; i < D; ++i) fold.base(sel_mul * (ab.c[i] - O.c[i]));
    fold.base(sel_bool * A.c[0] * (A.c[0] - one));
#pragma unroll
    for (int i = 1; i < D; ++i) fold.base(sel_bool * A.c[i]);
#pragma unroll
    for (int i = 0; i < D; ++i) fold.base(sel_muladd * (ab.c[i] + C.c[i] - O.c[i]));
    F next_sel_horner = v.PN(p + 4);
    VD<F, D> NA = loadD<D, F>(N, m), NB = loadD<D, F>(N, m + D), NC = loadD<D, F>(N, m + 2 * D), NO = loadD<D, F>(N, m + 3 * D);
    VD<F, D> out_next_b = mulD<PP, D, F>(O, NB, a.ext_w_mont);
    if (lane == 0) {
      F any_cur = F::zero(), any_next = F::zero(), sel_ge3_next = F::zero();
      for (int kk = 2; kk <= k_max; ++kk) any_cur += v.PL(extra_prep + kk - 2);
      for (int kk = 2; kk <= k_max; ++kk) any_next += v.PN(extra_prep + kk - 2);
      F next_sel_k2 = v.PN(extra_prep);
      for (int kk = 3; kk <= k_max; ++kk) sel_ge3_next += v.PN(extra_prep + kk - 2);
      const int b_sq_base = ac_base + 2 * D * (k_max - 1);
      VD<F, D> b_sq = loadD<D, F>(L, b_sq_base), b_sq_next = loadD<D, F>(N, b_sq_base);
      VD<F, D> bb = mulD<PP, D, F>(B, B, a.ext_w_mont);
#pragma unroll
      for (int i = 0; i < D; ++i) fold.base(any_cur * (b_sq.c[i] - bb.c[i]));
      VD<F, D> out_b_sq = mulD<PP, D, F>(O, b_sq_next, a.ext_w_mont), c0b = mulD<PP, D, F>(NC, NB, a.ext_w_mont), a0b = mulD<PP, D, F>(NA, NB, a.ext_w_mont);
      VD<F, D> a1n = loadD<D, F>(N, ac_base), c1n = loadD<D, F>(N, ac_base + D), int0n = loadD<D, F>(N, extra_main);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        F poly = out_b_sq.c[i] + c0b.c[i] - a0b.c[i] + c1n.c[i] - a1n.c[i];
        fold.base(next_sel_k2 * (poly - NO.c[i]));
        fold.base(sel_ge3_next * (poly - int0n.c[i]));
      }
      F next_sel_single = next_sel_horner - any_next;
#pragma unroll
      for (int i = 0; i < D; ++i) fold.base(next_sel_single * (out_next_b.c[i] + NC.c[i] - NA.c[i] - NO.c[i]));
      for (int kk = 3; kk <= k_max; ++kk) {
        F sel_kk = v.PL(extra_prep + kk - 2);
        int s = 2, slot = 0;
        while (s < kk) {
          VD<F, D> int_curr = loadD<D, F>(L, extra_main + slot * D);
          const int off_s = ac_base + 2 * D * (s - 1);
          VD<F, D> a_s = loadD<D, F>(L, off_s), c_s = loadD<D, F>(L, off_s + D);
          if (s + 1 < kk) {
            const int off_sp1 = ac_base + 2 * D * s;
            VD<F, D> a_sp1 = loadD<D, F>(L, off_sp1), c_sp1 = loadD<D, F>(L, off_sp1 + D);
            VD<F, D> int_b_sq = mulD<PP, D, F>(int_curr, b_sq, a.ext_w_mont), c_s_b = mulD<PP, D, F>(c_s, B, a.ext_w_mont), a_s_b = mulD<PP, D, F>(a_s, B, a.ext_w_mont);
            const bool to_out = s + 2 >= kk;
            VD<F, D> target = to_out ? O : loadD<D, F>(L, extra_main + (slot + 1) * D);
#pragma unroll
            for (int i = 0; i < D; ++i)
              fold.base(sel_kk * (int_b_sq.c[i] + c_s_b.c[i] - a_s_b.c[i] + c_sp1.c[i] - a_sp1.c[i] - target.c[i]));
            if (!to_out) slot += 1;
            s += 2;
          } else {
            VD<F, D> int_b = mulD<PP, D, F>(int_curr, B, a.ext_w_mont);
#pragma unroll
            for (int i = 0; i < D; ++i) fold.base(sel_kk * (int_b.c[i] + c_s.c[i] - a_s.c[i] - O.c[i]));
            s += 1;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i) fold.base(next_sel_horner * (out_next_b.c[i] + NC.c[i] - NA.c[i] - NO.c[i]));
    }
  }
}

// p3_poseidon2_air::eval over the permutation columns (poseidon2-circuit-air/src/air.rs:1137-1158)
template <class PP, class View, class Fold>
P3R_HD void poseidon2_perm_constraints(const View& v, const uint32_t* __restrict__ rc, Fold& fold) {
  using F = typename View::V;
  using B = Fp<PP>;
  constexpr int R = PP::SBOX_REGISTERS;
  F s[P2_WIDTH];
#pragma unroll
  for (int i = 0; i < P2_WIDTH; ++i) s[i] = v.L(i);
  p2_external_linear(s);
  int k = 0, col = P2_WIDTH;
  auto full_round = [&]() {
#pragma unroll
    for (int i = 0; i < P2_WIDTH; ++i) {
      F x = s[i] + Lift<F>::of(B::raw(rc[k + i]));
      if (R == 1) {
        F c3 = v.L(col + i);
        fold.base(c3 - x.sqr() * x);
        s[i] = c3.sqr() * x;
      } else {
        s[i] = x.sqr() * x;
      }
    }
    k += P2_WIDTH;
    col += P2_WIDTH * R;
    p2_external_linear(s);
#pragma unroll
    for (int i = 0; i < P2_WIDTH; i += 2) {
      F post0 = v.L(col + i), post1 = v.L(col + i + 1);
      fold.base2(s[i] - post0, s[i + 1] - post1);
      s[i] = post0;
      s[i + 1] = post1;
    }
    col += P2_WIDTH;
  };
  for (int r = 0; r < P2_HALF_FULL; ++r) full_round();
  for (int r = 0; r < PP::PARTIAL_ROUNDS; ++r) {
    F x = s[0] + Lift<F>::of(B::raw(rc[k++]));
    if (R == 1) {
      F c3 = v.L(col);
      fold.base(c3 - x.sqr() * x);
      s[0] = c3.sqr() * x;
      col += 1;
    } else {
      s[0] = x.sqr() * x;
    }
    F post = v.L(col);
    fold.base(s[0] - post);
    s[0] = post;
    col += 1;
    p2_internal_linear<PP, F>(s);
  }
  for (int r = 0; r < P2_HALF_FULL; ++r) full_round();
}

template <class PP, class View, class Fold>
P3R_HD void poseidon2_constraints(const View& v, typename View::V is_transition, const uint32_t* __restrict__ rc,
                                  Fold& fold) {
  using F = typename View::V;
  constexpr int pc = p2_perm_cols<PP>();
  constexpr int out_col = pc - P2_WIDTH;
  const F one = F::one();
  const F mmcs_bit = v.L(pc), index_sum = v.L(pc + 1), next_bit = v.N(pc), next_index_sum = v.N(pc + 1);
  fold.base(mmcs_bit * (one - mmcs_bit));
  // sponge chaining
  for (int l = 0; l < 4; ++l) {
    F gate = is_transition * v.PN(l * 4 + 2);
#pragma unroll
    for (int d = 0; d < 4; d += 2)
      fold.base2(gate * (v.N(l * 4 + d) - v.L(out_col + l * 4 + d)), gate * (v.N(l * 4 + d + 1) - v.L(out_col + l * 4 + d + 1)));
  }
  // Merkle chaining, left then right placement
  F is_left = one - next_bit;
  for (int i = 0; i < 2; ++i) {
    F gate = is_transition * (v.PN(i * 4 + 3) * is_left);
#pragma unroll
    for (int d = 0; d < 4; ++d) fold.base(gate * (v.N(i * 4 + d) - v.L(out_col + i * 4 + d)));
  }
  for (int i = 0; i < 2; ++i) {
    F gate = is_transition * (v.PN(i * 4 + 3) * next_bit);
#pragma unroll
    for (int d = 0; d < 4; ++d) fold.base(gate * (v.N((2 + i) * 4 + d) - v.L(out_col + i * 4 + d)));
  }
  fold.base(is_transition * (one - v.PN(22)) * v.PN(23) * (next_index_sum - (index_sum.dbl() + next_bit)));
  poseidon2_perm_constraints<PP>(v, rc, fold);
}

// The compact-D1 table (one witness per state element): air.rs:937-1031.
template <class PP, class View, class Fold>
P3R_HD void poseidon2_d1_constraints(const View& v, typename View::V is_transition, const uint32_t* __restrict__ rc,
                                     Fold& fold) {
  using F = typename View::V;
  constexpr int pc = p2_perm_cols<PP>();
  constexpr int out_col = pc - P2_WIDTH;
  const F one = F::one();
  const F mmcs_bit = v.L(pc), index_sum = v.L(pc + 1), next_bit = v.N(pc), next_index_sum = v.N(pc + 1);
  fold.base(mmcs_bit * (one - mmcs_bit));
  const F cap_tag = v.PN(8), next_new_start = v.PN(kP2D1Tail + 2), next_merkle = v.PN(kP2D1Tail + 3);
  const F not_merkle = one - next_merkle;
  // sponge chaining: rate limbs by their own selector, capacity by cap_chain_enable (first element += the length tag)
  for (int l = 0; l < 8; ++l) fold.base(is_transition * v.PN(10 + l) * (v.N(l) - v.L(out_col + l)));
  {
    const F gate = is_transition * (v.PN(9) * not_merkle);
    fold.base(gate * (v.N(8) - v.L(out_col + 8) - cap_tag));
    for (int l = 9; l < 16; ++l) fold.base(gate * (v.N(l) - v.L(out_col + l)));
  }
  // Merkle chaining: the running digest goes left or right by the next row's direction bit
  const F is_left = one - next_bit;
  for (int i = 0; i < 8; ++i) {
    const F sel = v.PN(18 + i), d = v.L(out_col + i);
    fold.base(is_transition * (sel * is_left) * (v.N(i) - d));
    fold.base(is_transition * (sel * next_bit) * (v.N(8 + i) - d));
  }
  // sponge chain starts: the capacity is the length tag, then zeros
  {
    const F gate = is_transition * next_new_start * not_merkle;
    fold.base(gate * (v.N(8) - cap_tag));
    for (int l = 9; l < 16; ++l) fold.base(gate * v.N(l));
  }
  fold.base(is_transition * (one - next_new_start) * next_merkle * (next_index_sum - (index_sum.dbl() + next_bit)));
  poseidon2_perm_constraints<PP>(v, rc, fold);
}

// The width-32 table: p3_poseidon2_air::eval over Poseidon2Cols<32> (rcw = the width-32 constant table, poseidon2.h)
template <class PP, class View, class Fold>
P3R_HD void poseidon2w_perm_constraints(const View& v, const uint32_t* __restrict__ rcw, Fold& fold) {
  using F = typename View::V;
  using B = Fp<PP>;
  constexpr int R = PP::SBOX_REGISTERS, W = P2W_WIDTH;
  const uint32_t* diag = rcw + p2w_num_rc<PP>();
  F s[W];
#pragma unroll
  for (int i = 0; i < W; ++i) s[i] = v.L(i);
  p2w_external_linear(s);
  int k = 0, col = W;
  auto full_round = [&]() {
#pragma unroll
    for (int i = 0; i < W; ++i) {
      F x = s[i] + Lift<F>::of(B::raw(rcw[k + i]));
      if (R == 1) {
        F c3 = v.L(col + i);
        fold.base(c3 - x.sqr() * x);
        s[i] = c3.sqr() * x;
      } else {
        s[i] = x.sqr() * x;
      }
    }
    k += W;
    col += W * R;
    p2w_external_linear(s);
#pragma unroll
    for (int i = 0; i < W; ++i) {
      F post = v.L(col + i);
      fold.base(s[i] - post);
      s[i] = post;
    }
    col += W;
  };
  for (int r = 0; r < P2_HALF_FULL; ++r) full_round();
  for (int r = 0; r < PP::PARTIAL_ROUNDS_W32; ++r) {
    F x = s[0] + Lift<F>::of(B::raw(rcw[k++]));
    if (R == 1) {
      F c3 = v.L(col);
      fold.base(c3 - x.sqr() * x);
      s[0] = c3.sqr() * x;
      col += 1;
    } else {
      s[0] = x.sqr() * x;
    }
    F post = v.L(col);
    fold.base(s[0] - post);
    s[0] = post;
    col += 1;
    p2w_internal_linear<PP, F>(s, diag);
  }
  for (int r = 0; r < P2_HALF_FULL; ++r) full_round();
}
// eval_arity4 (poseidon2-circuit-air/src/air.rs:1178-1342): booleanity of both direction bits, the product column,
// sponge chaining, the running-hash placement into chunk pos = bit + 2 * bit2, the base-four index accumulator
template <class PP, class View, class Fold>
P3R_HD void poseidon2w_constraints(const View& v, typename View::V is_transition, const uint32_t* __restrict__ rcw, Fold& fold) {
  using F = typename View::V;
  constexpr int pc = p2w_perm_cols<PP>();
  constexpr int out_col = pc - P2W_WIDTH;
  const F one = F::one();
  const F bit = v.L(pc), bit2 = v.L(pc + 1), bxb = v.L(pc + 2), index_sum = v.L(pc + 3);
  const F nbit = v.N(pc), nbit2 = v.N(pc + 1), nbxb = v.N(pc + 2), next_index_sum = v.N(pc + 3);
  fold.base(bit * (one - bit));
  fold.base(bit2 * (one - bit2));
  fold.base(bxb - bit * bit2);
  for (int l = 0; l < 8; ++l) {
    const F gate = is_transition * v.PN(l * 4 + 2);
#pragma unroll
    for (int d = 0; d < 4; ++d) fold.base(gate * (v.N(l * 4 + d) - v.L(out_col + l * 4 + d)));
  }
  const F h[4] = {one - nbit - nbit2 + nbxb, nbit - nbxb, nbit2 - nbxb, nbxb};
  for (int chunk = 0; chunk < 4; ++chunk)
    for (int slot = 0; slot < 2; ++slot) {
      const int g = chunk * 2 + slot;
      const F gate = is_transition * (v.PN(g * 4 + 3) * h[chunk]);
#pragma unroll
      for (int d = 0; d < 4; ++d) fold.base(gate * (v.N(g * 4 + d) - v.L(out_col + slot * 4 + d)));
    }
  fold.base(is_transition * (one - v.PN(kP2WTail + 2)) * v.PN(kP2WTail + 3) *
            (next_index_sum - (index_sum.dbl().dbl() + nbit + nbit2.dbl())));
  poseidon2w_perm_constraints<PP>(v, rcw, fold);
}

// Number of base constraints of an AIR (host side needs it to size the alpha-power table).
template <class PP>
__host__ __device__ inline int air_num_base_constraints(const AirParams& a) {
  switch (a.kind) {
    case AIR_ALU: {
      int n = 0;
      const int D = a.ext_d;
      for (int lane = 0; lane < a.lanes; ++lane) {
        n += 4 * D;
        if (lane == 0) {
          n += 4 * D;
          for (int kk = 3; kk <= a.horner_k; ++kk) {
            int s = 2;
            while (s < kk) { n += D; s += (s + 1 < kk) ? 2 : 1; }
          }
        } else {
          n += D;
        }
      }
      return n;
    }
    case AIR_POSEIDON2: {
      constexpr int R = PP::SBOX_REGISTERS;
      return (a.ext_d == 4 ? 1 + 16 + 8 + 8 + 1 : 1 + 8 + 8 + 16 + 8 + 1) +
             2 * P2_HALF_FULL * (P2_WIDTH * R + P2_WIDTH) + PP::PARTIAL_ROUNDS * (R + 1);
    }
    case AIR_POSEIDON2_W32: {
      constexpr int R = PP::SBOX_REGISTERS;
      return 3 + 32 + 32 + 1 + 2 * P2_HALF_FULL * (P2W_WIDTH * R + P2W_WIDTH) + PP::PARTIAL_ROUNDS_W32 * (R + 1);
    }
    default: return 0;
  }
}

}  // namespace p3r
