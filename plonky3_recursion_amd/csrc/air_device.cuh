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
#include "field.h"
#include "poseidon2.h"

namespace p3r {

enum AirKind { AIR_CONST = 0, AIR_PUBLIC = 1, AIR_ALU = 2, AIR_POSEIDON2 = 3, AIR_RECOMPOSE = 4 };

struct AirParams {
  int kind;
  int lanes;
  int horner_k;
  int coeff_lookups;
  int lookup_unpacked = 0;  // p3r_config.ext_choices & P3R_EXT_LOOKUP_UNPACKED
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

template <class F>
struct V4 {
  F c[4];
};
template <class V, class G>
P3R_HD V4<V> load4(G&& get, int col) {
  V4<V> r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.c[i] = get(col + i);
  return r;
}
// x*y in F[x]/(x^4 - W) (alu_air.rs:715-733)
template <class PP, class V>
P3R_HD V4<V> mul4(const V4<V>& a, const V4<V>& b) {
  const V W = Lift<V>::of(Fp<PP>::from_canonical(PP::EXT_W));
  V4<V> r;
  r.c[0] = a.c[0] * b.c[0] + W * (a.c[1] * b.c[3] + a.c[2] * b.c[2] + a.c[3] * b.c[1]);
  r.c[1] = a.c[0] * b.c[1] + a.c[1] * b.c[0] + W * (a.c[2] * b.c[3] + a.c[3] * b.c[2]);
  r.c[2] = a.c[0] * b.c[2] + a.c[1] * b.c[1] + a.c[2] * b.c[0] + W * (a.c[3] * b.c[3]);
  r.c[3] = a.c[0] * b.c[3] + a.c[1] * b.c[2] + a.c[2] * b.c[1] + a.c[3] * b.c[0];
  return r;
}

// ---------------------------------------------------------------- interactions (push order)
// sink.add(idx, v, mult): one bus tuple (idx, v0..v3) with signed multiplicity.
template <class PP, class View, class Sink>
P3R_HD void air_interactions(const AirParams& a, const View& v, Sink& sink) {
  using F = typename View::V;  // value type of the view
  auto L = [&](int c) { return v.L(c); };
  switch (a.kind) {
    case AIR_CONST:
    case AIR_PUBLIC:
      for (int lane = 0; lane < a.lanes; ++lane)
        sink.add(v.PL(lane * 2 + 1), load4<F>(L, lane * 4), v.PL(lane * 2));
      break;
    case AIR_RECOMPOSE: {
      const int plw = 2 + (a.coeff_lookups ? 8 : 0);
      for (int lane = 0; lane < a.lanes; ++lane) {
        sink.add(v.PL(lane * plw), load4<F>(L, lane * 4), v.PL(lane * plw + 1));
        if (a.coeff_lookups)
          for (int i = 0; i < 4; ++i) {
            V4<F> t;
            t.c[0] = v.L(lane * 4 + i);
            t.c[1] = t.c[2] = t.c[3] = F::zero();
            sink.add(v.PL(lane * plw + 2 + 2 * i), t, v.PL(lane * plw + 3 + 2 * i));
          }
      }
    } break;
    case AIR_ALU: {
      const int lanes = a.lanes, k_max = a.horner_k;
      for (int lane = 0; lane < lanes; ++lane) {
        const int m = lane * 16, p = lane * 13;
        F mult_a = v.PL(p), a_rd = v.PL(p + 11), c_rd = v.PL(p + 12);
        sink.add(v.PL(p + 5), load4<F>(L, m), mult_a * a_rd);
        sink.add(v.PL(p + 6), load4<F>(L, m + 4), v.PL(p + 9));
        sink.add(v.PL(p + 7), load4<F>(L, m + 8), mult_a * c_rd);
        sink.add(v.PL(p + 8), load4<F>(L, m + 12), v.PL(p + 10));
      }
      const int extra_main = lanes * 16, extra_prep = lanes * 13;
      const int ac_base = extra_main + ((k_max - 1) / 2) * 4;
      for (int t = 1; t < k_max; ++t) {
        const int sp = extra_prep + (k_max - 1) + 6 * (t - 1);
        const int off = ac_base + 8 * (t - 1);
        sink.add(v.PL(sp), load4<F>(L, off), v.PL(sp + 4));
        sink.add(v.PL(sp + 1), load4<F>(L, off + 4), v.PL(sp + 5));
      }
    } break;
    case AIR_POSEIDON2: {
      constexpr int R = PP::SBOX_REGISTERS;
      constexpr int pc = p2_perm_cols<PP>();
      constexpr int out_col = pc - P2_WIDTH;  // ending_full_rounds[3].post
      (void)R;
      F not_merkle = F::one() - v.PL(23);
      for (int l = 0; l < 4; ++l)
        sink.add(v.PL(l * 4), load4<F>(L, l * 4), -(v.PL(l * 4 + 1) * not_merkle));
      for (int l = 0; l < 2; ++l)
        sink.add(v.PL(16 + l * 2), load4<F>(L, out_col + l * 4), v.PL(16 + l * 2 + 1));
      V4<F> t;
      t.c[0] = v.L(pc + 1);
      t.c[1] = t.c[2] = t.c[3] = F::zero();
      sink.add(v.PL(20), t, -(v.PL(21) * v.PN(22)));
    } break;
  }
}

// ---------------------------------------------------------------- constraints
// fold.base(c): the next base-field constraint value, in declaration order.
template <class PP, class View, class Fold>
P3R_HD void alu_constraints(const AirParams& a, const View& v, Fold& fold) {
  using F = typename View::V;
  auto L = [&](int c) { return v.L(c); };
  auto N = [&](int c) { return v.N(c); };
  const int lanes = a.lanes, k_max = a.horner_k;
  const int extra_main = lanes * 16, extra_prep = lanes * 13;
  const int num_int = (k_max - 1) / 2;
  const int ac_base = extra_main + num_int * 4;
  const F one = F::one();
  for (int lane = 0; lane < lanes; ++lane) {
    const int m = lane * 16, p = lane * 13;
    V4<F> A = load4<F>(L, m), B = load4<F>(L, m + 4), C = load4<F>(L, m + 8), O = load4<F>(L, m + 12);
    F mult_a = v.PL(p), sel_add = v.PL(p + 1), sel_bool = v.PL(p + 2), sel_muladd = v.PL(p + 3),
      sel_horner = v.PL(p + 4);
    F sel_mul = -mult_a - sel_bool - sel_muladd - sel_horner - sel_add;
#pragma unroll
    for (int i = 0; i < 4; ++i) fold.base(sel_add * (A.c[i] + B.c[i] - O.c[i]));
    V4<F> ab = mul4<PP, F>(A, B);
#pragma unroll
    for (int i = 0; i < 4; ++i) fold.base(sel_mul * (ab.c[i] - O.c[i]));
    fold.base(sel_bool * A.c[0] * (A.c[0] - one));
#pragma unroll
    for (int i = 1; i < 4; ++i) fold.base(sel_bool * A.c[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) fold.base(sel_muladd * (ab.c[i] + C.c[i] - O.c[i]));
    F next_sel_horner = v.PN(p + 4);
    V4<F> NA = load4<F>(N, m), NB = load4<F>(N, m + 4), NC = load4<F>(N, m + 8), NO = load4<F>(N, m + 12);
    V4<F> out_next_b = mul4<PP, F>(O, NB);
    if (lane == 0) {
      F any_cur = F::zero(), any_next = F::zero(), sel_ge3_next = F::zero();
      for (int kk = 2; kk <= k_max; ++kk) any_cur += v.PL(extra_prep + kk - 2);
      for (int kk = 2; kk <= k_max; ++kk) any_next += v.PN(extra_prep + kk - 2);
      F next_sel_k2 = v.PN(extra_prep);
      for (int kk = 3; kk <= k_max; ++kk) sel_ge3_next += v.PN(extra_prep + kk - 2);
      const int b_sq_base = ac_base + 8 * (k_max - 1);
      V4<F> b_sq = load4<F>(L, b_sq_base), b_sq_next = load4<F>(N, b_sq_base);
      V4<F> bb = mul4<PP, F>(B, B);
#pragma unroll
      for (int i = 0; i < 4; ++i) fold.base(any_cur * (b_sq.c[i] - bb.c[i]));
      V4<F> out_b_sq = mul4<PP, F>(O, b_sq_next), c0b = mul4<PP, F>(NC, NB), a0b = mul4<PP, F>(NA, NB);
      V4<F> a1n = load4<F>(N, ac_base), c1n = load4<F>(N, ac_base + 4), int0n = load4<F>(N, extra_main);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        F poly = out_b_sq.c[i] + c0b.c[i] - a0b.c[i] + c1n.c[i] - a1n.c[i];
        fold.base(next_sel_k2 * (poly - NO.c[i]));
        fold.base(sel_ge3_next * (poly - int0n.c[i]));
      }
      F next_sel_single = next_sel_horner - any_next;
#pragma unroll
      for (int i = 0; i < 4; ++i) fold.base(next_sel_single * (out_next_b.c[i] + NC.c[i] - NA.c[i] - NO.c[i]));
      for (int kk = 3; kk <= k_max; ++kk) {
        F sel_kk = v.PL(extra_prep + kk - 2);
        int s = 2, slot = 0;
        while (s < kk) {
          V4<F> int_curr = load4<F>(L, extra_main + slot * 4);
          const int off_s = ac_base + 8 * (s - 1);
          V4<F> a_s = load4<F>(L, off_s), c_s = load4<F>(L, off_s + 4);
          if (s + 1 < kk) {
            const int off_sp1 = ac_base + 8 * s;
            V4<F> a_sp1 = load4<F>(L, off_sp1), c_sp1 = load4<F>(L, off_sp1 + 4);
            V4<F> int_b_sq = mul4<PP, F>(int_curr, b_sq), c_s_b = mul4<PP, F>(c_s, B), a_s_b = mul4<PP, F>(a_s, B);
            const bool to_out = s + 2 >= kk;
            V4<F> target = to_out ? O : load4<F>(L, extra_main + (slot + 1) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              fold.base(sel_kk * (int_b_sq.c[i] + c_s_b.c[i] - a_s_b.c[i] + c_sp1.c[i] - a_sp1.c[i] - target.c[i]));
            if (!to_out) slot += 1;
            s += 2;
          } else {
            V4<F> int_b = mul4<PP, F>(int_curr, B);
#pragma unroll
            for (int i = 0; i < 4; ++i) fold.base(sel_kk * (int_b.c[i] + c_s.c[i] - a_s.c[i] - O.c[i]));
            s += 1;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) fold.base(next_sel_horner * (out_next_b.c[i] + NC.c[i] - NA.c[i] - NO.c[i]));
    }
  }
}

template <class PP, class View, class Fold>
P3R_HD void poseidon2_constraints(const View& v, typename View::V is_transition, const uint32_t* __restrict__ rc,
                                  Fold& fold) {
  using F = typename View::V;
  using B = Fp<PP>;
  constexpr int R = PP::SBOX_REGISTERS;
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
  // inner permutation AIR
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

// Number of base constraints of an AIR (host side needs it to size the alpha-power table).
template <class PP>
__host__ __device__ inline int air_num_base_constraints(const AirParams& a) {
  switch (a.kind) {
    case AIR_ALU: {
      int n = 0;
      for (int lane = 0; lane < a.lanes; ++lane) {
        n += 4 + 4 + 4 + 4;
        if (lane == 0) {
          n += 4 + 8 + 4;
          for (int kk = 3; kk <= a.horner_k; ++kk) {
            int s = 2;
            while (s < kk) { n += 4; s += (s + 1 < kk) ? 2 : 1; }
          }
        } else {
          n += 4;
        }
      }
      return n;
    }
    case AIR_POSEIDON2: {
      constexpr int R = PP::SBOX_REGISTERS;
      return 1 + 16 + 8 + 8 + 1 + 2 * P2_HALF_FULL * (P2_WIDTH * R + P2_WIDTH) + PP::PARTIAL_ROUNDS * (R + 1);
    }
    default: return 0;
  }
}

}  // namespace p3r
