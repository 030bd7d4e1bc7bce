// Poseidon2 width-32 permutation in FP64, device only: the throughput form of the arity-4 MMCS
// (PaddingFreeSponge<Perm32, 32, 24, 8> leaves, TruncatedPermutation<Perm32, 4, 8, 32> levels; one permutation per
// lane).  Same representation and building blocks as the width-16 form (poseidon2_f64.hip.h: a state element is a
// double holding an integer congruent to the canonical value, only the S-box reduces in the full rounds).
//
// What differs from width 16: the internal diagonal is the caller's DATA (p3r_config.poseidon2_w32_diag), so its
// entries are general field elements and, in general, a partial round multiplies every lane by its entry with a full
// modular product (p2f_mulmod_s_add below: six instructions, the entry in a scalar register pair) instead of the
// one-to-three instruction forms a known diagonal allows: 11.3 k FP64 instructions per permutation (width 16: 3.3 k) for
// three times the rate.  The library's own diagonal gets those forms (next paragraph): 7.9 k.
//
// Constant table `tab` (doubles, p3r_ctx::rcd_w32()): [4][32] | [partial] | [4][32] round constants (canonical),
// then the diagonal as CENTRED integers (|d| <= P / 2).
//
// The built-in diagonal (round 5).  The diagonal is data, so in general every lane pays the full product.  The
// library's OWN default diagonal, however, is known at compile time (poseidon2_w32_default.inc): small integers and
// inverse powers of two, like the width-16 one.  When the configured diagonal IS the built-in one (p3r_create compares
// the 32 entries and launches the BUILTIN kernel instances) the partial rounds run per-lane forms fixed at compile time - one FMA for |d| <= 16, the
// three-instruction p2f_mul_2exp_neg_add for +-2^-k with k <= 12, p2f_mul_2exp_neg and an add for larger k - with the
// small-integer lanes reduced every 3 - 5 rounds (three wave-uniform branches per round, as in the width-16 kernel):
// ~95 instead of ~160 instructions per partial round.  Any other diagonal takes the general path.
// (Tried first, as the round-4 review proposed: forms selected at RUN time by wave-uniform branches on per-lane code words,
// for any structured diagonal.  Byte-exact and 2.3 x SLOWER - a commit of 2^22 x 64 in 16.1 ms against 7.0 ms: each lane's
// form becomes its own basic block, so the scheduler can no longer interleave the 32 independent dependency chains of a
// round and every chain runs at FP64 latency.  profiles/r05/w32_diag_ab.txt.)
//
// Magnitudes: inputs |x| <= P (fresh cells, or carried lanes that p2wf_permute reduces on the way in).  External layer: rows of
// circ(2 M4, M4, ..) sum to 7 * 9 = 63, so a full round's S-box sees |x| < 63 * 1.3 P + P < 2^38 - inside the domain
// of the narrow S-box (p2f_mulmod_k needs |a b| < 2^76).  Partial rounds: d_i * s_i reduced to < 0.7 P, the lane sum
// reduced to <= 0.5 P: no growth.
#pragma once
#include <utility>

#include "poseidon2_f64.hip.h"
#include "poseidon2_w32_default.inc"   // the built-in diagonal: its lane forms are compile-time facts here

namespace p3r {

#pragma clang fp contract(off)

__device__ __forceinline__ void p2wf_external_linear(double* s) {
#pragma unroll
  for (int i = 0; i < P2W_WIDTH; i += 4) p2f_mat4(s[i], s[i + 1], s[i + 2], s[i + 3]);
  double sum[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    sum[k] = ((s[k] + s[4 + k]) + (s[8 + k] + s[12 + k])) + ((s[16 + k] + s[20 + k]) + (s[24 + k] + s[28 + k]));
#pragma unroll
  for (int i = 0; i < P2W_WIDTH; ++i) s[i] += sum[i & 3];
}

// Exact product a * d + add mod P with a quotient kept together with its rounding constant (qm = MAGIC + q):
//   qm = fma(a / P, d, MAGIC)          = MAGIC + q exactly, q = rint(a d / P)       (|q| < 2^46)
//   t  = fma(qm, P_HI, -MAGIC * P_HI)  = q * P_HI exactly: one rounding of a value that is representable (46 + 7 bits);
//                                        MAGIC * P_HI is itself a 9-bit constant
//   e  = fma(a, d, -t)                 = (a d - q P) + q exactly
//   (e + addM) - qm                    = a d - q P + add, addM = add + MAGIC: both steps are integer sums below 2^53
// `neg_c` = -MAGIC * P_HI and `magic` are handed in as live vector registers (as literals the compiler re-materialises
// them in front of every use), `p_hi` = P_HI and the diagonal entry in scalar register pairs (one scalar operand per
// instruction on gfx9).
// No per-entry d / P: the quotient comes from (a / P) * d - one more multiplication (six instructions) and 64 fewer
// vector registers than round 4's five-instruction form with c = d / P per entry.  (a * INVP) * d carries two roundings instead of one: an error
// below 2^-20 on a quotient below 2^32, so q is still within one of the exact quotient's rounding and |result| < 0.7 P.
template <class PP>
__device__ __forceinline__ double p2f_mulmod_s_add(double a, double d, double addM, double magic, double neg_c, double p_hi) {
  const double ai = a * P2F64<PP>::INVP;
  const double qm = __builtin_fma(ai, d, magic);
  const double t = __builtin_fma(qm, p_hi, neg_c);
  const double e = __builtin_fma(a, d, -t);
  return (e + addM) - qm;
}

// s_i <- d_i s_i + sum(s)
template <class PP>
__device__ __forceinline__ void p2wf_internal_linear(double* s, const double* d, double magic, double neg_c, double p_hi) {
  double part[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    part[k] = ((s[k] + s[4 + k]) + (s[8 + k] + s[12 + k])) + ((s[16 + k] + s[20 + k]) + (s[24 + k] + s[28 + k]));
  const double sum = p2f_reduce<PP>((part[0] + part[1]) + (part[2] + part[3]));
  const double sumM = sum + P2F64<PP>::MAGIC;
#pragma unroll
  for (int i = 0; i < P2W_WIDTH; ++i) s[i] = p2f_mulmod_s_add<PP>(s[i], d[i], sumM, magic, neg_c, p_hi);
}

// ---- the built-in diagonal: lane forms as compile-time facts
struct P2WLaneForm { int form; int period; };   // form 0: small integer, 1: +-2^-k (addend form), 2: +-2^-k then add, 3: general
template <class PP>
constexpr P2WLaneForm p2w_default_form(int i) {
  const uint32_t d = PP::FIELD_ID == 0 ? kDefaultDiagW32_koala_bear[i] : kDefaultDiagW32_baby_bear[i];
  const int64_t c = d > PP::P / 2 ? (int64_t)d - (int64_t)PP::P : (int64_t)d;
  if (c != 0 && c >= -16 && c <= 16) {
    const int64_t a = c < 0 ? -c : c;
    return {0, a < 2 ? 0 : a <= 4 ? 5 : a <= 7 ? 4 : 3};
  }
  for (int k = 1; k <= PP::TWO_ADICITY; ++k) {
    const uint64_t t = ((uint64_t)d << k) % PP::P;
    if (t == 1 || t == PP::P - 1) return {k <= 12 ? 1 : 2, 0};
  }
  return {3, 0};
}
// the lane's factor as an FP64 value (the integer itself, or +-2^-k), and the first lane with the same magnitude: the
// magnitudes that are not inline constants of the ISA live in SCALAR registers (one pair per distinct magnitude, ~17; the
// sign is a source modifier), see P2FDiag in poseidon2_f64.hip.h for why not literals
template <class PP>
constexpr double p2w_default_factor(int i) {
  const uint32_t d = PP::FIELD_ID == 0 ? kDefaultDiagW32_koala_bear[i] : kDefaultDiagW32_baby_bear[i];
  const int64_t c = d > PP::P / 2 ? (int64_t)d - (int64_t)PP::P : (int64_t)d;
  if (c >= -16 && c <= 16) return (double)c;
  double m = 1.0;
  for (int k = 1; k <= PP::TWO_ADICITY; ++k) {
    m *= 0.5;
    const uint64_t t = ((uint64_t)d << k) % PP::P;
    if (t == 1) return m;
    if (t == PP::P - 1) return -m;
  }
  return 0.0;
}
constexpr double p2w_abs(double x) { return x < 0 ? -x : x; }
template <class PP>
constexpr int p2w_default_rep(int i) {
  for (int j = 0; j < i; ++j)
    if (p2w_abs(p2w_default_factor<PP>(j)) == p2w_abs(p2w_default_factor<PP>(i))) return j;
  return i;
}
template <class PP>
constexpr bool p2w_default_inline(int i) {
  const double a = p2w_abs(p2w_default_factor<PP>(i));
  return a == 0.5 || a == 1.0 || a == 2.0 || a == 4.0;
}
template <class PP, int I>
__device__ __forceinline__ void p2wf_pin_default(double* mk) {
  if constexpr (p2w_default_rep<PP>(I) == I && !p2w_default_inline<PP>(I)) {
    constexpr double mag = p2w_abs(p2w_default_factor<PP>(I));
    mk[I] = mag;
    asm volatile("" : "+s"(mk[I]));
  }
}
template <class PP, int... I>
__device__ __forceinline__ void p2wf_pin_defaults(double* mk, std::integer_sequence<int, I...>) { (p2wf_pin_default<PP, I>(mk), ...); }

template <class PP, int I>
__device__ __forceinline__ void p2wf_lane_default(double* s, const double* mk, double sum) {
  constexpr P2WLaneForm f = p2w_default_form<PP>(I);
  static_assert(f.form != 3, "the built-in width-32 diagonal is made of small integers and inverse powers of two");
  // (constexpr variables, not calls: a call outside a constant expression is compiled, and the 64-bit `%` loops of these
  // functions then run on the scalar unit every round)
  constexpr double lit = p2w_default_factor<PP>(I);
  constexpr bool inl = p2w_default_inline<PP>(I);
  constexpr int rep = p2w_default_rep<PP>(I);
  const double m = inl ? lit : lit < 0 ? -mk[rep] : mk[rep];
  if constexpr (f.form == 0) s[I] = __builtin_fma(s[I], m, sum);
  else if constexpr (f.form == 1) s[I] = p2f_mul_2exp_neg_add<PP>(s[I], m, sum);
  else s[I] = p2f_mul_2exp_neg<PP>(s[I], m) + sum;
}
// `period` names the growth class of a small-integer lane: 5 = |d| in 2..4, 4 = |d| in 5..7, 3 = |d| in 8..16
template <class PP, int PERIOD, int I>
__device__ __forceinline__ void p2wf_lane_reduce(double* s) {
  if constexpr (I > 0 && p2w_default_form<PP>(I).period == PERIOD) s[I] = p2f_reduce<PP>(s[I]);   // lane 0: the S-box reduces it
}
template <class PP, int I>
__device__ __forceinline__ void p2wf_lane_reduce_grown(double* s) {
  if constexpr (I > 0 && p2w_default_form<PP>(I).period > 0) s[I] = p2f_reduce<PP>(s[I]);
}
template <class PP, int... I>
__device__ __forceinline__ void p2wf_reduce_grown(double* s, std::integer_sequence<int, I...>) { (p2wf_lane_reduce_grown<PP, I>(s), ...); }
template <class PP, int... I>
__device__ __forceinline__ void p2wf_internal_linear_default(double* s, const double* mk, int r, std::integer_sequence<int, I...>) {
  // A lane multiplied by a small integer grows by that factor every round, from < 2^37 (63 * 0.7 P) at the first round and
  // from 2^30 after a reduction; kept below 2^47, so that the sum of all of them stays below 2^51: |d| <= 4 reduced at the
  // start of rounds 5, 13, 21, 29 (4^5 2^36.4, then 4^8 2^30), |d| <= 7 at 3, 8, 13, .. (7^3, then 7^5), |d| <= 16 at 2, 5, 8, ..
  // (16^2, then 16^3).  (Until this change: every 5 / 4 / 3 rounds from round 0.)
  if (r >= 5 && (r - 5) % 8 == 0) (p2wf_lane_reduce<PP, 5, I>(s), ...);
  if (r >= 3 && (r - 3) % 5 == 0) (p2wf_lane_reduce<PP, 4, I>(s), ...);
  if (r >= 2 && (r - 2) % 3 == 0) (p2wf_lane_reduce<PP, 3, I>(s), ...);
  double part[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    part[k] = ((s[k] + s[4 + k]) + (s[8 + k] + s[12 + k])) + ((s[16 + k] + s[20 + k]) + (s[24 + k] + s[28 + k]));
  const double sum = p2f_reduce<PP>((part[0] + part[1]) + (part[2] + part[3]));
  (p2wf_lane_default<PP, I>(s, mk, sum), ...);
}

// BUILTIN: the configured diagonal is the built-in one (its own kernel instance: the general path keeps 64 constants in
// vector registers and runs at one wave per SIMD; this one needs none)
// In: integers in [0, P] (p2f_load), except the lanes of CARRIED (bit i = lane i): unreduced outputs of a previous
// permutation, reduced here (poseidon2_f64.hip.h: p2f_permute).  Out: |.| < 63 * 1.3 P, NOT reduced: a digest goes through
// p2f_store, which reduces; a carried lane is reduced by the next permutation.  (Until round 5 all 32 outputs were reduced
// on the way out: 96 instructions, 72 of them on lanes that were overwritten or dropped.)
template <class PP, bool BUILTIN, unsigned CARRIED = 0xFFFFFFFFu>
__device__ __forceinline__ void p2wf_permute(double* s, const double* __restrict__ tab) {
#pragma unroll
  for (int i = 0; i < P2W_WIDTH; ++i)
    if (CARRIED >> i & 1u) s[i] = p2f_reduce<PP>(s[i]);
  const double* d = tab + p2w_num_rc<PP>();
  const P2FSboxK<PP> SK = p2f_sbox_consts<PP>();
  p2wf_external_linear(s);
  int k = 0;
  for (int r = 0; r < P2_HALF_FULL; ++r) {
#pragma unroll
    for (int i = 0; i < P2W_WIDTH; ++i) s[i] = p2f_sbox<PP>(s[i] + tab[k + i], SK);
    k += P2W_WIDTH;
    p2wf_external_linear(s);
  }
  if constexpr (BUILTIN) {
    double mk[P2W_WIDTH];
    p2wf_pin_defaults<PP>(mk, std::make_integer_sequence<int, P2W_WIDTH>{});
#pragma unroll 1
    for (int r = 0; r < PP::PARTIAL_ROUNDS_W32; ++r) {
      s[0] = p2f_sbox<PP>(s[0] + tab[k + r], SK);
      p2wf_internal_linear_default<PP>(s, mk, r, std::make_integer_sequence<int, P2W_WIDTH>{});
    }
    // whatever the small-integer lanes accumulated since their last reduction: back inside the full rounds' domain (the
    // other lanes are there already: +-2^-k lanes leave every round below 2^32, the d = 1 lane adds a reduced sum a round)
    p2wf_reduce_grown<PP>(s, std::make_integer_sequence<int, P2W_WIDTH>{});
  } else {
    // The 32 diagonal entries stay in SCALAR registers through the partial rounds (64 SGPRs) and the vector registers
    // hold the state and its temporaries only: four waves per SIMD.  (Round 4 kept d AND d / P, 64 constants, in vector
    // registers - as scalars they would have needed 128 SGPRs - and ran at one wave per SIMD on the argument that a
    // wave has 32 independent lanes in flight; the built-in diagonal's instance showed what that occupancy costs.)
    double dv[P2W_WIDTH];
#pragma unroll
    for (int i = 0; i < P2W_WIDTH; ++i) {
      dv[i] = d[i];
      asm volatile("" : "+s"(dv[i]));
    }
    double neg_c = -(P2F64<PP>::MAGIC * P2F64<PP>::P_HI);
    double magic = P2F64<PP>::MAGIC;
    double p_hi = P2F64<PP>::P_HI;
    asm volatile("" : "+v"(neg_c), "+v"(magic), "+s"(p_hi));
#pragma unroll 1
    for (int r = 0; r < PP::PARTIAL_ROUNDS_W32; ++r) {
      s[0] = p2f_sbox<PP>(s[0] + tab[k + r], SK);
      p2wf_internal_linear<PP>(s, dv, magic, neg_c, p_hi);
    }
  }
  k += PP::PARTIAL_ROUNDS_W32;
  for (int r = 0; r < P2_HALF_FULL; ++r) {
#pragma unroll
    for (int i = 0; i < P2W_WIDTH; ++i) s[i] = p2f_sbox<PP>(s[i] + tab[k + i], SK);
    k += P2W_WIDTH;
    p2wf_external_linear(s);
  }
}

#pragma clang fp contract(fast)

}  // namespace p3r
