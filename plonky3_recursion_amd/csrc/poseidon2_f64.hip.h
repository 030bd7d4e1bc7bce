// Poseidon2 width-16 permutation in FP64, device only: the throughput form used by MMCS leaf
// hashing and the wide 2-to-1 layers (one permutation per lane).
//
// Why FP64 for an integer permutation.  gfx950 issues v_add_f64 / v_mul_f64 / v_fma_f64 at the
// full DP rate (tools/microbench: profiles/r02/op_rates.txt), and an integer-valued double holds
// 53 bits, so the linear layers need NO modular reduction: a field addition is ONE instruction
// instead of three (add, sub, min on a 31-bit modulus in a 32-bit word has one bit of headroom),
// `d*s + sum` with a small integer d is one FMA, and the multiplications by 2^-k of the internal
// diagonal are three instructions (below).  Only the S-box reduces, and its reduction also
// absorbs whatever the linear layers accumulated.  Same round structure and constants as
// poseidon2.h; values are exact integers throughout, so the result is the same field element.
//
// Representation: a state element is a double holding an INTEGER congruent to the CANONICAL value
// (not the Montgomery form: the plain product of two Montgomery forms is not one), of either sign,
// magnitude < 2^53.  p2f_load / p2f_store convert from / to the Montgomery u32 of field.h.
//
// Exactness (every step below is exact integer arithmetic, |.| < 2^53):
//   a * b mod P:    p2f_mulmod_k below: the quotient from a * (b / P), the remainder through P = P_HI + 1 with P_HI a
//                   7-bit (4-bit) multiple of 2^24 (2^27), so q * P_HI is exact; needs |a b| < 2^76.
//   x mod P:        q = rint(x / P) (as x / P + 1.5 2^52 - 1.5 2^52); r = fma(-q, P, x) is exact because x - q P is an
//                   integer of magnitude <= P / 2 + slack.  Holds for |x| < 2^51.
//   x / 2^k:        for 2^k | P - 1 and ANY integer x:  x / 2^k  =  t - frac(t) * P  with t = x * 2^-k
//                   (x = 2^k F + low  =>  x / 2^k = F - low (P-1) / 2^k  mod P,  frac(t) = low / 2^k):
//                   v_mul_f64, v_fract_f64, v_fma_f64.  |result| <= |x| / 2^k + P.
//   growth:         the sum of a partial round is reduced (3 instructions), so a round adds at most
//                   0.7 P to a lane after its diagonal factor (|d| <= 4); the lanes with |d| >= 2 are
//                   reduced twice inside the partial rounds and once after them (p2f_permute).
#pragma once
#include "poseidon2.h"

#if defined(__FAST_MATH__) || defined(__FINITE_MATH_ONLY__) && __FINITE_MATH_ONLY__
#error "poseidon2_f64.hip.h relies on exact IEEE-754 double arithmetic (error-free products, magic-number rounding): do not build with -ffast-math / -Ofast"
#endif

namespace p3r {

template <class PP>
struct P2F64 {
  static constexpr double P = (double)PP::P;
  static constexpr double INVP = 1.0 / (double)PP::P;
  // x + MAGIC - MAGIC = x rounded to the nearest integer for |x| < 2^51: two instructions, the first one fused with the
  // product that forms x (v_rndne_f64 is full-rate on gfx950 too - tools/microbench/int_rates - and would be the same
  // count: v_mul_f64 + v_rndne_f64); p2f_mulmod_k never subtracts it back.
  static constexpr double MAGIC = 0x1.8p52;
  // P = P_HI + 1 with P_HI = c * 2^m (127 * 2^24, 15 * 2^27): q * P_HI is an exact double for q < 2^46
  static constexpr double P_HI = (double)(PP::P - 1);
};
#pragma clang fp contract(off)

template <class PP>
__device__ __forceinline__ double p2f_quot(double x) {
  return __builtin_fma(x, P2F64<PP>::INVP, P2F64<PP>::MAGIC) - P2F64<PP>::MAGIC;
}

// x mod P, |result| <= 0.5 P (+ rounding slack)
template <class PP>
__device__ __forceinline__ double p2f_reduce(double x) {
  const double q = p2f_quot<PP>(x);
  return __builtin_fma(-q, P2F64<PP>::P, x);
}
// x * m mod P where m = +-2^-k, 2^k | P - 1
template <class PP>
__device__ __forceinline__ double p2f_mul_2exp_neg(double x, double m) {
  const double t = x * m;
  const double f = __builtin_amdgcn_fract(t);
  return __builtin_fma(-f, P2F64<PP>::P, t);
}
// x * m + a mod P for k <= 8 and an INTEGER a with |a| < 2^40: the addend rides in the first FMA
// (a + x * 2^-k has at most 40 + 8 significant bits, so it is exact and frac(t) is unchanged)
template <class PP>
__device__ __forceinline__ double p2f_mul_2exp_neg_add(double x, double m, double a) {
  const double t = __builtin_fma(x, m, a);
  const double f = __builtin_amdgcn_fract(t);
  return __builtin_fma(-f, P2F64<PP>::P, t);
}

// a * b mod P given c = b / P (rounded; c is shared by every product with the same b: the S-box multiplies by x twice or
// three times).  Until round 5 in five instructions:
//   q = fma(a, c, MAGIC) - MAGIC = rint(a c);  t = q * P_HI;  e = fma(a, b, -t) = (a b - q P) + q exactly (an integer below
//   2^47, because q * P_HI is exact);  result = e - q.  Needs |a b| < 2^76 (q < 2^46); |result| < 0.7 P.
// Now in FOUR: the rounding constant is never subtracted from the quotient, it rides through the chain and cancels in the
// last step.
//   qm = fma(a, c, MAGIC)            = MAGIC + q exactly, q = rint(a c)                        (|q| < 2^46)
//   t  = fma(qm, P_HI, -MAGIC * P)   = q P_HI - MAGIC exactly: qm P_HI = MAGIC P_HI + q P_HI, and MAGIC P = MAGIC P_HI + MAGIC;
//                                      the result is a multiple of 2^24 below 2^77, 53 significant bits
//   e  = fma(a, b, -t)               = (a b - q P) + q + MAGIC exactly: an integer below 2^47 on top of MAGIC, inside [2^52, 2^53)
//   e - qm                           = a b - q P
// MAGIC P = 3 P 2^51 is a 33-bit constant.  The three constants must sit in registers (one scalar operand per instruction
// on gfx9, no 64-bit literals in the three-address forms): `k` = -MAGIC P in a vector register pair, P_HI and MAGIC in scalar
// ones, pinned by p2f_sbox_consts so that the compiler does not fold them back into literals (see P2FDiag below for what
// that costs).  Two instructions fewer per S-box of degree 3, four per S-box of degree 7.
template <class PP>
struct P2FSboxK {
  double k, p_hi, magic;
};
template <class PP>
__device__ __forceinline__ P2FSboxK<PP> p2f_sbox_consts() {
  P2FSboxK<PP> K;
  K.k = -(P2F64<PP>::MAGIC * P2F64<PP>::P);
  K.p_hi = P2F64<PP>::P_HI;
  K.magic = P2F64<PP>::MAGIC;
  asm volatile("" : "+v"(K.k), "+s"(K.p_hi), "+s"(K.magic));
  return K;
}
template <class PP>
__device__ __forceinline__ double p2f_mulmod_k(double a, double b, double c, const P2FSboxK<PP>& K) {
  const double qm = __builtin_fma(a, c, K.magic);
  const double t = __builtin_fma(qm, K.p_hi, K.k);
  const double e = __builtin_fma(a, b, -t);
  return e - qm;
}

// |x| < 2^38 (p2f_mulmod_k needs |a b| < 2^76).  (Until round 5 there was a second, "wide" form on the two-product
// p2f_mulmod for lanes that arrived unreduced; every caller reduces such lanes first now, which is cheaper.)
template <class PP>
__device__ __forceinline__ double p2f_sbox(double x, const P2FSboxK<PP>& K) {
  const double c = x * P2F64<PP>::INVP;
  const double x2 = p2f_mulmod_k<PP>(x, x, c, K);
  const double x3 = p2f_mulmod_k<PP>(x2, x, c, K);
  if (PP::SBOX_DEGREE == 3) return x3;
  const double x6 = p2f_mulmod_k<PP>(x3, x3, x3 * P2F64<PP>::INVP, K);
  return p2f_mulmod_k<PP>(x6, x, c, K);
}

__device__ __forceinline__ void p2f_mat4(double& x0, double& x1, double& x2, double& x3) {
  const double t01 = x0 + x1, t23 = x2 + x3;
  const double t0123 = t01 + t23;
  const double t01123 = t0123 + x1;
  const double t01233 = t0123 + x3;
  const double n3 = __builtin_fma(x0, 2.0, t01233);
  const double n1 = __builtin_fma(x2, 2.0, t01123);
  const double n0 = t01123 + t01;
  const double n2 = t01233 + t23;
  x0 = n0; x1 = n1; x2 = n2; x3 = n3;
}
// |out| <= 35 max|in|
__device__ __forceinline__ void p2f_external_linear(double* s) {
#pragma unroll
  for (int i = 0; i < P2_WIDTH; i += 4) p2f_mat4(s[i], s[i + 1], s[i + 2], s[i + 3]);
  double sum[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sum[k] = (s[k] + s[4 + k]) + (s[8 + k] + s[12 + k]);
#pragma unroll
  for (int i = 0; i < P2_WIDTH; ++i) s[i] += sum[i & 3];
}

// The diagonal's factors that are not inline constants of the ISA (3 and the inverse powers of two below 1/2), as
// SCALAR REGISTER values.  Written as literals they are only encodable in the two-address v_fmac_f64 form, whose
// addend register is overwritten: every `x * m + sum` then starts with a copy of `sum` (v_mov_b64), 7 extra
// instructions in an 80-instruction partial round.  From a register the three-address v_fma_f64 takes them.
template <class PP>
struct P2FDiag {
  double three, a, b, c, d;   // KoalaBear: 2^-8, 2^-3, 2^-4, 2^-24;  BabyBear: 2^-8, 2^-2, 2^-3, 2^-4 (2^-27 has no addend form)
};
template <class PP>
__device__ __forceinline__ P2FDiag<PP> p2f_diag_consts() {
  P2FDiag<PP> k;
  k.three = 3.0;
  k.a = 0x1p-8;
  k.b = PP::FIELD_ID == 0 ? 0x1p-3 : 0x1p-2;
  k.c = PP::FIELD_ID == 0 ? 0x1p-4 : 0x1p-3;
  k.d = PP::FIELD_ID == 0 ? 0x1p-24 : 0x1p-4;
  asm volatile("" : "+s"(k.three), "+s"(k.a), "+s"(k.b), "+s"(k.c), "+s"(k.d));
  return k;
}

// Diagonal of the internal layer as FP64 factors (poseidon2.h: p2_internal_linear); entries that
// are integers are applied by one FMA, the 2^-k ones by p2f_mul_2exp_neg.
template <class PP>
__device__ __forceinline__ void p2f_internal_linear(double* s, bool reduce_wide, const P2FDiag<PP>& K) {
  if (reduce_wide) {
    s[2] = p2f_reduce<PP>(s[2]);
    s[4] = p2f_reduce<PP>(s[4]);
    s[5] = p2f_reduce<PP>(s[5]);
    s[7] = p2f_reduce<PP>(s[7]);
    s[8] = p2f_reduce<PP>(s[8]);
  }
  double part = ((s[1] + s[2]) + (s[3] + s[4])) + ((s[5] + s[6]) + (s[7] + s[8]));
  part += ((s[9] + s[10]) + (s[11] + s[12])) + ((s[13] + s[14]) + s[15]);
  const double sum = p2f_reduce<PP>(part + s[0]);
  s[0] = __builtin_fma(s[0], -2.0, sum);
  s[1] = s[1] + sum;
  s[2] = __builtin_fma(s[2], 2.0, sum);
  s[3] = p2f_mul_2exp_neg_add<PP>(s[3], 0.5, sum);
  s[4] = __builtin_fma(s[4], K.three, sum);
  s[5] = __builtin_fma(s[5], 4.0, sum);
  s[6] = p2f_mul_2exp_neg_add<PP>(s[6], -0.5, sum);
  s[7] = __builtin_fma(s[7], -K.three, sum);
  s[8] = __builtin_fma(s[8], -4.0, sum);
  s[9] = p2f_mul_2exp_neg_add<PP>(s[9], K.a, sum);          // 2^-8
  if (PP::FIELD_ID == 0) {
    s[10] = p2f_mul_2exp_neg_add<PP>(s[10], K.b, sum);      // 2^-3
    s[11] = p2f_mul_2exp_neg<PP>(s[11], K.d) + sum;         // 2^-24
    s[12] = p2f_mul_2exp_neg_add<PP>(s[12], -K.a, sum);     // -2^-8
    s[13] = p2f_mul_2exp_neg_add<PP>(s[13], -K.b, sum);     // -2^-3
    s[14] = p2f_mul_2exp_neg_add<PP>(s[14], -K.c, sum);     // -2^-4
    s[15] = p2f_mul_2exp_neg<PP>(s[15], -K.d) + sum;        // -2^-24
  } else {
    s[10] = p2f_mul_2exp_neg_add<PP>(s[10], K.b, sum);      // 2^-2
    s[11] = p2f_mul_2exp_neg_add<PP>(s[11], K.c, sum);      // 2^-3
    s[12] = p2f_mul_2exp_neg<PP>(s[12], 0x1p-27) + sum;
    s[13] = p2f_mul_2exp_neg_add<PP>(s[13], -K.a, sum);     // -2^-8
    s[14] = p2f_mul_2exp_neg_add<PP>(s[14], -K.d, sum);     // -2^-4
    s[15] = p2f_mul_2exp_neg<PP>(s[15], -0x1p-27) + sum;
  }
}

// `rc`: the flat constant table of poseidon2.h as CANONICAL doubles.
// In: integers in [0, P] (p2f_load: a lazy REDC of one word), except the lanes of CARRIED (bit i = lane i), which may hold the unreduced outputs of
// a previous permutation (< 2^36) and are reduced first: three instructions per carried lane, after which EVERY S-box of
// the first full round is the narrow one (four instructions fewer per lane than the wide form that an unreduced lane,
// spread over the state by the first linear layer, used to force on all sixteen).
// Out: integers of magnitude < 2^36 (35 * 0.7 P), not reduced.
template <class PP, unsigned CARRIED = 0xFFFFu>
__device__ __forceinline__ void p2f_permute(double* s, const double* __restrict__ rc) {
  const P2FSboxK<PP> SK = p2f_sbox_consts<PP>();
#pragma unroll
  for (int i = 0; i < P2_WIDTH; ++i)
    if (CARRIED >> i & 1u) s[i] = p2f_reduce<PP>(s[i]);
  p2f_external_linear(s);   // |.| <= 35 P: inside the narrow S-box's 2^38 with the round constant added
  int k = 0;
  // (two rounds per iteration: the scalar loads of the second round's constants are in flight while the first one runs)
#pragma unroll 2
  for (int r = 0; r < P2_HALF_FULL; ++r) {
#pragma unroll
    for (int i = 0; i < P2_WIDTH; ++i) s[i] = p2f_sbox<PP>(s[i] + rc[k + i], SK);
    k += P2_WIDTH;
    p2f_external_linear(s);
  }
  const P2FDiag<PP> K = p2f_diag_consts<PP>();
  // The five lanes with |d| >= 2 (2, 3, 4, -3, -4) grow by up to four times a round from < 2^36: reduced at the start of
  // rounds 6 and 15 (2^36 * 4^6 and 2^30 * 4^9 stay below 2^49, so the lane sum stays below 2^52) and once after the last
  // round, after which every S-box of the last four full rounds is the narrow one.  (Until round 5: every fifth round
  // and a wide S-box for the five lanes, 80 instructions where this takes 45.)
  for (int r = 0; r < PP::PARTIAL_ROUNDS; ++r) {
    s[0] = p2f_sbox<PP>(s[0] + rc[k + r], SK);
    p2f_internal_linear<PP>(s, r == 6 || r == 15, K);
  }
  k += PP::PARTIAL_ROUNDS;
  s[2] = p2f_reduce<PP>(s[2]);
  s[4] = p2f_reduce<PP>(s[4]);
  s[5] = p2f_reduce<PP>(s[5]);
  s[7] = p2f_reduce<PP>(s[7]);
  s[8] = p2f_reduce<PP>(s[8]);
  // (two rounds per iteration: the scalar loads of the second round's constants are in flight while the first one runs)
#pragma unroll 2
  for (int r = 0; r < P2_HALF_FULL; ++r) {
#pragma unroll
    for (int i = 0; i < P2_WIDTH; ++i) s[i] = p2f_sbox<PP>(s[i] + rc[k + i], SK);
    k += P2_WIDTH;
    p2f_external_linear(s);
  }
}

// Montgomery u32 (field.h) -> canonical integer in a double: one REDC, one conversion.
template <class PP>
__device__ __forceinline__ double p2f_load(uint32_t mont) {
  using F = Fp<PP>;
  return (double)F::reduce64_lazy((uint64_t)mont);
}
// Any state element (|x| < 2^40) -> fully reduced Montgomery u32: x * 2^32 mod P.
template <class PP>
__device__ __forceinline__ uint32_t p2f_store(double x) {
  const double r = p2f_reduce<PP>(x * 0x1p32);
  const int32_t v = (int32_t)r;
  return (uint32_t)(v + ((v >> 31) & (int32_t)PP::P));
}

#pragma clang fp contract(fast)

}  // namespace p3r
