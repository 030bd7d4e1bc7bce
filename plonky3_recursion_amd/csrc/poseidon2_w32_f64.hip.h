// Poseidon2 width-32 permutation in FP64, device only: the throughput form of the arity-4 MMCS
// (PaddingFreeSponge<Perm32, 32, 24, 8> leaves, TruncatedPermutation<Perm32, 4, 8, 32> levels; one permutation per
// lane).  Same representation and building blocks as the width-16 form (poseidon2_f64.hip.h: a state element is a
// double holding an integer congruent to the canonical value, only the S-box reduces in the full rounds).
//
// What differs from width 16: the internal diagonal is the caller's DATA (p3r_config.poseidon2_w32_diag), so its
// entries are general field elements and a partial round multiplies every lane by its entry with the five-instruction
// product p2f_mulmod_c (the quotient factor d_i / P is part of the constant table), instead of the one-to-three
// instruction forms the known width-16 diagonal allows.  That makes a permutation ~11.5 k FP64 instructions
// (width 16: 3.9 k) for three times the rate - the same cost per absorbed cell.
//
// Constant table `tab` (doubles, p3r_ctx::rcd_w32()): [4][32] | [partial] | [4][32] round constants (canonical),
// then the diagonal as CENTRED integers (|d| <= P / 2), then d_i / P.
//
// Magnitudes: inputs |x| <= 0.5 P + slack (p2wf_permute reduces its outputs).  External layer: rows of
// circ(2 M4, M4, ..) sum to 7 * 9 = 63, so a full round's S-box sees |x| < 63 * 1.3 P + P < 2^38 - inside the domain
// of the narrow S-box (p2f_mulmod_c needs |a b| < 2^76).  Partial rounds: d_i * s_i reduced to < 0.7 P, the lane sum
// reduced to <= 0.5 P: no growth.
#pragma once
#include "poseidon2_f64.hip.h"

namespace p3r {

#pragma clang fp contract(off)

template <class PP>
constexpr int p2wf_table_len() { return p2w_num_rc<PP>() + 2 * P2W_WIDTH; }

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

// a * b + add mod P given c = b / P, in FIVE instructions including the addition (p2f_mulmod_c + one add is six): the
// rounding constant stays inside the quotient.
//   qm = fma(a, c, MAGIC)              = MAGIC + q exactly, q = rint(a c)           (|q| < 2^46)
//   t  = fma(qm, P_HI, -MAGIC * P_HI)  = q * P_HI exactly: one rounding of a value that is representable (46 + 7 bits);
//                                        MAGIC * P_HI is itself a 9-bit constant
//   e  = fma(a, b, -t)                 = (a b - q P) + q exactly
//   (e + addM) - qm                    = a b - q P + add, addM = add + MAGIC: both steps are integer sums below 2^53
// `neg_c` = -MAGIC * P_HI, handed in as a live register value: as a literal the compiler re-materialises it in front of
// every use (two v_mov_b32 feeding a v_fmac_f64), which costs more than the instruction the form saves; `p_hi` = P_HI in
// a scalar register pair for the same reason (as a literal it is only encodable in the two-address v_fmac form).
template <class PP>
__device__ __forceinline__ double p2f_mulmod_c_add(double a, double b, double c, double addM, double neg_c, double p_hi) {
  const double qm = __builtin_fma(a, c, P2F64<PP>::MAGIC);
  const double t = __builtin_fma(qm, p_hi, neg_c);
  const double e = __builtin_fma(a, b, -t);
  return (e + addM) - qm;
}

// s_i <- d_i s_i + sum(s)
template <class PP>
__device__ __forceinline__ void p2wf_internal_linear(double* s, const double* __restrict__ d, const double* __restrict__ c, double neg_c, double p_hi) {
  double part[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    part[k] = ((s[k] + s[4 + k]) + (s[8 + k] + s[12 + k])) + ((s[16 + k] + s[20 + k]) + (s[24 + k] + s[28 + k]));
  const double sum = p2f_reduce<PP>((part[0] + part[1]) + (part[2] + part[3]));
  const double sumM = sum + P2F64<PP>::MAGIC;
#pragma unroll
  for (int i = 0; i < P2W_WIDTH; ++i) s[i] = p2f_mulmod_c_add<PP>(s[i], d[i], c[i], sumM, neg_c, p_hi);
}

// In: integers |x| <= 0.5 P + slack.  Out: the same (reduced, not canonical: either sign).
template <class PP>
__device__ __forceinline__ void p2wf_permute(double* s, const double* __restrict__ tab) {
  const double* d = tab + p2w_num_rc<PP>();
  p2wf_external_linear(s);
  int k = 0;
  for (int r = 0; r < P2_HALF_FULL; ++r) {
#pragma unroll
    for (int i = 0; i < P2W_WIDTH; ++i) s[i] = p2f_sbox<PP>(s[i] + tab[k + i]);
    k += P2W_WIDTH;
    p2wf_external_linear(s);
  }
  {
    // The 64 diagonal constants stay in VECTOR registers through the partial rounds.  As scalars they need 128 SGPRs at
    // once: hoisted out of the loop they spill into vector lanes (v_readlane per use: +50 % instructions, measured 0.39
    // ns per permutation), loaded inside the round the waves wait on the scalar cache every round (0.60 ns).  One wave
    // has 32 independent lanes of work in flight, so the lower occupancy costs nothing.
    double dv[P2W_WIDTH], cv[P2W_WIDTH];
#pragma unroll
    for (int i = 0; i < P2W_WIDTH; ++i) {
      dv[i] = d[i];
      cv[i] = d[P2W_WIDTH + i];
      asm volatile("" : "+v"(dv[i]), "+v"(cv[i]));
    }
    double neg_c = -(P2F64<PP>::MAGIC * P2F64<PP>::P_HI);
    double p_hi = P2F64<PP>::P_HI;
    asm volatile("" : "+v"(neg_c), "+s"(p_hi));
    for (int r = 0; r < PP::PARTIAL_ROUNDS_W32; ++r) {
      s[0] = p2f_sbox<PP>(s[0] + tab[k + r]);
      p2wf_internal_linear<PP>(s, dv, cv, neg_c, p_hi);
    }
  }
  k += PP::PARTIAL_ROUNDS_W32;
  for (int r = 0; r < P2_HALF_FULL; ++r) {
#pragma unroll
    for (int i = 0; i < P2W_WIDTH; ++i) s[i] = p2f_sbox<PP>(s[i] + tab[k + i]);
    k += P2W_WIDTH;
    p2wf_external_linear(s);
  }
#pragma unroll
  for (int i = 0; i < P2W_WIDTH; ++i) s[i] = p2f_reduce<PP>(s[i]);
}

#pragma clang fp contract(fast)

}  // namespace p3r
