// Poseidon2 width-16 permutation for KoalaBear (x^3, 4+20+4 rounds) and BabyBear
// (x^7, 4+13+4 rounds), as one host/device template.
//
// Reference anchors:
//   round structure / S-box degrees  circuit/src/ops/poseidon2_perm/config.rs:56-122
//   round-constant sources            poseidon2-circuit-air/src/public_types.rs:48-54,220-226
//   sponge / compression use          recursion/src/pcs/mmcs.rs:17-26,75-172,
//                                     circuit/src/ops/mmcs.rs:117-160
// The linear layers (external 4x4 circulant-of-M4, internal diagonal) are those of the
// un-vendored p3-poseidon2 / p3-{koala,baby}-bear 0.6 crates.
//
// The round constants are DATA supplied through p3r_config (include/p3r.h): a flat table
//   [4][16] external-initial | [PARTIAL] internal | [4][16] external-final
// in Montgomery form on the device.  Nothing in this file bakes in constant values.
#pragma once
#include "field.h"

namespace p3r {

constexpr int P2_WIDTH = 16;
constexpr int P2_RATE = 8;
constexpr int P2_DIGEST = 8;
constexpr int P2_HALF_FULL = 4;

template <class PP>
constexpr int p2_num_constants() { return 2 * P2_HALF_FULL * P2_WIDTH + PP::PARTIAL_ROUNDS; }

// Number of columns of the upstream Poseidon2Cols<.., WIDTH=16, ..> struct:
// inputs | 4 x {sbox regs[16][R], post[16]} | PARTIAL x {sbox regs[R], post_sbox} | 4 x {...}.
template <class PP>
constexpr int p2_perm_cols() {
  return P2_WIDTH + 2 * P2_HALF_FULL * (P2_WIDTH * PP::SBOX_REGISTERS + P2_WIDTH) +
         PP::PARTIAL_ROUNDS * (PP::SBOX_REGISTERS + 1);
}

// y = M4 * x with M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]].
template <class F>
P3R_HD void p2_mat4(F& x0, F& x1, F& x2, F& x3) {
  F t01 = x0 + x1, t23 = x2 + x3;
  F t0123 = t01 + t23;
  F t01123 = t0123 + x1;
  F t01233 = t0123 + x3;
  F n3 = t01233 + x0.dbl();  // 3x0 + x1 + x2 + 2x3
  F n1 = t01123 + x2.dbl();  // x0 + 2x1 + 3x2 + x3
  F n0 = t01123 + t01;       // 2x0 + 3x1 + x2 + x3
  F n2 = t01233 + t23;       // x0 + x1 + 2x2 + 3x3
  x0 = n0; x1 = n1; x2 = n2; x3 = n3;
}

template <class F>
P3R_HD void p2_external_linear(F* s) {
#pragma unroll
  for (int i = 0; i < P2_WIDTH; i += 4) p2_mat4(s[i], s[i + 1], s[i + 2], s[i + 3]);
  F sum[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sum[k] = s[k] + s[4 + k] + s[8 + k] + s[12 + k];
#pragma unroll
  for (int i = 0; i < P2_WIDTH; ++i) s[i] += sum[i & 3];
}

// Multiply a Montgomery-form value by 2^-k.  Both primes are c*2^m + 1 (KoalaBear 127*2^24 + 1,
// BabyBear 15*2^27 + 1), so 2^-m = -c and, for k <= m,
//     x / 2^k  =  (x >> k)  -  c * ((x mod 2^k) << (m - k))      (mod P),
// with both terms below P: shifts, one small multiply, one conditional correction - no REDC.
// (Montgomery form is linear, so the same map applies to the representative.)
template <class F>
P3R_HD F p2_div_2exp(F x, int k) {
  using PP = typename F::Params;
  constexpr int M = PP::TWO_ADICITY;                 // 24 / 27
  constexpr uint32_t C = (PP::P - 1) >> M;           // 127 / 15
  const uint32_t hi = x.v >> k;
  const uint32_t t = (x.v << (32 - k)) >> (32 - M);  // (x mod 2^k) << (M - k), below 2^M
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t m = M <= 24 ? __umul24(t, C) : (t << 4) - t;  // C = 15 = 2^4 - 1 when M = 27
#else
  const uint32_t m = t * C;
#endif
  const uint32_t d = hi - m, d2 = d + PP::P;
  return F::raw(d < d2 ? d : d2);
}

template <class PP>
P3R_HD Fp4<PP> p2_div_2exp(Fp4<PP> x, int k) {
  Fp4<PP> r;
  for (int i = 0; i < 4; ++i) r.c[i] = p2_div_2exp(x.c[i], k);
  return r;
}
template <class PP>
P3R_HD Fp5<PP> p2_div_2exp(Fp5<PP> x, int k) {
  Fp5<PP> r;
  for (int i = 0; i < 5; ++i) r.c[i] = p2_div_2exp(x.c[i], k);
  return r;
}

// s_i <- v_i * s_i + sum(s), diagonal v per field:
//  KoalaBear: [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/8, 1/2^24, -1/2^8, -1/8, -1/16, -1/2^24]
//  BabyBear : [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27]
template <class PP, class F>
P3R_HD void p2_internal_linear(F* s) {
  F part = s[1];
#pragma unroll
  for (int i = 2; i < P2_WIDTH; ++i) part += s[i];
  F sum = part + s[0];
  s[0] = part - s[0];  // -2*s0 + sum
  s[1] = s[1] + sum;
  s[2] = s[2].dbl() + sum;
  s[3] = s[3].halve() + sum;
  s[4] = s[4].dbl() + s[4] + sum;
  s[5] = s[5].dbl().dbl() + sum;
  s[6] = sum - s[6].halve();
  s[7] = sum - (s[7].dbl() + s[7]);
  s[8] = sum - s[8].dbl().dbl();
  s[9] = p2_div_2exp(s[9], 8) + sum;
  if (PP::FIELD_ID == 0) {
    s[10] = p2_div_2exp(s[10], 3) + sum;
    s[11] = p2_div_2exp(s[11], 24) + sum;
    s[12] = sum - p2_div_2exp(s[12], 8);
    s[13] = sum - p2_div_2exp(s[13], 3);
    s[14] = sum - p2_div_2exp(s[14], 4);
    s[15] = sum - p2_div_2exp(s[15], 24);
  } else {
    s[10] = p2_div_2exp(s[10], 2) + sum;
    s[11] = p2_div_2exp(s[11], 3) + sum;
    s[12] = p2_div_2exp(s[12], 27) + sum;
    s[13] = sum - p2_div_2exp(s[13], 8);
    s[14] = sum - p2_div_2exp(s[14], 4);
    s[15] = sum - p2_div_2exp(s[15], 27);
  }
}

template <class PP, class F>
P3R_HD F p2_sbox(F x) {
  F x3 = x.cube();
  if (PP::SBOX_DEGREE == 3) return x3;
  return x3.sqr_times(x);  // x^7
}

// Trace sink used by the circuit-table fill (K3): receives every committed cell in
// Poseidon2Cols order. NullSink turns the same code into the plain permutation.
struct P2NullSink {
  template <class F> P3R_HD void put(F) {}
};

// Full permutation; `rc` is the flat constant table in Montgomery form.
// Emits, in Poseidon2Cols order after the inputs: per full round [sbox regs x16][post x16],
// per partial round [sbox reg][post_sbox].
template <class PP, class F, class Sink>
P3R_HD void p2_permute_traced(F* s, const uint32_t* __restrict__ rc, Sink& sink) {
  p2_external_linear(s);
  int k = 0;
  for (int r = 0; r < P2_HALF_FULL; ++r) {
#pragma unroll
    for (int i = 0; i < P2_WIDTH; ++i) {
      F x = s[i] + F::raw(rc[k + i]);
      if (PP::SBOX_REGISTERS == 1) {
        F x3 = x.cube();
        sink.put(x3);
        s[i] = x3.sqr_times(x);
      } else {
        s[i] = p2_sbox<PP>(x);
      }
    }
    k += P2_WIDTH;
    p2_external_linear(s);
#pragma unroll
    for (int i = 0; i < P2_WIDTH; ++i) sink.put(s[i]);
  }
  for (int r = 0; r < PP::PARTIAL_ROUNDS; ++r) {
    F x = s[0] + F::raw(rc[k + r]);
    if (PP::SBOX_REGISTERS == 1) {
      F x3 = x.cube();
      sink.put(x3);
      s[0] = x3.sqr_times(x);
    } else {
      s[0] = p2_sbox<PP>(x);
    }
    sink.put(s[0]);
    p2_internal_linear<PP>(s);
  }
  k += PP::PARTIAL_ROUNDS;
  for (int r = 0; r < P2_HALF_FULL; ++r) {
#pragma unroll
    for (int i = 0; i < P2_WIDTH; ++i) {
      F x = s[i] + F::raw(rc[k + i]);
      if (PP::SBOX_REGISTERS == 1) {
        F x3 = x.cube();
        sink.put(x3);
        s[i] = x3.sqr_times(x);
      } else {
        s[i] = p2_sbox<PP>(x);
      }
    }
    k += P2_WIDTH;
    p2_external_linear(s);
#pragma unroll
    for (int i = 0; i < P2_WIDTH; ++i) sink.put(s[i]);
  }
}

template <class PP, class F>
P3R_HD void p2_permute(F* s, const uint32_t* __restrict__ rc) {
  P2NullSink sink;
  p2_permute_traced<PP>(s, rc, sink);
}

// ---- width 32: the permutation of the arity-4 MMCS (Poseidon2{Koala,Baby}Bear<32>, Poseidon2Config::*_D4_W32;
// circuit-prover/tests/arity4_mmcs.rs:42-47).  Same round structure; the external layer runs over eight M4 blocks;
// the internal diagonal is DATA like the round constants (upstream's GenericPoseidon2LinearLayers<32> lives in
// un-vendored crates): the constant table is [4][32] external-initial | [PARTIAL_W32] internal | [4][32]
// external-final | [32] diagonal, Montgomery on the device, and follows the width-16 table in p3r_ctx::rc.
constexpr int P2W_WIDTH = 32;
template <class PP>
constexpr int p2w_num_rc() { return 2 * P2_HALF_FULL * P2W_WIDTH + PP::PARTIAL_ROUNDS_W32; }
template <class PP>
constexpr int p2w_num_constants() { return p2w_num_rc<PP>() + P2W_WIDTH; }
template <class PP>
constexpr int p2w_perm_cols() {
  return P2W_WIDTH + 2 * P2_HALF_FULL * (P2W_WIDTH * PP::SBOX_REGISTERS + P2W_WIDTH) + PP::PARTIAL_ROUNDS_W32 * (PP::SBOX_REGISTERS + 1);
}
template <class F>
P3R_HD void p2w_external_linear(F* s) {
#pragma unroll
  for (int i = 0; i < P2W_WIDTH; i += 4) p2_mat4(s[i], s[i + 1], s[i + 2], s[i + 3]);
  F sum[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sum[k] = s[k];
#pragma unroll
    for (int b = 1; b < P2W_WIDTH / 4; ++b) sum[k] += s[4 * b + k];
  }
#pragma unroll
  for (int i = 0; i < P2W_WIDTH; ++i) s[i] += sum[i & 3];
}
// s_i <- d_i * s_i + sum(s), d = the diagonal of the constant table (base-field words lifted into F)
template <class PP, class F>
P3R_HD void p2w_internal_linear(F* s, const uint32_t* __restrict__ diag) {
  F sum = s[0];
#pragma unroll
  for (int i = 1; i < P2W_WIDTH; ++i) sum += s[i];
#pragma unroll
  for (int i = 0; i < P2W_WIDTH; ++i) s[i] = s[i] * Lift<F>::of(Fp<PP>::raw(diag[i])) + sum;
}
// Full permutation; `rcw` = the width-32 constant table.  Cells in Poseidon2Cols order, as p2_permute_traced.
template <class PP, class F, class Sink>
P3R_HD void p2w_permute_traced(F* s, const uint32_t* __restrict__ rcw, Sink& sink) {
  const uint32_t* diag = rcw + p2w_num_rc<PP>();
  p2w_external_linear(s);
  int k = 0;
  auto full_round = [&]() {
#pragma unroll
    for (int i = 0; i < P2W_WIDTH; ++i) {
      F x = s[i] + F::raw(rcw[k + i]);
      if (PP::SBOX_REGISTERS == 1) {
        F x3 = x.cube();
        sink.put(x3);
        s[i] = x3.sqr_times(x);
      } else {
        s[i] = p2_sbox<PP>(x);
      }
    }
    k += P2W_WIDTH;
    p2w_external_linear(s);
#pragma unroll
    for (int i = 0; i < P2W_WIDTH; ++i) sink.put(s[i]);
  };
  for (int r = 0; r < P2_HALF_FULL; ++r) full_round();
  for (int r = 0; r < PP::PARTIAL_ROUNDS_W32; ++r) {
    F x = s[0] + F::raw(rcw[k + r]);
    if (PP::SBOX_REGISTERS == 1) {
      F x3 = x.cube();
      sink.put(x3);
      s[0] = x3.sqr_times(x);
    } else {
      s[0] = p2_sbox<PP>(x);
    }
    sink.put(s[0]);
    p2w_internal_linear<PP>(s, diag);
  }
  k += PP::PARTIAL_ROUNDS_W32;
  for (int r = 0; r < P2_HALF_FULL; ++r) full_round();
}

// ---- the FP64 table of the width-32 permutation (poseidon2_w32_f64.hip.h): the round constants, then the 32 entries of
// the internal diagonal, as doubles (centred: entries above P / 2 as negative numbers)
}  // namespace p3r
