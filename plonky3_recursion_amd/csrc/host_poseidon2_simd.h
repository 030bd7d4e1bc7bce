// Host-side Poseidon2 for the transcript (K11): the DuplexChallenger is a strictly sequential
// sponge on the host (as in the reference), and for the 2^14..2^16-row layers of real verifier
// circuits its ~560 permutations per proof - mostly the absorption of the opened values - are a
// visible part of a 4-5 ms proof.  On x86-64 hosts with AVX-512 the whole 16-element state is ONE
// vector register: S-boxes of a full round are 16 parallel Montgomery products, the external layer
// is three in-lane shuffles (M4 is circulant) plus two 128-bit-lane swaps, and in the partial
// rounds element 0 stays in a scalar register while the other fifteen take one vector product with
// the diagonal.  Same arithmetic as poseidon2.h (Montgomery form), bit-identical results;
// host_permute() falls back to the scalar template when the CPU lacks AVX-512.
#pragma once
#include <cstdlib>

#include "poseidon2.h"

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
#define P3R_HOST_AVX512 1
#endif

namespace p3r {

#if defined(P3R_HOST_AVX512)
#define P3R_AVX512_FN __attribute__((target("avx512f,avx512dq")))

template <class PP>
struct P2Avx512 {
  using F = Fp<PP>;
  // P3R_HOST_SIMD=0 in the environment forces the scalar template (tests compare the two)
  static bool supported() {
    static const bool ok = [] {
      const char* e = getenv("P3R_HOST_SIMD");
      if (e && e[0] == '0') return false;
      return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    }();
    return ok;
  }
  // Internal diagonal in Montgomery form (poseidon2.h, p2_internal_linear); lane 0 is zero because
  // element 0 is kept in a scalar register during the partial rounds.
  static const uint32_t* diag() {
    static const struct Table {
      alignas(64) uint32_t d[16];
      Table() {
        auto inv2k = [](int k) { return F::from_u64(uint64_t(1) << k).inv(); };
        const F two = F::from_canonical(2), three = F::from_canonical(3), four = F::from_canonical(4);
        F v[16] = {F::zero(), F::one(), two, inv2k(1), three, four, -inv2k(1), -three, -four, inv2k(8)};
        if (PP::FIELD_ID == 0) {
          v[10] = inv2k(3); v[11] = inv2k(24); v[12] = -inv2k(8); v[13] = -inv2k(3); v[14] = -inv2k(4); v[15] = -inv2k(24);
        } else {
          v[10] = inv2k(2); v[11] = inv2k(3); v[12] = inv2k(27); v[13] = -inv2k(8); v[14] = -inv2k(4); v[15] = -inv2k(27);
        }
        for (int i = 0; i < 16; ++i) d[i] = v[i].v;
      }
    } t;
    return t.d;
  }

  P3R_AVX512_FN static inline __m512i addm(__m512i a, __m512i b, __m512i p) {
    const __m512i s = _mm512_add_epi32(a, b);
    return _mm512_min_epu32(s, _mm512_sub_epi32(s, p));
  }
  // a * b * 2^-32 mod P per lane, result in [0, 2P): needs a * b < P * 2^32 (one factor below P,
  // the other any u32, or both below 2^31.5)
  P3R_AVX512_FN static inline __m512i mul_lazy(__m512i a, __m512i b, __m512i p, __m512i neg_mu) {
    const __m512i xe = _mm512_mul_epu32(a, b);
    const __m512i xo = _mm512_mul_epu32(_mm512_srli_epi64(a, 32), _mm512_srli_epi64(b, 32));
    const __m512i re = _mm512_add_epi64(_mm512_mul_epu32(_mm512_mul_epu32(xe, neg_mu), p), xe);
    const __m512i ro = _mm512_add_epi64(_mm512_mul_epu32(_mm512_mul_epu32(xo, neg_mu), p), xo);
    return _mm512_mask_blend_epi32(0xAAAA, _mm512_srli_epi64(re, 32), ro);
  }
  P3R_AVX512_FN static inline __m512i mulm(__m512i a, __m512i b, __m512i p, __m512i neg_mu) {
    const __m512i r = mul_lazy(a, b, p, neg_mu);
    return _mm512_min_epu32(r, _mm512_sub_epi32(r, p));
  }
  P3R_AVX512_FN static inline __m512i sbox(__m512i x, __m512i p, __m512i neg_mu) {
    const __m512i x2 = mul_lazy(x, x, p, neg_mu);  // [0, 2P): fine as a factor next to x < P
    if (PP::SBOX_DEGREE == 3) return mulm(x2, x, p, neg_mu);
    const __m512i x2r = _mm512_min_epu32(x2, _mm512_sub_epi32(x2, p));
    const __m512i x3 = mulm(x2r, x, p, neg_mu);
    const __m512i x4 = mul_lazy(x2r, x2r, p, neg_mu);
    return mulm(x4, x3, p, neg_mu);  // x^7
  }
  // y_i = 2 x_i + 3 x_{i+1} + x_{i+2} + x_{i+3} inside each group of four (M4 is circulant), then
  // every element gets the sum of the four groups at its position added.
  P3R_AVX512_FN static inline __m512i external(__m512i x, __m512i p) {
    const __m512i b = _mm512_shuffle_epi32(x, (_MM_PERM_ENUM)_MM_SHUFFLE(0, 3, 2, 1));
    const __m512i c = _mm512_shuffle_epi32(x, (_MM_PERM_ENUM)_MM_SHUFFLE(1, 0, 3, 2));
    const __m512i d = _mm512_shuffle_epi32(x, (_MM_PERM_ENUM)_MM_SHUFFLE(2, 1, 0, 3));
    const __m512i ab = addm(x, b, p);
    const __m512i y = addm(addm(addm(ab, ab, p), b, p), addm(c, d, p), p);
    __m512i t = addm(y, _mm512_shuffle_i32x4(y, y, _MM_SHUFFLE(2, 3, 0, 1)), p);
    t = addm(t, _mm512_shuffle_i32x4(t, t, _MM_SHUFFLE(1, 0, 3, 2)), p);
    return addm(y, t, p);
  }
  // sum of the sixteen lanes as an integer (< 16 P, no reduction on the way)
  P3R_AVX512_FN static inline uint64_t hsum(__m512i v) {
    const __m512i lo = _mm512_and_si512(v, _mm512_set1_epi64(0xFFFFFFFFll));
    const __m512i s = _mm512_add_epi64(lo, _mm512_srli_epi64(v, 32));
    return (uint64_t)_mm512_reduce_add_epi64(s);
  }

  P3R_AVX512_FN static void permute(F* state, const uint32_t* rc) {
    const __m512i p = _mm512_set1_epi32((int)PP::P), neg_mu = _mm512_set1_epi64((long long)F::NEG_MU);
    __m512i s = _mm512_loadu_si512(state);
    s = external(s, p);
    int k = 0;
    for (int r = 0; r < P2_HALF_FULL; ++r) {
      s = sbox(addm(s, _mm512_loadu_si512(rc + k), p), p, neg_mu);
      k += P2_WIDTH;
      s = external(s, p);
    }
    {
      // partial rounds: element 0 in a scalar register, lane 0 of the vector held at zero
      F s0 = F::raw((uint32_t)_mm_cvtsi128_si32(_mm512_castsi512_si128(s)));
      s = _mm512_maskz_mov_epi32(0xFFFE, s);
      const __m512i dg = _mm512_load_si512(diag());
      for (int r = 0; r < PP::PARTIAL_ROUNDS; ++r) {
        const F sb = p2_sbox<PP>(s0 + F::raw(rc[k + r]));
        const F rest = F::raw((uint32_t)(hsum(s) % PP::P));  // the other fifteen elements
        const F sum = rest + sb;
        s0 = rest - sb;  // -2 * sb + sum
        const __m512i prod = mulm(s, dg, p, neg_mu);
        s = _mm512_maskz_mov_epi32(0xFFFE, addm(prod, _mm512_set1_epi32((int)sum.v), p));
      }
      k += PP::PARTIAL_ROUNDS;
      s = _mm512_mask_set1_epi32(s, 0x0001, (int)s0.v);
    }
    for (int r = 0; r < P2_HALF_FULL; ++r) {
      s = sbox(addm(s, _mm512_loadu_si512(rc + k), p), p, neg_mu);
      k += P2_WIDTH;
      s = external(s, p);
    }
    _mm512_storeu_si512(state, s);
  }
};
#endif  // P3R_HOST_AVX512

// The permutation of the host transcript and of the native verifier's hashing.
template <class PP>
inline void host_permute(Fp<PP>* state, const uint32_t* rc) {
#if defined(P3R_HOST_AVX512)
  if (P2Avx512<PP>::supported()) return P2Avx512<PP>::permute(state, rc);
#endif
  p2_permute<PP>(state, rc);
}

}  // namespace p3r
