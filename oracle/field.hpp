// ORACLE (test infrastructure, not product code): plain canonical-form arithmetic for the
// KoalaBear / BabyBear prime fields and their degree-4 binomial extension.
//
// PARITY UNPINNED: the reference's arithmetic lives in the un-vendored p3-* 0.6 crates
// (no Cargo.lock, cargo/rustc absent), and the reference holds no golden vectors, so this
// restatement is pinned only against the in-tree protocol restatements cited per function.
//
// Deliberately written differently from plonky3_recursion_amd/csrc/field.h (no Montgomery
// form, u64 `%` reduction) so that it is an independent check of the device arithmetic.
//
//   moduli                    circuit-prover/src/batch_stark_prover.rs:76-78
//   x^4 = W extension mul     circuit-prover/src/air/alu_air.rs:715-733
//   W extraction              circuit-prover/src/field_params.rs:46-53
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <vector>

namespace orc {

struct KoalaBear {
  static constexpr uint32_t P = 0x7f000001u;
  static constexpr uint32_t GEN = 3;
  static constexpr int TWO_ADICITY = 24;
  static constexpr uint32_t W = 3;
  static constexpr int SBOX_DEGREE = 3;
  static constexpr int SBOX_REGS = 0;
  static constexpr int PARTIAL = 20;
  static constexpr int PARTIAL_W32 = 31;   // Poseidon2Config::KOALA_BEAR_D4_W32 (circuit/src/ops/poseidon2_perm/config.rs:164-172)
};
struct BabyBear {
  static constexpr uint32_t P = 0x78000001u;
  static constexpr uint32_t GEN = 31;
  static constexpr int TWO_ADICITY = 27;
  static constexpr uint32_t W = 11;
  static constexpr int SBOX_DEGREE = 7;
  static constexpr int SBOX_REGS = 1;
  static constexpr int PARTIAL = 13;
  static constexpr int PARTIAL_W32 = 30;   // Poseidon2Config::BABY_BEAR_D4_W32 (config.rs:88-100)
};

template <class FP>
struct Fe {
  using Params = FP;
  static constexpr uint32_t P = FP::P;
  uint32_t v = 0;  // canonical
  Fe() = default;
  explicit Fe(uint64_t x) : v((uint32_t)(x % P)) {}
  static Fe zero() { return Fe(); }
  static Fe one() { return Fe(1); }
  static Fe from_i64(int64_t x) {
    int64_t m = x % (int64_t)P;
    if (m < 0) m += P;
    return Fe((uint64_t)m);
  }
  friend Fe operator+(Fe a, Fe b) { return Fe((uint64_t)a.v + b.v); }
  friend Fe operator-(Fe a, Fe b) { return Fe((uint64_t)a.v + P - b.v); }
  friend Fe operator*(Fe a, Fe b) { return Fe((uint64_t)a.v * b.v); }
  Fe operator-() const { return Fe((uint64_t)(P - v)); }
  Fe& operator+=(Fe o) { return *this = *this + o; }
  Fe& operator-=(Fe o) { return *this = *this - o; }
  Fe& operator*=(Fe o) { return *this = *this * o; }
  bool operator==(Fe o) const { return v == o.v; }
  bool operator!=(Fe o) const { return v != o.v; }
  Fe pow(uint64_t e) const {
    Fe r = one(), b = *this;
    for (; e; e >>= 1) {
      if (e & 1) r *= b;
      b *= b;
    }
    return r;
  }
  Fe inv() const {
    if (v == 0) throw std::runtime_error("inverse of zero");
    return pow((uint64_t)P - 2);
  }
  static Fe generator() { return Fe(FP::GEN); }
  // F::two_adic_generator(bits): GEN^((P-1)/2^bits)
  static Fe two_adic_generator(int bits) {
    if (bits > FP::TWO_ADICITY) throw std::runtime_error("two-adicity exceeded");
    return generator().pow(((uint64_t)P - 1) >> bits);
  }
};

// The STARK's CHALLENGE field.  Degree 4: F[x]/(x^4 - W), coefficient order 1, x, x^2, x^3 (every BASELINE
// configuration).  Degree 5: F[x]/(x^5 + x^2 - 1), p3's QuinticTrinomialExtensionField, the challenge field of
// `koala_bear_quintic_params` (test-utils/src/lib.rs:414-460; recursive_fibonacci.rs --quintic).  The degree is a
// process-wide setting of this test oracle (challenge_degree(), set by the C entry points from orc_params before a
// prove / verify); the name Fe4 stays.
inline int& challenge_degree() {
  static int d = 4;
  return d;
}
template <class FP>
struct Fe4 {
  using F = Fe<FP>;
  std::array<F, 5> c{};
  static int deg() { return challenge_degree(); }
  Fe4() = default;
  explicit Fe4(F b) { c[0] = b; }
  Fe4(F a, F b, F cc, F d) { c = {a, b, cc, d, F::zero()}; }
  static Fe4 zero() { return Fe4(); }
  static Fe4 one() { return Fe4(F::one()); }
  friend Fe4 operator+(Fe4 a, const Fe4& b) {
    for (int i = 0; i < 5; ++i) a.c[i] += b.c[i];
    return a;
  }
  friend Fe4 operator-(Fe4 a, const Fe4& b) {
    for (int i = 0; i < 5; ++i) a.c[i] -= b.c[i];
    return a;
  }
  Fe4 operator-() const {
    Fe4 r;
    for (int i = 0; i < 5; ++i) r.c[i] = -c[i];
    return r;
  }
  friend Fe4 operator*(const Fe4& a, const Fe4& b) {
    Fe4 r;
    if (deg() == 5) {
      F t[9];
      for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) t[i + j] += a.c[i] * b.c[j];
      // x^5 = 1 - x^2, x^6 = x - x^3, x^7 = x^2 - x^4, x^8 = x^3 + x^2 - 1
      r.c[0] = t[0] + t[5] - t[8];
      r.c[1] = t[1] + t[6];
      r.c[2] = t[2] - t[5] + t[7] + t[8];
      r.c[3] = t[3] - t[6] + t[8];
      r.c[4] = t[4] - t[7];
      return r;
    }
    // schoolbook with wrap-around x^4 -> W
    const F w(FP::W);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        F t = a.c[i] * b.c[j];
        if (i + j >= 4) r.c[i + j - 4] += w * t;
        else r.c[i + j] += t;
      }
    return r;
  }
  friend Fe4 operator*(Fe4 a, F b) {
    for (int i = 0; i < 5; ++i) a.c[i] *= b;
    return a;
  }
  Fe4& operator+=(const Fe4& o) { return *this = *this + o; }
  Fe4& operator-=(const Fe4& o) { return *this = *this - o; }
  Fe4& operator*=(const Fe4& o) { return *this = *this * o; }
  bool operator==(const Fe4& o) const { return c == o.c; }
  bool operator!=(const Fe4& o) const { return !(c == o.c); }
  bool is_zero() const { return c[0].v == 0 && c[1].v == 0 && c[2].v == 0 && c[3].v == 0 && c[4].v == 0; }
  Fe4 pow(uint64_t e) const {
    Fe4 r = one(), b = *this;
    for (; e; e >>= 1) {
      if (e & 1) r *= b;
      b *= b;
    }
    return r;
  }
  // Degree 4: a^-1 = a^(p^4 - 2) through the Frobenius norm: a^-1 = a^(r-1) / N(a), r = 1 + p + p^2 + p^3.
  // Degree 5: the solution of (multiplication-by-a matrix) x = 1 over the base field (Gauss-Jordan).
  Fe4 inv() const {
    if (is_zero()) throw std::runtime_error("inverse of zero");
    if (deg() == 5) {
      F m[5][6];
      Fe4 col = *this, x;
      x.c[1] = F::one();
      for (int j = 0; j < 5; ++j) {
        for (int i = 0; i < 5; ++i) m[i][j] = col.c[i];
        col = col * x;
      }
      for (int i = 0; i < 5; ++i) m[i][5] = i == 0 ? F::one() : F::zero();
      for (int k = 0; k < 5; ++k) {
        int piv = k;
        while (piv < 5 && m[piv][k].v == 0) ++piv;
        if (piv == 5) throw std::runtime_error("singular multiplication matrix");
        for (int j = 0; j < 6; ++j) std::swap(m[k][j], m[piv][j]);
        const F s = m[k][k].inv();
        for (int j = 0; j < 6; ++j) m[k][j] *= s;
        for (int i = 0; i < 5; ++i) {
          if (i == k) continue;
          const F f = m[i][k];
          for (int j = 0; j < 6; ++j) m[i][j] -= f * m[k][j];
        }
      }
      Fe4 r;
      for (int i = 0; i < 5; ++i) r.c[i] = m[i][5];
      return r;
    }
    Fe4 f1 = frobenius(), f2 = f1.frobenius(), f3 = f2.frobenius();
    Fe4 prod = f1 * f2 * f3;  // a^(r-1)
    Fe4 norm = *this * prod;  // in the base field
    return prod * norm.c[0].inv();
  }
  // x -> x^p : coefficient i is multiplied by (W^((p-1)/4))^i  (degree 4 only)
  Fe4 frobenius() const {
    const F z = F(FP::W).pow(((uint64_t)FP::P - 1) / 4);
    Fe4 r;
    F zi = F::one();
    for (int i = 0; i < 4; ++i) {
      r.c[i] = c[i] * zi;
      zi *= z;
    }
    return r;
  }
};

// Element of the CIRCUIT's extension field with the degree chosen at run time: F[x]/(x^4 - W) or, for D = 5,
// F[x]/(x^5 + x^2 - 1) (p3's QuinticTrinomialExtensionField; reduction as in alu_air.rs:737-765).  Used by the
// trace builders; the STARK's challenge field is Fe4 whatever the circuit's degree.
template <class FP>
struct FeX {
  using F = Fe<FP>;
  int D = 4;
  uint32_t W = 0;   // binomial x^D = W; 0: the field's own rule for the degree (W of the quartic, the quintic trinomial)
  std::array<F, 8> c{};
  explicit FeX(int d = 4, uint32_t w = 0) : D(d), W(w) {}
  static FeX zero(int d) { return FeX(d); }
  friend FeX operator+(FeX a, const FeX& b) { for (int i = 0; i < a.D; ++i) a.c[i] += b.c[i]; return a; }
  friend FeX operator-(FeX a, const FeX& b) { for (int i = 0; i < a.D; ++i) a.c[i] -= b.c[i]; return a; }
  friend FeX operator*(const FeX& a, const FeX& b) {
    FeX r(a.D, a.W);
    if (a.D == 5 && a.W == 0) {
      F t[9];
      for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) t[i + j] += a.c[i] * b.c[j];
      // x^5 = 1 - x^2, x^6 = x - x^3, x^7 = x^2 - x^4, x^8 = x^3 + x^2 - 1
      r.c[0] = t[0] + t[5] - t[8];
      r.c[1] = t[1] + t[6];
      r.c[2] = t[2] - t[5] + t[7] + t[8];
      r.c[3] = t[3] - t[6] + t[8];
      r.c[4] = t[4] - t[7];
      return r;
    }
    const F w(a.W ? a.W : FP::W);
    for (int i = 0; i < a.D; ++i)
      for (int j = 0; j < a.D; ++j) {
        F t = a.c[i] * b.c[j];
        if (i + j >= a.D) r.c[i + j - a.D] += w * t;
        else r.c[i + j] += t;
      }
    return r;
  }
};

inline uint32_t bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}
inline int log2_strict(size_t n) {
  int l = 0;
  while ((size_t(1) << l) < n) ++l;
  if ((size_t(1) << l) != n) throw std::runtime_error("not a power of two");
  return l;
}

}  // namespace orc
