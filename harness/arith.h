// Arithmetic of the synthetic-workload generator: its own, so that test inputs never depend on the
// code under test (the product's csrc/) nor on the checker (oracle/).  Canonical u32 values,
// u64 `%` products, the degree-4 binomial extension by the schoolbook rule, and the Poseidon2
// width-16 permutation with M4 written out as an integer matrix.
//   moduli / extension      circuit-prover/src/batch_stark_prover.rs:76-78, air/alu_air.rs:715-733
//   round structure         circuit/src/ops/poseidon2_perm/config.rs:56-122
//   linear layers           SURVEY.md appendix A (p3-poseidon2 / p3-{koala,baby}-bear 0.6)
#pragma once
#include <cstdint>

namespace syn {

struct KoalaBearParams {
  static constexpr uint32_t P = 0x7f000001u, EXT_W = 3;
  static constexpr int SBOX_DEGREE = 3, PARTIAL_ROUNDS = 20;
  // internal diagonal as (sign, log2 of the denominator, small numerator): v = sign * num / 2^k
  static constexpr int DIAG[16][3] = {{-1, 0, 2}, {1, 0, 1}, {1, 0, 2}, {1, 1, 1}, {1, 0, 3}, {1, 0, 4}, {-1, 1, 1}, {-1, 0, 3},
                                      {-1, 0, 4}, {1, 8, 1}, {1, 3, 1}, {1, 24, 1}, {-1, 8, 1}, {-1, 3, 1}, {-1, 4, 1}, {-1, 24, 1}};
};
struct BabyBearParams {
  static constexpr uint32_t P = 0x78000001u, EXT_W = 11;
  static constexpr int SBOX_DEGREE = 7, PARTIAL_ROUNDS = 13;
  static constexpr int DIAG[16][3] = {{-1, 0, 2}, {1, 0, 1}, {1, 0, 2}, {1, 1, 1}, {1, 0, 3}, {1, 0, 4}, {-1, 1, 1}, {-1, 0, 3},
                                      {-1, 0, 4}, {1, 8, 1}, {1, 2, 1}, {1, 3, 1}, {1, 27, 1}, {-1, 8, 1}, {-1, 4, 1}, {-1, 27, 1}};
};

template <class PP>
struct Fp {
  static constexpr uint32_t P = PP::P;
  uint32_t v = 0;  // canonical
  static Fp from_canonical(uint32_t x) { Fp r; r.v = x % P; return r; }
  uint32_t to_canonical() const { return v; }
  static Fp zero() { return Fp(); }
  static Fp one() { return from_canonical(1); }
  friend Fp operator+(Fp a, Fp b) { uint32_t s = a.v + b.v; return from_canonical(s >= P ? s - P : s); }
  friend Fp operator-(Fp a, Fp b) { return from_canonical(a.v >= b.v ? a.v - b.v : a.v + P - b.v); }
  friend Fp operator*(Fp a, Fp b) { Fp r; r.v = (uint32_t)((uint64_t)a.v * b.v % P); return r; }
  Fp operator-() const { return from_canonical(v ? P - v : 0); }
  bool operator==(Fp o) const { return v == o.v; }
  Fp pow(uint64_t e) const {
    Fp r = one(), b = *this;
    for (; e; e >>= 1, b = b * b)
      if (e & 1) r = r * b;
    return r;
  }
  Fp inv() const { return pow((uint64_t)P - 2); }
};

template <class PP>
struct Fp4 {
  using F = Fp<PP>;
  static constexpr int DEG = 4;
  F c[4];
  static Fp4 zero() { return Fp4(); }
  static Fp4 one() { Fp4 r; r.c[0] = F::one(); return r; }
  static Fp4 from_base(F b) { Fp4 r; r.c[0] = b; return r; }
  friend Fp4 operator+(Fp4 a, const Fp4& b) { for (int i = 0; i < 4; ++i) a.c[i] = a.c[i] + b.c[i]; return a; }
  friend Fp4 operator-(Fp4 a, const Fp4& b) { for (int i = 0; i < 4; ++i) a.c[i] = a.c[i] - b.c[i]; return a; }
  friend Fp4 operator*(const Fp4& a, const Fp4& b) {
    Fp4 r;
    const F w = F::from_canonical(PP::EXT_W);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        const F t = a.c[i] * b.c[j];
        if (i + j >= 4) r.c[i + j - 4] = r.c[i + j - 4] + w * t;
        else r.c[i + j] = r.c[i + j] + t;
      }
    return r;
  }
  bool operator==(const Fp4& o) const { return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2] && c[3] == o.c[3]; }
  // a^-1 = a^(p^4 - 2): square-and-multiply over the 124-bit exponent
  Fp4 inv() const {
    const unsigned __int128 p = PP::P;
    unsigned __int128 e = p * p * p * p - 2;
    Fp4 r = one(), b = *this;
    for (; e; e >>= 1, b = b * b)
      if (e & 1) r = r * b;
    return r;
  }
};

// The base field as a "degree-1 extension": the element type of base-field circuits (CircuitBuilder<F>).
template <class PP>
struct Fp1 {
  using F = Fp<PP>;
  static constexpr int DEG = 1;
  F c[1];
  static Fp1 zero() { return Fp1(); }
  static Fp1 one() { Fp1 r; r.c[0] = F::one(); return r; }
  static Fp1 from_base(F b) { Fp1 r; r.c[0] = b; return r; }
  friend Fp1 operator+(Fp1 a, const Fp1& b) { a.c[0] = a.c[0] + b.c[0]; return a; }
  friend Fp1 operator-(Fp1 a, const Fp1& b) { a.c[0] = a.c[0] - b.c[0]; return a; }
  friend Fp1 operator*(const Fp1& a, const Fp1& b) { Fp1 r; r.c[0] = a.c[0] * b.c[0]; return r; }
  bool operator==(const Fp1& o) const { return c[0] == o.c[0]; }
  Fp1 inv() const { Fp1 r; r.c[0] = c[0].inv(); return r; }
};

// A binomial extension F[x] / (x^D - W) of any degree (the reference's circuit degrees 2, 6, 8:
// batch_stark_prover/tests.rs:486 runs D = 8 over KoalaBear).  Inverse by Gauss-Jordan on the multiplication matrix.
template <class PP, int D_, uint32_t W_>
struct FpBin {
  using F = Fp<PP>;
  static constexpr int DEG = D_;
  static constexpr uint32_t W = W_;
  F c[D_];
  static FpBin zero() { return FpBin(); }
  static FpBin one() { FpBin r; r.c[0] = F::one(); return r; }
  static FpBin from_base(F b) { FpBin r; r.c[0] = b; return r; }
  friend FpBin operator+(FpBin a, const FpBin& b) { for (int i = 0; i < D_; ++i) a.c[i] = a.c[i] + b.c[i]; return a; }
  friend FpBin operator-(FpBin a, const FpBin& b) { for (int i = 0; i < D_; ++i) a.c[i] = a.c[i] - b.c[i]; return a; }
  friend FpBin operator*(const FpBin& a, const FpBin& b) {
    FpBin r;
    const F w = F::from_canonical(W_);
    for (int i = 0; i < D_; ++i)
      for (int j = 0; j < D_; ++j) {
        const F t = a.c[i] * b.c[j];
        if (i + j >= D_) r.c[i + j - D_] = r.c[i + j - D_] + w * t;
        else r.c[i + j] = r.c[i + j] + t;
      }
    return r;
  }
  bool operator==(const FpBin& o) const {
    for (int i = 0; i < D_; ++i) if (!(c[i] == o.c[i])) return false;
    return true;
  }
  FpBin inv() const {
    F m[D_][D_ + 1];
    FpBin col = *this, xgen;
    xgen.c[1] = F::one();
    for (int j = 0; j < D_; ++j) {
      for (int i = 0; i < D_; ++i) m[i][j] = col.c[i];
      col = col * xgen;
    }
    for (int i = 0; i < D_; ++i) m[i][D_] = i == 0 ? F::one() : F::zero();
    for (int k = 0; k < D_; ++k) {
      int piv = k;
      while (piv < D_ && m[piv][k] == F::zero()) ++piv;
      if (piv == D_) return zero();  // zero, or a zero divisor when x^D - W is reducible
      for (int j = 0; j <= D_; ++j) { F t = m[k][j]; m[k][j] = m[piv][j]; m[piv][j] = t; }
      const F s = m[k][k].inv();
      for (int j = 0; j <= D_; ++j) m[k][j] = m[k][j] * s;
      for (int i = 0; i < D_; ++i) {
        if (i == k) continue;
        const F f = m[i][k];
        for (int j = 0; j <= D_; ++j) m[i][j] = m[i][j] - f * m[k][j];
      }
    }
    FpBin r;
    for (int i = 0; i < D_; ++i) r.c[i] = m[i][D_];
    return r;
  }
};

// The degree-5 extension F[x] / (x^5 + x^2 - 1) of KoalaBear (QuinticTrinomialExtensionField; the circuit
// field of the reference's D = 5 unit tests, circuit-prover/src/batch_stark_prover.rs tests.rs:844-1029 and
// air/alu_air.rs:735-760): schoolbook product of degree 8, then x^5 = 1 - x^2 applied from the top.
template <class PP>
struct Fp5 {
  using F = Fp<PP>;
  static constexpr int DEG = 5;
  F c[5];
  static Fp5 zero() { return Fp5(); }
  static Fp5 one() { Fp5 r; r.c[0] = F::one(); return r; }
  static Fp5 from_base(F b) { Fp5 r; r.c[0] = b; return r; }
  friend Fp5 operator+(Fp5 a, const Fp5& b) { for (int i = 0; i < 5; ++i) a.c[i] = a.c[i] + b.c[i]; return a; }
  friend Fp5 operator-(Fp5 a, const Fp5& b) { for (int i = 0; i < 5; ++i) a.c[i] = a.c[i] - b.c[i]; return a; }
  friend Fp5 operator*(const Fp5& a, const Fp5& b) {
    F t[9];
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) t[i + j] = t[i + j] + a.c[i] * b.c[j];
    for (int k = 8; k >= 5; --k) {  // x^k = x^(k-5) - x^(k-3)
      t[k - 5] = t[k - 5] + t[k];
      t[k - 3] = t[k - 3] - t[k];
    }
    Fp5 r;
    for (int i = 0; i < 5; ++i) r.c[i] = t[i];
    return r;
  }
  bool operator==(const Fp5& o) const {
    for (int i = 0; i < 5; ++i) if (!(c[i] == o.c[i])) return false;
    return true;
  }
  // a^-1 by solving (multiplication-by-a matrix) * x = 1 over the base field (Gauss-Jordan, 5 x 6)
  Fp5 inv() const {
    F m[5][6];
    Fp5 col = *this, xgen;
    xgen.c[1] = F::one();
    for (int j = 0; j < 5; ++j) {
      for (int i = 0; i < 5; ++i) m[i][j] = col.c[i];
      col = col * xgen;
    }
    for (int i = 0; i < 5; ++i) m[i][5] = i == 0 ? F::one() : F::zero();
    for (int k = 0; k < 5; ++k) {
      int piv = k;
      while (piv < 5 && m[piv][k] == F::zero()) ++piv;
      if (piv == 5) return zero();  // a = 0
      for (int j = 0; j < 6; ++j) { F t = m[k][j]; m[k][j] = m[piv][j]; m[piv][j] = t; }
      const F s = m[k][k].inv();
      for (int j = 0; j < 6; ++j) m[k][j] = m[k][j] * s;
      for (int i = 0; i < 5; ++i) {
        if (i == k) continue;
        const F f = m[i][k];
        for (int j = 0; j < 6; ++j) m[i][j] = m[i][j] - f * m[k][j];
      }
    }
    Fp5 r;
    for (int i = 0; i < 5; ++i) r.c[i] = m[i][5];
    return r;
  }
};

template <class PP>
constexpr int p2_num_constants() { return 2 * 4 * 16 + PP::PARTIAL_ROUNDS; }

// state <- circ(2 M4, M4, M4, M4) * state, M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
template <class F>
void p2_external(F* s) {
  static const int M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
  F y[16];
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < 4; ++i) {
      uint64_t acc = 0;  // 7 * P < 2^34
      for (int j = 0; j < 4; ++j) acc += (uint64_t)M4[i][j] * s[4 * b + j].v;
      y[4 * b + i].v = (uint32_t)(acc % F::P);
    }
  for (int i = 0; i < 16; ++i) s[i] = y[i] + y[i & 3] + y[4 + (i & 3)] + y[8 + (i & 3)] + y[12 + (i & 3)];
}
template <class PP, class F>
const F* p2_diag() {
  static F d[16];
  static const bool init = [] {
    for (int i = 0; i < 16; ++i) {
      d[i] = F::from_canonical((uint32_t)PP::DIAG[i][2]) * F::from_canonical(2).pow((uint64_t)PP::DIAG[i][1]).inv();
      if (PP::DIAG[i][0] < 0) d[i] = -d[i];
    }
    return true;
  }();
  (void)init;
  return d;
}
template <class PP, class F>
void p2_internal(F* s) {
  const F* d = p2_diag<PP, F>();
  F sum = F::zero();
  for (int i = 0; i < 16; ++i) sum = sum + s[i];
  for (int i = 0; i < 16; ++i) s[i] = s[i] * d[i] + sum;
}
// `rc`: flat canonical table [4][16] external-initial | [PARTIAL] internal | [4][16] external-final
template <class PP, class F>
F p2_sbox(F x) {
  const F x3 = x * x * x;
  return PP::SBOX_DEGREE == 3 ? x3 : x3 * x3 * x;
}
template <class PP, class F>
void p2_permute(F* s, const uint32_t* rc) {
  p2_external(s);
  int k = 0;
  auto full = [&]() {
    for (int i = 0; i < 16; ++i) s[i] = p2_sbox<PP>(s[i] + F::from_canonical(rc[k + i]));
    k += 16;
    p2_external(s);
  };
  for (int r = 0; r < 4; ++r) full();
  for (int r = 0; r < PP::PARTIAL_ROUNDS; ++r) {
    s[0] = p2_sbox<PP>(s[0] + F::from_canonical(rc[k++]));
    p2_internal<PP>(s);
  }
  for (int r = 0; r < 4; ++r) full();
}

}  // namespace syn
