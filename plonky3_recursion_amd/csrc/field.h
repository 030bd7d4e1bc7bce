// 31-bit Montgomery prime fields (KoalaBear, BabyBear) and their degree-4 binomial
// extension, shared by host orchestration code and gfx950 device kernels.
//
// Reference anchors (the arithmetic itself lives in the un-vendored p3-monty-31 /
// p3-koala-bear / p3-baby-bear 0.6 crates):
//   moduli                      circuit-prover/src/batch_stark_prover.rs:76-78
//   x^4 = W binomial extension   circuit-prover/src/air/alu_air.rs:715-733,
//                                circuit-prover/src/field_params.rs:46-53
//   coset shift = F::GENERATOR   recursion/src/pcs/fri/verifier.rs:960
//
// Device representation: Montgomery form x*2^32 mod P held in a u32 in [0, P).
// Everything that crosses the C ABI is canonical (see include/p3r.h).
#pragma once
#include <type_traits>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define P3R_HD __host__ __device__ __forceinline__
#else
#define P3R_HD inline
#endif

namespace p3r {

#if defined(__HIPCC__)
// A pointer that a kernel reads out of a job list in memory is a generic ("flat") address to the
// compiler: 64-bit vector addresses and flat loads.  Kernel-argument pointers are known to be
// global; as_global() says the same about a loaded one (every buffer here is hipMalloc'ed), which
// gives scalar-base addressing back.
template <class T>
using gptr = T __attribute__((address_space(1)))*;
template <class T>
__device__ __forceinline__ gptr<T> as_global(T* p) {
  return (gptr<T>)p;
}

// The job a workgroup of a job-list launch belongs to: the last one whose first block is not past it.  A ZK commit of
// quotient chunks is forty matrices, the openings of a proof twenty-five to sixty: a linear walk over their descriptors is
// that many dependent scalar loads before the workgroup starts; a bisection takes five or six.
template <class Job>
__device__ __forceinline__ int find_job(const Job* __restrict__ jobs, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (blockIdx.x >= jobs[mid].block0) lo = mid; else hi = mid - 1;
  }
  return lo;
}
#endif

constexpr uint32_t inv_mod_2_32(uint32_t p) {
  // Newton iteration: x <- x*(2 - p*x); doubles the number of correct low bits.
  uint32_t x = 1;
  for (int i = 0; i < 6; ++i) x *= 2u - p * x;
  return x;
}
constexpr uint32_t pow_mod(uint32_t b, uint64_t e, uint32_t p) {
  uint64_t r = 1, x = b % p;
  while (e) {
    if (e & 1) r = r * x % p;
    x = x * x % p;
    e >>= 1;
  }
  return (uint32_t)r;
}

// Field parameter packs. GEN is the multiplicative generator upstream uses as the
// LDE coset shift; the 2-adic generator is GEN^((P-1)/2^TWO_ADICITY), which
// reproduces the upstream tables (0x6ac49f88 for KoalaBear, 0x1a427a41 for BabyBear).
struct KoalaBearParams {
  static constexpr uint32_t P = 0x7f000001u;  // 2^31 - 2^24 + 1
  static constexpr uint32_t GEN = 3;
  static constexpr int TWO_ADICITY = 24;
  static constexpr uint32_t EXT_W = 3;  // x^4 = 3
  static constexpr int SBOX_DEGREE = 3;
  static constexpr int SBOX_REGISTERS = 0;
  static constexpr int PARTIAL_ROUNDS = 20;
  static constexpr int PARTIAL_ROUNDS_W32 = 31;  // Poseidon2Config::KOALA_BEAR_D4_W32 (circuit/src/ops/poseidon2_perm/config.rs:164-172)
  static constexpr int FIELD_ID = 0;
};
struct BabyBearParams {
  static constexpr uint32_t P = 0x78000001u;  // 2^31 - 2^27 + 1
  static constexpr uint32_t GEN = 31;
  static constexpr int TWO_ADICITY = 27;
  static constexpr uint32_t EXT_W = 11;  // x^4 = 11
  static constexpr int SBOX_DEGREE = 7;
  static constexpr int SBOX_REGISTERS = 1;
  static constexpr int PARTIAL_ROUNDS = 13;
  static constexpr int PARTIAL_ROUNDS_W32 = 30;  // Poseidon2Config::BABY_BEAR_D4_W32 (config.rs:88-100)
  static constexpr int FIELD_ID = 1;
};

template <class PP>
struct Fp {
  using Params = PP;
  static constexpr uint32_t P = PP::P;
  static constexpr uint32_t MU = inv_mod_2_32(PP::P);          // P*MU == 1 mod 2^32
  static constexpr uint32_t R1 = (uint32_t)((1ull << 32) % PP::P);  // mont(1)
  static constexpr uint32_t R2 = (uint32_t)((uint64_t)R1 * R1 % PP::P);

  uint32_t v;  // Montgomery form, < P

  static P3R_HD uint32_t mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
  }
  static constexpr uint32_t NEG_MU = 0u - MU;                   // P*NEG_MU == -1 mod 2^32
  // x < P*2^32  ->  x * 2^-32 mod P
  static P3R_HD uint32_t reduce64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // REDC as ONE 64-bit multiply-add: x + q*P has a zero low word, its high word is < 2P.
    // On gfx950 v_mad_u64_u32 makes the Montgomery product 33 % cheaper than the mul_lo /
    // mul_hi / borrow form (tools/microbench/int_rates.hip: 7.7 vs 5.8 T products/s).
    const uint32_t q = (uint32_t)x * NEG_MU;
    const uint32_t r = (uint32_t)(((uint64_t)q * P + x) >> 32);
    const uint32_t r2 = r - P;
    return r < r2 ? r : r2;
#else
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    uint32_t t = lo * MU;
    uint32_t u = mulhi(t, P);
    uint32_t r = hi - u;
    return hi < u ? r + P : r;
#endif
  }
  static P3R_HD uint32_t reduce(uint32_t lo, uint32_t hi) { return reduce64(((uint64_t)hi << 32) | lo); }
  // REDC without the final conditional subtraction: a value in [0, 2P) congruent to x * 2^-32.
  // Usable wherever the next operation is a product with a fully reduced factor (2P * P < P * 2^32).
  static P3R_HD uint32_t reduce64_lazy(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t q = (uint32_t)x * NEG_MU;
    return (uint32_t)(((uint64_t)q * P + x) >> 32);
#else
    return reduce64(x);
#endif
  }
  // (a * a) * x with the square left unreduced in between
  P3R_HD Fp sqr_times(Fp x) const {
    const uint32_t sq = reduce64_lazy((uint64_t)v * v);
    return raw(reduce64((uint64_t)sq * x.v));
  }
  // a * a * a with the square left unreduced in between
  P3R_HD Fp cube() const {
    const uint32_t sq = reduce64_lazy((uint64_t)v * v);
    return raw(reduce64((uint64_t)sq * v));
  }
  static P3R_HD Fp raw(uint32_t m) { Fp r; r.v = m; return r; }
  static P3R_HD Fp zero() { return raw(0); }
  static P3R_HD Fp one() { return raw(R1); }
  static P3R_HD Fp from_canonical(uint32_t x) {  // x < P
    return raw(reduce64((uint64_t)x * R2));
  }
  static P3R_HD Fp from_u64(uint64_t x) { return from_canonical((uint32_t)(x % P)); }
  P3R_HD uint32_t to_canonical() const { return reduce(v, 0); }

  friend P3R_HD Fp operator+(Fp a, Fp b) {
    uint32_t s = a.v + b.v;
    return raw(s >= P ? s - P : s);
  }
  friend P3R_HD Fp operator-(Fp a, Fp b) {
    uint32_t d = a.v - b.v;
    return raw(a.v < b.v ? d + P : d);
  }
  friend P3R_HD Fp operator*(Fp a, Fp b) { return raw(reduce64((uint64_t)a.v * b.v)); }
  // a1*b1 + a2*b2 with ONE reduction: 2*P^2 < P*2^32, so the sum of two products is still inside
  // REDC's input range (inner products and constraint folds pair their terms)
  static P3R_HD Fp dot2(Fp a1, Fp b1, Fp a2, Fp b2) {
    return raw(reduce64((uint64_t)a1.v * b1.v + (uint64_t)a2.v * b2.v));
  }
  P3R_HD Fp operator-() const { return raw(v ? P - v : 0); }
  P3R_HD Fp& operator+=(Fp o) { *this = *this + o; return *this; }
  P3R_HD Fp& operator-=(Fp o) { *this = *this - o; return *this; }
  P3R_HD Fp& operator*=(Fp o) { *this = *this * o; return *this; }
  P3R_HD bool operator==(Fp o) const { return v == o.v; }
  P3R_HD bool operator!=(Fp o) const { return v != o.v; }
  P3R_HD Fp dbl() const { return *this + *this; }
  P3R_HD Fp sqr() const { return *this * *this; }
  P3R_HD Fp halve() const {
    // (v + (v odd ? P : 0)) / 2; Montgomery form is linear so halving commutes.
    uint32_t t = (v & 1) ? v + P : v;  // < 2^32 since P < 2^31
    return raw(t >> 1);
  }
  P3R_HD Fp pow(uint64_t e) const {
    Fp r = one(), b = *this;
    while (e) {
      if (e & 1) r *= b;
      b = b.sqr();
      e >>= 1;
    }
    return r;
  }
  P3R_HD Fp inv() const { return pow((uint64_t)P - 2); }

  static P3R_HD Fp generator() { return from_canonical(PP::GEN); }
  // Generator of the order-2^bits subgroup (upstream two_adic_generator(bits)).
  static P3R_HD Fp two_adic_generator(int bits) {
    Fp g = from_canonical(PP::GEN).pow(((uint64_t)P - 1) >> PP::TWO_ADICITY);
    for (int i = PP::TWO_ADICITY; i > bits; --i) g = g.sqr();
    return g;
  }
};

// Degree-4 binomial extension F[x]/(x^4 - W), basis 1,x,x^2,x^3, flattened in that
// order wherever an extension element is stored as 4 base elements.
template <class PP>
struct Fp4 {
  using F = Fp<PP>;
  F c[4];

  static P3R_HD Fp4 zero() { Fp4 r; r.c[0] = r.c[1] = r.c[2] = r.c[3] = F::zero(); return r; }
  static P3R_HD Fp4 one() { Fp4 r = zero(); r.c[0] = F::one(); return r; }
  static P3R_HD Fp4 from_base(F b) { Fp4 r = zero(); r.c[0] = b; return r; }
  static P3R_HD F w() { return F::from_canonical(PP::EXT_W); }

  friend P3R_HD Fp4 operator+(Fp4 a, Fp4 b) {
    Fp4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] + b.c[i];
    return r;
  }
  friend P3R_HD Fp4 operator-(Fp4 a, Fp4 b) {
    Fp4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] - b.c[i];
    return r;
  }
  P3R_HD Fp4 operator-() const {
    Fp4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = -c[i];
    return r;
  }
  friend P3R_HD Fp4 operator*(Fp4 a, Fp4 b) {
    // W folded into a's high coefficients, then every output coefficient is two paired products
    const F W = w();
    const F wa1 = W * a.c[1], wa2 = W * a.c[2], wa3 = W * a.c[3];
    Fp4 r;
    r.c[0] = F::dot2(a.c[0], b.c[0], wa1, b.c[3]) + F::dot2(wa2, b.c[2], wa3, b.c[1]);
    r.c[1] = F::dot2(a.c[0], b.c[1], a.c[1], b.c[0]) + F::dot2(wa2, b.c[3], wa3, b.c[2]);
    r.c[2] = F::dot2(a.c[0], b.c[2], a.c[1], b.c[1]) + F::dot2(a.c[2], b.c[0], wa3, b.c[3]);
    r.c[3] = F::dot2(a.c[0], b.c[3], a.c[1], b.c[2]) + F::dot2(a.c[2], b.c[1], a.c[3], b.c[0]);
    return r;
  }
  friend P3R_HD Fp4 operator*(Fp4 a, F b) {
    Fp4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] * b;
    return r;
  }
  // a1*b1 + a2*b2 for extension a's and base b's, one reduction per coefficient
  static P3R_HD Fp4 dot2_base(const Fp4& a1, F b1, const Fp4& a2, F b2) {
    Fp4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = F::dot2(a1.c[i], b1, a2.c[i], b2);
    return r;
  }
  P3R_HD Fp4& operator+=(Fp4 o) { *this = *this + o; return *this; }
  P3R_HD Fp4& operator-=(Fp4 o) { *this = *this - o; return *this; }
  P3R_HD Fp4& operator*=(Fp4 o) { *this = *this * o; return *this; }
  P3R_HD bool operator==(const Fp4& o) const {
    return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2] && c[3] == o.c[3];
  }
  P3R_HD bool is_zero() const { return (c[0].v | c[1].v | c[2].v | c[3].v) == 0; }
  P3R_HD Fp4 sqr() const { return *this * *this; }
  P3R_HD Fp4 dbl() const { Fp4 r; for (int i = 0; i < 4; ++i) r.c[i] = c[i].dbl(); return r; }
  P3R_HD Fp4 halve() const { Fp4 r; for (int i = 0; i < 4; ++i) r.c[i] = c[i].halve(); return r; }
  P3R_HD Fp4 pow(uint64_t e) const {
    Fp4 r = one(), b = *this;
    while (e) {
      if (e & 1) r *= b;
      b = b.sqr();
      e >>= 1;
    }
    return r;
  }
  // Inverse through the norm to the quadratic subfield F[y]/(y^2 - W), y = x^2:
  // a = A + x*B with A = a0 + a2*y, B = a1 + a3*y; a^-1 = (A - x*B) / (A^2 - y*B^2).
  // The inverse in two halves, so that callers inverting many elements can share ONE base-field
  // inversion among them (Montgomery's trick on the norms): norm() is the element's norm down to
  // the base field, inv_given(1/norm()) finishes.  inv() = inv_given(norm().inv()).
  struct Norm { F n0, n1, d; };
  P3R_HD Norm norm() const {
    const F W = w();
    // A^2 = (a0^2 + W a2^2) + (2 a0 a2) y ; B^2 = (a1^2 + W a3^2) + (2 a1 a3) y
    F A0 = c[0] * c[0] + W * (c[2] * c[2]);
    F A1 = (c[0] * c[2]).dbl();
    F B0 = c[1] * c[1] + W * (c[3] * c[3]);
    F B1 = (c[1] * c[3]).dbl();
    // N = A^2 - y*B^2 = (A0 - W*B1) + (A1 - B0) y ;  norm of N down to F:  n0^2 - W n1^2
    Norm r;
    r.n0 = A0 - W * B1;
    r.n1 = A1 - B0;
    r.d = r.n0 * r.n0 - W * (r.n1 * r.n1);
    return r;
  }
  P3R_HD Fp4 inv() const {
    const Norm nm = norm();
    return inv_given(nm, nm.d.inv());
  }
  P3R_HD Fp4 inv_given(const Norm& nm, F d) const {
    const F W = w();
    // 1/N = (n0 - n1 y) / (n0^2 - W n1^2)
    F i0 = nm.n0 * d;
    F i1 = -(nm.n1 * d);
    // (A - xB) * (i0 + i1 y):  A*(i0+i1 y) = (a0 i0 + W a2 i1) + (a0 i1 + a2 i0) y
    Fp4 r;
    r.c[0] = c[0] * i0 + W * (c[2] * i1);
    r.c[2] = c[0] * i1 + c[2] * i0;
    r.c[1] = -(c[1] * i0 + W * (c[3] * i1));
    r.c[3] = -(c[1] * i1 + c[3] * i0);
    return r;
  }
};

// Degree-5 extension F[x]/(x^5 + x^2 - 1) of KoalaBear (p3-field's QuinticTrinomialExtensionField): the field of
// D = 5 circuits.  Only what trace generation needs (the ALU table's packed-Horner intermediates): ring
// operations plus inv (Itoh-Tsujii) - also the STARK's challenge field under challenge_degree = 5 (Chal<PP, 5>).
template <class PP>
inline constexpr bool kHasQuintic = PP::P == 0x7f000001u;
// The base field as the element type of base-field circuits (D = 1: CircuitBuilder<F>).
template <class PP>
struct Fp1 {
  using F = Fp<PP>;
  F c[1];
  static P3R_HD Fp1 zero() { Fp1 r; r.c[0] = F::zero(); return r; }
  friend P3R_HD Fp1 operator+(Fp1 a, Fp1 b) { Fp1 r; r.c[0] = a.c[0] + b.c[0]; return r; }
  friend P3R_HD Fp1 operator-(Fp1 a, Fp1 b) { Fp1 r; r.c[0] = a.c[0] - b.c[0]; return r; }
  friend P3R_HD Fp1 operator*(Fp1 a, Fp1 b) { Fp1 r; r.c[0] = a.c[0] * b.c[0]; return r; }
  static P3R_HD Fp1 one() { Fp1 r; r.c[0] = F::one(); return r; }
  static P3R_HD Fp1 from_base(F b) { Fp1 r; r.c[0] = b; return r; }
  P3R_HD bool operator==(const Fp1& o) const { return c[0] == o.c[0]; }
  P3R_HD Fp1 inv() const { Fp1 r; r.c[0] = c[0].inv(); return r; }
};
// Calls fn(std::integral_constant<int, D>) for the circuit extension degree d of a context: 1, 4, or 5 (KoalaBear) -
// the degrees the device runner computes in.
template <class PP, class Fn>
inline void dispatch_ext_degree(int d, Fn&& fn) {
  if (d == 1) fn(std::integral_constant<int, 1>{});
  else if (d == 5) { if constexpr (kHasQuintic<PP>) fn(std::integral_constant<int, 5>{}); }
  else fn(std::integral_constant<int, 4>{});
}
// The same over every degree the AIR statements are written for: + the binomial extensions of degree 2, 6, 8
// (W at run time).
template <class PP, class Fn>
inline void dispatch_air_degree(int d, Fn&& fn) {
  if (d == 2) fn(std::integral_constant<int, 2>{});
  else if (d == 6) fn(std::integral_constant<int, 6>{});
  else if (d == 8) fn(std::integral_constant<int, 8>{});
  else dispatch_ext_degree<PP>(d, fn);
}
inline bool ext_degree_is_binomial_generic(uint32_t d) { return d == 2 || d == 6 || d == 8; }

// Frobenius constants of F[x]/(x^5 + x^2 - 1): coefficient J of x^(I * p^K) in Montgomery form, evaluated at compile
// time (the map a -> a^(p^K) is F-linear, so it is a 5 x 5 matrix over the base field whose column I is this power).
namespace quintic {
struct Q5 { uint64_t c[5]; };
constexpr Q5 q5_mul(Q5 a, Q5 b, uint64_t p) {
  uint64_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) t[i + j] = (t[i + j] + a.c[i] * b.c[j] % p) % p;
  for (int k = 8; k >= 5; --k) {  // x^k = x^(k-5) - x^(k-3)
    t[k - 5] = (t[k - 5] + t[k]) % p;
    t[k - 3] = (t[k - 3] + p - t[k]) % p;
  }
  return Q5{{t[0], t[1], t[2], t[3], t[4]}};
}
constexpr Q5 q5_pow(Q5 b, uint64_t e, uint64_t p) {
  Q5 r{{1, 0, 0, 0, 0}};
  while (e) {
    if (e & 1) r = q5_mul(r, b, p);
    b = q5_mul(b, b, p);
    e >>= 1;
  }
  return r;
}
constexpr uint32_t frob_entry(uint32_t p, int k, int i, int j) {
  Q5 x{{0, 1, 0, 0, 0}};
  for (int s = 0; s < k; ++s) x = q5_pow(x, p, p);   // x^(p^k)
  const Q5 xi = q5_pow(x, (uint64_t)i, p);
  return (uint32_t)((xi.c[j] << 32) % p);              // Montgomery form
}
}  // namespace quintic
template <class PP, int K, int I, int J>
inline constexpr uint32_t kFrob5 = quintic::frob_entry(PP::P, K, I, J);

template <class PP>
struct Fp5 {
  using F = Fp<PP>;
  F c[5];
  static P3R_HD Fp5 zero() { Fp5 r; for (int i = 0; i < 5; ++i) r.c[i] = F::zero(); return r; }
  friend P3R_HD Fp5 operator+(Fp5 a, Fp5 b) { Fp5 r; for (int i = 0; i < 5; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
  friend P3R_HD Fp5 operator-(Fp5 a, Fp5 b) { Fp5 r; for (int i = 0; i < 5; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
  friend P3R_HD Fp5 operator*(Fp5 a, Fp5 b) {
    // schoolbook coefficients c0..c8 (paired products share a reduction), then x^5 = 1 - x^2 from the top:
    // r0 = c0 + c5 - c8, r1 = c1 + c6, r2 = c2 - c5 + c7 + c8, r3 = c3 - c6 + c8, r4 = c4 - c7
    const F c5 = F::dot2(a.c[1], b.c[4], a.c[2], b.c[3]) + F::dot2(a.c[3], b.c[2], a.c[4], b.c[1]);
    const F c6 = F::dot2(a.c[2], b.c[4], a.c[3], b.c[3]) + a.c[4] * b.c[2];
    const F c7 = F::dot2(a.c[3], b.c[4], a.c[4], b.c[3]);
    const F c8 = a.c[4] * b.c[4];
    const F c58 = c5 - c8;
    Fp5 r;
    r.c[0] = a.c[0] * b.c[0] + c58;
    r.c[1] = F::dot2(a.c[0], b.c[1], a.c[1], b.c[0]) + c6;
    r.c[2] = F::dot2(a.c[0], b.c[2], a.c[1], b.c[1]) + a.c[2] * b.c[0] - c58 + c7;
    r.c[3] = F::dot2(a.c[0], b.c[3], a.c[1], b.c[2]) + F::dot2(a.c[2], b.c[1], a.c[3], b.c[0]) - c6 + c8;
    r.c[4] = F::dot2(a.c[0], b.c[4], a.c[1], b.c[3]) + F::dot2(a.c[2], b.c[2], a.c[3], b.c[1]) + a.c[4] * b.c[0] - c7;
    return r;
  }
  friend P3R_HD Fp5 operator*(Fp5 a, F b) { Fp5 r; for (int i = 0; i < 5; ++i) r.c[i] = a.c[i] * b; return r; }
  P3R_HD Fp5 operator-() const { Fp5 r; for (int i = 0; i < 5; ++i) r.c[i] = -c[i]; return r; }
  P3R_HD Fp5& operator+=(Fp5 o) { *this = *this + o; return *this; }
  P3R_HD Fp5& operator-=(Fp5 o) { *this = *this - o; return *this; }
  P3R_HD Fp5& operator*=(Fp5 o) { *this = *this * o; return *this; }
  P3R_HD bool is_zero() const { return (c[0].v | c[1].v | c[2].v | c[3].v | c[4].v) == 0; }
  P3R_HD Fp5 sqr() const { return *this * *this; }
  P3R_HD Fp5 dbl() const { Fp5 r; for (int i = 0; i < 5; ++i) r.c[i] = c[i].dbl(); return r; }
  P3R_HD Fp5 halve() const { Fp5 r; for (int i = 0; i < 5; ++i) r.c[i] = c[i].halve(); return r; }
  P3R_HD Fp5 pow(uint64_t e) const {
    Fp5 r = one(), b = *this;
    while (e) {
      if (e & 1) r *= b;
      b = b.sqr();
      e >>= 1;
    }
    return r;
  }
  static P3R_HD Fp5 dot2_base(const Fp5& a1, F b1, const Fp5& a2, F b2) {
    Fp5 r;
    for (int i = 0; i < 5; ++i) r.c[i] = F::dot2(a1.c[i], b1, a2.c[i], b2);
    return r;
  }
  static P3R_HD Fp5 one() { Fp5 r = zero(); r.c[0] = F::one(); return r; }
  static P3R_HD Fp5 from_base(F b) { Fp5 r = zero(); r.c[0] = b; return r; }
  P3R_HD bool operator==(const Fp5& o) const {
    return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2] && c[3] == o.c[3] && c[4] == o.c[4];
  }
  // a^(p^K), K = 1, 2: the Frobenius map as a matrix-vector product with compile-time constants.
  template <int K, int J>
  P3R_HD F frobenius_coeff() const {
    const F t = F::dot2(c[1], F::raw(kFrob5<PP, K, 1, J>), c[2], F::raw(kFrob5<PP, K, 2, J>)) +
                F::dot2(c[3], F::raw(kFrob5<PP, K, 3, J>), c[4], F::raw(kFrob5<PP, K, 4, J>));
    return J == 0 ? c[0] + t : t;
  }
  template <int K>
  P3R_HD Fp5 frobenius() const {
    Fp5 r;
    r.c[0] = frobenius_coeff<K, 0>();
    r.c[1] = frobenius_coeff<K, 1>();
    r.c[2] = frobenius_coeff<K, 2>();
    r.c[3] = frobenius_coeff<K, 3>();
    r.c[4] = frobenius_coeff<K, 4>();
    return r;
  }
  // Constant coefficient of a * b (all that is needed of a product known to lie in the base field).
  static P3R_HD F mul_c0(const Fp5& a, const Fp5& b) {
    return a.c[0] * b.c[0] + F::dot2(a.c[1], b.c[4], a.c[2], b.c[3]) + F::dot2(a.c[3], b.c[2], a.c[4], b.c[1]) -
           a.c[4] * b.c[4];
  }
  // Itoh-Tsujii: with r = 1 + p + p^2 + p^3 + p^4, a^r = N(a) lies in the base field and a^-1 = a^(r-1) / N(a).
  // a^(r-1) = a^(p + p^2 + p^3 + p^4) = t * t^(p^2) with t = a^p * a^(p^2): three Frobenius maps, two products.
  // norm_cofactor() returns a^(r-1); norm = mul_c0(a, cofactor).  Zero maps to zero (callers check first).
  P3R_HD Fp5 norm_cofactor() const {
    const Fp5 a1 = frobenius<1>();
    const Fp5 t = a1 * a1.template frobenius<1>();
    return t * t.template frobenius<2>();
  }
  P3R_HD Fp5 inv() const {
    const Fp5 co = norm_cofactor();
    return co * mul_c0(*this, co).inv();
  }
};

// element type of a circuit of extension degree D
template <class PP, int D> struct CircuitExt;
template <class PP> struct CircuitExt<PP, 1> { using type = Fp1<PP>; };
template <class PP> struct CircuitExt<PP, 4> { using type = Fp4<PP>; };
template <class PP> struct CircuitExt<PP, 5> { using type = Fp5<PP>; };

// Embedding of a base-field constant into the value type a generic routine computes in
// (the base field itself for the prover's kernels, the extension for evaluations at zeta).
template <class V> struct Lift;
template <class PP> struct Lift<Fp<PP>> { static P3R_HD Fp<PP> of(Fp<PP> x) { return x; } };
template <class PP> struct Lift<Fp4<PP>> { static P3R_HD Fp4<PP> of(Fp<PP> x) { return Fp4<PP>::from_base(x); } };
template <class PP> struct Lift<Fp5<PP>> { static P3R_HD Fp5<PP> of(Fp<PP> x) { return Fp5<PP>::from_base(x); } };
// The STARK's challenge field: the quartic binomial extension (every BASELINE configuration), or - DC = 5 -
// KoalaBear's quintic trinomial extension (test-utils koala_bear_quintic_params, recursive_fibonacci --quintic).
template <class PP, int DC> struct Chal;
template <class PP> struct Chal<PP, 4> { using type = Fp4<PP>; };
template <class PP> struct Chal<PP, 5> { using type = Fp5<PP>; };

P3R_HD uint32_t bit_reverse(uint32_t x, int bits) {
  if (bits == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x) >> (32 - bits);
#else
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
#endif
}

}  // namespace p3r
