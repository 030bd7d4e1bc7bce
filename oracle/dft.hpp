// ORACLE (test infrastructure): radix-2 DFT / coset LDE in canonical arithmetic.
// PARITY UNPINNED (see field.hpp).
//
// Restates what TwoAdicFriPcs::commit does with Radix2DitParallel
// (circuit-prover/src/config.rs:55,131): interpolate each column over the size-h subgroup,
// evaluate over shift * <w_{h*2^added_bits}>, and store the rows bit-reversed.  The row
// order is pinned by the query-point formula x = GENERATOR * g^{rev(index)}
// (recursion/src/pcs/fri/verifier.rs:921-981).
//
// Written as the textbook in-place decimation-in-time butterfly network (bit-reversal
// permutation first), i.e. a different factorisation from the device's LDS-tiled four-step
// passes, so agreement is a real check.
#pragma once
#include "hash.hpp"

namespace orc {

// In-place DFT of a (natural order in, natural order out): a[k] <- sum_n a[n] root^{nk}.
template <class FP>
void dft_inplace(std::vector<Fe<FP>>& a, Fe<FP> root) {
  using F = Fe<FP>;
  const size_t n = a.size();
  const int ln = log2_strict(n);
  for (size_t i = 0; i < n; ++i) {
    size_t j = bitrev((uint32_t)i, ln);
    if (i < j) std::swap(a[i], a[j]);
  }
  for (int s = 1; s <= ln; ++s) {
    size_t m = size_t(1) << s;
    F wm = root.pow(n >> s);
    for (size_t k = 0; k < n; k += m) {
      F w = F::one();
      for (size_t j = 0; j < m / 2; ++j) {
        F t = w * a[k + j + m / 2];
        F u = a[k + j];
        a[k + j] = u + t;
        a[k + j + m / 2] = u - t;
        w *= wm;
      }
    }
  }
}

template <class FP>
std::vector<Fe<FP>> idft(std::vector<Fe<FP>> evals) {
  using F = Fe<FP>;
  const size_t n = evals.size();
  dft_inplace<FP>(evals, F::two_adic_generator(log2_strict(n)).inv());
  F ninv = F((uint64_t)n).inv();
  for (auto& x : evals) x *= ninv;
  return evals;
}

// Evaluate coefficients on shift * <w_m>, m = coeffs.size() << added_bits, natural order.
template <class FP>
std::vector<Fe<FP>> coset_dft(const std::vector<Fe<FP>>& coeffs, int added_bits, Fe<FP> shift) {
  using F = Fe<FP>;
  const size_t m = coeffs.size() << added_bits;
  std::vector<F> a(m);
  F s = F::one();
  for (size_t i = 0; i < coeffs.size(); ++i) {
    a[i] = coeffs[i] * s;
    s *= shift;
  }
  dft_inplace<FP>(a, F::two_adic_generator(log2_strict(m)));
  return a;
}

// coset_lde_batch(mat, added_bits, shift).bit_reverse_rows()
template <class FP>
Matrix<FP> coset_lde_bitrev(const Matrix<FP>& evals, int added_bits, Fe<FP> shift) {
  using F = Fe<FP>;
  const size_t h = evals.h, m = h << added_bits;
  const int lm = log2_strict(m);
  Matrix<FP> out(m, evals.w);
#pragma omp parallel for schedule(dynamic) if (h >= 1024)
  for (size_t c = 0; c < evals.w; ++c) {
    std::vector<F> col(h);
    for (size_t r = 0; r < h; ++r) col[r] = evals.at(r, c);
    auto e = coset_dft<FP>(idft<FP>(col), added_bits, shift);
    for (size_t i = 0; i < m; ++i) out.at(i, c) = e[bitrev((uint32_t)i, lm)];
  }
  return out;
}

}  // namespace orc
