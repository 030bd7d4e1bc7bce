// FRI reduced openings with the quotient denominators hoisted out of the per-matrix pass:
//   inv_{z}[r] = 1 / (z - x_r),  x_r = gen * w_h^{bitrev(r)}          (k_fri_inv_points)
//   ro[r] += sum_p off_p * (V_p - sum_c alpha^c M[c][r]) * inv_{z_p}[r]   (k_fri_reduce_pre)
// One inverse vector per distinct (height, point) serves every matrix committed at that height
// (recursion/src/pcs/fri/verifier.rs:1122-1345 caches the same quantity per (height, z)).
#pragma once
#include "kernels_stark.cuh"

namespace p3r {

template <class PP>
__global__ void __launch_bounds__(kBlock)
k_fri_inv_points(size_t h, int log_h, uint32_t gen, uint32_t w_h, E4 z, uint32_t* __restrict__ inv /* [4][h] */) {
  using F = Fp<PP>;
  using E = Fp4<PP>;
  size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= h) return;
  F x = F::raw(gen) * F::raw(w_h).pow(bit_reverse((uint32_t)r, log_h));
  E v = (e4_load<PP>(z) - E::from_base(x)).inv();
#pragma unroll
  for (int k = 0; k < 4; ++k) inv[(size_t)k * h + r] = v.c[k].v;
}

struct FriReducePreArgs {
  const uint32_t* mat;  // bit-reversed LDE [w][h]
  size_t h;
  int w;
  const uint32_t* apow;  // alpha^c, 4 words each
  int n_points;
  const uint32_t* inv[2];  // [4][h] each
  E4 v[2], off[2];
  uint32_t* ro;  // [4][h], accumulated in place
};
template <class PP>
__global__ void __launch_bounds__(kBlock) k_fri_reduce_pre(FriReducePreArgs a) {
  using F = Fp<PP>;
  using E = Fp4<PP>;
  size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= a.h) return;
  E S = E::zero();
  auto apow = [&](int c) {
    E ap;
#pragma unroll
    for (int k = 0; k < 4; ++k) ap.c[k] = F::raw(a.apow[4 * c + k]);
    return ap;
  };
  int c = 0;
  for (; c + 1 < a.w; c += 2)  // two columns per reduction
    S += E::dot2_base(apow(c), F::raw(a.mat[(size_t)c * a.h + r]), apow(c + 1), F::raw(a.mat[(size_t)(c + 1) * a.h + r]));
  if (c < a.w) S += apow(c) * F::raw(a.mat[(size_t)c * a.h + r]);
  E acc;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc.c[k] = F::raw(a.ro[(size_t)k * a.h + r]);
  for (int p = 0; p < a.n_points; ++p) {
    E inv;
#pragma unroll
    for (int k = 0; k < 4; ++k) inv.c[k] = F::raw(a.inv[p][(size_t)k * a.h + r]);
    acc += e4_load<PP>(a.off[p]) * (e4_load<PP>(a.v[p]) - S) * inv;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) a.ro[(size_t)k * a.h + r] = acc.c[k].v;
}

}  // namespace p3r
