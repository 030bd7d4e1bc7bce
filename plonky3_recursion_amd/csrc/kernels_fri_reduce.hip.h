// FRI reduced openings with the quotient denominators hoisted out of the per-matrix pass:
//   inv_{z}[r] = 1 / (z - x_r),  x_r = gen * w_h^{bitrev(r)}          (k_fri_inv_points)
//   ro[r] += sum_p off_p * (V_p - sum_c alpha^c M[c][r]) * inv_{z_p}[r]   (k_fri_reduce_pre)
// One inverse vector per distinct (height, point) serves every matrix committed at that height
// (recursion/src/pcs/fri/verifier.rs:1122-1345 caches the same quantity per (height, z)).
#pragma once
#include "kernels_stark.hip.h"

namespace p3r {

// Both kernels run once per proof over job lists (one job per distinct (height, point), resp. per
// height); a block finds its job by walking the list of first-block indices.
template <int DC>
struct FriInvJobT {
  uint32_t* inv;  // [DC][h]
  uint64_t h;
  int log_h;
  uint32_t w_h;
  uint32_t w_4;     // primitive 4th root of unity w_h^(h/4) (h >= 4)
  EW<DC> z;
  uint32_t block0;  // first block; a lane owns four consecutive rows
};
using FriInvJob = FriInvJobT<4>;
template <class PP, int DC = 4>
__global__ void __launch_bounds__(kBlock)
k_fri_inv_points(const FriInvJobT<DC>* __restrict__ jobs, int n_jobs, uint32_t gen) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  const int j = find_job(jobs, n_jobs);
  const FriInvJobT<DC>& job = jobs[j];
  // four consecutive rows per lane: their inversions share one base-field inversion (inv4)
  const size_t h = job.h, r0 = ((size_t)(blockIdx.x - job.block0) * kBlock + threadIdx.x) * 4;
  if (r0 >= h) return;
  const E z = e4_load<PP, DC>(job.z);
  E x[4], v[4];
  // rows r0..r0+3 differ in their two low bits, i.e. in the two HIGH bits of the bit-reversed
  // exponent: x, -x, ix, -ix with i = w_h^(h/4)   (h >= 4, r0 a multiple of 4)
  const F x0 = F::raw(gen) * F::raw(job.w_h).pow(bit_reverse((uint32_t)r0, job.log_h));
  const F x2 = x0 * F::raw(job.w_4);
  x[0] = z - E::from_base(x0);
  x[1] = z + E::from_base(x0);
  x[2] = z - E::from_base(x2);
  x[3] = z + E::from_base(x2);
  inv4<PP>(x, v);
  const gptr<uint32_t> inv = as_global(job.inv);
#pragma unroll
  for (int m = 0; m < 4; ++m)
    if (r0 + m < h)
#pragma unroll
      for (int k = 0; k < DC; ++k) inv[(size_t)k * h + r0 + m] = v[m].c[k].v;
}

// V = sum_c alpha^c * value_c over the opened values of one (matrix, point): one workgroup per
// job, straight from the device copy of the opened values (the host never forms these sums).
struct FriVsumJob {
  const uint32_t* vals;  // [w][4]
  uint32_t* out;         // [4]
  int w;
};
template <class PP, int DC = 4>
__global__ void __launch_bounds__(kBlock)
k_fri_vsum(const FriVsumJob* __restrict__ jobs, const uint32_t* __restrict__ apow_tab /* alpha^c, DC words each */) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  __shared__ uint32_t sh[kBlock / 64][DC];
  const FriVsumJob job = jobs[blockIdx.x];
  const gptr<const uint32_t> vals = as_global(job.vals);
  E acc = E::zero();
  for (int c = threadIdx.x; c < job.w; c += kBlock) {
    E ap, v;
#pragma unroll
    for (int k = 0; k < DC; ++k) {
      ap.c[k] = F::raw(apow_tab[DC * c + k]);
      v.c[k] = F::raw(vals[DC * c + k]);
    }
    acc += ap * v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < DC; ++k) {
    F x = acc.c[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += F::raw(__shfl_down(x.v, off));
    if (lane == 0) sh[wave][k] = x.v;
  }
  __syncthreads();
  if (threadIdx.x < DC) {
    F x = F::zero();
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; ++wv) x += F::raw(sh[wv][threadIdx.x]);
    as_global(job.out)[threadIdx.x] = x.v;
  }
}

// One committed matrix and its opening points.
template <int DC>
struct FriReduceMatT {
  const uint32_t* mat;  // bit-reversed LDE [w][h]
  int w, n_points;
  const uint32_t* inv[2];  // [DC][h] each
  const uint32_t* v[2];    // [DC]: the k_fri_vsum result of this matrix and point
  EW<DC> off[2];
};
using FriReduceMat = FriReduceMatT<4>;
// All matrices of one height: lane r owns ro[r] and adds every matrix's term to it, so ro is
// written once and needs no zero fill.
struct FriReduceJob {
  uint32_t* ro;  // [4][h]
  uint64_t h;
  uint32_t mat0, n_mats;  // range in the matrix list
  uint32_t block0;
};
template <class PP, int DC = 4>
__global__ void __launch_bounds__(kBlock)
k_fri_reduce_pre(const FriReduceJob* __restrict__ jobs, int n_jobs, const FriReduceMatT<DC>* __restrict__ mats,
                 const uint32_t* __restrict__ apow_tab /* alpha^c, DC words each */) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  const int j = find_job(jobs, n_jobs);
  const FriReduceJob job = jobs[j];
  const size_t h = job.h, r = (size_t)(blockIdx.x - job.block0) * kBlock + threadIdx.x;
  if (r >= h) return;
  auto apow = [&](int c) {
    E ap;
#pragma unroll
    for (int k = 0; k < DC; ++k) ap.c[k] = F::raw(apow_tab[DC * c + k]);
    return ap;
  };
  E acc = E::zero();
  for (uint32_t m = 0; m < job.n_mats; ++m) {
    const FriReduceMatT<DC>& a = mats[job.mat0 + m];
    const gptr<const uint32_t> mat = as_global(a.mat);
    const int w = a.w;
    E S = E::zero(), S2 = E::zero();
    int c = 0;
    // four column loads in flight, two columns per reduction, two accumulators: the pass is bound by HBM latency x
    // occupancy, not by issue (1.22 -> 1.10 ms at 2^20 rows; eight in flight: 1.14 ms)
    for (; c + 3 < w; c += 4) {
      const F m0 = F::raw(mat[(size_t)c * h + r]), m1 = F::raw(mat[(size_t)(c + 1) * h + r]);
      const F m2 = F::raw(mat[(size_t)(c + 2) * h + r]), m3 = F::raw(mat[(size_t)(c + 3) * h + r]);
      S += E::dot2_base(apow(c), m0, apow(c + 1), m1);
      S2 += E::dot2_base(apow(c + 2), m2, apow(c + 3), m3);
    }
    for (; c + 1 < w; c += 2)
      S += E::dot2_base(apow(c), F::raw(mat[(size_t)c * h + r]), apow(c + 1), F::raw(mat[(size_t)(c + 1) * h + r]));
    if (c < w) S += apow(c) * F::raw(mat[(size_t)c * h + r]);
    S += S2;
    for (int p = 0; p < a.n_points; ++p) {
      E inv;
#pragma unroll
      for (int k = 0; k < DC; ++k) inv.c[k] = F::raw(as_global(a.inv[p])[(size_t)k * h + r]);
      E V;
#pragma unroll
      for (int k = 0; k < DC; ++k) V.c[k] = F::raw(as_global(a.v[p])[k]);
      acc += e4_load<PP, DC>(a.off[p]) * (V - S) * inv;
    }
  }
#pragma unroll
  for (int k = 0; k < DC; ++k) as_global(job.ro)[(size_t)k * h + r] = acc.c[k].v;
}

}  // namespace p3r
