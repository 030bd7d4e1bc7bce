// K6 for the arity-4 MMCS: leaf hashing and tree levels over the width-32 permutation (poseidon2_w32_f64.hip.h).
// Layout as the binary tree's (kernels.hip.h): digests SoA [8][n], matrices column-major.
#pragma once
#include "kernels.hip.h"
#include "poseidon2_w32_f64.hip.h"

namespace p3r {

constexpr int P2W_RATE = 24;   // PaddingFreeSponge<Perm32, 32, 24, 8>

// One absorb loop for both leaf kernels: cell (g, i) of the concatenated row is load(g).
template <class PP, bool BUILTIN, class Load>
__device__ __forceinline__ void p2wf_sponge(double* s, int wtot, const double* __restrict__ tab, Load&& load) {
#pragma unroll
  for (int k = 0; k < P2W_WIDTH; ++k) s[k] = 0.0;
  int g = 0;
  for (; g + P2W_RATE <= wtot; g += P2W_RATE) {
#pragma unroll
    for (int j = 0; j < P2W_RATE; ++j) s[j] = load(g + j);
    p2wf_permute<PP, BUILTIN, 0xFF000000u>(s, tab);   // the capacity lanes are carried
  }
  const int rem = wtot - g;
  if (rem > 0) {
#pragma unroll
    for (int j = 0; j < P2W_RATE; ++j)
      if (j < rem) s[j] = load(g + j);
    p2wf_permute<PP, BUILTIN, 0xFFFFFFFFu>(s, tab);   // and the rate lanes past `rem`
  }
}

struct HashRowsJob4 {
  const uint32_t* const* cols;
  uint32_t* dig;     // [8][h_alloc]
  uint64_t h;
  uint64_t h_alloc;  // width of the digest layer (a layer of 2 is padded to 4)
  int wtot;
  uint32_t block0;   // first block of this job
};
// Leaf digests of several height classes in one launch (the job list of k_mmcs_hash_rows): overwrite-mode sponge of
// rate 24 over the concatenated row (recursion/src/pcs/mmcs.rs:963-985 add_arity4_leaf_digest_from_base).
// BUILTIN: the instance for the built-in diagonal (poseidon2_w32_f64.hip.h).
template <class PP, bool BUILTIN>
__global__ void __launch_bounds__(kBlock)
k_mmcs4_hash_rows(const HashRowsJob4* __restrict__ jobs, int n_jobs, const double* __restrict__ tab) {
  const int jb = find_job(jobs, n_jobs);
  const gptr<const uint32_t* const> cols = as_global(jobs[jb].cols);
  const gptr<uint32_t> dig = as_global(jobs[jb].dig);
  const size_t h = jobs[jb].h, stride = jobs[jb].h_alloc;
  const size_t i = (size_t)(blockIdx.x - jobs[jb].block0) * kBlock + threadIdx.x;
  if (i >= h) return;
  double s[P2W_WIDTH];
  p2wf_sponge<PP, BUILTIN>(s, jobs[jb].wtot, tab, [&](int g) { return p2f_load<PP>(as_global(cols[g])[i]); });
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) dig[(size_t)k * stride + i] = p2f_store<PP>(s[k]);
}

// FRI commit-phase leaves: strided views over the folded vector (k_mmcs_hash_rows_strided).  dig: [8][h_alloc].
template <class PP, bool BUILTIN>
__global__ void __launch_bounds__(kBlock)
k_mmcs4_hash_rows_strided(const uint32_t* const* __restrict__ cols, int wtot, size_t h, size_t stride, uint32_t* __restrict__ dig,
                          size_t h_alloc, const double* __restrict__ tab) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= h) return;
  double s[P2W_WIDTH];
  p2wf_sponge<PP, BUILTIN>(s, wtot, tab, [&](int g) { return p2f_load<PP>(cols[g][i * stride]); });
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) dig[(size_t)k * h_alloc + i] = p2f_store<PP>(s[k]);
}

// One level: node i of the new layer = perm(c_0 || .. || c_3)[0..8] over its `step` children prev[step i ..], the
// chunks above `step` zero (a bridge level; recursion/src/pcs/mmcs.rs:1023-1075); with `inj`, the row digests of
// the matrices of the new layer's height enter as one more compression (node, inj[i], 0, 0) in the same lane.
// Nodes n_logical .. n_out of the new layer are the zero digests that pad a layer of 2 to 4.
// prev: [8][n_prev], inj: [8][n_logical] or null, out: [8][n_out].
template <class PP, bool BUILTIN>
__global__ void __launch_bounds__(kBlock)
k_mmcs4_compress(const uint32_t* __restrict__ prev, size_t n_prev, int step, const uint32_t* __restrict__ inj,
                 uint32_t* __restrict__ out, size_t n_logical, size_t n_out, const double* __restrict__ tab) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_out) return;
  if (i >= n_logical) {
#pragma unroll
    for (int k = 0; k < P2_DIGEST; ++k) out[(size_t)k * n_out + i] = 0;
    return;
  }
  double s[P2W_WIDTH];
  if (step == 4) {
#pragma unroll
    for (int k = 0; k < P2_DIGEST; ++k) {
      const uint4 c = *reinterpret_cast<const uint4*>(prev + (size_t)k * n_prev + 4 * i);   // the four children are adjacent
      s[k] = p2f_load<PP>(c.x);
      s[P2_DIGEST + k] = p2f_load<PP>(c.y);
      s[2 * P2_DIGEST + k] = p2f_load<PP>(c.z);
      s[3 * P2_DIGEST + k] = p2f_load<PP>(c.w);
    }
  } else {
#pragma unroll
    for (int k = 0; k < P2_DIGEST; ++k) {
      const uint2 c = *reinterpret_cast<const uint2*>(prev + (size_t)k * n_prev + 2 * i);
      s[k] = p2f_load<PP>(c.x);
      s[P2_DIGEST + k] = p2f_load<PP>(c.y);
      s[2 * P2_DIGEST + k] = 0.0;
      s[3 * P2_DIGEST + k] = 0.0;
    }
  }
  p2wf_permute<PP, BUILTIN, 0u>(s, tab);
  if (inj) {
#pragma unroll
    for (int k = 0; k < P2_DIGEST; ++k) {
      s[P2_DIGEST + k] = p2f_load<PP>(inj[(size_t)k * n_logical + i]);
      s[2 * P2_DIGEST + k] = 0.0;
      s[3 * P2_DIGEST + k] = 0.0;
    }
    p2wf_permute<PP, BUILTIN, 0x000000FFu>(s, tab);   // the node's own digest is carried
  }
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) out[(size_t)k * n_out + i] = p2f_store<PP>(s[k]);
}

}  // namespace p3r

// (the lane-cooperative width-32 permutation coop32_permute lives in kernels_coop.hip.h, next to the width-16 form)
#include "kernels_coop.hip.h"

namespace p3r {

// One level, 32 lanes per node (k_mmcs4_compress is the one-node-per-lane form): lane e holds word e & 7 of child
// e >> 3.  For the levels small enough to be latency-bound.
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_mmcs4_compress_coop(const uint32_t* __restrict__ prev, size_t n_prev, int step, const uint32_t* __restrict__ inj,
                      uint32_t* __restrict__ out, size_t n_logical, size_t n_out, const uint32_t* __restrict__ rcw) {
  using F = Fp<PP>;
  const size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x, node = gid >> 5;
  const int e = (int)(gid & 31), chunk = e >> 3, k = e & 7;
  if (node >= n_out) return;   // whole 32-lane groups
  if (node >= n_logical) {
    if (e < P2_DIGEST) out[(size_t)k * n_out + node] = 0;
    return;
  }
  const Coop32Rc<PP> rc = coop32_load_rc<PP>(rcw, e);
  F s = chunk < step ? F::raw(prev[(size_t)k * n_prev + (size_t)step * node + chunk]) : F::zero();
  s = coop32_permute<PP>(s, e, rc);
  if (inj) {
    if (e >= P2_DIGEST) s = e < 2 * P2_DIGEST ? F::raw(inj[(size_t)k * n_logical + node]) : F::zero();
    s = coop32_permute<PP>(s, e, rc);
  }
  if (e < P2_DIGEST) out[(size_t)k * n_out + node] = s.v;
}

// FRI commit-phase leaves of the later phases, 32 lanes per row (k_mmcs4_hash_rows_strided is one row per lane)
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_mmcs4_hash_rows_strided_coop(const uint32_t* const* __restrict__ cols, int wtot, size_t h, size_t stride,
                               uint32_t* __restrict__ dig, size_t h_alloc, const uint32_t* __restrict__ rcw) {
  using F = Fp<PP>;
  const size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x, row = gid >> 5;
  const int e = (int)(gid & 31);
  if (row >= h) return;   // whole 32-lane groups
  const Coop32Rc<PP> rc = coop32_load_rc<PP>(rcw, e);
  F s = F::zero();
  for (int g = 0; g < wtot; g += P2W_RATE) {
    if (e < P2W_RATE && g + e < wtot) s = F::raw(cols[g + e][row * stride]);
    s = coop32_permute<PP>(s, e, rc);
  }
  if (e < P2_DIGEST) dig[(size_t)e * h_alloc + row] = s.v;
}

}  // namespace p3r
