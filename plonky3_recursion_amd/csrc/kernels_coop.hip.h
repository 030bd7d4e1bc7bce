// Lane-cooperative Poseidon2 for LATENCY-bound work (small Merkle layers): one state element per
// lane, 16 lanes (one DPP "row") per permutation, 4 permutations per wavefront.  The one-state-
// per-lane kernels of kernels.hip.h are ~9 k dependent instructions (~25 us) per layer however few
// nodes the layer has; here a permutation is ~1.3 k dependent instructions because the S-boxes of
// a round run in parallel and the linear layers are DPP rotations:
//   external:  M4 is circulant, y_i = 2 x_i + 3 x_{i+1} + x_{i+2} + x_{i+3} inside a quad
//              (quad_perm), then the 4 quads are summed with row_ror:4 / row_ror:8;
//   internal:  sum over the row with row_ror:8,4,2,1, then s_i <- d_i * s_i + sum.
// Same arithmetic as poseidon2.h (Montgomery form), so results are bit-identical.
#pragma once
#include "kernels.hip.h"

namespace p3r {

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
constexpr int DPP_QUAD_NEXT1 = 0x39;  // quad_perm:[1,2,3,0]
constexpr int DPP_QUAD_NEXT2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int DPP_QUAD_NEXT3 = 0x93;  // quad_perm:[3,0,1,2]
constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;

template <class F>
__device__ __forceinline__ F coop_external(F x) {
  F b = F::raw(dpp<DPP_QUAD_NEXT1>(x.v)), c = F::raw(dpp<DPP_QUAD_NEXT2>(x.v)), d = F::raw(dpp<DPP_QUAD_NEXT3>(x.v));
  F ab = x + b;
  F y = ab.dbl() + b + (c + d);  // 2x + 3b + c + d
  F t = y + F::raw(dpp<DPP_ROW_ROR4>(y.v));
  t = t + F::raw(dpp<DPP_ROW_ROR8>(t.v));  // sum of the 4 quads at this quad position
  return y + t;
}
template <class F>
__device__ __forceinline__ F coop_row_sum(F x) {
  F t = x + F::raw(dpp<DPP_ROW_ROR8>(x.v));
  t = t + F::raw(dpp<DPP_ROW_ROR4>(t.v));
  t = t + F::raw(dpp<DPP_ROW_ROR2>(t.v));
  return t + F::raw(dpp<DPP_ROW_ROR1>(t.v));
}

// Value of row-lane 0 on every lane of the row; `x0` must be zero on lanes 1..15, so the three
// combining steps are plain ORs (no modular reduction on this dependent path).
constexpr int DPP_QUAD_BCAST0 = 0x00;  // quad_perm:[0,0,0,0]
__device__ __forceinline__ uint32_t coop_bcast0(uint32_t x0) {
  uint32_t q = dpp<DPP_QUAD_BCAST0>(x0);  // lanes 0..3 (the other quads read their own zero)
  q |= dpp<DPP_ROW_ROR4>(q);
  return q | dpp<DPP_ROW_ROR8>(q);
}

// The round constants one lane of a row needs, loaded ONCE per kernel: its own element of the eight
// full rounds (vector registers) and the partial rounds' single constants (uniform: scalar registers).
// Loaded inside the round loops they were a dependent memory read per round - 28 cache latencies per
// permutation, most of the time of a latency-bound Merkle level.
template <class PP>
struct CoopRc {
  uint32_t full[2 * P2_HALF_FULL];
  uint32_t part[PP::PARTIAL_ROUNDS];
};
template <class PP>
__device__ __forceinline__ CoopRc<PP> coop_load_rc(const uint32_t* __restrict__ rc, int elem) {
  CoopRc<PP> c;
#pragma unroll
  for (int r = 0; r < P2_HALF_FULL; ++r) c.full[r] = rc[r * P2_WIDTH + elem];
#pragma unroll
  for (int r = 0; r < PP::PARTIAL_ROUNDS; ++r) c.part[r] = rc[P2_HALF_FULL * P2_WIDTH + r];
#pragma unroll
  for (int r = 0; r < P2_HALF_FULL; ++r)
    c.full[P2_HALF_FULL + r] = rc[P2_HALF_FULL * P2_WIDTH + PP::PARTIAL_ROUNDS + r * P2_WIDTH + elem];
  return c;
}

// `s`: this lane's state element (lane & 15 = element index).  `diag`: Montgomery internal
// diagonal of this lane.  Round constants as loaded by coop_load_rc (layout of poseidon2.h).
template <class PP>
__device__ __forceinline__ Fp<PP> coop_permute(Fp<PP> s, int elem, Fp<PP> diag, const CoopRc<PP>& rc) {
  using F = Fp<PP>;
  s = coop_external(s);
#pragma unroll
  for (int r = 0; r < P2_HALF_FULL; ++r) {
    s = p2_sbox<PP>(s + F::raw(rc.full[r]));
    s = coop_external(s);
  }
#pragma unroll
  for (int r = 0; r < PP::PARTIAL_ROUNDS; ++r) {
    // sum(s') = S-box output of element 0 + the sum of the other fifteen: the latter and the
    // products d_i * s_i do not wait for the S-box, only its broadcast and two additions do
    const F rest = coop_row_sum(elem == 0 ? F::zero() : s);
    const F sb = p2_sbox<PP>(s + F::raw(rc.part[r]));
    const F sum = rest + F::raw(coop_bcast0(elem == 0 ? sb.v : 0u));
    s = (elem == 0 ? sb : s) * diag + sum;
  }
#pragma unroll
  for (int r = 0; r < P2_HALF_FULL; ++r) {
    s = p2_sbox<PP>(s + F::raw(rc.full[P2_HALF_FULL + r]));
    s = coop_external(s);
  }
  return s;
}

// Leaf hashing of a SMALL strided matrix (FRI commit-phase leaves of the later phases), sixteen
// lanes per row: a row is only 1-4 permutations, so with one row per lane the launch is one
// permutation latency of the 7.8 k-instruction kind (25 us) however few rows there are; the
// lane-cooperative permutation brings it to a few microseconds.  Same overwrite-mode sponge as
// k_mmcs_hash_rows_strided (kernels_stark.hip.h).
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_mmcs_hash_rows_strided_coop(const uint32_t* const* __restrict__ cols, int wtot, size_t h, size_t stride,
                              uint32_t* __restrict__ dig, const uint32_t* __restrict__ rc,
                              const uint32_t* __restrict__ diag) {
  using F = Fp<PP>;
  const size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t row = gid >> 4;
  const int e = (int)(gid & 15);
  const bool live = row < h;  // whole 16-lane rows are live or not
  const F d = F::raw(diag[e]);
  const CoopRc<PP> rcs = coop_load_rc<PP>(rc, e);
  F s = F::zero();
  for (int g = 0; g < wtot; g += P2_RATE) {
    if (live && e < P2_RATE && g + e < wtot) s = F::raw(cols[g + e][row * stride]);
    s = coop_permute<PP>(s, e, d, rcs);
  }
  if (live && e < P2_DIGEST) dig[(size_t)e * h + row] = s.v;
}

// Several levels of a Merkle tree per launch.  A workgroup owns `local` consecutive digests of the
// input layer (32 by default, at most kSubtreeNodes; p3r_core.hip::subtree_nodes says why) and
// everything above them: a barrier per level instead of a launch
// (each of these levels is one permutation latency; the launches between them cost more than the
// work), LDS hand-off between levels.  Every level is also written to its own layer buffer -
// queries read siblings from them.  A level may carry an injection (digests of the shorter
// matrices of the commit, circuit/src/ops/mmcs.rs:117-160): node = compress(compress(l, r), inj).
// Used for layers of at most coop_max_nodes() nodes; the last launch of a tree is a single workgroup.
constexpr int kSubtreeBlock = 1024;  // the largest workgroup: 64 nodes in flight
constexpr int kSubtreeNodes = 256;
constexpr int kSubtreeLevels = 8;
struct SubtreeArgs {
  const uint32_t* in;  // [8][n_in]
  uint32_t n_in;       // power of two; a multiple of `local`, or smaller (one workgroup)
  uint32_t local;      // digests of the input layer per workgroup (power of two <= kSubtreeNodes);
                       // launched with 8 * local lanes (at least a wave): one 16-lane row per level-0 node
  int n_levels;        // <= kSubtreeLevels, <= log2(min(n_in, local))
  uint32_t* out[kSubtreeLevels];        // out[l]: [8][n_in >> (l+1)]
  const uint32_t* inj[kSubtreeLevels];  // inj[l]: [8][n_in >> (l+1)] or null
  // Optional (FRI commit phase, single-workgroup launch that ends at the root): the transcript step
  // of fri_transcript_step below runs right behind the root instead of in a launch of its own.
  uint32_t* t_state;  // [16] sponge state, or null
  uint32_t* t_beta;   // [DC]
  uint32_t* t_cap;    // [8]
  int t_dc;           // words of the folding challenge (the challenge degree); 0 = 4
};

// One step of the FRI commit-phase transcript on the device (DuplexChallenger<F, Perm, 16, 8>,
// recursion/src/challenger/circuit.rs:97-156,337-386) for cap_height 0 and no commit-phase proof
// of work: observe the 8 words of the phase's Merkle root (a full rate block: overwrite
// state[0..8], state[8] += 8, permute), then sample the folding challenge (an extension element
// pops state[7], [6], [5], [4](, [3] over the quintic challenge field)).  Keeps the commit phase free of host round trips; the host
// replays the same steps on its own transcript afterwards from `cap_out`.
// Called by the 16 lanes of one DPP row; `root_word` = lane j's word of the root (j < 8).
template <class PP>
__device__ __forceinline__ void fri_transcript_step(int j, uint32_t root_word, uint32_t* __restrict__ state,
                                                    uint32_t* __restrict__ beta_out, uint32_t* __restrict__ cap_out,
                                                    const CoopRc<PP>& rc, const uint32_t* __restrict__ diag, int dc = 4) {
  using F = Fp<PP>;
  F s = j < P2_RATE ? F::raw(root_word) : F::raw(state[j]);
  if (j < P2_RATE) cap_out[j] = s.v;
  if (j == P2_RATE) s += F::from_canonical(P2_RATE);
  s = coop_permute<PP>(s, j, F::raw(diag[j]), rc);
  state[j] = s.v;
  if (j >= P2_RATE - dc && j < P2_RATE) beta_out[7 - j] = s.v;
}
template <class PP>
__global__ void __launch_bounds__(kSubtreeBlock)
k_mmcs_subtree(SubtreeArgs a, const uint32_t* __restrict__ rc, const uint32_t* __restrict__ diag) {
  using F = Fp<PP>;
  __shared__ uint32_t buf[2][P2_DIGEST * kSubtreeNodes / 2];
  const int elem = threadIdx.x & 15;
  const uint32_t group = threadIdx.x >> 4, k = elem & 7;
  const F d = F::raw(diag[elem]);
  const CoopRc<PP> rcs = coop_load_rc<PP>(rc, elem);
  uint32_t n = a.n_in < a.local ? a.n_in : a.local;  // local nodes
  uint32_t n_glob = a.n_in, first = blockIdx.x * n;  // layer size, this workgroup's first node
#pragma unroll
  for (int l = 0; l < kSubtreeLevels; ++l) {
    if (l < a.n_levels) {
      const uint32_t nn = n / 2, nn_glob = n_glob / 2, first_out = first / 2;
      const uint32_t* cur = buf[(l + 1) & 1];
      uint32_t* nxt = buf[l & 1];
      // a 16-lane row works on one node: lanes 0..7 hold the left child, 8..15 the right one
      for (uint32_t node = group; node < nn; node += blockDim.x / 16) {
        const uint32_t child = 2 * node + (elem >> 3);
        F s = F::raw(l == 0 ? a.in[(size_t)k * n_glob + first + child] : cur[k * n + child]);
        s = coop_permute<PP>(s, elem, d, rcs);
        if (a.inj[l]) {
          if (elem >= P2_DIGEST) s = F::raw(a.inj[l][(size_t)k * nn_glob + first_out + node]);
          s = coop_permute<PP>(s, elem, d, rcs);
        }
        if (elem < P2_DIGEST) {
          nxt[k * nn + node] = s.v;
          a.out[l][(size_t)k * nn_glob + first_out + node] = s.v;
        }
      }
      __syncthreads();
      n = nn;
      n_glob = nn_glob;
      first = first_out;
    }
  }
  if (a.t_state && threadIdx.x < P2_WIDTH) {
    // the root is the single node of the last level: word k at buf[(n_levels - 1) & 1][k]
    const uint32_t* root = buf[(a.n_levels - 1) & 1];
    fri_transcript_step<PP>((int)threadIdx.x, threadIdx.x < P2_RATE ? root[threadIdx.x] : 0u, a.t_state, a.t_beta,
                            a.t_cap, rcs, diag, a.t_dc ? a.t_dc : 4);
  }
}

// The same step as a launch of its own, for a phase whose tree has no level above its single leaf
// (the root is the leaf digest, no k_mmcs_subtree launch to ride on).
template <class PP>
__global__ void __launch_bounds__(64)
k_fri_transcript_step(const uint32_t* __restrict__ root /* [8] */, uint32_t* __restrict__ state /* [16] */,
                      uint32_t* __restrict__ beta_out /* [DC] */, uint32_t* __restrict__ cap_out /* [8] */,
                      const uint32_t* __restrict__ rc, const uint32_t* __restrict__ diag, int dc) {
  const int j = threadIdx.x;
  if (j >= P2_WIDTH) return;  // one 16-lane row
  fri_transcript_step<PP>(j, j < P2_RATE ? root[j] : 0u, state, beta_out, cap_out, coop_load_rc<PP>(rc, j), diag, dc);
}

// ---- lane-cooperative width-32 permutation for LATENCY-bound levels (coop_permute above is the width-16 form): one
// state element per lane, 32 lanes (two DPP rows) per permutation, two permutations per wavefront.  With one
// permutation per lane a level of an arity-4 tree costs one 7.8 - 11.3 k-instruction permutation (13 - 19 us) however few nodes
// it has; here the 32 S-boxes of a round run side by side and the linear layers are DPP rotations plus ONE cross-row
// step (v_permlane16_swap, new on gfx950: it exchanges the odd rows of one register with the even rows of another,
// so swap(t, t) yields (row0, row0, row2, row2) and (row1, row1, row3, row3), whose sum is the pair sum in every row).


template <class F>
__device__ __forceinline__ F coop32_pair_sum(F t) {
  const auto r = __builtin_amdgcn_permlane16_swap(t.v, t.v, false, false);
  return F::raw(r[0]) + F::raw(r[1]);
}
// external layer circ(2 M4, M4, .., M4) over eight quads: M4 inside the quad, then the sum of the eight quads
template <class F>
__device__ __forceinline__ F coop32_external(F x) {
  const F b = F::raw(dpp<DPP_QUAD_NEXT1>(x.v)), c = F::raw(dpp<DPP_QUAD_NEXT2>(x.v)), d = F::raw(dpp<DPP_QUAD_NEXT3>(x.v));
  const F ab = x + b;
  const F y = ab.dbl() + b + (c + d);  // 2x + 3b + c + d
  F t = y + F::raw(dpp<DPP_ROW_ROR4>(y.v));
  t = t + F::raw(dpp<DPP_ROW_ROR8>(t.v));
  return y + coop32_pair_sum(t);
}
template <class F>
__device__ __forceinline__ F coop32_sum(F x) { return coop32_pair_sum(coop_row_sum(x)); }
// the value of lane 0 of the group on all 32 lanes; `x0` is zero on the other 31
__device__ __forceinline__ uint32_t coop32_bcast0(uint32_t x0) {
  const uint32_t q = coop_bcast0(x0);
  const auto r = __builtin_amdgcn_permlane16_swap(q, q, false, false);
  return r[0] | r[1];
}

template <class PP>
struct Coop32Rc {
  uint32_t full[2 * P2_HALF_FULL];
  uint32_t part[PP::PARTIAL_ROUNDS_W32];
  uint32_t diag;
};
// `rcw`: the width-32 constant table in Montgomery form (p3r_ctx::rc + p2_num_constants: round constants | diagonal)
template <class PP>
__device__ __forceinline__ Coop32Rc<PP> coop32_load_rc(const uint32_t* __restrict__ rcw, int elem) {
  Coop32Rc<PP> c;
#pragma unroll
  for (int r = 0; r < P2_HALF_FULL; ++r) c.full[r] = rcw[r * P2W_WIDTH + elem];
#pragma unroll
  for (int r = 0; r < PP::PARTIAL_ROUNDS_W32; ++r) c.part[r] = rcw[P2_HALF_FULL * P2W_WIDTH + r];
#pragma unroll
  for (int r = 0; r < P2_HALF_FULL; ++r)
    c.full[P2_HALF_FULL + r] = rcw[P2_HALF_FULL * P2W_WIDTH + PP::PARTIAL_ROUNDS_W32 + r * P2W_WIDTH + elem];
  c.diag = rcw[p2w_num_rc<PP>() + elem];
  return c;
}
template <class PP>
__device__ __forceinline__ Fp<PP> coop32_permute(Fp<PP> s, int elem, const Coop32Rc<PP>& rc) {
  using F = Fp<PP>;
  const F diag = F::raw(rc.diag);
  s = coop32_external(s);
#pragma unroll
  for (int r = 0; r < P2_HALF_FULL; ++r) {
    s = p2_sbox<PP>(s + F::raw(rc.full[r]));
    s = coop32_external(s);
  }
#pragma unroll 1
  for (int r = 0; r < PP::PARTIAL_ROUNDS_W32; ++r) {
    const F rest = coop32_sum(elem == 0 ? F::zero() : s);
    const F sb = p2_sbox<PP>(s + F::raw(rc.part[r]));
    const F sum = rest + F::raw(coop32_bcast0(elem == 0 ? sb.v : 0u));
    s = (elem == 0 ? sb : s) * diag + sum;
  }
#pragma unroll
  for (int r = 0; r < P2_HALF_FULL; ++r) {
    s = p2_sbox<PP>(s + F::raw(rc.full[P2_HALF_FULL + r]));
    s = coop32_external(s);
  }
  return s;
}

}  // namespace p3r
