// gfx950 kernels of the ZK (HidingFriPcs) configuration, p3r_config.zk: HBM-bound fills and interleaves.
//   k_zk_randomize   HidingFriPcs::commit on one matrix: h x w -> 2h x (w + R), original rows at the even positions
//   k_zk_mask_last   the dependent quotient mask t_{C-1} = -(1 / k_{C-1}) sum_c k_c t_c over the common coset
//   k_zk_chunk       one randomised quotient chunk: even rows q_c, odd rows q_c - 2 t_c on the odd coset
// What the verifier enforces of them: recursion/src/verifier/batch_stark.rs:629-661 (extended trace domain),
// :701-735 (quotient chunk domains, randomised opening domains), verifier/quotient.rs (chunk recomposition).
#pragma once
#include "kernels.hip.h"
#include "zk_rand.h"

namespace p3r {

// One job = one matrix; the jobs of a commit are one launch.  Cells are column-major: src [w][h], dst [w + R][2h].
// mode 0: random fill; 1: zero fill (the preprocessed round).  src == nullptr (w = 0): every cell is random (the
// random round).  Random cell (r, c) of the 2h x (w + R) matrix is cell r * (w + R) + c of the job's stream.
struct ZkRandomizeJob {
  const uint32_t* src;
  uint32_t* dst;
  uint64_t h2;          // 2h
  uint32_t w, w2;       // original / randomised width
  uint32_t zero_fill;
  uint32_t stream;      // zk_stream_id(round, matrix): which stream of the proof's generator (zk_rand.h)
  uint32_t block0;      // first block of this job: blocks cover (column, 256 rows) tiles, rows fastest
};
template <class PP>
__global__ void __launch_bounds__(kBlock) k_zk_randomize(const ZkRandomizeJob* __restrict__ jobs, int n_jobs, ZkKey key) {
  int jb = 0;
  while (jb + 1 < n_jobs && blockIdx.x >= jobs[jb + 1].block0) ++jb;
  const ZkRandomizeJob& j = jobs[jb];
  const uint64_t tiles_per_col = (j.h2 + kBlock - 1) / kBlock;
  const uint64_t t = blockIdx.x - j.block0;
  const uint32_t c = (uint32_t)(t / tiles_per_col);
  const uint64_t r = (t % tiles_per_col) * kBlock + threadIdx.x;
  if (r >= j.h2) return;
  uint32_t v;
  if (!(r & 1) && c < j.w) v = as_global(j.src)[(uint64_t)c * (j.h2 >> 1) + (r >> 1)];
  else v = j.zero_fill ? 0u : zk_rand_mont<PP>(key, j.stream, r * j.w2 + c);
  as_global(j.dst)[(uint64_t)c * j.h2 + r] = v;
}

// Quotient masks (prove_impl.hip.h step 4, ZK): the C - 1 independent masks t_c are n x DC random evaluations over
// the common coset U (matrices [DC][n], cell r * DC + k of the chunk's mask stream); the last one is their combination.
constexpr int kZkMaxChunks = 8;
struct ZkMaskArgs {
  uint32_t* t[kZkMaxChunks];    // [DC][n] each; t[C - 1] is written
  uint32_t stream[kZkMaxChunks];
  ZkKey key;
  uint32_t coef[kZkMaxChunks];  // -(k_c / k_{C-1}), Montgomery
  uint64_t n;
  int C, DC;
};
template <class PP>
__global__ void __launch_bounds__(kBlock) k_zk_masks(ZkMaskArgs a) {
  using F = Fp<PP>;
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;   // cell k * n + r
  if (i >= a.n * a.DC) return;
  const uint64_t k = i / a.n, r = i % a.n;
  F acc = F::zero();
  for (int c = 0; c + 1 < a.C; ++c) {
    const F v = F::raw(zk_rand_mont<PP>(a.key, a.stream[c], r * a.DC + k));
    as_global(a.t[c])[i] = v.v;
    acc += v * F::raw(a.coef[c]);
  }
  as_global(a.t[a.C - 1])[i] = acc.v;
}

// One randomised chunk matrix [DC + R][2n]: row 2r = q_c over its coset (natural order), row 2r + 1 = q_c - 2 t_c over
// the odd coset - both given with rows in bit-reversed order (coset_lde_batch with no added bits) -, the R codeword
// columns random (cell row * (DC + R) + col of the chunk's stream).
struct ZkChunkJob {
  const uint32_t* q;       // [DC][n] chunk evaluations, natural order
  const uint32_t* q_odd;   // [DC][n] over the odd coset, bit-reversed rows
  const uint32_t* t_odd;   // [DC][n] mask over the odd coset, bit-reversed rows
  uint32_t* dst;           // [DC + R][2n]
  uint64_t n;
  int log_n, DC, R;
  uint32_t stream;
  uint32_t block0;
};
template <class PP>
__global__ void __launch_bounds__(kBlock) k_zk_chunk(const ZkChunkJob* __restrict__ jobs, int n_jobs, ZkKey key) {
  using F = Fp<PP>;
  int jb = 0;
  while (jb + 1 < n_jobs && blockIdx.x >= jobs[jb + 1].block0) ++jb;
  const ZkChunkJob& j = jobs[jb];
  const uint64_t h2 = 2 * j.n, tiles_per_col = (h2 + kBlock - 1) / kBlock;
  const uint64_t t = blockIdx.x - j.block0;
  const uint32_t c = (uint32_t)(t / tiles_per_col);
  const uint64_t r = (t % tiles_per_col) * kBlock + threadIdx.x;
  if (r >= h2) return;
  uint32_t v;
  if ((int)c >= j.DC) {
    v = zk_rand_mont<PP>(key, j.stream, r * (uint64_t)(j.DC + j.R) + c);
  } else if (!(r & 1)) {
    v = as_global(j.q)[(uint64_t)c * j.n + (r >> 1)];
  } else {
    const uint64_t br = bit_reverse((uint32_t)(r >> 1), j.log_n);
    const F q = F::raw(as_global(j.q_odd)[(uint64_t)c * j.n + br]), m = F::raw(as_global(j.t_odd)[(uint64_t)c * j.n + br]);
    v = (q - m.dbl()).v;
  }
  as_global(j.dst)[(uint64_t)c * h2 + r] = v;
}

// Salts of a hiding MMCS (p3r_config.mmcs_salt_elems): cell (r, c) of the h x S salt matrix of one committed matrix is
// cell r * S + c of its stream.  Plain column-major matrices [S][h] (stride 1), or the strided layout of a FRI
// commit-phase leaf matrix: dst[c * h * stride + r * stride].
struct ZkSaltJob {
  uint32_t* dst;
  uint64_t h;
  uint32_t S, stride;
  uint32_t stream;
  uint32_t block0;
};
template <class PP>
__global__ void __launch_bounds__(kBlock) k_zk_salts(const ZkSaltJob* __restrict__ jobs, int n_jobs, ZkKey key) {
  int jb = 0;
  while (jb + 1 < n_jobs && blockIdx.x >= jobs[jb + 1].block0) ++jb;
  const ZkSaltJob& j = jobs[jb];
  const uint64_t tiles_per_col = (j.h + kBlock - 1) / kBlock;
  const uint64_t t = blockIdx.x - j.block0;
  const uint32_t c = (uint32_t)(t / tiles_per_col);
  const uint64_t r = (t % tiles_per_col) * kBlock + threadIdx.x;
  if (r >= j.h) return;
  as_global(j.dst)[((uint64_t)c * j.h + r) * j.stride] = zk_rand_mont<PP>(key, j.stream, r * j.S + c);
}

}  // namespace p3r
