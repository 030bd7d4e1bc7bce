// gfx950 device kernels for the batch-STARK hot path.
//
// HBM layout used by every kernel here (DESIGN.md "Data layout"):
//   * a matrix of height h (power of two) and width w is COLUMN-MAJOR: d[c*h + r],
//     each cell a Montgomery-form u32.  One lane owns one row, so every per-column
//     access of a wavefront is one fully coalesced 256-byte request.
//   * low-degree extensions keep upstream's committed order: row i is the evaluation at
//     GENERATOR * w^{bitrev(i)} (recursion/src/pcs/fri/verifier.rs:921-981).
//   * digests are struct-of-arrays: dig[k*n + i], k < 8.
#pragma once
#include "field.h"
#include "poseidon2.h"
#include "poseidon2_f64.hip.h"
#include "kernels_ntt.hip.h"

namespace p3r {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------
// Boundary layout conversion: row-major canonical (ABI)  <->  column-major Montgomery.
// 64x64 tiles staged through LDS so both sides are coalesced.
// ---------------------------------------------------------------------------------
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_rowmajor_to_colmajor(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                       uint32_t h, uint32_t w, int to_monty) {
  __shared__ uint32_t tile[64][65];
  using F = Fp<PP>;
  const uint32_t r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (uint32_t rr = ty; rr < 64; rr += 4) {
    uint32_t r = r0 + rr, c = c0 + tx;
    if (r < h && c < w) tile[rr][tx] = src[(size_t)r * w + c];
  }
  __syncthreads();
  for (uint32_t cc = ty; cc < 64; cc += 4) {
    uint32_t r = r0 + tx, c = c0 + cc;
    if (r < h && c < w) {
      uint32_t v = tile[tx][cc];
      dst[(size_t)c * h + r] = to_monty ? F::from_canonical(v).v : v;
    }
  }
}

template <class PP>
__global__ void __launch_bounds__(kBlock)
k_colmajor_to_rowmajor(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                       uint32_t h, uint32_t w, int from_monty) {
  __shared__ uint32_t tile[64][65];
  using F = Fp<PP>;
  const uint32_t r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (uint32_t cc = ty; cc < 64; cc += 4) {
    uint32_t r = r0 + tx, c = c0 + cc;
    if (r < h && c < w) {
      uint32_t v = src[(size_t)c * h + r];
      tile[cc][tx] = from_monty ? F::raw(v).to_canonical() : v;
    }
  }
  __syncthreads();
  for (uint32_t rr = ty; rr < 64; rr += 4) {
    uint32_t r = r0 + rr, c = c0 + tx;
    if (r < h && c < w) dst[(size_t)r * w + c] = tile[tx][rr];
  }
}

template <class PP>
__global__ void __launch_bounds__(kBlock)
k_convert_inplace(uint32_t* __restrict__ d, size_t n, int to_monty) {
  using F = Fp<PP>;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) d[i] = to_monty ? F::from_canonical(d[i]).v : F::raw(d[i]).to_canonical();
}

// ---------------------------------------------------------------------------------
// Poseidon2, one permutation per lane (state register-resident, 16 VGPRs).
// ---------------------------------------------------------------------------------

// Plain batch permutation: states column-major [16][n] in and out (the perms/s metric).
// FP64 form (poseidon2_f64.hip.h): `rcd` = the constants as canonical doubles.
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_p2_permute_batch(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n,
                   const double* __restrict__ rcd) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double s[P2_WIDTH];
#pragma unroll
  for (int k = 0; k < P2_WIDTH; ++k) s[k] = p2f_load<PP>(in[(size_t)k * n + i]);
  p2f_permute<PP, 0u>(s, rcd);
#pragma unroll
  for (int k = 0; k < P2_WIDTH; ++k) out[(size_t)k * n + i] = p2f_store<PP>(s[k]);
}

// K3 pass 1: the MMCS index accumulator is the segmented affine recurrence
//   acc_i = (i>0 && merkle_i && !new_start_i) ? 2*acc_{i-1} + bit_i : mmcs_index_sum_i
// (poseidon2-circuit-air/src/air.rs:401-412).  Each row is the map x -> a*x + b with
// a in {0,2}; maps compose associatively so the column is a 3-kernel block scan.
template <class PP>
struct Affine {
  Fp<PP> a, b;
};
template <class PP>
__device__ __forceinline__ Affine<PP> affine_compose(Affine<PP> first, Affine<PP> then) {
  Affine<PP> r;
  r.a = then.a * first.a;
  r.b = then.a * first.b + then.b;
  return r;
}
template <class PP>
__device__ __forceinline__ Affine<PP> p2_row_affine(size_t i, const uint8_t* new_start,
                                                    const uint8_t* merkle_path,
                                                    const uint8_t* mmcs_bit,
                                                    const uint32_t* mmcs_index_sum_mont,
                                                    const uint8_t* mmcs_bit2 = nullptr) {
  using F = Fp<PP>;
  Affine<PP> m;
  if (i > 0 && merkle_path[i] && !new_start[i]) {
    if (mmcs_bit2) {
      // arity-4 rows (the width-32 table): base four, acc <- 4 acc + bit + 2 bit2 (air.rs:401-412)
      m.a = F::one().dbl().dbl();
      m.b = F::from_canonical((uint32_t)(mmcs_bit[i] ? 1 : 0) + (mmcs_bit2[i] ? 2u : 0u));
    } else {
      m.a = F::one().dbl();
      m.b = mmcs_bit[i] ? F::one() : F::zero();
    }
  } else {
    m.a = F::zero();
    m.b = F::raw(mmcs_index_sum_mont[i]);
  }
  return m;
}

constexpr int kScanItems = 4;                       // rows per thread
constexpr int kScanTile = kBlock * kScanItems;      // rows per block

// mode 0: write the block aggregate; mode 1: scan block aggregates in place (one block);
// mode 2: apply (exclusive block prefix from `agg`) and write the accumulator column.
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_p2_acc_scan(int mode, size_t n, const uint8_t* __restrict__ new_start,
              const uint8_t* __restrict__ merkle_path, const uint8_t* __restrict__ mmcs_bit,
              const uint32_t* __restrict__ index_sum_mont, uint32_t* __restrict__ agg,
              size_t n_blocks, uint32_t* __restrict__ acc_out, const uint8_t* __restrict__ mmcs_bit2 = nullptr) {
  using F = Fp<PP>;
  __shared__ uint32_t sa[kBlock], sb[kBlock];
  const int tid = threadIdx.x;
  Affine<PP> loc[kScanItems];
  Affine<PP> run;
  run.a = F::one();
  run.b = F::zero();
  if (mode == 1) {
    // single block: thread t owns aggregates [t*per, (t+1)*per)
    size_t per = (n_blocks + kBlock - 1) / kBlock;
    size_t lo = (size_t)tid * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    for (size_t j = lo; j < hi; ++j) {
      Affine<PP> m;
      m.a = F::raw(agg[2 * j]);
      m.b = F::raw(agg[2 * j + 1]);
      run = affine_compose(run, m);
    }
  } else {
    size_t base = (size_t)blockIdx.x * kScanTile + (size_t)tid * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      size_t i = base + k;
      if (i < n) {
        loc[k] = p2_row_affine<PP>(i, new_start, merkle_path, mmcs_bit, index_sum_mont, mmcs_bit2);
      } else {
        loc[k].a = F::one();
        loc[k].b = F::zero();
      }
      run = affine_compose(run, loc[k]);
    }
  }
  // inclusive Hillis-Steele scan of per-thread aggregates across the block
  sa[tid] = run.a.v;
  sb[tid] = run.b.v;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {
    Affine<PP> prev, cur;
    bool has = tid >= off;
    if (has) {
      prev.a = F::raw(sa[tid - off]);
      prev.b = F::raw(sb[tid - off]);
    }
    cur.a = F::raw(sa[tid]);
    cur.b = F::raw(sb[tid]);
    __syncthreads();
    if (has) {
      cur = affine_compose(prev, cur);
      sa[tid] = cur.a.v;
      sb[tid] = cur.b.v;
    }
    __syncthreads();
  }
  Affine<PP> excl;  // composition of everything before this thread inside the block
  if (tid == 0) {
    excl.a = F::one();
    excl.b = F::zero();
  } else {
    excl.a = F::raw(sa[tid - 1]);
    excl.b = F::raw(sb[tid - 1]);
  }
  if (mode == 0) {
    if (tid == kBlock - 1) {
      agg[2 * (size_t)blockIdx.x] = sa[tid];
      agg[2 * (size_t)blockIdx.x + 1] = sb[tid];
    }
  } else if (mode == 1) {
    // rewrite aggregates as EXCLUSIVE prefixes
    size_t per = (n_blocks + kBlock - 1) / kBlock;
    size_t lo = (size_t)tid * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    Affine<PP> p = excl;
    for (size_t j = lo; j < hi; ++j) {
      Affine<PP> m;
      m.a = F::raw(agg[2 * j]);
      m.b = F::raw(agg[2 * j + 1]);
      agg[2 * j] = p.a.v;
      agg[2 * j + 1] = p.b.v;
      p = affine_compose(p, m);
    }
  } else {
    Affine<PP> p;
    p.a = F::raw(agg[2 * (size_t)blockIdx.x]);
    p.b = F::raw(agg[2 * (size_t)blockIdx.x + 1]);
    p = affine_compose(p, excl);
    size_t base = (size_t)blockIdx.x * kScanTile + (size_t)tid * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      size_t i = base + k;
      p = affine_compose(p, loc[k]);
      // The scan starts from acc = 0 before row 0; p.b is the map applied to 0.
      if (i < n) acc_out[i] = p.b.v;
    }
  }
}

// K3 pass 2: one lane per circuit row; writes every Poseidon2Cols cell plus the two
// circuit columns (mmcs_bit, mmcs_index_sum) column-major
// (poseidon2-circuit-air/src/air.rs:414-434,454-506; column order SURVEY.md appendix A).
template <class F>
struct P2ColSink {
  uint32_t* p;
  size_t stride;
  __device__ __forceinline__ void put(F x) {
    *p = x.v;
    p += stride;
  }
};

template <class PP>
__global__ void __launch_bounds__(kBlock)
k_p2_trace_fill(const uint32_t* __restrict__ inputs /* [16][n] mont */,
                const uint8_t* __restrict__ mmcs_bit, const uint32_t* __restrict__ acc /* [n] mont */,
                uint32_t* __restrict__ trace /* [cols][n] */, size_t n,
                const uint32_t* __restrict__ rc) {
  using F = Fp<PP>;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  F s[P2_WIDTH];
  P2ColSink<F> sink{trace + i, n};
#pragma unroll
  for (int k = 0; k < P2_WIDTH; ++k) {
    s[k] = F::raw(inputs[(size_t)k * n + i]);
    sink.put(s[k]);
  }
  p2_permute_traced<PP>(s, rc, sink);
  sink.put(mmcs_bit[i] ? F::one() : F::zero());
  sink.put(F::raw(acc[i]));
}
// the width-32 table: [Poseidon2Cols<32> | mmcs_bit | mmcs_bit2 | mmcs_bit * mmcs_bit2 | mmcs_index_sum]
// (poseidon2-circuit-air/src/air.rs:414-434 arity-4 branch); rcw = the width-32 constant table
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_p2w_trace_fill(const uint32_t* __restrict__ inputs /* [32][n] mont */, const uint8_t* __restrict__ mmcs_bit,
                 const uint8_t* __restrict__ mmcs_bit2, const uint32_t* __restrict__ acc /* [n] mont */,
                 uint32_t* __restrict__ trace /* [cols][n] */, size_t n, const uint32_t* __restrict__ rcw) {
  using F = Fp<PP>;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  F s[P2W_WIDTH];
  P2ColSink<F> sink{trace + i, n};
#pragma unroll
  for (int k = 0; k < P2W_WIDTH; ++k) {
    s[k] = F::raw(inputs[(size_t)k * n + i]);
    sink.put(s[k]);
  }
  p2w_permute_traced<PP>(s, rcw, sink);
  const bool b0 = mmcs_bit[i], b1 = mmcs_bit2[i];
  sink.put(b0 ? F::one() : F::zero());
  sink.put(b1 ? F::one() : F::zero());
  sink.put((b0 && b1) ? F::one() : F::zero());
  sink.put(F::raw(acc[i]));
}

// K6 leaf hashing: PaddingFreeSponge<Perm,16,8,8> over the concatenation of row i of
// every same-height matrix (recursion/src/pcs/mmcs.rs:17-26,75-172,355-425).  Overwrite
// mode: a chunk of k<=8 cells overwrites state[0..k]; an exact multiple of 8 does not
// trigger an extra permutation; the digest is state[0..8].
// `cols[g]` is the device address of concatenated column g (length h).
// A commit hashes the rows of every height class in ONE launch (job list, widest rows first): the
// short tables of a layer do not fill the chip on their own, and for 2^14..2^16-row layers each
// launch saved is a visible fraction of the proof.
struct HashRowsJob {
  const uint32_t* const* cols;
  uint32_t* dig;  // [8][h]
  uint64_t h;
  int wtot;
  uint32_t block0;  // first block of this job
};
template <class PP>
__global__ void __launch_bounds__(kBlock, 5)
k_mmcs_hash_rows(const HashRowsJob* __restrict__ jobs, int n_jobs, const double* __restrict__ rcd) {
  const int jb = find_job(jobs, n_jobs);
  const gptr<const uint32_t* const> cols = as_global(jobs[jb].cols);
  const gptr<uint32_t> dig = as_global(jobs[jb].dig);
  const size_t h = jobs[jb].h;
  const int wtot = jobs[jb].wtot;
  size_t i = (size_t)(blockIdx.x - jobs[jb].block0) * kBlock + threadIdx.x;
  if (i >= h) return;
  // the sponge state lives in FP64 between permutations (poseidon2_f64.hip.h): absorbed cells are
  // converted on the way in, the capacity half is carried unreduced, the digest is reduced once
  double s[P2_WIDTH];
#pragma unroll
  for (int k = 0; k < P2_WIDTH; ++k) s[k] = 0.0;
  int g = 0;
  for (; g + P2_RATE <= wtot; g += P2_RATE) {
#pragma unroll
    for (int j = 0; j < P2_RATE; ++j) s[j] = p2f_load<PP>(as_global(cols[g + j])[i]);
    p2f_permute<PP, 0xFF00u>(s, rcd);   // the capacity half is carried
  }
  int rem = wtot - g;
  if (rem > 0) {
#pragma unroll
    for (int j = 0; j < P2_RATE; ++j)
      if (j < rem) s[j] = p2f_load<PP>(as_global(cols[g + j])[i]);
    p2f_permute<PP, 0xFFFFu>(s, rcd);   // and the rate lanes past `rem` (reducing a fresh cell is harmless)
  }
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) dig[(size_t)k * h + i] = p2f_store<PP>(s[k]);
}

// K6 tree layers: TruncatedPermutation<Perm,2,8,16>: perm(left || right)[0..8]
// (circuit/src/ops/mmcs.rs:117-160).  Node i of the new layer = compress(prev[2i], prev[2i+1]); when
// the commit has matrices of this layer's height their row digests `inj` are folded in by the same
// lane, node = compress(compress(l, r), inj[i]) - the intermediate digest never leaves registers.
// prev: [8][2n], inj: [8][n] or null, out: [8][n].
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_mmcs_compress(const uint32_t* __restrict__ prev, const uint32_t* __restrict__ inj, uint32_t* __restrict__ out,
                size_t n, const double* __restrict__ rcd) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double s[P2_WIDTH];
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) {
    const uint2 lr = *reinterpret_cast<const uint2*>(prev + (size_t)k * 2 * n + 2 * i);  // siblings are adjacent
    s[k] = p2f_load<PP>(lr.x);
    s[P2_DIGEST + k] = p2f_load<PP>(lr.y);
  }
  p2f_permute<PP, 0u>(s, rcd);
  if (inj) {
#pragma unroll
    for (int k = 0; k < P2_DIGEST; ++k) s[P2_DIGEST + k] = p2f_load<PP>(inj[(size_t)k * n + i]);
    p2f_permute<PP, 0x00FFu>(s, rcd);   // the node's own digest is carried
  }
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) out[(size_t)k * n + i] = p2f_store<PP>(s[k]);
}

}  // namespace p3r
