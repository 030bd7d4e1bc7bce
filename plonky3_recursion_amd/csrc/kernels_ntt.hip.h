// K5: radix-2 NTT as LDS-staged tiles (DESIGN.md "NTT / LDE").
//
// A polynomial of N = N1*N2 cells is viewed as [N1][N2] (idx = n1*N2 + n2).  One launch
// transforms along ONE of the two dimensions for a tile of T lines; the size-R sub-NTT runs
// decimation-in-frequency (result k lands in row bitrev(k)), up to four stages at a time in
// registers (radix-16), with Montgomery twiddles read from an LDS copy of the table.
//   * the FIRST stage group reads its 16 cells straight from global memory (with the optional
//     coset pre-scaling) and the LAST one, for strided passes, writes straight back (with the
//     optional four-step twiddle / scaling), so a 2^10-point sub-NTT makes only two LDS round
//     trips and two barriers;
//   * tiles are 2^13 cells with 512 lanes, four tiles per CU, so the global-memory phases of
//     one tile overlap the arithmetic of the others.
// Reference: TwoAdicSubgroupDft::coset_lde_batch as used by TwoAdicFriPcs::commit
// (circuit-prover/src/config.rs:55,131); row order recursion/src/pcs/fri/verifier.rs:921-981.
#pragma once
#include "field.h"

namespace p3r {

struct NttPass {
  const uint32_t* in;
  uint32_t* out;
  uint64_t in_col_stride;    // cells between consecutive polynomials (grid.y)
  uint64_t out_col_stride;
  uint64_t out_coset_stride; // cells between consecutive cosets (grid.z)
  int log_n1, log_n2;
  int sub_dim;   // 0: transform along n1 (stride-N2 lines), 1: along n2 (contiguous lines)
  int log_t;     // lines per tile
  int out_mode;  // 0 keep row order (bit-reversed); 1 natural order, same geometry;
                 // 2 natural order, transposed: out[n2*N1 + k1] (sub_dim 0 only)
  const uint32_t* tw_sub;   // w_R^i in Montgomery form, i < R/2
  const uint32_t* tw4_lo;   // optional 4-step twiddles: w_N^x = hi[x >> 10] * lo[x & 1023]
  const uint32_t* tw4_hi;
  const uint32_t* pre_a;    // optional per-coset input scaling pre_a[z][n1] * pre_b[z][n2]
  const uint32_t* pre_b;
  uint32_t scale;           // Montgomery; multiplied into every output when use_scale
  int use_scale;
  int inverse;
  // position in a job-list launch: this pass owns (polynomials << (log_gx + log_gz)) blocks
  // starting at block0; 2^log_gx tiles per polynomial and coset, 2^log_gz cosets
  uint32_t block0;
  int log_gx, log_gz;
};

constexpr int kNttBlock = 512;  // launches use tile_cells/16 lanes (2^13-cell tiles)

__device__ __forceinline__ uint32_t lds_addr(uint32_t r, uint32_t t, uint32_t T) {
  return r * (T + 1) + (r >> 5) + t;
}

// Per-block view of the pass: where a tile cell lives in global memory and what is applied to
// it on the way in and out.
template <class PP>
struct NttTileIo {
  using F = Fp<PP>;
  const NttPass& a;
  gptr<const uint32_t> in;
  gptr<uint32_t> out;
  gptr<const uint32_t> pre_a;
  gptr<const uint32_t> pre_b;
  gptr<const uint32_t> tw4_lo;
  gptr<const uint32_t> tw4_hi;
  uint32_t line0, N1, N2;
  int log_r;
  __device__ __forceinline__ F load(uint32_t r, uint32_t t) const {
    uint32_t n1, n2;
    if (a.sub_dim == 0) { n1 = r; n2 = line0 + t; } else { n1 = line0 + t; n2 = r; }
    F v = F::raw(in[(size_t)n1 * N2 + n2]);
    if (pre_a) v = v * (F::raw(pre_a[n1]) * F::raw(pre_b[n2]));
    return v;
  }
  // `r` = row of the DIF result inside the tile (holds sub-NTT output bitrev(r))
  __device__ __forceinline__ void store_row(uint32_t r, uint32_t t, F v) const {
    const uint32_t k = bit_reverse(r, log_r);
    const uint32_t rho = a.out_mode == 0 ? r : k;
    const uint32_t line = line0 + t;
    if (tw4_lo) {
      uint32_t x = k * line;  // < N
      v = v * (F::raw(tw4_hi[x >> 10]) * F::raw(tw4_lo[x & 1023]));
    }
    if (a.use_scale) v = v * F::raw(a.scale);
    size_t o;
    if (a.sub_dim == 0) o = a.out_mode == 2 ? (size_t)line * N1 + rho : (size_t)rho * N2 + line;
    else o = (size_t)line * N2 + rho;
    out[o] = v.v;
  }
};

// LOGM consecutive DIF stages (s .. s+LOGM-1) in registers: a lane owns the 2^LOGM rows
// r0 + j*q (q = R >> (s+LOGM)) of one tile column.  Stage s+u pairs (j, j + M/2^(u+1)) with
// twiddle w_R^{i << (s+u)}, i = jj*q + low.
//   FROM_GLOBAL: this is the first group (s == 0): cells come from global memory.
//   TO_GLOBAL  : this is the last group (q == 1): results go to global memory.
template <class PP, int LOGM, bool FROM_GLOBAL, bool TO_GLOBAL>
__device__ __forceinline__ void ntt_stage_group(uint32_t* tile, const uint32_t* tws, const NttTileIo<PP>& io, int s,
                                                int log_r, int log_t, uint32_t tid) {
  using F = Fp<PP>;
  constexpr int M = 1 << LOGM;
  const uint32_t T = 1u << log_t;
  const int lq = log_r - s - LOGM;
  const uint32_t q = 1u << lq;
  const int log_items_col = log_r - LOGM;
  const uint32_t items = (1u << log_items_col) << log_t;
  for (uint32_t e = tid; e < items; e += blockDim.x) {
    uint32_t t, b;
    if (FROM_GLOBAL && io.a.sub_dim == 1) {  // contiguous lines: lanes run along the rows
      b = e & ((1u << log_items_col) - 1);
      t = e >> log_items_col;
    } else {
      t = e & (T - 1);
      b = e >> log_t;
    }
    const uint32_t low = b & (q - 1), high = b >> lq;
    const uint32_t r0 = (high << (lq + LOGM)) | low;
    F x[M];
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = FROM_GLOBAL ? io.load(r0 + j * q, t) : F::raw(tile[lds_addr(r0 + j * q, t, T)]);
#pragma unroll
    for (int u = 0; u < LOGM; ++u) {
      const int half = M >> (u + 1);
      const uint32_t base_idx = low << (s + u);
#pragma unroll
      for (int jj = 0; jj < M / 2; ++jj) {
        if (jj < half) {
          const uint32_t idx = base_idx + ((uint32_t)jj << (lq + s + u));
          const uint32_t tw = tws[idx];
#pragma unroll
          for (int blk = 0; blk < M; blk += 2 * half) {
            F p = x[blk + jj], c = x[blk + jj + half];
            x[blk + jj] = p + c;
            // (p - c + P) < 2P < 2^32 times a Montgomery twiddle < P: within REDC's input range
            x[blk + jj + half] = F::raw(F::reduce64((uint64_t)(p.v + (PP::P - c.v)) * tw));
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      if (TO_GLOBAL) io.store_row(r0 + j * q, t, x[j]);
      else tile[lds_addr(r0 + j * q, t, T)] = x[j].v;
    }
  }
  if (!TO_GLOBAL) __syncthreads();
}

template <class PP, bool FROM_GLOBAL, bool TO_GLOBAL>
__device__ __forceinline__ void ntt_group_dispatch(int logm, uint32_t* tile, const uint32_t* tws,
                                                   const NttTileIo<PP>& io, int s, int log_r, int log_t, uint32_t tid) {
  switch (logm) {
    case 4: ntt_stage_group<PP, 4, FROM_GLOBAL, TO_GLOBAL>(tile, tws, io, s, log_r, log_t, tid); break;
    case 3: ntt_stage_group<PP, 3, FROM_GLOBAL, TO_GLOBAL>(tile, tws, io, s, log_r, log_t, tid); break;
    case 2: ntt_stage_group<PP, 2, FROM_GLOBAL, TO_GLOBAL>(tile, tws, io, s, log_r, log_t, tid); break;
    default: ntt_stage_group<PP, 1, FROM_GLOBAL, TO_GLOBAL>(tile, tws, io, s, log_r, log_t, tid); break;
  }
}

// One launch runs the same pass of SEVERAL matrices (all tables of a commit): `jobs` lists them,
// a block finds its pass by walking the first-block indices.  The loops below stride by
// blockDim, so a pass may get more lanes than its tile has 16-cell items.
template <class PP>
__global__ void __launch_bounds__(kNttBlock, 8) k_ntt_tile(const NttPass* __restrict__ jobs, int n_jobs) {
  using F = Fp<PP>;
  extern __shared__ uint32_t lds[];
  const int jb = find_job(jobs, n_jobs);
  const NttPass a = jobs[jb];
  // tile (x) fastest, then coset (z), then polynomial (y); x and z counts are powers of two
  const uint32_t local = blockIdx.x - a.block0;
  const uint32_t bx = local & ((1u << a.log_gx) - 1);
  const uint32_t bz = (local >> a.log_gx) & ((1u << a.log_gz) - 1);
  const uint32_t by = local >> (a.log_gx + a.log_gz);
  const uint32_t tid = threadIdx.x;
  const int log_r = a.sub_dim == 0 ? a.log_n1 : a.log_n2;
  const uint32_t R = 1u << log_r, T = 1u << a.log_t;
  const uint32_t N1 = 1u << a.log_n1, N2 = 1u << a.log_n2;
  uint32_t* tile = lds;
  uint32_t* tws = lds + ((R * (T + 1) + (R >> 5) + 2) & ~1u);  // twiddle copy, after the padded tile
  NttTileIo<PP> io{a,
                   as_global(a.in) + (size_t)by * a.in_col_stride,
                   as_global(a.out) + (size_t)by * a.out_col_stride + (size_t)bz * a.out_coset_stride,
                   a.pre_a ? as_global(a.pre_a) + (size_t)bz * N1 : nullptr,
                   a.pre_b ? as_global(a.pre_b) + (size_t)bz * N2 : nullptr,
                   as_global(a.tw4_lo), as_global(a.tw4_hi),
                   bx * T, N1, N2, log_r};
  const gptr<const uint32_t> tw_sub = as_global(a.tw_sub);
  for (uint32_t i = tid; i < R / 2; i += blockDim.x) tws[i] = tw_sub[i];
  __syncthreads();
  const uint32_t E = R << a.log_t;

  // stage-group plan: groups of 4 stages, remainder last
  const int n_groups = (log_r + 3) / 4;
  // the last group may go straight to global memory when a lane's rows map to coalesced stores
  const bool direct_store = a.sub_dim == 0 && a.out_mode != 2 && n_groups >= 2;
  if (log_r == 0) {
    for (uint32_t e = tid; e < E; e += blockDim.x) io.store_row(0, e, io.load(0, e));
    return;
  }
  int s = 0;
  {
    const int m0 = log_r >= 4 ? 4 : log_r;
    ntt_group_dispatch<PP, true, false>(m0, tile, tws, io, 0, log_r, a.log_t, tid);
    s = m0;
  }
  while (log_r - s > 4) {
    ntt_stage_group<PP, 4, false, false>(tile, tws, io, s, log_r, a.log_t, tid);
    s += 4;
  }
  if (log_r - s > 0) {
    const int ml = log_r - s;
    if (direct_store) {
      ntt_group_dispatch<PP, false, true>(ml, tile, tws, io, s, log_r, a.log_t, tid);
      return;
    }
    ntt_group_dispatch<PP, false, false>(ml, tile, tws, io, s, log_r, a.log_t, tid);
  }
  // ---- store through LDS (lanes run along the unit-stride global dimension) ----
  for (uint32_t e = tid; e < E; e += blockDim.x) {
    uint32_t rho, t;  // rho: row index in the OUTPUT geometry
    const bool lanes_along_t = (a.sub_dim == 0 && a.out_mode != 2);
    if (lanes_along_t) { t = e & (T - 1); rho = e >> a.log_t; }
    else { rho = e & (R - 1); t = e >> log_r; }
    const uint32_t r = a.out_mode == 0 ? rho : bit_reverse(rho, log_r);  // tile row
    io.store_row(r, t, F::raw(tile[lds_addr(r, t, T)]));
  }
}

}  // namespace p3r
