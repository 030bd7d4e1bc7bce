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
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_p2_permute_batch(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n,
                   const uint32_t* __restrict__ rc) {
  using F = Fp<PP>;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  F s[P2_WIDTH];
#pragma unroll
  for (int k = 0; k < P2_WIDTH; ++k) s[k] = F::raw(in[(size_t)k * n + i]);
  p2_permute<PP>(s, rc);
#pragma unroll
  for (int k = 0; k < P2_WIDTH; ++k) out[(size_t)k * n + i] = s[k].v;
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
                                                    const uint32_t* mmcs_index_sum_mont) {
  using F = Fp<PP>;
  Affine<PP> m;
  if (i > 0 && merkle_path[i] && !new_start[i]) {
    m.a = F::one().dbl();
    m.b = mmcs_bit[i] ? F::one() : F::zero();
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
              size_t n_blocks, uint32_t* __restrict__ acc_out) {
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
        loc[k] = p2_row_affine<PP>(i, new_start, merkle_path, mmcs_bit, index_sum_mont);
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

// K6 leaf hashing: PaddingFreeSponge<Perm,16,8,8> over the concatenation of row i of
// every same-height matrix (recursion/src/pcs/mmcs.rs:17-26,75-172,355-425).  Overwrite
// mode: a chunk of k<=8 cells overwrites state[0..k]; an exact multiple of 8 does not
// trigger an extra permutation; the digest is state[0..8].
// `cols[g]` is the device address of concatenated column g (length h).
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_mmcs_hash_rows(const uint32_t* const* __restrict__ cols, int wtot, size_t h,
                 uint32_t* __restrict__ dig /* [8][h] */, const uint32_t* __restrict__ rc) {
  using F = Fp<PP>;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= h) return;
  F s[P2_WIDTH];
#pragma unroll
  for (int k = 0; k < P2_WIDTH; ++k) s[k] = F::zero();
  int g = 0;
  for (; g + P2_RATE <= wtot; g += P2_RATE) {
#pragma unroll
    for (int j = 0; j < P2_RATE; ++j) s[j] = F::raw(cols[g + j][i]);
    p2_permute<PP>(s, rc);
  }
  int rem = wtot - g;
  if (rem > 0) {
#pragma unroll
    for (int j = 0; j < P2_RATE; ++j)
      if (j < rem) s[j] = F::raw(cols[g + j][i]);
    p2_permute<PP>(s, rc);
  }
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) dig[(size_t)k * h + i] = s[k].v;
}

// K6 tree layers: TruncatedPermutation<Perm,2,8,16>: perm(left || right)[0..8]
// (circuit/src/ops/mmcs.rs:117-160).  left digest i = L[k*nl + i*lmul + ladd], same for right:
//   plain layer        L = R = prev, lmul = rmul = 2, ladd = 0, radd = 1
//   injection          L = compressed layer, R = digests of the shorter matrices, mul 1
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_mmcs_compress(const uint32_t* __restrict__ L, size_t nl, int lmul, int ladd,
                const uint32_t* __restrict__ R, size_t nr, int rmul, int radd,
                uint32_t* __restrict__ out, size_t n, const uint32_t* __restrict__ rc) {
  using F = Fp<PP>;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  F s[P2_WIDTH];
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) {
    s[k] = F::raw(L[(size_t)k * nl + i * lmul + ladd]);
    s[P2_DIGEST + k] = F::raw(R[(size_t)k * nr + i * rmul + radd]);
  }
  p2_permute<PP>(s, rc);
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) out[(size_t)k * n + i] = s[k].v;
}

// ---------------------------------------------------------------------------------
// K5: radix-2 NTT as LDS-staged tiles (DESIGN.md "NTT / LDE").
// A polynomial of N = N1*N2 cells is viewed as [N1][N2] (idx = n1*N2 + n2).  One launch
// transforms along ONE of the two dimensions for a tile of T lines; the size-R sub-NTT
// runs decimation-in-frequency inside LDS (result k lands in LDS row bitrev(k)).
// ---------------------------------------------------------------------------------
struct NttPass {
  const uint32_t* in;
  uint32_t* out;
  uint64_t in_col_stride;    // cells between consecutive polynomials (grid.y)
  uint64_t out_col_stride;
  uint64_t out_coset_stride; // cells between consecutive cosets (grid.z)
  int log_n1, log_n2;
  int sub_dim;   // 0: transform along n1 (stride-N2 lines), 1: along n2 (contiguous lines)
  int log_t;     // lines per tile
  int out_mode;  // 0 keep LDS row order (bit-reversed); 1 natural order, same geometry;
                 // 2 natural order, transposed: out[n2*N1 + k1] (sub_dim 0 only)
  const uint32_t* tw_sub;   // Shoup pairs (w_R^i canonical, floor(w*2^32/P)) for i < R/2
  const uint32_t* tw4_lo;   // optional 4-step twiddles: w_N^x = hi[x >> 10] * lo[x & 1023]
  const uint32_t* tw4_hi;
  const uint32_t* pre_a;    // optional per-coset input scaling pre_a[z][n1] * pre_b[z][n2]
  const uint32_t* pre_b;
  uint32_t scale;           // Montgomery; multiplied into every output when use_scale
  int use_scale;
  int inverse;              // direction (selects root16)
  uint32_t root16[8];       // w_16^k (or its inverse), k < 8: constants of the register radix-16
};

__device__ __forceinline__ uint32_t lds_addr(uint32_t r, uint32_t t, uint32_t T) {
  return r * (T + 1) + (r >> 5) + t;
}

constexpr int kNttBlock = 1024;  // upper bound; launches use tile_cells/16 lanes (one radix-16 group per lane)
// Shoup product: a (any u32) times a fixed w < P given w' = floor(w * 2^32 / P); 3 multiplies.
// The data stays in Montgomery form (x*R) while w is canonical: (x*R)*w = (x*w)*R.
template <class PP>
__device__ __forceinline__ Fp<PP> shoup_mul(uint32_t a, uint32_t w, uint32_t wp) {
  uint32_t q = __umulhi(a, wp);
  uint32_t r = a * w - q * PP::P;  // in [0, 2P)
  uint32_t r2 = r - PP::P;
  return Fp<PP>::raw(r < r2 ? r : r2);
}

// LOGM consecutive DIF stages (s .. s+LOGM-1) of the size-R sub-NTT done in registers: a lane
// owns the 2^LOGM rows r0 + j*q (q = R >> (s+LOGM)) of one tile column.  Stage s+u pairs
// (j, j + M/2^(u+1)) with twiddle w_R^{i << (s+u)}, i = jj*q + low, read as a (w, w') Shoup pair
// from the LDS copy of the twiddle table.
template <class PP, int LOGM>
__device__ __forceinline__ void ntt_stage_group(uint32_t* tile, const uint32_t* tws, int s, int log_r, int log_t,
                                                uint32_t tid) {
  using F = Fp<PP>;
  constexpr int M = 1 << LOGM;
  const uint32_t T = 1u << log_t;
  const int lq = log_r - s - LOGM;
  const uint32_t q = 1u << lq;
  const uint32_t items = (1u << (log_r - LOGM)) << log_t;
  for (uint32_t e = tid; e < items; e += blockDim.x) {
    const uint32_t t = e & (T - 1), b = e >> log_t;
    const uint32_t low = b & (q - 1), high = b >> lq;
    const uint32_t r0 = (high << (lq + LOGM)) | low;
    F x[M];
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = F::raw(tile[lds_addr(r0 + j * q, t, T)]);
#pragma unroll
    for (int u = 0; u < LOGM; ++u) {
      const int half = M >> (u + 1);
      const uint32_t base_idx = low << (s + u);
#pragma unroll
      for (int jj = 0; jj < M / 2; ++jj) {
        if (jj < half) {
          const uint32_t idx = base_idx + ((uint32_t)jj << (lq + s + u));
          const uint2 tw = *reinterpret_cast<const uint2*>(&tws[2 * idx]);
#pragma unroll
          for (int blk = 0; blk < M; blk += 2 * half) {
            F p = x[blk + jj], c = x[blk + jj + half];
            x[blk + jj] = p + c;
            x[blk + jj + half] = shoup_mul<PP>(p.v + (PP::P - c.v), tw.x, tw.y);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) tile[lds_addr(r0 + j * q, t, T)] = x[j].v;
  }
  __syncthreads();
}

template <class PP>
__global__ void __launch_bounds__(kNttBlock) k_ntt_tile(NttPass a) {
  using F = Fp<PP>;
  extern __shared__ uint32_t lds[];
  const uint32_t tid = threadIdx.x;
  const int log_r = a.sub_dim == 0 ? a.log_n1 : a.log_n2;
  const uint32_t R = 1u << log_r, T = 1u << a.log_t;
  const uint32_t N1 = 1u << a.log_n1, N2 = 1u << a.log_n2;
  uint32_t* tile = lds;
  uint32_t* tws = lds + ((R * (T + 1) + (R >> 5) + 2) & ~1u);  // 8-byte aligned (w, w') pairs
  const uint32_t line0 = blockIdx.x * T;  // first line of the tile (n2 for sub_dim 0, n1 for 1)
  const uint32_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
  uint32_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride +
                  (size_t)blockIdx.z * a.out_coset_stride;
  const uint32_t* pre_a = a.pre_a ? a.pre_a + (size_t)blockIdx.z * N1 : nullptr;
  const uint32_t* pre_b = a.pre_b ? a.pre_b + (size_t)blockIdx.z * N2 : nullptr;

  for (uint32_t i = tid; i < R; i += blockDim.x) tws[i] = a.tw_sub[i];

  const uint32_t E = R << a.log_t;
  // ---- load (lanes run along the unit-stride global dimension) ----
  for (uint32_t e = tid; e < E; e += blockDim.x) {
    uint32_t r, t, n1, n2;
    if (a.sub_dim == 0) {
      t = e & (T - 1); r = e >> a.log_t; n1 = r; n2 = line0 + t;
    } else {
      r = e & (R - 1); t = e >> log_r; n1 = line0 + t; n2 = r;
    }
    F v = F::raw(in[(size_t)n1 * N2 + n2]);
    if (pre_a) v = v * (F::raw(pre_a[n1]) * F::raw(pre_b[n2]));
    tile[lds_addr(r, t, T)] = v.v;
  }
  __syncthreads();
  // ---- DIF butterflies, up to four stages per LDS round trip (register radix-16) ----
  {
    int s = 0;
    while (log_r - s >= 4) { ntt_stage_group<PP, 4>(tile, tws, s, log_r, a.log_t, tid); s += 4; }
    if (log_r - s == 3) ntt_stage_group<PP, 3>(tile, tws, s, log_r, a.log_t, tid);
    else if (log_r - s == 2) ntt_stage_group<PP, 2>(tile, tws, s, log_r, a.log_t, tid);
    else if (log_r - s == 1) ntt_stage_group<PP, 1>(tile, tws, s, log_r, a.log_t, tid);
  }
  // ---- store ----
  const F scale = F::raw(a.scale);
  for (uint32_t e = tid; e < E; e += blockDim.x) {
    uint32_t rho, t;  // rho: row index in the OUTPUT geometry
    bool lanes_along_t = (a.sub_dim == 0 && a.out_mode != 2);
    if (lanes_along_t) {
      t = e & (T - 1); rho = e >> a.log_t;
    } else {
      rho = e & (R - 1); t = e >> log_r;
    }
    uint32_t r = a.out_mode == 0 ? rho : bit_reverse(rho, log_r);  // LDS row
    uint32_t k = a.out_mode == 0 ? bit_reverse(rho, log_r) : rho;  // sub-NTT output index
    F v = F::raw(tile[lds_addr(r, t, T)]);
    uint32_t line = line0 + t;
    if (a.tw4_lo) {
      uint32_t x = k * line;  // < N
      v = v * (F::raw(a.tw4_hi[x >> 10]) * F::raw(a.tw4_lo[x & 1023]));
    }
    if (a.use_scale) v = v * scale;
    size_t o;
    if (a.sub_dim == 0) {
      o = a.out_mode == 2 ? (size_t)line * N1 + rho : (size_t)rho * N2 + line;
    } else {
      o = (size_t)line * N2 + rho;
    }
    out[o] = v.v;
  }
}

}  // namespace p3r
