// K5, lean form of the two FORWARD passes of a coset LDE (the bulk of the NTT work: 4 cosets):
//   k_ntt_fwd_col   pass 1: size-R1 transforms along the strided dimension of the coefficient
//                   vector, all cosets, input scaled by the coset shift powers, output multiplied
//                   by the four-step twiddles, rows left in bit-reversed order;
//   k_ntt_fwd_line  pass 2: contiguous size-R2 transforms in place, rows left in bit-reversed order.
// Same arithmetic and data movement as k_ntt_tile (kernels_ntt.hip.h) - identical outputs - but the
// geometry is a template parameter: one 16-cell item per lane and stage group, every shift and
// stride a compile-time constant, the twiddles of the last stage group immediates, no per-cell
// 64-bit address arithmetic.  (rocprof, round 2: k_ntt_tile spends 10.7 VALU instructions per cell
// and stage where the butterfly itself needs 5; profiles/r02/pmc_sq.json.)
//
// Tile: 2^13 cells = 512 lanes x 16 cells, four tiles per CU.
#pragma once
#include "kernels_ntt.hip.h"

namespace p3r {

constexpr int kNtt2LogTile = 13;
constexpr int kNtt2Lanes = 512;

// w_{2^log_r}^idx in Montgomery form, at compile time (the twiddles of a last stage group are the
// 16th roots of unity and their squares: immediates instead of LDS reads)
template <class PP>
constexpr uint32_t ntt2_root_mont(int log_r, uint32_t idx, bool inverse) {
  uint32_t g = pow_mod(PP::GEN, ((uint64_t)PP::P - 1) >> log_r, PP::P);
  if (inverse) g = pow_mod(g, (uint64_t)PP::P - 2, PP::P);
  const uint64_t w = pow_mod(g, idx, PP::P);
  return (uint32_t)((w << 32) % PP::P);
}

// twiddles of a LAST stage group (S .. S+ML-1, q = 1): v[u][jj] = w_R^(jj << (S+u))
template <class PP, int LOG_R, int S, int ML, bool INV>
struct Ntt2LastTw {
  uint32_t v[4][8];
  constexpr Ntt2LastTw() : v() {
    for (int u = 0; u < ML; ++u)
      for (int jj = 0; jj < ((1 << ML) >> (u + 1)); ++jj) v[u][jj] = ntt2_root_mont<PP>(LOG_R, (uint32_t)jj << (S + u), INV);
  }
};

// DIF butterfly: (p, c) <- (p + c, (p - c) * tw)
template <class PP>
__device__ __forceinline__ void ntt2_bfly(Fp<PP>& p, Fp<PP>& c, uint32_t tw) {
  using F = Fp<PP>;
  const F s = p + c;
  // (p - c + P) < 2P < 2^32 times a Montgomery twiddle < P: within REDC's input range
  c = F::raw(F::reduce64((uint64_t)(p.v + (PP::P - c.v)) * tw));
  p = s;
}
template <class PP>
__device__ __forceinline__ void ntt2_bfly_one(Fp<PP>& p, Fp<PP>& c) {  // twiddle 1
  const Fp<PP> s = p + c, d = p - c;
  p = s;
  c = d;
}

// ML stages (S .. S+ML-1) of a size-2^LOG_R DIF transform on the 16 registers of a lane.
// M = 2^ML cells form an item (rows r0 + j*q, q = 2^(LOG_R-S-ML)); a lane holds 16/M items in
// x[i*M + j].  LAST (q == 1): the twiddles are compile-time constants; otherwise `tws` is the
// table w_R^i (i < R/2) and `low` the item's position below q.  INV: the constants are powers of w_R^-1
// (the table is the caller's, already inverse).
template <class PP, int LOG_R, int S, int ML, bool LAST, bool INV = false, class TW = const uint32_t*>
__device__ __forceinline__ void ntt2_stages(Fp<PP>* x, TW tws, uint32_t low) {
  constexpr int M = 1 << ML;
  constexpr int LQ = LOG_R - S - ML;
  static_assert(!LAST || LQ == 0, "the last group ends the transform");
  constexpr Ntt2LastTw<PP, LOG_R, S, ML, INV> kLast{};
#pragma unroll
  for (int u = 0; u < ML; ++u) {
    const int half = M >> (u + 1);
#pragma unroll
    for (int jj = 0; jj < M / 2; ++jj) {
      if (jj < half) {
        uint32_t tw = 0;
        const bool one = LAST && jj == 0;
        if constexpr (!LAST) tw = tws[(low << (S + u)) + ((uint32_t)jj << (LQ + S + u))];
#pragma unroll
        for (int it = 0; it < 16 / M; ++it) {
#pragma unroll
          for (int blk = 0; blk < M; blk += 2 * half) {
            Fp<PP>& p = x[it * M + blk + jj];
            Fp<PP>& c = x[it * M + blk + jj + half];
            if (LAST) {
              if (one) ntt2_bfly_one<PP>(p, c);
              else ntt2_bfly<PP>(p, c, kLast.v[u][jj]);
            } else {
              ntt2_bfly<PP>(p, c, tw);
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ pass 2: contiguous lines, in place
// data: consecutive lines of R = 2^LOG_R cells; a tile is 2^13 consecutive cells (2^(13-LOG_R) lines).
// LDS position of cell r of line t: t*LINE + r + (r >> 4): one pad word per 16 cells, so that the
// last group (16 consecutive cells per lane) and the strided groups are both conflict-free.
template <int LOG_R>
__device__ __forceinline__ uint32_t ntt2_line_pos(uint32_t t, uint32_t r) {
  constexpr uint32_t LINE = (1u << LOG_R) + (1u << (LOG_R > 4 ? LOG_R - 4 : 0));
  return t * LINE + r + (r >> 4);
}

// one middle stage group (stages S..S+3) of the line pass, LDS to LDS
template <class PP, int LOG_R, int S>
__device__ __forceinline__ void ntt2_line_middle(uint32_t* tile, const uint32_t* tws, Fp<PP>* x, uint32_t tid) {
  using F = Fp<PP>;
  constexpr uint32_t R = 1u << LOG_R;
  constexpr int LQ = LOG_R - S - 4;
  const uint32_t it = tid & ((R / 16) - 1), t = tid >> (LOG_R - 4);
  const uint32_t low = it & ((1u << LQ) - 1), high = it >> LQ;
  const uint32_t r0 = (high << (LQ + 4)) | low;
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = F::raw(tile[ntt2_line_pos<LOG_R>(t, r0 + ((uint32_t)j << LQ))]);
  ntt2_stages<PP, LOG_R, S, 4, false>(x, tws, low);
#pragma unroll
  for (int j = 0; j < 16; ++j) tile[ntt2_line_pos<LOG_R>(t, r0 + ((uint32_t)j << LQ))] = x[j].v;
  __syncthreads();
}

struct NttLineJob {
  uint32_t* data;     // in place
  const uint32_t* tw; // w_R^i, i < R/2, Montgomery
  uint32_t block0;    // first block of this job; a job owns cells / tile blocks
  uint32_t log_r;     // line length (read by the mixed-length launch only)
};

// words of LDS a line tile needs (data with one pad word per 16 cells; the twiddle table is R/2 more)
constexpr uint32_t ntt2_line_tile_words(int log_r, int log_tile) {
  return ((1u << log_r) + (1u << (log_r - 4))) << (log_tile - log_r);
}

// the work of one workgroup: tile `local` of job `job` (tile / tws: LDS of at least
// ntt2_line_tile_words(LOG_R, LOG_TILE) / R/2 words)
template <class PP, int LOG_R, int LOG_TILE>
__device__ __forceinline__ void ntt2_line_body(const NttLineJob& job, uint32_t local, uint32_t* tile, uint32_t* tws) {
  using F = Fp<PP>;
  static_assert(LOG_R >= 5 && LOG_R <= LOG_TILE && LOG_TILE <= 13, "line length");
  constexpr uint32_t R = 1u << LOG_R;
  constexpr uint32_t LANES = 1u << (LOG_TILE - 4);
  constexpr int G = (LOG_R + 3) / 4;             // stage groups: 4, 4, ..., remainder last
  constexpr int ML_LAST = LOG_R - 4 * (G - 1);
  const gptr<uint32_t> data = as_global(job.data) + ((size_t)local << LOG_TILE);
  const gptr<const uint32_t> twg = as_global(job.tw);
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < R / 2; i += LANES) tws[i] = twg[i];
  F x[16];
  // ---- group 0 (stages 0..3) straight from global memory: item b of line t = cells b + j*(R/16)
  {
    constexpr int LQ = LOG_R - 4;  // q = R/16 items per line
    const uint32_t b = tid & ((1u << LQ) - 1), t = tid >> LQ;
    const gptr<uint32_t> src = data + (t << LOG_R) + b;
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = F::raw(src[(uint32_t)j << LQ]);
    __syncthreads();  // twiddle table
    ntt2_stages<PP, LOG_R, 0, 4, LOG_R == 4>(x, tws, b);
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[ntt2_line_pos<LOG_R>(t, b + ((uint32_t)j << LQ))] = x[j].v;
  }
  __syncthreads();
  // ---- middle groups (stages 4..7, and 8..11 for 2^13-cell lines)
  if constexpr (G >= 3) ntt2_line_middle<PP, LOG_R, 4>(tile, tws, x, tid);
  if constexpr (G >= 4) ntt2_line_middle<PP, LOG_R, 8>(tile, tws, x, tid);
  // ---- last group: 16 consecutive cells per lane, constant twiddles
  {
    constexpr int S = 4 * (G - 1);
    const uint32_t it = tid & ((R / 16) - 1), t = tid >> (LOG_R - 4);
    const uint32_t r0 = it << 4;
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = F::raw(tile[ntt2_line_pos<LOG_R>(t, r0 + j)]);
    ntt2_stages<PP, LOG_R, S, ML_LAST, true>(x, (const uint32_t*)nullptr, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[ntt2_line_pos<LOG_R>(t, r0 + j)] = x[j].v;
  }
  __syncthreads();
  // ---- copy out, lanes along the line
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t cell = tid + (uint32_t)j * LANES;
    data[cell] = tile[ntt2_line_pos<LOG_R>(cell >> LOG_R, cell & (R - 1))];
  }
}

template <class PP, int LOG_R, int LOG_TILE = kNtt2LogTile>
__global__ void __launch_bounds__(1 << (LOG_TILE - 4)) k_ntt_fwd_line(const NttLineJob* __restrict__ jobs, int n_jobs) {
  __shared__ uint32_t tile[ntt2_line_tile_words(LOG_R, LOG_TILE)];
  __shared__ uint32_t tws[(1u << LOG_R) / 2];  // (reading the table through L1 instead frees LDS for another tile per CU but measured 25 % slower)
  const int jb = find_job(jobs, n_jobs);
  ntt2_line_body<PP, LOG_R, LOG_TILE>(jobs[jb], blockIdx.x - jobs[jb].block0, tile, tws);
}

// Lines of different lengths (2^5 .. 2^12 cells, 2^12-cell tiles) in ONE launch: the tables of a small
// layer have four different heights, and at 2^14..2^16 rows a pass per height is a launch of a few
// dozen workgroups that costs more to start than to run.  Used below kNtt2MixedMaxBlocks workgroups.
template <class PP>
__global__ void __launch_bounds__(256) k_ntt_fwd_line_mixed(const NttLineJob* __restrict__ jobs, int n_jobs) {
  __shared__ uint32_t tile[ntt2_line_tile_words(12, 12)];  // the same for every line length
  __shared__ uint32_t tws[(1u << 12) / 2];
  const int jb = find_job(jobs, n_jobs);
  const NttLineJob& job = jobs[jb];
  const uint32_t local = blockIdx.x - job.block0;
  switch (job.log_r) {
    case 5: ntt2_line_body<PP, 5, 12>(job, local, tile, tws); break;
    case 6: ntt2_line_body<PP, 6, 12>(job, local, tile, tws); break;
    case 7: ntt2_line_body<PP, 7, 12>(job, local, tile, tws); break;
    case 8: ntt2_line_body<PP, 8, 12>(job, local, tile, tws); break;
    case 9: ntt2_line_body<PP, 9, 12>(job, local, tile, tws); break;
    case 10: ntt2_line_body<PP, 10, 12>(job, local, tile, tws); break;
    case 11: ntt2_line_body<PP, 11, 12>(job, local, tile, tws); break;
    default: ntt2_line_body<PP, 12, 12>(job, local, tile, tws); break;
  }
}

// ------------------------------------------------------------------ column passes (strided dimension)
// One column is viewed as [N1 rows][N2] (index n1*N2 + n2); a tile is [R = N1 rows][T = 2^13/R columns n2]
// and the size-R transform runs along the rows (DIF: tile row r ends up holding output k1 = bitrev(r)).
//   NTT2_FWD   forward pass 1 of the LDE, every coset z: input scaled by s_z^(N2*n1 + n2), output times
//              the four-step twiddle w_N^(k1*n2), stored at row r (bit-reversed order kept):
//              out[z][r*N2 + n2]
//   NTT2_INV1  inverse pass 1: output times w_N^-(k1*n2), stored TRANSPOSED in natural order,
//              out[n2*N1 + k1] (through LDS, so that lanes run along k1)
//   NTT2_INV2  inverse pass 2: output times `scale` (1/N), natural order, out[k1*N2 + n2]
enum { NTT2_FWD = 0, NTT2_INV1 = 1, NTT2_INV2 = 2 };
struct NttColJob {
  const uint32_t* in;
  uint32_t* out;
  const uint32_t* tw;      // w_R^(+-i), i < R/2
  const uint32_t* tw4_lo;  // w_N^(+-x) = hi[x >> 10] * lo[x & 1023]
  const uint32_t* tw4_hi;
  const uint32_t* pre_a;   // [cosets][N1]: s_z^(N2*n1)
  const uint32_t* pre_b;   // [cosets][N2]: s_z^n2
  uint64_t in_col_stride, out_col_stride, out_coset_stride;
  int log_n2, log_cosets;
  int log_r;        // sub-transform size (read by the mixed-size launch only)
  uint32_t scale;   // Montgomery (NTT2_INV2)
  uint32_t block0;  // tile fastest, then coset, then column - or, with xcd_map, see k_ntt_col
  uint32_t xcd_map; // the cosets of one tile on blocks b, b+8, b+16, ...: one XCD, one L2
};

// LOG_TILE: 13 (one 16-row item per lane and stage group) or 14 (two items per lane: twice the
// columns per tile, i.e. twice the contiguous bytes per row - what the inverse passes of tall
// matrices need, whose 2^10 / 2^11-row tiles are only 8 / 4 columns wide at 2^13 cells).
constexpr uint32_t ntt2_col_tile_words(int log_r, int log_tile) {
  return (1u << log_r) * ((1u << (log_tile - log_r)) + 1) + ((1u << log_r) >> 5) + 2;
}

// the work of one workgroup: tile `local` of job `a` (tile / tws: LDS of at least
// ntt2_col_tile_words(LOG_R, LOG_TILE) / R/2 words)
template <class PP, int LOG_R, int MODE, int LOG_TILE>
__device__ __forceinline__ void ntt2_col_body(const NttColJob& a, uint32_t local, uint32_t* tile, uint32_t* tws) {
  using F = Fp<PP>;
  static_assert(LOG_R >= 5 && LOG_R <= 12, "sub-transform size");
  static_assert(LOG_TILE == 13 || LOG_TILE == 14, "tile size");
  constexpr bool INV = MODE != NTT2_FWD;
  constexpr uint32_t R = 1u << LOG_R;
  constexpr int LOG_T = LOG_TILE - LOG_R;
  constexpr uint32_t T = 1u << LOG_T;
  constexpr int ITEMS = 1 << (LOG_TILE - kNtt2LogTile);
  constexpr int G = (LOG_R + 3) / 4;
  constexpr int ML_LAST = LOG_R - 4 * (G - 1);
  const int log_gx = a.log_n2 - LOG_T;
  uint32_t bx, bz, by;
  if (a.xcd_map) {
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each XCD has its
    // own L2): the 2^log_cosets cosets of a tile read the SAME coefficient tile, so they are placed on
    // blocks 8 apart - the tile comes from HBM once and from that XCD's L2 for the other cosets.
    const uint32_t xcd = local & 7, slot = local >> 3;
    bz = slot & ((1u << a.log_cosets) - 1);
    const uint32_t tile_lin = ((slot >> a.log_cosets) << 3) | xcd;
    bx = tile_lin & ((1u << log_gx) - 1);
    by = tile_lin >> log_gx;
  } else {
    bx = local & ((1u << log_gx) - 1);
    bz = (local >> log_gx) & ((1u << a.log_cosets) - 1);
    by = local >> (log_gx + a.log_cosets);
  }
  const uint32_t tid = threadIdx.x;
  const gptr<const uint32_t> twg = as_global(a.tw);
  for (uint32_t i = tid; i < R / 2; i += kNtt2Lanes) tws[i] = twg[i];
  F x[16];
  // ---- group 0 from global memory: rows it + j*(R/16)
#pragma unroll
  for (int item = 0; item < ITEMS; ++item) {
    const uint32_t e = tid + (uint32_t)item * kNtt2Lanes;
    const uint32_t t = e & (T - 1), it = e >> LOG_T;  // column inside the tile, 16-row item
    const uint32_t n2 = (bx << LOG_T) + t;
    constexpr int LQ = LOG_R - 4;
    const gptr<const uint32_t> src = as_global(a.in) + (size_t)by * a.in_col_stride + n2;
    if constexpr (MODE == NTT2_FWD) {
      // scaled by the row part s_z^(N2*n1) of the coset shift power; the column part s_z^n2 is the same
      // for the whole size-R transform of this column and rides in the output twiddle
      const gptr<const uint32_t> pa = as_global(a.pre_a) + ((size_t)bz << LOG_R);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const uint32_t n1 = it + ((uint32_t)j << LQ);
        x[j] = F::raw(src[(size_t)n1 << a.log_n2]) * F::raw(pa[n1]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = F::raw(src[(size_t)(it + ((uint32_t)j << LQ)) << a.log_n2]);
    }
    if (item == 0) __syncthreads();  // twiddle table
    ntt2_stages<PP, LOG_R, 0, 4, false, INV>(x, tws, it);
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[lds_addr(it + ((uint32_t)j << LQ), t, T)] = x[j].v;
  }
  __syncthreads();
  if constexpr (G >= 3) {
#pragma unroll
    for (int item = 0; item < ITEMS; ++item) {
      const uint32_t e = tid + (uint32_t)item * kNtt2Lanes;
      const uint32_t t = e & (T - 1), it = e >> LOG_T;
      constexpr int S = 4;
      constexpr int LQ = LOG_R - S - 4;
      const uint32_t low = it & ((1u << LQ) - 1), high = it >> LQ;
      const uint32_t r0 = (high << (LQ + 4)) | low;
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = F::raw(tile[lds_addr(r0 + ((uint32_t)j << LQ), t, T)]);
      ntt2_stages<PP, LOG_R, S, 4, false, INV>(x, tws, low);
#pragma unroll
      for (int j = 0; j < 16; ++j) tile[lds_addr(r0 + ((uint32_t)j << LQ), t, T)] = x[j].v;
    }
    __syncthreads();
  }
  // ---- last group: rows 16*it .. 16*it+15
  constexpr int S_LAST = 4 * (G - 1);
  const gptr<uint32_t> dst = as_global(a.out) + (size_t)by * a.out_col_stride + (size_t)bz * a.out_coset_stride;
#pragma unroll
  for (int item = 0; item < ITEMS; ++item) {
    const uint32_t e = tid + (uint32_t)item * kNtt2Lanes;
    const uint32_t t = e & (T - 1), it = e >> LOG_T;
    const uint32_t n2 = (bx << LOG_T) + t;
    const uint32_t r0 = it << 4;
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = F::raw(tile[lds_addr(r0 + j, t, T)]);
    ntt2_stages<PP, LOG_R, S_LAST, ML_LAST, true, INV>(x, (const uint32_t*)nullptr, 0);
    if constexpr (MODE == NTT2_FWD) {
      // straight to global memory with the four-step twiddle w_N^(k1*n2), rows in place.  Row r0 + j
      // holds k1 = bitrev(r0 + j) = bj * R/16 + K0 with bj = bitrev4(j), K0 = bitrev(it): the sixteen
      // twiddles of a lane are A * g^bj, A = s_z^n2 * w_N^(K0*n2), g = w_N^(n2*R/16) - two table lookups
      // and a product chain per lane instead of two gathers per cell (four chains of four, step g^4).
      const gptr<const uint32_t> lo = as_global(a.tw4_lo), hi = as_global(a.tw4_hi);
      auto w_pow = [&](uint32_t e) {  // w_N^e, canonical Montgomery
        const uint32_t v = F::reduce64_lazy((uint64_t)hi[e >> 10] * lo[e & 1023]);
        return F::raw(min(v, v - PP::P));
      };
      const F pb = F::raw(as_global(a.pre_b)[((size_t)bz << a.log_n2) + n2]);
      const F g = w_pow(n2 << (LOG_R - 4));
      const F g2 = g * g, g4 = g2 * g2;
      F head[4];
      head[0] = w_pow(bit_reverse(it, LOG_R - 4) * n2) * pb;
      head[1] = head[0] * g;
      head[2] = head[0] * g2;
      head[3] = head[1] * g2;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          constexpr int kRev4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
          const int j = kRev4[4 * m + q];  // bitrev4 is an involution: bj = 4m + q sits in row r0 + bitrev4(bj)
          dst[((size_t)(r0 + j) << a.log_n2) + n2] = (x[j] * head[q]).v;
          if (m < 3) head[q] = head[q] * g4;
        }
      }
    } else if constexpr (MODE == NTT2_INV2) {
      // natural row order, scaled: row k1 = bitrev(r)
      const F sc = F::raw(a.scale);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const uint32_t k1 = bit_reverse(r0 + j, LOG_R);
        dst[((size_t)k1 << a.log_n2) + n2] = (x[j] * sc).v;
      }
    } else {
      // transposed: back through LDS so that consecutive lanes write consecutive k1
#pragma unroll
      for (int j = 0; j < 16; ++j) tile[lds_addr(r0 + j, t, T)] = x[j].v;
    }
  }
  if constexpr (MODE == NTT2_INV1) {
    __syncthreads();
    const gptr<const uint32_t> lo = as_global(a.tw4_lo), hi = as_global(a.tw4_hi);
#pragma unroll
    for (int j = 0; j < 16 * ITEMS; ++j) {
      const uint32_t e = tid + (uint32_t)j * kNtt2Lanes;
      const uint32_t k1 = e & (R - 1), tt = e >> LOG_R;
      const uint32_t col = (bx << LOG_T) + tt;
      const uint32_t xk = k1 * col;
      const F tw = F::raw(F::reduce64_lazy((uint64_t)hi[xk >> 10] * lo[xk & 1023]));
      const F v = F::raw(tile[lds_addr(bit_reverse(k1, LOG_R), tt, T)]);
      dst[((size_t)col << LOG_R) + k1] = (v * tw).v;
    }
  }
}

template <class PP, int LOG_R, int MODE, int LOG_TILE = kNtt2LogTile>
__global__ void __launch_bounds__(kNtt2Lanes, LOG_TILE == 13 ? 8 : 4) k_ntt_col(const NttColJob* __restrict__ jobs, int n_jobs) {
  __shared__ uint32_t tile[ntt2_col_tile_words(LOG_R, LOG_TILE)];
  __shared__ uint32_t tws[(1u << LOG_R) / 2];
  const int jb = find_job(jobs, n_jobs);
  ntt2_col_body<PP, LOG_R, MODE, LOG_TILE>(jobs[jb], blockIdx.x - jobs[jb].block0, tile, tws);
}

// Column passes of different sub-transform sizes (2^13-cell tiles) in ONE launch, for small layers
// (see k_ntt_fwd_line_mixed).
constexpr uint32_t kNtt2MixedMaxBlocks = 2048;
template <class PP, int MODE>
__global__ void __launch_bounds__(kNtt2Lanes) k_ntt_col_mixed(const NttColJob* __restrict__ jobs, int n_jobs) {
  __shared__ uint32_t tile[ntt2_col_tile_words(12, 13)];  // the largest: 2^12 rows x 2 columns
  __shared__ uint32_t tws[(1u << 12) / 2];
  const int jb = find_job(jobs, n_jobs);
  const NttColJob& a = jobs[jb];
  const uint32_t local = blockIdx.x - a.block0;
  switch (a.log_r) {
    case 5: ntt2_col_body<PP, 5, MODE, 13>(a, local, tile, tws); break;
    case 6: ntt2_col_body<PP, 6, MODE, 13>(a, local, tile, tws); break;
    case 7: ntt2_col_body<PP, 7, MODE, 13>(a, local, tile, tws); break;
    case 8: ntt2_col_body<PP, 8, MODE, 13>(a, local, tile, tws); break;
    case 9: ntt2_col_body<PP, 9, MODE, 13>(a, local, tile, tws); break;
    case 10: ntt2_col_body<PP, 10, MODE, 13>(a, local, tile, tws); break;
    case 11: ntt2_col_body<PP, 11, MODE, 13>(a, local, tile, tws); break;
    default: ntt2_col_body<PP, 12, MODE, 13>(a, local, tile, tws); break;
  }
}

}  // namespace p3r
