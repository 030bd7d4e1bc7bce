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
  const int jb = find_job(jobs, n_jobs);
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
  const int jb = find_job(jobs, n_jobs);
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
  const int jb = find_job(jobs, n_jobs);
  const ZkSaltJob& j = jobs[jb];
  const uint64_t tiles_per_col = (j.h + kBlock - 1) / kBlock;
  const uint64_t t = blockIdx.x - j.block0;
  const uint32_t c = (uint32_t)(t / tiles_per_col);
  const uint64_t r = (t % tiles_per_col) * kBlock + threadIdx.x;
  if (r >= j.h) return;
  as_global(j.dst)[((uint64_t)c * j.h + r) * j.stride] = zk_rand_mont<PP>(key, j.stream, r * j.S + c);
}


// ---- the same fills, one ChaCha block per EIGHT cells (round 6) ---------------------------------------------------------
// k_zk_randomize / k_zk_salts compute cell (r, c) in the lane that stores it - lanes run along a column, the cells of a
// ChaCha block along a row - so every random cell pays for a whole block (~ 400 instructions, 5.7 ms of randomisation and
// 9.0 ms of salts per 2^20-row proof).  Here a workgroup owns a tile of 2^log_tr rows x all columns: each lane computes
// whole blocks, draws their eight cells by static index and parks them in LDS by (row, column); the tile is then written
// out column by column, lanes along the rows.  Same values, same bytes (zk_rand_canonical remains the definition and the
// slow path of a cell whose two candidates are refused).
#ifndef P3R_ZK_TILE_WORDS
#define P3R_ZK_TILE_WORDS 6144   // 24 KB of LDS per workgroup: six workgroups per CU (tools/microbench/zk_fill_check.hip: 48 KB tiles
                                // take 1.10 / 0.74 ms for 2^22 x 64 / 2^20 x 168 cells, 24 KB 0.75 / 0.52, 16 KB 0.82 / 0.64)
#endif
constexpr uint32_t kZkTileWords = P3R_ZK_TILE_WORDS;
struct ZkTileJob {
  const uint32_t* src;   // mode 0: the matrix being randomised, [w][rows / 2]
  uint32_t* dst;
  uint64_t rows;         // rows of the index space: 2h (randomise), h (salts)
  uint32_t w, w2;        // mode 0: original / randomised width; otherwise 0 / number of columns
  uint32_t mode;         // 0: even rows keep src for c < w, everything else random; 1: every cell random
                         // (the zero fill of the preprocessed round draws nothing: it stays with k_zk_randomize)
  uint32_t stride;       // element stride of dst inside a column (1: plain column-major; FRI commit-phase salts: the arity)
  uint32_t stream;
  uint32_t log_tr;       // rows per tile
  uint32_t block0;
};
// rows per tile: as many as the LDS budget holds at this width (row pitch odd: the write-out reads a column of the tile)
inline uint32_t zk_tile_log_rows(uint64_t rows, uint32_t w2) {
  const uint32_t pitch = w2 | 1u;
  uint32_t log_tr = rows >= 2 ? 1 : 0;
  while (log_tr >= 1 && log_tr < 11 && (uint64_t(2) << log_tr) * pitch <= kZkTileWords && (uint64_t(2) << log_tr) <= rows) ++log_tr;
  return log_tr;
}
// the eight cells of block b of a stream that fall in [lo, hi): emit(idx, Montgomery value).  A cell whose two candidates
// are refused (2^-14 KoalaBear, 2^-8 BabyBear) is drawn again by the definition, in ONE loop behind the eight (a call in
// each of them would be inlined eight times, or cost every lane the calling convention's spills)
template <class PP, class Emit>
__device__ __forceinline__ void zk_block_cells(const ZkKey& key, uint32_t stream, uint64_t b, uint64_t lo, uint64_t hi, Emit&& emit) {
  uint32_t blk[16];
  zk_chacha8_block(key.k, (uint32_t)b, stream, key.nonce_lo, key.nonce_hi, blk);
  uint32_t refused = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint64_t idx = 8 * b + k;
    if (idx < lo || idx >= hi) continue;
    uint32_t v = blk[2 * k] & 0x7FFFFFFFu;
    if (v >= PP::P) v = blk[2 * k + 1] & 0x7FFFFFFFu;
    if (v >= PP::P) { refused |= 1u << k; continue; }
    emit(idx, Fp<PP>::from_canonical(v).v);
  }
  while (refused) {
    const int k = __ffs(refused) - 1;
    refused &= refused - 1;
    emit(8 * b + k, Fp<PP>::from_canonical(zk_rand_canonical<PP>(key, stream, 8 * b + k)).v);
  }
}
template <class PP>
__global__ void __launch_bounds__(kBlock) k_zk_fill_tiles(const ZkTileJob* __restrict__ jobs, int n_jobs, ZkKey key) {
  __shared__ uint32_t tile[kZkTileWords];
  const int jb = find_job(jobs, n_jobs);
  const ZkTileJob& j = jobs[jb];
  const uint32_t w2 = j.w2, pitch = w2 | 1u, tr = 1u << j.log_tr, tid = threadIdx.x;
  const uint64_t r0 = (uint64_t)(blockIdx.x - j.block0) << j.log_tr;
  if (j.mode == 1) {
    // one contiguous range of the stream
    const uint64_t lo = r0 * w2, hi = lo + (uint64_t)tr * w2;
    for (uint64_t b = (lo >> 3) + tid; b <= ((hi - 1) >> 3); b += kBlock)
      zk_block_cells<PP>(key, j.stream, b, lo, hi, [&](uint64_t idx, uint32_t v) {
        const uint32_t off = (uint32_t)(idx - lo), rr = off / w2;
        tile[rr * pitch + (off - rr * w2)] = v;
      });
  } else {
    // odd rows: all w2 cells; even rows: the codeword columns [w, w2).  Items (block of the row, row), rows fastest:
    // the lanes of a wave work on the same block position of neighbouring rows
    const uint32_t half = tr >> 1;
    for (uint32_t par = 0; par < 2; ++par) {
      const uint32_t c_lo = par ? j.w : 0u;
      if (c_lo >= w2) continue;
      const uint32_t kb_n = (w2 - c_lo + 7) / 8 + 1;
      for (uint32_t it = tid; it < kb_n * half; it += kBlock) {
        const uint32_t kb = it / half, q = it - kb * half, rr = 2 * q + (par ? 0u : 1u);
        const uint64_t lo = (r0 + rr) * w2 + c_lo, hi = (r0 + rr + 1) * w2, b = (lo >> 3) + kb;
        if (8 * b >= hi) continue;
        zk_block_cells<PP>(key, j.stream, b, lo, hi, [&](uint64_t idx, uint32_t v) { tile[rr * pitch + c_lo + (uint32_t)(idx - lo)] = v; });
      }
    }
  }
  __syncthreads();
  // write-out: column by column, lanes along the rows of the tile
  const uint64_t total = (uint64_t)w2 << j.log_tr;
  for (uint64_t i = tid; i < total; i += kBlock) {
    const uint32_t c = (uint32_t)(i >> j.log_tr), rr = (uint32_t)i & (tr - 1);
    const uint64_t r = r0 + rr;
    uint32_t v;
    if (j.mode == 0 && !(r & 1) && c < j.w) v = as_global(j.src)[(uint64_t)c * (j.rows >> 1) + (r >> 1)];
    else v = tile[rr * pitch + c];
    as_global(j.dst)[((uint64_t)c * j.rows + r) * j.stride] = v;
  }
}

}  // namespace p3r
