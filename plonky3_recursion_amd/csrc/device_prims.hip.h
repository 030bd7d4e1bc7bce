// Exclusive sums, a maximum and a stable radix sort over device arrays: the data-parallel primitives of the preparation
// pass (prep_device.hip), written for gfx950's 64-wide wavefronts.  (They were hipCUB calls until round 5.)
//
//   exclusive_sum   three launches: per-tile sums, a one-block scan of the tile sums, per-tile scan with the tile's offset.
//                   The input is a functor of the index (flags read off the op list), so reading it twice costs less than
//                   a materialised flag array would.
//   reduce_max      wave maximum by cross-lane shuffles, one atomicMax per wave.
//   sort_pairs      least-significant-digit radix sort, 8 bits per pass, stable: per-block digit histograms, one exclusive
//                   sum over the [digit][block] counts, then a scatter in which every wave walks its keys in order and
//                   ranks the 64 keys of a step against each other with eight ballots (the lanes that share a digit:
//                   AND of the eight bit votes), so equal digits keep their input order without a single atomic.
#pragma once
#include "context.h"

namespace p3r {
namespace prims {

constexpr int kPB = 256;                        // threads per block: four wavefronts
constexpr int kWaves = kPB / 64;
constexpr int kScanItems = 8;                   // consecutive items per thread
constexpr int kScanTile = kPB * kScanItems;
constexpr int kRadixBits = 8, kRadix = 1 << kRadixBits;
constexpr int kSortItems = 16;                  // keys per lane
constexpr int kSortTile = kPB * kSortItems;     // keys per block; a wave owns a contiguous quarter of it
static_assert(kRadix == kPB, "one thread per digit when the histograms are written and rebased");

template <class T>
__device__ __forceinline__ T wave_inclusive_sum(T v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T u = __shfl_up(v, d, 64);
    if (lane >= d) v += u;
  }
  return v;
}
// exclusive prefix of `v` over the block's threads and the block's total; `sh`: kWaves cells of LDS
template <class T>
__device__ __forceinline__ T block_exclusive_sum(T v, T* sh, T& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const T inc = wave_inclusive_sum(v, lane);
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  T base = 0, all = 0;
#pragma unroll
  for (int k = 0; k < kWaves; ++k) {
    const T c = sh[k];
    if (k < w) base += c;
    all += c;
  }
  total = all;
  __syncthreads();   // `sh` is free again when this returns
  return base + inc - v;
}

template <class T, class Fn>
__global__ void __launch_bounds__(kPB) k_tile_sums(Fn fn, size_t n, T* __restrict__ sums) {
  __shared__ T sh[kWaves];
  const size_t i0 = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  T v = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (i0 + j < n) v += (T)fn(i0 + j);
  T total;
  (void)block_exclusive_sum(v, sh, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one block: sums[0 .. n) -> their exclusive prefix sums, in place
template <class T>
__global__ void __launch_bounds__(kPB) k_scan_sums(T* __restrict__ sums, size_t n) {
  __shared__ T sh[kWaves];
  T carry = 0;
  for (size_t c0 = 0; c0 < n; c0 += kPB) {
    const size_t i = c0 + threadIdx.x;
    const T v = i < n ? sums[i] : 0;
    T total;
    const T ex = block_exclusive_sum(v, sh, total);
    if (i < n) sums[i] = carry + ex;
    carry += total;
  }
}
template <class T, class Fn>
__global__ void __launch_bounds__(kPB) k_tile_scan(Fn fn, size_t n, const T* __restrict__ sums, T* __restrict__ out) {
  __shared__ T sh[kWaves];
  const size_t i0 = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  T item[kScanItems];
  T v = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    item[j] = i0 + j < n ? (T)fn(i0 + j) : 0;
    v += item[j];
  }
  T total;
  T run = sums[blockIdx.x] + block_exclusive_sum(v, sh, total);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    if (i0 + j < n) out[i0 + j] = run;
    run += item[j];
  }
}

// out[i] = fn(0) + .. + fn(i - 1), i < n.  `out` may be the array `fn` reads (every thread holds its items before it writes).
template <class T, class Fn>
void exclusive_sum(hipStream_t s, Fn fn, size_t n, T* out) {
  if (!n) return;
  const size_t tiles = (n + kScanTile - 1) / kScanTile;
  DevBuf sums(tiles * (sizeof(T) / 4));
  T* ps = reinterpret_cast<T*>(sums.p);
  hipLaunchKernelGGL((k_tile_sums<T, Fn>), dim3((unsigned)tiles), dim3(kPB), 0, s, fn, n, ps);
  hipLaunchKernelGGL((k_scan_sums<T>), dim3(1), dim3(kPB), 0, s, ps, tiles);
  hipLaunchKernelGGL((k_tile_scan<T, Fn>), dim3((unsigned)tiles), dim3(kPB), 0, s, fn, n, ps, out);
  P3R_HIP(hipGetLastError());
}
struct LoadU32 {
  const uint32_t* p;
  __device__ uint32_t operator()(size_t i) const { return p[i]; }
};

template <class Fn>
__global__ void __launch_bounds__(kPB) k_reduce_max(Fn fn, size_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * kPB + threadIdx.x; i < n; i += (size_t)gridDim.x * kPB) m = max(m, (uint32_t)fn(i));
#pragma unroll
  for (int d = 32; d; d >>= 1) m = max(m, (uint32_t)__shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// *out = max(0, fn(0), .., fn(n - 1))
template <class Fn>
void reduce_max(hipStream_t s, Fn fn, size_t n, uint32_t* out) {
  P3R_HIP(hipMemsetAsync(out, 0, 4, s));
  if (!n) return;
  const unsigned blocks = (unsigned)std::min<size_t>((n + kPB - 1) / kPB, 2048);
  hipLaunchKernelGGL((k_reduce_max<Fn>), dim3(blocks), dim3(kPB), 0, s, fn, n, out);
  P3R_HIP(hipGetLastError());
}

// counts[digit][block]
static __global__ void __launch_bounds__(kPB) k_sort_hist(const uint32_t* __restrict__ keys, size_t n, int shift, uint32_t mask,
                                                          uint32_t* __restrict__ counts, uint32_t n_blocks) {
  __shared__ uint32_t h[kRadix];
  h[threadIdx.x] = 0;
  __syncthreads();
  const size_t t0 = (size_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const size_t i = t0 + (size_t)j * kPB + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  counts[(size_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}
// offs: the exclusive sums of counts, i.e. where the first key of (digit, block) goes
static __global__ void __launch_bounds__(kPB) k_sort_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n, int shift,
                                                             uint32_t mask, const uint32_t* __restrict__ offs, uint32_t n_blocks) {
  __shared__ uint32_t wh[kWaves][kRadix];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < kWaves; ++k) wh[k][threadIdx.x] = 0;
  __syncthreads();
  const size_t w0 = (size_t)blockIdx.x * kSortTile + (size_t)w * (kSortTile / kWaves);
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const size_t i = w0 + (size_t)j * 64 + lane;
    if (i < n) atomicAdd(&wh[w][(keys_in[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  {  // counts -> where each wave's first key of digit `threadIdx.x` goes
    uint32_t base = offs[(size_t)threadIdx.x * n_blocks + blockIdx.x];
#pragma unroll
    for (int k = 0; k < kWaves; ++k) {
      const uint32_t c = wh[k][threadIdx.x];
      wh[k][threadIdx.x] = base;
      base += c;
    }
  }
  __syncthreads();
  volatile uint32_t* next = wh[w];   // other lanes of the wave advance it between this lane's reads
  for (int j = 0; j < kSortItems; ++j) {
    const size_t i = w0 + (size_t)j * 64 + lane;
    const bool act = i < n;
    const uint32_t key = act ? keys_in[i] : 0u, val = act ? vals_in[i] : 0u;
    const uint32_t d = (key >> shift) & mask;
    uint64_t peers = __ballot(act);   // the active lanes of this step whose digit is d
#pragma unroll
    for (int b = 0; b < kRadixBits; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t vote = __ballot(bit);
      peers &= bit ? vote : ~vote;
    }
    const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull)), cnt = (uint32_t)__popcll(peers);
    const uint32_t base = act ? next[d] : 0u;
    // one wave, one instruction stream: every lane's read above is issued before the write below
    if (act && rank == cnt - 1) next[d] = base + cnt;
    if (act) {
      keys_out[base + rank] = key;
      vals_out[base + rank] = val;
    }
  }
}

// (keys_out, vals_out) = (keys_in, vals_in) sorted by the low `bits` bits of the key, equal keys in input order.
// The inputs are left as they were; in and out must not overlap.
inline void sort_pairs(hipStream_t s, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                       int bits) {
  if (!n) return;
  if (bits < 1 || bits > 32) fail(P3R_EINVAL, "sort_pairs: 1 <= bits <= 32");
  const int passes = (bits + kRadixBits - 1) / kRadixBits;
  const uint32_t n_blocks = (uint32_t)((n + kSortTile - 1) / kSortTile);
  DevBuf counts((size_t)kRadix * n_blocks), tk, tv;
  if (passes > 1) {
    tk.alloc(n);
    tv.alloc(n);
  }
  const uint32_t *src_k = keys_in, *src_v = vals_in;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;   // the last pass lands in the caller's arrays
    uint32_t *dst_k = to_out ? keys_out : tk.p, *dst_v = to_out ? vals_out : tv.p;
    const int shift = p * kRadixBits;
    const uint32_t mask = (1u << std::min(kRadixBits, bits - shift)) - 1u;   // key bits at and above `bits` are not part of the order
    hipLaunchKernelGGL(k_sort_hist, dim3(n_blocks), dim3(kPB), 0, s, src_k, n, shift, mask, counts.p, n_blocks);
    exclusive_sum<uint32_t>(s, LoadU32{counts.p}, (size_t)kRadix * n_blocks, counts.p);
    hipLaunchKernelGGL(k_sort_scatter, dim3(n_blocks), dim3(kPB), 0, s, src_k, src_v, dst_k, dst_v, n, shift, mask, counts.p, n_blocks);
    src_k = dst_k;
    src_v = dst_v;
  }
  P3R_HIP(hipGetLastError());
}

}  // namespace prims
}  // namespace p3r
