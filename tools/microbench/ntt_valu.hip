// Is the forward line pass of the LDE (k_ntt_fwd_line, csrc/kernels_ntt2.hip.h) bound by the arithmetic it issues, or by
// what surrounds it (LDS exchange, barriers, the load / store phases of a workgroup)?
//   line      the shipped kernel: 2^12-cell lines, three stage groups, two LDS exchanges + the copy-out exchange
//   regs      the same butterflies and twiddle reads (twelve stages on the sixteen registers of a lane, the twiddle
//             table in LDS), WITHOUT the data exchanges and their barriers: the arithmetic alone (its output is not a
//             transform - only the time matters)
//   regs_f64  the butterflies of `regs` in FP64 (seven operations: add, sub, and a five-operation product against
//             (w, w / P)) with u32 <-> f64 conversions at the ends
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../plonky3_recursion_amd/csrc -I../../include ntt_valu.hip -o ntt_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels_ntt2.hip.h"
#include "poseidon2_f64.hip.h"
using namespace p3r;
using PP = KoalaBearParams;
using F = Fp<PP>;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int LOG_R = 12;
__global__ void __launch_bounds__(256) k_regs(uint32_t* __restrict__ data, const uint32_t* __restrict__ tw) {
  __shared__ uint32_t tws[(1u << LOG_R) / 2];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < (1u << LOG_R) / 2; i += 256) tws[i] = tw[i];
  uint32_t* d = data + ((size_t)blockIdx.x << LOG_R);
  F x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = F::raw(d[tid + 256 * j]);
  __syncthreads();
  ntt2_stages<PP, LOG_R, 0, 4, false>(x, tws, tid);
  ntt2_stages<PP, LOG_R, 4, 4, false>(x, tws, tid & 15);
  ntt2_stages<PP, LOG_R, 8, 4, true>(x, (const uint32_t*)nullptr, 0);
#pragma unroll
  for (int j = 0; j < 16; ++j) d[tid + 256 * j] = x[j].v;
}

// FP64 butterfly: (p, c) <- (p + c, (p - c) * w); no reduction on the sum path (12 stages add 12 bits to 2^31)
__device__ __forceinline__ void bfly_f64(double& p, double& c, double w, double wq) {
  const double s = p + c, d = p - c;
  const double q = __builtin_fma(d, wq, P2F64<PP>::MAGIC) - P2F64<PP>::MAGIC;
  const double t = q * P2F64<PP>::P_HI;
  const double e = __builtin_fma(d, w, -t);
  c = e - q;
  p = s;
}
__global__ void __launch_bounds__(256) k_regs_f64(uint32_t* __restrict__ data, const double2* __restrict__ tw) {
  __shared__ double2 tws[(1u << LOG_R) / 2];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < (1u << LOG_R) / 2; i += 256) tws[i] = tw[i];
  uint32_t* d = data + ((size_t)blockIdx.x << LOG_R);
  double x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = (double)d[tid + 256 * j];
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const uint32_t low = g == 0 ? tid : g == 1 ? (tid & 15) : 0;
    const int S = 4 * g, LQ = LOG_R - S - 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int half = 8 >> u;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (jj < half) {
          const double2 w = tws[((low << (S + u)) + ((uint32_t)jj << (LQ + S + u))) & ((1u << (LOG_R - 1)) - 1)];
#pragma unroll
          for (int blk = 0; blk < 16; blk += 2 * half) bfly_f64(x[blk + jj], x[blk + jj + half], w.x, w.y);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double r = p2f_reduce<PP>(x[j]);
    const int32_t v = (int32_t)r;
    d[tid + 256 * j] = (uint32_t)(v + ((v >> 31) & (int32_t)PP::P));
  }
}


// peak rate of the butterfly itself: sixteen registers, no memory in the loop; twiddles in registers (TWREG) or
// read from the LDS table as the passes do
template <bool TWREG>
__global__ void __launch_bounds__(256) k_peak(uint32_t* __restrict__ data, const uint32_t* __restrict__ tw, int iters) {
  __shared__ uint32_t tws[(1u << LOG_R) / 2];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < (1u << LOG_R) / 2; i += 256) tws[i] = tw[i];
  F x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = F::raw(data[tid + 256 * j]);
  __syncthreads();
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = tws[tid + 256 * j];
  for (int it = 0; it < iters; ++it) {
    if (TWREG) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int half = 8 >> u;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
          if (jj < half)
#pragma unroll
            for (int blk = 0; blk < 16; blk += 2 * half) ntt2_bfly<PP>(x[blk + jj], x[blk + jj + half], w[(jj + u) & 7]);
      }
    } else {
      ntt2_stages<PP, LOG_R, 0, 4, false>(x, tws, (tid + it) & 255);
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) data[tid + 256 * j] = x[j].v;
}
// `regs` with the arithmetic done REP times per load: if the time scales with REP the pass is bound by its arithmetic
template <int REP>
__global__ void __launch_bounds__(256) k_regs_rep(uint32_t* __restrict__ data, const uint32_t* __restrict__ tw) {
  __shared__ uint32_t tws[(1u << LOG_R) / 2];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < (1u << LOG_R) / 2; i += 256) tws[i] = tw[i];
  uint32_t* d = data + ((size_t)blockIdx.x << LOG_R);
  F x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = F::raw(d[tid + 256 * j]);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    ntt2_stages<PP, LOG_R, 0, 4, false>(x, tws, tid);
    ntt2_stages<PP, LOG_R, 4, 4, false>(x, tws, tid & 15);
    ntt2_stages<PP, LOG_R, 8, 4, true>(x, (const uint32_t*)nullptr, 0);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) d[tid + 256 * j] = x[j].v;
}
__global__ void __launch_bounds__(256) k_copy(uint32_t* __restrict__ data) {
  uint32_t* d = data + ((size_t)blockIdx.x << LOG_R);
  uint32_t x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = d[threadIdx.x + 256 * j];
#pragma unroll
  for (int j = 0; j < 16; ++j) d[threadIdx.x + 256 * j] = x[j] + 1;
}

// `regs` as PERSISTENT workgroups whose NEXT tile streams from HBM straight into LDS (global_load_lds_dwordx4: no
// destination registers, so the bytes in flight are not bounded by the register file) while the current tile is
// transformed: does decoupling the loads from the lanes close the gap between a pass and max(arithmetic, copy)?
//   wait vmcnt(0) (the tile issued one iteration ago + the previous stores) -> barrier -> issue the next tile into the
//   other buffer -> read this lane's 16 cells from LDS -> barrier -> 12 stages -> 16 stores
template <int SKEW>
__global__ void __launch_bounds__(256) k_regs_glds(uint32_t* __restrict__ data, const uint32_t* __restrict__ tw, uint32_t n_tiles) {
  // SKEW: the persistent workgroups of a CU start a fraction of a tile period apart (blockIdx / 256 is the slot on
  // the CU under round-robin placement), so that they do not all load, then all compute, then all store together
  if (SKEW) {
    const int slot = blockIdx.x / 256;
    for (int k = 0; k < slot; ++k) __builtin_amdgcn_s_sleep(SKEW);
  }
  __shared__ uint32_t tws[(1u << LOG_R) / 2];
  __shared__ __attribute__((aligned(16))) uint32_t buf[2][1u << LOG_R];
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  for (uint32_t i = tid; i < (1u << LOG_R) / 2; i += 256) tws[i] = tw[i];
  auto issue = [&](uint32_t tile, int b) {
    const uint32_t* src = data + ((size_t)tile << LOG_R);
#pragma unroll
    for (int j = 0; j < 4; ++j)   // 256 lanes x 16 B = 4 KB per step; a wave's 64 lanes land in 1 KB of consecutive LDS
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (j * 256 + tid) * 4),
                                       (__attribute__((address_space(3))) void*)(&buf[b][(j * 256 + wave * 64) * 4]), 16, 0, 0);
  };
  uint32_t tile = blockIdx.x;
  if (tile < n_tiles) issue(tile, 0);
  int cur = 0;
  for (; tile < n_tiles; tile += gridDim.x, cur ^= 1) {
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this tile is in LDS
    __builtin_amdgcn_s_barrier();
    if (tile + gridDim.x < n_tiles) issue(tile + gridDim.x, cur ^ 1);
    F x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = F::raw(buf[cur][tid + 256 * j]);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the reads above are done before anyone refills this buffer
    __builtin_amdgcn_s_barrier();
    ntt2_stages<PP, LOG_R, 0, 4, false>(x, tws, tid);
    ntt2_stages<PP, LOG_R, 4, 4, false>(x, tws, tid & 15);
    ntt2_stages<PP, LOG_R, 8, 4, true>(x, (const uint32_t*)nullptr, 0);
    uint32_t* d = data + ((size_t)tile << LOG_R);
#pragma unroll
    for (int j = 0; j < 16; ++j) d[tid + 256 * j] = x[j].v;
  }
}

template <class Fn>
float time_ms(Fn&& launch, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int log_cells = argc > 1 ? atoi(argv[1]) : 28;  // 2^20 rows x 64 columns x 4 cosets
  const size_t cells = size_t(1) << log_cells;
  uint32_t* data; CK(hipMalloc(&data, cells * 4));
  std::vector<uint32_t> h(cells);
  for (size_t i = 0; i < cells; ++i) h[i] = (uint32_t)((i * 2654435761ull) % PP::P);
  CK(hipMemcpy(data, h.data(), cells * 4, hipMemcpyHostToDevice));
  std::vector<uint32_t> tw(1u << (LOG_R - 1));
  std::vector<double2> twd(1u << (LOG_R - 1));
  F root = F::two_adic_generator(LOG_R), x = F::one();
  for (size_t i = 0; i < tw.size(); ++i) { tw[i] = x.v; twd[i] = double2{(double)x.to_canonical(), (double)x.to_canonical() / (double)PP::P}; x *= root; }
  uint32_t* dtw; CK(hipMalloc(&dtw, tw.size() * 4)); CK(hipMemcpy(dtw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
  double2* dtwd; CK(hipMalloc(&dtwd, twd.size() * 16)); CK(hipMemcpy(dtwd, twd.data(), twd.size() * 16, hipMemcpyHostToDevice));
  NttLineJob job{data, dtw, 0, LOG_R};
  NttLineJob* djob; CK(hipMalloc(&djob, sizeof job)); CK(hipMemcpy(djob, &job, sizeof job, hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)(cells >> LOG_R);
  const double bf = (double)cells * LOG_R / 2;
  float t;
  t = time_ms([&] { hipLaunchKernelGGL((k_ntt_fwd_line<PP, LOG_R, 12>), dim3(blocks), dim3(256), 0, 0, djob, 1); });
  printf("line      %.3f ms  %.2f T butterflies/s  %.0f GB/s\n", t, bf / t / 1e9, cells * 8.0 / t / 1e6);
  t = time_ms([&] { hipLaunchKernelGGL(k_regs, dim3(blocks), dim3(256), 0, 0, data, dtw); });
  printf("regs      %.3f ms  %.2f T butterflies/s  %.0f GB/s\n", t, bf / t / 1e9, cells * 8.0 / t / 1e6);
  for (unsigned per_cu : {2u, 3u, 4u}) {
    t = time_ms([&] { hipLaunchKernelGGL(k_regs_glds<0>, dim3(256 * per_cu), dim3(256), 0, 0, data, dtw, blocks); });
    printf("regs_glds %.3f ms  %.2f T butterflies/s  %.0f GB/s   (%u persistent workgroups per CU)\n", t, bf / t / 1e9, cells * 8.0 / t / 1e6, per_cu);
    t = time_ms([&] { hipLaunchKernelGGL(k_regs_glds<40>, dim3(256 * per_cu), dim3(256), 0, 0, data, dtw, blocks); });
    printf("  skewed  %.3f ms   (slot k starts k x 1.1 us late)\n", t);
    t = time_ms([&] { hipLaunchKernelGGL(k_regs_glds<100>, dim3(256 * per_cu), dim3(256), 0, 0, data, dtw, blocks); });
    printf("  skewed  %.3f ms   (slot k starts k x 2.7 us late)\n", t);
  }
  t = time_ms([&] { hipLaunchKernelGGL(k_regs_f64, dim3(blocks), dim3(256), 0, 0, data, dtwd); });
  printf("regs_f64  %.3f ms  %.2f T butterflies/s  %.0f GB/s\n", t, bf / t / 1e9, cells * 8.0 / t / 1e6);
  t = time_ms([&] { hipLaunchKernelGGL(k_regs_rep<2>, dim3(blocks), dim3(256), 0, 0, data, dtw); });
  printf("regs x2   %.3f ms\n", t);
  t = time_ms([&] { hipLaunchKernelGGL(k_regs_rep<4>, dim3(blocks), dim3(256), 0, 0, data, dtw); });
  printf("regs x4   %.3f ms\n", t);
  t = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, data); });
  printf("copy      %.3f ms  %.0f GB/s\n", t, cells * 8.0 / t / 1e6);
  {
    const int iters = 512;
    const unsigned pb = 256 * 8;
    const double pbf = (double)pb * 256 * iters * 32;
    t = time_ms([&] { hipLaunchKernelGGL(k_peak<true>, dim3(pb), dim3(256), 0, 0, data, dtw, iters); });
    printf("peak butterflies, twiddles in registers  %.3f ms  %.2f T/s\n", t, pbf / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL(k_peak<false>, dim3(pb), dim3(256), 0, 0, data, dtw, iters); });
    printf("peak butterflies, twiddles from LDS      %.3f ms  %.2f T/s\n", t, pbf / t / 1e9);
  }
  CK(hipGetLastError());
  return 0;
}
