// Integer / FP64 VALU issue-rate microbenchmark for gfx950: the ceiling the Poseidon2 and NTT
// kernels are priced against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, UNROLL = 16;

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a[UNROLL];
  double d[UNROLL];
  uint32_t b = seed | 1u;
  double kmagic = 0x1.8p52, kphi = 2130706432.0, kk = -(0x1.8p52 * 2130706433.0);
  asm volatile("" : "+s"(kmagic), "+s"(kphi), "+v"(kk));
  for (int i = 0; i < UNROLL; ++i) { a[i] = threadIdx.x * 2654435761u + i + seed; d[i] = (double)a[i]; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      if (OP == 0) a[i] = a[i] + a[(i + 1) & 15];                                  // v_add_u32
      if (OP == 1) a[i] = a[i] * a[(i + 1) & 15];                                  // v_mul_lo_u32
      if (OP == 2) a[i] = __umulhi(a[i], a[(i + 1) & 15]) | 0x80000001u;                         // v_mul_hi_u32
      if (OP == 3) { uint64_t p = (uint64_t)a[i] * b + a[i]; a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; }  // mad_u64_u32
      if (OP == 4) a[i] = (a[i] << 7) + a[(i + 1) & 15];        // v_mul_u32_u24
      if (OP == 5) d[i] = __builtin_fma(d[i], 1.0000001, 0.5);        // v_fma_f64
      if (OP == 6) { uint32_t t = a[i] + a[(i + 1) & 15]; a[i] = min(t, t - 0x7f000001u); }                   // add, sub, min
      if (OP == 7) { uint32_t lo = a[i] * b, hi = __umulhi(a[i], b); uint32_t t = lo * 0x81000001u; uint32_t u = __umulhi(t, 0x7f000001u); uint32_t r = hi - u; a[i] = hi < u ? r + 0x7f000001u : r; }  // Montgomery product
      if (OP == 9) {  // Shoup product by a fixed w (w' = floor(w 2^32 / P)): 1 mul_hi + 2 mul_lo
        uint32_t q = __umulhi(a[i], 0x9A3C5E71u);
        uint32_t r = a[i] * b - q * 0x7f000001u;
        a[i] = min(r, r - 0x7f000001u) + i;
      }
      if (OP == 10) d[i] = d[i] + d[(i + 1) & 15];                                  // v_add_f64
      if (OP == 11) d[i] = d[i] * 1.0000001;                                        // v_mul_f64
      if (OP == 12) d[i] = __builtin_amdgcn_fract(d[i]) + 1.5;                      // v_fract_f64 (+ v_add_f64)
      if (OP == 13) d[i] = __builtin_rint(d[i]) * 1.0000001;                        // v_rndne_f64 (+ v_mul_f64)
      if (OP == 14) {  // FP64 product mod P, five instructions (poseidon2_f64.hip.h: p2f_mulmod_c), c = x / P given
        const double x = d[i], c = d[(i + 1) & 15];
        const double q = __builtin_fma(x, c, 0x1.8p52) - 0x1.8p52;
        const double t = q * 2130706432.0;
        d[i] = __builtin_fma(x, x, -t) - q;
      }
      if (OP == 15) {  // the same in four (p2f_mulmod_k): the rounding constant rides through the chain
        const double x = d[i], c = d[(i + 1) & 15];
        const double qm = __builtin_fma(x, c, kmagic);
        const double t = __builtin_fma(qm, kphi, kk);
        d[i] = __builtin_fma(x, x, -t) - qm;
      }
      if (OP == 8) {  // Montgomery product through two 64-bit multiply-adds: x = a*b; y = q*P + x; r = y >> 32
        uint64_t x = (uint64_t)a[i] * b;
        uint32_t q = (uint32_t)x * 0x7EFFFFFFu;
        uint64_t y = (uint64_t)q * 0x7f000001u + x;
        uint32_t r = (uint32_t)(y >> 32);
        a[i] = min(r, r - 0x7f000001u);
      }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < UNROLL; ++i) s ^= a[i] ^ (uint32_t)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(uint32_t* out, const char* name, double ops_per_iter) {
  const int blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double total = 5.0 * blocks * 256 * (double)ITER * UNROLL * ops_per_iter;
  double rate = total / (ms * 1e-3) / 1e12;
  printf("%-28s %8.2f T lane-ops/s  (%.3f ms)\n", name, rate, ms / 5);
  return rate;
}

int main() {
  uint32_t* out;
  CK(hipMalloc(&out, 256 * 8 * 256 * 4));
  run<0>(out, "v_add_u32", 1);
  run<1>(out, "v_mul_lo_u32", 1);
  run<2>(out, "v_mul_hi_u32", 1);
  run<3>(out, "mad_u64_u32 (+xor)", 1);
  run<4>(out, "v_lshl_add_u32", 1);
  run<5>(out, "v_fma_f64", 1);
  run<6>(out, "modular add (add,sub,min)", 1);
  run<7>(out, "Montgomery product", 1);
  run<8>(out, "Montgomery via mad_u64_u32", 1);
  run<9>(out, "Shoup product (+add)", 1);
  run<10>(out, "v_add_f64", 1);
  run<11>(out, "v_mul_f64", 1);
  run<12>(out, "v_fract_f64 + v_add_f64", 2);
  run<13>(out, "v_rndne_f64 + v_mul_f64", 2);
  run<14>(out, "FP64 product mod P, 5 instr.", 1);
  run<15>(out, "FP64 product mod P, 4 instr.", 1);
  return 0;
}
