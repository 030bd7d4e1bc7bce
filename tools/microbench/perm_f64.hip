// FP64 issue rates on gfx950 and the FP64 Poseidon2 permutation (csrc/poseidon2_f64.hip.h) against the
// integer Montgomery one (csrc/poseidon2.h): bit-equality on random and edge-case states, and
// permutations per second of both.  Output is committed under profiles/.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include "../../plonky3_recursion_amd/csrc/poseidon2_f64.hip.h"
using namespace p3r;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048, UNROLL = 16;

template <int OP>
__global__ void __launch_bounds__(256) k_rate(double* out, double seed) {
  double d[UNROLL];
  uint64_t u[UNROLL];
  for (int i = 0; i < UNROLL; ++i) { d[i] = seed + threadIdx.x + i * 0.25; u[i] = threadIdx.x * 77u + i; }
  const double c = seed * 1.0000001;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      if (OP == 0) d[i] = d[i] + d[(i + 1) & 15];
      if (OP == 1) d[i] = d[i] * c;
      if (OP == 2) d[i] = __builtin_fma(d[i], c, d[(i + 1) & 15]);
      if (OP == 3) d[i] = __builtin_rint(d[i]) + 0.0;  // +0.0 folds away; keeps a dependency on rint only
      if (OP == 4) d[i] = __builtin_amdgcn_fract(d[i]);
      if (OP == 5) d[i] = (double)(uint32_t)u[i], u[i] += 3;            // v_cvt_f64_u32 (+ int add)
      if (OP == 6) u[i] = (uint64_t)(int32_t)d[i] + u[(i + 1) & 15];     // v_cvt_i32_f64 (+ int add)
      if (OP == 7) u[i] = (u[i] << 1) + u[(i + 1) & 15];                // v_lshl_add_u64
      if (OP == 9) d[i] = __builtin_ldexp(d[i], (int)(u[0] & 3) - 1);       // v_ldexp_f64
      if (OP == 8) u[i] = u[i] + u[(i + 1) & 15];                       // 64-bit add (add_co/addc or lshl_add_u64)
    }
  }
  double s = 0;
  for (int i = 0; i < UNROLL; ++i) s += d[i] + (double)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void rate(double* out, const char* name) {
  const int blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 3.0);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 3.0 + r);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double total = 5.0 * blocks * 256 * (double)ITER * UNROLL;
  printf("%-34s %8.2f T lane-ops/s\n", name, total / (ms * 1e-3) / 1e12);
}

constexpr int CHAIN = 4;  // permutations per lane per launch (amortises the loads)
template <class PP>
__global__ void __launch_bounds__(256) k_int(const uint32_t* in, uint32_t* out, size_t n, const uint32_t* rc) {
  using F = Fp<PP>;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  F s[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) s[k] = F::raw(in[(size_t)k * n + i]);
  for (int c = 0; c < CHAIN; ++c) p2_permute<PP>(s, rc);
#pragma unroll
  for (int k = 0; k < 16; ++k) out[(size_t)k * n + i] = s[k].v;
}
template <class PP>
__global__ void __launch_bounds__(256) k_f64(const uint32_t* in, uint32_t* out, size_t n, const double* rc) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) s[k] = p2f_load<PP>(in[(size_t)k * n + i]);
  for (int c = 0; c < CHAIN; ++c) p2f_permute<PP>(s, rc);
#pragma unroll
  for (int k = 0; k < 16; ++k) out[(size_t)k * n + i] = p2f_store<PP>(s[k]);
}

template <class PP>
int perm(const char* name) {
  using F = Fp<PP>;
  const size_t n = 1 << 22;
  const int NC = p2_num_constants<PP>();
  std::mt19937_64 g(7);
  std::vector<uint32_t> rc_m(NC);
  std::vector<double> rc_d(NC);
  for (int i = 0; i < NC; ++i) {
    uint32_t c = (uint32_t)(g() % PP::P);
    if (i == 3) c = PP::P - 1;
    if (i == 5) c = 0;
    rc_m[i] = F::from_canonical(c).v;
    rc_d[i] = (double)c;
  }
  std::vector<uint32_t> in(16 * n);
  for (auto& x : in) x = F::from_canonical((uint32_t)(g() % PP::P)).v;
  // edge cases: all zero, all P-1, mixed extremes
  for (int k = 0; k < 16; ++k) {
    in[(size_t)k * n + 0] = 0;
    in[(size_t)k * n + 1] = F::from_canonical(PP::P - 1).v;
    in[(size_t)k * n + 2] = F::from_canonical((k & 1) ? PP::P - 1 : 0).v;
    in[(size_t)k * n + 3] = F::from_canonical((k & 1) ? 1 : PP::P - 1).v;
    in[(size_t)k * n + 4] = F::from_canonical((PP::P - 1) / 2 + (k & 1)).v;
  }
  uint32_t *d_in, *d_o1, *d_o2, *d_rc;
  double* d_rcd;
  CK(hipMalloc(&d_in, 16 * n * 4)); CK(hipMalloc(&d_o1, 16 * n * 4)); CK(hipMalloc(&d_o2, 16 * n * 4));
  CK(hipMalloc(&d_rc, NC * 4)); CK(hipMalloc(&d_rcd, NC * 8));
  CK(hipMemcpy(d_in, in.data(), 16 * n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rc, rc_m.data(), NC * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rcd, rc_d.data(), NC * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b, c;
  hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
  const int blocks = (int)(n / 256), REP = 5;
  hipLaunchKernelGGL(k_int<PP>, dim3(blocks), dim3(256), 0, 0, d_in, d_o1, n, d_rc);
  hipLaunchKernelGGL(k_f64<PP>, dim3(blocks), dim3(256), 0, 0, d_in, d_o2, n, d_rcd);
  hipEventRecord(a);
  for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(k_int<PP>, dim3(blocks), dim3(256), 0, 0, d_in, d_o1, n, d_rc);
  hipEventRecord(b);
  for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(k_f64<PP>, dim3(blocks), dim3(256), 0, 0, d_in, d_o2, n, d_rcd);
  hipEventRecord(c);
  CK(hipEventSynchronize(c));
  float ms1, ms2;
  hipEventElapsedTime(&ms1, a, b); hipEventElapsedTime(&ms2, b, c);
  std::vector<uint32_t> o1(16 * n), o2(16 * n);
  CK(hipMemcpy(o1.data(), d_o1, 16 * n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o2.data(), d_o2, 16 * n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < 16 * n; ++i) bad += o1[i] != o2[i];
  const double perms = (double)REP * n * CHAIN;
  printf("%s: int %.2f G perm/s, f64 %.2f G perm/s, mismatching words %zu of %zu\n", name,
         perms / (ms1 * 1e-3) / 1e9, perms / (ms2 * 1e-3) / 1e9, bad, 16 * n);
  hipFree(d_in); hipFree(d_o1); hipFree(d_o2); hipFree(d_rc); hipFree(d_rcd);
  return bad != 0;
}

int main() {
  double* out;
  CK(hipMalloc(&out, 256 * 8 * 256 * 8));
  rate<0>(out, "v_add_f64");
  rate<1>(out, "v_mul_f64");
  rate<2>(out, "v_fma_f64");
  rate<3>(out, "v_rndne_f64");
  rate<4>(out, "v_fract_f64");
  rate<5>(out, "v_cvt_f64_u32 (+add)");
  rate<6>(out, "v_cvt_i32_f64 (+64-bit add)");
  rate<7>(out, "v_lshl_add_u64");
  rate<8>(out, "64-bit integer add");
  rate<9>(out, "v_ldexp_f64");
  int bad = perm<KoalaBearParams>("koala-bear");
  bad |= perm<BabyBearParams>("baby-bear");
  return bad;
}
