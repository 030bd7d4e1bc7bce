// Does gfx950 overlap FP64 and 32-bit integer VALU work of one wave stream / of co-resident waves?
// k<0>: 16 independent v_fma_f64 chains; k<1>: 16 independent integer modular adds; k<2>: both interleaved.
// If time(k2) ~ max(time(k0), time(k1)) the pipes overlap; if ~ sum they share the issue slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 4096;
template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, double seed) {
  double d[16];
  uint32_t a[16];
  for (int i = 0; i < 16; ++i) { d[i] = seed + threadIdx.x + i; a[i] = threadIdx.x * 2654435761u + i; }
  const double c = seed * 1.0000001;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE != 1) d[i] = __builtin_fma(d[i], c, d[(i + 1) & 15]);
      if (MODE != 0) { uint32_t t = a[i] + a[(i + 1) & 15]; a[i] = min(t, t - 0x7f000001u); }
    }
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += d[i] + a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
float run(double* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, out, 3.0);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, out, 3.0 + r);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  double* out; hipMalloc(&out, 2048 * 256 * 8);
  float f = run<0>(out), i = run<1>(out), m = run<2>(out);
  printf("fp64 only %.3f ms, int only (3 instr per step) %.3f ms, interleaved %.3f ms  (sum %.3f, max %.3f)\n", f, i, m, f + i, f > i ? f : i);
  return 0;
}
