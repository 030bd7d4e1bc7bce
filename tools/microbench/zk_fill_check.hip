// The tiled ZK fills (k_zk_fill_tiles: one ChaCha block per eight cells) against the per-cell kernels they replace
// (k_zk_randomize, k_zk_salts) on a list of shapes, both fields; and their times on a 2^21 x 64 matrix.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I plonky3_recursion_amd/csrc -o tools/microbench/zk_fill_check tools/microbench/zk_fill_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
#include "kernels_zk.hip.h"
using namespace p3r;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_count_diff(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long local = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) local += a[i] != b[i];
  if (local) atomicAdd(count, local);
}
// `--big`: cell indices beyond 2^32 (2^24 rows x 300 columns of one stream, 2^23 x 330 randomised), compared on the device
template <class PP>
int run_big(const char* name) {
  ZkKey key{};
  for (int i = 0; i < 8; ++i) key.k[i] = 0x85EBCA6Bu * (i + 3);
  int bad = 0;
  for (int mode = 1; mode >= 0; --mode) {
    const uint64_t rows = mode ? (uint64_t(1) << 24) : (uint64_t(1) << 24);
    const uint32_t w = mode ? 0 : 328, w2 = mode ? 300 : 330;
    const size_t cells = (size_t)rows * w2;
    uint32_t *src = nullptr, *a, *b;
    if (!mode) { CK(hipMalloc(&src, (rows / 2) * w * 4)); CK(hipMemset(src, 3, (rows / 2) * w * 4)); }
    CK(hipMalloc(&a, cells * 4)); CK(hipMalloc(&b, cells * 4));
    ZkTileJob t{}; t.src = src; t.dst = b; t.rows = rows; t.w = w; t.w2 = w2; t.mode = mode; t.stride = 1; t.stream = zk_stream_id(2, 1);
    t.log_tr = zk_tile_log_rows(rows, w2);
    ZkTileJob* dt; CK(hipMalloc(&dt, sizeof t)); CK(hipMemcpy(dt, &t, sizeof t, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_zk_fill_tiles<PP>, dim3((unsigned)(rows >> t.log_tr)), dim3(kBlock), 0, 0, dt, 1, key);
    if (mode) {
      // the per-cell salts kernel covers (column, 256-row) tiles: 300 x 65536 blocks
      ZkSaltJob j{}; j.dst = a; j.h = rows; j.S = w2; j.stride = 1; j.stream = t.stream;
      ZkSaltJob* dj; CK(hipMalloc(&dj, sizeof j)); CK(hipMemcpy(dj, &j, sizeof j, hipMemcpyHostToDevice));
      // (a grid holds fewer than 2^32 threads: two launches, the second one's block offset carried by a wrapped block0)
      const uint32_t all = (uint32_t)(w2 * ((rows + kBlock - 1) / kBlock)), g1 = all / 2;
      hipLaunchKernelGGL(k_zk_salts<PP>, dim3(g1), dim3(kBlock), 0, 0, dj, 1, key);
      j.block0 = 0u - g1;
      ZkSaltJob* dj2; CK(hipMalloc(&dj2, sizeof j)); CK(hipMemcpy(dj2, &j, sizeof j, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_zk_salts<PP>, dim3(all - g1), dim3(kBlock), 0, 0, dj2, 1, key);
    } else {
      ZkRandomizeJob j{}; j.src = src; j.dst = a; j.h2 = rows; j.w = w; j.w2 = w2; j.stream = t.stream;
      ZkRandomizeJob* dj; CK(hipMalloc(&dj, sizeof j)); CK(hipMemcpy(dj, &j, sizeof j, hipMemcpyHostToDevice));
      const uint32_t all = (uint32_t)(w2 * ((rows + kBlock - 1) / kBlock)), g1 = all / 2;
      hipLaunchKernelGGL(k_zk_randomize<PP>, dim3(g1), dim3(kBlock), 0, 0, dj, 1, key);
      j.block0 = 0u - g1;
      ZkRandomizeJob* dj2; CK(hipMalloc(&dj2, sizeof j)); CK(hipMemcpy(dj2, &j, sizeof j, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_zk_randomize<PP>, dim3(all - g1), dim3(kBlock), 0, 0, dj2, 1, key);
    }
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    unsigned long long* dcount; CK(hipMalloc(&dcount, 8)); CK(hipMemset(dcount, 0, 8));
    hipLaunchKernelGGL(k_count_diff, dim3(8192), dim3(256), 0, 0, a, b, cells, dcount);
    unsigned long long diff = 0;
    CK(hipMemcpy(&diff, dcount, 8, hipMemcpyDeviceToHost));
    printf("%s BIG mode %d: 2^24 rows x %u columns (%zu cells, indices to %.2f x 2^32): %llu cells differ\n", name, mode, w2, cells,
           (double)cells / 4294967296.0, diff);
    bad += diff != 0;
    hipFree(a); hipFree(b); if (src) hipFree(src);
  }
  return bad;
}

template <class PP>
int run(const char* name) {
  ZkKey key{};
  for (int i = 0; i < 8; ++i) key.k[i] = 0x9E3779B9u * (i + 1);
  key.nonce_lo = 7;
  struct Shape { uint64_t h; uint32_t w, R; int mode; uint32_t stride; };   // mode 0 randomise (h = trace rows), 1 salts / random round
  const Shape shapes[] = {{1, 3, 2, 0, 1}, {2, 1, 1, 0, 1}, {64, 5, 2, 0, 1}, {1024, 300, 2, 0, 1}, {4096, 16, 3, 0, 1}, {256, 7, 2, 0, 1}, {2048, 9, 1, 0, 1}, {8, 4, 4, 0, 1}, {32768, 31, 2, 0, 1},
                          {1, 0, 4, 1, 1}, {2, 0, 4, 1, 1}, {512, 0, 4, 1, 1}, {4096, 0, 5, 1, 1}, {1024, 0, 4, 1, 4}, {2048, 0, 6, 1, 1},
                          {8192, 0, 1, 1, 2}, {512, 0, 37, 1, 1}, {128, 2, 1, 0, 1}, {16384, 166, 2, 0, 1}};
  int bad = 0;
  for (const Shape& s : shapes) {
    const uint64_t rows = s.mode == 1 ? s.h : 2 * s.h;
    const uint32_t w2 = s.w + s.R;
    const size_t cells = (size_t)rows * w2 * s.stride;
    uint32_t *src = nullptr, *a = nullptr, *b = nullptr;
    std::vector<uint32_t> hs((size_t)s.h * (s.w ? s.w : 1));
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = (uint32_t)((i * 2654435761u) % PP::P);
    CK(hipMalloc(&src, hs.size() * 4)); CK(hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&a, cells * 4)); CK(hipMalloc(&b, cells * 4));
    CK(hipMemset(a, 0xAB, cells * 4)); CK(hipMemset(b, 0xAB, cells * 4));
    ZkTileJob t{};
    t.src = s.mode == 1 ? nullptr : src; t.dst = b; t.rows = rows; t.w = s.mode == 1 ? 0 : s.w; t.w2 = w2; t.mode = s.mode; t.stride = s.stride;
    t.stream = zk_stream_id(1, 3); t.log_tr = zk_tile_log_rows(rows, w2); t.block0 = 0;
    ZkTileJob* dt; CK(hipMalloc(&dt, sizeof t)); CK(hipMemcpy(dt, &t, sizeof t, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_zk_fill_tiles<PP>, dim3((unsigned)(rows >> t.log_tr)), dim3(kBlock), 0, 0, dt, 1, key);
    if (s.mode == 1) {
      ZkSaltJob j{}; j.dst = a; j.h = rows; j.S = w2; j.stride = s.stride; j.stream = t.stream; j.block0 = 0;
      ZkSaltJob* dj; CK(hipMalloc(&dj, sizeof j)); CK(hipMemcpy(dj, &j, sizeof j, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_zk_salts<PP>, dim3((unsigned)(w2 * ((rows + kBlock - 1) / kBlock))), dim3(kBlock), 0, 0, dj, 1, key);
    } else {
      ZkRandomizeJob j{}; j.src = src; j.dst = a; j.h2 = rows; j.w = s.w; j.w2 = w2; j.zero_fill = s.mode == 2; j.stream = t.stream; j.block0 = 0;
      ZkRandomizeJob* dj; CK(hipMalloc(&dj, sizeof j)); CK(hipMemcpy(dj, &j, sizeof j, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_zk_randomize<PP>, dim3((unsigned)(w2 * ((rows + kBlock - 1) / kBlock))), dim3(kBlock), 0, 0, dj, 1, key);
    }
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> ha(cells), hb(cells);
    CK(hipMemcpy(ha.data(), a, cells * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, cells * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < cells; ++i) diff += ha[i] != hb[i];
    printf("%s h %llu w %u R %u mode %d stride %u (tile rows %u): %zu of %zu cells differ\n", name, (unsigned long long)s.h, s.w, s.R, s.mode,
           s.stride, 1u << t.log_tr, diff, cells);
    bad += diff != 0;
    hipFree(src); hipFree(a); hipFree(b);
  }
  // times
  for (int shape = 0; shape < 2; ++shape) {
    const uint64_t h = shape ? (1 << 19) : (1 << 21); const uint32_t w = shape ? 166 : 62, R = 2, w2 = w + R; const uint64_t rows = 2 * h;
    uint32_t *src, *a; CK(hipMalloc(&src, h * w * 4)); CK(hipMalloc(&a, rows * w2 * 4)); CK(hipMemset(src, 1, h * w * 4));
    ZkTileJob t{}; t.src = src; t.dst = a; t.rows = rows; t.w = w; t.w2 = w2; t.mode = 0; t.stride = 1; t.stream = 5; t.log_tr = zk_tile_log_rows(rows, w2);
    ZkTileJob* dt; CK(hipMalloc(&dt, sizeof t)); CK(hipMemcpy(dt, &t, sizeof t, hipMemcpyHostToDevice));
    ZkRandomizeJob j{}; j.src = src; j.dst = a; j.h2 = rows; j.w = w; j.w2 = w2; j.stream = 5;
    ZkRandomizeJob* dj; CK(hipMalloc(&dj, sizeof j)); CK(hipMemcpy(dj, &j, sizeof j, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms_new = 0, ms_old = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_zk_fill_tiles<PP>, dim3((unsigned)(rows >> t.log_tr)), dim3(kBlock), 0, 0, dt, 1, key);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_new, e0, e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_zk_randomize<PP>, dim3((unsigned)(w2 * ((rows + kBlock - 1) / kBlock))), dim3(kBlock), 0, 0, dj, 1, key);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_old, e0, e1);
    }
    printf("%s randomise %llu x %u -> %llu x %u: tiled %.3f ms, per cell %.3f ms\n", name, (unsigned long long)h, w, (unsigned long long)rows, w2, ms_new, ms_old);
    const uint32_t S = 4;
    ZkTileJob ts{}; ts.dst = a; ts.rows = rows; ts.w2 = S; ts.mode = 1; ts.stride = 1; ts.stream = 6; ts.log_tr = zk_tile_log_rows(rows, S);
    CK(hipMemcpy(dt, &ts, sizeof ts, hipMemcpyHostToDevice));
    ZkSaltJob sj{}; sj.dst = a; sj.h = rows; sj.S = S; sj.stride = 1; sj.stream = 6;
    ZkSaltJob* dsj; CK(hipMalloc(&dsj, sizeof sj)); CK(hipMemcpy(dsj, &sj, sizeof sj, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_zk_fill_tiles<PP>, dim3((unsigned)(rows >> ts.log_tr)), dim3(kBlock), 0, 0, dt, 1, key);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_new, e0, e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_zk_salts<PP>, dim3((unsigned)(S * ((rows + kBlock - 1) / kBlock))), dim3(kBlock), 0, 0, dsj, 1, key);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_old, e0, e1);
    }
    printf("%s salts %llu x 4: tiled %.3f ms, per cell %.3f ms\n", name, (unsigned long long)rows, ms_new, ms_old);
  }
  return bad;
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc > 1 && std::string(argv[1]) == "--big") {
    int b = run_big<KoalaBearParams>("koala-bear");
    b += run_big<BabyBearParams>("baby-bear");
    printf(b ? "MISMATCH\n" : "all shapes identical\n");
    return b != 0;
  }
  int bad = run<KoalaBearParams>("koala-bear");
  bad += run<BabyBearParams>("baby-bear");
  printf(bad ? "MISMATCH\n" : "all shapes identical\n");
  return bad != 0;
}
