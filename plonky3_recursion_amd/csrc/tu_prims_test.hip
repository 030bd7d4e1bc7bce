// Test seam for device_prims.hip.h.  Built into the knobs library only (plonky3_recursion_amd/knobs/libp3r_hip.so,
// -DP3R_TUNING_KNOBS: what tests and tuning tools load); the product library neither compiles nor exports it.
// Host arrays in, host arrays out, on the context's stream and pool.  tests/test_gpu_device_prims.py.
#include "device_prims.hip.h"

namespace p3r {
namespace {
struct LoadU64 {
  const uint64_t* p;
  __device__ uint64_t operator()(size_t i) const { return p[i]; }
};
template <class Fn>
int seam(p3r_ctx* ctx, Fn&& fn) {
  try {
    if (!ctx) return P3R_EINVAL;
    (void)hipSetDevice(ctx->cfg.device);
    tls_pool() = ctx->pool;
    fn();
    return P3R_OK;
  } catch (const Error& e) {
    ctx->err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    ctx->err = e.what();
    return P3R_EINVAL;
  }
}
}  // namespace
}  // namespace p3r

using namespace p3r;

extern "C" {
int p3r_test_exclusive_sum_u32(p3r_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, int in_place) {
  return seam(ctx, [&] {
    if (!n) return;
    DevBuf a(n), b(n);
    P3R_HIP(hipMemcpyAsync(a.p, in, n * 4, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* dst = in_place ? a.p : b.p;
    prims::exclusive_sum<uint32_t>(ctx->stream, prims::LoadU32{a.p}, n, dst);
    P3R_HIP(copy_sync(ctx->stream, out, dst, n * 4, hipMemcpyDeviceToHost));
  });
}
int p3r_test_exclusive_sum_u64(p3r_ctx* ctx, const uint64_t* in, uint64_t* out, size_t n) {
  return seam(ctx, [&] {
    if (!n) return;
    DevBuf a(2 * n), b(2 * n);
    P3R_HIP(hipMemcpyAsync(a.p, in, n * 8, hipMemcpyHostToDevice, ctx->stream));
    prims::exclusive_sum<uint64_t>(ctx->stream, LoadU64{reinterpret_cast<const uint64_t*>(a.p)}, n, reinterpret_cast<uint64_t*>(b.p));
    P3R_HIP(copy_sync(ctx->stream, out, b.p, n * 8, hipMemcpyDeviceToHost));
  });
}
int p3r_test_reduce_max(p3r_ctx* ctx, const uint32_t* in, size_t n, uint32_t* out) {
  return seam(ctx, [&] {
    DevBuf a(n + 1), m(1);
    if (n) P3R_HIP(hipMemcpyAsync(a.p, in, n * 4, hipMemcpyHostToDevice, ctx->stream));
    prims::reduce_max(ctx->stream, prims::LoadU32{a.p}, n, m.p);
    P3R_HIP(copy_sync(ctx->stream, out, m.p, 4, hipMemcpyDeviceToHost));
  });
}
int p3r_test_sort_pairs(p3r_ctx* ctx, const uint32_t* keys, const uint32_t* vals, size_t n, int bits, uint32_t* keys_out, uint32_t* vals_out) {
  return seam(ctx, [&] {
    if (!n) return;
    DevBuf k(n), v(n), ko(n), vo(n);
    P3R_HIP(hipMemcpyAsync(k.p, keys, n * 4, hipMemcpyHostToDevice, ctx->stream));
    P3R_HIP(hipMemcpyAsync(v.p, vals, n * 4, hipMemcpyHostToDevice, ctx->stream));
    prims::sort_pairs(ctx->stream, k.p, ko.p, v.p, vo.p, n, bits);
    P3R_HIP(hipMemcpyAsync(keys_out, ko.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    P3R_HIP(copy_sync(ctx->stream, vals_out, vo.p, n * 4, hipMemcpyDeviceToHost));
  });
}
}
