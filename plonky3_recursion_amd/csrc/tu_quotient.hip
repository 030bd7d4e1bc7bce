// K8: the quotient kernel, one instance per (field, circuit degree, challenge degree) - the AIR constraint systems
// are inlined into it, which makes it the largest piece of device code of the library.  Own translation unit (tu_api.h).
#include "tu_api.h"

namespace p3r {

template <class PP, int DC>
void launch_quotient(p3r_ctx* ctx, unsigned blocks, const QuotientArgs* d_jobs, int n_jobs, const LookupChT<DC>& lc) {
  dispatch_air_degree<PP>((int)ctx->cfg.ext_degree, [&](auto dc) {
    hipLaunchKernelGGL((k_quotient<PP, decltype(dc)::value, DC>), dim3(blocks), dim3(kBlock), 0, ctx->stream, d_jobs, n_jobs, lc,
                       ctx->rc.p);
  });
  P3R_HIP(hipGetLastError());
}

template void launch_quotient<KoalaBearParams, 4>(p3r_ctx*, unsigned, const QuotientArgs*, int, const LookupChT<4>&);
template void launch_quotient<KoalaBearParams, 5>(p3r_ctx*, unsigned, const QuotientArgs*, int, const LookupChT<5>&);
template void launch_quotient<BabyBearParams, 4>(p3r_ctx*, unsigned, const QuotientArgs*, int, const LookupChT<4>&);

}  // namespace p3r
