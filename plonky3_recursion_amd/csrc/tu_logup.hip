// K7: the LogUp fraction kernel, one instance per (field, circuit degree, challenge degree) - every AIR's bus
// interactions are inlined into it.  Own translation unit (tu_api.h).
#include "tu_api.h"

namespace p3r {

template <class PP, int DC>
void launch_logup_aux(p3r_ctx* ctx, unsigned blocks, const LogupJob* d_jobs, int n_jobs, const LookupChT<DC>& lc) {
  dispatch_air_degree<PP>((int)ctx->cfg.ext_degree, [&](auto dc) {
    hipLaunchKernelGGL((k_logup_aux<PP, decltype(dc)::value, DC>), dim3(blocks), dim3(kBlock), 0, ctx->stream, d_jobs, n_jobs, lc);
  });
  P3R_HIP(hipGetLastError());
}

template void launch_logup_aux<KoalaBearParams, 4>(p3r_ctx*, unsigned, const LogupJob*, int, const LookupChT<4>&);
template void launch_logup_aux<KoalaBearParams, 5>(p3r_ctx*, unsigned, const LogupJob*, int, const LookupChT<5>&);
template void launch_logup_aux<BabyBearParams, 4>(p3r_ctx*, unsigned, const LogupJob*, int, const LookupChT<4>&);

}  // namespace p3r
