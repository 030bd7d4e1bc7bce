// K6 of the arity-4 MMCS: the three kernels over the width-32 FP64 permutation and their launches.  Own translation
// unit (tu_api.h): the permutation is 11 k instructions of straight-line code per instance.
#include "tu_api.h"
#include "kernels_mmcs4.hip.h"
#include "profile.h"

namespace p3r {

// Up to this many nodes / rows the lane-cooperative form wins: a launch of the one-permutation-per-lane form is one
// permutation latency (13 us with the built-in diagonal's 7.8 k instructions, 19 us with the general 11.3 k) whatever its
// size up to ~64 K nodes, the cooperative one ~5 us per pass of 16 K nodes; above, the FP64 form's throughput wins (a
// lane-cooperative permutation is 32 lanes x ~1.7 k integer instructions against 7.8 - 11.3 k FP64 instructions of one lane).
constexpr size_t kCoop4MaxNodes = 32768, kCoop4MaxRows = 32768;

template <class PP>
void mmcs4_hash_rows(p3r_ctx* ctx, const std::vector<std::vector<const p3r_dmat*>>& classes, const std::vector<uint32_t*>& digs,
                     const std::vector<size_t>& allocs) {
  std::vector<HashRowsJob4> jobs;
  for (size_t c = 0; c < classes.size(); ++c) {
    std::vector<const uint32_t*> cols;
    for (const p3r_dmat* m : classes[c])
      for (size_t k = 0; k < m->w; ++k) cols.push_back(m->d + k * m->h);
    HashRowsJob4 j{};
    j.cols = col_table(ctx, cols);
    j.dig = digs[c];
    j.h = classes[c][0]->h;
    j.h_alloc = allocs[c];
    j.wtot = (int)cols.size();
    jobs.push_back(j);
  }
  // widest rows first: their blocks run longest
  std::stable_sort(jobs.begin(), jobs.end(), [](const HashRowsJob4& a, const HashRowsJob4& b) { return a.wtot > b.wtot; });
  uint32_t blocks = 0;
  double perms = 0;
  for (auto& j : jobs) {
    j.block0 = blocks;
    blocks += (uint32_t)((j.h + kBlock - 1) / kBlock);
    perms += (double)j.h * ((j.wtot + P2W_RATE - 1) / P2W_RATE);   // width-32 permutations of the rate-24 sponge
  }
  prof_count(ctx, "hash_rows_perms", perms);
  const auto* d_jobs = static_cast<const HashRowsJob4*>(const_table(ctx, jobs.data(), jobs.size() * sizeof(HashRowsJob4)));
  ProfScope ps(ctx, "mmcs_hash_rows");
  const auto kern = ctx->w32_diag_builtin ? &k_mmcs4_hash_rows<PP, true> : &k_mmcs4_hash_rows<PP, false>;   // poseidon2_w32_f64.hip.h
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_jobs, (int)jobs.size(), ctx->rcd_w32());
  P3R_HIP(hipGetLastError());
}

template <class PP>
void mmcs4_hash_rows_strided(p3r_ctx* ctx, const uint32_t* const* dcols, int wtot, size_t rows, size_t stride, uint32_t* dig,
                             size_t alloc) {
  ProfScope ps(ctx, "mmcs_hash_rows_strided");
  if (rows <= kCoop4MaxRows) {   // latency-bound: 32 lanes per row
    hipLaunchKernelGGL(k_mmcs4_hash_rows_strided_coop<PP>, dim3((unsigned)((rows * 32 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       ctx->stream, dcols, wtot, rows, stride, dig, alloc, ctx->rc.p + p2_num_constants<PP>());
    P3R_HIP(hipGetLastError());
    return;
  }
  const auto kern = ctx->w32_diag_builtin ? &k_mmcs4_hash_rows_strided<PP, true> : &k_mmcs4_hash_rows_strided<PP, false>;   // poseidon2_w32_f64.hip.h
  hipLaunchKernelGGL(kern, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, dcols,
                     wtot, rows, stride, dig, alloc, ctx->rcd_w32());
  P3R_HIP(hipGetLastError());
}

template <class PP>
void mmcs4_compress(p3r_ctx* ctx, const uint32_t* prev, size_t n_prev, int step, const uint32_t* inj, uint32_t* out,
                    size_t n_logical, size_t n_out) {
  ProfScope ps(ctx, "mmcs_compress");
  if (n_out <= kCoop4MaxNodes) {   // latency-bound: 32 lanes per node
    hipLaunchKernelGGL(k_mmcs4_compress_coop<PP>, dim3((unsigned)((n_out * 32 + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                       prev, n_prev, step, inj, out, n_logical, n_out, ctx->rc.p + p2_num_constants<PP>());
    P3R_HIP(hipGetLastError());
    return;
  }
  const auto kern = ctx->w32_diag_builtin ? &k_mmcs4_compress<PP, true> : &k_mmcs4_compress<PP, false>;   // poseidon2_w32_f64.hip.h
  hipLaunchKernelGGL(kern, dim3((unsigned)((n_out + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, prev, n_prev,
                     step, inj, out, n_logical, n_out, ctx->rcd_w32());
  P3R_HIP(hipGetLastError());
}

#define P3R_MMCS4_INSTANCES(PP)                                                                                              \
  template void mmcs4_hash_rows<PP>(p3r_ctx*, const std::vector<std::vector<const p3r_dmat*>>&, const std::vector<uint32_t*>&, \
                                    const std::vector<size_t>&);                                                              \
  template void mmcs4_hash_rows_strided<PP>(p3r_ctx*, const uint32_t* const*, int, size_t, size_t, uint32_t*, size_t);        \
  template void mmcs4_compress<PP>(p3r_ctx*, const uint32_t*, size_t, int, const uint32_t*, uint32_t*, size_t, size_t);
P3R_MMCS4_INSTANCES(KoalaBearParams)
P3R_MMCS4_INSTANCES(BabyBearParams)

}  // namespace p3r
