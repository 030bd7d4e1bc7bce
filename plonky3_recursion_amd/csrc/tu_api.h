// Functions that cross the translation units of libp3r_hip.so.  The library is several units so that they build
// side by side (one unit took ten minutes): p3r_core.hip (C ABI, MMCS, prover sequencing, circuit boundary),
// tu_lde.hip (K5: NTT tables, passes and the coset LDE), tu_quotient.hip / tu_logup.hip (the two kernels with the
// AIR constraint systems inlined, one instance per circuit degree and challenge degree), prep_device.hip.
// Kernels never call across units; only these host entry points do.
#pragma once
#include <memory>
#include <vector>

#include "context.h"
#include "kernels_stark.hip.h"

namespace p3r {

inline std::unique_ptr<p3r_dmat> dmat_alloc(size_t h, size_t w) {
  log2_exact(h, "matrix height");
  auto m = std::make_unique<p3r_dmat>();
  m->buf.alloc(h * w);
  m->d = m->buf.p;
  m->h = h;
  m->w = w;
  return m;
}

// Device copy of a small read-only table (see p3r_ctx::const_tables).
inline const void* const_table(p3r_ctx* ctx, const void* data, size_t bytes) {
  std::string key(static_cast<const char*>(data), bytes);
  auto it = ctx->const_tables.find(key);
  if (it == ctx->const_tables.end()) {
    DevBuf b((bytes + 3) / 4);
    P3R_HIP(ctx->stage.upload(ctx->stream, b.p, data, bytes));
    it = ctx->const_tables.emplace(std::move(key), std::move(b)).first;
  }
  return it->second.p;
}
inline const uint32_t* const* col_table(p3r_ctx* ctx, const std::vector<const uint32_t*>& cols) {
  return static_cast<const uint32_t* const*>(const_table(ctx, cols.data(), cols.size() * sizeof(void*)));
}

// K5 for a batch of matrices (tu_lde.hip).  in: h x w evaluations over the subgroup (natural order, column-major
// Montgomery).  Returns (h << added_bits) x w, rows in bit-reversed order over shift * <w_{h << added_bits}>.
struct LdeItem {
  const p3r_dmat* in;
  uint32_t shift;  // canonical coset shift
};
template <class PP>
std::vector<std::unique_ptr<p3r_dmat>> coset_lde_batch(p3r_ctx* ctx, const std::vector<LdeItem>& items, int added_bits);
// once per context: kernel attributes of the unit's kernels
template <class PP>
void lde_init(p3r_ctx* ctx);

// K7 / K8 launches (tu_logup.hip, tu_quotient.hip): the instance of the context's circuit degree
template <class PP, int DC>
void launch_logup_aux(p3r_ctx* ctx, unsigned blocks, const LogupJob* d_jobs, int n_jobs, const LookupChT<DC>& lc);
template <class PP, int DC>
void launch_quotient(p3r_ctx* ctx, unsigned blocks, const QuotientArgs* d_jobs, int n_jobs, const LookupChT<DC>& lc);

// K6 of the arity-4 MMCS (tu_mmcs4.hip; kernels_mmcs4.hip.h).  classes[c] = the matrices of one height, digs[c] =
// [8][allocs[c]] with allocs[c] >= that height.
template <class PP>
void mmcs4_hash_rows(p3r_ctx* ctx, const std::vector<std::vector<const p3r_dmat*>>& classes, const std::vector<uint32_t*>& digs,
                     const std::vector<size_t>& allocs);
template <class PP>
void mmcs4_hash_rows_strided(p3r_ctx* ctx, const uint32_t* const* dcols, int wtot, size_t rows, size_t stride, uint32_t* dig,
                             size_t alloc);
template <class PP>
void mmcs4_compress(p3r_ctx* ctx, const uint32_t* prev, size_t n_prev, int step, const uint32_t* inj, uint32_t* out,
                    size_t n_logical, size_t n_out);

}  // namespace p3r
