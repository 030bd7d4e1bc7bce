// C ABI (include/p3r.h): context, device matrices, Poseidon2 (K3), coset LDE (K5), MMCS (K6).
// Host orchestration only; all arithmetic on data runs in the gfx950 kernels of kernels.cuh.
#include "context.h"
#include "kernels.cuh"
#include "kernels_stark.cuh"
#include "kernels_coop.cuh"
#include "kernels_ntt2.cuh"
#include "poseidon2_rc_default.inc"
#include "profile.h"
#include "run_schedule.h"
#include "prep_device.h"

#include <algorithm>
#include <future>
#include <numeric>
#include <unordered_map>

using namespace p3r;

namespace {

thread_local std::string g_create_error;

template <class Fn>
int guard(p3r_ctx* ctx, Fn&& fn) {
  try {
    // the calling thread's current device may have been changed by the embedding framework
    if (ctx) {
      (void)hipSetDevice(ctx->cfg.device);
      tls_pool() = ctx->pool;
      // the table cache is only trimmed here, between API calls: a call keeps device pointers into
      // it while it assembles its launches (e.g. column tables referenced from a job list)
      if (ctx->const_tables.size() > 4096) ctx->const_tables.clear();
    }
    fn();
    return P3R_OK;
  } catch (const Error& e) {
    if (ctx) ctx->err = e.what(); else g_create_error = e.what();
    return e.code;
  } catch (const std::exception& e) {
    if (ctx) ctx->err = e.what(); else g_create_error = e.what();
    return P3R_EINVAL;
  }
}

#define P3R_FIELD_CALL(ctx, fn, ...)                                              \
  ((ctx)->cfg.field == P3R_FIELD_KOALA_BEAR ? fn<KoalaBearParams>(__VA_ARGS__)    \
                                            : fn<BabyBearParams>(__VA_ARGS__))

inline unsigned blocks_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// ------------------------------------------------------------------ matrices
template <class PP>
std::unique_ptr<p3r_dmat> upload(p3r_ctx* ctx, const uint32_t* rowmajor, size_t h, size_t w) {
  log2_exact(h, "matrix height");
  if (w == 0) fail(P3R_EINVAL, "matrix width must be positive");
  auto m = std::make_unique<p3r_dmat>();
  m->buf.alloc(h * w);
  m->d = m->buf.p;
  m->h = h;
  m->w = w;
  DevBuf stage(h * w);
  P3R_HIP(hipMemcpyAsync(stage.p, rowmajor, h * w * 4, hipMemcpyHostToDevice, ctx->stream));
  dim3 grid((unsigned)((h + 63) / 64), (unsigned)((w + 63) / 64));
  hipLaunchKernelGGL(k_rowmajor_to_colmajor<PP>, grid, dim3(kBlock), 0, ctx->stream, stage.p,
                     m->d, (uint32_t)h, (uint32_t)w, 1);
  P3R_HIP(hipGetLastError());
  // the caller's (pageable) host buffer must be fully consumed before we return
  P3R_HIP(hipStreamSynchronize(ctx->stream));
  return m;
}

template <class PP>
void download(p3r_ctx* ctx, const p3r_dmat* m, uint32_t* rowmajor_out) {
  DevBuf stage(m->h * m->w);
  dim3 grid((unsigned)((m->h + 63) / 64), (unsigned)((m->w + 63) / 64));
  hipLaunchKernelGGL(k_colmajor_to_rowmajor<PP>, grid, dim3(kBlock), 0, ctx->stream, m->d,
                     stage.p, (uint32_t)m->h, (uint32_t)m->w, 1);
  P3R_HIP(hipGetLastError());
  P3R_HIP(hipMemcpyAsync(rowmajor_out, stage.p, m->h * m->w * 4, hipMemcpyDeviceToHost,
                         ctx->stream));
  P3R_HIP(hipStreamSynchronize(ctx->stream));
}

std::unique_ptr<p3r_dmat> dmat_alloc(size_t h, size_t w) {
  log2_exact(h, "matrix height");
  auto m = std::make_unique<p3r_dmat>();
  m->buf.alloc(h * w);
  m->d = m->buf.p;
  m->h = h;
  m->w = w;
  return m;
}

// ------------------------------------------------------------------ Poseidon2
template <class PP>
void permute_dmat(p3r_ctx* ctx, p3r_dmat* s) {
  if (s->w != P2_WIDTH) fail(P3R_EINVAL, "state matrix must have width 16, got %zu", s->w);
  ProfScope ps(ctx, "p2_permute_batch");
  hipLaunchKernelGGL(k_p2_permute_batch<PP>, dim3(blocks_for(s->h)), dim3(kBlock), 0, ctx->stream,
                     s->d, s->d, s->h, ctx->rcd());
  P3R_HIP(hipGetLastError());
}

// Device-resident Poseidon2CircuitRow batch (inputs column-major Montgomery, flags as bytes).
template <class PP>
std::unique_ptr<p3r_p2_dev> p2_rows_upload(p3r_ctx* ctx, const p3r_p2_rows* rows) {
  const size_t n = rows->n;
  log2_exact(n, "Poseidon2 row count (callers pad to a power of two)");
  if (!rows->input_values || !rows->new_start || !rows->merkle_path || !rows->mmcs_bit ||
      !rows->mmcs_index_sum)
    fail(P3R_EINVAL, "p3r_p2_rows has a NULL field");
  auto d = std::make_unique<p3r_p2_dev>();
  d->n = n;
  d->flags.alloc((3 * n + 3) / 4 + 1);
  uint8_t* f8 = reinterpret_cast<uint8_t*>(d->flags.p);
  P3R_HIP(hipMemcpyAsync(f8, rows->new_start, n, hipMemcpyHostToDevice, ctx->stream));
  P3R_HIP(hipMemcpyAsync(f8 + n, rows->merkle_path, n, hipMemcpyHostToDevice, ctx->stream));
  P3R_HIP(hipMemcpyAsync(f8 + 2 * n, rows->mmcs_bit, n, hipMemcpyHostToDevice, ctx->stream));
  d->seed.alloc(n);
  P3R_HIP(hipMemcpyAsync(d->seed.p, rows->mmcs_index_sum, n * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_convert_inplace<PP>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream,
                     d->seed.p, n, 1);
  P3R_HIP(hipGetLastError());
  d->inputs = upload<PP>(ctx, rows->input_values, n, P2_WIDTH);  // syncs the stream
  return d;
}

template <class PP>
std::unique_ptr<p3r_dmat> trace_fill(p3r_ctx* ctx, const p3r_p2_dev* rows) {
  const size_t n = rows->n;
  const uint8_t* f8 = reinterpret_cast<const uint8_t*>(rows->flags.p);
  DevBuf acc(n);
  const size_t n_blocks = (n + kScanTile - 1) / kScanTile;
  DevBuf agg(2 * n_blocks);
  {
    ProfScope ps(ctx, "p2_acc_scan");
    hipLaunchKernelGGL(k_p2_acc_scan<PP>, dim3((unsigned)n_blocks), dim3(kBlock), 0, ctx->stream, 0,
                       n, f8, f8 + n, f8 + 2 * n, rows->seed.p, agg.p, n_blocks, acc.p);
    hipLaunchKernelGGL(k_p2_acc_scan<PP>, dim3(1), dim3(kBlock), 0, ctx->stream, 1, n, f8, f8 + n,
                       f8 + 2 * n, rows->seed.p, agg.p, n_blocks, acc.p);
    hipLaunchKernelGGL(k_p2_acc_scan<PP>, dim3((unsigned)n_blocks), dim3(kBlock), 0, ctx->stream, 2,
                       n, f8, f8 + n, f8 + 2 * n, rows->seed.p, agg.p, n_blocks, acc.p);
  }
  const size_t width = p2_perm_cols<PP>() + 2;
  auto trace = dmat_alloc(n, width);
  {
    ProfScope ps(ctx, "p2_trace_fill");
    hipLaunchKernelGGL(k_p2_trace_fill<PP>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream,
                       rows->inputs->d, f8 + 2 * n, acc.p, trace->d, n, ctx->rc.p);
  }
  P3R_HIP(hipGetLastError());
  return trace;
}

// ------------------------------------------------------------------ NTT tables
template <class PP>
const uint32_t* get_tw_sub(p3r_ctx* ctx, int log_r, int inverse) {
  auto key = std::make_pair(log_r, inverse);
  auto it = ctx->tw_sub.find(key);
  if (it != ctx->tw_sub.end()) return it->second.p;
  using F = Fp<PP>;
  size_t half = log_r ? (size_t(1) << (log_r - 1)) : 1;
  std::vector<uint32_t> t(half);
  F root = F::two_adic_generator(log_r);
  if (inverse) root = root.inv();
  F x = F::one();
  for (size_t i = 0; i < half; ++i) {
    t[i] = x.v;  // Montgomery form
    x *= root;
  }
  DevBuf d(half);
  P3R_HIP(copy_sync(ctx->stream, d.p, t.data(), half * 4, hipMemcpyHostToDevice));
  return ctx->tw_sub.emplace(key, std::move(d)).first->second.p;
}

template <class PP>
std::pair<const uint32_t*, const uint32_t*> get_tw4(p3r_ctx* ctx, int log_n, int inverse) {
  auto key = std::make_pair(log_n, inverse);
  auto it = ctx->tw4.find(key);
  if (it == ctx->tw4.end()) {
    using F = Fp<PP>;
    F root = F::two_adic_generator(log_n);
    if (inverse) root = root.inv();
    size_t n_hi = log_n > 10 ? (size_t(1) << (log_n - 10)) : 1;
    std::vector<uint32_t> lo(1024), hi(n_hi);
    F x = F::one();
    for (size_t i = 0; i < 1024; ++i) {
      lo[i] = x.v;
      x *= root;
    }
    F step = x;  // root^1024
    x = F::one();
    for (size_t i = 0; i < n_hi; ++i) {
      hi[i] = x.v;
      x *= step;
    }
    DevBuf dlo(1024), dhi(n_hi);
    P3R_HIP(copy_sync(ctx->stream, dlo.p, lo.data(), 1024 * 4, hipMemcpyHostToDevice));
    P3R_HIP(copy_sync(ctx->stream, dhi.p, hi.data(), n_hi * 4, hipMemcpyHostToDevice));
    it = ctx->tw4.emplace(key, std::make_pair(std::move(dlo), std::move(dhi))).first;
  }
  return {it->second.first.p, it->second.second.p};
}

// Per-coset input scaling for the forward pass: output block z of the bit-reversed LDE is
// the coset shift * w_{N<<b}^{bitrev_b(z)} * <w_N>, so cell k of the coefficient vector is
// multiplied by s_z^k = s_z^{N2*n1} * s_z^{n2}.
template <class PP>
std::pair<const uint32_t*, const uint32_t*> get_pre(p3r_ctx* ctx, int log_n, int log_n1,
                                                    int log_n2, int added_bits, uint32_t shift) {
  auto key = std::make_tuple(log_n, added_bits, shift);
  auto it = ctx->pre.find(key);
  if (it == ctx->pre.end()) {
    using F = Fp<PP>;
    const size_t B = size_t(1) << added_bits, N1 = size_t(1) << log_n1, N2 = size_t(1) << log_n2;
    std::vector<uint32_t> a(B * N1), b(B * N2);
    F wbig = F::two_adic_generator(log_n + added_bits);
    for (size_t z = 0; z < B; ++z) {
      F s = F::from_canonical(shift) * wbig.pow(bit_reverse((uint32_t)z, added_bits));
      F x = F::one();
      for (size_t i = 0; i < N2; ++i) {
        b[z * N2 + i] = x.v;
        x *= s;
      }
      F step = x;  // s^N2
      x = F::one();
      for (size_t i = 0; i < N1; ++i) {
        a[z * N1 + i] = x.v;
        x *= step;
      }
    }
    DevBuf da(a.size()), db(b.size());
    P3R_HIP(copy_sync(ctx->stream, da.p, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    P3R_HIP(copy_sync(ctx->stream, db.p, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    it = ctx->pre.emplace(key, std::make_pair(std::move(da), std::move(db))).first;
  }
  return {it->second.first.p, it->second.second.p};
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

// One pass of one matrix inside a job-list launch.
struct NttJob {
  NttPass pass;
  size_t ncols, ncosets;
};
// Runs the listed passes in ONE launch (they must be independent of each other).
template <class PP>
void launch_ntt(p3r_ctx* ctx, std::vector<NttJob>& jobs, const char* name) {
  if (jobs.empty()) return;
  static const int log_tile = tuning_knob("P3R_NTT_LOG_TILE") ? atoi(tuning_knob("P3R_NTT_LOG_TILE")) : 13;  // 2^13 cells, 512 lanes: 4 tiles per CU overlap their phases
  std::vector<NttPass> passes;
  size_t lds_max = 0;
  unsigned threads_max = 64;
  uint64_t blocks = 0;
  for (NttJob& j : jobs) {
    NttPass& a = j.pass;
    const int log_r = a.sub_dim == 0 ? a.log_n1 : a.log_n2;
    const int log_lines = a.sub_dim == 0 ? a.log_n2 : a.log_n1;
    int log_t = std::max(0, std::min(log_tile, 13) - log_r);
    if (a.sub_dim == 0 && log_t > 5) log_t = 5;  // 128-byte segments are enough when strided
    log_t = std::min(log_t, log_lines);
    a.log_t = log_t;
    const size_t R = size_t(1) << log_r, T = size_t(1) << log_t;
    const size_t lds = (R * (T + 1) + (R >> 5) + 2 + R + 2) * sizeof(uint32_t);
    if (lds > 160 * 1024) fail(P3R_EUNSUPPORTED, "NTT tile of 2^%d rows does not fit LDS", log_r);
    lds_max = std::max(lds_max, lds);
    threads_max = std::max(threads_max, (unsigned)std::min<size_t>(kNttBlock, (R * T) >> 4));
    a.block0 = (uint32_t)blocks;
    a.log_gx = log_lines - log_t;
    a.log_gz = log2_exact(j.ncosets, "coset count");
    blocks += (uint64_t)j.ncols << (a.log_gx + a.log_gz);
    passes.push_back(a);
  }
  if (blocks >= (uint64_t(1) << 31)) fail(P3R_EUNSUPPORTED, "NTT launch of %llu tiles", (unsigned long long)blocks);
  const auto* d_jobs = static_cast<const NttPass*>(const_table(ctx, passes.data(), passes.size() * sizeof(NttPass)));
  ProfScope ps(ctx, name);
  hipLaunchKernelGGL(k_ntt_tile<PP>, dim3((unsigned)blocks), dim3(threads_max), lds_max, ctx->stream, d_jobs,
                     (int)passes.size());
  P3R_HIP(hipGetLastError());
}

// The lean forward passes (kernels_ntt2.cuh): jobs grouped by the compile-time sub-transform size.
template <class PP, int LOG_R, int MODE, int LOG_TILE>
void launch_col_r(p3r_ctx* ctx, std::vector<NttColJob>& jobs, uint32_t blocks) {
  const auto* d = static_cast<const NttColJob*>(const_table(ctx, jobs.data(), jobs.size() * sizeof(NttColJob)));
  ProfScope ps(ctx, MODE == NTT2_FWD ? "ntt_forward_1" : MODE == NTT2_INV1 ? "ntt_inverse_1" : "ntt_inverse_2");
  hipLaunchKernelGGL((k_ntt_col<PP, LOG_R, MODE, LOG_TILE>), dim3(blocks), dim3(kNtt2Lanes), 0, ctx->stream, d, (int)jobs.size());
  P3R_HIP(hipGetLastError());
}
template <class PP, int LOG_R, int LOG_TILE>
void launch_fwd_line_r(p3r_ctx* ctx, std::vector<NttLineJob>& jobs, uint32_t blocks) {
  const auto* d = static_cast<const NttLineJob*>(const_table(ctx, jobs.data(), jobs.size() * sizeof(NttLineJob)));
  ProfScope ps(ctx, "ntt_forward_2");
  hipLaunchKernelGGL((k_ntt_fwd_line<PP, LOG_R, LOG_TILE>), dim3(blocks), dim3(1u << (LOG_TILE - 4)), 0, ctx->stream, d, (int)jobs.size());
  P3R_HIP(hipGetLastError());
}
constexpr int kNtt2MinLogR = 5, kNtt2MaxLogR = 12, kNtt2MaxLineLogR = 13;
// jobs grouped by (sub-transform size, tile size): key = log_r * 2 + (log_tile - 13)
// Several sub-transform sizes, all on 2^13-cell tiles and few workgroups in total (the tables of a small
// layer): one launch of the mixed-size kernel instead of one per size.
template <class JOB>
bool merge_small_launches(std::map<int, std::pair<std::vector<JOB>, uint64_t>>& by_r, std::vector<JOB>& all, uint32_t& blocks) {
  static const bool off = tuning_knob("P3R_NTT_NO_MIXED") != nullptr;
  if (off || by_r.size() < 2) return false;
  uint64_t total = 0;
  for (auto& kv : by_r) {
    if (kv.first & 1) return false;  // a 2^14-cell column tile / a 2^13-cell line tile: own launch
    total += kv.second.second;
  }
  if (total > kNtt2MixedMaxBlocks) return false;
  uint32_t base = 0;
  for (auto& kv : by_r) {
    for (JOB j : kv.second.first) {
      j.block0 += base;
      all.push_back(j);
    }
    base += (uint32_t)kv.second.second;
  }
  blocks = base;
  return true;
}
template <class PP, int MODE>
void launch_col(p3r_ctx* ctx, std::map<int, std::pair<std::vector<NttColJob>, uint64_t>>& by_r) {
  {
    std::vector<NttColJob> all;
    uint32_t blocks = 0;
    if (merge_small_launches(by_r, all, blocks)) {
      const auto* d = static_cast<const NttColJob*>(const_table(ctx, all.data(), all.size() * sizeof(NttColJob)));
      ProfScope ps(ctx, MODE == NTT2_FWD ? "ntt_forward_1" : MODE == NTT2_INV1 ? "ntt_inverse_1" : "ntt_inverse_2");
      hipLaunchKernelGGL((k_ntt_col_mixed<PP, MODE>), dim3(blocks), dim3(kNtt2Lanes), 0, ctx->stream, d, (int)all.size());
      P3R_HIP(hipGetLastError());
      return;
    }
  }
  for (auto& kv : by_r) {
    auto& jobs = kv.second.first;
    if (kv.second.second >= (uint64_t(1) << 31)) fail(P3R_EUNSUPPORTED, "NTT launch of %llu tiles", (unsigned long long)kv.second.second);
    const uint32_t blocks = (uint32_t)kv.second.second;
    const int log_r = kv.first >> 1, big = kv.first & 1;
#define P3R_COL_CASE(R)                                                                         \
  case R:                                                                                       \
    if (big) launch_col_r<PP, R, MODE, 14>(ctx, jobs, blocks); \
    else launch_col_r<PP, R, MODE, 13>(ctx, jobs, blocks);                                      \
    break;
    switch (log_r) {
      P3R_COL_CASE(5) P3R_COL_CASE(6) P3R_COL_CASE(7) P3R_COL_CASE(8) P3R_COL_CASE(9) P3R_COL_CASE(10) P3R_COL_CASE(11)
      P3R_COL_CASE(12)
      default: fail(P3R_EUNSUPPORTED, "NTT column pass of 2^%d rows", log_r);
    }
#undef P3R_COL_CASE
  }
}
template <class PP>
void launch_fwd_line(p3r_ctx* ctx, std::map<int, std::pair<std::vector<NttLineJob>, uint64_t>>& by_r) {
  {
    // the line map's low key bit is set for 2^12-cell tiles (the mixed kernel's), clear for 2^13-cell ones
    std::map<int, std::pair<std::vector<NttLineJob>, uint64_t>> flipped;
    bool all_small = true;
    for (auto& kv : by_r) all_small = all_small && (kv.first & 1);
    std::vector<NttLineJob> all;
    uint32_t blocks = 0;
    if (all_small) {
      for (auto& kv : by_r) flipped[kv.first ^ 1] = kv.second;
      if (merge_small_launches(flipped, all, blocks)) {
        const auto* d = static_cast<const NttLineJob*>(const_table(ctx, all.data(), all.size() * sizeof(NttLineJob)));
        ProfScope ps(ctx, "ntt_forward_2");
        hipLaunchKernelGGL((k_ntt_fwd_line_mixed<PP>), dim3(blocks), dim3(256), 0, ctx->stream, d, (int)all.size());
        P3R_HIP(hipGetLastError());
        return;
      }
    }
  }
  for (auto& kv : by_r) {
    auto& jobs = kv.second.first;
    if (kv.second.second >= (uint64_t(1) << 31)) fail(P3R_EUNSUPPORTED, "NTT launch of %llu tiles", (unsigned long long)kv.second.second);
    const uint32_t blocks = (uint32_t)kv.second.second;
    const int log_r = kv.first >> 1, small = kv.first & 1;
#define P3R_LINE_CASE(R)                                            \
  case R:                                                           \
    if (small) launch_fwd_line_r<PP, R, 12>(ctx, jobs, blocks);     \
    else launch_fwd_line_r<PP, R, 13>(ctx, jobs, blocks);           \
    break;
    switch (log_r) {
      P3R_LINE_CASE(5) P3R_LINE_CASE(6) P3R_LINE_CASE(7) P3R_LINE_CASE(8) P3R_LINE_CASE(9) P3R_LINE_CASE(10)
      P3R_LINE_CASE(11) P3R_LINE_CASE(12)
      case 13: launch_fwd_line_r<PP, 13, 13>(ctx, jobs, blocks); break;
      default: fail(P3R_EUNSUPPORTED, "forward NTT line pass of 2^%d cells", log_r);
    }
#undef P3R_LINE_CASE
  }
}

// K5 for a batch of matrices (all tables of a commit): every matrix goes through the same passes,
// and pass k of all of them is one launch.
// in: h x w evaluations over the subgroup (natural order, column-major Montgomery).
// Returns (h << added_bits) x w, rows in bit-reversed order over shift * <w_{h<<added_bits}>.
struct LdeItem {
  const p3r_dmat* in;
  uint32_t shift;  // canonical coset shift
};
template <class PP>
std::vector<std::unique_ptr<p3r_dmat>> coset_lde_batch(p3r_ctx* ctx, const std::vector<LdeItem>& items,
                                                       int added_bits) {
  using F = Fp<PP>;
  const size_t B = size_t(1) << added_bits;
  std::vector<std::unique_ptr<p3r_dmat>> outs;
  std::vector<DevBuf> scratch;  // coefficient vectors and transposition buffers
  // phase 1/2: inverse transform (small matrices: 1 = inverse, 2 = forward); 3/4: forward of the rest
  std::vector<NttJob> phase[4];
  std::map<int, std::pair<std::vector<NttColJob>, uint64_t>> fwd_col;    // sub-transform size -> (jobs, blocks)
  std::map<int, std::pair<std::vector<NttLineJob>, uint64_t>> fwd_line;
  std::map<int, std::pair<std::vector<NttColJob>, uint64_t>> inv1, inv2;
  static const bool lean_fwd = !tuning_knob("P3R_NTT_OLD");
  static const int fwd_la_cap = tuning_knob("P3R_NTT_FWD_LOG_N1") ? atoi(tuning_knob("P3R_NTT_FWD_LOG_N1")) : 8;
  for (const LdeItem& it : items) {
    const p3r_dmat* in = it.in;
    const uint32_t shift = it.shift;
    const int log_n = log2_exact(in->h, "LDE input height");
    if (log_n + added_bits > PP::TWO_ADICITY)
      fail(P3R_EINVAL, "LDE of 2^%d rows exceeds the field's two-adicity (%d)", log_n + added_bits,
           PP::TWO_ADICITY);
    if (shift == 0 || shift >= PP::P) fail(P3R_EINVAL, "coset shift must be a non-zero canonical element");
    const size_t N = in->h, w = in->w;
    outs.push_back(dmat_alloc(N * B, w));
    p3r_dmat* out = outs.back().get();
    scratch.emplace_back(N * w);
    uint32_t* coef = scratch.back().p;
    const uint32_t inv_n = F::from_canonical((uint32_t)(N % PP::P)).inv().v;

    NttPass p{};
    if (log_n <= 11) {
      // single pass each way: whole polynomial in one LDS tile
      p.in = in->d; p.out = coef;
      p.in_col_stride = N; p.out_col_stride = N; p.out_coset_stride = 0;
      p.log_n1 = 0; p.log_n2 = log_n; p.sub_dim = 1; p.out_mode = 1;
      p.tw_sub = get_tw_sub<PP>(ctx, log_n, 1); p.inverse = 1;
      p.scale = inv_n; p.use_scale = 1;
      phase[0].push_back({p, w, 1});
      auto pre = get_pre<PP>(ctx, log_n, 0, log_n, added_bits, shift);
      p = NttPass{};
      p.in = coef; p.out = out->d;
      p.in_col_stride = N; p.out_col_stride = N * B; p.out_coset_stride = N;
      p.log_n1 = 0; p.log_n2 = log_n; p.sub_dim = 1; p.out_mode = 0;
      p.tw_sub = get_tw_sub<PP>(ctx, log_n, 0);
      p.pre_a = pre.first; p.pre_b = pre.second;
      phase[1].push_back({p, w, B});
      continue;
    }
    const int la = log_n / 2, lb = log_n - la;  // N1 = 2^la (strided dim), N2 = 2^lb
    scratch.emplace_back(N * w);
    uint32_t* tmp = scratch.back().p;
    auto tw4i = get_tw4<PP>(ctx, log_n, 1);
    const bool lean_inv = lean_fwd && la >= kNtt2MinLogR && lb <= kNtt2MaxLogR && la >= kNtt2LogTile - lb && lb >= kNtt2LogTile - la;
    if (lean_inv) {
      NttColJob j1{};
      j1.in = in->d; j1.out = tmp;
      j1.tw = get_tw_sub<PP>(ctx, la, 1);
      j1.tw4_lo = tw4i.first; j1.tw4_hi = tw4i.second;
      j1.in_col_stride = N; j1.out_col_stride = N;
      j1.log_n2 = lb; j1.log_r = la;
      // 2^14-cell tiles (two items per lane) when the 2^13 tile would be narrower than 16 columns
      const int big1 = (kNtt2LogTile - la < 4 && lb >= kNtt2LogTile + 1 - la) ? 1 : 0;
      auto& q1 = inv1[la * 2 + big1];
      j1.block0 = (uint32_t)q1.second;
      q1.second += (uint64_t)w << (lb - (kNtt2LogTile + big1 - la));
      q1.first.push_back(j1);
      NttColJob j2{};   // tmp viewed as [N2 rows][N1]: size-N2 transforms along the rows
      j2.in = tmp; j2.out = coef;
      j2.tw = get_tw_sub<PP>(ctx, lb, 1);
      j2.in_col_stride = N; j2.out_col_stride = N;
      j2.log_n2 = la; j2.log_r = lb;
      j2.scale = inv_n;
      const int big2 = (kNtt2LogTile - lb < 4 && la >= kNtt2LogTile + 1 - lb) ? 1 : 0;
      auto& q2 = inv2[lb * 2 + big2];
      j2.block0 = (uint32_t)q2.second;
      q2.second += (uint64_t)w << (la - (kNtt2LogTile + big2 - lb));
      q2.first.push_back(j2);
    } else {
    // inverse pass 1: size-N1 transforms along n1, twiddle, transposed store tmp[n2*N1 + k1]
    p.in = in->d; p.out = tmp;
    p.in_col_stride = N; p.out_col_stride = N;
    p.log_n1 = la; p.log_n2 = lb; p.sub_dim = 0; p.out_mode = 2;
    p.tw_sub = get_tw_sub<PP>(ctx, la, 1); p.inverse = 1;
    p.tw4_lo = tw4i.first; p.tw4_hi = tw4i.second;
    phase[0].push_back({p, w, 1});
    // inverse pass 2: tmp viewed as [N2][N1]; size-N2 transforms along its first dim,
    // natural row order -> coefficient k1 + N1*k2 lands at k2*N1 + k1
    p = NttPass{};
    p.in = tmp; p.out = coef;
    p.in_col_stride = N; p.out_col_stride = N;
    p.log_n1 = lb; p.log_n2 = la; p.sub_dim = 0; p.out_mode = 1;
    p.tw_sub = get_tw_sub<PP>(ctx, lb, 1); p.inverse = 1;
    p.scale = inv_n; p.use_scale = 1;
    phase[1].push_back({p, w, 1});
    }
    // forward pass 1 (all cosets): scale by s_z^k, size-N1 transforms along n1, twiddle, in place rows.
    // The forward transform has its own split: its strided pass wants few rows per tile (long
    // contiguous segments per row), its second pass is contiguous whatever N2 is.
    // (measured: 2^8 x 2^12 beats 2^10 x 2^10 at n = 2^20; past 2^12 contiguous points per line the
    // balanced split is better again)
    // With the lean kernels the contiguous pass takes lines of up to 2^13 cells (one tile), so the strided
    // pass keeps 2^8 rows (128-byte segments) up to 2^21 rows and grows only beyond that (2^22: 2^9 rows,
    // 64-byte segments; the balanced 2^11 x 2^11 split moved 16-byte segments).
    const int la_f = lean_fwd ? std::max(std::min(log_n / 2, fwd_la_cap), log_n - kNtt2MaxLineLogR)
                              : (log_n - fwd_la_cap <= 12 ? std::min(log_n / 2, fwd_la_cap) : log_n / 2);
    const int lb_f = log_n - la_f;
    auto pre = get_pre<PP>(ctx, log_n, la_f, lb_f, added_bits, shift);
    auto tw4f = get_tw4<PP>(ctx, log_n, 0);
    if (lean_fwd && la_f >= kNtt2MinLogR && la_f <= kNtt2MaxLogR && lb_f >= kNtt2MinLogR && lb_f <= kNtt2MaxLineLogR &&
        lb_f >= kNtt2LogTile - la_f) {
      // lean kernels (kernels_ntt2.cuh): the same two passes with compile-time geometry
      NttColJob cj{};
      cj.in = coef; cj.out = out->d;
      cj.tw = get_tw_sub<PP>(ctx, la_f, 0);
      cj.tw4_lo = tw4f.first; cj.tw4_hi = tw4f.second;
      cj.pre_a = pre.first; cj.pre_b = pre.second;
      cj.in_col_stride = N; cj.out_col_stride = N * B; cj.out_coset_stride = N;
      cj.log_n2 = lb_f; cj.log_cosets = added_bits; cj.log_r = la_f;
      // 2^14-cell tiles when the 2^13 tile would be narrower than 32 columns (measured: slower at 2^8 rows
      // x 32 columns, faster from 2^9 rows on)
      const int bigf = (kNtt2LogTile - la_f < 5 && lb_f >= kNtt2LogTile + 1 - la_f) ? 1 : 0;
      auto& fc = fwd_col[la_f * 2 + bigf];
      cj.block0 = (uint32_t)fc.second;
      {
        static const bool no_xcd = tuning_knob("P3R_NTT_NO_XCD_MAP") != nullptr;
        const uint64_t tiles = (uint64_t)w << (lb_f - (kNtt2LogTile + bigf - la_f));
        cj.xcd_map = (!no_xcd && added_bits > 0 && (cj.block0 & 7) == 0 && (tiles & 7) == 0) ? 1 : 0;
      }
      fc.second += (uint64_t)w << (lb_f - (kNtt2LogTile + bigf - la_f) + added_bits);
      fc.first.push_back(cj);
      NttLineJob lj{};
      lj.data = out->d;
      lj.tw = get_tw_sub<PP>(ctx, lb_f, 0);
      lj.log_r = (uint32_t)lb_f;
      // lines of up to 2^12 cells on 2^12-cell tiles (256 lanes, six workgroups per CU): measured 10 % faster
      // than 2^13-cell tiles at the same waves per CU - the pass is VALU-bound (it does not slow down with
      // a third fewer waves) and smaller workgroups wait less at their barriers.  P3R_NTT_LINE_LOG_TILE=13: tuning
      static const int line_log_tile = tuning_knob("P3R_NTT_LINE_LOG_TILE") ? atoi(tuning_knob("P3R_NTT_LINE_LOG_TILE")) : 12;
      const int small = (line_log_tile == 12 && lb_f <= 12) ? 1 : 0;
      auto& fl = fwd_line[lb_f * 2 + small];
      lj.block0 = (uint32_t)fl.second;
      fl.second += ((uint64_t)w * N * B) >> (kNtt2LogTile - small);
      fl.first.push_back(lj);
      continue;
    }
    p = NttPass{};
    p.in = coef; p.out = out->d;
    p.in_col_stride = N; p.out_col_stride = N * B; p.out_coset_stride = N;
    p.log_n1 = la_f; p.log_n2 = lb_f; p.sub_dim = 0; p.out_mode = 0;
    p.tw_sub = get_tw_sub<PP>(ctx, la_f, 0);
    p.tw4_lo = tw4f.first; p.tw4_hi = tw4f.second;
    p.pre_a = pre.first; p.pre_b = pre.second;
    phase[2].push_back({p, w, B});
    // forward pass 2: contiguous size-N2 transforms, in place, bit-reversed rows kept.
    // The B cosets of a column are contiguous, so they are just B*N1 lines of N2 cells.
    p = NttPass{};
    p.in = out->d; p.out = out->d;
    p.in_col_stride = N * B; p.out_col_stride = N * B;
    p.log_n1 = la_f + added_bits; p.log_n2 = lb_f; p.sub_dim = 1; p.out_mode = 0;
    p.tw_sub = get_tw_sub<PP>(ctx, lb_f, 0);
    phase[3].push_back({p, w, 1});
  }
  launch_ntt<PP>(ctx, phase[0], "ntt_inverse_1");
  launch_col<PP, NTT2_INV1>(ctx, inv1);
  launch_ntt<PP>(ctx, phase[1], "ntt_inverse_2");
  launch_col<PP, NTT2_INV2>(ctx, inv2);
  launch_ntt<PP>(ctx, phase[2], "ntt_forward_1");
  launch_col<PP, NTT2_FWD>(ctx, fwd_col);
  launch_ntt<PP>(ctx, phase[3], "ntt_forward_2");
  launch_fwd_line<PP>(ctx, fwd_line);
  return outs;
}
template <class PP>
std::unique_ptr<p3r_dmat> coset_lde(p3r_ctx* ctx, const p3r_dmat* in, int added_bits, uint32_t shift) {
  return std::move(coset_lde_batch<PP>(ctx, {{in, shift}}, added_bits)[0]);
}

// ------------------------------------------------------------------ MMCS
// Row digests of several height classes in one launch: classes[c] = the matrices of one height
// (their rows are concatenated in the given order), digs[c] = [8][h_c].
template <class PP>
void hash_rows(p3r_ctx* ctx, const std::vector<std::vector<const p3r_dmat*>>& classes,
               const std::vector<uint32_t*>& digs) {
  std::vector<HashRowsJob> jobs;
  for (size_t c = 0; c < classes.size(); ++c) {
    std::vector<const uint32_t*> cols;
    for (const p3r_dmat* m : classes[c])
      for (size_t k = 0; k < m->w; ++k) cols.push_back(m->d + k * m->h);
    HashRowsJob j{};
    j.cols = col_table(ctx, cols);
    j.dig = digs[c];
    j.h = classes[c][0]->h;
    j.wtot = (int)cols.size();
    jobs.push_back(j);
  }
  // widest rows first: their blocks run longest
  std::stable_sort(jobs.begin(), jobs.end(),
                   [](const HashRowsJob& a, const HashRowsJob& b) { return a.wtot > b.wtot; });
  uint32_t blocks = 0;
  for (auto& j : jobs) {
    j.block0 = blocks;
    blocks += blocks_for(j.h);
  }
  const auto* d_jobs =
      static_cast<const HashRowsJob*>(const_table(ctx, jobs.data(), jobs.size() * sizeof(HashRowsJob)));
  ProfScope ps(ctx, "mmcs_hash_rows");
  hipLaunchKernelGGL(k_mmcs_hash_rows<PP>, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_jobs, (int)jobs.size(),
                     ctx->rcd());
  P3R_HIP(hipGetLastError());
}

// One 2-to-1 layer, one permutation per lane: for layers large enough to fill the chip.  Smaller
// ones are latency-bound and go through mmcs_subtree below (16 lanes per node, several levels per launch).
// A lane-cooperative permutation costs 16 lanes x ~1.2 k instructions against ~3.9 k FP64 operations of one
// lane: it is the faster way through a level only while the level is latency-bound, i.e. up to about
// one 16-lane row per SIMD and pass (4096 nodes a pass on 256 CUs; a pass is ~2.7 us, a launch of the
// one-permutation-per-lane kernel ~11 us whatever its size).  P3R_COOP_MAX_NODES / _LEAF_ROWS: tuning.
// (tuning knobs are rounded down to a power of two: the kernels index by shifts and halvings)
inline size_t env_pow2(const char* name, size_t dflt, size_t lo, size_t hi) {
  const char* e = tuning_knob(name);
  size_t v = e ? (size_t)atol(e) : dflt;
  v = std::min(std::max(v, lo), hi);
  while (v & (v - 1)) v &= v - 1;
  return v;
}
inline size_t coop_max_nodes() {
  static const size_t v = env_pow2("P3R_COOP_MAX_NODES", 16384, 1, size_t(1) << 30);
  return v;
}
inline size_t coop_max_leaf_rows() {
  static const size_t v = env_pow2("P3R_COOP_MAX_LEAF_ROWS", 8192, 1, size_t(1) << 30);
  return v;
}
// Digests per workgroup of a k_mmcs_subtree launch: with 32, level 0 is one pass of 16 rows - one wave
// per SIMD - and the five levels of the launch are all latency-bound; with 256 (eight levels per launch)
// levels 0 and 1 queued 8 and 4 waves per SIMD on the few CUs that had a workgroup.
inline size_t subtree_nodes() {
  static const size_t v = env_pow2("P3R_SUBTREE_NODES", 32, 2, kSubtreeNodes);
  return v;
}
template <class PP>
void launch_compress(p3r_ctx* ctx, const uint32_t* prev, const uint32_t* inj, uint32_t* out, size_t n) {
  ProfScope ps(ctx, "mmcs_compress");
  hipLaunchKernelGGL(k_mmcs_compress<PP>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, prev, inj, out, n,
                     ctx->rcd());
  P3R_HIP(hipGetLastError());
}

// log2(subtree_nodes()) levels above the `n`-digest layer at the back of `tree->layers` in one launch
// (k_mmcs_subtree), for layers small enough to be latency-bound.  `inject`: height -> digests to
// fold in at that height (may be null).  Returns the size of the new back layer, or `n` when
// the layer is too large for this path.
// `step`: FRI commit phase only - when this launch ends at the root (one workgroup, cap of one
// digest) the transcript step runs inside it and `step->done` is set.
struct TranscriptStep {
  uint32_t *state, *beta, *cap;
  bool done = false;
  int dc = 4;   // words of the folding challenge (the challenge degree)
};
template <class PP>
size_t mmcs_subtree(p3r_ctx* ctx, p3r_tree* tree, size_t n, const std::map<size_t, DevBuf>* inject,
                    TranscriptStep* step = nullptr) {
  const size_t cap_n = size_t(1) << tree->cap_height;
  if (n / 2 > coop_max_nodes() || n <= cap_n) return n;
  SubtreeArgs a{};
  a.in = tree->layers.back().p;
  a.n_in = (uint32_t)n;
  const size_t local = std::min<size_t>(n, subtree_nodes());
  a.local = (uint32_t)local;
  size_t nn = n, shrink = local;
  while (shrink > 1 && nn > cap_n && a.n_levels < kSubtreeLevels) {
    nn /= 2;
    shrink /= 2;
    tree->layers.emplace_back(P2_DIGEST * nn);
    a.out[a.n_levels] = tree->layers.back().p;
    if (inject) {
      auto it = inject->find(nn);
      if (it != inject->end()) a.inj[a.n_levels] = it->second.p;
    }
    ++a.n_levels;
  }
  if (step && nn == 1 && n == local) {
    a.t_state = step->state;
    a.t_beta = step->beta;
    a.t_cap = step->cap;
    a.t_dc = step->dc;
    step->done = true;
  }
  ProfScope ps(ctx, "mmcs_compress");
  const unsigned lanes = (unsigned)std::min<size_t>(std::max<size_t>(local * 8, 64), kSubtreeBlock);
  hipLaunchKernelGGL(k_mmcs_subtree<PP>, dim3((unsigned)(n / local)), dim3(lanes), 0, ctx->stream, a,
                     ctx->rc.p, ctx->p2_diag.p);
  P3R_HIP(hipGetLastError());
  return nn;
}

template <class PP>
void mmcs_commit(p3r_ctx* ctx, p3r_tree* tree, uint32_t* cap_out) {
  using F = Fp<PP>;
  const auto& mats = tree->mats;
  if (mats.empty()) fail(P3R_EINVAL, "MMCS commit needs at least one matrix");
  // tallest first, stable (recursion/src/pcs/mmcs.rs:355-425)
  std::vector<size_t> order(mats.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](size_t a, size_t b) { return mats[a]->h > mats[b]->h; });
  const size_t hmax = mats[order[0]]->h;
  tree->log_max_h = log2_exact(hmax, "matrix height");
  tree->cap_height = (int)ctx->cfg.cap_height;
  if (tree->cap_height > tree->log_max_h)
    fail(P3R_EINVAL, "cap_height %d exceeds log2 of the tallest matrix (%d)", tree->cap_height,
         tree->log_max_h);
  tree->total_width = 0;
  for (auto* m : mats) tree->total_width += m->w;

  auto at_height = [&](size_t h) {
    std::vector<const p3r_dmat*> v;
    for (size_t i : order)
      if (mats[i]->h == h) v.push_back(mats[i]);
    return v;
  };
  // leaf digests of every height class up front, in one launch
  std::vector<size_t> class_h;
  for (size_t i : order)
    if (class_h.empty() || class_h.back() != mats[i]->h) class_h.push_back(mats[i]->h);
  tree->layers.clear();
  tree->layers.emplace_back(P2_DIGEST * hmax);
  std::map<size_t, DevBuf> inject;  // height -> digests of the matrices of that height
  {
    std::vector<std::vector<const p3r_dmat*>> classes;
    std::vector<uint32_t*> digs;
    for (size_t h : class_h) {
      classes.push_back(at_height(h));
      if (h == hmax) digs.push_back(tree->layers[0].p);
      else digs.push_back(inject.emplace(h, DevBuf(P2_DIGEST * h)).first->second.p);
    }
    hash_rows<PP>(ctx, classes, digs);
  }
  size_t n = hmax;
  const size_t cap_n = size_t(1) << tree->cap_height;
  while (n > cap_n) {
    const size_t after = mmcs_subtree<PP>(ctx, tree, n, &inject);
    if (after != n) {
      n = after;
      continue;
    }
    const size_t nn = n / 2;
    DevBuf next(P2_DIGEST * nn);
    const uint32_t* prev = tree->layers.back().p;
    auto inj = inject.find(nn);
    launch_compress<PP>(ctx, prev, inj != inject.end() ? inj->second.p : nullptr, next.p, nn);
    tree->layers.push_back(std::move(next));
    n = nn;
  }
  // cap: digest-major canonical
  std::vector<uint32_t> soa(P2_DIGEST * cap_n);
  P3R_HIP(fetch_small(ctx, tree->layers.back().p, soa.size(), soa.data()));
  for (size_t j = 0; j < cap_n; ++j)
    for (int k = 0; k < P2_DIGEST; ++k)
      cap_out[j * P2_DIGEST + k] = F::raw(soa[(size_t)k * cap_n + j]).to_canonical();
}

template <class PP>
void mmcs_open(p3r_ctx* ctx, const p3r_tree* tree, size_t index, uint32_t* opened, uint32_t* proof) {
  using F = Fp<PP>;
  if (index >> tree->log_max_h) fail(P3R_EINVAL, "open index %zu out of range", index);
  size_t off = 0;
  for (const p3r_dmat* m : tree->mats) {
    int lh = log2_exact(m->h, "matrix height");
    size_t row = index >> (tree->log_max_h - lh);
    // strided gather of one row: w scattered 4-byte cells
    P3R_HIP(hipMemcpy2DAsync(opened + off, 4, m->d + row, m->h * 4, 4, m->w, hipMemcpyDeviceToHost,
                             ctx->stream));
    off += m->w;
  }
  const int depth = tree->log_max_h - tree->cap_height;
  for (int l = 0; l < depth; ++l) {
    size_t n = size_t(1) << (tree->log_max_h - l);
    size_t sib = (index >> l) ^ 1;
    P3R_HIP(hipMemcpy2DAsync(proof + (size_t)l * P2_DIGEST, 4, tree->layers[l].p + sib, n * 4, 4,
                             P2_DIGEST, hipMemcpyDeviceToHost, ctx->stream));
  }
  P3R_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < off; ++i) opened[i] = F::raw(opened[i]).to_canonical();
  for (size_t i = 0; i < (size_t)depth * P2_DIGEST; ++i) proof[i] = F::raw(proof[i]).to_canonical();
}

template <class PP>
void init_ctx(p3r_ctx* ctx) {
  using F = Fp<PP>;
  const size_t nrc = p2_num_constants<PP>();
  const uint32_t* src = ctx->cfg.poseidon2_rc;
  if (src) {
    if (ctx->cfg.poseidon2_rc_len != nrc)
      fail(P3R_EINVAL, "poseidon2_rc_len is %u, the field needs %zu constants",
           ctx->cfg.poseidon2_rc_len, nrc);
  } else {
    src = PP::FIELD_ID == 0 ? kDefaultRc_koala_bear : kDefaultRc_baby_bear;
  }
  ctx->rc_canonical.assign(src, src + nrc);
  std::vector<uint32_t> mont(nrc);
  for (size_t i = 0; i < nrc; ++i) {
    if (src[i] >= PP::P) fail(P3R_EINVAL, "round constant %zu is not canonical", i);
    mont[i] = F::from_canonical(src[i]).v;
  }
  ctx->rc_mont_host = mont;
  ctx->rc.alloc(nrc);
  P3R_HIP(copy_sync(ctx->stream, ctx->rc.p, mont.data(), nrc * 4, hipMemcpyHostToDevice));
  std::vector<double> rcd(src, src + nrc);
  ctx->rc_f64.alloc(2 * nrc);
  P3R_HIP(copy_sync(ctx->stream, ctx->rc_f64.p, rcd.data(), nrc * 8, hipMemcpyHostToDevice));
  {
    auto inv2k = [](int k) { return F::from_u64(uint64_t(1) << k).inv(); };
    const F two = F::from_canonical(2), three = F::from_canonical(3), four = F::from_canonical(4);
    F d[16] = {-two, F::one(), two, inv2k(1), three, four, -inv2k(1), -three, -four, inv2k(8), inv2k(3), inv2k(24),
               -inv2k(8), -inv2k(3), -inv2k(4), -inv2k(24)};
    if (PP::FIELD_ID == 1) {
      F b[16] = {-two, F::one(), two, inv2k(1), three, four, -inv2k(1), -three, -four, inv2k(8), inv2k(2), inv2k(3),
                 inv2k(27), -inv2k(8), -inv2k(4), -inv2k(27)};
      for (int i = 0; i < 16; ++i) d[i] = b[i];
    }
    uint32_t dm[16];
    for (int i = 0; i < 16; ++i) dm[i] = d[i].v;
    ctx->p2_diag.alloc(16);
    P3R_HIP(copy_sync(ctx->stream, ctx->p2_diag.p, dm, sizeof dm, hipMemcpyHostToDevice));
  }
  ctx->partial_rounds = PP::PARTIAL_ROUNDS;
  ctx->cfg.poseidon2_rc = nullptr;  // caller's pointer is not retained
  if (ctx->cfg.fri_log_arities) ctx->fri_log_arities.assign(ctx->cfg.fri_log_arities, ctx->cfg.fri_log_arities + ctx->cfg.fri_log_arities_len);
  ctx->cfg.fri_log_arities = nullptr;
  if (!ctx->proof_layout.set(ctx->cfg.proof_layout, ctx->cfg.proof_layout_len))
    fail(P3R_EINVAL, "proof_layout must be 18 bytes: three permutations batch[5] | fri[5] | opened[8]");
  ctx->cfg.proof_layout = nullptr;
  P3R_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_tile<PP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

}  // namespace

#include "verify_impl.h"
#include "prove_impl.cuh"
#include "layer_impl.cuh"
#include "circuit_impl.cuh"

// =============================================================================== C ABI
extern "C" {

p3r_ctx* p3r_create(const p3r_config* cfg) {
  p3r_ctx* ctx = nullptr;
  int rc = guard(nullptr, [&] {
    if (!cfg) fail(P3R_EINVAL, "cfg is NULL");
    if (cfg->abi_version != P3R_ABI_VERSION)
      fail(P3R_EINVAL, "abi_version %u != %u", cfg->abi_version, P3R_ABI_VERSION);
    if (cfg->field != P3R_FIELD_KOALA_BEAR && cfg->field != P3R_FIELD_BABY_BEAR)
      fail(P3R_EUNSUPPORTED, "unsupported field id %u", cfg->field);
    // circuit extension degree: 4 (binomial) on both fields; 5 = the KoalaBear quintic trinomial extension, proved
    // under the same D = 4 STARK configuration (batch_stark_prover/tests.rs:844-1029), primitive tables only
    if (cfg->challenge_degree != 0 && cfg->challenge_degree != 4 &&
        !(cfg->challenge_degree == 5 && cfg->field == P3R_FIELD_KOALA_BEAR))
      fail(P3R_EUNSUPPORTED, "UnsupportedChallengeDegree(%u): 4, or 5 over KoalaBear", cfg->challenge_degree);
    if (p3r::ext_degree_is_binomial_generic(cfg->ext_degree)) {
      // binomial extension x^D = W of degree 2 / 6 / 8: W is the caller's (BinomiallyExtendable<D>::W of its field
      // crate; the proof carries it as w_binomial)
      const uint32_t P = cfg->field == P3R_FIELD_KOALA_BEAR ? p3r::KoalaBearParams::P : p3r::BabyBearParams::P;
      if (cfg->ext_w == 0 || cfg->ext_w >= P) fail(P3R_EINVAL, "MissingWForExtension: ext_degree %u needs ext_w in 1..p-1", cfg->ext_degree);
    } else if (cfg->ext_degree != 1 && cfg->ext_degree != 4 && !(cfg->ext_degree == 5 && cfg->field == P3R_FIELD_KOALA_BEAR))
      fail(P3R_EUNSUPPORTED, "UnsupportedExtDegree(%u): 1, 2, 4, 6, 8, or 5 over KoalaBear", cfg->ext_degree);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
      fail(P3R_ENODEV, "no HIP device available (%s); this library has no CPU fallback",
           e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
      fail(P3R_ENODEV, "device %d out of range (%d devices)", cfg->device, ndev);
    P3R_HIP(hipSetDevice(cfg->device));
    auto c = std::make_unique<p3r_ctx>();
    c->cfg = *cfg;
    tls_pool() = c->pool;
    P3R_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    P3R_FIELD_CALL(c, init_ctx, c.get());
    ctx = c.release();
  });
  return rc == P3R_OK ? ctx : nullptr;
}

void p3r_destroy(p3r_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->cfg.device);
  (void)hipStreamSynchronize(ctx->stream);
  prof_clear(ctx);
  (void)hipStreamDestroy(ctx->stream);
  std::shared_ptr<DevPool> pool = ctx->pool;
  delete ctx;
  pool->trim();  // cached blocks go back to the driver; blocks still owned by live objects follow later
  if (tls_pool() == pool) tls_pool().reset();
}

const char* p3r_last_error(const p3r_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

uint32_t p3r_poseidon2_trace_width(const p3r_ctx* ctx) {
  return ctx->cfg.field == P3R_FIELD_KOALA_BEAR ? p2_perm_cols<KoalaBearParams>() + 2
                                                : p2_perm_cols<BabyBearParams>() + 2;
}
uint32_t p3r_poseidon2_num_constants(const p3r_ctx* ctx) {
  return ctx->cfg.field == P3R_FIELD_KOALA_BEAR ? p2_num_constants<KoalaBearParams>()
                                                : p2_num_constants<BabyBearParams>();
}
int p3r_poseidon2_round_constants(const p3r_ctx* ctx, uint32_t* out) {
  if (!ctx || !out) return P3R_EINVAL;
  std::copy(ctx->rc_canonical.begin(), ctx->rc_canonical.end(), out);
  return P3R_OK;
}

int p3r_trim(p3r_ctx* ctx, uint64_t* freed_bytes) {
  if (!ctx) return P3R_EINVAL;
  return guard(ctx, [&] {
    P3R_HIP(hipStreamSynchronize(ctx->stream));
    if (freed_bytes) {
      std::lock_guard<std::mutex> g(ctx->pool->mu);  // a sibling's out-of-memory path may be trimming this pool
      *freed_bytes = ctx->pool->cached_bytes;
    }
    ctx->pool->trim();
    ctx->const_tables.clear();  // the job lists were keyed to the addresses the pool handed out
  });
}

int p3r_sync(p3r_ctx* ctx) {
  return guard(ctx, [&] { P3R_HIP(hipStreamSynchronize(ctx->stream)); });
}

p3r_dmat* p3r_dmat_upload(p3r_ctx* ctx, const uint32_t* rowmajor, size_t h, size_t w) {
  p3r_dmat* out = nullptr;
  guard(ctx, [&] {
    if (!rowmajor) fail(P3R_EINVAL, "rowmajor is NULL");
    out = P3R_FIELD_CALL(ctx, upload, ctx, rowmajor, h, w).release();
  });
  return out;
}
p3r_dmat* p3r_dmat_alloc(p3r_ctx* ctx, size_t h, size_t w) {
  p3r_dmat* out = nullptr;
  guard(ctx, [&] { out = dmat_alloc(h, w).release(); });
  return out;
}
int p3r_dmat_download(p3r_ctx* ctx, const p3r_dmat* m, uint32_t* out) {
  return guard(ctx, [&] {
    if (!m || !out) fail(P3R_EINVAL, "NULL argument");
    P3R_FIELD_CALL(ctx, download, ctx, m, out);
  });
}
size_t p3r_dmat_height(const p3r_dmat* m) { return m->h; }
size_t p3r_dmat_width(const p3r_dmat* m) { return m->w; }
void p3r_dmat_free(p3r_ctx* ctx, p3r_dmat* m) {
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  delete m;
}

int p3r_poseidon2_permute_dmat(p3r_ctx* ctx, p3r_dmat* states) {
  return guard(ctx, [&] {
    if (!states) fail(P3R_EINVAL, "states is NULL");
    P3R_FIELD_CALL(ctx, permute_dmat, ctx, states);
  });
}

int p3r_poseidon2_permute_batch(p3r_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n) {
  return guard(ctx, [&] {
    if (!in || !out) fail(P3R_EINVAL, "NULL argument");
    if (n == 0) return;
    // pad the batch to a power of two of lanes; the tail rows are zeros and are dropped
    size_t np = 1;
    while (np < n) np <<= 1;
    std::vector<uint32_t> padded;
    const uint32_t* src = in;
    if (np != n) {
      padded.assign(np * P2_WIDTH, 0);
      memcpy(padded.data(), in, n * P2_WIDTH * 4);
      src = padded.data();
    }
    auto m = P3R_FIELD_CALL(ctx, upload, ctx, src, np, (size_t)P2_WIDTH);
    P3R_FIELD_CALL(ctx, permute_dmat, ctx, m.get());
    if (np != n) {
      P3R_FIELD_CALL(ctx, download, ctx, m.get(), padded.data());
      memcpy(out, padded.data(), n * P2_WIDTH * 4);
    } else {
      P3R_FIELD_CALL(ctx, download, ctx, m.get(), out);
    }
  });
}

p3r_p2_dev* p3r_p2_rows_upload(p3r_ctx* ctx, const p3r_p2_rows* rows) {
  p3r_p2_dev* out = nullptr;
  guard(ctx, [&] {
    if (!rows) fail(P3R_EINVAL, "rows is NULL");
    out = P3R_FIELD_CALL(ctx, p2_rows_upload, ctx, rows).release();
  });
  return out;
}
void p3r_p2_rows_free(p3r_ctx* ctx, p3r_p2_dev* rows) {
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  delete rows;
}
p3r_dmat* p3r_poseidon2_trace_fill_dev(p3r_ctx* ctx, const p3r_p2_dev* rows) {
  p3r_dmat* out = nullptr;
  guard(ctx, [&] {
    if (!rows) fail(P3R_EINVAL, "rows is NULL");
    out = P3R_FIELD_CALL(ctx, trace_fill, ctx, rows).release();
  });
  return out;
}
p3r_dmat* p3r_poseidon2_trace_fill_dmat(p3r_ctx* ctx, const p3r_p2_rows* rows) {
  p3r_dmat* out = nullptr;
  guard(ctx, [&] {
    if (!rows) fail(P3R_EINVAL, "rows is NULL");
    auto d = P3R_FIELD_CALL(ctx, p2_rows_upload, ctx, rows);
    out = P3R_FIELD_CALL(ctx, trace_fill, ctx, d.get()).release();
  });
  return out;
}

int p3r_poseidon2_trace_fill(p3r_ctx* ctx, const p3r_p2_rows* rows, uint32_t* trace_out) {
  return guard(ctx, [&] {
    if (!rows || !trace_out) fail(P3R_EINVAL, "NULL argument");
    auto d = P3R_FIELD_CALL(ctx, p2_rows_upload, ctx, rows);
    auto t = P3R_FIELD_CALL(ctx, trace_fill, ctx, d.get());
    P3R_FIELD_CALL(ctx, download, ctx, t.get(), trace_out);
  });
}

p3r_dmat* p3r_coset_lde_dmat(p3r_ctx* ctx, const p3r_dmat* evals, uint32_t added_bits,
                             uint32_t shift) {
  p3r_dmat* out = nullptr;
  guard(ctx, [&] {
    if (!evals) fail(P3R_EINVAL, "evals is NULL");
    out = P3R_FIELD_CALL(ctx, coset_lde, ctx, evals, (int)added_bits, shift).release();
  });
  return out;
}

int p3r_coset_lde(p3r_ctx* ctx, const uint32_t* evals, size_t h, size_t w, uint32_t added_bits,
                  uint32_t shift, uint32_t* out) {
  return guard(ctx, [&] {
    if (!evals || !out) fail(P3R_EINVAL, "NULL argument");
    auto in = P3R_FIELD_CALL(ctx, upload, ctx, evals, h, w);
    auto lde = P3R_FIELD_CALL(ctx, coset_lde, ctx, in.get(), (int)added_bits, shift);
    P3R_FIELD_CALL(ctx, download, ctx, lde.get(), out);
  });
}

int p3r_mmcs_commit_dmat(p3r_ctx* ctx, const p3r_dmat* const* mats, size_t n_mats,
                         uint32_t* cap_out, p3r_tree** tree_out) {
  return guard(ctx, [&] {
    if (!mats || !cap_out || n_mats == 0) fail(P3R_EINVAL, "bad arguments");
    auto tree = std::make_unique<p3r_tree>();
    tree->mats.assign(mats, mats + n_mats);
    P3R_FIELD_CALL(ctx, mmcs_commit, ctx, tree.get(), cap_out);
    if (tree_out) *tree_out = tree.release();
  });
}

int p3r_mmcs_commit(p3r_ctx* ctx, const p3r_matrix* mats, size_t n_mats, uint32_t* cap_out,
                    p3r_tree** tree_out) {
  return guard(ctx, [&] {
    if (!mats || !cap_out || n_mats == 0) fail(P3R_EINVAL, "bad arguments");
    auto tree = std::make_unique<p3r_tree>();
    for (size_t i = 0; i < n_mats; ++i) {
      if (!mats[i].values) fail(P3R_EINVAL, "matrix %zu has NULL values", i);
      tree->owned.push_back(
          P3R_FIELD_CALL(ctx, upload, ctx, mats[i].values, mats[i].height, mats[i].width));
      tree->mats.push_back(tree->owned.back().get());
    }
    P3R_FIELD_CALL(ctx, mmcs_commit, ctx, tree.get(), cap_out);
    if (tree_out) *tree_out = tree.release();
  });
}

int p3r_mmcs_open(p3r_ctx* ctx, const p3r_tree* tree, size_t index, uint32_t* opened_values,
                  uint32_t* proof_out) {
  return guard(ctx, [&] {
    if (!tree || !opened_values || !proof_out) fail(P3R_EINVAL, "NULL argument");
    P3R_FIELD_CALL(ctx, mmcs_open, ctx, tree, index, opened_values, proof_out);
  });
}
size_t p3r_tree_log_max_height(const p3r_tree* t) { return (size_t)t->log_max_h; }
size_t p3r_tree_total_width(const p3r_tree* t) { return t->total_width; }
void p3r_tree_free(p3r_ctx* ctx, p3r_tree* tree) {
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  delete tree;
}

int p3r_time_permute_dmat(p3r_ctx* ctx, p3r_dmat* states, int iters, double* ms_per_launch) {
  return guard(ctx, [&] {
    if (!states || !ms_per_launch || iters <= 0) fail(P3R_EINVAL, "bad arguments");
    hipEvent_t a, b;
    P3R_HIP(hipEventCreate(&a));
    P3R_HIP(hipEventCreate(&b));
    P3R_FIELD_CALL(ctx, permute_dmat, ctx, states);  // warm-up
    P3R_HIP(hipEventRecord(a, ctx->stream));
    for (int i = 0; i < iters; ++i) P3R_FIELD_CALL(ctx, permute_dmat, ctx, states);
    P3R_HIP(hipEventRecord(b, ctx->stream));
    P3R_HIP(hipEventSynchronize(b));
    float ms = 0;
    P3R_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_per_launch = (double)ms / iters;
  });
}

p3r_prep* p3r_prep_create(p3r_ctx* ctx, const p3r_air_desc* airs, const p3r_matrix* prep_mats,
                          size_t n_instances, uint32_t* commit_out) {
  p3r_prep* out = nullptr;
  guard(ctx, [&] {
    if (!airs || !prep_mats || !commit_out || n_instances == 0) fail(P3R_EINVAL, "bad arguments");
    for (size_t i = 0; i < n_instances; ++i)
      if (!prep_mats[i].values) fail(P3R_EINVAL, "preprocessed matrix %zu has NULL values", i);
    auto prep = P3R_FIELD_CALL(ctx, prep_create, ctx, airs, prep_mats, n_instances);
    std::copy(prep->cap_canonical.begin(), prep->cap_canonical.end(), commit_out);
    out = prep.release();
  });
  return out;
}
void p3r_prep_free(p3r_ctx* ctx, p3r_prep* prep) {
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  delete prep;
}

static int emit_proof(std::vector<uint8_t>&& bytes, uint8_t* buf, size_t cap, size_t* len) {
  *len = bytes.size();
  if (bytes.size() > cap) fail(P3R_EBUFFER, "proof needs %zu bytes, buffer holds %zu", bytes.size(), cap);
  memcpy(buf, bytes.data(), bytes.size());
  return 0;
}

int p3r_prove_batch(p3r_ctx* ctx, const p3r_prep* prep, const p3r_dmat* const* main_traces,
                    size_t n_instances, uint32_t flags, uint8_t* proof_buf, size_t proof_cap,
                    size_t* proof_len) {
  return guard(ctx, [&] {
    if (!prep || !main_traces || !proof_len || (!proof_buf && proof_cap)) fail(P3R_EINVAL, "bad arguments");
    for (size_t i = 0; i < n_instances; ++i)
      if (!main_traces[i]) fail(P3R_EINVAL, "main trace %zu is NULL", i);
    auto bytes = P3R_FIELD_CALL(ctx, prove_batch_any, ctx, prep, main_traces, n_instances,
                                (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0);
    emit_proof(std::move(bytes), proof_buf, proof_cap, proof_len);
  });
}

int p3r_prove_batch_host(p3r_ctx* ctx, const p3r_prep* prep, const p3r_matrix* main_traces,
                         size_t n_instances, uint32_t flags, uint8_t* proof_buf, size_t proof_cap,
                         size_t* proof_len) {
  return guard(ctx, [&] {
    if (!prep || !main_traces || !proof_len || (!proof_buf && proof_cap)) fail(P3R_EINVAL, "bad arguments");
    std::vector<std::unique_ptr<p3r_dmat>> owned;
    std::vector<const p3r_dmat*> ptrs;
    for (size_t i = 0; i < n_instances; ++i) {
      if (!main_traces[i].values) fail(P3R_EINVAL, "main trace %zu has NULL values", i);
      owned.push_back(P3R_FIELD_CALL(ctx, upload, ctx, main_traces[i].values, main_traces[i].height,
                                     main_traces[i].width));
      ptrs.push_back(owned.back().get());
    }
    auto bytes = P3R_FIELD_CALL(ctx, prove_batch_any, ctx, prep, ptrs.data(), n_instances,
                                (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0);
    emit_proof(std::move(bytes), proof_buf, proof_cap, proof_len);
  });
}

p3r_layer* p3r_layer_create(p3r_ctx* ctx, const p3r_layer_desc* desc, uint32_t* commit_out) {
  p3r_layer* out = nullptr;
  guard(ctx, [&] {
    if (!desc || !commit_out) fail(P3R_EINVAL, "NULL argument");
    out = P3R_FIELD_CALL(ctx, layer_create, ctx, desc, commit_out).release();
  });
  return out;
}
void p3r_layer_free(p3r_ctx* ctx, p3r_layer* layer) {
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  delete layer;
}
int p3r_layer_table_heights(const p3r_layer* L, size_t h[5]) {
  if (!L || !h) return P3R_EINVAL;
  h[0] = L->h_const; h[1] = L->h_public; h[2] = L->h_alu; h[3] = L->h_p2; h[4] = L->h_recompose;
  return P3R_OK;
}
int p3r_layer_recompose_coeff_height(const p3r_layer* L, size_t* h) {
  if (!L || !h) return P3R_EINVAL;
  *h = L->h_recompose_coeff;
  return P3R_OK;
}
int p3r_layer_recompose_kind(const p3r_layer* L, uint32_t* coeff_lookups) {
  if (!L || !coeff_lookups) return P3R_EINVAL;
  *coeff_lookups = L->recompose_coeff ? 1u : 0u;
  return P3R_OK;
}
int p3r_layer_effective_lanes(const p3r_layer* L, uint32_t* public_lanes, uint32_t* alu_lanes) {
  if (!L || !public_lanes || !alu_lanes) return P3R_EINVAL;
  *public_lanes = L->public_lanes;
  *alu_lanes = L->alu_lanes;
  return P3R_OK;
}
p3r_dtraces* p3r_traces_upload(p3r_ctx* ctx, const p3r_layer* layer, const p3r_traces* traces) {
  p3r_dtraces* out = nullptr;
  guard(ctx, [&] {
    if (!layer || !traces) fail(P3R_EINVAL, "NULL argument");
    out = P3R_FIELD_CALL(ctx, traces_upload, ctx, layer, traces).release();
  });
  return out;
}
void p3r_traces_free(p3r_ctx* ctx, p3r_dtraces* t) {
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  delete t;
}
int p3r_prove_all_tables_resident(p3r_ctx* ctx, const p3r_layer* layer, const p3r_dtraces* traces,
                                  uint32_t flags, uint8_t* proof_buf, size_t proof_cap, size_t* proof_len) {
  return guard(ctx, [&] {
    if (!layer || !traces || !proof_len || (!proof_buf && proof_cap)) fail(P3R_EINVAL, "bad arguments");
    auto bytes = P3R_FIELD_CALL(ctx, prove_all_tables, ctx, layer, traces,
                                (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0);
    emit_proof(std::move(bytes), proof_buf, proof_cap, proof_len);
  });
}
int p3r_prove_all_tables(p3r_ctx* ctx, const p3r_layer* layer, const p3r_traces* traces, uint32_t flags,
                         uint8_t* proof_buf, size_t proof_cap, size_t* proof_len) {
  return guard(ctx, [&] {
    if (!layer || !traces || !proof_len || (!proof_buf && proof_cap)) fail(P3R_EINVAL, "bad arguments");
    auto d = P3R_FIELD_CALL(ctx, traces_upload, ctx, layer, traces);
    auto bytes = P3R_FIELD_CALL(ctx, prove_all_tables, ctx, layer, d.get(),
                                (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0);
    emit_proof(std::move(bytes), proof_buf, proof_cap, proof_len);
  });
}
p3r_dmat* p3r_layer_build_main_trace(p3r_ctx* ctx, const p3r_layer* layer, const p3r_dtraces* traces,
                                     uint32_t table) {
  p3r_dmat* out = nullptr;
  guard(ctx, [&] {
    if (!layer || !traces || table > 5) fail(P3R_EINVAL, "bad arguments");
    if (layer->slot_of((int)table) < 0) fail(P3R_EINVAL, "table %u has no rows and is not part of the batch", table);
    auto m = P3R_FIELD_CALL(ctx, build_main_traces, ctx, layer, traces);
    P3R_HIP(hipStreamSynchronize(ctx->stream));
    out = m[table].release();
  });
  return out;
}

// ---- native verifier (verify_impl.h): host code, no device needed ----
int p3r_verify_batch(const p3r_config* cfg, const p3r_air_desc* airs, size_t n_airs,
                     const uint32_t* preprocessed_commitment, const uint32_t* degree_bits, const uint8_t* proof,
                     size_t proof_len, uint32_t flags, char* err_buf, size_t err_cap) {
  auto report = [&](const char* msg) {
    if (err_buf && err_cap) snprintf(err_buf, err_cap, "%s", msg);
  };
  try {
    if (!cfg || !airs || !preprocessed_commitment || !degree_bits || (!proof && proof_len)) { report("NULL argument"); return P3R_EINVAL; }
    if (cfg->abi_version != P3R_ABI_VERSION) { report("ABI version mismatch"); return P3R_EINVAL; }
    if (cfg->challenge_degree != 0 && cfg->challenge_degree != 4 && cfg->challenge_degree != 5) { report("UnsupportedChallengeDegree"); return P3R_EUNSUPPORTED; }
    const bool generic_d = p3r::ext_degree_is_binomial_generic(cfg->ext_degree);
    if (generic_d && cfg->ext_w == 0) { report("MissingWForExtension"); return P3R_EINVAL; }
    if (!generic_d && cfg->ext_degree != 1 && cfg->ext_degree != 4 && !(cfg->ext_degree == 5 && cfg->field == P3R_FIELD_KOALA_BEAR)) { report("UnsupportedDegree"); return P3R_EUNSUPPORTED; }
    p3r::VerifyParams prm{(int)cfg->log_blowup, (int)cfg->max_log_arity, (int)cfg->cap_height, (int)cfg->log_final_poly_len,
                          (int)cfg->commit_pow_bits, (int)cfg->query_pow_bits, (int)cfg->num_queries, {}};
    if (cfg->fri_log_arities) prm.fri_log_arities.assign(cfg->fri_log_arities, cfg->fri_log_arities + cfg->fri_log_arities_len);
    if (!prm.layout.set(cfg->proof_layout, cfg->proof_layout_len)) { report("proof_layout must be 18 bytes: three permutations"); return P3R_EINVAL; }
    std::vector<p3r::AirParams> a(n_airs);
    for (size_t i = 0; i < n_airs; ++i) {
      if (airs[i].kind > P3R_AIR_RECOMPOSE || !airs[i].lanes) { report("bad AIR descriptor"); return P3R_EINVAL; }
      a[i] = {(int)airs[i].kind, (int)airs[i].lanes, (int)airs[i].horner_packed_steps, (int)airs[i].coeff_lookups,
              (cfg->ext_choices & P3R_EXT_LOOKUP_UNPACKED) ? 1 : 0, (int)cfg->ext_degree, 0u};
    }
    const bool canonical = (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0;
    std::vector<uint32_t> cap(preprocessed_commitment, preprocessed_commitment + ((size_t)P2_DIGEST << cfg->cap_height));
    const std::vector<uint32_t> want_db(degree_bits, degree_bits + n_airs);
    auto run = [&](auto tag) {
      using PP = decltype(tag);
      const size_t nrc = p2_num_constants<PP>();
      const uint32_t* src = cfg->poseidon2_rc;
      if (src && cfg->poseidon2_rc_len != nrc) p3r::vfail("poseidon2_rc_len is %u, the field needs %zu constants", cfg->poseidon2_rc_len, nrc);
      if (!src) src = PP::FIELD_ID == 0 ? kDefaultRc_koala_bear : kDefaultRc_baby_bear;
      auto airs_pp = a;
      for (auto& x : airs_pp) x.ext_w_mont = generic_d ? p3r::Fp<PP>::from_canonical(cfg->ext_w).v : 0u;
      if (cfg->challenge_degree == 5) {
        if constexpr (p3r::kHasQuintic<PP>)
          p3r::verify_batch<PP, 5>(prm, std::vector<uint32_t>(src, src + nrc), airs_pp, cap, want_db, proof, proof_len, canonical);
        else
          p3r::vfail("UnsupportedChallengeDegree: the quintic challenge field is KoalaBear's");
      } else {
        p3r::verify_batch<PP>(prm, std::vector<uint32_t>(src, src + nrc), airs_pp, cap, want_db, proof, proof_len, canonical);
      }
    };
    if (cfg->field == P3R_FIELD_KOALA_BEAR) run(p3r::KoalaBearParams{});
    else if (cfg->field == P3R_FIELD_BABY_BEAR) run(p3r::BabyBearParams{});
    else { report("unknown field"); return P3R_EINVAL; }
    return P3R_OK;
  } catch (const p3r::VerifyFailure& e) {
    report(e.what());
    return P3R_EINVAL;
  } catch (const std::exception& e) {
    report(e.what());
    return P3R_EINVAL;
  }
}

int p3r_batch_proof_len(uint32_t field, const uint8_t* bytes, size_t len, uint32_t flags, size_t* proof_len,
                        char* err_buf, size_t err_cap) {
  return p3r_batch_proof_len_layout(field, bytes, len, flags, nullptr, proof_len, err_buf, err_cap);
}
int p3r_batch_proof_len_layout(uint32_t field, const uint8_t* bytes, size_t len, uint32_t flags,
                               const uint8_t* proof_layout, size_t* proof_len, char* err_buf, size_t err_cap) {
  try {
    if (!bytes || !proof_len) throw std::runtime_error("NULL argument");
    const bool canonical = (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0;
    p3r::ProofLayout PL;
    if (!PL.set(proof_layout, 18)) throw std::runtime_error("proof_layout must be three permutations batch[5] | fri[5] | opened[8]");
    if (field == P3R_FIELD_KOALA_BEAR && (flags & P3R_PROOF_QUINTIC_CHALLENGE))
      (void)p3r::parse_proof<p3r::KoalaBearParams, 5>(bytes, len, canonical, proof_len, PL);
    else if (field == P3R_FIELD_KOALA_BEAR) (void)p3r::parse_proof<p3r::KoalaBearParams>(bytes, len, canonical, proof_len, PL);
    else if (field == P3R_FIELD_BABY_BEAR) (void)p3r::parse_proof<p3r::BabyBearParams>(bytes, len, canonical, proof_len, PL);
    else throw std::runtime_error("unknown field");
    return P3R_OK;
  } catch (const std::exception& e) {
    if (err_buf && err_cap) snprintf(err_buf, err_cap, "%s", e.what());
    return P3R_EINVAL;
  }
}

int p3r_batch_stark_proof_parse(uint32_t field, const uint8_t* bytes, size_t len, uint32_t flags,
                                const uint8_t* proof_layout, p3r_batch_stark_meta* out, char* err_buf,
                                size_t err_cap) {
  try {
    if (!bytes || !out) throw std::runtime_error("NULL argument");
    const auto t0 = std::chrono::steady_clock::now();
    const bool canonical = (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0;
    p3r::ProofLayout PL;
    if (!PL.set(proof_layout, 18)) throw std::runtime_error("proof_layout must be three permutations batch[5] | fri[5] | opened[8]");
    const int dc = (flags & P3R_PROOF_QUINTIC_CHALLENGE) ? 5 : 4;
    if (field == P3R_FIELD_KOALA_BEAR) p3r::parse_batch_stark_meta<p3r::KoalaBearParams>(bytes, len, canonical, PL, dc, out);
    else if (field == P3R_FIELD_BABY_BEAR) p3r::parse_batch_stark_meta<p3r::BabyBearParams>(bytes, len, canonical, PL, dc, out);
    else throw std::runtime_error("unknown field");
    out->parse_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return P3R_OK;
  } catch (const std::exception& e) {
    if (err_buf && err_cap) snprintf(err_buf, err_cap, "%s", e.what());
    return P3R_EINVAL;
  }
}

// ---- circuit boundary (circuit_impl.cuh) ----
p3r_circuit* p3r_circuit_create(p3r_ctx* ctx, const p3r_circuit_desc* desc, uint32_t* commit_out) {
  p3r_circuit* out = nullptr;
  guard(ctx, [&] {
    if (!desc || !commit_out) fail(P3R_EINVAL, "NULL argument");
    if (p3r::ext_degree_is_binomial_generic(ctx->cfg.ext_degree))
      fail(P3R_EUNSUPPORTED, "UnsupportedDegree(%u): the device runner computes in degree 1, 4 and 5; hand the layer over as Traces",
           ctx->cfg.ext_degree);
    out = P3R_FIELD_CALL(ctx, circuit_create, ctx, desc, commit_out).release();
  });
  return out;
}
void p3r_circuit_free(p3r_ctx* ctx, p3r_circuit* circuit) {
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  delete circuit;
}
const p3r_layer* p3r_circuit_layer(const p3r_circuit* circuit) { return circuit ? circuit->layer.get() : nullptr; }
int p3r_circuit_counts(const p3r_circuit* circuit, p3r_layer_desc_counts* out) {
  if (!circuit || !out) return P3R_EINVAL;
  *out = circuit->counts;
  return P3R_OK;
}
int p3r_circuit_levels(const p3r_circuit* circuit, size_t* n_levels) {
  if (!circuit || !n_levels) return P3R_EINVAL;
  *n_levels = circuit->sched.levels;
  return P3R_OK;
}
int p3r_circuit_prepared_on_device(const p3r_circuit* circuit) { return circuit && circuit->prepared_on_device ? 1 : 0; }
p3r_dinputs* p3r_circuit_inputs_upload(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_circuit_inputs* inputs) {
  p3r_dinputs* out = nullptr;
  guard(ctx, [&] {
    if (!circuit || !inputs) fail(P3R_EINVAL, "NULL argument");
    out = P3R_FIELD_CALL(ctx, circuit_inputs_upload, ctx, circuit, inputs).release();
  });
  return out;
}
void p3r_circuit_inputs_free(p3r_ctx* ctx, p3r_dinputs* inputs) {
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  delete inputs;
}
p3r_dtraces* p3r_circuit_run_resident(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_dinputs* inputs) {
  p3r_dtraces* out = nullptr;
  guard(ctx, [&] {
    if (!circuit || !inputs) fail(P3R_EINVAL, "NULL argument");
    out = P3R_FIELD_CALL(ctx, circuit_run, ctx, circuit, inputs).release();
  });
  return out;
}
p3r_dtraces* p3r_circuit_run(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_circuit_inputs* inputs) {
  p3r_dtraces* out = nullptr;
  guard(ctx, [&] {
    if (!circuit || !inputs) fail(P3R_EINVAL, "NULL argument");
    auto d = P3R_FIELD_CALL(ctx, circuit_inputs_upload, ctx, circuit, inputs);
    out = P3R_FIELD_CALL(ctx, circuit_run, ctx, circuit, d.get()).release();
  });
  return out;
}
static void prove_next_layer_impl(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_dinputs* d, uint32_t flags,
                                  uint8_t* proof_buf, size_t proof_cap, size_t* proof_len) {
  // The run is enqueued, not awaited: proving starts behind it on the stream, and the run's error
  // word lands in a pinned host word that is read once the proof is done.  (A failed run leaves
  // garbage VALUES in the traces, never a bad address, so proving over them is harmless; its error
  // takes precedence over whatever the prover made of the garbage.)
  uint32_t* run_err = nullptr;
  P3R_HIP(ctx->stage.words(&run_err));
  *run_err = 0xFFFFFFFFu;
  auto t = P3R_FIELD_CALL(ctx, circuit_run, ctx, circuit, d, run_err);
  std::vector<uint8_t> bytes;
  try {
    bytes = P3R_FIELD_CALL(ctx, prove_all_tables, ctx, circuit->layer.get(), t.get(),
                           (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0);
  } catch (...) {
    if (hipStreamSynchronize(ctx->stream) == hipSuccess) run_raise_error(*run_err);
    throw;
  }
  run_raise_error(*run_err);  // prove_all_tables returns with the stream drained
  emit_proof(std::move(bytes), proof_buf, proof_cap, proof_len);
}
int p3r_prove_next_layer(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_circuit_inputs* inputs,
                         uint32_t flags, uint8_t* proof_buf, size_t proof_cap, size_t* proof_len) {
  return guard(ctx, [&] {
    if (!circuit || !inputs || !proof_len || (!proof_buf && proof_cap)) fail(P3R_EINVAL, "bad arguments");
    auto d = P3R_FIELD_CALL(ctx, circuit_inputs_upload, ctx, circuit, inputs);
    prove_next_layer_impl(ctx, circuit, d.get(), flags, proof_buf, proof_cap, proof_len);
  });
}
int p3r_prove_next_layer_resident(p3r_ctx* ctx, const p3r_circuit* circuit, const p3r_dinputs* inputs,
                                  uint32_t flags, uint8_t* proof_buf, size_t proof_cap, size_t* proof_len) {
  return guard(ctx, [&] {
    if (!circuit || !inputs || !proof_len || (!proof_buf && proof_cap)) fail(P3R_EINVAL, "bad arguments");
    prove_next_layer_impl(ctx, circuit, inputs, flags, proof_buf, proof_cap, proof_len);
  });
}
int p3r_dtraces_get(p3r_ctx* ctx, const p3r_layer* layer, const p3r_dtraces* traces, uint32_t which,
                    uint32_t* out, size_t out_len) {
  return guard(ctx, [&] {
    if (!layer || !traces || (!out && out_len)) fail(P3R_EINVAL, "NULL argument");
    P3R_FIELD_CALL(ctx, dtraces_get, ctx, layer, traces, which, out, out_len);
  });
}

int p3r_profile_enable(p3r_ctx* ctx, int on) {
  return guard(ctx, [&] {
    P3R_HIP(hipStreamSynchronize(ctx->stream));
    prof_clear(ctx);
    ctx->prof_enabled = on != 0;
  });
}

int p3r_profile_read(p3r_ctx* ctx, p3r_profile_entry* out, size_t cap, size_t* n_out) {
  return guard(ctx, [&] {
    if (!out || !n_out) fail(P3R_EINVAL, "NULL argument");
    P3R_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<p3r_profile_entry> acc;
    for (auto& r : ctx->prof) {
      float ms = 0;
      P3R_HIP(hipEventElapsedTime(&ms, r.a, r.b));
      auto it = std::find_if(acc.begin(), acc.end(),
                             [&](const p3r_profile_entry& e) { return !strcmp(e.name, r.name); });
      if (it == acc.end()) {
        p3r_profile_entry e{};
        strncpy(e.name, r.name, sizeof e.name - 1);
        acc.push_back(e);
        it = acc.end() - 1;
      }
      it->total_ms += ms;
      it->launches += 1;
    }
    for (auto& kv : ctx->stage_ms) {
      p3r_profile_entry e{};
      snprintf(e.name, sizeof e.name, "stage:%s", kv.first.c_str());
      e.total_ms = kv.second;
      acc.push_back(e);
    }
    if (acc.size() > cap) fail(P3R_EBUFFER, "need room for %zu profile entries", acc.size());
    std::copy(acc.begin(), acc.end(), out);
    *n_out = acc.size();
  });
}

}  // extern "C"
