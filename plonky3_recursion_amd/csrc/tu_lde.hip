// K5: NTT tables, the passes of kernels_ntt.hip.h / kernels_ntt2.hip.h and the coset LDE of a batch of matrices
// (TwoAdicSubgroupDft::coset_lde_batch as TwoAdicFriPcs::commit uses it, circuit-prover/src/config.rs:55,131).
// Own translation unit (tu_api.h).
#include "tu_api.h"
#include "kernels_ntt2.hip.h"
#include "profile.h"

#include <algorithm>
#include <map>

namespace p3r {
namespace {

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

// The lean forward passes (kernels_ntt2.hip.h): jobs grouped by the compile-time sub-transform size.
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

}  // namespace

// K5 for a batch of matrices (all tables of a commit): every matrix goes through the same passes,
// and pass k of all of them is one launch.
// in: h x w evaluations over the subgroup (natural order, column-major Montgomery).
// Returns (h << added_bits) x w, rows in bit-reversed order over shift * <w_{h<<added_bits}>.
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
      // lean kernels (kernels_ntt2.hip.h): the same two passes with compile-time geometry
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
void lde_init(p3r_ctx*) {
  P3R_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_tile<PP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

template std::vector<std::unique_ptr<p3r_dmat>> coset_lde_batch<KoalaBearParams>(p3r_ctx*, const std::vector<LdeItem>&, int);
template std::vector<std::unique_ptr<p3r_dmat>> coset_lde_batch<BabyBearParams>(p3r_ctx*, const std::vector<LdeItem>&, int);
template void lde_init<KoalaBearParams>(p3r_ctx*);
template void lde_init<BabyBearParams>(p3r_ctx*);

}  // namespace p3r
