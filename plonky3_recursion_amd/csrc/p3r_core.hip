// C ABI (include/p3r.h): context, device matrices, Poseidon2 (K3), coset LDE (K5), MMCS (K6).
// Host orchestration only; all arithmetic on data runs in the gfx950 kernels of kernels.hip.h.
#include "context.h"
#include "kernels.hip.h"
#include "kernels_stark.hip.h"
#include "kernels_coop.hip.h"
#include "tu_api.h"
#include "profile.h"
#include "run_schedule.h"
#include "prep_device.h"
#include "host_abi.h"

#include <sys/random.h>

#include <algorithm>
#include <cerrno>
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

// the width-32 permutation is about to be used: its constants must be the caller's, or acknowledged as unpinned (p3r.h)
inline void require_w32_constants(const p3r_ctx* ctx) {
  if (ctx->w32_unacknowledged)
    fail(P3R_EINVAL, "the width-32 permutation with poseidon2_w32_rc / poseidon2_w32_diag NULL: the built-in constants are self-generated, "
                     "not upstream's - pass the caller's, or acknowledge with P3R_EXT_UNPINNED_W32_DEFAULTS");
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

// width-32 permutation over n row-major states (canonical in / out): the unit seam of p3r_poseidon2_w32_permute_batch
template <class PP>
__global__ void __launch_bounds__(kBlock) k_p2w_permute_rows(uint32_t* __restrict__ s, size_t n, const uint32_t* __restrict__ rcw) {
  using F = Fp<PP>;
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  F x[P2W_WIDTH];
#pragma unroll
  for (int k = 0; k < P2W_WIDTH; ++k) x[k] = F::from_canonical(s[i * P2W_WIDTH + k]);
  P2NullSink sink;
  p2w_permute_traced<PP>(x, rcw, sink);
#pragma unroll
  for (int k = 0; k < P2W_WIDTH; ++k) s[i * P2W_WIDTH + k] = x[k].to_canonical();
}
template <class PP>
void permute_w32_rows(p3r_ctx* ctx, uint32_t* d, size_t n) {
  hipLaunchKernelGGL(k_p2w_permute_rows<PP>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, d, n, ctx->rc.p + p2_num_constants<PP>());
  P3R_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ coset LDE (tu_lde.hip)
template <class PP>
std::unique_ptr<p3r_dmat> coset_lde(p3r_ctx* ctx, const p3r_dmat* in, int added_bits, uint32_t shift) {
  return std::move(coset_lde_batch<PP>(ctx, {{in, shift}}, added_bits)[0]);
}

// The second stream of the two-stream commit experiment (knobs build only: P3R_COMMIT_OVERLAP, prove_impl.hip.h): created
// on first use, so that a product context owns one stream and no idle events.
inline void ensure_side_streams(p3r_ctx* ctx) {
  if (ctx->stream2) return;
  P3R_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  int lo = 0, hi = 0;
  P3R_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = the numerically greatest = lowest priority
  P3R_HIP(hipStreamCreateWithPriority(&ctx->stream2_low, hipStreamNonBlocking, lo));
  P3R_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  P3R_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
}

// ------------------------------------------------------------------ MMCS
// Row digests of several height classes in one launch: classes[c] = the matrices of one height
// (their rows are concatenated in the given order), digs[c] = [8][h_c].
// `side`: launch on the ctx's second stream, after everything the main stream has queued so far (the job tables go up
// through the main stream); the caller joins with hash_rows_join before anything reads the digests.
template <class PP>
void hash_rows(p3r_ctx* ctx, const std::vector<std::vector<const p3r_dmat*>>& classes,
               const std::vector<uint32_t*>& digs, int side = 0) {
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
  double perms = 0;
  for (auto& j : jobs) {
    j.block0 = blocks;
    blocks += blocks_for(j.h);
    perms += (double)j.h * ((j.wtot + P2_RATE - 1) / P2_RATE);
  }
  prof_count(ctx, "hash_rows_perms", perms);
  const auto* d_jobs =
      static_cast<const HashRowsJob*>(const_table(ctx, jobs.data(), jobs.size() * sizeof(HashRowsJob)));
  if (side) {
    ensure_side_streams(ctx);
    hipStream_t s2 = side == 2 ? ctx->stream2_low : ctx->stream2;
    P3R_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    P3R_HIP(hipStreamWaitEvent(s2, ctx->ev_fork, 0));
    hipLaunchKernelGGL(k_mmcs_hash_rows<PP>, dim3(blocks), dim3(kBlock), 0, s2, d_jobs, (int)jobs.size(), ctx->rcd());
    P3R_HIP(hipGetLastError());
    P3R_HIP(hipEventRecord(ctx->ev_join, s2));
    return;
  }
  ProfScope ps(ctx, "mmcs_hash_rows");
  hipLaunchKernelGGL(k_mmcs_hash_rows<PP>, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_jobs, (int)jobs.size(),
                     ctx->rcd());
  P3R_HIP(hipGetLastError());
}
inline void hash_rows_join(p3r_ctx* ctx) { P3R_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); }

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

// The levels of an arity-4 tree above its leaf-digest layer (tree->layers[0], tree->levels set): one launch each.
// `inject`: height -> row digests of the matrices of that height (null: none, the FRI commit-phase trees).
template <class PP>
void mmcs4_build_levels(p3r_ctx* ctx, p3r_tree* tree, const std::map<size_t, DevBuf>* inject) {
  for (const Mmcs4Level& lv : tree->levels) {
    DevBuf next(P2_DIGEST * lv.padded_next);
    const uint32_t* inj = nullptr;
    if (lv.inject_h) {
      auto it = inject ? inject->find(lv.inject_h) : decltype(inject->end()){};
      if (!inject || it == inject->end() || lv.inject_h != lv.logical_next)
        fail(P3R_EINVAL, "arity-4 MMCS: matrix heights must be powers of two");
      inj = it->second.p;
    }
    mmcs4_compress<PP>(ctx, tree->layers.back().p, tree->layer_n.back(), lv.step, inj, next.p, lv.logical_next, lv.padded_next);
    tree->layers.push_back(std::move(next));
    tree->layer_n.push_back(lv.padded_next);
  }
}

// The arity-4 tree (mmcs4.h; kernels_mmcs4.hip.h): leaf digests of every height class in one launch, then one launch
// per level.  tree->layers[l] is [8][tree->layer_n[l]].
template <class PP>
void mmcs_commit4(p3r_ctx* ctx, p3r_tree* tree, uint32_t* cap_out) {
  using F = Fp<PP>;
  const auto& mats = tree->mats;
  std::vector<size_t> heights, class_h;
  for (auto* m : mats) heights.push_back(m->h);
  std::vector<size_t> order(mats.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return mats[a]->h > mats[b]->h; });
  const size_t hmax = mats[order[0]]->h;
  tree->arity = 4;
  tree->log_max_h = log2_exact(hmax, "matrix height");
  tree->cap_height = 0;
  tree->total_width = 0;
  for (auto* m : mats) tree->total_width += m->w;
  tree->levels = mmcs4_schedule(heights);
  for (size_t i : order)
    if (class_h.empty() || class_h.back() != mats[i]->h) class_h.push_back(mats[i]->h);
  tree->layers.clear();
  tree->layer_n.clear();
  const size_t n0 = mmcs4_padded_len(hmax);
  tree->layers.emplace_back(P2_DIGEST * n0);
  tree->layer_n.push_back(n0);
  if (n0 != hmax) P3R_HIP(fill_async(ctx->stream, tree->layers[0].p, 0, P2_DIGEST * n0 * 4));
  std::map<size_t, DevBuf> inject;
  {
    std::vector<std::vector<const p3r_dmat*>> classes;
    std::vector<uint32_t*> digs;
    std::vector<size_t> allocs;
    for (size_t h : class_h) {
      std::vector<const p3r_dmat*> v;
      for (size_t i : order)
        if (mats[i]->h == h) v.push_back(mats[i]);
      classes.push_back(std::move(v));
      if (h == hmax) { digs.push_back(tree->layers[0].p); allocs.push_back(n0); }
      else { digs.push_back(inject.emplace(h, DevBuf(P2_DIGEST * h)).first->second.p); allocs.push_back(h); }
    }
    mmcs4_hash_rows<PP>(ctx, classes, digs, allocs);
  }
  mmcs4_build_levels<PP>(ctx, tree, &inject);
  uint32_t root[P2_DIGEST];
  P3R_HIP(fetch_small(ctx, tree->layers.back().p, P2_DIGEST, root));
  for (int k = 0; k < P2_DIGEST; ++k) cap_out[k] = F::raw(root[k]).to_canonical();
}

// `pre`: leaf digests of some height classes computed already (height -> [8][h] digests; hash_rows on the side stream,
// joined by the caller): those classes are not hashed again.
template <class PP>
void mmcs_commit(p3r_ctx* ctx, p3r_tree* tree, uint32_t* cap_out, std::map<size_t, DevBuf>* pre = nullptr) {
  using F = Fp<PP>;
  const auto& mats = tree->mats;
  if (mats.empty()) fail(P3R_EINVAL, "MMCS commit needs at least one matrix");
  if (ctx->cfg.mmcs_arity == 4) {
    if (pre && !pre->empty()) fail(P3R_EHIP, "internal: pre-hashed classes under the arity-4 MMCS");
    return mmcs_commit4<PP>(ctx, tree, cap_out);
  }
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
  std::map<size_t, DevBuf> inject;  // height -> digests of the matrices of that height
  {
    std::vector<std::vector<const p3r_dmat*>> classes;
    std::vector<uint32_t*> digs;
    for (size_t h : class_h) {
      auto done = pre ? pre->find(h) : decltype(pre->end()){};
      const bool have = pre && done != pre->end();
      if (h == hmax) {
        if (have) tree->layers.push_back(std::move(done->second));
        else tree->layers.emplace_back(P2_DIGEST * hmax);
      } else if (have) {
        inject.emplace(h, std::move(done->second));
      }
      if (have) continue;
      classes.push_back(at_height(h));
      if (h == hmax) digs.push_back(tree->layers[0].p);
      else digs.push_back(inject.emplace(h, DevBuf(P2_DIGEST * h)).first->second.p);
    }
    if (!classes.empty()) hash_rows<PP>(ctx, classes, digs);
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
  size_t depth = (size_t)(tree->log_max_h - tree->cap_height);
  if (tree->arity == 4) {
    // step - 1 siblings per level, ascending position, the node's own left out (recursion/src/pcs/mmcs.rs:1413-1461)
    depth = 0;
    for (size_t l = 0; l < tree->levels.size(); ++l) {
      const size_t step = tree->levels[l].step, idx = index >> tree->levels[l].bits, pos = idx & (step - 1);
      for (size_t j = 0; j < step; ++j) {
        if (j == pos) continue;
        P3R_HIP(hipMemcpy2DAsync(proof + depth * P2_DIGEST, 4, tree->layers[l].p + (idx - pos + j), tree->layer_n[l] * 4, 4,
                                 P2_DIGEST, hipMemcpyDeviceToHost, ctx->stream));
        ++depth;
      }
    }
  } else {
    for (size_t l = 0; l < depth; ++l) {
      size_t n = size_t(1) << (tree->log_max_h - l);
      size_t sib = (index >> l) ^ 1;
      P3R_HIP(hipMemcpy2DAsync(proof + l * P2_DIGEST, 4, tree->layers[l].p + sib, n * 4, 4,
                               P2_DIGEST, hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  P3R_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < off; ++i) opened[i] = F::raw(opened[i]).to_canonical();
  for (size_t i = 0; i < depth * P2_DIGEST; ++i) proof[i] = F::raw(proof[i]).to_canonical();
}

template <class PP>
void init_ctx(p3r_ctx* ctx) {
  using F = Fp<PP>;
  const size_t nrc = p2_num_constants<PP>();
  const std::vector<uint32_t> table = constants_table<PP>(ctx->cfg, [](const char* what, uint32_t got, size_t want) {
    fail(P3R_EINVAL, "%s is %u, the field needs %zu constants", what, got, want);
  });
  const uint32_t* src = table.data();
  ctx->rc_canonical = table;   // width-16 constants first (nrc of them), then the width-32 table
  std::vector<uint32_t> mont(table.size());
  for (size_t i = 0; i < table.size(); ++i) {
    if (table[i] >= PP::P) fail(P3R_EINVAL, "permutation constant %zu is not canonical", i);
    mont[i] = F::from_canonical(table[i]).v;
  }
  ctx->rc_mont_host = mont;
  ctx->rc.alloc(mont.size());
  P3R_HIP(copy_sync(ctx->stream, ctx->rc.p, mont.data(), mont.size() * 4, hipMemcpyHostToDevice));
  // FP64 tables: the width-16 round constants, then the width-32 table of poseidon2_w32_f64.hip.h (round constants |
  // the diagonal as centred integers | diagonal / P)
  std::vector<double> rcd(src, src + nrc);
  ctx->rcd_w32_at = nrc;
  {
    const size_t nrcw = p2w_num_rc<PP>();
    const uint32_t* w = src + nrc;
    rcd.insert(rcd.end(), w, w + nrcw);
    for (int i = 0; i < P2W_WIDTH; ++i) {
      const uint32_t d = w[nrcw + i];
      rcd.push_back(d > PP::P / 2 ? (double)d - (double)PP::P : (double)d);
    }
    // the kernels have the lane forms of the BUILT-IN diagonal compiled in: those instances are launched when the
    // configured diagonal is the built-in one, entry by entry (poseidon2_w32_f64.hip.h); any other diagonal is data
    const uint32_t* builtin = PP::FIELD_ID == 0 ? kDefaultDiagW32_koala_bear : kDefaultDiagW32_baby_bear;
    const bool is_builtin = std::equal(builtin, builtin + P2W_WIDTH, w + nrcw) && !tuning_knob("P3R_W32_GENERAL_DIAG");
    ctx->w32_diag_builtin = is_builtin;
  }
  ctx->rc_f64.alloc(2 * rcd.size());
  P3R_HIP(copy_sync(ctx->stream, ctx->rc_f64.p, rcd.data(), rcd.size() * 8, hipMemcpyHostToDevice));
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
  ctx->cfg.poseidon2_rc = nullptr;  // caller's pointers are not retained
  ctx->w32_unacknowledged = (!ctx->cfg.poseidon2_w32_rc || !ctx->cfg.poseidon2_w32_diag) && !(ctx->cfg.ext_choices & P3R_EXT_UNPINNED_W32_DEFAULTS);
  if (ctx->w32_unacknowledged && ctx->cfg.mmcs_arity == 4)
    fail(P3R_EINVAL, "mmcs_arity = 4 with poseidon2_w32_rc / poseidon2_w32_diag NULL: the built-in width-32 constants are self-generated, "
                     "not upstream's - pass the caller's, or acknowledge with P3R_EXT_UNPINNED_W32_DEFAULTS");
  ctx->cfg.poseidon2_w32_rc = nullptr;
  ctx->cfg.poseidon2_w32_diag = nullptr;
  if (ctx->cfg.fri_log_arities) ctx->fri_log_arities.assign(ctx->cfg.fri_log_arities, ctx->cfg.fri_log_arities + ctx->cfg.fri_log_arities_len);
  ctx->cfg.fri_log_arities = nullptr;
  if (!ctx->proof_layout.set(ctx->cfg.proof_layout, ctx->cfg.proof_layout_len))
    fail(P3R_EINVAL, "proof_layout must be 18 bytes: three permutations batch[5] | fri[5] | opened[8]");
  ctx->cfg.proof_layout = nullptr;
  lde_init<PP>(ctx);
}

}  // namespace

#include "prove_impl.hip.h"
#include "layer_impl.hip.h"
#include "circuit_impl.hip.h"

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
    if (cfg->mmcs_arity != 0 && cfg->mmcs_arity != 2 && cfg->mmcs_arity != 4)
      fail(P3R_EINVAL, "mmcs_arity must be 2 or 4 (got %u)", cfg->mmcs_arity);
    // the cap of an arity-4 tree strips whole compression steps (recursion/src/pcs/mmcs.rs:1143-1156); the reference's
    // arity-4 configurations run with the default cap_height 0 (recursive_aggregation.rs:77), and only that is built
    if (cfg->mmcs_arity == 4 && cfg->cap_height != 0) fail(P3R_EUNSUPPORTED, "arity-4 MMCS: cap_height must be 0");
    if (cfg->zk > 1) fail(P3R_EINVAL, "zk must be 0 or 1 (got %u)", cfg->zk);
    if (cfg->zk && cfg->num_random_codewords > 8) fail(P3R_EINVAL, "num_random_codewords must be in 1..8 (0 selects 2)");
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
    // (the side streams of the two-stream commit experiment are created on first use: ensure_side_streams; the knobs build
    // can create them with the context, as round 5 did - the A/B of profiles/r06/README.md on small-layer throughput)
    if (tuning_knob("P3R_FORCE_SIDE_STREAMS")) ensure_side_streams(c.get());
    {
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) c->n_cus = cus;
    }
    if (cfg->mmcs_salt_elems > 16) fail(P3R_EINVAL, "mmcs_salt_elems must be in 0..16 (got %u)", cfg->mmcs_salt_elems);
    if (cfg->zk || cfg->mmcs_salt_elems) {
      // the key the context draws its masks (and the salts of a hiding MMCS) with (zk_rand.h): the caller's, mixed with 128 bits of operating-system
      // entropy unless reproducible proofs were asked for
      uint32_t entropy[4] = {0, 0, 0, 0};
      const bool deterministic = cfg->ext_choices & P3R_EXT_ZK_DETERMINISTIC;
      if (!deterministic) {
        size_t got = 0;
        while (got < sizeof entropy) {
          const ssize_t r = getrandom(reinterpret_cast<uint8_t*>(entropy) + got, sizeof entropy - got, 0);
          if (r < 0) { if (errno == EINTR) continue; fail(P3R_EHIP, "getrandom failed (errno %d): no entropy for the ZK key", errno); }
          got += (size_t)r;
        }
      }
      zk_context_key(cfg->zk_key, entropy, deterministic, c->zk_key);
    }
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
  if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
  if (ctx->stream2_low) { (void)hipStreamSynchronize(ctx->stream2_low); (void)hipStreamDestroy(ctx->stream2_low); }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
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
  std::copy(ctx->rc_canonical.begin(), ctx->rc_canonical.begin() + p3r_poseidon2_num_constants(ctx), out);   // the width-16 table
  return P3R_OK;
}

uint64_t p3r_zk_nonce(const p3r_ctx* ctx) { return ctx ? ctx->zk_nonce : 0; }
int p3r_zk_set_nonce(p3r_ctx* ctx, uint64_t nonce) {
  if (!ctx) return P3R_EINVAL;
  if (!(ctx->cfg.ext_choices & P3R_EXT_ZK_DETERMINISTIC)) {
    ctx->err = "p3r_zk_set_nonce needs P3R_EXT_ZK_DETERMINISTIC: replaying a nonce repeats the masks of an earlier proof";
    return P3R_EINVAL;
  }
  ctx->zk_nonce = nonce;
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

// unit seams of the width-32 permutation (ABI 6): the permutation itself, and the table's trace fill
int p3r_poseidon2_w32_permute_batch(p3r_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n) {
  return guard(ctx, [&] {
    require_w32_constants(ctx);
    if (!in || !out) fail(P3R_EINVAL, "NULL argument");
    if (n == 0) return;
    DevBuf d(n * P2W_WIDTH);
    P3R_HIP(hipMemcpyAsync(d.p, in, n * P2W_WIDTH * 4, hipMemcpyHostToDevice, ctx->stream));
    P3R_FIELD_CALL(ctx, permute_w32_rows, ctx, d.p, n);
    P3R_HIP(copy_sync(ctx->stream, out, d.p, n * P2W_WIDTH * 4, hipMemcpyDeviceToHost));
  });
}
int p3r_poseidon2_w32_trace_fill(p3r_ctx* ctx, const p3r_p2w_rows* rows, uint32_t* trace_out) {
  return guard(ctx, [&] {
    require_w32_constants(ctx);
    if (!rows || !trace_out || !rows->input_values || !rows->new_start || !rows->merkle_path || !rows->mmcs_bit || !rows->mmcs_bit2 ||
        !rows->mmcs_index_sum)
      fail(P3R_EINVAL, "NULL argument");
    const size_t n = rows->n;
    log2_exact(n, "width-32 Poseidon2 row count (callers pad to a power of two)");
    std::vector<uint8_t> f(4 * n);
    std::copy(rows->new_start, rows->new_start + n, f.begin());
    std::copy(rows->merkle_path, rows->merkle_path + n, f.begin() + n);
    std::copy(rows->mmcs_bit, rows->mmcs_bit + n, f.begin() + 2 * n);
    std::copy(rows->mmcs_bit2, rows->mmcs_bit2 + n, f.begin() + 3 * n);
    auto run = [&](auto tag) {
      using PP = decltype(tag);
      auto d = p2w_rows_upload<PP>(ctx, n, n, rows->input_values, f.data(), rows->mmcs_index_sum);
      auto t = trace_fill_w32<PP>(ctx, d.get());
      download<PP>(ctx, t.get(), trace_out);
    };
    if (ctx->cfg.field == P3R_FIELD_KOALA_BEAR) run(KoalaBearParams{}); else run(BabyBearParams{});
  });
}
uint32_t p3r_poseidon2_w32_trace_width(const p3r_ctx* ctx) {
  return (ctx->cfg.field == P3R_FIELD_KOALA_BEAR ? p2w_perm_cols<KoalaBearParams>() : p2w_perm_cols<BabyBearParams>()) + 4;
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
size_t p3r_tree_proof_len(const p3r_tree* t) {
  return t->arity == 4 ? p3r::mmcs4_proof_len(t->levels) : (size_t)(t->log_max_h - t->cap_height);
}
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

// A proof that does not fit the caller's buffer is KEPT (p3r_take_proof): proving again to learn nothing but the size
// would double the call's cost, and under zk = 1 - or a hiding MMCS - the second proof is another one, of another length.
static int emit_proof(p3r_ctx* ctx, std::vector<uint8_t>&& bytes, uint8_t* buf, size_t cap, size_t* len) {
  *len = bytes.size();
  ctx->pending_proof.clear();
  if (bytes.size() > cap) {
    const size_t need = bytes.size();
    ctx->pending_proof = std::move(bytes);
    fail(P3R_EBUFFER, "proof needs %zu bytes, buffer holds %zu (p3r_take_proof hands it over)", need, cap);
  }
  memcpy(buf, bytes.data(), bytes.size());
  return 0;
}

int p3r_take_proof(p3r_ctx* ctx, uint8_t* proof_buf, size_t proof_cap, size_t* proof_len) {
  return guard(ctx, [&] {
    if (!proof_len || (!proof_buf && proof_cap)) fail(P3R_EINVAL, "bad arguments");
    if (ctx->pending_proof.empty()) fail(P3R_EINVAL, "no proof is waiting: the last prove call on this context did not return P3R_EBUFFER");
    *proof_len = ctx->pending_proof.size();
    if (ctx->pending_proof.size() > proof_cap)
      fail(P3R_EBUFFER, "proof needs %zu bytes, buffer holds %zu", ctx->pending_proof.size(), proof_cap);
    memcpy(proof_buf, ctx->pending_proof.data(), ctx->pending_proof.size());
    std::vector<uint8_t>().swap(ctx->pending_proof);
  });
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
    emit_proof(ctx, std::move(bytes), proof_buf, proof_cap, proof_len);
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
    emit_proof(ctx, std::move(bytes), proof_buf, proof_cap, proof_len);
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
int p3r_layer_p2w_height(const p3r_layer* L, size_t* h) {
  if (!L || !h) return P3R_EINVAL;
  *h = L->h_p2w;
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
    emit_proof(ctx, std::move(bytes), proof_buf, proof_cap, proof_len);
  });
}
int p3r_prove_all_tables(p3r_ctx* ctx, const p3r_layer* layer, const p3r_traces* traces, uint32_t flags,
                         uint8_t* proof_buf, size_t proof_cap, size_t* proof_len) {
  return guard(ctx, [&] {
    if (!layer || !traces || !proof_len || (!proof_buf && proof_cap)) fail(P3R_EINVAL, "bad arguments");
    auto d = P3R_FIELD_CALL(ctx, traces_upload, ctx, layer, traces);
    auto bytes = P3R_FIELD_CALL(ctx, prove_all_tables, ctx, layer, d.get(),
                                (flags & P3R_PROVE_CANONICAL_FIELD_ENCODING) != 0);
    emit_proof(ctx, std::move(bytes), proof_buf, proof_cap, proof_len);
  });
}
p3r_dmat* p3r_layer_build_main_trace(p3r_ctx* ctx, const p3r_layer* layer, const p3r_dtraces* traces,
                                     uint32_t table) {
  p3r_dmat* out = nullptr;
  guard(ctx, [&] {
    if (!layer || !traces || table > 6) fail(P3R_EINVAL, "bad arguments");
    if (layer->slot_of((int)table) < 0) fail(P3R_EINVAL, "table %u has no rows and is not part of the batch", table);
    auto m = P3R_FIELD_CALL(ctx, build_main_traces, ctx, layer, traces);
    P3R_HIP(hipStreamSynchronize(ctx->stream));
    out = m[table].release();
  });
  return out;
}

// ---- circuit boundary (circuit_impl.hip.h) ----
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
  emit_proof(ctx, std::move(bytes), proof_buf, proof_cap, proof_len);
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
