// prove_batch on the GPU: the body the reference delegates to p3_batch_stark::prove_batch
// (call site circuit-prover/src/batch_stark_prover.rs:1595).  Included into p3r_core.hip.
//
// Stage order and transcript are pinned by the in-tree circuit verifier:
//   transcript head, commitments, challenges   recursion/src/verifier/batch_stark.rs:521-627
//   rounds / opening points                     recursion/src/verifier/batch_stark.rs:645-852
//   opened-value observation order              recursion/src/verifier/batch_stark.rs:1114-1276
//   quotient domain / chunks                    recursion/src/verifier/batch_stark.rs:701-717,
//                                               recursion/src/verifier/quotient.rs:60-140
//   FRI challenges, arities, PoW, queries       recursion/src/pcs/fri/targets.rs:748-866
//   reduced openings / folding / final poly     recursion/src/pcs/fri/verifier.rs:562-781,887-915,1068-1356
//   ZK (p3r_config.zk, HidingFriPcs)            recursion/src/verifier/batch_stark.rs:424-428,487-490,536,623-661,701-735,
//                                               855-864,1116-1260; pcs/fri/targets.rs:1076-1130 - the prover side is
//                                               un-vendored and randomised: written from those acceptance conditions
// All matrices stay resident in HBM; only commitments, opened values, the final polynomial
// and the query answers cross to the host.
#include "host_transcript.h"
#include "kernels_stark.hip.h"
#include "kernels_fri_reduce.hip.h"
#include "kernels_zk.hip.h"
#include "open_impl.hip.h"

struct p3r_prep {
  std::vector<p3r::AirParams> airs;
  std::vector<std::unique_ptr<p3r_dmat>> traces;  // preprocessed traces (K7 and openings read them)
  std::vector<std::unique_ptr<p3r_dmat>> evals;   // ZK: the committed evaluations, 2h x (w + R), zero-padded (else empty)
  std::vector<std::unique_ptr<p3r_dmat>> ldes;    // their bit-reversed LDEs (K8, FRI)
  std::vector<size_t> heights;                    // BASE trace heights
  int zk_codewords = -1;                          // R the preparation was committed with (0: not ZK)
  const p3r_dmat* committed(size_t i) const { return evals.empty() ? traces[i].get() : evals[i].get(); }
  std::unique_ptr<p3r_tree> tree;
  std::vector<uint32_t> cap_canonical;
};

namespace {

template <class PP, int DC = 4>
EW<DC> to_e4(const typename Chal<PP, DC>::type& e) { return e4_store<PP, DC>(e); }

template <class PP, int DC = 4>
std::vector<typename Chal<PP, DC>::type> download_ef(p3r_ctx* ctx, const uint32_t* dev, size_t count) {
  std::vector<uint32_t> raw(count * DC);
  P3R_HIP(fetch_small(ctx, dev, raw.size(), raw.data()));
  std::vector<typename Chal<PP, DC>::type> out(count);
  for (size_t i = 0; i < count; ++i)
    for (int k = 0; k < DC; ++k) out[i].c[k] = Fp<PP>::raw(raw[i * DC + k]);
  return out;
}

// Merkle layers above a leaf-digest layer (no injections): used by the FRI commit phase.
template <class PP>
void build_plain_layers(p3r_ctx* ctx, p3r_tree* tree, size_t n_leaves, TranscriptStep* step = nullptr) {
  size_t n = n_leaves;
  const size_t cap_n = size_t(1) << tree->cap_height;
  while (n > cap_n) {
    const size_t after = mmcs_subtree<PP>(ctx, tree, n, nullptr, step);
    if (after != n) {
      n = after;
      continue;
    }
    const size_t nn = n / 2;
    DevBuf next(P2_DIGEST * nn);
    const uint32_t* prev = tree->layers.back().p;
    launch_compress<PP>(ctx, prev, nullptr, next.p, nn);
    tree->layers.push_back(std::move(next));
    n = nn;
  }
}

template <class PP>
std::vector<uint32_t> download_cap_mont(p3r_ctx* ctx, const p3r_tree* tree) {
  const size_t cap_n = size_t(1) << tree->cap_height;
  std::vector<uint32_t> soa(P2_DIGEST * cap_n), cap(P2_DIGEST * cap_n);
  P3R_HIP(fetch_small(ctx, tree->layers.back().p, soa.size(), soa.data()));
  for (size_t j = 0; j < cap_n; ++j)
    for (int k = 0; k < P2_DIGEST; ++k) cap[j * P2_DIGEST + k] = soa[(size_t)k * cap_n + j];
  return cap;
}

// The context's generator key with a proof's nonce (zk_rand.h).
inline ZkKey zk_key_of(const p3r_ctx* ctx, uint64_t nonce) {
  ZkKey k{};
  for (int i = 0; i < 8; ++i) k.k[i] = ctx->zk_key[i];
  k.nonce_lo = (uint32_t)nonce; k.nonce_hi = (uint32_t)(nonce >> 32);
  return k;
}

// the tiled fills of kernels_zk.hip.h (one ChaCha block per eight cells); P3R_ZK_CELL_FILL=1 (knobs build): the per-cell kernels
inline bool zk_cell_fill() {
  static const bool on = tuning_knob("P3R_ZK_CELL_FILL") != nullptr;
  return on;
}
template <class PP>
void launch_zk_tiles(p3r_ctx* ctx, std::vector<ZkTileJob>& jobs, const ZkKey& key, const char* family) {
  uint64_t blocks = 0;
  for (auto& j : jobs) {
    if (2 * (uint64_t)(j.w2 | 1u) > kZkTileWords) fail(P3R_EUNSUPPORTED, "ZK fill of a matrix of %u columns", j.w2);
    j.log_tr = zk_tile_log_rows(j.rows, j.w2);
    j.block0 = (uint32_t)blocks;
    blocks += j.rows >> j.log_tr;
  }
  if (blocks >= (uint64_t(1) << 31)) fail(P3R_EUNSUPPORTED, "ZK fill launch of %llu tiles", (unsigned long long)blocks);
  if (!blocks) return;
  DevBuf d_jobs((jobs.size() * sizeof(ZkTileJob) + 3) / 4);
  P3R_HIP(ctx->stage.upload(ctx->stream, d_jobs.p, jobs.data(), jobs.size() * sizeof(ZkTileJob)));
  ProfScope ps(ctx, family);
  hipLaunchKernelGGL(k_zk_fill_tiles<PP>, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, reinterpret_cast<const ZkTileJob*>(d_jobs.p),
                     (int)jobs.size(), key);
  P3R_HIP(hipGetLastError());
}

// Salt matrices of a hiding MMCS for the matrices of one batch (commit order): h x S each, one launch.
template <class PP>
std::vector<std::unique_ptr<p3r_dmat>> draw_salts(p3r_ctx* ctx, const std::vector<const p3r_dmat*>& mats, int salt_round, const ZkKey& key) {
  const uint32_t S = ctx->cfg.mmcs_salt_elems;
  std::vector<std::unique_ptr<p3r_dmat>> out;
  if (!zk_cell_fill()) {
    std::vector<ZkTileJob> tj;
    for (size_t i = 0; i < mats.size(); ++i) {
      out.push_back(dmat_alloc(mats[i]->h, S));
      ZkTileJob j{};
      j.dst = out.back()->d; j.rows = mats[i]->h; j.w2 = S; j.mode = 1; j.stride = 1;
      j.stream = zk_stream_id(salt_round, i);
      tj.push_back(j);
    }
    launch_zk_tiles<PP>(ctx, tj, key, "mmcs_salts");
    return out;
  }
  std::vector<ZkSaltJob> jobs;
  uint64_t blocks = 0;
  for (size_t i = 0; i < mats.size(); ++i) {
    out.push_back(dmat_alloc(mats[i]->h, S));
    ZkSaltJob j{};
    j.dst = out.back()->d; j.h = mats[i]->h; j.S = S; j.stride = 1;
    j.stream = zk_stream_id(salt_round, i);
    j.block0 = (uint32_t)blocks;
    blocks += (uint64_t)S * ((j.h + kBlock - 1) / kBlock);
    jobs.push_back(j);
  }
  if (blocks >= (uint64_t(1) << 31)) fail(P3R_EUNSUPPORTED, "salt launch of %llu tiles", (unsigned long long)blocks);
  DevBuf d_jobs((jobs.size() * sizeof(ZkSaltJob) + 3) / 4);
  P3R_HIP(ctx->stage.upload(ctx->stream, d_jobs.p, jobs.data(), jobs.size() * sizeof(ZkSaltJob)));
  ProfScope ps(ctx, "mmcs_salts");
  hipLaunchKernelGGL(k_zk_salts<PP>, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, reinterpret_cast<const ZkSaltJob*>(d_jobs.p),
                     (int)jobs.size(), key);
  P3R_HIP(hipGetLastError());
  return out;
}

// `salt_round` >= 0 under a hiding MMCS (p3r_config.mmcs_salt_elems > 0): the stream round of this batch's salts.
template <class PP>
std::unique_ptr<p3r_tree> commit_dmats(p3r_ctx* ctx, const std::vector<const p3r_dmat*>& mats,
                                       std::vector<uint32_t>& cap_mont, std::map<size_t, DevBuf>* pre = nullptr,
                                       int salt_round = -1, const ZkKey* key = nullptr) {
  auto tree = std::make_unique<p3r_tree>();
  tree->mats = mats;
  if (ctx->cfg.mmcs_salt_elems) {
    if (salt_round < 0 || !key) fail(P3R_EHIP, "internal: a commit without its salt stream under a hiding MMCS");
    if (pre && !pre->empty()) fail(P3R_EHIP, "internal: pre-hashed classes under a hiding MMCS");
    tree->salt_elems = (int)ctx->cfg.mmcs_salt_elems;
    tree->salt_owned = draw_salts<PP>(ctx, mats, salt_round, *key);
    tree->mats.clear();
    for (size_t i = 0; i < mats.size(); ++i) { tree->mats.push_back(mats[i]); tree->mats.push_back(tree->salt_owned[i].get()); }
  }
  std::vector<uint32_t> cap_canon((size_t)P2_DIGEST << ctx->cfg.cap_height);
  mmcs_commit<PP>(ctx, tree.get(), cap_canon.data(), pre);
  cap_mont.resize(cap_canon.size());
  for (size_t i = 0; i < cap_canon.size(); ++i) cap_mont[i] = Fp<PP>::from_canonical(cap_canon[i]).v;
  return tree;
}

// Coset LDE + MMCS commitment of one round's matrices (TwoAdicFriPcs::commit).  The product runs them back to back on the
// ctx's stream.  The `knobs` build can take the leaf hashing of ONE height class off the critical path instead
// (P3R_COMMIT_OVERLAP=1): the class with the most permutations is extended first and hashed on the ctx's second stream
// while the main stream extends the other classes - the hash is VALU-bound (0.96 busy), the LDE passes leave a third of
// their time to memory phases.  Measured (profiles/r05/commit_overlap_ab.txt): between two contexts the pair gains
// 0.55 ms of a 1.9 ms LDE, inside the prover 0.0 - 0.2 ms of 28 (paired, 40 proofs per form alternating in one
// process): below the 0.3 ms a change has to earn, so it is not the product's path.  Same digests, same tree, same bytes
// (tests/test_gpu_cpp_host.py::test_two_stream_commit_gives_the_same_proof).
template <class PP>
std::unique_ptr<p3r_tree> lde_and_commit(p3r_ctx* ctx, const std::vector<LdeItem>& items, int log_blowup,
                                         std::vector<std::unique_ptr<p3r_dmat>>& ldes, std::vector<uint32_t>& cap_mont,
                                         int salt_round = -1, const ZkKey* key = nullptr) {
  // (read per call, not once: tools/ab_commit_overlap.py alternates the forms proof by proof inside one process)
  const bool off = tuning_knob("P3R_COMMIT_OVERLAP") == nullptr;
  // A/B forms (knobs build): 1 = the same split of the LDE and of the hash launch on ONE stream (what the split costs by
  // itself), 2 = the hash on a lowest-priority stream (the LDE's workgroups go first wherever both are waiting)
  const int mode = tuning_knob("P3R_COMMIT_OVERLAP_MODE") ? atoi(tuning_knob("P3R_COMMIT_OVERLAP_MODE")) : 0;
  std::map<size_t, uint64_t> perms, cells;   // per height class of the LDEs
  for (auto& it : items) {
    cells[it.in->h] += (uint64_t)it.in->h * it.in->w;
    perms[it.in->h] += it.in->w;             // columns for now
  }
  size_t pick = 0;
  uint64_t best = 0, total_cells = 0;
  for (auto& kv : perms) {
    kv.second = (uint64_t)kv.first * ((kv.second + P2_RATE - 1) / P2_RATE);
    if (kv.second > best) { best = kv.second; pick = kv.first; }
    total_cells += cells[kv.first];
  }
  const bool overlap = !off && !ctx->prof_enabled && ctx->cfg.mmcs_arity != 4 && !ctx->cfg.mmcs_salt_elems && perms.size() >= 2 &&
                       total_cells - cells[pick] >= (uint64_t(1) << 22) && best >= (uint64_t(1) << 20);
  ldes.clear();
  ldes.resize(items.size());
  std::vector<const p3r_dmat*> ptrs(items.size());
  if (!overlap) {
    host_mark("lde: enqueue");
    auto out = coset_lde_batch<PP>(ctx, items, log_blowup);
    host_mark("lde: enqueued; commit: enqueue");
    for (size_t i = 0; i < items.size(); ++i) { ldes[i] = std::move(out[i]); ptrs[i] = ldes[i].get(); }
    return commit_dmats<PP>(ctx, ptrs, cap_mont, nullptr, salt_round, key);
  }
  std::vector<LdeItem> first, rest;
  std::vector<size_t> first_at, rest_at;
  for (size_t i = 0; i < items.size(); ++i) {
    (items[i].in->h == pick ? first : rest).push_back(items[i]);
    (items[i].in->h == pick ? first_at : rest_at).push_back(i);
  }
  {
    auto out = coset_lde_batch<PP>(ctx, first, log_blowup);
    for (size_t k = 0; k < first.size(); ++k) { ldes[first_at[k]] = std::move(out[k]); ptrs[first_at[k]] = ldes[first_at[k]].get(); }
  }
  std::map<size_t, DevBuf> pre;
  // `pre` is written by the side stream: an exception below must not release it while that kernel still runs
  struct DrainSideStream {
    p3r_ctx* ctx;
    bool armed = false;
    ~DrainSideStream() {
      if (!armed) return;
      if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
      if (ctx->stream2_low) (void)hipStreamSynchronize(ctx->stream2_low);
    }
  } drain_on_unwind{ctx};
  {
    std::vector<const p3r_dmat*> cls;   // commit order within the class = the caller's order (stable)
    for (size_t i : first_at) cls.push_back(ptrs[i]);
    const size_t h = cls[0]->h;
    uint32_t* dig = pre.emplace(h, DevBuf(P2_DIGEST * h)).first->second.p;
    hash_rows<PP>(ctx, {cls}, {dig}, /*side=*/mode == 1 ? 0 : mode == 2 ? 2 : 1);
    drain_on_unwind.armed = mode != 1;
  }
  {
    auto out = coset_lde_batch<PP>(ctx, rest, log_blowup);
    for (size_t k = 0; k < rest.size(); ++k) { ldes[rest_at[k]] = std::move(out[k]); ptrs[rest_at[k]] = ldes[rest_at[k]].get(); }
  }
  if (mode != 1) hash_rows_join(ctx);
  auto tree = commit_dmats<PP>(ctx, ptrs, cap_mont, &pre);
  drain_on_unwind.armed = false;   // joined on the main stream: `pre` is released in stream order from here on
  return tree;
}

// Proof-of-work grinding on the device: the smallest witness w such that, after observing w, the
// low `bits` bits of the next sample are zero (recursion/src/challenger/circuit.rs:409-430).
// Leaves the host challenger in the post-check state.  bits == 0: witness 0, transcript untouched.
template <class PP, class Challenger>
Fp<PP> grind_witness(p3r_ctx* ctx, Challenger& ch, int bits) {
  using F = Fp<PP>;
  if (bits == 0) return F::zero();
  if (bits > 30) fail(P3R_EINVAL, "proof-of-work bits must be <= 30");
  GrindArgs g{};
  for (int k = 0; k < 16; ++k) g.state[k] = ch.state[k].v;
  g.n_pending = (int)ch.in_buf.size();
  for (int k = 0; k < g.n_pending; ++k) g.pending[k] = ch.in_buf[k].v;
  g.bits = bits;
  g.rc = ctx->rc.p;
  DevBuf res(1);
  g.result = res.p;
  uint32_t found = 0xFFFFFFFFu;
  // 4x the expected number of tries per launch: a miss (e^-4) costs one more round trip, while a
  // larger batch makes every proof pay for permutations past the witness
  const uint32_t batch = 1u << std::min<uint32_t>(std::max<uint32_t>(bits + 2, 12), 24);
  // 2^(bits+10) candidates all miss with probability e^-1024; stop there instead of sweeping the field
  const uint64_t limit = std::min<uint64_t>(PP::P, (uint64_t(1) << std::min<uint32_t>(bits + 10, 31)) + batch);
  for (uint64_t base = 0; base < limit && found == 0xFFFFFFFFu; base += batch) {
    P3R_HIP(fill_async(ctx->stream, res.p, 0xFF, 4));
    g.base = (uint32_t)base;
    ProfScope ps(ctx, "grind");
    hipLaunchKernelGGL(k_grind<PP>, dim3(batch / kBlock), dim3(kBlock), 0, ctx->stream, g);
    P3R_HIP(fetch_small(ctx, res.p, 1, &found));
  }
  if (found == 0xFFFFFFFFu) fail(P3R_EINVAL, "proof-of-work search found no witness");
  const F w = F::from_canonical(found);
  if (!ch.check_witness(bits, w)) fail(P3R_EHIP, "device PoW witness rejected on host");
  return w;
}

inline int zk_codewords(const p3r_config& cfg) { return cfg.zk ? (cfg.num_random_codewords ? (int)cfg.num_random_codewords : 2) : 0; }

// HidingFriPcs::commit on a batch of matrices (kernels_zk.hip.h): h x w -> 2h x (w + R) each, one launch.
// src[i] == nullptr: a fully random 2 * h[i] x (w[i] + R) matrix (w[i] = the width before the codeword columns).
template <class PP>
std::vector<std::unique_ptr<p3r_dmat>> zk_randomize(p3r_ctx* ctx, const std::vector<const p3r_dmat*>& src, const std::vector<size_t>& h,
                                                    const std::vector<size_t>& w, int R, const std::vector<uint32_t>& streams, const ZkKey& key,
                                                    bool zero_fill) {
  std::vector<std::unique_ptr<p3r_dmat>> out;
  std::vector<ZkRandomizeJob> jobs;
  uint64_t blocks = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    out.push_back(dmat_alloc(2 * h[i], w[i] + (size_t)R));
    ZkRandomizeJob j{};
    j.src = src[i] ? src[i]->d : nullptr;
    j.dst = out.back()->d;
    j.h2 = 2 * h[i];
    j.w = src[i] ? (uint32_t)w[i] : 0u;
    j.w2 = (uint32_t)(w[i] + (size_t)R);
    j.zero_fill = zero_fill ? 1u : 0u;
    j.stream = streams[i];
    if ((uint64_t)j.h2 * j.w2 >= (uint64_t(1) << 35)) fail(P3R_EUNSUPPORTED, "ZK: a matrix of 2^35 cells or more");
    j.block0 = (uint32_t)blocks;
    blocks += (uint64_t)j.w2 * ((j.h2 + kBlock - 1) / kBlock);
    jobs.push_back(j);
  }
  if (blocks >= (uint64_t(1) << 31)) fail(P3R_EUNSUPPORTED, "ZK randomisation launch of %llu tiles", (unsigned long long)blocks);
  if (jobs.empty()) return out;
  if (!zk_cell_fill() && !zero_fill) {   // (a zero fill draws nothing: the per-cell kernel is the cheap one there)
    std::vector<ZkTileJob> tj;
    for (auto& j : jobs) {
      ZkTileJob t{};
      t.src = j.src; t.dst = j.dst; t.rows = j.h2; t.w = j.w; t.w2 = j.w2; t.stride = 1; t.stream = j.stream;
      t.mode = j.src ? 0u : 1u;   // no source matrix (the random round): every cell is random
      tj.push_back(t);
    }
    launch_zk_tiles<PP>(ctx, tj, key, "zk_randomize");
    return out;
  }
  DevBuf d_jobs((jobs.size() * sizeof(ZkRandomizeJob) + 3) / 4);
  P3R_HIP(ctx->stage.upload(ctx->stream, d_jobs.p, jobs.data(), jobs.size() * sizeof(ZkRandomizeJob)));
  ProfScope ps(ctx, "zk_randomize");
  hipLaunchKernelGGL(k_zk_randomize<PP>, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                     reinterpret_cast<const ZkRandomizeJob*>(d_jobs.p), (int)jobs.size(), key);
  P3R_HIP(hipGetLastError());
  return out;
}

// `dev_traces`: the preprocessed traces already in HBM (column-major, Montgomery: the device-side preparation
// writes them there); otherwise they are uploaded from the host matrices `mats`.
template <class PP>
std::unique_ptr<p3r_prep> prep_create(p3r_ctx* ctx, const p3r_air_desc* airs, const p3r_matrix* mats, size_t n,
                                      std::vector<std::unique_ptr<p3r_dmat>>* dev_traces = nullptr) {
  using F = Fp<PP>;
  auto prep = std::make_unique<p3r_prep>();
  std::vector<const p3r_dmat*> ptrs;
  std::vector<LdeItem> items;
  for (size_t i = 0; i < n; ++i) {
    AirParams a{(int)airs[i].kind, (int)airs[i].lanes, (int)airs[i].horner_packed_steps, (int)airs[i].coeff_lookups,
                (ctx->cfg.ext_choices & P3R_EXT_LOOKUP_UNPACKED) ? 1 : 0, (int)ctx->cfg.ext_degree,
                ext_degree_is_binomial_generic(ctx->cfg.ext_degree) ? Fp<PP>::from_canonical(ctx->cfg.ext_w).v : 0u};
    if (ext_degree_is_binomial_generic(ctx->cfg.ext_degree) && a.kind == AIR_POSEIDON2)
      fail(P3R_EUNSUPPORTED, "instance %zu: UnsupportedDegree(%d): no Poseidon2 table for this circuit degree", i, a.ext_d);
    if (a.kind < 0 || a.kind > AIR_POSEIDON2_W32) fail(P3R_EINVAL, "instance %zu: unknown AIR kind %d", i, a.kind);
    if (a.kind == AIR_POSEIDON2_W32) require_w32_constants(ctx);
    if (a.kind == AIR_POSEIDON2_W32 && ctx->cfg.ext_degree != 4)
      fail(P3R_EUNSUPPORTED, "instance %zu: the width-32 Poseidon2 table belongs to D = 4 circuits", i);
    if (a.lanes < 1) fail(P3R_EINVAL, "instance %zu: lanes must be positive", i);
    if (a.kind == AIR_ALU && (a.horner_k < 2 || a.horner_k > 8))
      fail(P3R_EINVAL, "instance %zu: horner_packed_steps must be in 2..8", i);
    const size_t width = dev_traces ? (*dev_traces)[i]->w : mats[i].width, height = dev_traces ? (*dev_traces)[i]->h : mats[i].height;
    if ((int)width != air_prep_width_of(a))
      fail(P3R_EINVAL, "instance %zu: preprocessed width %zu, the AIR expects %d", i, width, air_prep_width_of(a));
    {
      // known at preparation time: a quotient of 2^(log_chunks) cosets (twice as many under ZK) is evaluated ON the
      // committed LDE here, so log_chunks must not exceed log_blowup.  Upstream re-evaluates on the larger domain
      // instead; that is a capability this prover lacks, not a caller error - said so, and said early
      const LookupLayout lay = lookup_layout(a, ctx->cfg.zk ? 1 : 0);
      if (lay.log_chunks > (int)ctx->cfg.log_blowup)
        fail(P3R_EUNSUPPORTED, "instance %zu: quotient domain larger than the LDE: constraint degree needs log_blowup >= %d%s, log_blowup is %u "
             "(upstream extends the traces to the quotient domain in this case; this prover does not)", i, lay.log_chunks,
             ctx->cfg.zk ? " under zk = 1" : "", ctx->cfg.log_blowup);
    }
    prep->airs.push_back(a);
    prep->heights.push_back(height);
    auto m = dev_traces ? std::move((*dev_traces)[i]) : upload<PP>(ctx, mats[i].values, mats[i].height, mats[i].width);
    items.push_back({m.get(), PP::GEN});
    prep->traces.push_back(std::move(m));
  }
  prep->zk_codewords = zk_codewords(ctx->cfg);
  if (ctx->cfg.zk) {
    // commit_preprocessing of a hiding PCS: the same extended domain and codeword columns, padded with ZEROS - public
    // data, and the commitment stays a function of the circuit shape (p3r.h: p3r_config.zk)
    std::vector<const p3r_dmat*> src;
    std::vector<size_t> hs, ws;
    for (auto& t : prep->traces) { src.push_back(t.get()); hs.push_back(t->h); ws.push_back(t->w); }
    prep->evals = zk_randomize<PP>(ctx, src, hs, ws, prep->zk_codewords, std::vector<uint32_t>(n, 0), ZkKey{}, true);
    for (size_t i = 0; i < n; ++i) items[i].in = prep->evals[i].get();
  }
  prep->ldes = coset_lde_batch<PP>(ctx, items, (int)ctx->cfg.log_blowup);
  for (auto& l : prep->ldes) ptrs.push_back(l.get());
  std::vector<uint32_t> cap_mont;
  {
    // a hiding MMCS salts the preprocessed commitment too; it is made once per circuit: the nonce of `proof 0`
    const ZkKey prep_key = zk_key_of(ctx, 0);
    prep->tree = commit_dmats<PP>(ctx, ptrs, cap_mont, nullptr, kSaltRound + ZK_ROUND_PREP, &prep_key);
  }
  prep->cap_canonical.resize(cap_mont.size());
  for (size_t i = 0; i < cap_mont.size(); ++i) prep->cap_canonical[i] = F::raw(cap_mont[i]).to_canonical();
  return prep;
}

struct RoundTrees {
  const p3r_tree* tree;
};

// DC: the degree of the challenge field (p3r_config.challenge_degree): 4, or 5 = KoalaBear's quintic trinomial
// extension.  Extension vectors on the device are DC planes ([DC][n]); the proof writes DC words per element.
template <class PP, int DC = 4>
std::vector<uint8_t> prove_batch(p3r_ctx* ctx, const p3r_prep* prep, const p3r_dmat* const* mains, size_t ni,
                                 bool canonical_encoding) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  const p3r_config& cfg = ctx->cfg;
  const int log_blowup = (int)cfg.log_blowup;
  host_timeline_begin();
  {
    // the waits of this proof repeat those of the last proof of the same shape (HostPost::post)
    uint64_t shape = 0x9E3779B97F4A7C15ull ^ ni;
    for (size_t i = 0; i < ni; ++i) shape = (shape ^ (mains[i]->h * 31 + mains[i]->w)) * 0x100000001B3ull;
    // (not while profiling: the stage marks drain the stream, the results are there when they are asked for)
    if (ctx->prof_enabled) ctx->post.end_proof(); else ctx->post.begin_proof(shape);
  }
  struct EndProofWaits {
    HostPost& p;
    ~EndProofWaits() { p.end_proof(); }
  } end_proof_waits{ctx->post};
  if (ni != prep->airs.size()) fail(P3R_EINVAL, "%zu traces for %zu preprocessed instances", ni, prep->airs.size());
  const int p2w = p2_perm_cols<PP>() + 2;
  // ZK (HidingFriPcs): every committed matrix lives over the extended trace domain (log_e = log_n + 1) with R random
  // codeword columns; `nonce` keys this proof's random values (zk_rand.h)
  const int zk = cfg.zk ? 1 : 0, R = zk_codewords(cfg);
  if (prep->zk_codewords != R) fail(P3R_EINVAL, "the preprocessed data was committed under another ZK setting");
  const uint64_t nonce = (zk || cfg.mmcs_salt_elems) ? ctx->zk_nonce++ : 0;   // (the salts of a hiding MMCS are per proof too)
  ZkKey zk_key{};
  for (int i = 0; i < 8; ++i) zk_key.k[i] = ctx->zk_key[i];
  zk_key.nonce_lo = (uint32_t)nonce; zk_key.nonce_hi = (uint32_t)(nonce >> 32);
  auto zkey = [&](int round, size_t mat) { return zk_stream_id(round, mat); };
  std::vector<LookupLayout> layouts(ni);
  std::vector<int> log_n(ni), log_e(ni);
  for (size_t i = 0; i < ni; ++i) {
    const AirParams& a = prep->airs[i];
    if (mains[i]->h != prep->heights[i])
      fail(P3R_EINVAL, "instance %zu: trace height %zu != preprocessed height %zu", i, mains[i]->h, prep->heights[i]);
    if ((int)mains[i]->w != air_width_of(a, p2w, p2w_perm_cols<PP>() + 4))
      fail(P3R_EINVAL, "instance %zu: trace width %zu, the AIR expects %d", i, mains[i]->w, air_width_of(a, p2w, p2w_perm_cols<PP>() + 4));
    log_n[i] = log2_exact(mains[i]->h, "trace height");
    log_e[i] = log_n[i] + zk;
    layouts[i] = lookup_layout(a, zk);
    if (layouts[i].log_chunks > log_blowup) fail(P3R_EUNSUPPORTED, "quotient domain larger than the LDE (zk %d, log_blowup %d)", zk, log_blowup);
    if ((1 << (layouts[i].log_chunks + zk)) > 8) fail(P3R_EUNSUPPORTED, "more than 8 quotient chunks");
  }
  const F gen = F::generator();
  std::vector<F> rc_host(ctx->rc_canonical.size());
  std::vector<uint32_t> rc_mont(ctx->rc_canonical.size());
  for (size_t i = 0; i < rc_mont.size(); ++i) rc_mont[i] = F::from_canonical(ctx->rc_canonical[i]).v;
  HostChallenger<PP, DC> ch(rc_mont.data());
  ProofWriter<PP, DC> W;
  W.canonical = canonical_encoding;

  prof_stage(ctx, "main_lde_commit");
  // ---- 1. main LDEs + commitment
  std::vector<LdeItem> lde_items;
  std::vector<std::unique_ptr<p3r_dmat>> main_r;   // ZK: the randomised traces (what is committed and opened)
  std::vector<const p3r_dmat*> main_ev(mains, mains + ni);
  if (zk) {
    std::vector<size_t> hs, ws;
    std::vector<uint32_t> keys;
    for (size_t i = 0; i < ni; ++i) { hs.push_back(mains[i]->h); ws.push_back(mains[i]->w); keys.push_back(zkey(ZK_ROUND_MAIN, i)); }
    main_r = zk_randomize<PP>(ctx, main_ev, hs, ws, R, keys, zk_key, false);
    for (size_t i = 0; i < ni; ++i) main_ev[i] = main_r[i].get();
  }
  for (size_t i = 0; i < ni; ++i) lde_items.push_back({main_ev[i], PP::GEN});
  std::vector<std::unique_ptr<p3r_dmat>> main_lde;
  std::vector<const p3r_dmat*> ptrs;
  std::vector<uint32_t> main_cap, perm_cap, quot_cap, rand_cap;
  auto main_tree = lde_and_commit<PP>(ctx, lde_items, log_blowup, main_lde, main_cap, kSaltRound + ZK_ROUND_MAIN, &zk_key);

  prof_stage(ctx, "transcript_head");
  // ---- 2. transcript head
  ch.observe_base_as_ext(ni);
  for (size_t i = 0; i < ni; ++i) {
    ch.observe_base_as_ext(log_e[i]);   // extended degree bits, base degree bits, width, chunk count (:538-558)
    ch.observe_base_as_ext(log_n[i]);
    ch.observe_base_as_ext(mains[i]->w);
    ch.observe_base_as_ext(uint64_t(1) << (layouts[i].log_chunks + zk));
  }
  for (uint32_t v : main_cap) ch.observe(F::raw(v));
  for (size_t i = 0; i < ni; ++i) ch.observe_base_as_ext(air_prep_width_of(prep->airs[i]));
  for (uint32_t v : prep->cap_canonical) ch.observe(F::from_canonical(v));

  prof_stage(ctx, "logup_aux_commit");
  // ---- 3. LogUp: challenges, aux traces, commitment, terminals
  bool any_lookup = false;
  for (auto& L : layouts) any_lookup |= L.n_groups > 0;
  LookupChT<DC> lc{};
  if (any_lookup) {
    E alpha_l = ch.sample_ext(), beta_l = ch.sample_ext();
    E bp = E::one();
    // the widest tuple on the bus is (idx, v_0..v_{D-1}): gamma = beta^(D+1)
    const int tuple_w = (int)ctx->cfg.ext_degree + 1;
    for (int j = 0; j < tuple_w; ++j) { lc.beta_pow[j] = to_e4<PP, DC>(bp); bp *= beta_l; }
    lc.prefix = to_e4<PP, DC>(alpha_l + bp);  // alpha + beta^(D+1), bus id 0
  }
  std::vector<std::unique_ptr<p3r_dmat>> aux(ni), aux_lde(ni), aux_r(ni);
  std::vector<const p3r_dmat*> aux_ev(ni, nullptr);   // the committed permutation evaluations (ZK: randomised)
  std::vector<E> terminals(ni, E::zero());
  std::vector<int> perm_insts;
  std::unique_ptr<p3r_tree> perm_tree;
  if (any_lookup) {
    DevBuf totals(DC * ni);
    P3R_HIP(fill_async(ctx->stream, totals.p, 0, 4 * DC * ni));
    // aux traces of all tables: fractions per row, then the running sum as a three-phase scan
    std::vector<LogupJob> jobs;
    std::vector<DevBuf> scratch;
    uint32_t row_blocks = 0, tiles = 0;
    for (size_t i = 0; i < ni; ++i) {
      const auto& L = layouts[i];
      if (!L.n_groups) continue;
      const size_t n = mains[i]->h;
      aux[i] = dmat_alloc(n, (size_t)L.aux_width() * DC);
      LogupJob j{};
      j.air = prep->airs[i];
      j.main = mains[i]->d;
      // the preprocessed TRACE (not the LDE): kept from preparation for this pass and the openings
      j.prep = prep->traces[i]->d;
      j.aux = aux[i]->d;
      j.n = n;
      j.pair = L.pair;
      j.n_tiles = (uint32_t)((n + kScanTile - 1) / kScanTile);
      scratch.emplace_back(DC * n);
      j.rowsum = scratch.back().p;
      scratch.emplace_back(DC * (size_t)j.n_tiles);
      j.agg = scratch.back().p;
      j.total = totals.p + DC * i;
      j.block0 = row_blocks;
      j.tile0 = tiles;
      row_blocks += blocks_for(n);
      tiles += j.n_tiles;
      jobs.push_back(j);
      perm_insts.push_back((int)i);
    }
    {
      scratch.emplace_back((jobs.size() * sizeof(LogupJob) + 3) / 4);
      const auto* d_jobs = reinterpret_cast<const LogupJob*>(scratch.back().p);
      P3R_HIP(ctx->stage.upload(ctx->stream, scratch.back().p, jobs.data(), jobs.size() * sizeof(LogupJob)));
      const int nj = (int)jobs.size();
      ProfScope ps(ctx, "logup_aux");
      launch_logup_aux<PP, DC>(ctx, row_blocks, d_jobs, nj, lc);
      hipLaunchKernelGGL((k_ef_scan<PP, DC>), dim3(tiles), dim3(kBlock), 0, ctx->stream, 0, d_jobs, nj);
      hipLaunchKernelGGL((k_ef_scan<PP, DC>), dim3((unsigned)nj), dim3(kBlock), 0, ctx->stream, 1, d_jobs, nj);
      hipLaunchKernelGGL((k_ef_scan<PP, DC>), dim3(tiles), dim3(kBlock), 0, ctx->stream, 2, d_jobs, nj);
      P3R_HIP(hipGetLastError());
    }
    for (int i : perm_insts) aux_ev[i] = aux[i].get();
    if (zk) {
      std::vector<const p3r_dmat*> src;
      std::vector<size_t> hs, ws;
      std::vector<uint32_t> keys;
      for (size_t k = 0; k < perm_insts.size(); ++k) {
        const p3r_dmat* a = aux[perm_insts[k]].get();
        src.push_back(a); hs.push_back(a->h); ws.push_back(a->w); keys.push_back(zkey(ZK_ROUND_PERM, k));
      }
      auto rs = zk_randomize<PP>(ctx, src, hs, ws, R, keys, zk_key, false);
      for (size_t k = 0; k < perm_insts.size(); ++k) {
        aux_r[perm_insts[k]] = std::move(rs[k]);
        aux_ev[perm_insts[k]] = aux_r[perm_insts[k]].get();
      }
    }
    lde_items.clear();
    for (int i : perm_insts) lde_items.push_back({aux_ev[i], PP::GEN});
    std::vector<std::unique_ptr<p3r_dmat>> ldes;
    perm_tree = lde_and_commit<PP>(ctx, lde_items, log_blowup, ldes, perm_cap, kSaltRound + ZK_ROUND_PERM, &zk_key);
    for (size_t k = 0; k < perm_insts.size(); ++k) aux_lde[perm_insts[k]] = std::move(ldes[k]);
    {
      // every table's global sum in one transfer
      auto all = download_ef<PP, DC>(ctx, totals.p, ni);
      for (int i : perm_insts) terminals[i] = all[i];
    }
    for (uint32_t v : perm_cap) ch.observe(F::raw(v));
    for (int i : perm_insts) ch.observe_ext(terminals[i]);
  }

  prof_stage(ctx, "quotient_commit");
  // ---- 4. alpha, quotient chunks, commitment
  const E alpha = ch.sample_ext();
  struct Chunk { int inst; F shift; std::unique_ptr<p3r_dmat> evals, lde; };
  std::vector<Chunk> chunks;
  std::vector<std::unique_ptr<p3r_dmat>> chunk_bufs_keep;
  // one table of alpha powers for all AIRs: constraint k of N is weighted alpha^(N-1-k)
  auto n_constraints = [&](size_t i) {
    return air_num_base_constraints<PP>(prep->airs[i]) + (layouts[i].n_groups ? layouts[i].n_groups + 3 : 0);
  };
  int n_max = 1;
  for (size_t i = 0; i < ni; ++i) n_max = std::max(n_max, n_constraints(i));
  DevBuf d_apow((size_t)n_max * DC);
  {
    std::vector<uint32_t> apow((size_t)n_max * DC);
    E p = E::one();
    for (int k = 0; k < n_max; ++k) {
      for (int c = 0; c < DC; ++c) apow[DC * k + c] = p.c[c].v;
      p *= alpha;
    }
    P3R_HIP(ctx->stage.upload(ctx->stream, d_apow.p, apow.data(), apow.size() * 4));
  }
  std::vector<QuotientArgs> quot_jobs;  // every table's quotient in one launch
  uint32_t quot_blocks = 0;
  for (size_t i = 0; i < ni; ++i) {
    const AirParams& a = prep->airs[i];
    const auto& L = layouts[i];
    // quotient domain of 2^(log_qd + is_zk) chunk cosets of the BASE trace size (batch_stark.rs:701-717)
    const int lq = L.log_chunks + zk, C = 1 << lq;
    const size_t n = mains[i]->h;
    const int n_base = air_num_base_constraints<PP>(a);
    QuotientArgs q{};
    q.n_constraints = n_constraints(i);
    q.air = a;
    q.main = main_lde[i]->d;
    q.prep = prep->ldes[i]->d;
    q.aux = L.n_groups ? aux_lde[i]->d : nullptr;
    q.lde_h = main_lde[i]->h;
    q.log_n = log_n[i];
    q.log_chunks = lq;
    q.apow = d_apow.p;
    q.n_base = n_base; q.n_groups = L.n_groups; q.pair = L.pair;
    for (int k = 0; k < DC; ++k) q.terminal[k] = terminals[i].c[k].v;
    q.gen = gen.v;
    const F wq = F::two_adic_generator(log_n[i] + lq);
    q.w_q = wq.v;
    q.g_inv = F::two_adic_generator(log_n[i]).inv().v;
    const F gen_n = gen.pow(n), w_c = F::two_adic_generator(lq);
    for (int c = 0; c < C; ++c) {
      F zh = gen_n * w_c.pow(c) - F::one();
      q.zh[c] = zh.v;
      q.zh_inv[c] = zh.inv().v;
    }
    auto chunk_buf = dmat_alloc(n, (size_t)DC * C);  // [C][DC][n]
    q.out = chunk_buf->d;
    q.block0 = quot_blocks;
    quot_blocks += blocks_for(n << lq);
    quot_jobs.push_back(q);
    for (int c = 0; c < C; ++c) {
      Chunk ck;
      ck.inst = (int)i;
      ck.shift = gen * wq.pow(c);
      ck.evals = std::make_unique<p3r_dmat>();
      ck.evals->d = chunk_buf->d + (size_t)c * DC * n;  // view: n x DC column-major
      ck.evals->h = n;
      ck.evals->w = DC;
      chunks.push_back(std::move(ck));
    }
    chunk_bufs_keep.push_back(std::move(chunk_buf));
  }
  {
    DevBuf d_quot((quot_jobs.size() * sizeof(QuotientArgs) + 3) / 4);
    P3R_HIP(ctx->stage.upload(ctx->stream, d_quot.p, quot_jobs.data(), quot_jobs.size() * sizeof(QuotientArgs)));
    ProfScope ps(ctx, "quotient");
    launch_quotient<PP, DC>(ctx, quot_blocks, reinterpret_cast<const QuotientArgs*>(d_quot.p), (int)quot_jobs.size(), lc);
    P3R_HIP(hipGetLastError());
  }
  std::vector<std::unique_ptr<p3r_dmat>> zk_keep;   // masks, coset moves and the randomised chunk matrices
  if (zk) {
    // HidingFriPcs::commit_quotient from the acceptance side.  The verifier opens chunk c over
    // natural_domain_for_degree(2n) (batch_stark.rs:719-727) and recomposes quotient(zeta) = sum_c zp_c(zeta) q'_c(zeta),
    // zp_c(x) = prod_{j != c} Z_j(x) / Z_j(s_c), Z_j(x) = (x / s_j)^n - 1 (verifier/quotient.rs).  q'_c = q_c + Z_c t_c
    // leaves the sum unchanged iff sum_c k_c t_c = 0, k_c = prod_{j != c} 1 / Z_j(s_c): C - 1 independent random masks
    // t_c of degree < n and t_{C-1} = -(1 / k_{C-1}) sum_{c < C-1} k_c t_c.  Each mask is n random evaluations over the
    // common coset U = u <g_n>, u = s_{C-1} g_2n; the committed matrix of chunk c is q'_c over s_c <g_2n>: even rows
    // the chunk evaluations, odd rows q_c - 2 t_c on s_c g_2n <g_n> (Z_c = g_2n^n - 1 = -2 there), plus R random
    // codeword columns.  q_c and t_c reach the odd coset by a coset move (the LDE with no added bits).
    std::vector<LdeItem> q_moves, t_moves;
    size_t k0 = 0;
    for (size_t i = 0; i < ni; ++i) {
      const int lq = layouts[i].log_chunks + zk, C = 1 << lq;
      const size_t n = mains[i]->h;
      const F wq = F::two_adic_generator(log_n[i] + lq), g2 = F::two_adic_generator(log_n[i] + 1);
      std::vector<F> sh(C), kc(C);
      for (int c = 0; c < C; ++c) sh[c] = gen * wq.pow(c);
      for (int c = 0; c < C; ++c) {
        F d = F::one();
        for (int j = 0; j < C; ++j)
          if (j != c) d *= (sh[c] * sh[j].inv()).pow(n) - F::one();
        kc[c] = d.inv();
      }
      const F u = sh[C - 1] * g2, neg_inv_last = -(kc[C - 1].inv());
      ZkMaskArgs ma{};
      ma.n = n; ma.C = C; ma.DC = DC; ma.key = zk_key;
      for (int c = 0; c < C; ++c) {
        zk_keep.push_back(dmat_alloc(n, DC));
        ma.t[c] = zk_keep.back()->d;
        ma.stream[c] = zkey(ZK_ROUND_QMASK, k0 + c);
        ma.coef[c] = (kc[c] * neg_inv_last).v;
        t_moves.push_back({zk_keep.back().get(), (sh[c] * g2 * u.inv()).to_canonical()});
        q_moves.push_back({chunks[k0 + c].evals.get(), g2.to_canonical()});
      }
      {
        ProfScope ps(ctx, "zk_masks");
        hipLaunchKernelGGL(k_zk_masks<PP>, dim3(blocks_for(n * DC)), dim3(kBlock), 0, ctx->stream, ma);
      }
      P3R_HIP(hipGetLastError());
      k0 += C;
    }
    auto q_odd = coset_lde_batch<PP>(ctx, q_moves, 0), t_odd = coset_lde_batch<PP>(ctx, t_moves, 0);
    std::vector<ZkChunkJob> jobs;
    uint64_t blocks = 0;
    for (size_t k = 0; k < chunks.size(); ++k) {
      const size_t n = chunks[k].evals->h;
      auto m = dmat_alloc(2 * n, (size_t)DC + R);
      ZkChunkJob j{};
      j.q = chunks[k].evals->d; j.q_odd = q_odd[k]->d; j.t_odd = t_odd[k]->d; j.dst = m->d;
      j.n = n; j.log_n = log_n[chunks[k].inst]; j.DC = DC; j.R = R;
      j.stream = zkey(ZK_ROUND_QUOTIENT, k);
      j.block0 = (uint32_t)blocks;
      blocks += (uint64_t)(DC + R) * ((2 * n + kBlock - 1) / kBlock);
      jobs.push_back(j);
      chunks[k].evals = std::move(m);   // the committed evaluations: 2n x (DC + R) over s_c <g_2n>
    }
    if (blocks >= (uint64_t(1) << 31)) fail(P3R_EUNSUPPORTED, "ZK chunk launch of %llu tiles", (unsigned long long)blocks);
    DevBuf d_jobs((jobs.size() * sizeof(ZkChunkJob) + 3) / 4);
    P3R_HIP(ctx->stage.upload(ctx->stream, d_jobs.p, jobs.data(), jobs.size() * sizeof(ZkChunkJob)));
    {
      ProfScope ps(ctx, "zk_chunks");
      hipLaunchKernelGGL(k_zk_chunk<PP>, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                         reinterpret_cast<const ZkChunkJob*>(d_jobs.p), (int)jobs.size(), zk_key);
    }
    P3R_HIP(hipGetLastError());
    for (auto& m : q_odd) zk_keep.push_back(std::move(m));   // the stream still reads them
    for (auto& m : t_odd) zk_keep.push_back(std::move(m));
  }
  {
    // commit evaluates each chunk polynomial on gen*<w>: shift = GENERATOR / domain shift
    lde_items.clear();
    for (auto& ck : chunks) lde_items.push_back({ck.evals.get(), (gen * ck.shift.inv()).to_canonical()});
    auto ldes = coset_lde_batch<PP>(ctx, lde_items, log_blowup);
    for (size_t k = 0; k < chunks.size(); ++k) chunks[k].lde = std::move(ldes[k]);
  }
  ptrs.clear();
  for (auto& ck : chunks) ptrs.push_back(ck.lde.get());
  auto quot_tree = commit_dmats<PP>(ctx, ptrs, quot_cap, nullptr, kSaltRound + ZK_ROUND_QUOTIENT, &zk_key);
  for (uint32_t v : quot_cap) ch.observe(F::raw(v));
  // ZK: the random round - per instance a fully random matrix of Challenge::DIMENSION (+ R) columns over the extended
  // trace domain, opened at zeta; its commitment is observed after the quotient's (batch_stark.rs:623-625)
  std::vector<std::unique_ptr<p3r_dmat>> rand_ev, rand_lde;
  std::unique_ptr<p3r_tree> rand_tree;
  if (zk) {
    std::vector<const p3r_dmat*> src(ni, nullptr);
    std::vector<size_t> hs, ws(ni, (size_t)DC);
    std::vector<uint32_t> keys;
    for (size_t i = 0; i < ni; ++i) { hs.push_back(mains[i]->h); keys.push_back(zkey(ZK_ROUND_RANDOM, i)); }
    rand_ev = zk_randomize<PP>(ctx, src, hs, ws, R, keys, zk_key, false);
    lde_items.clear();
    for (auto& m : rand_ev) lde_items.push_back({m.get(), PP::GEN});
    rand_lde = coset_lde_batch<PP>(ctx, lde_items, log_blowup);
    ptrs.clear();
    for (auto& m : rand_lde) ptrs.push_back(m.get());
    rand_tree = commit_dmats<PP>(ctx, ptrs, rand_cap, nullptr, kSaltRound + ZK_ROUND_RANDOM, &zk_key);
    for (uint32_t v : rand_cap) ch.observe(F::raw(v));
  }
  const E zeta = ch.sample_ext();

  prof_stage(ctx, "openings");
  // ---- 5. openings, observed in round / matrix / point order
  // Rounds: [random,] main, quotient, preprocessed, permutation (batch_stark.rs:645-852).  Every committed matrix is
  // opened in full; under ZK the last R values of each opening are the random codewords' - HidingFriPcs splits them off
  // into the opening proof's first item and the verifier merges them back before observing (:855-864, :1116-1260).
  struct Item { int round, mat; const p3r_dmat* lde; int log_h; std::vector<E> z; std::vector<std::vector<E>> vals; size_t job; };
  std::vector<Item> items;
  std::vector<std::vector<std::vector<E>>> o_main(ni), o_prep(ni), o_perm(ni);
  std::vector<std::vector<E>> o_chunks(chunks.size()), o_rand(ni);
  const int r_main = zk, r_quot = zk + 1, r_prep = zk + 2, r_perm = zk + 3;
  Opener<PP, DC> op(ctx);  // keeps the opened values on the device for the reduced openings
  {
    if (zk)
      for (size_t i = 0; i < ni; ++i) {
        size_t j = op.open(rand_ev[i]->d, rand_ev[i]->h, (int)rand_ev[i]->w, F::one(), {zeta});
        items.push_back({0, (int)i, rand_lde[i].get(), log_e[i], {zeta}, {}, j});
      }
    for (size_t i = 0; i < ni; ++i) {
      std::vector<E> pts{zeta};
      // zeta * g of the BASE trace domain (:663-700)
      if (air_uses_next(prep->airs[i])) pts.push_back(zeta * F::two_adic_generator(log_n[i]));
      size_t j = op.open(main_ev[i]->d, main_ev[i]->h, (int)main_ev[i]->w, F::one(), pts);
      items.push_back({r_main, (int)i, main_lde[i].get(), log_e[i], pts, {}, j});
    }
    for (size_t k = 0; k < chunks.size(); ++k) {
      auto& ck = chunks[k];
      size_t j = op.open(ck.evals->d, ck.evals->h, (int)ck.evals->w, ck.shift, {zeta});
      items.push_back({r_quot, (int)k, ck.lde.get(), log_e[ck.inst], {zeta}, {}, j});
    }
    for (size_t i = 0; i < ni; ++i) {
      std::vector<E> pts{zeta, zeta * F::two_adic_generator(log_n[i])};
      const p3r_dmat* pt = prep->committed(i);
      size_t j = op.open(pt->d, pt->h, (int)pt->w, F::one(), pts);
      items.push_back({r_prep, (int)i, prep->ldes[i].get(), log_e[i], pts, {}, j});
    }
    for (size_t k = 0; k < perm_insts.size(); ++k) {
      int i = perm_insts[k];
      std::vector<E> pts{zeta, zeta * F::two_adic_generator(log_n[i])};
      size_t j = op.open(aux_ev[i]->d, aux_ev[i]->h, (int)aux_ev[i]->w, F::one(), pts);
      items.push_back({r_perm, (int)k, aux_lde[i].get(), log_e[i], pts, {}, j});
    }
    auto all = op.finish();
    // the proof's own fields hold the values of the AIR's columns: the openings without their R codeword values
    auto head = [&](const std::vector<std::vector<E>>& v) {
      std::vector<std::vector<E>> o;
      for (auto& pv : v) o.emplace_back(pv.begin(), pv.end() - R);
      return o;
    };
    for (auto& it : items) {
      it.vals = all[it.job];
      if (it.round == r_main) o_main[it.mat] = head(it.vals);
      else if (it.round == r_quot) o_chunks[it.mat] = head(it.vals)[0];
      else if (it.round == r_prep) o_prep[it.mat] = head(it.vals);
      else if (it.round == r_perm) o_perm[perm_insts[it.mat]] = head(it.vals);
      else o_rand[it.mat] = head(it.vals)[0];
    }
  }
  // 1/(zeta - x) vectors of the reduced openings depend on the opening points only: the device
  // computes them while the host absorbs the opened values into the transcript
  std::map<std::array<uint64_t, 6>, uint32_t*> inv_cache;  // (log_height, z) -> 1/(z - x_r)
  auto inv_key = [](int lh, const E& z) {
    std::array<uint64_t, 6> key{(uint64_t)lh, 0, 0, 0, 0, 0};
    for (int k = 0; k < DC; ++k) key[1 + k] = z.c[k].v;
    return key;
  };
  std::vector<DevBuf> inv_keep;
  {
    std::vector<FriInvJobT<DC>> inv_jobs;
    uint32_t inv_blocks = 0;
    for (auto& it : items) {
      const int lh = it.log_h + log_blowup;
      for (size_t p = 0; p < it.z.size(); ++p) {
        const auto key = inv_key(lh, it.z[p]);
        if (inv_cache.count(key)) continue;
        inv_keep.emplace_back((size_t)DC << lh);
        FriInvJobT<DC> j{};
        j.inv = inv_keep.back().p;
        j.h = uint64_t(1) << lh;
        j.log_h = lh;
        if (lh < 2) fail(P3R_EINVAL, "FRI: an LDE of fewer than four rows");
        j.w_h = F::two_adic_generator(lh).v;
        j.w_4 = F::two_adic_generator(2).v;
        j.z = to_e4<PP, DC>(it.z[p]);
        j.block0 = inv_blocks;
        inv_blocks += blocks_for((size_t(1) << lh) / 4);
        inv_jobs.push_back(j);
        inv_cache.emplace(key, j.inv);
      }
    }
    inv_keep.emplace_back((inv_jobs.size() * sizeof(FriInvJobT<DC>) + 3) / 4);
    P3R_HIP(ctx->stage.upload(ctx->stream, inv_keep.back().p, inv_jobs.data(), inv_jobs.size() * sizeof(FriInvJobT<DC>)));
    ProfScope ps(ctx, "fri_inv_points");
    hipLaunchKernelGGL((k_fri_inv_points<PP, DC>), dim3(inv_blocks), dim3(kBlock), 0, ctx->stream,
                       reinterpret_cast<const FriInvJobT<DC>*>(inv_keep.back().p), (int)inv_jobs.size(), gen.v);
    P3R_HIP(hipGetLastError());
  }
  for (auto& it : items)
    for (auto& pv : it.vals)
      for (auto& v : pv) ch.observe_ext(v);
  // Self-check before anything is serialised: the opened values must satisfy the verifier's
  // out-of-domain identity, folded constraints(zeta) / Z_H(zeta) == quotient(zeta), per instance
  // (the same routine the native verifier runs).  Traces that violate a constraint or unbalance a
  // lookup fail it with overwhelming probability: P3R_EINVAL instead of an unverifiable proof -
  // the counterpart of prove_batch's internal constraint check (include/p3r.h).  Host work of a few
  // extension-field evaluations per table, done while the device computes the 1/(z - x) vectors.
  {
    E l_beta_pow[kMaxExtD + 1];
    for (int j = 0; j <= kMaxExtD; ++j) l_beta_pow[j] = e4_load<PP, DC>(lc.beta_pow[j]);
    const E l_prefix = e4_load<PP, DC>(lc.prefix);
    std::vector<std::vector<std::vector<E>>> inst_chunks(ni);
    for (size_t k = 0; k < chunks.size(); ++k) inst_chunks[chunks[k].inst].push_back(o_chunks[k]);
    static const std::vector<E> empty;
    for (size_t i = 0; i < ni; ++i) {
      const bool lk = layouts[i].n_groups > 0;
      ZetaInstance<PP, DC> zi{&o_main[i][0], o_main[i].size() > 1 ? &o_main[i][1] : nullptr, &o_prep[i][0], &o_prep[i][1],
                          lk ? &o_perm[i][0] : &empty, lk ? &o_perm[i][1] : &empty, &inst_chunks[i],
                          lk ? &terminals[i] : nullptr};
      try {
        check_instance_at_zeta<PP, DC>(prep->airs[i], layouts[i], log_n[i], zi, alpha, zeta, l_prefix, l_beta_pow,
                                   ctx->rc_mont_host.data(), i, zk);
      } catch (const VerifyFailure& e) {
        fail(P3R_EINVAL, "the traces do not satisfy the constraints (%s)", e.what());
      }
    }
    E tsum = E::zero();
    for (int i : perm_insts) tsum += terminals[i];
    if (!tsum.is_zero()) fail(P3R_EINVAL, "the traces do not satisfy the constraints (global lookup sum is not zero)");
  }

  prof_stage(ctx, "fri_reduce");
  // ---- 6. FRI batching challenge and per-height reduced openings
  const E fri_alpha = ch.sample_ext();
  size_t max_w = 1;
  for (auto& it : items) max_w = std::max(max_w, it.lde->w);
  std::vector<E> fa_pow(max_w + 1);
  fa_pow[0] = E::one();
  for (size_t c = 1; c <= max_w; ++c) fa_pow[c] = fa_pow[c - 1] * fri_alpha;
  DevBuf d_fapow(max_w * DC);
  {
    std::vector<uint32_t> h(max_w * DC);
    for (size_t c = 0; c < max_w; ++c)
      for (int k = 0; k < DC; ++k) h[c * DC + k] = fa_pow[c].c[k].v;
    P3R_HIP(ctx->stage.upload(ctx->stream, d_fapow.p, h.data(), h.size() * 4));
  }
  // One pass per height over all its matrices (kernels_fri_reduce.hip.h); alpha powers restart per
  // height and run on across that height's matrices and points in `items` order.
  std::map<int, std::pair<E, DevBuf>> ros;  // log_height -> (alpha power, ro planes [DC][h])
  {
    std::map<int, std::vector<FriReduceMatT<DC>>> by_height;
    std::vector<DevBuf> keep;
    std::vector<FriVsumJob> vsum_jobs;
    DevBuf vsums(2 * DC * items.size());
    for (auto& it : items) {
      const int lh = it.log_h + log_blowup;
      auto f = ros.find(lh);
      if (f == ros.end()) f = ros.emplace(lh, std::make_pair(E::one(), DevBuf((size_t)DC << lh))).first;
      FriReduceMatT<DC> a{};
      a.mat = it.lde->d;
      a.w = (int)it.lde->w;
      a.n_points = (int)it.z.size();
      E ap = f->second.first;
      for (size_t p = 0; p < it.z.size(); ++p) {
        a.inv[p] = inv_cache.at(inv_key(lh, it.z[p]));
        // V = sum_c alpha^c * opened value c, formed on the device (k_fri_vsum)
        a.v[p] = vsums.p + DC * vsum_jobs.size();
        vsum_jobs.push_back({op.values_dev(it.job, (int)p), vsums.p + DC * vsum_jobs.size(), (int)it.vals[p].size()});
        a.off[p] = to_e4<PP, DC>(ap);
        ap *= fa_pow[it.lde->w];
      }
      f->second.first = ap;
      by_height[lh].push_back(a);
    }
    std::vector<FriReduceMatT<DC>> mats;
    std::vector<FriReduceJob> jobs;
    uint32_t blocks = 0;
    for (auto& kv : by_height) {
      FriReduceJob j{};
      j.ro = ros[kv.first].second.p;
      j.h = uint64_t(1) << kv.first;
      j.mat0 = (uint32_t)mats.size();
      j.n_mats = (uint32_t)kv.second.size();
      j.block0 = blocks;
      blocks += blocks_for(size_t(1) << kv.first);
      jobs.push_back(j);
      mats.insert(mats.end(), kv.second.begin(), kv.second.end());
    }
    // the three job lists travel in one transfer
    const size_t b_mats = mats.size() * sizeof(FriReduceMatT<DC>), b_jobs = jobs.size() * sizeof(FriReduceJob),
                 b_vsum = vsum_jobs.size() * sizeof(FriVsumJob);
    const size_t o_jobs = (b_mats + 15) & ~size_t(15), o_vsum = (o_jobs + b_jobs + 15) & ~size_t(15);
    std::vector<unsigned char> blob(o_vsum + b_vsum);
    std::memcpy(blob.data(), mats.data(), b_mats);
    std::memcpy(blob.data() + o_jobs, jobs.data(), b_jobs);
    std::memcpy(blob.data() + o_vsum, vsum_jobs.data(), b_vsum);
    keep.emplace_back((blob.size() + 3) / 4);
    unsigned char* d_blob = reinterpret_cast<unsigned char*>(keep.back().p);
    P3R_HIP(ctx->stage.upload(ctx->stream, d_blob, blob.data(), blob.size()));
    const auto* d_mats = reinterpret_cast<const FriReduceMatT<DC>*>(d_blob);
    const auto* d_jobs = reinterpret_cast<const FriReduceJob*>(d_blob + o_jobs);
    const auto* d_vsum = reinterpret_cast<const FriVsumJob*>(d_blob + o_vsum);
    {
      ProfScope ps(ctx, "fri_reduce");
      hipLaunchKernelGGL((k_fri_vsum<PP, DC>), dim3((unsigned)vsum_jobs.size()), dim3(kBlock), 0, ctx->stream, d_vsum,
                         d_fapow.p);
      hipLaunchKernelGGL((k_fri_reduce_pre<PP, DC>), dim3(blocks), dim3(kBlock), 0, ctx->stream, d_jobs, (int)jobs.size(),
                         d_mats, d_fapow.p);
    }
    P3R_HIP(hipGetLastError());
  }

  prof_stage(ctx, "fri_commit_phase");
  // ---- 7. FRI commit phase
  std::vector<int> heights;
  for (auto& kv : ros) heights.push_back(kv.first);
  std::sort(heights.rbegin(), heights.rend());
  const int log_max = heights[0];
  const int log_final = (int)cfg.log_final_poly_len + log_blowup;
  struct Phase { int la; size_t rows; DevBuf folded_in; std::unique_ptr<p3r_tree> tree; std::vector<uint32_t> cap; };
  std::vector<Phase> phases;
  DevBuf folded = std::move(ros[log_max].second);
  size_t next_h = 1;
  int log_cur = log_max;
  std::vector<F> commit_pow_witnesses;
  // The folding challenges stay on the device.  With a one-digest cap and no commit-phase proof of
  // work the transcript steps between the phases run there too (k_fri_transcript_step), so the whole
  // commit phase is enqueued without a host round trip; the host replays them afterwards.
  const bool device_transcript = cfg.cap_height == 0 && cfg.commit_pow_bits == 0 && ch.in_buf.empty();
  constexpr size_t kMaxPhases = 32;
  DevBuf d_tstate(P2_WIDTH), d_phase((DC + P2_DIGEST) * kMaxPhases);  // challenges, then roots
  uint32_t* const d_betas = d_phase.p;
  uint32_t* const d_caps = d_phase.p + DC * kMaxPhases;
  if (device_transcript) {
    uint32_t st[P2_WIDTH];
    for (int k = 0; k < P2_WIDTH; ++k) st[k] = ch.state[k].v;
    P3R_HIP(ctx->stage.upload(ctx->stream, d_tstate.p, st, sizeof st));
  }
  while (log_cur > log_final) {
    int log_next = next_h < heights.size() ? heights[next_h] : -1;
    int la = fri_log_arity(ctx->fri_log_arities, phases.size(), (int)cfg.max_log_arity, log_cur, log_final, log_next);
    if (la < 0) fail(P3R_EINVAL, "fri_log_arities does not fit the proof: phase %zu at height 2^%d", phases.size(), log_cur);
    if (la > 3) fail(P3R_EUNSUPPORTED, "max_log_arity > 3 is not supported");
    const size_t arity = size_t(1) << la, rows = (size_t(1) << log_cur) >> la, n_in = size_t(1) << log_cur;
    Phase ph;
    ph.la = la;
    ph.rows = rows;
    // leaves: row r = the 2^la sibling evaluations, EF flattened -> column (j*DC+k) = plane k, offset j, stride arity
    ph.tree = std::make_unique<p3r_tree>();
    ph.tree->cap_height = (int)cfg.cap_height;
    ph.tree->log_max_h = log_cur - la;
    if (ph.tree->cap_height > ph.tree->log_max_h)
      fail(P3R_EINVAL, "cap_height %d exceeds the height of FRI commit phase %zu (2^%d rows)", ph.tree->cap_height,
           phases.size(), ph.tree->log_max_h);
    const bool arity4 = cfg.mmcs_arity == 4;
    const size_t n_leaf = arity4 ? mmcs4_padded_len(rows) : rows;
    ph.tree->layers.emplace_back(P2_DIGEST * n_leaf);
    {
      std::vector<const uint32_t*> cols;
      for (size_t j = 0; j < arity; ++j)
        for (int k = 0; k < DC; ++k) cols.push_back(folded.p + (size_t)k * n_in + j);
      if (cfg.mmcs_salt_elems) {
        // ExtensionMmcs over a hiding MMCS: the flattened row, then its salt (recursion/src/pcs/mmcs.rs:470-486).  The salt
        // columns take the leaf kernels' strided layout: column c of row r at [c * n_in + r * arity]
        const uint32_t S = cfg.mmcs_salt_elems;
        p3r_tree& T = *ph.tree;
        T.salt_elems = (int)S; T.phase_salt_stride = arity; T.phase_rows = rows;
        T.phase_salts.alloc((size_t)S * n_in);
        ZkSaltJob j{};
        j.dst = T.phase_salts.p; j.h = rows; j.S = S; j.stride = (uint32_t)arity;
        j.stream = zk_stream_id(kSaltRoundFri, phases.size());
        j.block0 = 0;
        DevBuf d_job((sizeof(ZkSaltJob) + 3) / 4);
        if (!zk_cell_fill()) {
          std::vector<ZkTileJob> tj(1);
          tj[0].dst = j.dst; tj[0].rows = rows; tj[0].w2 = S; tj[0].mode = 1; tj[0].stride = (uint32_t)arity; tj[0].stream = j.stream;
          launch_zk_tiles<PP>(ctx, tj, zk_key, "mmcs_salts");
        } else {
          P3R_HIP(ctx->stage.upload(ctx->stream, d_job.p, &j, sizeof j));
          ProfScope ps(ctx, "mmcs_salts");
          hipLaunchKernelGGL(k_zk_salts<PP>, dim3((unsigned)(S * ((rows + kBlock - 1) / kBlock))), dim3(kBlock), 0, ctx->stream,
                             reinterpret_cast<const ZkSaltJob*>(d_job.p), 1, zk_key);
        }
        P3R_HIP(hipGetLastError());
        for (uint32_t c = 0; c < S; ++c) cols.push_back(T.phase_salts.p + (size_t)c * n_in);
      }
      const uint32_t* const* dcols = col_table(ctx, cols);
      if (arity4) {
        // ExtensionMmcs over the arity-4 MMCS: the same flattened rows under the width-32 sponge
        ph.tree->arity = 4;
        ph.tree->levels = mmcs4_schedule({rows});
        ph.tree->layer_n.assign(1, n_leaf);
        if (n_leaf != rows) P3R_HIP(fill_async(ctx->stream, ph.tree->layers[0].p, 0, P2_DIGEST * n_leaf * 4));
        mmcs4_hash_rows_strided<PP>(ctx, dcols, (int)cols.size(), rows, arity, ph.tree->layers[0].p, n_leaf);
      } else {
        ProfScope ps(ctx, "mmcs_hash_rows_strided");
        if (rows <= coop_max_leaf_rows())  // latency-bound: sixteen lanes per row
          hipLaunchKernelGGL(k_mmcs_hash_rows_strided_coop<PP>, dim3(blocks_for(rows * 16)), dim3(kBlock), 0,
                             ctx->stream, dcols, (int)cols.size(), rows, arity, ph.tree->layers[0].p, ctx->rc.p,
                             ctx->p2_diag.p);
        else
          hipLaunchKernelGGL(k_mmcs_hash_rows_strided<PP>, dim3(blocks_for(rows)), dim3(kBlock), 0, ctx->stream,
                             dcols, (int)cols.size(), rows, arity, ph.tree->layers[0].p, ctx->rcd());
      }
      P3R_HIP(hipGetLastError());
    }
    const size_t pi = phases.size();
    if (pi >= kMaxPhases) fail(P3R_EUNSUPPORTED, "more than %zu FRI commit phases", kMaxPhases);
    TranscriptStep step{d_tstate.p, d_betas + DC * pi, d_caps + P2_DIGEST * pi};
    step.dc = DC;
    if (arity4) mmcs4_build_levels<PP>(ctx, ph.tree.get(), nullptr);
    else build_plain_layers<PP>(ctx, ph.tree.get(), rows, device_transcript ? &step : nullptr);
    if (device_transcript) {
      if (!step.done)  // a tree whose root is not produced by a single-workgroup launch (one leaf)
        hipLaunchKernelGGL(k_fri_transcript_step<PP>, dim3(1), dim3(64), 0, ctx->stream, ph.tree->layers.back().p,
                           step.state, step.beta, step.cap, ctx->rc.p, ctx->p2_diag.p, DC);
    } else {
      ph.cap = download_cap_mont<PP>(ctx, ph.tree.get());
      for (uint32_t v : ph.cap) ch.observe(F::raw(v));
      commit_pow_witnesses.push_back(grind_witness<PP>(ctx, ch, (int)cfg.commit_pow_bits));
      const E beta = ch.sample_ext();
      uint32_t bw[DC];
      for (int k = 0; k < DC; ++k) bw[k] = beta.c[k].v;
      P3R_HIP(ctx->stage.upload(ctx->stream, d_betas + DC * pi, bw, sizeof bw));
    }
    DevBuf out(DC * rows);
    FriFoldArgs fa{};
    fa.in = folded.p; fa.out = out.p; fa.rows = rows; fa.la = la; fa.log_rows = log_cur - la;
    fa.beta = d_betas + DC * pi;
    const bool roll = next_h < heights.size() && heights[next_h] == log_cur - la;
    fa.roll = roll ? ros[heights[next_h]].second.p : nullptr;
    fa.w_inv = F::two_adic_generator(log_cur).inv().v;
    const F omega = F::two_adic_generator(la);
    for (int s = 0; s < la; ++s) {
      F om_s = omega.pow(uint64_t(1) << s);
      for (int j = 0; j < (int)(arity >> (s + 1)); ++j)
        fa.tw_inv[s][j] = om_s.pow(bit_reverse(2 * j, la - s)).inv().v;
    }
    fa.neg_half = (-(F::from_canonical(2).inv())).v;
    {
      ProfScope ps(ctx, "fri_fold");
      hipLaunchKernelGGL((k_fri_fold<PP, DC>), dim3(blocks_for(rows)), dim3(kBlock), 0, ctx->stream, fa);
    }
    P3R_HIP(hipGetLastError());
    if (roll) ++next_h;
    ph.folded_in = std::move(folded);
    folded = std::move(out);
    log_cur -= la;
    phases.push_back(std::move(ph));
  }
  if (next_h != heights.size()) fail(P3R_EINVAL, "FRI: an input height was never rolled in");
  // final polynomial (host: <= 2^(log_final) extension elements)
  std::vector<E> final_poly;
  {
    const size_t m = size_t(1) << log_cur;
    std::vector<uint32_t> raw(DC * m), phase_words(d_phase.n);
    if (device_transcript && !phases.empty()) P3R_HIP(fetch_small(ctx, d_phase.p, phase_words.size(), phase_words.data()));
    P3R_HIP(fetch_small(ctx, folded.p, raw.size(), raw.data()));
    const uint32_t* betas = phase_words.data();
    const uint32_t* caps = betas + DC * kMaxPhases;
    if (device_transcript) {
      // replay the commit phase on the host transcript; the challenges must be the device's
      for (size_t pi = 0; pi < phases.size(); ++pi) {
        phases[pi].cap.assign(caps + P2_DIGEST * pi, caps + P2_DIGEST * (pi + 1));
        for (uint32_t v : phases[pi].cap) ch.observe(F::raw(v));
        commit_pow_witnesses.push_back(grind_witness<PP>(ctx, ch, 0));
        const E beta = ch.sample_ext();
        for (int k = 0; k < DC; ++k)
          if (beta.c[k].v != betas[DC * pi + k]) fail(P3R_EHIP, "internal: device and host FRI transcripts disagree");
      }
    }
    // inverse DFT, decimation in time: the rows are already in bit-reversed order, the
    // coefficients come out in natural order
    std::vector<E> coeffs(m);
    for (size_t i = 0; i < m; ++i)
      for (int k = 0; k < DC; ++k) coeffs[i].c[k] = F::raw(raw[(size_t)k * m + i]);
    const F w_inv = F::two_adic_generator(log_cur).inv(), m_inv = F::from_u64(m).inv();
    for (size_t len = 2; len <= m; len <<= 1) {
      const F w_len = w_inv.pow(m / len);
      for (size_t i = 0; i < m; i += len) {
        F x = F::one();
        for (size_t j = 0; j < len / 2; ++j) {
          const E u = coeffs[i + j], v = coeffs[i + j + len / 2] * x;
          coeffs[i + j] = u + v;
          coeffs[i + j + len / 2] = u - v;
          x *= w_len;
        }
      }
    }
    for (auto& c : coeffs) c = c * m_inv;
    const size_t flen = size_t(1) << cfg.log_final_poly_len;
    for (size_t i = flen; i < m; ++i)
      if (!coeffs[i].is_zero())
        fail(P3R_EINVAL, "FRI final polynomial has degree >= 2^%u: the traces do not satisfy the constraints",
             cfg.log_final_poly_len);
    final_poly.assign(coeffs.begin(), coeffs.begin() + flen);
  }
  for (auto& c : final_poly) ch.observe_ext(c);
  for (auto& ph : phases) ch.observe(F::from_canonical((uint32_t)ph.la));
  // query proof of work: smallest witness (device search)
  const F query_pow_witness = grind_witness<PP>(ctx, ch, (int)cfg.query_pow_bits);

  prof_stage(ctx, "queries");
  // ---- 8. queries: one gather launch for every opened row / sibling of every query
  std::vector<const p3r_tree*> round_trees;
  if (zk) round_trees.push_back(rand_tree.get());
  round_trees.push_back(main_tree.get()); round_trees.push_back(quot_tree.get()); round_trees.push_back(prep->tree.get());
  if (any_lookup) round_trees.push_back(perm_tree.get());
  const int n_rounds = (int)round_trees.size();
  std::vector<size_t> indices(cfg.num_queries);
  for (auto& ix : indices) ix = ch.sample_bits(log_max);
  // the item list is the same for every query (kernels_stark.hip.h, k_gather); offsets are relative
  // to a query's block of `words_per_query` words in the output
  std::vector<QueryItem> q_items;
  uint32_t cursor = 0;
  auto push = [&](const uint32_t* base, uint64_t stride, uint32_t count, uint32_t shift, uint32_t flip, uint32_t mul) {
    q_items.push_back({base, stride, count, cursor, shift, flip, mul});
    const uint32_t at = cursor;
    cursor += count;
    return at;
  };
  // an opening proof inside a query's block: binary tree - `depth` sibling digests in a row at proof_at; arity-4 tree
  // (mmcs4.h) - per level the siblings at positions pos ^ 1 .. pos ^ (step - 1), written out in ascending position
  struct QPath {
    uint32_t proof_at = 0;
    int depth = 0;
    std::vector<uint32_t> lv_at;
    const p3r_tree* t = nullptr;
  };
  struct QRound { std::vector<std::pair<uint32_t, uint32_t>> rows; QPath path; uint32_t tree_shift; };
  struct QPhase { uint32_t sib_at[8]; QPath path; int shift; uint32_t salt_at = 0; };
  // `base_shift`: the tree's index = query index >> base_shift
  auto push_path = [&](const p3r_tree* t, uint32_t base_shift) {
    QPath qp;
    qp.t = t;
    qp.proof_at = cursor;
    if (t->arity == 4) {
      for (size_t l = 0; l < t->levels.size(); ++l) {
        qp.lv_at.push_back(cursor);
        for (int f = 1; f < t->levels[l].step; ++f)
          push(t->layers[l].p, t->layer_n[l], P2_DIGEST, base_shift + (uint32_t)t->levels[l].bits, (uint32_t)f, 1);
      }
      qp.depth = (int)mmcs4_proof_len(t->levels);
    } else {
      qp.depth = t->log_max_h - t->cap_height;
      for (int l = 0; l < qp.depth; ++l)
        push(t->layers[l].p, size_t(1) << (t->log_max_h - l), P2_DIGEST, base_shift + l, 1, 1);
    }
    return qp;
  };
  std::vector<QRound> qrounds;
  std::vector<QPhase> qphases;
  for (int r = 0; r < n_rounds; ++r) {
    const p3r_tree* t = round_trees[r];
    const uint32_t tree_shift = (uint32_t)(log_max - t->log_max_h);  // tree index = query index >> tree_shift
    QRound qr;
    for (const p3r_dmat* m : t->mats) {
      const int lh = log2_exact(m->h, "height");
      qr.rows.emplace_back(push(m->d, m->h, (uint32_t)m->w, (uint32_t)(log_max - lh), 0, 1), (uint32_t)m->w);
    }
    qr.tree_shift = tree_shift;
    qr.path = push_path(t, tree_shift);
    qrounds.push_back(std::move(qr));
  }
  {
    int shift = 0;  // the phase's index = query index >> shift
    for (auto& ph : phases) {
      const size_t arity = size_t(1) << ph.la, n_in = ph.rows << ph.la;
      QPhase qp;
      qp.shift = shift;
      // sibling j of the row: the DC planes of one extension element, at row * arity + j
      for (size_t j = 0; j < arity; ++j)
        qp.sib_at[j] = push(ph.folded_in.p + j, n_in, DC, (uint32_t)(shift + ph.la), 0, (uint32_t)arity);
      if (ph.tree->salt_elems)   // the salt of the opened row (strided layout: fri commit phase above)
        qp.salt_at = push(ph.tree->phase_salts.p, n_in, (uint32_t)ph.tree->salt_elems, (uint32_t)(shift + ph.la), 0, (uint32_t)arity);
      qp.path = push_path(ph.tree.get(), (uint32_t)(shift + ph.la));
      qphases.push_back(std::move(qp));
      shift += ph.la;
    }
  }
  const uint32_t words_per_query = cursor;
  const uint32_t* gathered = nullptr;  // in the ctx's pinned landing area, read in place below
  {
    const auto* d_items = static_cast<const QueryItem*>(const_table(ctx, q_items.data(), q_items.size() * sizeof(QueryItem)));
    std::vector<uint32_t> idx32(indices.begin(), indices.end());
    DevBuf d_idx(idx32.size()), d_out((size_t)words_per_query * indices.size());
    hipError_t e = ctx->stage.upload(ctx->stream, d_idx.p, idx32.data(), idx32.size() * 4);
    if (e == hipSuccess) {
      ProfScope ps(ctx, "query_gather");
      hipLaunchKernelGGL(k_gather<PP>, dim3((unsigned)q_items.size(), (unsigned)indices.size()), dim3(64), 0, ctx->stream,
                         d_items, d_idx.p, words_per_query, d_out.p);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = ctx->landing.fetch(ctx->stream, d_out.p, d_out.n * 4, &gathered);
    P3R_HIP(e);
  }

  prof_stage(ctx, "serialize");
  // ---- 9. serialise BatchProof; the order of the fields of each struct is ctx->proof_layout
  // (host_transcript.h::ProofLayout, identity by default)
  const ProofLayout& PL = ctx->proof_layout;
  auto write_commitments = [&] {
    W.cap_mont(main_cap);
    if (any_lookup) { W.byte(1); W.cap_mont(perm_cap); } else W.byte(0);
    W.cap_mont(quot_cap);
    if (zk) { W.byte(1); W.cap_mont(rand_cap); } else W.byte(0);  // random commitment: Some iff Pcs::ZK (batch_stark.rs:424-428)
  };
  auto write_opened = [&] {
    W.varint(ni);
    size_t ck = 0;
    for (size_t i = 0; i < ni; ++i) {
      const size_t C = size_t(1) << (layouts[i].log_chunks + zk);
      for (int f = 0; f < 8; ++f) {
        switch (PL.opened[f]) {
          case 0: W.vec_ef(o_main[i][0]); break;
          case 1: if (o_main[i].size() == 2) { W.byte(1); W.vec_ef(o_main[i][1]); } else W.byte(0); break;
          case 2: W.byte(1); W.vec_ef(o_prep[i][0]); break;
          case 3: W.byte(1); W.vec_ef(o_prep[i][1]); break;
          case 4:
            W.varint(C);
            for (size_t c = 0; c < C; ++c) W.vec_ef(o_chunks[ck++]);
            break;
          case 5: if (zk) { W.byte(1); W.vec_ef(o_rand[i]); } else W.byte(0); break;  // random: Option<Vec<Challenge>>
          case 6: if (!o_perm[i].empty()) W.vec_ef(o_perm[i][0]); else W.varint(0); break;
          default: if (!o_perm[i].empty()) W.vec_ef(o_perm[i][1]); else W.varint(0); break;
        }
      }
    }
  };
  auto write_path = [&](const QPath& qp, const uint32_t* g, size_t tree_index) {
    W.varint(qp.depth);
    if (qp.t->arity != 4) {
      W.words(g + qp.proof_at, (size_t)qp.depth * P2_DIGEST);
      return;
    }
    for (size_t l = 0; l < qp.t->levels.size(); ++l) {
      const size_t step = qp.t->levels[l].step, pos = (tree_index >> qp.t->levels[l].bits) & (step - 1);
      for (size_t j = 0; j < step; ++j)
        if (j != pos) W.words(g + qp.lv_at[l] + ((j ^ pos) - 1) * P2_DIGEST, P2_DIGEST);
    }
  };
  auto write_queries = [&] {
    W.varint(indices.size());
    for (size_t qi = 0; qi < indices.size(); ++qi) {
      const uint32_t* g = gathered + (size_t)qi * words_per_query;  // this query's answers
      W.varint(qrounds.size());
      for (auto& qr : qrounds) {
        // a hiding MMCS's tree lists [M0, S0, M1, S1, ..]: opened values are the rows of the even entries, and the opening
        // proof is the tuple (salts = rows of the odd entries, siblings) (SaltedMmcsProof, mmcs.rs:763-768)
        const size_t step = qr.path.t->salt_elems ? 2 : 1;
        W.varint(qr.rows.size() / step);
        for (size_t k = 0; k < qr.rows.size(); k += step) {
          W.varint(qr.rows[k].second);
          W.words(g + qr.rows[k].first, qr.rows[k].second);
        }
        if (step == 2) {
          W.varint(qr.rows.size() / 2);
          for (size_t k = 1; k < qr.rows.size(); k += 2) {
            W.varint(qr.rows[k].second);
            W.words(g + qr.rows[k].first, qr.rows[k].second);
          }
        }
        write_path(qr.path, g, indices[qi] >> qr.tree_shift);
      }
      W.varint(qphases.size());
      for (size_t p = 0; p < phases.size(); ++p) {
        auto& qp = qphases[p];
        const size_t arity = size_t(1) << phases[p].la;
        const size_t pos = (indices[qi] >> qp.shift) & (arity - 1);  // the query's own position in the row
        W.byte((uint8_t)phases[p].la);
        W.varint(arity - 1);
        for (size_t j = 0; j < arity; ++j)
          if (j != pos) W.words(g + qp.sib_at[j], DC);
        if (qp.path.t->salt_elems) {   // (salts of the phase's one matrix, siblings)
          W.varint(1);
          W.varint((size_t)qp.path.t->salt_elems);
          W.words(g + qp.salt_at, (size_t)qp.path.t->salt_elems);
        }
        write_path(qp.path, g, indices[qi] >> (qp.shift + phases[p].la));
      }
    }
  };
  auto write_fri = [&] {
    if (zk) {
      // HidingFriPcs::Proof = (OpenedValues<Challenge>, FriProof): rounds -> matrices -> points -> the R codeword values
      // (pcs/fri/targets.rs:956-1005); a tuple has no framing of its own
      W.varint((size_t)n_rounds);
      size_t at = 0;
      for (int r = 0; r < n_rounds; ++r) {
        size_t n_m = 0;
        while (at + n_m < items.size() && items[at + n_m].round == r) ++n_m;
        W.varint(n_m);
        for (size_t m = 0; m < n_m; ++m) {
          const Item& it = items[at + m];
          W.varint(it.vals.size());
          for (auto& pv : it.vals) W.vec_ef(std::vector<E>(pv.end() - R, pv.end()));
        }
        at += n_m;
      }
    }
    for (int f = 0; f < 5; ++f) {
      switch (PL.fri[f]) {
        case 0:
          W.varint(phases.size());
          for (auto& ph : phases) W.cap_mont(ph.cap);
          break;
        case 1:
          W.varint(commit_pow_witnesses.size());
          for (auto& w : commit_pow_witnesses) W.fe(w);
          break;
        case 2: write_queries(); break;
        case 3: W.vec_ef(final_poly); break;
        default: W.fe(query_pow_witness); break;
      }
    }
  };
  for (int f = 0; f < 5; ++f) {
    switch (PL.batch[f]) {
      case 0: write_commitments(); break;
      case 1: write_opened(); break;
      case 2: write_fri(); break;
      case 3:
        W.varint(ni);
        for (size_t i = 0; i < ni; ++i) {
          if (layouts[i].n_groups) { W.byte(1); W.ef(terminals[i]); } else W.byte(0);
        }
        break;
      default:
        W.varint(ni);
        for (size_t i = 0; i < ni; ++i) W.varint(log_e[i]);   // ZK: the extended degree bits
        break;
    }
  }
  P3R_HIP(hipStreamSynchronize(ctx->stream));
  prof_stage(ctx, nullptr);
  host_timeline_dump();
  return W.take();
}

// prove_batch over the context's challenge field (p3r_config.challenge_degree)
template <class PP>
std::vector<uint8_t> prove_batch_any(p3r_ctx* ctx, const p3r_prep* prep, const p3r_dmat* const* mains, size_t ni,
                                     bool canonical_encoding) {
  if (ctx->cfg.challenge_degree == 5) {
    if constexpr (kHasQuintic<PP>) return prove_batch<PP, 5>(ctx, prep, mains, ni, canonical_encoding);
    else fail(P3R_EUNSUPPORTED, "UnsupportedChallengeDegree: the quintic challenge field is KoalaBear's");
  }
  return prove_batch<PP, 4>(ctx, prep, mains, ni, canonical_encoding);
}

}  // namespace
