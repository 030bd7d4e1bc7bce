// prove_all_tables on the GPU (included into p3r_core.hip): Traces -> per-table matrices
// (K1, K2, K3) -> prove_batch, plus the cached per-circuit-shape setup (K4).
//
//   BatchStarkProver::prove_all_tables / prove     circuit-prover/src/batch_stark_prover.rs:1203-1222,1275-1642
//   build_next_layer_prep / NextLayerPrepCache      recursion/src/recursion.rs:295-298,342-394
//   get_airs_and_degrees_with_prep                  circuit-prover/src/common.rs:127-390
//   AluAir::compute_schedule / trace_to_matrix /
//          build_scheduled_preprocessed_trace       circuit-prover/src/air/alu_air.rs:349-463,497-608,613-677
//   ConstAir / WitnessSendAir / RecomposeAir trace_to_matrix
//                                                   const_air.rs:88-127, public_air.rs:127-169, recompose_air.rs:96-119
//   Poseidon2 preprocessed rows + padding           poseidon2-circuit-air/src/air.rs:613-649,697-794
//   Poseidon2 filler rows                           circuit-prover/src/batch_stark_prover/poseidon2.rs:1125-1140
//
// The ALU lane schedule depends only on the preprocessed op list, so it is computed once on
// the host at setup and kept in HBM as a scatter plan; building the ALU main trace from the
// per-proof values is then one fully parallel kernel (a packed-Horner row's previous lane-0
// output is itself a trace value, so there is no sequential dependency between rows).

namespace {

// (AluPlanEntry / PLAN_*: run_schedule.h)

// K1: one lane per trace row.  D = the circuit's extension degree; elements are VD<F, D> and products go through
// mulD (the same rule the AIR states: Fp1 / x^4 = W / quintic trinomial / generic binomial with W = w_mont).
template <class PP, int D>
struct AluElem {
  Fp<PP> c[D];
  static __device__ __forceinline__ AluElem zero() { AluElem r; for (int i = 0; i < D; ++i) r.c[i] = Fp<PP>::zero(); return r; }
};
template <class PP, int D>
__global__ void __launch_bounds__(kBlock)
k_alu_trace(const AluPlanEntry* __restrict__ plan, const uint32_t* __restrict__ prev_src /* per row */,
            const uint32_t* __restrict__ values /* [n_ops][4 D] Montgomery */, size_t rows, size_t h, int lanes,
            int k_max, uint32_t* __restrict__ out /* [width][h] */, uint32_t w_mont) {
  using F = Fp<PP>;
  using E = VD<F, D>;
  auto zero = [] { E r; for (int i = 0; i < D; ++i) r.c[i] = F::zero(); return r; };
  auto mul = [&](const E& x, const E& y) { return mulD<PP, D, F>(x, y, w_mont); };
  auto fma = [&](const E& x, const E& y, const E& p, const E& m) {   // x * y + p - m
    E r = mul(x, y);
    for (int i = 0; i < D; ++i) r.c[i] = r.c[i] + p.c[i] - m.c[i];
    return r;
  };
  size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= rows) return;
  auto val = [&](uint32_t op, int operand) {
    E e;
#pragma unroll
    for (int d = 0; d < D; ++d) e.c[d] = F::raw(values[(size_t)op * (4 * D) + operand * D + d]);
    return e;
  };
  auto put = [&](int col, const E& e) {
#pragma unroll
    for (int d = 0; d < D; ++d) out[(size_t)(col + d) * h + row] = e.c[d].v;
  };
  const int num_int = (k_max - 1) / 2;
  for (int lane = 0; lane < lanes; ++lane) {
    AluPlanEntry en = plan[row * lanes + lane];
    const int base = lane * 4 * D;
    if (en.kind == PLAN_OP) {
      for (int o = 0; o < 4; ++o) put(base + o * D, val(en.first, o));
    } else if (en.kind == PLAN_PACKED) {
      const int k = en.k;
      for (int o = 0; o < 3; ++o) put(base + o * D, val(en.first, o));
      put(base + 3 * D, val(en.first + k - 1, 3));
      if (lane == 0) {
        const int extra = lanes * 4 * D;
        const uint32_t ps = prev_src[row];
        E acc = ps == 0xFFFFFFFFu ? zero() : val(ps, 3);
        const E b = val(en.first, 1);
        int step = 0;
        for (int s = 0; s < num_int; ++s) {
          uint32_t i0 = en.first + step;
          if (step + 1 < k) {
            E o0 = fma(acc, b, val(i0, 2), val(i0, 0));
            acc = fma(o0, b, val(i0 + 1, 2), val(i0 + 1, 0));
            step += 2;
          } else {
            acc = fma(acc, b, val(i0, 2), val(i0, 0));
            step += 1;
          }
          put(extra + s * D, acc);
        }
        const int ac_base = extra + num_int * D;
        for (int t = 1; t < k; ++t) {
          put(ac_base + 2 * D * (t - 1), val(en.first + t, 0));
          put(ac_base + 2 * D * (t - 1) + D, val(en.first + t, 2));
        }
        put(ac_base + 2 * D * (k_max - 1), mul(b, b));
      }
    }
  }
}

// K2: flat per-op values (row-major, `w` cells per row) -> zero-padded column-major matrix.
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_flat_to_colmajor(const uint32_t* __restrict__ src, size_t n_flat, uint32_t* __restrict__ dst, size_t h, int w) {
  size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= h * (size_t)w) return;
  size_t c = t / h, r = t % h;
  size_t s = r * w + c;
  dst[t] = s < n_flat ? src[s] : 0u;
}

struct HostAluSchedule {
  std::vector<AluPlanEntry> entries;  // rows * lanes
  std::vector<uint32_t> prev_src;     // per row
  size_t rows = 0;
};

// AluAir::compute_schedule (alu_air.rs:349-463); prep13 canonical, 13 cells per op.
inline HostAluSchedule alu_schedule(const uint32_t* prep13, size_t n_ops, int lanes, int pack_k) {
  HostAluSchedule S;
  auto is_h = [&](size_t i) { return prep13[i * 13 + 4] == 1; };
  bool any = false;
  for (size_t i = 0; i < n_ops; ++i) any = any || is_h(i);
  std::vector<AluPlanEntry>& sched = S.entries;
  if (!any) {
    for (size_t i = 0; i < n_ops; ++i) sched.push_back({(uint32_t)i, PLAN_OP, 1, 0});
  } else {
    std::vector<std::vector<size_t>> chains;
    std::vector<size_t> cur, non_chain;
    for (size_t i = 0; i < n_ops; ++i) {
      if (is_h(i)) cur.push_back(i);
      else { if (!cur.empty()) { chains.push_back(cur); cur.clear(); } non_chain.push_back(i); }
    }
    if (!cur.empty()) chains.push_back(cur);
    size_t nc = 0;
    auto fill_row = [&]() {
      while (sched.size() % lanes != 0) {
        if (nc < non_chain.size()) sched.push_back({(uint32_t)non_chain[nc++], PLAN_OP, 1, 0});
        else sched.push_back({0, PLAN_SEP, 1, 0});
      }
    };
    sched.push_back({0, PLAN_SEP, 1, 0});
    fill_row();
    for (size_t ci = 0; ci < chains.size(); ++ci) {
      const auto& chain = chains[ci];
      if (ci > 0) { fill_row(); sched.push_back({0, PLAN_SEP, 1, 0}); fill_row(); }
      size_t i = 0;
      while (i < chain.size()) {
        size_t k_try = std::min<size_t>(chain.size() - i, (size_t)pack_k), best = 1;
        for (size_t k = k_try; k >= 2; --k) {
          bool ok = true;
          for (size_t j = 1; j < k && ok; ++j) ok = chain[i + j] == chain[i] + j;
          for (size_t j = 1; j < k && ok; ++j) ok = prep13[chain[i + j] * 13 + 6] == prep13[chain[i] * 13 + 6];
          if (ok) { best = k; break; }
        }
        if (best >= 2) { sched.push_back({(uint32_t)chain[i], PLAN_PACKED, (uint8_t)best, 0}); i += best; }
        else { sched.push_back({(uint32_t)chain[i], PLAN_OP, 1, 0}); i += 1; }
        fill_row();
      }
    }
    fill_row();
    while (nc < non_chain.size()) sched.push_back({(uint32_t)non_chain[nc++], PLAN_OP, 1, 0});
    fill_row();
  }
  while (sched.size() % lanes != 0) sched.push_back({0, PLAN_SEP, 1, 0});  // unscheduled tail: zero cells
  S.rows = std::max<size_t>(sched.size() / lanes, 1);
  sched.resize(S.rows * lanes, AluPlanEntry{0, PLAN_SEP, 1, 0});
  // previous lane-0 output feeding each row's packed-Horner accumulator (alu_air.rs:513-589)
  S.prev_src.assign(S.rows, 0xFFFFFFFFu);
  uint32_t prev = 0xFFFFFFFFu;
  for (size_t r = 0; r < S.rows; ++r) {
    S.prev_src[r] = prev;
    const AluPlanEntry& e = sched[r * lanes];
    if (!any) continue;  // unscheduled traces never use it
    if (e.kind == PLAN_OP) prev = e.first;
    else if (e.kind == PLAN_PACKED) prev = e.first + e.k - 1;
    else prev = 0xFFFFFFFFu;
  }
  return S;
}

inline size_t padded_height(size_t rows, size_t min_height) {
  size_t h = 1;
  while (h < std::max<size_t>(rows, 1)) h <<= 1;
  size_t mh = 1;
  while (mh < min_height) mh <<= 1;
  return std::max(h, mh);
}

}  // namespace

// Everything that depends only on the circuit shape (CircuitProverData + the ALU schedule).
struct p3r_layer {
  p3r_layer_desc_counts counts{};
  uint32_t public_lanes = 1, alu_lanes = 1, horner_k = 2, recompose_lanes = 1, min_height = 1;
  size_t h_const = 0, h_public = 0, h_alu = 0, h_p2 = 0, h_recompose = 0, alu_rows = 0;
  size_t h_recompose_coeff = 0;   // table 5: the `recompose/coeff` table of a layer that holds both kinds
  // A non-primitive table with no rows is not part of the batch (poseidon2.rs:1089-1092,
  // recompose.rs:77-80: `batch_instance_*` returns None); the primitive three always are.
  bool has_p2 = true, has_recompose = true, has_recompose_coeff = false;
  bool recompose_coeff = false;  // table 4 is the "recompose/coeff" variant: per-coefficient bus tuples (recompose_air.rs:196-226)
  // table 6: the width-32 Poseidon2 table (arity-4 MMCS rows), proved right after the width-16 one
  bool has_p2w = false;
  size_t h_p2w = 0;
  int slot_of(int table) const {  // position of table 0..6 among the proved instances, -1 if absent
    if (table < 3) return table;
    if (table == 3) return has_p2 ? 3 : -1;
    int base = has_p2 ? 4 : 3;
    if (table == 6) return has_p2w ? base : -1;
    if (has_p2w) base += 1;
    if (table == 4) return has_recompose ? base : -1;
    return has_recompose_coeff ? base + (has_recompose ? 1 : 0) : -1;
  }
  std::unique_ptr<p3r_prep> prep;
  p3r::DevBuf alu_plan, alu_prev_src;
};

// Per-proof inputs resident in HBM (flattened Traces<EF>).
struct p3r_dtraces {
  p3r::DevBuf const_values, public_values, alu_values, recompose_values;  // Montgomery, row-major
  size_t n_const = 0, n_public = 0, n_alu = 0, n_recompose = 0;
  size_t n_recompose_coeff = 0;   // rows of table 5: they follow the n_recompose rows of table 4 in recompose_values
  std::unique_ptr<p3r_p2_dev> p2;  // padded to the table height with filler rows
  std::unique_ptr<p3r_p2_dev> p2w; // rows of the width-32 table (flags: new_start | merkle_path | mmcs_bit | mmcs_bit2)
};
// tables in PROOF order: [Const, Public, Alu, Poseidon2, Poseidon2-W32, Recompose, Recompose/coeff]
constexpr int kTableOrder[7] = {0, 1, 2, 3, 6, 4, 5};

namespace {

template <class PP>
std::unique_ptr<p3r_layer> layer_create(p3r_ctx* ctx, const p3r_layer_desc* d, uint32_t* commit_out) {
  const uint32_t P = PP::P;
  auto L = std::make_unique<p3r_layer>();
  L->counts = d->counts;
  L->public_lanes = d->public_lanes; L->alu_lanes = d->alu_lanes; L->horner_k = d->horner_packed_steps;
  L->recompose_lanes = d->recompose_lanes; L->min_height = d->min_trace_height;
  if (!L->public_lanes || !L->alu_lanes || !L->recompose_lanes) fail(P3R_EINVAL, "lane counts must be positive");
  // reduce_lanes_if_dummy (batch_stark_prover.rs:1305-1318, common.rs:150-160): a Public / ALU table
  // that holds at most the dummy op is proved with one lane, and the proof records that packing.
  if (d->counts.n_public <= 1) L->public_lanes = 1;
  if (d->counts.n_alu <= 1) L->alu_lanes = 1;
  L->has_p2 = d->counts.n_p2 > 0;
  L->has_recompose = d->counts.n_recompose > 0;
  L->has_recompose_coeff = d->counts.n_recompose_coeff > 0;
  L->has_p2w = d->counts.n_p2w > 0;
  if (L->has_p2w && ctx->cfg.ext_degree != 4) fail(P3R_EUNSUPPORTED, "the width-32 Poseidon2 table belongs to D = 4 circuits");
  if (L->has_p2w && !d->p2w_prep) fail(P3R_EINVAL, "p2w_prep is NULL");
  if (L->has_recompose_coeff && d->recompose_coeff_lookups)
    fail(P3R_EINVAL, "recompose_coeff_lookups = 1 names the ONE Recompose table; with a second table (n_recompose_coeff) recompose_prep is the plain kind");
  if (L->horner_k < 2 || L->horner_k > 8) fail(P3R_EINVAL, "horner_packed_steps must be in 2..8");
  const uint32_t ext_d = ctx->cfg.ext_degree;
  if (ext_degree_is_binomial_generic(ext_d) && L->has_p2)
    fail(P3R_EUNSUPPORTED, "UnsupportedDegree(%u): no Poseidon2 table for this circuit degree (Const, Public, ALU, Recompose)", ext_d);
  L->recompose_coeff = d->recompose_coeff_lookups != 0;
  const int rec_plw = 2 + (L->recompose_coeff ? 2 * (int)ext_d : 0);
  const auto& c = d->counts;
  auto check = [&](const uint32_t* p, size_t n, const char* what) {
    if (n && !p) fail(P3R_EINVAL, "%s is NULL", what);
    host_parallel_for(n, size_t(1) << 20, [&](size_t i0, size_t i1) {
      for (size_t i = i0; i < i1; ++i)
        if (p[i] >= P) fail(P3R_EINVAL, "%s[%zu] is not canonical", what, i);
    });
  };
  prof_stage(ctx, "prep_lc_checks");
  check(d->alu_prep13, c.n_alu * 13, "alu_prep13");
  const size_t mh = L->min_height;
  auto lanes_prep = [&](const uint32_t* prep, size_t n_ops, int per_op, int lanes, size_t& h_out) {
    size_t rows = std::max<size_t>((n_ops + lanes - 1) / lanes, 1);
    h_out = padded_height(rows, mh);
    std::vector<uint32_t> m(h_out * (size_t)lanes * per_op, 0);
    std::copy(prep, prep + n_ops * per_op, m.begin());
    return m;
  };
  const int rec2_plw = 2 + 2 * (int)ext_d;
  std::vector<std::vector<uint32_t>> mats(7);
  std::unique_ptr<uint32_t, decltype(&free)> alu_mat(nullptr, &free);   // table 2 (the largest), filled in parallel
  p3r_air_desc airs[7] = {{P3R_AIR_CONST, 1, 2, 0},
                          {P3R_AIR_PUBLIC, L->public_lanes, 2, 0},
                          {P3R_AIR_ALU, L->alu_lanes, L->horner_k, 0},
                          {P3R_AIR_POSEIDON2, 1, 2, 0},
                          {P3R_AIR_RECOMPOSE, L->recompose_lanes, 2, L->recompose_coeff ? 1u : 0u},
                          {P3R_AIR_RECOMPOSE, L->recompose_lanes, 2, 1u},
                          {P3R_AIR_POSEIDON2_W32, 1, 2, 0}};
  // every table but the ALU one on a second host thread (they share nothing with it)
  auto other_tables = std::async(std::launch::async, [&] {
    check(d->const_prep, c.n_const * 2, "const_prep");
    check(d->public_prep, c.n_public * 2, "public_prep");
    check(d->recompose_prep, c.n_recompose * rec_plw, "recompose_prep");
    check(d->recompose_coeff_prep, c.n_recompose_coeff * rec2_plw, "recompose_coeff_prep");
    check(d->p2_out_ctl, c.n_p2 * (ext_d == 4 ? 2 : 8), "p2_out_ctl");
    mats[0] = lanes_prep(d->const_prep, c.n_const, 2, 1, L->h_const);
    mats[1] = lanes_prep(d->public_prep, c.n_public, 2, (int)L->public_lanes, L->h_public);
    // Poseidon2 preprocessed rows (air.rs:697-794, non-compact D=4 layout) + padding (:613-649)
    if (L->has_p2 && ext_d != 4) {
      // compact-D1 rows (air.rs:730-763; header executor.rs:720-741) + the same padding rule
      L->h_p2 = padded_height(c.n_p2, mh);
      std::vector<uint32_t>& m = mats[3];
      constexpr size_t W = kP2D1PrepWidth;
      m.assign(L->h_p2 * W, 0);
      auto scaled = [&](uint32_t wid) { return (uint32_t)((uint64_t)wid * ext_d % P); };
      for (size_t r = 0; r < c.n_p2; ++r) {
        uint32_t* o = &m[r * W];
        const bool ns = d->p2_new_start[r], mp = d->p2_merkle_path[r], en = d->p2_mmcs_ctl_enabled[r];
        const uint8_t* ctl = d->p2_in_ctl + r * 16;
        if (!mp)
          for (int l = 8; l < 16; ++l)
            if (ctl[l]) fail(P3R_EINVAL, "Poseidon2 row %zu: capacity input slots must be empty on compact D=1 sponge rows", r);
        for (int l = 0; l < 8; ++l) o[l] = ctl[l] != 0;
        o[8] = d->p2_absorb_len ? d->p2_absorb_len[r] : 0u;
        o[9] = !ns;
        for (int l = 0; l < 8; ++l) o[10 + l] = !ns && !mp && !ctl[l];
        for (int l = 0; l < 8; ++l) o[18 + l] = !ns && mp && !ctl[l];
        for (int l = 0; l < 16; ++l) o[kP2D1Hdr + l] = scaled(d->p2_input_indices[r * 16 + l]);
        for (int l = 0; l < 8; ++l) o[kP2D1Hdr + 16 + l] = scaled(d->p2_output_indices[r * 8 + l]);
        for (int l = 0; l < 8; ++l) o[kP2D1Hdr + 24 + l] = d->p2_out_ctl[r * 8 + l];
        o[kP2D1Tail] = scaled(d->p2_mmcs_index_sum_idx[r]);
        o[kP2D1Tail + 1] = en && mp;
        o[kP2D1Tail + 2] = ns;
        o[kP2D1Tail + 3] = mp;
      }
      if (L->h_p2 > c.n_p2) m[c.n_p2 * W + kP2D1Tail + 2] = 1;
    } else if (L->has_p2) {
      L->h_p2 = padded_height(c.n_p2, mh);
      std::vector<uint32_t>& m = mats[3];
      m.assign(L->h_p2 * 24, 0);
      auto scaled = [&](uint32_t wid) { return (uint32_t)((uint64_t)wid * 4 % P); };
      for (size_t r = 0; r < c.n_p2; ++r) {
        uint32_t* o = &m[r * 24];
        const bool ns = d->p2_new_start[r], mp = d->p2_merkle_path[r], en = d->p2_mmcs_ctl_enabled[r];
        for (int l = 0; l < 4; ++l) {
          const bool ctl = d->p2_in_ctl[r * 4 + l];
          o[l * 4] = scaled(d->p2_input_indices[r * 4 + l]);
          o[l * 4 + 1] = ctl;
          o[l * 4 + 2] = !ns && !mp && !ctl;
          o[l * 4 + 3] = !ns && mp && !ctl;
        }
        for (int l = 0; l < 2; ++l) {
          o[16 + 2 * l] = scaled(d->p2_output_indices[r * 2 + l]);
          o[17 + 2 * l] = d->p2_out_ctl[r * 2 + l];
        }
        o[20] = scaled(d->p2_mmcs_index_sum_idx[r]);
        o[21] = en && mp;
        o[22] = ns;
        o[23] = mp;
      }
      if (L->h_p2 > c.n_p2) m[c.n_p2 * 24 + 22] = 1;
    }
    if (L->has_recompose)
      mats[4] = lanes_prep(d->recompose_prep, c.n_recompose, rec_plw, (int)L->recompose_lanes, L->h_recompose);
    if (L->has_recompose_coeff)
      mats[5] = lanes_prep(d->recompose_coeff_prep, c.n_recompose_coeff, rec2_plw, (int)L->recompose_lanes, L->h_recompose_coeff);
    if (L->has_p2w) {
      // the caller's assembled rows + BaseAir::preprocessed_trace padding (air.rs:613-649): zero rows, the first one
      // with new_start = 1 at width - 2
      check(d->p2w_prep, c.n_p2w * (size_t)kP2WPrepWidth, "p2w_prep");
      L->h_p2w = padded_height(c.n_p2w, mh);
      mats[6].assign(L->h_p2w * (size_t)kP2WPrepWidth, 0);
      std::copy(d->p2w_prep, d->p2w_prep + c.n_p2w * (size_t)kP2WPrepWidth, mats[6].begin());
      if (L->h_p2w > c.n_p2w) mats[6][c.n_p2w * (size_t)kP2WPrepWidth + kP2WTail + 2] = 1;
    }
  });
  // ALU: schedule + scheduled preprocessed trace (alu_air.rs:613-677)
  {
    const int lanes = (int)L->alu_lanes, k_max = (int)L->horner_k, pw = lanes * 13 + 7 * (k_max - 1);
    prof_stage(ctx, "prep_lc_alu_schedule");
    HostAluSchedule S = alu_schedule(d->alu_prep13, c.n_alu, lanes, k_max);
    prof_stage(ctx, "prep_lc_alu_matrix");
    L->alu_rows = S.rows;
    L->h_alu = padded_height(S.rows, mh);
    // zero pages from calloc, first touched by the thread that fills them
    alu_mat.reset(static_cast<uint32_t*>(calloc(L->h_alu * (size_t)pw, sizeof(uint32_t))));
    if (!alu_mat) fail(P3R_ENOMEM, "host allocation of the ALU preprocessed matrix failed");
    uint32_t* m = alu_mat.get();
    auto mulmod = [&](uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)a * b % P); };
    host_parallel_for(S.entries.size(), size_t(1) << 15, [&](size_t pos0, size_t pos1) {
    for (size_t pos = pos0; pos < pos1; ++pos) {
      const auto& en = S.entries[pos];
      size_t row = pos / lanes, lane = pos % lanes, base = row * pw + lane * 13;
      if (en.kind == PLAN_OP) {
        std::copy(d->alu_prep13 + (size_t)en.first * 13, d->alu_prep13 + (size_t)en.first * 13 + 13, m + base);
      } else if (en.kind == PLAN_PACKED && lane == 0) {
        const int k = en.k;
        const uint32_t* src0 = d->alu_prep13 + (size_t)en.first * 13;
        const uint32_t* last = d->alu_prep13 + (size_t)(en.first + k - 1) * 13;
        std::copy(src0, src0 + 13, m + base);
        m[base + 8] = last[8];
        m[base + 10] = last[10];
        m[base + 9] = mulmod(m[base + 9], (uint32_t)k);
        const uint32_t mult_a = m[base];
        size_t extra = row * pw + (size_t)lanes * 13;
        m[extra + (k - 2)] = 1;
        for (int t = 1; t < k; ++t) {
          const uint32_t* st = d->alu_prep13 + (size_t)(en.first + t) * 13;
          size_t p = extra + (k_max - 1) + 6 * (t - 1);
          m[p] = st[5]; m[p + 1] = st[7]; m[p + 2] = st[11]; m[p + 3] = st[12];
          m[p + 4] = mulmod(mult_a, st[11]);
          m[p + 5] = mulmod(mult_a, st[12]);
        }
      }
    }
    });
    prof_stage(ctx, "prep_lc_plan_upload");
    L->alu_plan.alloc((S.entries.size() * sizeof(AluPlanEntry) + 3) / 4);
    P3R_HIP(copy_sync(ctx->stream, L->alu_plan.p, S.entries.data(), S.entries.size() * sizeof(AluPlanEntry), hipMemcpyHostToDevice));
    L->alu_prev_src.alloc(S.prev_src.size());
    P3R_HIP(copy_sync(ctx->stream, L->alu_prev_src.p, S.prev_src.data(), S.prev_src.size() * 4, hipMemcpyHostToDevice));
  }
  prof_stage(ctx, "prep_lc_other_tables_wait");
  other_tables.get();
  prof_stage(ctx, "prep_lc_upload_lde_commit");
  const int widths[7] = {2, (int)L->public_lanes * 2, (int)L->alu_lanes * 13 + 7 * ((int)L->horner_k - 1),
                         ext_d == 4 ? 24 : kP2D1PrepWidth, (int)L->recompose_lanes * rec_plw, (int)L->recompose_lanes * rec2_plw,
                         kP2WPrepWidth};
  const size_t heights[7] = {L->h_const, L->h_public, L->h_alu, L->h_p2, L->h_recompose, L->h_recompose_coeff, L->h_p2w};
  p3r_matrix pm[7];
  p3r_air_desc present_airs[7];
  size_t n_present = 0;
  for (int i : kTableOrder) {
    if (L->slot_of(i) < 0) continue;
    present_airs[n_present] = airs[i];
    pm[n_present++] = {i == 2 ? alu_mat.get() : mats[i].data(), heights[i], (size_t)widths[i]};
  }
  L->prep = prep_create<PP>(ctx, present_airs, pm, n_present);
  std::copy(L->prep->cap_canonical.begin(), L->prep->cap_canonical.end(), commit_out);
  return L;
}

// The same object from the device-side preparation (prep_device.hip): the preprocessed traces and the ALU scatter
// plan are already in HBM; what is left is their LDE and commitment.
template <class PP>
std::unique_ptr<p3r_layer> layer_from_device(p3r_ctx* ctx, DevPrep& R, const p3r_circuit_desc* d, uint32_t* commit_out) {
  auto L = std::make_unique<p3r_layer>();
  L->counts = R.counts;
  L->public_lanes = R.public_lanes; L->alu_lanes = R.alu_lanes; L->horner_k = d->horner_packed_steps;
  L->recompose_lanes = d->recompose_lanes; L->min_height = d->min_trace_height;
  if (!d->public_lanes || !d->alu_lanes || !d->recompose_lanes) fail(P3R_EINVAL, "lane counts must be positive");
  if (L->horner_k < 2 || L->horner_k > 8) fail(P3R_EINVAL, "horner_packed_steps must be in 2..8");
  L->has_p2 = R.counts.n_p2 > 0;
  L->has_recompose = R.counts.n_recompose > 0;
  L->has_recompose_coeff = R.counts.n_recompose_coeff > 0;
  L->recompose_coeff = R.recompose_coeff;
  L->has_p2w = R.counts.n_p2w > 0;
  L->h_p2w = R.h[6];
  L->h_const = R.h[0]; L->h_public = R.h[1]; L->h_alu = R.h[2]; L->h_p2 = R.h[3]; L->h_recompose = R.h[4];
  L->h_recompose_coeff = R.h[5];
  L->alu_rows = R.alu_rows;
  L->alu_plan = std::move(R.alu_plan);
  L->alu_prev_src = std::move(R.alu_prev_src);
  const p3r_air_desc airs[7] = {{P3R_AIR_CONST, 1, 2, 0},
                                {P3R_AIR_PUBLIC, L->public_lanes, 2, 0},
                                {P3R_AIR_ALU, L->alu_lanes, L->horner_k, 0},
                                {P3R_AIR_POSEIDON2, 1, 2, 0},
                                {P3R_AIR_RECOMPOSE, L->recompose_lanes, 2, L->recompose_coeff ? 1u : 0u},
                                {P3R_AIR_RECOMPOSE, L->recompose_lanes, 2, 1u},
                                {P3R_AIR_POSEIDON2_W32, 1, 2, 0}};
  p3r_air_desc present[7];
  std::vector<std::unique_ptr<p3r_dmat>> traces;
  size_t n = 0;
  for (int i : kTableOrder) {
    if (L->slot_of(i) < 0) continue;
    present[n++] = airs[i];
    traces.push_back(std::move(R.prep[i]));
  }
  L->prep = prep_create<PP>(ctx, present, nullptr, n, &traces);
  std::copy(L->prep->cap_canonical.begin(), L->prep->cap_canonical.end(), commit_out);
  return L;
}

template <class PP>
DevBuf upload_mont(p3r_ctx* ctx, const uint32_t* host, size_t n, const char* what) {
  DevBuf b(std::max<size_t>(n, 1));
  if (n) {
    if (!host) fail(P3R_EINVAL, "%s is NULL", what);
    P3R_HIP(hipMemcpyAsync(b.p, host, n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_convert_inplace<PP>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, b.p, n, 1);
    P3R_HIP(hipGetLastError());
    P3R_HIP(hipStreamSynchronize(ctx->stream));
  }
  return b;
}

// rows of the width-32 table on the device: h padded rows of which the first n carry data; flags = new_start[h] |
// merkle_path[h] | mmcs_bit[h] | mmcs_bit2[h]
template <class PP>
std::unique_ptr<p3r_p2_dev> p2w_rows_upload(p3r_ctx* ctx, size_t h, size_t n, const uint32_t* inputs /* h x 32 */, const uint8_t* flags /* 4 h */,
                                            const uint32_t* index_sum /* h */) {
  log2_exact(h, "width-32 Poseidon2 row count (padded)");
  for (size_t i = 0; i < n * 32; ++i) if (inputs[i] >= PP::P) fail(P3R_EINVAL, "p2w.input_values[%zu] is not canonical", i);
  for (size_t i = 0; i < n; ++i) if (index_sum[i] >= PP::P) fail(P3R_EINVAL, "p2w.mmcs_index_sum[%zu] is not canonical", i);
  auto dw = std::make_unique<p3r_p2_dev>();
  dw->n = h;
  dw->flags.alloc(h + 1);
  P3R_HIP(hipMemcpyAsync(dw->flags.p, flags, 4 * h, hipMemcpyHostToDevice, ctx->stream));
  dw->seed.alloc(h);
  P3R_HIP(hipMemcpyAsync(dw->seed.p, index_sum, h * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_convert_inplace<PP>, dim3(blocks_for(h)), dim3(kBlock), 0, ctx->stream, dw->seed.p, h, 1);
  P3R_HIP(hipGetLastError());
  dw->inputs = upload<PP>(ctx, inputs, h, P2W_WIDTH);  // syncs the stream
  return dw;
}

template <class PP>
std::unique_ptr<p3r_dtraces> traces_upload(p3r_ctx* ctx, const p3r_layer* L, const p3r_traces* t) {
  const auto& c = L->counts;
  if (t->n_const != c.n_const || t->n_public != c.n_public || t->n_alu != c.n_alu || t->p2.n != c.n_p2 ||
      t->n_recompose != c.n_recompose || t->n_recompose_coeff != c.n_recompose_coeff)
    fail(P3R_EINVAL, "trace row counts do not match the prepared circuit shape");
  auto d = std::make_unique<p3r_dtraces>();
  d->n_const = c.n_const; d->n_public = c.n_public; d->n_alu = c.n_alu; d->n_recompose = c.n_recompose;
  d->n_recompose_coeff = c.n_recompose_coeff;
  const size_t D = ctx->cfg.ext_degree;  // values are n x D (Const, Public), n x 4D (ALU: a, b, c, out)
  d->const_values = upload_mont<PP>(ctx, t->const_values, c.n_const * D, "const_values");
  d->public_values = upload_mont<PP>(ctx, t->public_values, c.n_public * D, "public_values");
  d->alu_values = upload_mont<PP>(ctx, t->alu_values, c.n_alu * 4 * D, "alu_values");
  if (!c.n_recompose_coeff) {
    d->recompose_values = upload_mont<PP>(ctx, t->recompose_values, c.n_recompose * D, "recompose_values");
  } else {
    // one device array: the rows of table 4, then those of table 5
    if (!t->recompose_coeff_values || (c.n_recompose && !t->recompose_values)) fail(P3R_EINVAL, "recompose values are NULL");
    std::vector<uint32_t> both((c.n_recompose + c.n_recompose_coeff) * D);
    std::copy(t->recompose_values, t->recompose_values + c.n_recompose * D, both.begin());
    std::copy(t->recompose_coeff_values, t->recompose_coeff_values + c.n_recompose_coeff * D, both.begin() + c.n_recompose * D);
    d->recompose_values = upload_mont<PP>(ctx, both.data(), both.size(), "recompose_values");
  }
  if (!L->has_p2) return d;
  // Poseidon2 rows padded with fillers: new_start = true, zero state (poseidon2.rs:1125-1140)
  const size_t h = L->h_p2, n = c.n_p2;
  std::vector<uint32_t> in(h * 16, 0), idx(h, 0);
  std::vector<uint8_t> ns(h, 1), mp(h, 0), bit(h, 0);
  if (n) {
    if (!t->p2.input_values || !t->p2.new_start || !t->p2.merkle_path || !t->p2.mmcs_bit || !t->p2.mmcs_index_sum)
      fail(P3R_EINVAL, "Poseidon2 rows have a NULL field");
    std::copy(t->p2.input_values, t->p2.input_values + n * 16, in.begin());
    std::copy(t->p2.mmcs_index_sum, t->p2.mmcs_index_sum + n, idx.begin());
    std::copy(t->p2.new_start, t->p2.new_start + n, ns.begin());
    std::copy(t->p2.merkle_path, t->p2.merkle_path + n, mp.begin());
    std::copy(t->p2.mmcs_bit, t->p2.mmcs_bit + n, bit.begin());
  }
  p3r_p2_rows rows{h, in.data(), ns.data(), mp.data(), bit.data(), idx.data()};
  d->p2 = p2_rows_upload<PP>(ctx, &rows);
  if (L->has_p2w) {
    // rows of the width-32 table, padded with the same fillers
    const size_t hw = L->h_p2w, nw = c.n_p2w;
    const p3r_p2w_rows& w = t->p2w;
    if (w.n != nw) fail(P3R_EINVAL, "trace row counts do not match the prepared circuit shape (width-32 Poseidon2 table)");
    if (!w.input_values || !w.new_start || !w.merkle_path || !w.mmcs_bit || !w.mmcs_bit2 || !w.mmcs_index_sum)
      fail(P3R_EINVAL, "width-32 Poseidon2 rows have a NULL field");
    std::vector<uint32_t> win(hw * 32, 0), widx(hw, 0);
    std::vector<uint8_t> f(4 * hw, 0);
    std::fill(f.begin(), f.begin() + hw, 1);   // new_start = 1 on fillers
    std::copy(w.input_values, w.input_values + nw * 32, win.begin());
    std::copy(w.mmcs_index_sum, w.mmcs_index_sum + nw, widx.begin());
    std::copy(w.new_start, w.new_start + nw, f.begin());
    std::copy(w.merkle_path, w.merkle_path + nw, f.begin() + hw);
    std::copy(w.mmcs_bit, w.mmcs_bit + nw, f.begin() + 2 * hw);
    std::copy(w.mmcs_bit2, w.mmcs_bit2 + nw, f.begin() + 3 * hw);
    d->p2w = p2w_rows_upload<PP>(ctx, hw, nw, win.data(), f.data(), widx.data());
  }
  return d;
}

// K3 for the width-32 table: base-four accumulator scan, then one permutation per row
template <class PP>
std::unique_ptr<p3r_dmat> trace_fill_w32(p3r_ctx* ctx, const p3r_p2_dev* rows) {
  const size_t n = rows->n;
  const uint8_t* f8 = reinterpret_cast<const uint8_t*>(rows->flags.p);
  DevBuf acc(n);
  const size_t n_blocks = (n + kScanTile - 1) / kScanTile;
  DevBuf agg(2 * n_blocks);
  {
    ProfScope ps(ctx, "p2_acc_scan");
    for (int mode = 0; mode < 3; ++mode)
      hipLaunchKernelGGL(k_p2_acc_scan<PP>, dim3(mode == 1 ? 1u : (unsigned)n_blocks), dim3(kBlock), 0, ctx->stream, mode, n, f8, f8 + n,
                         f8 + 2 * n, rows->seed.p, agg.p, n_blocks, acc.p, f8 + 3 * n);
  }
  auto trace = dmat_alloc(n, (size_t)p2w_perm_cols<PP>() + 4);
  {
    ProfScope ps(ctx, "p2_trace_fill");
    hipLaunchKernelGGL(k_p2w_trace_fill<PP>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, rows->inputs->d, f8 + 2 * n, f8 + 3 * n,
                       acc.p, trace->d, n, ctx->rc.p + p2_num_constants<PP>());
  }
  P3R_HIP(hipGetLastError());
  return trace;
}

// K1 + K2 + K3: the main-trace matrices, indexed by table 0..4 (absent tables stay null).
template <class PP>
std::vector<std::unique_ptr<p3r_dmat>> build_main_traces(p3r_ctx* ctx, const p3r_layer* L, const p3r_dtraces* t) {
  std::vector<std::unique_ptr<p3r_dmat>> m(7);
  const int D = (int)ctx->cfg.ext_degree;
  auto flat = [&](const uint32_t* src, size_t n_ops, size_t h, int w, int per_op) {
    auto out = dmat_alloc(h, (size_t)w);
    ProfScope ps(ctx, "trace_to_matrix");
    hipLaunchKernelGGL(k_flat_to_colmajor<PP>, dim3(blocks_for(h * w)), dim3(kBlock), 0, ctx->stream, src,
                       n_ops * per_op, out->d, h, w);
    return out;
  };
  m[0] = flat(t->const_values.p, t->n_const, L->h_const, D, D);
  m[1] = flat(t->public_values.p, t->n_public, L->h_public, (int)L->public_lanes * D, D);
  {
    const int lanes = (int)L->alu_lanes, k = (int)L->horner_k;
    const int width = (lanes * 4 + (k - 1) / 2 + 2 * (k - 1) + 1) * D;
    m[2] = dmat_alloc(L->h_alu, (size_t)width);
    P3R_HIP(fill_async(ctx->stream, m[2]->d, 0, L->h_alu * (size_t)width * 4));
    ProfScope ps(ctx, "alu_trace");
    const auto* plan = reinterpret_cast<const AluPlanEntry*>(L->alu_plan.p);
    const uint32_t w_mont = ext_degree_is_binomial_generic((uint32_t)D) ? Fp<PP>::from_canonical(ctx->cfg.ext_w).v : 0u;
    dispatch_air_degree<PP>(D, [&](auto dc) {
      hipLaunchKernelGGL((k_alu_trace<PP, decltype(dc)::value>), dim3(blocks_for(L->alu_rows)), dim3(kBlock), 0, ctx->stream, plan,
                         L->alu_prev_src.p, t->alu_values.p, L->alu_rows, L->h_alu, lanes, k, m[2]->d, w_mont);
    });
  }
  if (L->has_p2) m[3] = trace_fill<PP>(ctx, t->p2.get());
  if (L->has_p2w) m[6] = trace_fill_w32<PP>(ctx, t->p2w.get());
  if (L->has_recompose)
    m[4] = flat(t->recompose_values.p, t->n_recompose, L->h_recompose, (int)L->recompose_lanes * D, D);
  if (L->has_recompose_coeff)
    m[5] = flat(t->recompose_values.p + t->n_recompose * (size_t)D, t->n_recompose_coeff, L->h_recompose_coeff,
                (int)L->recompose_lanes * D, D);
  P3R_HIP(hipGetLastError());
  return m;
}

template <class PP>
std::vector<uint8_t> prove_all_tables(p3r_ctx* ctx, const p3r_layer* L, const p3r_dtraces* t, bool canonical) {
  prof_stage(ctx, "build_traces");
  auto mains = build_main_traces<PP>(ctx, L, t);
  const p3r_dmat* ptrs[7];
  size_t n = 0;
  for (int i : kTableOrder)
    if (mains[i]) ptrs[n++] = mains[i].get();
  return prove_batch_any<PP>(ctx, L->prep.get(), ptrs, n, canonical);
}

}  // namespace
