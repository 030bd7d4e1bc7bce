// gfx950 kernels for the proving stages after the main-trace commit:
//   K7  LogUp auxiliary (permutation) trace       k_logup_aux, k_ef_scan
//   K8  quotient evaluation                        k_quotient
//   K9  openings at zeta / zeta*g                   k_bary_weights, k_open_dot, k_open_reduce
//   K10 FRI reduced openings, folds, grinding       k_fri_fold, k_grind (reduced openings: kernels_fri_reduce.hip.h)
//   query answers                                   k_gather
// Protocol anchors are listed in p3r_prove.hip next to the host code that sequences them.
#pragma once
#include "air_device.hip.h"
#include "kernels.hip.h"

namespace p3r {

// Extension elements (the challenge field, DC = 4 or 5 coefficients) travel to kernels by value as Montgomery words.
template <int DC>
struct EW {
  uint32_t c[DC];
};
using E4 = EW<4>;
template <class PP, int DC = 4>
__host__ __device__ __forceinline__ typename Chal<PP, DC>::type e4_load(const EW<DC>& e) {
  typename Chal<PP, DC>::type r;
  for (int i = 0; i < DC; ++i) r.c[i] = Fp<PP>::raw(e.c[i]);
  return r;
}
template <class PP, int DC = 4>
__host__ __device__ __forceinline__ EW<DC> e4_store(const typename Chal<PP, DC>::type& e) {
  EW<DC> r;
  for (int i = 0; i < DC; ++i) r.c[i] = e.c[i].v;
  return r;
}

// LogUp challenges: denominator = prefix + sum_j beta^j * field_j, tuple = (idx, v_0..v_{D-1}) for circuit
// extension degree D, prefix = alpha + beta^(D+1) (recursion/src/verifier/batch_stark.rs:1086-1100).
template <int DC>
struct LookupChT {
  EW<DC> prefix;
  EW<DC> beta_pow[kMaxExtD + 1];
};
using LookupCh = LookupChT<4>;
template <class PP, int D, int DC>
__device__ __forceinline__ typename Chal<PP, DC>::type lookup_denom(const LookupChT<DC>& lc, Fp<PP> idx, const VD<Fp<PP>, D>& v) {
  typename Chal<PP, DC>::type d = e4_load<PP, DC>(lc.prefix) + e4_load<PP, DC>(lc.beta_pow[0]) * idx;
#pragma unroll
  for (int j = 0; j < D; ++j) d += e4_load<PP, DC>(lc.beta_pow[j + 1]) * v.c[j];
  return d;
}

// ------------------------------------------------------------------ K7: aux trace
// One lane per trace row: fraction columns f_g = sum_{k in g} m_k / d_k for each packed
// lookup group (groups are `pair` + 1 consecutive interactions: singletons, pairs, or - under a ZK configuration's
// budget - triples; the host checks that the packing computed from the degree budget has this shape), plus the row total.
template <class PP, int DC = 4>
struct AuxSink {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  const LookupChT<DC>& lc;
  gptr<uint32_t> aux;  // [DC*aw][n]
  size_t n, row;
  int pair;
  int cnt = 0;
  E cur, total;
  __device__ AuxSink(const LookupChT<DC>& l, gptr<uint32_t> a, size_t n_, size_t r, int p)
      : lc(l), aux(a), n(n_), row(r), pair(p), cur(E::zero()), total(E::zero()) {}
  __device__ __forceinline__ void flush() {
    int g = pair == 1 ? (cnt - 1) / 2 : pair == 2 ? (cnt - 1) / 3 : cnt - 1;
#pragma unroll
    for (int k = 0; k < DC; ++k) aux[(size_t)((g + 1) * DC + k) * n + row] = cur.c[k].v;
    total += cur;
    cur = E::zero();
  }
  template <int D>
  __device__ __forceinline__ void add(F idx, const VD<F, D>& v, F mult) {
    if (mult.v != 0) cur += lookup_denom<PP, D, DC>(lc, idx, v).inv() * mult;
    ++cnt;
    if (!pair || (pair == 1 ? (cnt & 1) == 0 : cnt % 3 == 0)) flush();
  }
  __device__ __forceinline__ void finish() {
    if (pair == 1 ? (cnt & 1) != 0 : pair == 2 && cnt % 3 != 0) flush();
  }
};

// The LogUp pass of every table of a proof is one launch of each kernel below over this job list.
constexpr int kMaxChalD = 5;   // widest challenge field (words per extension element)
struct LogupJob {
  AirParams air;
  const uint32_t* main;  // trace [w][n]
  const uint32_t* prep;  // preprocessed trace [w_prep][n]
  uint32_t* aux;         // [4*aux_w][n]: column 0 = running sum, then the fraction columns
  uint32_t* rowsum;      // [4][n] scratch: sum of the row's fractions
  uint32_t* agg;         // [n_tiles][4] scratch of the scan
  uint32_t* total;       // [4]: the table's global sum
  uint64_t n;
  int pair;
  uint32_t n_tiles;      // scan tiles of kScanTile rows
  uint32_t block0;       // first block in the row launch (k_logup_aux)
  uint32_t tile0;        // first block in the tile launches (k_ef_scan modes 0 and 2)
};

template <class PP, int D = 4, int DC = 4>
__global__ void __launch_bounds__(kBlock)
k_logup_aux(const LogupJob* __restrict__ jobs, int n_jobs, LookupChT<DC> lc) {
  const int jb = find_job(jobs, n_jobs);
  const LogupJob& job = jobs[jb];
  const size_t n = job.n, r = (size_t)(blockIdx.x - job.block0) * kBlock + threadIdx.x;
  if (r >= n) return;
  RowView<PP> v{as_global(job.main), as_global(job.prep), n, r, r + 1 == n ? 0 : r + 1};
  AuxSink<PP, DC> sink(lc, as_global(job.aux), n, r, job.pair);
  air_interactions<PP, D>(job.air, v, sink);
  sink.finish();
  const gptr<uint32_t> rowsum = as_global(job.rowsum);
#pragma unroll
  for (int k = 0; k < DC; ++k) rowsum[(size_t)k * n + r] = sink.total.c[k].v;
}

// Exclusive prefix sum of extension elements (running LogUp sum, aux column 0) per job.
// mode 0: tile totals -> agg; mode 1: exclusive scan of agg (one block per job), grand total ->
// total; mode 2: write exclusive prefixes.
template <class PP, int DC = 4>
__global__ void __launch_bounds__(kBlock) k_ef_scan(int mode, const LogupJob* __restrict__ jobs, int n_jobs) {
  using E = typename Chal<PP, DC>::type;
  using F = Fp<PP>;
  __shared__ uint32_t sh[DC][kBlock];
  const int tid = threadIdx.x;
  int jb = 0;
  if (mode == 1) jb = blockIdx.x;
  else
    while (jb + 1 < n_jobs && blockIdx.x >= jobs[jb + 1].tile0) ++jb;
  const size_t n = jobs[jb].n, n_blocks = jobs[jb].n_tiles;
  const uint32_t tile = mode == 1 ? 0 : blockIdx.x - jobs[jb].tile0;
  const gptr<const uint32_t> in = as_global(jobs[jb].rowsum);
  const gptr<uint32_t> agg = as_global(jobs[jb].agg), out = as_global(jobs[jb].aux),
                       total = as_global(jobs[jb].total);
  E loc[kScanItems];
  E run = E::zero();
  if (mode == 1) {
    size_t per = (n_blocks + kBlock - 1) / kBlock;
    size_t lo = (size_t)tid * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    for (size_t j = lo; j < hi; ++j) {
      E m;
      for (int k = 0; k < DC; ++k) m.c[k] = F::raw(agg[DC * j + k]);
      run += m;
    }
  } else {
    size_t base = (size_t)tile * kScanTile + (size_t)tid * kScanItems;
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) {
      size_t i = base + q;
      loc[q] = E::zero();
      if (i < n)
        for (int k = 0; k < DC; ++k) loc[q].c[k] = F::raw(in[(size_t)k * n + i]);
      run += loc[q];
    }
  }
  for (int k = 0; k < DC; ++k) sh[k][tid] = run.c[k].v;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {
    E prev = E::zero(), cur;
    bool has = tid >= off;
    for (int k = 0; k < DC; ++k) {
      if (has) prev.c[k] = F::raw(sh[k][tid - off]);
      cur.c[k] = F::raw(sh[k][tid]);
    }
    __syncthreads();
    if (has) {
      cur += prev;
      for (int k = 0; k < DC; ++k) sh[k][tid] = cur.c[k].v;
    }
    __syncthreads();
  }
  E excl = E::zero();
  if (tid > 0)
    for (int k = 0; k < DC; ++k) excl.c[k] = F::raw(sh[k][tid - 1]);
  if (mode == 0) {
    if (tid == kBlock - 1)
      for (int k = 0; k < DC; ++k) agg[DC * (size_t)tile + k] = sh[k][tid];
  } else if (mode == 1) {
    size_t per = (n_blocks + kBlock - 1) / kBlock;
    size_t lo = (size_t)tid * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    E p = excl;
    for (size_t j = lo; j < hi; ++j) {
      E m;
      for (int k = 0; k < DC; ++k) m.c[k] = F::raw(agg[DC * j + k]);
      for (int k = 0; k < DC; ++k) agg[DC * j + k] = p.c[k].v;
      p += m;
    }
    if (tid == kBlock - 1)
      for (int k = 0; k < DC; ++k) total[k] = sh[k][tid];
  } else {
    E p;
    for (int k = 0; k < DC; ++k) p.c[k] = F::raw(agg[DC * (size_t)tile + k]);
    p += excl;
    size_t base = (size_t)tile * kScanTile + (size_t)tid * kScanItems;
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) {
      size_t i = base + q;
      if (i < n)
        for (int k = 0; k < DC; ++k) out[(size_t)k * n + i] = p.c[k].v;
      p += loc[q];
    }
  }
}

// ------------------------------------------------------------------ K8: quotient
struct QuotientArgs {
  AirParams air;
  const uint32_t* main;   // bit-reversed LDEs, height lde_h
  const uint32_t* prep;
  const uint32_t* aux;    // nullable
  size_t lde_h;
  int log_n;              // trace height
  int log_chunks;         // log2 of quotient chunks C (ZK: one more than the AIR's log_quotient_chunks, batch_stark.rs:701-717)
  const uint32_t* apow;   // alpha^j as 4 words each, j = 0..: one table for all AIRs of a proof
  int n_constraints;      // N: constraint k (base constraints first) is weighted alpha^(N-1-k)
  int n_base, n_groups, pair;
  uint32_t terminal[kMaxChalD];   // the instance's LogUp terminal, DC words
  uint32_t gen;           // coset shift (Montgomery)
  uint32_t w_q;           // generator of the quotient domain (size n*C)
  uint32_t g_inv;         // inverse trace-domain generator
  uint32_t zh[8];         // Z_H on the C cosets:  gen^n * w_C^c - 1
  uint32_t zh_inv[8];
  uint32_t* out;          // [C][DC][n] chunk evaluations, natural order
  uint32_t block0;        // first block of this table in the launch (all tables of a proof share it)
};

template <class PP, int DC = 4>
struct BaseFold {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  gptr<const uint32_t> apow;  // alpha^j, ascending
  int k;                      // exponent of the next constraint's weight: N-1, N-2, ...
  E acc;
  __device__ BaseFold(gptr<const uint32_t> a, int n) : apow(a), k(n - 1), acc(E::zero()) {}
  __device__ __forceinline__ E pw() {
    E p;
#pragma unroll
    for (int i = 0; i < DC; ++i) p.c[i] = F::raw(apow[DC * k + i]);
    --k;
    return p;
  }
  __device__ __forceinline__ void base(F c) { acc += pw() * c; }
  // two consecutive base constraints: their products share one reduction per coefficient
  __device__ __forceinline__ void base2(F c0, F c1) {
    const E p0 = pw(), p1 = pw();
    acc += E::dot2_base(p0, c0, p1, c1);
  }
  __device__ __forceinline__ void ext(const E& c) { acc += pw() * c; }
};

// Collects the LogUp group constraints:  f_g * prod d_k - sum_k m_k prod_{l!=k} d_l.
template <class PP, int DC = 4>
struct QuotSink {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  const QuotientArgs& q;
  const LookupChT<DC>& lc;
  gptr<const uint32_t> aux;
  BaseFold<PP, DC>& fold;
  size_t row;
  int cnt = 0, held = 0;   // held: interactions of the open group (pairs / triples)
  E d0, d1, sum_f;
  F m0, m1;
  __device__ QuotSink(const QuotientArgs& q_, const LookupChT<DC>& lc_, BaseFold<PP, DC>& f, size_t r)
      : q(q_), lc(lc_), aux(as_global(q_.aux)), fold(f), row(r), d0(E::zero()), d1(E::zero()), sum_f(E::zero()),
        m0(F::zero()), m1(F::zero()) {}
  __device__ __forceinline__ E aux_at(int col, size_t r) const {
    E e;
#pragma unroll
    for (int k = 0; k < DC; ++k) e.c[k] = F::raw(aux[(size_t)(col * DC + k) * q.lde_h + r]);
    return e;
  }
  template <int D>
  __device__ __forceinline__ void add(F idx, const VD<F, D>& v, F mult) {
    E d = lookup_denom<PP, D, DC>(lc, idx, v);
    ++cnt;
    if (!q.pair) {
      E f = aux_at(cnt, row);
      fold.ext(f * d - E::from_base(mult));
      sum_f += f;
    } else if (q.pair == 1) {
      if (cnt & 1) {
        d0 = d;
        m0 = mult;
        held = 1;
      } else {
        E f = aux_at(cnt / 2, row);
        fold.ext(f * d0 * d - (d * m0 + d0 * mult));
        sum_f += f;
        held = 0;
      }
    } else {   // triples: f d0 d1 d2 - (m0 d1 d2 + m1 d0 d2 + m2 d0 d1)
      if (held == 0) {
        d0 = d; m0 = mult; held = 1;
      } else if (held == 1) {
        d1 = d; m1 = mult; held = 2;
      } else {
        E f = aux_at(cnt / 3, row);
        const E d01 = d0 * d1;
        fold.ext(f * d01 * d - ((d1 * m0 + d0 * m1) * d + d01 * mult));
        sum_f += f;
        held = 0;
      }
    }
  }
  __device__ __forceinline__ void finish() {
    if (!held) return;
    const int G = q.pair + 1;
    E f = aux_at((cnt + G - 1) / G, row);
    if (held == 1) fold.ext(f * d0 - E::from_base(m0));
    else fold.ext(f * d0 * d1 - (d1 * m0 + d0 * m1));
    sum_f += f;
  }
};

// One launch for all tables of a proof: `jobs` lists them, a block finds its table by walking the
// first-block indices.  The LogUp challenges and the round constants are the same for every table.
template <class PP, int D = 4, int DC = 4>
__global__ void __launch_bounds__(kBlock)
k_quotient(const QuotientArgs* __restrict__ jobs, int n_jobs, LookupChT<DC> lc, const uint32_t* __restrict__ rc) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  const int jb = find_job(jobs, n_jobs);
  const QuotientArgs& q = jobs[jb];
  const int lq = q.log_n + q.log_chunks;
  const size_t qn = size_t(1) << lq, C = size_t(1) << q.log_chunks;
  size_t j = (size_t)(blockIdx.x - q.block0) * kBlock + threadIdx.x;  // LDE row
  if (j >= qn) return;
  const uint32_t i = bit_reverse((uint32_t)j, lq);        // natural index on the quotient coset
  const uint32_t i_next = (uint32_t)((i + C) & (qn - 1));
  RowView<PP> v{as_global(q.main), as_global(q.prep), q.lde_h, j, bit_reverse(i_next, lq)};
  const F x = F::raw(q.gen) * F::raw(q.w_q).pow(i);
  const uint32_t c = i & (uint32_t)(C - 1);
  const F zh = F::raw(q.zh[c]), g_inv = F::raw(q.g_inv);
  const F is_transition = x - g_inv;
  BaseFold<PP, DC> fold(as_global(q.apow), q.n_constraints);
  if (q.air.kind == AIR_ALU) alu_constraints<PP, D>(q.air, v, fold);
  if (q.air.kind == AIR_POSEIDON2) {
    if constexpr (D == 4) poseidon2_constraints<PP>(v, is_transition, rc, fold);
    else if constexpr (D == 1 || D == 5) poseidon2_d1_constraints<PP>(v, is_transition, rc, fold);
  }
  if (q.air.kind == AIR_POSEIDON2_W32) {
    // (the width-32 constant table follows the width-16 one in p3r_ctx::rc)
    if constexpr (D == 4) poseidon2w_constraints<PP>(v, is_transition, rc + p2_num_constants<PP>(), fold);
  }
  if (q.aux) {
    const F is_first = zh * (x - F::one()).inv();
    const F is_last = zh * is_transition.inv();
    QuotSink<PP, DC> sink(q, lc, fold, j);
    air_interactions<PP, D>(q.air, v, sink);
    sink.finish();
    E s = sink.aux_at(0, j), s_next = sink.aux_at(0, v.nxt);
    fold.ext(s * is_first);
    fold.ext((s_next - s - sink.sum_f) * is_transition);
    E terminal;
#pragma unroll
    for (int k = 0; k < DC; ++k) terminal.c[k] = F::raw(q.terminal[k]);
    fold.ext((s + sink.sum_f - terminal) * is_last);
  }
  E quot = fold.acc * F::raw(q.zh_inv[c]);
  const size_t n = size_t(1) << q.log_n, r = i >> q.log_chunks;
#pragma unroll
  for (int k = 0; k < DC; ++k) as_global(q.out)[((size_t)c * DC + k) * n + r] = quot.c[k].v;
}

// ------------------------------------------------------------------ K9: openings
// All openings of one proof run as three launches over job lists (a recursion layer opens ~20
// matrices at 1-2 points each; per-matrix launches are latency-bound for 2^14..2^16-row layers).
// A block finds its job by walking the (short) list of first-block indices.
//
// Inverses of four extension elements with ONE base-field inversion (Montgomery's trick on their
// norms): the tower inverse costs ~25 products plus a ~56-product exponentiation in the base
// field, and the latter is what the four share.  A zero among them (never, for z outside the base
// field) falls back to separate inversions so that it cannot poison its neighbours.
template <class PP>
__device__ __forceinline__ void inv4(const Fp5<PP> (&x)[4], Fp5<PP> (&out)[4]) {
  // the quintic field's norm is N(a) = a * a^(r-1) (field.h: norm_cofactor); the four norms share the inversion
  using F = Fp<PP>;
  Fp5<PP> co[4];
  F nm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    co[i] = x[i].norm_cofactor();
    nm[i] = Fp5<PP>::mul_c0(x[i], co[i]);
  }
  const F p01 = nm[0] * nm[1], p012 = p01 * nm[2], p0123 = p012 * nm[3];
  if (p0123.v == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = x[i].inv();
    return;
  }
  F t = p0123.inv();
  const F i3 = t * p012;
  t = t * nm[3];
  const F i2 = t * p01;
  t = t * nm[2];
  const F i1 = t * nm[0], i0 = t * nm[1];
  out[0] = co[0] * i0;
  out[1] = co[1] * i1;
  out[2] = co[2] * i2;
  out[3] = co[3] * i3;
}
template <class PP>
__device__ __forceinline__ void inv4(const Fp4<PP> (&x)[4], Fp4<PP> (&out)[4]) {
  using F = Fp<PP>;
  typename Fp4<PP>::Norm nm[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) nm[i] = x[i].norm();
  const F p01 = nm[0].d * nm[1].d, p012 = p01 * nm[2].d, p0123 = p012 * nm[3].d;
  if (p0123.v == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = x[i].inv();
    return;
  }
  F t = p0123.inv();
  const F i3 = t * p012;
  t = t * nm[3].d;
  const F i2 = t * p01;
  t = t * nm[2].d;
  const F i1 = t * nm[0].d, i0 = t * nm[1].d;
  out[0] = x[0].inv_given(nm[0], i0);
  out[1] = x[1].inv_given(nm[1], i1);
  out[2] = x[2].inv_given(nm[2], i2);
  out[3] = x[3].inv_given(nm[3], i3);
}

// Barycentric weights over the trace subgroup:  L_i(z) = w^i (z^n - 1) / (n (z - w^i)).
// `scale` = (z^n - 1)/n is supplied by the host.
template <int DC>
struct BaryJobT {
  uint32_t* out;  // [DC][n]
  uint64_t n;
  uint32_t w_n;
  EW<DC> z, scale;
  uint32_t block0;  // first block of this job
};
using BaryJob = BaryJobT<4>;
template <class PP, int DC = 4>
__global__ void __launch_bounds__(kBlock) k_bary_weights(const BaryJobT<DC>* __restrict__ jobs, int n_jobs) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  const int j = find_job(jobs, n_jobs);
  const BaryJobT<DC>& b = jobs[j];
  // four consecutive points per lane: their inversions share one base-field inversion (inv4)
  const size_t i0 = ((size_t)(blockIdx.x - b.block0) * kBlock + threadIdx.x) * 4;
  if (i0 >= b.n) return;
  const F w1 = F::raw(b.w_n);
  const E z = e4_load<PP, DC>(b.z), scale = e4_load<PP, DC>(b.scale);
  F wi[4];
  E x[4], inv[4];
  wi[0] = w1.pow(i0);
#pragma unroll
  for (int m = 1; m < 4; ++m) wi[m] = wi[m - 1] * w1;
#pragma unroll
  for (int m = 0; m < 4; ++m) x[m] = z - E::from_base(wi[m]);
  inv4<PP>(x, inv);
  const gptr<uint32_t> out = as_global(b.out);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (i0 + m < b.n) {
      const E r = inv[m] * scale * wi[m];
#pragma unroll
      for (int k = 0; k < DC; ++k) out[(size_t)k * b.n + i0 + m] = r.c[k].v;
    }
  }
}

constexpr int kOpenCols = 8;      // matrix columns sharing one pass over the weights
constexpr int kOpenRows = 8192;   // rows per block for tall matrices (the host shrinks it for short ones)
struct OpenJob {
  const uint32_t* mat;  // [w][n] column-major, natural order
  const uint32_t *wt0, *wt1;  // weights per point ([DC][n]); wt1 null for a single point
  uint32_t* partial;    // [P][n_chunks][w][DC]
  uint64_t n;
  int w, n_chunks, rows_per_block, col_groups;
  uint32_t block0;      // first block of this job in the dot launch
  uint32_t out0;        // first output word of this job ([P][w][DC]) in the reduce launch
};
// partial[p][chunk][col] = sum over the chunk's rows of weights_p[row] * M[col][row].
// All accumulator indexing is compile-time (register resident); the block reduction is a
// wave shuffle tree followed by a 4-wave LDS combine.
template <class PP, int P, int DC>
__device__ __forceinline__ void open_dot_block(const OpenJob& job, int col_group, int chunk,
                                               uint32_t (*sh)[2 * kOpenCols * DC]) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  constexpr int NV = P * kOpenCols * DC;
  const gptr<const uint32_t> mat = as_global(job.mat);
  const gptr<const uint32_t> wt0 = as_global(job.wt0);
  const gptr<const uint32_t> wt1 = as_global(job.wt1);
  const size_t n = job.n;
  const int w = job.w, c0 = col_group * kOpenCols;
  size_t r0 = (size_t)chunk * job.rows_per_block, r1 = r0 + job.rows_per_block < n ? r0 + job.rows_per_block : n;
  E acc[P][kOpenCols];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int c = 0; c < kOpenCols; ++c) acc[p][c] = E::zero();
  // two rows per step: their products share one reduction per coefficient
  for (size_t r = r0 + threadIdx.x; r < r1; r += 2 * kBlock) {
    const size_t rb = r + kBlock;
    const bool has_b = rb < r1;
    E wa[P], wb[P];
#pragma unroll
    for (int k = 0; k < DC; ++k) {
      wa[0].c[k] = F::raw(wt0[(size_t)k * n + r]);
      wb[0].c[k] = has_b ? F::raw(wt0[(size_t)k * n + rb]) : F::zero();
    }
    if (P == 2)
#pragma unroll
      for (int k = 0; k < DC; ++k) {
        wa[P - 1].c[k] = F::raw(wt1[(size_t)k * n + r]);
        wb[P - 1].c[k] = has_b ? F::raw(wt1[(size_t)k * n + rb]) : F::zero();
      }
#pragma unroll
    for (int c = 0; c < kOpenCols; ++c) {
      const bool col = c0 + c < w;
      const F ma = col ? F::raw(mat[(size_t)(c0 + c) * n + r]) : F::zero();
      const F mb = col && has_b ? F::raw(mat[(size_t)(c0 + c) * n + rb]) : F::zero();
#pragma unroll
      for (int p = 0; p < P; ++p) acc[p][c] += E::dot2_base(wa[p], ma, wb[p], mb);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int c = 0; c < kOpenCols; ++c)
#pragma unroll
      for (int k = 0; k < DC; ++k) {
        F v = acc[p][c].c[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += F::raw(__shfl_down(v.v, off));
        if (lane == 0) sh[wave][(p * kOpenCols + c) * DC + k] = v.v;
      }
  __syncthreads();
  if ((int)threadIdx.x < NV) {
    F s = F::zero();
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; ++wv) s += F::raw(sh[wv][threadIdx.x]);
    const int p = threadIdx.x / (kOpenCols * DC), rem = threadIdx.x % (kOpenCols * DC), c = rem / DC, k = rem % DC;
    if (c0 + c < w) as_global(job.partial)[(((size_t)p * job.n_chunks + chunk) * w + c0 + c) * DC + k] = s.v;
  }
}
template <class PP, int DC = 4>
__global__ void __launch_bounds__(kBlock) k_open_dot(const OpenJob* __restrict__ jobs, int n_jobs) {
  __shared__ uint32_t sh[kBlock / 64][2 * kOpenCols * DC];
  const int j = find_job(jobs, n_jobs);
  const OpenJob job = jobs[j];
  const int local = (int)(blockIdx.x - job.block0);
  const int col_group = local % job.col_groups, chunk = local / job.col_groups;
  if (job.wt1) open_dot_block<PP, 2, DC>(job, col_group, chunk, sh);
  else open_dot_block<PP, 1, DC>(job, col_group, chunk, sh);
}
// out[out0 + (p*w + c)*DC + k] = sum over chunks of partial[p][chunk][c][k]
template <class PP, int DC = 4>
__global__ void __launch_bounds__(kBlock)
k_open_reduce(const OpenJob* __restrict__ jobs, int n_jobs, uint32_t total, uint32_t* __restrict__ out) {
  using F = Fp<PP>;
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= total) return;
  int j = 0, hi = n_jobs - 1;   // the last job whose first output is not past t (bisection: context.h::find_job)
  while (j < hi) {
    const int mid = (j + hi + 1) >> 1;
    if (t >= jobs[mid].out0) j = mid; else hi = mid - 1;
  }
  const OpenJob& job = jobs[j];
  const uint32_t local = t - job.out0;
  const uint32_t per_point = (uint32_t)job.w * DC, p = local / per_point, rem = local % per_point;
  F s = F::zero();
  const gptr<const uint32_t> partial = as_global(job.partial);
  for (int ch = 0; ch < job.n_chunks; ++ch)
    s += F::raw(partial[((size_t)p * job.n_chunks + ch) * per_point + rem]);
  out[t] = s.v;
}

// ------------------------------------------------------------------ K10: FRI
// One commit-phase fold of arity 2^la: la sequential arity-2 folds with beta, beta^2, ...
// (recursion/src/pcs/fri/verifier.rs:562-781), then the roll-in  + beta^{2^la} * ro.
struct FriFoldArgs {
  const uint32_t* in;   // [4][rows << la]
  uint32_t* out;        // [4][rows]
  size_t rows;
  int la, log_rows;
  const uint32_t* beta; // the phase's folding challenge, 4 words on the device (written by the
                        // device-side transcript step, or uploaded by the host)
  const uint32_t* roll; // nullable [4][rows]
  uint32_t w_inv;       // inverse generator of the domain of size rows << la
  uint32_t tw_inv[3][4];  // tw_inv[s][j] = (w_arity^{2^s})^{-bitrev(2j, la - s)}
  uint32_t neg_half;
};
template <class PP, int DC = 4>
__global__ void __launch_bounds__(kBlock) k_fri_fold(FriFoldArgs a) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= a.rows) return;
  const size_t n_in = a.rows << a.la;
  E e[8];
  const int arity = 1 << a.la;
  for (int j = 0; j < arity; ++j)
#pragma unroll
    for (int k = 0; k < DC; ++k) e[j].c[k] = F::raw(a.in[(size_t)k * n_in + (r << a.la) + j]);
  F ss_inv = F::raw(a.w_inv).pow(bit_reverse((uint32_t)r, a.log_rows));
  E b;
#pragma unroll
  for (int k = 0; k < DC; ++k) b.c[k] = F::raw(a.beta[k]);
  const F nh = F::raw(a.neg_half);
  int len = arity;
  for (int s = 0; s < a.la; ++s) {
    for (int j = 0; j < len / 2; ++j) {
      F x0_inv = ss_inv * F::raw(a.tw_inv[s][j]);
      // e0 + (beta - x0)(e1 - e0)(-1/2)/x0  =  e0 + (beta/x0 - 1)(e1 - e0)(-1/2)
      E t = b * x0_inv - E::one();
      e[j] = e[2 * j] + t * (e[2 * j + 1] - e[2 * j]) * nh;
    }
    len /= 2;
    ss_inv = ss_inv.sqr();
    b = b.sqr();
  }
  E res = e[0];
  if (a.roll) {
    E ro;
#pragma unroll
    for (int k = 0; k < DC; ++k) ro.c[k] = F::raw(a.roll[(size_t)k * a.rows + r]);
    res += b * ro;  // b is beta^(2^la) after the la squarings above
  }
#pragma unroll
  for (int k = 0; k < DC; ++k) a.out[(size_t)k * a.rows + r] = res.c[k].v;
}

// Leaf hashing for column sets that are strided views (FRI commit-phase leaves are the
// rows of the `rows x (arity*4)` matrix laid over the folded vector).
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_mmcs_hash_rows_strided(const uint32_t* const* __restrict__ cols, int wtot, size_t h, size_t stride,
                         uint32_t* __restrict__ dig, const double* __restrict__ rcd) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= h) return;
  double s[P2_WIDTH];
#pragma unroll
  for (int k = 0; k < P2_WIDTH; ++k) s[k] = 0.0;
  int g = 0;
  for (; g + P2_RATE <= wtot; g += P2_RATE) {
#pragma unroll
    for (int j = 0; j < P2_RATE; ++j) s[j] = p2f_load<PP>(cols[g + j][i * stride]);
    p2f_permute<PP, 0xFF00u>(s, rcd);
  }
  int rem = wtot - g;
  if (rem > 0) {
#pragma unroll
    for (int j = 0; j < P2_RATE; ++j)
      if (j < rem) s[j] = p2f_load<PP>(cols[g + j][i * stride]);
    p2f_permute<PP, 0xFFFFu>(s, rcd);
  }
#pragma unroll
  for (int k = 0; k < P2_DIGEST; ++k) dig[(size_t)k * h + i] = p2f_store<PP>(s[k]);
}

// Proof-of-work grinding: candidate witness w = base + tid; smallest valid one wins
// (check = low `bits` bits of the next sample are zero, recursion/src/challenger/circuit.rs:409-430).
struct GrindArgs {
  uint32_t state[16];   // sponge state before the witness is observed (Montgomery)
  uint32_t pending[8];  // buffered, not yet absorbed inputs (Montgomery)
  int n_pending;
  int bits;
  uint32_t base;
  const uint32_t* rc;
  uint32_t* result;     // atomicMin target, initialised to 0xFFFFFFFF
};
template <class PP>
__global__ void __launch_bounds__(kBlock) k_grind(GrindArgs a) {
  using F = Fp<PP>;
  uint32_t cand = a.base + blockIdx.x * kBlock + threadIdx.x;
  if (cand >= PP::P) return;
  F s[P2_WIDTH];
#pragma unroll
  for (int k = 0; k < P2_WIDTH; ++k) s[k] = F::raw(a.state[k]);
  const int n_abs = a.n_pending + 1;
#pragma unroll
  for (int k = 0; k < P2_RATE; ++k) {
    if (k < a.n_pending) s[k] = F::raw(a.pending[k]);
    else if (k == a.n_pending) s[k] = F::from_canonical(cand);
    else s[k] = F::zero();
  }
  s[P2_RATE] += F::from_canonical((uint32_t)n_abs);
  p2_permute<PP>(s, a.rc);
  uint32_t sample = s[P2_RATE - 1].to_canonical();
  if ((sample & ((1u << a.bits) - 1)) == 0) atomicMin(a.result, cand);
}

// Query answers: copy `count` strided cells into a contiguous staging buffer.
// Query answers.  What a query opens is the same for every query of a proof - one row of every
// committed matrix, one sibling digest per tree level, the sibling evaluations and the Merkle path
// of every FRI phase - and only WHERE depends on the query index, through a shift and (for tree
// siblings) a flipped low bit.  The item list is therefore static per proof shape (it stays on the
// device, content-keyed); a proof uploads its query indices and launches (items x queries) blocks.
struct QueryItem {
  const uint32_t* base;
  uint64_t stride;   // words between consecutive elements of the item
  uint32_t count;    // elements
  uint32_t dst;      // offset inside a query's block of the output
  uint32_t shift;    // position = ((index >> shift) ^ flip) * mul
  uint32_t flip;     // 1 for tree siblings
  uint32_t mul;
};
template <class PP>
__global__ void __launch_bounds__(64)
k_gather(const QueryItem* __restrict__ items, const uint32_t* __restrict__ indices, uint32_t words_per_query,
         uint32_t* __restrict__ out) {
  const QueryItem it = items[blockIdx.x];
  const uint32_t q = blockIdx.y;
  const size_t pos = (size_t)((indices[q] >> it.shift) ^ it.flip) * it.mul;
  const gptr<const uint32_t> src = as_global(it.base) + pos;
  uint32_t* dst = out + (size_t)q * words_per_query + it.dst;
  for (uint32_t i = threadIdx.x; i < it.count; i += 64) dst[i] = src[(size_t)i * it.stride];
}

}  // namespace p3r

