// ORACLE (test infrastructure): Poseidon2 width-16, PaddingFreeSponge<_,16,8,8>,
// TruncatedPermutation<_,2,8,16>, DuplexChallenger<_,_,16,8> and MerkleTreeMmcs; Poseidon2 width-32 with
// PaddingFreeSponge<_,32,24,8>, TruncatedPermutation<_,4,8,32> and the arity-4 MerkleTreeMmcs<..,4,8>.
// PARITY UNPINNED (see field.hpp).  Each function cites the in-tree restatement it follows; for the arity-4 tree
// that is its in-circuit verifier (recursion/src/pcs/mmcs.rs:866-1316) - the native builder is an un-vendored crate.
#pragma once
#include <algorithm>
#include <memory>
#include <numeric>

#include "field.hpp"

namespace orc {

constexpr int WIDTH = 16, RATE = 8, DIGEST = 8, HALF_FULL = 4;

// Poseidon2 width 32 (Poseidon2{Koala,Baby}Bear<32>: the permutation of the arity-4 MMCS, circuit-prover/tests/arity4_mmcs.rs:42-47):
// same round structure as width 16 - external layer circ(2 M4, M4, .., M4) over eight blocks, internal layer
// diag + all-ones - with 31 / 30 partial rounds.  Round constants AND the internal diagonal are the caller's data
// (upstream's live in un-vendored crates and are not restated in-tree).
constexpr int WIDTH32 = 32;
template <class FP>
struct Poseidon2W32 {
  using F = Fe<FP>;
  std::vector<F> rc;                       // [4][32] | [PARTIAL_W32] | [4][32]
  std::array<F, WIDTH32> diag;
  static constexpr int num_constants() { return 2 * HALF_FULL * WIDTH32 + FP::PARTIAL_W32; }
  static constexpr int perm_cols() {
    return WIDTH32 + 2 * HALF_FULL * (WIDTH32 * FP::SBOX_REGS + WIDTH32) + FP::PARTIAL_W32 * (FP::SBOX_REGS + 1);
  }
  Poseidon2W32(const uint32_t* rc_canonical, const uint32_t* diag_canonical) {
    rc.resize(num_constants());
    for (int i = 0; i < num_constants(); ++i) rc[i] = F(rc_canonical[i]);
    for (int i = 0; i < WIDTH32; ++i) diag[i] = F(diag_canonical[i]);
  }
  static F ext_entry(int i, int j) {
    static const int M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    return F((uint64_t)M4[i % 4][j % 4] * ((i / 4 == j / 4) ? 2 : 1));
  }
  void external(std::array<F, WIDTH32>& s) const {
    std::array<F, WIDTH32> o{};
    for (int i = 0; i < WIDTH32; ++i)
      for (int j = 0; j < WIDTH32; ++j) o[i] += ext_entry(i, j) * s[j];
    s = o;
  }
  void internal(std::array<F, WIDTH32>& s) const {
    F sum = F::zero();
    for (auto& x : s) sum += x;
    for (int i = 0; i < WIDTH32; ++i) s[i] = s[i] * diag[i] + sum;
  }
  void permute(std::array<F, WIDTH32>& s, std::vector<F>* cells = nullptr) const {
    if (cells) for (auto& x : s) cells->push_back(x);
    external(s);
    int k = 0;
    auto full = [&]() {
      for (int i = 0; i < WIDTH32; ++i) {
        F x = s[i] + rc[k + i];
        if (FP::SBOX_REGS == 1 && cells) cells->push_back(x * x * x);
        s[i] = x.pow(FP::SBOX_DEGREE);
      }
      k += WIDTH32;
      external(s);
      if (cells) for (auto& x : s) cells->push_back(x);
    };
    for (int r = 0; r < HALF_FULL; ++r) full();
    for (int r = 0; r < FP::PARTIAL_W32; ++r) {
      F x = s[0] + rc[k++];
      if (FP::SBOX_REGS == 1 && cells) cells->push_back(x * x * x);
      s[0] = x.pow(FP::SBOX_DEGREE);
      if (cells) cells->push_back(s[0]);
      internal(s);
    }
    for (int r = 0; r < HALF_FULL; ++r) full();
  }
  // PaddingFreeSponge<Perm32, 32, 24, 8>, overwrite mode: the leaf hash of the arity-4 MMCS
  // (`MyHashArity4`, recursion/examples/recursive_aggregation.rs:1026-1029; in-circuit restatement
  // add_hash_base_coeffs_overwrite called from recursion/src/pcs/mmcs.rs:963-985)
  static constexpr int RATE32 = 24;
  std::array<F, DIGEST> hash(const std::vector<F>& in) const {
    std::array<F, WIDTH32> s{};
    for (size_t i = 0; i < in.size();) {
      const size_t take = std::min<size_t>(RATE32, in.size() - i);
      for (size_t j = 0; j < take; ++j) s[j] = in[i + j];
      permute(s);
      i += take;
    }
    std::array<F, DIGEST> d;
    std::copy(s.begin(), s.begin() + DIGEST, d.begin());
    return d;
  }
  // TruncatedPermutation<Perm32, 4, 8, 32>: perm(c0 || c1 || c2 || c3)[0..8] (recursion/src/pcs/mmcs.rs:1010-1075: chunk k
  // spans lanes [8k, 8k + 8))
  std::array<F, DIGEST> compress4(const std::array<std::array<F, DIGEST>, 4>& c) const {
    std::array<F, WIDTH32> s;
    for (int k = 0; k < 4; ++k) std::copy(c[k].begin(), c[k].end(), s.begin() + DIGEST * k);
    permute(s);
    std::array<F, DIGEST> d;
    std::copy(s.begin(), s.begin() + DIGEST, d.begin());
    return d;
  }
};

// Level schedule of an arity-4 Merkle tree over mixed-height matrices (recursion/src/pcs/mmcs.rs:866-960:
// padded_len, arity4_path_schedule - "matching native arity_schedule").  A level compresses `step` children: 4, or 2
// (a bridge) when a shorter matrix has to be injected half way to the next quaternary layer; a layer of 2 logical
// nodes is padded to 4 with zero digests.  `inject_h`: the height of the matrices whose row digests are folded in
// after the level (0: none).
struct Arity4Step {
  int step;
  size_t logical_next, padded_next, inject_h;
};
inline size_t npt(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
inline size_t padded_len(size_t raw, size_t n) { return raw <= 1 ? raw : (raw >= n ? (raw + n - 1) / n * n : n); }
inline std::vector<Arity4Step> arity4_schedule(std::vector<size_t> heights, size_t num_roots) {
  std::stable_sort(heights.begin(), heights.end(), [](size_t a, size_t b) { return a > b; });
  const size_t max_height = heights.at(0), leaf_npt = npt(max_height);
  size_t at = 0;   // the leaf hash consumes every matrix of the tallest class
  while (at < heights.size() && npt(heights[at]) == leaf_npt) ++at;
  std::vector<Arity4Step> steps;
  size_t curr = padded_len(max_height, 4);
  while (curr > num_roots) {
    int step;
    if (curr < 4) step = 2;
    else {
      const size_t target = npt(curr / 4);
      bool intermediate = false;
      for (size_t k = at; k < heights.size(); ++k) intermediate |= npt(heights[k]) > target;
      step = intermediate ? 2 : 4;
    }
    const size_t logical_next = curr / step;
    curr = padded_len(logical_next, 4);
    size_t inject = 0;
    if (at < heights.size() && npt(heights[at]) == npt(logical_next)) {
      inject = heights[at];
      while (at < heights.size() && heights[at] == inject) ++at;
    }
    steps.push_back({step, logical_next, curr, inject});
  }
  return steps;
}

template <class FP>
struct Poseidon2 {
  using F = Fe<FP>;
  std::shared_ptr<Poseidon2W32<FP>> w32;   // set when the layer holds a width-32 table (AIR_POSEIDON2_W32)
  // flat: [4][16] external-initial | [PARTIAL] internal | [4][16] external-final
  // (poseidon2-circuit-air/src/public_types.rs:48-54,220-226 name the upstream statics)
  std::vector<F> rc;
  std::array<F, WIDTH> diag;            // internal diagonal v_i
  std::array<std::array<F, WIDTH>, WIDTH> ext;  // external matrix circ(2*M4, M4, M4, M4)

  static constexpr int num_constants() { return 2 * HALF_FULL * WIDTH + FP::PARTIAL; }
  // Poseidon2Cols width: inputs | 4x{sbox[16][R], post[16]} | P x {sbox[R], post_sbox} | 4x{..}
  static constexpr int perm_cols() {
    return WIDTH + 2 * HALF_FULL * (WIDTH * FP::SBOX_REGS + WIDTH) + FP::PARTIAL * (FP::SBOX_REGS + 1);
  }

  explicit Poseidon2(const uint32_t* rc_canonical) {
    rc.resize(num_constants());
    for (int i = 0; i < num_constants(); ++i) rc[i] = F(rc_canonical[i]);
    static const int M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    for (int i = 0; i < WIDTH; ++i)
      for (int j = 0; j < WIDTH; ++j)
        ext[i][j] = F((uint64_t)M4[i % 4][j % 4] * ((i / 4 == j / 4) ? 2 : 1));
    // SURVEY.md appendix A (upstream p3-{koala,baby}-bear internal diagonals)
    auto inv2k = [](int k) { return F(uint64_t(1) << k).inv(); };
    const F two(2), three(3), four(4);
    if (FP::P == 0x7f000001u) {
      diag = {-two, F::one(), two, inv2k(1), three, four, -inv2k(1), -three, -four, inv2k(8),
              inv2k(3), inv2k(24), -inv2k(8), -inv2k(3), -inv2k(4), -inv2k(24)};
    } else {
      diag = {-two, F::one(), two, inv2k(1), three, four, -inv2k(1), -three, -four, inv2k(8),
              inv2k(2), inv2k(3), inv2k(27), -inv2k(8), -inv2k(4), -inv2k(27)};
    }
  }

  void external(std::array<F, WIDTH>& s) const {
    std::array<F, WIDTH> o{};
    for (int i = 0; i < WIDTH; ++i)
      for (int j = 0; j < WIDTH; ++j) o[i] += ext[i][j] * s[j];
    s = o;
  }
  void internal(std::array<F, WIDTH>& s) const {
    F sum = F::zero();
    for (auto& x : s) sum += x;
    for (int i = 0; i < WIDTH; ++i) s[i] = s[i] * diag[i] + sum;
  }
  static F sbox(F x) { return x.pow(FP::SBOX_DEGREE); }

  // Permutation; when `cells` is non-null every committed trace cell is appended in
  // Poseidon2Cols order (p3_poseidon2_air::generate_trace_rows_for_perm, called at
  // poseidon2-circuit-air/src/air.rs:497-505).
  void permute(std::array<F, WIDTH>& s, std::vector<F>* cells = nullptr) const {
    if (cells) for (auto& x : s) cells->push_back(x);
    external(s);
    int k = 0;
    auto full = [&]() {
      for (int i = 0; i < WIDTH; ++i) {
        F x = s[i] + rc[k + i];
        if (FP::SBOX_REGS == 1 && cells) cells->push_back(x * x * x);
        s[i] = sbox(x);
      }
      k += WIDTH;
      external(s);
      if (cells) for (auto& x : s) cells->push_back(x);
    };
    for (int r = 0; r < HALF_FULL; ++r) full();
    for (int r = 0; r < FP::PARTIAL; ++r) {
      F x = s[0] + rc[k++];
      if (FP::SBOX_REGS == 1 && cells) cells->push_back(x * x * x);
      s[0] = sbox(x);
      if (cells) cells->push_back(s[0]);
      internal(s);
    }
    for (int r = 0; r < HALF_FULL; ++r) full();
  }

  // PaddingFreeSponge<Perm,16,8,8>::hash_iter, overwrite mode
  // (recursion/src/pcs/mmcs.rs:17-26 and the chunk loop :75-172).
  std::array<F, DIGEST> hash(const std::vector<F>& in) const {
    std::array<F, WIDTH> s{};
    size_t i = 0;
    while (i < in.size()) {
      size_t take = std::min<size_t>(RATE, in.size() - i);
      for (size_t j = 0; j < take; ++j) s[j] = in[i + j];
      permute(s);
      i += take;
    }
    std::array<F, DIGEST> d;
    std::copy(s.begin(), s.begin() + DIGEST, d.begin());
    return d;
  }
  // TruncatedPermutation<Perm,2,8,16>: perm(left || right)[0..8]
  // (circuit/src/ops/mmcs.rs:117-160).
  std::array<F, DIGEST> compress(const std::array<F, DIGEST>& l, const std::array<F, DIGEST>& r) const {
    std::array<F, WIDTH> s;
    std::copy(l.begin(), l.end(), s.begin());
    std::copy(r.begin(), r.end(), s.begin() + DIGEST);
    permute(s);
    std::array<F, DIGEST> d;
    std::copy(s.begin(), s.begin() + DIGEST, d.begin());
    return d;
  }
};

// DuplexChallenger<F, Perm, 16, 8> with the 0.6 prefix-free padding
// (recursion/src/challenger/circuit.rs:97-156 duplexing, :337-364 observe/sample,
//  :366-386 extension elements, :388-430 sample_bits / PoW).
template <class FP>
struct Challenger {
  using F = Fe<FP>;
  using EF = Fe4<FP>;
  const Poseidon2<FP>* perm;
  std::array<F, WIDTH> state{};
  std::vector<F> in_buf, out_buf;
  explicit Challenger(const Poseidon2<FP>* p) : perm(p) {}

  void duplexing() {
    size_t n = in_buf.size();
    for (size_t i = 0; i < n; ++i) state[i] = in_buf[i];
    in_buf.clear();
    if (n > 0) {
      for (size_t i = n; i < RATE; ++i) state[i] = F::zero();
      state[RATE] += F((uint64_t)n);
    }
    perm->permute(state);
    out_buf.assign(state.begin(), state.begin() + RATE);
  }
  void observe(F x) {
    out_buf.clear();
    in_buf.push_back(x);
    if (in_buf.size() == RATE) duplexing();
  }
  void observe_slice(const std::vector<F>& xs) { for (auto x : xs) observe(x); }
  template <size_t N> void observe_arr(const std::array<F, N>& xs) { for (auto x : xs) observe(x); }
  void observe_ext(const EF& x) { for (int i = 0; i < EF::deg(); ++i) observe(x.c[i]); }
  // observe_base_as_algebra_element: [v,0,0,0(,0)] (recursion/src/verifier/batch_stark.rs:521-523)
  void observe_base_as_ext(F v) { observe_ext(EF(v)); }
  F sample() {
    if (!in_buf.empty() || out_buf.empty()) duplexing();
    F x = out_buf.back();
    out_buf.pop_back();
    return x;
  }
  EF sample_ext() {
    EF e;
    for (int i = 0; i < EF::deg(); ++i) e.c[i] = sample();
    return e;
  }
  uint32_t sample_bits(int bits) { return sample().v & ((bits >= 32) ? ~0u : ((1u << bits) - 1)); }
  bool check_witness(int bits, F witness) {
    if (bits == 0) return true;
    observe(witness);
    return sample_bits(bits) == 0;
  }
  // Deterministic grind: smallest canonical witness (upstream searches in parallel and may
  // return any valid witness - SURVEY.md appendix A "PoW"; DESIGN.md states this choice).
  // `forced`: witnesses to use instead of searching, in grind order (tools/resolve_pins.py: a proof made elsewhere may
  // carry any valid witness - upstream searches in parallel -, and the rest of its transcript hangs off that choice)
  const std::vector<uint32_t>* forced = nullptr;
  size_t forced_at = 0;
  F grind(int bits) {
    if (bits == 0) return F::zero();
    if (forced && forced_at < forced->size()) {
      const F w((*forced)[forced_at++]);
      if (!check_witness(bits, w)) throw std::runtime_error("forced proof-of-work witness is not valid at this point of the transcript");
      return w;
    }
    for (uint32_t w = 0; w < FP::P; ++w) {
      Challenger c = *this;
      if (c.check_witness(bits, F(w))) {
        check_witness(bits, F(w));
        return F(w);
      }
    }
    throw std::runtime_error("grind failed");
  }
};

// ------------------------------------------------------------------ ZK randomness
// Keyed counter-based generator shared with the device (csrc/zk_rand.h is the device's statement of it): ChaCha with 8
// rounds (RFC 8439's block function with four double rounds) under a 256-bit key; input words 12..15 of a block are
// [counter, stream, nonce_lo, nonce_hi], nonce = proofs made so far.  Streams: (round << 20) | matrix, rounds 0 random,
// 1 main, 2 quotient, 4 permutation, 5 quotient masks.  Cell idx of a stream: rejection sampling on 31-bit words -
// words 2j, 2j + 1 (j = idx mod 8) of block idx / 8, then the words of fallback blocks
// [idx mod 2^32, stream | f << 24 | (idx >> 32) << 27], f = 1..7, the first value below p.
struct ZkStream {
  std::array<uint32_t, 8> key{};
  uint64_t nonce = 0;
  uint32_t stream = 0;
};
inline std::array<uint32_t, 16> zk_chacha8(const std::array<uint32_t, 8>& key, uint32_t w12, uint32_t w13, uint64_t nonce) {
  std::array<uint32_t, 16> in = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
  for (int i = 0; i < 8; ++i) in[4 + i] = key[i];
  in[12] = w12; in[13] = w13; in[14] = (uint32_t)nonce; in[15] = (uint32_t)(nonce >> 32);
  std::array<uint32_t, 16> x = in;
  auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
  auto quarter = [&](int a, int b, int c, int d) {
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
    x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
    x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
  };
  for (int dr = 0; dr < 4; ++dr) {
    for (int c = 0; c < 4; ++c) quarter(c, 4 + c, 8 + c, 12 + c);                                   // columns
    for (int c = 0; c < 4; ++c) quarter(c, 4 + (c + 1) % 4, 8 + (c + 2) % 4, 12 + (c + 3) % 4);     // diagonals
  }
  for (int i = 0; i < 16; ++i) x[i] += in[i];
  return x;
}
template <class FP>
Fe<FP> zk_rand(const ZkStream& s, uint64_t idx) {
  const auto first = zk_chacha8(s.key, (uint32_t)(idx >> 3), s.stream, s.nonce);
  const size_t j = idx & 7;
  for (size_t t = 0; t < 2; ++t) {
    const uint32_t v = first[2 * j + t] & 0x7FFFFFFFu;
    if (v < FP::P) return Fe<FP>(v);
  }
  uint32_t last = 0;
  for (uint32_t f = 1; f <= 7; ++f) {
    const auto more = zk_chacha8(s.key, (uint32_t)idx, s.stream | (f << 24) | ((uint32_t)(idx >> 32) << 27), s.nonce);
    for (uint32_t w : more) {
      last = w & 0x7FFFFFFFu;
      if (last < FP::P) return Fe<FP>(last);
    }
  }
  return Fe<FP>(last % FP::P);
}

// Row-major matrix of base elements.
template <class FP>
struct Matrix {
  using F = Fe<FP>;
  size_t h = 0, w = 0;
  std::vector<F> v;
  Matrix() = default;
  Matrix(size_t h_, size_t w_) : h(h_), w(w_), v(h_ * w_) {}
  F& at(size_t r, size_t c) { return v[r * w + c]; }
  const F& at(size_t r, size_t c) const { return v[r * w + c]; }
  std::vector<F> row(size_t r) const { return std::vector<F>(v.begin() + r * w, v.begin() + (r + 1) * w); }
};

// MerkleTreeMmcs over mixed-height matrices (recursion/src/pcs/mmcs.rs:355-425: group
// same-height matrices, tallest first, stable; concatenate their rows into one leaf
// preimage; circuit/src/ops/mmcs.rs:19-70,112-185: after compressing a level, if shorter
// matrices have that height, digest = compress(digest, hash(rows))).
template <class FP>
struct MerkleTree {
  using F = Fe<FP>;
  using Digest = std::array<F, DIGEST>;
  std::vector<const Matrix<FP>*> mats;  // commit order
  std::vector<std::vector<Digest>> layers;
  int log_max_h = 0, cap_height = 0;
  int arity = 2;                    // 4: MerkleTreeMmcs<.., 4, 8> over the width-32 permutation
  std::vector<Arity4Step> sched;    // arity 4: one entry per level
  // MerkleTreeHidingMmcs (recursion/src/pcs/mmcs.rs:315-413,430-510: "commits [row | salt] per matrix"): one salt matrix
  // (height x SALT_ELEMS) per committed matrix, in commit order; a leaf preimage is the concatenation, over the matrices of
  // the height class in tallest-first stable order, of [row | salt].  Empty: the plain MerkleTreeMmcs.
  std::vector<Matrix<FP>> salts;
  // cell (r, c) of the salt matrix of matrix k: cell r * elems + c of the stream (round << 20 | first_mat + k) of the
  // keyed generator above (the device's salts are the same values: csrc/zk_rand.h)
  struct SaltSpec { int elems = 0; ZkStream base; int round = 0; size_t first_mat = 0; };
  void draw_salts(const SaltSpec* sp) {
    salts.clear();
    if (!sp || sp->elems <= 0) return;
    for (size_t k = 0; k < mats.size(); ++k) {
      Matrix<FP> m(mats[k]->h, (size_t)sp->elems);
      ZkStream st = sp->base;
      st.stream = ((uint32_t)sp->round << 20) | (uint32_t)(sp->first_mat + k);
      for (size_t r = 0; r < m.h; ++r)
        for (size_t c = 0; c < m.w; ++c) m.at(r, c) = zk_rand<FP>(st, r * m.w + c);
      salts.push_back(std::move(m));
    }
  }
  std::vector<F> salted_row(size_t k, size_t i) const {
    auto r = mats[k]->row(i);
    if (!salts.empty()) { auto s = salts[k].row(i); r.insert(r.end(), s.begin(), s.end()); }
    return r;
  }

  std::vector<Digest> cap() const { return layers.back(); }

  static const Poseidon2W32<FP>& wide(const Poseidon2<FP>& p2) {
    if (!p2.w32) throw std::runtime_error("the arity-4 MMCS needs the constants of the width-32 permutation");
    return *p2.w32;
  }
  static Digest zero_digest() { Digest d; d.fill(F::zero()); return d; }

  // Arity 4 (recursion/src/pcs/mmcs.rs:866-1316 is the in-tree statement of what such a tree is: leaf = W32 sponge
  // over the rows of the tallest matrices; a level compresses 4 (or 2, zero-padded to 4) children; an injected
  // matrix enters as one more compression (node, its row digest, 0, 0); logical layers of 2 are padded to 4).
  static MerkleTree commit4(const Poseidon2<FP>& p2, const std::vector<const Matrix<FP>*>& mats, int cap_height,
                            const SaltSpec* salt = nullptr) {
    if (cap_height != 0) throw std::runtime_error("arity-4 MMCS: cap_height must be 0");
    const auto& w = wide(p2);
    MerkleTree t;
    t.mats = mats;
    t.arity = 4;
    t.draw_salts(salt);
    std::vector<size_t> order(mats.size()), heights;
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return mats[a]->h > mats[b]->h; });
    for (auto* m : mats) heights.push_back(m->h);
    const size_t hmax = mats[order[0]]->h;
    t.log_max_h = log2_strict(hmax);
    t.sched = arity4_schedule(heights, 1);
    auto rows_at = [&](size_t h, size_t i) {
      std::vector<F> cat;
      for (size_t k : order)
        if (mats[k]->h == h) {
          auto r = t.salted_row(k, i);
          cat.insert(cat.end(), r.begin(), r.end());
        }
      return cat;
    };
    std::vector<Digest> cur(padded_len(hmax, 4), zero_digest());
#pragma omp parallel for schedule(static) if (hmax >= 1024)
    for (size_t i = 0; i < hmax; ++i) cur[i] = w.hash(rows_at(hmax, i));
    t.layers.push_back(cur);
    for (const Arity4Step& st : t.sched) {
      std::vector<Digest> nxt(st.padded_next, zero_digest());
#pragma omp parallel for schedule(static) if (st.logical_next >= 1024)
      for (size_t i = 0; i < st.logical_next; ++i) {
        std::array<Digest, 4> c{zero_digest(), zero_digest(), zero_digest(), zero_digest()};
        for (int k = 0; k < st.step; ++k) c[k] = cur[(size_t)st.step * i + k];
        nxt[i] = w.compress4(c);
        if (st.inject_h) nxt[i] = w.compress4({nxt[i], w.hash(rows_at(st.inject_h, i)), zero_digest(), zero_digest()});
      }
      t.layers.push_back(nxt);
      cur = nxt;
    }
    return t;
  }
  // siblings of one level in ascending position, the opened node's own position left out
  // (`Proof = Vec<[F; 8]>`, grouped per level by set_arity4_opening_private_data, recursion/src/pcs/mmcs.rs:1413-1461)
  void open4(size_t index, std::vector<std::vector<F>>& opened, std::vector<Digest>& proof,
             std::vector<std::vector<F>>* salts_out = nullptr) const {
    opened.clear();
    proof.clear();
    for (auto* m : mats) opened.push_back(m->row(index >> (log_max_h - log2_strict(m->h))));
    open_salts(index, salts_out);
    size_t idx = index;
    for (size_t l = 0; l < sched.size(); ++l) {
      const size_t step = sched[l].step, pos = idx % step, base = idx - pos;
      for (size_t j = 0; j < step; ++j)
        if (j != pos) proof.push_back(layers[l][base + j]);
      idx /= step;
    }
  }
  static bool verify4(const Poseidon2<FP>& p2, const std::vector<Digest>& cap, int cap_height,
                      const std::vector<std::pair<size_t, size_t>>& dims, size_t index,
                      const std::vector<std::vector<F>>& opened, const std::vector<Digest>& proof,
                      const std::vector<std::vector<F>>* salts = nullptr) {
    if (cap_height != 0 || cap.size() != 1) return false;
    if (salts && salts->size() != dims.size()) return false;
    const auto& w = wide(p2);
    std::vector<size_t> order(dims.size()), heights;
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return dims[a].first > dims[b].first; });
    for (auto& d : dims) heights.push_back(d.first);
    const size_t hmax = dims[order[0]].first;
    if (index >= hmax) return false;
    const auto sched = arity4_schedule(heights, 1);
    auto cat_at = [&](size_t h) {
      std::vector<F> cat;
      for (size_t k : order)
        if (dims[k].first == h) {
          if (opened[k].size() != dims[k].second) throw std::runtime_error("opened width mismatch");
          cat.insert(cat.end(), opened[k].begin(), opened[k].end());
          if (salts) cat.insert(cat.end(), (*salts)[k].begin(), (*salts)[k].end());   // [row | salt] per matrix
        }
      return cat;
    };
    size_t want = 0;
    for (auto& st : sched) want += st.step - 1;
    if (proof.size() != want) return false;
    Digest d = w.hash(cat_at(hmax));
    size_t idx = index, at = 0;
    for (auto& st : sched) {
      const size_t pos = idx % st.step;
      std::array<Digest, 4> c{zero_digest(), zero_digest(), zero_digest(), zero_digest()};
      for (size_t j = 0; j < (size_t)st.step; ++j) c[j] = j == pos ? d : proof[at++];
      d = w.compress4(c);
      idx /= st.step;
      if (st.inject_h) d = w.compress4({d, w.hash(cat_at(st.inject_h)), zero_digest(), zero_digest()});
    }
    return idx == 0 && cap[0] == d;
  }

  static MerkleTree commit(const Poseidon2<FP>& p2, const std::vector<const Matrix<FP>*>& mats,
                           int cap_height, int arity = 2, const SaltSpec* salt = nullptr) {
    if (arity == 4) return commit4(p2, mats, cap_height, salt);
    MerkleTree t;
    t.mats = mats;
    t.cap_height = cap_height;
    t.draw_salts(salt);
    std::vector<size_t> order(mats.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](size_t a, size_t b) { return mats[a]->h > mats[b]->h; });
    size_t hmax = mats[order[0]]->h;
    t.log_max_h = log2_strict(hmax);
    if (cap_height > t.log_max_h) throw std::runtime_error("cap_height exceeds log2 of the tallest matrix");
    auto rows_at = [&](size_t h, size_t i) {
      std::vector<F> cat;
      for (size_t k : order)
        if (mats[k]->h == h) {
          auto r = t.salted_row(k, i);
          cat.insert(cat.end(), r.begin(), r.end());
        }
      return cat;
    };
    auto any_at = [&](size_t h) {
      for (auto* m : mats) if (m->h == h) return true;
      return false;
    };
    std::vector<Digest> cur(hmax);
    // rows and nodes of one layer are independent: OpenMP over them (cpu_baseline, bench.py)
#pragma omp parallel for schedule(static) if (hmax >= 1024)
    for (size_t i = 0; i < hmax; ++i) cur[i] = p2.hash(rows_at(hmax, i));
    t.layers.push_back(cur);
    while (cur.size() > (size_t(1) << cap_height)) {
      size_t nn = cur.size() / 2;
      std::vector<Digest> nxt(nn);
      bool inj = any_at(nn);
#pragma omp parallel for schedule(static) if (nn >= 1024)
      for (size_t i = 0; i < nn; ++i) {
        nxt[i] = p2.compress(cur[2 * i], cur[2 * i + 1]);
        if (inj) nxt[i] = p2.compress(nxt[i], p2.hash(rows_at(nn, i)));
      }
      t.layers.push_back(nxt);
      cur = nxt;
    }
    return t;
  }

  // open_batch(index): rows (index >> (log_max_h - log_h)) of every matrix in commit order,
  // sibling digests bottom-up.
  // the salt rows of the opened rows, per matrix in commit order (the first half of `Proof = (salts, siblings)`,
  // recursion/src/pcs/mmcs.rs:763-790)
  void open_salts(size_t index, std::vector<std::vector<F>>* salts_out) const {
    if (!salts_out) return;
    salts_out->clear();
    for (size_t k = 0; k < salts.size(); ++k) salts_out->push_back(salts[k].row(index >> (log_max_h - log2_strict(mats[k]->h))));
  }
  void open(size_t index, std::vector<std::vector<F>>& opened, std::vector<Digest>& proof,
            std::vector<std::vector<F>>* salts_out = nullptr) const {
    if (arity == 4) return open4(index, opened, proof, salts_out);
    opened.clear();
    proof.clear();
    open_salts(index, salts_out);
    for (auto* m : mats) opened.push_back(m->row(index >> (log_max_h - log2_strict(m->h))));
    for (int l = 0; l < log_max_h - cap_height; ++l) proof.push_back(layers[l][(index >> l) ^ 1]);
  }

  // verify_batch (recursion/src/pcs/mmcs.rs:319-426): dims = (height, width) per matrix in
  // commit order.
  static bool verify(const Poseidon2<FP>& p2, const std::vector<Digest>& cap, int cap_height,
                     const std::vector<std::pair<size_t, size_t>>& dims, size_t index,
                     const std::vector<std::vector<F>>& opened, const std::vector<Digest>& proof, int arity = 2,
                     const std::vector<std::vector<F>>* salts = nullptr) {
    if (arity == 4) return verify4(p2, cap, cap_height, dims, index, opened, proof, salts);
    if (salts && salts->size() != dims.size()) return false;
    std::vector<size_t> order(dims.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](size_t a, size_t b) { return dims[a].first > dims[b].first; });
    size_t hmax = dims[order[0]].first;
    int log_max = log2_strict(hmax);
    if ((int)proof.size() != log_max - cap_height) return false;
    auto cat_at = [&](size_t h, bool& any) {
      std::vector<F> cat;
      any = false;
      for (size_t k : order)
        if (dims[k].first == h) {
          any = true;
          if (opened[k].size() != dims[k].second) throw std::runtime_error("opened width mismatch");
          cat.insert(cat.end(), opened[k].begin(), opened[k].end());
          if (salts) cat.insert(cat.end(), (*salts)[k].begin(), (*salts)[k].end());   // [row | salt] per matrix (mmcs.rs:375-389)
        }
      return cat;
    };
    bool any;
    Digest d = p2.hash(cat_at(hmax, any));
    size_t idx = index;
    size_t h = hmax;
    for (auto& sib : proof) {
      d = (idx & 1) ? p2.compress(sib, d) : p2.compress(d, sib);
      idx >>= 1;
      h >>= 1;
      auto cat = cat_at(h, any);
      if (any) d = p2.compress(d, p2.hash(cat));
    }
    return idx < cap.size() && cap[idx] == d;
  }
};

}  // namespace orc
