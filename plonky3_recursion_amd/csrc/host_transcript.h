// Host-side serial pieces of the prover: DuplexChallenger, LogUp lookup layout, postcard writer.
// They are tiny and strictly sequential in the reference too.
//
//   DuplexChallenger<F, Perm, 16, 8> + 0.6 prefix-free padding
//        recursion/src/challenger/circuit.rs:97-156 (duplexing), :337-364 (observe / sample),
//        :366-386 (extension elements), :388-430 (sample_bits, PoW check)
//   lookup packing budget      circuit-prover/src/batch_stark_prover.rs:925-941
//   proof field order          recursion/src/types/proof.rs:403-409,452-457,527-534,585-589,
//                              recursion/src/pcs/fri/targets.rs:104-110
#pragma once
#include "proof_layout.h"
#include <cstring>
#include <algorithm>
#include <vector>

#include "air_device.hip.h"
#include "context.h"
#include "host_poseidon2_simd.h"

namespace p3r {

template <class PP, int DC = 4>
struct HostChallenger {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;   // the challenge field: DC sampled / observed coefficients per element
  const uint32_t* rc;  // Montgomery
  F state[P2_WIDTH];
  std::vector<F> in_buf, out_buf;
  explicit HostChallenger(const uint32_t* rc_) : rc(rc_) {
    for (auto& s : state) s = F::zero();
  }
  void duplexing() {
    size_t n = in_buf.size();
    for (size_t i = 0; i < n; ++i) state[i] = in_buf[i];
    in_buf.clear();
    if (n > 0) {
      for (size_t i = n; i < (size_t)P2_RATE; ++i) state[i] = F::zero();
      state[P2_RATE] += F::from_canonical((uint32_t)n);
    }
    host_permute<PP>(state, rc);
    out_buf.assign(state, state + P2_RATE);
  }
  void observe(F x) {
    out_buf.clear();
    in_buf.push_back(x);
    if (in_buf.size() == (size_t)P2_RATE) duplexing();
  }
  void observe_ext(const E& e) { for (int i = 0; i < DC; ++i) observe(e.c[i]); }
  void observe_base_as_ext(uint64_t v) { observe_ext(E::from_base(F::from_u64(v))); }
  void observe_digest_canonical(const uint32_t* d) { for (int i = 0; i < P2_DIGEST; ++i) observe(F::from_canonical(d[i])); }
  F sample() {
    if (!in_buf.empty() || out_buf.empty()) duplexing();
    F x = out_buf.back();
    out_buf.pop_back();
    return x;
  }
  E sample_ext() {
    E e;
    for (int i = 0; i < DC; ++i) e.c[i] = sample();
    return e;
  }
  uint32_t sample_bits(int bits) { return sample().to_canonical() & ((1u << bits) - 1); }
  bool check_witness(int bits, F w) {
    if (bits == 0) return true;
    observe(w);
    return sample_bits(bits) == 0;
  }
};

// ---- lookup layout (same rule as DESIGN.md "LogUp"): greedy same-bus packing under the
// degree budget 2^log_chunks + 1 - is_zk; log_chunks = log2_ceil(max(degree + is_zk, 2) - 1)
// (circuit-prover/src/batch_stark_prover.rs:931-939).
inline std::vector<int> interaction_mult_degrees(const AirParams& a) {
  std::vector<int> d;
  switch (a.kind) {
    case AIR_CONST: d = {1}; break;
    case AIR_PUBLIC: d.assign(a.lanes, 1); break;
    case AIR_RECOMPOSE: d.assign(a.lanes * (1 + (a.coeff_lookups ? a.ext_d : 0)), 1); break;
    case AIR_ALU:
      for (int l = 0; l < a.lanes; ++l) { d.push_back(2); d.push_back(1); d.push_back(2); d.push_back(1); }
      for (int t = 1; t < a.horner_k; ++t) { d.push_back(1); d.push_back(1); }
      break;
    case AIR_POSEIDON2:
      if (a.ext_d == 4) { d = {2, 2, 2, 2, 1, 1, 2}; break; }
      // compact D1: 8 rate sends (in_ctl * not_merkle), 8 output receives, the accumulator send
      d.assign(8, 2); d.insert(d.end(), 8, 1); d.push_back(2);
      break;
    case AIR_POSEIDON2_W32: d.assign(8 + 6 + 2, 1); break;   // in_ctl sends, out_ctl receives, the two direction-bit reads
  }
  return d;
}
struct LookupLayout {
  // pair: interactions per packed group minus one (0 singletons, 1 pairs, 2 triples - the budget of a ZK configuration);
  // the last group may be smaller
  int n_interactions = 0, n_groups = 0, pair = 0, log_chunks = 0;
  int aux_width() const { return n_groups ? n_groups + 1 : 0; }
};
inline LookupLayout lookup_layout(const AirParams& a, int is_zk = 0) {
  auto md = interaction_mult_degrees(a);
  auto gdeg = [&](int first, int K) {
    int deg = 1 + K;
    for (int k = 0; k < K; ++k) deg = std::max(deg, md[first + k] + K - 1);
    return deg;
  };
  int max_deg = 0;
  if (a.kind == AIR_ALU || a.kind == AIR_POSEIDON2 || a.kind == AIR_POSEIDON2_W32) max_deg = 3;
  for (size_t i = 0; i < md.size(); ++i) max_deg = std::max(max_deg, gdeg((int)i, 1));
  max_deg = std::max(max_deg + is_zk, 2);
  LookupLayout L;
  L.n_interactions = (int)md.size();
  while ((1 << L.log_chunks) < max_deg - 1) ++L.log_chunks;
  const int budget = (1 << L.log_chunks) + 1 - is_zk;
  // greedy packing; the device kernels support the two shapes it produces for these AIRs
  std::vector<int> sizes;
  int first = 0, K = 0;
  for (int i = 0; i < (int)md.size(); ++i) {
    if (K > 0 && (a.lookup_unpacked || gdeg(first, K + 1) > budget)) { sizes.push_back(K); first = i; K = 1; }
    else { if (K == 0) first = i; ++K; }
  }
  if (K) sizes.push_back(K);
  L.n_groups = (int)sizes.size();
  // uniform groups of G = 1, 2 or 3 interactions, the last one possibly smaller: the shapes the greedy rule
  // produces for these AIRs, and the ones the device kernels take
  const int G = sizes.empty() ? 1 : sizes[0];
  bool uniform = G >= 1 && G <= 3;
  for (size_t g = 0; g < sizes.size(); ++g)
    uniform = uniform && (sizes[g] == G || (g + 1 == sizes.size() && sizes[g] >= 1 && sizes[g] < G));
  if (!uniform) fail(P3R_EUNSUPPORTED, "lookup packing shape not supported by the device kernels");
  L.pair = G - 1;
  return L;
}

// FRI folding schedule: log2 arity of commit phase `phase` at height 2^log_cur.  Rule (no override):
// fold as far as max_log_arity allows without passing the next roll-in height or the final height.
// With an explicit schedule (p3r_config.fri_log_arities) the entry is used if it is legal; -1 otherwise.
inline int fri_log_arity(const std::vector<uint8_t>& schedule, size_t phase, int max_log_arity, int log_cur, int log_final,
                         int log_next_input) {
  int limit = log_cur - log_final;
  if (log_next_input >= 0 && log_next_input < log_cur) limit = std::min(limit, log_cur - log_next_input);
  if (schedule.empty()) return std::max(std::min(max_log_arity, limit), 1);
  if (phase >= schedule.size()) return -1;
  const int la = schedule[phase];
  return (la >= 1 && la <= limit && la <= max_log_arity) ? la : -1;
}

// p2_width: the field's width-16 Poseidon2 table (perm columns + 2); its width-32 table has p2w_width columns
inline int air_width_of(const AirParams& a, int p2_width, int p2w_width = 0) {
  switch (a.kind) {
    case AIR_CONST: return a.ext_d;
    case AIR_PUBLIC: return a.lanes * a.ext_d;
    case AIR_ALU: return (a.lanes * 4 + (a.horner_k - 1) / 2 + 2 * (a.horner_k - 1) + 1) * a.ext_d;
    case AIR_POSEIDON2: return p2_width;
    case AIR_RECOMPOSE: return a.lanes * a.ext_d;
    case AIR_POSEIDON2_W32: return p2w_width;
  }
  return 0;
}
inline int air_prep_width_of(const AirParams& a) {
  switch (a.kind) {
    case AIR_CONST: return 2;
    case AIR_PUBLIC: return a.lanes * 2;
    case AIR_ALU: return a.lanes * 13 + 7 * (a.horner_k - 1);
    case AIR_POSEIDON2: return a.ext_d == 4 ? 24 : kP2D1PrepWidth;
    case AIR_RECOMPOSE: return a.lanes * (2 + (a.coeff_lookups ? 2 * a.ext_d : 0));
    case AIR_POSEIDON2_W32: return kP2WPrepWidth;
  }
  return 0;
}
inline bool air_uses_next(const AirParams& a) { return a.kind == AIR_ALU || a.kind == AIR_POSEIDON2 || a.kind == AIR_POSEIDON2_W32; }

// ---- postcard writer. Field elements are written as the Montgomery word by default
// (p3-monty-31's serde form), or canonical when `canonical` is set.
template <class PP, int DC = 4>
struct ProofWriter {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  // `out` is grown in large steps and written through a cursor (a proof is ~10^5 varints; a
  // push_back per byte made serialisation a visible part of small proofs); take() trims it.
  std::vector<uint8_t> out;
  size_t len = 0;
  bool canonical = false;
  uint8_t* room(size_t n) {
    if (len + n > out.size()) out.resize(std::max(out.size() * 2, len + n + 4096));
    return out.data() + len;
  }
  void byte(uint8_t b) { *room(1) = b; ++len; }
  void varint(uint64_t v) {
    uint8_t* p = room(10);
    size_t i = 0;
    while (v >= 0x80) { p[i++] = (uint8_t)(v | 0x80); v >>= 7; }
    p[i++] = (uint8_t)v;
    len += i;
  }
  std::vector<uint8_t> take() {
    out.resize(len);
    return std::move(out);
  }
  void fe(F x) {
    const uint32_t v = canonical ? x.to_canonical() : x.v;
    if (v < (1u << 28)) return varint(v);
    uint8_t* p = room(5);  // 7 of 8 field elements: five bytes, no loop
    p[0] = (uint8_t)(v | 0x80);
    p[1] = (uint8_t)((v >> 7) | 0x80);
    p[2] = (uint8_t)((v >> 14) | 0x80);
    p[3] = (uint8_t)((v >> 21) | 0x80);
    p[4] = (uint8_t)(v >> 28);
    len += 5;
  }
  // A run of field elements given as Montgomery words (rows of a query answer, digests, extension
  // elements).  On x86-64 with BMI2 the five bytes of a varint are one bit-deposit and one store.
  void words(const uint32_t* mont, size_t n) {
#if defined(P3R_HOST_AVX512)
    static const bool bmi2 = __builtin_cpu_supports("bmi2") && !(getenv("P3R_HOST_SIMD") && getenv("P3R_HOST_SIMD")[0] == '0');
    if (bmi2 && !canonical) {
      len += words_bmi2(room(5 * n + 8), mont, n);
      return;
    }
#endif
    for (size_t i = 0; i < n; ++i) fe(F::raw(mont[i]));
  }
#if defined(P3R_HOST_AVX512)
  __attribute__((target("bmi2"))) static size_t words_bmi2(uint8_t* p, const uint32_t* v, size_t n) {
    uint8_t* const p0 = p;
    for (size_t i = 0; i < n; ++i) {
      const uint32_t x = v[i];
      // 7 bits per byte; continuation bits on the bytes below the last non-zero group
      const uint64_t spread = _pdep_u64(x, 0x0000000F7F7F7F7Full);
      const int bytes = x < (1u << 7) ? 1 : x < (1u << 14) ? 2 : x < (1u << 21) ? 3 : x < (1u << 28) ? 4 : 5;
      const uint64_t cont = 0x0000000080808080ull & ((uint64_t(1) << (8 * (bytes - 1))) - 1);
      const uint64_t enc = spread | cont;
      std::memcpy(p, &enc, 8);  // 8-byte store, `bytes` of them count (the caller reserved the slack)
      p += bytes;
    }
    return (size_t)(p - p0);
  }
#endif
  static_assert(sizeof(E) == 4 * DC && sizeof(F) == 4, "extension elements are DC contiguous words");
  void ef(const E& e) { words(&e.c[0].v, DC); }
  void vec_ef(const std::vector<E>& v) {
    varint(v.size());
    if (!v.empty()) words(&v[0].c[0].v, DC * v.size());
  }
  void digest_mont(const uint32_t* d) { words(d, P2_DIGEST); }
  void cap_mont(const std::vector<uint32_t>& cap) {
    varint(cap.size() / P2_DIGEST);
    for (size_t i = 0; i < cap.size(); i += P2_DIGEST) digest_mont(&cap[i]);
  }
};

}  // namespace p3r
