// Native verifier for the proofs prove_impl.hip.h produces: `verify_batch` behind
// `BatchStarkProver::verify_all_tables` (circuit-prover/src/batch_stark_prover.rs:1230-1268,
// 1649-1727).  Host code: verification is a few thousand permutations and is serial in the
// reference too.  It replays the prover's transcript and checks, following the in-tree circuit
// verifier (the only in-tree statement of what p3-batch-stark / p3-fri 0.6 verify):
//   transcript order, domains, opening points      recursion/src/verifier/batch_stark.rs:521-852
//   folded constraints * 1/Z_H == quotient(zeta)   :886-1021, recursion/src/verifier/quotient.rs:60-
//   selectors at zeta                               recursion/src/pcs/fri/targets.rs:868-908
//   LogUp constraints, global sum of terminals      batch_stark.rs:1026-1112
//   MMCS openings (mixed heights, injection, cap)   recursion/src/pcs/mmcs.rs:319-426, circuit/src/ops/mmcs.rs:81-209
//   FRI: reduced openings, folds, roll-ins, final polynomial, proofs of work
//                                                   recursion/src/pcs/fri/verifier.rs:424-465,562-781,887-981,1068-1356
//   ZK (HidingFriPcs, p3r_config.zk)                batch_stark.rs:424-428 (randomisation presence), :487-490,536 (degree
//                                                   bits), :623-661 (random commitment, its round), :701-735 (quotient
//                                                   domains), :855-864,1116-1260 and pcs/fri/targets.rs:1076-1130 (the
//                                                   opening proof's random opened values, merged into every point)
// The AIR statements are the ones the quotient kernel evaluates (air_device.hip.h), instantiated
// over the extension field at zeta.
#pragma once
#include <array>
#include <cstdarg>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/p3r.h"
#include "host_transcript.h"

namespace p3r {

struct VerifyFailure : std::runtime_error {
  using std::runtime_error::runtime_error;
};
[[noreturn]] inline void vfail(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw VerifyFailure(buf);
}

// ---- postcard reader (mirror of ProofWriter)
template <class PP, int DC = 4>
struct ProofReader {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  const uint8_t* p;
  const uint8_t* end;
  bool canonical;
  uint8_t byte() {
    if (p >= end) vfail("proof truncated");
    return *p++;
  }
  bool flag() {  // Option / bool tag: postcard admits 0 and 1 only
    const uint8_t b = byte();
    if (b > 1) vfail("invalid option tag %u", b);
    return b != 0;
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      uint8_t b = byte();
      if (shift == 63 && b > 1) vfail("varint overflows 64 bits");  // postcard: the tenth byte carries one bit
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) {
        if (b == 0 && shift > 0) vfail("non-canonical varint");
        return v;
      }
    }
    vfail("malformed varint");
  }
  size_t len(size_t max) {
    uint64_t v = varint();
    if (v > max) vfail("length %llu exceeds the bound %zu", (unsigned long long)v, max);
    return (size_t)v;
  }
  F fe() {
    uint64_t v = varint();
    if (v >= PP::P) vfail("field element out of range");
    return canonical ? F::from_canonical((uint32_t)v) : F::raw((uint32_t)v);
  }
  E ef() { E e; for (int i = 0; i < DC; ++i) e.c[i] = fe(); return e; }
  std::vector<E> vec_ef(size_t max = 1u << 16) {
    std::vector<E> v(len(max));
    for (auto& e : v) e = ef();
    return v;
  }
  using Digest = std::array<F, P2_DIGEST>;
  Digest digest() { Digest d; for (auto& x : d) x = fe(); return d; }
  std::vector<Digest> cap() {
    std::vector<Digest> c(len(1u << 16));
    for (auto& d : c) d = digest();
    return c;
  }
};

template <class PP, int DC = 4>
struct ParsedProof {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  using Digest = std::array<F, P2_DIGEST>;
  using Cap = std::vector<Digest>;
  Cap main_cap, quot_cap;
  std::optional<Cap> perm_cap, rand_cap;
  struct Inst {
    std::vector<E> main_local, prep_local, prep_next, perm_local, perm_next;
    std::optional<std::vector<E>> main_next, random;
    std::vector<std::vector<E>> chunks;
  };
  // HidingFriPcs::Proof = (OpenedValues<Challenge>, FriProof): rounds -> matrices -> points -> random codeword values
  std::vector<std::vector<std::vector<std::vector<E>>>> fri_random;
  std::vector<Inst> insts;
  std::vector<Cap> commit_caps;
  std::vector<F> commit_pow;
  // `salts`: MerkleTreeHidingMmcs - the opening proof is the tuple (per-matrix salts, sibling digests)
  // (SaltedMmcsProof, recursion/src/pcs/mmcs.rs:763-768); empty under the plain MMCS
  struct QRound { std::vector<std::vector<F>> rows; std::vector<Digest> path; std::vector<std::vector<F>> salts; };
  struct QPhase { int la; std::vector<E> sibs; std::vector<Digest> path; std::vector<std::vector<F>> salts; };
  struct Query { std::vector<QRound> rounds; std::vector<QPhase> phases; };
  std::vector<Query> queries;
  std::vector<E> final_poly;
  F query_pow;
  std::vector<std::optional<E>> terminals;
  std::vector<int> degree_bits;
};

// `consumed`: when given, trailing bytes are allowed and the length of the BatchProof is returned
// (the outer BatchStarkProof appends its metadata after it, batch_stark_prover.rs:610-636).
template <class PP, int DC = 4>
// `zk`: the proof type is the hiding PCS's - a property of the configuration (SC::Pcs in the reference), not of the bytes.
// `salted`: the MMCSs are hiding ones (p3r_config.mmcs_salt_elems > 0): every opening proof carries its salts first.
ParsedProof<PP, DC> parse_proof(const uint8_t* bytes, size_t n, bool canonical, size_t* consumed = nullptr,
                            const ProofLayout& PL = ProofLayout{}, bool zk = false, bool salted = false) {
  ProofReader<PP, DC> R{bytes, bytes + n, canonical};
  ParsedProof<PP, DC> P;
  auto read_commitments = [&] {
    P.main_cap = R.cap();
    if (R.flag()) P.perm_cap = R.cap();
    P.quot_cap = R.cap();
    if (R.flag()) P.rand_cap = R.cap();
  };
  auto read_opened = [&] {
    P.insts.resize(R.len(64));
    for (auto& in : P.insts) {
      for (int f = 0; f < 8; ++f) {
        switch (PL.opened[f]) {
          case 0: in.main_local = R.vec_ef(); break;
          case 1: if (R.flag()) in.main_next = R.vec_ef(); break;
          case 2: if (!R.flag()) vfail("preprocessed_local missing"); in.prep_local = R.vec_ef(); break;
          case 3: if (!R.flag()) vfail("preprocessed_next missing"); in.prep_next = R.vec_ef(); break;
          case 4:
            in.chunks.resize(R.len(8));
            for (auto& c : in.chunks) c = R.vec_ef();
            break;
          case 5: if (R.flag()) in.random = R.vec_ef(); break;
          case 6: in.perm_local = R.vec_ef(); break;
          default: in.perm_next = R.vec_ef(); break;
        }
      }
    }
  };
  auto read_queries = [&] {
    P.queries.resize(R.len(1024));
    for (auto& q : P.queries) {
      q.rounds.resize(R.len(8));
      for (auto& r : q.rounds) {
        r.rows.resize(R.len(256));
        for (auto& row : r.rows) {
          row.resize(R.len(1u << 16));
          for (auto& x : row) x = R.fe();
        }
        if (salted) {
          r.salts.resize(R.len(256));
          for (auto& sl : r.salts) { sl.resize(R.len(64)); for (auto& x : sl) x = R.fe(); }
        }
        r.path.resize(R.len(64));
        for (auto& d : r.path) d = R.digest();
      }
      q.phases.resize(R.len(64));
      for (auto& ph : q.phases) {
        ph.la = R.byte();
        ph.sibs.resize(R.len(16));
        for (auto& e : ph.sibs) e = R.ef();
        if (salted) {
          ph.salts.resize(R.len(4));
          for (auto& sl : ph.salts) { sl.resize(R.len(64)); for (auto& x : sl) x = R.fe(); }
        }
        ph.path.resize(R.len(64));
        for (auto& d : ph.path) d = R.digest();
      }
    }
  };
  auto read_fri = [&] {
    if (zk) {
      P.fri_random.resize(R.len(8));
      for (auto& rd : P.fri_random) {
        rd.resize(R.len(256));
        for (auto& m : rd) {
          m.resize(R.len(2));
          for (auto& pt : m) pt = R.vec_ef(16);
        }
      }
    }
    for (int f = 0; f < 5; ++f) {
      switch (PL.fri[f]) {
        case 0:
          P.commit_caps.resize(R.len(64));
          for (auto& c : P.commit_caps) c = R.cap();
          break;
        case 1:
          P.commit_pow.resize(R.len(64));
          for (auto& w : P.commit_pow) w = R.fe();
          break;
        case 2: read_queries(); break;
        case 3: P.final_poly = R.vec_ef(); break;
        default: P.query_pow = R.fe(); break;
      }
    }
  };
  for (int f = 0; f < 5; ++f) {
    switch (PL.batch[f]) {
      case 0: read_commitments(); break;
      case 1: read_opened(); break;
      case 2: read_fri(); break;
      case 3:
        P.terminals.resize(R.len(64));
        for (auto& t : P.terminals)
          if (R.flag()) t = R.ef();
        break;
      default:
        P.degree_bits.resize(R.len(64));
        for (auto& d : P.degree_bits) d = (int)R.len(40);
        break;
    }
  }
  if (consumed) *consumed = (size_t)(R.p - bytes);
  else if (R.p != R.end) vfail("%zu trailing bytes after the proof", (size_t)(R.end - R.p));
  return P;
}

// ---- framing-only walk of a BatchProof: what a parent node of the aggregation tree needs before it can
// hand a child to its verifier circuit (the postcard framing is sound, every field element is in range) -
// no containers are built.  Runs of field elements take the five-byte fast path (a Montgomery word is
// >= 2^28 fifteen times out of sixteen): one 8-byte load, one bit-extract, one compare.
template <class PP>
struct ProofSkimmer {
  const uint8_t* p;
  const uint8_t* end;
  uint8_t byte() {
    if (p >= end) vfail("proof truncated");
    return *p++;
  }
  bool flag() {
    const uint8_t b = byte();
    if (b > 1) vfail("invalid option tag %u", b);
    return b != 0;
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      uint8_t b = byte();
      if (shift == 63 && b > 1) vfail("varint overflows 64 bits");  // postcard: the tenth byte carries one bit
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) {
        if (b == 0 && shift > 0) vfail("non-canonical varint");
        return v;
      }
    }
    vfail("malformed varint");
  }
  size_t len(size_t max) {
    uint64_t v = varint();
    if (v > max) vfail("length %llu exceeds the bound %zu", (unsigned long long)v, max);
    return (size_t)v;
  }
  void fes_scalar(size_t k) {
    for (size_t i = 0; i < k; ++i)
      if (varint() >= PP::P) vfail("field element out of range");
  }
#if defined(P3R_HOST_AVX512)
  __attribute__((target("bmi2"))) void fes_bmi2(size_t k) {
    const uint8_t* q = p;
    size_t i = 0;
    constexpr uint64_t M = 0x8080808080ull, T = 0x0080808080ull, X = 0x7F7F7F7F7Full;
    while (i < k) {
      // four at a time while all four are five-byte words (their positions are then known up front)
      while (i + 4 <= k && q + 23 <= end) {
        uint64_t x0, x1, x2, x3;
        std::memcpy(&x0, q, 8); std::memcpy(&x1, q + 5, 8); std::memcpy(&x2, q + 10, 8); std::memcpy(&x3, q + 15, 8);
        const bool five = ((x0 & M) == T) & ((x1 & M) == T) & ((x2 & M) == T) & ((x3 & M) == T) &
                          (((x0 >> 32) & 0x7F) != 0) & (((x1 >> 32) & 0x7F) != 0) & (((x2 >> 32) & 0x7F) != 0) &
                          (((x3 >> 32) & 0x7F) != 0);
        if (!five) break;
        if ((_pext_u64(x0, X) >= PP::P) | (_pext_u64(x1, X) >= PP::P) | (_pext_u64(x2, X) >= PP::P) |
            (_pext_u64(x3, X) >= PP::P))
          vfail("field element out of range");
        q += 20;
        i += 4;
      }
      // a shorter word among the next four (one word in sixteen is below 2^28): one at a time, then retry
      for (int j = 0; j < 4 && i < k; ++j, ++i) {
        uint64_t x = 0;
        if (q + 8 <= end) std::memcpy(&x, q, 8);
        if ((x & M) == T && ((x >> 32) & 0x7F)) {  // five bytes, canonical
          if (_pext_u64(x, X) >= PP::P) vfail("field element out of range");
          q += 5;
        } else {
          p = q;
          if (varint() >= PP::P) vfail("field element out of range");
          q = p;
        }
      }
    }
    p = q;
  }
#endif
  void fes(size_t k) {
#if defined(P3R_HOST_AVX512)
    static const bool bmi2 = __builtin_cpu_supports("bmi2") && !(getenv("P3R_HOST_SIMD") && getenv("P3R_HOST_SIMD")[0] == '0');
    if (bmi2) return fes_bmi2(k);
#endif
    fes_scalar(k);
  }
  int dc = 4;   // words per extension element (the challenge degree)
  void vec_ef(size_t max = 1u << 16) { fes((size_t)dc * len(max)); }
  void cap() { fes(P2_DIGEST * len(1u << 16)); }
};

// Same grammar as parse_proof; returns the length of the BatchProof at the head of `bytes`.
template <class PP>
size_t skim_proof(const uint8_t* bytes, size_t n, const ProofLayout& PL = ProofLayout{}, int dc = 4, bool zk = false, bool salted = false) {
  ProofSkimmer<PP> R{bytes, bytes + n};
  R.dc = dc;
  auto opened = [&] {
    const size_t ni = R.len(64);
    for (size_t i = 0; i < ni; ++i)
      for (int f = 0; f < 8; ++f) {
        switch (PL.opened[f]) {
          case 0: R.vec_ef(); break;
          case 1: if (R.flag()) R.vec_ef(); break;
          case 2: if (!R.flag()) vfail("preprocessed_local missing"); R.vec_ef(); break;
          case 3: if (!R.flag()) vfail("preprocessed_next missing"); R.vec_ef(); break;
          case 4: { const size_t nc = R.len(8); for (size_t c = 0; c < nc; ++c) R.vec_ef(); break; }
          case 5: if (R.flag()) R.vec_ef(); break;
          default: R.vec_ef(); break;
        }
      }
  };
  auto queries = [&] {
    const size_t nq = R.len(1024);
    for (size_t q = 0; q < nq; ++q) {
      const size_t nr = R.len(8);
      for (size_t r = 0; r < nr; ++r) {
        const size_t rows = R.len(256);
        for (size_t k = 0; k < rows; ++k) R.fes(R.len(1u << 16));
        if (salted) { const size_t ns = R.len(256); for (size_t k = 0; k < ns; ++k) R.fes(R.len(64)); }
        R.fes(P2_DIGEST * R.len(64));
      }
      const size_t nph = R.len(64);
      for (size_t k = 0; k < nph; ++k) {
        (void)R.byte();
        R.fes((size_t)dc * R.len(16));
        if (salted) { const size_t ns = R.len(4); for (size_t k = 0; k < ns; ++k) R.fes(R.len(64)); }
        R.fes(P2_DIGEST * R.len(64));
      }
    }
  };
  auto fri = [&] {
    if (zk) {
      const size_t nr = R.len(8);
      for (size_t r = 0; r < nr; ++r) {
        const size_t nm = R.len(256);
        for (size_t m = 0; m < nm; ++m) {
          const size_t np = R.len(2);
          for (size_t k = 0; k < np; ++k) R.vec_ef(16);
        }
      }
    }
    for (int f = 0; f < 5; ++f) {
      switch (PL.fri[f]) {
        case 0: { const size_t nc = R.len(64); for (size_t c = 0; c < nc; ++c) R.cap(); break; }
        case 1: R.fes(R.len(64)); break;
        case 2: queries(); break;
        case 3: R.vec_ef(); break;
        default: R.fes(1); break;
      }
    }
  };
  for (int f = 0; f < 5; ++f) {
    switch (PL.batch[f]) {
      case 0:
        R.cap();
        if (R.flag()) R.cap();
        R.cap();
        if (R.flag()) R.cap();
        break;
      case 1: opened(); break;
      case 2: fri(); break;
      case 3: { const size_t nt = R.len(64); for (size_t t = 0; t < nt; ++t) if (R.flag()) R.fes(dc); break; }
      default: { const size_t nd = R.len(64); for (size_t d = 0; d < nd; ++d) (void)R.len(40); break; }
    }
  }
  return (size_t)(R.p - bytes);
}

// The metadata fields that follow the inner BatchProof (BatchStarkProof, batch_stark_prover.rs:610-636).
template <class PP>
void parse_batch_stark_meta(const uint8_t* bytes, size_t len, bool canonical, const ProofLayout& PL, int dc,
                            p3r_batch_stark_meta* M, bool zk = false, bool salted = false) {
  std::memset(M, 0, sizeof *M);
  M->proof_len = skim_proof<PP>(bytes, len, PL, dc, zk, salted);
  ProofSkimmer<PP> R{bytes + M->proof_len, bytes + len};
  auto u32 = [&](const char* what) {
    const uint64_t v = R.varint();
    if (v > 0xFFFFFFFFull) vfail("%s does not fit 32 bits", what);
    return (uint32_t)v;
  };
  auto fe = [&]() -> uint32_t {
    const uint64_t v = R.varint();
    if (v >= PP::P) vfail("field element out of range");
    return canonical ? (uint32_t)v : Fp<PP>::raw((uint32_t)v).to_canonical();
  };
  auto str = [&](char (&dst)[64]) {  // NpoTypeId(String): valid UTF-8 (serde rejects anything else), no NUL (the C struct ends there)
    const size_t k = R.len(63);
    if ((size_t)(R.end - R.p) < k) vfail("proof metadata truncated");
    for (size_t i = 0; i < k;) {
      const uint8_t b = R.p[i];
      size_t n = b < 0x80 ? 1 : (b >> 5) == 6 ? 2 : (b >> 4) == 14 ? 3 : (b >> 3) == 30 ? 4 : 0;
      bool ok = n != 0 && b != 0 && i + n <= k;
      uint32_t cp = n == 1 ? b : n == 2 ? (b & 0x1F) : n == 3 ? (b & 0x0F) : (b & 0x07);
      for (size_t j = 1; ok && j < n; ++j) {
        ok = (R.p[i + j] >> 6) == 2;
        cp = (cp << 6) | (R.p[i + j] & 0x3F);
      }
      // shortest form only, no surrogates, nothing beyond U+10FFFF
      if (ok && ((n == 2 && cp < 0x80) || (n == 3 && (cp < 0x800 || (cp >= 0xD800 && cp < 0xE000))) ||
                 (n == 4 && (cp < 0x10000 || cp > 0x10FFFF))))
        ok = false;
      if (!ok) vfail("invalid UTF-8 in an operation type name");
      i += n;
    }
    std::memcpy(dst, R.p, k);
    dst[k] = 0;
    R.p += k;
  };
  // TablePacking { public_lanes, alu_lanes, npo_lanes, min_trace_height, horner_packed_steps } (packing.rs:9-27)
  M->public_lanes = u32("public_lanes");
  M->alu_lanes = u32("alu_lanes");
  M->n_npo_lanes = (uint32_t)R.len(P3R_META_MAX_NPO);
  for (uint32_t i = 0; i < M->n_npo_lanes; ++i) { str(M->npo_lanes[i].op_type); M->npo_lanes[i].lanes = u32("npo lanes"); }
  M->min_trace_height = u32("min_trace_height");
  M->horner_packed_steps = u32("horner_packed_steps");
  for (int i = 0; i < 3; ++i) M->rows[i] = R.varint();  // RowCounts([usize; 3]): no length prefix
  M->alu_variant = u32("alu_variant");
  if (M->alu_variant > 1) vfail("unknown AirVariant %u", M->alu_variant);
  M->ext_degree = u32("ext_degree");
  M->has_w_binomial = R.flag();
  if (M->has_w_binomial) M->w_binomial = fe();
  M->alu_quintic_trinomial = R.flag();
  M->n_non_primitives = (uint32_t)R.len(P3R_META_MAX_NPO);
  for (uint32_t i = 0; i < M->n_non_primitives; ++i) {
    p3r_npo_table_entry& e = M->non_primitives[i];
    str(e.op_type);
    e.rows = R.varint();
    e.lanes = u32("npo lanes");
    e.n_public_values = (uint32_t)R.len(8);
    for (uint32_t k = 0; k < e.n_public_values; ++k) e.public_values[k] = fe();
    e.air_variant = u32("air_variant");
    if (e.air_variant > 1) vfail("unknown AirVariant %u", e.air_variant);
  }
  M->has_stark_common = R.flag();
  if (M->has_stark_common) {
    M->cap_len = (uint32_t)R.len(P3R_META_MAX_CAP);
    for (uint32_t i = 0; i < 8 * M->cap_len; ++i) M->commitment[i] = fe();
    const size_t n_inst = R.len(P3R_META_MAX_INSTANCES);
    for (size_t i = 0; i < n_inst; ++i) {
      if (!R.flag()) continue;
      (void)R.varint();  // matrix index
      M->preprocessed_widths[M->n_instances] = u32("preprocessed width");
      M->degree_bits[M->n_instances] = (uint32_t)R.len(40);
      M->n_instances++;
    }
    const size_t n_map = R.len(P3R_META_MAX_INSTANCES);
    for (size_t i = 0; i < n_map; ++i) (void)R.varint();  // matrix_to_instance
  }
  if (R.p != R.end) vfail("%zu trailing bytes after the proof metadata", (size_t)(R.end - R.p));
  // structural invariants a derived Deserialize bypasses (batch_stark_prover.rs:666-681, packing.rs:140-161)
  const uint32_t d = M->ext_degree;
  if (!(d == 1 || d == 2 || d == 4 || d == 5 || d == 6 || d == 8)) vfail("UnsupportedExtDegree(%u)", d);
  for (int i = 0; i < 3; ++i)
    if (!M->rows[i]) vfail("ZeroRowCount");  // RowCounts::validate, batch_stark_prover.rs:475-479
  // TablePacking::validate, packing.rs:140-161
  if (!M->public_lanes) vfail("ZeroLanes(\"public_lanes\")");
  if (!M->alu_lanes) vfail("ZeroLanes(\"alu_lanes\")");
  for (uint32_t i = 0; i < M->n_npo_lanes; ++i)
    if (!M->npo_lanes[i].lanes) vfail("ZeroNpoLanes(%s)", M->npo_lanes[i].op_type);
  if (!M->min_trace_height || (M->min_trace_height & (M->min_trace_height - 1))) vfail("BadMinTraceHeight(%u)", M->min_trace_height);
  if (M->horner_packed_steps < 2) vfail("BadHornerPackedSteps(%u)", M->horner_packed_steps);
  // NonPrimitiveTableEntry::validate, batch_stark_prover.rs:292-300
  for (uint32_t i = 0; i < M->n_non_primitives; ++i)
    if (!M->non_primitives[i].lanes) vfail("ZeroNpoLanes(%s)", M->non_primitives[i].op_type);
}

// ---- the opened values of one instance as an AIR view over the extension field
template <class PP, int DC = 4>
struct ZetaView {
  using V = typename Chal<PP, DC>::type;
  const std::vector<V>*ml, *mn, *pl, *pn;
  V L(int c) const { return (*ml).at(c); }
  V N(int c) const { return (*mn).at(c); }
  V PL(int c) const { return (*pl).at(c); }
  V PN(int c) const { return (*pn).at(c); }
};

// acc <- acc * alpha + c over every constraint, base-field constraints first
// (recursion/src/traits/air.rs:162-182)
template <class PP, int DC = 4>
struct ZetaFold {
  using E = typename Chal<PP, DC>::type;
  E alpha, acc = E::zero();
  int count = 0;
  void base(const E& c) { acc = acc * alpha + c; ++count; }
  void base2(const E& c0, const E& c1) { base(c0); base(c1); }
  void ext(const E& c) { base(c); }
};

// LogUp group constraints at zeta: f_g * prod d_k - sum_k m_k prod_{l != k} d_l (same grouping as QuotSink)
template <class PP, int DC = 4>
struct ZetaLookupSink {
  using E = typename Chal<PP, DC>::type;
  E prefix;
  E beta_pow[kMaxExtD + 1];
  const std::vector<E>& aux;  // EF aux columns at zeta: [0] running sum, [g + 1] fraction of group g
  ZetaFold<PP, DC>& fold;
  int pair, cnt = 0, held = 0;
  E d0 = E::zero(), m0 = E::zero(), d1 = E::zero(), m1 = E::zero(), sum_f = E::zero();
  template <int D>
  E denom(const E& idx, const VD<E, D>& v) const {
    E d = prefix + beta_pow[0] * idx;
    for (int j = 0; j < D; ++j) d += beta_pow[j + 1] * v.c[j];
    return d;
  }
  template <int D>
  void add(const E& idx, const VD<E, D>& v, const E& mult) {
    const E d = denom(idx, v);
    ++cnt;
    if (!pair) {
      const E f = aux.at(cnt);
      fold.ext(f * d - mult);
      sum_f += f;
    } else if (held == 0) {
      d0 = d; m0 = mult; held = 1;
    } else if (pair == 1) {
      const E f = aux.at(cnt / 2);
      fold.ext(f * d0 * d - (d * m0 + d0 * mult));
      sum_f += f;
      held = 0;
    } else if (held == 1) {
      d1 = d; m1 = mult; held = 2;
    } else {   // triples (the budget of a ZK configuration)
      const E f = aux.at(cnt / 3);
      fold.ext(f * d0 * d1 * d - (m0 * d1 * d + m1 * d0 * d + mult * d0 * d1));
      sum_f += f;
      held = 0;
    }
  }
  void finish() {
    if (!held) return;
    const int G = pair + 1;
    const E f = aux.at((cnt + G - 1) / G);
    if (held == 1) fold.ext(f * d0 - m0);
    else fold.ext(f * d0 * d1 - (m0 * d1 + m1 * d0));
    sum_f += f;
  }
};

// ---- MMCS
template <class PP>
std::array<Fp<PP>, P2_DIGEST> sponge_hash(const std::vector<Fp<PP>>& row, const uint32_t* rc) {
  using F = Fp<PP>;
  F s[P2_WIDTH];
  for (auto& x : s) x = F::zero();
  size_t g = 0;
  // overwrite-mode PaddingFreeSponge, rate 8 (recursion/src/pcs/mmcs.rs:38-179)
  for (; g < row.size(); g += P2_RATE) {
    for (size_t j = 0; j < (size_t)P2_RATE && g + j < row.size(); ++j) s[j] = row[g + j];
    host_permute<PP>(s, rc);
  }
  std::array<F, P2_DIGEST> d;
  for (int k = 0; k < P2_DIGEST; ++k) d[k] = s[k];
  return d;
}
template <class PP>
std::array<Fp<PP>, P2_DIGEST> compress2(const std::array<Fp<PP>, P2_DIGEST>& l, const std::array<Fp<PP>, P2_DIGEST>& r,
                                        const uint32_t* rc) {
  Fp<PP> s[P2_WIDTH];
  for (int k = 0; k < P2_DIGEST; ++k) { s[k] = l[k]; s[P2_DIGEST + k] = r[k]; }
  host_permute<PP>(s, rc);
  std::array<Fp<PP>, P2_DIGEST> d;
  for (int k = 0; k < P2_DIGEST; ++k) d[k] = s[k];
  return d;
}

// verify_batch of one commitment: `log_heights[m]` / rows[m] in COMMIT order; index addresses the
// tallest matrix.  (recursion/src/pcs/mmcs.rs:319-426)
template <class PP>
void mmcs_verify(const std::vector<std::array<Fp<PP>, P2_DIGEST>>& cap, int cap_height, const std::vector<int>& log_heights,
                 const std::vector<std::vector<Fp<PP>>>& rows, size_t index,
                 const std::vector<std::array<Fp<PP>, P2_DIGEST>>& path, const uint32_t* rc, const char* what) {
  using F = Fp<PP>;
  if (rows.size() != log_heights.size()) vfail("%s: %zu opened rows for %zu matrices", what, rows.size(), log_heights.size());
  int log_max = 0;
  for (int lh : log_heights) log_max = std::max(log_max, lh);
  if (cap_height > log_max || cap.size() != (size_t(1) << cap_height)) vfail("%s: bad cap", what);
  if ((int)path.size() != log_max - cap_height) vfail("%s: opening proof has %zu siblings, expected %d", what, path.size(), log_max - cap_height);
  if (index >> log_max) vfail("%s: index out of range", what);
  auto concat = [&](int lh) {
    std::vector<F> r;
    for (size_t m = 0; m < rows.size(); ++m)
      if (log_heights[m] == lh) r.insert(r.end(), rows[m].begin(), rows[m].end());
    return r;
  };
  auto any_at = [&](int lh) {
    for (int x : log_heights) if (x == lh) return true;
    return false;
  };
  auto node = sponge_hash<PP>(concat(log_max), rc);
  size_t idx = index;
  for (int lvl = 0; lvl < log_max - cap_height; ++lvl) {
    node = (idx & 1) ? compress2<PP>(path[lvl], node, rc) : compress2<PP>(node, path[lvl], rc);
    idx >>= 1;
    const int lh = log_max - lvl - 1;
    if (any_at(lh)) node = compress2<PP>(node, sponge_hash<PP>(concat(lh), rc), rc);  // injection
  }
  if (node != cap.at(idx)) vfail("%s: Merkle root mismatch", what);
}

// Arity-4 MMCS (mmcs4.h): leaf = width-32 sponge of rate 24 over the rows of the tallest matrices; per level step - 1
// siblings in ascending position around the running digest at `pos = index mod step`, chunks above `step` zero; an
// injected class as one more compression (node, digest of its rows, 0, 0).  recursion/src/pcs/mmcs.rs:1165-1316 is
// the same walk in circuit form.  `rcw`: the width-32 constant table (Montgomery), rc + p2_num_constants.
template <class PP>
std::array<Fp<PP>, P2_DIGEST> sponge_hash_w32(const std::vector<Fp<PP>>& row, const uint32_t* rcw) {
  using F = Fp<PP>;
  F s[P2W_WIDTH];
  for (auto& x : s) x = F::zero();
  struct { void put(F) {} } sink;
  for (size_t g = 0; g < row.size(); g += 24) {
    for (size_t j = 0; j < 24 && g + j < row.size(); ++j) s[j] = row[g + j];
    p2w_permute_traced<PP>(s, rcw, sink);
  }
  std::array<F, P2_DIGEST> d;
  for (int k = 0; k < P2_DIGEST; ++k) d[k] = s[k];
  return d;
}
template <class PP>
std::array<Fp<PP>, P2_DIGEST> compress4(const std::array<Fp<PP>, P2_DIGEST>* c, const uint32_t* rcw) {
  using F = Fp<PP>;
  F s[P2W_WIDTH];
  for (int j = 0; j < 4; ++j)
    for (int k = 0; k < P2_DIGEST; ++k) s[j * P2_DIGEST + k] = c[j][k];
  struct { void put(F) {} } sink;
  p2w_permute_traced<PP>(s, rcw, sink);
  std::array<F, P2_DIGEST> d;
  for (int k = 0; k < P2_DIGEST; ++k) d[k] = s[k];
  return d;
}
template <class PP>
void mmcs_verify4(const std::vector<std::array<Fp<PP>, P2_DIGEST>>& cap, int cap_height, const std::vector<int>& log_heights,
                  const std::vector<std::vector<Fp<PP>>>& rows, size_t index,
                  const std::vector<std::array<Fp<PP>, P2_DIGEST>>& path, const uint32_t* rcw, const char* what) {
  using F = Fp<PP>;
  using Digest = std::array<F, P2_DIGEST>;
  if (rows.size() != log_heights.size() || rows.empty()) vfail("%s: %zu opened rows for %zu matrices", what, rows.size(), log_heights.size());
  if (cap_height != 0 || cap.size() != 1) vfail("%s: bad cap (the arity-4 MMCS has a one-digest cap)", what);
  int log_max = 0;
  std::vector<size_t> heights;
  for (int lh : log_heights) {
    log_max = std::max(log_max, lh);
    heights.push_back(size_t(1) << lh);
  }
  if (index >> log_max) vfail("%s: index out of range", what);
  const std::vector<Mmcs4Level> levels = mmcs4_schedule(heights);
  if (path.size() != mmcs4_proof_len(levels))
    vfail("%s: opening proof has %zu siblings, expected %zu", what, path.size(), mmcs4_proof_len(levels));
  auto concat = [&](size_t h) {
    std::vector<F> r;
    for (size_t m = 0; m < rows.size(); ++m)
      if (heights[m] == h) r.insert(r.end(), rows[m].begin(), rows[m].end());
    return r;
  };
  Digest zero;
  zero.fill(F::zero());
  Digest node = sponge_hash_w32<PP>(concat(size_t(1) << log_max), rcw);
  size_t at = 0;
  for (const Mmcs4Level& lv : levels) {
    const size_t pos = (index >> lv.bits) & (size_t)(lv.step - 1);
    Digest c[4] = {zero, zero, zero, zero};
    for (size_t j = 0; j < (size_t)lv.step; ++j) c[j] = j == pos ? node : path[at++];
    node = compress4<PP>(c, rcw);
    if (lv.inject_h) {
      Digest in[4] = {node, sponge_hash_w32<PP>(concat(lv.inject_h), rcw), zero, zero};
      node = compress4<PP>(in, rcw);
    }
  }
  if (node != cap[0]) vfail("%s: Merkle root mismatch", what);
}

// ---- the out-of-domain identity of one instance: folded constraints(zeta) / Z_H(zeta) == quotient(zeta)
// (recursion/src/verifier/batch_stark.rs:886-1017, verifier/quotient.rs:60-140).  Shared by the
// verifier and by the prover's self-check before it serialises a proof (prove_impl.hip.h): the
// counterpart of prove_batch's debug constraint check, at the cost of one evaluation per table.
template <class PP, int DC = 4>
struct ZetaInstance {
  using E = typename Chal<PP, DC>::type;
  const std::vector<E>* main_local;
  const std::vector<E>* main_next;  // null when the AIR reads no next row
  const std::vector<E>* prep_local;
  const std::vector<E>* prep_next;
  const std::vector<E>* perm_local;
  const std::vector<E>* perm_next;
  const std::vector<std::vector<E>>* chunks;
  const E* terminal;  // null without lookups
};
// log_n_i: the BASE trace degree bits; is_zk: the quotient domain has 2^(log_chunks + is_zk) chunk cosets of the base
// trace size (batch_stark.rs:701-717).
template <class PP, int DC = 4>
void check_instance_at_zeta(const AirParams& air, const LookupLayout& L, int log_n_i, const ZetaInstance<PP, DC>& in,
                            typename Chal<PP, DC>::type alpha, typename Chal<PP, DC>::type zeta,
                            typename Chal<PP, DC>::type l_prefix, const typename Chal<PP, DC>::type* l_beta_pow,
                            const uint32_t* rc_mont, size_t i, int is_zk = 0) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  const F gen = F::generator();
  {
    const size_t n = size_t(1) << log_n_i;
    const F g = F::two_adic_generator(log_n_i), g_inv = g.inv();
    // selectors on the (unshifted) trace domain: Z_H = zeta^n - 1 (pcs/fri/targets.rs:868-908)
    const E zh = zeta.pow(n) - E::one();
    if (zh.is_zero()) vfail("zeta lies in the trace domain");
    const E is_transition = zeta - E::from_base(g_inv);
    const E is_first = zh * (zeta - E::one()).inv();
    const E is_last = zh * is_transition.inv();
    static const std::vector<E> none;
    ZetaView<PP, DC> v{in.main_local, in.main_next ? in.main_next : &none, in.prep_local, in.prep_next};
    ZetaFold<PP, DC> fold;
    fold.alpha = alpha;
    const bool generic = ext_degree_is_binomial_generic((uint32_t)air.ext_d);
    if (!(air.ext_d == 1 || air.ext_d == 4 || (air.ext_d == 5 && kHasQuintic<PP>) || (generic && air.kind != AIR_POSEIDON2)))
      vfail("instance %zu: no AIR of kind %d for circuit extension degree %d", i, air.kind, air.ext_d);
    if (air.kind == AIR_ALU) {
      dispatch_air_degree<PP>(air.ext_d, [&](auto dc) { alu_constraints<PP, decltype(dc)::value>(air, v, fold); });
    } else if (air.kind == AIR_POSEIDON2) {
      if (air.ext_d != 4) poseidon2_d1_constraints<PP>(v, is_transition, rc_mont, fold);
      else poseidon2_constraints<PP>(v, is_transition, rc_mont, fold);
    } else if (air.kind == AIR_POSEIDON2_W32) {
      if (air.ext_d != 4) vfail("instance %zu: the width-32 Poseidon2 table belongs to D = 4 circuits", i);
      poseidon2w_constraints<PP>(v, is_transition, rc_mont + p2_num_constants<PP>(), fold);
    }
    if (fold.count != air_num_base_constraints<PP>(air)) vfail("instance %zu: constraint count mismatch", i);
    if (L.n_groups) {
      // EF aux columns from their 4 base-column openings: sum_k x^k * col_k(zeta)
      auto ef_cols = [&](const std::vector<E>& flat) {
        std::vector<E> out(flat.size() / DC, E::zero());
        for (size_t c = 0; c < out.size(); ++c)
          for (int k = 0; k < DC; ++k) {
            E basis = E::zero();
            basis.c[k] = F::one();
            out[c] += basis * flat[c * DC + k];
          }
        return out;
      };
      const std::vector<E> aux_l = ef_cols(*in.perm_local), aux_n = ef_cols(*in.perm_next);
      ZetaLookupSink<PP, DC> sink{l_prefix, {l_beta_pow[0], l_beta_pow[1], l_beta_pow[2], l_beta_pow[3], l_beta_pow[4],
                                         l_beta_pow[5], l_beta_pow[6], l_beta_pow[7], l_beta_pow[8]},
                              aux_l, fold, L.pair};
      dispatch_air_degree<PP>(air.ext_d, [&](auto dc) { air_interactions<PP, decltype(dc)::value>(air, v, sink); });
      sink.finish();
      if (sink.cnt != L.n_interactions) vfail("instance %zu: interaction count mismatch", i);
      const E s = aux_l[0], s_next = aux_n[0], terminal = *in.terminal;
      fold.ext(s * is_first);
      fold.ext((s_next - s - sink.sum_f) * is_transition);
      fold.ext((s + sink.sum_f - terminal) * is_last);
    }
    // quotient(zeta) = sum_c L_c(zeta) * Q_c(zeta) over the 2^log_chunks cosets of the quotient domain
    // (recursion/src/verifier/quotient.rs:60-)
    const int lq = L.log_chunks + is_zk;
    const size_t C = size_t(1) << lq;
    if (in.chunks->size() != C) vfail("instance %zu: %zu quotient chunks, expected %zu", i, in.chunks->size(), C);
    const F wq = F::two_adic_generator(log_n_i + lq);
    std::vector<F> shifts(C);
    for (size_t c = 0; c < C; ++c) shifts[c] = gen * wq.pow(c);
    E quotient = E::zero();
    for (size_t c = 0; c < C; ++c) {
      // vanishing polynomial of coset c' at x: (x / shift_c')^n - 1
      E num = E::one();
      F den = F::one();
      for (size_t o = 0; o < C; ++o) {
        if (o == c) continue;
        num *= (zeta * shifts[o].inv()).pow(n) - E::one();
        den *= (shifts[c] * shifts[o].inv()).pow(n) - F::one();
      }
      E qc = E::zero();
      for (int k = 0; k < DC; ++k) {
        E basis = E::zero();
        basis.c[k] = F::one();
        qc += basis * (*in.chunks)[c][k];
      }
      quotient += num * den.inv() * qc;
    }
    if (!(fold.acc * zh.inv() == quotient)) vfail("instance %zu: constraints do not match the quotient at zeta (OodEvaluationMismatch)", i);
  }
}

// ---- the whole verification
struct VerifyParams {
  int log_blowup, max_log_arity, cap_height, log_final_poly_len, commit_pow_bits, query_pow_bits, num_queries;
  std::vector<uint8_t> fri_log_arities;  // explicit folding schedule (p3r_config), empty: the rule
  ProofLayout layout;                    // field order of the serialised structs (p3r_config.proof_layout)
  int mmcs_arity = 2;                    // 4: the arity-4 MMCS over the width-32 permutation (p3r_config.mmcs_arity)
  int zk = 0;                            // HidingFriPcs (p3r_config.zk)
  int num_random_codewords = 0;          // random codeword columns per committed matrix (p3r_config.num_random_codewords)
  int mmcs_salt_elems = 0;               // MerkleTreeHidingMmcs: salt elements per committed row (p3r_config.mmcs_salt_elems)
};

// MerkleTreeHidingMmcs::verify_batch through the plain walk: the leaf preimage of a height class is the concatenation of
// [row | salt] per matrix (recursion/src/pcs/mmcs.rs:375-389 for base rows, :470-486 for flattened extension rows), i.e.
// the plain preimage of the matrices widened by their salts.
template <class F>
std::vector<std::vector<F>> salted_rows(const std::vector<std::vector<F>>& rows, const std::vector<std::vector<F>>& salts, int salt_elems,
                                        const char* what) {
  if (!salt_elems) {
    if (!salts.empty()) vfail("%s: salts under a non-hiding MMCS", what);
    return rows;
  }
  if (salts.size() != rows.size()) vfail("%s: %zu salts for %zu matrices", what, salts.size(), rows.size());
  std::vector<std::vector<F>> out(rows);
  for (size_t m = 0; m < rows.size(); ++m) {
    if ((int)salts[m].size() != salt_elems) vfail("%s: a salt of %zu elements, expected %d", what, salts[m].size(), salt_elems);
    out[m].insert(out[m].end(), salts[m].begin(), salts[m].end());
  }
  return out;
}

template <class PP, int DC = 4>
void verify_batch(const VerifyParams& prm, const std::vector<uint32_t>& rc_canonical, const std::vector<AirParams>& airs,
                  const std::vector<uint32_t>& prep_cap_canonical, const std::vector<uint32_t>& expected_degree_bits,
                  const uint8_t* bytes, size_t n_bytes, bool canonical) {
  using F = Fp<PP>;
  using E = typename Chal<PP, DC>::type;
  using Digest = std::array<F, P2_DIGEST>;
  const int zk = prm.zk ? 1 : 0, R = zk ? prm.num_random_codewords : 0;
  if (zk && (R < 1 || R > 8)) vfail("num_random_codewords must be in 1..8");
  if (prm.mmcs_salt_elems < 0 || prm.mmcs_salt_elems > 16) vfail("mmcs_salt_elems must be in 0..16");
  const ParsedProof<PP, DC> P = parse_proof<PP, DC>(bytes, n_bytes, canonical, nullptr, prm.layout, zk != 0, prm.mmcs_salt_elems != 0);
  const size_t ni = airs.size();
  // the width-16 round constants, followed by the width-32 table (poseidon2.h; csrc/p3r_core.hip::constants_table)
  if (rc_canonical.size() != (size_t)p2_num_constants<PP>() + (size_t)p2w_num_constants<PP>()) vfail("wrong number of permutation constants");
  std::vector<uint32_t> rc(rc_canonical.size());
  for (size_t i = 0; i < rc.size(); ++i) rc[i] = F::from_canonical(rc_canonical[i]).v;
  const int p2w = p2_perm_cols<PP>() + 2;
  if (P.insts.size() != ni || P.degree_bits.size() != ni || P.terminals.size() != ni)
    vfail("proof has %zu instances, the verifier was given %zu AIRs", P.insts.size(), ni);
  const int cap_h = prm.cap_height, lb = prm.log_blowup;
  std::vector<Digest> prep_cap(size_t(1) << cap_h);
  if (prep_cap_canonical.size() != prep_cap.size() * P2_DIGEST) vfail("preprocessed commitment has the wrong size");
  for (size_t j = 0; j < prep_cap.size(); ++j)
    for (int k = 0; k < P2_DIGEST; ++k) prep_cap[j][k] = F::from_canonical(prep_cap_canonical[j * P2_DIGEST + k]);

  // ---- shapes
  // randomisation must match the PCS's ZK setting (batch_stark.rs:424-428: RandomizationError)
  if (P.rand_cap.has_value() != (zk != 0)) vfail("RandomizationError: random commitment presence does not match the ZK setting");
  for (auto& in : P.insts)
    if (in.random.has_value() != (zk != 0)) vfail("RandomizationError: random opened values presence does not match the ZK setting");
  std::vector<LookupLayout> layouts(ni);
  std::vector<int> log_n(ni), log_e(ni), width(ni), prep_w(ni);   // base / extended (committed) degree bits
  bool any_lookup = false;
  std::vector<int> perm_insts;
  for (size_t i = 0; i < ni; ++i) {
    const auto& in = P.insts[i];
    layouts[i] = lookup_layout(airs[i], zk);
    log_e[i] = P.degree_bits[i];
    // the domains are the verifier's, not the prover's: recursion/src/verifier/batch_stark.rs:793 (ZK: the preprocessed
    // metadata holds the EXTENDED degree bits, recursion.rs:374)
    if (expected_degree_bits.size() != ni || (uint32_t)log_e[i] != expected_degree_bits[i])
      vfail("InvalidProofShape: instance %zu declares degree_bits %d, the preprocessed metadata has %u", i, log_e[i],
            i < expected_degree_bits.size() ? expected_degree_bits[i] : 0u);
    if (log_e[i] < zk) vfail("InvalidProofShape: extended degree bits smaller than the ZK adjustment");   // :536
    log_n[i] = log_e[i] - zk;
    if (log_e[i] + lb > PP::TWO_ADICITY) vfail("instance %zu: degree too large", i);
    width[i] = air_width_of(airs[i], p2w, p2w_perm_cols<PP>() + 4);
    prep_w[i] = air_prep_width_of(airs[i]);
    // quotient_degree = 1 << (log_qd + is_zk) (:487-496)
    const size_t aw = (size_t)layouts[i].aux_width() * DC, C = size_t(1) << (layouts[i].log_chunks + zk);
    if (in.random && in.random->size() != (size_t)DC) vfail("RandomizationError: instance %zu: %zu random opened values", i, in.random->size());
    if (in.main_local.size() != (size_t)width[i]) vfail("instance %zu: %zu main openings, the AIR has %d columns", i, in.main_local.size(), width[i]);
    if (air_uses_next(airs[i]) != in.main_next.has_value()) vfail("instance %zu: main next-row openings do not match the AIR", i);
    if (in.main_next && in.main_next->size() != (size_t)width[i]) vfail("instance %zu: bad main next width", i);
    if (in.prep_local.size() != (size_t)prep_w[i] || in.prep_next.size() != (size_t)prep_w[i]) vfail("instance %zu: bad preprocessed opening width", i);
    if (in.perm_local.size() != aw || in.perm_next.size() != aw) vfail("instance %zu: bad permutation opening width", i);
    if (in.chunks.size() != C) vfail("instance %zu: %zu quotient chunks, expected %zu", i, in.chunks.size(), C);
    for (auto& c : in.chunks) if (c.size() != (size_t)DC) vfail("instance %zu: bad quotient chunk width", i);
    if ((layouts[i].n_groups > 0) != P.terminals[i].has_value()) vfail("instance %zu: lookup terminal presence mismatch", i);
    if (layouts[i].n_groups) { any_lookup = true; perm_insts.push_back((int)i); }
  }
  if (any_lookup != P.perm_cap.has_value()) vfail("permutation commitment presence mismatch");

  // ---- transcript (same order as prove_batch)
  HostChallenger<PP, DC> ch(rc.data());
  auto observe_cap = [&](const std::vector<Digest>& cap) { for (auto& d : cap) for (auto x : d) ch.observe(x); };
  ch.observe_base_as_ext(ni);
  for (size_t i = 0; i < ni; ++i) {
    ch.observe_base_as_ext(log_e[i]);
    ch.observe_base_as_ext(log_n[i]);
    ch.observe_base_as_ext(width[i]);
    ch.observe_base_as_ext(uint64_t(1) << (layouts[i].log_chunks + zk));
  }
  observe_cap(P.main_cap);
  for (size_t i = 0; i < ni; ++i) ch.observe_base_as_ext(prep_w[i]);
  observe_cap(prep_cap);
  E l_prefix = E::zero(), l_beta_pow[kMaxExtD + 1];
  for (auto& b : l_beta_pow) b = E::zero();
  if (any_lookup) {
    const E alpha_l = ch.sample_ext(), beta_l = ch.sample_ext();
    E bp = E::one();
    // gamma = beta^W, W = the widest bus tuple = 1 + circuit extension degree (get_perm_challenges,
    // recursion/src/verifier/batch_stark.rs:1086-1100)
    int tuple_w = 2;
    for (auto& a : airs) tuple_w = std::max(tuple_w, std::min(a.ext_d, kMaxExtD) + 1);
    for (int j = 0; j < tuple_w; ++j) { l_beta_pow[j] = bp; bp *= beta_l; }
    l_prefix = alpha_l + bp;
    observe_cap(*P.perm_cap);
    for (int i : perm_insts) ch.observe_ext(*P.terminals[i]);
  }
  const E alpha = ch.sample_ext();
  observe_cap(P.quot_cap);
  if (zk) observe_cap(*P.rand_cap);   // :623-625
  const E zeta = ch.sample_ext();
  // input batches in round order ([random,] main, quotient, preprocessed, permutation): matrices (log LDE height,
  // width), points and claimed values.  ZK: every committed matrix has R more columns - the random codewords - whose
  // values at each point are the opening proof's first item; they are appended to the point's values before anything
  // is observed or reduced (merge_hiding_random_openings, pcs/fri/targets.rs:1076-1130; batch_stark.rs:1116-1260).
  struct Mat { int log_h; int w; std::vector<E> z; std::vector<std::vector<E>> vals; };
  std::vector<std::vector<Mat>> rounds;
  if (zk) {
    rounds.emplace_back();
    for (size_t i = 0; i < ni; ++i) rounds.back().push_back({log_e[i] + lb, DC, {zeta}, {*P.insts[i].random}});   // :645-661
  }
  rounds.emplace_back();
  for (size_t i = 0; i < ni; ++i) {
    Mat m{log_e[i] + lb, width[i], {zeta}, {P.insts[i].main_local}};
    // zeta * g of the BASE trace domain (:663-700)
    if (P.insts[i].main_next) { m.z.push_back(zeta * F::two_adic_generator(log_n[i])); m.vals.push_back(*P.insts[i].main_next); }
    rounds.back().push_back(m);
  }
  rounds.emplace_back();
  for (size_t i = 0; i < ni; ++i)   // randomized_quotient_domains: natural_domain_for_degree(size << is_zk) (:719-727)
    for (auto& c : P.insts[i].chunks) rounds.back().push_back({log_e[i] + lb, DC, {zeta}, {c}});
  rounds.emplace_back();
  for (size_t i = 0; i < ni; ++i)
    rounds.back().push_back({log_e[i] + lb, prep_w[i], {zeta, zeta * F::two_adic_generator(log_n[i])},
                             {P.insts[i].prep_local, P.insts[i].prep_next}});
  if (any_lookup) {
    rounds.emplace_back();
    for (int i : perm_insts)
      rounds.back().push_back({log_e[i] + lb, layouts[i].aux_width() * DC, {zeta, zeta * F::two_adic_generator(log_n[i])},
                               {P.insts[i].perm_local, P.insts[i].perm_next}});
  }
  if (zk) {
    if (P.fri_random.size() != rounds.size()) vfail("InvalidProofShape: hiding FRI proof: random rounds count does not match commitments");
    for (size_t r = 0; r < rounds.size(); ++r) {
      if (P.fri_random[r].size() != rounds[r].size()) vfail("InvalidProofShape: hiding FRI proof: random matrices count does not match (round %zu)", r);
      for (size_t m = 0; m < rounds[r].size(); ++m) {
        Mat& M = rounds[r][m];
        if (P.fri_random[r][m].size() != M.z.size()) vfail("InvalidProofShape: hiding FRI proof: random points count does not match (round %zu matrix %zu)", r, m);
        for (size_t p = 0; p < M.z.size(); ++p) {
          const auto& extra = P.fri_random[r][m][p];
          if (extra.size() != (size_t)R) vfail("InvalidProofShape: hiding FRI proof: %zu random codeword values, expected %d", extra.size(), R);
          M.vals[p].insert(M.vals[p].end(), extra.begin(), extra.end());
        }
        M.w += R;
      }
    }
  }
  for (auto& r : rounds) for (auto& m : r) for (auto& pv : m.vals) for (auto& v : pv) ch.observe_ext(v);

  // ---- constraints at zeta
  const F gen = F::generator();
  E terminal_sum = E::zero();
  for (size_t i = 0; i < ni; ++i) {
    const auto& in = P.insts[i];
    ZetaInstance<PP, DC> zi{&in.main_local, in.main_next ? &*in.main_next : nullptr, &in.prep_local, &in.prep_next,
                        &in.perm_local, &in.perm_next, &in.chunks, P.terminals[i] ? &*P.terminals[i] : nullptr};
    check_instance_at_zeta<PP, DC>(airs[i], layouts[i], log_n[i], zi, alpha, zeta, l_prefix, l_beta_pow, rc.data(), i, zk);
    if (layouts[i].n_groups) terminal_sum += *P.terminals[i];
  }
  if (any_lookup && !terminal_sum.is_zero()) vfail("global lookup sum is not zero");

  // ---- FRI transcript
  const E fri_alpha = ch.sample_ext();
  std::vector<const std::vector<Digest>*> round_caps;
  if (zk) round_caps.push_back(&*P.rand_cap);
  round_caps.push_back(&P.main_cap); round_caps.push_back(&P.quot_cap); round_caps.push_back(&prep_cap);
  if (any_lookup) round_caps.push_back(&*P.perm_cap);
  int log_max = 0;
  std::vector<int> heights;
  for (auto& r : rounds) for (auto& m : r) { log_max = std::max(log_max, m.log_h); heights.push_back(m.log_h); }
  std::sort(heights.rbegin(), heights.rend());
  heights.erase(std::unique(heights.begin(), heights.end()), heights.end());
  const int log_final = prm.log_final_poly_len + lb;
  // the arity schedule the prover must have used (the FRI prover's rule, prove_impl.hip.h step 7)
  std::vector<int> las;
  {
    size_t next_h = 1;
    int cur = log_max;
    while (cur > log_final) {
      int log_next = next_h < heights.size() ? heights[next_h] : -1;
      const int la = fri_log_arity(prm.fri_log_arities, las.size(), prm.max_log_arity, cur, log_final, log_next);
      if (la < 0) vfail("FRI: the configured folding schedule does not fit the proof (phase %zu)", las.size());
      if (next_h < heights.size() && heights[next_h] == cur - la) ++next_h;
      cur -= la;
      las.push_back(la);
    }
    if (next_h != heights.size()) vfail("FRI: an input height is never rolled in");
    if (cur != log_final) vfail("FRI: fold schedule does not end at the final polynomial length");
  }
  if (P.commit_caps.size() != las.size() || P.commit_pow.size() != las.size()) vfail("FRI: %zu commit phases, expected %zu", P.commit_caps.size(), las.size());
  if (P.final_poly.size() != (size_t(1) << prm.log_final_poly_len)) vfail("FRI: final polynomial has the wrong length");
  std::vector<E> betas;
  for (size_t p = 0; p < las.size(); ++p) {
    observe_cap(P.commit_caps[p]);
    if (!ch.check_witness(prm.commit_pow_bits, P.commit_pow[p])) vfail("FRI: invalid commit-phase proof of work");
    betas.push_back(ch.sample_ext());
  }
  for (auto& c : P.final_poly) ch.observe_ext(c);
  for (int la : las) ch.observe(F::from_canonical((uint32_t)la));
  if (!ch.check_witness(prm.query_pow_bits, P.query_pow)) vfail("FRI: invalid query proof of work");
  if ((int)P.queries.size() != prm.num_queries) vfail("FRI: %zu queries, expected %d", P.queries.size(), prm.num_queries);

  // ---- queries
  size_t max_w = 1;
  for (auto& r : rounds) for (auto& m : r) max_w = std::max(max_w, (size_t)m.w);
  std::vector<E> fa_pow(max_w + 1);
  fa_pow[0] = E::one();
  for (size_t c = 1; c <= max_w; ++c) fa_pow[c] = fa_pow[c - 1] * fri_alpha;
  for (size_t qi = 0; qi < P.queries.size(); ++qi) {
    const auto& Q = P.queries[qi];
    const size_t index = ch.sample_bits(log_max);
    if (Q.rounds.size() != rounds.size()) vfail("query %zu: %zu input batches, expected %zu", qi, Q.rounds.size(), rounds.size());
    // reduced openings per height, alpha powers restart per height (pcs/fri/verifier.rs:1068-1356)
    std::map<int, std::pair<E, E>> ro;  // log_h -> (next alpha power, value)
    for (size_t r = 0; r < rounds.size(); ++r) {
      std::vector<int> lhs;
      for (auto& m : rounds[r]) lhs.push_back(m.log_h);
      int r_max = 0;
      for (int x : lhs) r_max = std::max(r_max, x);
      const auto& qr = Q.rounds[r];
      if (qr.rows.size() != rounds[r].size()) vfail("query %zu: batch %zu opens %zu matrices, expected %zu", qi, r, qr.rows.size(), rounds[r].size());
      for (size_t m = 0; m < rounds[r].size(); ++m)
        if (qr.rows[m].size() != (size_t)rounds[r][m].w) vfail("query %zu: batch %zu matrix %zu has the wrong width", qi, r, m);
      const auto leaf_rows = salted_rows<F>(qr.rows, qr.salts, prm.mmcs_salt_elems, "input batch");
      if (prm.mmcs_arity == 4) mmcs_verify4<PP>(*round_caps[r], cap_h, lhs, leaf_rows, index >> (log_max - r_max), qr.path, rc.data() + p2_num_constants<PP>(), "input batch");
      else mmcs_verify<PP>(*round_caps[r], cap_h, lhs, leaf_rows, index >> (log_max - r_max), qr.path, rc.data(), "input batch");
      for (size_t m = 0; m < rounds[r].size(); ++m) {
        const Mat& M = rounds[r][m];
        auto it = ro.find(M.log_h);
        if (it == ro.end()) it = ro.emplace(M.log_h, std::make_pair(E::one(), E::zero())).first;
        const size_t ridx = index >> (log_max - M.log_h);
        const F x = gen * F::two_adic_generator(M.log_h).pow(bit_reverse((uint32_t)ridx, M.log_h));  // verifier.rs:921-981
        E S = E::zero();
        for (int c = 0; c < M.w; ++c) S += fa_pow[c] * qr.rows[m][c];
        for (size_t p = 0; p < M.z.size(); ++p) {
          E Vp = E::zero();
          for (int c = 0; c < M.w; ++c) Vp += fa_pow[c] * M.vals[p][c];
          it->second.second += it->second.first * (Vp - S) * (M.z[p] - E::from_base(x)).inv();
          it->second.first *= fa_pow[M.w];
        }
      }
    }
    // fold chain (pcs/fri/verifier.rs:562-781)
    if (Q.phases.size() != las.size()) vfail("query %zu: %zu commit-phase openings, expected %zu", qi, Q.phases.size(), las.size());
    if (!ro.count(log_max)) vfail("query %zu: no reduced opening at the maximum height", qi);
    E folded = ro[log_max].second;
    size_t idx = index;
    int cur = log_max;
    const F neg_half = -(F::from_canonical(2).inv());
    for (size_t p = 0; p < las.size(); ++p) {
      const auto& ph = Q.phases[p];
      const int la = las[p];
      const size_t arity = size_t(1) << la, pos = idx & (arity - 1), row = idx >> la;
      if (ph.la != la || ph.sibs.size() != arity - 1) vfail("query %zu phase %zu: wrong arity", qi, p);
      std::vector<E> e(arity);
      for (size_t j = 0, s = 0; j < arity; ++j) e[j] = j == pos ? folded : ph.sibs[s++];
      // the leaf is the row of 2^la sibling evaluations, extension elements flattened
      std::vector<F> leaf;
      for (auto& x : e) for (int k = 0; k < DC; ++k) leaf.push_back(x.c[k]);
      const auto leaf_rows = salted_rows<F>({leaf}, ph.salts, prm.mmcs_salt_elems, "FRI commit phase");
      if (prm.mmcs_arity == 4) mmcs_verify4<PP>(P.commit_caps[p], cap_h, {cur - la}, leaf_rows, row, ph.path, rc.data() + p2_num_constants<PP>(), "FRI commit phase");
      else mmcs_verify<PP>(P.commit_caps[p], cap_h, {cur - la}, leaf_rows, row, ph.path, rc.data(), "FRI commit phase");
      F ss_inv = F::two_adic_generator(cur).inv().pow(bit_reverse((uint32_t)row, cur - la));
      E b = betas[p];
      const F omega = F::two_adic_generator(la);
      size_t len = arity;
      for (int s = 0; s < la; ++s) {
        const F om_s = omega.pow(uint64_t(1) << s);
        for (size_t j = 0; j < len / 2; ++j) {
          const F x0_inv = ss_inv * om_s.pow(bit_reverse((uint32_t)(2 * j), la - s)).inv();
          const E t = b * x0_inv - E::one();  // arity2_fold_at_point: e0 + (beta - x0)(e1 - e0)(-1/2)/x0
          e[j] = e[2 * j] + t * (e[2 * j + 1] - e[2 * j]) * neg_half;
        }
        len /= 2;
        ss_inv = ss_inv.sqr();
        b = b.sqr();
      }
      folded = e[0];
      cur -= la;
      idx = row;
      auto it = ro.find(cur);
      if (it != ro.end() && cur != log_max) folded += betas[p].pow(arity) * it->second.second;  // roll-in
    }
    // final polynomial at x = w^{bitrev(idx)} of the size-2^cur domain (verifier.rs:887-915)
    const F x = F::two_adic_generator(cur).pow(bit_reverse((uint32_t)idx, cur));
    E eval = E::zero();
    for (size_t k = P.final_poly.size(); k-- > 0;) eval = eval * x + P.final_poly[k];
    if (!(eval == folded)) vfail("query %zu: final polynomial mismatch", qi);
  }
}

}  // namespace p3r
