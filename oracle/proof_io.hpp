// ORACLE (test infrastructure): postcard (varint little-endian) encoding of BatchProof, the
// byte string "bit-exact proof bytes" refers to (recursion/examples/common/mod.rs:144-147).
// PARITY UNPINNED: struct field ORDER follows the destructuring patterns in
// recursion/src/types/proof.rs:403-409,452-457,527-534,585-589 and
// recursion/src/pcs/fri/targets.rs:104-110; the serde derives themselves live in the
// un-vendored p3-batch-stark / p3-fri / p3-commit 0.6 crates.
//
// Field elements: p3-monty-31 serialises the internal Montgomery word as a u32
// ("faster to serialize in monty form"); FIELD_ENCODING_MONTY reproduces that, the canonical
// alternative is kept behind the same switch (DESIGN.md "EXT choices").
#pragma once
#include "stark.hpp"

namespace orc {

enum { FIELD_ENCODING_MONTY = 0, FIELD_ENCODING_CANONICAL = 1 };

template <class FP>
struct Writer {
  std::vector<uint8_t> out;
  int enc = FIELD_ENCODING_MONTY;
  void byte(uint8_t b) { out.push_back(b); }
  void varint(uint64_t v) {
    while (v >= 0x80) { out.push_back((uint8_t)(v | 0x80)); v >>= 7; }
    out.push_back((uint8_t)v);
  }
  void fe(Fe<FP> x) {
    uint64_t v = x.v;
    if (enc == FIELD_ENCODING_MONTY) v = ((uint64_t)x.v << 32) % FP::P;
    varint(v);
  }
  void ef(const Fe4<FP>& x) { for (int i = 0; i < Fe4<FP>::deg(); ++i) fe(x.c[i]); }
  void digest(const std::array<Fe<FP>, DIGEST>& d) { for (auto c : d) fe(c); }
  void cap(const std::vector<std::array<Fe<FP>, DIGEST>>& c) { varint(c.size()); for (auto& d : c) digest(d); }
  void vec_fe(const std::vector<Fe<FP>>& v) { varint(v.size()); for (auto x : v) fe(x); }
  void vec_ef(const std::vector<Fe4<FP>>& v) { varint(v.size()); for (auto& x : v) ef(x); }
  void some() { byte(1); }
  void none() { byte(0); }
};

// Field order of the serialised structs (twin of p3r_config.proof_layout): three permutations,
// BatchProof {0 commitments, 1 opened_values, 2 opening_proof, 3 global_lookup_data, 4 degree_bits},
// FriProof {0 commit_phase_commits, 1 commit_pow_witnesses, 2 query_proofs, 3 final_poly, 4 query_pow_witness},
// OpenedValues {0 trace_local, 1 trace_next, 2 preprocessed_local, 3 preprocessed_next, 4 quotient_chunks,
// 5 random, 6 permutation_local, 7 permutation_next}.  Identity = the order of the in-tree destructuring
// patterns (recursion/src/types/proof.rs:403-409,452-457,527-534,585-589, pcs/fri/targets.rs:104-110).
struct Layout {
  uint8_t batch[5] = {0, 1, 2, 3, 4};
  uint8_t fri[5] = {0, 1, 2, 3, 4};
  uint8_t opened[8] = {0, 1, 2, 3, 4, 5, 6, 7};
};

template <class FP>
// `salted`: the MMCSs are MerkleTreeHidingMmcs - an opening proof is the tuple (Vec<Vec<F>> salts, Vec<[F; 8]> siblings)
// (`SaltedMmcsProof`, recursion/src/pcs/mmcs.rs:763-768); like `zk` a property of the configuration, not of the bytes
std::vector<uint8_t> serialize_proof(const BatchProof<FP>& p, int enc = FIELD_ENCODING_MONTY, const Layout& L = Layout{},
                                     bool salted = false) {
  Writer<FP> w;
  w.enc = enc;
  auto commitments = [&] {  // { main, permutation?, quotient_chunks, random? }
    w.cap(p.main_commit);
    if (p.has_permutation) { w.some(); w.cap(p.permutation_commit); } else w.none();
    w.cap(p.quotient_commit);
    if (p.has_random) { w.some(); w.cap(p.random_commit); } else w.none();
  };
  auto opened = [&] {  // { instances: Vec<OpenedValuesWithLookups> }
    w.varint(p.opened.size());
    for (auto& ov : p.opened)
      for (int k = 0; k < 8; ++k) switch (L.opened[k]) {
        case 0: w.vec_ef(ov.trace_local); break;
        case 1: if (ov.has_trace_next) { w.some(); w.vec_ef(ov.trace_next); } else w.none(); break;
        case 2: w.some(); w.vec_ef(ov.preprocessed_local); break;
        case 3: w.some(); w.vec_ef(ov.preprocessed_next); break;
        case 4:
          w.varint(ov.quotient_chunks.size());
          for (auto& c : ov.quotient_chunks) w.vec_ef(c);
          break;
        case 5: if (ov.has_random) { w.some(); w.vec_ef(ov.random); } else w.none(); break;  // random: Option<Vec<Challenge>>
        case 6: w.vec_ef(ov.permutation_local); break;
        default: w.vec_ef(ov.permutation_next); break;
      }
  };
  const auto& f = p.fri;
  auto queries = [&] {
    w.varint(f.query_proofs.size());
    for (auto& q : f.query_proofs) {
      w.varint(q.input_proof.size());
      for (auto& bo : q.input_proof) {
        w.varint(bo.opened_values.size());
        for (auto& r : bo.opened_values) w.vec_fe(r);
        if (salted) { w.varint(bo.salts.size()); for (auto& sl : bo.salts) w.vec_fe(sl); }
        w.varint(bo.opening_proof.size());
        for (auto& d : bo.opening_proof) w.digest(d);
      }
      w.varint(q.commit_phase_openings.size());
      for (auto& s : q.commit_phase_openings) {
        w.byte(s.log_arity);
        w.vec_ef(s.sibling_values);
        if (salted) { w.varint(s.salts.size()); for (auto& sl : s.salts) w.vec_fe(sl); }
        w.varint(s.opening_proof.size());
        for (auto& d : s.opening_proof) w.digest(d);
      }
    }
  };
  auto fri = [&] {  // opening_proof: FriProof; HidingFriPcs::Proof = (OpenedValues<Challenge>, FriProof) - a tuple, no framing
    if (p.has_random) {
      w.varint(p.fri_random.size());
      for (auto& rd : p.fri_random) {
        w.varint(rd.size());
        for (auto& m : rd) {
          w.varint(m.size());
          for (auto& pt : m) w.vec_ef(pt);
        }
      }
    }
    for (int k = 0; k < 5; ++k) switch (L.fri[k]) {
      case 0:
        w.varint(f.commit_phase_commits.size());
        for (auto& c : f.commit_phase_commits) w.cap(c);
        break;
      case 1: w.vec_fe(f.commit_pow_witnesses); break;
      case 2: queries(); break;
      case 3: w.vec_ef(f.final_poly); break;
      default: w.fe(f.query_pow_witness); break;
    }
  };
  for (int k = 0; k < 5; ++k) switch (L.batch[k]) {
    case 0: commitments(); break;
    case 1: opened(); break;
    case 2: fri(); break;
    case 3:  // lookup_terminals: Vec<Option<EF>>
      w.varint(p.has_terminal.size());
      for (size_t i = 0; i < p.has_terminal.size(); ++i) {
        if (p.has_terminal[i]) { w.some(); w.ef(p.lookup_terminals[i]); } else w.none();
      }
      break;
    default:  // degree_bits: Vec<usize>
      w.varint(p.degree_bits.size());
      for (auto d : p.degree_bits) w.varint(d);
      break;
  }
  return w.out;
}

template <class FP>
struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  int enc = FIELD_ENCODING_MONTY;
  uint8_t byte() { if (p >= end) throw std::runtime_error("proof bytes truncated"); return *p++; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      uint8_t b = byte();
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    throw std::runtime_error("varint too long");
  }
  size_t len() { uint64_t n = varint(); if (n > (uint64_t)(end - p)) throw std::runtime_error("length exceeds input"); return (size_t)n; }
  Fe<FP> fe() {
    uint64_t v = varint();
    if (v >= FP::P) throw std::runtime_error("field element out of range");
    if (enc == FIELD_ENCODING_MONTY) {
      // v = x * 2^32 mod P  ->  x = v * (2^32)^-1
      static const Fe<FP> rinv = Fe<FP>((uint64_t(1) << 32) % FP::P).inv();
      return Fe<FP>(v) * rinv;
    }
    return Fe<FP>(v);
  }
  Fe4<FP> ef() { Fe4<FP> e; for (int i = 0; i < Fe4<FP>::deg(); ++i) e.c[i] = fe(); return e; }
  std::array<Fe<FP>, DIGEST> digest() { std::array<Fe<FP>, DIGEST> d; for (auto& c : d) c = fe(); return d; }
  std::vector<std::array<Fe<FP>, DIGEST>> cap() { size_t n = len(); std::vector<std::array<Fe<FP>, DIGEST>> c(n); for (auto& d : c) d = digest(); return c; }
  std::vector<Fe<FP>> vec_fe() { size_t n = len(); std::vector<Fe<FP>> v(n); for (auto& x : v) x = fe(); return v; }
  std::vector<Fe4<FP>> vec_ef() { size_t n = len(); std::vector<Fe4<FP>> v(n); for (auto& x : v) x = ef(); return v; }
  bool option() { uint8_t b = byte(); if (b > 1) throw std::runtime_error("bad option tag"); return b == 1; }
};

template <class FP>
// `zk`: the proof type is the hiding PCS's (the opening proof is the tuple above): a property of the configuration, as
// SC::Pcs is in the reference - not something the bytes announce.
BatchProof<FP> deserialize_proof(const uint8_t* data, size_t n, int enc = FIELD_ENCODING_MONTY, const Layout& L = Layout{},
                                 bool zk = false, bool salted = false) {
  Reader<FP> r{data, data + n, enc};
  BatchProof<FP> p;
  auto commitments = [&] {
    p.main_commit = r.cap();
    p.has_permutation = r.option();
    if (p.has_permutation) p.permutation_commit = r.cap();
    p.quotient_commit = r.cap();
    p.has_random = r.option();
    if (p.has_random) p.random_commit = r.cap();
  };
  auto opened = [&] {
    size_t ni = r.len();
    p.opened.resize(ni);
    for (auto& ov : p.opened)
      for (int k = 0; k < 8; ++k) switch (L.opened[k]) {
        case 0: ov.trace_local = r.vec_ef(); break;
        case 1: ov.has_trace_next = r.option(); if (ov.has_trace_next) ov.trace_next = r.vec_ef(); break;
        case 2: if (r.option()) ov.preprocessed_local = r.vec_ef(); break;
        case 3: if (r.option()) ov.preprocessed_next = r.vec_ef(); break;
        case 4: {
          size_t nc = r.len();
          ov.quotient_chunks.resize(nc);
          for (auto& c : ov.quotient_chunks) c = r.vec_ef();
          break;
        }
        case 5: ov.has_random = r.option(); if (ov.has_random) ov.random = r.vec_ef(); break;
        case 6: ov.permutation_local = r.vec_ef(); break;
        default: ov.permutation_next = r.vec_ef(); break;
      }
  };
  auto& f = p.fri;
  auto queries = [&] {
    size_t nq = r.len();
    f.query_proofs.resize(nq);
    for (auto& q : f.query_proofs) {
      size_t nb = r.len();
      q.input_proof.resize(nb);
      for (auto& bo : q.input_proof) {
        size_t nm = r.len();
        bo.opened_values.resize(nm);
        for (auto& row : bo.opened_values) row = r.vec_fe();
        if (salted) { bo.salts.resize(r.len()); for (auto& sl : bo.salts) sl = r.vec_fe(); }
        size_t nd = r.len();
        bo.opening_proof.resize(nd);
        for (auto& d : bo.opening_proof) d = r.digest();
      }
      size_t ns = r.len();
      q.commit_phase_openings.resize(ns);
      for (auto& s : q.commit_phase_openings) {
        s.log_arity = r.byte();
        s.sibling_values = r.vec_ef();
        if (salted) { s.salts.resize(r.len()); for (auto& sl : s.salts) sl = r.vec_fe(); }
        size_t nd = r.len();
        s.opening_proof.resize(nd);
        for (auto& d : s.opening_proof) d = r.digest();
      }
    }
  };
  auto fri = [&] {
    if (zk) {
      p.fri_random.resize(r.len());
      for (auto& rd : p.fri_random) {
        rd.resize(r.len());
        for (auto& m : rd) {
          m.resize(r.len());
          for (auto& pt : m) pt = r.vec_ef();
        }
      }
    }
    for (int k = 0; k < 5; ++k) switch (L.fri[k]) {
      case 0: {
        size_t np = r.len();
        f.commit_phase_commits.resize(np);
        for (auto& c : f.commit_phase_commits) c = r.cap();
        break;
      }
      case 1: f.commit_pow_witnesses = r.vec_fe(); break;
      case 2: queries(); break;
      case 3: f.final_poly = r.vec_ef(); break;
      default: f.query_pow_witness = r.fe(); break;
    }
  };
  for (int k = 0; k < 5; ++k) switch (L.batch[k]) {
    case 0: commitments(); break;
    case 1: opened(); break;
    case 2: fri(); break;
    case 3: {
      size_t nt = r.len();
      p.has_terminal.assign(nt, false);
      p.lookup_terminals.assign(nt, Fe4<FP>::zero());
      for (size_t i = 0; i < nt; ++i) {
        p.has_terminal[i] = r.option();
        if (p.has_terminal[i]) p.lookup_terminals[i] = r.ef();
      }
      break;
    }
    default: {
      size_t nd = r.len();
      for (size_t i = 0; i < nd; ++i) p.degree_bits.push_back((size_t)r.varint());
      break;
    }
  }
  if (r.p != r.end) throw std::runtime_error("trailing bytes after proof");
  return p;
}

}  // namespace orc
