// Driver of the sanitizer build (tests/san/Makefile): a structure-aware mutator over REAL proofs and circuits, run
// in-process against the ASan + UBSan build of the library's host-only code (san_host.hip).  Test infrastructure.
//
//   san_driver proofs  <case file> <iterations> <seed>     proof bytes  -> p3r_batch_stark_proof_parse, p3r_batch_proof_len_layout,
//                                                           p3r::BatchStarkProof::from_postcard / to_postcard (include/p3r.hpp),
//                                                           p3r_verify_batch (when the bytes still parse)
//   san_driver mmcs    <case file> <iterations> <seed>     p3r_mmcs_verify on the openings inside the case's proof, with mutated
//                                                           shapes, indices, caps and configurations
//   san_driver circuit <case file> <iterations> <seed>     p3r_circuit_desc     -> validate_circuit + circuit_tables + build_schedule
//
// The mutator knows the postcard grammar of BatchStarkProof (the inner BatchProof of recursion/src/types/proof.rs:403-409 +
// the metadata of circuit-prover/src/batch_stark_prover.rs:610-636): it edits lengths, option tags, field elements,
// varint encodings, vector contents (drop / duplicate / swap / truncate), strings and the packing / row-count fields the
// validation rules are about (batch_stark_prover.rs:459-488,666-681, packing.rs:140-161), besides unstructured byte flips.
// A mutant must be REFUSED or - if it is still the same proof with different metadata - verify; it must never be accepted
// with different inner bytes, and nothing may trip a sanitizer (the process aborts: -fno-sanitize-recover).
// Exit code 0 and one summary line on success.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/p3r.hpp"

extern "C" int san_circuit_host_prep(const p3r_circuit_desc* d, uint32_t field, uint32_t ext_degree, char* err, size_t err_cap);

namespace {

struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint64_t below(uint64_t n) { return n ? next() % n : 0; }
};

std::vector<uint8_t> read_file(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  std::vector<uint8_t> b;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
  fclose(f);
  return b;
}
struct In {
  const uint8_t* p;
  const uint8_t* e;
  template <class T> T get() {
    if ((size_t)(e - p) < sizeof(T)) { fprintf(stderr, "case file truncated\n"); exit(2); }
    T v; memcpy(&v, p, sizeof(T)); p += sizeof(T); return v;
  }
  std::vector<uint32_t> words(size_t n) { std::vector<uint32_t> v(n); for (auto& x : v) x = get<uint32_t>(); return v; }
};

void put_varint(std::vector<uint8_t>& o, uint64_t v) {
  while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; }
  o.push_back((uint8_t)v);
}

// ---- the grammar, as a tree of byte spans
enum Kind { K_STRUCT, K_VEC, K_OPT, K_FE, K_VARINT, K_BYTE, K_STR, K_LEN, K_TAG };
struct Node { Kind kind; size_t b, e; std::vector<int> kids; };
struct Tree {
  const std::vector<uint8_t>& d;
  std::vector<Node> n;
  size_t at = 0;
  bool ok = true;
  int commit_pow = -1, query_pow = -1, inner_end = -1;   // nodes: Vec of commit-phase PoW witnesses, the query witness; last node of the inner proof
  explicit Tree(const std::vector<uint8_t>& data) : d(data) {}
  int open(Kind k) { n.push_back({k, at, at, {}}); return (int)n.size() - 1; }
  void close(int id) { n[id].e = at; }
  uint64_t varint_raw() {
    uint64_t v = 0;
    for (int s = 0; s < 70; s += 7) {
      if (at >= d.size()) { ok = false; return 0; }
      uint8_t b = d[at++];
      if (s < 64) v |= (uint64_t)(b & 0x7F) << s;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  int leaf(Kind k, uint64_t* out = nullptr) {
    int id = open(k);
    uint64_t v = 0;
    if (k == K_BYTE || k == K_TAG) { if (at >= d.size()) ok = false; else v = d[at++]; }
    else v = varint_raw();
    close(id);
    if (out) *out = v;
    return id;
  }
  template <class F> int vec(F item, size_t cap = 1u << 20) {
    int id = open(K_VEC);
    uint64_t len = 0;
    { const int k = leaf(K_LEN, &len); n[id].kids.push_back(k); }   // (leaf() may reallocate `n`)
    if (len > cap) ok = false;
    for (uint64_t i = 0; ok && i < len; ++i) { int k = item(); n[id].kids.push_back(k); }
    close(id);
    return id;
  }
  template <class F> int opt(F item) {
    int id = open(K_OPT);
    uint64_t t = 0;
    { const int k = leaf(K_TAG, &t); n[id].kids.push_back(k); }
    if (t > 1) ok = false;
    if (ok && t == 1) { int k = item(); n[id].kids.push_back(k); }
    close(id);
    return id;
  }
  template <class F> int strct(F body) { int id = open(K_STRUCT); body(id); close(id); return id; }
  void add(int parent, int kid) { n[parent].kids.push_back(kid); }   // `kid` is evaluated by the caller BEFORE this call: pass a value, not T.leaf(..) inline with n[..]
};

int parse_outer(Tree& T, int dc, bool zk, bool salted = false) {
  auto fe = [&] { return T.leaf(K_FE); };
  auto ef = [&] { return T.strct([&](int id) { for (int i = 0; i < dc && T.ok; ++i) T.add(id, fe()); }); };
  auto vec_ef = [&] { return T.vec(ef); };
  auto digest = [&] { return T.strct([&](int id) { for (int i = 0; i < 8 && T.ok; ++i) T.add(id, fe()); }); };
  auto cap = [&] { return T.vec(digest); };
  auto str = [&] {
    int id = T.open(K_STR);
    uint64_t len = 0;
    { const int k = T.leaf(K_LEN, &len); T.n[id].kids.push_back(k); }
    if (len > T.d.size() - T.at) T.ok = false; else T.at += len;
    T.close(id);
    return id;
  };
  return T.strct([&](int root) {
    // ---- inner BatchProof
    T.add(root, T.strct([&](int c) { T.add(c, cap()); T.add(c, T.opt(cap)); T.add(c, cap()); T.add(c, T.opt(cap)); }));
    T.add(root, T.vec([&] {
      return T.strct([&](int o) {
        T.add(o, vec_ef()); T.add(o, T.opt(vec_ef)); T.add(o, T.opt(vec_ef)); T.add(o, T.opt(vec_ef));
        T.add(o, T.vec(vec_ef)); T.add(o, T.opt(vec_ef)); T.add(o, vec_ef()); T.add(o, vec_ef());
      });
    }));
    T.add(root, T.strct([&](int f) {
      if (zk) T.add(f, T.vec([&] { return T.vec([&] { return T.vec(vec_ef); }); }));
      T.add(f, T.vec(cap));
      T.commit_pow = T.vec(fe);
      T.add(f, T.commit_pow);
      T.add(f, T.vec([&] {
        return T.strct([&](int q) {
          // a hiding MMCS's opening proof is the tuple (salts, siblings): one more Vec<Vec<F>> in front of the digests
          T.add(q, T.vec([&] { return T.strct([&](int b) {
            T.add(b, T.vec([&] { return T.vec(fe); }));
            if (salted) T.add(b, T.vec([&] { return T.vec(fe); }));
            T.add(b, T.vec(digest)); }); }));
          T.add(q, T.vec([&] { return T.strct([&](int s) {
            T.add(s, T.leaf(K_BYTE)); T.add(s, vec_ef());
            if (salted) T.add(s, T.vec([&] { return T.vec(fe); }));
            T.add(s, T.vec(digest)); }); }));
        });
      }));
      T.add(f, vec_ef());
      T.query_pow = fe();
      T.add(f, T.query_pow);
    }));
    T.add(root, T.vec([&] { return T.opt(ef); }));
    T.inner_end = T.vec([&] { return T.leaf(K_VARINT); });
    T.add(root, T.inner_end);
    // ---- metadata (batch_stark_prover.rs:610-636)
    T.add(root, T.strct([&](int m) {
      T.add(m, T.leaf(K_VARINT)); T.add(m, T.leaf(K_VARINT));                                   // public_lanes, alu_lanes
      T.add(m, T.vec([&] { return T.strct([&](int e) { T.add(e, str()); T.add(e, T.leaf(K_VARINT)); }); }));   // npo_lanes
      T.add(m, T.leaf(K_VARINT)); T.add(m, T.leaf(K_VARINT));                                   // min_trace_height, horner_packed_steps
      for (int i = 0; i < 3; ++i) T.add(m, T.leaf(K_VARINT));                                   // RowCounts
      T.add(m, T.leaf(K_VARINT)); T.add(m, T.leaf(K_VARINT));                                   // alu_variant, ext_degree
      T.add(m, T.opt(fe)); T.add(m, T.leaf(K_TAG));                                             // w_binomial, alu_quintic_trinomial
      T.add(m, T.vec([&] {                                                                       // non_primitives
        return T.strct([&](int e) { T.add(e, str()); T.add(e, T.leaf(K_VARINT)); T.add(e, T.leaf(K_VARINT)); T.add(e, T.vec(fe)); T.add(e, T.leaf(K_VARINT)); });
      }));
      T.add(m, T.opt([&] {                                                                       // stark_common
        return T.strct([&](int c) {
          T.add(c, cap());
          T.add(c, T.vec([&] { return T.opt([&] { return T.strct([&](int i) { for (int k = 0; k < 3; ++k) T.add(i, T.leaf(K_VARINT)); }); }); }));
          T.add(c, T.vec([&] { return T.leaf(K_VARINT); }));
        });
      }));
    }));
  });
}

struct Case {
  p3r_config cfg{};
  std::vector<p3r_air_desc> airs;
  std::vector<uint32_t> cap, degree_bits;
  std::vector<uint8_t> outer;
  uint32_t flags = 0;
  int dc = 4;
};
Case load_case(const char* path) {
  auto raw = read_file(path);
  In in{raw.data(), raw.data() + raw.size()};
  if (raw.size() < 8 || memcmp(in.p, "P3RSAN1", 8) != 0) { fprintf(stderr, "%s is not a proof case\n", path); exit(2); }
  in.p += 8;
  Case c;
  c.cfg.abi_version = P3R_ABI_VERSION;
  c.cfg.field = in.get<uint32_t>(); c.cfg.ext_degree = in.get<uint32_t>(); c.cfg.log_blowup = in.get<uint32_t>();
  c.cfg.max_log_arity = in.get<uint32_t>(); c.cfg.cap_height = in.get<uint32_t>(); c.cfg.log_final_poly_len = in.get<uint32_t>();
  c.cfg.commit_pow_bits = in.get<uint32_t>(); c.cfg.query_pow_bits = in.get<uint32_t>(); c.cfg.num_queries = in.get<uint32_t>();
  c.cfg.challenge_degree = in.get<uint32_t>(); c.cfg.mmcs_arity = in.get<uint32_t>(); c.cfg.zk = in.get<uint32_t>();
  c.cfg.num_random_codewords = in.get<uint32_t>();
  c.cfg.mmcs_salt_elems = in.get<uint32_t>();
  c.cfg.ext_choices = P3R_EXT_UNPINNED_W32_DEFAULTS;   // the cases are made with the library's built-in width-32 constants
  const uint32_t na = in.get<uint32_t>();
  for (uint32_t i = 0; i < na; ++i) { p3r_air_desc a{}; a.kind = in.get<uint32_t>(); a.lanes = in.get<uint32_t>(); a.horner_packed_steps = in.get<uint32_t>(); a.coeff_lookups = in.get<uint32_t>(); c.airs.push_back(a); }
  c.cap = in.words(in.get<uint32_t>());
  c.degree_bits = in.words(in.get<uint32_t>());
  const uint64_t len = in.get<uint64_t>();
  if ((uint64_t)(in.e - in.p) < len) { fprintf(stderr, "case file truncated\n"); exit(2); }
  c.outer.assign(in.p, in.p + len);
  c.dc = c.cfg.challenge_degree == 5 ? 5 : 4;
  c.flags = (c.dc == 5 ? P3R_PROOF_QUINTIC_CHALLENGE : 0) | (c.cfg.zk ? P3R_PROOF_ZK : 0) | (c.cfg.mmcs_salt_elems ? P3R_PROOF_SALTED : 0);
  return c;
}

const uint64_t kSpecial[] = {0, 1, 2, 3, 7, 8, 16, 63, 64, 65, 127, 128, 255, 256, 1023, 0x7F000000ull, 0x7F000001ull, 0x78000000ull, 0x78000001ull,
                             0x7FFFFFFFull, 0x80000000ull, 0xFFFFFFFFull, 0x100000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull};

// one structured edit of `d` (tree T is of the ORIGINAL bytes); returns the mutant
std::vector<uint8_t> mutate(const std::vector<uint8_t>& d, const Tree& T, Rng& r) {
  std::vector<uint8_t> o;
  auto splice = [&](size_t b, size_t e, const std::vector<uint8_t>& with) {
    o.assign(d.begin(), d.begin() + b);
    o.insert(o.end(), with.begin(), with.end());
    o.insert(o.end(), d.begin() + e, d.end());
  };
  const int op = (int)r.below(20);
  if (op == 0) {   // unstructured: flip / overwrite / insert / delete a few bytes
    o = d;
    for (int k = 0, n = 1 + (int)r.below(3); k < n && !o.empty(); ++k) {
      const size_t i = r.below(o.size());
      switch (r.below(4)) {
        case 0: o[i] ^= (uint8_t)(1u << r.below(8)); break;
        case 1: o[i] = (uint8_t)r.next(); break;
        case 2: o.insert(o.begin() + i, (uint8_t)r.next()); break;
        default: o.erase(o.begin() + i); break;
      }
    }
    return o;
  }
  if (op == 1) {   // truncate at a field boundary / append
    const Node& x = T.n[r.below(T.n.size())];
    o.assign(d.begin(), d.begin() + (r.below(2) ? x.b : x.e));
    if (r.below(4) == 0) for (int k = 0, n = 1 + (int)r.below(9); k < n; ++k) o.push_back((uint8_t)r.next());
    return o;
  }
  // pick a node of a kind the remaining operations apply to
  for (int tries = 0; tries < 64; ++tries) {
    const Node& x = T.n[r.below(T.n.size())];
    std::vector<uint8_t> w;
    switch (x.kind) {
      case K_FE: case K_VARINT: case K_LEN: {
        uint64_t v = kSpecial[r.below(sizeof kSpecial / sizeof *kSpecial)];
        if (r.below(3) == 0) v = r.next() >> r.below(64);
        if (r.below(8) == 0) {   // a non-canonical (over-long) encoding of a small value
          w.push_back((uint8_t)((v & 0x7F) | 0x80));
          for (int k = 0, n = (int)r.below(10); k < n; ++k) w.push_back(0x80);
          w.push_back(0);
        } else put_varint(w, v);
        splice(x.b, x.e, w);
        return o;
      }
      case K_TAG: case K_BYTE: {
        w.push_back(r.below(2) ? (uint8_t)r.below(4) : (uint8_t)r.next());
        splice(x.b, x.e, w);
        return o;
      }
      case K_OPT: {
        const Node& tag = T.n[x.kids[0]];
        if (x.kids.size() == 2) { w.push_back(0); splice(x.b, r.below(2) ? x.e : tag.e, w); }   // None, payload dropped or left behind
        else { w.push_back(1); if (r.below(2)) for (int k = 0, n = (int)r.below(40); k < n; ++k) w.push_back((uint8_t)r.next()); splice(x.b, x.e, w); }
        return o;
      }
      case K_VEC: {
        const size_t n = x.kids.size() - 1;
        const Node& len = T.n[x.kids[0]];
        const int what = (int)r.below(5);
        if (n == 0 && what != 4) break;
        auto elem = [&](size_t i) { const Node& e = T.n[x.kids[1 + i]]; return std::make_pair(e.b, e.e); };
        put_varint(w, what == 0 ? n - 1 : what == 1 ? n + 1 : n);
        if (what == 0) {          // drop one element
          const size_t k = r.below(n);
          for (size_t i = 0; i < n; ++i) if (i != k) w.insert(w.end(), d.begin() + elem(i).first, d.begin() + elem(i).second);
        } else if (what == 1) {   // duplicate one
          const size_t k = r.below(n);
          for (size_t i = 0; i < n; ++i) {
            w.insert(w.end(), d.begin() + elem(i).first, d.begin() + elem(i).second);
            if (i == k) w.insert(w.end(), d.begin() + elem(i).first, d.begin() + elem(i).second);
          }
        } else if (what == 2) {   // swap two
          if (n < 2) break;
          size_t a = r.below(n), b = r.below(n);
          for (size_t i = 0; i < n; ++i) { const size_t s = i == a ? b : i == b ? a : i; w.insert(w.end(), d.begin() + elem(s).first, d.begin() + elem(s).second); }
        } else if (what == 3) {   // keep the length, lose the tail
          const size_t keep = r.below(n);
          for (size_t i = 0; i < keep; ++i) w.insert(w.end(), d.begin() + elem(i).first, d.begin() + elem(i).second);
        } else {                  // claim one more element than there is
          w.clear();
          put_varint(w, n + 1 + r.below(3));
          w.insert(w.end(), d.begin() + len.e, d.begin() + x.e);
        }
        splice(x.b, x.e, w);
        return o;
      }
      case K_STR: {
        const Node& len = T.n[x.kids[0]];
        std::string s(d.begin() + len.e, d.begin() + x.e);
        switch (r.below(6)) {
          case 0: s = ""; break;
          case 1: s.assign(1 + r.below(300), 'a'); break;
          case 2: if (!s.empty()) s[r.below(s.size())] = (char)(0x80 | r.below(0x80)); break;   // invalid UTF-8
          case 3: s += '\0'; break;
          case 4: s = "recompose"; break;
          default: s = "poseidon2_perm/koala_bear_d4_w32"; break;
        }
        put_varint(w, r.below(8) ? s.size() : s.size() + 1 + r.below(100));
        w.insert(w.end(), s.begin(), s.end());
        splice(x.b, x.e, w);
        return o;
      }
      default: break;
    }
  }
  o = d;
  if (!o.empty()) o[r.below(o.size())] ^= 0x40;
  return o;
}

// The inner proof with the proof-of-work witnesses of ZERO-bit grinds blanked: `check_pow_witness` leaves the transcript
// unchanged when no proof of work is required (recursion/src/challenger/circuit.rs:409-430: "When no PoW is required, keep
// challenger state unchanged"), so such a witness is not bound by the proof - in the reference as here.  Any other accepted
// difference is a finding.
bool normalised_inner(const std::vector<uint8_t>& d, int dc, bool zk, bool salted, bool blank_commit, bool blank_query, std::vector<uint8_t>& out) {
  Tree T(d);
  parse_outer(T, dc, zk, salted);
  if (!T.ok || T.commit_pow < 0 || T.query_pow < 0 || T.inner_end < 0) return false;
  out.clear();
  size_t at = 0;
  auto blank = [&](const Node& x) { out.insert(out.end(), d.begin() + at, d.begin() + x.b); out.push_back(0); at = x.e; };
  if (blank_commit) for (size_t i = 1; i < T.n[T.commit_pow].kids.size(); ++i) blank(T.n[T.n[T.commit_pow].kids[i]]);
  if (blank_query) blank(T.n[T.query_pow]);
  const size_t end = T.n[T.inner_end].e;
  out.insert(out.end(), d.begin() + at, d.begin() + end);
  return true;
}

int run_proofs(const char* path, uint64_t iters, uint64_t seed) {
  Case c = load_case(path);
  char err[512];
  p3r_batch_stark_meta meta;
  if (p3r_batch_stark_proof_parse(c.cfg.field, c.outer.data(), c.outer.size(), c.flags, nullptr, &meta, err, sizeof err) != P3R_OK) {
    fprintf(stderr, "the seed proof does not parse: %s\n", err);
    return 1;
  }
  const std::vector<uint8_t> inner(c.outer.begin(), c.outer.begin() + meta.proof_len);
  if (p3r_verify_batch(&c.cfg, c.airs.data(), c.airs.size(), c.cap.data(), c.degree_bits.data(), inner.data(), inner.size(), 0, err, sizeof err) != P3R_OK) {
    fprintf(stderr, "the seed proof is rejected: %s\n", err);
    return 1;
  }
  Tree T(c.outer);
  parse_outer(T, c.dc, c.cfg.zk != 0, c.cfg.mmcs_salt_elems != 0);
  if (!T.ok || T.at != c.outer.size()) { fprintf(stderr, "the mutator's grammar does not cover the seed proof (%zu of %zu bytes)\n", T.at, c.outer.size()); return 1; }
  Rng r{seed};
  uint64_t parsed = 0, verified_same = 0, rejected_parse = 0, rejected_verify = 0, hpp_ok = 0, unbound_pow = 0;
  const bool blank_c = c.cfg.commit_pow_bits == 0, blank_q = c.cfg.query_pow_bits == 0;
  std::vector<uint8_t> seed_norm;
  if (!normalised_inner(c.outer, c.dc, c.cfg.zk != 0, c.cfg.mmcs_salt_elems != 0, blank_c, blank_q, seed_norm)) { fprintf(stderr, "cannot normalise the seed proof\n"); return 1; }
  for (uint64_t it = 0; it < iters; ++it) {
    std::vector<uint8_t> m = mutate(c.outer, T, r);
    if (r.below(16) == 0) { Tree T2(m); parse_outer(T2, c.dc, c.cfg.zk != 0, c.cfg.mmcs_salt_elems != 0); if (T2.ok) m = mutate(m, T2, r); }   // two edits
    if (m == c.outer) continue;
    // exact-size heap copy: ASan then sees any read past the end
    uint8_t* buf = (uint8_t*)malloc(m.size() ? m.size() : 1);
    if (!m.empty()) memcpy(buf, m.data(), m.size());   // (an empty mutant has a null data(): memcpy's arguments must not be - UBSan, at 10 x the default volume)
    size_t plen = 0;
    (void)p3r_batch_proof_len_layout(c.cfg.field, buf, m.size(), c.flags, nullptr, &plen, err, sizeof err);
    const int rc = p3r_batch_stark_proof_parse(c.cfg.field, buf, m.size(), c.flags, nullptr, &meta, err, sizeof err);
    if (rc != P3R_OK) { ++rejected_parse; free(buf); continue; }
    ++parsed;
    // the C++ mirror (include/p3r.hpp): parse -> object -> bytes must reproduce what was parsed
    try {
      auto p = p3r::BatchStarkProof::from_postcard(std::vector<uint8_t>(buf, buf + m.size()), (p3r::Field)c.cfg.field, true, (uint32_t)c.dc, c.cfg.zk != 0, c.cfg.mmcs_salt_elems != 0);
      (void)p.airs();
      ++hpp_ok;
    } catch (const std::exception&) {}
    const bool same_inner = meta.proof_len == inner.size() && memcmp(buf, inner.data(), inner.size()) == 0;
    uint8_t* ib = (uint8_t*)malloc(meta.proof_len ? meta.proof_len : 1);
    memcpy(ib, buf, meta.proof_len);
    const int vrc = p3r_verify_batch(&c.cfg, c.airs.data(), c.airs.size(), c.cap.data(), c.degree_bits.data(), ib, meta.proof_len, 0, err, sizeof err);
    free(ib);
    free(buf);
    if (vrc == P3R_OK) {
      std::vector<uint8_t> mn;
      if (!same_inner && (blank_c || blank_q) && normalised_inner(m, c.dc, c.cfg.zk != 0, c.cfg.mmcs_salt_elems != 0, blank_c, blank_q, mn) && mn == seed_norm) {
        ++unbound_pow;   // differs only in a witness of a zero-bit grind
      } else if (!same_inner) {
        fprintf(stderr, "iteration %llu (seed %llu): a mutant with different proof bytes was ACCEPTED\n", (unsigned long long)it, (unsigned long long)seed);
        if (const char* dump = getenv("P3R_SAN_DUMP")) {   // the mutant, for tests/proof_codec.py to diff against the seed
          if (FILE* f = fopen(dump, "wb")) { fwrite(m.data(), 1, m.size(), f); fclose(f); }
        }
        return 1;
      } else ++verified_same;
    } else {
      if (same_inner) { fprintf(stderr, "iteration %llu: the unchanged inner proof was rejected: %s\n", (unsigned long long)it, err); return 1; }
      ++rejected_verify;
    }
  }
  printf("{\"mode\": \"proofs\", \"iterations\": %llu, \"rejected_by_parser\": %llu, \"parsed\": %llu, \"rejected_by_verifier\": %llu, "
         "\"same_proof_other_metadata_verified\": %llu, \"zero_bit_pow_witness_changed_verified\": %llu, \"hpp_round_trips\": %llu}\n",
         (unsigned long long)iters, (unsigned long long)rejected_parse, (unsigned long long)parsed, (unsigned long long)rejected_verify,
         (unsigned long long)verified_same, (unsigned long long)unbound_pow, (unsigned long long)hpp_ok);
  return 0;
}

// p3r_mmcs_verify: the openings of the case's first query, then mutated shapes / indices / caps / configurations
int run_mmcs(const char* path, uint64_t iters, uint64_t seed) {
  Case c = load_case(path);
  Rng r{seed};
  char err[512];
  uint64_t ok = 0, bad = 0;
  const uint32_t P = c.cfg.field == P3R_FIELD_KOALA_BEAR ? 0x7F000001u : 0x78000001u;
  for (uint64_t it = 0; it < iters; ++it) {
    p3r_config cfg = c.cfg;
    const size_t n_mats = 1 + r.below(4);
    std::vector<size_t> hs(n_mats), ws(n_mats);
    size_t total = 0;
    // sizes are either small (the buffers below cover every read of a well-formed call) or absurd (the call must be
    // refused BEFORE anything is read: widths beyond 2^24, proofs longer than any tree's, caps taller than the tree)
    for (size_t m = 0; m < n_mats; ++m) {
      hs[m] = r.below(12) ? size_t(1) << r.below(12) : (size_t)r.next();
      ws[m] = r.below(12) ? 1 + r.below(40) : (size_t(1) << 24) + 1 + (size_t)(r.next() >> r.below(40));
      if (ws[m] <= 40) total += ws[m];
    }
    switch (r.below(8)) {
      case 0: cfg.cap_height = (uint32_t)r.below(80); break;
      case 1: cfg.mmcs_arity = (uint32_t)r.below(6); break;
      case 2: cfg.cap_height = 0; cfg.mmcs_arity = 4; break;
      case 3: cfg.field = (uint32_t)r.below(3); break;
      default: break;
    }
    const size_t proof_len = r.below(10) ? r.below(16) : 193 + (size_t)(r.next() >> r.below(60));
    const size_t cap_digests = cfg.cap_height < 12 ? size_t(1) << cfg.cap_height : 1;
    std::vector<uint32_t> cap(8 * cap_digests), opened(total ? total : 1), proof(8 * (proof_len < 16 ? proof_len : 0) + 8);
    for (auto& x : cap) x = r.below(9) ? (uint32_t)r.below(P) : (uint32_t)r.next();
    for (auto& x : opened) x = r.below(30) ? (uint32_t)r.below(P) : (uint32_t)r.next();
    for (auto& x : proof) x = (uint32_t)r.below(P);
    const int rc = p3r_mmcs_verify(&cfg, cap.data(), n_mats, hs.data(), ws.data(), (size_t)r.next() >> r.below(64), opened.data(),
                                   proof.data(), proof_len, err, sizeof err);
    (rc == P3R_OK ? ok : bad)++;
  }
  printf("{\"mode\": \"mmcs\", \"iterations\": %llu, \"accepted\": %llu, \"refused\": %llu}\n", (unsigned long long)iters, (unsigned long long)ok, (unsigned long long)bad);
  return ok ? 1 : 0;   // random digests never open a random cap
}

int run_circuit(const char* path, uint64_t iters, uint64_t seed) {
  auto raw = read_file(path);
  In in{raw.data(), raw.data() + raw.size()};
  if (raw.size() < 8 || memcmp(in.p, "P3RSANC", 8) != 0) { fprintf(stderr, "%s is not a circuit case\n", path); return 2; }
  in.p += 8;
  const uint32_t field = in.get<uint32_t>(), ext_degree = in.get<uint32_t>();
  p3r_circuit_desc base{};
  base.witness_count = in.get<uint32_t>();
  base.public_lanes = in.get<uint32_t>(); base.alu_lanes = in.get<uint32_t>(); base.horner_packed_steps = in.get<uint32_t>();
  base.recompose_lanes = in.get<uint32_t>(); base.min_trace_height = in.get<uint32_t>();
  const uint64_t n_ops = in.get<uint64_t>();
  std::vector<uint32_t> ops = in.words(8 * n_ops);
  std::vector<uint32_t> ext = in.words(in.get<uint64_t>()), pub = in.words(in.get<uint64_t>()), priv = in.words(in.get<uint64_t>());
  std::vector<uint32_t> rew = in.words(2 * in.get<uint64_t>());
  static_assert(sizeof(p3r_op) == 32, "p3r_op is eight words");
  char err[512];
  auto run = [&](const p3r_circuit_desc& d, const std::vector<uint32_t>& o, const std::vector<uint32_t>& e, const std::vector<uint32_t>& pu,
                 const std::vector<uint32_t>& pr, const std::vector<uint32_t>& rw, uint32_t dg) {
    // exact-size heap copies: an over-read is a sanitizer report
    auto dup = [](const std::vector<uint32_t>& v) { uint32_t* p = (uint32_t*)malloc(v.size() * 4 + (v.empty() ? 4 : 0)); if (!v.empty()) memcpy(p, v.data(), v.size() * 4); return p; };
    uint32_t *po = dup(o), *pe = dup(e), *ppu = dup(pu), *ppr = dup(pr), *prw = dup(rw);
    p3r_circuit_desc x = d;
    x.n_ops = o.size() / 8; x.ops = (const p3r_op*)po; x.n_ext = e.size(); x.ext = pe; x.n_public = pu.size(); x.public_rows = ppu;
    x.n_private = pr.size(); x.private_input_rows = ppr; x.n_rewrite = rw.size() / 2; x.witness_rewrite = prw;
    const int rc = san_circuit_host_prep(&x, field, dg, err, sizeof err);
    free(po); free(pe); free(ppu); free(ppr); free(prw);
    return rc;
  };
  if (run(base, ops, ext, pub, priv, rew, ext_degree) != 0) { fprintf(stderr, "the seed circuit is refused: %s\n", err); return 1; }
  Rng r{seed};
  uint64_t ok = 0, bad = 0;
  const uint32_t nw = base.witness_count;
  const uint32_t special[] = {0, 1, 2, nw - 1, nw, nw + 1, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu, (uint32_t)ext.size(), (uint32_t)ext.size() - 1,
                              (uint32_t)ext.size() + 1, (uint32_t)n_ops, (uint32_t)n_ops - 1, 4, 5, 8, 16, 31, 124, 125};
  auto sp = [&] { return r.below(4) ? special[r.below(sizeof special / sizeof *special)] : (uint32_t)r.next(); };
  for (uint64_t it = 0; it < iters; ++it) {
    p3r_circuit_desc d = base;
    auto o = ops; auto e = ext; auto pu = pub; auto pr = priv; auto rw = rew;
    uint32_t dg = ext_degree;
    for (int k = 0, n = 1 + (int)r.below(3); k < n; ++k) {
      switch (r.below(14)) {
        case 0: case 1: case 2: case 3: if (!o.empty()) o[8 * r.below(o.size() / 8) + r.below(8)] = sp(); break;      // one field of one op
        case 4: if (!o.empty()) { const size_t i = 8 * r.below(o.size() / 8); o[i + 6] = sp(); o[i + 7] = sp(); } break;   // ext_off + ext_len (overflow)
        case 5: if (!e.empty()) e[r.below(e.size())] = sp(); break;
        case 6: d.witness_count = sp(); break;
        case 7: switch (r.below(5)) { case 0: d.public_lanes = sp(); break; case 1: d.alu_lanes = sp(); break; case 2: d.horner_packed_steps = sp(); break;
                                      case 3: d.recompose_lanes = sp(); break; default: d.min_trace_height = sp(); break; } break;
        case 8: if (!pu.empty()) pu[r.below(pu.size())] = sp(); else pu.push_back(sp()); break;
        case 9: if (!pr.empty()) pr[r.below(pr.size())] = sp(); else pr.push_back(sp()); break;
        case 10: rw.push_back(sp()); rw.push_back(sp()); break;
        case 11: if (o.size() >= 16) { const size_t a = 8 * r.below(o.size() / 8), b = 8 * r.below(o.size() / 8); for (int j = 0; j < 8; ++j) std::swap(o[a + j], o[b + j]); } break;
        case 12: if (o.size() >= 8) o.resize(8 * r.below(o.size() / 8)); break;                                        // truncated op list
        default: dg = (uint32_t[]){1, 4, 5, 0, 2, 7}[r.below(6)]; break;
      }
    }
    (run(d, o, e, pu, pr, rw, dg) == 0 ? ok : bad)++;
  }
  printf("{\"mode\": \"circuit\", \"iterations\": %llu, \"prepared\": %llu, \"refused\": %llu}\n", (unsigned long long)iters, (unsigned long long)ok, (unsigned long long)bad);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc != 5) { fprintf(stderr, "usage: %s proofs|mmcs|circuit <case file> <iterations> <seed>\n", argv[0]); return 2; }
  const uint64_t iters = strtoull(argv[3], nullptr, 10), seed = strtoull(argv[4], nullptr, 10);
  if (!strcmp(argv[1], "proofs")) return run_proofs(argv[2], iters, seed);
  if (!strcmp(argv[1], "mmcs")) return run_mmcs(argv[2], iters, seed);
  if (!strcmp(argv[1], "circuit")) return run_circuit(argv[2], iters, seed);
  return 2;
}
