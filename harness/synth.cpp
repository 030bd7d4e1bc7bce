// Synthetic recursion-layer workload generator (bench / test harness; neither product nor oracle).
//
// Produces the inputs the reference's `prove_all_tables(&Traces, &CircuitProverData)` consumes
// (circuit-prover/src/batch_stark_prover.rs:1203-1222), for the table mix SURVEY.md section 8d
// prescribes: Const, Public, ALU (Add / Mul / MulAdd / BoolCheck / HornerAcc chains),
// Poseidon2 circuit rows (sponge chains, Merkle paths) and Recompose, with witness indices and
// signed LogUp multiplicities assigned so that every send has matching receives
// (circuit-prover/src/common.rs:198-368 conventions: creators carry +n_reads, readers -1).
//
// Row semantics of the Poseidon2 table follow circuit/src/ops/poseidon_perm/executor.rs
// (SURVEY.md appendix C).  Values are uniform field elements from splitmix64.
//
// All arrays are exposed as flat canonical u32 through syn_get().
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "arith.h"  // the generator's own arithmetic: independent of the product and of the oracle

using namespace syn;

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
  double unit() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct Workload {
  std::map<std::string, std::vector<uint32_t>> arr;
  std::string err;
};

// shape knobs for the reference's edge cases: non-primitive tables without rows are left out of the
// batch, Public / ALU tables holding at most the dummy op run with one lane
// (batch_stark_prover.rs:1305-1318, tables/alu.rs:69-73)
enum { SYN_NO_POSEIDON2 = 1, SYN_NO_RECOMPOSE = 2, SYN_SINGLE_PUBLIC = 4, SYN_NO_ALU = 8,
       // sponge rows read only witnesses that no permutation produced: the sponge chains are independent of each
       // other (the leaf hashes of a verifier circuit: one chain per opened row), Merkle chains may still start
       // from a sponge's digest
       SYN_INDEPENDENT_SPONGES = 16,
       // the Recompose table is the "recompose/coeff" variant (per-coefficient bus tuples)
       SYN_RECOMPOSE_COEFF = 32,
       // both Recompose tables in one layer: every op is drawn `recompose` or `recompose/coeff`; the rows of the second
       // kind go to recompose_coeff_values / recompose_coeff_prep and counts[6]
       SYN_RECOMPOSE_BOTH = 64,
       // a width-32 Poseidon2 table next to the width-16 one (Poseidon2Config::*_D4_W32: the permutation of the arity-4
       // MMCS, circuit-prover/tests/arity4_mmcs.rs): arity-4 Merkle chains (4-to-1 compressions, injection levels with
       // zero pads, bridge levels) and rate-24 sponge chains.  D = 4 only; arrays p2w_*, counts[7].  The rows enter at
       // the prove_all_tables boundary (they are not ops of the flattened circuit)
       SYN_P2_W32 = 128,
       // with SYN_P2_W32: the width-32 rows ARE ops of the flattened circuit (P3R_OP_POSEIDON2_W32_PERM, kind 11), with the
       // executor's exact row semantics (poseidon_perm/executor.rs:92-235,947-966): no index-accumulator value, Merkle
       // chains that continue a leaf sponge (new_start = 0 on every compression row, the shape add_arity4_compression_row
       // emits, recursion/src/pcs/mmcs.rs:1013-1075) next to chains that start from a CTL-loaded digest; arrays
       // pdw_op_ids / pdw_siblings hold the private data (three sibling digests per Merkle row).  Bit 12: the bits 8-11
       // carry the extension degree
       SYN_P2_W32_OPS = 4096 };

// constants of the width-32 permutation (syn_set_w32): rc = [4][32] | partial | [4][32], diag = [32], canonical
static const uint32_t* g_w32_rc = nullptr;
static const uint32_t* g_w32_diag = nullptr;
template <class PP, class F>
void p2w_permute(F* s, const uint32_t* rc, const uint32_t* diag) {
  constexpr int W = 32;
  constexpr int PARTIAL = PP::P == 0x7f000001u ? 31 : 30;   // config.rs:88-100, :164-172
  static const int M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
  auto external = [&]() {
    F o[W];
    for (int i = 0; i < W; ++i) {
      F acc = F::zero();
      for (int j = 0; j < W; ++j) acc = acc + F::from_canonical((uint32_t)(M4[i % 4][j % 4] * (i / 4 == j / 4 ? 2 : 1))) * s[j];
      o[i] = acc;
    }
    for (int i = 0; i < W; ++i) s[i] = o[i];
  };
  auto sbox = [&](F x) { const F x3 = x * x * x; return PP::SBOX_DEGREE == 3 ? x3 : x3 * x3 * x; };
  external();
  int k = 0;
  auto full = [&]() {
    for (int i = 0; i < W; ++i) s[i] = sbox(s[i] + F::from_canonical(rc[k + i]));
    k += W;
    external();
  };
  for (int r = 0; r < 4; ++r) full();
  for (int r = 0; r < PARTIAL; ++r) {
    s[0] = sbox(s[0] + F::from_canonical(rc[k++]));
    F sum = F::zero();
    for (int i = 0; i < W; ++i) sum = sum + s[i];
    for (int i = 0; i < W; ++i) s[i] = s[i] * F::from_canonical(diag[i]) + sum;
  }
  for (int r = 0; r < 4; ++r) full();
}

enum { OP_ADD = 0, OP_MUL = 1, OP_BOOL = 2, OP_MULADD = 3, OP_HORNER = 4 };

// circuit op kinds of include/p3r.h (p3r_op_kind)
enum : uint32_t { C_CONST = 0, C_PUBLIC = 1, C_ADD = 2, C_MUL = 3, C_BOOL = 4, C_MULADD = 5, C_HORNER = 6,
                  C_HINT_EXT = 7, C_HINT_BIN = 8, C_P2 = 9, C_RECOMPOSE = 10, C_P2W = 11 };
constexpr uint32_t NO_W = 0xFFFFFFFFu;

template <class PP, class E>
void generate(Workload& W, int log_h, uint64_t seed, int horner_chain_len, int sponge_chain_len,
              int merkle_depth, const uint32_t* rc_canonical, uint32_t flags) {
  using F = Fp<PP>;
  constexpr int D = E::DEG;  // circuit extension degree: witness indices on the bus are scaled by D
  if ((D == 2 || D == 6 || D == 8) && !(flags & SYN_NO_POSEIDON2))
    throw std::runtime_error("ext_degree 2 / 6 / 8: no Poseidon2 table, pass SYN_NO_POSEIDON2");
  const uint32_t P = PP::P;
  const size_t H = size_t(1) << log_h;
  Rng rng(seed);
  auto rf = [&]() { return F::from_canonical((uint32_t)(rng.next() % P)); };
  auto re = [&]() { E e; for (int i = 0; i < D; ++i) e.c[i] = rf(); return e; };


  // ---- the circuit being built (flattened Circuit<EF>, include/p3r.h) ----
  auto& c_ops = W.arr["ops"];   // n x 8: kind, a, b, c, out, aux, ext_off, ext_len
  auto& c_ext = W.arr["ext"];
  auto push_op = [&](uint32_t kind, uint32_t a, uint32_t b, uint32_t c, uint32_t out, uint32_t aux,
                     const std::vector<uint32_t>& ext) {
    const uint32_t off = (uint32_t)c_ext.size();
    c_ext.insert(c_ext.end(), ext.begin(), ext.end());
    const uint32_t row[8] = {kind, a, b, c, out, aux, off, (uint32_t)ext.size()};
    c_ops.insert(c_ops.end(), row, row + 8);
  };
  uint32_t next_npo_id = 0;

  // witness table
  std::vector<E> wval;
  std::vector<uint32_t> reads;
  auto create = [&](const E& v) { wval.push_back(v); reads.push_back(0); return (uint32_t)(wval.size() - 1); };
  auto pick = [&]() { uint32_t w = rng.below((uint32_t)wval.size()); reads[w]++; return w; };
  auto put_e = [&](std::vector<uint32_t>& dst, const E& e) { for (int i = 0; i < D; ++i) dst.push_back(e.c[i].to_canonical()); };
  auto canon4 = [&](const E& e) { std::vector<uint32_t> v; put_e(v, e); return v; };

  // ---- Const (H/16 rows) and Public (H/2 ops) ----
  if ((flags & SYN_SINGLE_PUBLIC) && !(flags & SYN_NO_POSEIDON2))
    throw std::runtime_error("SYN_SINGLE_PUBLIC needs SYN_NO_POSEIDON2 (Merkle accumulators are public inputs)");
  const size_t n_const = std::max<size_t>(H / 16, 8);
  // H/2 ops in total: a few percent of them are added later, as targets of Poseidon2 outputs
  const size_t n_public = (flags & SYN_SINGLE_PUBLIC) ? 1 : std::max<size_t>(H / 2 - H / 16, 2);
  std::vector<uint32_t> const_w, public_w, base_valued;
  {
    // witness 0 is the zero constant (the lowerer's ExprId::ZERO), witness 1 is one; every other
    // constant alternates between a base-field value (coefficients of Recompose ops) and a full
    // extension value
    const_w.push_back(create(E::zero()));
    const_w.push_back(create(E::one()));
    while (const_w.size() < n_const) {
      const bool base = const_w.size() & 1;
      uint32_t w = create(base ? E::from_base(rf()) : re());
      if (base) base_valued.push_back(w);
      const_w.push_back(w);
    }
    for (uint32_t w : const_w) push_op(C_CONST, 0, 0, 0, w, 0, canon4(wval[w]));
  }

  // ---- Poseidon2 chain plan first (its accumulator values need Public witnesses) ----
  const size_t n_p2 = (flags & SYN_NO_POSEIDON2) ? 0 : std::max<size_t>(H / 2, 4);
  struct P2Plan { bool new_start, merkle, bit, mmcs_ctl; uint32_t acc; };
  std::vector<P2Plan> plan;
  {
    // ~70% Merkle rows, ~30% sponge rows
    while (plan.size() < n_p2) {
      bool merkle = rng.unit() < 0.7;
      size_t len = merkle ? (size_t)merkle_depth : (size_t)sponge_chain_len;
      len = std::min(len, n_p2 - plan.size());
      uint32_t acc = 0;
      bool ctl = merkle && rng.unit() < 0.5;
      for (size_t j = 0; j < len; ++j) {
        P2Plan p{};
        p.new_start = j == 0;
        p.merkle = merkle;
        p.bit = merkle ? (rng.next() & 1) : false;
        if (merkle) acc = (j == 0) ? 0u : ((acc * 2 + (p.bit ? 1 : 0)) % P);
        p.acc = acc;
        p.mmcs_ctl = ctl;
        plan.push_back(p);
      }
    }
  }
  // ends of Merkle chains with mmcs_ctl: the row sends (idx, acc); it is followed by a
  // new_start row (or by the table padding, whose first row carries new_start = 1).
  std::vector<int64_t> acc_wid(n_p2, -1);
  for (size_t r = 0; r < n_p2; ++r) {
    bool last_of_chain = (r + 1 == n_p2) || plan[r + 1].new_start;
    if (plan[r].merkle && plan[r].mmcs_ctl && last_of_chain) {
      E v = E::from_base(F::from_canonical(plan[r].acc));
      uint32_t w = create(v);
      reads[w]++;  // read by the Poseidon2 table's accumulator send
      acc_wid[r] = w;
      public_w.push_back(w);
    }
  }
  while (public_w.size() < n_public) public_w.push_back(create(re()));
  auto& public_rows = W.arr["public_rows"]; auto& in_public = W.arr["in_public_values"];
  for (size_t i = 0; i < public_w.size(); ++i) {
    push_op(C_PUBLIC, 0, 0, 0, public_w[i], (uint32_t)i, {});
    public_rows.push_back(public_w[i]);
    put_e(in_public, wval[public_w[i]]);
  }

  // ---- private inputs: set before the run, claimed on the bus by their first ALU use
  // (circuit.rs:247-250,351-359) ----
  const size_t n_private = (flags & SYN_NO_ALU) ? 0 : std::max<size_t>(H / 64, 2);
  std::vector<uint32_t> private_w;
  auto& private_rows = W.arr["private_rows"]; auto& in_private = W.arr["in_private_values"];
  for (size_t i = 0; i < n_private; ++i) {
    uint32_t w = create(re());
    private_w.push_back(w);
    private_rows.push_back(w);
    put_e(in_private, wval[w]);
  }
  // witnesses other tables may read right now (private inputs only after an ALU op claimed them)
  std::vector<uint32_t> pickable;
  for (uint32_t w : const_w) pickable.push_back(w);
  for (uint32_t w : public_w) pickable.push_back(w);
  auto pickp_noread = [&]() { return pickable[rng.below((uint32_t)pickable.size())]; };
  auto pickp = [&]() { uint32_t w = pickp_noread(); reads[w]++; return w; };

  // ---- Recompose (H/4 rows): packs D base-field witnesses into one extension witness
  // (ops/recompose.rs:115-170).  Plain table: the coefficient witnesses are not looked up.  SYN_RECOMPOSE_COEFF: the
  // "recompose/coeff" variant (recompose_air.rs:196-226, batch_stark_prover/recompose.rs:341-352) - every coefficient
  // is a bus tuple (D*idx, c, 0, ..) whose multiplicity is the read count of a coefficient no other table defines
  // (a hint output in the reference) and 0 otherwise ----
  const size_t n_rec = (flags & SYN_NO_RECOMPOSE) ? 0 : std::max<size_t>(H / 4, 2);
  const bool rec_both = (flags & SYN_RECOMPOSE_BOTH) != 0;
  const bool rec_coeff = (flags & SYN_RECOMPOSE_COEFF) != 0 || rec_both;
  std::vector<uint32_t> rec_w;
  std::vector<uint8_t> rec_dup;      // the output lands on a witness another table already defined: a reader (-1)
  std::vector<std::vector<uint32_t>> rec_ins;
  // coefficient witnesses whose only creator is a recompose/coeff row: the outputs of an ExtDecompositionHint
  // (circuit_builder.rs:1438-1463: decompose -> recompose/coeff -> connect).  Only Poseidon2 sponge inputs read them
  // afterwards: an ALU op would claim an undefined hint output as ITS creation (circuit.rs:351-359), and the
  // recompose/coeff preprocessing does not mark them defined (ops/recompose.rs:174-192)
  std::vector<uint32_t> rec_owned;
  std::vector<uint8_t> is_rec_out;   // 1: made by a plain `recompose` row, 2: by a `recompose/coeff` row
  std::vector<uint8_t> rec_kind;     // 1: the row belongs to the `recompose/coeff` table
  auto& rec_values = W.arr["recompose_values"];
  auto& rec2_values = W.arr["recompose_coeff_values"];
  for (size_t i = 0; i < n_rec; ++i) {
    std::vector<uint32_t> ins(D);
    E v;
    uint32_t w;
    bool dup = false;
    // RECOMPOSE_BOTH: about half the ops are the plain kind (their coefficients are never hint outputs: a plain row
    // does not put them on the bus)
    const bool coeff_row = rec_coeff && !(rec_both && rng.unit() < 0.5);
    if (coeff_row && rng.unit() < 0.4) {
      const uint32_t src = pickp_noread();
      v = wval[src];
      for (int k = 0; k < D; ++k) {
        ins[k] = create(E::from_base(v.c[k]));
        rec_owned.push_back(ins[k]);
      }
      push_op(C_HINT_EXT, src, 0, 0, 0, 0, ins);
      // connect(x, reconstructed): the op's output IS the decomposed witness - unless another row of this table
      // made it (dup_npo_outputs is kept per witness: both rows would become readers, circuit.rs:464-491)
      // (dup_npo_outputs is per op type: with two tables only an output of the SAME table - this row is of the
      // coefficient kind - is excluded; decomposing what a plain `recompose` row made and connecting back is the
      // builder's own pattern, circuit_builder.rs:1438-1463)
      const uint8_t made_by = src < is_rec_out.size() ? is_rec_out[src] : 0;
      dup = rng.unit() < 0.5 && (made_by == 0 || (rec_both && made_by == 1));
      if (dup) { w = src; reads[src]++; } else w = create(v);
    } else {
      for (int k = 0; k < D; ++k) {
        ins[k] = base_valued[rng.below((uint32_t)base_valued.size())];
        v.c[k] = wval[ins[k]].c[0];
      }
      w = create(v);
    }
    const bool second = rec_both && coeff_row;
    for (int k = 0; k < D; ++k) (second ? rec2_values : rec_values).push_back(v.c[k].to_canonical());
    rec_kind.push_back(second);
    rec_w.push_back(w);
    if (!dup) { is_rec_out.resize(wval.size(), 0); is_rec_out[w] = coeff_row ? 2 : 1; }
    rec_dup.push_back(dup);
    if (!dup) pickable.push_back(w);
    rec_ins.push_back(ins);
    push_op(C_RECOMPOSE, next_npo_id++, 0, 0, w, coeff_row ? 1u : 0u, ins);
  }
  // a sponge row may absorb an owned coefficient instead of an ordinary witness
  auto sponge_owned = [&](uint32_t& w) {
    if (rec_owned.empty() || rng.unit() >= 0.25) return false;
    w = rec_owned[rng.below((uint32_t)rec_owned.size())];
    return true;
  };

  // ---- Poseidon2 rows (executor semantics: ops/poseidon_perm/executor.rs:921-972) ----
  auto& p2_inputs = W.arr["p2_inputs"];      // n x 16
  auto& p2_flags = W.arr["p2_flags"];        // n x 4: new_start, merkle_path, mmcs_bit, mmcs_ctl_enabled
  auto& p2_index_sum = W.arr["p2_mmcs_index_sum"];
  auto& p2_in_ctl = W.arr["p2_in_ctl"];      // n x 4
  auto& p2_in_idx = W.arr["p2_input_indices"];
  auto& p2_out_ctl = W.arr["p2_out_ctl"];    // n x 2, canonical multiplicity
  auto& p2_out_idx = W.arr["p2_output_indices"];
  auto& p2_acc_idx = W.arr["p2_mmcs_index_sum_idx"];
  auto& pd_ids = W.arr["pd_op_ids"]; auto& pd_sib = W.arr["pd_siblings"];
  struct OutFix { size_t row; int limb; uint32_t wid; };
  std::vector<OutFix> out_fix;
  if constexpr (D == 4) {
    F state[16];
    for (auto& x : state) x = F::zero();
    const uint32_t sponge_pick_limit = (uint32_t)pickable.size();
    auto sponge_pick = [&](bool count_read) {
      uint32_t w;
      if (!(count_read && sponge_owned(w)))
        w = (flags & SYN_INDEPENDENT_SPONGES) ? pickable[rng.below(sponge_pick_limit)] : pickp_noread();
      if (count_read) reads[w]++;
      return w;
    };
    // The outputs of some one-row sponges land on witnesses a Public op already defined with the
    // same value: such an output is a READER on the bus (out_ctl = -1; circuit.rs:464-491,
    // dup_npo_outputs; batch_stark_prover.rs:225-238)
    for (size_t r = 0; r < n_p2; ++r) {
      const auto& p = plan[r];
      const bool last_of_chain = (r + 1 == n_p2) || plan[r + 1].new_start;
      F in[16];
      uint32_t in_ctl[4] = {0, 0, 0, 0}, in_idx[4] = {0, 0, 0, 0};
      std::vector<uint32_t> ext(7, NO_W);
      const bool onto_public = !p.merkle && last_of_chain && rng.unit() < 0.2;
      if (p.new_start) {
        for (auto& x : in) x = F::zero();
        for (int l = 0; l < 2; ++l) {
          // sponge: rate limbs read from the witness bus; Merkle: the leaf digest is named by
          // witness index without a bus read (executor.rs:786-790)
          uint32_t w = p.merkle ? pickp_noread() : sponge_pick(true);
          in_ctl[l] = 1; in_idx[l] = w; ext[l] = w;
          for (int d = 0; d < 4; ++d) in[l * 4 + d] = wval[w].c[d];
        }
      } else if (!p.merkle) {
        for (int i = 0; i < 16; ++i) in[i] = state[i];  // full previous output carried
        for (int l = 0; l < 2; ++l) {
          if (rng.unit() < 0.5) {
            uint32_t w = sponge_pick(true);
            in_ctl[l] = 1; in_idx[l] = w; ext[l] = w;
            for (int d = 0; d < 4; ++d) in[l * 4 + d] = wval[w].c[d];
          }
        }
      } else {
        for (int i = 0; i < 8; ++i) in[i] = state[i];  // previous digest carried in the rate limbs
        for (int i = 8; i < 16; ++i) in[i] = F::zero();
      }
      if (p.merkle) {
        // sibling = private data in the capacity limbs, then the direction bit swaps the halves
        // (executor.rs:166-234); the bit itself is the zero / one constant witness
        pd_ids.push_back(next_npo_id);
        for (int i = 8; i < 16; ++i) { in[i] = rf(); pd_sib.push_back(in[i].to_canonical()); }
        if (p.bit) for (int i = 0; i < 8; ++i) std::swap(in[i], in[8 + i]);
        ext[5] = p.bit ? 1u : 0u;
      }
      if (acc_wid[r] >= 0) ext[4] = (uint32_t)acc_wid[r];
      for (int i = 0; i < 16; ++i) p2_inputs.push_back(in[i].to_canonical());
      for (int i = 0; i < 16; ++i) state[i] = in[i];
      p2_permute<PP>(state, rc_canonical);
      const bool en = acc_wid[r] >= 0;
      p2_flags.push_back(p.new_start); p2_flags.push_back(p.merkle); p2_flags.push_back(p.bit);
      p2_flags.push_back(en);
      p2_index_sum.push_back(en ? p.acc : 0u);
      for (int l = 0; l < 4; ++l) { p2_in_ctl.push_back(in_ctl[l]); p2_in_idx.push_back(in_idx[l]); }
      ext[6] = 2;
      for (int l = 0; l < 2; ++l) {
        if (onto_public) {
          E v; for (int d = 0; d < 4; ++d) v.c[d] = state[l * 4 + d];
          uint32_t w = create(v);
          push_op(C_PUBLIC, 0, 0, 0, w, (uint32_t)public_w.size(), {});
          public_w.push_back(w);
          public_rows.push_back(w);
          put_e(in_public, v);
          reads[w]++;
          pickable.push_back(w);
          p2_out_idx.push_back(w);
          p2_out_ctl.push_back(P - 1);
          ext.push_back(w);
        } else if (last_of_chain) {
          E v; for (int d = 0; d < 4; ++d) v.c[d] = state[l * 4 + d];
          uint32_t w = create(v);
          pickable.push_back(w);
          out_fix.push_back({r, l, w});
          p2_out_idx.push_back(w);
          p2_out_ctl.push_back(0);  // patched below once read counts are known
          ext.push_back(w);
        } else {
          p2_out_idx.push_back(0);
          p2_out_ctl.push_back(0);
          ext.push_back(NO_W);
        }
      }
      p2_acc_idx.push_back(en ? (uint32_t)acc_wid[r] : 0u);
      push_op(C_P2, next_npo_id++, 0, 0, 0, (p.new_start ? 1u : 0u) | (p.merkle ? 2u : 0u), ext);
    }
  } else {
    // ---- compact-D1 Poseidon2 rows of a D = 5 circuit (KOALA_BEAR_D1_W16 on the 5-slot witness bus): every state
    // element is its own witness, holding a base-field value (tuples (idx, v, 0, 0, 0, 0)).  Row semantics follow the
    // AIR (poseidon2-circuit-air/src/air.rs:937-1031) and the base-mode executor
    // (circuit/src/ops/poseidon_perm/executor.rs:600-700): sponge rows take CTL inputs on the rate only, the capacity
    // chains (first capacity element += the length tag); Merkle rows carry digest | sibling, swapped by the direction
    // bit, and never read the bus.
    auto& p2_absorb = W.arr["p2_absorb_len"];
    std::vector<uint32_t> base_w;   // base-valued witnesses a sponge may read
    for (uint32_t w : base_valued) base_w.push_back(w);
    for (size_t i = 0; i < std::max<size_t>(H / 8, 4); ++i) {
      // base-valued public inputs
      uint32_t w = create(E::from_base(rf()));
      push_op(C_PUBLIC, 0, 0, 0, w, (uint32_t)public_w.size(), {});
      public_w.push_back(w); public_rows.push_back(w); put_e(in_public, wval[w]);
      pickable.push_back(w); base_w.push_back(w); base_valued.push_back(w);
    }
    F state[16];
    for (auto& x : state) x = F::zero();
    for (size_t r = 0; r < n_p2; ++r) {
      const auto& p = plan[r];
      const bool last_of_chain = (r + 1 == n_p2) || plan[r + 1].new_start;
      F in[16];
      uint32_t in_ctl[16] = {0}, in_idx[16] = {0};
      uint32_t tag = 0;
      if (!p.merkle) {
        tag = rng.below(4) == 0 ? 0u : 1u + rng.below(8);
        for (int i = 0; i < 16; ++i) in[i] = p.new_start ? F::zero() : state[i];
        for (int i = 0; i < 8; ++i)
          if (rng.unit() < (p.new_start ? 0.75 : 0.5)) {
            uint32_t w;
            if (!sponge_owned(w)) w = base_w[rng.below((uint32_t)base_w.size())];
            reads[w]++;
            in_ctl[i] = 1; in_idx[i] = w; in[i] = wval[w].c[0];
          }
        in[8] = in[8] + F::from_canonical(tag);
      } else {
        F digest[8];
        for (int i = 0; i < 8; ++i) {
          if (p.new_start) {
            // the leaf digest is named by witness index, without a bus read (Merkle rows send nothing)
            const uint32_t w = base_w[rng.below((uint32_t)base_w.size())];
            in_ctl[i] = 1; in_idx[i] = w; digest[i] = wval[w].c[0];
          } else {
            digest[i] = state[i];
          }
        }
        for (int i = 0; i < 8; ++i) {
          in[p.bit ? 8 + i : i] = digest[i];
          in[p.bit ? i : 8 + i] = rf();   // sibling: private data
        }
        // (the preprocessed row names the op's input SLOTS, 0..7 for the leaf digest, whatever the direction bit:
        // executor.rs preprocess_inputs; the swap happens on the values)
        pd_ids.push_back(next_npo_id);
        for (int i = 0; i < 8; ++i) pd_sib.push_back(in[p.bit ? i : 8 + i].to_canonical());
      }
      for (int i = 0; i < 16; ++i) { p2_inputs.push_back(in[i].to_canonical()); state[i] = in[i]; }
      p2_permute<PP>(state, rc_canonical);
      const bool en = acc_wid[r] >= 0;
      p2_flags.push_back(p.new_start); p2_flags.push_back(p.merkle); p2_flags.push_back(p.bit); p2_flags.push_back(en);
      p2_index_sum.push_back(en ? p.acc : 0u);
      p2_absorb.push_back(tag);
      for (int i = 0; i < 16; ++i) { p2_in_ctl.push_back(in_ctl[i]); p2_in_idx.push_back(in_idx[i]); }
      const bool onto_public = !p.merkle && last_of_chain && rng.unit() < 0.2;
      uint32_t row_out[8];
      for (int l = 0; l < 8; ++l) {
        row_out[l] = NO_W;
        if (last_of_chain && rng.unit() < 0.6) {
          const uint32_t w = create(E::from_base(state[l]));
          row_out[l] = w;
          if (onto_public) {
            // a Public op already defined this witness with the same value: the output is a reader on the bus
            push_op(C_PUBLIC, 0, 0, 0, w, (uint32_t)public_w.size(), {});
            public_w.push_back(w); public_rows.push_back(w); put_e(in_public, wval[w]);
            reads[w]++;
            p2_out_ctl.push_back(P - 1);
          } else {
            out_fix.push_back({r, l, w});
            p2_out_ctl.push_back(0);  // patched below once read counts are known
          }
          pickable.push_back(w); base_valued.push_back(w);
          p2_out_idx.push_back(w);
        } else {
          p2_out_idx.push_back(0);
          p2_out_ctl.push_back(0);
        }
      }
      p2_acc_idx.push_back(en ? (uint32_t)acc_wid[r] : 0u);
      // the op: ext = [in0..in15, mmcs_index_sum, mmcs_bit, n_out = 8, out0..out7], b = absorb_len (include/p3r.h)
      std::vector<uint32_t> ext(19 + 8, NO_W);
      for (int i = 0; i < 16; ++i) if (in_ctl[i]) ext[i] = in_idx[i];
      if (en) ext[16] = (uint32_t)acc_wid[r];
      if (p.merkle) ext[17] = p.bit ? 1u : 0u;   // the zero / one constant witnesses
      ext[18] = 8;
      for (int l = 0; l < 8; ++l) ext[19 + l] = row_out[l];
      push_op(C_P2, next_npo_id++, tag, 0, 0, (p.new_start ? 1u : 0u) | (p.merkle ? 2u : 0u), ext);
    }
  }

  // ---- the width-32 Poseidon2 table (SYN_P2_W32): H/4 rows of arity-4 Merkle chains and rate-24 sponge chains ----
  // Row semantics (poseidon2-circuit-air/src/air.rs:1178-1342, recursion/src/pcs/mmcs.rs:1013-1075): the state is 8
  // limbs of 4 elements, a chunk is CAPACITY_EXT = 2 limbs.  A Merkle continuation row carries the previous digest
  // (output limbs 0, 1) in chunk pos = bit + 2 * bit2; the other chunks are free siblings, or - on an injection
  // level (pos = 0) - chunk 1 an injected digest and chunks 2, 3 zero pads, all three CTL-loaded (in_ctl = 1); the
  // two direction bits are read from the zero / one constant witnesses on every Merkle row.  A sponge row overwrites
  // the six rate limbs from the bus and chains the capacity limbs.  The last row of a chain exposes its six rate
  // output limbs (out_ctl).
  size_t n_p2w = 0;
  if (flags & SYN_P2_W32) {
    if constexpr (D != 4) throw std::runtime_error("SYN_P2_W32: the width-32 table belongs to D = 4 circuits");
    else {
    if (!g_w32_rc || !g_w32_diag) throw std::runtime_error("SYN_P2_W32: call syn_set_w32 first");
    n_p2w = std::max<size_t>(H / 4, 4);
    auto& w_inputs = W.arr["p2w_inputs"];   // n x 32
    auto& w_flags = W.arr["p2w_flags"];     // n x 4: new_start, merkle_path, mmcs_bit, mmcs_bit2
    auto& w_sum = W.arr["p2w_mmcs_index_sum"];
    auto& w_prep = W.arr["p2w_prep"];       // n x 48, assembled Poseidon2PreprocessedRow<8, 6>
    struct WFix { size_t cell; uint32_t wid; };
    std::vector<WFix> w_out_fix;            // out_ctl cells patched with the final read counts
    F st[32];
    for (auto& x : st) x = F::zero();
    size_t r = 0;
    const uint32_t ZERO_W = const_w[0], ONE_W = const_w[1];
    const bool as_ops = flags & SYN_P2_W32_OPS;
    auto& pdw_ids = W.arr["pdw_op_ids"]; auto& pdw_sib = W.arr["pdw_siblings"];
    while (r < n_p2w) {
      const bool merkle_chain = rng.unit() < 0.7;
      // as ops: half of the Merkle chains continue a leaf sponge of 1-3 rows (new_start only on its first row)
      const size_t prefix = (as_ops && merkle_chain && rng.unit() < 0.5) ? 1 + rng.below(3) : 0;
      size_t len = std::min<size_t>((merkle_chain ? std::max(2, merkle_depth / 2) : std::max(1, sponge_chain_len)) + prefix, n_p2w - r);
      uint32_t acc = 0;
      for (size_t j = 0; j < len; ++j, ++r) {
        const bool merkle = merkle_chain && j >= prefix;
        const bool ns = j == 0, last = j + 1 == len;
        F in[32];
        uint32_t in_ctl[8] = {0}, in_idx[8] = {0};
        bool bit = false, bit2 = false;
        auto load = [&](int limb, uint32_t w) {   // CTL-loaded limb: a bus read of witness w
          in_ctl[limb] = 1; in_idx[limb] = w; reads[w]++;
          for (int d = 0; d < 4; ++d) in[limb * 4 + d] = wval[w].c[d];
        };
        if (merkle) {
          const bool inject = !ns && rng.unit() < 0.3, bridge = !ns && !inject && rng.unit() < 0.2;
          uint32_t pos = (inject || bridge) ? (bridge ? rng.below(2) : 0u) : rng.below(4);
          bit = pos & 1; bit2 = pos & 2;
          for (int i = 0; i < 32; ++i) in[i] = rf();          // free siblings (private data)
          if (as_ops) {   // set_private_data: the three chunks other than pos, ascending (fill_sibling_data)
            pdw_ids.push_back(next_npo_id);
            for (uint32_t chunk = 0; chunk < 4; ++chunk)
              if (chunk != pos) for (int i = 0; i < 8; ++i) pdw_sib.push_back(in[8 * chunk + i].to_canonical());
          }
          if (ns) {
            // the leaf digest enters chunk pos from the bus
            load(2 * pos, pickp_noread());
            load(2 * pos + 1, pickp_noread());
          } else {
            for (int i = 0; i < 8; ++i) in[8 * pos + i] = st[i];   // running digest, bound by the placement constraint
          }
          if (inject) {
            load(2, pickp_noread()); load(3, pickp_noread());   // injected digest in chunk 1
            for (int l = 4; l < 8; ++l) load(l, ZERO_W);         // zero pads
          } else if (bridge) {
            for (int l = 4; l < 8; ++l) load(l, ZERO_W);         // a step-2 level: chunks 2, 3 are pads
          }
          reads[bit ? ONE_W : ZERO_W]++;
          reads[bit2 ? ONE_W : ZERO_W]++;
          acc = ns ? pos : (uint32_t)(((uint64_t)acc * 4 + pos) % P);
        } else {
          if (ns) for (int i = 0; i < 32; ++i) in[i] = F::zero();
          else for (int i = 0; i < 32; ++i) in[i] = st[i];      // capacity (and un-overwritten rate) limbs chain
          const int take = ns ? 6 : 1 + (int)rng.below(6);
          for (int l = 0; l < take; ++l) load(l, pickp_noread());
        }
        for (int i = 0; i < 32; ++i) { w_inputs.push_back(in[i].to_canonical()); st[i] = in[i]; }
        p2w_permute<PP>(st, g_w32_rc, g_w32_diag);
        w_flags.push_back(ns); w_flags.push_back(merkle); w_flags.push_back(bit); w_flags.push_back(bit2);
        w_sum.push_back(merkle && !as_ops ? acc : 0u);   // the executor's row carries no accumulator value
        std::vector<uint32_t> ext(12 + 6, NO_W);         // [in0..7, mmcs_index_sum, bit, bit2, n_out, out0..5]
        for (int l = 0; l < 8; ++l) if (in_ctl[l]) ext[l] = in_idx[l];
        if (merkle) { ext[9] = bit ? ONE_W : ZERO_W; ext[10] = bit2 ? ONE_W : ZERO_W; }
        ext[11] = 6;
        // preprocessed row
        for (int l = 0; l < 8; ++l) {
          w_prep.push_back(in_idx[l] * D);
          w_prep.push_back(in_ctl[l]);
          w_prep.push_back((!ns && !merkle && !in_ctl[l]) ? 1u : 0u);
          w_prep.push_back((!ns && merkle && !in_ctl[l]) ? 1u : 0u);
        }
        for (int l = 0; l < 6; ++l) {
          if (last) {
            E v; for (int d = 0; d < 4; ++d) v.c[d] = st[l * 4 + d];
            const uint32_t w = create(v);
            pickable.push_back(w);
            ext[12 + l] = w;
            w_prep.push_back(w * D);
            w_out_fix.push_back({w_prep.size(), w});
            w_prep.push_back(0);
          } else {
            w_prep.push_back(0); w_prep.push_back(0);
          }
        }
        // arity-4: the accumulator slots carry the witnesses of the two direction bits (air.rs:1848-1870)
        w_prep.push_back((merkle ? (bit ? ONE_W : ZERO_W) : 0u) * D);
        w_prep.push_back((merkle ? (bit2 ? ONE_W : ZERO_W) : 0u) * D);
        w_prep.push_back(ns);
        w_prep.push_back(merkle);
        if (as_ops) push_op(C_P2W, next_npo_id++, 0, 0, 0, (ns ? 1u : 0u) | (merkle ? 2u : 0u), ext);
      }
    }
    W.arr["p2w_out_fix"].clear();
    for (auto& f : w_out_fix) { W.arr["p2w_out_fix"].push_back((uint32_t)f.cell); W.arr["p2w_out_fix"].push_back(f.wid); }
    }
  }

  // ---- ALU ops: 3 lanes x ~H rows; Horner chains ride lane 0 (alu_air.rs:349-463) ----
  // Op mix: 45% Add, 30% Mul, 10% MulAdd, 14% HornerAcc (chains of
  // length U{4..horner_chain_len}), 1% BoolCheck; a few percent of the Add / Mul ops run
  // backwards (the runner solves for b, runner.rs:341-385), claim a private input or consume
  // hint outputs.  Only the LAST output of a Horner run is ever read by another op:
  // intermediate outputs of packed rows never reach the bus (alu_air.rs:630-667).
  auto& alu_values = W.arr["alu_values"];  // n x 4D (a,b,c,out)
  // a_state / c_state: 0 skip, 1 reader, 2 creator (circuit.rs:341-379)
  struct AluOp { int kind; uint32_t a, b, c, out; uint8_t a_state, c_state; bool b_creator, out_creator, has_c; };
  std::vector<AluOp> ops;
  std::vector<uint32_t> bool_w = {0, 1};
  {
    const size_t lanes = 3;
    auto emit = [&](int kind, uint32_t a, uint32_t b, uint32_t c, uint32_t out, uint32_t aux, uint8_t a_state,
                    uint8_t c_state, bool b_creator, bool out_creator) {
      const bool has_c = c != NO_W;
      if (!b_creator) reads[b]++;
      if (!out_creator) reads[out]++;
      if (a_state == 1) reads[a]++;
      if (c_state == 1) reads[c]++;
      ops.push_back({kind, a, b, has_c ? c : 0u, out, a_state, c_state, b_creator, out_creator, has_c});
      // AluOpRecord values (runner.rs:317-453)
      E av = wval[a], bv = wval[b], cv = E::zero(), ov = wval[out];
      if (kind == OP_BOOL) { bv = E::zero(); cv = av; }
      else if (kind == OP_MULADD || kind == OP_HORNER) cv = has_c ? wval[c] : E::zero();
      put_e(alu_values, av); put_e(alu_values, bv); put_e(alu_values, cv); put_e(alu_values, ov);
      static const uint32_t kinds[5] = {C_ADD, C_MUL, C_BOOL, C_MULADD, C_HORNER};
      push_op(kinds[kind], a, b, c, out, aux, {});
    };
    // forward op with already-defined operands
    auto fwd = [&](int kind, uint32_t a, uint32_t b, uint32_t c, const E& outv, bool out_pickable, uint32_t aux = NO_W) {
      uint32_t o = create(outv);
      emit(kind, a, b, c, o, aux, 1, c != NO_W ? 1 : 0, false, true);
      if (out_pickable) pickable.push_back(o);
      return o;
    };
    const double avg_len = horner_chain_len > 4 ? (4 + horner_chain_len) / 2.0 : horner_chain_len;
    const double p_chain = horner_chain_len > 0 ? 0.14 / (avg_len * 0.86 + 0.14) : 0.0;
    size_t chain_rows = 1, nonchain = 0;  // leading separator row
    auto rows_now = [&]() {
      size_t fill = 2 * chain_rows;
      return chain_rows + (nonchain > fill ? (nonchain - fill + lanes - 1) / lanes : 0);
    };
    const size_t target = (flags & SYN_NO_ALU) ? 0 : (H > 16 ? H - 4 : H - 1);
    size_t next_private = 0;
    while (rows_now() < target) {
      double u = rng.unit();
      if (u < p_chain) {
        int len = horner_chain_len > 4 ? 4 + (int)rng.below((uint32_t)horner_chain_len - 3) : horner_chain_len;
        size_t need = (size_t)(len + 3) / 4 + 1;
        if (rows_now() + need + 1 >= target) { len = 2; need = 2; }
        uint32_t b = pickp_noread();
        E acc = E::zero();
        uint32_t acc_w = 0;  // the zero constant seeds the accumulator
        for (int j = 0; j < len; ++j) {
          uint32_t a = pickp_noread(), c = pickp_noread();
          acc = acc * wval[b] + wval[c] - wval[a];
          acc_w = fwd(OP_HORNER, a, b, c, acc, j == len - 1, acc_w);
        }
        chain_rows += need;
        // a non-Horner op ends the run so the next chain is a separate chain
        uint32_t a = pickp_noread(), bb = pickp_noread();
        fwd(OP_ADD, a, bb, NO_W, wval[a] + wval[bb], true);
        nonchain += 1;
        continue;
      }
      double v = rng.unit();
      if (v < 0.03 && next_private < private_w.size()) {
        // a private input is claimed by its first ALU use: as `a` (a_state = creator), as `b` of
        // a forward op (b_is_private_creator) or as `c` of a MulAdd (c_state = creator)
        uint32_t pw = private_w[next_private++];
        int role = (int)rng.below(3);
        if (role == 0) {
          uint32_t b = pickp_noread();
          uint32_t o = create(wval[pw] + wval[b]);
          emit(OP_ADD, pw, b, NO_W, o, NO_W, 2, 0, false, true);
          pickable.push_back(o);
        } else if (role == 1) {
          uint32_t a = pickp_noread();
          uint32_t o = create(wval[a] * wval[pw]);
          emit(OP_MUL, a, pw, NO_W, o, NO_W, 1, 0, true, true);
          pickable.push_back(o);
        } else {
          uint32_t a = pickp_noread(), b = pickp_noread();
          uint32_t o = create(wval[a] * wval[b] + wval[pw]);
          emit(OP_MULADD, a, b, pw, o, NO_W, 1, 2, false, true);
          pickable.push_back(o);
        }
        pickable.push_back(pw);
      } else if (v < 0.06) {
        // backward op (the lowering of sub / div): a and out are known, the runner solves for b
        uint32_t a = pickp_noread(), o = pickp_noread();
        const bool mul = rng.unit() < 0.5 && !(wval[a] == E::zero());
        uint32_t b = create(mul ? wval[o] * wval[a].inv() : wval[o] - wval[a]);
        emit(mul ? OP_MUL : OP_ADD, a, b, NO_W, o, NO_W, 1, 0, true, false);
        pickable.push_back(b);
      } else if (v < 0.07) {
        // ExtDecompositionHint: D hint outputs, each claimed by the Add that first uses it
        uint32_t src = pickp_noread();
        std::vector<uint32_t> outs(D);
        for (int i = 0; i < D; ++i) outs[i] = create(E::from_base(wval[src].c[i]));
        push_op(C_HINT_EXT, src, 0, 0, 0, 0, outs);
        for (int i = 0; i < D; ++i) {
          uint32_t b = pickp_noread();
          uint32_t o = create(wval[outs[i]] + wval[b]);
          emit(OP_ADD, outs[i], b, NO_W, o, NO_W, 2, 0, false, true);
          pickable.push_back(o);
          pickable.push_back(outs[i]);
          base_valued.push_back(outs[i]);
          nonchain += 1;
        }
        continue;
      } else if (v < 0.075) {
        // BinaryDecompositionHint of the low coefficient: 8 bits, each bool-checked the way the
        // lowerer emits it (a = c = bit, b = zero constant; lowerer/state.rs:336-351)
        uint32_t src = pickp_noread();
        std::vector<uint32_t> outs(8);
        const uint32_t val = wval[src].c[0].to_canonical();
        for (int i = 0; i < 8; ++i) outs[i] = create(E::from_base(F::from_canonical((val >> i) & 1)));
        push_op(C_HINT_BIN, src, 0, 0, 0, 0, outs);
        for (int i = 0; i < 8; ++i) {
          // an Add claims the hint output first (a_state = creator); the lowerer-style BoolCheck
          // then reads it twice (a and c)
          uint32_t z = pickp_noread();
          uint32_t o1 = create(wval[outs[i]] + wval[z]);
          emit(OP_ADD, outs[i], z, NO_W, o1, NO_W, 2, 0, false, true);
          pickable.push_back(o1);
          uint32_t o = create(wval[outs[i]]);
          emit(OP_BOOL, outs[i], 0, outs[i], o, NO_W, 1, 1, false, true);
          pickable.push_back(o);
          bool_w.push_back(o);
          nonchain += 2;
        }
        continue;
      } else if (v < 0.45 / 0.86) {
        uint32_t a = pickp_noread(), b = pickp_noread();
        fwd(OP_ADD, a, b, NO_W, wval[a] + wval[b], true);
      } else if (v < 0.75 / 0.86) {
        uint32_t a = pickp_noread(), b = pickp_noread();
        fwd(OP_MUL, a, b, NO_W, wval[a] * wval[b], true);
      } else if (v < 0.85 / 0.86) {
        uint32_t a = pickp_noread(), b = pickp_noread(), c = pickp_noread();
        // one MulAdd in four keeps the witness of the fused a*b product (intermediate_out)
        uint32_t io = rng.unit() < 0.25 ? create(wval[a] * wval[b]) : NO_W;
        fwd(OP_MULADD, a, b, c, wval[a] * wval[b] + wval[c], true, io);
      } else {
        uint32_t a = bool_w[rng.below((uint32_t)bool_w.size())];
        uint32_t o = fwd(OP_BOOL, a, 0, a, wval[a], true);
        bool_w.push_back(o);
      }
      nonchain += 1;
    }
    // private inputs nobody claimed yet (UnclaimedPrivateInput otherwise, circuit.rs:497-503)
    while (next_private < private_w.size()) {
      uint32_t pw = private_w[next_private++], b = pickp_noread();
      uint32_t o = create(wval[pw] + wval[b]);
      emit(OP_ADD, pw, b, NO_W, o, NO_W, 2, 0, false, true);
      pickable.push_back(o);
    }
  }
  // ALU-dedup leftovers: duplicates the runner fills from their canonical witness at the end
  auto& rewrite = W.arr["rewrite"];
  for (size_t i = 0; i < std::max<size_t>(H / 256, 1) && !(flags & SYN_NO_ALU); ++i) {
    uint32_t canon = pickp_noread();
    uint32_t dup = create(wval[canon]);
    rewrite.push_back(dup); rewrite.push_back(canon);
  }

  // ---- emit tables ----
  auto& const_values = W.arr["const_values"]; auto& const_prep = W.arr["const_prep"];
  for (uint32_t w : const_w) { put_e(const_values, wval[w]); const_prep.push_back(reads[w]); const_prep.push_back(w * D); }
  auto& public_values = W.arr["public_values"]; auto& public_prep = W.arr["public_prep"];
  for (uint32_t w : public_w) { put_e(public_values, wval[w]); public_prep.push_back(reads[w]); public_prep.push_back(w * D); }
  auto& rec_prep = W.arr["recompose_prep"];
  auto& rec2_prep = W.arr["recompose_coeff_prep"];
  size_t n_rec_second = 0;
  {
    std::vector<char> owned(wval.size(), 0);
    for (uint32_t w : rec_owned) owned[w] = 1;
    for (size_t i = 0; i < rec_w.size(); ++i) {
      auto& dst = rec_kind[i] ? rec2_prep : rec_prep;
      n_rec_second += rec_kind[i];
      dst.push_back(rec_w[i] * D); dst.push_back(rec_dup[i] ? P - 1 : reads[rec_w[i]]);
      if (rec_both ? rec_kind[i] != 0 : rec_coeff)
        for (int k = 0; k < D; ++k) {
          const uint32_t c = rec_ins[i][k];
          dst.push_back(c * D);
          dst.push_back(owned[c] ? reads[c] : 0u);
        }
    }
  }
  for (auto& f : out_fix) p2_out_ctl[f.row * (D == 4 ? 2 : 8) + f.limb] = reads[f.wid];
  if (n_p2w) {
    auto& fix = W.arr["p2w_out_fix"];
    auto& w_prep = W.arr["p2w_prep"];
    for (size_t i = 0; i + 1 < fix.size(); i += 2) w_prep[fix[i]] = reads[fix[i + 1]] % P;
    W.arr.erase("p2w_out_fix");
  }
  // ALU per-op preprocessed, 13 columns (AluPrepLaneCols, alu_columns.rs:9-46; common.rs:198-281)
  auto& alu_prep = W.arr["alu_prep13"];
  const uint32_t neg1 = P - 1;
  auto neg = [&](uint32_t x) { return (P - x % P) % P; };
  for (auto& o : ops) {
    const uint32_t a_col = o.a_state == 1 ? 1u : o.a_state == 2 ? neg(reads[o.a]) : 0u;
    const uint32_t c_col = o.c_state == 1 ? 1u : o.c_state == 2 ? neg(reads[o.c]) : 0u;
    uint32_t row[13] = {neg1, o.kind == OP_ADD, o.kind == OP_BOOL, o.kind == OP_MULADD, o.kind == OP_HORNER,
                        o.a * D, o.b * D, o.c * D, o.out * D,
                        o.b_creator ? reads[o.b] % P : neg1, o.out_creator ? reads[o.out] % P : neg1, a_col, c_col};
    alu_prep.insert(alu_prep.end(), row, row + 13);
  }
  if (ops.empty()) {
    // AluTrace::from_records / get_airs_and_degrees_with_prep add one all-zero dummy op
    // (tables/alu.rs:69-73, common.rs:283-286)
    alu_prep.insert(alu_prep.end(), 13, 0u);
    alu_values.insert(alu_values.end(), 4 * D, 0u);
  }
  W.arr["counts"] = {(uint32_t)const_w.size(), (uint32_t)public_w.size(), (uint32_t)std::max<size_t>(ops.size(), 1),
                     (uint32_t)n_p2, (uint32_t)(rec_w.size() - n_rec_second), (uint32_t)wval.size(), (uint32_t)n_rec_second,
                     (uint32_t)n_p2w};
}

}  // namespace

extern "C" {

// flags bits 8..11: circuit extension degree (0 = 4).  5 = the KoalaBear quintic trinomial extension, 1 = base-field
// circuits (Poseidon2 rows are the compact-D1 ones in both).
void* syn_generate(int field, int log_h, uint64_t seed, int horner_chain_len, int sponge_chain_len,
                   int merkle_depth, const uint32_t* rc_canonical, uint32_t flags) {
  auto* W = new Workload();
  const uint32_t ext_degree = (flags >> 8) & 15u ? (flags >> 8) & 15u : 4u;
  flags = (flags & 0xFFu) | (flags & SYN_P2_W32_OPS);
  if (flags & SYN_P2_W32_OPS) flags |= SYN_P2_W32;
  try {
    if (ext_degree == 5 && field == 0) generate<KoalaBearParams, Fp5<KoalaBearParams>>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else if (ext_degree == 1 && field == 0) generate<KoalaBearParams, Fp1<KoalaBearParams>>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else if (ext_degree == 1 && field == 1) generate<BabyBearParams, Fp1<BabyBearParams>>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    // binomial extensions of degree 2 / 6 / 8 with W = 3 (KoalaBear) / 11 (BabyBear): the W is data (tests pass the same
    // value to the prover and the oracle); primitive tables and Recompose
    else if (ext_degree == 8 && field == 0) generate<KoalaBearParams, FpBin<KoalaBearParams, 8, 3>>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else if (ext_degree == 2 && field == 0) generate<KoalaBearParams, FpBin<KoalaBearParams, 2, 3>>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else if (ext_degree == 6 && field == 1) generate<BabyBearParams, FpBin<BabyBearParams, 6, 11>>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else if (ext_degree == 8 && field == 1) generate<BabyBearParams, FpBin<BabyBearParams, 8, 11>>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else if (ext_degree != 4) throw std::runtime_error("ext_degree must be 1, 4, 5 over KoalaBear, or 2 / 6 / 8 (koala-bear: 2, 8; baby-bear: 6, 8)");
    else if (field == 0) generate<KoalaBearParams, Fp4<KoalaBearParams>>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else if (field == 1) generate<BabyBearParams, Fp4<BabyBearParams>>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else throw std::runtime_error("unknown field");
  } catch (const std::exception& e) {
    W->err = e.what();
  }
  return W;
}
// constants of the width-32 permutation for SYN_P2_W32 (the pointers must stay valid during syn_generate)
void syn_set_w32(const uint32_t* rc_canonical, const uint32_t* diag_canonical) { g_w32_rc = rc_canonical; g_w32_diag = diag_canonical; }
const char* syn_error(void* h) { return static_cast<Workload*>(h)->err.c_str(); }
int syn_get(void* h, const char* name, const uint32_t** ptr, size_t* len) {
  auto* W = static_cast<Workload*>(h);
  auto it = W->arr.find(name);
  if (it == W->arr.end()) return -1;
  *ptr = it->second.data();
  *len = it->second.size();
  return 0;
}
void syn_free(void* h) { delete static_cast<Workload*>(h); }

}  // extern "C"
