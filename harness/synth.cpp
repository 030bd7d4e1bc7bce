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

#include "../plonky3_recursion_amd/csrc/field.h"
#include "../plonky3_recursion_amd/csrc/poseidon2.h"

using namespace p3r;

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
enum { SYN_NO_POSEIDON2 = 1, SYN_NO_RECOMPOSE = 2, SYN_SINGLE_PUBLIC = 4, SYN_NO_ALU = 8 };

enum { OP_ADD = 0, OP_MUL = 1, OP_BOOL = 2, OP_MULADD = 3, OP_HORNER = 4 };

template <class PP>
void generate(Workload& W, int log_h, uint64_t seed, int horner_chain_len, int sponge_chain_len,
              int merkle_depth, const uint32_t* rc_canonical, uint32_t flags) {
  using F = Fp<PP>;
  using E = Fp4<PP>;
  const uint32_t P = PP::P;
  const size_t H = size_t(1) << log_h;
  Rng rng(seed);
  auto rf = [&]() { return F::from_canonical((uint32_t)(rng.next() % P)); };
  auto re = [&]() { E e; for (int i = 0; i < 4; ++i) e.c[i] = rf(); return e; };

  std::vector<uint32_t> rc_m(p2_num_constants<PP>());
  for (size_t i = 0; i < rc_m.size(); ++i) rc_m[i] = F::from_canonical(rc_canonical[i]).v;

  // witness table
  std::vector<E> wval;
  std::vector<uint32_t> reads;
  auto create = [&](const E& v) { wval.push_back(v); reads.push_back(0); return (uint32_t)(wval.size() - 1); };
  auto pick = [&]() { uint32_t w = rng.below((uint32_t)wval.size()); reads[w]++; return w; };
  auto put_e = [&](std::vector<uint32_t>& dst, const E& e) { for (int i = 0; i < 4; ++i) dst.push_back(e.c[i].to_canonical()); };

  // ---- Const (H/16 rows) and Public (H/2 ops) ----
  if ((flags & SYN_SINGLE_PUBLIC) && !(flags & SYN_NO_POSEIDON2))
    throw std::runtime_error("SYN_SINGLE_PUBLIC needs SYN_NO_POSEIDON2 (Merkle accumulators are public inputs)");
  const size_t n_const = std::max<size_t>(H / 16, 2);
  const size_t n_public = (flags & SYN_SINGLE_PUBLIC) ? 1 : std::max<size_t>(H / 2, 2);
  std::vector<uint32_t> const_w, public_w;
  {
    // witness 0 is the zero constant, witness 1 is one (handy for bool ops)
    const_w.push_back(create(E::zero()));
    const_w.push_back(create(E::one()));
    while (const_w.size() < n_const) const_w.push_back(create(re()));
  }

  // ---- Poseidon2 chain plan first (its accumulator values need Public witnesses) ----
  const size_t n_p2 = (flags & SYN_NO_POSEIDON2) ? 0 : std::max<size_t>(H / 2, 4);
  struct P2Plan { bool new_start, merkle, bit, mmcs_ctl; uint32_t acc; };
  std::vector<P2Plan> plan;
  {
    // leave room for the trailing partial chain; ~70% Merkle rows, ~30% sponge rows
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
  std::vector<uint32_t> public_vals_w;
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

  // ---- Recompose (H/4 rows): creates an extension witness from 4 base coefficients ----
  const size_t n_rec = (flags & SYN_NO_RECOMPOSE) ? 0 : std::max<size_t>(H / 4, 2);
  std::vector<uint32_t> rec_w;
  for (size_t i = 0; i < n_rec; ++i) rec_w.push_back(create(re()));

  // ---- Poseidon2 rows ----
  auto& p2_inputs = W.arr["p2_inputs"];      // n x 16
  auto& p2_flags = W.arr["p2_flags"];        // n x 4: new_start, merkle_path, mmcs_bit, mmcs_ctl_enabled
  auto& p2_index_sum = W.arr["p2_mmcs_index_sum"];
  auto& p2_in_ctl = W.arr["p2_in_ctl"];      // n x 4
  auto& p2_in_idx = W.arr["p2_input_indices"];
  auto& p2_out_ctl = W.arr["p2_out_ctl"];    // n x 2, canonical multiplicity (n_reads)
  auto& p2_out_idx = W.arr["p2_output_indices"];
  auto& p2_acc_idx = W.arr["p2_mmcs_index_sum_idx"];
  struct OutFix { size_t row; int limb; uint32_t wid; };
  std::vector<OutFix> out_fix;
  {
    F state[16];
    for (auto& x : state) x = F::zero();
    for (size_t r = 0; r < n_p2; ++r) {
      const auto& p = plan[r];
      F in[16];
      uint32_t in_ctl[4] = {0, 0, 0, 0}, in_idx[4] = {0, 0, 0, 0};
      if (p.new_start) {
        for (auto& x : in) x = F::zero();
        if (p.merkle) {
          for (auto& x : in) x = rf();  // leaf digest + sibling: private data, no CTL on Merkle rows
        } else {
          for (int l = 0; l < 2; ++l) {  // rate limbs fed from the witness bus
            uint32_t w = pick();
            in_ctl[l] = 1; in_idx[l] = w;
            for (int d = 0; d < 4; ++d) in[l * 4 + d] = wval[w].c[d];
          }
        }
      } else if (!p.merkle) {
        for (int i = 0; i < 16; ++i) in[i] = state[i];  // full previous output carried
        for (int l = 0; l < 2; ++l) {
          if (rng.unit() < 0.5) {
            uint32_t w = pick();
            in_ctl[l] = 1; in_idx[l] = w;
            for (int d = 0; d < 4; ++d) in[l * 4 + d] = wval[w].c[d];
          }
        }
      } else {
        // previous digest (limbs 0..1) placed left or right of a fresh sibling
        F sib[8];
        for (auto& x : sib) x = rf();
        for (int i = 0; i < 8; ++i) {
          in[p.bit ? 8 + i : i] = state[i];
          in[p.bit ? i : 8 + i] = sib[i];
        }
      }
      for (int i = 0; i < 16; ++i) p2_inputs.push_back(in[i].to_canonical());
      for (int i = 0; i < 16; ++i) state[i] = in[i];
      p2_permute<PP>(state, rc_m.data());
      bool last_of_chain = (r + 1 == n_p2) || plan[r + 1].new_start;
      p2_flags.push_back(p.new_start); p2_flags.push_back(p.merkle); p2_flags.push_back(p.bit);
      p2_flags.push_back(p.mmcs_ctl);
      p2_index_sum.push_back(p.merkle ? p.acc : 0u);
      for (int l = 0; l < 4; ++l) { p2_in_ctl.push_back(in_ctl[l]); p2_in_idx.push_back(in_idx[l]); }
      for (int l = 0; l < 2; ++l) {
        if (last_of_chain) {
          E v; for (int d = 0; d < 4; ++d) v.c[d] = state[l * 4 + d];
          uint32_t w = create(v);
          out_fix.push_back({r, l, w});
          p2_out_idx.push_back(w);
        } else {
          p2_out_idx.push_back(0);
        }
        p2_out_ctl.push_back(0);  // patched below once read counts are known
      }
      p2_acc_idx.push_back(acc_wid[r] >= 0 ? (uint32_t)acc_wid[r] : 0u);
    }
  }

  // ---- ALU ops: 3 lanes x ~H rows; Horner chains ride lane 0 (alu_air.rs:349-463) ----
  // Op mix: 45% Add, 30% Mul, 10% MulAdd, 14% HornerAcc (chains of
  // length U{4..horner_chain_len}), 1% BoolCheck.  Only the LAST output of a Horner run is
  // ever read by another op: intermediate outputs of packed rows never reach the bus
  // (alu_air.rs:630-667), so they are created non-pickable.
  auto& alu_values = W.arr["alu_values"];  // n x 16 (a,b,c,out)
  struct AluOp { int kind; uint32_t a, b, c, out; bool c_rd; };
  std::vector<AluOp> ops;
  {
    const size_t lanes = 3;
    std::vector<uint32_t> pickable(wval.size());
    for (size_t i = 0; i < pickable.size(); ++i) pickable[i] = (uint32_t)i;
    auto pickp = [&]() { uint32_t w = pickable[rng.below((uint32_t)pickable.size())]; reads[w]++; return w; };
    auto emit = [&](int kind, uint32_t a, uint32_t b, uint32_t c, const E& outv, bool c_rd, bool out_pickable) {
      uint32_t o = create(outv);
      if (out_pickable) pickable.push_back(o);
      ops.push_back({kind, a, b, c, o, c_rd});
      put_e(alu_values, wval[a]); put_e(alu_values, wval[b]);
      put_e(alu_values, c_rd ? wval[c] : E::zero()); put_e(alu_values, outv);
    };
    const double avg_len = horner_chain_len > 4 ? (4 + horner_chain_len) / 2.0 : horner_chain_len;
    const double p_chain = horner_chain_len > 0 ? 0.14 / (avg_len * 0.86 + 0.14) : 0.0;
    size_t chain_rows = 1, nonchain = 0;  // leading separator row
    auto rows_now = [&]() {
      size_t fill = 2 * chain_rows;
      return chain_rows + (nonchain > fill ? (nonchain - fill + lanes - 1) / lanes : 0);
    };
    const size_t target = (flags & SYN_NO_ALU) ? 0 : (H > 16 ? H - 4 : H - 1);
    while (rows_now() < target) {
      double u = rng.unit();
      if (u < p_chain) {
        int len = horner_chain_len > 4 ? 4 + (int)rng.below((uint32_t)horner_chain_len - 3) : horner_chain_len;
        size_t need = (size_t)(len + 3) / 4 + 1;
        if (rows_now() + need + 1 >= target) { len = 2; need = 2; }
        uint32_t b = pickp();
        reads[b] += (uint32_t)len - 1;
        E acc = E::zero();
        for (int j = 0; j < len; ++j) {
          uint32_t a = pickp(), c = pickp();
          acc = acc * wval[b] + wval[c] - wval[a];
          emit(OP_HORNER, a, b, c, acc, true, j == len - 1);
        }
        chain_rows += need;
        // a non-Horner op ends the run so the next chain is a separate chain
        uint32_t a = pickp(), bb = pickp();
        emit(OP_ADD, a, bb, 0, wval[a] + wval[bb], false, true);
        nonchain += 1;
        continue;
      }
      double v = rng.unit();
      if (v < 0.45 / 0.86) {
        uint32_t a = pickp(), b = pickp();
        emit(OP_ADD, a, b, 0, wval[a] + wval[b], false, true);
      } else if (v < 0.75 / 0.86) {
        uint32_t a = pickp(), b = pickp();
        emit(OP_MUL, a, b, 0, wval[a] * wval[b], false, true);
      } else if (v < 0.85 / 0.86) {
        uint32_t a = pickp(), b = pickp(), c = pickp();
        emit(OP_MULADD, a, b, c, wval[a] * wval[b] + wval[c], true, true);
      } else {
        uint32_t a = rng.below(2);  // witness 0 (zero) or 1 (one)
        reads[a]++;
        uint32_t b = pickp();
        emit(OP_BOOL, a, b, 0, E::zero(), false, true);
      }
      nonchain += 1;
    }
  }

  // ---- emit tables ----
  auto& const_values = W.arr["const_values"]; auto& const_prep = W.arr["const_prep"];
  for (uint32_t w : const_w) { put_e(const_values, wval[w]); const_prep.push_back(reads[w]); const_prep.push_back(w * 4); }
  auto& public_values = W.arr["public_values"]; auto& public_prep = W.arr["public_prep"];
  for (uint32_t w : public_w) { put_e(public_values, wval[w]); public_prep.push_back(reads[w]); public_prep.push_back(w * 4); }
  auto& rec_values = W.arr["recompose_values"]; auto& rec_prep = W.arr["recompose_prep"];
  for (uint32_t w : rec_w) { put_e(rec_values, wval[w]); rec_prep.push_back(w * 4); rec_prep.push_back(reads[w]); }
  for (auto& f : out_fix) p2_out_ctl[f.row * 2 + f.limb] = reads[f.wid];
  // ALU per-op preprocessed, 13 columns (AluPrepLaneCols, alu_columns.rs:9-46)
  auto& alu_prep = W.arr["alu_prep13"];
  const uint32_t neg1 = P - 1;
  for (auto& o : ops) {
    uint32_t row[13] = {neg1, o.kind == OP_ADD, o.kind == OP_BOOL, o.kind == OP_MULADD, o.kind == OP_HORNER,
                        o.a * 4, o.b * 4, (o.c_rd ? o.c : 0) * 4, o.out * 4, neg1, reads[o.out] % P,
                        1u, o.c_rd ? 1u : 0u};
    alu_prep.insert(alu_prep.end(), row, row + 13);
  }
  if (ops.empty()) {
    // AluTrace::from_records / get_airs_and_degrees_with_prep add one all-zero dummy op
    // (tables/alu.rs:69-73, common.rs:283-286)
    alu_prep.insert(alu_prep.end(), 13, 0u);
    alu_values.insert(alu_values.end(), 16, 0u);
    ops.push_back({OP_ADD, 0, 0, 0, 0, false});
  }
  W.arr["counts"] = {(uint32_t)const_w.size(), (uint32_t)public_w.size(), (uint32_t)ops.size(),
                     (uint32_t)n_p2, (uint32_t)rec_w.size(), (uint32_t)wval.size()};
}

}  // namespace

extern "C" {

void* syn_generate(int field, int log_h, uint64_t seed, int horner_chain_len, int sponge_chain_len,
                   int merkle_depth, const uint32_t* rc_canonical, uint32_t flags) {
  auto* W = new Workload();
  try {
    if (field == 0) generate<KoalaBearParams>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else if (field == 1) generate<BabyBearParams>(*W, log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth, rc_canonical, flags);
    else throw std::runtime_error("unknown field");
  } catch (const std::exception& e) {
    W->err = e.what();
  }
  return W;
}
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
