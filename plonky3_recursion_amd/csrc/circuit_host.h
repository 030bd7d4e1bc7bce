// Host side of the circuit boundary: validation of a flattened Circuit<EF>, the preprocessed columns
// (Circuit::generate_preprocessed_columns + get_airs_and_degrees_with_prep + the two NPO preprocessors) and the
// execution schedule, as plain host code over HostCircuit - no device, no p3r_ctx.  Included into p3r_core.hip through
// circuit_impl.hip.h (which cites the reference items), where it is the host restatement behind P3R_PREP_HOST and the
// error path of the device-side preparation; and into the sanitizer build of the host-only code (tests/san/), because a
// parent node of an aggregation tree runs it on circuits derived from other ranks' bytes.
#pragma once
#include <future>
#include <numeric>
#include <unordered_map>
#include <string>
#include <vector>

#include "context.h"
#include "run_schedule.h"

using namespace p3r;

namespace {

// ---------------------------------------------------------------- preprocessing (host, once)
struct CircuitTables {
  p3r_layer_desc_counts counts{};
  std::vector<uint32_t> const_prep, public_prep, alu_prep13, recompose_prep;
  // ops of the "recompose/coeff" kind (aux = 1): the layer's ONE Recompose table when the circuit has no plain
  // Recompose op (recompose_coeff: recompose_prep holds them), its second table otherwise
  std::vector<uint32_t> recompose_coeff_prep;
  bool recompose_coeff = false;
  std::vector<uint8_t> p2_new_start, p2_merkle_path, p2_mmcs_ctl_enabled, p2_in_ctl;
  std::vector<uint32_t> p2_input_indices, p2_out_ctl, p2_output_indices, p2_mmcs_index_sum_idx;
  std::vector<uint8_t> p2_absorb_len;   // base-mode rows (circuits of degree 1 / 5)
  // rows of the width-32 table (P3R_OP_POSEIDON2_W32_PERM), assembled: Poseidon2PreprocessedRow<8, 6>, 48 columns
  std::vector<uint32_t> p2w_prep;
};


// Recompose ops come in two kinds: "recompose" (aux = 0) and "recompose/coeff" (aux = 1:
// NpoTypeId::recompose_with_coeff_lookups, circuit/src/ops/npo.rs:48-60).  Each kind is its own table
// (recompose_table_provers(lanes, true), batch_stark_prover.rs:1914-1932).

// Poseidon2 op layout by circuit degree: D = 4 -> four input limbs of four elements, two (or four) output limbs;
// otherwise base mode -> sixteen one-element slots, eight (or sixteen) outputs.  ext = [in.., index_sum, bit, n_out, out..]
struct P2Shape {
  uint32_t il, ol, ol_full;
  explicit P2Shape(uint32_t D) : il(D == 4 ? 4 : 16), ol(D == 4 ? 2 : 8), ol_full(D == 4 ? 4 : 16) {}
};

struct HostCircuit {
  uint32_t witness_count = 0;
  std::vector<p3r_op> ops;
  std::vector<uint32_t> ext, public_rows, private_rows, rewrite;
  const uint32_t* ext_of(const p3r_op& op) const { return ext.data() + op.ext_off; }
};

// Sizes first, before anything is allocated from them (both preparations size tables by witness_count): a witness id that
// no op field, ext entry, public / private row or rewrite pair can name does not exist for the circuit, so a count beyond
// what the arrays can reference is a malformed description, not a large circuit - and would otherwise be a multi-gigabyte
// allocation on the host and in HBM.
inline void check_circuit_sizes(const p3r_circuit_desc& d) {
  const uint64_t nameable = 5 * (uint64_t)d.n_ops + d.n_ext + d.n_public + d.n_private + 2 * (uint64_t)d.n_rewrite;
  if (d.witness_count >= (1u << 31)) fail(P3R_EINVAL, "witness_count %u is too large", d.witness_count);
  if ((uint64_t)d.witness_count > nameable + 16)
    fail(P3R_EINVAL, "witness_count %u exceeds the %llu witness ids the circuit's arrays can name", d.witness_count, (unsigned long long)nameable);
  if (d.n_ops >= (size_t(1) << 28) || d.n_ext >= (size_t(1) << 31)) fail(P3R_EINVAL, "circuit too large (%zu ops, %zu ext words)", d.n_ops, d.n_ext);
}

inline bool op_is_alu(uint32_t k) { return k >= P3R_OP_ALU_ADD && k <= P3R_OP_ALU_HORNER_ACC; }

// D: the circuit's extension degree (p3r_config.ext_degree): constants carry D coefficients, an ExtDecompositionHint
// has D outputs, Recompose packs D coefficients.
inline void validate_circuit(const HostCircuit& c, uint32_t D = 4) {
  const uint32_t nw = c.witness_count;
  // bit 31 of a stored witness id and the top bits of the error word are used as flags
  if (nw >= (1u << 31)) fail(P3R_EINVAL, "witness_count %u is too large", nw);
  if (c.ops.size() >= (size_t(1) << 28)) fail(P3R_EINVAL, "%zu ops is too many", c.ops.size());
  auto wid = [&](uint32_t w, size_t i, const char* what) {
    if (w >= nw) fail(P3R_EINVAL, "op %zu: %s witness %u out of bounds (witness_count %u)", i, what, w, nw);
  };
  auto opt = [&](uint32_t w, size_t i, const char* what) { if (w != kNoW) wid(w, i, what); };
  host_parallel_for(c.ops.size(), size_t(1) << 16, [&](size_t i0, size_t i1) {
  for (size_t i = i0; i < i1; ++i) {
    const p3r_op& op = c.ops[i];
    if ((size_t)op.ext_off + op.ext_len > c.ext.size()) fail(P3R_EINVAL, "op %zu: ext slice out of range", i);
    const uint32_t* e = c.ext_of(op);
    switch (op.kind) {
      case P3R_OP_CONST:
        wid(op.out, i, "out");
        if (op.ext_len != D) fail(P3R_EINVAL, "op %zu: a constant carries %u coefficients", i, D);
        break;
      case P3R_OP_PUBLIC: wid(op.out, i, "out"); break;
      case P3R_OP_ALU_ADD: case P3R_OP_ALU_MUL: case P3R_OP_ALU_BOOL_CHECK: case P3R_OP_ALU_MUL_ADD:
      case P3R_OP_ALU_HORNER_ACC:
        wid(op.a, i, "a"); wid(op.b, i, "b"); wid(op.out, i, "out"); opt(op.c, i, "c"); opt(op.aux, i, "intermediate_out");
        if (op.kind == P3R_OP_ALU_HORNER_ACC && (op.c == kNoW || op.aux == kNoW))
          fail(P3R_EINVAL, "op %zu: HornerAcc requires c and the accumulator witness", i);
        break;
      case P3R_OP_HINT_EXT_DECOMPOSITION:
        wid(op.a, i, "input");
        if (op.ext_len != D) fail(P3R_EINVAL, "op %zu: ExtDecompositionHint expects %u outputs, got %u", i, D, op.ext_len);
        for (uint32_t k = 0; k < op.ext_len; ++k) wid(e[k], i, "hint output");
        break;
      case P3R_OP_HINT_BINARY_DECOMPOSITION:
        wid(op.a, i, "input");
        if (op.ext_len > 31 * D) fail(P3R_EINVAL, "op %zu: BinaryDecompositionTooManyBits (%u > %u)", i, op.ext_len, 31 * D);
        for (uint32_t k = 0; k < op.ext_len; ++k) wid(e[k], i, "hint output");
        break;
      case P3R_OP_POSEIDON2_PERM: {
        const P2Shape sh(D);
        const uint32_t hdr = sh.il + 3;   // inputs, mmcs_index_sum, mmcs_bit, n_out
        if (op.ext_len < hdr || (e[hdr - 1] != sh.ol && e[hdr - 1] != sh.ol_full) || op.ext_len != hdr + e[hdr - 1])
          fail(P3R_EINVAL, "op %zu: Poseidon2 perm expects %u input limbs, mmcs_index_sum, mmcs_bit and %u or %u outputs", i,
               sh.il, sh.ol, sh.ol_full);
        for (uint32_t k = 0; k < sh.il + 2; ++k) opt(e[k], i, "poseidon2 input");
        for (uint32_t k = 0; k < e[hdr - 1]; ++k) opt(e[hdr + k], i, "poseidon2 output");
        if ((op.aux & 2) && e[sh.il + 1] == kNoW)
          fail(P3R_EINVAL, "op %zu: mmcs_bit must be provided when merkle_path=true", i);
        if (D != 4 && !(op.aux & 2))   // executor.rs:712-725
          for (uint32_t k = sh.ol; k < sh.il; ++k)
            if (e[k] != kNoW)
              fail(P3R_EINVAL, "op %zu: NonPrimitiveOpLayoutMismatch: capacity input slots must be empty on compact D=1 sponge rows", i);
        if (D != 4 && op.b > 255) fail(P3R_EINVAL, "op %zu: absorb_len %u does not fit the length tag", i, op.b);
        if (op.a >= c.ops.size()) fail(P3R_EINVAL, "op %zu: NonPrimitiveOpId(%u) out of range", i, op.a);
        break;
      }
      case P3R_OP_POSEIDON2_W32_PERM: {
        // validate_ext_inputs / validate_ext_outputs for the arity-4 shape (executor.rs:493-577)
        if (D != 4) fail(P3R_EUNSUPPORTED, "op %zu: the width-32 Poseidon2 table belongs to D = 4 circuits", i);
        if (op.ext_len < kW32Hdr || (e[kW32NOutSlot] != kW32Rate && e[kW32NOutSlot] != kW32In) || op.ext_len != kW32Hdr + e[kW32NOutSlot])
          fail(P3R_EINVAL, "op %zu: width-32 Poseidon2 perm expects 8 input limbs, mmcs_index_sum, mmcs_bit, mmcs_bit2 and 6 or 8 outputs", i);
        for (uint32_t k = 0; k < kW32NOutSlot; ++k) opt(e[k], i, "poseidon2 input");
        for (uint32_t k = 0; k < e[kW32NOutSlot]; ++k) opt(e[kW32Hdr + k], i, "poseidon2 output");
        if (e[kW32IdxSlot] != kNoW)
          fail(P3R_EUNSUPPORTED, "op %zu: mmcs_index_sum on a width-32 row (the arity-4 table has no index accumulator bus)", i);
        if ((op.aux & 2) && e[kW32BitSlot] == kNoW) fail(P3R_EINVAL, "op %zu: mmcs_bit must be provided when merkle_path=true", i);
        if ((op.aux & 2) && e[kW32Bit2Slot] == kNoW) fail(P3R_EINVAL, "op %zu: mmcs_bit2 must be provided when merkle_path=true", i);
        if (!(op.aux & 2) && (e[kW32BitSlot] != kNoW || e[kW32Bit2Slot] != kNoW))
          fail(P3R_EUNSUPPORTED, "op %zu: a direction bit on a width-32 sponge row", i);
        if (op.a >= c.ops.size()) fail(P3R_EINVAL, "op %zu: NonPrimitiveOpId(%u) out of range", i, op.a);
        break;
      }
      case P3R_OP_RECOMPOSE:
        wid(op.out, i, "out");
        if (op.a >= c.ops.size()) fail(P3R_EINVAL, "op %zu: NonPrimitiveOpId(%u) out of range", i, op.a);
        if (op.ext_len != D) fail(P3R_EINVAL, "op %zu: recompose expects 1 input group with %u witnesses", i, D);
        if (op.aux > 1 && op.aux != kNoW)
          fail(P3R_EINVAL, "op %zu: recompose aux %u (0 or P3R_NO_WITNESS = `recompose`, 1 = `recompose/coeff`)", i, op.aux);
        for (uint32_t k = 0; k < D; ++k) wid(e[k], i, "coefficient");
        break;
      default: fail(P3R_EUNSUPPORTED, "op %zu: kind %u has no table in this backend", i, op.kind);
    }
  }
  });
  for (uint32_t w : c.public_rows) if (w >= nw) fail(P3R_EINVAL, "public row witness %u out of bounds", w);
  for (uint32_t w : c.private_rows) if (w >= nw) fail(P3R_EINVAL, "private row witness %u out of bounds", w);
  for (uint32_t w : c.rewrite) if (w >= nw) fail(P3R_EINVAL, "witness_rewrite entry %u out of bounds", w);
}

// Bus roles of one ALU op (circuit.rs:337-391): 0 skip / 1 reader / 2 creator for a and c.
struct AluRoles { uint8_t a_state, c_state, b_creator, out_creator; };

template <class PP>
CircuitTables circuit_tables(const HostCircuit& c, uint32_t D = 4) {
  constexpr uint32_t P = PP::P, NEG1 = P - 1;
  auto scaled = [&](uint32_t w) { return (uint32_t)(((uint64_t)w * D) % P); };
  CircuitTables T;
  std::vector<uint32_t> reads(c.witness_count, 0);
  std::vector<uint8_t> defined(c.witness_count, 0), is_private(c.witness_count, 0), is_hint(c.witness_count, 0);
  // dup_npo_outputs is kept per op type (circuit.rs:464-491): one map per Recompose kind
  std::vector<uint8_t> dup_p2(c.witness_count, 0), dup_rec(c.witness_count, 0), dup_rec_coeff(c.witness_count, 0);
  std::vector<uint8_t> dup_p2w;   // the width-32 table is its own op type
  std::vector<const p3r_op*> p2ws;
  for (uint32_t w : c.private_rows) is_private[w] = 1;
  {
    // hint outputs not also produced by a Const / Public op (circuit.rs:263-284)
    std::vector<uint8_t> cp(c.witness_count, 0);
    for (auto& op : c.ops)
      if (op.kind == P3R_OP_CONST || op.kind == P3R_OP_PUBLIC) cp[op.out] = 1;
    for (auto& op : c.ops)
      if (op.kind == P3R_OP_HINT_EXT_DECOMPOSITION || op.kind == P3R_OP_HINT_BINARY_DECOMPOSITION)
        for (uint32_t k = 0; k < op.ext_len; ++k) {
          const uint32_t w = c.ext_of(op)[k];
          if (!cp[w]) is_hint[w] = 1;
        }
  }
  const P2Shape sh(D);
  // pass 1: who creates, who reads
  std::vector<AluRoles> roles;
  std::vector<const p3r_op*> consts, publics, alus, p2s, recs, recs_coeff;
  for (auto& op : c.ops) {
    switch (op.kind) {
      case P3R_OP_CONST: consts.push_back(&op); defined[op.out] = 1; break;
      case P3R_OP_PUBLIC: publics.push_back(&op); defined[op.out] = 1; break;
      case P3R_OP_HINT_EXT_DECOMPOSITION: case P3R_OP_HINT_BINARY_DECOMPOSITION: break;
      case P3R_OP_POSEIDON2_PERM: {
        const uint32_t* e = c.ext_of(op);
        const bool merkle = op.aux & 2;
        for (uint32_t l = 0; l < sh.il; ++l)
          if (e[l] != kNoW && !merkle) reads[e[l]]++;  // Merkle rows name the limb without a bus read
        for (uint32_t l = 0; l < sh.ol; ++l) {
          const uint32_t w = e[sh.il + 3 + l];
          if (w == kNoW) continue;
          if (defined[w]) { dup_p2[w] = 1; reads[w]++; } else defined[w] = 1;
        }
        p2s.push_back(&op);
        break;
      }
      case P3R_OP_POSEIDON2_W32_PERM: {
        // arity-4 shape (executor.rs:777-793,880-893): every named input limb is a bus read - Merkle rows too (their
        // AIR sends a bare in_ctl) - and a Merkle row reads the witnesses of its two direction bits
        const uint32_t* e = c.ext_of(op);
        if (dup_p2w.empty()) dup_p2w.assign(c.witness_count, 0);
        for (uint32_t l = 0; l < kW32In; ++l)
          if (e[l] != kNoW) reads[e[l]]++;
        for (uint32_t l = 0; l < kW32Rate; ++l) {
          const uint32_t w = e[kW32Hdr + l];
          if (w == kNoW) continue;
          if (defined[w]) { dup_p2w[w] = 1; reads[w]++; } else defined[w] = 1;
        }
        if (op.aux & 2) { reads[e[kW32BitSlot]]++; reads[e[kW32Bit2Slot]]++; }
        p2ws.push_back(&op);
        break;
      }
      case P3R_OP_RECOMPOSE: {
        const bool coeff = op.aux == 1u;
        if (defined[op.out]) { (coeff ? dup_rec_coeff : dup_rec)[op.out] = 1; reads[op.out]++; } else defined[op.out] = 1;
        (coeff ? recs_coeff : recs).push_back(&op);
        break;
      }
      default: {  // ALU
        const bool out_def = defined[op.out], b_def = defined[op.b];
        auto state_of = [&](uint32_t w) -> uint8_t {
          if (defined[w]) return 1;
          return ((is_private[w] || is_hint[w]) && !(!out_def && w == op.out)) ? 2 : 0;
        };
        AluRoles r{};
        r.a_state = state_of(op.a);
        r.c_state = op.c != kNoW ? state_of(op.c) : 0;
        const bool out_backward = out_def || is_hint[op.out];
        r.out_creator = !out_def;
        r.b_creator = (!b_def && is_private[op.b]) || (out_backward && !b_def);
        if (!r.b_creator) reads[op.b]++;
        if (!r.out_creator) reads[op.out]++;
        if (r.a_state == 1) reads[op.a]++;
        if (r.c_state == 1) reads[op.c]++;
        if (r.out_creator) defined[op.out] = 1;
        if (r.b_creator) defined[op.b] = 1;
        if (r.a_state == 2) defined[op.a] = 1;
        if (r.c_state == 2) defined[op.c] = 1;
        roles.push_back(r);
        alus.push_back(&op);
      }
    }
  }
  for (uint32_t w : c.private_rows)
    if (!defined[w]) fail(P3R_EINVAL, "UnclaimedPrivateInput { witness_id: WitnessId(%u) }", w);
  // the accumulator of a Merkle chain is read when the row is followed by a chain boundary
  // (the first padding row counts as one: batch_stark_prover.rs:149-176)
  {
    const size_t n = p2s.size();
    size_t h = 1;
    while (h < n) h <<= 1;
    for (size_t r = 0; r < n; ++r) {
      const uint32_t* e = c.ext_of(*p2s[r]);
      if (e[sh.il] == kNoW || !(p2s[r]->aux & 2)) continue;
      const bool next_ns = r + 1 < n ? (p2s[r + 1]->aux & 1) : (h > n ? true : (p2s[0]->aux & 1));
      if (next_ns) reads[e[sh.il]]++;
    }
  }
  // pass 2: signed multiplicities.  The tables are independent of each other once the read counts are
  // known: the Poseidon2 / Recompose / Const / Public rows are filled on a second host thread while this
  // one fills the ALU rows (the largest table).
  auto mult = [&](uint32_t w) { return reads[w] % P; };
  T.counts.n_const = consts.size();
  T.counts.n_public = publics.size();
  T.counts.n_alu = std::max<size_t>(alus.size(), 1);
  T.counts.n_p2 = p2s.size();
  T.counts.n_p2w = p2ws.size();
  // a table without rows is not proved: a circuit whose Recompose ops are all of the coefficient kind has ONE
  // Recompose table, `recompose/coeff`, in the first slot (p3r_layer_desc.recompose_coeff_lookups)
  T.recompose_coeff = recs.empty() && !recs_coeff.empty();
  if (T.recompose_coeff) recs.swap(recs_coeff);
  T.counts.n_recompose = recs.size();
  T.counts.n_recompose_coeff = recs_coeff.size();
  auto small_tables = std::async(std::launch::async, [&] {
    T.const_prep.reserve(2 * consts.size());
    for (auto* op : consts) { T.const_prep.push_back(mult(op->out)); T.const_prep.push_back(scaled(op->out)); }
    T.public_prep.reserve(2 * publics.size());
    for (auto* op : publics) { T.public_prep.push_back(mult(op->out)); T.public_prep.push_back(scaled(op->out)); }
    const size_t np = p2s.size();
    T.p2_new_start.resize(np); T.p2_merkle_path.resize(np); T.p2_mmcs_ctl_enabled.resize(np);
    const uint32_t il = sh.il, ol = sh.ol;
    T.p2_in_ctl.resize(il * np); T.p2_input_indices.resize(il * np);
    T.p2_output_indices.resize(ol * np); T.p2_out_ctl.resize(ol * np); T.p2_mmcs_index_sum_idx.resize(np);
    if (D != 4) T.p2_absorb_len.resize(np);
    for (size_t r = 0; r < np; ++r) {
      const p3r_op* op = p2s[r];
      const uint32_t* e = c.ext_of(*op);
      T.p2_new_start[r] = op->aux & 1;
      T.p2_merkle_path[r] = (op->aux >> 1) & 1;
      T.p2_mmcs_ctl_enabled[r] = e[il] != kNoW;
      for (uint32_t l = 0; l < il; ++l) {
        T.p2_in_ctl[il * r + l] = e[l] != kNoW;
        T.p2_input_indices[il * r + l] = e[l] != kNoW ? e[l] : 0;
      }
      for (uint32_t l = 0; l < ol; ++l) {
        const uint32_t w = e[il + 3 + l];
        T.p2_output_indices[ol * r + l] = w != kNoW ? w : 0;
        T.p2_out_ctl[ol * r + l] = w == kNoW ? 0 : dup_p2[w] ? NEG1 : mult(w);
      }
      T.p2_mmcs_index_sum_idx[r] = e[il] != kNoW ? e[il] : 0;
      if (D != 4) T.p2_absorb_len[r] = (uint8_t)op->b;
    }
    // the width-32 table: Poseidon2PreprocessedRow<8, 6> assembled (preprocess_inputs / _outputs / _flags for
    // is_arity4_shape, executor.rs:770-893, then phase 2 of poseidon_preprocess_for_prover, batch_stark_prover.rs:177-243;
    // phase 1 skips the arity-4 op types, :121-127)
    T.p2w_prep.resize(48 * p2ws.size());
    for (size_t r = 0; r < p2ws.size(); ++r) {
      const p3r_op* op = p2ws[r];
      const uint32_t* e = c.ext_of(*op);
      const bool ns = op->aux & 1, merkle = op->aux & 2;
      uint32_t* row = &T.p2w_prep[48 * r];
      for (uint32_t l = 0; l < kW32In; ++l) {
        const bool named = e[l] != kNoW;
        row[4 * l] = named ? scaled(e[l]) : 0;
        row[4 * l + 1] = named;
        row[4 * l + 2] = !ns && !merkle && !named;
        row[4 * l + 3] = !ns && merkle && !named;
      }
      for (uint32_t l = 0; l < kW32Rate; ++l) {
        const uint32_t w = e[kW32Hdr + l];
        row[32 + 2 * l] = w != kNoW ? scaled(w) : 0;
        row[33 + 2 * l] = w == kNoW ? 0 : dup_p2w[w] ? NEG1 : mult(w);
      }
      if (merkle) {   // the accumulator slots carry the two bit witnesses
        row[44] = scaled(e[kW32BitSlot]);
        row[45] = scaled(e[kW32Bit2Slot]);
      } else {
        row[44] = 0;   // mmcs_index_sum is refused on width-32 rows (validate_circuit): idx 0, mmcs_merkle_flag 0
        row[45] = 0;
      }
      row[46] = ns;
      row[47] = merkle;
    }
    // recompose.rs:293-356: [D * out, mult]; the coefficient variant appends (D * coeff, mult) per coefficient, where
    // only a hint output is created here (its reads), any other coefficient is named with multiplicity 0
    auto rows_of = [&](const std::vector<const p3r_op*>& list, std::vector<uint32_t>& dst) {
      for (auto* op : list) {
        const bool coeff = op->aux == 1u;
        dst.push_back(scaled(op->out));
        dst.push_back((coeff ? dup_rec_coeff : dup_rec)[op->out] ? NEG1 : mult(op->out));
        if (!coeff) continue;
        for (uint32_t k = 0; k < D; ++k) {
          const uint32_t w = c.ext_of(*op)[k];
          dst.push_back(scaled(w));
          dst.push_back(is_hint[w] ? mult(w) : 0u);
        }
      }
    };
    rows_of(recs, T.recompose_prep);
    rows_of(recs_coeff, T.recompose_coeff_prep);
  });
  T.alu_prep13.resize(13 * alus.size());
  host_parallel_for(alus.size(), size_t(1) << 16, [&](size_t i0, size_t i1) {
  for (size_t i = i0; i < i1; ++i) {
    const p3r_op& op = *alus[i];
    const AluRoles& r = roles[i];
    const uint32_t c_w = op.c != kNoW ? op.c : 0;
    auto reader_col = [&](uint8_t st, uint32_t w) { return st == 1 ? 1u : st == 2 ? (P - mult(w)) % P : 0u; };
    const uint32_t row[13] = {NEG1,
                              op.kind == P3R_OP_ALU_ADD, op.kind == P3R_OP_ALU_BOOL_CHECK,
                              op.kind == P3R_OP_ALU_MUL_ADD, op.kind == P3R_OP_ALU_HORNER_ACC,
                              scaled(op.a), scaled(op.b), scaled(c_w), scaled(op.out),
                              r.b_creator ? mult(op.b) : NEG1, r.out_creator ? mult(op.out) : NEG1,
                              reader_col(r.a_state, op.a), reader_col(r.c_state, c_w)};
    std::copy(row, row + 13, T.alu_prep13.begin() + 13 * i);
  }
  });
  if (alus.empty()) T.alu_prep13.assign(13, 0);  // the dummy row of an empty ALU table (common.rs:283-286)
  small_tables.get();
  return T;
}

// ---------------------------------------------------------------- execution schedule (host, once)
// (RunOp / RunP2 / RunSchedule: run_schedule.h, shared with the device-side preparation)
inline RunSchedule build_schedule(const HostCircuit& c, uint32_t D = 4) {
  const P2Shape sh(D);
  RunSchedule S;
  const uint32_t nw = c.witness_count;
  std::vector<uint8_t> set(nw, 0);
  std::vector<uint32_t> wlevel(nw, 0);
  for (uint32_t w : c.public_rows) set[w] = 1;
  for (uint32_t w : c.private_rows) set[w] = 1;
  auto defer = [&](const char* fmt, auto... args) {
    if (!S.deferred_error.empty()) return;
    char buf[256];
    snprintf(buf, sizeof buf, fmt, args...);
    S.deferred_error = buf;
  };
  struct Tmp { uint32_t level; bool is_p2; uint32_t idx; };
  std::vector<Tmp> order;
  struct OpenP2 { uint32_t level; std::vector<RunP2> rows; std::vector<RunP2B> rows_b; };
  std::vector<OpenP2> p2open;        // every segment; open_normal / open_merkle index the growing ones
  int open_normal = -1, open_merkle = -1;
  uint32_t n_p2_rows = 0;
  struct OpenChain { uint32_t level, first, n, acc_w, b_w, last_out; size_t last_op; };
  std::vector<OpenChain> chains;     // closed + (last one possibly) open
  bool chain_open = false;
  std::vector<RunOp> light;
  uint32_t n_alu = 0, n_rec = 0, n_pub = 0;
  // trace rows of the "recompose/coeff" ops follow those of the plain ops in the one recompose_values array
  uint32_t n_rec_plain_total = 0, n_rec_coeff = 0;
  for (auto& op : c.ops) n_rec_plain_total += op.kind == P3R_OP_RECOMPOSE && op.aux != 1u;
  uint32_t last_normal = kNoW, last_merkle = kNoW;
  // the width-32 table is its own op type with its own chain state (PoseidonExecutionState per op_type); an arity-4
  // sponge row seeds the Merkle state too (update_chain_state, executor.rs:462-491)
  struct OpenP2W { uint32_t level; std::vector<RunP2W> rows; };
  std::vector<OpenP2W> p2wopen;
  std::vector<uint32_t> p2w_seg_of_row;
  uint32_t w_last_normal = kNoW, w_last_merkle = kNoW, n_p2w_rows = 0;
  uint32_t max_op_id = 0;
  bool any_npo = false, any_w32 = false;
  for (auto& op : c.ops)
    if (op.kind == P3R_OP_POSEIDON2_PERM || op.kind == P3R_OP_RECOMPOSE || op.kind == P3R_OP_POSEIDON2_W32_PERM) {
      max_op_id = std::max(max_op_id, op.a); any_npo = true;
      any_w32 = any_w32 || op.kind == P3R_OP_POSEIDON2_W32_PERM;
    }
  if (any_npo) S.p2_row_of_op_id.assign((size_t)max_op_id + 1, kNoW);
  if (any_w32) S.p2w_row_of_op_id.assign((size_t)max_op_id + 1, kNoW);

  std::vector<uint32_t> written;
  light.reserve(c.ops.size());
  order.reserve(c.ops.size());
  S.dev_ext.reserve(c.ext.size());
  for (size_t i = 0; i < c.ops.size(); ++i) {
    const p3r_op& op = c.ops[i];
    const uint32_t* e = c.ext_of(op);
    uint32_t lvl = 0;
    auto need = [&](uint32_t w) {  // read of a witness that must already be set
      if (!set[w]) defer("WitnessNotSet { witness_id: WitnessId(%u) } at op %zu", w, i);
      lvl = std::max(lvl, wlevel[w]);
    };
    // a write: fresh slot, or a comparison against what is there (then the op also depends on it)
    auto put = [&](uint32_t w) -> bool {
      if (set[w]) { lvl = std::max(lvl, wlevel[w]); return true; }
      return false;
    };
    RunOp r{};
    r.kind_flags = op.kind;
    r.a = op.a; r.b = op.b; r.c = op.c; r.out = op.out; r.aux = op.aux; r.op_idx = (uint32_t)i;
    written.clear();  // (one vector for the whole walk: a heap allocation per op was a third of this function)
    switch (op.kind) {
      case P3R_OP_CONST:
        S.const_rows.push_back(op.out);
        r.ext_off = (uint32_t)S.dev_ext.size();
        S.dev_ext.insert(S.dev_ext.end(), e, e + op.ext_len);
        if (put(op.out)) r.kind_flags |= RUN_CHECK_OUT; else written.push_back(op.out);
        break;
      case P3R_OP_PUBLIC:
        if (!set[op.out]) defer("PublicInputNotSet { witness_id: WitnessId(%u) }", op.out);
        S.public_out.push_back(op.out);
        ++n_pub;
        continue;  // nothing to execute
      case P3R_OP_ALU_ADD: case P3R_OP_ALU_MUL:
        need(op.a);
        if (set[op.b]) {
          need(op.b);
          if (put(op.out)) r.kind_flags |= RUN_CHECK_OUT; else written.push_back(op.out);
        } else {
          need(op.out);
          r.kind_flags |= RUN_BACKWARD;
          written.push_back(op.b);
        }
        r.rec = n_alu++;
        break;
      case P3R_OP_ALU_BOOL_CHECK:
        need(op.a);
        if (put(op.out)) r.kind_flags |= RUN_CHECK_OUT; else written.push_back(op.out);
        r.rec = n_alu++;
        break;
      case P3R_OP_ALU_MUL_ADD:
        need(op.a); need(op.b);
        if (op.aux != kNoW) { if (put(op.aux)) r.kind_flags |= RUN_CHECK_AUX; else written.push_back(op.aux); }
        if (op.c != kNoW && op.c != op.aux) need(op.c);
        // `out` may be the witness intermediate_out just wrote
        if (op.aux != kNoW && op.out == op.aux) r.kind_flags |= RUN_CHECK_OUT;
        else if (put(op.out)) r.kind_flags |= RUN_CHECK_OUT; else written.push_back(op.out);
        r.rec = n_alu++;
        break;
      case P3R_OP_ALU_HORNER_ACC: {
        const bool ready = set[op.aux] && set[op.a] && set[op.b] && set[op.c] && !set[op.out] &&
                           op.out != op.a && op.out != op.b && op.out != op.c && op.out != op.aux;
        if (ready) {
          r.rec = n_alu++;
          // extend the open chain when this step continues it and its operands are ready in time
          if (chain_open) {
            OpenChain& ch = chains.back();
            if (ch.last_op + 1 == i && op.aux == ch.last_out && op.b == ch.b_w &&
                std::max(wlevel[op.a], wlevel[op.c]) < ch.level) {
              ch.n++; ch.last_out = op.out; ch.last_op = i;
              S.chain_ops.push_back(r);
              set[op.out] = 1; wlevel[op.out] = ch.level;
              continue;
            }
          }
          const uint32_t l = 1 + std::max(std::max(wlevel[op.aux], wlevel[op.b]), std::max(wlevel[op.a], wlevel[op.c]));
          chains.push_back({l, (uint32_t)S.chain_ops.size(), 1, op.aux, op.b, op.out, i});
          chain_open = true;
          S.chain_ops.push_back(r);
          set[op.out] = 1; wlevel[op.out] = l;
          continue;
        }
        need(op.aux); need(op.a); need(op.b); need(op.c);
        if (put(op.out)) r.kind_flags |= RUN_CHECK_OUT; else written.push_back(op.out);
        r.rec = n_alu++;
        break;
      }
      case P3R_OP_HINT_EXT_DECOMPOSITION: case P3R_OP_HINT_BINARY_DECOMPOSITION:
        need(op.a);
        r.ext_off = (uint32_t)S.dev_ext.size();
        r.kind_flags |= op.ext_len << 16;
        for (uint32_t k = 0; k < op.ext_len; ++k) {
          uint32_t w = e[k];
          bool dup_in_op = false;
          for (uint32_t j = 0; j < k; ++j) dup_in_op |= e[j] == w;
          if (dup_in_op || put(w)) S.dev_ext.push_back(w | RUN_CHECK_BIT); else { S.dev_ext.push_back(w); written.push_back(w); }
        }
        break;
      case P3R_OP_RECOMPOSE:
        for (uint32_t k = 0; k < op.ext_len; ++k) need(e[k]);
        r.ext_off = (uint32_t)S.dev_ext.size();
        S.dev_ext.insert(S.dev_ext.end(), e, e + op.ext_len);
        if (put(op.out)) r.kind_flags |= RUN_CHECK_OUT; else written.push_back(op.out);
        r.rec = op.aux == 1u ? n_rec_plain_total + n_rec_coeff++ : n_rec++;
        break;
      case P3R_OP_POSEIDON2_PERM: {
        const bool new_start = op.aux & 1, merkle = op.aux & 2;
        const uint32_t il = sh.il, n_out = e[il + 2];
        RunP2 q{};
        RunP2B qb{};
        const uint32_t row = n_p2_rows++;
        if (D == 4) {
          q.flags = (op.aux & 3) | (n_out << 8);
          q.op_idx = (uint32_t)i; q.row = row; q.prev_row = kNoW;
          for (uint32_t l = 0; l < 4; ++l) q.in[l] = e[l];
          q.idx_w = e[4]; q.bit_w = e[5];
        } else {
          qb.flags = (op.aux & 3) | (n_out << 8);
          qb.op_idx = (uint32_t)i; qb.row = row; qb.prev_row = kNoW; qb.absorb_len = op.b;
          for (uint32_t l = 0; l < 16; ++l) qb.in[l] = e[l];
          qb.idx_w = e[16]; qb.bit_w = e[17];
        }
        for (uint32_t l = 0; l < il + 2; ++l) if (e[l] != kNoW) need(e[l]);
        int& open = merkle ? open_merkle : open_normal;
        if (!new_start) {
          const uint32_t prev = merkle ? last_merkle : last_normal;
          if (prev == kNoW) defer("Poseidon2ChainMissingPreviousState { operation_index: NonPrimitiveOpId(%u) }", op.a);
          q.prev_row = qb.prev_row = prev;
        }
        for (uint32_t l = 0; l < sh.ol_full; ++l) {
          const uint32_t ow = l < n_out ? e[il + 3 + l] : kNoW;
          if (D == 4) q.out[l] = ow; else qb.out[l] = ow;
          if (ow == kNoW) continue;
          bool earlier = false;
          for (uint32_t j = 0; j < l; ++j) earlier |= e[il + 3 + j] == ow;
          if (earlier || put(ow)) { if (D == 4) q.flags |= 1u << (4 + l); else qb.check_mask |= 1u << l; }
          else written.push_back(ow);
        }
        if (S.p2_row_of_op_id[op.a] != kNoW || (!S.p2w_row_of_op_id.empty() && S.p2w_row_of_op_id[op.a] != kNoW))
          fail(P3R_EINVAL, "duplicate NonPrimitiveOpId(%u)", op.a);
        S.p2_row_of_op_id[op.a] = row;
        S.p2_row_merkle.push_back(merkle);
        // `lvl` = highest level among the witnesses this row reads (or compares against)
        const bool chained = !new_start && open >= 0 && (D == 4 ? q.prev_row : qb.prev_row) != kNoW;
        if (chained && lvl < p2open[open].level) {
          if (D == 4) p2open[open].rows.push_back(q); else p2open[open].rows_b.push_back(qb);  // continues the open run of its mode
        } else {
          uint32_t seg_level = lvl + 1;
          if (!new_start && open >= 0) seg_level = std::max(seg_level, p2open[open].level + 1);
          p2open.push_back({seg_level, {}, {}});
          if (D == 4) p2open.back().rows.push_back(q); else p2open.back().rows_b.push_back(qb);
          open = (int)p2open.size() - 1;
        }
        if (merkle) last_merkle = row; else last_normal = row;
        for (uint32_t w : written) { set[w] = 1; wlevel[w] = p2open[open].level; }
        continue;
      }
      case P3R_OP_POSEIDON2_W32_PERM: {
        const bool new_start = op.aux & 1, merkle = op.aux & 2;
        const uint32_t n_out = e[kW32NOutSlot];
        RunP2W q{};
        const uint32_t row = n_p2w_rows++;
        q.flags = (op.aux & 3) | (n_out << 8);
        q.op_idx = (uint32_t)i; q.row = row; q.prev_row = kNoW;
        for (uint32_t l = 0; l < kW32In; ++l) { q.in[l] = e[l]; if (e[l] != kNoW) need(e[l]); }
        q.bit_w = e[kW32BitSlot]; q.bit2_w = e[kW32Bit2Slot];
        if (q.bit_w != kNoW) need(q.bit_w);
        if (q.bit2_w != kNoW) need(q.bit2_w);
        if (!new_start) {
          const uint32_t prev = merkle ? w_last_merkle : w_last_normal;
          if (prev == kNoW) defer("Poseidon2ChainMissingPreviousState { operation_index: NonPrimitiveOpId(%u) }", op.a);
          q.prev_row = prev;
        }
        for (uint32_t l = 0; l < kW32In; ++l) {
          const uint32_t ow = l < n_out ? e[kW32Hdr + l] : kNoW;
          q.out[l] = ow;
          if (ow == kNoW) continue;
          bool earlier = false;
          for (uint32_t j = 0; j < l; ++j) earlier |= e[kW32Hdr + j] == ow;
          if (earlier || put(ow)) q.flags |= 1u << (16 + l);
          else written.push_back(ow);
        }
        if (S.p2_row_of_op_id[op.a] != kNoW || S.p2w_row_of_op_id[op.a] != kNoW) fail(P3R_EINVAL, "duplicate NonPrimitiveOpId(%u)", op.a);
        S.p2w_row_of_op_id[op.a] = row;
        S.p2w_row_merkle.push_back(merkle);
        // joins the segment that ends in its predecessor when everything it reads is ready before that segment starts
        int seg = -1;
        if (q.prev_row != kNoW) {
          const uint32_t ps = p2w_seg_of_row[q.prev_row];
          if (p2wopen[ps].rows.back().row == q.prev_row && lvl < p2wopen[ps].level) { seg = (int)ps; q.prev_in_seg = 1; }
        }
        if (seg < 0) {
          uint32_t seg_level = lvl + 1;
          if (q.prev_row != kNoW) seg_level = std::max(seg_level, p2wopen[p2w_seg_of_row[q.prev_row]].level + 1);
          p2wopen.push_back({seg_level, {}});
          seg = (int)p2wopen.size() - 1;
        }
        p2wopen[seg].rows.push_back(q);
        p2w_seg_of_row.push_back((uint32_t)seg);
        if (merkle) w_last_merkle = row; else w_last_normal = w_last_merkle = row;
        for (uint32_t w : written) { set[w] = 1; wlevel[w] = p2wopen[seg].level; }
        continue;
      }
      default: fail(P3R_EUNSUPPORTED, "op %zu: unsupported kind %u", i, op.kind);
    }
    lvl += 1;
    for (uint32_t w : written) { set[w] = 1; wlevel[w] = lvl; }
    order.push_back({lvl, false, (uint32_t)light.size()});
    light.push_back(r);
  }
  // ALU-dedup leftovers (runner.rs:199-216)
  uint32_t max_level = 0;
  for (auto& t : order) max_level = std::max(max_level, t.level);
  for (auto& ch : chains) max_level = std::max(max_level, ch.level);
  for (auto& sg : p2open) max_level = std::max(max_level, sg.level);
  for (auto& sg : p2wopen) max_level = std::max(max_level, sg.level);
  std::unordered_map<uint32_t, uint32_t> canon_of;
  for (size_t k = 0; k + 1 < c.rewrite.size(); k += 2) canon_of.emplace(c.rewrite[k], c.rewrite[k + 1]);
  for (size_t k = 0; k + 1 < c.rewrite.size(); k += 2) {
    const uint32_t dup = c.rewrite[k];
    uint32_t cur = c.rewrite[k + 1];
    // follow the chain to its root (WitnessId::resolve, circuit/src/types.rs:20-27); a cycle never ends there either
    for (size_t hops = 0; hops <= canon_of.size(); ++hops) {
      auto it = canon_of.find(cur);
      if (it == canon_of.end()) break;
      cur = it->second;
    }
    if (!set[cur]) continue;
    S.rewrite_pairs.insert(S.rewrite_pairs.end(), {dup, cur, (uint32_t)set[dup]});
    set[dup] = 1;
  }
  for (uint32_t w = 0; w < nw; ++w)
    if (!set[w]) { defer("WitnessNotSetForIndex { index: %u }", w); break; }
  // sort by level (stable: circuit order inside a level)
  S.levels = max_level;
  S.light_off.assign(max_level + 2, 0);
  S.p2seg_off.assign(max_level + 2, 0);
  for (auto& t : order) S.light_off[t.level + 1]++;
  for (auto& sg : p2open) S.p2seg_off[sg.level + 1]++;
  for (size_t l = 1; l < S.light_off.size(); ++l) { S.light_off[l] += S.light_off[l - 1]; S.p2seg_off[l] += S.p2seg_off[l - 1]; }
  S.light.resize(light.size());
  {
    std::vector<uint32_t> lp(S.light_off.begin(), S.light_off.end() - 1);
    for (auto& t : order) S.light[lp[t.level]++] = light[t.idx];
  }
  {
    // rows of a segment contiguous, segments in level order
    std::vector<uint32_t> by_level(p2open.size());
    std::iota(by_level.begin(), by_level.end(), 0u);
    std::stable_sort(by_level.begin(), by_level.end(), [&](uint32_t x, uint32_t y) { return p2open[x].level < p2open[y].level; });
    for (uint32_t k : by_level) {
      if (D == 4) {
        S.p2segs.push_back({(uint32_t)S.p2.size(), (uint32_t)p2open[k].rows.size()});
        S.p2.insert(S.p2.end(), p2open[k].rows.begin(), p2open[k].rows.end());
      } else {
        S.p2segs.push_back({(uint32_t)S.p2b.size(), (uint32_t)p2open[k].rows_b.size()});
        S.p2b.insert(S.p2b.end(), p2open[k].rows_b.begin(), p2open[k].rows_b.end());
      }
    }
  }
  if (!p2wopen.empty()) {
    S.p2wseg_off.assign(max_level + 2, 0);
    for (auto& sg : p2wopen) S.p2wseg_off[sg.level + 1]++;
    for (size_t l = 1; l < S.p2wseg_off.size(); ++l) S.p2wseg_off[l] += S.p2wseg_off[l - 1];
    std::vector<uint32_t> by_level(p2wopen.size());
    std::iota(by_level.begin(), by_level.end(), 0u);
    std::stable_sort(by_level.begin(), by_level.end(), [&](uint32_t x, uint32_t y) { return p2wopen[x].level < p2wopen[y].level; });
    for (uint32_t k : by_level) {
      S.p2wsegs.push_back({(uint32_t)S.p2w.size(), (uint32_t)p2wopen[k].rows.size()});
      S.p2w.insert(S.p2w.end(), p2wopen[k].rows.begin(), p2wopen[k].rows.end());
    }
  }
  (void)n_pub;
  S.chain_off.assign(max_level + 2, 0);
  for (auto& ch : chains) S.chain_off[ch.level + 1]++;
  for (size_t l = 1; l < S.chain_off.size(); ++l) S.chain_off[l] += S.chain_off[l - 1];
  S.chains.resize(chains.size());
  S.chain_long.assign(max_level + 2, 0);
  {
    // a chain longer than kLongChain steps is scanned by a whole workgroup, the others by one wave
    std::vector<uint32_t> cp(S.chain_off.begin(), S.chain_off.end() - 1);
    for (auto& ch : chains)
      if (ch.n > 128) { S.chains[cp[ch.level]++] = {ch.first, ch.n, ch.acc_w, ch.b_w}; S.chain_long[ch.level]++; }
    for (auto& ch : chains)
      if (ch.n <= 128) S.chains[cp[ch.level]++] = {ch.first, ch.n, ch.acc_w, ch.b_w};
  }
  finish_segments(S);
  S.n_alu_records = n_alu;
  return S;
}

}  // namespace
