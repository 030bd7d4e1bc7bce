// The caller side of prove_next_layer on the GPU (included into p3r_core.hip): a flattened
// Circuit<EF> is prepared once (preprocessed columns + execution schedule) and then RUN in HBM:
// the witness table and the resulting Traces never leave the device before prove_all_tables.
//
//   Circuit::generate_preprocessed_columns::<4>      circuit/src/circuit.rs:237-510
//   NPO executors' preprocess()                      circuit/src/ops/poseidon_perm/executor.rs:770-920,
//                                                    circuit/src/ops/recompose.rs:172-192
//   get_airs_and_degrees_with_prep                   circuit-prover/src/common.rs:127-390
//   poseidon_preprocess_for_prover                   circuit-prover/src/batch_stark_prover.rs:97-246
//   recompose_preprocess_for_op                      circuit-prover/src/batch_stark_prover/recompose.rs:294-358
//   CircuitRunner::{set_public_inputs, set_private_inputs, set_private_data, execute_all, run}
//                                                    circuit/src/tables/runner.rs:83-510
//   PoseidonPermExecutor::execute (D=4, width 16)    circuit/src/ops/poseidon_perm/executor.rs:921-972
//   RecomposeExecutor::execute                       circuit/src/ops/recompose.rs:115-170
//   Ext / BinaryDecompositionHint                    circuit/src/builder/circuit_builder.rs:1659-1810
//
// The reference runner is a sequential interpreter.  Which witnesses are set when an op runs
// is a property of the circuit, not of the inputs, so everything the interpreter decides at
// run time - the direction of an Add / Mul (forward, or solving for `b`), whether a write
// lands on a fresh slot or must equal the value already there - is decided here once, at
// setup.  Ops are then levelised (an op's level = 1 + the highest level among the ops that
// produce what it reads, including the Poseidon2 chaining through the previous permutation of
// the same mode) and every level runs as one launch: one lane per ALU / hint / recompose op,
// sixteen lanes per Poseidon2 permutation (kernels_coop.hip.h).  Ops write their trace records
// (AluOpRecord, Poseidon2CircuitRow, RecomposeCircuitRow) as they execute.

#include "circuit_host.h"

namespace {

// ---------------------------------------------------------------- kernels
// The witness table holds D Montgomery words per witness (row-major, [witness_count][D]); E = the circuit's element
// type (Fp1 / Fp4 / Fp5, field.h).  A D = 4 row is one 16-byte access.
template <class PP, int D = 4>
__device__ __forceinline__ typename CircuitExt<PP, D>::type w_load(const uint32_t* __restrict__ w, uint32_t id) {
  typename CircuitExt<PP, D>::type e;
  if constexpr (D == 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(w + (size_t)id * 4);
    e.c[0] = Fp<PP>::raw(v.x); e.c[1] = Fp<PP>::raw(v.y); e.c[2] = Fp<PP>::raw(v.z); e.c[3] = Fp<PP>::raw(v.w);
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k) e.c[k] = Fp<PP>::raw(w[(size_t)id * D + k]);
  }
  return e;
}
template <class PP, int D = 4>
__device__ __forceinline__ void w_store(uint32_t* __restrict__ w, uint32_t id, const typename CircuitExt<PP, D>::type& e) {
  if constexpr (D == 4) {
    *reinterpret_cast<uint4*>(w + (size_t)id * 4) = make_uint4(e.c[0].v, e.c[1].v, e.c[2].v, e.c[3].v);
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k) w[(size_t)id * D + k] = e.c[k].v;
  }
}
// one AluOpRecord operand ([a, b, c, out] x D words per op)
template <class PP, int D, class E>
__device__ __forceinline__ void rec_store(uint32_t* __restrict__ alu_values, uint32_t rec, int operand, const E& e) {
  uint32_t* dst = alu_values + ((size_t)rec * 4 + operand) * D;
  if constexpr (D == 4) {
    *reinterpret_cast<uint4*>(dst) = make_uint4(e.c[0].v, e.c[1].v, e.c[2].v, e.c[3].v);
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k) dst[k] = e.c[k].v;
  }
}
__device__ __forceinline__ void run_error(uint32_t* err, uint32_t op_idx, uint32_t code) {
  // the runner stops at the FIRST failing op of the sequence: keep the smallest op index
  atomicMin(err, (op_idx << 3) | code);
}
template <class PP, int D = 4>
__device__ __forceinline__ void w_put(uint32_t* __restrict__ w, uint32_t id, const typename CircuitExt<PP, D>::type& e,
                                      bool check, uint32_t* err, uint32_t op_idx) {
  if (check) {
    if (!(w_load<PP, D>(w, id) == e)) run_error(err, op_idx, RUN_ERR_CONFLICT);
  } else {
    w_store<PP, D>(w, id, e);
  }
}

struct RunArgs {
  const RunOp* light;   // sorted by level
  const RunP2* p2;
  const RunP2B* p2b;    // base-mode rows (circuits of degree 1 / 5) instead
  uint32_t* w;          // witness table, [witness_count][4] Montgomery
  const uint32_t* ext;
  uint32_t* alu_values;
  uint32_t* rec_values;
  uint32_t* p2_inputs;
  size_t p2_h;
  uint8_t* p2_flags;
  uint32_t* p2_seed;
  uint32_t* p2_out;     // permutation outputs by row (chain state)
  const int32_t* pd_slot;
  const uint32_t* siblings;
  const uint32_t* rc;
  const uint32_t* diag;
  uint32_t* err;
  const uint32_t* light_off;  // per level, device copies of the schedule offsets
  const RunSchedule::P2Seg* p2segs;
  const uint32_t* p2seg_off;
  // rows of the width-32 table (P3R_OP_POSEIDON2_W32_PERM); p2wseg_off == nullptr: the circuit has none
  const RunP2W* p2w;
  uint32_t* p2w_inputs;   // [32][p2w_h]
  size_t p2w_h;
  uint8_t* p2w_flags;     // new_start[h] | merkle_path[h] | mmcs_bit[h] | mmcs_bit2[h]
  uint32_t* p2w_out;      // permutation outputs by row (chain state)
  const int32_t* pdw_slot;
  const uint32_t* siblings_w;   // [n][24]
  const uint32_t* rcw;    // constants of the width-32 permutation (round constants | diagonal), Montgomery
  const RunSchedule::P2Seg* p2wsegs;
  const uint32_t* p2wseg_off;
};

// One ALU / hint / recompose / const op.
template <class PP, int D = 4>
__device__ __forceinline__ void run_light_op(const RunArgs& A, const RunOp& op) {
  using F = Fp<PP>;
  using E = typename CircuitExt<PP, D>::type;
  uint32_t* __restrict__ w = A.w;
  const uint32_t* __restrict__ ext = A.ext;
  uint32_t* err = A.err;
  const uint32_t kind = op.kind_flags & 0xFF;
  const bool chk_out = op.kind_flags & RUN_CHECK_OUT;
  auto record = [&](const E& a, const E& b, const E& c, const E& o) {  // AluOpRecord (runner.rs:317-453)
    rec_store<PP, D>(A.alu_values, op.rec, 0, a);
    rec_store<PP, D>(A.alu_values, op.rec, 1, b);
    rec_store<PP, D>(A.alu_values, op.rec, 2, c);
    rec_store<PP, D>(A.alu_values, op.rec, 3, o);
  };
  switch (kind) {
    case P3R_OP_CONST: {
      E v;
      for (int k = 0; k < D; ++k) v.c[k] = F::raw(ext[op.ext_off + k]);  // stored in Montgomery form
      w_put<PP, D>(w, op.out, v, chk_out, err, op.op_idx);
      break;
    }
    case P3R_OP_ALU_ADD: case P3R_OP_ALU_MUL: {
      const E a = w_load<PP, D>(w, op.a);
      E b, o;
      if (op.kind_flags & RUN_BACKWARD) {
        o = w_load<PP, D>(w, op.out);
        if (kind == P3R_OP_ALU_ADD) b = o - a;
        else {
          if (a == E::zero()) { run_error(err, op.op_idx, RUN_ERR_DIV0); b = E::zero(); }
          else b = o * a.inv();
        }
        w_store<PP, D>(w, op.b, b);
      } else {
        b = w_load<PP, D>(w, op.b);
        o = kind == P3R_OP_ALU_ADD ? a + b : a * b;
        w_put<PP, D>(w, op.out, o, chk_out, err, op.op_idx);
      }
      record(a, b, E::zero(), o);
      break;
    }
    case P3R_OP_ALU_BOOL_CHECK: {
      const E a = w_load<PP, D>(w, op.a);
      w_put<PP, D>(w, op.out, a, chk_out, err, op.op_idx);
      record(a, E::zero(), a, a);
      break;
    }
    case P3R_OP_ALU_MUL_ADD: {
      const E a = w_load<PP, D>(w, op.a), b = w_load<PP, D>(w, op.b), ab = a * b;
      if (op.aux != kNoW) w_put<PP, D>(w, op.aux, ab, op.kind_flags & RUN_CHECK_AUX, err, op.op_idx);
      const E c = op.c != kNoW ? (op.c == op.aux ? ab : w_load<PP, D>(w, op.c)) : E::zero();
      const E o = ab + c;
      if (op.aux != kNoW && op.out == op.aux) { if (!(o == ab)) run_error(err, op.op_idx, RUN_ERR_CONFLICT); }
      else w_put<PP, D>(w, op.out, o, chk_out, err, op.op_idx);
      record(a, b, c, o);
      break;
    }
    case P3R_OP_ALU_HORNER_ACC: {
      const E acc = w_load<PP, D>(w, op.aux), a = w_load<PP, D>(w, op.a), b = w_load<PP, D>(w, op.b), c = w_load<PP, D>(w, op.c);
      const E o = acc * b + c - a;
      w_put<PP, D>(w, op.out, o, chk_out, err, op.op_idx);
      record(a, b, c, o);
      break;
    }
    case P3R_OP_HINT_EXT_DECOMPOSITION: {
      const E v = w_load<PP, D>(w, op.a);
      for (int k = 0; k < D; ++k) {
        const uint32_t t = ext[op.ext_off + k];
        w_put<PP, D>(w, t & ~RUN_CHECK_BIT, E::from_base(v.c[k]), t & RUN_CHECK_BIT, err, op.op_idx);
      }
      break;
    }
    case P3R_OP_HINT_BINARY_DECOMPOSITION: {
      const E v = w_load<PP, D>(w, op.a);
      const uint32_t n_out = (op.kind_flags >> 16) & 0xFF;
      uint32_t o = 0;
      for (int k = 0; k < D && o < n_out; ++k) {
        const uint32_t val = v.c[k].to_canonical();
        for (int bit = 0; bit < 31 && o < n_out; ++bit, ++o) {
          const uint32_t t = ext[op.ext_off + o];
          const E e = ((val >> bit) & 1) ? E::one() : E::zero();
          w_put<PP, D>(w, t & ~RUN_CHECK_BIT, e, t & RUN_CHECK_BIT, err, op.op_idx);
        }
      }
      break;
    }
    case P3R_OP_RECOMPOSE: {
      E v;
      for (int k = 0; k < D; ++k) {
        v.c[k] = w_load<PP, D>(w, ext[op.ext_off + k]).c[0];
        A.rec_values[(size_t)op.rec * D + k] = v.c[k].v;
      }
      w_put<PP, D>(w, op.out, v, chk_out, err, op.op_idx);
      break;
    }
    default: break;
  }
}

// One segment of chained Poseidon2 permutations: lane j of a 16-lane group owns state element j
// (limb j/4, coefficient j%4) and walks the rows of the segment with the state in a register.
// Whole 16-lane groups call this together (DPP inside coop_permute); `live` masks the tail groups.
template <class PP>
__device__ __forceinline__ void run_p2_segment(const RunArgs& A, RunSchedule::P2Seg seg, int j, bool live) {
  using F = Fp<PP>;
  using E = Fp4<PP>;
  uint32_t* __restrict__ w = A.w;
  uint32_t* err = A.err;
  if (!live) seg.n = 0;
  const CoopRc<PP> rcs = coop_load_rc<PP>(A.rc, j);
  const F diag_j = F::raw(A.diag[j]);
  F carried = F::zero();  // output of the previous row of this segment
  RunP2 next = seg.n ? A.p2[seg.first] : RunP2{};
  for (uint32_t k = 0; k < seg.n; ++k) {
    const RunP2 q = next;
    if (k + 1 < seg.n) next = A.p2[seg.first + k + 1];  // in flight while this row is permuted
    const bool new_start = q.flags & 1, merkle = q.flags & 2;
    // init_chain_state (executor.rs:103-139): Merkle rows carry the rate limbs only
    F s = F::zero();
    if (!new_start && (!merkle || j < 8)) s = k ? carried : F::raw(A.p2_out[(size_t)q.prev_row * 16 + j]);
    // fill_sibling_data (:166-201): private sibling in the capacity limbs
    const int32_t slot = A.pd_slot[q.row];
    if (merkle && slot >= 0 && j >= 8) s = F::raw(A.siblings[(size_t)slot * 8 + (j - 8)]);
    // apply_witness_values (:207-219)
    const uint32_t in_w = q.in[j >> 2];
    if (in_w != kNoW) s = F::raw(w[(size_t)in_w * 4 + (j & 3)]);
    // resolve_mmcs_bit (:283-338)
    bool bit = false;
    if (q.bit_w != kNoW) {
      const E v = w_load<PP>(w, q.bit_w);
      if (v == E::one()) bit = true;
      else if (!(v == E::zero()) && j == 0) run_error(err, q.op_idx, RUN_ERR_MMCS_BIT);
    }
    // apply_merkle_swap (:227-234): the two halves trade places
    {
      const uint32_t other = __shfl_xor(s.v, 8);
      if (merkle && bit) s = F::raw(other);
    }
    // Poseidon2CircuitRow (build_trace_row :364-417, trace.rs:188-233)
    A.p2_inputs[(size_t)j * A.p2_h + q.row] = s.v;
    if (j == 0) {
      A.p2_flags[q.row] = q.flags & 1;
      A.p2_flags[A.p2_h + q.row] = (q.flags >> 1) & 1;
      A.p2_flags[2 * A.p2_h + q.row] = bit;
      uint32_t seed = 0;
      if (q.idx_w != kNoW) {
        const E v = w_load<PP>(w, q.idx_w);
        if (v.c[1].v | v.c[2].v | v.c[3].v) run_error(err, q.op_idx, RUN_ERR_INDEX_SUM);
        seed = v.c[0].v;
      }
      A.p2_seed[q.row] = seed;
    }
    s = coop_permute<PP>(s, j, diag_j, rcs);
    carried = s;
    A.p2_out[(size_t)q.row * 16 + j] = s.v;
    const uint32_t n_out = (q.flags >> 8) & 7;
    const uint32_t l = (uint32_t)j >> 2;
    if (l < n_out && q.out[l] != kNoW) {
      uint32_t* slot_w = w + (size_t)q.out[l] * 4 + (j & 3);
      if (q.flags & (1u << (4 + l))) { if (*slot_w != s.v) run_error(err, q.op_idx, RUN_ERR_CONFLICT); }
      else *slot_w = s.v;
    }
  }
}

// One segment of width-32 permutations (P3R_OP_POSEIDON2_W32_PERM): lane j of a 32-lane group owns state element j
// (limb j/4, coefficient j%4; chunk j/8 of the arity-4 shape).  PoseidonPermExecutor::execute for is_arity4()
// (poseidon_perm/executor.rs:92-235,947-966): no swap - the running digest is PLACED into chunk pos = bit + 2 bit2.
template <class PP>
__device__ __forceinline__ void run_p2w_segment(const RunArgs& A, RunSchedule::P2Seg seg, int j, bool live) {
  using F = Fp<PP>;
  using E = Fp4<PP>;
  uint32_t* __restrict__ w = A.w;
  uint32_t* err = A.err;
  if (!live) seg.n = 0;
  const Coop32Rc<PP> rcs = coop32_load_rc<PP>(A.rcw, j);
  F carried = F::zero();  // output of the previous row of this segment
  RunP2W next = seg.n ? A.p2w[seg.first] : RunP2W{};
  for (uint32_t k = 0; k < seg.n; ++k) {
    const RunP2W q = next;
    if (k + 1 < seg.n) next = A.p2w[seg.first + k + 1];  // in flight while this row is permuted
    const bool new_start = q.flags & 1, merkle = q.flags & 2;
    // resolve_mmcs_bit / resolve_mmcs_bit2 (:283-338)
    bool bit = false, bit2 = false;
    if (q.bit_w != kNoW) {
      const E v = w_load<PP>(w, q.bit_w);
      if (v == E::one()) bit = true;
      else if (!(v == E::zero()) && j == 0) run_error(err, q.op_idx, RUN_ERR_MMCS_BIT);
    }
    if (q.bit2_w != kNoW) {
      const E v = w_load<PP>(w, q.bit2_w);
      if (v == E::one()) bit2 = true;
      else if (!(v == E::zero()) && j == 0) run_error(err, q.op_idx, RUN_ERR_MMCS_BIT);
    }
    const int pos = (int)bit + 2 * (int)bit2, chunk = j >> 3;
    // previous output of this row's chain: still in registers when the predecessor is the row before, else in memory
    // (it was written by an earlier level); the digest = its first eight elements goes to the lanes of chunk `pos`
    F prev = F::zero();
    if (!new_start) {
      if (q.prev_in_seg) prev = merkle ? F::raw(__shfl(carried.v, j & 7, 32)) : carried;
      else prev = F::raw(A.p2w_out[(size_t)q.prev_row * 32 + (merkle ? (j & 7) : j)]);
    }
    // init_chain_state + place_arity4_running_hash (:103-160)
    F s = F::zero();
    if (!new_start && (!merkle || chunk == pos)) s = prev;
    // fill_sibling_data (:166-201): the three other chunks in ascending order
    const int32_t slot = A.pdw_slot[q.row];
    if (merkle && slot >= 0 && chunk != pos)
      s = F::raw(A.siblings_w[(size_t)slot * 24 + (size_t)(chunk - (chunk > pos ? 1 : 0)) * 8 + (j & 7)]);
    // apply_witness_values (:207-219)
    const uint32_t in_w = q.in[j >> 2];
    if (in_w != kNoW) s = F::raw(w[(size_t)in_w * 4 + (j & 3)]);
    // Poseidon2CircuitRow (build_trace_row :364-417): mmcs_index_sum stays zero (no accumulator witness on this table)
    A.p2w_inputs[(size_t)j * A.p2w_h + q.row] = s.v;
    if (j == 0) {
      A.p2w_flags[q.row] = new_start;
      A.p2w_flags[A.p2w_h + q.row] = merkle;
      A.p2w_flags[2 * A.p2w_h + q.row] = bit;
      A.p2w_flags[3 * A.p2w_h + q.row] = bit2;
    }
    s = coop32_permute<PP>(s, j, rcs);
    carried = s;
    A.p2w_out[(size_t)q.row * 32 + j] = s.v;
    const uint32_t n_out = (q.flags >> 8) & 15;
    const uint32_t l = (uint32_t)j >> 2;
    if (l < n_out && q.out[l] != kNoW) {
      uint32_t* slot_w = w + (size_t)q.out[l] * 4 + (j & 3);
      if (q.flags & (1u << (16 + l))) { if (*slot_w != s.v) run_error(err, q.op_idx, RUN_ERR_CONFLICT); }
      else *slot_w = s.v;
    }
  }
}

// The same for the base-mode permutation rows of a circuit of degree D = 1 / 5 (one witness per state element,
// poseidon_perm/executor.rs:600-700 for sponge rows - the chained state, CTL inputs on any slot the op names, the
// length tag added to the first capacity element - and :924-970 for Merkle rows, sixteen one-element limbs).
// A witness is a D-word element; the permutation acts on its constant term (LiftPermToQuintic) and an output is
// written back as a base-field element of the circuit's field.
template <class PP, int D>
__device__ __forceinline__ void run_p2_segment_base(const RunArgs& A, RunSchedule::P2Seg seg, int j, bool live) {
  using F = Fp<PP>;
  uint32_t* __restrict__ w = A.w;
  uint32_t* err = A.err;
  if (!live) seg.n = 0;
  const CoopRc<PP> rcs = coop_load_rc<PP>(A.rc, j);
  const F diag_j = F::raw(A.diag[j]);
  F carried = F::zero();
  for (uint32_t k = 0; k < seg.n; ++k) {
    const RunP2B& q = A.p2b[seg.first + k];
    const uint32_t flags = q.flags;
    const bool new_start = flags & 1, merkle = flags & 2;
    F s = F::zero();
    if (!new_start && (!merkle || j < 8)) s = k ? carried : F::raw(A.p2_out[(size_t)q.prev_row * 16 + j]);
    const int32_t slot = A.pd_slot[q.row];
    if (merkle && slot >= 0 && j >= 8) s = F::raw(A.siblings[(size_t)slot * 8 + (j - 8)]);
    const uint32_t in_w = q.in[j];
    if (in_w != kNoW) s = F::raw(w[(size_t)in_w * D]);
    // the prefix-free length tag of a sponge row (executor.rs:681-683)
    if (!merkle && j == 8 && q.absorb_len) s = s + F::from_canonical(q.absorb_len);
    bool bit = false;
    if (q.bit_w != kNoW) {
      const auto v = w_load<PP, D>(w, q.bit_w);
      using E = typename CircuitExt<PP, D>::type;
      if (v == E::one()) bit = true;
      else if (!(v == E::zero()) && j == 0) run_error(err, q.op_idx, RUN_ERR_MMCS_BIT);
    }
    {
      const uint32_t other = __shfl_xor(s.v, 8);
      if (merkle && bit) s = F::raw(other);
    }
    A.p2_inputs[(size_t)j * A.p2_h + q.row] = s.v;
    if (j == 0) {
      A.p2_flags[q.row] = flags & 1;
      A.p2_flags[A.p2_h + q.row] = (flags >> 1) & 1;
      A.p2_flags[2 * A.p2_h + q.row] = bit;
      uint32_t seed = 0;
      if (q.idx_w != kNoW) {
        const auto v = w_load<PP, D>(w, q.idx_w);
        uint32_t high = 0;
        for (int c = 1; c < D; ++c) high |= v.c[c].v;
        if (high) run_error(err, q.op_idx, RUN_ERR_INDEX_SUM);
        seed = v.c[0].v;
      }
      A.p2_seed[q.row] = seed;
    }
    s = coop_permute<PP>(s, j, diag_j, rcs);
    carried = s;
    A.p2_out[(size_t)q.row * 16 + j] = s.v;
    const uint32_t n_out = (flags >> 8) & 31;
    if ((uint32_t)j < n_out && q.out[j] != kNoW) {
      uint32_t* slot_w = w + (size_t)q.out[j] * D;
      if (q.check_mask & (1u << j)) {
        bool same = slot_w[0] == s.v;
        for (int c = 1; c < D; ++c) same = same && slot_w[c] == 0;
        if (!same) run_error(err, q.op_idx, RUN_ERR_CONFLICT);
      } else {
        slot_w[0] = s.v;
        for (int c = 1; c < D; ++c) slot_w[c] = 0;
      }
    }
  }
}

// Horner chains of one level: one WAVE per chain evaluates acc_j = acc_{j-1}*b + (c_j - a_j) as an
// affine scan - every lane folds its slice locally, the slice maps (b^len, value) are combined
// with a shuffle scan across the wave, then every lane replays its slice from its incoming
// accumulator, writing the outputs and the AluOpRecords (runner.rs:430-453).
// `block`: index among the chain blocks of the launch, four chains (waves) per block.
template <class PP, int D = 4>
__device__ __forceinline__ void run_chains(const RunArgs& A, const RunOp* __restrict__ steps,
                                           const RunSchedule::ChainSeg* __restrict__ segs, uint32_t n_segs,
                                           uint32_t block) {
  using F = Fp<PP>;
  using E = typename CircuitExt<PP, D>::type;
  const uint32_t chain = block * (kBlock / 64) + (threadIdx.x >> 6);
  if (chain >= n_segs) return;  // whole waves leave together
  const RunSchedule::ChainSeg seg = segs[chain];
  const uint32_t t = threadIdx.x & 63;
  uint32_t* __restrict__ w = A.w;
  const E b = w_load<PP, D>(w, seg.b_w);
  const uint32_t per = (seg.n + 63) / 64;
  const uint32_t i0 = min(t * per, seg.n), i1 = min(i0 + per, seg.n);
  // local fold from a zero accumulator: value V, multiplier M = b^(i1 - i0)
  E M = E::one(), V = E::zero();
  for (uint32_t i = i0; i < i1; ++i) {
    const RunOp op = steps[seg.first + i];
    V = V * b + w_load<PP, D>(w, op.c) - w_load<PP, D>(w, op.a);
    M = M * b;
  }
  auto up = [&](const E& e, int d) { E r; for (int k = 0; k < D; ++k) r.c[k] = F::raw(__shfl_up(e.c[k].v, d)); return r; };
  // inclusive scan of the maps x -> x*M + V (composition: the earlier slice is applied first)
  for (int d = 1; d < 64; d <<= 1) {
    const E pm = up(M, d), pv = up(V, d);
    if ((int)t >= d) {
      V = pv * M + V;
      M = pm * M;
    }
  }
  // incoming accumulator of this slice = (maps of all earlier slices)(acc0)
  E acc = w_load<PP, D>(w, seg.acc_w);
  {
    const E pm = up(M, 1), pv = up(V, 1);
    if (t > 0) acc = acc * pm + pv;
  }
  for (uint32_t i = i0; i < i1; ++i) {
    const RunOp op = steps[seg.first + i];
    const E a = w_load<PP, D>(w, op.a), c = w_load<PP, D>(w, op.c);
    acc = acc * b + c - a;
    w_store<PP, D>(w, op.out, acc);
    rec_store<PP, D>(A.alu_values, op.rec, 0, a);
    rec_store<PP, D>(A.alu_values, op.rec, 1, b);
    rec_store<PP, D>(A.alu_values, op.rec, 2, c);
    rec_store<PP, D>(A.alu_values, op.rec, 3, acc);
  }
}

// One WIDE level of the schedule in one launch, longest-running blocks first: a Poseidon2
// permutation segment per 16 lanes, then a short Horner chain per wave, then one light op per lane.
template <class PP, int D = 4>
__global__ void __launch_bounds__(kBlock)
k_run_level(RunArgs A, uint32_t p2_begin, uint32_t n_p2, uint32_t p2_blocks, const RunOp* __restrict__ chain_steps,
            const RunSchedule::ChainSeg* __restrict__ chains, uint32_t n_chains, uint32_t chain_blocks,
            uint32_t light_begin, uint32_t n_light, uint32_t p2w_begin = 0, uint32_t n_p2w = 0, uint32_t p2w_blocks = 0) {
  if constexpr (D == 4) {
    // width-32 permutation segments ride at the END of the grid (the existing block ranges keep their indices)
    if (p2w_blocks && blockIdx.x >= gridDim.x - p2w_blocks) {
      const uint32_t g = (blockIdx.x - (gridDim.x - p2w_blocks)) * kBlock + threadIdx.x;
      const bool live = (g >> 5) < n_p2w;
      const RunSchedule::P2Seg sg = live ? A.p2wsegs[p2w_begin + (g >> 5)] : RunSchedule::P2Seg{0, 0};
      run_p2w_segment<PP>(A, sg, (int)(g & 31), live);
      return;
    }
  }
  if (blockIdx.x < p2_blocks) {
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    const bool live = (g >> 4) < n_p2;
    const RunSchedule::P2Seg sg = live ? A.p2segs[p2_begin + (g >> 4)] : RunSchedule::P2Seg{0, 0};
    if constexpr (D == 4) run_p2_segment<PP>(A, sg, (int)(g & 15), live);
    else run_p2_segment_base<PP, D>(A, sg, (int)(g & 15), live);
    return;
  }
  if (blockIdx.x < p2_blocks + chain_blocks) {
    run_chains<PP, D>(A, chain_steps, chains, n_chains, blockIdx.x - p2_blocks);
    return;
  }
  const uint32_t i = (blockIdx.x - p2_blocks - chain_blocks) * kBlock + threadIdx.x;
  if (i < n_light) run_light_op<PP, D>(A, A.light[light_begin + i]);   // (a trailing width-32 block returned above)
}

// Long chains: the same scan with a whole workgroup per chain - slices are folded per lane, combined
// by a shuffle scan inside each wave, the sixteen wave totals by one more shuffle scan through LDS.
constexpr int kLongChainBlock = 1024;
template <class PP, int D = 4>
__global__ void __launch_bounds__(kLongChainBlock)
k_run_chains_block(RunArgs A, const RunOp* __restrict__ steps, const RunSchedule::ChainSeg* __restrict__ segs) {
  using F = Fp<PP>;
  using E = typename CircuitExt<PP, D>::type;
  constexpr int kWaves = kLongChainBlock / 64;
  __shared__ uint32_t s_m[kWaves][D], s_v[kWaves][D];
  const RunSchedule::ChainSeg seg = segs[blockIdx.x];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t* __restrict__ w = A.w;
  const E b = w_load<PP, D>(w, seg.b_w);
  const uint32_t per = (seg.n + kLongChainBlock - 1) / kLongChainBlock;
  const uint32_t i0 = min(t * per, seg.n), i1 = min(i0 + per, seg.n);
  E M = E::one(), V = E::zero();
  for (uint32_t i = i0; i < i1; ++i) {
    const RunOp op = steps[seg.first + i];
    V = V * b + w_load<PP, D>(w, op.c) - w_load<PP, D>(w, op.a);
    M = M * b;
  }
  auto up = [&](const E& e, int d) { E r; for (int k = 0; k < D; ++k) r.c[k] = F::raw(__shfl_up(e.c[k].v, d)); return r; };
  for (int d = 1; d < 64; d <<= 1) {  // inclusive scan inside the wave
    const E pm = up(M, d), pv = up(V, d);
    if ((int)lane >= d) { V = pv * M + V; M = pm * M; }
  }
  if (lane == 63)
    for (int k = 0; k < D; ++k) { s_m[wave][k] = M.c[k].v; s_v[wave][k] = V.c[k].v; }
  __syncthreads();
  if (wave == 0) {  // inclusive scan of the wave totals, kept as EXCLUSIVE prefixes per wave
    E tm = E::one(), tv = E::zero();
    if (lane < kWaves)
      for (int k = 0; k < D; ++k) { tm.c[k] = F::raw(s_m[lane][k]); tv.c[k] = F::raw(s_v[lane][k]); }
    for (int d = 1; d < kWaves; d <<= 1) {
      const E pm = up(tm, d), pv = up(tv, d);
      if ((int)lane >= d) { tv = pv * tm + tv; tm = pm * tm; }
    }
    const E em = up(tm, 1), ev = up(tv, 1);
    if (lane < kWaves)
      for (int k = 0; k < D; ++k) {
        s_m[lane][k] = lane ? em.c[k].v : E::one().c[k].v;
        s_v[lane][k] = lane ? ev.c[k].v : 0u;
      }
  }
  __syncthreads();
  // incoming accumulator: acc0 through the earlier waves, then through the earlier lanes of this wave
  E acc = w_load<PP, D>(w, seg.acc_w);
  {
    E wm, wv;
    for (int k = 0; k < D; ++k) { wm.c[k] = F::raw(s_m[wave][k]); wv.c[k] = F::raw(s_v[wave][k]); }
    acc = acc * wm + wv;
    const E pm = up(M, 1), pv = up(V, 1);
    if (lane > 0) acc = acc * pm + pv;
  }
  for (uint32_t i = i0; i < i1; ++i) {
    const RunOp op = steps[seg.first + i];
    const E a = w_load<PP, D>(w, op.a), c = w_load<PP, D>(w, op.c);
    acc = acc * b + c - a;
    w_store<PP, D>(w, op.out, acc);
    rec_store<PP, D>(A.alu_values, op.rec, 0, a);
    rec_store<PP, D>(A.alu_values, op.rec, 1, b);
    rec_store<PP, D>(A.alu_values, op.rec, 2, c);
    rec_store<PP, D>(A.alu_values, op.rec, 3, acc);
  }
}

// Runs of NARROW levels (each at most kNarrowBlock light ops and kNarrowBlock/16 permutations)
// inside ONE workgroup: the level barrier is a __syncthreads() instead of a kernel boundary.
// All waves of a workgroup share the CU's vector L1, so workgroup-scope ordering is enough for
// the witness table and the chain state to be seen by the next level.  The light-op records of
// a chunk of levels (they are contiguous: the schedule is sorted by level) are staged in LDS by
// one coalesced copy, so a level does not start with a cold HBM read of its own ops.
constexpr int kNarrowBlock = 1024;
constexpr uint32_t kNarrowLightCap = 1400;  // light-op records per chunk (56 KB of LDS)
template <class PP, int D = 4>
__global__ void __launch_bounds__(kNarrowBlock)
k_run_levels_narrow(RunArgs A, const uint32_t* __restrict__ chunk_bounds, uint32_t n_chunks) {
  __shared__ RunOp s_light[kNarrowLightCap];
  const uint32_t t = threadIdx.x;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const uint32_t l0 = chunk_bounds[c], l1 = chunk_bounds[c + 1];
    const uint32_t lb0 = A.light_off[l0];
    {
      const uint32_t nw = (A.light_off[l1] - lb0) * (uint32_t)(sizeof(RunOp) / 4);
      const uint32_t* src = reinterpret_cast<const uint32_t*>(A.light + lb0);
      uint32_t* dst = reinterpret_cast<uint32_t*>(s_light);
      for (uint32_t i = t; i < nw; i += kNarrowBlock) dst[i] = src[i];
    }
    __syncthreads();
    for (uint32_t l = l0; l < l1; ++l) {
      const uint32_t lb = A.light_off[l], nl = A.light_off[l + 1] - lb;
      const uint32_t pb = A.p2seg_off[l], np = A.p2seg_off[l + 1] - pb;
      if (t < nl) run_light_op<PP, D>(A, s_light[lb - lb0 + t]);
      const bool live = (t >> 4) < np;
      if (__any(live)) {
        const RunSchedule::P2Seg sg = live ? A.p2segs[pb + (t >> 4)] : RunSchedule::P2Seg{0, 0};
        if constexpr (D == 4) run_p2_segment<PP>(A, sg, (int)(t & 15), live);
        else run_p2_segment_base<PP, D>(A, sg, (int)(t & 15), live);
      }
      if constexpr (D == 4) {
        if (A.p2wseg_off) {
          const uint32_t wb = A.p2wseg_off[l], nwseg = A.p2wseg_off[l + 1] - wb;
          const bool wlive = (t >> 5) < nwseg;
          if (__any(wlive)) {
            const RunSchedule::P2Seg sg = wlive ? A.p2wsegs[wb + (t >> 5)] : RunSchedule::P2Seg{0, 0};
            run_p2w_segment<PP>(A, sg, (int)(t & 31), wlive);
          }
        }
      }
      __syncthreads();
    }
  }
}

// witness[rows[i]] = values[i]  (set_public_inputs / set_private_inputs, runner.rs:83-122)
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_run_scatter(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ values, size_t n, uint32_t* __restrict__ w,
              uint32_t D) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n * D) return;
  w[(size_t)rows[i / D] * D + (i % D)] = values[i];
}
// out[i] = witness[rows[i]]  (PublicTraceBuilder, tables/public.rs:44-62)
template <class PP>
__global__ void __launch_bounds__(kBlock)
k_run_gather(const uint32_t* __restrict__ rows, size_t n, const uint32_t* __restrict__ w, uint32_t* __restrict__ out,
             uint32_t D) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n * D) return;
  out[i] = w[(size_t)rows[i / D] * D + (i % D)];
}
// duplicates left behind by ALU deduplication take the value of their canonical witness
template <class PP, int D = 4>
__global__ void __launch_bounds__(kBlock)
k_run_rewrite(const uint32_t* __restrict__ triples, size_t n, uint32_t* __restrict__ w, uint32_t* __restrict__ err) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t dst = triples[3 * i], src = triples[3 * i + 1];
  w_put<PP, D>(w, dst, w_load<PP, D>(w, src), triples[3 * i + 2] != 0, err, 0x1FFFFFFFu);
}

}  // namespace

// ---------------------------------------------------------------- the prepared circuit
struct p3r_circuit {
  // `sched` always holds the per-level offsets, the launch plan and the counts; its large arrays (light ops,
  // permutation rows, chain steps) are only populated by the host-side preparation, which uploads them - the
  // device-side preparation (prep_device.hip) builds them in HBM
  RunSchedule sched;
  uint32_t witness_count = 0;
  bool prepared_on_device = false;
  size_t n_public_rows = 0, n_private_rows = 0, n_rewrite = 0, n_op_ids = 0;
  p3r_layer_desc_counts counts{};
  std::unique_ptr<p3r_layer> layer;
  p3r::DevBuf d_light, d_p2, d_ext, d_const_values, d_public_rows, d_private_rows, d_public_out, d_rewrite;
  p3r::DevBuf d_light_off, d_p2seg_off, d_p2segs, d_chunk_bounds, d_chain_ops, d_chains;
  p3r::DevBuf d_row_of_op_id;  // NonPrimitiveOpId -> Poseidon2 row, bit 31 = Merkle row; kNoW: no permutation
  // rows of the width-32 table (P3R_OP_POSEIDON2_W32_PERM; host-side preparation only)
  p3r::DevBuf d_p2w, d_p2wsegs, d_p2wseg_off;
  p3r::DevBuf d_roww_of_op_id; // NonPrimitiveOpId -> width-32 row, bit 31 = Merkle row
};

// Inputs of one run made resident in HBM (public / private values, Merkle siblings).
struct p3r_dinputs {
  p3r::DevBuf pub, priv, sib, slot;
  p3r::DevBuf sib_w, slot_w;   // private data of the width-32 Merkle rows: [n][24], row -> position
};

namespace {

template <class PP>
std::unique_ptr<p3r_circuit> circuit_create(p3r_ctx* ctx, const p3r_circuit_desc* d, uint32_t* commit_out) {
  auto C = std::make_unique<p3r_circuit>();
  HostCircuit h;
  auto need = [&](const void* p, size_t n, const char* what) { if (n && !p) fail(P3R_EINVAL, "%s is NULL", what); };
  need(d->ops, d->n_ops, "ops"); need(d->ext, d->n_ext, "ext"); need(d->public_rows, d->n_public, "public_rows");
  need(d->private_input_rows, d->n_private, "private_input_rows"); need(d->witness_rewrite, d->n_rewrite, "witness_rewrite");
  check_circuit_sizes(*d);
  C->witness_count = d->witness_count;
  C->n_public_rows = d->n_public;
  C->n_private_rows = d->n_private;
  // packing parameters first: both preparations divide by them (TablePacking::validate, packing.rs:140-161)
  if (!d->public_lanes || !d->alu_lanes || !d->recompose_lanes) fail(P3R_EINVAL, "lane counts must be positive");
  if (d->horner_packed_steps < 2 || d->horner_packed_steps > 8) fail(P3R_EINVAL, "horner_packed_steps must be in 2..8");
  // Device-side preparation (prep_device.hip): the op list crosses PCIe once; preprocessed columns, ALU lane
  // schedule and execution schedule are built in HBM.  A circuit it flags (malformed, unclaimed private input,
  // a witness nobody sets ...) goes through the host restatement below, which raises the reference's error.
  // (circuit degrees 1, 4 and 5 - the degrees the runner computes in - and both Recompose kinds)
  const uint32_t ext_d = ctx->cfg.ext_degree;
  const bool host_prep = getenv("P3R_PREP_HOST") != nullptr;   // read per call: the equality tests flip it
  if (!host_prep) {
    prof_stage(ctx, "prep_device");
    DevPrep R;
    if (devprep_circuit(ctx, d, R)) {
      prof_stage(ctx, "prep_layer_create");
      C->counts = R.counts;
      C->layer = layer_from_device<PP>(ctx, R, d, commit_out);
      C->sched = std::move(R.sched);
      C->n_rewrite = R.n_rewrite;
      C->n_op_ids = R.n_op_ids;
      C->d_light = std::move(R.d_light); C->d_p2 = std::move(R.d_p2); C->d_ext = std::move(R.d_ext);
      C->d_const_values = std::move(R.d_const_values); C->d_public_rows = std::move(R.d_public_rows);
      C->d_private_rows = std::move(R.d_private_rows); C->d_public_out = std::move(R.d_public_out);
      C->d_rewrite = std::move(R.d_rewrite); C->d_light_off = std::move(R.d_light_off);
      C->d_p2seg_off = std::move(R.d_p2seg_off); C->d_p2segs = std::move(R.d_p2segs);
      C->d_chunk_bounds = std::move(R.d_chunk_bounds); C->d_chain_ops = std::move(R.d_chain_ops);
      C->d_chains = std::move(R.d_chains); C->d_row_of_op_id = std::move(R.d_row_of_op_id);
      C->d_p2w = std::move(R.d_p2w); C->d_p2wsegs = std::move(R.d_p2wsegs); C->d_p2wseg_off = std::move(R.d_p2wseg_off);
      C->d_roww_of_op_id = std::move(R.d_roww_of_op_id);
      C->prepared_on_device = true;
      prof_stage(ctx, nullptr);
      return C;
    }
  }
  prof_stage(ctx, "prep_validate");
  h.witness_count = d->witness_count;
  h.ops.assign(d->ops, d->ops + d->n_ops);
  h.ext.assign(d->ext, d->ext + d->n_ext);
  h.public_rows.assign(d->public_rows, d->public_rows + d->n_public);
  h.private_rows.assign(d->private_input_rows, d->private_input_rows + d->n_private);
  h.rewrite.assign(d->witness_rewrite, d->witness_rewrite + 2 * d->n_rewrite);
  validate_circuit(h, ext_d);
  for (auto& op : h.ops)
    if (op.kind == P3R_OP_CONST)
      for (uint32_t k = 0; k < ext_d; ++k)
        if (h.ext_of(op)[k] >= PP::P) fail(P3R_EINVAL, "constant of witness %u is not canonical", op.out);

  // the execution schedule depends on the circuit alone (host vectors only, no device work): it is built
  // on a second host thread while this one builds and commits the preprocessed columns
  std::future<RunSchedule> sched_job = std::async(std::launch::async, [&h, ext_d] { return build_schedule(h, ext_d); });
  struct JoinOnUnwind {  // an error below must not leave the worker reading `h` after it is gone
    std::future<RunSchedule>& f;
    ~JoinOnUnwind() { if (f.valid()) f.wait(); }
  } join_on_unwind{sched_job};

  // CircuitProverData: preprocessed columns -> LDE + commitment (build_next_layer_prep)
  prof_stage(ctx, "prep_circuit_tables");
  CircuitTables T = circuit_tables<PP>(h, ext_d);
  C->counts = T.counts;
  p3r_layer_desc ld{};
  ld.counts = T.counts;
  ld.public_lanes = d->public_lanes; ld.alu_lanes = d->alu_lanes; ld.horner_packed_steps = d->horner_packed_steps;
  ld.recompose_lanes = d->recompose_lanes; ld.min_trace_height = d->min_trace_height;
  ld.const_prep = T.const_prep.data(); ld.public_prep = T.public_prep.data(); ld.alu_prep13 = T.alu_prep13.data();
  ld.recompose_prep = T.recompose_prep.data();
  ld.p2_new_start = T.p2_new_start.data(); ld.p2_merkle_path = T.p2_merkle_path.data();
  ld.p2_mmcs_ctl_enabled = T.p2_mmcs_ctl_enabled.data(); ld.p2_in_ctl = T.p2_in_ctl.data();
  ld.p2_input_indices = T.p2_input_indices.data(); ld.p2_out_ctl = T.p2_out_ctl.data();
  ld.p2_output_indices = T.p2_output_indices.data(); ld.p2_mmcs_index_sum_idx = T.p2_mmcs_index_sum_idx.data();
  if (ext_d != 4 && !T.p2_absorb_len.empty()) ld.p2_absorb_len = T.p2_absorb_len.data();
  ld.recompose_coeff_prep = T.recompose_coeff_prep.data();
  ld.recompose_coeff_lookups = T.recompose_coeff ? 1u : 0u;
  if (T.counts.n_p2w) ld.p2w_prep = T.p2w_prep.data();
  prof_stage(ctx, "prep_layer_create");
  C->layer = layer_create<PP>(ctx, &ld, commit_out);

  // execution schedule
  prof_stage(ctx, "prep_build_schedule");  // = what is left of it when the commitment is done
  C->sched = sched_job.get();
  prof_stage(ctx, "prep_upload_schedule");
  RunSchedule& S = C->sched;
  // constants travel in Montgomery form; hint output lists keep their flag bit
  std::vector<uint32_t> ext_m = S.dev_ext;
  std::vector<uint32_t> const_vals;
  for (auto& r : S.light)
    if ((r.kind_flags & 0xFF) == P3R_OP_CONST)
      for (uint32_t k = 0; k < ext_d; ++k) ext_m[r.ext_off + k] = Fp<PP>::from_canonical(S.dev_ext[r.ext_off + k]).v;
  for (auto& op : h.ops)
    if (op.kind == P3R_OP_CONST)
      for (uint32_t k = 0; k < ext_d; ++k) const_vals.push_back(Fp<PP>::from_canonical(h.ext_of(op)[k]).v);
  auto up = [&](DevBuf& b, const void* src, size_t bytes) {
    b.alloc(std::max<size_t>((bytes + 3) / 4, 1));
    if (bytes) P3R_HIP(copy_sync(ctx->stream, b.p, src, bytes, hipMemcpyHostToDevice));
  };
  up(C->d_light, S.light.data(), S.light.size() * sizeof(RunOp));
  if (ext_d == 4) up(C->d_p2, S.p2.data(), S.p2.size() * sizeof(RunP2));
  else up(C->d_p2, S.p2b.data(), S.p2b.size() * sizeof(RunP2B));
  up(C->d_ext, ext_m.data(), ext_m.size() * 4);
  up(C->d_const_values, const_vals.data(), const_vals.size() * 4);
  up(C->d_public_rows, h.public_rows.data(), h.public_rows.size() * 4);
  up(C->d_private_rows, h.private_rows.data(), h.private_rows.size() * 4);
  up(C->d_public_out, S.public_out.data(), S.public_out.size() * 4);
  up(C->d_rewrite, S.rewrite_pairs.data(), S.rewrite_pairs.size() * 4);
  up(C->d_light_off, S.light_off.data(), S.light_off.size() * 4);
  up(C->d_p2seg_off, S.p2seg_off.data(), S.p2seg_off.size() * 4);
  up(C->d_p2segs, S.p2segs.data(), S.p2segs.size() * sizeof(RunSchedule::P2Seg));
  up(C->d_chunk_bounds, S.chunk_bounds.data(), S.chunk_bounds.size() * 4);
  up(C->d_chain_ops, S.chain_ops.data(), S.chain_ops.size() * sizeof(RunOp));
  up(C->d_chains, S.chains.data(), S.chains.size() * sizeof(RunSchedule::ChainSeg));
  if (!S.p2wseg_off.empty()) {
    up(C->d_p2w, S.p2w.data(), S.p2w.size() * sizeof(RunP2W));
    up(C->d_p2wsegs, S.p2wsegs.data(), S.p2wsegs.size() * sizeof(RunSchedule::P2Seg));
    up(C->d_p2wseg_off, S.p2wseg_off.data(), S.p2wseg_off.size() * 4);
    std::vector<uint32_t> ids(S.p2w_row_of_op_id);
    for (auto& v : ids)
      if (v != kNoW && S.p2w_row_merkle[v]) v |= 1u << 31;
    up(C->d_roww_of_op_id, ids.data(), ids.size() * 4);
  }
  {
    std::vector<uint32_t> ids(S.p2_row_of_op_id);
    for (auto& v : ids)
      if (v != kNoW && S.p2_row_merkle[v]) v |= 1u << 31;
    C->n_op_ids = ids.size();
    up(C->d_row_of_op_id, ids.data(), ids.size() * 4);
  }
  C->n_rewrite = S.rewrite_pairs.size() / 3;
  // the large host arrays are on the device now
  S.light = {}; S.p2 = {}; S.p2b = {}; S.chain_ops = {}; S.dev_ext = {}; S.p2_row_of_op_id = {}; S.p2_row_merkle = {};
  S.p2w = {}; S.p2w_row_of_op_id = {}; S.p2w_row_merkle = {};
  prof_stage(ctx, nullptr);
  return C;
}

inline const char* run_error_text(uint32_t code) {
  switch (code) {
    case RUN_ERR_CONFLICT: return "WitnessConflict";
    case RUN_ERR_DIV0: return "DivisionByZero";
    case RUN_ERR_MMCS_BIT: return "IncorrectNonPrimitiveOpPrivateData: expected boolean mmcs_bit (0 or 1)";
    case RUN_ERR_INDEX_SUM: return "IncorrectNonPrimitiveOpPrivateData: expected base field mmcs_index_sum";
    default: return "unknown";
  }
}

// set_private_data (runner.rs:124-176) against the device-resident NonPrimitiveOpId -> row table: pass 1 claims the
// row for the smallest position k naming it, pass 2 reports, per position, what the sequential loop would have
// said; the smallest failing position wins.
__global__ void __launch_bounds__(kBlock)
k_pd_claim(const uint32_t* __restrict__ ids, size_t n, const uint32_t* __restrict__ row_of_id, size_t n_ids, uint32_t* __restrict__ slot) {
  const size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const uint32_t id = ids[k];
  if (id >= n_ids) return;
  const uint32_t v = row_of_id[id];
  if (v == kNoW) return;
  atomicMin(&slot[v & 0x7FFFFFFFu], (uint32_t)k);
}
__global__ void __launch_bounds__(kBlock)
k_pd_check(const uint32_t* __restrict__ ids, size_t n, const uint32_t* __restrict__ row_of_id, size_t n_ids,
           const uint32_t* __restrict__ slot, uint32_t* __restrict__ err) {
  const size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const uint32_t id = ids[k];
  uint32_t code = 0;
  if (id >= n_ids || row_of_id[id] == kNoW) code = 1;          // NonPrimitiveOpIdOutOfRange
  else if (slot[row_of_id[id] & 0x7FFFFFFFu] != k) code = 2;   // private data already set
  else if (!(row_of_id[id] >> 31)) code = 3;                   // not a Merkle row
  if (code) atomicMin(err, ((uint32_t)k << 2) | code);
}

// set_public_inputs / set_private_inputs / set_private_data (runner.rs:83-176): validate, upload.
template <class PP>
std::unique_ptr<p3r_dinputs> circuit_inputs_upload(p3r_ctx* ctx, const p3r_circuit* C, const p3r_circuit_inputs* in) {
  if (C->n_public_rows && !in->public_values) fail(P3R_EINVAL, "PublicInputLengthMismatch: public_values is NULL");
  if (C->n_private_rows && !in->private_values) fail(P3R_EINVAL, "PrivateInputLengthMismatch: private_values is NULL");
  const size_t n_p2 = C->counts.n_p2, n_pd = in->n_private_data;
  if (n_pd && (!in->private_data_op_ids || !in->private_data_siblings)) fail(P3R_EINVAL, "private data arrays are NULL");
  auto D = std::make_unique<p3r_dinputs>();
  D->pub = upload_mont<PP>(ctx, in->public_values, C->n_public_rows * ctx->cfg.ext_degree, "public_values");
  D->priv = upload_mont<PP>(ctx, in->private_values, C->n_private_rows * ctx->cfg.ext_degree, "private_values");
  // the two widths are two op types with their own row tables: the same claim / check passes per list
  auto private_data = [&](size_t n_pd, const uint32_t* ids_host, const uint32_t* sib_host, size_t per_op, size_t n_rows,
                          const DevBuf& row_of_id, const DevBuf& other_row_of_id, DevBuf& sib, DevBuf& slot, const char* what) {
    sib = upload_mont<PP>(ctx, sib_host, n_pd * per_op, what);
    slot.alloc(std::max<size_t>(n_rows, 1));
    P3R_HIP(fill_async(ctx->stream, slot.p, 0xFF, std::max<size_t>(n_rows, 1) * 4));  // -1: no private data
    if (!n_pd) return;
    if (!row_of_id.p)   // the circuit has no permutation of this width at all
      fail(P3R_EINVAL, "NonPrimitiveOpIdOutOfRange { op_id: %u, max_ops: %zu } (%s)", ids_host[0], C->n_op_ids, what);
    DevBuf ids(n_pd), err(1);
    P3R_HIP(hipMemcpyAsync(ids.p, ids_host, n_pd * 4, hipMemcpyHostToDevice, ctx->stream));
    P3R_HIP(fill_async(ctx->stream, err.p, 0xFF, 4));
    hipLaunchKernelGGL(k_pd_claim, dim3(blocks_for(n_pd)), dim3(kBlock), 0, ctx->stream, ids.p, n_pd, row_of_id.p, C->n_op_ids, slot.p);
    hipLaunchKernelGGL(k_pd_check, dim3(blocks_for(n_pd)), dim3(kBlock), 0, ctx->stream, ids.p, n_pd, row_of_id.p, C->n_op_ids, slot.p, err.p);
    uint32_t e = 0;
    P3R_HIP(copy_sync(ctx->stream, &e, err.p, 4, hipMemcpyDeviceToHost));
    if (e != 0xFFFFFFFFu) {
      const uint32_t id = ids_host[e >> 2];
      switch (e & 3) {
        case 1: {
          uint32_t other = kNoW;
          if (other_row_of_id.p && id < C->n_op_ids) P3R_HIP(copy_sync(ctx->stream, &other, other_row_of_id.p + id, 4, hipMemcpyDeviceToHost));
          if (other != kNoW)
            fail(P3R_EINVAL, "IncorrectNonPrimitiveOpPrivateData: NonPrimitiveOpId(%u) is a permutation of the other width (%s)", id, what);
          fail(P3R_EINVAL, "NonPrimitiveOpIdOutOfRange { op_id: %u, max_ops: %zu }", id, C->n_op_ids);
        }
        case 2: fail(P3R_EINVAL, "IncorrectNonPrimitiveOpPrivateData: private data already set for NonPrimitiveOpId(%u)", id);
        default: fail(P3R_EINVAL, "IncorrectNonPrimitiveOpPrivateData: private data provided for non-Merkle operation NonPrimitiveOpId(%u)", id);
      }
    }
  };
  private_data(n_pd, in->private_data_op_ids, in->private_data_siblings, 8, n_p2, C->d_row_of_op_id, C->d_roww_of_op_id, D->sib, D->slot,
               "private_data_siblings");
  const size_t n_pdw = in->n_private_data_w32;
  if (n_pdw && (!in->private_data_w32_op_ids || !in->private_data_w32_siblings)) fail(P3R_EINVAL, "width-32 private data arrays are NULL");
  private_data(n_pdw, in->private_data_w32_op_ids, in->private_data_w32_siblings, 24, C->counts.n_p2w, C->d_roww_of_op_id, C->d_row_of_op_id,
               D->sib_w, D->slot_w, "private_data_w32_siblings");
  return D;
}

// Decodes the error word of a run (first failing op in circuit order, or all ones).
inline void run_raise_error(uint32_t e) {
  if (e != 0xFFFFFFFFu) {
    const uint32_t idx = e >> 3;
    if (idx == 0x1FFFFFFFu) fail(P3R_EINVAL, "WitnessConflict while applying witness_rewrite");
    fail(P3R_EINVAL, "%s at op %u", run_error_text(e & 7), idx);
  }
}

// CircuitRunner::run: returns the Traces (HBM-resident) of one execution.
// `deferred_err`: when given, the run is only enqueued; its error word travels to that pinned host
// word behind it on the stream, and the caller passes it to run_raise_error after its next
// synchronisation, before trusting anything derived from the traces.
template <class PP>
std::unique_ptr<p3r_dtraces> circuit_run(p3r_ctx* ctx, const p3r_circuit* C, const p3r_dinputs* in,
                                         uint32_t* deferred_err = nullptr) {
  const RunSchedule& S = C->sched;
  const p3r_layer* L = C->layer.get();
  if (!S.deferred_error.empty()) fail(P3R_EINVAL, "%s", S.deferred_error.c_str());
  const size_t n_p2 = C->counts.n_p2;
  prof_stage(ctx, "run_circuit");
  auto T = std::make_unique<p3r_dtraces>();
  const auto& cn = C->counts;
  T->n_const = cn.n_const; T->n_public = cn.n_public; T->n_alu = cn.n_alu; T->n_recompose = cn.n_recompose;
  const uint32_t ext_d = ctx->cfg.ext_degree;
  DevBuf w((size_t)std::max<uint32_t>(C->witness_count, 1) * ext_d);
  DevBuf err(1), p2_out(std::max<size_t>(n_p2, 1) * 16);
  const DevBuf &d_pub = in->pub, &d_priv = in->priv, &d_sib = in->sib, &d_slot = in->slot;
  P3R_HIP(fill_async(ctx->stream, err.p, 0xFF, 4));
  if (C->n_public_rows)
    hipLaunchKernelGGL(k_run_scatter<PP>, dim3(blocks_for(C->n_public_rows * ext_d)), dim3(kBlock), 0, ctx->stream,
                       C->d_public_rows.p, d_pub.p, C->n_public_rows, w.p, ext_d);
  if (C->n_private_rows)
    hipLaunchKernelGGL(k_run_scatter<PP>, dim3(blocks_for(C->n_private_rows * ext_d)), dim3(kBlock), 0, ctx->stream,
                       C->d_private_rows.p, d_priv.p, C->n_private_rows, w.p, ext_d);
  // trace buffers
  T->const_values.alloc(std::max<size_t>(cn.n_const * ext_d, 1));
  if (cn.n_const)
    P3R_HIP(copy_async_kernel(ctx->stream, T->const_values.p, C->d_const_values.p, cn.n_const * ext_d * 4));
  T->public_values.alloc(std::max<size_t>(cn.n_public * ext_d, 1));
  T->alu_values.alloc(std::max<size_t>(cn.n_alu * 4 * ext_d, 1));
  if (S.n_alu_records == 0)  // the dummy op of an empty table; otherwise every record is written by its op
    P3R_HIP(fill_async(ctx->stream, T->alu_values.p, 0, T->alu_values.n * 4));
  T->recompose_values.alloc(std::max<size_t>((cn.n_recompose + cn.n_recompose_coeff) * ext_d, 1));
  T->n_recompose_coeff = cn.n_recompose_coeff;
  uint32_t* p2_inputs = nullptr; uint8_t* p2_flags = nullptr; uint32_t* p2_seed = nullptr;
  size_t p2_h = 0;
  if (L->has_p2) {
    // padded with fillers: new_start = 1, zero state (poseidon2.rs:1125-1140)
    p2_h = L->h_p2;
    auto d = std::make_unique<p3r_p2_dev>();
    d->n = p2_h;
    d->inputs = dmat_alloc(p2_h, P2_WIDTH);
    P3R_HIP(fill_async(ctx->stream, d->inputs->d, 0, p2_h * P2_WIDTH * 4));
    d->flags.alloc((3 * p2_h + 3) / 4 + 1);
    p2_flags = reinterpret_cast<uint8_t*>(d->flags.p);
    P3R_HIP(fill_async(ctx->stream, p2_flags, 1, p2_h));
    P3R_HIP(fill_async(ctx->stream, p2_flags + p2_h, 0, 2 * p2_h));
    d->seed.alloc(p2_h);
    P3R_HIP(fill_async(ctx->stream, d->seed.p, 0, p2_h * 4));
    p2_inputs = d->inputs->d; p2_seed = d->seed.p;
    T->p2 = std::move(d);
  }
  // rows of the width-32 table: the same fillers
  uint32_t* p2w_inputs = nullptr; uint8_t* p2w_flags = nullptr;
  size_t p2w_h = 0;
  DevBuf p2w_out(std::max<size_t>(cn.n_p2w, 1) * 32);
  if (L->has_p2w) {
    p2w_h = L->h_p2w;
    auto d = std::make_unique<p3r_p2_dev>();
    d->n = p2w_h;
    d->inputs = dmat_alloc(p2w_h, P2W_WIDTH);
    P3R_HIP(fill_async(ctx->stream, d->inputs->d, 0, p2w_h * P2W_WIDTH * 4));
    d->flags.alloc(p2w_h + 1);
    p2w_flags = reinterpret_cast<uint8_t*>(d->flags.p);
    P3R_HIP(fill_async(ctx->stream, p2w_flags, 1, p2w_h));
    P3R_HIP(fill_async(ctx->stream, p2w_flags + p2w_h, 0, 3 * p2w_h));
    d->seed.alloc(p2w_h);
    P3R_HIP(fill_async(ctx->stream, d->seed.p, 0, p2w_h * 4));   // no accumulator witness on this table
    p2w_inputs = d->inputs->d;
    T->p2w = std::move(d);
  }
  {
    ProfScope ps(ctx, "run_levels");
    RunArgs A{};
    A.light = reinterpret_cast<const RunOp*>(C->d_light.p);
    A.p2 = reinterpret_cast<const RunP2*>(C->d_p2.p);
    A.p2b = reinterpret_cast<const RunP2B*>(C->d_p2.p);
    A.w = w.p; A.ext = C->d_ext.p; A.alu_values = T->alu_values.p; A.rec_values = T->recompose_values.p;
    A.p2_inputs = p2_inputs; A.p2_h = p2_h; A.p2_flags = p2_flags; A.p2_seed = p2_seed; A.p2_out = p2_out.p;
    A.pd_slot = reinterpret_cast<const int32_t*>(d_slot.p); A.siblings = d_sib.p;
    A.rc = ctx->rc.p; A.diag = ctx->p2_diag.p; A.err = err.p;
    A.light_off = C->d_light_off.p; A.p2seg_off = C->d_p2seg_off.p;
    A.p2segs = reinterpret_cast<const RunSchedule::P2Seg*>(C->d_p2segs.p);
    const bool has_w32 = !S.p2wseg_off.empty();
    if (has_w32) {
      A.p2w = reinterpret_cast<const RunP2W*>(C->d_p2w.p);
      A.p2w_inputs = p2w_inputs; A.p2w_h = p2w_h; A.p2w_flags = p2w_flags; A.p2w_out = p2w_out.p;
      A.pdw_slot = reinterpret_cast<const int32_t*>(in->slot_w.p); A.siblings_w = in->sib_w.p;
      A.rcw = ctx->rc.p + p2_num_constants<PP>();
      A.p2wsegs = reinterpret_cast<const RunSchedule::P2Seg*>(C->d_p2wsegs.p);
      A.p2wseg_off = C->d_p2wseg_off.p;
    }
    dispatch_ext_degree<PP>((int)ext_d, [&](auto dc) {
    constexpr int DD = decltype(dc)::value;
    for (const auto& seg : S.segments) {
      if (seg.narrow) {
        hipLaunchKernelGGL((k_run_levels_narrow<PP, DD>), dim3(1), dim3(kNarrowBlock), 0, ctx->stream, A,
                           C->d_chunk_bounds.p + seg.chunk_begin, seg.n_chunks);
        continue;
      }
      const uint32_t l = seg.l0;
      const uint32_t nl = S.light_off[l + 1] - S.light_off[l], np = S.p2seg_off[l + 1] - S.p2seg_off[l];
      const uint32_t lb = (nl + kBlock - 1) / kBlock, pb = (np * 16 + kBlock - 1) / kBlock;
      const uint32_t nc = S.chain_off[l + 1] - S.chain_off[l], n_long = S.chain_long[l];
      const RunOp* steps = reinterpret_cast<const RunOp*>(C->d_chain_ops.p);
      const RunSchedule::ChainSeg* segs = reinterpret_cast<const RunSchedule::ChainSeg*>(C->d_chains.p) + S.chain_off[l];
      const uint32_t n_short = nc - n_long, cb = (n_short + kBlock / 64 - 1) / (kBlock / 64);
      const uint32_t npw = has_w32 ? S.p2wseg_off[l + 1] - S.p2wseg_off[l] : 0u, pwb = (npw * 32 + kBlock - 1) / kBlock;
      // chains only read operands of lower levels, so they share the launch with the level's other ops
      if (lb + pb + cb + pwb)
        hipLaunchKernelGGL((k_run_level<PP, DD>), dim3(pb + cb + lb + pwb), dim3(kBlock), 0, ctx->stream, A, S.p2seg_off[l], np, pb,
                           steps, segs + n_long, n_short, cb, S.light_off[l], nl, has_w32 ? S.p2wseg_off[l] : 0u, npw, pwb);
      if (n_long)
        hipLaunchKernelGGL((k_run_chains_block<PP, DD>), dim3(n_long), dim3(kLongChainBlock), 0, ctx->stream, A, steps, segs);
    }
    if (C->n_rewrite)
      hipLaunchKernelGGL((k_run_rewrite<PP, DD>), dim3(blocks_for(C->n_rewrite)), dim3(kBlock), 0, ctx->stream,
                         C->d_rewrite.p, C->n_rewrite, w.p, err.p);
    });
    if (cn.n_public)
      hipLaunchKernelGGL(k_run_gather<PP>, dim3(blocks_for(cn.n_public * ext_d)), dim3(kBlock), 0, ctx->stream,
                         C->d_public_out.p, cn.n_public, w.p, T->public_values.p, ext_d);
    P3R_HIP(hipGetLastError());
  }
  if (deferred_err) {
    P3R_HIP(hipMemcpyAsync(deferred_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  } else {
    uint32_t e = 0;
    P3R_HIP(copy_sync(ctx->stream, &e, err.p, 4, hipMemcpyDeviceToHost));
    run_raise_error(e);
  }
  return T;
}

template <class PP>
void dtraces_get(p3r_ctx* ctx, const p3r_layer* L, const p3r_dtraces* t, uint32_t which, uint32_t* out, size_t out_len) {
  const auto& c = L->counts;
  auto plain_at = [&](const uint32_t* src, size_t n) {
    if (out_len != n) fail(P3R_EBUFFER, "array holds %zu values, caller asked for %zu", n, out_len);
    if (!n) return;
    DevBuf tmp(n);
    P3R_HIP(hipMemcpyAsync(tmp.p, src, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_convert_inplace<PP>, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, tmp.p, n, 0);
    P3R_HIP(copy_sync(ctx->stream, out, tmp.p, n * 4, hipMemcpyDeviceToHost));
  };
  switch (which) {
    case P3R_TRACES_CONST_VALUES: plain_at(t->const_values.p, c.n_const * ctx->cfg.ext_degree); break;
    case P3R_TRACES_PUBLIC_VALUES: plain_at(t->public_values.p, c.n_public * ctx->cfg.ext_degree); break;
    case P3R_TRACES_ALU_VALUES: plain_at(t->alu_values.p, c.n_alu * 4 * ctx->cfg.ext_degree); break;
    case P3R_TRACES_RECOMPOSE_VALUES: plain_at(t->recompose_values.p, c.n_recompose * ctx->cfg.ext_degree); break;
    case P3R_TRACES_RECOMPOSE_COEFF_VALUES:   // the rows of the second table follow those of the first
      plain_at(t->recompose_values.p + c.n_recompose * ctx->cfg.ext_degree, c.n_recompose_coeff * ctx->cfg.ext_degree);
      break;
    case P3R_TRACES_P2_INPUT_VALUES: {
      if (out_len != c.n_p2 * 16) fail(P3R_EBUFFER, "array holds %zu values, caller asked for %zu", c.n_p2 * 16, out_len);
      if (!c.n_p2) break;
      std::vector<uint32_t> full(t->p2->n * 16);
      download<PP>(ctx, t->p2->inputs.get(), full.data());  // row-major [h][16], canonical
      std::copy(full.begin(), full.begin() + c.n_p2 * 16, out);
      break;
    }
    case P3R_TRACES_P2_FLAGS: {
      if (out_len != c.n_p2 * 3) fail(P3R_EBUFFER, "array holds %zu values, caller asked for %zu", c.n_p2 * 3, out_len);
      if (!c.n_p2) break;
      const size_t h = t->p2->n;
      std::vector<uint8_t> f(3 * h);
      P3R_HIP(copy_sync(ctx->stream, f.data(), t->p2->flags.p, 3 * h, hipMemcpyDeviceToHost));
      for (size_t r = 0; r < c.n_p2; ++r)
        for (int k = 0; k < 3; ++k) out[r * 3 + k] = f[k * h + r];
      break;
    }
    case P3R_TRACES_P2_MMCS_INDEX_SUM: {
      if (out_len != c.n_p2) fail(P3R_EBUFFER, "array holds %zu values, caller asked for %zu", c.n_p2, out_len);
      if (!c.n_p2) break;
      plain_at(t->p2->seed.p, c.n_p2);
      break;
    }
    case P3R_TRACES_P2W_INPUT_VALUES: {
      if (out_len != c.n_p2w * 32) fail(P3R_EBUFFER, "array holds %zu values, caller asked for %zu", c.n_p2w * 32, out_len);
      if (!c.n_p2w) break;
      std::vector<uint32_t> full(t->p2w->n * 32);
      download<PP>(ctx, t->p2w->inputs.get(), full.data());  // row-major [h][32], canonical
      std::copy(full.begin(), full.begin() + c.n_p2w * 32, out);
      break;
    }
    case P3R_TRACES_P2W_FLAGS: {
      if (out_len != c.n_p2w * 4) fail(P3R_EBUFFER, "array holds %zu values, caller asked for %zu", c.n_p2w * 4, out_len);
      if (!c.n_p2w) break;
      const size_t h = t->p2w->n;
      std::vector<uint8_t> f(4 * h);
      P3R_HIP(copy_sync(ctx->stream, f.data(), t->p2w->flags.p, 4 * h, hipMemcpyDeviceToHost));
      for (size_t r = 0; r < c.n_p2w; ++r)
        for (int k = 0; k < 4; ++k) out[r * 4 + k] = f[k * h + r];
      break;
    }
    case P3R_TRACES_P2W_MMCS_INDEX_SUM: {
      if (out_len != c.n_p2w) fail(P3R_EBUFFER, "array holds %zu values, caller asked for %zu", c.n_p2w, out_len);
      if (!c.n_p2w) break;
      plain_at(t->p2w->seed.p, c.n_p2w);
      break;
    }
    default: fail(P3R_EINVAL, "unknown traces array %u", which);
  }
}

}  // namespace
