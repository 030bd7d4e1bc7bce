// Structures of the prepared circuit shared by the host-side preparation (circuit_impl.hip.h, layer_impl.hip.h)
// and the device-side one (prep_device.hip): the execution schedule of the verifier circuit
// (CircuitRunner::run, circuit/src/tables/runner.rs:195-253, as a static levelised plan) and the ALU lane
// schedule (AluAir::compute_schedule, circuit-prover/src/air/alu_air.rs:349-463, as a scatter plan).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/p3r.h"

namespace p3r {

constexpr uint32_t kNoW = P3R_NO_WITNESS;

// Preprocessed row of the compact D = 1 Poseidon2 table (poseidon2-circuit-air/src/air.rs:730-763): 8 in_ctl, the
// length tag, cap_chain_enable, 8 + 8 chain selectors | at kP2D1Hdr: 16 input indices, 8 output indices, 8 out_ctl |
// at kP2D1Tail: mmcs_index_sum index, mmcs_ctl_enabled, new_start, merkle_path
constexpr int kP2D1Hdr = 26, kP2D1Tail = 58, kP2D1PrepWidth = 62;

// ext layout of a width-32 permutation op: [in0..in7, mmcs_index_sum, mmcs_bit, mmcs_bit2, n_out, out0..]
constexpr uint32_t kW32In = 8, kW32Rate = 6, kW32IdxSlot = 8, kW32BitSlot = 9, kW32Bit2Slot = 10, kW32NOutSlot = 11, kW32Hdr = 12;
// preprocessed row of that table: Poseidon2PreprocessedRow<8, 6> (= air_device.hip.h::kP2WPrepWidth, tail at 44)
constexpr int kP2WPrepCols = 48;

// ALU plan entry kinds
enum { PLAN_SEP = 0, PLAN_OP = 1, PLAN_PACKED = 2 };

struct AluPlanEntry {
  uint32_t first;  // op index
  uint8_t kind, k;
  uint16_t pad;
};

enum : uint32_t {
  RUN_BACKWARD = 1u << 8,    // Add / Mul solving for b (runner.rs:341-385)
  RUN_CHECK_OUT = 1u << 9,   // `out` already holds a value: compare instead of write (set_witness, :473-510)
  RUN_CHECK_AUX = 1u << 10,  // same for MulAdd's intermediate_out
  RUN_CHECK_BIT = 1u << 31,  // on a hint-output entry of the device ext array
};

struct RunOp {  // ALU / hint / recompose / const-check ops, one lane each
  uint32_t kind_flags;  // bits 0-7 p3r_op_kind, 8-10 RUN_*, 16-23 ext_len
  uint32_t a, b, c, out, aux;
  uint32_t rec;      // ALU record / recompose row this op fills
  uint32_t ext_off;  // into the device ext array
  uint32_t op_idx;   // position in the circuit (error reports)
  uint32_t pad;
};

struct RunP2 {  // one Poseidon2 permutation, sixteen lanes
  uint32_t in[4], idx_w, bit_w, out[4];
  uint32_t flags;  // bit 0 new_start, 1 merkle_path, 4-7 output is a check, 8-10 number of outputs
  uint32_t row, prev_row, op_idx;
};

// The same for a circuit of extension degree 1 / 5: a base-mode permutation (Poseidon2Config::*_D1_W16) has one
// witness per state element - sixteen input slots, eight or sixteen outputs - and a sponge length tag
// (circuit/src/ops/poseidon_perm/executor.rs:600-700).
struct RunP2B {
  uint32_t in[16], out[16], idx_w, bit_w;
  uint32_t flags;       // bit 0 new_start, 1 merkle_path, 8-12 number of outputs
  uint32_t check_mask;  // bit l: output l lands on a witness that already holds a value (compare)
  uint32_t row, prev_row, op_idx, absorb_len;
};

// A permutation of the WIDTH-32 table in a D = 4 circuit (P3R_OP_POSEIDON2_W32_PERM: the arity-4 compression shape,
// poseidon_perm/executor.rs:92-235): eight input limbs, two direction bits, up to eight output limbs; thirty-two lanes.
struct RunP2W {
  uint32_t in[8], bit_w, bit2_w, out[8];
  uint32_t flags;       // bit 0 new_start, 1 merkle_path, 8-11 number of outputs, 16-23 output l is a check
  uint32_t row, prev_row, op_idx;
  uint32_t prev_in_seg; // 1: prev_row is the row just before this one in its segment (the state is still in registers)
};

enum : uint32_t { RUN_ERR_CONFLICT = 1, RUN_ERR_DIV0 = 2, RUN_ERR_MMCS_BIT = 3, RUN_ERR_INDEX_SUM = 4 };

struct RunSchedule {
  uint32_t n_alu_records = 0;  // AluOpRecords the run writes (0: the ALU table holds its dummy op only)
  std::vector<RunOp> light;
  // Poseidon2 permutations: a run of rows chained through the sponge / Merkle state whose witness
  // inputs are all ready when the run starts is ONE segment, executed row after row by one 16-lane
  // group with the state kept in registers (no launch, barrier or memory round trip per row)
  struct P2Seg { uint32_t first, n; };
  std::vector<RunP2> p2;                    // rows, segment by segment
  std::vector<RunP2B> p2b;                  // the same for base-mode rows (circuits of degree 1 / 5): one of the two is empty
  std::vector<P2Seg> p2segs;                // sorted by level
  std::vector<uint32_t> light_off, p2seg_off;  // per level, size levels + 1
  // rows of the width-32 table (D = 4 circuits that hold P3R_OP_POSEIDON2_W32_PERM ops): same segment rule - a row
  // joins the segment whose last row is its predecessor when its witnesses are ready in time
  std::vector<RunP2W> p2w;
  std::vector<P2Seg> p2wsegs;
  std::vector<uint32_t> p2wseg_off;         // per level, size levels + 1 (empty: no width-32 rows)
  std::vector<uint32_t> p2w_row_of_op_id;   // NonPrimitiveOpId -> width-32 row (or kNoW)
  std::vector<uint8_t> p2w_row_merkle;
  std::vector<uint32_t> dev_ext;
  std::vector<uint32_t> const_rows;         // const op -> witness, in table order (static Const trace)
  std::vector<uint32_t> public_out;         // public table row -> witness
  std::vector<uint32_t> rewrite_pairs;      // (dst, src, check) triples applied after the last level
  std::vector<uint32_t> p2_row_of_op_id;    // NonPrimitiveOpId -> Poseidon2 row (or kNoW)
  std::vector<uint8_t> p2_row_merkle;
  std::string deferred_error;               // what run() reports for a circuit that cannot complete
  size_t levels = 0;
  // launches: a wide level each, or a run of consecutive narrow levels [l0, l1) in one workgroup
  struct Segment { uint32_t l0, l1; bool narrow; uint32_t chunk_begin, n_chunks; };
  std::vector<Segment> segments;
  std::vector<uint32_t> chunk_bounds;       // per narrow segment: n_chunks + 1 level boundaries
  // Horner chains: runs of consecutive HornerAcc ops threaded through the accumulator with one
  // shared multiplier b are an affine recurrence acc <- acc*b + (c - a); each run is ONE scan
  // (run_chains) at one level instead of one level per step.
  struct ChainSeg { uint32_t first, n, acc_w, b_w; };
  std::vector<RunOp> chain_ops;             // steps of all chains, chain by chain
  std::vector<ChainSeg> chains;             // sorted by level; within a level the long ones first
  std::vector<uint32_t> chain_off;          // per level
  std::vector<uint32_t> chain_long;         // per level: how many of its chains get a whole workgroup
};

// Launch plan from the per-level counts: a wide level each, or a run of consecutive narrow levels in one
// workgroup (k_run_levels_narrow), chunked so that the light-op records of a chunk fit its LDS staging buffer.
inline void finish_segments(RunSchedule& S) {
  for (uint32_t l = 1; l <= S.levels; ++l) {
    const uint32_t nl = S.light_off[l + 1] - S.light_off[l], np = S.p2seg_off[l + 1] - S.p2seg_off[l];
    const uint32_t nc = S.chain_off[l + 1] - S.chain_off[l];
    const uint32_t npw = S.p2wseg_off.empty() ? 0u : S.p2wseg_off[l + 1] - S.p2wseg_off[l];
    if (!nl && !np && !nc && !npw) continue;
    const bool narrow = nl <= 1024 && np <= 64 && npw <= 32 && !nc;
    if (narrow && !S.segments.empty() && S.segments.back().narrow && S.segments.back().l1 == l) S.segments.back().l1 = l + 1;
    else S.segments.push_back({l, l + 1, narrow, 0, 0});
  }
  for (auto& seg : S.segments) {
    if (!seg.narrow) continue;
    seg.chunk_begin = (uint32_t)S.chunk_bounds.size();
    uint32_t start = seg.l0;
    S.chunk_bounds.push_back(start);
    for (uint32_t l = seg.l0; l < seg.l1; ++l) {
      // light-op records of levels [start, l + 1) must fit the LDS staging buffer (kNarrowLightCap)
      if (S.light_off[l + 1] - S.light_off[start] > 1400) {
        S.chunk_bounds.push_back(l);
        start = l;
        seg.n_chunks++;
      }
    }
    S.chunk_bounds.push_back(seg.l1);
    seg.n_chunks++;
  }
}

}  // namespace p3r
