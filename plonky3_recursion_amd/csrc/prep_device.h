// Device-side preparation of a verifier circuit: what `prove_next_layer` pays with prep = None
// (recursion/src/recursion.rs:452-501) before the proof itself -
//   Circuit::generate_preprocessed_columns::<D>      circuit/src/circuit.rs:237-510   (D = 1, 4, 5: the context's ext_degree)
//   get_airs_and_degrees_with_prep                   circuit-prover/src/common.rs:127-390
//   poseidon_preprocess_for_prover                   circuit-prover/src/batch_stark_prover.rs:97-246
//   AluAir::compute_schedule + build_scheduled_preprocessed_trace
//                                                    circuit-prover/src/air/alu_air.rs:349-463,613-677
// and the static execution schedule of the device CircuitRunner (circuit_impl.hip.h).  The host restatement of
// the same steps (circuit_impl.hip.h::circuit_tables / build_schedule, layer_impl.hip.h::layer_create) walks the
// 4.8 M ops of a 2^20-row layer on one or two host threads (400 ms); here every step is a map, a scan, a
// histogram or a short fixed-point iteration over the op list in HBM, and the preprocessed traces are
// written where their LDE and commitment read them - nothing but the op list crosses PCIe.
// The host path stays: it is what raises the reference's errors (a circuit the device pass flags is handed to
// it), and P3R_PREP_HOST=1 selects it for A/B runs and the equality tests.
#pragma once
#include "context.h"
#include "run_schedule.h"

namespace p3r {

struct DevPrep {
  // ---- CircuitProverData inputs: per-table preprocessed traces (column-major, Montgomery, padded)
  p3r_layer_desc_counts counts{};
  uint32_t public_lanes = 1, alu_lanes = 1;  // effective (reduce_lanes_if_dummy, batch_stark_prover.rs:1305-1318)
  size_t h[7] = {0, 0, 0, 0, 0, 0, 0};       // padded heights, 0 = table absent; [5] = the second Recompose table, [6] = width-32 Poseidon2
  size_t alu_rows = 0;
  std::unique_ptr<p3r_dmat> prep[7];
  bool recompose_coeff = false;              // slot 4 holds the `recompose/coeff` kind (the circuit has no plain Recompose op)
  DevBuf alu_plan, alu_prev_src;
  // ---- execution schedule: the large arrays stay on the device, `sched` carries the per-level offsets,
  // the launch plan and the counts the runner needs on the host
  RunSchedule sched;
  DevBuf d_light, d_p2, d_ext, d_const_values, d_public_rows, d_private_rows, d_public_out, d_rewrite;
  DevBuf d_light_off, d_p2seg_off, d_p2segs, d_chunk_bounds, d_chain_ops, d_chains;
  DevBuf d_row_of_op_id;  // NonPrimitiveOpId -> Poseidon2 row, bit 31 = Merkle row; kNoW: not a permutation
  DevBuf d_p2w, d_p2wsegs, d_p2wseg_off, d_roww_of_op_id;  // the same for the rows of the width-32 table (empty: the circuit has none)
  size_t n_op_ids = 0, n_public_rows = 0, n_private_rows = 0, n_rewrite = 0;
};

// false: the device pass met something the host path must report (a malformed circuit, an unclaimed private
// input, a witness that is never set ...): `out` is unusable, the caller runs the host preparation, which raises.
bool devprep_circuit(p3r_ctx* ctx, const p3r_circuit_desc* d, DevPrep& out);

}  // namespace p3r
