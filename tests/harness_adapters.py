"""Adapters from the synthetic-workload arrays (harness/synth.cpp) to the prover's input types."""
import numpy as np

from plonky3_recursion_amd.prover import Circuit, CircuitInputs, CircuitPrep, Traces


def traces_from_arrays(a, ext_degree=4) -> Traces:
    fl = a["p2_flags"].reshape(-1, 4)
    d = ext_degree
    return Traces(
        const_values=a["const_values"].reshape(-1, d),
        public_values=a["public_values"].reshape(-1, d),
        alu_values=a["alu_values"].reshape(-1, 4 * d),
        p2_input_values=a["p2_inputs"].reshape(-1, 16),
        p2_new_start=fl[:, 0].astype(np.uint8),
        p2_merkle_path=fl[:, 1].astype(np.uint8),
        p2_mmcs_bit=fl[:, 2].astype(np.uint8),
        p2_mmcs_index_sum=a["p2_mmcs_index_sum"],
        recompose_values=a["recompose_values"].reshape(-1, d),
        # RECOMPOSE_BOTH: the rows of the second Recompose table (`recompose/coeff`)
        recompose_coeff_values=a["recompose_coeff_values"].reshape(-1, d) if len(a.get("recompose_coeff_values", ())) else None,
        # P2_W32: the rows of the width-32 Poseidon2 table
        **(_p2w_traces(a) if len(a.get("p2w_inputs", ())) else {}),
    )


def _p2w_traces(a):
    fl = a["p2w_flags"].reshape(-1, 4)
    return dict(p2w_input_values=a["p2w_inputs"].reshape(-1, 32), p2w_new_start=fl[:, 0].astype(np.uint8),
                p2w_merkle_path=fl[:, 1].astype(np.uint8), p2w_mmcs_bit=fl[:, 2].astype(np.uint8),
                p2w_mmcs_bit2=fl[:, 3].astype(np.uint8), p2w_mmcs_index_sum=a["p2w_mmcs_index_sum"])


def circuit_prep_from_arrays(a, ext_degree=4, recompose_coeff_lookups=False) -> CircuitPrep:
    fl = a["p2_flags"].reshape(-1, 4)
    il, ol = (4, 2) if ext_degree == 4 else (16, 8)
    return CircuitPrep(
        const_prep=a["const_prep"].reshape(-1, 2),
        public_prep=a["public_prep"].reshape(-1, 2),
        alu_prep13=a["alu_prep13"].reshape(-1, 13),
        recompose_prep=a["recompose_prep"].reshape(-1, 2 + (2 * ext_degree if recompose_coeff_lookups else 0)),
        recompose_coeff_lookups=recompose_coeff_lookups,
        p2_new_start=fl[:, 0].astype(np.uint8),
        p2_merkle_path=fl[:, 1].astype(np.uint8),
        p2_mmcs_ctl_enabled=fl[:, 3].astype(np.uint8),
        p2_in_ctl=a["p2_in_ctl"].reshape(-1, il).astype(np.uint8),
        p2_input_indices=a["p2_input_indices"].reshape(-1, il),
        p2_out_ctl=a["p2_out_ctl"].reshape(-1, ol),
        p2_output_indices=a["p2_output_indices"].reshape(-1, ol),
        p2_mmcs_index_sum_idx=a["p2_mmcs_index_sum_idx"],
        p2_absorb_len=a["p2_absorb_len"].astype(np.uint8) if ext_degree != 4 and "p2_absorb_len" in a else None,
        recompose_coeff_prep=a["recompose_coeff_prep"].reshape(-1, 2 + 2 * ext_degree) if len(a.get("recompose_coeff_prep", ())) else None,
        p2w_prep=a["p2w_prep"].reshape(-1, 48) if len(a.get("p2w_prep", ())) else None,
    )


def circuit_from_arrays(a) -> Circuit:
    return Circuit(witness_count=int(a["counts"][5]), ops=a["ops"].reshape(-1, 8), ext=a["ext"],
                   public_rows=a["public_rows"], private_input_rows=a["private_rows"],
                   witness_rewrite=a["rewrite"].reshape(-1, 2))


def circuit_inputs_from_arrays(a, ext_degree=4) -> CircuitInputs:
    return CircuitInputs(public_values=a["in_public_values"].reshape(-1, ext_degree),
                         private_values=a["in_private_values"].reshape(-1, ext_degree),
                         private_data_op_ids=a["pd_op_ids"], private_data_siblings=a["pd_siblings"].reshape(-1, 8),
                         private_data_w32_op_ids=a.get("pdw_op_ids", np.zeros(0, np.uint32)),
                         private_data_w32_siblings=np.asarray(a.get("pdw_siblings", np.zeros(0, np.uint32))).reshape(-1, 24))


def split_aggregation_inputs(inputs: CircuitInputs):
    """A synthetic aggregation circuit verifies two proofs; its inputs split into the share of the left
    and of the right proof (first / second half of the public and private values; Merkle siblings by
    non-primitive op id, the right half's ids relative to the left verifier's op count).  Returns
    (left, right, left_non_primitive_ops) with pack_aggregation_inputs(left, right, n) == inputs."""
    pub = np.asarray(inputs.public_values, np.uint32).reshape(-1, 4)
    prv = np.asarray(inputs.private_values, np.uint32).reshape(-1, 4)
    ids = np.asarray(inputs.private_data_op_ids, np.uint32).reshape(-1)
    sib = np.asarray(inputs.private_data_siblings, np.uint32).reshape(-1, 8)
    hp, hv = pub.shape[0] // 2, prv.shape[0] // 2
    hs = ids.shape[0] // 2
    n_left = int(ids[hs]) if hs < ids.shape[0] else 0   # ids are ascending: everything below belongs to the left
    hs = int(np.searchsorted(ids, n_left))
    left = CircuitInputs(public_values=pub[:hp], private_values=prv[:hv], private_data_op_ids=ids[:hs],
                         private_data_siblings=sib[:hs])
    right = CircuitInputs(public_values=pub[hp:], private_values=prv[hv:],
                          private_data_op_ids=ids[hs:] - np.uint32(n_left), private_data_siblings=sib[hs:])
    return left, right, n_left
