"""Adapters from the synthetic-workload arrays (harness/synth.cpp) to the prover's input types."""
import numpy as np

from plonky3_recursion_amd.prover import Circuit, CircuitInputs, CircuitPrep, Traces


def traces_from_arrays(a) -> Traces:
    fl = a["p2_flags"].reshape(-1, 4)
    return Traces(
        const_values=a["const_values"].reshape(-1, 4),
        public_values=a["public_values"].reshape(-1, 4),
        alu_values=a["alu_values"].reshape(-1, 16),
        p2_input_values=a["p2_inputs"].reshape(-1, 16),
        p2_new_start=fl[:, 0].astype(np.uint8),
        p2_merkle_path=fl[:, 1].astype(np.uint8),
        p2_mmcs_bit=fl[:, 2].astype(np.uint8),
        p2_mmcs_index_sum=a["p2_mmcs_index_sum"],
        recompose_values=a["recompose_values"].reshape(-1, 4),
    )


def circuit_prep_from_arrays(a) -> CircuitPrep:
    fl = a["p2_flags"].reshape(-1, 4)
    return CircuitPrep(
        const_prep=a["const_prep"].reshape(-1, 2),
        public_prep=a["public_prep"].reshape(-1, 2),
        alu_prep13=a["alu_prep13"].reshape(-1, 13),
        recompose_prep=a["recompose_prep"].reshape(-1, 2),
        p2_new_start=fl[:, 0].astype(np.uint8),
        p2_merkle_path=fl[:, 1].astype(np.uint8),
        p2_mmcs_ctl_enabled=fl[:, 3].astype(np.uint8),
        p2_in_ctl=a["p2_in_ctl"].reshape(-1, 4).astype(np.uint8),
        p2_input_indices=a["p2_input_indices"].reshape(-1, 4),
        p2_out_ctl=a["p2_out_ctl"].reshape(-1, 2),
        p2_output_indices=a["p2_output_indices"].reshape(-1, 2),
        p2_mmcs_index_sum_idx=a["p2_mmcs_index_sum_idx"],
    )


def circuit_from_arrays(a) -> Circuit:
    return Circuit(witness_count=int(a["counts"][5]), ops=a["ops"].reshape(-1, 8), ext=a["ext"],
                   public_rows=a["public_rows"], private_input_rows=a["private_rows"],
                   witness_rewrite=a["rewrite"].reshape(-1, 2))


def circuit_inputs_from_arrays(a) -> CircuitInputs:
    return CircuitInputs(public_values=a["in_public_values"].reshape(-1, 4),
                         private_values=a["in_private_values"].reshape(-1, 4),
                         private_data_op_ids=a["pd_op_ids"], private_data_siblings=a["pd_siblings"].reshape(-1, 8))
