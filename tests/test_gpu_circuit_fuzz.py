"""GPU: the static execution schedule (levels, Horner-chain scans, Poseidon2 segments, fused narrow
levels, writes that become comparisons) and the host-side preprocessing against the oracle's
sequential runner / preprocessing on random circuits of arbitrary dependency structure."""
import numpy as np
import pytest

import circuit_fuzz
import circuit_lib as cl
import layer_lib
import oracle_lib

pytestmark = pytest.mark.gpu
FRI = dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=2, num_queries=3)


@pytest.mark.parametrize("field,seeds,n_ops", [("koala-bear", range(0, 24), 300), ("baby-bear", range(100, 108), 300),
                                               ("koala-bear", range(200, 204), 3000)])
def test_random_circuits_run_and_preprocess_like_the_oracle(oracle, field, seeds, n_ops):
    import plonky3_recursion_amd as p3r
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking(public_lanes=2, alu_lanes=2, recompose_lanes=2).with_fri_params(FRI["log_final_poly_len"],
                                                                                          FRI["log_blowup"])
    P = oracle_lib.MODULUS[field]
    for seed in seeds:
        circuit, inputs = circuit_fuzz.random_circuit(seed, n_ops=n_ops, modulus=P)
        oc = cl.OracleCircuit(oracle, circuit).preprocess(P)
        oc.run(field, inputs)
        want = oc.workload_arrays()
        pc = p3r.PreparedCircuit(ctx, p3r.Circuit(circuit.witness_count, circuit.ops, circuit.ext, circuit.public_rows,
                                                  circuit.private_rows, circuit.rewrite.reshape(-1, 2)), tp)
        res = pc.run(p3r.CircuitInputs(inputs.public_values.reshape(-1, 4), inputs.private_values.reshape(-1, 4),
                                       inputs.pd_op_ids, inputs.pd_siblings.reshape(-1, 8)))
        for name, key in (("const_values", "const_values"), ("public_values", "public_values"), ("alu_values", "alu_values"),
                          ("recompose_values", "recompose_values"), ("p2_input_values", "p2_inputs"),
                          ("p2_mmcs_index_sum", "p2_mmcs_index_sum")):
            assert np.array_equal(res.download(name).reshape(-1), want[key]), (seed, name)
        assert np.array_equal(res.download("p2_flags"), want["p2_flags"].reshape(-1, 4)[:, :3]), seed
        L = layer_lib.OracleLayer(oracle, field, want, layer_lib.params(**FRI),
                                  packing=dict(public_lanes=2, alu_lanes=2, recompose_lanes=2))
        assert np.array_equal(pc.circuit_prover_data.preprocessed_commitment, L.prep_commit()), seed
        res.free()
        pc.free()
    ctx.close()
