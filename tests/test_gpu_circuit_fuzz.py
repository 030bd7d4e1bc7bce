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


@pytest.mark.parametrize("field,seeds,n_ops", [("koala-bear", range(300, 316), 400), ("baby-bear", range(400, 406), 400),
                                               ("koala-bear", range(500, 502), 3000)])
def test_random_circuits_with_width32_rows(oracle, field, seeds, n_ops):
    """The same with chains of the WIDTH-32 table (P3R_OP_POSEIDON2_W32_PERM) mixed in: sponge and Merkle rows in any order
    (both chain states of the op type), rows that wait for witnesses of later levels (segments that split), outputs read by
    ALU ops; rows, flags, and the 48-column preprocessed rows with their multiplicities through the commitment."""
    import plonky3_recursion_amd as p3r
    ctx = p3r.Context(field=field, allow_unpinned_w32_defaults=True, **FRI)
    tp = p3r.TablePacking(public_lanes=2, alu_lanes=2, recompose_lanes=2).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    P = oracle_lib.MODULUS[field]
    saw_rows = 0
    for seed in seeds:
        circuit, inputs = circuit_fuzz.random_circuit(seed, n_ops=n_ops, modulus=P, w32=True)
        oc = cl.OracleCircuit(oracle, circuit).preprocess(P)
        oc.run(field, inputs)
        want = oc.workload_arrays()
        saw_rows += int(want["counts"][7])
        pc = p3r.PreparedCircuit(ctx, p3r.Circuit(circuit.witness_count, circuit.ops, circuit.ext, circuit.public_rows,
                                                  circuit.private_rows, circuit.rewrite.reshape(-1, 2)), tp)
        cin = p3r.CircuitInputs(inputs.public_values.reshape(-1, 4), inputs.private_values.reshape(-1, 4), inputs.pd_op_ids,
                                inputs.pd_siblings.reshape(-1, 8), inputs.pdw_op_ids, inputs.pdw_siblings.reshape(-1, 24))
        res = pc.run(cin)
        for name, key in (("alu_values", "alu_values"), ("recompose_values", "recompose_values"), ("p2_input_values", "p2_inputs"),
                          ("p2w_input_values", "p2w_inputs"), ("p2w_flags", "p2w_flags"), ("p2w_mmcs_index_sum", "p2w_mmcs_index_sum")):
            assert np.array_equal(res.download(name).reshape(-1), want[key]), (seed, name)
        L = layer_lib.OracleLayer(oracle, field, want, layer_lib.params(**FRI), packing=dict(public_lanes=2, alu_lanes=2, recompose_lanes=2))
        assert np.array_equal(pc.circuit_prover_data.preprocessed_commitment, L.prep_commit()), seed
        # (these circuits exercise the runner and the preparation, not the AIRs: a random op list is not a provable statement)
        res.free()
        pc.free()
    assert saw_rows > 50 * len(seeds) // 4
    ctx.close()
