"""GPU parity for the circuit boundary: the device CircuitRunner (levelised schedule) and the
host-side preprocessing against the oracle's sequential restatement, then proof bytes of
prove_next_layer(circuit, inputs) against the oracle proving the oracle-run traces."""
import numpy as np
import pytest

import circuit_lib as cl
import harness_lib
import layer_lib
import oracle_lib

pytestmark = pytest.mark.gpu

FRI = dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)


def setup(oracle, field, log_h, flags=0, packing=None, **gen):
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    gen.setdefault("horner_chain_len", 16)
    gen.setdefault("sponge_chain_len", 3)
    gen.setdefault("merkle_depth", 5)
    a = harness_lib.generate(field, log_h, seed=31 + log_h, flags=flags, **gen)
    oc = cl.OracleCircuit(oracle, cl.Circuit.from_arrays(a)).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, cl.Inputs.from_arrays(a))
    prm = layer_lib.params(**FRI)
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking(**(packing or {})).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(a), p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=tp))
    return a, oc, prm, ctx, cache, wl.circuit_inputs_from_arrays(a)


SHAPES = [0, harness_lib.NO_POSEIDON2, harness_lib.NO_RECOMPOSE,
          harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE | harness_lib.SINGLE_PUBLIC,
          harness_lib.NO_POSEIDON2 | harness_lib.NO_ALU]


@pytest.mark.parametrize("field,log_h,flags", [("koala-bear", 7, f) for f in SHAPES] + [("baby-bear", 8, 0),
                                                                                       ("koala-bear", 10, 0)])
def test_device_runner_and_preprocessing_match_oracle(oracle, field, log_h, flags):
    import plonky3_recursion_amd as p3r
    a, oc, prm, ctx, cache, inputs = setup(oracle, field, log_h, flags)
    want = oc.workload_arrays()
    pc = cache.prepared_circuit
    assert [pc.circuit_prover_data.rows[k] for k in ("const", "public", "alu", "poseidon2", "recompose")] == \
        [int(x) for x in want["counts"][:5]]
    res = pc.run(inputs)
    assert np.array_equal(res.download("const_values").reshape(-1), want["const_values"])
    assert np.array_equal(res.download("public_values").reshape(-1), want["public_values"])
    assert np.array_equal(res.download("alu_values").reshape(-1), want["alu_values"])
    assert np.array_equal(res.download("recompose_values").reshape(-1), want["recompose_values"])
    if want["counts"][3]:
        assert np.array_equal(res.download("p2_input_values").reshape(-1), want["p2_inputs"])
        assert np.array_equal(res.download("p2_flags"), want["p2_flags"].reshape(-1, 4)[:, :3])
        assert np.array_equal(res.download("p2_mmcs_index_sum").reshape(-1), want["p2_mmcs_index_sum"])
    # preprocessing: the commitment binds every preprocessed column and multiplicity
    L = layer_lib.OracleLayer(oracle, field, want, prm)
    assert np.array_equal(pc.circuit_prover_data.preprocessed_commitment, L.prep_commit())
    # proof bytes, both through prove_next_layer(circuit inputs) and the one-call C entry point
    out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=inputs), ctx, p3r.FriRecursionBackend(),
                               p3r.ProveNextLayerParams(table_packing=pc.packing), prep=cache)
    proof = L.prove()
    assert out.proof.proof == proof
    assert pc.prove(inputs) == proof
    L.verify(out.proof.proof)
    assert pc.levels >= 2
    res.free()
    pc.free()
    ctx.close()


def test_runner_surface_and_errors(oracle):
    """CircuitRunner API (runner.rs:83-253) and its CircuitError paths on the device."""
    import plonky3_recursion_amd as p3r
    a, oc, prm, ctx, cache, inputs = setup(oracle, "koala-bear", 7)
    pc = cache.prepared_circuit
    r = pc.circuit.runner(pc)
    r.set_public_inputs(inputs.public_values)
    r.set_private_inputs(inputs.private_values)
    for op_id, sib in zip(inputs.private_data_op_ids, inputs.private_data_siblings):
        r.set_private_data(int(op_id), sib)
    res = r.run()
    assert np.array_equal(res.download("alu_values").reshape(-1), oc.workload_arrays()["alu_values"])
    res.free()
    # a public input that contradicts what the circuit computes from it: WitnessConflict, as in the oracle
    bad = p3r.CircuitInputs(inputs.public_values.copy(), inputs.private_values, inputs.private_data_op_ids,
                            inputs.private_data_siblings)
    P = oracle_lib.MODULUS["koala-bear"]
    p2_out_public = np.nonzero(a["p2_out_ctl"] == P - 1)[0]
    assert len(p2_out_public), "harness should route some Poseidon2 outputs onto public inputs"
    wid = int(a["p2_output_indices"][p2_out_public[0]])
    pos = int(np.nonzero(a["public_rows"] == wid)[0][0])
    bad.public_values[pos, 0] = (int(bad.public_values[pos, 0]) + 1) % P
    with pytest.raises(p3r.P3rError, match="WitnessConflict"):
        pc.run(bad)
    # the same through prove_next_layer, where the run is not awaited before proving starts: the
    # run's error is what the caller sees, and the prover is usable afterwards
    with pytest.raises(p3r.P3rError, match="WitnessConflict"):
        pc.prove(bad)
    assert pc.prove(inputs) == pc.prove(inputs)
    with pytest.raises(RuntimeError, match="WitnessConflict"):
        cl.OracleCircuit(oracle, cl.Circuit.from_arrays(a)).run("koala-bear", cl.Inputs(
            bad.public_values, bad.private_values, bad.private_data_op_ids, bad.private_data_siblings))
    # length mismatch, unknown op id, private data on a sponge row, double set
    with pytest.raises(p3r.P3rError, match="PublicInputLengthMismatch"):
        pc.run(p3r.CircuitInputs(inputs.public_values[:-1], inputs.private_values))
    with pytest.raises(p3r.P3rError, match="NonPrimitiveOpIdOutOfRange"):
        pc.run(p3r.CircuitInputs(inputs.public_values, inputs.private_values, [1 << 20], np.zeros((1, 8), np.uint32)))
    ops = a["ops"].reshape(-1, 8)
    sponge = ops[(ops[:, 0] == cl.OP_P2) & ((ops[:, 5] & 2) == 0)]
    with pytest.raises(p3r.P3rError, match="non-Merkle"):
        pc.run(p3r.CircuitInputs(inputs.public_values, inputs.private_values, [int(sponge[0, 1])],
                                 np.zeros((1, 8), np.uint32)))
    mid = int(inputs.private_data_op_ids[0])
    with pytest.raises(p3r.P3rError, match="already set"):
        pc.run(p3r.CircuitInputs(inputs.public_values, inputs.private_values, [mid, mid], np.zeros((2, 8), np.uint32)))
    pc.free()
    ctx.close()


def test_static_circuit_errors(oracle):
    import plonky3_recursion_amd as p3r
    ctx = p3r.Context(field="koala-bear", **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    N = cl.NO_W
    ops = [[cl.OP_CONST, 0, 0, N, 0, N, 0, 4], [cl.OP_CONST, 0, 0, N, 1, N, 4, 4], [cl.OP_PUBLIC, 0, 0, N, 2, 0, 0, 0],
           [cl.OP_ADD, 2, 1, N, 3, N, 0, 0]]
    ext = [0, 0, 0, 0, 5, 0, 0, 0]
    mk = lambda wc, ops, **kw: p3r.Circuit(witness_count=wc, ops=np.array(ops, dtype=np.uint32), ext=np.array(ext, np.uint32),
                                           public_rows=np.array([2], np.uint32), **kw)
    pc = p3r.PreparedCircuit(ctx, mk(4, ops), tp)
    res = pc.run(p3r.CircuitInputs(public_values=[[3, 0, 0, 0]]))
    assert res.download("alu_values").tolist() == [[3, 0, 0, 0, 5, 0, 0, 0, 0, 0, 0, 0, 8, 0, 0, 0]]
    res.free()
    pc.free()
    # a witness nobody writes: reported by run(), like the reference (runner.rs:218-221)
    pc = p3r.PreparedCircuit(ctx, mk(5, ops), tp)
    with pytest.raises(p3r.P3rError, match="WitnessNotSetForIndex"):
        pc.run(p3r.CircuitInputs(public_values=[[3, 0, 0, 0]]))
    pc.free()
    # division by zero in a backward Mul (a = witness 0 = 0)
    pc = p3r.PreparedCircuit(ctx, mk(5, ops + [[cl.OP_MUL, 0, 4, N, 3, N, 0, 0]]), tp)
    with pytest.raises(p3r.P3rError, match="DivisionByZero"):
        pc.run(p3r.CircuitInputs(public_values=[[3, 0, 0, 0]]))
    pc.free()
    # an unclaimed private input and an out-of-range witness are rejected at preparation
    with pytest.raises(p3r.P3rError, match="UnclaimedPrivateInput"):
        p3r.PreparedCircuit(ctx, mk(5, ops, private_input_rows=np.array([4], np.uint32)), tp)
    with pytest.raises(p3r.P3rError, match="out of bounds"):
        p3r.PreparedCircuit(ctx, mk(3, ops), tp)
    ctx.close()
