"""BASELINE config 0 (`recursive_fibonacci --field koala-bear --n 1000`): the example's base circuit
through the circuit boundary - table shapes of the reference (Const 2 rows, Public 1, ALU 999 ops at
height 1024 with TablePacking::new(1, 1) and min height 256), oracle prove, both verifiers."""
import numpy as np
import pytest

import circuit_lib as cl
import fib_lib
import layer_lib
import oracle_lib

FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15,
           num_queries=54)   # recursive_fibonacci.rs:71-147
PACKING = dict(public_lanes=1, alu_lanes=1, horner_packed_steps=2)   # TablePacking::new(1, 1): K keeps its default 2


def oracle_run(oracle, field, n=1000):
    circuit, inputs, fib = fib_lib.fibonacci_circuit(n, oracle_lib.MODULUS[field])
    oc = cl.OracleCircuit(oracle, circuit).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, inputs)
    return circuit, inputs, fib, oc


def test_fibonacci_base_circuit_tables_prove_and_verify(oracle):
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    circuit, inputs, fib, oc = oracle_run(oracle, field)
    w = oc.workload_arrays()
    assert [int(x) for x in w["counts"][:5]] == [2, 1, 999, 0, 0]
    alu = w["alu_values"].reshape(-1, 4, 4)
    assert alu[-1, 3, 0] == fib and not alu[:, :, 1:].any()            # F(1000) lands on the public witness
    p13 = w["alu_prep13"].reshape(-1, 13)
    P = oracle_lib.MODULUS[field]
    assert p13[-1, 10] == P - 1 and (p13[:-1, 10] != P - 1).all()       # only the connected output is a reader
    assert w["public_prep"].tolist() == [1, 4]                           # expected_result is read once (by that Add)
    prm = layer_lib.params(**FRI)
    L = layer_lib.OracleLayer(oracle, field, w, prm, packing=PACKING)
    tables = L.tables()
    assert [t["kind"] for t in tables] == ["const", "public", "alu"]     # no non-primitive tables in the batch
    assert [t["main"].shape[0] for t in tables] == [256, 256, 1024]
    proof = L.prove()
    L.verify(proof)
    cfg, keep = p3r.make_config(field, **{k: FRI[k] for k in FRI})
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"]) for t in tables]
    p3r.verify_batch(cfg, airs, L.prep_commit(), [int(t["main"].shape[0]).bit_length() - 1 for t in tables], proof)
    # the wrong expected_result is a WitnessConflict at run time (connect is enforced by the runner)
    bad = cl.Inputs(public_values=[(fib + 1) % P, 0, 0, 0])
    with pytest.raises(RuntimeError, match="WitnessConflict"):
        cl.OracleCircuit(oracle, circuit).run(field, bad)


@pytest.mark.gpu
@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
def test_fibonacci_base_circuit_on_device(oracle, field):
    import plonky3_recursion_amd as p3r
    circuit, inputs, fib, oc = oracle_run(oracle, field)
    w = oc.workload_arrays()
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking(public_lanes=1, alu_lanes=1, horner_packed_steps=2).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    pcirc = p3r.Circuit(circuit.witness_count, circuit.ops, circuit.ext, circuit.public_rows)
    cache = p3r.build_next_layer_prep(ctx, pcirc, p3r.FriRecursionBackend(), p3r.ProveNextLayerParams(table_packing=tp))
    pc = cache.prepared_circuit
    assert pc.circuit_prover_data.table_heights == [256, 256, 1024, 0, 0]
    assert pc.levels == 1000            # a chain of dependent additions: one level per op (+ the constants)
    pin = p3r.CircuitInputs(public_values=inputs.public_values.reshape(-1, 4))
    res = pc.run(pin)
    assert np.array_equal(res.download("alu_values").reshape(-1), w["alu_values"])
    res.free()
    out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=pin), ctx, p3r.FriRecursionBackend(),
                               p3r.ProveNextLayerParams(table_packing=tp), prep=cache)
    L = layer_lib.OracleLayer(oracle, field, w, layer_lib.params(**FRI), packing=PACKING)
    assert out.proof.proof == L.prove()
    cache.prover.verify_all_tables(out.proof)
    assert out.proof.non_primitives == () and out.proof.rows == (2, 1, 999)
    bad = p3r.CircuitInputs(public_values=[[(fib + 1) % oracle_lib.MODULUS[field], 0, 0, 0]])
    with pytest.raises(p3r.P3rError, match="WitnessConflict"):
        pc.run(bad)
    pc.free()
    ctx.close()
