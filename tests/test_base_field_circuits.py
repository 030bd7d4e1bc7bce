"""D = 1: base-field circuits (`CircuitBuilder<F>`), what the reference proves first in every example - the base proof
of BASELINE config 0 is `prove_all_tables` over D = 1 traces of the Fibonacci circuit (recursive_fibonacci.rs:315-337,
batch_stark_prover/tests.rs:433).  Bus tuples are (idx, v); the Poseidon2 table of a base-field circuit is the
compact-D1 one on a 1-slot witness bus (`poseidon_d1_witness_bus_dim(1)`, batch_stark_prover.rs:84-90).
CPU: oracle round trips and the native verifier.  GPU: proof bytes against the oracle."""
import numpy as np
import pytest

import circuit_lib as cl
import fib_lib
import harness_lib
import layer_lib
import oracle_lib

FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15,
           num_queries=54)   # recursive_fibonacci.rs:71-147
SMALL = dict(horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)


def fibonacci_base_workload(oracle, field, n=1000):
    """The example's base circuit as D = 1 arrays: run through the oracle's circuit runner (which computes in the D = 4
    embedding), then drop the zero coefficients and unscale the witness indices."""
    circuit, inputs, fib = fib_lib.fibonacci_circuit(n, oracle_lib.MODULUS[field])
    oc = cl.OracleCircuit(oracle, circuit).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, inputs)
    w = {k: np.array(v, dtype=np.uint32) for k, v in oc.workload_arrays().items()}
    for name, per in (("const_values", 1), ("public_values", 1), ("alu_values", 4)):
        v = w[name].reshape(-1, per, 4)
        assert not v[:, :, 1:].any()
        w[name] = np.ascontiguousarray(v[:, :, 0]).reshape(-1)
    for name in ("const_prep", "public_prep"):
        p = w[name].reshape(-1, 2)
        assert (p[:, 1] % 4 == 0).all()
        p[:, 1] //= 4
    p13 = w["alu_prep13"].reshape(-1, 13)
    assert (p13[:, 5:9] % 4 == 0).all()
    p13[:, 5:9] //= 4
    return w, fib


def native_verify(field, prm, tables, cap, proof, ext_degree=1, coeff=0):
    import plonky3_recursion_amd as p3r
    cfg, keep = p3r.make_config(field, prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, ext_degree=ext_degree)
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"],
                 coeff_lookups=coeff if t["kind"] == "recompose" else 0) for t in tables]
    p3r.verify_batch(cfg, airs, cap, [int(t["main"].shape[0]).bit_length() - 1 for t in tables], proof)


def test_fibonacci_base_proof_as_the_reference_runs_it(oracle):
    """BASELINE config 0's base proof: D = 1 traces, TablePacking::new(1, 1), min height 256."""
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    w, fib = fibonacci_base_workload(oracle, field)
    assert [int(x) for x in w["counts"][:5]] == [2, 1, 999, 0, 0] and w["alu_values"].reshape(-1, 4)[-1, 3] == fib
    prm = layer_lib.params(**FRI)
    # TablePacking::new(1, 1): horner_packed_steps keeps its default 2 (packing.rs:36-47)
    L = layer_lib.OracleLayer(oracle, field, w, prm, packing=dict(public_lanes=1, alu_lanes=1, horner_packed_steps=2, ext_degree=1))
    tables = L.tables()
    # D = 1 widths: Const 1, Public 1, ALU 4 + (0 + 2 + 1) = 7 at K = 2 (alu_air.rs:320-325)
    assert [(t["kind"], t["main"].shape) for t in tables] == [("const", (256, 1)), ("public", (256, 1)), ("alu", (1024, 7))]
    proof = L.prove()
    L.verify(proof)
    native_verify(field, prm, tables, L.prep_commit(), proof)
    with pytest.raises(p3r.P3rError):
        native_verify(field, prm, tables, L.prep_commit(), proof, ext_degree=4)
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 2
    with pytest.raises(p3r.P3rError):
        native_verify(field, prm, tables, L.prep_commit(), bytes(bad))


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
@pytest.mark.parametrize("flags,coeff", [(harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE, 0), (0, 0),
                                         (harness_lib.RECOMPOSE_COEFF, 1)])
def test_base_field_layers_roundtrip_and_native_verifier(oracle, field, flags, coeff):
    import plonky3_recursion_amd as p3r
    prm = layer_lib.params(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=3, num_queries=5)
    arrs = harness_lib.generate(field, 7, seed=41, flags=flags, ext_degree=1, **SMALL)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=dict(ext_degree=1, recompose_coeff_lookups=coeff))
    tables = L.tables()
    t = {x["kind"]: x for x in tables}
    assert t["alu"]["main"].shape[1] == 3 * 4 + (1 + 6 + 1) and t["const"]["main"].shape[1] == 1
    if "poseidon2" in t:
        assert t["poseidon2"]["prep"].shape[1] == 62
    pf = L.prove()
    L.verify(pf)
    native_verify(field, prm, tables, L.prep_commit(), pf, coeff=coeff)
    for frac in (0.1, 0.5, 0.9):
        bad = bytearray(pf)
        bad[int(len(bad) * frac)] ^= 1
        with pytest.raises(RuntimeError):
            L.verify(bytes(bad))
        with pytest.raises(p3r.P3rError):
            native_verify(field, prm, tables, L.prep_commit(), bytes(bad), coeff=coeff)


def test_base_field_unsatisfied_product_is_rejected(oracle):
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    arrs = harness_lib.generate("koala-bear", 6, seed=4, flags=harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE,
                                ext_degree=1, **SMALL)
    v = arrs["alu_values"].reshape(-1, 4)
    k = arrs["alu_prep13"].reshape(-1, 13)
    mul = next(i for i, r in enumerate(k) if not (r[1] or r[2] or r[3] or r[4]) and v[i, 0] and v[i, 1])
    v[mul, 3] = (int(v[mul, 3]) + 1) % 0x7F000001
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(ext_degree=1))
    with pytest.raises(RuntimeError, match="constraints do not match|final polynomial|terminals"):
        L.verify(L.prove())


# ---------------------------------------------------------------------------------------------------- GPU
def gpu_setup(oracle, field, arrs, prm, packing, coeff=False):
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    ctx = p3r.Context(field=field, log_blowup=prm.log_blowup, max_log_arity=prm.max_log_arity, cap_height=prm.cap_height,
                      log_final_poly_len=prm.log_final_poly_len, commit_pow_bits=prm.commit_pow_bits,
                      query_pow_bits=prm.query_pow_bits, num_queries=prm.num_queries, ext_degree=1)
    tp = pv.TablePacking(public_lanes=packing.get("public_lanes", 1), alu_lanes=packing.get("alu_lanes", 3),
                         horner_packed_steps=packing.get("horner_packed_steps", 4),
                         recompose_lanes=packing.get("recompose_lanes", 1))
    tp.with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs, ext_degree=1, recompose_coeff_lookups=coeff),
                                     pv.FriRecursionBackend(), pv.ProveNextLayerParams(table_packing=tp))
    return ctx, cache, wl.traces_from_arrays(arrs, ext_degree=1)


@pytest.mark.gpu
def test_gpu_fibonacci_base_proof_matches_the_oracle(oracle):
    field = "koala-bear"
    w, fib = fibonacci_base_workload(oracle, field)
    for name in ("p2_inputs", "p2_flags", "p2_mmcs_index_sum", "p2_in_ctl", "p2_input_indices", "p2_out_ctl",
                 "p2_output_indices", "p2_mmcs_index_sum_idx", "recompose_values", "recompose_prep"):
        w.setdefault(name, np.zeros(0, np.uint32))
    prm = layer_lib.params(**FRI)
    packing = dict(public_lanes=1, alu_lanes=1, horner_packed_steps=2)   # TablePacking::new(1, 1)
    L = layer_lib.OracleLayer(oracle, field, w, prm, packing=dict(packing, ext_degree=1))
    ctx, cache, traces = gpu_setup(oracle, field, w, prm, packing)
    cpd = cache.circuit_prover_data
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    proof = cache.prover.prove_all_tables(traces, cpd)
    assert proof.proof == L.prove()
    assert proof.ext_degree == 1 and proof.w_binomial is None and not proof.alu_quintic_trinomial
    assert proof.rows == (2, 1, 999) and proof.degree_bits == (8, 8, 10)
    cache.prover.verify_all_tables(proof)
    cpd.free()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("field,log_h,kw,packing,flags", [
    ("koala-bear", 6, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=4, num_queries=5), {}, 0),
    ("baby-bear", 8, dict(log_blowup=1, max_log_arity=3, log_final_poly_len=2, cap_height=1, query_pow_bits=5, num_queries=5),
     dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3, recompose_lanes=2), harness_lib.RECOMPOSE_COEFF),
    ("koala-bear", 9, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=3, commit_pow_bits=2, query_pow_bits=6,
                           num_queries=6), dict(alu_lanes=4, horner_packed_steps=5),
     harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE),
    ("baby-bear", 7, dict(log_blowup=2, max_log_arity=1, log_final_poly_len=1, query_pow_bits=4, num_queries=6),
     dict(alu_lanes=1, horner_packed_steps=2), harness_lib.NO_RECOMPOSE),
])
def test_gpu_base_field_layers_match_the_oracle(oracle, field, log_h, kw, packing, flags):
    from plonky3_recursion_amd import prover as pv
    coeff = bool(flags & harness_lib.RECOMPOSE_COEFF)
    arrs = harness_lib.generate(field, log_h, seed=13 + log_h, flags=flags, ext_degree=1, horner_chain_len=20,
                                sponge_chain_len=3, merkle_depth=5)
    prm = layer_lib.params(**kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=dict(packing, ext_degree=1, recompose_coeff_lookups=int(coeff)))
    ctx, cache, traces = gpu_setup(oracle, field, arrs, prm, packing, coeff)
    cpd = cache.circuit_prover_data
    tables = L.tables()
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    res = pv.ResidentTraces(ctx, cpd, traces)
    slot = 0
    for i, h in enumerate(cpd.table_heights):
        if not h:
            continue
        got = cache.prover.build_main_trace(res, cpd, i).download()
        assert np.array_equal(got, tables[slot]["main"]), tables[slot]["kind"]
        slot += 1
    proof = cache.prover.prove_all_tables(res, cpd)
    assert proof.proof == L.prove()
    L.verify(proof.proof)
    cache.prover.verify_all_tables(pv.BatchStarkProof.from_postcard(proof.to_postcard(), field))
    res.free()
    cpd.free()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
def test_gpu_config_0_base_layer_end_to_end(oracle, field):
    """BASELINE config 0's base layer the way the example runs it, on the device: the `CircuitBuilder<F>` Fibonacci
    circuit (n = 1000) is prepared (get_airs_and_degrees_with_prep::<_, F, 1>), RUN (runner.run()) and proved
    (prove_all_tables) in one prove_next_layer call under ext_degree = 1; the proof is byte-identical to the oracle's
    proof of the same D = 1 traces and the native verifier accepts it."""
    import plonky3_recursion_amd as p3r
    w, fib = fibonacci_base_workload(oracle, field)
    circuit, inputs, fib2 = fib_lib.fibonacci_circuit(1000, oracle_lib.MODULUS[field], ext_degree=1)
    assert fib == fib2
    prm = layer_lib.params(**FRI)
    packing = dict(public_lanes=1, alu_lanes=1, horner_packed_steps=2)   # TablePacking::new(1, 1)
    L = layer_lib.OracleLayer(oracle, field, w, prm, packing=dict(packing, ext_degree=1))
    ctx = p3r.Context(field=field, ext_degree=1, **FRI)
    tp = p3r.TablePacking(**packing).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    pcirc = p3r.Circuit(circuit.witness_count, circuit.ops, circuit.ext, circuit.public_rows)
    cache = p3r.build_next_layer_prep(ctx, pcirc, p3r.FriRecursionBackend(), p3r.ProveNextLayerParams(table_packing=tp))
    pc = cache.prepared_circuit
    assert np.array_equal(pc.circuit_prover_data.preprocessed_commitment, L.prep_commit())
    pin = p3r.CircuitInputs(public_values=np.array([[fib]], dtype=np.uint32))
    res = pc.run(pin)
    assert np.array_equal(res.download("alu_values").reshape(-1), w["alu_values"])
    res.free()
    out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=pin), ctx, p3r.FriRecursionBackend(),
                               p3r.ProveNextLayerParams(table_packing=tp), prep=cache)
    assert out.proof.proof == L.prove()
    assert out.proof.ext_degree == 1 and out.proof.rows == (2, 1, 999)
    cache.prover.verify_all_tables(out.proof)
    # the wrong expected_result is a WitnessConflict at run time
    with pytest.raises(p3r.P3rError, match="WitnessConflict"):
        pc.run(p3r.CircuitInputs(public_values=np.array([[(fib + 1) % oracle_lib.MODULUS[field]]], dtype=np.uint32)))
    pc.free()
    ctx.close()
