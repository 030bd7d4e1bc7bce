"""The STARK over KoalaBear's quintic CHALLENGE field (`Challenge = QuinticTrinomialExtensionField<KoalaBear>`,
test-utils koala_bear_quintic_params; recursive_fibonacci --quintic; recursion/tests/
fibonacci_batch_stark_prover_quintic.rs): LogUp fractions, constraint folding, openings, FRI and the transcript run
over five-coefficient elements; the base-field side (traces, LDE, MMCS) is unchanged.
CPU: the oracle's prover against its verifier and the product's native verifier.  GPU: proof bytes."""
import numpy as np
import pytest

import harness_lib
import layer_lib

FIELD = "koala-bear"
SMALL = dict(horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)
# circuit degree x table mix: the quintic recursion backend's own (D = 5), the base proof (D = 1), and D = 4 tables
MIXES = [(5, harness_lib.RECOMPOSE_COEFF, 1), (1, harness_lib.NO_RECOMPOSE, 0), (4, 0, 0)]


def native_verify(prm, tables, cap, proof, d, coeff=0, challenge_degree=5, canonical=False):
    import plonky3_recursion_amd as p3r
    cfg, keep = p3r.make_config(FIELD, prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, ext_degree=d,
                                challenge_degree=challenge_degree)
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"],
                 coeff_lookups=coeff if t["kind"] == "recompose" else 0) for t in tables]
    p3r.verify_batch(cfg, airs, cap, [int(t["main"].shape[0]).bit_length() - 1 for t in tables], proof, canonical)


@pytest.mark.parametrize("d,flags,coeff", MIXES)
@pytest.mark.parametrize("log_h,kw", [
    (5, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=0)),
    (7, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2)),
    (7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, cap_height=2, commit_pow_bits=2)),
])
def test_oracle_and_native_verifier_over_the_quintic_challenge_field(oracle, d, flags, coeff, log_h, kw):
    import plonky3_recursion_amd as p3r
    prm = layer_lib.params(query_pow_bits=3, num_queries=5, challenge_degree=5, **kw)
    arrs = harness_lib.generate(FIELD, log_h, seed=90 + log_h, flags=flags, ext_degree=d, **SMALL)
    packing = dict(ext_degree=d, recompose_coeff_lookups=coeff)
    L = layer_lib.OracleLayer(oracle, FIELD, arrs, prm, packing=packing)
    pf = L.prove()
    L.verify(pf)
    assert L.prove() == pf
    tables, cap = L.tables(), L.prep_commit()
    native_verify(prm, tables, cap, pf, d, coeff)
    native_verify(prm, tables, cap, L.prove(field_encoding=1), d, coeff, canonical=True)
    # five words per extension element: the quartic verifier cannot even frame it, and vice versa
    with pytest.raises(p3r.P3rError):
        native_verify(prm, tables, cap, pf, d, coeff, challenge_degree=4)
    prm4 = layer_lib.params(query_pow_bits=3, num_queries=5, **kw)
    pf4 = layer_lib.OracleLayer(oracle, FIELD, arrs, prm4, packing=packing).prove()
    assert len(pf) > len(pf4)
    with pytest.raises(p3r.P3rError):
        native_verify(prm, tables, cap, pf4, d, coeff)
    for pos in range(7, len(pf), max(len(pf) // 25, 1)):
        bad = bytearray(pf)
        bad[pos] ^= 1
        with pytest.raises(RuntimeError):
            L.verify(bytes(bad))
        with pytest.raises(p3r.P3rError):
            native_verify(prm, tables, cap, bytes(bad), d, coeff)


def test_quintic_challenge_is_koala_bears(oracle):
    import plonky3_recursion_amd as p3r
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=3)
    arrs = harness_lib.generate("baby-bear", 6, seed=1, **SMALL)
    L = layer_lib.OracleLayer(oracle, "baby-bear", arrs, prm)
    cfg, keep = p3r.make_config("baby-bear", prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, challenge_degree=5)
    tables = L.tables()
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"]) for t in tables]
    with pytest.raises(p3r.P3rError, match="UnsupportedChallengeDegree"):
        p3r.verify_batch(cfg, airs, L.prep_commit(), [int(t["main"].shape[0]).bit_length() - 1 for t in tables], L.prove())


# ---- GPU: the device prover over the quintic challenge field ------------------------------------------------------
GPU_CASES = [
    (6, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=4, num_queries=5), None),
    (8, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=5, num_queries=5),
     dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3)),
    (7, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=1, query_pow_bits=4, num_queries=6),
     dict(alu_lanes=1, horner_packed_steps=2)),
    (9, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=3, cap_height=2, commit_pow_bits=2, query_pow_bits=6,
             num_queries=6), dict(alu_lanes=4, horner_packed_steps=5)),
    (11, dict(log_blowup=1, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=8),
     dict(alu_lanes=3, horner_packed_steps=8)),
    (13, dict(log_blowup=3, max_log_arity=3, log_final_poly_len=0, query_pow_bits=3, num_queries=4),
     dict(alu_lanes=2, horner_packed_steps=4, recompose_lanes=2)),
]


def gpu_setup(oracle, d, flags, coeff, log_h, kw, packing):
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    arrs = harness_lib.generate(FIELD, log_h, seed=31 + log_h, flags=flags, ext_degree=d, horner_chain_len=20,
                                sponge_chain_len=3, merkle_depth=5)
    prm = layer_lib.params(challenge_degree=5, **kw)
    packing = packing or {}
    L = layer_lib.OracleLayer(oracle, FIELD, arrs, prm, packing=dict(packing, ext_degree=d, recompose_coeff_lookups=coeff))
    ctx = p3r.Context(field=FIELD, log_blowup=prm.log_blowup, max_log_arity=prm.max_log_arity,
                      cap_height=prm.cap_height, log_final_poly_len=prm.log_final_poly_len,
                      commit_pow_bits=prm.commit_pow_bits, query_pow_bits=prm.query_pow_bits,
                      num_queries=prm.num_queries, ext_degree=d, challenge_degree=5)
    tp = pv.TablePacking(public_lanes=packing.get("public_lanes", 1), alu_lanes=packing.get("alu_lanes", 3),
                         horner_packed_steps=packing.get("horner_packed_steps", 4),
                         recompose_lanes=packing.get("recompose_lanes", 1))
    tp.with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs, ext_degree=d, recompose_coeff_lookups=bool(coeff)),
                                     pv.FriRecursionBackend(), pv.ProveNextLayerParams(table_packing=tp))
    return arrs, L, ctx, cache, wl.traces_from_arrays(arrs, ext_degree=d)


@pytest.mark.gpu
@pytest.mark.parametrize("d,flags,coeff", MIXES)
@pytest.mark.parametrize("log_h,kw,packing", GPU_CASES)
def test_device_proof_bytes_over_the_quintic_challenge_field(oracle, d, flags, coeff, log_h, kw, packing):
    from plonky3_recursion_amd import prover as pv
    arrs, L, ctx, cache, traces = gpu_setup(oracle, d, flags, coeff, log_h, kw, packing)
    cpd = cache.circuit_prover_data
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    out = pv.prove_next_layer(pv.RecursionInput(traces=traces), ctx, pv.FriRecursionBackend(),
                              pv.ProveNextLayerParams(table_packing=cpd.packing), prep=cache)
    want = L.prove()
    assert out.proof.proof == want
    res = pv.ResidentTraces(ctx, cpd, traces)
    assert cache.prover.prove_all_tables(res, cpd).proof == want
    assert cache.prover.prove_all_tables(traces, cpd, canonical_field_encoding=True).proof == L.prove(field_encoding=1)
    p = out.proof
    assert p.ext_degree == d
    cache.prover.verify_all_tables(p)
    back = pv.BatchStarkProof.from_postcard(p.to_postcard(), FIELD, challenge_degree=5)
    assert back.to_postcard() == p.to_postcard()
    cache.prover.verify_all_tables(back)
    with pytest.raises(Exception):   # five-word elements do not frame as a quartic proof
        pv.BatchStarkProof.from_postcard(p.to_postcard(), FIELD)
    res.free()
    cpd.free()
    ctx.close()


@pytest.mark.gpu
def test_quintic_challenge_self_check_refuses_a_broken_trace(oracle):
    import plonky3_recursion_amd as p3r
    arrs, L, ctx, cache, traces = gpu_setup(oracle, 5, harness_lib.RECOMPOSE_COEFF, 1, 6,
                                            dict(log_final_poly_len=1, query_pow_bits=3, num_queries=4), None)
    v = traces.alu_values.copy()
    v[3, 19] = (int(v[3, 19]) + 1) % 0x7F000001
    traces.alu_values = v
    with pytest.raises(p3r.P3rError, match="do not satisfy"):
        cache.prover.prove_all_tables(traces, cache.circuit_prover_data)
    cache.circuit_prover_data.free()
    ctx.close()


@pytest.mark.gpu
def test_quintic_challenge_context_is_koala_bears():
    import plonky3_recursion_amd as p3r
    with pytest.raises(p3r.P3rError, match="UnsupportedChallengeDegree"):
        p3r.Context(field="baby-bear", challenge_degree=5)
    with pytest.raises(p3r.P3rError, match="UnsupportedChallengeDegree"):
        p3r.Context(field="koala-bear", challenge_degree=3)


# ---- CPU: six-table layers (`recompose` next to `recompose/coeff`) through the oracle and the native verifier ------
@pytest.mark.parametrize("d,dc", [(5, 5), (5, 4), (4, 4), (1, 4)])
def test_six_table_layer_oracle_and_native_verifier(oracle, d, dc):
    """A backend with coefficient lookups registers both Recompose table provers (batch_stark_prover.rs:1914-1932);
    the statement the native verifier rebuilds lists `recompose` before `recompose/coeff`.  Declaring the second
    table as the plain kind, or swapping the two, is a different statement."""
    import plonky3_recursion_amd as p3r
    prm = layer_lib.params(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=5,
                           challenge_degree=dc)
    arrs = harness_lib.generate(FIELD, 7, seed=404, flags=harness_lib.RECOMPOSE_BOTH, ext_degree=d, **SMALL)
    L = layer_lib.OracleLayer(oracle, FIELD, arrs, prm, packing=dict(ext_degree=d))
    tables, cap = L.tables(), L.prep_commit()
    assert [t["kind"] for t in tables[-2:]] == ["recompose", "recompose"]
    assert tables[-2]["prep"].shape[1] == 2 and tables[-1]["prep"].shape[1] == 2 + 2 * d
    pf = L.prove()
    L.verify(pf)
    cfg, keep = p3r.make_config(FIELD, prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, ext_degree=d, challenge_degree=dc)
    db = [int(t["main"].shape[0]).bit_length() - 1 for t in tables]

    def airs(kinds):
        out = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"], coeff_lookups=0) for t in tables]
        out[-2]["coeff_lookups"], out[-1]["coeff_lookups"] = kinds
        return out
    p3r.verify_batch(cfg, airs((0, 1)), cap, db, pf)
    for wrong in ((0, 0), (1, 1), (1, 0)):
        with pytest.raises(p3r.P3rError):
            p3r.verify_batch(cfg, airs(wrong), cap, db, pf)


# ---- the reference's own quintic test, first half (recursion/tests/fibonacci_batch_stark_prover_quintic.rs:62-113) ----
def fibonacci_quintic_workload(oracle, n):
    """Fibonacci(n) built with CircuitBuilder<Challenge> (Challenge = the quintic field): D = 5 arrays derived from the
    oracle's circuit runner, which computes in the D = 4 embedding - values are base-field elements either way."""
    import circuit_lib as cl
    import fib_lib
    import oracle_lib
    circuit, inputs, fib = fib_lib.fibonacci_circuit(n, oracle_lib.MODULUS[FIELD])
    oc = cl.OracleCircuit(oracle, circuit).preprocess(oracle_lib.MODULUS[FIELD])
    oc.run(FIELD, inputs)
    w = {k: np.array(v, dtype=np.uint32) for k, v in oc.workload_arrays().items()}
    for name, per in (("const_values", 1), ("public_values", 1), ("alu_values", 4)):
        v = w[name].reshape(-1, per, 4)
        assert not v[:, :, 1:].any()
        out = np.zeros((v.shape[0], per, 5), np.uint32)
        out[:, :, 0] = v[:, :, 0]
        w[name] = out.reshape(-1)
    for name in ("const_prep", "public_prep"):
        p = w[name].reshape(-1, 2)
        p[:, 1] = p[:, 1] // 4 * 5
    p13 = w["alu_prep13"].reshape(-1, 13)
    p13[:, 5:9] = p13[:, 5:9] // 4 * 5
    return w, fib


@pytest.mark.gpu
def test_fibonacci_batch_verifier_quintic_koala_inner_proof(oracle):
    """`test_fibonacci_batch_verifier_quintic_koala`: n = 48, `CircuitBuilder::<Challenge>`, TablePacking::new(2, 4),
    get_airs_and_degrees_with_prep::<MyConfig, _, 5>, runner.run(), prove_all_tables, verify_all_tables::<Challenge>
    under koala_bear_quintic_params - here as one prove_next_layer call on the device (ext_degree = 5,
    challenge_degree = 5), proof bytes against the oracle.  (FriParameters::new_testing's scalars are upstream's;
    small test parameters of the same kind are used.)"""
    import fib_lib
    import oracle_lib
    import plonky3_recursion_amd as p3r
    fri = dict(log_blowup=2, max_log_arity=1, cap_height=0, log_final_poly_len=0, commit_pow_bits=1, query_pow_bits=1,
               num_queries=2)
    w, fib = fibonacci_quintic_workload(oracle, 48)
    assert [int(x) for x in w["counts"][:5]] == [2, 1, 47, 0, 0]
    prm = layer_lib.params(challenge_degree=5, **fri)
    packing = dict(public_lanes=2, alu_lanes=4, horner_packed_steps=2)   # TablePacking::new(2, 4)
    L = layer_lib.OracleLayer(oracle, FIELD, w, prm, packing=dict(packing, ext_degree=5))
    circuit, inputs, fib2 = fib_lib.fibonacci_circuit(48, oracle_lib.MODULUS[FIELD], ext_degree=5)
    assert fib == fib2 == 4807526976 % oracle_lib.MODULUS[FIELD]
    ctx = p3r.Context(field=FIELD, ext_degree=5, challenge_degree=5, **fri)
    tp = p3r.TablePacking(**packing).with_fri_params(fri["log_final_poly_len"], fri["log_blowup"])
    pcirc = p3r.Circuit(circuit.witness_count, circuit.ops, circuit.ext, circuit.public_rows)
    cache = p3r.build_next_layer_prep(ctx, pcirc, p3r.FriRecursionBackendD5(), p3r.ProveNextLayerParams(table_packing=tp))
    assert np.array_equal(cache.prepared_circuit.circuit_prover_data.preprocessed_commitment, L.prep_commit())
    pin = p3r.CircuitInputs(public_values=np.array([[fib, 0, 0, 0, 0]], dtype=np.uint32))
    out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=pin), ctx, p3r.FriRecursionBackendD5(),
                               p3r.ProveNextLayerParams(table_packing=tp), prep=cache)
    want = L.prove()
    assert out.proof.proof == want
    L.verify(want)
    p = out.proof
    assert p.ext_degree == 5 and p.alu_quintic_trinomial and p.w_binomial is None and p.rows == (2, 1, 47)
    assert p.non_primitives == ()
    cache.prover.verify_all_tables(p)
    back = p3r.BatchStarkProof.from_postcard(p.to_postcard(), FIELD, challenge_degree=5)
    cache.prover.verify_all_tables(back)
    with pytest.raises(p3r.P3rError, match="WitnessConflict"):
        cache.prepared_circuit.run(p3r.CircuitInputs(public_values=np.array([[fib + 1, 0, 0, 0, 0]], dtype=np.uint32)))
    cache.prepared_circuit.free()
    ctx.close()
