"""Circuits over the binomial extensions of degree 2, 6 and 8 (x^D = W): the remaining circuit degrees of the
reference's `UnsupportedExtDegree` rule (batch_stark_prover.rs:666-681; `test_koalabear_batch_stark_extension_field_d8`,
tests.rs:486).  W is the caller's (`p3r_config.ext_w`, carried by the proof as `w_binomial`).  Primitive tables and
Recompose at the prove_all_tables boundary.  CPU: oracle round trips + the native verifier; GPU: bytes vs the oracle."""
import numpy as np
import pytest

import harness_lib
import layer_lib

NO_P2 = harness_lib.NO_POSEIDON2
CASES = [("koala-bear", 8, 3), ("koala-bear", 2, 3), ("baby-bear", 6, 11), ("baby-bear", 8, 11)]
SMALL = dict(horner_chain_len=12)


def native_verify(field, prm, tables, cap, proof, d, w):
    import plonky3_recursion_amd as p3r
    cfg, keep = p3r.make_config(field, prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, ext_degree=d, ext_w=w)
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"]) for t in tables]
    p3r.verify_batch(cfg, airs, cap, [int(t["main"].shape[0]).bit_length() - 1 for t in tables], proof)


@pytest.mark.parametrize("field,d,w", CASES)
def test_binomial_layers_roundtrip_and_native_verifier(oracle, field, d, w):
    import plonky3_recursion_amd as p3r
    prm = layer_lib.params(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=3, num_queries=5)
    arrs = harness_lib.generate(field, 7, seed=5 + d, flags=NO_P2, ext_degree=d, **SMALL)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=dict(ext_degree=d, ext_w=w))
    tables = L.tables()
    t = {x["kind"]: x for x in tables}
    assert set(t) == {"const", "public", "alu", "recompose"}
    assert t["alu"]["main"].shape[1] == (3 * 4 + 1 + 6 + 1) * d and t["recompose"]["main"].shape[1] == d
    pf = L.prove()
    L.verify(pf)
    native_verify(field, prm, tables, L.prep_commit(), pf, d, w)
    # another W is another multiplication rule
    with pytest.raises(p3r.P3rError):
        native_verify(field, prm, tables, L.prep_commit(), pf, d, w + 1)
    with pytest.raises(p3r.P3rError, match="MissingWForExtension"):
        native_verify(field, prm, tables, L.prep_commit(), pf, d, 0)
    for frac in (0.2, 0.7):
        bad = bytearray(pf)
        bad[int(len(bad) * frac)] ^= 1
        with pytest.raises(p3r.P3rError):
            native_verify(field, prm, tables, L.prep_commit(), bytes(bad), d, w)


def test_d8_product_is_checked(oracle):
    """The shape of the reference's D = 8 test (x * y * z == expected): a wrong top coefficient of one product."""
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    arrs = harness_lib.generate("koala-bear", 6, seed=2, flags=NO_P2 | harness_lib.NO_RECOMPOSE, ext_degree=8, **SMALL)
    v = arrs["alu_values"].reshape(-1, 32)
    k = arrs["alu_prep13"].reshape(-1, 13)
    mul = next(i for i, r in enumerate(k) if not (r[1] or r[2] or r[3] or r[4]) and v[i, 7] and v[i, 15])
    v[mul, 31] = (int(v[mul, 31]) + 1) % 0x7F000001
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(ext_degree=8, ext_w=3))
    with pytest.raises(RuntimeError, match="constraints do not match|final polynomial|terminals"):
        L.verify(L.prove())


@pytest.mark.gpu
@pytest.mark.parametrize("field,d,w", CASES)
def test_gpu_binomial_layers_match_the_oracle(oracle, field, d, w):
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=5)
    packing = dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3, recompose_lanes=2) if d == 8 else {}
    arrs = harness_lib.generate(field, 8, seed=9 + d, flags=NO_P2, ext_degree=d, horner_chain_len=20)
    prm = layer_lib.params(**kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=dict(packing, ext_degree=d, ext_w=w))
    ctx = p3r.Context(field=field, ext_degree=d, ext_w=w, **kw)
    tp = pv.TablePacking(public_lanes=packing.get("public_lanes", 1), alu_lanes=packing.get("alu_lanes", 3),
                         horner_packed_steps=packing.get("horner_packed_steps", 4),
                         recompose_lanes=packing.get("recompose_lanes", 1)).with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs, ext_degree=d), pv.FriRecursionBackend(),
                                     pv.ProveNextLayerParams(table_packing=tp))
    cpd = cache.circuit_prover_data
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    traces = wl.traces_from_arrays(arrs, ext_degree=d)
    res = pv.ResidentTraces(ctx, cpd, traces)
    tables = L.tables()
    slot = 0
    for i, h in enumerate(cpd.table_heights):
        if h:
            assert np.array_equal(cache.prover.build_main_trace(res, cpd, i).download(), tables[slot]["main"]), tables[slot]["kind"]
            slot += 1
    proof = cache.prover.prove_all_tables(res, cpd)
    assert proof.proof == L.prove()
    assert proof.ext_degree == d and proof.w_binomial == w and not proof.alu_quintic_trinomial
    back = pv.BatchStarkProof.from_postcard(proof.to_postcard(), field)
    assert back.w_binomial == w
    cache.prover.verify_all_tables(back)
    # the circuit boundary computes in degree 1, 4 and 5
    with pytest.raises(p3r.P3rError, match="UnsupportedDegree"):
        pv.PreparedCircuit(ctx, wl.circuit_from_arrays(arrs), tp)
    res.free()
    cpd.free()
    ctx.close()
    with pytest.raises(p3r.P3rError, match="MissingWForExtension"):
        p3r.Context(field=field, ext_degree=d, **kw)
