"""GPU parity for D = 5 layers: KoalaBear circuits over the quintic trinomial extension x^5 + x^2 - 1 - the
primitive tables (Const, Public, ALU) and the compact-D1 Poseidon2 table - proved under the D = 4 STARK
configuration as the reference's unit tests do
(circuit-prover/src/batch_stark_prover/tests.rs:844-1029).  Matrices, preprocessed commitment and proof bytes
against the CPU oracle; the proof's metadata; what the D = 5 context refuses."""
import numpy as np
import pytest

import harness_lib
import layer_lib

pytestmark = pytest.mark.gpu

PRIMITIVE = harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE


def setup(oracle, log_h, kw, packing=None, flags=PRIMITIVE, ext_degree=5, **gen):
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    gen.setdefault("horner_chain_len", 20)
    gen.setdefault("sponge_chain_len", 3)
    gen.setdefault("merkle_depth", 5)
    arrs = harness_lib.generate("koala-bear", log_h, seed=11 + log_h, flags=flags, ext_degree=ext_degree, **gen)
    prm = layer_lib.params(**kw)
    packing = packing or {}
    coeff = bool(flags & harness_lib.RECOMPOSE_COEFF)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm,
                              packing=dict(packing, ext_degree=ext_degree, recompose_coeff_lookups=int(coeff)))
    ctx = p3r.Context(field="koala-bear", log_blowup=prm.log_blowup, max_log_arity=prm.max_log_arity,
                      cap_height=prm.cap_height, log_final_poly_len=prm.log_final_poly_len,
                      commit_pow_bits=prm.commit_pow_bits, query_pow_bits=prm.query_pow_bits,
                      num_queries=prm.num_queries, ext_degree=ext_degree)
    tp = pv.TablePacking(public_lanes=packing.get("public_lanes", 1), alu_lanes=packing.get("alu_lanes", 3),
                         horner_packed_steps=packing.get("horner_packed_steps", 4),
                         recompose_lanes=packing.get("recompose_lanes", 1))
    tp.with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs, ext_degree=ext_degree, recompose_coeff_lookups=coeff),
                                     pv.FriRecursionBackend(), pv.ProveNextLayerParams(table_packing=tp))
    return arrs, L, ctx, cache, wl.traces_from_arrays(arrs, ext_degree=ext_degree)


CASES = [
    (6, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=4, num_queries=5), None),
    (8, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=5, num_queries=5),
     dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3)),
    (7, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=1, query_pow_bits=4, num_queries=6),
     dict(alu_lanes=1, horner_packed_steps=2)),
    (9, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=3, cap_height=2, commit_pow_bits=2, query_pow_bits=6,
             num_queries=6), dict(alu_lanes=4, horner_packed_steps=5)),
    (10, dict(log_blowup=1, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=8),
     dict(alu_lanes=3, horner_packed_steps=8)),
]


WITH_P2 = harness_lib.NO_RECOMPOSE   # + the compact-D1 Poseidon2 table (KOALA_BEAR_D1_W16 on the 5-slot witness bus)
BACKEND_D5 = harness_lib.RECOMPOSE_COEFF   # + Recompose with coefficient lookups: the D = 5 backend's mix (fri.rs:741-852)
KINDS = {PRIMITIVE: [], WITH_P2: ["poseidon2"], BACKEND_D5: ["poseidon2", "recompose"], 0: ["poseidon2", "recompose"]}
NPO_NAMES = {"poseidon2": "poseidon2_perm/koala_bear_d1_w16"}


@pytest.mark.parametrize("flags", [PRIMITIVE, WITH_P2, BACKEND_D5, 0])
@pytest.mark.parametrize("log_h,kw,packing", CASES)
def test_quintic_layer_matrices_commitment_and_proof(oracle, log_h, kw, packing, flags):
    from plonky3_recursion_amd import prover as pv
    arrs, L, ctx, cache, traces = setup(oracle, log_h, kw, packing, flags=flags)
    tables = L.tables()
    cpd = cache.circuit_prover_data
    assert [t["kind"] for t in tables] == ["const", "public", "alu"] + KINDS[flags]
    assert [h for h in cpd.table_heights if h] == [t["main"].shape[0] for t in tables]
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    res = pv.ResidentTraces(ctx, cpd, traces)
    for i, t in enumerate(tables):
        got = cache.prover.build_main_trace(res, cpd, i).download()
        assert got.shape == t["main"].shape, t["kind"]
        assert np.array_equal(got, t["main"]), t["kind"]
    assert np.array_equal(res.download("alu_values"), traces.alu_values)
    out = pv.prove_next_layer(pv.RecursionInput(traces=traces), ctx, pv.FriRecursionBackend(),
                              pv.ProveNextLayerParams(table_packing=cpd.packing), prep=cache)
    want = L.prove()
    assert out.proof.proof == want
    L.verify(out.proof.proof)
    assert cache.prover.prove_all_tables(res, cpd).proof == want
    assert cache.prover.prove_all_tables(traces, cpd, canonical_field_encoding=True).proof == L.prove(field_encoding=1)
    # the metadata the reference writes next to the proof (batch_stark_prover.rs:1597-1641)
    p = out.proof
    assert p.ext_degree == 5 and p.w_binomial is None and p.alu_quintic_trinomial
    rec_name = "recompose/coeff" if flags == BACKEND_D5 else "recompose"
    assert [e.op_type for e in p.non_primitives] == [NPO_NAMES.get(k, rec_name) for k in KINDS[flags]]
    cache.prover.verify_all_tables(p)
    back = pv.BatchStarkProof.from_postcard(p.to_postcard(), "koala-bear")
    assert back.to_postcard() == p.to_postcard() and back.ext_degree == 5 and back.alu_quintic_trinomial
    cache.prover.verify_all_tables(back)
    res.free()
    cpd.free()
    ctx.close()


def test_quintic_unsatisfied_trace_is_reported(oracle):
    """The prover's self-check at zeta applies the trinomial rule: a product reduced the binomial way is refused."""
    import plonky3_recursion_amd as p3r
    arrs, L, ctx, cache, traces = setup(oracle, 6, dict(log_final_poly_len=1, query_pow_bits=3, num_queries=4))
    v = traces.alu_values.copy()
    v[3, 19] = (int(v[3, 19]) + 1) % 0x7F000001     # top coefficient of an output
    traces.alu_values = v
    with pytest.raises(p3r.P3rError, match="do not satisfy"):
        cache.prover.prove_all_tables(traces, cache.circuit_prover_data)
    cache.circuit_prover_data.free()
    ctx.close()


def test_compact_d1_chain_break_is_reported(oracle):
    """A chained capacity element of a sponge continuation row changed: the prover's self-check at zeta refuses."""
    import plonky3_recursion_amd as p3r
    arrs, L, ctx, cache, traces = setup(oracle, 6, dict(log_final_poly_len=1, query_pow_bits=3, num_queries=4), flags=WITH_P2)
    fl = arrs["p2_flags"].reshape(-1, 4)
    r = next(r for r in range(1, len(fl)) if not fl[r, 0] and not fl[r, 1])
    v = traces.p2_input_values.copy()
    v[r, 12] = (int(v[r, 12]) + 1) % 0x7F000001
    traces.p2_input_values = v
    with pytest.raises(p3r.P3rError, match="do not satisfy"):
        cache.prover.prove_all_tables(traces, cache.circuit_prover_data)
    cache.circuit_prover_data.free()
    ctx.close()


def test_what_a_quintic_context_refuses(oracle):
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    with pytest.raises(p3r.P3rError, match="UnsupportedExtDegree"):
        p3r.Context(field="baby-bear", ext_degree=5)
    with pytest.raises(p3r.P3rError, match="UnsupportedExtDegree"):
        p3r.Context(field="koala-bear", ext_degree=3)
    ctx = p3r.Context(field="koala-bear", log_final_poly_len=1, query_pow_bits=3, num_queries=4, ext_degree=5)
    tp = pv.TablePacking().with_fri_params(1, 2)
    # 4 x 2-limb Poseidon2 rows are a D = 4 layer's
    arrs4 = harness_lib.generate("koala-bear", 6, seed=1, horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)
    prep4 = wl.circuit_prep_from_arrays(arrs4)
    with pytest.raises(p3r.P3rError, match="ext_degree 5"):
        pv.build_next_layer_prep(ctx, prep4, pv.FriRecursionBackend(), pv.ProveNextLayerParams(table_packing=tp))
    arrs5 = harness_lib.generate("koala-bear", 6, seed=1, horner_chain_len=12, flags=PRIMITIVE, ext_degree=5)
    # D = 4 shaped values under a D = 5 context
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs5, ext_degree=5), pv.FriRecursionBackend(),
                                     pv.ProveNextLayerParams(table_packing=tp))
    bad = wl.traces_from_arrays(harness_lib.generate("koala-bear", 6, seed=1, horner_chain_len=12, flags=PRIMITIVE))
    with pytest.raises(p3r.P3rError, match="shape"):
        cache.prover.prove_all_tables(bad, cache.circuit_prover_data)
    cache.circuit_prover_data.free()
    ctx.close()


def test_quintic_layer_at_2_16_rows_verifies(oracle):
    """Past the sizes the oracle's prover finishes in seconds: the GPU proof of a 2^16-row D = 5 layer is accepted by
    the oracle's verifier (from the statement alone) and by the native verifier."""
    from plonky3_recursion_amd import prover as pv
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    prm = layer_lib.params(query_pow_bits=8, num_queries=20)
    arrs = harness_lib.generate("koala-bear", 16, seed=5, horner_chain_len=64, sponge_chain_len=6, merkle_depth=20,
                                flags=BACKEND_D5, ext_degree=5)
    ctx = p3r.Context(field="koala-bear", query_pow_bits=8, num_queries=20, ext_degree=5)
    tp = pv.TablePacking().with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs, ext_degree=5, recompose_coeff_lookups=True),
                                     pv.FriRecursionBackend(), pv.ProveNextLayerParams(table_packing=tp))
    proof = cache.prover.prove_all_tables(wl.traces_from_arrays(arrs, ext_degree=5), cache.circuit_prover_data)
    cache.prover.verify_all_tables(proof)
    airs = [dict(a, ext_degree=5) for a in proof.airs()]
    assert [a["kind"] for a in airs] == [0, 1, 2, 3, 4] and airs[4]["coeff_lookups"] == 1
    layer_lib.oracle_verify_statement(oracle, "koala-bear", prm, airs, proof.preprocessed_commitment, proof.proof)
    bad = bytearray(proof.proof)
    bad[len(bad) // 2] ^= 4
    with pytest.raises(RuntimeError):
        layer_lib.oracle_verify_statement(oracle, "koala-bear", prm, airs, proof.preprocessed_commitment, bytes(bad))
    cache.circuit_prover_data.free()
    ctx.close()


def test_recompose_coeff_variant_under_degree_four(oracle):
    """"recompose/coeff" is not tied to D = 5: a D = 4 backend registers it when its permutation is a D1 one
    (backend/fri.rs:693-721).  Same bytes as the oracle, metadata names the variant, native verifier accepts."""
    from plonky3_recursion_amd import prover as pv
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=5, num_queries=5)
    arrs, L, ctx, cache, traces = setup(oracle, 8, kw, dict(recompose_lanes=2), flags=harness_lib.RECOMPOSE_COEFF,
                                        ext_degree=4)
    cpd = cache.circuit_prover_data
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    proof = cache.prover.prove_all_tables(traces, cpd)
    assert proof.proof == L.prove()
    assert [e.op_type for e in proof.non_primitives] == ["poseidon2_perm/koala_bear_d4_w16", "recompose/coeff"]
    assert proof.preprocessed_widths[-1] == 2 * (2 + 8)
    cache.prover.verify_all_tables(pv.BatchStarkProof.from_postcard(proof.to_postcard(), "koala-bear"))
    cpd.free()
    ctx.close()


@pytest.mark.parametrize("ext_degree", [5, 4, 1])
@pytest.mark.parametrize("log_h,kw,packing", CASES[:3])
def test_layer_with_both_recompose_tables(oracle, log_h, kw, packing, ext_degree):
    """`recompose` and `recompose/coeff` in one layer (recompose_table_provers(lanes, true), batch_stark_prover.rs:
    1914-1932): six tables at the prove_all_tables boundary - matrices, commitment, proof bytes, metadata."""
    from plonky3_recursion_amd import prover as pv
    arrs, L, ctx, cache, traces = setup(oracle, log_h, kw, packing, flags=harness_lib.RECOMPOSE_BOTH, ext_degree=ext_degree)
    tables = L.tables()
    cpd = cache.circuit_prover_data
    assert [t["kind"] for t in tables] == ["const", "public", "alu", "poseidon2", "recompose", "recompose"]
    assert cpd.rows["recompose"] == int(arrs["counts"][4]) and cpd.rows["recompose_coeff"] == int(arrs["counts"][6]) > 0
    assert cpd.table_heights + [cpd.recompose_coeff_height] == [t["main"].shape[0] for t in tables]
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    res = pv.ResidentTraces(ctx, cpd, traces)
    for i, t in enumerate(tables):
        got = cache.prover.build_main_trace(res, cpd, i).download()
        assert np.array_equal(got, t["main"]), (i, t["kind"])
    want = L.prove()
    p = cache.prover.prove_all_tables(res, cpd)
    assert p.proof == want
    assert cache.prover.prove_all_tables(traces, cpd, canonical_field_encoding=True).proof == L.prove(field_encoding=1)
    assert [e.op_type for e in p.non_primitives[-2:]] == ["recompose", "recompose/coeff"]
    cache.prover.verify_all_tables(p)
    back = pv.BatchStarkProof.from_postcard(p.to_postcard(), "koala-bear")
    assert back.to_postcard() == p.to_postcard()
    cache.prover.verify_all_tables(back)
    L.verify(p.proof)
    # a second table next to one that already is the coefficient kind is refused
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    prep = wl.circuit_prep_from_arrays(arrs, ext_degree=ext_degree)
    prep.recompose_coeff_lookups = True
    prep.recompose_prep = np.zeros((len(prep.recompose_prep), 2 + 2 * ext_degree), np.uint32)
    with pytest.raises(p3r.P3rError, match="plain kind"):
        pv.CircuitProverData(ctx, prep, cpd.packing)
    res.free()
    cpd.free()
    ctx.close()


@pytest.mark.parametrize("d,dc,packing", [
    (5, 5, dict(public_lanes=1, alu_lanes=8, horner_packed_steps=2)),                      # TablePacking::new(1, 8)
    (4, 4, dict(public_lanes=1, alu_lanes=8, horner_packed_steps=2, recompose_lanes=2)),
    (5, 5, dict(public_lanes=3, alu_lanes=6, horner_packed_steps=3, recompose_lanes=2)),
    (1, 4, dict(public_lanes=4, alu_lanes=8, horner_packed_steps=8)),
])
def test_wide_packings_of_a_six_table_layer(oracle, d, dc, packing):
    """The verifier circuit of the reference's quintic test is proved with `TablePacking::new(1, 8)` - eight ALU lanes -
    and both Recompose tables (fibonacci_batch_stark_prover_quintic.rs:173-181, :236-238): wide rows, six tables, the
    quintic configuration; proof bytes against the oracle."""
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
    arrs = harness_lib.generate("koala-bear", 9, seed=5, flags=harness_lib.RECOMPOSE_BOTH, ext_degree=d, horner_chain_len=20,
                                sponge_chain_len=3, merkle_depth=5)
    prm = layer_lib.params(challenge_degree=dc, **kw)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(packing, ext_degree=d))
    ctx = p3r.Context(field="koala-bear", ext_degree=d, challenge_degree=dc, **kw)
    tp = pv.TablePacking(**packing).with_fri_params(kw["log_final_poly_len"], kw["log_blowup"])
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs, ext_degree=d), pv.FriRecursionBackend(),
                                     pv.ProveNextLayerParams(table_packing=tp))
    cpd = cache.circuit_prover_data
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    p = cache.prover.prove_all_tables(wl.traces_from_arrays(arrs, ext_degree=d), cpd)
    assert p.proof == L.prove()
    assert p.table_packing.alu_lanes == packing["alu_lanes"]
    cache.prover.verify_all_tables(p)
    cpd.free()
    ctx.close()
