"""GPU parity for the prove_all_tables boundary: K1/K2/K3 matrices, preprocessed commitment and
final proof bytes vs the CPU oracle, through the host-side mirror of the reference API."""
import numpy as np
import pytest

import harness_lib
import layer_lib

pytestmark = pytest.mark.gpu


def setup(oracle, field, log_h, kw, packing=None, **gen):
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    gen.setdefault("horner_chain_len", 20)
    gen.setdefault("sponge_chain_len", 3)
    gen.setdefault("merkle_depth", 5)
    arrs = harness_lib.generate(field, log_h, seed=7 + log_h, **gen)
    prm = layer_lib.params(**kw)
    packing = packing or {}
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=dict(packing))
    ctx = p3r.Context(field=field, log_blowup=prm.log_blowup, max_log_arity=prm.max_log_arity,
                      cap_height=prm.cap_height, log_final_poly_len=prm.log_final_poly_len,
                      commit_pow_bits=prm.commit_pow_bits, query_pow_bits=prm.query_pow_bits,
                      num_queries=prm.num_queries)
    tp = pv.TablePacking(public_lanes=packing.get("public_lanes", 1), alu_lanes=packing.get("alu_lanes", 3),
                         horner_packed_steps=packing.get("horner_packed_steps", 4),
                         recompose_lanes=packing.get("recompose_lanes", 1))
    tp.with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs), pv.FriRecursionBackend(),
                                     pv.ProveNextLayerParams(table_packing=tp))
    return arrs, L, ctx, cache, wl.traces_from_arrays(arrs)


CASES = [
    ("koala-bear", 6, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=4, num_queries=5), None),
    ("koala-bear", 8, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=5, num_queries=5),
     dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3, recompose_lanes=2)),
    ("baby-bear", 7, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=1, query_pow_bits=4, num_queries=6),
     dict(alu_lanes=1, horner_packed_steps=2)),
    ("koala-bear", 9, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=3, query_pow_bits=6, num_queries=6),
     dict(alu_lanes=4, horner_packed_steps=5)),
]


@pytest.mark.parametrize("field,log_h,kw,packing", CASES)
def test_layer_matrices_commitment_and_proof(oracle, field, log_h, kw, packing):
    from plonky3_recursion_amd import prover as pv
    arrs, L, ctx, cache, traces = setup(oracle, field, log_h, kw, packing)
    tables = L.tables()
    cpd = cache.circuit_prover_data
    assert cpd.table_heights == [t["main"].shape[0] for t in tables]
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    res = pv.ResidentTraces(ctx, cpd, traces)
    for i, t in enumerate(tables):
        got = cache.prover.build_main_trace(res, cpd, i).download()
        assert got.shape == t["main"].shape, t["kind"]
        assert np.array_equal(got, t["main"]), t["kind"]
    out = pv.prove_next_layer(pv.RecursionInput(traces=traces), ctx, pv.FriRecursionBackend(),
                              pv.ProveNextLayerParams(table_packing=cpd.packing), prep=cache)
    want = L.prove()
    assert out.proof.proof == want
    L.verify(out.proof.proof)
    # resident inputs give the same bytes; proving twice is deterministic
    again = cache.prover.prove_all_tables(res, cpd)
    assert again.proof == want
    res.free()
    cpd.free()
    ctx.close()


def test_no_horner_ops_unscheduled_alu(oracle):
    """Without HornerAcc ops the ALU table is laid out unscheduled (alu_air.rs:367-369,592-599)."""
    from plonky3_recursion_amd import prover as pv
    arrs, L, ctx, cache, traces = setup(oracle, "koala-bear", 6,
                                        dict(log_final_poly_len=1, query_pow_bits=3, num_queries=4), None,
                                        horner_chain_len=0)
    assert not np.any(arrs["alu_prep13"].reshape(-1, 13)[:, 4])
    out = cache.prover.prove_all_tables(traces, cache.circuit_prover_data)
    assert out.proof == L.prove()
    cache.circuit_prover_data.free()
    ctx.close()


def test_row_count_mismatch_is_reported(oracle):
    import plonky3_recursion_amd as p3r
    arrs, L, ctx, cache, traces = setup(oracle, "koala-bear", 6,
                                        dict(log_final_poly_len=1, query_pow_bits=3, num_queries=4), None)
    traces.alu_values = traces.alu_values[:-1]
    with pytest.raises(p3r.P3rError):
        cache.prover.prove_all_tables(traces, cache.circuit_prover_data)
    cache.circuit_prover_data.free()
    ctx.close()


def test_cap_taller_than_a_fri_commit_phase_tree_is_refused(oracle):
    """A FRI commit-phase tree of 4 leaves has no 8-digest cap: prover and oracle both refuse the
    configuration (found by tools/prove_sweep.py; the prover used to read past the tree)."""
    import plonky3_recursion_amd as p3r
    kw = dict(log_blowup=2, max_log_arity=3, cap_height=3, log_final_poly_len=0, commit_pow_bits=0, query_pow_bits=2,
              num_queries=2)
    arrs, L, ctx, cache, traces = setup(oracle, "koala-bear", 5, kw, None)
    with pytest.raises(RuntimeError, match="cap_height"):
        L.prove()
    with pytest.raises(p3r.P3rError, match="cap_height"):
        cache.prover.prove_all_tables(traces, cache.circuit_prover_data)
    cache.circuit_prover_data.free()
    ctx.close()


EDGE_SHAPES = [
    harness_lib.NO_POSEIDON2,
    harness_lib.NO_RECOMPOSE,
    harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE | harness_lib.SINGLE_PUBLIC,
    harness_lib.NO_POSEIDON2 | harness_lib.NO_ALU,
]


@pytest.mark.parametrize("flags", EDGE_SHAPES)
def test_absent_tables_and_dummy_lane_reduction(oracle, flags):
    """Tables without rows are left out of the batch, dummy-only Public / ALU tables use one lane
    (batch_stark_prover.rs:1305-1318, poseidon2.rs:1089-1092, recompose.rs:77-80); the effective
    packing is what the proof wrapper records (:1617-1622)."""
    from plonky3_recursion_amd import prover as pv
    arrs, L, ctx, cache, traces = setup(oracle, "koala-bear", 6,
                                        dict(log_blowup=1, log_final_poly_len=1, query_pow_bits=3, num_queries=4),
                                        dict(public_lanes=2, alu_lanes=3), horner_chain_len=8, flags=flags)
    tables = L.tables()
    cpd = cache.circuit_prover_data
    assert [h for h in cpd.table_heights if h] == [t["main"].shape[0] for t in tables]
    assert [i for i, h in enumerate(cpd.table_heights) if h] == [t["kind_id"] for t in tables]
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    res = pv.ResidentTraces(ctx, cpd, traces)
    for t in tables:
        got = cache.prover.build_main_trace(res, cpd, t["kind_id"]).download()
        assert np.array_equal(got, t["main"]), t["kind"]
    for absent in set(range(5)) - {t["kind_id"] for t in tables}:
        with pytest.raises(pv.P3rError):
            cache.prover.build_main_trace(res, cpd, absent)
    out = cache.prover.prove_all_tables(res, cpd)
    assert out.proof == L.prove()
    L.verify(out.proof)
    by_kind = {t["kind"]: t for t in tables}
    assert out.table_packing.public_lanes == by_kind["public"]["lanes"]
    assert out.table_packing.alu_lanes == by_kind["alu"]["lanes"]
    assert len(out.non_primitives) == len(tables) - 3
    assert len(out.degree_bits) == len(out.preprocessed_widths) == len(tables)
    assert out.preprocessed_widths == tuple(t["prep"].shape[1] for t in tables)
    out.to_postcard()
    res.free()
    cpd.free()
    ctx.close()


@pytest.mark.parametrize("field,log_h,kw,packing", CASES[:3])
def test_prove_then_verify_all_tables_natively(oracle, field, log_h, kw, packing):
    """The reference's own test pattern (prove_all_tables -> verify_all_tables,
    circuit-prover/src/batch_stark_prover/tests.rs) with the product's native verifier; a tampered
    proof and tampered metadata are rejected."""
    import dataclasses
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    arrs, L, ctx, cache, traces = setup(oracle, field, log_h, kw, packing)
    proof = cache.prover.prove_all_tables(traces, cache.circuit_prover_data)
    cache.prover.verify_all_tables(proof)
    canon = cache.prover.prove_all_tables(traces, cache.circuit_prover_data, canonical_field_encoding=True)
    cache.prover.verify_all_tables(canon)
    p3r.verify_all_tables(ctx.cfg, proof)   # free function: no prover object, no GPU work
    bad = bytearray(proof.proof)
    bad[len(bad) // 2] ^= 2
    with pytest.raises(p3r.P3rError):
        cache.prover.verify_all_tables(dataclasses.replace(proof, proof=bytes(bad)))
    with pytest.raises(p3r.P3rError, match="BadHornerPackedSteps"):
        cache.prover.verify_all_tables(dataclasses.replace(
            proof, table_packing=dataclasses.replace(proof.table_packing, horner_packed_steps=1)))
    with pytest.raises(p3r.P3rError):   # a different lane count is a different statement
        cache.prover.verify_all_tables(dataclasses.replace(
            proof, table_packing=dataclasses.replace(proof.table_packing, alu_lanes=proof.table_packing.alu_lanes + 1)))
    cache.circuit_prover_data.free()
    ctx.close()


def test_concurrent_contexts_on_one_gpu(oracle):
    """Several p3r_ctx (own stream, own memory pool) driven from separate host threads at the same
    time: every proof equals the single-threaded one (aggregation leaves sharing a GPU)."""
    import threading
    from plonky3_recursion_amd import prover as pv
    jobs = []
    for i, (field, log_h) in enumerate([("koala-bear", 9), ("baby-bear", 8), ("koala-bear", 10), ("koala-bear", 8)]):
        arrs, L, ctx, cache, traces = setup(oracle, field, log_h, dict(log_final_poly_len=2, query_pow_bits=6, num_queries=8),
                                            None)
        want = cache.prover.prove_all_tables(traces, cache.circuit_prover_data).proof
        assert want == L.prove()
        jobs.append((ctx, cache, traces, want))
    errors = []

    def run(job):
        ctx, cache, traces, want = job
        try:
            for _ in range(12):
                if cache.prover.prove_all_tables(traces, cache.circuit_prover_data).proof != want:
                    errors.append("proof changed under concurrency")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for ctx, cache, traces, want in jobs:
        cache.circuit_prover_data.free()
        ctx.close()


def test_trim_releases_the_pool_and_proofs_stay_identical(oracle):
    """p3r_trim (include/p3r.h): cached device memory goes back to the driver; the next proof
    re-allocates and produces the same bytes."""
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    field = "koala-bear"
    fri = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=2, commit_pow_bits=0, query_pow_bits=4,
               num_queries=6)
    a = harness_lib.generate(field, 10, seed=4, horner_chain_len=20, sponge_chain_len=4, merkle_depth=6)
    ctx = p3r.Context(field=field, **fri)
    tp = p3r.TablePacking().with_fri_params(fri["log_final_poly_len"], fri["log_blowup"])
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    inputs = wl.circuit_inputs_from_arrays(a)
    first = pc.prove(inputs)
    freed = ctx.trim()
    assert freed > 0
    assert ctx.trim() < freed          # nothing (or only the dropped job tables) left to release
    assert pc.prove(inputs) == first
    other = p3r.Context(field=field, **fri)   # a sibling ctx on the same GPU keeps working
    pc2 = p3r.PreparedCircuit(other, wl.circuit_from_arrays(a), tp)
    assert pc2.prove(inputs) == first
    pc2.free()
    other.close()
    pc.free()
    ctx.close()
