"""GPU: the circuit boundary (prepare -> run on the device -> prove) for circuits of extension degree 1 and 5 -
`CircuitBuilder<F>` and `CircuitBuilder<QuinticTrinomialExtensionField<KoalaBear>>` circuits: the primitive ops, the
hints, Recompose and base-mode Poseidon2 permutations (one witness per state element: sponge chains with their
length tags, Merkle paths with private siblings).  The synthetic generator (its own arithmetic, harness/arith.h)
supplies the op list AND the traces and preprocessed columns a sequential run must produce; the device's run, its
preprocessed commitment and the proof bytes are compared with those and with the oracle proving the generator's
traces."""
import numpy as np
import pytest

import harness_lib
import layer_lib

pytestmark = pytest.mark.gpu

FRI = dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
NO_P2 = harness_lib.NO_POSEIDON2


def setup(oracle, field, ext_degree, log_h, flags, packing=None, **gen):
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    gen.setdefault("horner_chain_len", 16)
    gen.setdefault("sponge_chain_len", 3)
    gen.setdefault("merkle_depth", 5)
    a = harness_lib.generate(field, log_h, seed=57 + log_h, flags=flags, ext_degree=ext_degree, **gen)
    prm = layer_lib.params(**FRI)
    ctx = p3r.Context(field=field, ext_degree=ext_degree, **FRI)
    tp = p3r.TablePacking(**(packing or {})).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(a), p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=tp))
    return a, prm, ctx, cache, wl.circuit_inputs_from_arrays(a, ext_degree)


CASES = [("koala-bear", 5, 7, NO_P2 | harness_lib.NO_RECOMPOSE, None),
         ("koala-bear", 5, 9, NO_P2 | harness_lib.NO_RECOMPOSE, dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3)),
         ("koala-bear", 5, 10, NO_P2 | harness_lib.NO_RECOMPOSE, dict(alu_lanes=4, horner_packed_steps=5)),
         ("koala-bear", 1, 8, NO_P2 | harness_lib.NO_RECOMPOSE, None),
         ("baby-bear", 1, 9, NO_P2 | harness_lib.NO_RECOMPOSE, dict(alu_lanes=1, horner_packed_steps=2)),
         ("koala-bear", 5, 7, NO_P2 | harness_lib.NO_RECOMPOSE | harness_lib.SINGLE_PUBLIC, None),
         # + Recompose ops (D base-field witnesses packed into one element)
         ("koala-bear", 5, 8, NO_P2, dict(recompose_lanes=2)),
         ("baby-bear", 1, 7, NO_P2, None),
         # + base-mode Poseidon2 permutations
         ("koala-bear", 5, 7, harness_lib.NO_RECOMPOSE, None),
         ("koala-bear", 5, 9, 0, dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3)),
         ("koala-bear", 1, 8, 0, None),
         ("baby-bear", 1, 8, harness_lib.NO_RECOMPOSE, dict(alu_lanes=4, horner_packed_steps=6)),
         ("koala-bear", 5, 10, harness_lib.INDEPENDENT_SPONGES, None),
         # Recompose ops of the `recompose/coeff` kind: hint outputs created by the recompose rows, read by sponge inputs
         ("koala-bear", 5, 8, harness_lib.RECOMPOSE_COEFF, None),
         ("koala-bear", 5, 10, harness_lib.RECOMPOSE_COEFF, dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3, recompose_lanes=2)),
         ("baby-bear", 1, 8, harness_lib.RECOMPOSE_COEFF, None),
         ("koala-bear", 5, 7, harness_lib.RECOMPOSE_COEFF | NO_P2, None),
         # both kinds in one circuit: two Recompose tables, `recompose` then `recompose/coeff` (what a verifier circuit
         # of the D = 5 backend holds: plain recomposition in the challenger, the coefficient kind for decomposition links)
         ("koala-bear", 5, 8, harness_lib.RECOMPOSE_BOTH, None),
         ("koala-bear", 5, 10, harness_lib.RECOMPOSE_BOTH, dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3, recompose_lanes=2)),
         ("baby-bear", 1, 8, harness_lib.RECOMPOSE_BOTH, None),
         ("koala-bear", 4, 8, harness_lib.RECOMPOSE_BOTH, None),
         ("baby-bear", 4, 7, harness_lib.RECOMPOSE_BOTH | NO_P2, None),
         # wide levels, long Horner chains (the workgroup scan), deep Merkle paths
         ("koala-bear", 5, 13, 0, None)]


@pytest.mark.parametrize("field,ext_degree,log_h,flags,packing", CASES)
def test_device_runner_for_base_field_and_quintic_circuits(oracle, field, ext_degree, log_h, flags, packing):
    import plonky3_recursion_amd as p3r
    a, prm, ctx, cache, inputs = setup(oracle, field, ext_degree, log_h, flags, packing, horner_chain_len=200 if log_h >= 10 else 16,
                                       merkle_depth=12 if log_h >= 13 else 5)
    pc = cache.prepared_circuit
    assert pc.prepared_on_device           # every circuit degree the runner computes in is prepared on the device (round 4)
    assert [pc.circuit_prover_data.rows[k] for k in ("const", "public", "alu", "poseidon2", "recompose")] == \
        [int(x) for x in a["counts"][:5]]
    res = pc.run(inputs)
    assert np.array_equal(res.download("const_values").reshape(-1), a["const_values"])
    assert np.array_equal(res.download("public_values").reshape(-1), a["public_values"])
    got = res.download("alu_values")
    assert np.array_equal(got.reshape(-1), a["alu_values"]), np.argwhere(got.reshape(-1) != a["alu_values"])[:4]
    assert np.array_equal(res.download("recompose_values").reshape(-1), a["recompose_values"])
    both = bool(flags & harness_lib.RECOMPOSE_BOTH)
    if both:
        assert pc.circuit_prover_data.rows["recompose_coeff"] == int(a["counts"][6]) > 0
        assert np.array_equal(res.download("recompose_coeff_values").reshape(-1), a["recompose_coeff_values"])
    if a["counts"][3]:
        assert np.array_equal(res.download("p2_input_values").reshape(-1), a["p2_inputs"])
        assert np.array_equal(res.download("p2_flags"), a["p2_flags"].reshape(-1, 4)[:, :3])
        assert np.array_equal(res.download("p2_mmcs_index_sum").reshape(-1), a["p2_mmcs_index_sum"])
    # the commitment binds the preprocessed columns derived from the op list: bus roles, multiplicities, indices x D
    coeff = int(bool(flags & harness_lib.RECOMPOSE_COEFF))
    L = layer_lib.OracleLayer(oracle, field, a, prm, packing=dict(packing or {}, ext_degree=ext_degree, recompose_coeff_lookups=coeff))
    assert np.array_equal(pc.circuit_prover_data.preprocessed_commitment, L.prep_commit())
    out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=inputs), ctx, p3r.FriRecursionBackend(),
                               p3r.ProveNextLayerParams(table_packing=pc.packing), prep=cache)
    proof = L.prove()
    assert out.proof.proof == proof and pc.prove(inputs) == proof
    assert out.proof.ext_degree == ext_degree and out.proof.alu_quintic_trinomial == (ext_degree == 5)
    if both:
        assert [e.op_type for e in out.proof.non_primitives[-2:]] == ["recompose", "recompose/coeff"]
        assert [e.rows for e in out.proof.non_primitives[-2:]] == [int(a["counts"][4]), int(a["counts"][6])]
        back = p3r.BatchStarkProof.from_postcard(out.proof.to_postcard(), field)
        assert back.to_postcard() == out.proof.to_postcard()
        cache.prover.verify_all_tables(back)
    elif a["counts"][4]:
        assert out.proof.non_primitives[-1].op_type == ("recompose/coeff" if coeff else "recompose")
    cache.prover.verify_all_tables(out.proof)
    res.free()
    pc.free()
    ctx.close()


def test_quintic_division_by_zero_and_conflict_are_reported(oracle):
    """The runner's error paths over Fp5: a backward Mul through a zero divisor, and a public input that contradicts
    what an op computes (WitnessConflict)."""
    import plonky3_recursion_amd as p3r
    a, prm, ctx, cache, inputs = setup(oracle, "koala-bear", 5, 7, NO_P2 | harness_lib.NO_RECOMPOSE)
    pc = cache.prepared_circuit
    pub = inputs.public_values.copy()
    pub[:, 4] = (pub[:, 4].astype(np.uint64) + 1) % 0x7F000001     # the top coefficient of every public input
    bad = p3r.CircuitInputs(public_values=pub, private_values=inputs.private_values)
    try:
        res = pc.run(bad)
    except p3r.P3rError as e:
        assert "WitnessConflict" in str(e) or "DivisionByZero" in str(e)
    else:
        # no op re-derives a public input in this draw: the run goes through, with different traces
        assert not np.array_equal(res.download("alu_values").reshape(-1), a["alu_values"])
        res.free()
    pc.free()
    ctx.close()


def test_layouts_of_the_other_degree_are_refused(oracle):
    """A D = 4 op list (four-limb permutations, four-coefficient constants) under a D = 5 context, and a sponge row
    that feeds its capacity from a witness (NonPrimitiveOpLayoutMismatch, executor.rs:712-725)."""
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    a4 = harness_lib.generate("koala-bear", 6, seed=3, horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)
    ctx = p3r.Context(field="koala-bear", ext_degree=5, **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    with pytest.raises(p3r.P3rError):
        p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a4), tp)
    a5 = harness_lib.generate("koala-bear", 6, seed=3, horner_chain_len=12, sponge_chain_len=3, merkle_depth=4,
                              flags=harness_lib.NO_RECOMPOSE, ext_degree=5)
    ops = a5["ops"].reshape(-1, 8)
    r = next(i for i in range(len(ops)) if ops[i, 0] == 9 and not (ops[i, 5] & 2))      # a sponge permutation
    a5["ext"][ops[r, 6] + 12] = 0                                                          # capacity slot 12 <- witness 0
    with pytest.raises(p3r.P3rError, match="capacity input slots must be empty"):
        p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a5), tp)
    ctx.close()
