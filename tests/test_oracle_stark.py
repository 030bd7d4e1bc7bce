"""CPU: the oracle's batch-STARK prover against its own verifier (a restatement of the in-tree
circuit verifier, recursion/src/verifier/batch_stark.rs + pcs/fri/verifier.rs) on synthetic
recursion layers; prove -> verify round trips and the reference's negative-test patterns
(tampered proof bytes, unsatisfied trace, unbalanced lookup: circuit-prover/src/batch_stark_prover/tests.rs,
recursion/tests/test_lookups.rs)."""
import numpy as np
import pytest

import harness_lib
import layer_lib

SMALL = dict(horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
@pytest.mark.parametrize("log_h,kw", [
    (5, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=0)),
    (7, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2)),
    (7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, cap_height=2)),
])
def test_prove_verify_roundtrip(oracle, field, log_h, kw):
    arrs = harness_lib.generate(field, log_h, seed=log_h, **SMALL)
    prm = layer_lib.params(query_pow_bits=3, num_queries=5, **kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    pf = L.prove()
    L.verify(pf)
    assert L.prove() == pf  # deterministic (smallest PoW witness)
    # canonical field encoding round trip
    L.verify(L.prove(field_encoding=1), field_encoding=1)
    # every 97th byte flipped must be rejected
    for pos in range(7, len(pf), max(len(pf) // 40, 1)):
        bad = bytearray(pf)
        bad[pos] ^= 1
        with pytest.raises(RuntimeError):
            L.verify(bytes(bad))


def test_table_shapes_follow_reference_formulas(oracle):
    arrs = harness_lib.generate("koala-bear", 6, **SMALL)
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=3)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm)
    t = {x["kind"]: x for x in L.tables()}
    # SURVEY.md appendix B / shape_golden.rs: D=4, alu_lanes=3, K=4
    assert t["const"]["main"].shape[1] == 4 and t["const"]["prep"].shape[1] == 2
    assert t["alu"]["main"].shape[1] == 3 * 16 + (1 + 6 + 1) * 4 == 80
    assert t["alu"]["prep"].shape[1] == 3 * 13 + 7 * 3 == 60
    assert t["poseidon2"]["main"].shape[1] == 166 and t["poseidon2"]["prep"].shape[1] == 24
    assert t["recompose"]["main"].shape[1] == 4 and t["recompose"]["prep"].shape[1] == 2
    mh = layer_lib.min_trace_height(prm)  # packing.rs:100-106
    assert all(x["main"].shape[0] >= mh and x["main"].shape[0] == x["prep"].shape[0] for x in L.tables())
    # preprocessed padding rule of the Poseidon2 table (air.rs:644-646)
    n_p2 = int(arrs["counts"][3])
    prep = t["poseidon2"]["prep"]
    if prep.shape[0] > n_p2:
        assert prep[n_p2, 22] == 1 and not prep[n_p2 + 1:].any()


@pytest.mark.parametrize("packing", [dict(alu_lanes=1, horner_packed_steps=2), dict(alu_lanes=2, horner_packed_steps=3, public_lanes=2),
                                     dict(alu_lanes=4, horner_packed_steps=6, recompose_lanes=2)])
def test_lane_and_pack_variants(oracle, packing):
    arrs = harness_lib.generate("koala-bear", 6, seed=3, horner_chain_len=17, sponge_chain_len=3, merkle_depth=4)
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=3)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=packing)
    L.verify(L.prove())


def test_unsatisfied_trace_is_rejected(oracle):
    arrs = harness_lib.generate("koala-bear", 6, **SMALL)
    arrs["alu_values"][5] = (int(arrs["alu_values"][5]) + 1) % 0x7F000001
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm)
    with pytest.raises(RuntimeError, match="constraints do not match|final polynomial"):
        L.verify(L.prove())


def test_unbalanced_lookup_is_rejected(oracle):
    arrs = harness_lib.generate("koala-bear", 6, **SMALL)
    arrs["const_prep"][2] = int(arrs["const_prep"][2]) + 1  # one extra read that nobody performs
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm)
    with pytest.raises(RuntimeError, match="terminals do not sum to zero"):
        L.verify(L.prove())


def test_poseidon2_chain_break_is_rejected(oracle):
    arrs = harness_lib.generate("koala-bear", 6, **SMALL)
    fl = arrs["p2_flags"].reshape(-1, 4)
    # find a sponge continuation row and corrupt a chained (non-CTL) capacity input
    rows = [r for r in range(1, len(fl)) if not fl[r, 0] and not fl[r, 1]]
    assert rows
    arrs["p2_inputs"][rows[0] * 16 + 12] = (int(arrs["p2_inputs"][rows[0] * 16 + 12]) + 1) % 0x7F000001
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm)
    with pytest.raises(RuntimeError):
        L.verify(L.prove())


@pytest.mark.parametrize("lanes,want", [(1, (28, 20)), (2, (44, 33))])
def test_alu_widths_match_reference_shape_goldens(oracle, lanes, want):
    """The only literal goldens in the reference: circuit-prover/src/air/shape_golden.rs:48-61
    (D=4, default horner pack K=2): (main_width, preprocessed_width) = (28, 20) / (44, 33)."""
    arrs = harness_lib.generate("koala-bear", 5, seed=2, horner_chain_len=6, sponge_chain_len=2, merkle_depth=3)
    prm = layer_lib.params(log_blowup=1, max_log_arity=1, log_final_poly_len=0, query_pow_bits=1, num_queries=2)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(alu_lanes=lanes, horner_packed_steps=2))
    alu = [t for t in L.tables() if t["kind"] == "alu"][0]
    assert (alu["main"].shape[1], alu["prep"].shape[1]) == want
    const, public = L.tables()[0], L.tables()[1]
    assert (const["main"].shape[1], const["prep"].shape[1]) == (4, 2)     # shape_golden.rs: D4 const/public
    assert (public["main"].shape[1], public["prep"].shape[1]) == (4, 2)
    L.verify(L.prove())


def test_field_moduli_match_reference_constants():
    # circuit-prover/src/batch_stark_prover.rs:76-78, tests.rs:694-733
    assert oracle_lib_moduli() == {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}
    assert 0x7F000001 == 2**31 - 2**24 + 1 and 0x78000001 == 2**31 - 2**27 + 1


def oracle_lib_moduli():
    import oracle_lib
    return dict(oracle_lib.MODULUS)


EDGE_SHAPES = [
    (harness_lib.NO_POSEIDON2, ["const", "public", "alu", "recompose"]),
    (harness_lib.NO_RECOMPOSE, ["const", "public", "alu", "poseidon2"]),
    (harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE | harness_lib.SINGLE_PUBLIC, ["const", "public", "alu"]),
    (harness_lib.NO_POSEIDON2 | harness_lib.NO_ALU, ["const", "public", "alu", "recompose"]),
]


@pytest.mark.parametrize("flags,kinds", EDGE_SHAPES)
def test_absent_tables_and_dummy_lane_reduction(oracle, flags, kinds):
    """Non-primitive tables without rows are left out of the batch (poseidon2.rs:1089-1092,
    recompose.rs:77-80); Public / ALU tables holding at most the dummy op use one lane
    (batch_stark_prover.rs:1305-1318)."""
    arrs = harness_lib.generate("koala-bear", 6, seed=9, horner_chain_len=8, sponge_chain_len=3, merkle_depth=4,
                                flags=flags)
    prm = layer_lib.params(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(public_lanes=2, alu_lanes=3))
    tables = L.tables()
    assert [t["kind"] for t in tables] == kinds
    by_kind = {t["kind"]: t for t in tables}
    assert by_kind["public"]["lanes"] == (1 if flags & harness_lib.SINGLE_PUBLIC else 2)
    assert by_kind["alu"]["lanes"] == (1 if flags & harness_lib.NO_ALU else 3)
    proof = L.prove()
    L.verify(proof)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    with pytest.raises(RuntimeError):
        L.verify(bytes(bad))


@pytest.mark.parametrize("kw,msg", [
    # a FRI commit-phase tree of 4 leaves has no 8-digest cap
    (dict(log_blowup=2, max_log_arity=3, cap_height=3, log_final_poly_len=0, query_pow_bits=2, num_queries=2), "cap_height"),
    # an explicit folding step above max_log_arity: the verifier's bound on a step's arity
    (dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=1, num_queries=2,
          fri_log_arities=[1, 1, 3, 1, 1, 1, 1, 1]), "fri_log_arities"),
])
def test_configurations_without_a_proof_are_refused(oracle, kw, msg):
    """Found by tools/prove_sweep.py: these used to yield bytes the verifier then rejected."""
    arrs = harness_lib.generate("koala-bear", 5, seed=11, horner_chain_len=8, sponge_chain_len=2, merkle_depth=3)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, layer_lib.params(**kw))
    with pytest.raises(RuntimeError, match=msg):
        L.prove()
