"""CPU (no GPU needed): the product's native verifier (`p3r_verify_batch`, host code in
libp3r_hip.so) against proofs made by the CPU oracle's prover - two independently written
implementations of the protocol must agree: it accepts what the oracle proves (both fields, FRI
parameter sets, cap heights, lanes, absent tables, commit-phase proof of work, both field
encodings) and rejects tampered bytes, a wrong preprocessed commitment, wrong AIR shapes and the
oracle's negative cases."""
import numpy as np
import pytest

import harness_lib
import layer_lib

SMALL = dict(horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)


def verify(field, prm, tables, cap, proof, canonical=False, degree_bits=None):
    import plonky3_recursion_amd as p3r
    if degree_bits is None:  # the verifier's own metadata: log2 of every table's (padded) height
        degree_bits = [int(t["main"].shape[0]).bit_length() - 1 for t in tables]
    cfg, keep = p3r.make_config(field, prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, mmcs_arity=prm.mmcs_arity or 2, allow_unpinned_w32_defaults=True)
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"]) for t in tables]
    p3r.verify_batch(cfg, airs, cap, degree_bits, proof, canonical)


CASES = [
    ("koala-bear", 5, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=0, query_pow_bits=3, num_queries=4), None, 0),
    ("koala-bear", 7, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, query_pow_bits=5, num_queries=6), None, 0),
    ("koala-bear", 7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, cap_height=2, query_pow_bits=4, num_queries=5),
     dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3, recompose_lanes=2), 0),
    ("baby-bear", 6, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=4, num_queries=5), None, 0),
    ("baby-bear", 7, dict(log_blowup=1, max_log_arity=3, log_final_poly_len=0, commit_pow_bits=3, query_pow_bits=5, num_queries=4),
     dict(alu_lanes=1, horner_packed_steps=2), 0),
    ("koala-bear", 6, dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4), None,
     harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE | harness_lib.SINGLE_PUBLIC),
    ("koala-bear", 6, dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4),
     dict(alu_lanes=4, horner_packed_steps=5), harness_lib.NO_POSEIDON2 | harness_lib.NO_ALU),
    # the arity-4 MMCS over the width-32 permutation (p3r_config.mmcs_arity = 4): oracle prover, native verifier
    ("koala-bear", 7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=5, mmcs_arity=4), None, 0),
    ("baby-bear", 6, dict(log_blowup=1, max_log_arity=3, log_final_poly_len=0, commit_pow_bits=3, query_pow_bits=3, num_queries=4, mmcs_arity=4),
     dict(alu_lanes=2, horner_packed_steps=3), 0),
    ("koala-bear", 7, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=1, query_pow_bits=3, num_queries=4, mmcs_arity=4), None,
     harness_lib.P2_W32),
]


@pytest.mark.parametrize("field,log_h,kw,packing,flags", CASES)
def test_native_verifier_accepts_oracle_proofs_and_rejects_tampering(oracle, field, log_h, kw, packing, flags):
    import plonky3_recursion_amd as p3r
    arrs = harness_lib.generate(field, log_h, seed=50 + log_h, flags=flags, **SMALL)
    prm = layer_lib.params(**kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=packing)
    tables, cap = L.tables(), L.prep_commit()
    proof = L.prove()
    verify(field, prm, tables, cap, proof)
    verify(field, prm, tables, cap, L.prove(field_encoding=1), canonical=True)
    # every part of the proof is bound: flip one bit at several depths
    for frac in (0.02, 0.3, 0.55, 0.8, 0.97):
        bad = bytearray(proof)
        bad[int(len(bad) * frac)] ^= 1
        with pytest.raises(p3r.P3rError):
            verify(field, prm, tables, cap, bytes(bad))
    with pytest.raises(p3r.P3rError):
        verify(field, prm, tables, cap, proof[:-1])
    with pytest.raises(p3r.P3rError):
        verify(field, prm, tables, cap, proof + b"\x00")
    wrong_cap = cap.copy()
    wrong_cap[0, 0] ^= 1
    with pytest.raises(p3r.P3rError, match="root mismatch|proof of work|quotient"):
        verify(field, prm, tables, wrong_cap, proof)
    # a different AIR shape (lanes) or FRI parameter is not the statement that was proved
    other = [dict(t) for t in tables]
    other[2]["lanes"] += 1
    with pytest.raises(p3r.P3rError):
        verify(field, prm, other, cap, proof)
    prm2 = layer_lib.params(**dict(kw, num_queries=kw["num_queries"] + 1))
    with pytest.raises(p3r.P3rError):
        verify(field, prm2, tables, cap, proof)
    # the trace domains are the verifier's (recursion/src/verifier/batch_stark.rs:793): a proof whose
    # degree_bits differ from the preprocessed metadata is InvalidProofShape, whatever else it holds
    db = [int(t["main"].shape[0]).bit_length() - 1 for t in tables]
    for i in range(len(db)):
        for delta in (1, -1):
            other_db = list(db)
            other_db[i] += delta
            with pytest.raises(p3r.P3rError, match="InvalidProofShape"):
                verify(field, prm, tables, cap, proof, degree_bits=other_db)
    with pytest.raises(p3r.P3rError):
        verify(field, prm, tables, cap, proof, degree_bits=db[:-1])


def corrupt_and_prove(oracle, mutate):
    arrs = harness_lib.generate("koala-bear", 6, seed=3, **SMALL)
    mutate(arrs)
    prm = layer_lib.params(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm)
    return prm, L.tables(), L.prep_commit(), L.prove()


def test_native_verifier_rejects_unsatisfied_trace_and_unbalanced_lookup(oracle):
    import plonky3_recursion_amd as p3r

    def bad_alu(a):
        a["alu_values"][16 * 5 + 12] = (int(a["alu_values"][16 * 5 + 12]) + 1) % 0x7F000001

    def bad_mult(a):
        a["const_prep"][2 * 3] = (int(a["const_prep"][2 * 3]) + 1) % 0x7F000001

    for mutate, why in ((bad_alu, "quotient|lookup"), (bad_mult, "lookup|quotient")):
        prm, tables, cap, proof = corrupt_and_prove(oracle, mutate)
        with pytest.raises(p3r.P3rError, match=why):
            verify("koala-bear", prm, tables, cap, proof)


@pytest.mark.parametrize("canonical", [False, True])
def test_batch_stark_proof_wire_round_trip_and_verify(oracle, canonical):
    """BatchStarkProof -> postcard bytes -> BatchStarkProof (metadata validated,
    batch_stark_prover.rs:666-681) -> verify_all_tables, with the inner BatchProof made by the oracle."""
    import dataclasses
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    field = "koala-bear"
    arrs = harness_lib.generate(field, 6, seed=8, **SMALL)
    prm = layer_lib.params(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
    packing = dict(public_lanes=2, alu_lanes=3, horner_packed_steps=4, recompose_lanes=2)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=packing)
    tables, cap = L.tables(), L.prep_commit()
    inner = L.prove(field_encoding=1 if canonical else 0)
    tp = pv.TablePacking(min_trace_height=layer_lib.min_trace_height(prm), **packing)
    counts = [int(x) for x in arrs["counts"]]
    proof = pv.BatchStarkProof(
        proof=inner, table_packing=tp, rows=tuple(counts[:3]), w_binomial=3,
        non_primitives=(pv.NonPrimitiveTableEntry("poseidon2_perm/koala_bear_d4_w16", tables[3]["main"].shape[0], 1),
                        pv.NonPrimitiveTableEntry("recompose", counts[4], 2)),
        preprocessed_commitment=cap, preprocessed_widths=tuple(t["prep"].shape[1] for t in tables),
        degree_bits=tuple(t["main"].shape[0].bit_length() - 1 for t in tables),
        monty_r=0 if canonical else 1, modulus=0x7F000001)
    wire = proof.to_postcard()
    back = pv.BatchStarkProof.from_postcard(wire, field, canonical_field_encoding=canonical)
    assert back.proof == inner and back.to_postcard() == wire
    assert back.table_packing == tp and back.rows == proof.rows and back.non_primitives == proof.non_primitives
    assert np.array_equal(back.preprocessed_commitment, cap)
    cfg, keep = p3r.make_config(field, prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, allow_unpinned_w32_defaults=True)
    p3r.verify_all_tables(cfg, back)
    # the extension metadata must be the verifier's (batch_stark_prover.rs:1245-1263) and the
    # stark_common degree_bits are what the inner proof has to declare
    with pytest.raises(p3r.P3rError, match="ExtDegreeMismatch"):
        p3r.verify_all_tables(cfg, dataclasses.replace(back, ext_degree=2))
    with pytest.raises(p3r.P3rError, match="BinomialWMismatch"):
        p3r.verify_all_tables(cfg, dataclasses.replace(back, w_binomial=11))
    with pytest.raises(p3r.P3rError, match="QuinticReductionMismatch"):
        p3r.verify_all_tables(cfg, dataclasses.replace(back, alu_quintic_trinomial=True))
    with pytest.raises(p3r.P3rError, match="InvalidProofShape"):
        p3r.verify_all_tables(cfg, dataclasses.replace(back, degree_bits=(back.degree_bits[0] + 1,) + tuple(back.degree_bits[1:])))
    with pytest.raises(p3r.P3rError, match="InvalidProofShape"):
        p3r.verify_all_tables(cfg, dataclasses.replace(back, degree_bits=tuple(back.degree_bits[:-1])))
    # metadata tampering: a zero lane count is refused at decode time, trailing bytes too
    bad_tp = dataclasses.replace(tp, alu_lanes=0)
    with pytest.raises(p3r.P3rError, match="ZeroLanes"):
        pv.BatchStarkProof.from_postcard(dataclasses.replace(proof, table_packing=bad_tp).to_postcard(), field, canonical)
    with pytest.raises(p3r.P3rError, match="trailing"):
        pv.BatchStarkProof.from_postcard(wire + b"\x00", field, canonical)
    with pytest.raises(p3r.P3rError):
        pv.BatchStarkProof.from_postcard(wire[:len(inner) + 3], field, canonical)


def test_native_verifier_parser_survives_mutations(oracle):
    """Random byte edits, truncations and insertions: always a clean rejection (bounded lengths,
    strict option tags and varints), never an acceptance of different bytes."""
    import random
    import plonky3_recursion_amd as p3r
    arrs = harness_lib.generate("koala-bear", 5, seed=8, horner_chain_len=8, sponge_chain_len=3, merkle_depth=3)
    prm = layer_lib.params(log_blowup=1, max_log_arity=2, log_final_poly_len=0, query_pow_bits=2, num_queries=3)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm)
    tables, cap, proof = L.tables(), L.prep_commit(), L.prove()
    verify("koala-bear", prm, tables, cap, proof)
    rng = random.Random(1)
    for _ in range(600):
        b = bytearray(proof)
        k = rng.random()
        if k < 0.4:
            for _ in range(rng.randint(1, 4)):
                b[rng.randrange(len(b))] = rng.randrange(256)
        elif k < 0.6:
            b = b[:rng.randrange(len(b))]
        elif k < 0.8:
            i = rng.randrange(len(b))
            b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 8)))
        else:
            b[rng.randrange(min(len(b), 400))] = 0xFF
        if bytes(b) == proof:
            continue
        with pytest.raises(p3r.P3rError):
            verify("koala-bear", prm, tables, cap, bytes(b))


LAYOUT = [4, 0, 2, 1, 3] + [3, 4, 0, 2, 1] + [0, 2, 3, 1, 6, 7, 4, 5]   # some other order of every struct's fields


@pytest.mark.parametrize("ext_choices,arities,layout", [(1, None, None), (0, [1, 1, 1, 1, 1], None), (1, [1, 1, 1, 1, 1], LAYOUT),
                                                        (0, None, LAYOUT)])
def test_selectable_protocol_details_oracle_and_native_verifier_agree(oracle, ext_choices, arities, layout):
    """The [EXT] switches (include/p3r.h: ext_choices, fri_log_arities): LogUp
    without same-bus packing and an explicit FRI folding schedule.  The oracle proves under the
    switched rules, both verifiers accept under the same rules and reject under the default ones."""
    import plonky3_recursion_amd as p3r
    field, log_h = "koala-bear", 6
    arrs = harness_lib.generate(field, log_h, seed=91, **SMALL)
    kw = dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
    base = layer_lib.params(**kw)
    prm = layer_lib.params(ext_choices=ext_choices, fri_log_arities=arities, proof_layout=layout, **kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    tables, cap = L.tables(), L.prep_commit()
    proof = L.prove()
    L.verify(proof)
    db = [int(t["main"].shape[0]).bit_length() - 1 for t in tables]
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"]) for t in tables]

    def native(ext, ar, lay=layout):
        cfg, keep = p3r.make_config(field, base.log_blowup, base.max_log_arity, base.cap_height, base.log_final_poly_len,
                                    base.commit_pow_bits, base.query_pow_bits, base.num_queries, ext_choices=ext,
                                    fri_log_arities=ar, proof_layout=lay, allow_unpinned_w32_defaults=True)
        p3r.verify_batch(cfg, airs, cap, db, proof)

    native(ext_choices, arities)
    default_proof = layer_lib.OracleLayer(oracle, field, arrs, base).prove()
    assert default_proof != proof
    if layout is not None:
        assert sorted(default_proof) == sorted(proof) or ext_choices or arities   # same content, other order
    with pytest.raises(p3r.P3rError):      # the default rules describe a different proof shape
        native(0, None, None)
    if layout is not None:
        from plonky3_recursion_amd import prover as pv
        n = pv.BatchStarkProof  # the wire parser finds the end of the inner proof under the same layout
        import ctypes as C
        from plonky3_recursion_amd import _lib
        lib = _lib.load()
        buf = (C.c_uint8 * (len(proof) + 3)).from_buffer_copy(proof + b"xyz")
        got, err = C.c_size_t(), C.create_string_buffer(256)
        lay = (C.c_uint8 * 18)(*layout)
        assert lib.p3r_batch_proof_len_layout(0, buf, len(proof) + 3, 0, lay, C.byref(got), err, 256) == 0 and got.value == len(proof)
        bad = (C.c_uint8 * 18)(*([0] * 18))
        assert lib.p3r_batch_proof_len_layout(0, buf, len(proof) + 3, 0, bad, C.byref(got), err, 256) != 0
    if arities is not None:
        with pytest.raises(p3r.P3rError):  # a schedule that skips the final height is refused
            native(ext_choices, [2, 2, 2, 2])
