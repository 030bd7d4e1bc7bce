"""CPU (no GPU): the ZK configuration - HidingFriPcs, `create_config_zk` (recursion/examples/common/mod.rs:511-553) -
on the verifier side.  The acceptance rules are in-tree and normative (recursion/src/verifier/batch_stark.rs:424-428
randomisation presence, :487-490,536 degree bits and chunk count, :623-661 random commitment and round, :701-735 quotient
domains, :855-864 + :1116-1260 and pcs/fri/targets.rs:1076-1130 the opening proof's random opened values); the prover
side is un-vendored and randomised, so there is no byte parity to claim: the checks here are that BOTH verifiers (the
oracle's and the product's native one, two independently written restatements of those rules) accept the oracle
prover's ZK proofs and refuse every violation of a rule."""
import copy

import numpy as np
import pytest

import harness_lib
import layer_lib
import proof_codec

SMALL = dict(horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)
P = {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}


def native_verify(field, prm, tables, cap, proof, degree_bits=None, zk=None, codewords=None):
    import plonky3_recursion_amd as p3r
    zk = prm.zk if zk is None else zk
    if degree_bits is None:   # the preprocessed metadata holds the EXTENDED degree bits (recursion.rs:374)
        degree_bits = [int(t["main"].shape[0]).bit_length() - 1 + zk for t in tables]
    cfg, keep = p3r.make_config(field, prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, mmcs_arity=prm.mmcs_arity or 2,
                                zk=zk, num_random_codewords=prm.num_random_codewords if codewords is None else codewords,
                                challenge_degree=prm.challenge_degree or 4, allow_unpinned_w32_defaults=True)
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"]) for t in tables]
    p3r.verify_batch(cfg, airs, cap, degree_bits, proof)


CASES = [
    # (field, log2 rows, FRI parameters, packing, harness flags, codewords)
    ("koala-bear", 5, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4), None, 0, 2),
    ("koala-bear", 6, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, cap_height=1, query_pow_bits=4, num_queries=5),
     dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3, recompose_lanes=2), 0, 2),
    ("baby-bear", 5, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=0, commit_pow_bits=3, query_pow_bits=3, num_queries=4), None, 0, 2),
    ("baby-bear", 6, dict(log_blowup=3, max_log_arity=1, log_final_poly_len=1, query_pow_bits=3, num_queries=3),
     dict(alu_lanes=1, horner_packed_steps=2), 0, 1),
    ("koala-bear", 5, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4), None,
     harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE | harness_lib.SINGLE_PUBLIC, 3),
    # the arity-4 MMCS under ZK (the reference has not wired `--arity4 --zk`, recursive_aggregation.rs:196-199: nothing
    # in the verifier rules couples the two, and both verifiers here take the combination)
    ("koala-bear", 5, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4, mmcs_arity=4), None, 0, 2),
]


@pytest.mark.parametrize("field,log_h,kw,packing,flags,codewords", CASES)
def test_zk_proofs_are_accepted_by_both_verifiers(oracle, field, log_h, kw, packing, flags, codewords):
    import plonky3_recursion_amd as p3r
    arrs = harness_lib.generate(field, log_h, seed=70 + log_h, flags=flags, **SMALL)
    prm = layer_lib.params(zk=1, num_random_codewords=codewords, zk_seed=11, **kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=packing)
    tables, cap = L.tables(), L.prep_commit()
    proof = L.prove()
    L.verify(proof)
    native_verify(field, prm, tables, cap, proof)
    d = proof_codec.decode(proof, zk=True)
    assert d["_consumed"] == len(proof) and proof_codec.encode(d) == proof
    # the shape the rules prescribe: extended degree bits, doubled chunk counts, a random commitment, one random opened
    # vector of Challenge::DIMENSION values per instance, `codewords` values per (round, matrix, point)
    assert d["degree_bits"] == [int(t["main"].shape[0]).bit_length() for t in tables]
    assert d["commitments"]["random"] is not None
    assert all(o["random"] is not None and len(o["random"]) == 4 for o in d["opened"])
    assert all(len(pt) == codewords for rd in d["opening_proof"]["random_opened_values"] for m in rd for pt in m)
    assert len(d["opening_proof"]["random_opened_values"]) == 5 if any(len(o["permutation_local"]) for o in d["opened"]) else 4
    # the same statement without ZK: half the chunks per instance
    prm0 = layer_lib.params(**kw)
    L0 = layer_lib.OracleLayer(oracle, field, arrs, prm0, packing=packing)
    d0 = proof_codec.decode(L0.prove())
    for o, o0 in zip(d["opened"], d0["opened"]):
        assert len(o["quotient_chunks"]) in (2 * len(o0["quotient_chunks"]), 4 * len(o0["quotient_chunks"]))
    # the preprocessed commitment of a ZK configuration is another commitment (extended domain, codeword columns) ...
    assert not np.array_equal(cap, L0.prep_commit())
    # ... that does not depend on the seed or on the proofs made so far
    prm_b = layer_lib.params(zk=1, num_random_codewords=codewords, zk_seed=12, zk_nonce=5, **kw)
    Lb = layer_lib.OracleLayer(oracle, field, arrs, prm_b, packing=packing)
    assert np.array_equal(cap, Lb.prep_commit())
    # two proofs of one statement differ (the PCS's RNG advances; another seed is another sequence), all are accepted,
    # and the sequence is reproducible
    proof_b = Lb.prove()
    prm_c = layer_lib.params(zk=1, num_random_codewords=codewords, zk_seed=11, zk_nonce=1, **kw)
    proof_c = layer_lib.OracleLayer(oracle, field, arrs, prm_c, packing=packing).prove()
    assert len({proof, proof_b, proof_c}) == 3
    for pf in (proof_b, proof_c):
        L.verify(pf)
        native_verify(field, prm, tables, cap, pf)
    assert L.prove() == proof
    # nothing of the witness-dependent openings repeats between two proofs: the opened trace values are those of
    # different randomised polynomials
    dc = proof_codec.decode(proof_c, zk=True)
    assert all(a["trace_local"] != b["trace_local"] for a, b in zip(d["opened"], dc["opened"]))
    # bit flips anywhere are refused by both
    for frac in (0.03, 0.31, 0.52, 0.77, 0.96):
        bad = bytearray(proof)
        bad[int(len(bad) * frac)] ^= 1
        with pytest.raises(RuntimeError):
            L.verify(bytes(bad))
        with pytest.raises(p3r.P3rError):
            native_verify(field, prm, tables, cap, bytes(bad))


def both_reject(L, field, prm, tables, cap, proof_bytes, match, zk=None):
    import plonky3_recursion_amd as p3r
    with pytest.raises(RuntimeError, match=match):
        L.verify(proof_bytes)
    with pytest.raises(p3r.P3rError, match=match):
        native_verify(field, prm, tables, cap, proof_bytes, zk=zk)


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
def test_every_zk_acceptance_rule_has_a_negative(oracle, field):
    import plonky3_recursion_amd as p3r
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
    arrs = harness_lib.generate(field, 5, seed=91, **SMALL)
    prm = layer_lib.params(zk=1, zk_seed=3, **kw)
    prm0 = layer_lib.params(**kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    L0 = layer_lib.OracleLayer(oracle, field, arrs, prm0)
    tables, cap, cap0 = L.tables(), L.prep_commit(), L0.prep_commit()
    proof, proof0 = L.prove(), L0.prove()
    native_verify(field, prm, tables, cap, proof)
    native_verify(field, prm0, tables, cap0, proof0)
    d = proof_codec.decode(proof, zk=True)

    def mutated(fn, zk=True):
        m = copy.deepcopy(d)
        fn(m)
        return proof_codec.encode(m, zk=zk)

    # a proof of the other PCS type does not even parse (SC::Pcs is a type in the reference) ...
    with pytest.raises(RuntimeError):
        L.verify(proof0)
    with pytest.raises(RuntimeError):
        L0.verify(proof)
    with pytest.raises(p3r.P3rError):
        native_verify(field, prm, tables, cap, proof0)
    with pytest.raises(p3r.P3rError):
        native_verify(field, prm0, tables, cap0, proof)
    # ... and batch_stark.rs:424-428 (RandomizationError) holds for well-formed bytes too: a non-ZK proof dressed in the
    # hiding PCS's proof type (empty random opened values) under the ZK configuration
    d0 = proof_codec.decode(proof0)
    d0["opening_proof"]["random_opened_values"] = []
    both_reject(L, field, prm, tables, cap, proof_codec.encode(d0, zk=True), "RandomizationError")
    # the random commitment / one instance's random opened values missing; a random vector of the wrong length (:506-511)
    both_reject(L, field, prm, tables, cap, mutated(lambda m: m["commitments"].__setitem__("random", None)), "RandomizationError")
    both_reject(L, field, prm, tables, cap, mutated(lambda m: m["opened"][2].__setitem__("random", None)), "RandomizationError")
    both_reject(L, field, prm, tables, cap, mutated(lambda m: m["opened"][1]["random"].pop()), "RandomizationError")
    # under the NON-ZK configuration a random commitment / random opened values are refused the same way
    m0 = copy.deepcopy(proof_codec.decode(proof0))
    m0["commitments"]["random"] = d["commitments"]["random"]
    both_reject(L0, field, prm0, tables, cap0, proof_codec.encode(m0), "RandomizationError", zk=0)
    m0 = copy.deepcopy(proof_codec.decode(proof0))
    m0["opened"][0]["random"] = d["opened"][0]["random"]
    both_reject(L0, field, prm0, tables, cap0, proof_codec.encode(m0), "RandomizationError", zk=0)
    # quotient_degree = 1 << (log_qd + is_zk) (:487-496): the non-ZK chunk count is refused
    both_reject(L, field, prm, tables, cap, mutated(lambda m: m["opened"][2].__setitem__("quotient_chunks", m["opened"][2]["quotient_chunks"][:4])), "chunk")
    # base_db = ext_db - is_zk (:536): degree bits the preprocessed metadata does not hold
    bad = mutated(lambda m: m["degree_bits"].__setitem__(0, m["degree_bits"][0] - 1))
    with pytest.raises(RuntimeError):
        L.verify(bad)
    with pytest.raises(p3r.P3rError, match="InvalidProofShape"):
        native_verify(field, prm, tables, cap, bad)
    with pytest.raises(p3r.P3rError, match="InvalidProofShape"):   # the verifier's metadata is the BASE degree: not this proof's
        native_verify(field, prm, tables, cap, proof, degree_bits=[int(t["main"].shape[0]).bit_length() - 1 for t in tables])
    # the opening proof's random opened values (merge_hiding_random_openings, pcs/fri/targets.rs:1076-1130): rounds,
    # matrices, points must match the commitments; every point carries num_random_codewords values
    rov = lambda m: m["opening_proof"]["random_opened_values"]   # noqa: E731
    both_reject(L, field, prm, tables, cap, mutated(lambda m: rov(m).pop()), "random rounds count")
    both_reject(L, field, prm, tables, cap, mutated(lambda m: rov(m)[1].pop()), "random matrices count")
    both_reject(L, field, prm, tables, cap, mutated(lambda m: rov(m)[1][2].pop()), "random points count")
    both_reject(L, field, prm, tables, cap, mutated(lambda m: rov(m)[2][3][0].pop()), "codeword")
    # they are part of the transcript and of the reduced openings (:855-864, :1116-1260): a changed value is refused
    both_reject(L, field, prm, tables, cap, mutated(lambda m: rov(m)[0][0][0][0].__setitem__(0, (rov(m)[0][0][0][0][0] + 1) % P[field])), ".")
    both_reject(L, field, prm, tables, cap, mutated(lambda m: m["opened"][0]["random"][0].__setitem__(1, (m["opened"][0]["random"][0][1] + 1) % P[field])), ".")
    # a verifier configured with another codeword count does not accept the proof
    with pytest.raises(p3r.P3rError):
        native_verify(field, prm, tables, cap, proof, codewords=3)
    # the random commitment is observed (:623-625): another one changes zeta and everything after it
    both_reject(L, field, prm, tables, cap, mutated(lambda m: m["commitments"]["random"][0].__setitem__(0, (m["commitments"]["random"][0][0] + 1) % P[field])), ".")
    # swapping two quotient chunks (their opening domains coincide, their recomposition weights do not, :701-735)
    def swap(m):
        c = m["opened"][2]["quotient_chunks"]
        c[0], c[1] = c[1], c[0]
    both_reject(L, field, prm, tables, cap, mutated(swap), ".")


def test_zk_parameter_checks(oracle):
    import plonky3_recursion_amd as p3r
    # a quotient of 2^(log_qd + 1) chunks needs log_qd <= log_blowup: degree-3 constraints under ZK have log_qd = 2
    arrs = harness_lib.generate("koala-bear", 5, seed=5, **SMALL)
    prm = layer_lib.params(zk=1, log_blowup=1, max_log_arity=1, log_final_poly_len=0, query_pow_bits=2, num_queries=3)
    with pytest.raises(RuntimeError, match="quotient domain larger than the LDE"):
        layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm).prove()
    with pytest.raises(RuntimeError, match="num_random_codewords"):
        layer_lib.OracleLayer(oracle, "koala-bear", arrs, layer_lib.params(zk=1, num_random_codewords=9)).prove()
    cfg, keep = p3r.make_config("koala-bear", zk=1, num_random_codewords=9, allow_unpinned_w32_defaults=True)
    with pytest.raises(p3r.P3rError, match="num_random_codewords"):
        p3r.verify_batch(cfg, [dict(kind=0)], np.zeros((1, 8), np.uint32), [5], b"\x00")


# circuit degree x table mix x challenge degree: the quintic recursion backend's own tables (D = 5, recompose/coeff) under the
# quartic and the quintic challenge field, and the base proof (D = 1)
DEGREE_MIXES = [(5, harness_lib.RECOMPOSE_COEFF, 1, 4), (5, harness_lib.RECOMPOSE_COEFF, 1, 5), (1, harness_lib.NO_RECOMPOSE, 0, 4),
                (4, 0, 0, 5)]


def native_verify_d(prm, tables, cap, proof, d, coeff, cd):
    import plonky3_recursion_amd as p3r
    cfg, keep = p3r.make_config("koala-bear", prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, ext_degree=d, challenge_degree=cd,
                                zk=prm.zk, num_random_codewords=prm.num_random_codewords, allow_unpinned_w32_defaults=True)
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"],
                 coeff_lookups=coeff if t["kind"] == "recompose" else 0) for t in tables]
    p3r.verify_batch(cfg, airs, cap, [int(t["main"].shape[0]).bit_length() - 1 + prm.zk for t in tables], proof)


@pytest.mark.parametrize("d,flags,coeff,cd", DEGREE_MIXES)
def test_zk_over_the_other_circuit_degrees_and_the_quintic_challenge_field(oracle, d, flags, coeff, cd):
    """`random` opened vectors have Challenge::DIMENSION = 5 entries under the quintic challenge field, the random round
    has 5 + R columns, the masked chunks 5 + R; compact-D1 Poseidon2 rows pack their 17 interactions in triples."""
    import plonky3_recursion_amd as p3r
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
    prm = layer_lib.params(zk=1, zk_seed=21, challenge_degree=cd, **kw)
    arrs = harness_lib.generate("koala-bear", 6, seed=60 + d + cd, flags=flags, ext_degree=d, **SMALL)
    packing = dict(ext_degree=d, recompose_coeff_lookups=coeff)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=packing)
    tables, cap, proof = L.tables(), L.prep_commit(), L.prove()
    L.verify(proof)
    native_verify_d(prm, tables, cap, proof, d, coeff, cd)
    dec = proof_codec.decode(proof, dc=cd, zk=True)
    assert dec["_consumed"] == len(proof)
    assert all(len(o["random"]) == cd and all(len(v) == cd for v in o["random"]) for o in dec["opened"])
    for frac in (0.1, 0.45, 0.9):
        bad = bytearray(proof)
        bad[int(len(bad) * frac)] ^= 1
        with pytest.raises(RuntimeError):
            L.verify(bytes(bad))
        with pytest.raises(p3r.P3rError):
            native_verify_d(prm, tables, cap, bytes(bad), d, coeff, cd)
    prm0 = layer_lib.params(challenge_degree=cd, **kw)
    with pytest.raises(p3r.P3rError):
        native_verify_d(prm0, tables, cap, proof, d, coeff, cd)
