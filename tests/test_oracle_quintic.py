"""CPU: circuits over the KoalaBear quintic trinomial extension (D = 5, x^5 + x^2 - 1) through the oracle.
The reference proves such circuits under the D = 4 STARK configuration - the circuit field only changes the
width of the witness tuples on the bus (D + 1), the table widths and the ALU multiplication rule
(circuit-prover/src/batch_stark_prover/tests.rs:844-1029, air/alu_air.rs:735-760)."""
import numpy as np
import pytest

import harness_lib
import layer_lib

P = 0x7F000001
PRIMITIVE = harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE


def quintic_mul(x, y):
    t = [0] * 9
    for i in range(5):
        for j in range(5):
            t[i + j] = (t[i + j] + int(x[i]) * int(y[j])) % P
    for k in range(8, 4, -1):   # x^k = x^(k-5) - x^(k-3)
        t[k - 5] = (t[k - 5] + t[k]) % P
        t[k - 3] = (t[k - 3] - t[k]) % P
    return t[:5]


def layer(oracle, log_h, seed, prm, packing=None, **kw):
    arrs = harness_lib.generate("koala-bear", log_h, seed=seed, horner_chain_len=kw.pop("horner_chain_len", 12),
                                flags=PRIMITIVE, ext_degree=5, **kw)
    packing = dict(packing or {}, ext_degree=5)
    return arrs, layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=packing)


def test_generator_values_follow_the_trinomial_rule():
    arrs = harness_lib.generate("koala-bear", 5, seed=1, horner_chain_len=6, flags=PRIMITIVE, ext_degree=5)
    v = arrs["alu_values"].reshape(-1, 20)
    kinds = arrs["alu_prep13"].reshape(-1, 13)
    seen = 0
    for row, k in zip(v, kinds):
        a, b, c, out = row[0:5], row[5:10], row[10:15], row[15:20]
        if not (k[1] or k[2] or k[3] or k[4]):      # Mul: a * b = out
            assert quintic_mul(a, b) == [int(x) for x in out]
            seen += 1
    assert seen > 10
    # x^5 = 1 - x^2 on the generator's arithmetic via a known product: (x^4) * (x) = 1 - x^2
    assert quintic_mul([0, 0, 0, 0, 1], [0, 1, 0, 0, 0]) == [1, 0, P - 1, 0, 0]


@pytest.mark.parametrize("log_h,kw", [
    (5, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=0)),
    (7, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2)),
    (7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, cap_height=2)),
])
def test_prove_verify_roundtrip(oracle, log_h, kw):
    prm = layer_lib.params(query_pow_bits=3, num_queries=5, **kw)
    arrs, L = layer(oracle, log_h, log_h, prm)
    pf = L.prove()
    L.verify(pf)
    assert L.prove() == pf
    for pos in range(7, len(pf), max(len(pf) // 40, 1)):
        bad = bytearray(pf)
        bad[pos] ^= 1
        with pytest.raises(RuntimeError):
            L.verify(bytes(bad))


def test_table_shapes(oracle):
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=3)
    arrs, L = layer(oracle, 6, 2, prm)
    t = {x["kind"]: x for x in L.tables()}
    assert set(t) == {"const", "public", "alu"}
    # D = 5: Const / Public carry 5 value columns per lane; ALU 4 operands x 5 per lane + (1 + 6 + 1) x 5 for the
    # packed Horner steps (alu_columns.rs:52-127)
    assert t["const"]["main"].shape[1] == 5 and t["const"]["prep"].shape[1] == 2
    assert t["public"]["main"].shape[1] == 5
    assert t["alu"]["main"].shape[1] == 3 * 20 + (1 + 6 + 1) * 5
    assert t["alu"]["prep"].shape[1] == 3 * 13 + 7 * 3
    # witness indices are scaled by D on the bus (circuit.rs:237-510 with D = 5)
    assert (arrs["const_prep"].reshape(-1, 2)[:, 1] % 5 == 0).all()


@pytest.mark.parametrize("packing", [dict(alu_lanes=1, horner_packed_steps=2), dict(alu_lanes=2, horner_packed_steps=3, public_lanes=2)])
def test_lane_and_pack_variants(oracle, packing):
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=3)
    _, L = layer(oracle, 6, 3, prm, packing=packing, horner_chain_len=17)
    L.verify(L.prove())


def test_unsatisfied_trace_is_rejected(oracle):
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    arrs = harness_lib.generate("koala-bear", 6, seed=4, horner_chain_len=12, flags=PRIMITIVE, ext_degree=5)
    # the top coefficient of an operand: invisible to a D = 4 rule
    arrs["alu_values"][4] = (int(arrs["alu_values"][4]) + 1) % P
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(ext_degree=5))
    with pytest.raises(RuntimeError, match="constraints do not match|final polynomial|terminals"):
        L.verify(L.prove())


def test_degree_four_rule_does_not_verify_a_quintic_layer(oracle):
    """The same proof checked as a D = 4 statement is rejected (the verifier's AIR list carries D)."""
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=3)
    arrs, L = layer(oracle, 6, 5, prm)
    pf = L.prove()
    airs5 = [dict(kind=x["kind_id"], lanes=x["lanes"], horner_packed_steps=x["horner_k"], ext_degree=5) for x in L.tables()]
    layer_lib.oracle_verify_statement(oracle, "koala-bear", prm, airs5, L.prep_commit(), pf)
    airs4 = [dict(a, ext_degree=4) for a in airs5]
    with pytest.raises(RuntimeError):
        layer_lib.oracle_verify_statement(oracle, "koala-bear", prm, airs4, L.prep_commit(), pf)


def test_refused_shapes(oracle):
    with pytest.raises(RuntimeError, match="ext_degree"):
        harness_lib.generate("baby-bear", 5, flags=PRIMITIVE, ext_degree=5)


# ---------------------------------------------------------------- the product's native verifier (host code, no GPU)
def native_verify(prm, tables, cap, proof, ext_degree=5, field="koala-bear", canonical=False):
    import plonky3_recursion_amd as p3r
    degree_bits = [int(t["main"].shape[0]).bit_length() - 1 for t in tables]
    cfg, keep = p3r.make_config(field, prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, ext_degree=ext_degree)
    airs = [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"]) for t in tables]
    p3r.verify_batch(cfg, airs, cap, degree_bits, proof, canonical)


@pytest.mark.parametrize("log_h,kw,packing", [
    (5, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=0, query_pow_bits=3, num_queries=4), None),
    (7, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, query_pow_bits=5, num_queries=6), None),
    (7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, cap_height=2, query_pow_bits=4, num_queries=5),
     dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3)),
    (6, dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, commit_pow_bits=3, query_pow_bits=3, num_queries=4),
     dict(alu_lanes=4, horner_packed_steps=6)),
])
def test_native_verifier_accepts_quintic_oracle_proofs(oracle, log_h, kw, packing):
    import plonky3_recursion_amd as p3r
    prm = layer_lib.params(**kw)
    arrs, L = layer(oracle, log_h, 70 + log_h, prm, packing=packing, horner_chain_len=17)
    tables, cap = L.tables(), L.prep_commit()
    proof = L.prove()
    native_verify(prm, tables, cap, proof)
    native_verify(prm, tables, cap, L.prove(field_encoding=1), canonical=True)
    for frac in (0.02, 0.3, 0.55, 0.8, 0.97):
        bad = bytearray(proof)
        bad[int(len(bad) * frac)] ^= 1
        with pytest.raises(p3r.P3rError):
            native_verify(prm, tables, cap, bytes(bad))
    # the statement carries the circuit field: the D = 4 verifier rejects (table widths differ), and D = 5 is
    # KoalaBear's
    with pytest.raises(p3r.P3rError):
        native_verify(prm, tables, cap, proof, ext_degree=4)
    with pytest.raises(p3r.P3rError, match="UnsupportedDegree"):
        native_verify(prm, tables, cap, proof, field="baby-bear")


def test_native_verifier_rejects_unsatisfied_quintic_trace(oracle):
    import plonky3_recursion_amd as p3r
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    arrs = harness_lib.generate("koala-bear", 6, seed=4, horner_chain_len=12, flags=PRIMITIVE, ext_degree=5)
    v = arrs["alu_values"].reshape(-1, 20)
    kinds = arrs["alu_prep13"].reshape(-1, 13)
    mul = next(i for i, k in enumerate(kinds) if not (k[1] or k[2] or k[3] or k[4]) and v[i, 4] and v[i, 9])
    # out = a * b under the BINOMIAL-like wrap x^5 = 1 (no - x^2 term): a product a D = 5 rule must refuse
    a, b = v[mul, 0:5], v[mul, 5:10]
    t = [0] * 9
    for i in range(5):
        for j in range(5):
            t[i + j] = (t[i + j] + int(a[i]) * int(b[j])) % P
    v[mul, 15:20] = [(t[k] + (t[k + 5] if k + 5 < 9 else 0)) % P for k in range(5)]
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(ext_degree=5))
    with pytest.raises(p3r.P3rError, match="quotient|final polynomial|lookup sum"):
        native_verify(prm, L.tables(), L.prep_commit(), L.prove())


def test_quintic_metadata_on_the_wire(oracle):
    """BatchStarkProof of a D = 5 layer: ext_degree 5, no binomial W, the trinomial flag
    (batch_stark_prover.rs:610-636, 1245-1263)."""
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd.prover import BatchStarkProof, TablePacking, verify_all_tables
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=3)
    arrs, L = layer(oracle, 6, 9, prm)
    tables, cap = L.tables(), L.prep_commit()
    tp = TablePacking(public_lanes=1, alu_lanes=3, horner_packed_steps=4, min_trace_height=layer_lib.min_trace_height(prm))
    c = arrs["counts"]
    bsp = BatchStarkProof(proof=L.prove(), table_packing=tp, rows=(int(c[0]), int(c[1]), int(c[2])), ext_degree=5,
                          w_binomial=None, alu_quintic_trinomial=True, preprocessed_commitment=cap,
                          preprocessed_widths=tuple(t["prep"].shape[1] for t in tables),
                          degree_bits=tuple(int(t["main"].shape[0]).bit_length() - 1 for t in tables),
                          monty_r=1, modulus=P)
    wire = bsp.to_postcard()
    back = BatchStarkProof.from_postcard(wire, "koala-bear")
    assert back.ext_degree == 5 and back.w_binomial is None and back.alu_quintic_trinomial
    assert back.to_postcard() == wire
    cfg5, keep5 = p3r.make_config("koala-bear", prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                  prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, ext_degree=5)
    verify_all_tables(cfg5, back)
    cfg4, keep4 = p3r.make_config("koala-bear", prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                  prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries)
    with pytest.raises(p3r.P3rError, match="ExtDegreeMismatch"):
        verify_all_tables(cfg4, back)
    import dataclasses
    with pytest.raises(p3r.P3rError, match="BinomialWMismatch"):
        verify_all_tables(cfg5, dataclasses.replace(back, w_binomial=3))
    with pytest.raises(p3r.P3rError, match="QuinticReductionMismatch"):
        verify_all_tables(cfg5, dataclasses.replace(back, alu_quintic_trinomial=False))


# ---------------------------------------------------------------- compact-D1 Poseidon2 table of a D = 5 circuit
D1 = dict(horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)


def layer_d1(oracle, log_h, seed, prm, packing=None, mutate=None):
    arrs = harness_lib.generate("koala-bear", log_h, seed=seed, flags=harness_lib.NO_RECOMPOSE, ext_degree=5, **D1)
    if mutate:
        mutate(arrs)
    return arrs, layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(packing or {}, ext_degree=5))


@pytest.mark.parametrize("log_h,kw", [
    (5, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=0)),
    (7, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2)),
    (7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, cap_height=2)),
])
def test_compact_d1_poseidon2_roundtrip(oracle, log_h, kw):
    prm = layer_lib.params(query_pow_bits=3, num_queries=5, **kw)
    arrs, L = layer_d1(oracle, log_h, log_h, prm)
    t = {x["kind"]: x for x in L.tables()}
    # KoalaBearD1Width16: the same 166 main columns as the D = 4 table, 62 preprocessed ones
    # (poseidon-circuit-cols/src/preprocessed.rs:137-145: 26 header + 16 + 8 + 8 + 4)
    assert t["poseidon2"]["main"].shape[1] == 166 and t["poseidon2"]["prep"].shape[1] == 62
    n = int(arrs["counts"][3])
    prep = t["poseidon2"]["prep"]
    assert (prep[:n, 26:42] % 5 == 0).all()                       # witness indices scaled by D = 5
    assert np.array_equal(prep[:n, 8], arrs["p2_absorb_len"]) and (log_h < 7 or arrs["p2_absorb_len"].any())
    if prep.shape[0] > n:
        assert prep[n, 60] == 1 and not prep[n + 1:].any()        # first padding row: chain boundary
    pf = L.prove()
    L.verify(pf)
    assert L.prove() == pf
    for pos in range(7, len(pf), max(len(pf) // 40, 1)):
        bad = bytearray(pf)
        bad[pos] ^= 1
        with pytest.raises(RuntimeError):
            L.verify(bytes(bad))
    # the product's native verifier agrees
    native_verify(prm, L.tables(), L.prep_commit(), pf)


def sponge_continuation_rows(arrs):
    fl = arrs["p2_flags"].reshape(-1, 4)
    return [r for r in range(1, len(fl)) if not fl[r, 0] and not fl[r, 1]]


@pytest.mark.parametrize("what", ["capacity", "rate", "tag", "merkle", "accumulator", "bus"])
def test_compact_d1_poseidon2_rejects(oracle, what):
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)

    def mutate(arrs):
        fl = arrs["p2_flags"].reshape(-1, 4)
        inp = arrs["p2_inputs"].reshape(-1, 16)
        ctl = arrs["p2_in_ctl"].reshape(-1, 16)
        if what == "capacity":      # a chained capacity element (never witness-fed)
            r = sponge_continuation_rows(arrs)[0]
            inp[r, 12] = (int(inp[r, 12]) + 1) % P
        elif what == "rate":        # a chained rate element
            r, i = next((r, i) for r in sponge_continuation_rows(arrs) for i in range(8) if not ctl[r, i])
            inp[r, i] = (int(inp[r, i]) + 1) % P
        elif what == "tag":         # the length tag bound into the first capacity element
            # (row 0 is nobody's "next" row under when_transition: its tag is not bound)
            r = next(r for r in range(1, len(fl)) if not fl[r, 1] and arrs["p2_absorb_len"][r])
            arrs["p2_absorb_len"][r] = int(arrs["p2_absorb_len"][r]) + 1
        elif what == "merkle":      # the running digest of a Merkle continuation row
            r = next(r for r in range(1, len(fl)) if fl[r, 1] and not fl[r, 0])
            side = 8 if fl[r, 2] else 0
            inp[r, side + 3] = (int(inp[r, side + 3]) + 1) % P
        elif what == "accumulator":  # the public witness the path's mmcs_index_sum is sent to
            r = next(r for r in range(len(fl)) if fl[r, 3])
            w = int(arrs["p2_mmcs_index_sum_idx"][r])
            pos = int(np.nonzero(arrs["public_prep"].reshape(-1, 2)[:, 1] == 5 * w)[0][0])
            pv = arrs["public_values"].reshape(-1, 5)
            assert pv[pos, 0] == arrs["p2_mmcs_index_sum"][r]
            pv[pos, 0] = (int(pv[pos, 0]) + 1) % P
        else:                        # an output multiplicity
            oc = arrs["p2_out_ctl"]
            k = next(k for k in range(len(oc)) if 0 < oc[k] < P - 1)
            oc[k] = int(oc[k]) + 1

    arrs, L = layer_d1(oracle, 6, 21, prm, mutate=mutate)
    with pytest.raises(RuntimeError, match="constraints do not match|final polynomial|terminals do not sum"):
        L.verify(L.prove())


# ---------------------------------------------------------------- Recompose over D coefficients, "recompose/coeff"
@pytest.mark.parametrize("ext_degree,coeff", [(5, 0), (5, 1), (4, 1)])
def test_recompose_variants_roundtrip_and_native_verifier(oracle, ext_degree, coeff):
    """The D = 5 backend's table mix (backend/fri.rs:741-852): Const / Public / ALU / compact-D1 Poseidon2 /
    Recompose with coefficient lookups; and the coefficient variant under D = 4 (what a D = 4 circuit with a D1
    permutation registers, fri.rs:693-721)."""
    import plonky3_recursion_amd as p3r
    prm = layer_lib.params(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=3, num_queries=5)
    arrs = harness_lib.generate("koala-bear", 7, seed=31, flags=harness_lib.RECOMPOSE_COEFF if coeff else 0,
                                ext_degree=ext_degree, **D1)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm,
                              packing=dict(ext_degree=ext_degree, recompose_coeff_lookups=coeff, recompose_lanes=2))
    tables = L.tables()
    t = {x["kind"]: x for x in tables}
    assert set(t) == {"const", "public", "alu", "poseidon2", "recompose"}
    d = ext_degree
    assert t["recompose"]["main"].shape[1] == 2 * d and t["recompose"]["prep"].shape[1] == 2 * (2 + (2 * d if coeff else 0))
    if coeff:
        assert arrs["recompose_prep"].reshape(-1, 2 + 2 * d)[:, 3::2].any()     # some coefficient multiplicities
    pf = L.prove()
    L.verify(pf)
    cfg, keep = p3r.make_config("koala-bear", prm.log_blowup, prm.max_log_arity, prm.cap_height, prm.log_final_poly_len,
                                prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, ext_degree=ext_degree)
    airs = [dict(kind=x["kind_id"], lanes=x["lanes"], horner_packed_steps=x["horner_k"],
                 coeff_lookups=coeff if x["kind"] == "recompose" else 0) for x in tables]
    db = [int(x["main"].shape[0]).bit_length() - 1 for x in tables]
    p3r.verify_batch(cfg, airs, L.prep_commit(), db, pf)
    for frac in (0.1, 0.5, 0.9):
        bad = bytearray(pf)
        bad[int(len(bad) * frac)] ^= 1
        with pytest.raises(p3r.P3rError):
            p3r.verify_batch(cfg, airs, L.prep_commit(), db, bytes(bad))
    if coeff:
        # without the coefficient tuples it is another statement
        plain = [dict(a, coeff_lookups=0) for a in airs]
        with pytest.raises(p3r.P3rError):
            p3r.verify_batch(cfg, plain, L.prep_commit(), db, pf)


def test_recompose_coeff_unbalanced_is_rejected(oracle):
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    arrs = harness_lib.generate("koala-bear", 6, seed=33, flags=harness_lib.RECOMPOSE_COEFF, ext_degree=5, **D1)
    rp = arrs["recompose_prep"].reshape(-1, 12)
    r, i = next((r, i) for r in range(len(rp)) for i in range(5) if rp[r, 3 + 2 * i])
    rp[r, 3 + 2 * i] += 1
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(ext_degree=5, recompose_coeff_lookups=1))
    with pytest.raises(RuntimeError, match="terminals do not sum"):
        L.verify(L.prove())
