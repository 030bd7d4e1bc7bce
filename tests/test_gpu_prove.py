"""GPU parity for the full prove_batch: proof BYTES from the HIP path equal the CPU oracle's on
the same synthetic recursion layer, and the oracle's verifier (a restatement of the in-tree
circuit verifier) accepts them."""
import numpy as np
import pytest

import harness_lib
import layer_lib

pytestmark = pytest.mark.gpu


def make_ctx(field, prm):
    import plonky3_recursion_amd as p3r
    return p3r.Context(field=field, log_blowup=prm.log_blowup, max_log_arity=prm.max_log_arity,
                       cap_height=prm.cap_height, log_final_poly_len=prm.log_final_poly_len,
                       commit_pow_bits=prm.commit_pow_bits, query_pow_bits=prm.query_pow_bits,
                       num_queries=prm.num_queries)


def airs_of(tables):
    return [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"], coeff_lookups=0)
            for t in tables]


CASES = [
    ("koala-bear", 5, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=0, query_pow_bits=3, num_queries=4)),
    ("koala-bear", 7, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, query_pow_bits=5, num_queries=6)),
    ("koala-bear", 8, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=3, cap_height=2, query_pow_bits=4, num_queries=5)),
    ("baby-bear", 6, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=4, num_queries=5)),
    ("baby-bear", 8, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=6, num_queries=4)),
    ("koala-bear", 10, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=5, query_pow_bits=8, num_queries=8)),
    # commit-phase proof of work (one grind per FRI phase) and no query PoW
    ("koala-bear", 7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, commit_pow_bits=4, query_pow_bits=0, num_queries=5)),
    ("baby-bear", 7, dict(log_blowup=1, max_log_arity=3, log_final_poly_len=0, commit_pow_bits=3, query_pow_bits=5, num_queries=4)),
]


@pytest.mark.parametrize("field,log_h,kw", CASES)
def test_prove_batch_bytes_equal_oracle(oracle, field, log_h, kw):
    arrs = harness_lib.generate(field, log_h, seed=100 + log_h, horner_chain_len=20, sponge_chain_len=3, merkle_depth=5)
    prm = layer_lib.params(**kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    tables = L.tables()
    ctx = make_ctx(field, prm)
    cap, pd = ctx.prep_create(airs_of(tables), [t["prep"] for t in tables])
    assert np.array_equal(cap, L.prep_commit())
    got = ctx.prove_batch(pd, [t["main"] for t in tables])
    L.verify(got)  # the oracle verifier accepts the GPU proof
    want = L.prove()
    assert len(got) == len(want)
    assert got == want
    # canonical field encoding switch
    got_c = ctx.prove_batch(pd, [t["main"] for t in tables], canonical_field_encoding=True)
    assert got_c == L.prove(field_encoding=1)
    pd.free()
    ctx.close()


def test_prove_rejects_unsatisfied_trace(oracle):
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    arrs = harness_lib.generate(field, 6, horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)
    prm = layer_lib.params(log_final_poly_len=1, query_pow_bits=2, num_queries=4)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    tables = L.tables()
    ctx = make_ctx(field, prm)
    cap, pd = ctx.prep_create(airs_of(tables), [t["prep"] for t in tables])
    mains = [t["main"].copy() for t in tables]
    # break the output of an Add op in lane 0 (preprocessed column 1 = sel_add): a + b != out
    row = int(np.nonzero(tables[2]["prep"][:, 1] == 1)[0][0])
    mains[2][row, 12] = (int(mains[2][row, 12]) + 1) % 0x7F000001
    # the prover checks the out-of-domain identity on its own opened values before it serialises
    # anything (include/p3r.h: P3R_EINVAL "do not satisfy the constraints", the counterpart of
    # prove_batch's internal constraint check): no unverifiable proof leaves the library
    with pytest.raises(p3r.P3rError, match="do not satisfy the constraints"):
        ctx.prove_batch(pd, mains)
    # an unbalanced lookup (a multiplicity off by one in the preprocessed data) is caught the same way
    preps = [t["prep"].copy() for t in tables]
    preps[0][3, 0] = (int(preps[0][3, 0]) + 1) % 0x7F000001
    cap2, pd2 = ctx.prep_create(airs_of(tables), preps)
    with pytest.raises(p3r.P3rError, match="do not satisfy the constraints"):
        ctx.prove_batch(pd2, [t["main"] for t in tables])
    pd2.free()
    # the satisfied traces still prove and verify
    L.verify(ctx.prove_batch(pd, [t["main"] for t in tables]))
    # shape errors are reported, not crashed on
    with pytest.raises(p3r.P3rError):
        ctx.prove_batch(pd, mains[:3])
    with pytest.raises(p3r.P3rError):
        ctx.prove_batch(pd, [m[:, :-1] for m in mains])
    pd.free()
    ctx.close()


LAYOUT = [4, 0, 2, 1, 3] + [3, 4, 0, 2, 1] + [0, 2, 3, 1, 6, 7, 4, 5]


@pytest.mark.parametrize("ext_choices,arities,layout", [(1, None, None), (0, [1, 1, 1, 1, 1, 1], None),
                                                        (1, [1, 1, 1, 1, 1, 1], LAYOUT), (0, None, LAYOUT)])
def test_selectable_protocol_details_bit_exact(oracle, ext_choices, arities, layout):
    """ext_choices / fri_log_arities of p3r_config (the [EXT] switches of DESIGN.md section 4): the HIP
    prover under the switched rules produces the oracle's bytes under the same rules."""
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    arrs = harness_lib.generate(field, 7, seed=17, horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)
    kw = dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=5)
    prm = layer_lib.params(ext_choices=ext_choices, fri_log_arities=arities, proof_layout=layout, **kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    tables = L.tables()
    ctx = p3r.Context(field=field, ext_choices=ext_choices, fri_log_arities=arities, proof_layout=layout, **kw)
    cap, pd = ctx.prep_create(airs_of(tables), [t["prep"] for t in tables])
    assert np.array_equal(cap, L.prep_commit())
    got = ctx.prove_batch(pd, [t["main"] for t in tables])
    assert got == L.prove()
    L.verify(got)
    assert got != layer_lib.OracleLayer(oracle, field, arrs, layer_lib.params(**kw)).prove()
    pd.free()
    ctx.close()


def test_bad_selectable_details_are_refused():
    import plonky3_recursion_amd as p3r
    with pytest.raises(p3r.P3rError, match="proof_layout"):
        p3r.Context(field="koala-bear", proof_layout=[0] * 18)            # not three permutations
    with pytest.raises(p3r.P3rError, match="proof_layout"):
        p3r.Context(field="koala-bear", proof_layout=list(range(5)))       # wrong length
