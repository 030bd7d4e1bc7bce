"""GPU: the ZK configuration (p3r_config.zk = 1: HidingFriPcs, create_config_zk of recursion/examples/common/mod.rs:
511-553) through the HIP prover.  The prover side of HidingFriPcs is un-vendored and its proofs are randomised, so
byte parity with upstream is UNDEFINED; what is checked: both verifiers (the oracle's and the
native one, restatements of recursion/src/verifier/batch_stark.rs:424-428,487-490,536,623-661,701-735,855-864,
1116-1260) accept the HIP proofs; two proofs of one input differ; the two PCS types refuse each other's proofs; and -
the random values being a counter-based function of (seed, proof number, cell) that the oracle restates - the HIP
proof equals the oracle's byte for byte when both are given the same seed and proof number, which pins every kernel of
the ZK path (randomisation, masks, coset moves, the eight-chunk quotient, triple-packed LogUp)."""
import numpy as np
import pytest

import harness_lib
import layer_lib
import proof_codec
from test_zk import CASES as ZK_CASES, native_verify

pytestmark = pytest.mark.gpu


def make_ctx(field, prm, **extra):
    import plonky3_recursion_amd as p3r
    return p3r.Context(field=field, log_blowup=prm.log_blowup, max_log_arity=prm.max_log_arity,
                       cap_height=prm.cap_height, log_final_poly_len=prm.log_final_poly_len,
                       commit_pow_bits=prm.commit_pow_bits, query_pow_bits=prm.query_pow_bits,
                       num_queries=prm.num_queries, mmcs_arity=prm.mmcs_arity or 2, zk=prm.zk,
                       num_random_codewords=prm.num_random_codewords, zk_key=list(prm.zk_key), zk_deterministic=True,
                       allow_unpinned_w32_defaults=True, **extra)


def airs_of(tables):
    return [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"], coeff_lookups=0) for t in tables]


GPU_CASES = ZK_CASES + [
    ("koala-bear", 9, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=3, query_pow_bits=6, num_queries=6), None, 0, 2),
    ("baby-bear", 8, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, query_pow_bits=5, num_queries=5),
     dict(alu_lanes=4, horner_packed_steps=5), 0, 2),
    # tall enough for the two-pass NTT on the extended domain and the coset moves
    ("koala-bear", 12, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=4, query_pow_bits=6, num_queries=6), None, 0, 2),
]


@pytest.mark.parametrize("field,log_h,kw,packing,flags,codewords", GPU_CASES)
def test_zk_prove_batch(oracle, field, log_h, kw, packing, flags, codewords):
    import plonky3_recursion_amd as p3r
    arrs = harness_lib.generate(field, log_h, seed=70 + log_h, flags=flags, horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)
    prm = layer_lib.params(zk=1, num_random_codewords=codewords, zk_seed=11, **kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=packing)
    tables = L.tables()
    ctx = make_ctx(field, prm)
    cap, pd = ctx.prep_create(airs_of(tables), [t["prep"] for t in tables])
    assert np.array_equal(cap, L.prep_commit())
    assert ctx.zk_nonce == 0
    first = ctx.prove_batch(pd, [t["main"] for t in tables])
    second = ctx.prove_batch(pd, [t["main"] for t in tables])
    assert ctx.zk_nonce == 2
    assert first != second   # the PCS's RNG advanced (lengths differ too: field elements are varints)
    for pf in (first, second):
        L.verify(pf)
        native_verify(field, prm, tables, cap, pf)
    # same seed, same proof number: the oracle's bytes
    assert first == L.prove()
    prm1 = layer_lib.params(zk=1, num_random_codewords=codewords, zk_seed=11, zk_nonce=1, **kw)
    assert second == layer_lib.OracleLayer(oracle, field, arrs, prm1, packing=packing).prove()
    # replay: the nonce is the caller's to set
    ctx.zk_nonce = 0
    assert ctx.prove_batch(pd, [t["main"] for t in tables]) == first
    # the non-ZK verifier configuration refuses it, the ZK one refuses a non-ZK proof of the same traces
    prm0 = layer_lib.params(**kw)
    with pytest.raises(p3r.P3rError):
        native_verify(field, prm0, tables, cap, first, zk=0, degree_bits=[int(t["main"].shape[0]).bit_length() for t in tables])
    ctx0 = make_ctx(field, prm0)
    cap0, pd0 = ctx0.prep_create(airs_of(tables), [t["prep"] for t in tables])
    plain = ctx0.prove_batch(pd0, [t["main"] for t in tables])
    with pytest.raises(p3r.P3rError):
        native_verify(field, prm, tables, cap, plain)
    with pytest.raises(RuntimeError):
        L.verify(plain)
    # unsatisfied traces are refused before anything is serialised, as without ZK
    mains = [t["main"].copy() for t in tables]
    if tables[2]["kind"] == "alu" and np.any(tables[2]["prep"][:, 1] == 1):
        row = int(np.nonzero(tables[2]["prep"][:, 1] == 1)[0][0])
        d = 1 if field == "koala-bear" else 1
        mains[2][row, 12] = (int(mains[2][row, 12]) + d) % oracle_modulus(field)
        with pytest.raises(p3r.P3rError, match="do not satisfy the constraints"):
            ctx.prove_batch(pd, mains)
    pd0.free()
    ctx0.close()
    pd.free()
    ctx.close()


def oracle_modulus(field):
    return {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}[field]


@pytest.mark.parametrize("field,log_h", [("koala-bear", 7), ("baby-bear", 6)])
def test_zk_prove_next_layer_and_wire_round_trip(oracle, field, log_h):
    """The layer boundary under ZK: prove_next_layer -> BatchStarkProof (extended degree bits in stark_common,
    recursion.rs:374) -> postcard -> parse (P3R_PROOF_ZK) -> verify_all_tables."""
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=4, num_queries=5)
    arrs = harness_lib.generate(field, log_h, seed=7 + log_h, horner_chain_len=20, sponge_chain_len=3, merkle_depth=5)
    prm = layer_lib.params(zk=1, zk_seed=99, **kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    ctx = make_ctx(field, prm)
    tp = pv.TablePacking().with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs), pv.FriRecursionBackend(),
                                     pv.ProveNextLayerParams(table_packing=tp))
    cpd = cache.circuit_prover_data
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    traces = wl.traces_from_arrays(arrs)
    out = pv.prove_next_layer(pv.RecursionInput(traces=traces), ctx, pv.FriRecursionBackend(),
                              pv.ProveNextLayerParams(table_packing=cpd.packing), prep=cache)
    assert out.proof.proof == L.prove()
    L.verify(out.proof.proof)
    assert out.proof.degree_bits == tuple(int(h).bit_length() for h in cpd.table_heights if h)
    cache.prover.verify_all_tables(out.proof)
    wire = out.proof.to_postcard()
    back = pv.BatchStarkProof.from_postcard(wire, field, zk=True)
    assert back.proof == out.proof.proof and back.to_postcard() == wire
    assert back.degree_bits == out.proof.degree_bits and back.preprocessed_widths == out.proof.preprocessed_widths
    assert back.non_primitives == out.proof.non_primitives and back.table_packing == out.proof.table_packing
    assert np.array_equal(back.preprocessed_commitment, out.proof.preprocessed_commitment)
    cache.prover.verify_all_tables(back)
    with pytest.raises(p3r.P3rError):
        pv.BatchStarkProof.from_postcard(wire, field)   # the non-ZK proof type does not parse it
    # a second layer proof from the same cache differs and verifies
    out2 = pv.prove_next_layer(pv.RecursionInput(traces=traces), ctx, pv.FriRecursionBackend(),
                               pv.ProveNextLayerParams(table_packing=cpd.packing), prep=cache)
    assert out2.proof.proof != out.proof.proof
    cache.prover.verify_all_tables(out2.proof)
    cpd.free()
    ctx.close()


@pytest.mark.parametrize("d,flags,coeff,cd", [(5, harness_lib.RECOMPOSE_COEFF, 1, 5), (5, harness_lib.RECOMPOSE_BOTH, 0, 4),
                                              (1, harness_lib.NO_RECOMPOSE, 0, 4)])
def test_zk_circuit_degrees_and_quintic_challenge_on_the_device(oracle, d, flags, coeff, cd):
    """ZK proofs of D = 5 / D = 1 layers, under the quartic and the quintic challenge field, through prove_next_layer:
    the oracle's bytes (same seed and proof number), both verifiers."""
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    from test_zk import native_verify_d
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=5)
    prm = layer_lib.params(zk=1, zk_seed=8, challenge_degree=cd, **kw)
    arrs = harness_lib.generate("koala-bear", 8, seed=30 + d + cd, flags=flags, ext_degree=d, horner_chain_len=12, sponge_chain_len=3,
                                merkle_depth=4)
    both = bool(flags & harness_lib.RECOMPOSE_BOTH) and not coeff
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(ext_degree=d, recompose_coeff_lookups=coeff))
    ctx = make_ctx("koala-bear", prm, ext_degree=d, challenge_degree=cd)
    tp = pv.TablePacking().with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs, ext_degree=d, recompose_coeff_lookups=bool(coeff)),
                                     pv.FriRecursionBackend(), pv.ProveNextLayerParams(table_packing=tp))
    cpd = cache.circuit_prover_data
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    out = pv.prove_next_layer(pv.RecursionInput(traces=wl.traces_from_arrays(arrs, d)), ctx, pv.FriRecursionBackend(),
                              pv.ProveNextLayerParams(table_packing=cpd.packing), prep=cache)
    assert out.proof.proof == L.prove()
    L.verify(out.proof.proof)
    cache.prover.verify_all_tables(out.proof)
    if not both:
        native_verify_d(prm, L.tables(), L.prep_commit(), out.proof.proof, d, coeff, cd)
    back = pv.BatchStarkProof.from_postcard(out.proof.to_postcard(), "koala-bear", challenge_degree=cd, zk=True)
    cache.prover.verify_all_tables(back)
    cpd.free()
    ctx.close()
