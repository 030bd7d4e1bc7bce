"""GPU parity for the prover's own arity-4 MMCS (p3r_config.mmcs_arity = 4: MerkleTreeMmcs<.., 4, 8> over the width-32
permutation, `recursive_aggregation --arity4`): commitments, opening proofs and whole-proof BYTES from the HIP path equal
the oracle's (oracle/hash.hpp: MerkleTree::commit4 / open4), both native verifiers accept, tampering is rejected."""
import numpy as np
import pytest

import harness_lib
import layer_lib
import oracle_lib

pytestmark = pytest.mark.gpu

P = {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}
FIELDS = ["koala-bear", "baby-bear"]


def rand(rng, field, shape):
    return rng.integers(0, P[field], size=shape, dtype=np.uint32)


SHAPES = [
    [(1, 5)],
    [(2, 3)],
    [(8, 30)],
    [(64, 7), (64, 26)],
    [(16, 9), (8, 4)],
    [(16, 9), (8, 4), (4, 50)],
    [(32, 3), (8, 24), (2, 25)],
    [(4, 6), (64, 5), (16, 48), (64, 1), (1, 2)],
    [(128, 11), (64, 3), (32, 3), (16, 3), (8, 3), (4, 3), (2, 3), (1, 3)],
    [(1 << 12, 70), (1 << 11, 5), (1 << 9, 24), (1 << 12, 1), (1 << 6, 49)],
]


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("shapes", SHAPES)
def test_commit_and_open_vs_oracle(oracle, field, shapes):
    import plonky3_recursion_amd as p3r
    c = p3r.Context(field=field, mmcs_arity=4, allow_unpinned_w32_defaults=True)
    rng = np.random.default_rng(11)
    mats = [rand(rng, field, s) for s in shapes]
    cap, tree = c.commit(mats)
    ocap, otree = oracle.commit4(field, mats)
    assert np.array_equal(cap, ocap)
    hmax = max(s[0] for s in shapes)
    for index in sorted({0, 1 % hmax, hmax - 1, hmax // 2, int(rng.integers(0, hmax)), int(rng.integers(0, hmax))}):
        opened, proof = tree.open_batch(index)
        oo, op = otree.open(index)
        assert np.array_equal(opened, oo) and np.array_equal(proof, op)
        assert oracle.verify4(field, cap, shapes, index, opened, proof)
        p3r.mmcs_verify(c.cfg, cap, shapes, index, opened, proof)          # the native host verifier
        if proof.shape[0]:
            bad = proof.copy()
            bad[int(rng.integers(0, bad.shape[0])), 3] ^= 1
            with pytest.raises(p3r.P3rError, match="root mismatch"):
                p3r.mmcs_verify(c.cfg, cap, shapes, index, opened, bad)
            with pytest.raises(p3r.P3rError, match="siblings"):
                p3r.mmcs_verify(c.cfg, cap, shapes, index, opened, proof[:-1])
        bad = opened.copy()
        bad[0] ^= 1
        with pytest.raises(p3r.P3rError, match="root mismatch"):
            p3r.mmcs_verify(c.cfg, cap, shapes, index, bad, proof)
    tree.free()
    c.close()


def test_reference_round_trip_pattern_on_the_device(oracle):
    """circuit-prover/tests/arity4_mmcs.rs: a 64 x 4 matrix (three quaternary levels, three siblings a level), indices
    0, 1, 2, 3, 27, 63; the openings must be the ones the W32 TABLE's rows then consume - checked here by feeding the
    device's opening into the synthetic table rows' own recomputation of the root (the oracle's permutation seam)."""
    import plonky3_recursion_amd as p3r
    from test_oracle_arity4_mmcs import REF_HEIGHT, REF_INDICES, REF_WIDTH, compress4, w32_hash
    for field in FIELDS:
        c = p3r.Context(field=field, mmcs_arity=4, allow_unpinned_w32_defaults=True)
        mat = rand(np.random.default_rng(64), field, (REF_HEIGHT, REF_WIDTH))
        cap, tree = c.commit([mat])
        for index in REF_INDICES:
            opened, proof = tree.open_batch(index)
            assert proof.shape == (9, 8)
            node = w32_hash(oracle, field, opened)
            for level in range(3):
                pos = (index >> (2 * level)) & 3
                sibs = list(proof[3 * level:3 * level + 3])
                node = compress4(oracle, field, [node if k == pos else sibs.pop(0) for k in range(4)])
            assert np.array_equal(node, cap[0])
        tree.free()
        c.close()


def test_tall_tree_every_level_kind(oracle):
    """2^17 leaves: a bridge level at the bottom (2^16 injected), a second bridge 2^14 -> 2^13 (injected), a plain
    injection at 2^9 = 2^13 / 4^2, and a padded top (2^9 -> 128 -> 32 -> 8 -> 2: the last level compresses a layer of 2
    padded to 4)."""
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    c = p3r.Context(field=field, mmcs_arity=4, allow_unpinned_w32_defaults=True)
    rng = np.random.default_rng(12)
    shapes = [(1 << 17, 3), (1 << 16, 9), (1 << 13, 2), (1 << 17, 30), (1 << 9, 5), (1 << 13, 8)]
    mats = [rand(rng, field, s) for s in shapes]
    cap, tree = c.commit(mats)
    ocap, otree = oracle.commit4(field, mats)
    assert np.array_equal(cap, ocap)
    sched = oracle.schedule4([s[0] for s in shapes])
    assert sched == [(2, 1 << 16), (4, 0), (2, 1 << 13), (4, 0), (4, 1 << 9), (4, 0), (4, 0), (4, 0), (4, 0), (4, 0)]
    for index in (0, 1, 4097, 77777, (1 << 17) - 1):
        opened, proof = tree.open_batch(index)
        oo, op = otree.open(index)
        assert np.array_equal(opened, oo) and np.array_equal(proof, op)
        p3r.mmcs_verify(c.cfg, cap, shapes, index, opened, proof)
    tree.free()
    c.close()


def test_binary_and_arity4_verifiers_do_not_accept_each_other(oracle):
    import plonky3_recursion_amd as p3r
    field = "baby-bear"
    rng = np.random.default_rng(13)
    shapes = [(64, 5), (16, 3)]
    mats = [rand(rng, field, s) for s in shapes]
    c4, c2 = p3r.Context(field=field, mmcs_arity=4, allow_unpinned_w32_defaults=True), p3r.Context(field=field)
    cap4, t4 = c4.commit(mats)
    cap2, t2 = c2.commit(mats)
    assert not np.array_equal(cap4, cap2)
    o4, p4 = t4.open_batch(37)
    o2, p2 = t2.open_batch(37)
    assert np.array_equal(o4, o2) and p4.shape[0] == 3 + 3 + 3 and p2.shape[0] == 6   # 64 -(4)-> 16 (inject) -(4)-> 4 -(4)-> 1: three siblings a level
    p3r.mmcs_verify(c2.cfg, cap2, shapes, 37, o2, p2)
    with pytest.raises(p3r.P3rError):
        p3r.mmcs_verify(c2.cfg, cap4, shapes, 37, o4, p4)
    with pytest.raises(p3r.P3rError):
        p3r.mmcs_verify(c4.cfg, cap2, shapes, 37, o2, p2)
    for t in (t4, t2):
        t.free()
    c4.close()
    c2.close()


def test_arity4_refuses_a_cap(oracle):
    import plonky3_recursion_amd as p3r
    with pytest.raises(p3r.P3rError, match="cap_height must be 0"):
        p3r.Context(field="koala-bear", mmcs_arity=4, cap_height=1, allow_unpinned_w32_defaults=True)
    with pytest.raises(p3r.P3rError, match="mmcs_arity"):
        p3r.Context(field="koala-bear", mmcs_arity=3, allow_unpinned_w32_defaults=True)


def test_custom_width32_constants_change_the_tree(oracle):
    """The constants are the caller's data (p3r_config.poseidon2_w32_rc / _diag): a different table gives the oracle's
    tree for THAT table."""
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    rc, diag = oracle_lib.default_w32(field)
    rng = np.random.default_rng(14)
    rc2 = rand(rng, field, rc.shape)
    diag2 = rand(rng, field, diag.shape)     # a general diagonal: every entry a full field element
    shapes = [(32, 29), (8, 3)]
    mats = [rand(rng, field, s) for s in shapes]
    c = p3r.Context(field=field, mmcs_arity=4, poseidon2_w32_rc=rc2, poseidon2_w32_diag=diag2, allow_unpinned_w32_defaults=True)
    cap, tree = c.commit(mats)
    ocap, otree = oracle.commit4(field, mats, w32=(rc2, diag2))
    dcap, _ = oracle.commit4(field, mats)
    assert np.array_equal(cap, ocap) and not np.array_equal(cap, dcap)
    opened, proof = tree.open_batch(21)
    oo, op = otree.open(21)
    assert np.array_equal(opened, oo) and np.array_equal(proof, op)
    p3r.mmcs_verify(c.cfg, cap, shapes, 21, opened, proof)
    tree.free()
    c.close()


def make_ctx(field, prm, **kw):
    import plonky3_recursion_amd as p3r
    return p3r.Context(field=field, log_blowup=prm.log_blowup, max_log_arity=prm.max_log_arity,
                       cap_height=prm.cap_height, log_final_poly_len=prm.log_final_poly_len,
                       commit_pow_bits=prm.commit_pow_bits, query_pow_bits=prm.query_pow_bits,
                       num_queries=prm.num_queries, mmcs_arity=prm.mmcs_arity, **kw, allow_unpinned_w32_defaults=True)


def airs_of(tables):
    return [dict(kind=t["kind_id"], lanes=t["lanes"], horner_packed_steps=t["horner_k"], coeff_lookups=0)
            for t in tables]


CASES = [
    ("koala-bear", 5, 0, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=0, query_pow_bits=3, num_queries=4)),
    ("koala-bear", 7, 0, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, query_pow_bits=5, num_queries=6)),
    ("baby-bear", 6, 0, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=4, num_queries=5)),
    ("baby-bear", 8, 0, dict(log_blowup=1, max_log_arity=2, log_final_poly_len=2, query_pow_bits=6, num_queries=4)),
    ("koala-bear", 10, 0, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=5, query_pow_bits=8, num_queries=8)),
    # commit-phase proof of work: the transcript steps between the FRI phases run on the host
    ("koala-bear", 7, 0, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, commit_pow_bits=4, query_pow_bits=0, num_queries=5)),
    # a layer that PROVES width-32 rows under the arity-4 MMCS: the recursion's own configuration
    ("koala-bear", 8, harness_lib.P2_W32, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=5)),
]


@pytest.mark.parametrize("field,log_h,flags,kw", CASES)
def test_prove_batch_bytes_equal_oracle(oracle, field, log_h, flags, kw):
    import plonky3_recursion_amd as p3r
    arrs = harness_lib.generate(field, log_h, seed=300 + log_h, flags=flags, horner_chain_len=20, sponge_chain_len=3, merkle_depth=5)
    prm = layer_lib.params(mmcs_arity=4, **kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    tables = L.tables()
    ctx = make_ctx(field, prm)
    cap, pd = ctx.prep_create(airs_of(tables), [t["prep"] for t in tables])
    assert np.array_equal(cap, L.prep_commit())
    got = ctx.prove_batch(pd, [t["main"] for t in tables])
    L.verify(got)                               # the oracle verifier accepts the GPU proof
    want = L.prove()
    assert len(got) == len(want) and got == want
    # the native verifier: accepts; under the binary configuration the same bytes are refused
    db = [int(np.log2(t["main"].shape[0])) for t in tables]
    p3r.verify_batch(ctx.cfg, airs_of(tables), cap, db, got)
    cfg2, keep = p3r.make_config(field, prm.log_blowup, prm.max_log_arity, 0, prm.log_final_poly_len, prm.commit_pow_bits,
                                 prm.query_pow_bits, prm.num_queries, allow_unpinned_w32_defaults=True)
    with pytest.raises(p3r.P3rError):
        p3r.verify_batch(cfg2, airs_of(tables), cap, db, got)
    bad = bytearray(got)
    bad[len(bad) - 40] ^= 1
    with pytest.raises(p3r.P3rError):
        p3r.verify_batch(ctx.cfg, airs_of(tables), cap, db, bytes(bad))
    pd.free()
    ctx.close()


def test_prove_next_layer_arity4(oracle):
    """The whole hot path - circuit run, device preparation, table build, prove - under the arity-4 MMCS: bytes equal the
    oracle's, both verifiers accept."""
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    field, log_h = "koala-bear", 9
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=5, num_queries=6)
    arrs = harness_lib.generate(field, log_h, seed=41, horner_chain_len=24, sponge_chain_len=4, merkle_depth=6)
    prm = layer_lib.params(mmcs_arity=4, **kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    ctx = make_ctx(field, prm)
    tp = p3r.TablePacking().with_fri_params(kw["log_final_poly_len"], kw["log_blowup"])
    cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(arrs), p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=tp))
    assert cache.prepared_circuit.prepared_on_device
    out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=wl.circuit_inputs_from_arrays(arrs)), ctx,
                               p3r.FriRecursionBackend(), p3r.ProveNextLayerParams(table_packing=tp), prep=cache)
    assert np.array_equal(cache.circuit_prover_data.preprocessed_commitment, L.prep_commit())
    assert out.proof.proof == L.prove()
    cache.prover.verify_all_tables(out.proof)
    layer_lib.oracle_verify_statement(oracle, field, prm, out.proof.airs(), cache.circuit_prover_data.preprocessed_commitment,
                                      out.proof.proof)
    ctx.close()
