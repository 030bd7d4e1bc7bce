"""The oracle's arity-4 MMCS (oracle/hash.hpp: arity4_schedule, MerkleTree::commit4 / open4 / verify4), CPU only.

What the reference holds for this path is the in-circuit verifier (recursion/src/pcs/mmcs.rs:866-1316): the level
schedule (`arity4_path_schedule`, `padded_len`), the shape of a compression row (chunk `pos = bit + 2 bit2` holds the
running digest, an injected digest sits in chunk 1, unused chunks are zero) and the grouping of the proof's siblings
(step - 1 per level).  The schedule cases below are worked by hand from those rules; the tree itself is checked through
its own open -> verify round trip, an independent numpy recomputation of the root from the permutation seam, and
rejections."""
import numpy as np
import pytest

import oracle_lib

FIELDS = ["koala-bear", "baby-bear"]
P = {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.Oracle()


# (heights) -> [(step, injected height)], by hand from arity4_path_schedule with num_roots = 1:
#   curr = padded_len(max, 4); step = 2 iff a not-yet-injected matrix is taller than npt(curr / 4); the next layer
#   (logical curr / step, padded to a multiple of 4, 2 -> 4) takes the matrices of its own height
SCHEDULES = [
    ([1], []),
    ([2], [(4, 0)]),                                   # 2 leaves padded to 4
    ([4], [(4, 0)]),
    ([8], [(4, 0), (4, 0)]),                           # 8 -> 2 (padded to 4) -> 1
    ([16], [(4, 0), (4, 0)]),
    ([64, 64], [(4, 0), (4, 0), (4, 0)]),
    ([16, 4], [(4, 4), (4, 0)]),                       # a quaternary layer below: plain injection
    ([16, 8], [(2, 8), (4, 0), (4, 0)]),               # 8 lies between 16 and 4: a bridge level, then 8 -> 2 -> 1
    ([16, 8, 4], [(2, 8), (2, 4), (4, 0)]),            # two bridges
    ([32, 8, 2], [(4, 8), (4, 2), (4, 0)]),            # 32 -> 8 (inject) -> 2 (inject, padded to 4) -> 1
    ([64, 32, 16, 1], [(2, 32), (2, 16), (4, 0), (4, 1)]),
    ([8, 2, 1], [(4, 2), (4, 1)]),
    ([4, 2], [(2, 2), (4, 0)]),                        # 2 > npt(4 / 4) = 1: bridge to 2, padded to 4, then 4 -> 1
    ([8, 4], [(2, 4), (4, 0)]),
]


@pytest.mark.parametrize("heights,want", SCHEDULES)
def test_schedule_cases(oracle, heights, want):
    assert oracle.schedule4(heights) == want
    assert oracle.schedule4(heights[::-1]) == want    # commit order does not matter: tallest first, stable


def mats_of(rng, field, shapes):
    return [rng.integers(0, P[field], size=s, dtype=np.uint32) for s in shapes]


SHAPES = [
    [(1, 5)],
    [(2, 3)],
    [(8, 30)],
    [(64, 7), (64, 26)],
    [(16, 9), (8, 4)],
    [(16, 9), (8, 4), (4, 50)],
    [(32, 3), (8, 24), (2, 25)],
    [(4, 6), (64, 5), (16, 48), (64, 1), (1, 2)],      # commit order is not height order
    [(128, 11), (64, 3), (32, 3), (16, 3), (8, 3), (4, 3), (2, 3), (1, 3)],
]


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("shapes", SHAPES)
def test_open_verify_round_trip_and_rejections(oracle, field, shapes):
    rng = np.random.default_rng(7)
    mats = mats_of(rng, field, shapes)
    cap, tree = oracle.commit4(field, mats)
    hmax = max(s[0] for s in shapes)
    sched = oracle.schedule4([s[0] for s in shapes])
    assert tree.proof_len == sum(st - 1 for st, _ in sched)
    for index in sorted({0, 1 % hmax, hmax - 1, hmax // 2, int(rng.integers(0, hmax))}):
        opened, proof = tree.open(index)
        lm = int(np.log2(hmax))
        # opened rows: row index >> (log_max - log_h) of every matrix, in commit order
        want = np.concatenate([m[index >> (lm - int(np.log2(m.shape[0])))] for m in mats])
        assert np.array_equal(opened, want)
        assert oracle.verify4(field, cap, shapes, index, opened, proof)
        if proof.shape[0]:
            bad = proof.copy()
            bad[int(rng.integers(0, bad.shape[0])), int(rng.integers(0, 8))] ^= 1
            assert not oracle.verify4(field, cap, shapes, index, opened, bad)
            assert not oracle.verify4(field, cap, shapes, index, opened, proof[:-1])
        bad = opened.copy()
        bad[int(rng.integers(0, bad.shape[0]))] ^= 1
        assert not oracle.verify4(field, cap, shapes, index, bad, proof)
        if hmax > 1:
            assert not oracle.verify4(field, cap, shapes, (index + 1) % hmax, opened, proof) or hmax == 1
        assert not oracle.verify4(field, cap, shapes, hmax, opened, proof)   # index out of range


def w32_hash(oracle, field, row):
    """PaddingFreeSponge<Perm32, 32, 24, 8> through the permutation seam (orc_p2w_permute)."""
    s = np.zeros(32, dtype=np.uint32)
    for i in range(0, len(row), 24):
        chunk = row[i:i + 24]
        s[:len(chunk)] = chunk
        s = oracle.p2w_permute(field, s[None, :])[0]
    return s[:8].copy()


def compress4(oracle, field, chunks):
    s = np.zeros(32, dtype=np.uint32)
    for k, c in enumerate(chunks):
        s[8 * k:8 * k + 8] = c
    return oracle.p2w_permute(field, s[None, :])[0][:8].copy()


@pytest.mark.parametrize("field", FIELDS)
def test_root_recomputed_from_the_permutation_seam(oracle, field):
    """[(16, 9), (8, 4), (2, 30)]: bridge 16 -> 8 with an injection, 8 -> 2 with an injection (padded to 4), 4 -> 1."""
    rng = np.random.default_rng(3)
    shapes = [(16, 9), (8, 4), (2, 30)]
    mats = mats_of(rng, field, shapes)
    cap, tree = oracle.commit4(field, mats)
    assert oracle.schedule4([16, 8, 2]) == [(2, 8), (4, 2), (4, 0)]
    z = np.zeros(8, dtype=np.uint32)
    l0 = [w32_hash(oracle, field, mats[0][i]) for i in range(16)]
    l1 = [compress4(oracle, field, [compress4(oracle, field, [l0[2 * i], l0[2 * i + 1], z, z]), w32_hash(oracle, field, mats[1][i]), z, z])
          for i in range(8)]
    l2 = [compress4(oracle, field, [compress4(oracle, field, l1[4 * i:4 * i + 4]), w32_hash(oracle, field, mats[2][i]), z, z])
          for i in range(2)]
    root = compress4(oracle, field, [l2[0], l2[1], z, z])
    assert np.array_equal(cap[0], root)
    # the proof of leaf 13 = 0b1101: level 0 (step 2) sibling l0[12]; level 1 (step 4, index 6 -> pos 2 of group 4..7)
    # siblings l1[4], l1[5], l1[7]; level 2 (step 4 over the padded layer, index 1 -> pos 1) siblings l2[0], 0, 0
    _, proof = tree.open(13)
    want = [l0[12], l1[4], l1[5], l1[7], l2[0], z, z]
    assert np.array_equal(proof, np.array(want))


def test_arity4_needs_a_one_digest_cap(oracle):
    # the cap of an arity-4 tree strips whole compression steps (mmcs.rs:1143-1156); only cap_height = 0 is built
    import layer_lib
    import harness_lib
    arrs = harness_lib.generate("koala-bear", 6, seed=1)
    prm = layer_lib.params(log_blowup=1, log_final_poly_len=1, cap_height=1, query_pow_bits=2, num_queries=2, mmcs_arity=4)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm)
    with pytest.raises(RuntimeError, match="cap_height must be 0"):
        L.prove()


@pytest.mark.parametrize("field", FIELDS)
def test_arity4_layer_proof_round_trip(oracle, field):
    """prove_batch / verify_batch of a whole layer under the arity-4 MMCS; the proof differs from the binary one only
    in commitments, transcript and opening proofs (postcard structure unchanged: Vec<[F; 8]> of another length)."""
    import layer_lib
    import harness_lib
    arrs = harness_lib.generate(field, 7, seed=5)
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
    p4, p2 = layer_lib.params(mmcs_arity=4, **kw), layer_lib.params(**kw)
    L4, L2 = layer_lib.OracleLayer(oracle, field, arrs, p4), layer_lib.OracleLayer(oracle, field, arrs, p2)
    proof4, proof2 = L4.prove(), L2.prove()
    L4.verify(proof4)
    L2.verify(proof2)
    assert proof4 != proof2 and not np.array_equal(L4.prep_commit(), L2.prep_commit())
    with pytest.raises(RuntimeError):
        L2.verify(proof4, prep_cap=L4.prep_commit())
    with pytest.raises(RuntimeError):
        L4.verify(proof2, prep_cap=L2.prep_commit())
    bad = bytearray(proof4)
    bad[len(bad) // 2] ^= 1
    with pytest.raises(RuntimeError):
        L4.verify(bytes(bad))


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("arity", [2, 4])
def test_native_host_verifier_accepts_the_oracles_openings(oracle, field, arity):
    """p3r_mmcs_verify (host code of the C-ABI library, no GPU) against trees built by the oracle: two implementations
    of verify_batch agree on acceptance and on every rejection."""
    import plonky3_recursion_amd as p3r
    rng = np.random.default_rng(21 + arity)
    cfg, keep = p3r.make_config(field, mmcs_arity=arity, allow_unpinned_w32_defaults=True)
    for shapes in SHAPES:
        mats = mats_of(rng, field, shapes)
        cap, tree = oracle.commit4(field, mats) if arity == 4 else oracle.commit(field, mats)
        hmax = max(s[0] for s in shapes)
        for index in sorted({0, hmax - 1, int(rng.integers(0, hmax))}):
            opened, proof = tree.open(index)
            p3r.mmcs_verify(cfg, cap, shapes, index, opened, proof)
            if proof.shape[0]:
                bad = proof.copy()
                bad[int(rng.integers(0, bad.shape[0])), int(rng.integers(0, 8))] ^= 1
                with pytest.raises(p3r.P3rError, match="root mismatch"):
                    p3r.mmcs_verify(cfg, cap, shapes, index, opened, bad)
                with pytest.raises(p3r.P3rError, match="siblings"):
                    p3r.mmcs_verify(cfg, cap, shapes, index, opened, proof[1:])
            bad = opened.copy()
            bad[-1] ^= 2
            with pytest.raises(p3r.P3rError, match="root mismatch"):
                p3r.mmcs_verify(cfg, cap, shapes, index, bad, proof)
            with pytest.raises(p3r.P3rError, match="out of range"):
                p3r.mmcs_verify(cfg, cap, shapes, hmax, opened, proof)
    # and the other arity's verifier refuses the same opening
    other, keep2 = p3r.make_config(field, mmcs_arity=6 - arity, allow_unpinned_w32_defaults=True)
    with pytest.raises(p3r.P3rError):
        p3r.mmcs_verify(other, cap, shapes, index, opened, proof)


def test_native_verifier_config_rules():
    import plonky3_recursion_amd as p3r
    cfg, keep = p3r.make_config("koala-bear", mmcs_arity=4, cap_height=1, allow_unpinned_w32_defaults=True)
    z = np.zeros((2, 8), dtype=np.uint32)
    with pytest.raises(p3r.P3rError, match="cap_height must be 0"):
        p3r.mmcs_verify(cfg, z, [(4, 1)], 0, np.zeros(1, dtype=np.uint32), z)


# ---- the reference's own test of this path, as data: circuit-prover/tests/arity4_mmcs.rs:49-55 (a 64 x 4 matrix: three
# full quaternary levels 64 -> 16 -> 4 -> 1, three siblings per level), :131-137 (pos = (index >> 2 level) & 3, the three
# native siblings fill the remaining chunks in ascending order), :222-228 (indices 0, 1, 2, 3, 27, 63), :235-238
# (siblings[0][0] += 1 at index 27 must break the root)
REF_HEIGHT, REF_WIDTH, REF_INDICES = 64, 4, [0, 1, 2, 3, 27, 63]


@pytest.mark.parametrize("field", FIELDS)
def test_reference_round_trip_pattern(oracle, field):
    import plonky3_recursion_amd as p3r
    rng = np.random.default_rng(64)
    mat = rng.integers(0, P[field], size=(REF_HEIGHT, REF_WIDTH), dtype=np.uint32)
    cap, tree = oracle.commit4(field, [mat])
    cfg, keep = p3r.make_config(field, mmcs_arity=4, allow_unpinned_w32_defaults=True)
    for index in REF_INDICES:
        opened, proof = tree.open(index)
        assert proof.shape[0] % 3 == 0 and proof.shape[0] // 3 == 3      # arity4_mmcs.rs:99-104
        node = w32_hash(oracle, field, opened)
        for level in range(3):
            pos = (index >> (2 * level)) & 3
            sibs = list(proof[3 * level:3 * level + 3])
            chunks = [node if k == pos else sibs.pop(0) for k in range(4)]
            node = compress4(oracle, field, chunks)
        assert np.array_equal(node, cap[0])
        assert oracle.verify4(field, cap, [(REF_HEIGHT, REF_WIDTH)], index, opened, proof)
        p3r.mmcs_verify(cfg, cap, [(REF_HEIGHT, REF_WIDTH)], index, opened, proof)
    opened, proof = tree.open(27)
    bad = proof.copy()
    bad[0, 0] = (int(bad[0, 0]) + 1) % P[field]
    assert not oracle.verify4(field, cap, [(REF_HEIGHT, REF_WIDTH)], 27, opened, bad)
    with pytest.raises(p3r.P3rError, match="root mismatch"):
        p3r.mmcs_verify(cfg, cap, [(REF_HEIGHT, REF_WIDTH)], 27, opened, bad)
