"""The shapes of the reference's own MMCS tests (recursion/src/pcs/mmcs.rs:1621-2572: in-circuit verify_batch against the
native MerkleTreeMmcs over mixed-height matrices), as data: `test_all_openings` opens EVERY leaf index of

  commit_batch_stark_heights          :1880-1896   [512 x 1, 8 x 1, 4 x 1, 128 x 12, 4 x 3]   (batch STARK degree_bits
                                                    [7, 1, 0, 5, 0] at log_blowup 2)
  commit_same_height_matrices         :1898-1910   [8 x 4, 4 x 2, 4 x 3]
  commit_with_cap_height_1 / _2       :1912-1927   [8 x 3] cap 1;  [16 x 2, 4 x 3] cap 2
  commit_batch_stark_with_cap_height  :1929-1939   the batch STARK shapes with cap 2
  lifted_verify_full_cap_no_path      :1954-1962   [8 x 3] with cap_height = log_max_height: an empty path
  verify_tampered_proof_fails         :1964-       4 x (8 x 1) + 4 x (8 x 2): a tampered proof must fail

(`commit_either_order` uses heights 5 and 3: this library takes power-of-two heights only - LDE heights.)
Here: the oracle's tree against the native host verifier (p3r_mmcs_verify) for every leaf index, both MMCS arities
(the arity-4 tree has a one-digest cap); the GPU twin opens the same trees on the device and compares with the oracle."""
import zlib

import numpy as np
import pytest

import oracle_lib

P = {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}
BATCH_STARK = [(512, 1), (8, 1), (4, 1), (128, 12), (4, 3)]
CASES = [
    ("commit_batch_stark_heights", BATCH_STARK, 0),
    ("commit_same_height_matrices", [(8, 4), (4, 2), (4, 3)], 0),
    ("commit_with_cap_height_1", [(8, 3)], 1),
    ("commit_with_cap_height_2", [(16, 2), (4, 3)], 2),
    ("commit_batch_stark_with_cap_height", BATCH_STARK, 2),
    ("lifted_verify_full_cap_no_path", [(8, 3)], 3),
    ("verify_tampered_proof_fails", [(8, 1)] * 4 + [(8, 2)] * 4, 0),
]


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.Oracle()


def mats_of(name, field, shapes):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return [rng.integers(0, P[field], size=s, dtype=np.uint32) for s in shapes]


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
@pytest.mark.parametrize("name,shapes,cap_height", CASES)
def test_all_openings_oracle_tree_native_verifier(oracle, field, name, shapes, cap_height):
    import plonky3_recursion_amd as p3r
    mats = mats_of(name, field, shapes)
    hmax = max(s[0] for s in shapes)
    for arity in (2, 4):
        if arity == 4 and cap_height:
            continue   # a one-digest cap only
        cfg, keep = p3r.make_config(field, cap_height=cap_height, mmcs_arity=arity, allow_unpinned_w32_defaults=True)
        cap, tree = oracle.commit4(field, mats) if arity == 4 else oracle.commit(field, mats, cap_height)
        for index in range(hmax):
            opened, proof = tree.open(index)
            if arity == 2:
                assert proof.shape[0] == int(np.log2(hmax)) - cap_height
                assert oracle.verify(field, cap, shapes, index, opened, proof)
            else:
                assert oracle.verify4(field, cap, shapes, index, opened, proof)
            p3r.mmcs_verify(cfg, cap, shapes, index, opened, proof)
        # a tampered sibling / opened value / cap entry is refused (mmcs.rs:1964-2070)
        opened, proof = tree.open(hmax - 1)
        if proof.shape[0]:
            bad = proof.copy()
            bad[0, 0] ^= 1
            with pytest.raises(p3r.P3rError, match="root mismatch"):
                p3r.mmcs_verify(cfg, cap, shapes, hmax - 1, opened, bad)
        bad = opened.copy()
        bad[0] ^= 1
        with pytest.raises(p3r.P3rError, match="root mismatch"):
            p3r.mmcs_verify(cfg, cap, shapes, hmax - 1, bad, proof)
        badcap = cap.copy()
        badcap[-1, 7] ^= 1
        with pytest.raises(p3r.P3rError, match="root mismatch"):
            p3r.mmcs_verify(cfg, badcap, shapes, hmax - 1, opened, proof)


@pytest.mark.gpu
@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
@pytest.mark.parametrize("name,shapes,cap_height", CASES)
def test_all_openings_device_tree(oracle, field, name, shapes, cap_height):
    import plonky3_recursion_amd as p3r
    mats = mats_of(name, field, shapes)
    hmax = max(s[0] for s in shapes)
    for arity in (2, 4):
        if arity == 4 and cap_height:
            continue
        c = p3r.Context(field=field, cap_height=cap_height, mmcs_arity=arity, allow_unpinned_w32_defaults=True)
        cap, tree = c.commit(mats)
        ocap, otree = oracle.commit4(field, mats) if arity == 4 else oracle.commit(field, mats, cap_height)
        assert np.array_equal(cap, ocap)
        for index in range(0, hmax, 1 if hmax <= 64 else 7):
            opened, proof = tree.open_batch(index)
            oo, op = otree.open(index)
            assert np.array_equal(opened, oo) and np.array_equal(proof, op)
            p3r.mmcs_verify(c.cfg, cap, shapes, index, opened, proof)
        tree.free()
        c.close()


# ---- recursion/tests/recursive_arity4_mmcs.rs: the arity-4 MMCS against its in-circuit verifier.  Those tests pin the
# LENGTH of the native opening proof to the circuit's schedule (`assert_eq!(mmcs_op_ids.len(), opening_proof.len())`,
# :170, :335) - one op-id per sibling, step - 1 per level - on deterministic matrices, which are reproduced here:
#   round_trip_single_height            :125-198   1024 x 4, cell i = i; indices 0, 1, 2, 3, 5, 1023; five step-4 levels
#   round_trip_wide_leaf_multi_chunk    :200-273   1024 x 40 (two absorb chunks of the rate-24 sponge), same indices
#   mixed_height_matrices               :275-290   heights [512 x4, 4096 x2, 2048 x2, 8192 x2], width 1, cell i of matrix
#                                                  m = (m + 1) * 100000 + i; indices 0, 1, 5, 8191; the schedule must hold
#                                                  a step-2 bridge (:337-341); a tampered bridge sibling (:459-547), a
#                                                  flipped direction bit (:679-758) and a tampered injected value
#                                                  (:760-840) must all fail
#   native_parity_cap_heights           :658-676   heights 1024 and 512, indices 0, 1, 2, 3, 5, 27, last (cap_height 0 here)
def iota_matrix(height, width):
    return np.arange(height * width, dtype=np.uint32).reshape(height, width)


def mixed_height_matrices():
    heights = [512, 512, 512, 512, 4096, 4096, 2048, 2048, 8192, 8192]
    return [((m + 1) * 100_000 + np.arange(h, dtype=np.uint32)).reshape(h, 1) for m, h in enumerate(heights)]


ARITY4_REF = [
    ("single_height", [iota_matrix(1024, 4)], [0, 1, 2, 3, 5, 1023], [4] * 5),
    ("wide_leaf_multi_chunk", [iota_matrix(1024, 40)], [0, 1, 2, 3, 5, 1023], [4] * 5),
    ("mixed_heights_with_injection", mixed_height_matrices(), [0, 1, 5, 8191], [2, 2, 4, 4, 4, 4, 4, 4]),
    ("cap_parity_even_log2", [iota_matrix(1024, 4)], [0, 1, 2, 3, 5, 27, 1023], [4] * 5),
    ("cap_parity_odd_log2", [iota_matrix(512, 4)], [0, 1, 2, 3, 5, 27, 511], [4] * 5),   # 512 -> 128 -> 32 -> 8 -> 2 (padded to 4) -> 1
]


def check_arity4_reference_case(oracle, field, tree, cap, mats, indices, steps, verify_native):
    shapes = [m.shape for m in mats]
    assert [st for st, _ in oracle.schedule4([s[0] for s in shapes])] == steps
    want_len = sum(st - 1 for st in steps)
    for index in indices:
        opened, proof = tree(index)
        assert proof.shape[0] == want_len          # the reference's op-id count == native proof length
        assert oracle.verify4(field, cap, shapes, index, opened, proof)
        verify_native(shapes, index, opened, proof, True)
    # soundness negatives at the reference's index (the last leaf)
    index = indices[-1]
    opened, proof = tree(index)
    if 2 in steps:                                 # step2_bridge_tampered_sibling_fails: the single sibling of the bridge
        at = sum(st - 1 for st in steps[:steps.index(2)])
        bad = proof.copy()
        bad[at, 0] ^= 1
        assert not oracle.verify4(field, cap, shapes, index, opened, bad)
        verify_native(shapes, index, opened, bad, False)
    if len(mats) > 1:                              # tampered_injected_value_fails: a row of an injected (shorter) matrix
        bad = opened.copy()
        bad[0] ^= 1                                # matrix 0 of the mixed batch is one of the shortest ones
        assert not oracle.verify4(field, cap, shapes, index, bad, proof)
        verify_native(shapes, index, bad, proof, False)
    # flipped_direction_bit_fails: the same opening under an index that differs in one direction bit
    for bit in (0, 1, 3):
        other = index ^ (1 << bit)
        assert not oracle.verify4(field, cap, shapes, other, opened, proof)
        verify_native(shapes, other, opened, proof, False)


@pytest.mark.parametrize("name,mats,indices,steps", ARITY4_REF)
def test_arity4_reference_patterns_oracle_tree(oracle, name, mats, indices, steps):
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    cfg, keep = p3r.make_config(field, mmcs_arity=4, allow_unpinned_w32_defaults=True)
    cap, tree = oracle.commit4(field, mats)

    def native(shapes, index, opened, proof, ok):
        if ok:
            p3r.mmcs_verify(cfg, cap, shapes, index, opened, proof)
        else:
            with pytest.raises(p3r.P3rError, match="root mismatch"):
                p3r.mmcs_verify(cfg, cap, shapes, index, opened, proof)
    check_arity4_reference_case(oracle, field, tree.open, cap, mats, indices, steps, native)


@pytest.mark.gpu
@pytest.mark.parametrize("name,mats,indices,steps", ARITY4_REF)
def test_arity4_reference_patterns_device_tree(oracle, name, mats, indices, steps):
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    c = p3r.Context(field=field, mmcs_arity=4, allow_unpinned_w32_defaults=True)
    cap, tree = c.commit(mats)
    ocap, otree = oracle.commit4(field, mats)
    assert np.array_equal(cap, ocap)
    for index in indices:
        o1, p1 = tree.open_batch(index)
        o2, p2 = otree.open(index)
        assert np.array_equal(o1, o2) and np.array_equal(p1, p2)

    def native(shapes, index, opened, proof, ok):
        if ok:
            p3r.mmcs_verify(c.cfg, cap, shapes, index, opened, proof)
        else:
            with pytest.raises(p3r.P3rError, match="root mismatch"):
                p3r.mmcs_verify(c.cfg, cap, shapes, index, opened, proof)
    check_arity4_reference_case(oracle, field, tree.open_batch, cap, mats, indices, steps, native)
    tree.free()
    c.close()
