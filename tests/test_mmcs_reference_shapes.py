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
        cfg, keep = p3r.make_config(field, cap_height=cap_height, mmcs_arity=arity)
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
        c = p3r.Context(field=field, cap_height=cap_height, mmcs_arity=arity)
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
