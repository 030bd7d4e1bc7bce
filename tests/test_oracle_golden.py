"""CPU: the C++ oracle against the self-generated golden vectors (tests/golden/primitives.json,
made by the independent pure-Python restatement tools/pyref.py).  PARITY UNPINNED: the reference
holds no golden vectors for this path."""
import numpy as np
import pytest

FIELDS = [("koala-bear", "koala_bear"), ("baby-bear", "baby_bear")]


@pytest.mark.parametrize("field,key", FIELDS)
def test_trace_widths_match_survey(oracle, field, key):
    # 166 (KoalaBear) / 300 (BabyBear) incl. the 2 circuit columns
    assert oracle.trace_width(field) == {"koala-bear": 166, "baby-bear": 300}[field]


@pytest.mark.parametrize("field,key", FIELDS)
def test_permutation_kats(oracle, golden, field, key):
    g = golden["prim"][key]
    ins = np.array([k["in"] for k in g["permute"]], dtype=np.uint32)
    outs = np.array([k["out"] for k in g["permute"]], dtype=np.uint32)
    assert np.array_equal(oracle.permute(field, ins), outs)


@pytest.mark.parametrize("field,key", FIELDS)
def test_trace_cells_match_python(oracle, golden, field, key):
    g = golden["prim"][key]["perm_cells"]
    n = 4
    inputs = np.tile(np.array(g["in"], dtype=np.uint32), (n, 1))
    z = np.zeros(n, dtype=np.uint8)
    tr = oracle.trace_rows(field, inputs, z + 1, z, z, np.zeros(n, dtype=np.uint32))
    assert np.array_equal(tr[2, :-2], np.array(g["cells"], dtype=np.uint32))
    assert np.all(tr[:, -2:] == 0)


@pytest.mark.parametrize("field,key", FIELDS)
def test_sponge_and_compress_via_mmcs(oracle, golden, field, key):
    g = golden["prim"][key]
    # a height-1 matrix commits to hash(row): PaddingFreeSponge KATs at ragged widths
    for kat in g["sponge"]:
        cap, _ = oracle.commit(field, [np.array([kat["in"]], dtype=np.uint32)])
        assert cap[0].tolist() == kat["out"]
    # height-2 matrix of width 8: root = compress(hash(r0), hash(r1)); with width-8 rows the
    # sponge is one permutation, checked above, so check compress through its KAT inputs:
    c = g["compress"]
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import pyref
    f = pyref.FIELDS[key]
    rc = golden["rc"][key]
    rows = np.array([c["left"], c["right"]], dtype=np.uint32)
    cap, _ = oracle.commit(field, [rows])
    expect = pyref.compress(pyref.sponge_hash(c["left"], rc, f), pyref.sponge_hash(c["right"], rc, f), rc, f)
    assert cap[0].tolist() == expect
    assert pyref.compress(c["left"], c["right"], rc, f) == c["out"]


@pytest.mark.parametrize("field,key", FIELDS)
def test_challenger_transcript(oracle, golden, field, key):
    g = golden["prim"][key]["challenger"]
    out = oracle.challenger_script(field, g["ops"], g["args"])
    assert out.tolist() == g["out"]


@pytest.mark.parametrize("field,key", FIELDS)
def test_extension_field(oracle, golden, field, key):
    g = golden["prim"][key]["ext"]
    mul, inv = oracle.ext_ops(field, g["a"], g["b"])
    assert mul.tolist() == g["mul"] and inv.tolist() == g["inv_a"]


@pytest.mark.parametrize("field,key", FIELDS)
def test_coset_lde_small(oracle, golden, field, key):
    g = golden["prim"][key]["lde"]
    out = oracle.coset_lde(field, np.array(g["evals"], dtype=np.uint32), g["added_bits"], g["shift"])
    assert np.array_equal(out, np.array(g["lde"], dtype=np.uint32))


@pytest.mark.parametrize("field,key", FIELDS)
def test_lde_first_block_is_coset_of_trace_domain(oracle, field, key):
    # size-independent property: rows [0,h) of the bit-reversed LDE are the evaluations on
    # shift*H (bit-reversed), so an LDE with shift w_{4h}^0=1... is checked via linearity +
    # the identity "LDE with added_bits=0 and shift=1 is the bit-reversed input".
    rng = np.random.default_rng(7)
    p = {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}[field]
    h, w = 64, 5
    a = rng.integers(0, p, size=(h, w), dtype=np.uint32)
    out = oracle.coset_lde(field, a, 0, 1)
    rev = [int(format(i, "06b")[::-1], 2) for i in range(h)]
    assert np.array_equal(out, a[rev])
    b = rng.integers(0, p, size=(h, w), dtype=np.uint32)
    s = ((a.astype(np.uint64) + b) % p).astype(np.uint32)
    la, lb, ls = (oracle.coset_lde(field, x, 2, 3) for x in (a, b, s))
    assert np.array_equal(ls, ((la.astype(np.uint64) + lb) % p).astype(np.uint32))


@pytest.mark.parametrize("field,key", FIELDS)
@pytest.mark.parametrize("cap_height", [0, 1, 2])
def test_mmcs_mixed_heights_open_verify(oracle, field, key, cap_height):
    # mirrors recursion/src/pcs/mmcs.rs:1668-1776 (mixed-height matrices, cap heights 0..2)
    rng = np.random.default_rng(11)
    p = {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}[field]
    shapes = [(8, 3), (32, 5), (32, 9), (16, 8), (4, 1)]
    mats = [rng.integers(0, p, size=s, dtype=np.uint32) for s in shapes]
    cap, tree = oracle.commit(field, mats, cap_height)
    for index in (0, 5, 17, 31):
        opened, proof = tree.open(index)
        off = 0
        for m in mats:
            row = index >> (5 - int(np.log2(m.shape[0])))
            assert np.array_equal(opened[off:off + m.shape[1]], m[row])
            off += m.shape[1]
        assert oracle.verify(field, cap, shapes, index, opened, proof)
        bad = opened.copy()
        bad[3] = (int(bad[3]) + 1) % p
        assert not oracle.verify(field, cap, shapes, index, bad, proof)
        badp = proof.copy()
        badp[0, 0] = (int(badp[0, 0]) + 1) % p
        assert not oracle.verify(field, cap, shapes, index, opened, badp)
