"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle and the golden vectors.
Bit-exact: this is integer field arithmetic."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu

FIELDS = [("koala-bear", "koala_bear"), ("baby-bear", "baby_bear")]
P = oracle_lib.MODULUS


@pytest.fixture(scope="module", params=[f[0] for f in FIELDS])
def ctx(request):
    import plonky3_recursion_amd as p3r
    c = p3r.Context(field=request.param)
    yield c
    c.close()


def key_of(ctx):
    return dict(FIELDS)[ctx.field]


def rand(rng, field, shape):
    return rng.integers(0, P[field], size=shape, dtype=np.uint32)


def test_upload_download_roundtrip(ctx):
    rng = np.random.default_rng(1)
    for shape in [(1, 1), (2, 3), (64, 64), (128, 7), (256, 166), (1024, 65)]:
        a = rand(rng, ctx.field, shape)
        a[0, 0] = P[ctx.field] - 1
        assert np.array_equal(ctx.upload(a).download(), a)


def test_permute_golden(ctx, golden):
    g = golden["prim"][key_of(ctx)]
    ins = np.array([k["in"] for k in g["permute"]], dtype=np.uint32)
    outs = np.array([k["out"] for k in g["permute"]], dtype=np.uint32)
    assert np.array_equal(ctx.permute_batch(ins), outs)


@pytest.mark.parametrize("n", [1, 3, 64, 1000, 1 << 14])
def test_permute_vs_oracle(ctx, oracle, n):
    rng = np.random.default_rng(n)
    s = rand(rng, ctx.field, (n, 16))
    assert np.array_equal(ctx.permute_batch(s), oracle.permute(ctx.field, s))


def make_p2_rows(rng, field, n, adversarial=False):
    inputs = rand(rng, field, (n, 16))
    if adversarial:  # one very long Merkle chain: worst case for the accumulator scan
        new_start = np.zeros(n, dtype=np.uint8); new_start[0] = 1
        merkle = np.ones(n, dtype=np.uint8)
    else:
        new_start = (rng.random(n) < 0.15).astype(np.uint8)
        merkle = (rng.random(n) < 0.7).astype(np.uint8)
    bit = rng.integers(0, 2, size=n, dtype=np.uint8)
    idx = rand(rng, field, n)
    return inputs, new_start, merkle, bit, idx


@pytest.mark.parametrize("n,adv", [(1, False), (2, False), (32, False), (1024, False), (4096, True), (1 << 13, False)])
def test_trace_fill_vs_oracle(ctx, oracle, n, adv):
    rng = np.random.default_rng(100 + n)
    rows = make_p2_rows(rng, ctx.field, n, adv)
    got = ctx.generate_trace_rows(*rows)
    want = oracle.trace_rows(ctx.field, *rows)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_trace_fill_golden_cells(ctx, golden):
    g = golden["prim"][key_of(ctx)]["perm_cells"]
    n = 4
    inputs = np.tile(np.array(g["in"], dtype=np.uint32), (n, 1))
    z = np.zeros(n, dtype=np.uint8)
    tr = ctx.generate_trace_rows(inputs, z + 1, z, z, np.zeros(n, dtype=np.uint32))
    assert np.array_equal(tr[1, :-2], np.array(g["cells"], dtype=np.uint32))


def test_trace_fill_rejects_non_power_of_two(ctx):
    import plonky3_recursion_amd as p3r
    rng = np.random.default_rng(5)
    with pytest.raises(p3r.P3rError):
        ctx.generate_trace_rows(*make_p2_rows(rng, ctx.field, 3))


def test_lde_golden(ctx, golden):
    g = golden["prim"][key_of(ctx)]["lde"]
    out = ctx.coset_lde_batch(np.array(g["evals"], dtype=np.uint32), g["added_bits"], g["shift"])
    assert np.array_equal(out, np.array(g["lde"], dtype=np.uint32))


@pytest.mark.parametrize("log_h,w,added", [(1, 1, 1), (2, 3, 2), (5, 4, 2), (8, 7, 1), (10, 3, 2), (11, 2, 2),
                                            (12, 3, 2), (13, 2, 1), (14, 2, 2), (15, 1, 2),
                                            # every sub-transform size of the lean NTT kernels (kernels_ntt2.hip.h): forward
                                            # 2^6..2^8 x 2^6..2^11, inverse 2^6..2^10 column passes; 1, 2, 4, 8 cosets
                                            (12, 2, 0), (13, 3, 3), (16, 2, 2), (16, 1, 3), (17, 1, 2), (18, 1, 2), (19, 1, 1), (20, 1, 1),
                                            (21, 1, 2)])
def test_lde_vs_oracle(ctx, oracle, log_h, w, added):
    rng = np.random.default_rng(log_h * 31 + w)
    a = rand(rng, ctx.field, (1 << log_h, w))
    shift = oracle_lib.GENERATOR[ctx.field]
    assert np.array_equal(ctx.coset_lde_batch(a, added, shift), oracle.coset_lde(ctx.field, a, added, shift))


def test_lde_other_shift(ctx, oracle):
    rng = np.random.default_rng(77)
    a = rand(rng, ctx.field, (1 << 12, 2))
    shift = int(rand(rng, ctx.field, 1)[0]) | 1
    assert np.array_equal(ctx.coset_lde_batch(a, 2, shift), oracle.coset_lde(ctx.field, a, 2, shift))


def test_lde_large_properties(ctx):
    """2^20 rows: identity, linearity, and 'first block = shifted-coset evaluation' checks that
    need no oracle run (size-independent properties)."""
    rng = np.random.default_rng(2020)
    p = P[ctx.field]
    h, w = 1 << 20, 2
    a = rand(rng, ctx.field, (h, w))
    b = rand(rng, ctx.field, (h, w))
    s = ((a.astype(np.uint64) + b) % p).astype(np.uint32)
    g = oracle_lib.GENERATOR[ctx.field]
    la, lb, ls = (ctx.coset_lde_batch(x, 1, g) for x in (a, b, s))
    assert np.array_equal(ls, ((la.astype(np.uint64) + lb) % p).astype(np.uint32))
    # added_bits=0, shift=1 is a pure bit-reversal of the rows
    idx = np.arange(h, dtype=np.uint32)
    rev = np.zeros(h, dtype=np.uint32)
    for i in range(20):
        rev |= ((idx >> i) & 1) << (19 - i)
    assert np.array_equal(ctx.coset_lde_batch(a, 0, 1), a[rev])
    # a constant column stays constant on every coset
    c = np.full((h, 1), 12345, dtype=np.uint32)
    assert np.all(ctx.coset_lde_batch(c, 2, g) == 12345)


@pytest.mark.parametrize("cap_height", [0, 2])
def test_mmcs_vs_oracle(oracle, cap_height, ctx):
    import plonky3_recursion_amd as p3r
    c = p3r.Context(field=ctx.field, cap_height=cap_height)
    rng = np.random.default_rng(33)
    shapes = [(64, 3), (256, 5), (256, 9), (128, 8), (8, 1), (256, 16), (32, 31)]
    mats = [rand(rng, ctx.field, s) for s in shapes]
    cap, tree = c.commit(mats)
    ocap, otree = oracle.commit(ctx.field, mats, cap_height)
    assert np.array_equal(cap, ocap)
    for index in (0, 1, 77, 200, 255):
        opened, proof = tree.open_batch(index)
        oo, op = otree.open(index)
        assert np.array_equal(opened, oo) and np.array_equal(proof, op)
        assert oracle.verify(ctx.field, cap, shapes, index, opened, proof)
    tree.free()
    c.close()


@pytest.mark.parametrize("cap_height", [0, 1])
def test_mmcs_tall_mixed_heights(oracle, cap_height, ctx):
    """A tree tall enough for every layer regime: one permutation per lane above 32 K nodes (with an
    injection there), several workgroups of the 8-levels-per-launch kernel (injection inside), and
    the single-workgroup tail (injection inside, cap above the root)."""
    import plonky3_recursion_amd as p3r
    c = p3r.Context(field=ctx.field, cap_height=cap_height)
    rng = np.random.default_rng(35)
    shapes = [(1 << 17, 3), (1 << 16, 9), (1 << 12, 2), (1 << 17, 6), (32, 5), (1 << 12, 8)]
    mats = [rand(rng, ctx.field, s) for s in shapes]
    cap, tree = c.commit(mats)
    ocap, otree = oracle.commit(ctx.field, mats, cap_height)
    assert np.array_equal(cap, ocap)
    for index in (0, 1, 4097, 77777, (1 << 17) - 1):
        opened, proof = tree.open_batch(index)
        oo, op = otree.open(index)
        assert np.array_equal(opened, oo) and np.array_equal(proof, op)
        assert oracle.verify(ctx.field, cap, shapes, index, opened, proof)
    tree.free()
    c.close()


def test_mmcs_golden_sponge(ctx, golden):
    g = golden["prim"][key_of(ctx)]
    for kat in g["sponge"]:
        cap, tree = ctx.commit([np.array([kat["in"]], dtype=np.uint32)])
        assert cap[0].tolist() == kat["out"]


def test_commit_device_resident_pipeline(ctx, oracle):
    """trace fill -> LDE -> commit without leaving HBM equals the oracle's pipeline."""
    rng = np.random.default_rng(99)
    n = 512
    rows = make_p2_rows(rng, ctx.field, n)
    g = oracle_lib.GENERATOR[ctx.field]
    tr = ctx.generate_trace_rows_device(*rows)
    other = rand(rng, ctx.field, (128, 4))
    d_other = ctx.upload(other)
    lde_a = ctx.coset_lde_batch_device(tr, 2, g)
    lde_b = ctx.coset_lde_batch_device(d_other, 2, g)
    cap, tree = ctx.commit_device([lde_a, lde_b])
    o_tr = oracle.trace_rows(ctx.field, *rows)
    o_a = oracle.coset_lde(ctx.field, o_tr, 2, g)
    o_b = oracle.coset_lde(ctx.field, other, 2, g)
    ocap, otree = oracle.commit(ctx.field, [o_a, o_b], 0)
    assert np.array_equal(cap, ocap)
    opened, proof = tree.open_batch(1234)
    oo, op = otree.open(1234)
    assert np.array_equal(opened, oo) and np.array_equal(proof, op)
    tree.free()
