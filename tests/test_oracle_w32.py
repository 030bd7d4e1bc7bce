"""CPU: the width-32 Poseidon2 table of the arity-4 MMCS (Poseidon2Config::{KOALA,BABY}_BEAR_D4_W32,
poseidon2-circuit-air/src/air.rs:1178-1342 `eval_arity4`, circuit-prover/tests/arity4_mmcs.rs) in the oracle: layers that
hold it next to the width-16 table are proved and verified, the table has the reference's shape, and broken rows -
a wrong direction bit, a running hash placed in the wrong chunk, a tampered zero pad, a wrong index accumulator,
an unbalanced bus - are rejected.  (The permutation's constants are self-generated defaults: parity unpinned.)"""
import numpy as np
import pytest

import harness_lib
import layer_lib
import oracle_lib

FRI = dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
GEN = dict(horner_chain_len=16, sponge_chain_len=3, merkle_depth=6)


def layer(oracle, field, log_h=7, seed=11, flags=0, edit=None, **prm):
    a = harness_lib.generate(field, log_h, seed=seed, flags=harness_lib.P2_W32 | flags, **GEN)
    if edit:
        edit(a)
    return a, layer_lib.OracleLayer(oracle, field, a, layer_lib.params(**dict(FRI, **prm)))


@pytest.mark.parametrize("field,cols", [("koala-bear", 32 + 8 * 32 + 31 + 4), ("baby-bear", 32 + 8 * 64 + 30 * 2 + 4)])
def test_table_shape_and_round_trip(oracle, field, cols):
    a, L = layer(oracle, field)
    kinds = [t["kind"] for t in L.tables()]
    assert kinds == ["const", "public", "alu", "poseidon2", "poseidon2_w32", "recompose"]   # proved right after the width-16 table
    t = L.tables()[4]
    assert t["main"].shape[1] == cols and t["prep"].shape[1] == 48 and t["main"].shape[0] == t["prep"].shape[0]
    n = int(a["counts"][7])
    fl = a["p2w_flags"].reshape(-1, 4)
    pc = cols - 4
    # arity-4 layout: [Poseidon2Cols | bit | bit2 | bit * bit2 | index_sum]
    assert np.array_equal(t["main"][:n, pc], fl[:, 2]) and np.array_equal(t["main"][:n, pc + 1], fl[:, 3])
    assert np.array_equal(t["main"][:n, pc + 2], fl[:, 2] * fl[:, 3])
    # base-four accumulator on Merkle continuation rows
    acc = t["main"][:n, pc + 3].astype(np.int64)
    P = oracle_lib.MODULUS[field]
    for r in range(1, n):
        if fl[r, 1] and not fl[r, 0]:
            assert acc[r] == (4 * acc[r - 1] + fl[r, 2] + 2 * fl[r, 3]) % P
    # padding: filler rows are chain starts; the first padding row of the preprocessed trace carries new_start = 1
    if t["prep"].shape[0] > n:
        assert t["prep"][n, 46] == 1 and not t["prep"][n, :46].any() and not t["prep"][n + 1:].any()
    proof = L.prove()
    L.verify(proof)
    with pytest.raises(RuntimeError):
        L.verify(proof[:-1] + bytes([proof[-1] ^ 1]))


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
@pytest.mark.parametrize("prm", [dict(), dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, cap_height=1),
                                 dict(ext_choices=1)])
def test_parameter_sets(oracle, field, prm):
    a, L = layer(oracle, field, log_h=8, seed=5, **prm)
    L.verify(L.prove())


def _first(a, pred):
    fl = a["p2w_flags"].reshape(-1, 4)
    prep = a["p2w_prep"].reshape(-1, 48)
    for r in range(len(fl)):
        if pred(r, fl, prep):
            return r
    raise AssertionError("the synthetic layer has no such row")


def rejected(oracle, field, edit):
    a, L = layer(oracle, field, edit=edit)
    proof = L.prove()   # the oracle prover does not check constraints: its proof is what a cheating prover would send
    with pytest.raises(RuntimeError):
        L.verify(proof)


def test_broken_rows_are_rejected(oracle):
    field = "koala-bear"
    P = oracle_lib.MODULUS[field]

    def flip_bit(a):       # the running hash then sits in the wrong chunk
        r = _first(a, lambda r, fl, pr: fl[r, 1] and not fl[r, 0] and not pr[r, 4 * 2 + 1])
        a["p2w_flags"].reshape(-1, 4)[r, 2] ^= 1
    rejected(oracle, field, flip_bit)

    def break_placement(a):  # a continuation row whose chunk `pos` is not the previous digest
        r = _first(a, lambda r, fl, pr: fl[r, 1] and not fl[r, 0])
        fl = a["p2w_flags"].reshape(-1, 4)
        pos = int(fl[r, 2] + 2 * fl[r, 3])
        a["p2w_inputs"].reshape(-1, 32)[r, 8 * pos] = (int(a["p2w_inputs"].reshape(-1, 32)[r, 8 * pos]) + 1) % P
    rejected(oracle, field, break_placement)

    def tamper_pad(a):     # a CTL-loaded zero pad of an injection / bridge level made non-zero: the bus no longer balances
        r = _first(a, lambda r, fl, pr: fl[r, 1] and pr[r, 4 * 7 + 1])
        a["p2w_inputs"].reshape(-1, 32)[r, 4 * 7] = 1
    rejected(oracle, field, tamper_pad)

    def wrong_out_ctl(a):  # an exposed output with the wrong multiplicity
        r = _first(a, lambda r, fl, pr: pr[r, 32] != 0)
        pr = a["p2w_prep"].reshape(-1, 48)
        pr[r, 33] = (int(pr[r, 33]) + 1) % P
    rejected(oracle, field, wrong_out_ctl)

    def non_boolean_bit2(a):   # bit2 := 2 breaks booleanity (and the product column)
        r = _first(a, lambda r, fl, pr: fl[r, 1])
        a["p2w_flags"].reshape(-1, 4)[r, 3] = 1 - a["p2w_flags"].reshape(-1, 4)[r, 3]
    rejected(oracle, field, non_boolean_bit2)


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
@pytest.mark.parametrize("prm", [dict(), dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, cap_height=1)])
def test_native_verifier_agrees_with_the_oracle_on_width32_layers(oracle, field, prm):
    """Two independently written verifiers on the oracle's proofs of layers that hold the width-32 table (CPU only: the
    native verifier is host code of the C-ABI library): accepted by both; a broken row's proof rejected by both."""
    import plonky3_recursion_amd as p3r

    def native(L, proof):
        t = L.tables()
        cfg, keep = p3r.make_config(field, **{k: getattr(L.prm, k) for k in ("log_blowup", "max_log_arity", "cap_height", "log_final_poly_len",
                                                                            "commit_pow_bits", "query_pow_bits", "num_queries")}, allow_unpinned_w32_defaults=True)
        airs = [dict(kind=x["kind_id"], lanes=x["lanes"], horner_packed_steps=x["horner_k"]) for x in t]
        assert 5 in [a["kind"] for a in airs]
        p3r.verify_batch(cfg, airs, L.prep_commit(), [int(x["main"].shape[0]).bit_length() - 1 for x in t], proof)

    a, L = layer(oracle, field, log_h=8, seed=21, **prm)
    proof = L.prove()
    L.verify(proof)
    native(L, proof)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    with pytest.raises(p3r.P3rError):
        native(L, bytes(bad))

    def flip_bit(a):
        fl = a["p2w_flags"].reshape(-1, 4)
        r = next(r for r in range(len(fl)) if fl[r, 1] and not fl[r, 0])
        fl[r, 2] ^= 1
    a2, L2 = layer(oracle, field, log_h=8, seed=21, edit=flip_bit, **prm)
    cheat = L2.prove()
    with pytest.raises(RuntimeError):
        L2.verify(cheat)
    with pytest.raises(p3r.P3rError):
        native(L2, cheat)
