"""Wide randomised sweep of the prover against the oracle: layer sizes, table packings, FRI parameters,
cap heights, proof-of-work bits, folding schedules and the selectable protocol details, many more
combinations than tests/test_gpu_layer.py and tests/test_gpu_prove.py run.  For every draw the proof
bytes of the HIP prover must equal the oracle's, and both verifiers must accept them.

usage: python tools/prove_sweep.py [first_seed] [count] [max_log_h] [min_log_h]     (needs a GPU; oracle = checker)
"""
import random
import sys
import time

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import numpy as np

import harness_adapters as wl
import harness_lib
import layer_lib
import oracle_lib
import plonky3_recursion_amd as p3r
from plonky3_recursion_amd import prover as pv

P3R_EXT_LOOKUP_UNPACKED = 1


def draw(rng, max_log_h, min_log_h=5):
    log_blowup = rng.choice([1, 1, 2, 2, 3])
    log_final = rng.randint(0, 3)
    # the smallest table is 2^(log_final + log_blowup + 1) rows; the layer's largest table 2^log_h
    log_h = rng.randint(max(min_log_h, log_final + log_blowup + 2), max_log_h)
    max_log_arity = rng.randint(1, 3)
    kw = dict(log_blowup=log_blowup, max_log_arity=max_log_arity, cap_height=rng.randint(0, 3),
              log_final_poly_len=log_final, commit_pow_bits=rng.choice([0, 0, 2, 5]),
              query_pow_bits=rng.randint(0, 7), num_queries=rng.randint(1, 9))
    if rng.random() < 0.3:
        kw["ext_choices"] = P3R_EXT_LOOKUP_UNPACKED
    r = rng.random()
    if r < 0.08:
        # an explicit folding schedule drawn blindly: one that does not fit the proof must be refused by both sides
        kw["fri_log_arities"] = [rng.randint(1, 3) for _ in range(rng.randint(1, 8))]
    elif r < 0.35:
        kw["fri_log_arities"] = "fitting"   # replaced in one() by a random schedule that fits the table heights
    if rng.random() < 0.25:
        # serialisation order of the proof's fields: a permutation per struct (batch 5, fri 5, opened values 8)
        kw["proof_layout"] = sum((rng.sample(range(n), n) for n in (5, 5, 8)), [])
    packing = dict(public_lanes=rng.randint(1, 3), alu_lanes=rng.randint(1, 4),
                   horner_packed_steps=rng.randint(2, 5), recompose_lanes=rng.randint(1, 2))
    gen = dict(horner_chain_len=rng.choice([0, 5, 20, 60]), sponge_chain_len=rng.randint(1, 6),
               merkle_depth=rng.randint(1, 12))
    field = rng.choice(["koala-bear", "baby-bear"])
    return field, log_h, kw, packing, gen


def fitting_schedule(rng, oracle, field, arrs, kw, packing):
    """A random folding schedule that reaches every roll-in height and the final height (the rule of
    include/p3r.h: each step at most max_log_arity, the distance to the next input height and to the end)."""
    base = {k: v for k, v in kw.items() if k not in ("fri_log_arities", "proof_layout", "mmcs_arity")}
    L = layer_lib.OracleLayer(oracle, field, arrs, layer_lib.params(**base), packing=dict(packing))
    lb = kw["log_blowup"]
    # FRI inputs: the LDEs of the trace domains and of the quotient chunks all live at log2(h) + log_blowup
    # (ZK: over the extended domains, one bit taller)
    hs = sorted({int(np.log2(t["main"].shape[0])) + lb + int(kw.get("zk", 0)) for t in L.tables()}, reverse=True)
    log_final = kw["log_final_poly_len"] + lb
    cur, out, nxt = hs[0], [], 1
    while cur > log_final:
        limit = cur - log_final
        if nxt < len(hs):
            limit = min(limit, cur - hs[nxt])
        la = rng.randint(1, max(1, min(kw["max_log_arity"], limit)))
        cur -= la
        if nxt < len(hs) and hs[nxt] == cur:
            nxt += 1
        out.append(la)
    return out


def one(oracle, seed, max_log_h, min_log_h=5):
    rng = random.Random(seed)
    field, log_h, kw, packing, gen = draw(rng, max_log_h, min_log_h)
    # circuit extension degree and table variants, from a second stream so that the draws above keep their seeds:
    # D = 5 (KoalaBear quintic circuits: primitive tables, + compact-D1 Poseidon2, + Recompose, + recompose/coeff) in
    # a third of the KoalaBear draws; recompose/coeff under D = 4 now and then
    rng2 = random.Random(seed * 7919 + 1)
    ext_degree, flags = 4, 0
    if field == "koala-bear" and rng2.random() < 0.33:
        ext_degree = 5
        flags = rng2.choice([harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE, harness_lib.NO_RECOMPOSE, 0,
                             harness_lib.RECOMPOSE_COEFF])
    elif rng2.random() < 0.25:
        # base-field circuits (D = 1, both fields), the same table variants
        ext_degree = 1
        flags = rng2.choice([harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE, harness_lib.NO_RECOMPOSE, 0,
                             harness_lib.RECOMPOSE_COEFF])
    elif rng2.random() < 0.15:
        flags = harness_lib.RECOMPOSE_COEFF
    # third stream: both Recompose tables in one layer (any degree), and KoalaBear's quintic CHALLENGE field
    rng3 = random.Random(seed * 104729 + 2)
    if not (flags & harness_lib.NO_RECOMPOSE) and rng3.random() < 0.3:
        flags = (flags & ~harness_lib.RECOMPOSE_COEFF) | harness_lib.RECOMPOSE_BOTH
    challenge_degree = 5 if field == "koala-bear" and rng3.random() < 0.3 else 4
    kw["challenge_degree"] = challenge_degree
    # fourth stream (round 4): the prover's own arity-4 MMCS in a third of the draws (a one-digest cap: cap_height 0 in four
    # of five of them, otherwise the drawn cap - which both sides must refuse), and the width-32 Poseidon2 table in a
    # quarter of the D = 4 layers that hold a width-16 one
    rng4 = random.Random(seed * 15485863 + 3)
    if rng4.random() < 0.33:
        kw["mmcs_arity"] = 4
        if rng4.random() < 0.8:
            kw["cap_height"] = 0
    if ext_degree == 4 and not (flags & harness_lib.NO_POSEIDON2) and rng4.random() < 0.25:
        flags |= harness_lib.P2_W32
    # fifth stream (round 5): the ZK configuration (HidingFriPcs: p3r_config.zk) in a third of the draws, with 1 - 3 random
    # codewords, a random seed and a random proof number - the oracle is given the same three, so the bytes must agree;
    # degree-3 constraints need log_blowup >= 2 under ZK, the other draws must be refused by both sides
    rng5 = random.Random(seed * 32452843 + 4)
    zk_nonce = 0
    if rng5.random() < 0.33:
        kw["zk"], kw["num_random_codewords"], kw["zk_seed"] = 1, rng5.randint(1, 3), rng5.getrandbits(48)
        zk_nonce = rng5.randint(0, 5)
    # sixth stream (round 6): the hiding MMCS (p3r_config.mmcs_salt_elems: MerkleTreeHidingMmcs for input and commit-phase
    # trees) in a quarter of the draws, 1 - 5 salt elements, with or without ZK, under either arity; the oracle is given the
    # same key and proof number
    rng6 = random.Random(seed * 49979687 + 5)
    if rng6.random() < 0.25:
        kw["mmcs_salt_elems"] = rng6.randint(1, 5)
        if "zk_seed" not in kw:
            kw["zk_seed"] = rng6.getrandbits(48)
            zk_nonce = rng6.randint(0, 3)
    # seventh stream (round 6): the CIRCUIT seam in a third of the draws - the flattened circuit prepared on the device
    # (csrc/prep_device.hip), run by the device runner and proved in one call (p3r_prove_next_layer) instead of
    # prove_all_tables over the generator's traces - and, in a third of the D = 4 layers that hold a width-16 table, the
    # width-32 rows as OPS of that circuit (P3R_OP_POSEIDON2_W32_PERM: leaf sponges seeding 4-to-1 chains)
    rng7 = random.Random(seed * 67867967 + 6)
    via_circuit = rng7.random() < 0.33
    if ext_degree == 4 and not (flags & harness_lib.NO_POSEIDON2) and rng7.random() < 0.33:
        flags |= harness_lib.P2_W32_OPS
    if via_circuit and (flags & harness_lib.P2_W32):
        flags |= harness_lib.P2_W32_OPS      # (a table-only width-32 layer has no circuit that fills it)
    coeff = bool(flags & harness_lib.RECOMPOSE_COEFF)
    arrs = harness_lib.generate(field, log_h, seed=seed, flags=flags, ext_degree=ext_degree, **gen)
    packing_o = dict(packing, ext_degree=ext_degree, recompose_coeff_lookups=int(coeff))
    if kw.get("fri_log_arities") == "fitting":
        kw["fri_log_arities"] = fitting_schedule(rng, oracle, field, arrs, kw, packing_o)
    prm = layer_lib.params(zk_nonce=zk_nonce, **kw)
    desc = f"seed {seed}: {field} D={ext_degree} DC={challenge_degree} flags={flags}{' via-circuit' if via_circuit else ''} 2^{log_h} {kw} {packing} {gen}"
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=dict(packing_o))
    try:
        want_cap, want = L.prep_commit(), L.prove()
    except RuntimeError as e:       # a configuration the protocol has no proof for: the prover must refuse it too
        want_cap, want = None, str(e)
    tp = pv.TablePacking(**packing)
    tp.with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    ctx = None
    try:
        ctx = p3r.Context(field=field, ext_degree=ext_degree, **kw, allow_unpinned_w32_defaults=True)
        if kw.get("zk") or kw.get("mmcs_salt_elems"):
            ctx.zk_nonce = zk_nonce   # (zk_seed makes the context deterministic: the oracle is given the same key and nonce)
        backend = pv.FriRecursionBackendD5() if ext_degree == 5 else pv.FriRecursionBackend()
        if via_circuit:
            cache = pv.build_next_layer_prep(ctx, wl.circuit_from_arrays(arrs), backend, pv.ProveNextLayerParams(table_packing=tp))
            assert cache.prepared_circuit.prepared_on_device, "device preparation handed over: " + desc
            cpd = cache.circuit_prover_data
            out = pv.prove_next_layer(pv.RecursionInput(circuit_inputs=wl.circuit_inputs_from_arrays(arrs, ext_degree)), ctx, backend,
                                      pv.ProveNextLayerParams(table_packing=tp), prep=cache).proof
        else:
            cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs, ext_degree=ext_degree, recompose_coeff_lookups=coeff),
                                             backend, pv.ProveNextLayerParams(table_packing=tp))
            cpd = cache.circuit_prover_data
            out = cache.prover.prove_all_tables(wl.traces_from_arrays(arrs, ext_degree=ext_degree), cpd)
    except p3r.P3rError as e:
        assert want_cap is None, "prover refused (%s) what the oracle proves: %s" % (e, desc)
        if ctx is not None:
            ctx.close()
        return desc + "  [refused by both: " + want + "]", 0
    assert want_cap is not None, "prover accepted what the oracle refuses (%s): %s" % (want, desc)
    assert np.array_equal(cpd.preprocessed_commitment, want_cap), "prep commitment: " + desc
    assert out.proof == want, "proof bytes: " + desc
    L.verify(out.proof)
    cache.prover.verify_all_tables(out)
    if cache.prepared_circuit is not None:
        cache.prepared_circuit.free()
    else:
        cpd.free()
    ctx.close()
    return desc, len(want)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    max_log_h = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    min_log_h = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    oracle = oracle_lib.Oracle()
    t0 = time.time()
    for seed in range(first, first + count):
        desc, n = one(oracle, seed, max_log_h, min_log_h)
        print(f"ok  {desc}  ({n} B)", flush=True)
    print(f"{count} draws agree with the oracle byte for byte ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
