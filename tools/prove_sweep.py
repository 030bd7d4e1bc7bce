"""Wide randomised sweep of the prover against the oracle: layer sizes, table packings, FRI parameters,
cap heights, proof-of-work bits, folding schedules and the selectable protocol details, many more
combinations than tests/test_gpu_layer.py and tests/test_gpu_prove.py run.  For every draw the proof
bytes of the HIP prover must equal the oracle's, and both verifiers must accept them.

usage: python tools/prove_sweep.py [first_seed] [count] [max_log_h]     (needs a GPU; oracle = checker)
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


def draw(rng, max_log_h):
    log_blowup = rng.choice([1, 1, 2, 2, 3])
    log_final = rng.randint(0, 3)
    # the smallest table is 2^(log_final + log_blowup + 1) rows; the layer's largest table 2^log_h
    log_h = rng.randint(max(5, log_final + log_blowup + 2), max_log_h)
    max_log_arity = rng.randint(1, 3)
    kw = dict(log_blowup=log_blowup, max_log_arity=max_log_arity, cap_height=rng.randint(0, 3),
              log_final_poly_len=log_final, commit_pow_bits=rng.choice([0, 0, 2, 5]),
              query_pow_bits=rng.randint(0, 7), num_queries=rng.randint(1, 9))
    if rng.random() < 0.3:
        kw["ext_choices"] = P3R_EXT_LOOKUP_UNPACKED
    packing = dict(public_lanes=rng.randint(1, 3), alu_lanes=rng.randint(1, 4),
                   horner_packed_steps=rng.randint(2, 5), recompose_lanes=rng.randint(1, 2))
    gen = dict(horner_chain_len=rng.choice([0, 5, 20, 60]), sponge_chain_len=rng.randint(1, 6),
               merkle_depth=rng.randint(1, 12))
    field = rng.choice(["koala-bear", "baby-bear"])
    return field, log_h, kw, packing, gen


def one(oracle, seed, max_log_h):
    rng = random.Random(seed)
    field, log_h, kw, packing, gen = draw(rng, max_log_h)
    arrs = harness_lib.generate(field, log_h, seed=seed, **gen)
    prm = layer_lib.params(**kw)
    desc = f"seed {seed}: {field} 2^{log_h} {kw} {packing} {gen}"
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=dict(packing))
    try:
        want_cap, want = L.prep_commit(), L.prove()
    except RuntimeError as e:       # a configuration the protocol has no proof for: the prover must refuse it too
        want_cap, want = None, str(e)
    ctx = p3r.Context(field=field, **kw)
    tp = pv.TablePacking(**packing)
    tp.with_fri_params(prm.log_final_poly_len, prm.log_blowup)
    try:
        cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs), pv.FriRecursionBackend(),
                                         pv.ProveNextLayerParams(table_packing=tp))
        cpd = cache.circuit_prover_data
        out = cache.prover.prove_all_tables(wl.traces_from_arrays(arrs), cpd)
    except p3r.P3rError as e:
        assert want_cap is None, "prover refused (%s) what the oracle proves: %s" % (e, desc)
        ctx.close()
        return desc + "  [refused by both: " + want + "]", 0
    assert want_cap is not None, "prover accepted what the oracle refuses (%s): %s" % (want, desc)
    assert np.array_equal(cpd.preprocessed_commitment, want_cap), "prep commitment: " + desc
    assert out.proof == want, "proof bytes: " + desc
    L.verify(out.proof)
    cache.prover.verify_all_tables(out)
    cpd.free()
    ctx.close()
    return desc, len(want)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    max_log_h = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    oracle = oracle_lib.Oracle()
    t0 = time.time()
    for seed in range(first, first + count):
        desc, n = one(oracle, seed, max_log_h)
        print(f"ok  {desc}  ({n} B)", flush=True)
    print(f"{count} draws agree with the oracle byte for byte ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
