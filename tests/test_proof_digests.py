"""Drift pins: sha256 of the proof bytes (both encodings) and of the preprocessed commitment of eight small layers -
every circuit degree, both challenge fields, five- and six-table layers - as committed in tests/golden/proof_digests.json
(tools/gen_proof_digests.py).  The oracle, the generator and the device prover evolve together; a change of the bytes of
an existing configuration has to show up as a change of this fixture.  `workload` tells a generator change apart."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_proof_digests", os.path.join(ROOT, "tools", "gen_proof_digests.py"))
gpd = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gpd)
PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "proof_digests.json")))["cases"]


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.mark.parametrize("case", gpd.CASES, ids=[c[0] for c in gpd.CASES])
def test_oracle_reproduces_the_committed_digests(oracle, case):
    pin = PINS[case[0]]
    arrs, prm, L = gpd.layer(oracle, case)
    assert gpd.workload_digest(arrs) == pin["workload"], "the generator's arrays changed (harness/synth.cpp)"
    assert sha(L.prep_commit().tobytes()) == pin["prep_commit"]
    proof = L.prove()
    assert len(proof) == pin["proof_bytes"] and sha(proof) == pin["proof"]
    assert sha(L.prove(field_encoding=1)) == pin["proof_canonical"]
    L.verify(proof)


@pytest.mark.gpu
@pytest.mark.parametrize("case", gpd.CASES, ids=[c[0] for c in gpd.CASES])
def test_device_reproduces_the_committed_digests(oracle, case):
    import harness_adapters as wl
    import harness_lib
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    name, field, log_h, seed, flags, d, dc, fri, packing = case
    pin = PINS[name]
    arrs = harness_lib.generate(field, log_h, seed=seed, flags=flags, ext_degree=d, **gpd.GEN)
    assert gpd.workload_digest(arrs) == pin["workload"]
    pk = dict(packing)
    ext_w = pk.pop("ext_w", 0)
    ctx = p3r.Context(field=field, ext_degree=d, ext_w=ext_w, challenge_degree=dc, **fri, allow_unpinned_w32_defaults=True)
    tp = pv.TablePacking(**pk).with_fri_params(fri["log_final_poly_len"], fri["log_blowup"])
    coeff = bool(flags & harness_lib.RECOMPOSE_COEFF)
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs, ext_degree=d, recompose_coeff_lookups=coeff),
                                     pv.FriRecursionBackend(), pv.ProveNextLayerParams(table_packing=tp))
    cpd = cache.circuit_prover_data
    assert sha(np.ascontiguousarray(cpd.preprocessed_commitment).tobytes()) == pin["prep_commit"]
    traces = wl.traces_from_arrays(arrs, ext_degree=d)
    assert sha(cache.prover.prove_all_tables(traces, cpd).proof) == pin["proof"]
    assert sha(cache.prover.prove_all_tables(traces, cpd, canonical_field_encoding=True).proof) == pin["proof_canonical"]
    cpd.free()
    ctx.close()
