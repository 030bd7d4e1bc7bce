"""CPU: tools/resolve_pins.py - the configuration search that turns the first run of tools/rust_pin into a `p3r_config`
(or into the name of the first structure no switch reproduces).  Without cargo there is no reference-made fixture to feed
it, so it is fed ORACLE proofs made under non-default switches, dressed as fixtures: it has to recover every switch
(field encoding, LogUp packing, FRI folding schedule, struct field order), to say so when the only difference is the
proof-of-work witness, and to name the stage where a foreign permutation's transcript parts ways."""
import os
import sys

import numpy as np
import pytest

import fib_lib
import circuit_lib as cl
import layer_lib
import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import resolve_pins  # noqa: E402

FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=2, commit_pow_bits=0, query_pow_bits=6, num_queries=8)
LAYOUT = [4, 0, 2, 1, 3] + [3, 4, 0, 2, 1] + [0, 2, 3, 1, 6, 7, 4, 5]
KEY = {"koala-bear": "koala_bear", "baby-bear": "baby_bear"}


def fake_fixture(oracle, field, rc=None, enc=0, fri=FRI, n=40, **switches):
    """What tools/rust_pin::fibonacci_layer would write if the reference behaved like the oracle under `switches`."""
    rc = oracle_lib.default_rc(field) if rc is None else rc
    circuit, inputs, fib = fib_lib.fibonacci_circuit(n, oracle_lib.MODULUS[field])
    oc = cl.OracleCircuit(oracle, circuit).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, inputs, rc=rc)
    packing = dict(public_lanes=1, alu_lanes=1, horner_packed_steps=2)
    L = layer_lib.OracleLayer(oracle, field, oc.workload_arrays(), layer_lib.params(**fri, **switches), packing=packing, rc=rc)
    return dict(field=KEY[field], n=n, fib=int(fib), fri=dict(fri), packing=packing, rc=[int(x) for x in rc],
                batch_proof_postcard_hex=L.prove(field_encoding=enc).hex())


@pytest.mark.parametrize("field,enc,switches", [
    ("koala-bear", 0, {}),
    ("baby-bear", 1, dict(ext_choices=1)),
    ("koala-bear", 0, dict(fri_log_arities=[1, 1, 1, 1, 1, 1])),
    ("koala-bear", 1, dict(proof_layout=LAYOUT)),
    ("baby-bear", 0, dict(ext_choices=1, fri_log_arities=[1, 1, 2], proof_layout=LAYOUT)),
])
def test_switches_are_recovered(oracle, field, enc, switches):
    fx = fake_fixture(oracle, field, enc=enc, **switches)
    out = resolve_pins.resolve(fx, oracle, log=lambda *a: None)
    assert out["resolved"], out
    cfg = out["config"]
    assert cfg["canonical_field_encoding"] == bool(enc)
    assert cfg["ext_choices"] == switches.get("ext_choices", 0)
    if "fri_log_arities" in switches:   # the phases the proof really has (a longer list's tail is not used)
        import proof_codec as pc
        d = pc.decode(bytes.fromhex(fx["batch_proof_postcard_hex"]) if "proof_layout" not in switches else
                      bytes.fromhex(fake_fixture(oracle, field, enc=enc, **{k: v for k, v in switches.items() if k != "proof_layout"})["batch_proof_postcard_hex"]))
        assert cfg["fri_log_arities"] == [st["log_arity"] for st in d["opening_proof"]["query_proofs"][0]["commit_phase_openings"]]
    else:
        assert cfg["fri_log_arities"] is None
    assert cfg["proof_layout"] == switches.get("proof_layout")
    assert cfg["poseidon2_rc"] == "built-in" and not out["notes"]
    # the configuration it prints does reproduce the bytes
    prm = layer_lib.params(ext_choices=cfg["ext_choices"], fri_log_arities=cfg["fri_log_arities"], proof_layout=cfg["proof_layout"], **FRI)
    assert cfg["proof_layout"] == switches.get("proof_layout")
    circuit, inputs, fib = fib_lib.fibonacci_circuit(fx["n"], oracle_lib.MODULUS[field])
    oc = cl.OracleCircuit(oracle, circuit).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, inputs)
    L = layer_lib.OracleLayer(oracle, field, oc.workload_arrays(), prm, packing=fx["packing"])
    assert L.prove(field_encoding=enc).hex() == fx["batch_proof_postcard_hex"]


def test_fixture_round_constants_are_used_and_reported(oracle):
    """Upstream's statics differ from the built-in table: the search runs on the fixture's and says so."""
    field = "koala-bear"
    rc = oracle_lib.default_rc(field).copy()
    rc[5] = (int(rc[5]) + 1) % oracle_lib.MODULUS[field]
    out = resolve_pins.resolve(fake_fixture(oracle, field, rc=rc), oracle, log=lambda *a: None)
    assert out["resolved"] and out["config"]["poseidon2_rc"] == "fixture" and "round constants differ" in out["notes"][0]


def test_other_pow_witness_is_diagnosed(oracle):
    """A reference that returns another valid proof-of-work witness (upstream searches in parallel): nothing after the
    witness agrees, and the tool says that this is the ONLY difference."""
    field = "koala-bear"
    base = fake_fixture(oracle, field)
    import proof_codec as pc
    d = pc.decode(bytes.fromhex(base["batch_proof_postcard_hex"]))
    P, R = oracle_lib.MODULUS[field], 1 << 32
    smallest = d["opening_proof"]["query_pow_witness"] * pow(R, -1, P) % P
    # the next valid witness: force candidates upwards until the oracle accepts one
    other = None
    for w in range(smallest + 1, smallest + 4000):
        try:
            fx = fake_fixture(oracle, field, forced_pow=[w])
            other = fx
            break
        except RuntimeError:
            continue
    assert other is not None and other["batch_proof_postcard_hex"] != base["batch_proof_postcard_hex"]
    out = resolve_pins.resolve(other, oracle, log=lambda *a: None)
    assert not out["resolved"]
    assert "query_pow_witness" in out["diagnosis"] or "query_proofs" in out["diagnosis"]
    assert "only difference is the PoW witness rule" in out["diagnosis"], out["diagnosis"]


def test_a_different_transcript_is_localised(oracle):
    """Proof bytes made with OTHER round constants than the fixture claims (= a Poseidon2 whose linear layers differ from
    this repo's): the very first commitment differs, and that is what is reported."""
    field = "baby-bear"
    rc2 = oracle_lib.default_rc(field).copy()
    rc2[0] = (int(rc2[0]) + 1) % oracle_lib.MODULUS[field]
    fx = fake_fixture(oracle, field, rc=rc2)
    fx["rc"] = [int(x) for x in oracle_lib.default_rc(field)]
    out = resolve_pins.resolve(fx, oracle, log=lambda *a: None)
    assert not out["resolved"] and "commitments" in out["diagnosis"], out


def test_zk_fixture_is_declined(oracle):
    fx = fake_fixture(oracle, "koala-bear", zk=1, zk_seed=1)
    fx["zk"] = {"num_random_codewords": 2, "seed": 1}
    out = resolve_pins.resolve(fx, oracle, log=lambda *a: None)
    assert not out["resolved"] and "randomised" in out["diagnosis"]
