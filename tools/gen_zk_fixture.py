#!/usr/bin/env python3
"""tests/golden/zk_fibonacci_layer_for_rust_<field>.json: a ZK proof made HERE (CPU oracle prover, the same bytes the
HIP prover emits: tests/test_gpu_zk.py) for the reference's verifier to judge.  A randomised proof has no byte parity
to claim; what can be pinned is ACCEPTANCE, in both directions:
    Rust prover  -> this repo's verifiers   tests/golden/rust_fibonacci_zk_layer_<field>.json (tools/rust_pin), checked by
                                            tests/test_rust_pins.py::test_rust_zk_proof_is_accepted
    this prover  -> Rust's verify_all_tables   THIS fixture, read by `cargo run -- zk-accept` (tools/rust_pin), which
                                            writes tests/golden/rust_zk_acceptance.json for
                                            tests/test_rust_pins.py::test_rust_accepts_our_zk_proof
The circuit is the Fibonacci(n = 100) layer of the other fixtures (recursive_fibonacci.rs:315-337 over the degree-4
extension) under the FRI parameters of tools/rust_pin with create_config_zk's two random codewords.  Round constants:
rust_primitives.json's when it exists (upstream's statics), else this repo's self-generated defaults - recorded in the
fixture, since a Rust verifier can only accept a proof made with its own permutation.
Run: python tools/gen_zk_fixture.py     (CPU only; data, not source)"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=2, commit_pow_bits=0, query_pow_bits=6, num_queries=8)
FIB_N, SEED = 100, 1
W = {"koala-bear": 3, "baby-bear": 11}


def make(field, key):
    import circuit_lib as cl
    import fib_lib
    import layer_lib
    import oracle_lib
    from plonky3_recursion_amd import prover as pv
    orc = oracle_lib.Oracle()
    rust = os.path.join(ROOT, "tests", "golden", "rust_primitives.json")
    if os.path.exists(rust):
        rc, rc_src = np.array(json.load(open(rust))["fields"][key]["rc"], dtype=np.uint32), "rust_primitives.json (upstream statics)"
    else:
        rc, rc_src = oracle_lib.default_rc(field), "self-generated defaults (tests/golden/poseidon2_rc_default.json): UNPINNED"
    circuit, inputs, fib = fib_lib.fibonacci_circuit(FIB_N, oracle_lib.MODULUS[field])
    oc = cl.OracleCircuit(orc, circuit).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, inputs, rc=rc)
    prm = layer_lib.params(zk=1, num_random_codewords=2, zk_seed=SEED, **FRI)
    packing = dict(public_lanes=1, alu_lanes=1, horner_packed_steps=2)
    L = layer_lib.OracleLayer(orc, field, oc.workload_arrays(), prm, packing=packing, rc=rc)
    tables, cap, inner = L.tables(), L.prep_commit(), L.prove()
    L.verify(inner)
    tp = pv.TablePacking(min_trace_height=layer_lib.min_trace_height(prm), **packing)
    proof = pv.BatchStarkProof(
        proof=inner, table_packing=tp, rows=(2, 1, FIB_N - 1), w_binomial=W[field], non_primitives=(),
        preprocessed_commitment=cap, preprocessed_widths=tuple(t["prep"].shape[1] for t in tables),
        degree_bits=tuple(int(t["main"].shape[0]).bit_length() for t in tables),   # EXTENDED degree bits (recursion.rs:374)
        monty_r=1, modulus=oracle_lib.MODULUS[field])
    outer = proof.to_postcard()
    return {
        "provenance": "tools/gen_zk_fixture.py: made by this repo's CPU oracle prover under p3r_config.zk = 1; NOT a reference output",
        "field": key, "n": FIB_N, "fib": int(fib), "fri": FRI, "zk": {"num_random_codewords": 2, "seed": SEED, "nonce": 0},
        "packing": packing, "round_constants": rc_src, "rc": rc.tolist(),
        "degree_bits": list(proof.degree_bits),
        "batch_stark_proof_postcard_hex": outer.hex(), "batch_proof_len": len(inner),   # the inner BatchProof is the prefix
        "sha256": hashlib.sha256(outer).hexdigest(),
    }


if __name__ == "__main__":
    for field, key in (("koala-bear", "koala_bear"), ("baby-bear", "baby_bear")):
        out = make(field, key)
        path = os.path.join(ROOT, "tests", "golden", f"zk_fibonacci_layer_for_rust_{key}.json")
        with open(path, "w") as fh:
            json.dump(out, fh)
        print(path, len(out["batch_stark_proof_postcard_hex"]) // 2, "bytes", out["sha256"][:16])
