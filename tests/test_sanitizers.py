"""CPU: the host-only code of libp3r_hip.so - proof parsers, native verifier, `Mmcs::verify_batch`, the host side of the
circuit boundary, include/p3r.hpp's `from_postcard` - built WITHOUT device code under AddressSanitizer and under UBSan
(tests/san/Makefile: `hipcc --cuda-host-only -fsanitize=address | undefined -fno-sanitize-recover=all`: two builds, the
combined one takes twenty minutes to compile) and driven by a
structure-aware mutator over real proofs and circuits (tests/san/san_driver.cpp).  A parent node of an aggregation tree
runs exactly this code on bytes from other ranks (plonky3_recursion_amd/aggregation.py: `_recv_bytes` ->
`BatchStarkProof.from_postcard` -> `verify_all_tables` / `prove_aggregation_layer`); the rules it has to enforce on
them are circuit-prover/src/batch_stark_prover.rs:459-488,666-681 and packing.rs:140-161.  Any sanitizer report aborts
the driver; an accepted mutant with different proof bytes fails it.

P3R_SAN_ITERS: mutants per proof case under AddressSanitizer (default 25000; four cases per field = 10^5 mutants per
field); the UBSan build (-O0: twenty times slower per mutant) sees a tenth of that."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import harness_lib
import layer_lib
import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "san")
DRIVERS = {"asan": os.path.join(SAN, "_build", "san_driver_asan"), "ubsan": os.path.join(SAN, "_build", "san_driver_ubsan")}
ITERS = int(os.environ.get("P3R_SAN_ITERS", "25000"))
SMALL = dict(horner_chain_len=8, sponge_chain_len=3, merkle_depth=3)
W = {"koala-bear": 3, "baby-bear": 11}
P2_NAME = {"koala-bear": "poseidon2_perm/koala_bear_d4_w16", "baby-bear": "poseidon2_perm/baby_bear_d4_w16"}


@pytest.fixture(scope="module")
def drivers():
    r = subprocess.run(["make", "-j2", "-C", SAN], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return DRIVERS


def run(driver, mode, case, iters, seed, jobs=4):
    """`jobs` driver processes with different seeds; returns their summaries.  LeakSanitizer is off (it needs ptrace,
    and there is nothing of interest to leak: every buffer is a std::vector or freed by the driver)."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    procs = [subprocess.Popen([driver, mode, case, str(max(iters // jobs, 1)), str(seed + 7919 * j)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for j in range(jobs)]
    out = []
    for p in procs:
        so, se = p.communicate()
        assert p.returncode == 0, (mode, case, p.returncode, so[-500:], se[-6000:])
        out.append(json.loads(so.strip().splitlines()[-1]))
    return out


def write_proof_case(path, oracle, field, prm, packing, flags=0, log_h=5):
    from plonky3_recursion_amd import prover as pv
    arrs = harness_lib.generate(field, log_h, seed=40 + log_h, flags=flags, **SMALL)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm, packing=packing)
    tables, cap, inner = L.tables(), L.prep_commit(), L.prove()
    counts = [int(x) for x in arrs["counts"]]
    tp = pv.TablePacking(min_trace_height=layer_lib.min_trace_height(prm), **packing)
    kinds = [t["kind"] for t in tables]
    npo = []
    if "poseidon2" in kinds:
        npo.append(pv.NonPrimitiveTableEntry(P2_NAME[field], tables[kinds.index("poseidon2")]["main"].shape[0], 1))
    if "recompose" in kinds:
        npo.append(pv.NonPrimitiveTableEntry("recompose", counts[4], packing["recompose_lanes"]))
    proof = pv.BatchStarkProof(
        proof=inner, table_packing=tp, rows=tuple(max(c, 1) for c in counts[:3]), w_binomial=W[field], non_primitives=tuple(npo),
        preprocessed_commitment=cap, preprocessed_widths=tuple(t["prep"].shape[1] for t in tables),
        degree_bits=tuple(int(t["main"].shape[0]).bit_length() - 1 + prm.zk for t in tables), monty_r=1, modulus=oracle_lib.MODULUS[field])
    outer = proof.to_postcard()
    with open(path, "wb") as fh:
        fh.write(b"P3RSAN1\0")
        fh.write(struct.pack("<14I", oracle_lib.FIELD_IDS[field], 4, prm.log_blowup, prm.max_log_arity, prm.cap_height,
                             prm.log_final_poly_len, prm.commit_pow_bits, prm.query_pow_bits, prm.num_queries, prm.challenge_degree or 4,
                             prm.mmcs_arity or 2, prm.zk, prm.num_random_codewords if prm.zk else 0, prm.mmcs_salt_elems))
        fh.write(struct.pack("<I", len(tables)))
        for t in tables:
            fh.write(struct.pack("<4I", t["kind_id"], t["lanes"], t["horner_k"], 0))
        c = np.ascontiguousarray(cap, dtype=np.uint32).reshape(-1)
        fh.write(struct.pack("<I", c.size) + c.tobytes())
        fh.write(struct.pack("<I", len(proof.degree_bits)) + np.array(proof.degree_bits, dtype=np.uint32).tobytes())
        fh.write(struct.pack("<Q", len(outer)) + outer)
    return len(outer)


PACK = dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3, recompose_lanes=2)
CASES = [
    ("plain", dict(log_blowup=1, max_log_arity=2, log_final_poly_len=0, query_pow_bits=2, num_queries=3), dict(PACK), 0),
    ("cap_pow", dict(log_blowup=1, max_log_arity=1, cap_height=2, log_final_poly_len=1, commit_pow_bits=2, query_pow_bits=2, num_queries=2),
     dict(public_lanes=1, alu_lanes=3, horner_packed_steps=4, recompose_lanes=1), 0),
    ("zk", dict(log_blowup=2, max_log_arity=2, log_final_poly_len=0, query_pow_bits=2, num_queries=2, zk=1, zk_seed=5), dict(PACK), 0),
    # the hiding MMCS under HidingFriPcs (recursion/tests/zk_hiding_mmcs.rs): opening proofs are (salts, siblings)
    ("zk_salted", dict(log_blowup=2, max_log_arity=2, log_final_poly_len=0, query_pow_bits=2, num_queries=2, zk=1, zk_seed=6, mmcs_salt_elems=4), dict(PACK), 0),
    ("arity4_no_npo", dict(log_blowup=1, max_log_arity=3, log_final_poly_len=0, query_pow_bits=2, num_queries=2, mmcs_arity=4),
     dict(PACK), harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE),
]


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
@pytest.mark.parametrize("name,kw,packing,flags", CASES)
def test_mutated_proofs_under_sanitizers(drivers, oracle, tmp_path, field, name, kw, packing, flags):
    case = str(tmp_path / f"{name}.case")
    n = write_proof_case(case, oracle, field, layer_lib.params(**kw), packing, flags)
    for san, iters in (("asan", ITERS), ("ubsan", max(ITERS // 10, 200))):
        res = run(drivers[san], "proofs", case, iters, seed=sum(map(ord, field + name + san)))
        total = {k: sum(r[k] for r in res) for k in res[0] if k != "mode"}
        print(san, field, name, n, "bytes:", total)
        # the mutator reaches past the parser (structure-preserving edits) and into the verifier; metadata-only edits verify
        assert total["parsed"] > total["iterations"] // 50
        assert total["rejected_by_verifier"] > 0 and total["rejected_by_parser"] > 0
        if name == "plain":
            res = run(drivers[san], "mmcs", case, 40000 if san == "asan" else 4000, seed=3)
            assert sum(r["refused"] for r in res) == sum(r["iterations"] for r in res)


@pytest.mark.parametrize("field,ext_degree", [("koala-bear", 4), ("baby-bear", 4)])
def test_mutated_circuits_under_sanitizers(drivers, tmp_path, field, ext_degree):
    """p3r_circuit_desc with zero / huge lane counts, witness ids out of range, `ext_off + ext_len` overflow, unknown op
    kinds, truncated op lists, rewrites of unknown witnesses: validate_circuit + the host preparation refuse or prepare,
    under the sanitizers."""
    arrs = harness_lib.generate(field, 6, seed=77, **SMALL)
    ops = np.ascontiguousarray(arrs["ops"], dtype=np.uint32).reshape(-1, 8)
    case = str(tmp_path / "circuit.case")
    with open(case, "wb") as fh:
        fh.write(b"P3RSANC\0")
        fh.write(struct.pack("<8I", oracle_lib.FIELD_IDS[field], ext_degree, int(arrs["counts"][5]), 1, 3, 4, 1, 32))
        for key, per in (("ops", 8), ("ext", 1), ("public_rows", 1), ("private_rows", 1), ("rewrite", 2)):
            a = np.ascontiguousarray(arrs[key], dtype=np.uint32).reshape(-1)
            fh.write(struct.pack("<Q", a.size // per) + a.tobytes())
    for san, iters in (("asan", max(ITERS // 2, 4000)), ("ubsan", max(ITERS // 10, 400))):
        res = run(drivers[san], "circuit", case, iters, seed=11)
        total = {k: sum(r[k] for r in res) for k in res[0] if k != "mode"}
        print(san, field, len(ops), "ops:", total)
        assert total["refused"] > 0 and total["prepared"] > 0
