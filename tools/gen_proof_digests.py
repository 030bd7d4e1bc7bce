"""Digests of the oracle's proof bytes for a fixed list of small layers -> tests/golden/proof_digests.json.
   python3 tools/gen_proof_digests.py
Not a parity pin (the reference holds no proof bytes): a DRIFT pin.  The oracle, the generator and
the device prover change together from round to round; these digests make a change of the proof bytes of an existing
configuration visible in review instead of silently re-agreeing with itself.  `workload` is the digest of the
generator's arrays, so a generator change is told apart from a prover change."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [
    # name, field, log_h, seed, flags, circuit degree, challenge degree, FRI parameters, packing
    ("d4_default", "koala-bear", 7, 11, 0, 4, 4, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=6), {}),
    ("d4_babybear_cap2", "baby-bear", 8, 12, 0, 4, 4, dict(log_blowup=1, max_log_arity=3, log_final_poly_len=1, cap_height=2, commit_pow_bits=2, query_pow_bits=3, num_queries=5),
     dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3)),
    ("d4_recompose_coeff", "koala-bear", 7, 13, 32, 4, 4, dict(log_blowup=2, max_log_arity=1, log_final_poly_len=1, query_pow_bits=3, num_queries=4), {}),
    ("d1_base_field", "baby-bear", 7, 14, 0, 1, 4, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4), dict(alu_lanes=1, horner_packed_steps=2)),
    ("d5_backend_tables", "koala-bear", 8, 15, 64, 5, 4, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=5), {}),
    ("d5_quintic_challenge", "koala-bear", 8, 16, 64, 5, 5, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=5),
     dict(public_lanes=1, alu_lanes=8, horner_packed_steps=2)),
    ("d1_quintic_challenge", "koala-bear", 7, 17, 2, 1, 5, dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4), {}),
    ("d4_arity4_mmcs", "koala-bear", 7, 19, 0, 4, 4, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4, mmcs_arity=4), {}),
    ("d4_arity4_mmcs_w32_table", "baby-bear", 7, 20, 128, 4, 4, dict(log_blowup=1, max_log_arity=3, log_final_poly_len=1, query_pow_bits=3, num_queries=4, mmcs_arity=4), {}),
    # round 6: the width-32 rows as ops of the circuit (flag 4096 | 128), under the arity-4 MMCS: the arrays are the generator's own books
    ("d4_arity4_w32_ops", "koala-bear", 7, 21, 4096 | 128, 4, 4, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4, mmcs_arity=4), {}),
    ("d8_binomial", "koala-bear", 7, 18, 1, 8, 4, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4), dict(ext_w=3)),
]
GEN = dict(horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)


def layer(oracle, case):
    import harness_lib
    import layer_lib
    name, field, log_h, seed, flags, d, dc, fri, packing = case
    arrs = harness_lib.generate(field, log_h, seed=seed, flags=flags, ext_degree=d, **GEN)
    prm = layer_lib.params(challenge_degree=dc, **fri)
    pk = dict(packing, ext_degree=d, recompose_coeff_lookups=1 if flags & harness_lib.RECOMPOSE_COEFF else 0)
    return arrs, prm, layer_lib.OracleLayer(oracle, field, arrs, prm, packing=pk)


def workload_digest(arrs):
    """sha256 over the generator's arrays.  Arrays added after the pins were made (the width-32 Poseidon2 table's,
    round 4) are left out while they are empty, and `counts` is hashed without its trailing zero entries for such
    tables - so the pins of the older layers keep telling a generator change apart from a layout extension."""
    late = ("p2w_inputs", "p2w_flags", "p2w_mmcs_index_sum", "p2w_prep", "pdw_op_ids", "pdw_siblings")   # (pdw_*: round 6)
    h = hashlib.sha256()
    for k in sorted(arrs):
        a = arrs[k]
        if k in late and not len(a):
            continue
        if k == "counts" and len(a) > 7 and not a[7:].any():
            a = a[:7]
        h.update(k.encode())
        h.update(a.tobytes())
    return h.hexdigest()


def main():
    import oracle_lib
    oracle = oracle_lib.Oracle()
    out = {"provenance": "tools/gen_proof_digests.py: sha256 of the oracle's prove_batch bytes (Montgomery and canonical "
                         "encodings) and of the preprocessed commitment; a drift pin, not a parity pin",
           "cases": {}}
    for case in CASES:
        arrs, prm, L = layer(oracle, case)
        out["cases"][case[0]] = {
            "workload": workload_digest(arrs),
            "prep_commit": hashlib.sha256(L.prep_commit().tobytes()).hexdigest(),
            "proof": hashlib.sha256(L.prove()).hexdigest(),
            "proof_canonical": hashlib.sha256(L.prove(field_encoding=1)).hexdigest(),
            "proof_bytes": len(L.prove()),
        }
    path = os.path.join(ROOT, "tests", "golden", "proof_digests.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
