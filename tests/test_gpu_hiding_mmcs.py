"""GPU: the hiding MMCS (p3r_config.mmcs_salt_elems; MerkleTreeHidingMmcs for the input and the FRI commit-phase MMCS -
recursion/tests/zk_hiding_mmcs.rs, rules recursion/src/pcs/mmcs.rs:315-510,763-790) on the device prover: salt matrices
drawn on the device from the context's keyed generator, leaf preimages [row | salt] per matrix (the plain kernels over the
matrix list [M0, S0, M1, S1, ..]), commit-phase leaves with their salts in the strided layout, opening proofs
(salts, siblings).  Under a fixed key the bytes equal the oracle's; both verifiers accept; with and without ZK; both arities."""
import numpy as np
import pytest

import harness_lib
import layer_lib
from test_hiding_mmcs import native_verify

pytestmark = pytest.mark.gpu
GEN = dict(horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)
CASES = [
    ("koala-bear", 7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4), 4),
    ("koala-bear", 9, dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, cap_height=1, query_pow_bits=4, num_queries=5, zk=1), 4),
    ("baby-bear", 8, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=0, commit_pow_bits=3, query_pow_bits=3, num_queries=4, zk=1), 4),
    ("baby-bear", 6, dict(log_blowup=1, max_log_arity=1, log_final_poly_len=1, query_pow_bits=3, num_queries=3), 1),
    ("koala-bear", 8, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4, mmcs_arity=4, zk=1), 4),
    ("baby-bear", 7, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4, mmcs_arity=4), 7),
    # tall enough for the two-pass NTTs, the FP64 leaf kernel over salted classes and multi-launch Merkle levels
    ("koala-bear", 12, dict(log_blowup=2, max_log_arity=2, log_final_poly_len=4, query_pow_bits=6, num_queries=6, zk=1), 4),
]


@pytest.mark.parametrize("field,log_h,kw,salt", CASES)
def test_salted_proof_bytes_equal_oracle(oracle, field, log_h, kw, salt):
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    arrs = harness_lib.generate(field, log_h, seed=100 + log_h, **GEN)
    prm = layer_lib.params(mmcs_salt_elems=salt, zk_seed=21, **kw)
    L = layer_lib.OracleLayer(oracle, field, arrs, prm)
    ctx = p3r.Context(field=field, mmcs_salt_elems=salt, zk_seed=21, allow_unpinned_w32_defaults=True, **kw)
    tp = p3r.TablePacking().with_fri_params(kw["log_final_poly_len"], kw["log_blowup"])
    cpd = p3r.CircuitProverData(ctx, wl.circuit_prep_from_arrays(arrs), tp)
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())     # salted too, same key: the same commitment
    prover = p3r.BatchStarkProver(ctx)
    traces = wl.traces_from_arrays(arrs)
    first = prover.prove_all_tables(traces, cpd)
    assert first.proof == L.prove()
    second = prover.prove_all_tables(traces, cpd)
    assert second.proof != first.proof and ctx.zk_nonce == 2          # fresh salts per proof (the nonce counts under salts too)
    prm1 = layer_lib.params(mmcs_salt_elems=salt, zk_seed=21, zk_nonce=1, **kw)
    assert second.proof == layer_lib.OracleLayer(oracle, field, arrs, prm1).prove()
    tables, cap = L.tables(), L.prep_commit()
    for pf in (first.proof, second.proof):
        L.verify(pf)
        native_verify(field, prm, tables, cap, pf)
    prover.verify_all_tables(first)
    # the wire form: the parser is told the proof type (P3R_PROOF_SALTED), as it is told ZK
    wire = first.to_postcard()
    back = p3r.BatchStarkProof.from_postcard(wire, field, zk=bool(kw.get("zk")), salted=True)
    assert back.proof == first.proof
    prover.verify_all_tables(back)
    with pytest.raises(p3r.P3rError):
        p3r.BatchStarkProof.from_postcard(wire, field, zk=bool(kw.get("zk")))
    cpd.free()
    ctx.close()


def test_salted_prove_next_layer_keyed_by_the_system(oracle):
    """The production form: no key given - the library keys the context from the operating system -, the circuit boundary,
    a hiding MMCS under HidingFriPcs (the configuration of recursion/tests/zk_hiding_mmcs.rs).  Both verifiers accept."""
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    kw = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=5, num_queries=6)
    arrs = harness_lib.generate(field, 10, seed=111, **GEN)
    ctx = p3r.Context(field=field, zk=1, mmcs_salt_elems=4, **kw)
    tp = p3r.TablePacking().with_fri_params(kw["log_final_poly_len"], kw["log_blowup"])
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(arrs), tp)
    inputs = wl.circuit_inputs_from_arrays(arrs)
    a, b = pc.prove(inputs), pc.prove(inputs)
    assert a != b
    prover = p3r.BatchStarkProver(ctx)
    prm = layer_lib.params(zk=1, mmcs_salt_elems=4, **kw)
    cpd = pc.circuit_prover_data
    for raw in (a, b):
        proof = prover.wrap_proof(raw, cpd)
        prover.verify_all_tables(proof)
        layer_lib.oracle_verify_statement(oracle, field, prm, proof.airs(), cpd.preprocessed_commitment, raw)
    # another context draws another key: another preprocessed commitment for the same circuit
    ctx2 = p3r.Context(field=field, zk=1, mmcs_salt_elems=4, **kw)
    pc2 = p3r.PreparedCircuit(ctx2, wl.circuit_from_arrays(arrs), tp)
    assert not np.array_equal(pc2.circuit_prover_data.preprocessed_commitment, cpd.preprocessed_commitment)
    pc2.free(); ctx2.close()
    pc.free(); ctx.close()
