"""GPU: the sizes the bench line and BASELINE.json are quoted on - KoalaBear 2^20 rows (the metric)
and BabyBear 2^22 rows (config 5) - with the default FRI parameters.  They take the branches the
smaller tests do not (balanced four-step NTT splits, 4 M / 16 M-leaf trees, > 2^31-cell matrices,
multi-GB pools): the proof of prove_next_layer is checked by BOTH verifiers (native
`verify_all_tables`, and the oracle's restatement of the in-tree circuit verifier), tampering is
rejected, and the LDE + MMCS commit of a tall narrow matrix is compared bit for bit with the oracle."""
import numpy as np
import pytest

import harness_lib
import layer_lib
import oracle_lib

pytestmark = pytest.mark.gpu

FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0,
           query_pow_bits=15, num_queries=54)  # the reference examples' defaults (= bench.py)
GEN = dict(horner_chain_len=64, sponge_chain_len=8, merkle_depth=20)


@pytest.mark.parametrize("field,log_h", [("koala-bear", 20), ("baby-bear", 22)])
def test_headline_layer_proves_and_both_verifiers_accept(oracle, field, log_h):
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    arrs = harness_lib.generate(field, log_h, seed=0x5EED0000, **GEN)   # the bench workload
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    circuit = wl.circuit_from_arrays(arrs)
    cache = p3r.build_next_layer_prep(ctx, circuit, p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=tp))
    assert cache.prepared_circuit.prepared_on_device
    inputs = wl.circuit_inputs_from_arrays(arrs)
    del arrs
    out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=inputs), ctx, p3r.FriRecursionBackend(),
                               p3r.ProveNextLayerParams(table_packing=tp), prep=cache)
    cpd = cache.circuit_prover_data
    assert max(cpd.table_heights) == 1 << log_h
    # a second prove of the same inputs gives the same bytes (no stale pooled memory at this size)
    assert cache.prepared_circuit.prove(inputs) == out.proof.proof
    # the host restatement of the preparation (csrc/circuit_impl.hip.h) agrees with the device pass at this size:
    # same preprocessed commitment, same schedule depth, same proof
    import os
    os.environ["P3R_PREP_HOST"] = "1"
    try:
        host_pc = p3r.PreparedCircuit(ctx, circuit, tp)
    finally:
        os.environ.pop("P3R_PREP_HOST", None)
    assert not host_pc.prepared_on_device
    assert np.array_equal(host_pc.circuit_prover_data.preprocessed_commitment, cpd.preprocessed_commitment)
    assert host_pc.levels == cache.prepared_circuit.levels
    assert host_pc.prove(inputs) == out.proof.proof
    host_pc.free()
    del circuit
    # 1. native verifier (host code of the C-ABI library), through the reference's entry point
    cache.prover.verify_all_tables(out.proof)
    # 2. oracle verifier, from the statement the proof metadata rebuilds
    prm = layer_lib.params(**FRI)
    layer_lib.oracle_verify_statement(oracle, field, prm, out.proof.airs(), cpd.preprocessed_commitment,
                                      out.proof.proof)
    # both reject a flipped bit deep inside the query section, and a wrong preprocessed commitment
    bad = bytearray(out.proof.proof)
    bad[(len(bad) * 2) // 3] ^= 4
    import dataclasses
    with pytest.raises(p3r.P3rError):
        cache.prover.verify_all_tables(dataclasses.replace(out.proof, proof=bytes(bad)))
    with pytest.raises(RuntimeError):
        layer_lib.oracle_verify_statement(oracle, field, prm, out.proof.airs(), cpd.preprocessed_commitment, bytes(bad))
    wrong = np.array(cpd.preprocessed_commitment, dtype=np.uint32).copy()
    wrong.reshape(-1)[3] ^= 1
    with pytest.raises(RuntimeError):
        layer_lib.oracle_verify_statement(oracle, field, prm, out.proof.airs(), wrong, out.proof.proof)
    cache.prepared_circuit.free()
    ctx.close()


def test_headline_zk_layer_proves_and_both_verifiers_accept(oracle):
    """The headline layer (KoalaBear, 2^20 rows, the bench workload) under the ZK configuration - HidingFriPcs, two random
    codewords, every commitment over 2^21-row extended domains (2^23-row LDEs), eight masked quotient chunks - keyed the way a
    deployment keys it: no key given, the library draws one from the operating system (so no byte can be compared with the
    CPU oracle's prover; tests/test_gpu_zk.py does that under P3R_EXT_ZK_DETERMINISTIC at sizes the oracle finishes).
    Both verifiers accept, two proofs of one input differ, tampering is rejected, a non-ZK verifier refuses the proof."""
    import dataclasses
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    field, log_h = "koala-bear", 20
    arrs = harness_lib.generate(field, log_h, seed=0x5EED0000, **GEN)
    ctx = p3r.Context(field=field, zk=1, num_random_codewords=2, **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(arrs), p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=tp))
    pc = cache.prepared_circuit
    inputs = pc.upload_inputs(wl.circuit_inputs_from_arrays(arrs))
    del arrs
    first, second = pc.prove(inputs), pc.prove(inputs)
    assert first != second and ctx.zk_nonce == 2      # (lengths differ too: field elements are varint-encoded)
    with pytest.raises(p3r.P3rError, match="DETERMINISTIC"):
        ctx.zk_nonce = 0          # replaying a proof counter repeats the masks: refused outside the deterministic mode
    cpd = pc.circuit_prover_data
    prover = p3r.BatchStarkProver(ctx)
    prm = layer_lib.params(zk=1, num_random_codewords=2, **FRI)
    for raw in (first, second):
        proof = prover.wrap_proof(raw, cpd)
        prover.verify_all_tables(proof)                                                                   # native verifier
        layer_lib.oracle_verify_statement(oracle, field, prm, proof.airs(), cpd.preprocessed_commitment, raw)   # the oracle's
    proof = prover.wrap_proof(first, cpd)
    assert max(proof.degree_bits) == log_h + 1            # the EXTENDED degree bits (recursion.rs:374)
    for at in (len(first) // 5, (len(first) * 2) // 3):
        bad = bytearray(first)
        bad[at] ^= 4
        with pytest.raises(p3r.P3rError):
            prover.verify_all_tables(dataclasses.replace(proof, proof=bytes(bad)))
        with pytest.raises(RuntimeError):
            layer_lib.oracle_verify_statement(oracle, field, prm, proof.airs(), cpd.preprocessed_commitment, bytes(bad))
    # the PCS types refuse each other: a verifier of the non-hiding configuration cannot read this proof
    with pytest.raises(RuntimeError):
        layer_lib.oracle_verify_statement(oracle, field, layer_lib.params(**FRI), proof.airs(), cpd.preprocessed_commitment, first)
    inputs.free()
    pc.free()
    ctx.close()


@pytest.mark.parametrize("field,log_h,width", [("koala-bear", 20, 8), ("baby-bear", 22, 2)])
def test_headline_lde_and_commit_bit_exact(oracle, field, log_h, width):
    """coset_lde_batch (blow-up 4) + MerkleTreeMmcs::commit of a 2^log_h x width matrix: every LDE cell,
    the commitment, and opened rows with their Merkle paths against the CPU oracle (OpenMP)."""
    import plonky3_recursion_amd as p3r
    p = oracle_lib.MODULUS[field]
    g = oracle_lib.GENERATOR[field]
    rng = np.random.default_rng(log_h)
    m = rng.integers(0, p, size=(1 << log_h, width), dtype=np.uint32)
    m[0, :] = p - 1
    m[-1, :] = 0
    ctx = p3r.Context(field=field, **FRI)
    dm = ctx.upload(m)
    lde = ctx.coset_lde_batch_device(dm, 2, g)
    cap, tree = ctx.commit_device([lde])
    got = lde.download()
    want = oracle.coset_lde(field, m, 2, g)
    assert np.array_equal(got, want)
    del got
    ocap, otree = oracle.commit(field, [want], 0)
    assert np.array_equal(cap, ocap)
    n = want.shape[0]
    for index in (0, 1, n // 2 - 1, n // 2, n - 1, 0x2AAAAA % n, 0x155555 % n, 123457 % n):
        opened, proof = tree.open_batch(index)
        o_opened, o_proof = otree.open(index)
        assert np.array_equal(opened, o_opened)
        assert np.array_equal(proof, o_proof)
        assert oracle.verify(field, cap, [want.shape], index, opened, proof)
    tree.free()
    ctx.close()
