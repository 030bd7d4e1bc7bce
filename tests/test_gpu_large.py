"""GPU: larger layers where the CPU oracle PROVER would take minutes - the GPU proof is checked by
the oracle VERIFIER (restatement of the in-tree circuit verifier) instead, against the GPU's own
preprocessed commitment (itself compared with the oracle's at small sizes in test_gpu_layer.py)."""
import numpy as np
import pytest

import harness_lib
import layer_lib

pytestmark = pytest.mark.gpu

FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0,
           query_pow_bits=15, num_queries=54)  # the reference examples' defaults


def prove_on_gpu(field, log_h, seed=11, **gen):
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    gen.setdefault("horner_chain_len", 64)
    gen.setdefault("sponge_chain_len", 8)
    gen.setdefault("merkle_depth", 20)
    arrs = harness_lib.generate(field, log_h, seed=seed, **gen)
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    cache = p3r.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(arrs), p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=tp))
    out = p3r.prove_next_layer(p3r.RecursionInput(traces=wl.traces_from_arrays(arrs)), ctx, p3r.FriRecursionBackend(),
                               p3r.ProveNextLayerParams(table_packing=tp), prep=cache)
    return arrs, ctx, cache, out


@pytest.mark.parametrize("field,log_h", [("koala-bear", 14), ("baby-bear", 13), ("koala-bear", 17)])
def test_default_fri_params_proof_verifies(oracle, field, log_h):
    arrs, ctx, cache, out = prove_on_gpu(field, log_h)
    L = layer_lib.OracleLayer(oracle, field, arrs, layer_lib.params(**FRI))
    L.verify(out.proof.proof, prep_cap=cache.circuit_prover_data.preprocessed_commitment)
    # a flipped byte deep inside the query section is rejected
    bad = bytearray(out.proof.proof)
    bad[(len(bad) * 2) // 3] ^= 4
    with pytest.raises(RuntimeError):
        L.verify(bytes(bad), prep_cap=cache.circuit_prover_data.preprocessed_commitment)
    cache.circuit_prover_data.free()
    ctx.close()


def test_keccak_like_mix_long_chains(oracle):
    """Config-2 style knobs: very long Horner and sponge chains."""
    arrs, ctx, cache, out = prove_on_gpu("koala-bear", 15, seed=5, horner_chain_len=2600, sponge_chain_len=330,
                                         merkle_depth=20)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, layer_lib.params(**FRI))
    L.verify(out.proof.proof, prep_cap=cache.circuit_prover_data.preprocessed_commitment)
    cache.circuit_prover_data.free()
    ctx.close()


def test_five_chained_layers_reuse_one_prep(oracle):
    """Config 3: successive proves over one NextLayerPrepCache (recursive_fibonacci.rs:413-440)."""
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    field, log_h = "koala-bear", 12
    arrs, ctx, cache, out = prove_on_gpu(field, log_h)
    L = layer_lib.OracleLayer(oracle, field, arrs, layer_lib.params(**FRI))
    cap = cache.circuit_prover_data.preprocessed_commitment
    inp = out.into_recursion_input(wl.traces_from_arrays(arrs))
    proofs = [out.proof.proof]
    for _ in range(4):
        out = p3r.prove_next_layer(inp, ctx, p3r.FriRecursionBackend(),
                                   p3r.ProveNextLayerParams(table_packing=cache.circuit_prover_data.packing), prep=cache)
        proofs.append(out.proof.proof)
        inp = out.into_recursion_input(inp.traces)
    assert all(p == proofs[0] for p in proofs)  # same traces, same shape -> identical bytes
    L.verify(proofs[-1], prep_cap=cap)
    cache.circuit_prover_data.free()
    ctx.close()


@pytest.mark.parametrize("field,log_h,gen", [
    ("koala-bear", 15, dict()),
    # config-2 style knobs: Horner chains of thousands of steps (one scan each), 330-deep sponge chains
    ("koala-bear", 15, dict(horner_chain_len=2600, sponge_chain_len=330)),
    ("baby-bear", 13, dict(horner_chain_len=300, sponge_chain_len=40)),
])
def test_circuit_run_and_prove_at_scale(oracle, field, log_h, gen):
    """prove_next_layer from the circuit: the device runner's Traces equal the oracle's sequential
    run (wide levels, long chain scans, fused narrow levels), the proof is accepted by the oracle
    verifier against the GPU's own preprocessed commitment."""
    import circuit_lib as cl
    import oracle_lib
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    gen = dict(dict(horner_chain_len=64, sponge_chain_len=8, merkle_depth=20), **gen)
    a = harness_lib.generate(field, log_h, seed=77, **gen)
    oc = cl.OracleCircuit(oracle, cl.Circuit.from_arrays(a)).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, cl.Inputs.from_arrays(a))
    want = oc.workload_arrays()
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(a), p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=tp))
    pc = cache.prepared_circuit
    inputs = pc.upload_inputs(wl.circuit_inputs_from_arrays(a))
    res = pc.run(inputs)
    for name, key in (("const_values", "const_values"), ("public_values", "public_values"), ("alu_values", "alu_values"),
                      ("recompose_values", "recompose_values"), ("p2_input_values", "p2_inputs"),
                      ("p2_mmcs_index_sum", "p2_mmcs_index_sum")):
        assert np.array_equal(res.download(name).reshape(-1), want[key]), name
    assert np.array_equal(res.download("p2_flags"), want["p2_flags"].reshape(-1, 4)[:, :3])
    res.free()
    out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=inputs), ctx, p3r.FriRecursionBackend(),
                               p3r.ProveNextLayerParams(table_packing=tp), prep=cache)
    assert pc.prove(inputs) == out.proof.proof
    L = layer_lib.OracleLayer(oracle, field, want, layer_lib.params(**FRI))
    L.verify(out.proof.proof, prep_cap=cache.circuit_prover_data.preprocessed_commitment)
    inputs.free()
    pc.free()
    ctx.close()
