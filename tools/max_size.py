"""The largest layers one 288 GB MI355X takes: prove_next_layer (circuit run + prove) at 2^log_h rows
with the default FRI parameters, the proof checked by the native verifier and by the oracle's verifier
(from the statement alone - the oracle PROVER would need hours at these sizes).

usage: python tools/max_size.py [field] [log_h ...] [--quintic] [--arity4]
--arity4: the prover's own arity-4 MMCS over the width-32 permutation (p3r_config.mmcs_arity = 4).
--quintic: a D = 5 circuit (base-mode Poseidon2, both Recompose kinds: six tables) under KoalaBear's quintic
challenge field.
"""
import sys
import time

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import torch

import harness_adapters as wl
import harness_lib
import layer_lib
import oracle_lib
import plonky3_recursion_amd as p3r

FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0,
           query_pow_bits=15, num_queries=54)
GEN = dict(horner_chain_len=64, sponge_chain_len=8, merkle_depth=20)
QUINTIC = "--quintic" in sys.argv
if QUINTIC:
    sys.argv.remove("--quintic")
ARITY4 = "--arity4" in sys.argv
if ARITY4:
    sys.argv.remove("--arity4")
    FRI["mmcs_arity"] = 4
D, DC = (5, 5) if QUINTIC else (4, 4)
field = sys.argv[1] if len(sys.argv) > 1 else "koala-bear"
oracle = oracle_lib.Oracle()
for log_h in [int(a) for a in sys.argv[2:]] or [23]:
    t0 = time.time()
    arrs = harness_lib.generate(field, log_h, seed=0x5EED0000, flags=harness_lib.RECOMPOSE_BOTH if QUINTIC else 0,
                                ext_degree=D, **GEN)
    print("2^%d rows %s: workload generated in %.0f s" % (log_h, field, time.time() - t0), flush=True)
    ctx = p3r.Context(field=field, ext_degree=D, challenge_degree=DC, **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    t0 = time.time()
    cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(arrs), p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=tp))
    print("  prepared in %.1f s" % (time.time() - t0), flush=True)
    pc = cache.prepared_circuit
    res = pc.upload_inputs(wl.circuit_inputs_from_arrays(arrs, D))
    del arrs
    proof = pc.prove(res)
    ctx.sync()
    t0 = time.perf_counter()
    again = pc.prove(res)
    ctx.sync()
    ms = (time.perf_counter() - t0) * 1e3
    free, total = torch.cuda.mem_get_info(0)
    assert again == proof
    wrapped = cache.prover.wrap_proof(proof, pc.circuit_prover_data)
    cache.prover.verify_all_tables(wrapped)
    layer_lib.oracle_verify_statement(oracle, field, layer_lib.params(challenge_degree=DC, **FRI),
                                      [dict(a, ext_degree=D) for a in wrapped.airs()],
                                      pc.circuit_prover_data.preprocessed_commitment, proof)
    print("  %.1f ms per prove_next_layer, %d-byte proof accepted by both verifiers, table heights %s, "
          "%.1f GB of HBM in use" % (ms, len(proof), pc.circuit_prover_data.table_heights, (total - free) / 1e9),
          flush=True)
    res.free()
    pc.free()
    ctx.close()
