"""GPU: BASELINE config 4 - `prove_aggregation_layer` (recursion/src/recursion.rs:656-762) with its
AggregationPrepCache, and the 2-to-1 tree over two ranks with REAL proofs (gloo for the transport,
both ranks on the one GPU of the test box; the driver's multi-GPU run uses nccl = RCCL)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import harness_lib
import layer_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=2, commit_pow_bits=0,
           query_pow_bits=5, num_queries=10)
GEN = dict(horner_chain_len=20, sponge_chain_len=4, merkle_depth=6)


def test_prove_aggregation_layer_cache_and_fingerprint(oracle):
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    params, backend = p3r.ProveNextLayerParams(table_packing=tp), p3r.FriRecursionBackend()
    # two leaf proofs (the children)
    la = harness_lib.generate(field, 8, seed=21, **GEN)
    leaf_cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(la), backend, params)
    leaf = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=wl.circuit_inputs_from_arrays(la)), ctx, backend,
                                params, prep=leaf_cache)
    leaf_cache.prover.verify_all_tables(leaf.proof)
    # the aggregation node: twice the counts; inputs split into the two proofs' shares and packed back
    na = harness_lib.generate(field, 9, seed=22, **GEN)
    circuit = wl.circuit_from_arrays(na)
    whole = wl.circuit_inputs_from_arrays(na)
    left, right, n_left = wl.split_aggregation_inputs(whole)
    packed = p3r.pack_aggregation_inputs(left, right, n_left)
    for f in ("public_values", "private_values", "private_data_op_ids", "private_data_siblings"):
        assert np.array_equal(np.asarray(getattr(packed, f)).reshape(-1), np.asarray(getattr(whole, f)).reshape(-1)), f
    assert len(left.private_data_op_ids) and len(right.private_data_op_ids)
    slot = [None]
    L, R = (p3r.RecursionInput(prev_proof=leaf.proof, circuit_inputs=x) for x in (left, right))
    out1 = p3r.prove_aggregation_layer(L, R, circuit, ctx, backend, params, prep_cache=slot, left_non_primitive_ops=n_left)
    assert slot[0] is not None and slot[0].circuit_fingerprint == p3r.aggregation_circuit_fingerprint(circuit)
    first_cache = slot[0]
    out2 = p3r.prove_aggregation_layer(L, R, circuit, ctx, backend, params, prep_cache=slot, left_non_primitive_ops=n_left)
    assert slot[0] is first_cache                       # cache hit: nothing rebuilt
    assert out2.proof.proof == out1.proof.proof
    first_cache.prover.verify_all_tables(out1.proof)    # native verifier
    layer_lib.oracle_verify_statement(oracle, field, layer_lib.params(**FRI), out1.proof.airs(),
                                      out1.circuit_prover_data.preprocessed_commitment, out1.proof.proof)
    # the same bytes as the plain prove_next_layer of that circuit (the node is a layer like any other)
    pc = p3r.PreparedCircuit(ctx, circuit, tp)
    assert pc.prove(whole) == out1.proof.proof
    pc.free()
    # wire round trip of the node proof, then verification from the bytes alone
    back = p3r.BatchStarkProof.from_postcard(out1.proof.to_postcard(), field)
    p3r.verify_all_tables(ctx.cfg, back)
    # a different circuit (another fingerprint) ignores and replaces the cached entry
    oa = harness_lib.generate(field, 8, seed=23, **GEN)
    c2 = wl.circuit_from_arrays(oa)
    l2, r2, n2 = wl.split_aggregation_inputs(wl.circuit_inputs_from_arrays(oa))
    assert p3r.aggregation_circuit_fingerprint(c2) != first_cache.circuit_fingerprint
    out3 = p3r.prove_aggregation_layer(p3r.RecursionInput(circuit_inputs=l2), p3r.RecursionInput(circuit_inputs=r2), c2, ctx,
                                       backend, params, prep_cache=slot, left_non_primitive_ops=n2)
    assert slot[0] is not first_cache and slot[0].circuit_fingerprint == p3r.aggregation_circuit_fingerprint(c2)
    slot[0].prover.verify_all_tables(out3.proof)
    # wrong inputs for the cached circuit: the device runner reports the conflict, as the reference's runner does
    with pytest.raises(p3r.P3rError):
        p3r.prove_aggregation_layer(L, R, c2, ctx, backend, params, prep_cache=slot, left_non_primitive_ops=n_left)
    slot[0].prepared_circuit.free()
    leaf_cache.prepared_circuit.free()
    ctx.close()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,leaves", [(2, 4), (1, 8)])
def test_tree_of_real_proofs_root_verifies(world, leaves):
    """bench.py --tree: leaves -> root with prove_next_layer / prove_aggregation_layer on every node,
    children parsed and natively verified before their parent is proved, the root verified on rank 0."""
    env = dict(os.environ, P3R_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["bench.py", "--tree", "--gpus", str(world), "--tree-leaves", str(leaves), "--leaf-log-height", "10", "--steps", "1",
            "--warmup", "0", "--tree-verify-children", "--tree-level-barriers"]
    if world == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(free_port())] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["root_verified"] is True and line["n_gpus"] == world and line["scaling"] == "strong"
    assert line["config"]["nodes"] == 2 * leaves - 1
    assert len(line["rank0"]["level_wall_ms_last_step"]) == leaves.bit_length()
    assert line["rank0"]["child_verify_ms"] is not None


def test_default_bench_line_two_ranks_over_gloo():
    """The driver's multi-GPU launch of bench.py (independent layers, one per rank, weak scaling, proofs
    handed to rank 0) with two ranks sharing the test box's one GPU: gloo for the transport instead of
    nccl, everything else as the driver runs it.  Every rank's timed proof is verified."""
    env = dict(os.environ, P3R_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-height", "12",
           "--no-cpu-baseline", "--no-config2", "--no-small-layers"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["proof_verified"] is True
    assert line["config"]["independent_proofs"] == 2
    assert isinstance(line["root_handoff_ms"], float)
