"""GPU: BASELINE config 4 - `prove_aggregation_layer` (recursion/src/recursion.rs:656-762) with its
AggregationPrepCache, and the 2-to-1 tree over two ranks with REAL proofs (gloo for the transport,
both ranks on the one GPU of the test box; the driver's multi-GPU run uses nccl = RCCL)."""
import json
import os
import socket
import subprocess
import tempfile
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


def test_aggregation_output_outlives_its_cache_slot_and_content_is_part_of_the_key():
    """The CircuitProverData of a RecursionOutput stays valid without a cache and after its slot is replaced
    (the reference returns an Rc, recursion.rs:748-761), and a circuit with the same four lengths but other
    content is NOT served from the slot."""
    import gc
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    params, backend = p3r.ProveNextLayerParams(table_packing=tp), p3r.FriRecursionBackend()
    na = harness_lib.generate(field, 8, seed=31, **GEN)
    circuit = wl.circuit_from_arrays(na)
    left, right, n_left = wl.split_aggregation_inputs(wl.circuit_inputs_from_arrays(na))
    L, R = (p3r.RecursionInput(circuit_inputs=x) for x in (left, right))
    out = p3r.prove_aggregation_layer(L, R, circuit, ctx, backend, params, prep_cache=None, left_non_primitive_ops=n_left)
    gc.collect()
    assert out.circuit_prover_data.h, "circuit_prover_data died with the local prepared circuit"
    assert out.circuit_prover_data.table_heights[2] > 0
    p3r.BatchStarkProver(ctx, tp).verify_all_tables(out.proof)
    slot = [None]
    out1 = p3r.prove_aggregation_layer(L, R, circuit, ctx, backend, params, prep_cache=slot, left_non_primitive_ops=n_left)
    first = slot[0]
    # same lengths, different content: flip the value of one constant
    import copy
    c2 = copy.deepcopy(circuit)
    ops = np.asarray(c2.ops).reshape(-1, 8)
    k = int(np.nonzero(ops[:, 0] == p3r.prover.OP_CONST)[0][3])
    c2.ext = np.array(c2.ext, copy=True)
    c2.ext[ops[k, 6]] ^= 1
    assert p3r.aggregation_circuit_fingerprint(c2) != first.circuit_fingerprint
    fp1, fp2 = first.circuit_fingerprint, p3r.aggregation_circuit_fingerprint(c2)
    assert (fp1.witness_count, fp1.public_flat_len, fp1.private_flat_len, fp1.ops_len) == \
           (fp2.witness_count, fp2.public_flat_len, fp2.private_flat_len, fp2.ops_len)
    try:
        p3r.prove_aggregation_layer(L, R, c2, ctx, backend, params, prep_cache=slot, left_non_primitive_ops=n_left)
    except p3r.P3rError:
        pass                                    # the changed constant may conflict with the inputs: a run error, not a stale hit
    assert slot[0] is first or slot[0].circuit_fingerprint == fp2
    gc.collect()
    assert out1.circuit_prover_data.h            # the replaced / kept entry is still usable from the older output
    ctx.close()


DETAIL = os.path.join(tempfile.gettempdir(), f"p3r_bench_detail_{os.getpid()}.json")


def bench_result(out):
    """The full result of a bench.py run: the LAST stdout line is the driver's contract line (compact, every contract
    key, names its detail file); everything else is in the detail file rank 0 wrote."""
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("{"), "the contract line must be the last line of stdout, got: " + repr(out.stdout[-600:])
    line = json.loads(last)
    assert len(last) < 6000 and {"metric", "value", "unit", "n_gpus", "roofline", "cpu_baseline", "proof_verified",
                                 "proof_sha256", "config"} <= set(line), last[:300]
    with open(DETAIL) as fh:
        full = json.load(fh)
    os.remove(DETAIL)
    assert full["value"] == pytest.approx(line["value"], rel=1e-5) and full["n_gpus"] == line["n_gpus"]
    return full



def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,leaves,zk", [(2, 4, False), (1, 8, False), (2, 4, True)])
def test_tree_of_real_proofs_root_verifies(world, leaves, zk):
    """bench.py --tree: leaves -> root with prove_next_layer / prove_aggregation_layer on every node,
    children parsed and natively verified before their parent is proved, the root verified on rank 0.
    --zk: every proof under the hiding PCS (`recursive_aggregation --zk`): children cross ranks as ZK proofs."""
    env = dict(os.environ, P3R_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["bench.py", "--tree", "--gpus", str(world), "--tree-leaves", str(leaves), "--leaf-log-height", "10", "--steps", "1",
            "--warmup", "0", "--tree-verify-children", "--tree-level-barriers"] + (["--zk"] if zk else [])
    if world == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(free_port())] + args
    out = subprocess.run(cmd + ["--detail-out", DETAIL], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = bench_result(out)
    assert line["root_verified"] is True and line["n_gpus"] == world and line["scaling"] == "strong"
    assert line["config"]["nodes"] == 2 * leaves - 1
    assert len(line["rank0"]["level_wall_ms_last_step"]) == leaves.bit_length()
    assert line["rank0"]["child_verify_ms"] is not None
    assert line["config"]["zk"] is zk
    # the comparison a multi-GPU run gets: critical path and predicted walls from solo measurements
    pr = line["prediction"]
    assert line["critical_path_ms"] > 0 and set(pr["one_gpu_per_rank"]) == {"1", "2", "4", "8"}
    assert pr["one_gpu_per_rank"]["8"]["predicted_wall_ms"] <= pr["one_gpu_per_rank"]["1"]["predicted_wall_ms"]


def test_plain_bench_entry_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns the two ranks (before it touches a
    device), the line reports what the collective layer saw; a launcher whose WORLD_SIZE disagrees is an error."""
    env = dict(os.environ, P3R_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-height", "12",
           "--no-cpu-baseline", "--no-config2", "--no-small-layers"]
    out = subprocess.run(cmd + ["--detail-out", DETAIL], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = bench_result(out)
    assert line["n_gpus"] == 2 and line["proof_verified"] is True
    assert line["ranks"]["world_size"] == 2 and line["ranks"]["backend"] == "gloo" and len(line["ranks"]["devices"]) == 2
    # the forest form through the same plain entry: one tree per rank, rotated placement, every root verified
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--tree", "--trees", "0", "--tree-leaves", "4", "--leaf-log-height", "10",
           "--steps", "1", "--warmup", "0", "--tree-workers", "2"]
    out = subprocess.run(cmd + ["--detail-out", DETAIL], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = bench_result(out)
    assert line["n_gpus"] == 2 and line["config"]["trees"] == 2 and line["roots_verified"] == 2 and line["scaling"] == "weak"
    assert line["config"]["scheduler"] == "dependency-driven" and line["ranks"]["world_size"] == 2
    assert line["rank0"]["child_parse_ms"] is not None
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], capture_output=True, text=True, timeout=120, cwd=ROOT,
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode == 2 and "WORLD_SIZE" in bad.stderr
    # ranks[]: what the collective layer reports, the device and PCI address of every rank, its own time
    rk = line["ranks"]
    assert rk["backend"] == "gloo" and len(rk["per_rank"]) == 2 and rk["distinct_gpus"] == 1
    assert all(p["pci"] and p["ms"] is not None and p["ms"] > 0 for p in rk["per_rank"])
    assert rk["omp_num_threads"] and int(rk["omp_num_threads"]) <= max(1, (os.cpu_count() or 2) // 2)
    # one rank per GPU over RCCL needs as many GPUs as ranks: refused with a message, not a hang inside ncclCommInitRank
    import torch
    if torch.cuda.device_count() < 2:
        nccl_env = {k: v for k, v in env.items() if k != "P3R_BENCH_BACKEND"}
        few = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--log-height", "10"],
                             capture_output=True, text=True, timeout=300, cwd=ROOT, env=nccl_env)
        assert few.returncode != 0 and "visible GPUs" in few.stderr, few.stderr[-2000:]


def test_default_bench_line_two_ranks_over_gloo():
    """The driver's multi-GPU launch of bench.py (independent layers, one per rank, weak scaling, proofs
    handed to rank 0) with two ranks sharing the test box's one GPU: gloo for the transport instead of
    nccl, everything else as the driver runs it.  Every rank's timed proof is verified."""
    env = dict(os.environ, P3R_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-height", "12",
           "--no-cpu-baseline", "--no-config2", "--no-small-layers"]
    out = subprocess.run(cmd + ["--detail-out", DETAIL], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = bench_result(out)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["proof_verified"] is True
    assert line["config"]["independent_proofs"] == 2
    assert isinstance(line["root_handoff_ms"], float)


@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
def test_rccl_process_group_of_one(launcher):
    """First contact with RCCL on an MI355X before the driver's scaling run: world size 1 under the **nccl** backend -
    init_process_group("nccl", device_id=...), the barrier and the all_reduce(MAX) that bracket the timed region, the
    proof's hand-off as device tensors through the collective library, destroy_process_group - with the environment
    (HSA_ENABLE_IPC_MODE_LEGACY=0) and the launcher line the driver uses, and once through `--force-dist` alone.
    Nothing about scaling is claimed."""
    env = {k: v for k, v in os.environ.items() if k not in ("P3R_BENCH_BACKEND", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--log-height", "12", "--no-cpu-baseline",
            "--no-small-layers", "--force-dist"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port())] + args
    else:
        env.pop("MASTER_ADDR")
        cmd = [sys.executable] + args
    out = subprocess.run(cmd + ["--detail-out", DETAIL], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = bench_result(out)
    assert line["n_gpus"] == 1 and line["proof_verified"] is True
    rk = line["ranks"]
    assert rk["backend"] == "nccl" and rk["world_size"] == 1 and rk["distinct_gpus"] == 1
    assert isinstance(line["root_handoff_ms"], float), line["root_handoff_ms"]    # "failed: ..." would be a string
