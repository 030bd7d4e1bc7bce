"""CPU: aggregation-tree placement and proof hand-off over torch.distributed (gloo, world_size 2
and 4) with a stand-in prover - the N>1 path of SURVEY.md section 8e."""
import hashlib
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import hashlib, os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from plonky3_recursion_amd.aggregation import TreePlan, run_aggregation_tree, gather_proofs_to_root
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n_leaves = int(sys.argv[1])
plan = TreePlan(n_leaves, world)
leaf = lambda i: (b"leaf%%d|" %% i) * (1000 + i)                       # distinct, variable-length payloads
parent = lambda lvl, node, l, r: hashlib.sha256(l).digest() + hashlib.sha256(r).digest() + b"|%%d.%%d" %% (lvl, node)
root = run_aggregation_tree(plan, rank, leaf, parent, dist=dist)
if rank == 0:
    print("ROOT", root.hex())
else:
    assert root is None
allp = gather_proofs_to_root(b"proof-of-rank-%%d" %% rank * (rank + 1), dist, rank, world)
if rank == 0:
    print("GATHER", [p.decode() for p in allp] == ["proof-of-rank-%%d" %% r * (r + 1) for r in range(world)])
dist.barrier()
dist.destroy_process_group()
'''


FOREST_WORKER = r'''
import hashlib, os, sys, time
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from plonky3_recursion_amd.aggregation import TreePlan, run_aggregation_forest
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n_leaves, n_trees, workers = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
plans = [TreePlan(n_leaves, world, offset=t) for t in range(n_trees)]
def leaf(t, i):
    time.sleep(0.001 * ((7 * i + 3 * t + rank) %% 5))                  # completion order differs from submission order
    return (b"tree%%d-leaf%%d|" %% (t, i)) * (1000 + i)
def parent(t, lvl, node, l, r):
    return hashlib.sha256(l).digest() + hashlib.sha256(r).digest() + b"|%%d.%%d.%%d" %% (t, lvl, node)
seen = []
roots = run_aggregation_forest(plans, rank, leaf, parent, dist=dist, workers=workers,
                               on_node=lambda t, lvl, node, s: seen.append((t, lvl, node)))
mine = sum(len(p.my_nodes(l, rank)) for p in plans for l in range(p.levels))
assert len(seen) == mine, (len(seen), mine)
if rank == 0:
    for t, r in enumerate(roots):
        print("ROOT", t, r.hex())
else:
    assert roots is None
dist.barrier()
dist.destroy_process_group()
'''


def expected_forest_root(t, n_leaves):
    level = [(b"tree%d-leaf%d|" % (t, i)) * (1000 + i) for i in range(n_leaves)]
    lvl = 1
    while len(level) > 1:
        level = [hashlib.sha256(level[2 * j]).digest() + hashlib.sha256(level[2 * j + 1]).digest() + b"|%d.%d.%d" % (t, lvl, j)
                 for j in range(len(level) // 2)]
        lvl += 1
    return level[0].hex()


@pytest.mark.parametrize("world,n_leaves,n_trees,workers", [(2, 8, 1, 1), (4, 8, 4, 2), (2, 4, 3, 3), (3, 4, 2, 1), (8, 8, 8, 1)])
def test_forest_over_gloo(tmp_path, world, n_leaves, n_trees, workers):
    """Dependency-driven scheduler: K trees in flight with rotated placement, no level barrier, one comm
    thread per rank; every tree's root reaches rank 0 and equals the serial evaluation."""
    script = tmp_path / "forest_worker.py"
    script.write_text(FOREST_WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script), str(n_leaves), str(n_trees),
           str(workers)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    for t in range(n_trees):
        assert f"ROOT {t} {expected_forest_root(t, n_leaves)}" in out.stdout


def test_forest_single_process_and_errors():
    from plonky3_recursion_amd.aggregation import TreePlan, run_aggregation_forest
    leaf = lambda t, i: (b"tree%d-leaf%d|" % (t, i)) * (1000 + i)
    parent = lambda t, lvl, node, l, r: hashlib.sha256(l).digest() + hashlib.sha256(r).digest() + b"|%d.%d.%d" % (t, lvl, node)
    roots = run_aggregation_forest([TreePlan(8, 1, offset=t) for t in range(3)], 0, leaf, parent, workers=4)
    assert [r.hex() for r in roots] == [expected_forest_root(t, 8) for t in range(3)]

    def bad_parent(t, lvl, node, l, r):
        if (lvl, node) == (2, 1):
            raise RuntimeError("node failed")
        return parent(t, lvl, node, l, r)
    with pytest.raises(RuntimeError, match="node failed"):     # the failure reaches the caller, the pool is shut down
        run_aggregation_forest([TreePlan(8, 1)], 0, leaf, bad_parent, workers=2)


def test_rotated_placement_balances_the_forest():
    from plonky3_recursion_amd.aggregation import TreePlan
    world = 8
    plans = [TreePlan(8, world, offset=t) for t in range(world)]
    load = [sum(len(p.my_nodes(l, r)) for p in plans for l in range(p.levels)) for r in range(world)]
    assert load == [15] * world                                  # 8 trees x 15 proofs over 8 ranks
    assert sorted(p.owner(p.levels - 1, 0) for p in plans) == list(range(world))


def expected_root(n_leaves):
    level = [(b"leaf%d|" % i) * (1000 + i) for i in range(n_leaves)]
    lvl = 1
    while len(level) > 1:
        level = [hashlib.sha256(level[2 * j]).digest() + hashlib.sha256(level[2 * j + 1]).digest() + b"|%d.%d" % (lvl, j)
                 for j in range(len(level) // 2)]
        lvl += 1
    return level[0].hex()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# world 8: BASELINE config 4's placement (one leaf per rank, root on rank 0); world 1: a process group of one - the
# hand-off still goes through the collective library (bench.py --force-dist)
@pytest.mark.parametrize("world,n_leaves", [(2, 8), (4, 8), (2, 2), (3, 4), (8, 8), (1, 4)])
def test_tree_over_gloo(tmp_path, world, n_leaves):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script), str(n_leaves)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert f"ROOT {expected_root(n_leaves)}" in out.stdout
    assert "GATHER True" in out.stdout


def test_plan_placement():
    from plonky3_recursion_amd.aggregation import TreePlan
    plan = TreePlan(8, 8)
    assert plan.levels == 4
    assert [plan.owner(0, i) for i in range(8)] == list(range(8))       # one leaf per GPU
    assert [plan.owner(1, i) for i in range(4)] == [0, 2, 4, 6]          # parent sits on its left child
    assert [plan.owner(2, i) for i in range(2)] == [0, 4]
    assert plan.owner(3, 0) == 0                                         # root on GPU 0
    with pytest.raises(ValueError):
        TreePlan(6, 2)


def test_single_process_tree():
    from plonky3_recursion_amd.aggregation import TreePlan, run_aggregation_tree
    leaf = lambda i: (b"leaf%d|" % i) * (1000 + i)
    parent = lambda lvl, node, l, r: hashlib.sha256(l).digest() + hashlib.sha256(r).digest() + b"|%d.%d" % (lvl, node)
    assert run_aggregation_tree(TreePlan(8, 1), 0, leaf, parent).hex() == expected_root(8)


def test_wall_time_prediction_model():
    """aggregation.predict_forest_wall_ms: the scheduler's own event model on solo times - what `bench.py --tree` prints as the
    number a multi-GPU run is to be compared with."""
    from plonky3_recursion_amd.aggregation import TreePlan, predict_forest_wall_ms as predict
    leaf, node, msg = 3.0, 5.0, 1.0
    # one rank, one prover: the serial loop of recursive_aggregation.rs:447-475
    one = predict([TreePlan(8, 1)], leaf, node)
    assert one["wall_ms"] == pytest.approx(8 * leaf + 7 * node) and one["nodes"] == 15
    assert one["critical_path_ms"] == pytest.approx(leaf + 3 * node)
    # a GPU per leaf: the critical path - a leaf, then one node and one message per level
    eight = predict([TreePlan(8, 8)], leaf, node, msg)
    assert eight["wall_ms"] == pytest.approx(leaf + 3 * (node + msg)) == pytest.approx(eight["critical_path_ms"])
    # fewer GPUs than leaves/2: between the two, never better than the critical path
    walls = [predict([TreePlan(8, w)], leaf, node, msg)["wall_ms"] for w in (1, 2, 4, 8)]
    assert walls == sorted(walls, reverse=True) and walls[-1] >= eight["critical_path_ms"] - 1e-9
    # concurrent provers on one GPU: bounded by its capacity (2 proofs at once at full speed: half the serial time at best)
    shared = predict([TreePlan(8, 1)], leaf, node, workers=4, gpu_capacity=2.0)
    assert (8 * leaf + 7 * node) / 2 - 1e-9 <= shared["wall_ms"] < 8 * leaf + 7 * node
    # two ranks on ONE GPU share it; on two GPUs they do not
    same = predict([TreePlan(8, 2)], leaf, node, msg, gpu_of_rank={0: "g", 1: "g"}, gpu_capacity=1.0)
    apart = predict([TreePlan(8, 2)], leaf, node, msg)
    assert same["wall_ms"] > apart["wall_ms"]
    # K trees with rotated placement keep every rank busy: per-tree cost falls
    forest = predict([TreePlan(8, 4, offset=t) for t in range(4)], leaf, node, msg)
    assert forest["wall_ms"] / 4 < predict([TreePlan(8, 4)], leaf, node, msg)["wall_ms"]


def test_weak_scaling_prediction_of_the_plain_entry():
    """bench.py's `scaling_prediction`: independent proofs, so the step of N ranks is the slowest rank's solo step - equal
    to the solo step at N = 1, growing by less than 3 % to N = 8, and the aggregate rate is N proofs per step."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    p = bench.weak_scaling_prediction(28.0)["one_gpu_per_rank"]
    assert p["1"]["predicted_ms_per_step"] == 28.0
    steps = [p[str(n)]["predicted_ms_per_step"] for n in (1, 2, 4, 8)]
    assert steps == sorted(steps) and steps[-1] < 28.0 * 1.03
    for n in (1, 2, 4, 8):
        assert abs(p[str(n)]["predicted_proofs_per_s"] - n / p[str(n)]["predicted_ms_per_step"] * 1e3) < 1e-9
