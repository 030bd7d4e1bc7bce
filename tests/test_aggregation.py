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


@pytest.mark.parametrize("world,n_leaves", [(2, 8), (4, 8), (2, 2), (3, 4)])
def test_tree_over_gloo(tmp_path, world, n_leaves):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script), str(n_leaves)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
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
