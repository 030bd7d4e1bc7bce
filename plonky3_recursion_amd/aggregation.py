"""2-to-1 aggregation tree across the GPUs of one node.

Reference: `prove_aggregation_layer` (recursion/src/recursion.rs:656-762) proves one tree node;
the tree driver is the serial `for pair_idx` loop of recursion/examples/recursive_aggregation.rs:423-476,
which the book calls embarrassingly parallel (book/src/user_guide/aggregation.md).  Nodes of one
level are independent, so they shard over ranks with NO data-path collective; the only exchange is
moving finished child proofs (hundreds of kB) to the rank that proves the parent, and the final
root hand-off to rank 0.  One process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The node prover itself is `prove_aggregation_layer` (prover.py: AggregationPrepCache keyed by the
circuit fingerprint, the verifier circuit run on the device); `bench.py --tree` drives this
scheduler with real proofs, `tests/test_gpu_aggregation.py` checks the root.  What is NOT here:
BUILDING the parent's verifier circuit from the two child proofs - the reference's symbolic CPU
front end (out of scope): callers pass the circuit and its inputs.
"""
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np


@dataclass
class TreePlan:
    """Static placement: node i of level l runs on rank (i * 2^l + offset) % world, so a parent lives where
    its LEFT child lived and only the right child's proof moves.  `offset` rotates the placement: with K
    trees in flight tree t takes offset t, so the rank that proves the upper levels (and receives the
    root) differs from tree to tree and every rank ends up with the same number of proofs."""
    n_leaves: int
    world: int
    offset: int = 0

    def __post_init__(self):
        if self.n_leaves < 1 or self.n_leaves & (self.n_leaves - 1):
            raise ValueError("n_leaves must be a power of two")

    @property
    def levels(self):
        return self.n_leaves.bit_length()  # leaves are level 0, root is level log2(n)

    def nodes(self, level):
        return self.n_leaves >> level

    def owner(self, level, node):
        return ((node << level) + self.offset) % self.world

    def my_nodes(self, level, rank):
        return [i for i in range(self.nodes(level)) if self.owner(level, i) == rank]


def _send_bytes(dist, payload: bytes, dst: int, device):
    import torch
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    dist.send(n, dst)
    buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    dist.send(buf, dst)


def _recv_bytes(dist, src: int, device) -> bytes:
    import torch
    n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(n, src)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.recv(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def run_aggregation_tree(plan: TreePlan, rank: int, prove_leaf: Callable[[int], bytes],
                         prove_parent: Callable[[int, int, bytes, bytes], bytes], dist=None,
                         device="cpu", on_level: Optional[Callable[[int, float], None]] = None,
                         level_barrier: Optional[Callable[[], None]] = None, workers: int = 1) -> Optional[bytes]:
    """Proves every node this rank owns, level by level; returns the root proof on rank 0.

    prove_leaf(i) -> proof bytes of leaf i.
    prove_parent(level, node, left_proof, right_proof) -> proof bytes (the caller runs the verifier
    circuit over the two children and calls `prove_aggregation_layer`'s GPU part).
    `dist` is torch.distributed (None for a single process).  `on_level(level, seconds)` receives this
    rank's wall time per level; with `level_barrier` the levels are separated by that barrier, so
    the times are the level's wall time across ranks.  `workers` > 1: the nodes this rank owns at one
    level are proved concurrently by that many host threads (the callbacks must then be thread-safe,
    e.g. one p3r_ctx = one HIP stream per thread: layers of 2^14..2^16 rows do not fill an MI355X on
    their own, tools/concurrent_small.py)."""
    import time
    pool = None
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=workers)

    def run_all(fn, keys):
        if pool is None or len(keys) < 2:
            return {k: fn(k) for k in keys}
        return dict(zip(keys, pool.map(fn, keys)))

    try:
        return _run_levels(plan, rank, prove_leaf, prove_parent, dist, device, on_level, level_barrier, run_all)
    finally:
        if pool is not None:
            pool.shutdown()


def _run_levels(plan, rank, prove_leaf, prove_parent, dist, device, on_level, level_barrier, run_all):
    import time
    t0 = time.perf_counter()
    proofs = run_all(prove_leaf, plan.my_nodes(0, rank))
    if level_barrier:
        level_barrier()
    if on_level:
        on_level(0, time.perf_counter() - t0)
    for level in range(1, plan.levels):
        t0 = time.perf_counter()
        nxt = {}
        # ship right children to the parent's owner (left child already lives there)
        for node in range(plan.nodes(level)):
            parent_rank = plan.owner(level, node)
            right = 2 * node + 1
            right_rank = plan.owner(level - 1, right)
            if right_rank == parent_rank:
                continue
            if rank == right_rank:
                _send_bytes(dist, proofs[right], parent_rank, device)
            elif rank == parent_rank:
                proofs[right] = _recv_bytes(dist, right_rank, device)
        prev = proofs
        nxt = run_all(lambda node: prove_parent(level, node, prev[2 * node], prev[2 * node + 1]), plan.my_nodes(level, rank))
        proofs = nxt
        if level_barrier:
            level_barrier()
        if on_level:
            on_level(level, time.perf_counter() - t0)
    # final root hand-off to rank 0
    root_level = plan.levels - 1
    root_rank = plan.owner(root_level, 0)
    if root_rank == 0:
        return proofs.get(0) if rank == 0 else None
    if rank == root_rank:
        _send_bytes(dist, proofs[0], 0, device)
        return None
    if rank == 0:
        return _recv_bytes(dist, root_rank, device)
    return None


def run_aggregation_forest(plans, rank: int, prove_leaf: Callable[[int, int], bytes],
                           prove_parent: Callable[[int, int, int, bytes, bytes], bytes], dist=None, device="cpu",
                           workers: int = 1, on_node: Optional[Callable[[int, int, int, float], None]] = None,
                           encode: Callable = lambda proof: proof, decode: Callable = lambda data: data) -> Optional[List[bytes]]:
    """Dependency-driven scheduler for one or more 2-to-1 trees in flight (`plans`: one TreePlan per tree, all
    over the same world).  A parent is proved as soon as ITS two children are on its rank - there is no level
    barrier - by a pool of `workers` host threads (callbacks must be thread-safe for workers > 1: one p3r_ctx
    = one HIP stream per thread).  The reference's driver is the serial pair loop of
    recursive_aggregation.rs:447-475; the nodes it proves one after the other are independent.

    prove_leaf(tree, i) -> proof; prove_parent(tree, level, node, left, right) -> proof.  A proof is whatever the
    callbacks exchange (bytes by default).  `encode(proof) -> bytes` / `decode(bytes) -> proof` are applied only to
    proofs that change rank, by the communication thread: a child proved on this rank reaches its parent as the
    object its prover returned (the reference's driver never serialises a child either), a child that arrives over
    the wire is parsed on receipt, while the parent is still waiting for its other child or for a prover.
    Child proofs that must change rank, and each tree's root on its way to rank 0, are moved by ONE
    communication thread per rank that walks the (level, tree, node)-sorted list of this rank's messages:
    both ends of a message reach it after the same earlier messages, so blocking send/recv pairs match
    without tags (RCCL has none) and cannot deadlock.  Returns the root proofs (tree order) on rank 0.
    `on_node(tree, level, node, seconds_since_start)` is called when a node this rank proved completes."""
    import threading
    import time
    from concurrent.futures import Future, ThreadPoolExecutor
    n_trees = len(plans)
    L = plans[0].levels
    fut = {}                       # (tree, level, node) -> Future[bytes], for every proof this rank holds at some point
    pending = {}                   # parent key -> number of children still missing
    lock = threading.Lock()
    errors = []
    t_start = time.perf_counter()
    pool = ThreadPoolExecutor(max_workers=max(1, workers))

    def future_of(key):
        with lock:
            f = fut.get(key)
            if f is None:
                f = fut[key] = Future()
            return f

    def run_node(key, fn, *args):
        f = future_of(key)
        try:
            out = fn(*args)
            if on_node:
                on_node(key[0], key[1], key[2], time.perf_counter() - t_start)
            f.set_result(out)
        except BaseException as e:  # noqa: BLE001 - handed to the waiting thread
            errors.append(e)
            f.set_exception(e)

    def child_done(parent_key, _f):
        with lock:
            pending[parent_key] -= 1
            ready = pending[parent_key] == 0
        if not ready:
            return
        t, level, node = parent_key
        lf, rf = fut[(t, level - 1, 2 * node)], fut[(t, level - 1, 2 * node + 1)]
        if lf.exception() or rf.exception():
            future_of(parent_key).set_exception(lf.exception() or rf.exception())
            return
        pool.submit(run_node, parent_key, prove_parent, t, level, node, lf.result(), rf.result())

    messages = []                  # (level of the RECEIVING node, tree, node, src, dst, key of the proof that moves)
    for t, plan in enumerate(plans):
        for level in range(1, L):
            for node in range(plan.nodes(level)):
                dst, src = plan.owner(level, node), plan.owner(level - 1, 2 * node + 1)
                if dst == rank:
                    key = (t, level, node)
                    pending[key] = 2
                    for child in (2 * node, 2 * node + 1):
                        future_of((t, level - 1, child)).add_done_callback(lambda f, k=key: child_done(k, f))
                if src != dst:
                    messages.append((level, t, node, src, dst, (t, level - 1, 2 * node + 1)))
        root_rank = plan.owner(L - 1, 0)
        if root_rank != 0:
            messages.append((L, t, 0, root_rank, 0, (t, L - 1, 0)))
    messages.sort(key=lambda m: m[:3])
    mine = [m for m in messages if rank in (m[3], m[4])]

    def comm():
        done = 0
        try:
            if getattr(device, "type", None) == "cuda":   # the current device is per thread
                import torch
                torch.cuda.set_device(device)
            for _, _, _, src, dst, key in mine:
                if src == rank:
                    _send_bytes(dist, encode(future_of(key).result()), dst, device)
                else:
                    future_of(key).set_result(decode(_recv_bytes(dist, src, device)))
                done += 1
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            for _, _, _, src, _, key in mine[done:]:   # nobody may wait for a proof that will not arrive
                if src != rank and not future_of(key).done():
                    future_of(key).set_exception(e)

    comm_thread = threading.Thread(target=comm, name="p3r-tree-comm") if mine else None
    try:
        if comm_thread:
            comm_thread.start()
        for t, plan in enumerate(plans):
            for i in plan.my_nodes(0, rank):
                pool.submit(run_node, (t, 0, i), prove_leaf, t, i)
        roots = None
        if rank == 0:
            roots = [future_of((t, L - 1, 0)).result() for t in range(n_trees)]
        else:
            # this rank is done when everything it proves is proved and everything it sends is sent
            for t, plan in enumerate(plans):
                for level in range(L):
                    for node in plan.my_nodes(level, rank):
                        future_of((t, level, node)).result()
        if comm_thread:
            comm_thread.join()
        if errors:
            raise errors[0]
        return roots
    finally:
        pool.shutdown(wait=True)


def gather_proofs_to_root(proof: bytes, dist, rank: int, world: int, device="cpu") -> Optional[List[bytes]]:
    """Independent proofs (one per rank) handed to rank 0: used by bench.py's weak-scaling run."""
    if dist is None:
        return [proof]
    if world == 1:
        # a process group of one (bench.py --force-dist): the proof still makes the trip through the collective library -
        # length and payload as device tensors, broadcast from rank 0 - so that a one-GPU box exercises the same path
        import torch
        n = torch.tensor([len(proof)], dtype=torch.int64, device=device)
        dist.broadcast(n, 0)
        buf = torch.frombuffer(bytearray(proof), dtype=torch.uint8).to(device)
        dist.broadcast(buf, 0)
        return [bytes(buf[: int(n.item())].cpu().numpy().tobytes())]
    if rank == 0:
        out = [proof]
        for src in range(1, world):
            out.append(_recv_bytes(dist, src, device))
        return out
    _send_bytes(dist, proof, 0, device)
    return None


# ---------------------------------------------------------------------------------------------------------------------
# What a run should cost: the schedule of run_aggregation_forest replayed on measured per-proof and per-message times.
def predict_forest_wall_ms(plans, leaf_ms: float, node_ms: float, message_ms: float = 0.0, workers: int = 1,
                           gpu_of_rank=None, gpu_capacity: float = 1.0, handoff_ms: Optional[float] = None):
    """Wall time of `run_aggregation_forest(plans, ..)` predicted from solo measurements - the number a first multi-GPU run
    is compared with (recursion/examples/recursive_aggregation.rs:447-475 is the serial loop being parallelised).

      leaf_ms / node_ms   one prove_next_layer / prove_aggregation_layer alone on an idle GPU (latency, not throughput)
      message_ms          one child proof changing rank: serialise + send/recv + native parse (the receiving side's decode
                          overlaps its other work but not the parent that waits for it)
      workers             concurrent provers per rank
      gpu_of_rank         rank -> GPU identity (ranks sharing a GPU share its capacity); default: one GPU per rank
      gpu_capacity        how many solo-speed proofs a GPU sustains at once (measured: concurrent throughput x solo
                          latency) - a number, or (for leaves, for nodes): a proof alone uses 1 / capacity of the GPU, the
                          proofs in flight on one GPU run at full speed while their shares sum to at most 1 and are all
                          slowed by that sum beyond it
      handoff_ms          the root's trip to rank 0 (default: message_ms)

    The model is the scheduler's own: a node becomes ready when both children are on its rank (the left one is there when
    it finishes, the right one message_ms after it finishes unless it was proved on the same rank; one communication
    thread per rank moves messages in (level, tree, node) order), ready nodes take the first free prover of their rank in
    ready order, and the proofs in flight on one GPU share it (processor sharing above `gpu_capacity`).
    Returns dict(wall_ms, critical_path_ms, busy_ms_per_rank, nodes)."""
    import heapq
    world = plans[0].world
    L = plans[0].levels
    gpu_of_rank = gpu_of_rank or {r: r for r in range(world)}
    handoff_ms = message_ms if handoff_ms is None else handoff_ms
    work = lambda lv: leaf_ms if lv == 0 else node_ms   # noqa: E731
    caps = gpu_capacity if isinstance(gpu_capacity, (tuple, list)) else (gpu_capacity, gpu_capacity)
    share = lambda k: 1.0 / max(caps[0 if k[1] == 0 else 1], 1e-9)   # noqa: E731
    keys = [(t, lv, i) for t, p in enumerate(plans) for lv in range(L) for i in range(p.nodes(lv))]
    owner = {(t, lv, i): plans[t].owner(lv, i) for t, lv, i in keys}
    # critical path (no contention, no queueing): longest dependent chain including the messages on it
    cp = {}
    for t, lv, i in sorted(keys, key=lambda k: k[1]):
        if lv == 0:
            cp[(t, lv, i)] = leaf_ms
        else:
            l, r = (t, lv - 1, 2 * i), (t, lv - 1, 2 * i + 1)
            cp[(t, lv, i)] = node_ms + max(cp[l], cp[r] + (message_ms if owner[r] != owner[(t, lv, i)] else 0.0))
    critical = max(cp[(t, L - 1, 0)] + (handoff_ms if owner[(t, L - 1, 0)] != 0 else 0.0) for t in range(len(plans)))
    # event simulation
    arrived = {}                                  # child key -> time it is available on its parent's rank
    missing = {k: 2 for k in keys if k[1] > 0}
    ready = {r: [] for r in range(world)}         # per rank: heap of (ready time, seq, key)
    running = {}                                  # key -> remaining solo-ms
    free = {r: workers for r in range(world)}
    comm_free = {r: 0.0 for r in range(world)}    # when the rank's communication thread is next idle
    finish = {}
    busy = {r: 0.0 for r in range(world)}
    seq = 0
    for k in keys:
        if k[1] == 0:
            heapq.heappush(ready[owner[k]], (0.0, seq, k))
            seq += 1
    now = 0.0
    pending_arrivals = []                         # heap of (time, seq, parent key)

    def rate(k):
        g = gpu_of_rank[owner[k]]
        load = sum(share(q) for q in running if gpu_of_rank[owner[q]] == g)
        return min(1.0, 1.0 / load) if load else 1.0

    def start_ready():
        for r in range(world):
            while free[r] and ready[r] and ready[r][0][0] <= now + 1e-12:
                _, _, k = heapq.heappop(ready[r])
                running[k] = work(k[1])
                free[r] -= 1

    roots_at_0 = {}
    while len(finish) < len(keys) or pending_arrivals:
        start_ready()
        # next event: a running proof finishes, a message arrives, or a queued node becomes ready
        t_next, what = None, None
        for k, rem in running.items():
            tf = now + rem / rate(k)
            if t_next is None or tf < t_next:
                t_next, what = tf, ("finish", k)
        if pending_arrivals and (t_next is None or pending_arrivals[0][0] < t_next):
            t_next, what = pending_arrivals[0][0], ("arrive", None)
        for r in range(world):
            if free[r] and ready[r] and (t_next is None or ready[r][0][0] < t_next):
                t_next, what = ready[r][0][0], ("ready", None)
        if t_next is None:
            break
        dt = max(t_next - now, 0.0)
        for k in list(running):
            running[k] -= dt * rate(k)
            busy[owner[k]] += dt
        now = t_next
        if what[0] == "finish":
            k = what[1]
            del running[k]
            free[owner[k]] += 1
            finish[k] = now
            t, lv, i = k
            if lv == L - 1:
                roots_at_0[t] = now + (handoff_ms if owner[k] != 0 else 0.0)
                continue
            parent = (t, lv + 1, i // 2)
            if owner[parent] == owner[k]:
                at = now
            else:   # through both ranks' communication threads, in order
                at = max(now, comm_free[owner[k]], comm_free[owner[parent]]) + message_ms
                comm_free[owner[k]] = comm_free[owner[parent]] = at
            heapq.heappush(pending_arrivals, (at, seq, parent))
            seq += 1
        elif what[0] == "arrive":
            at, _, parent = heapq.heappop(pending_arrivals)
            missing[parent] -= 1
            if missing[parent] == 0:
                heapq.heappush(ready[owner[parent]], (at, seq, parent))
                seq += 1
    wall = max(list(roots_at_0.values()) + [now])
    return {"wall_ms": wall, "critical_path_ms": critical, "busy_ms_per_rank": [busy[r] for r in range(world)], "nodes": len(keys)}
