#!/usr/bin/env python3
"""bench.py - one JSON line for the driver (see DESIGN.md "Measurement").

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM.  Multi-GPU: independent proofs shard one per rank with no data-path
collective - weak scaling; only the barrier / max-over-ranks timing
crosses ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

P = {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}
GEN = {"koala-bear": 3, "baby-bear": 31}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec

# Synthetic recursion-layer table mix: heights as
# fractions of H, main widths at D=4, alu_lanes=3, K=4.
def table_shapes(field, log_h):
    h = 1 << log_h
    p2w = 166 if field == "koala-bear" else 300
    return [("const", max(h // 16, 1), 4), ("public", h // 2, 4), ("alu", h, 80),
            ("poseidon2", h // 2, p2w), ("recompose", h // 4, 4)]


def synth_inputs(field, log_h, seed):
    rng = np.random.default_rng(seed)
    p = P[field]
    mats = {}
    for name, hh, w in table_shapes(field, log_h):
        if name == "poseidon2":
            continue
        mats[name] = rng.integers(0, p, size=(hh, w), dtype=np.uint32)
    n = (1 << log_h) // 2
    rows = dict(
        inputs=rng.integers(0, p, size=(n, 16), dtype=np.uint32),
        new_start=(rng.random(n) < 0.05).astype(np.uint8),
        merkle_path=(rng.random(n) < 0.7).astype(np.uint8),
        mmcs_bit=rng.integers(0, 2, size=n, dtype=np.uint8),
        mmcs_index_sum=rng.integers(0, p, size=n, dtype=np.uint32),
    )
    return mats, rows


def perms_in_commit(field, log_h, log_blowup):
    """Poseidon2 permutations one commit-phase step executes (trace fill + leaves + tree)."""
    total = 0
    shapes = table_shapes(field, log_h)
    total += (1 << log_h) // 2  # K3: one per Poseidon2-table row
    by_h = {}
    for _, hh, w in shapes:
        by_h.setdefault(hh << log_blowup, 0)
        by_h[hh << log_blowup] += w
    hmax = max(by_h)
    for hh, w in by_h.items():
        total += hh * ((w + 7) // 8)  # leaf / injected-row sponges
        if hh != hmax:
            total += hh  # injection compress
    total += hmax - 1  # 2-to-1 compressions
    return total, by_h


def commit_phase_step(ctx, d_mats, d_rows, log_blowup, gen):
    """K3 trace fill + K5 LDE of every main table + K6 one MMCS over all of them."""
    tr = ctx.generate_trace_rows_resident(d_rows)
    order = ["const", "public", "alu", "poseidon2", "recompose"]
    srcs = dict(d_mats)
    srcs["poseidon2"] = tr
    ldes = [ctx.coset_lde_batch_device(srcs[k], log_blowup, gen) for k in order]
    cap, tree = ctx.commit_device(ldes)
    tree.free()
    for m in ldes:
        m.free()
    tr.free()
    return cap


def cpu_baseline(field, log_h_sample, log_blowup):
    """The oracle (kind "port", 1 thread) on a bounded sample of the same workload."""
    import oracle_lib
    orc = oracle_lib.Oracle()
    mats, rows = synth_inputs(field, log_h_sample, 1)
    t0 = time.perf_counter()
    tr = orc.trace_rows(field, rows["inputs"], rows["new_start"], rows["merkle_path"], rows["mmcs_bit"],
                        rows["mmcs_index_sum"])
    srcs = dict(mats)
    srcs["poseidon2"] = tr
    ldes = [orc.coset_lde(field, srcs[k], log_blowup, GEN[field])
            for k in ["const", "public", "alu", "poseidon2", "recompose"]]
    orc.commit(field, ldes, 0)
    dt = time.perf_counter() - t0
    nperm, _ = perms_in_commit(field, log_h_sample, log_blowup)
    return dt, nperm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-height", type=int, default=20)
    ap.add_argument("--field", default="koala-bear")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import plonky3_recursion_amd as p3r

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    field, log_h, log_blowup = args.field, args.log_height, 2
    ctx = p3r.Context(field=field, device=local_rank, log_blowup=log_blowup)
    mats, rows = synth_inputs(field, log_h, 0x5EED0000 + rank)
    d_mats = {k: ctx.upload(v) for k, v in mats.items()}
    d_rows = ctx.upload_p2_rows(rows["inputs"], rows["new_start"], rows["merkle_path"], rows["mmcs_bit"],
                                rows["mmcs_index_sum"])
    del mats, rows

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        commit_phase_step(ctx, d_mats, d_rows, log_blowup, GEN[field])
    ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        commit_phase_step(ctx, d_mats, d_rows, log_blowup, GEN[field])
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile_enable(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    if rank == 0:
        nperm, by_h = perms_in_commit(field, log_h, log_blowup)
        # dominant kernel: MMCS leaf hashing. Algorithmic bytes per launch = every LDE cell
        # read once (4 B) + 8 digest words written per row (DESIGN.md "K6").
        hash_ms, hash_launches = prof.get("mmcs_hash_rows", (0.0, 0))
        cells = sum(hh * w for hh, w in by_h.items())
        rows_hashed = sum(by_h)
        alg_bytes_per_step = 4 * cells + 32 * rows_hashed
        launches_per_step = hash_launches / args.steps if args.steps else 0
        avg_launch_ms = hash_ms / hash_launches if hash_launches else float("nan")
        achieved = (alg_bytes_per_step / launches_per_step) / (avg_launch_ms * 1e-3) / 1e9 if hash_launches else None
        line = {
            "metric": "prove_next_layer ms + Poseidon2 perms/s, KoalaBear 2^20-row circuit, 1/8 GPU",
            "value": ms_per_step,
            "unit": "ms",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": False,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 (31-bit Montgomery prime field)",
            "data": "synthetic",
            "config": {
                "workload": f"PARTIAL hot path: main-trace commit phase only (K3 Poseidon2 trace fill + K5 coset LDE + "
                            f"K6 MMCS commit) of the synthetic {field} 2^{log_h}-row recursion layer "
                            f"(tables const/public/alu/poseidon2/recompose, blowup 4); quotient/FRI not yet in the timed region",
                "field": field, "log_height": log_h, "log_blowup": log_blowup,
                "independent_proofs": world,
            },
            "poseidon2_perms_per_s": nperm * world / (ms_per_step * 1e-3),
            "poseidon2_perms_per_step": nperm,
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
            "roofline": {
                "kernel": "k_mmcs_hash_rows",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": None,
                "avg_launch_ms": avg_launch_ms,
                "algorithmic_bytes_per_launch": alg_bytes_per_step / launches_per_step if launches_per_step else None,
                "note": "Poseidon2 hashing is integer-VALU bound, not HBM bound (616 modmul per 32 B absorbed); "
                        "see DESIGN.md for the VALU ceiling next to this HBM figure",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            sample_log_h = 11
            cdt, cperm = cpu_baseline(field, sample_log_h, log_blowup)
            line["cpu_baseline"] = {
                "value": cdt * 1e3, "unit": "ms", "cores": 1, "kind": "port",
                "sample": f"same commit phase on the same table mix at 2^{sample_log_h} rows "
                          f"(1/{1 << (log_h - sample_log_h)} of the workload), oracle/ C++ restatement, 1 thread",
                "poseidon2_perms_per_s": cperm / cdt,
            }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
