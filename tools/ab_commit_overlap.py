#!/usr/bin/env python3
"""Paired A/B of the two-stream commit (prove_impl.hip.h::lde_and_commit) on the headline layer, inside ONE process of the
`knobs` build: the forms alternate proof by proof on the same context, inputs and memory, so box-to-box and run-to-run
drift (+- 0.5 ms between processes on these boxes) cancels.
    plain      (default, the product's path)  one LDE batch, one hash launch per commit
    overlap    P3R_COMMIT_OVERLAP=1           biggest hash class on the second stream while the rest is extended
    split      + P3R_COMMIT_OVERLAP_MODE=1    the same split of LDE and hash on ONE stream: what the split costs
    lowprio    + P3R_COMMIT_OVERLAP_MODE=2    the side stream at the lowest priority
usage (GPU box): P3R_LIB_PATH=plonky3_recursion_amd/knobs/libp3r_hip.so python tools/ab_commit_overlap.py [reps=30] [log_h=20]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("P3R_LIB_PATH", os.path.join(ROOT, "plonky3_recursion_amd", "knobs", "libp3r_hip.so"))
import harness_adapters as wl  # noqa: E402
import harness_lib  # noqa: E402
import plonky3_recursion_amd as p3r  # noqa: E402
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
log_h = int(sys.argv[2]) if len(sys.argv) > 2 else 20
FORMS = {"plain": {}, "overlap": {"P3R_COMMIT_OVERLAP": "1"}, "split": {"P3R_COMMIT_OVERLAP": "1", "P3R_COMMIT_OVERLAP_MODE": "1"},
         "lowprio": {"P3R_COMMIT_OVERLAP": "1", "P3R_COMMIT_OVERLAP_MODE": "2"}}
ctx = p3r.Context(field="koala-bear", **bench.FRI)
packing = p3r.TablePacking().with_fri_params(bench.FRI["log_final_poly_len"], bench.FRI["log_blowup"])
arrs = harness_lib.generate("koala-bear", log_h, seed=0x5EED0000, **bench.GEN_KNOBS)
pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(arrs), packing)
res = pc.upload_inputs(wl.circuit_inputs_from_arrays(arrs))
del arrs


def set_form(env):
    for k in ("P3R_COMMIT_OVERLAP", "P3R_COMMIT_OVERLAP_MODE"):
        os.environ.pop(k, None)
    os.environ.update(env)


times = {k: [] for k in FORMS}
digests = set()
for name, env in FORMS.items():   # warm every form once
    set_form(env)
    digests.add(hashlib.sha256(pc.prove(res)).hexdigest())
for r in range(reps):
    for name, env in FORMS.items():
        set_form(env)
        ctx.sync()
        t = time.perf_counter()
        pc.prove(res)
        ctx.sync()
        times[name].append((time.perf_counter() - t) * 1e3)
base = sorted(times["plain"])
print(f"headline layer 2^{log_h} rows, {reps} proofs per form, alternating; proofs identical: {len(digests) == 1}")
for name, v in times.items():
    s = sorted(v)
    paired = sorted(a - b for a, b in zip(v, times["plain"]))
    print(f"{name:8s} median {s[len(s) // 2]:.3f} ms  min {s[0]:.3f}  mean {sum(v) / len(v):.3f}   vs plain (paired median) {paired[len(paired) // 2]:+.3f} ms")
