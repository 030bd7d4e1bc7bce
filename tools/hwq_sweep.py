"""Small-layer throughput (bench.small_layer_throughput: N provers = N contexts = N HIP streams on one GPU) against the
number of hardware queues the HIP runtime maps streams onto (GPU_MAX_HW_QUEUES, read when the runtime starts; default 4).
usage: python tools/hwq_sweep.py   (needs a GPU; fresh process per point)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import bench, harness_adapters as wl; "
        "import plonky3_recursion_amd as p3r; tp = p3r.TablePacking().with_fri_params(bench.FRI['log_final_poly_len'], bench.FRI['log_blowup']); "
        "print(json.dumps(bench.small_layer_throughput(p3r, wl, tp, 'koala-bear', provers=int(sys.argv[1]))))") % (ROOT, ROOT)
for provers in (8, 16):
    for q in (None, "2", "4", "8", "16"):
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        if q:
            env["GPU_MAX_HW_QUEUES"] = q
        vals = []
        for _ in range(2):
            out = subprocess.run([sys.executable, "-c", CODE, str(provers)], capture_output=True, text=True, env=env)
            try:
                vals.append(json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])["proofs_per_s"])
            except Exception:
                vals.append(float("nan"))
        print(f"provers {provers:2d}  GPU_MAX_HW_QUEUES {q or 'unset':5s}: " + "  ".join(f"{v:7.1f}" for v in vals) + " proofs/s", flush=True)
