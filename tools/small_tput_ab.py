"""Item 7 of the round-5 review: small_layer_throughput read 770 proofs/s in round 4's bench line and 583 in round 5's
(eight provers, 2^15 rows).  A/B on ONE box: bench.py's own measurement in fresh processes, alternating
  plain   the product library's behaviour since round 6 - one HIP stream per context
  forced  round 5's behaviour - every context also creates the two side streams and two events of the two-stream commit
          experiment when it is created (knobs library, P3R_FORCE_SIDE_STREAMS=1; they stay idle)
usage: python tools/small_tput_ab.py [rounds]   (needs a GPU; prints one line per run and the medians)"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = os.path.join(ROOT, "plonky3_recursion_amd", "knobs", "libp3r_hip.so")
CODE = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import bench, harness_adapters as wl; "
        "import plonky3_recursion_amd as p3r; tp = p3r.TablePacking().with_fri_params(bench.FRI['log_final_poly_len'], bench.FRI['log_blowup']); "
        "print(json.dumps(bench.small_layer_throughput(p3r, wl, tp, 'koala-bear')))") % (ROOT, ROOT)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
res = {"plain": [], "forced": []}
for r in range(rounds):
    for name in ("plain", "forced"):
        env = dict(os.environ, P3R_LIB_PATH=KNOBS)
        env.pop("P3R_FORCE_SIDE_STREAMS", None)
        if name == "forced":
            env["P3R_FORCE_SIDE_STREAMS"] = "1"
        out = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=env)
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["all_proofs_identical"]
        res[name].append(line["proofs_per_s"])
        print(f"round {r} {name:6s}: {line['proofs_per_s']:7.1f} proofs/s", flush=True)
for name, v in res.items():
    print(f"{name:6s}: median {statistics.median(v):7.1f}  min {min(v):7.1f}  max {max(v):7.1f} proofs/s over {len(v)} fresh processes")
