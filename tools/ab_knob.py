#!/usr/bin/env python3
"""Same-box A/B of one tuning knob of the `knobs` build of the library (plonky3_recursion_amd/knobs/libp3r_hip.so) on the
headline layer: alternates `rounds` x (knob unset, knob set), each a fresh process of bench.py's timed region only.
usage (GPU box): python tools/ab_knob.py P3R_NO_COMMIT_OVERLAP[=value] [rounds=3] [extra bench flags ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
knob, _, value = sys.argv[1].partition("=")
value = value or "1"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = sys.argv[3:]
lib = os.path.join(ROOT, "plonky3_recursion_amd", "knobs", "libp3r_hip.so")
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-small-layers", "--no-config2", "--no-quintic", "--steps", "10"] + extra
res = {"unset": [], "set": []}
sha = set()
for r in range(rounds):
    for name, env in (("unset", {}), ("set", {knob: value})):
        e = dict(os.environ, P3R_LIB_PATH=lib)
        e.pop(knob, None)
        e.update(env)
        out = subprocess.run(cmd, capture_output=True, text=True, env=e)
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        res[name].append(line["ms_per_step"])
        sha.add(line["proof_sha256"])
        print(f"round {r} {knob} {name:5s}: {line['ms_per_step']:.3f} ms  verified={line['proof_verified']}", flush=True)
mean = {k: sum(v) / len(v) for k, v in res.items()}
print(f"{knob}: unset {mean['unset']:.3f} ms (min {min(res['unset']):.3f}), set {mean['set']:.3f} ms (min {min(res['set']):.3f}); "
      f"set - unset = {mean['set'] - mean['unset']:+.3f} ms; proofs identical: {len(sha) == 1}")
