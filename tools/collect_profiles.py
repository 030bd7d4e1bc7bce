#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh (gpurun_out/final/) into the committed summaries
under profiles/r01/: the bench line, rocprofv3's kernel stats and the per-kernel PMC traffic."""
import collections
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, OUT = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles", "r01")


def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1]


def pmc(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(path + "/*/*_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            acc[m.group(1) if m else r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    return acc


line = json.load(open(os.path.join(SRC, "bench_line.json")))
json.dump(line, open(os.path.join(OUT, "bench_line_final.json"), "w"), indent=1)
shutil.copy(newest(SRC + "/stats/*/*_kernel_stats.csv"), os.path.join(OUT, "prove_next_layer_final_kernel_stats.csv"))
fe, wr = pmc(SRC + "/pmc_fetch", "FETCH_SIZE"), pmc(SRC + "/pmc_write", "WRITE_SIZE")
# k_mmcs_hash_rows runs once per commit; its first launch of the run is the preprocessed commit of
# the circuit preparation, which is not part of a prove_next_layer: keep the proofs' launches only.
for acc in (fe, wr):
    acc["k_mmcs_hash_rows"] = acc["k_mmcs_hash_rows"][1:]
kern = {k: {"launches": len(fe.get(k, [])),
            "fetch_kb_per_launch": sum(fe.get(k, [0])) / max(len(fe.get(k, [])), 1),
            "write_kb_per_launch": sum(wr.get(k, [0])) / max(len(wr.get(k, [])), 1)} for k in sorted(set(fe) | set(wr))}
json.dump({
    "provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
                  "`python3 bench.py --no-cpu-baseline --no-config2 --steps 1 --warmup 0` (tools/profile_round.sh), MI355X, round 1 "
                  "final; counter unit KB; averages over every launch of the kernel in the run (circuit preparation + "
                  "3 prove_next_layer), except k_mmcs_hash_rows: the 9 launches of the 3 proofs (main / LogUp / quotient "
                  "commit each), without the preprocessed commit of the preparation",
    "note": "raw counter values. gfx950 tallies the 128-B requests of a coalesced streaming read at 64 B, so FETCH_SIZE is "
            "doubled before it is compared with bytes (MI355X_MICROARCH.md, HBM section). k_mmcs_hash_rows reads every "
            "LDE cell of a commit exactly once (4 B per lane) and writes one 32-B digest per row: per proof "
            "2 x FETCH + WRITE summed over its three launches is compared with 4*cells + 32*rows in bench.py.",
    "kernels": kern}, open(os.path.join(OUT, "pmc_traffic.json"), "w"), indent=1)
h = kern["k_mmcs_hash_rows"]
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(OUT, "prove_next_layer_final_kernel_stats.csv")))}
hs = next(v for k, v in stats.items() if "k_mmcs_hash_rows<" in k and "strided" not in k)
print("value ms", line["value"], "| hash PMC MB/launch (2*FETCH + WRITE)",
      (2 * h["fetch_kb_per_launch"] + h["write_kb_per_launch"]) * 1024 / 1e6,
      "| bench avg_launch_ms", line["roofline"]["avg_launch_ms"], "| rocprof avg ms", float(hs["AverageNs"]) / 1e6, "calls", hs["Calls"])
print("roofline", {k: line["roofline"][k] for k in ("achieved", "frac", "traffic", "algorithmic_bytes_per_launch")})
print("valu", line.get("valu_roofline"))
print("kernel_ms", {k: round(v, 2) for k, v in line["kernel_ms_per_step"].items()})
print("stage_ms", {k: round(v, 2) for k, v in line["stage_wall_ms_per_step"].items()})
print("cpu", line.get("cpu_baseline"))
