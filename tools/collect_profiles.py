#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh (gpurun_out/final/) into the committed summaries
under profiles/<round>/: the bench line, rocprofv3's kernel stats, the per-kernel PMC traffic and SQ
instruction counters, the microbenchmark outputs.   usage: python tools/collect_profiles.py r02"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = next((a for a in sys.argv[1:] if not a.startswith("--")), "r05")
SRC, OUT = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles", ROUND)
os.makedirs(OUT, exist_ok=True)


def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1]


def kname(full):
    m = re.search(r"(k_[a-z0-9_]+)", full)
    return m.group(1) if m else full[:40]


def hash_rows_summary():
    """pmc_hash_rows.json from the hashrows_* passes; bench.py reads the instruction count from it, so
    tools/profile_round.sh runs this (`--hash-rows-only`) BEFORE the bench of the same call."""
    # FP64 / VALU instructions per Poseidon2 permutation of k_mmcs_hash_rows: counter total over the launches of
    # tools/pmc_hash_rows.py (matrix of known shape) x 64 lanes / permutations of those launches
    hr = {"provenance": "rocprofv3 --pmc SQ_INSTS_VALU (and SQ_WAVES) --kernel-trace over `python3 tools/pmc_hash_rows.py <field>` "
                        "bench` (tools/profile_round.sh): the three commits (main, LogUp aux, quotient chunks) of bench.py's 2^20-row "
                        "layer, every commit one job-list launch of k_mmcs_hash_rows over all its height classes; "
                        "valu_insts_per_perm = SQ_INSTS_VALU x 64 / permutations of those launches",
          "fields": {}}
    for fld in ("koala-bear", "baby-bear"):
        try:
            meta = json.loads([ln for ln in open(os.path.join(SRC, "hashrows_%s_SQ_INSTS_VALU.log" % fld)) if ln.startswith("{")][-1])
            def total(counter):
                path = newest(SRC + "/hashrows_%s_%s/*/*_counter_collection.csv" % (fld, counter))
                vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                        if r["Counter_Name"] == counter and "k_mmcs_hash_rows<" in r["Kernel_Name"] and "strided" not in r["Kernel_Name"]]
                return sum(vals), len(vals)
            insts, n = total("SQ_INSTS_VALU")
            waves, _ = total("SQ_WAVES")
            perms = n * meta["perms_per_launch"]
            hr["fields"][fld] = dict(meta, launches_counted=n, SQ_INSTS_VALU=insts, SQ_WAVES=waves, permutations=perms,
                                     valu_insts_per_perm=insts * 64.0 / perms)
        except Exception as e:  # a missing pass must not lose the rest of the summaries
            print("pmc_hash_rows: %s: %r" % (fld, e))
    # width-32 permutation (arity-4 MMCS leaf kernel): instructions per permutation of the built-in-diagonal instance and
    # of the general one
    w32 = {}
    for which in ("builtin", "general"):
        try:
            meta = json.loads([ln for ln in open(os.path.join(SRC, "hashrows_w32_%s.log" % which)) if ln.startswith("{")][-1])
            path = newest(SRC + "/hashrows_w32_%s/*/*_counter_collection.csv" % which)
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                    if r["Counter_Name"] == "SQ_INSTS_VALU" and "k_mmcs4_hash_rows<" in r["Kernel_Name"]]
            w32[which] = dict(meta, launches_counted=len(vals), SQ_INSTS_VALU=sum(vals),
                              valu_insts_per_perm=sum(vals) * 64.0 / (len(vals) * meta["perms_per_launch"]))
        except Exception as e:
            print("pmc_hash_rows w32: %s: %r" % (which, e))
    if w32:
        sys.path.insert(0, ROOT)
        import bench
        hr["width32"] = w32
        hr["width32_sources"] = list(bench.W32_KERNEL_SOURCES)
        hr["width32_sources_sha256"] = bench.kernel_source_digest(bench.W32_KERNEL_SOURCES)
    if hr["fields"]:
        # bench.py::committed_valu_model refuses the count when the kernel's sources are not the ones it was measured on
        sys.path.insert(0, ROOT)
        import bench
        hr["kernel_sources"] = list(bench.HASH_KERNEL_SOURCES)
        hr["kernel_sources_sha256"] = bench.kernel_source_digest()
        json.dump(hr, open(os.path.join(OUT, "pmc_hash_rows.json"), "w"), indent=1)


if "--hash-rows-only" in sys.argv:
    hash_rows_summary()
    sys.exit(0)


def pmc(counter):
    """kernel -> list of per-launch counter values, in launch order"""
    acc = collections.defaultdict(list)
    try:
        path = newest(SRC + "/pmc_%s/*/*_counter_collection.csv" % counter)
    except IndexError:
        return acc
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


line = json.load(open(os.path.join(SRC, "bench_line.json")))
json.dump(line, open(os.path.join(OUT, "bench_line_final.json"), "w"), indent=1)
for f in ("bench_default_line.json", "bench_default_detail.json", "bench_default_time.txt", "bench_contract_line.json"):
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(OUT, f))
shutil.copy(newest(SRC + "/stats/*/*_kernel_stats.csv"), os.path.join(OUT, "prove_next_layer_final_kernel_stats.csv"))
for f in ("int_rates.txt", "perm_f64.txt", "valu_classes.txt"):
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(OUT, "microbench_" + f))
fe, wr = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
# k_mmcs_hash_rows runs once per commit; its first launch of the run is the preprocessed commit of
# the circuit preparation, which is not part of a prove_next_layer: keep the proofs' launches only.
for acc in (fe, wr):
    acc["k_mmcs_hash_rows"] = acc["k_mmcs_hash_rows"][1:]
kern = {k: {"launches": len(fe.get(k, [])),
            "fetch_kb_per_launch": sum(fe.get(k, [0])) / max(len(fe.get(k, [])), 1),
            "write_kb_per_launch": sum(wr.get(k, [0])) / max(len(wr.get(k, [])), 1)} for k in sorted(set(fe) | set(wr))}
json.dump({
    "provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
                  "`python3 bench.py --no-cpu-baseline --no-config2 --no-small-layers --steps 1 --warmup 0` "
                  "(tools/profile_round.sh), MI355X, round %s; counter unit KB; averages over every launch of the "
                  "kernel in the run, except k_mmcs_hash_rows: without the preprocessed commit of the preparation" % ROUND,
    "note": "raw counter values. gfx950 tallies the 128-B requests of a coalesced streaming read at 64 B, so FETCH_SIZE is "
            "doubled before it is compared with bytes (MI355X_MICROARCH.md, HBM section).",
    "kernels": kern}, open(os.path.join(OUT, "pmc_traffic.json"), "w"), indent=1)

# SQ counters: totals per kernel over the run (one pass per counter)
sq_names = ["SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
            "SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
            "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INST_CYCLES_VMEM",
            "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"]
sq = {c: pmc(c) for c in sq_names}
kernels = sorted(set().union(*[set(v) for v in sq.values()]))
table = {}
for k in kernels:
    row = {c: sum(sq[c].get(k, [])) for c in sq_names if k in sq[c]}
    row["launches"] = len(sq["SQ_WAVES"].get(k, [])) or len(next(iter(sq.values())).get(k, []))
    if row.get("SQ_WAVES"):
        row["valu_insts_per_wave"] = row.get("SQ_INSTS_VALU", 0) / row["SQ_WAVES"]
        row["lds_insts_per_wave"] = row.get("SQ_INSTS_LDS", 0) / row["SQ_WAVES"]
    table[k] = row
json.dump({"provenance": "rocprofv3 --pmc <counter> --kernel-trace, one pass per counter, same command as pmc_traffic.json; "
                         "sums over every launch of the kernel in the run (1 preparation + 3 prove_next_layer)",
           "proofs_in_run": table.get("k_quotient", {}).get("launches"),   # one k_quotient launch per prove_next_layer
           "kernels": table}, open(os.path.join(OUT, "pmc_sq.json"), "w"), indent=1)

hash_rows_summary()
for f in ("bench_line_babybear_2p22.json", "bench_line_tree_1gpu.json", "bench_line_tree_1gpu_4workers.json",
          "bench_line_forest_1gpu_4trees.json", "bench_line_2ranks_gloo.json", "bench_line_forest_2ranks_gloo.json", "spans.txt"):
    src = os.path.join(SRC, f)
    if os.path.exists(src) and os.path.getsize(src):
        if f.endswith(".json"):   # the ranks' collective layer may print banner lines next to the bench line
            shutil.copy(src, os.path.join(OUT, f))   # bench.py --detail-out: the full result dict
        else:
            shutil.copy(src, os.path.join(OUT, f))

h = kern.get("k_mmcs_hash_rows", {})
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(OUT, "prove_next_layer_final_kernel_stats.csv")))}
hs = next(v for k, v in stats.items() if "k_mmcs_hash_rows<" in k and "strided" not in k)
print("value ms", line["value"], "| hash PMC MB/launch (2*FETCH + WRITE)",
      (2 * h.get("fetch_kb_per_launch", 0) + h.get("write_kb_per_launch", 0)) * 1024 / 1e6,
      "| bench avg_launch_ms", line["roofline"]["avg_launch_ms"], "| rocprof avg ms", float(hs["AverageNs"]) / 1e6, "calls", hs["Calls"])
print("roofline", {k: line["roofline"].get(k) for k in ("bound", "achieved", "frac", "traffic", "valu_insts_per_perm")})
print("hbm side", line["roofline"].get("hbm"))
print("proof_roofline", {k: line.get("proof_roofline", {}).get(k) for k in ("valu_floor_ms", "hbm_floor_ms", "floor_ms", "frac")})
print("kernel_ms", {k: round(v, 2) for k, v in line["kernel_ms_per_step"].items()})
print("stage_ms", {k: round(v, 2) for k, v in line["stage_wall_ms_per_step"].items()})
print("cpu", line.get("cpu_baseline"))
for k in ("k_mmcs_hash_rows", "k_ntt_tile", "k_mmcs_compress"):
    print(k, table.get(k))
