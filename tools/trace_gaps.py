"""Launch inventory and idle gaps of one prove_next_layer from a rocprofv3 --kernel-trace CSV.

usage: python tools/trace_gaps.py <kernel_trace.csv> [step_from_end]
A step is delimited by consecutive k_grind launches (one per proof).
"""
import csv
import statistics
import sys
from collections import defaultdict


def short(name):
    return name.split("<")[0].split("::")[-1].split("(")[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    marks = [i for i, r in enumerate(rows) if "k_grind" in r["Kernel_Name"]]
    seg = rows[marks[-back - 1] + 1 : marks[-back] + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print(f"launches {len(seg)}  span {(t1 - t0) / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms")
    fam = defaultdict(lambda: [0, 0])
    for r in seg:
        f = fam[short(r["Kernel_Name"])]
        f[0] += 1
        f[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:34s} {v[0]:4d} {v[1] / 1e3:9.1f} us")
    gaps = [
        (int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), short(a["Kernel_Name"]), short(b["Kernel_Name"]))
        for a, b in zip(seg, seg[1:])
    ]
    print(f"idle {sum(g for g, _, _ in gaps if g > 0) / 1e6:.3f} ms, median gap {statistics.median(g for g, _, _ in gaps) / 1e3:.1f} us")
    for g in sorted(gaps, reverse=True)[:30]:
        print(f"  {g[0] / 1e3:8.1f} us  {g[1]} -> {g[2]}")


if __name__ == "__main__":
    main()
