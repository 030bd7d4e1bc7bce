"""Host<->device copies of ONE prove_next_layer from a rocprofv3 trace:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 tools/compress_trace.py run [log_h]
   python3 tools/copy_trace.py OUT
Prints, for the last proof of the run, every memory copy (direction, bytes, duration) and the blit kernels
(__amd_rocclr_copyBuffer / fillBuffer) with their durations, in stream order."""
import csv
import glob
import sys


def main(out):
    kt = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
    mc = glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True)
    rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_alu_trace" in r["Kernel_Name"]]
    t0 = int(rows[starts[-1]]["Start_Timestamp"])
    t1 = int(rows[-1]["End_Timestamp"])
    blit = [r for r in rows[starts[-1]:] if "rocclr" in r["Kernel_Name"]]
    print("last proof: %.2f ms of kernels span, %d kernel launches, %d blit kernels (%.1f us)" % (
        (t1 - t0) / 1e6, len(rows) - starts[-1], len(blit),
        sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in blit) / 1e3))
    by = {}
    for r in blit:
        k = r["Kernel_Name"][:40]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        by.setdefault(k, []).append(d)
    for k, v in by.items():
        print("  %-40s n=%3d  total %.1f us  max %.1f us" % (k, len(v), sum(v), max(v)))
    if mc:
        copies = [r for r in csv.DictReader(open(mc[0])) if t0 <= int(r["Start_Timestamp"]) <= t1]
        print("memory copies inside the proof: %d" % len(copies))
        agg = {}
        for r in copies:
            key = r.get("Direction", r.get("Kind", "?"))
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            agg.setdefault(key, []).append(d)
        for k, v in agg.items():
            print("  %-30s n=%3d  total %.1f us  max %.1f us" % (k, len(v), sum(v), max(v)))


if __name__ == "__main__":
    main(sys.argv[1])
