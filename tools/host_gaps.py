"""Where the GPU waits for the HOST inside one proof: the idle gaps of a rocprofv3 kernel trace.
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --no-cpu-baseline --no-config2 --no-small-layers --no-quintic --steps 6 --warmup 1
   python tools/host_gaps.py <dir>/**/*_kernel_trace.csv [min_gap_us=8]
Takes the fastest proof of the run (between two query-gather kernels), lists every gap of at
least min_gap_us with the dispatch before and after it, and sums the gaps by the KIND of dispatch that follows (a kernel of
the library, or one of the runtime's own: `copyBuffer` = hipMemcpyAsync, `fillBufferAligned` = hipMemsetAsync)."""
import collections
import csv
import glob
import re
import sys


def short(n):
    m = re.search(r"(k_[a-z0-9_]+|copyBuffer|fillBuffer\w*)", n)
    return m.group(1) if m else n[:32]


def main():
    paths = glob.glob(sys.argv[1], recursive=True)
    min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 8e3
    rows = list(csv.DictReader(open(paths[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gathers = [i for i, r in enumerate(rows) if "k_gather" in r["Kernel_Name"] or "k_query_gather" in r["Kernel_Name"]]
    # proofs of the timed region: consecutive gathers a proof's length apart; take the middle one
    spans = [(a, b, int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) for a, b in zip(gathers[:-1], gathers[1:])]
    # (the FASTEST whole proof: bench.py's per-kernel profiling pass and its warm-up proofs are slower)
    def is_proof(s):
        names = [r["Kernel_Name"] for r in rows[s[0] + 1:s[1] + 1]]
        return sum("k_mmcs_hash_rows<" in n for n in names) >= 3 and any("k_quotient" in n for n in names) and not any("k_roles" in n for n in names)
    a, b, span = min((s for s in spans if is_proof(s)), key=lambda s: s[2])
    seg = rows[a:b + 1]
    by_kind = collections.Counter()
    n_kind = collections.Counter()
    busy = 0
    print(f"# one steady-state proof: {span / 1e6:.3f} ms from the end of one query gather to the end of the next, {len(seg) - 1} dispatches")
    prev = seg[0]
    for r in seg[1:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        gap = s - int(prev["End_Timestamp"])
        kind = "runtime copy" if "copyBuffer" in r["Kernel_Name"] else "runtime fill" if "fillBuffer" in r["Kernel_Name"] else "kernel"
        n_kind[kind] += 1
        if gap > 0:
            by_kind[kind] += gap
        if gap >= min_gap:
            print(f"{gap / 1e3:8.1f} us idle   after {short(prev['Kernel_Name']):26s} before {short(r['Kernel_Name'])}")
        prev = r
    print(f"# busy {busy / 1e6:.3f} ms; idle before a library kernel {by_kind['kernel'] / 1e6:.3f} ms ({n_kind['kernel']} dispatches), "
          f"before a runtime copy {by_kind['runtime copy'] / 1e6:.3f} ms ({n_kind['runtime copy']}), "
          f"before a runtime fill {by_kind['runtime fill'] / 1e6:.3f} ms ({n_kind['runtime fill']})")


if __name__ == "__main__":
    main()
