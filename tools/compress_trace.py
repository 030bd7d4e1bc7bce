"""Per-launch view of the Merkle layers of ONE prove_next_layer from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/compress_trace.py run [log_h]
   python3 tools/compress_trace.py report OUT/<...>_kernel_trace.csv
For every k_mmcs_compress / k_mmcs_subtree launch of the last proof: grid size, duration, and - for the
one-permutation-per-lane kernel - nodes per second (an injected level runs two permutations per node)."""
import csv
import sys


def run(log_h):
    sys.path.insert(0, "tests"); sys.path.insert(0, ".")
    import bench
    import harness_adapters as wl
    import harness_lib
    import plonky3_recursion_amd as p3r
    ctx = p3r.Context(field="koala-bear", **bench.FRI)
    packing = p3r.TablePacking().with_fri_params(5, 2)
    a = harness_lib.generate("koala-bear", log_h, seed=0x5EED0000, **bench.GEN_KNOBS)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), packing)
    res = pc.upload_inputs(wl.circuit_inputs_from_arrays(a))
    for _ in range(3):
        pc.prove(res)
    ctx.sync()


def report(path):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last proof: from the last k_run_level burst on
    names = [r["Kernel_Name"] for r in rows]
    starts = [i for i, n in enumerate(names) if "k_alu_trace" in n]
    rows = rows[starts[-1]:]
    t_comp = t_sub = 0.0
    n_comp = n_sub = 0
    print("kernel              grid(WGs)   us     Mnodes/s")
    for r in rows:
        n = r["Kernel_Name"]
        if "k_mmcs_compress" not in n and "k_mmcs_subtree" not in n:
            continue
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        wgs = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
        if "compress" in n:
            t_comp += us; n_comp += 1
            print("compress          %9d %7.1f %9.0f" % (wgs, us, wgs * 256 / us))
        else:
            t_sub += us; n_sub += 1
            print("subtree           %9d %7.1f" % (wgs, us))
    print("compress: %d launches %.1f us; subtree: %d launches %.1f us" % (n_comp, t_comp, n_sub, t_sub))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    else:
        report(sys.argv[2])
