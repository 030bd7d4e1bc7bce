"""Stage timers of the circuit preparation (p3r_circuit_create) for one synthetic layer:
   python3 tools/prep_breakdown.py [ext_degree] [log_h] [flags]
Prints the prep_* stages of the context's profile (ms), the way bench.py's prep_miss_breakdown_ms reads them."""
import sys
import time

sys.path.insert(0, "tests"); sys.path.insert(0, ".")


def main():
    import bench
    import harness_adapters as wl
    import harness_lib
    import plonky3_recursion_amd as p3r
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    log_h = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    flags = int(sys.argv[3]) if len(sys.argv) > 3 else (harness_lib.RECOMPOSE_COEFF if d != 4 else 0)
    ctx = p3r.Context(field="koala-bear", ext_degree=d, **bench.FRI)
    packing = p3r.TablePacking().with_fri_params(5, 2)
    a = harness_lib.generate("koala-bear", log_h, seed=0x5EED0005, flags=flags, ext_degree=d, **bench.GEN_KNOBS)
    circ = wl.circuit_from_arrays(a)
    for rep in range(2):
        ctx.profile_enable(True)
        t = time.perf_counter()
        pc = p3r.PreparedCircuit(ctx, circ, packing)
        ctx.sync()
        ms = (time.perf_counter() - t) * 1e3
        prof = ctx.profile_read()
        ctx.profile_enable(False)
        print("rep %d: %.1f ms, on device: %s" % (rep, ms, pc.prepared_on_device))
        for k, v in prof.items():
            if k.startswith("stage:prep"):
                print("   %-28s %8.2f" % (k[6:], v[0]))
        pc.free()
    ctx.close()


if __name__ == "__main__":
    main()
