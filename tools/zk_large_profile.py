"""Per-proof times and the per-kernel / per-stage profile of a large ZK layer under the binary and the arity-4 MMCS.
   python tools/zk_large_profile.py [log_rows=21]"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import harness_lib, torch
import plonky3_recursion_amd as p3r
import harness_adapters as wl
FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15, num_queries=54)
lh = int(sys.argv[1]) if len(sys.argv) > 1 else 21
for kw in (dict(zk=1, num_random_codewords=2, zk_seed=9), dict(zk=1, num_random_codewords=2, zk_seed=9, mmcs_arity=4)):
    a = harness_lib.generate("koala-bear", lh, seed=3)
    ctx = p3r.Context(field="koala-bear", **FRI, **kw, allow_unpinned_w32_defaults=True)
    tp = p3r.TablePacking().with_fri_params(5, 2)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    res = pc.upload_inputs(wl.circuit_inputs_from_arrays(a))
    pc.prove(res)
    ts = []
    for i in range(4):
        t0 = time.time(); pc.prove(res); ts.append((time.time() - t0) * 1e3)
    ctx.profile_enable(True); pc.prove(res); prof = ctx.profile_read(); ctx.profile_enable(False)
    print(lh, kw, ["%.1f" % t for t in ts])
    print("  kernels:", ", ".join("%s %.2f" % (k, v[0]) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]) if not k.startswith("stage:") and v[0] >= 0.3))
    print("  stages:", ", ".join("%s %.2f" % (k[6:], v[0]) for k, v in prof.items() if k.startswith("stage:") and "count" not in k))
    res.free(); pc.free(); ctx.close()
