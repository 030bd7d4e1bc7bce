import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import harness_lib, torch
import plonky3_recursion_amd as p3r
import harness_adapters as wl
FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15, num_queries=54)
for field, lh, n in (("koala-bear", 20, 60), ("baby-bear", 18, 100)):
    a = harness_lib.generate(field, lh, seed=3)
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking().with_fri_params(5, 2)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    res = pc.upload_inputs(wl.circuit_inputs_from_arrays(a))
    first = pc.prove(res)
    t0 = time.time()
    for i in range(n):
        assert pc.prove(res) == first, i
    print(field, lh, "%d proves identical, %.1f ms each" % (n, (time.time() - t0) / n * 1e3), flush=True)
    res.free(); pc.free(); ctx.close()
