import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import harness_lib, torch
import plonky3_recursion_amd as p3r
import harness_adapters as wl
# usage: python tools/soak.py [mmcs_arity = 2 | 4]
ARITY = int(sys.argv[1]) if len(sys.argv) > 1 else 2
FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15, num_queries=54,
           mmcs_arity=ARITY)
a = harness_lib.generate("koala-bear", 16, seed=3)
ctx = p3r.Context(field="koala-bear", allow_unpinned_w32_defaults=True, **FRI)
tp = p3r.TablePacking().with_fri_params(5, 2)
pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
res = pc.upload_inputs(wl.circuit_inputs_from_arrays(a))
first = pc.prove(res)
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
for i in range(300):
    assert pc.prove(res) == first
free1 = torch.cuda.mem_get_info()[0]
print("mmcs_arity %d: 300 proves identical, %.1f ms each, free HBM change %.1f MB" % (ARITY, (time.time() - t0) / 300 * 1e3, (free0 - free1) / 1e6))
ctx.profile_enable(True)
pc.prove(res)
prof = ctx.profile_read()
ctx.profile_enable(False)
print("  kernel ms:", ", ".join("%s %.3f" % (k, v[0]) for k, v in prof.items() if not k.startswith("stage:") and v[0] >= 0.05))
