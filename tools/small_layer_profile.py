"""Where a small layer's time goes: per-stage wall clock and per-kernel-family GPU time of one
prove_next_layer (circuit run + prove) at 2^log_h rows, HIP-event bracketing on (the bracketing itself
costs time: compare the families with each other, take the total from bench.py's small_layers).

usage: python tools/small_layer_profile.py [log_h] [field]
"""
import sys
import time

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import bench
import harness_adapters as wl
import harness_lib
import plonky3_recursion_amd as p3r

log_h = int(sys.argv[1]) if len(sys.argv) > 1 else 15
field = sys.argv[2] if len(sys.argv) > 2 else "koala-bear"
ctx = p3r.Context(field=field)
packing = p3r.TablePacking().with_fri_params(5, 2)
arrs = harness_lib.generate(field, log_h, seed=0x5EED0000, **bench.GEN_KNOBS)
cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(arrs), p3r.FriRecursionBackend(),
                                  p3r.ProveNextLayerParams(table_packing=packing))
pc = cache.prepared_circuit
res = pc.upload_inputs(wl.circuit_inputs_from_arrays(arrs))
for _ in range(3):
    pc.prove(res)
ctx.sync()
steps = 20
t0 = time.perf_counter()
for _ in range(steps):
    pc.prove(res)
ctx.sync()
print("2^%d rows, %s: %.3f ms per prove_next_layer (no bracketing)" % (log_h, field, (time.perf_counter() - t0) / steps * 1e3))
ctx.profile_enable(True)
for _ in range(steps):
    pc.prove(res)
prof = ctx.profile_read()
ctx.profile_enable(False)
stages = {k[6:]: v for k, v in prof.items() if k.startswith("stage:")}
kern = {k: v for k, v in prof.items() if not k.startswith("stage:")}
print("stage wall (ms per step, bracketed):")
for k, (ms, n) in stages.items():
    print("  %-22s %7.3f" % (k, ms / steps))
print("kernel families (ms per step, launches per step):")
for k, (ms, n) in sorted(kern.items(), key=lambda kv: -kv[1][0]):
    print("  %-28s %7.3f  %5.1f" % (k, ms / steps, n / steps))
print("sum of kernel families %.3f ms" % (sum(v[0] for v in kern.values()) / steps))
