import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import harness_lib
import plonky3_recursion_amd as p3r
import harness_adapters as wl
FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15, num_queries=54)
CASES = [(16, {}), (20, {}), (18, dict(horner_chain_len=2600, sponge_chain_len=330)),
         (20, dict(horner_chain_len=2600, sponge_chain_len=330))]
if len(sys.argv) > 1:
    CASES = [CASES[int(sys.argv[1])]]
for log_h, knobs in CASES:
    t0 = time.time()
    a = harness_lib.generate("koala-bear", log_h, seed=1, **knobs)
    t1 = time.time()
    ctx = p3r.Context(field="koala-bear", **FRI)
    tp = p3r.TablePacking().with_fri_params(5, 2)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    t2 = time.time()
    inputs = wl.circuit_inputs_from_arrays(a)
    res = pc.run(inputs); ctx.sync(); res.free()
    ts = []
    for _ in range(5):
        s = time.time(); res = pc.run(inputs); ctx.sync(); ts.append(time.time() - s); res.free()
    ctx.profile_enable(True)
    res = pc.run(inputs); ctx.sync()
    prof = ctx.profile_read(); ctx.profile_enable(False)
    s = time.time(); proof = pc.prove(inputs); tp_ = time.time() - s
    s = time.time(); proof = pc.prove(inputs); tp_ = time.time() - s
    print(f"log_h={log_h} {knobs} ops={len(a['ops'])//8} levels={pc.levels} gen={t1-t0:.2f}s prepare={t2-t1:.2f}s "
          f"run_ms={[round(x*1e3,2) for x in ts]} run_levels_kernel_ms={prof.get('run_levels')} prove_next_layer_ms={tp_*1e3:.1f}", flush=True)
    res.free(); pc.free(); ctx.close()
