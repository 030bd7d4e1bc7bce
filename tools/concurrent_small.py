"""Throughput of small (production-size) layers: N independent provers on ONE GPU, one p3r_ctx
(stream + memory pool) and one host thread each."""
import sys, time, threading
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import harness_lib
import plonky3_recursion_amd as p3r
import harness_adapters as wl
FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15, num_queries=54)
LOG_H = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REPS = 40
a = harness_lib.generate("koala-bear", LOG_H, seed=3)
for n_ctx in (tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 4, 8)):
    workers = []
    for i in range(n_ctx):
        ctx = p3r.Context(field="koala-bear", **FRI)
        pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), p3r.TablePacking().with_fri_params(5, 2))
        res = pc.upload_inputs(wl.circuit_inputs_from_arrays(a))
        ref = pc.prove(res)
        workers.append((ctx, pc, res, ref))
    def run(w):
        ctx, pc, res, ref = w
        for _ in range(REPS):
            assert pc.prove(res) == ref
    ts = [threading.Thread(target=run, args=(w,)) for w in workers]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print(f"2^{LOG_H} rows, {n_ctx} concurrent provers: {n_ctx * REPS / dt:7.1f} proofs/s  ({dt / REPS * 1e3:.2f} ms per round of {n_ctx})", flush=True)
    for ctx, pc, res, ref in workers:
        res.free(); pc.free(); ctx.close()
