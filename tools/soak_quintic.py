"""Determinism / leak soak under the quintic configuration: N consecutive prove_next_layer calls of one six-table D = 5
circuit (2^16 rows) give the same bytes and leave the free HBM unchanged; then four provers in four host threads, one
context each, prove the same circuit concurrently - every proof the same bytes again.
   python3 tools/soak_quintic.py [n_sequential] [n_per_thread]"""
import sys
import threading
import time

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch

import harness_adapters as wl
import harness_lib
import plonky3_recursion_amd as p3r

FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15, num_queries=54)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50
a = harness_lib.generate("koala-bear", 16, seed=3, flags=harness_lib.RECOMPOSE_BOTH, ext_degree=5)
tp = p3r.TablePacking().with_fri_params(5, 2)


def prover():
    ctx = p3r.Context(field="koala-bear", ext_degree=5, challenge_degree=5, **FRI)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    return ctx, pc, pc.upload_inputs(wl.circuit_inputs_from_arrays(a, 5))


ctx, pc, res = prover()
first = pc.prove(res)
p3r.BatchStarkProver(ctx).verify_all_tables(p3r.BatchStarkProver(ctx).wrap_proof(first, pc.circuit_prover_data))
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
for i in range(N):
    assert pc.prove(res) == first
free1 = torch.cuda.mem_get_info()[0]
print("%d proves identical (verified), %.2f ms each, free HBM change %.1f MB" % (N, (time.time() - t0) / N * 1e3, (free0 - free1) / 1e6))
workers = [prover() for _ in range(4)]
bad = []


def run(w):
    c, p, r = w
    for _ in range(M):
        if p.prove(r) != first:
            bad.append(1)


t0 = time.time()
th = [threading.Thread(target=run, args=(w,)) for w in workers]
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.time() - t0
assert not bad
print("4 concurrent provers x %d proofs: all identical, %.0f proofs/s" % (M, 4 * M / dt))
