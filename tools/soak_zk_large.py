import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import harness_lib, torch
import plonky3_recursion_amd as p3r
import harness_adapters as wl
FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15, num_queries=54)
for field, lh, kw in (("koala-bear", 21, dict(zk=1, num_random_codewords=2, zk_seed=9)), ("koala-bear", 21, dict(zk=1, num_random_codewords=2, zk_seed=9, mmcs_arity=4)),
                      ("koala-bear", 21, dict(zk=1, num_random_codewords=2, mmcs_salt_elems=4))):   # the hiding MMCS, keyed by the OS
    a = harness_lib.generate(field, lh, seed=3)
    ctx = p3r.Context(field=field, **FRI, **kw, allow_unpinned_w32_defaults=True)
    tp = p3r.TablePacking().with_fri_params(5, 2)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    res = pc.upload_inputs(wl.circuit_inputs_from_arrays(a))
    free0 = torch.cuda.mem_get_info()[0]
    first = pc.prove(res)
    t0 = time.time()
    n = 3
    for i in range(n):
        pf = pc.prove(res)
    dt = (time.time() - t0) / n * 1e3
    prover = p3r.BatchStarkProver(ctx)
    prover.verify_all_tables(prover.wrap_proof(pf, pc.circuit_prover_data))
    assert pf != first
    print(field, "2^%d rows" % lh, kw, "%.1f ms per proof, %d bytes, verified; free HBM %.1f GB of %.1f GB" % (dt, len(pf), torch.cuda.mem_get_info()[0] / 1e9, torch.cuda.mem_get_info()[1] / 1e9), flush=True)
    res.free(); pc.free(); ctx.close()
