"""Mixed soak: several contexts on one GPU (plain, arity-4 MMCS with width-32 ops, ZK, ZK + hiding MMCS), layers of 2^14 .. 2^18
rows, proofs in random order for `seconds`; deterministic contexts must reproduce their bytes, every proof of a randomised
context must differ from its predecessor and every tenth is verified natively; free HBM is watched.
   python tools/soak_mixed.py [seconds=600]"""
import random
import sys
import time

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import harness_adapters as wl
import harness_lib
import torch
import plonky3_recursion_amd as p3r

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600
FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15, num_queries=54)
CONFIGS = [("plain", dict(), 0, (14, 16, 18)),
           ("arity4+w32ops", dict(mmcs_arity=4, allow_unpinned_w32_defaults=True), harness_lib.P2_W32_OPS, (14, 17)),
           ("zk", dict(zk=1, num_random_codewords=2), 0, (14, 17)),
           ("zk+hiding", dict(zk=1, num_random_codewords=2, mmcs_salt_elems=4), 0, (15, 18)),
           ("hiding only", dict(mmcs_salt_elems=3), 0, (16,))]
tp = p3r.TablePacking().with_fri_params(5, 2)
items = []
for name, kw, flags, sizes in CONFIGS:
    ctx = p3r.Context(field="koala-bear", **FRI, **kw)
    prover = p3r.BatchStarkProver(ctx)
    for lh in sizes:
        a = harness_lib.generate("koala-bear", lh, seed=100 + lh, flags=flags)
        pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
        rin = pc.upload_inputs(wl.circuit_inputs_from_arrays(a))
        first = pc.prove(rin)
        prover.verify_all_tables(prover.wrap_proof(first, pc.circuit_prover_data))
        items.append(dict(name=f"{name} 2^{lh}", randomised=bool(kw.get("zk") or kw.get("mmcs_salt_elems")), pc=pc, rin=rin, last=first,
                          prover=prover, n=1, ms=0.0))
rng = random.Random(1)
free0 = torch.cuda.mem_get_info()[0]
t_end = time.time() + seconds
total = 0
next_report = time.time() + 60
while time.time() < t_end:
    if time.time() >= next_report:
        print(f"  t = {seconds - (t_end - time.time()):5.0f} s: {total} proofs, free HBM change {(free0 - torch.cuda.mem_get_info()[0]) / 1e6:.1f} MB", flush=True)
        next_report += 60
    it = rng.choice(items)
    t0 = time.perf_counter()
    pf = it["pc"].prove(it["rin"])
    it["ms"] += (time.perf_counter() - t0) * 1e3
    if it["randomised"]:
        assert pf != it["last"], it["name"]
    else:
        assert pf == it["last"], it["name"]
    it["last"] = pf
    it["n"] += 1
    total += 1
    if it["n"] % 10 == 0:
        it["prover"].verify_all_tables(it["prover"].wrap_proof(pf, it["pc"].circuit_prover_data))
free1 = torch.cuda.mem_get_info()[0]
print(f"{total} proofs in {seconds:.0f} s over {len(items)} prepared circuits of {len(CONFIGS)} contexts; free HBM change {(free0 - free1) / 1e6:.1f} MB")
for it in items:
    print(f"  {it['name']:24s} {it['n'] - 1:5d} proofs, {it['ms'] / max(1, it['n'] - 1):8.2f} ms each")
