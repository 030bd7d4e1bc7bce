"""Per-pass time of coset_lde_batch (blow-up 4) on one tall matrix: python tools/microbench/lde_time.py [log_h] [width]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import plonky3_recursion_amd as p3r
log_h, w = (int(sys.argv[1]) if len(sys.argv) > 1 else 20), (int(sys.argv[2]) if len(sys.argv) > 2 else 64)
ctx = p3r.Context(field="koala-bear")
m = np.random.default_rng(0).integers(0, 0x7F000001, size=(1 << log_h, w), dtype=np.uint32)
dm = ctx.upload(m)
ctx.coset_lde_batch_device(dm, 2, 3).free()
ctx.profile_enable(True)
reps = 5
for _ in range(reps):
    ctx.coset_lde_batch_device(dm, 2, 3).free()
prof = ctx.profile_read()
cells = (1 << log_h) * w
tot = 0.0
for k, (ms, n) in prof.items():
    if k.startswith("ntt"):
        ms /= reps
        tot += ms
        by = {"ntt_inverse_1": 8, "ntt_inverse_2": 8, "ntt_forward_1": 20, "ntt_forward_2": 32}[k]
        print("%-16s %7.3f ms  %6.0f GB/s (%d B/cell)" % (k, ms, cells * by / ms / 1e6, by))
print("total %.3f ms, %.0f GB/s of 68 B/cell" % (tot, cells * 68 / tot / 1e6))
