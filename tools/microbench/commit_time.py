"""Time of MerkleTreeMmcs::commit on one tall matrix: python tools/microbench/commit_time.py [log_h] [width]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import plonky3_recursion_amd as p3r
log_h, w = (int(sys.argv[1]) if len(sys.argv) > 1 else 22), (int(sys.argv[2]) if len(sys.argv) > 2 else 64)
ctx = p3r.Context(field=sys.argv[3] if len(sys.argv) > 3 else "koala-bear")
m = np.random.default_rng(0).integers(0, 0x78000001, size=(1 << log_h, w), dtype=np.uint32)
dm = ctx.upload(m)
cap, tree = ctx.commit_device([dm]); tree.free()
ctx.profile_enable(True)
reps = 5
for _ in range(reps):
    cap, tree = ctx.commit_device([dm]); tree.free()
prof = ctx.profile_read()
rows = 1 << log_h
for k, (ms, n) in prof.items():
    ms /= reps
    perms = rows * ((w + 7) // 8) if k == "mmcs_hash_rows" else rows - 1
    print("%-16s %7.3f ms  %6.2f G perm/s (%d launches)" % (k, ms, perms / ms / 1e6, n // reps))
