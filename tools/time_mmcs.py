"""Commit time of one LDE-sized matrix under the binary and the arity-4 MMCS (leaf hashing / levels), per kernel family.
   python tools/time_mmcs.py [log_rows=22] [width=64]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky3_recursion_amd as p3r  # noqa: E402

log_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 22
width = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(1)
base = rng.integers(0, 0x7F000001, size=(1 << (log_rows - 2), width), dtype=np.uint32)
for arity in (2, 4):
    ctx = p3r.Context(field="koala-bear", mmcs_arity=arity, allow_unpinned_w32_defaults=True)
    lde = ctx.coset_lde_batch_device(ctx.upload(base), 2, 3)
    cap, tree = ctx.commit_device([lde])
    tree.free()
    ctx.sync()
    t = time.perf_counter()
    for _ in range(5):
        cap, tree = ctx.commit_device([lde])
        tree.free()
    ctx.sync()
    ms = (time.perf_counter() - t) / 5 * 1e3
    ctx.profile_enable(True)
    cap, tree = ctx.commit_device([lde])
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    tree.free()
    perms = (1 << log_rows) * ((width + 7) // 8) if arity == 2 else (1 << log_rows) * ((width + 23) // 24)
    print(f"arity {arity}: commit of 2^{log_rows} x {width}: {ms:.3f} ms; " +
          ", ".join(f"{k} {v[0]:.3f} ms" for k, v in prof.items() if not k.startswith("stage:")) +
          f"; leaf permutations {perms / 1e6:.1f} M")
    ctx.close()
