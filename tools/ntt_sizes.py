"""Per-pass time of one coset LDE (inverse + forward, blow-up 4) by input height: ns per OUTPUT cell and pass, so that the
heights whose passes fall off the 2^20-row rate stand out.
   python tools/ntt_sizes.py [field=koala-bear] [width=64] [first_log=16] [last_log=23]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky3_recursion_amd as p3r  # noqa: E402

field = sys.argv[1] if len(sys.argv) > 1 else "koala-bear"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 16
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 23
rng = np.random.default_rng(1)
ctx = p3r.Context(field=field)
print(f"# {field}, {width} columns, added_bits 2; ms per LDE and ns per 1000 output cells (h * 4 * width) per pass")
for log_n in range(lo, hi + 1):
    w = width if log_n <= 21 else max(8, width >> (log_n - 21))   # bound the memory of the tall ones
    base = ctx.upload(rng.integers(0, 0x78000001, size=(1 << log_n, w), dtype=np.uint32))
    out = ctx.coset_lde_batch_device(base, 2, 3)
    out.free()
    ctx.sync()
    ctx.profile_enable(True)
    for _ in range(3):
        out = ctx.coset_lde_batch_device(base, 2, 3)
        out.free()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    cells = (1 << log_n) * 4 * w
    parts = {k: v[0] / 3 for k, v in prof.items() if not k.startswith("stage:")}
    total = sum(parts.values())
    print(f"2^{log_n} x {w}: {total:8.3f} ms  {total * 1e9 / cells:7.1f} ns/kcell | " +
          "  ".join(f"{k} {v:.3f} ms ({v * 1e9 / cells:.1f})" for k, v in sorted(parts.items())))
    base.free()
    ctx.trim()
ctx.close()
