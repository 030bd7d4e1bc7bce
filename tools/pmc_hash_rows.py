#!/usr/bin/env python3
"""N launches of k_mmcs_hash_rows over one matrix of known shape, to be run under
`rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace` (tools/profile_round.sh): the permutation count of the run is
exact (rows x ceil(width / 8) per launch), so SQ_INSTS_VALU x 64 lanes / permutations is the kernel's FP64 /
VALU instruction count per Poseidon2 permutation - the figure bench.py's `valu_roofline` prices.
tools/collect_profiles.py turns the counter file into profiles/<round>/pmc_hash_rows.json.
usage: python3 tools/pmc_hash_rows.py <field> [log_rows] [width] [launches]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plonky3_recursion_amd as p3r  # noqa: E402

field = sys.argv[1] if len(sys.argv) > 1 else "koala-bear"
log_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 22
width = int(sys.argv[3]) if len(sys.argv) > 3 else 80
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ctx = p3r.Context(field=field)
rng = np.random.default_rng(1)
m = ctx.upload(rng.integers(0, ctx.p, size=(1 << log_rows, width), dtype=np.uint32))
for _ in range(launches):
    cap, tree = ctx.commit_device([m])
    tree.free()
ctx.sync()
print(json.dumps({"field": field, "rows": 1 << log_rows, "width": width, "launches": launches,
                  "perms_per_launch": (1 << log_rows) * ((width + 7) // 8)}))
ctx.close()
