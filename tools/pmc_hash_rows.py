#!/usr/bin/env python3
"""N launches of k_mmcs_hash_rows over one matrix of known shape, to be run under
`rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace` (tools/profile_round.sh): the permutation count of the run is
exact (rows x ceil(width / 8) per launch), so SQ_INSTS_VALU x 64 lanes / permutations is the kernel's FP64 /
VALU instruction count per Poseidon2 permutation - the figure bench.py's `valu_roofline` prices.
tools/collect_profiles.py turns the counter file into profiles/<round>/pmc_hash_rows.json.
usage: python3 tools/pmc_hash_rows.py <field> [log_rows] [width] [launches]
       python3 tools/pmc_hash_rows.py <field> w32 builtin|general    (the arity-4 leaf kernel over the width-32 permutation: the
           built-in diagonal's instance, or the general one, reached by configuring a random diagonal)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plonky3_recursion_amd as p3r  # noqa: E402

field = sys.argv[1] if len(sys.argv) > 1 else "koala-bear"
rng = np.random.default_rng(1)
if len(sys.argv) > 2 and sys.argv[2] == "w32":
    which = sys.argv[3] if len(sys.argv) > 3 else "builtin"
    kw = {}
    if which == "general":   # any diagonal that is not the built-in one takes the general kernel
        dflt = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon2_w32_default.json")))[field.replace("-", "_")]
        kw = dict(poseidon2_w32_rc=np.array(dflt["rc"], dtype=np.uint32).reshape(-1),
                  poseidon2_w32_diag=rng.integers(1, 0x78000001, size=32, dtype=np.uint32))
    ctx = p3r.Context(field=field, mmcs_arity=4, **kw, allow_unpinned_w32_defaults=True)
    log_rows, width, launches = 22, 96, 3
    m = ctx.upload(rng.integers(0, ctx.p, size=(1 << log_rows, width), dtype=np.uint32))
    for _ in range(launches):
        cap, tree = ctx.commit_device([m])
        tree.free()
    ctx.sync()
    print(json.dumps({"field": field, "kernel": "k_mmcs4_hash_rows", "diagonal": which, "rows": 1 << log_rows, "width": width,
                      "launches": launches, "perms_per_launch": (1 << log_rows) * ((width + 23) // 24)}))
    ctx.close()
    sys.exit(0)
ctx = p3r.Context(field=field)
if len(sys.argv) > 2 and sys.argv[2] == "bench":
    # the three commits of bench.py's 2^20-row layer (main, LogUp aux, quotient chunks; default packing): one job-list
    # launch per commit over every height class, as in a proof
    import bench
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    packing = p3r.TablePacking()
    k, B = packing.horner_packed_steps, 4
    heights = [1 << 16, 1 << 19, 1 << 20, 1 << 19, 1 << 18]
    widths = [4, 4 * packing.public_lanes, 16 * packing.alu_lanes + ((k - 1) // 2 + 2 * (k - 1) + 1) * 4,
              166 if field == "koala-bear" else 300, 4 * packing.recompose_lanes]
    names = ["const", "public", "alu", "poseidon2", "recompose"]
    aux = bench.lookup_aux_widths(packing.alu_lanes, k)
    commits = [[(heights[i] * B, widths[i]) for i in range(5)],
               [(heights[i] * B, aux[n][0] * 4) for i, n in enumerate(names)],
               [(heights[i] * B, 4) for i, n in enumerate(names) for _ in range(aux[n][1])]]
    perms = 0
    for mats in commits:
        by_h = {}
        for h, w in mats:
            by_h[h] = by_h.get(h, 0) + w
        perms += sum(h * ((w + 7) // 8) for h, w in by_h.items())
        dm = [ctx.upload(rng.integers(0, ctx.p, size=(h, w), dtype=np.uint32)) for h, w in mats]
        for _ in range(launches):
            cap, tree = ctx.commit_device(dm)
            tree.free()
        ctx.sync()
        for m in dm:
            m.free()
    print(json.dumps({"field": field, "shapes": "bench.py 2^20-row layer: main / aux / quotient commits", "commits": commits,
                      "launches": 3 * launches, "perms_per_launch": perms / 3.0}))
else:
    log_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 22
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    launches = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    m = ctx.upload(rng.integers(0, ctx.p, size=(1 << log_rows, width), dtype=np.uint32))
    for _ in range(launches):
        cap, tree = ctx.commit_device([m])
        tree.free()
    ctx.sync()
    print(json.dumps({"field": field, "rows": 1 << log_rows, "width": width, "launches": launches,
                      "perms_per_launch": (1 << log_rows) * ((width + 7) // 8)}))
ctx.close()
