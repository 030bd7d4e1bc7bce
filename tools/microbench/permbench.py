import sys; sys.path.insert(0,'.')
import numpy as np, plonky3_recursion_amd as p3r
for field in ("koala-bear","baby-bear"):
    ctx = p3r.Context(field=field)
    n = 1<<22
    d = ctx.upload(np.random.default_rng(0).integers(0, ctx.p, size=(n,16), dtype=np.uint32))
    ms = ctx.time_permute(d, 10)
    print(field, "permute_batch n=2^22: %.3f ms -> %.2f G perms/s" % (ms, n/ms/1e6))
    ctx.close()
