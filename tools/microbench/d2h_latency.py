import sys, time
sys.path.insert(0, ".")
import numpy as np
import plonky3_recursion_amd as p3r
ctx = p3r.Context(field="koala-bear")
m = ctx.upload(np.arange(8, dtype=np.uint32).reshape(1, 8))
for _ in range(100): m.download()
t0 = time.perf_counter()
n = 2000
for _ in range(n): m.download()
print("tiny D2H round trip (ctypes call + hipMemcpyAsync + hipStreamSynchronize): %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
import torch
a = torch.zeros(8, dtype=torch.int32, device="cuda")
h = torch.zeros(8, dtype=torch.int32).pin_memory()
for _ in range(100): h.copy_(a, non_blocking=True); torch.cuda.current_stream().synchronize()
t0 = time.perf_counter()
for _ in range(n): h.copy_(a, non_blocking=True); torch.cuda.current_stream().synchronize()
print("torch pinned D2H + stream sync: %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
