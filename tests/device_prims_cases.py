"""Run by tests/test_gpu_device_prims.py with P3R_LIB_PATH = the knobs build of the library (the only one that exports the
p3r_test_* seam of csrc/device_prims.hip.h): exclusive sums, maximum and the stable radix sort against numpy."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky3_recursion_amd as p3r  # noqa: E402

ctx = p3r.Context(field="koala-bear")
lib, h = ctx.lib, ctx.h
u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
lib.p3r_test_exclusive_sum_u32.argtypes = [C.c_void_p, u32p, u32p, C.c_size_t, C.c_int]
lib.p3r_test_exclusive_sum_u64.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t]
lib.p3r_test_reduce_max.argtypes = [C.c_void_p, u32p, C.c_size_t, u32p]
lib.p3r_test_sort_pairs.argtypes = [C.c_void_p, u32p, u32p, C.c_size_t, C.c_int, u32p, u32p]
for f in (lib.p3r_test_exclusive_sum_u32, lib.p3r_test_exclusive_sum_u64, lib.p3r_test_reduce_max, lib.p3r_test_sort_pairs):
    f.restype = C.c_int


def p32(a):
    return a.ctypes.data_as(u32p)


rng = np.random.default_rng(5)
SIZES = [1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, 100_003, (1 << 20) + 3, 3_000_001]
checked = 0
for n in SIZES:
    a = rng.integers(0, 1 << 9, size=n, dtype=np.uint32)
    want = (np.cumsum(a, dtype=np.uint64) - a).astype(np.uint32)
    for in_place in (0, 1):
        out = np.empty(n, dtype=np.uint32)
        ctx.check(lib.p3r_test_exclusive_sum_u32(h, p32(a), p32(out), n, in_place))
        assert np.array_equal(out, want), ("exclusive_sum u32", n, in_place)
    # packed pair of counters, as scan_pairs uses it (a flag in the low word, a cell count in the high one)
    b = rng.integers(0, 2, size=n, dtype=np.uint64) | (rng.integers(0, 125, size=n, dtype=np.uint64) << np.uint64(32))
    out64 = np.empty(n, dtype=np.uint64)
    ctx.check(lib.p3r_test_exclusive_sum_u64(h, b.ctypes.data_as(u64p), out64.ctypes.data_as(u64p), n))
    assert np.array_equal(out64, np.cumsum(b, dtype=np.uint64) - b), ("exclusive_sum u64", n)
    m = np.zeros(1, dtype=np.uint32)
    big = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    ctx.check(lib.p3r_test_reduce_max(h, p32(big), n, p32(m)))
    assert int(m[0]) == int(big.max()), ("reduce_max", n)
    checked += 4
m = np.full(1, 7, dtype=np.uint32)
ctx.check(lib.p3r_test_reduce_max(h, p32(m), 0, p32(m)))
assert int(m[0]) == 0

for n in SIZES:
    for bits in (1, 5, 8, 9, 16, 17, 24, 27, 32):
        if n > 200_000 and bits not in (5, 17, 32):
            continue
        for style in ("uniform", "few", "high_bits"):
            if style == "uniform":
                keys = rng.integers(0, 1 << bits, size=n, dtype=np.uint64).astype(np.uint32)
            elif style == "few":      # long runs of equal digits: stability carries the order
                keys = rng.integers(0, min(1 << bits, 3), size=n, dtype=np.uint64).astype(np.uint32)
            else:                     # bits above `bits` are not part of the order
                keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
            vals = np.arange(n, dtype=np.uint32)[::-1].copy()
            ko, vo = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
            keep_k, keep_v = keys.copy(), vals.copy()
            ctx.check(lib.p3r_test_sort_pairs(h, p32(keys), p32(vals), n, bits, p32(ko), p32(vo)))
            mask = np.uint32((1 << bits) - 1) if bits < 32 else np.uint32(0xFFFFFFFF)
            order = np.argsort(keys & mask, kind="stable")
            assert np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order]), ("sort_pairs", n, bits, style)
            assert np.array_equal(keys, keep_k) and np.array_equal(vals, keep_v)
            checked += 1
ctx.close()
print("device_prims ok", checked)
