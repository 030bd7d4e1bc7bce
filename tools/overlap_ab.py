#!/usr/bin/env python3
"""A/B for "two streams inside a commit" (VERDICT r4 item 4b): does the coset LDE of one height class hide inside the
VALU-bound leaf hashing of another when the two run on different HIP streams?

Two p3r contexts on one GPU are two streams with their own pools, so the experiment needs no library change: context H
commits (k_mmcs_hash_rows + Merkle levels) an already extended matrix while context L extends another one
(coset_lde_batch), from two host threads.  Shapes are the two big classes of the headline layer's main commit:
    hash   2^21 rows x 170 columns  (Poseidon2 + Public LDEs: 46 M permutations)
    LDE    2^20 rows x 80 columns -> 2^22 rows (the ALU trace)
Reported: each alone, both back to back on one stream, both concurrently; the saving is what a two-stream commit
could gain per such pair at best (the library would also have to split its one hash launch per commit in two).
usage (GPU box): python tools/overlap_ab.py [reps=5]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky3_recursion_amd as p3r  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(1)
P = 0x7F000001
H, L = p3r.Context(field="koala-bear"), p3r.Context(field="koala-bear")
hash_in = H.coset_lde_batch_device(H.upload(rng.integers(0, P, size=(1 << 19, 170), dtype=np.uint32)), 2, 3)   # 2^21 x 170
lde_in = L.upload(rng.integers(0, P, size=(1 << 20, 80), dtype=np.uint32))
lde_in_h = H.upload(rng.integers(0, P, size=(1 << 20, 80), dtype=np.uint32))


def do_hash(ctx=H):
    cap, tree = ctx.commit_device([hash_in])
    tree.free()


def do_lde(ctx=L, m=lde_in):
    ctx.coset_lde_batch_device(m, 2, 3).free()


def timed(fn):
    fn()
    H.sync(); L.sync()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        H.sync(); L.sync()
        best = min(best, (time.perf_counter() - t) * 1e3)
    return best


def both_concurrent():
    t1, t2 = threading.Thread(target=do_hash), threading.Thread(target=do_lde)
    t1.start(); t2.start(); t1.join(); t2.join()


def both_serial_one_stream():
    do_lde(H, lde_in_h)
    do_hash(H)


hash_ms, lde_ms = timed(do_hash), timed(do_lde)
serial_ms, conc_ms = timed(both_serial_one_stream), timed(both_concurrent)
print("leaf hashing + Merkle levels of 2^21 x 170 alone        %.3f ms" % hash_ms)
print("coset LDE 2^20 x 80 -> 2^22 alone                       %.3f ms" % lde_ms)
print("both, back to back on one stream                        %.3f ms" % serial_ms)
print("both, concurrently on two streams (two host threads)    %.3f ms" % conc_ms)
print("saving of the concurrent form                           %.3f ms = %.0f %% of the LDE" % (serial_ms - conc_ms, (serial_ms - conc_ms) / lde_ms * 100))
print("per proof: three commits, of which the main one has two classes of this size; the others are smaller - an upper")
print("bound on what a two-stream commit gains is about 1.5 x this saving")
H.close(); L.close()
