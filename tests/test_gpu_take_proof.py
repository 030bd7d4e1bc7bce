"""GPU: a proof that does not fit the caller's buffer is KEPT, not recomputed (p3r_take_proof).  The pattern every binding
of a growable byte vector uses - call, on P3R_EBUFFER resize to the reported length, call again - cost a second proof per
call, and under zk = 1 the second proof is ANOTHER one whose varint-encoded length may exceed the buffer again (found by
tools/soak_zk_large.py: a 1.06 MB arity-4 ZK proof against the wrappers' 1 MiB buffer alternated 174 / 348 ms)."""
import ctypes as C

import numpy as np
import pytest

import harness_adapters as wl
import harness_lib

pytestmark = pytest.mark.gpu
FRI = dict(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=4, num_queries=6)
EBUFFER, EINVAL = -6, -1


def _setup(**kw):
    import plonky3_recursion_amd as p3r
    a = harness_lib.generate("koala-bear", 8, seed=41, horner_chain_len=12, sponge_chain_len=3, merkle_depth=4)
    ctx = p3r.Context(field="koala-bear", **FRI, **kw)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    return p3r, ctx, pc, pc.upload_inputs(wl.circuit_inputs_from_arrays(a))


def _raw_prove(ctx, pc, rin, cap):
    buf = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t()
    rc = ctx.lib.p3r_prove_next_layer_resident(ctx.h, pc.h, rin.h, 0, C.cast(buf, C.POINTER(C.c_uint8)), cap, C.byref(n))
    return rc, n.value, buf.raw[:n.value] if rc == 0 else None


def _take(ctx, cap):
    buf = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t()
    rc = ctx.lib.p3r_take_proof(ctx.h, C.cast(buf, C.POINTER(C.c_uint8)), cap, C.byref(n))
    return rc, n.value, buf.raw[:n.value] if rc == 0 else None


def test_a_proof_that_did_not_fit_is_handed_over_not_recomputed():
    p3r, ctx, pc, rin = _setup()
    want = pc.prove(rin)
    assert _take(ctx, 1 << 20)[0] == EINVAL and "no proof is waiting" in ctx.lib.p3r_last_error(ctx.h).decode()
    rc, n, _ = _raw_prove(ctx, pc, rin, 100)
    assert rc == EBUFFER and n == len(want)
    rc, n2, _ = _take(ctx, n - 1)                 # still too small: the proof stays
    assert rc == EBUFFER and n2 == n
    rc, n3, got = _take(ctx, n)
    assert rc == 0 and got == want
    assert _take(ctx, n)[0] == EINVAL             # taken: gone
    # a prove call that fits drops a proof that was waiting
    assert _raw_prove(ctx, pc, rin, 100)[0] == EBUFFER
    rc, _, got = _raw_prove(ctx, pc, rin, 1 << 20)
    assert rc == 0 and got == want
    assert _take(ctx, 1 << 20)[0] == EINVAL
    rin.free(); pc.free(); ctx.close()


def test_the_python_wrapper_takes_instead_of_proving_twice_under_zk():
    """Under ZK every prove call advances the proof counter: a wrapper that retried would show it."""
    p3r, ctx, pc, rin = _setup(zk=1, num_random_codewords=2, zk_seed=5)
    ctx._proof_buf = C.create_string_buffer(64)   # far too small: the first call must come back through p3r_take_proof
    before = ctx.zk_nonce
    proof = pc.prove(rin)
    assert ctx.zk_nonce == before + 1, "one prove call, one proof"
    assert len(ctx._proof_buf) >= len(proof)
    prover = p3r.BatchStarkProver(ctx)
    prover.verify_all_tables(prover.wrap_proof(proof, pc.circuit_prover_data))
    second = pc.prove(rin)                         # the grown buffer now fits: no detour
    assert ctx.zk_nonce == before + 2 and second != proof
    rin.free(); pc.free(); ctx.close()


def test_predicted_waits_leave_the_bytes_alone_when_shapes_alternate():
    """The host sleeps through most of what a transcript round trip took in the previous proof of the same shape
    (csrc/context.h::HostPost::post).  Two shapes alternating on one context, and a profiled proof in between (no
    prediction there: the stage marks drain the stream), must keep giving the same bytes - a misprediction may cost time,
    never a result."""
    import plonky3_recursion_amd as p3r
    ctx = p3r.Context(field="koala-bear")      # the headline FRI parameters: waits of a millisecond and more at 2^16 rows
    tp = p3r.TablePacking().with_fri_params(5, 2)
    made = []
    for log_h in (16, 17):
        a = harness_lib.generate("koala-bear", log_h, seed=7 + log_h)
        pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
        rin = pc.upload_inputs(wl.circuit_inputs_from_arrays(a))
        made.append((pc, rin, pc.prove(rin)))
    for k in range(3):
        for pc, rin, want in made:
            assert pc.prove(rin) == want
        if k == 1:
            ctx.profile_enable(True)
            assert made[1][0].prove(made[1][1]) == made[1][2]
            assert any(name.startswith("stage:") for name in ctx.profile_read())
            ctx.profile_enable(False)
    for pc, rin, _ in made:
        rin.free(); pc.free()
    ctx.close()
