"""GPU: the device-side circuit preparation (csrc/prep_device.hip: preprocessed columns, ALU lane schedule and the
execution schedule built in HBM) against the host restatement of the same steps (P3R_PREP_HOST=1): same
preprocessed commitment, same schedule depth, same run traces, same proof bytes - on the synthetic layers
(every table shape, lane counts, Horner pack sizes, long chains), on random circuits of arbitrary dependency
structure, and on circuits the device pass must hand to the host path for its error."""
import os

import numpy as np
import pytest

import circuit_fuzz
import harness_lib
import oracle_lib

pytestmark = pytest.mark.gpu
FRI = dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
ARRAYS = ("const_values", "public_values", "alu_values", "recompose_values", "recompose_coeff_values", "p2_input_values", "p2_flags",
          "p2_mmcs_index_sum")


DEGREE_CASES = [("koala-bear", 5, 8, 0, None), ("koala-bear", 1, 8, 0, None), ("baby-bear", 1, 9, 0, dict(alu_lanes=2, horner_packed_steps=3)),
                ("koala-bear", 5, 9, harness_lib.RECOMPOSE_COEFF, dict(public_lanes=2, alu_lanes=2, horner_packed_steps=3, recompose_lanes=2)),
                ("koala-bear", 5, 10, harness_lib.RECOMPOSE_BOTH, dict(recompose_lanes=2)),
                ("baby-bear", 1, 8, harness_lib.RECOMPOSE_BOTH, None),
                ("koala-bear", 4, 9, harness_lib.RECOMPOSE_BOTH, dict(public_lanes=3, alu_lanes=4, horner_packed_steps=5)),
                ("baby-bear", 4, 8, harness_lib.RECOMPOSE_COEFF | harness_lib.NO_POSEIDON2, None),
                ("koala-bear", 5, 8, harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE, None),
                ("koala-bear", 5, 12, harness_lib.INDEPENDENT_SPONGES | harness_lib.RECOMPOSE_BOTH,
                 dict(public_lanes=2, alu_lanes=3, horner_packed_steps=4))]


def both_ways(ctx, circuit, tp, inputs, arrays=ARRAYS):
    """(commitment, levels, traces, proof) with the device-side and with the host-side preparation."""
    import plonky3_recursion_amd as p3r
    out = []
    for host in (False, True):
        if host:
            os.environ["P3R_PREP_HOST"] = "1"
        try:
            pc = p3r.PreparedCircuit(ctx, circuit, tp)
        finally:
            os.environ.pop("P3R_PREP_HOST", None)
        assert pc.prepared_on_device == (not host)
        res = pc.run(inputs)
        cpd = pc.circuit_prover_data
        traces = {k: res.download(k) for k in arrays if cpd.rows[p3r.prover.TRACES_ARRAYS[k][1]]}
        out.append((cpd.preprocessed_commitment.copy(), pc.levels, traces, pc.prove(inputs), list(cpd.table_heights),
                    cpd.effective_packing))
        res.free()
        pc.free()
    return out


def check_same(dev, host, what):
    assert np.array_equal(dev[0], host[0]), (what, "preprocessed commitment")
    assert dev[1] == host[1], (what, "schedule levels", dev[1], host[1])
    assert dev[4] == host[4] and dev[5] == host[5], (what, "table heights / effective packing")
    for k in host[2]:
        assert np.array_equal(dev[2][k], host[2][k]), (what, k)
    assert dev[3] == host[3], (what, "proof bytes")


SHAPES = [0, harness_lib.NO_POSEIDON2, harness_lib.NO_RECOMPOSE,
          harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE | harness_lib.SINGLE_PUBLIC,
          harness_lib.NO_POSEIDON2 | harness_lib.NO_ALU, harness_lib.INDEPENDENT_SPONGES]
PACKINGS = [None, dict(public_lanes=2, alu_lanes=1, horner_packed_steps=2, recompose_lanes=2),
            dict(public_lanes=3, alu_lanes=4, horner_packed_steps=5, recompose_lanes=1),
            dict(alu_lanes=2, horner_packed_steps=3)]


@pytest.mark.parametrize("field,log_h,flags,packing,gen", [
    *[("koala-bear", 7, f, None, {}) for f in SHAPES],
    *[("koala-bear", 8, 0, p, {}) for p in PACKINGS[1:]],
    ("baby-bear", 8, 0, None, {}),
    ("koala-bear", 10, 0, PACKINGS[2], dict(horner_chain_len=300, sponge_chain_len=70, merkle_depth=9)),
    ("koala-bear", 12, harness_lib.INDEPENDENT_SPONGES, None, dict(horner_chain_len=2600, sponge_chain_len=330, merkle_depth=20)),
    ("baby-bear", 13, 0, PACKINGS[3], dict(horner_chain_len=64, sponge_chain_len=8, merkle_depth=20)),
    # many workgroups per scan / sort / histogram: the bench's knobs at 2^17 rows (0.6 M ops), and config 2's at 2^16
    ("koala-bear", 17, 0, None, dict(horner_chain_len=64, sponge_chain_len=8, merkle_depth=20)),
    ("baby-bear", 16, harness_lib.INDEPENDENT_SPONGES, None, dict(horner_chain_len=2600, sponge_chain_len=330, merkle_depth=20)),
])
def test_device_preparation_equals_host_preparation(field, log_h, flags, packing, gen):
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    gen = dict(dict(horner_chain_len=16, sponge_chain_len=3, merkle_depth=5), **gen)
    a = harness_lib.generate(field, log_h, seed=77 + log_h, flags=flags, **gen)
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking(**(packing or {})).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    dev, host = both_ways(ctx, wl.circuit_from_arrays(a), tp, wl.circuit_inputs_from_arrays(a))
    check_same(dev, host, (field, log_h, flags, packing))
    ctx.close()


@pytest.mark.parametrize("field,ext_degree,log_h,flags,packing", DEGREE_CASES)
def test_device_preparation_for_circuit_degrees_1_and_5_and_both_recompose_kinds(field, ext_degree, log_h, flags, packing):
    """Round 4: the device pass covers the circuit degrees 1 / 4 / 5 (base-mode Poseidon2 rows with their 62-column
    compact preprocessed layout, D-scaled witness indices, D-coefficient constants / hints / Recompose inputs) and
    Recompose ops of the coefficient-lookup kind (one table, or the second of two): same commitment, schedule, run
    traces and proof bytes as the host restatement (backend/fri.rs:741-852, common.rs:127-390, air.rs:730-763)."""
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    a = harness_lib.generate(field, log_h, seed=91 + log_h, flags=flags, ext_degree=ext_degree,
                             horner_chain_len=200 if log_h >= 10 else 16, sponge_chain_len=3, merkle_depth=5)
    ctx = p3r.Context(field=field, ext_degree=ext_degree, **FRI)
    tp = p3r.TablePacking(**(packing or {})).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    dev, host = both_ways(ctx, wl.circuit_from_arrays(a), tp, wl.circuit_inputs_from_arrays(a, ext_degree))
    check_same(dev, host, (field, ext_degree, log_h, flags, packing))
    ctx.close()


@pytest.mark.parametrize("field,seeds,n_ops", [("koala-bear", range(300, 340), 300), ("baby-bear", range(400, 410), 300),
                                               ("koala-bear", range(500, 506), 3000)])
def test_device_preparation_on_random_circuits(field, seeds, n_ops):
    import plonky3_recursion_amd as p3r
    ctx = p3r.Context(field=field, **FRI)
    P = oracle_lib.MODULUS[field]
    for k, seed in enumerate(seeds):
        tp = p3r.TablePacking(public_lanes=1 + k % 3, alu_lanes=1 + k % 4, horner_packed_steps=2 + k % 4,
                              recompose_lanes=1 + k % 2).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
        c, i = circuit_fuzz.random_circuit(seed, n_ops=n_ops, modulus=P)
        circuit = p3r.Circuit(c.witness_count, c.ops, c.ext, c.public_rows, c.private_rows, c.rewrite.reshape(-1, 2))
        inputs = p3r.CircuitInputs(i.public_values.reshape(-1, 4), i.private_values.reshape(-1, 4), i.pd_op_ids,
                                   i.pd_siblings.reshape(-1, 8))
        # random circuits need not satisfy the AIRs: compare everything up to the traces, not proofs
        import plonky3_recursion_amd.prover as pv
        res = []
        for host in (False, True):
            if host:
                os.environ["P3R_PREP_HOST"] = "1"
            try:
                pc = p3r.PreparedCircuit(ctx, circuit, tp)
            finally:
                os.environ.pop("P3R_PREP_HOST", None)
            assert pc.prepared_on_device == (not host), seed
            r = pc.run(inputs)
            cpd = pc.circuit_prover_data
            res.append((cpd.preprocessed_commitment.copy(), pc.levels, list(cpd.table_heights),
                        {n: r.download(n) for n in ARRAYS if cpd.rows[pv.TRACES_ARRAYS[n][1]]}))
            r.free()
            pc.free()
        dev, host = res
        assert np.array_equal(dev[0], host[0]), (seed, "preprocessed commitment")
        assert dev[1] == host[1] and dev[2] == host[2], (seed, dev[1], host[1], dev[2], host[2])
        for n in host[3]:
            assert np.array_equal(dev[3][n], host[3][n]), (seed, n)
    ctx.close()


W32_ARRAYS = ARRAYS + ("p2w_input_values", "p2w_flags", "p2w_mmcs_index_sum")


@pytest.mark.parametrize("field,log_h,packing,gen,ctx_kw", [
    ("koala-bear", 7, None, {}, {}),
    ("baby-bear", 8, PACKINGS[1], {}, dict(mmcs_arity=4)),
    ("koala-bear", 10, PACKINGS[2], dict(horner_chain_len=300, sponge_chain_len=40, merkle_depth=11), dict(mmcs_arity=4)),
    ("baby-bear", 13, PACKINGS[3], dict(horner_chain_len=64, sponge_chain_len=8, merkle_depth=20), {}),
    ("koala-bear", 16, None, dict(horner_chain_len=64, sponge_chain_len=8, merkle_depth=20), dict(mmcs_arity=4)),
])
def test_device_preparation_of_width32_rows(field, log_h, packing, gen, ctx_kw):
    """Round 6: circuits that hold P3R_OP_POSEIDON2_W32_PERM ops (the MMCS rows of an `--arity4` verifier circuit) are
    prepared on the device too: the 48-column Poseidon2PreprocessedRow<8, 6> rows with their multiplicities
    (executor.rs:770-893, batch_stark_prover.rs:177-243), the op type's own chain state (sponge rows seed the Merkle
    state, executor.rs:462-491) as segments of the static schedule, the NonPrimitiveOpId -> row table for the private
    data of the Merkle rows - same commitment, schedule depth, run traces and proof bytes as the host restatement."""
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    gen = dict(dict(horner_chain_len=16, sponge_chain_len=3, merkle_depth=5), **gen)
    a = harness_lib.generate(field, log_h, seed=177 + log_h, flags=harness_lib.P2_W32_OPS, **gen)
    assert len(a["pdw_op_ids"]) > 0
    ctx = p3r.Context(field=field, allow_unpinned_w32_defaults=True, **FRI, **ctx_kw)
    tp = p3r.TablePacking(**(packing or {})).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    dev, host = both_ways(ctx, wl.circuit_from_arrays(a), tp, wl.circuit_inputs_from_arrays(a), W32_ARRAYS)
    assert "p2w_input_values" in host[2] and len(host[2]["p2w_input_values"])
    check_same(dev, host, (field, log_h, packing))
    ctx.close()


@pytest.mark.parametrize("field,seeds,n_ops", [("koala-bear", range(600, 640), 400), ("baby-bear", range(700, 710), 400),
                                               ("koala-bear", range(800, 804), 3000)])
def test_device_preparation_on_random_circuits_with_width32_rows(field, seeds, n_ops):
    """Sponge and Merkle rows of the width-32 table in any order, rows that wait for witnesses of later levels (a successor
    that opens its own segment, the other successor of the same row joining instead), outputs that land on set witnesses."""
    import plonky3_recursion_amd as p3r
    import plonky3_recursion_amd.prover as pv
    ctx = p3r.Context(field=field, allow_unpinned_w32_defaults=True, **FRI)
    P = oracle_lib.MODULUS[field]
    saw = 0
    for k, seed in enumerate(seeds):
        tp = p3r.TablePacking(public_lanes=1 + k % 3, alu_lanes=1 + k % 4, horner_packed_steps=2 + k % 4,
                              recompose_lanes=1 + k % 2).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
        c, i = circuit_fuzz.random_circuit(seed, n_ops=n_ops, modulus=P, w32=True)
        circuit = p3r.Circuit(c.witness_count, c.ops, c.ext, c.public_rows, c.private_rows, c.rewrite.reshape(-1, 2))
        inputs = p3r.CircuitInputs(i.public_values.reshape(-1, 4), i.private_values.reshape(-1, 4), i.pd_op_ids,
                                   i.pd_siblings.reshape(-1, 8), i.pdw_op_ids, i.pdw_siblings.reshape(-1, 24))
        res = []
        for host in (False, True):
            if host:
                os.environ["P3R_PREP_HOST"] = "1"
            try:
                pc = p3r.PreparedCircuit(ctx, circuit, tp)
            finally:
                os.environ.pop("P3R_PREP_HOST", None)
            assert pc.prepared_on_device == (not host), seed
            r = pc.run(inputs)
            cpd = pc.circuit_prover_data
            res.append((cpd.preprocessed_commitment.copy(), pc.levels, list(cpd.table_heights),
                        {n: r.download(n) for n in W32_ARRAYS if cpd.rows[pv.TRACES_ARRAYS[n][1]]}))
            r.free()
            pc.free()
        dev, host = res
        saw += int("p2w_input_values" in host[3])
        assert np.array_equal(dev[0], host[0]), (seed, "preprocessed commitment")
        assert dev[1] == host[1] and dev[2] == host[2], (seed, dev[1], host[1], dev[2], host[2])
        for n in host[3]:
            assert np.array_equal(dev[3][n], host[3][n]), (seed, n)
    assert saw > len(seeds) // 2
    ctx.close()


def test_flagged_circuits_get_the_host_paths_error():
    """What the device pass cannot describe it hands over: the error text is the host restatement's."""
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    ctx = p3r.Context(field=field, **FRI)
    tp = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    a = harness_lib.generate(field, 7, seed=5, horner_chain_len=16, sponge_chain_len=3, merkle_depth=5)
    good = wl.circuit_from_arrays(a)

    def broken(edit):
        import copy
        c = copy.deepcopy(good)
        c.ops = np.array(c.ops, copy=True).reshape(-1, 8)
        c.ext = np.array(c.ext, copy=True)
        edit(c)
        return c

    def messages(c):
        got = []
        for host in (False, True):
            if host:
                os.environ["P3R_PREP_HOST"] = "1"
            try:
                with pytest.raises(p3r.P3rError) as e:
                    pc = p3r.PreparedCircuit(ctx, c, tp)
                    pc.run(wl.circuit_inputs_from_arrays(a))   # deferred errors are reported by run, as the reference does
                got.append(str(e.value))
            finally:
                os.environ.pop("P3R_PREP_HOST", None)
        assert got[0] == got[1], got
        return got[0]

    ops = np.asarray(good.ops).reshape(-1, 8)
    k_alu = int(np.nonzero(ops[:, 0] == p3r.prover.OP_ALU_ADD)[0][5])
    k_const = int(np.nonzero(ops[:, 0] == p3r.prover.OP_CONST)[0][2])
    assert "out of bounds" in messages(broken(lambda c: c.ops.__setitem__((k_alu, 1), good.witness_count + 7)))
    assert "not canonical" in messages(broken(lambda c: c.ext.__setitem__(int(ops[k_const, 6]), 0x7F000001)))
    assert "kind" in messages(broken(lambda c: c.ops.__setitem__((k_alu, 0), 99)))
    # a private input nobody claims
    c = broken(lambda c: None)
    c.private_input_rows = np.concatenate([np.asarray(c.private_input_rows, np.uint32), [np.uint32(good.witness_count)]])
    c.witness_count = good.witness_count + 1
    assert "UnclaimedPrivateInput" in messages(c)
    ctx.close()


def test_bad_packing_is_refused_before_either_preparation(monkeypatch):
    """Lane counts of 0 and a Horner pack size outside 2..8 reach the C ABI unvalidated when a caller skips
    TablePacking::validate: both preparations divide by them, so p3r_circuit_create checks first (P3R_EINVAL,
    not a SIGFPE) - on the device path and on the host path."""
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    ctx = p3r.Context(field=field, **FRI)
    a = harness_lib.generate(field, 7, seed=5, horner_chain_len=16, sponge_chain_len=3, merkle_depth=5)
    circuit = wl.circuit_from_arrays(a)
    monkeypatch.setattr(p3r.TablePacking, "validate", lambda self: None)
    bad = [dict(public_lanes=0), dict(alu_lanes=0), dict(recompose_lanes=0), dict(horner_packed_steps=0),
           dict(horner_packed_steps=1), dict(horner_packed_steps=9)]
    for host in (False, True):
        if host:
            monkeypatch.setenv("P3R_PREP_HOST", "1")
        for kw in bad:
            tp = p3r.TablePacking(min_trace_height=8)
            for k, v in kw.items():
                setattr(tp, k, v)
            with pytest.raises(p3r.P3rError, match="lane counts must be positive|horner_packed_steps must be in 2..8") as e:
                p3r.PreparedCircuit(ctx, circuit, tp)
            assert e.value.code == -1   # P3R_EINVAL
    ctx.close()
