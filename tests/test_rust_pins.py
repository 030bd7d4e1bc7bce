"""Parity pins against the REAL reference (upstream p3-* 0.6 + Plonky3-recursion's circuit-prover).

The vectors are produced by tools/rust_pin (a source-only Rust crate: this repo's build image has no
cargo) on a machine that has the toolchain; until someone has run it the files are absent and every
test here is skipped.  Once present they turn the "[EXT]" choices of DESIGN.md section 4 from
recollection into checked facts:

  rust_primitives.json            upstream round constants, Poseidon2 / sponge / compression KATs, the
                                  DuplexChallenger transcript, extension-field arithmetic, coset LDE
  rust_fibonacci_layer_<f>.json   prove_all_tables bytes of the Fibonacci circuit over the extension

A failing test names the choice to flip (tools/rust_pin/README.md lists where each one lives)."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FIELDS = [("koala-bear", "koala_bear"), ("baby-bear", "baby_bear")]


def load(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} absent: run tools/rust_pin on a machine with cargo (tools/rust_pin/README.md)")
    with open(path) as fh:
        return json.load(fh)


@pytest.mark.parametrize("field,key", FIELDS)
def test_default_round_constants_are_upstream(golden, field, key):
    """poseidon2_rc_default.inc / tests/golden/poseidon2_rc_default.json vs the upstream statics
    (*_POSEIDON2_RC_16_*).  On failure: regenerate the default table from rust_primitives.json["rc"]."""
    rust = load("rust_primitives.json")["fields"][key]
    assert golden["rc"][key] == rust["rc"]


@pytest.mark.parametrize("field,key", FIELDS)
def test_oracle_primitives_match_upstream(oracle, field, key):
    g = load("rust_primitives.json")["fields"][key]
    rc = np.array(g["rc"], dtype=np.uint32)
    ins = np.array([k["in"] for k in g["permute"]], dtype=np.uint32)
    outs = np.array([k["out"] for k in g["permute"]], dtype=np.uint32)
    assert np.array_equal(oracle.permute(field, ins, rc=rc), outs), "Poseidon2 linear layers / round structure"
    for kat in g["sponge"]:
        cap, _ = oracle.commit(field, [np.array([kat["in"]], dtype=np.uint32)], rc=rc)
        assert cap[0].tolist() == kat["out"], "PaddingFreeSponge (overwrite mode, ragged last block)"
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pyref
    f = pyref.FIELDS[key]
    c = g["compress"]
    assert pyref.compress(c["left"], c["right"], g["rc"], f) == c["out"], "TruncatedPermutation"
    ch = g["challenger"]
    assert oracle.challenger_script(field, ch["ops"], ch["args"], rc=rc).tolist() == ch["out"], "DuplexChallenger"
    mul, inv = oracle.ext_ops(field, g["ext"]["a"], g["ext"]["b"])
    assert mul.tolist() == g["ext"]["mul"] and inv.tolist() == g["ext"]["inv_a"], "binomial extension x^4 = W"
    for bits, gen in g["two_adic_generators"].items():
        assert pyref.two_adic_generator(int(bits), f) == gen, "two-adic generator table"
    lde = g["lde"]
    out = oracle.coset_lde(field, np.array(lde["evals"], dtype=np.uint32), lde["added_bits"], lde["shift"])
    assert np.array_equal(out, np.array(lde["lde"], dtype=np.uint32)), "coset_lde_batch + bit_reverse_rows"


def _fib_layer(oracle, field, key):
    import fib_lib
    import circuit_lib as cl
    import layer_lib
    import oracle_lib
    g = load(f"rust_fibonacci_layer_{key}.json")
    rc = np.array(g["rc"], dtype=np.uint32)
    circuit, inputs, fib = fib_lib.fibonacci_circuit(g["n"], oracle_lib.MODULUS[field])
    assert fib == g["fib"]
    oc = cl.OracleCircuit(oracle, circuit).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, inputs)
    prm = layer_lib.params(**g["fri"])
    packing = dict(g["packing"])
    packing.setdefault("horner_packed_steps", 2)      # TablePacking::new(1, 1) keeps K = 2 (packing.rs:36-47)
    L = layer_lib.OracleLayer(oracle, field, oc.workload_arrays(), prm, packing=packing, rc=rc)
    return g, rc, circuit, inputs, prm, L


@pytest.mark.parametrize("field,key", FIELDS)
def test_rust_proof_is_accepted_and_reproduced(oracle, field, key):
    """The reference's own proof of the Fibonacci layer: the native verifier accepts its bytes, the
    BatchStarkProof wire format round-trips, and the oracle prover reproduces the inner proof bit for
    bit (transcript order, LogUp packing, FRI arity schedule, proof-of-work witness, postcard field
    order - a mismatch localises to one of these, see tools/rust_pin/README.md)."""
    import plonky3_recursion_amd as p3r
    g, rc, circuit, inputs, prm, L = _fib_layer(oracle, field, key)
    inner = bytes.fromhex(g["batch_proof_postcard_hex"])
    outer = bytes.fromhex(g["batch_stark_proof_postcard_hex"])
    tables = L.tables()
    assert [int(t["main"].shape[0]).bit_length() - 1 for t in tables] == g["degree_bits"], "table heights / min_trace_height"
    proof = p3r.BatchStarkProof.from_postcard(outer, field)
    assert proof.proof == inner and proof.to_postcard() == outer, "BatchStarkProof postcard layout"
    assert np.array_equal(proof.preprocessed_commitment, L.prep_commit()), "preprocessed columns / commitment"
    cfg, keep = p3r.make_config(field, poseidon2_rc=rc, **g["fri"])
    p3r.verify_all_tables(cfg, proof)
    L.verify(inner)
    assert L.prove() == inner, "prove_batch bytes"


@pytest.mark.gpu
@pytest.mark.parametrize("field,key", FIELDS)
def test_hip_prover_reproduces_the_rust_proof(oracle, field, key):
    import plonky3_recursion_amd as p3r
    g, rc, circuit, inputs, prm, L = _fib_layer(oracle, field, key)
    ctx = p3r.Context(field=field, poseidon2_rc=rc, **g["fri"])
    tp = p3r.TablePacking(public_lanes=1, alu_lanes=1, horner_packed_steps=g["packing"].get("horner_packed_steps", 2))
    tp.with_fri_params(g["fri"]["log_final_poly_len"], g["fri"]["log_blowup"])
    pc = p3r.PreparedCircuit(ctx, p3r.Circuit(circuit.witness_count, circuit.ops, circuit.ext, circuit.public_rows), tp)
    got = pc.prove(p3r.CircuitInputs(public_values=inputs.public_values.reshape(-1, 4)))
    assert got == bytes.fromhex(g["batch_proof_postcard_hex"])
    pc.free()
    ctx.close()


# ---- the tables the Fibonacci circuit does not reach (Poseidon2, Recompose, Mul / MulAdd, packed HornerAcc) --------
TABLE_OF = {"const": 0, "public": 1, "alu": 2, "poseidon2": 3, "recompose": 4}


def _npo_layer(oracle, field, key):
    import circuit_lib as cl
    import layer_lib
    import oracle_lib
    g = load(f"rust_npo_layer_{key}.json")
    rc = np.array(g["rc"], dtype=np.uint32)
    c = g["circuit"]
    circuit = cl.Circuit(c["witness_count"], np.array(c["ops"], dtype=np.uint32), c["ext"], c["public_rows"], c["private_rows"],
                         c["rewrite"])
    pd = g["inputs"]["private_data"]
    inputs = cl.Inputs(np.array(g["inputs"]["public_values"], dtype=np.uint32).reshape(-1), (),
                       [d["op_id"] for d in pd], np.array([d["sibling"] for d in pd], dtype=np.uint32).reshape(-1))
    oc = cl.OracleCircuit(oracle, circuit).preprocess(oracle_lib.MODULUS[field])
    return g, rc, circuit, inputs, oc, layer_lib.params(**g["fri"])


@pytest.mark.parametrize("field,key", FIELDS)
def test_rust_npo_layer_localises_every_table(oracle, field, key):
    """Stage by stage against the reference's own run of the NPO circuit, so that a mismatch names one choice
    (tools/rust_pin/README.md has the table): preprocessed columns -> runner -> per-table main traces ->
    preprocessed commitment -> native verifier on the Rust proof -> proof bytes."""
    import layer_lib
    import plonky3_recursion_amd as p3r
    g, rc, circuit, inputs, oc, prm = _npo_layer(oracle, field, key)
    want = g["preprocessed_columns"]
    # get_airs_and_degrees_with_prep: [const, public, alu] primitive columns, then the NPO maps
    prim = want["primitive"]
    assert oc.get("const_prep").tolist() == prim[0], "Const preprocessed columns (ext_mult, D*idx)"
    assert oc.get("public_prep").tolist() == prim[1], "Public preprocessed columns"
    assert oc.get("alu_prep13").tolist() == prim[2], "ALU preprocessed columns: bus roles and signed multiplicities"
    oc.run(field, inputs)
    got = oc.workload_arrays()
    assert got["alu_values"].reshape(-1, 16).tolist() == g["alu_trace_values"], "CircuitRunner: AluOpRecords"
    L = layer_lib.OracleLayer(oracle, field, got, prm, packing=dict(g["packing"]), rc=rc)
    tables = L.tables()
    mains = {m["table"]: m for m in g["main_traces"]}
    for t in tables:
        name = next((n for n in mains if n.lower().startswith(t["kind"])), None)   # "poseidon2_perm/..." for the permutation table
        assert name is not None, (t["kind"], list(mains))
        ref = np.array(mains[name]["values"], dtype=np.uint32).reshape(-1, mains[name]["width"])
        assert t["main"].shape == ref.shape, (t["kind"], "main trace shape", t["main"].shape, ref.shape)
        assert np.array_equal(t["main"], ref), (t["kind"], "main trace: column order (Poseidon2Cols interior order for the "
                                                "permutation table, lane schedule and packed-Horner columns for the ALU)")
    assert [int(t["main"].shape[0]).bit_length() - 1 for t in tables] == g["degree_bits"], "table heights"
    inner = bytes.fromhex(g["batch_proof_postcard_hex"])
    outer = bytes.fromhex(g["batch_stark_proof_postcard_hex"])
    proof = p3r.BatchStarkProof.from_postcard(outer, field)
    assert proof.proof == inner and proof.to_postcard() == outer, "BatchStarkProof postcard layout (npo_lanes, non_primitives)"
    assert np.array_equal(proof.preprocessed_commitment, L.prep_commit()), \
        "preprocessed commitment: Poseidon2 24-column layout / scheduled ALU preprocessed trace"
    cfg, keep = p3r.make_config(field, poseidon2_rc=rc, **g["fri"])
    p3r.verify_all_tables(cfg, proof)     # constraint order of the inner permutation AIR, CTL + packed-Horner lookups, LogUp packing
    L.verify(inner)
    assert L.prove() == inner, "prove_batch bytes"


@pytest.mark.gpu
@pytest.mark.parametrize("field,key", FIELDS)
def test_hip_prover_reproduces_the_rust_npo_proof(oracle, field, key):
    import plonky3_recursion_amd as p3r
    g, rc, circuit, inputs, oc, prm = _npo_layer(oracle, field, key)
    ctx = p3r.Context(field=field, poseidon2_rc=rc, **g["fri"])
    tp = p3r.TablePacking(**g["packing"]).with_fri_params(g["fri"]["log_final_poly_len"], g["fri"]["log_blowup"])
    pc = p3r.PreparedCircuit(ctx, p3r.Circuit(circuit.witness_count, circuit.ops, circuit.ext, circuit.public_rows), tp)
    got = pc.prove(p3r.CircuitInputs(public_values=inputs.public_values.reshape(-1, 4), private_data_op_ids=inputs.pd_op_ids,
                                     private_data_siblings=inputs.pd_siblings.reshape(-1, 8)))
    assert got == bytes.fromhex(g["batch_proof_postcard_hex"])
    pc.free()
    ctx.close()


# ---- D = 1: the example's base proof (CircuitBuilder<F>) --------------------------------------------------------------
def _base_layer(oracle, field, key):
    import layer_lib
    g = load(f"rust_fibonacci_base_layer_{key}.json")
    rc = np.array(g["rc"], dtype=np.uint32)
    prep = [np.array(c, dtype=np.uint32) for c in g["preprocessed_columns"]]      # [Const, Public, Alu] (PrimitiveOpType order)
    w = dict(const_values=np.array(g["const_values"], np.uint32), const_prep=prep[0],
             public_values=np.array(g["public_values"], np.uint32), public_prep=prep[1],
             alu_values=np.array(g["alu_values"], np.uint32), alu_prep13=prep[2])
    for name in ("p2_inputs", "p2_flags", "p2_mmcs_index_sum", "p2_in_ctl", "p2_input_indices", "p2_out_ctl",
                 "p2_output_indices", "p2_mmcs_index_sum_idx", "recompose_values", "recompose_prep"):
        w[name] = np.zeros(0, np.uint32)
    w["counts"] = np.array([len(w["const_values"]), len(w["public_values"]), len(w["alu_values"]) // 4, 0, 0, 0], np.uint32)
    prm = layer_lib.params(**g["fri"])
    L = layer_lib.OracleLayer(oracle, field, w, prm, packing=dict(g["packing"], ext_degree=1), rc=rc)
    return g, rc, w, prm, L


@pytest.mark.parametrize("field,key", FIELDS)
def test_rust_base_field_proof_is_accepted_and_reproduced(oracle, field, key):
    """`prove_all_tables` over D = 1 traces (recursive_fibonacci.rs:315-337): the reference's preprocessed columns and
    Traces go into the oracle as they are; bus tuples (idx, v), prefix alpha + beta^2."""
    import plonky3_recursion_amd as p3r
    g, rc, w, prm, L = _base_layer(oracle, field, key)
    inner, outer = bytes.fromhex(g["batch_proof_postcard_hex"]), bytes.fromhex(g["batch_stark_proof_postcard_hex"])
    proof = p3r.BatchStarkProof.from_postcard(outer, field)
    assert proof.ext_degree == 1 and proof.w_binomial is None and proof.proof == inner and proof.to_postcard() == outer
    assert np.array_equal(proof.preprocessed_commitment, L.prep_commit()), "D = 1 table layout / preprocessed commitment"
    cfg, keep = p3r.make_config(field, poseidon2_rc=rc, ext_degree=1, **g["fri"])
    p3r.verify_all_tables(cfg, proof)
    L.verify(inner)
    assert L.prove() == inner, "prove_batch bytes (D = 1)"


@pytest.mark.gpu
@pytest.mark.parametrize("field,key", FIELDS)
def test_hip_prover_reproduces_the_rust_base_field_proof(oracle, field, key):
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    g, rc, w, prm, L = _base_layer(oracle, field, key)
    ctx = p3r.Context(field=field, poseidon2_rc=rc, ext_degree=1, **g["fri"])
    tp = pv.TablePacking(**g["packing"]).with_fri_params(g["fri"]["log_final_poly_len"], g["fri"]["log_blowup"])
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(w, ext_degree=1), pv.FriRecursionBackend(),
                                     pv.ProveNextLayerParams(table_packing=tp))
    got = cache.prover.prove_all_tables(wl.traces_from_arrays(w, ext_degree=1), cache.circuit_prover_data)
    assert got.proof == bytes.fromhex(g["batch_proof_postcard_hex"])
    assert got.to_postcard() == bytes.fromhex(g["batch_stark_proof_postcard_hex"])
    cache.circuit_prover_data.free()
    ctx.close()


# ---- D = 5: quintic ALU + the compact-D1 Poseidon2 table ---------------------------------------------------------------
QUINTIC_FIXTURES = ["rust_quintic_layer_koala_bear.json", "rust_quintic_challenge_layer_koala_bear.json"]


def _quintic_layer(oracle, g=None, name=QUINTIC_FIXTURES[0]):
    import layer_lib
    g = g or load(name)
    rc = np.array(g["rc"], dtype=np.uint32)
    prim = [np.array(c, dtype=np.uint32) for c in g["preprocessed_columns"]["primitive"]]
    p2 = np.array(g["preprocessed_columns"]["non_primitive"]["poseidon2_perm/koala_bear_d1_w16"], np.uint32).reshape(-1, 62)
    main_p2 = next(np.array(m["values"], np.uint32).reshape(-1, m["width"]) for m in g["main_traces"]
                   if m["table"].startswith("poseidon2_perm/"))
    n = len(p2)
    # the committed 62-column rows back to p3r_layer_desc's fields (include/p3r.h; indices are stored scaled by 5)
    in_ctl = np.zeros((n, 16), np.uint32)
    in_ctl[:, :8] = p2[:, 0:8]
    w = dict(const_values=np.array(g["const_values"], np.uint32), const_prep=prim[0],
             public_values=np.array(g["public_values"], np.uint32), public_prep=prim[1],
             alu_values=np.array(g["alu_values"], np.uint32), alu_prep13=prim[2],
             p2_inputs=main_p2[:n, :16].reshape(-1), p2_in_ctl=in_ctl.reshape(-1), p2_absorb_len=p2[:, 8].copy(),
             p2_input_indices=(p2[:, 26:42] // 5).reshape(-1), p2_output_indices=(p2[:, 42:50] // 5).reshape(-1),
             p2_out_ctl=p2[:, 50:58].reshape(-1), p2_mmcs_index_sum_idx=p2[:, 58] // 5,
             p2_flags=np.stack([p2[:, 60], p2[:, 61], main_p2[:n, -2], p2[:, 59]], axis=1).reshape(-1),
             p2_mmcs_index_sum=main_p2[:n, -1].copy(),
             recompose_values=np.zeros(0, np.uint32), recompose_prep=np.zeros(0, np.uint32))
    w["counts"] = np.array([len(w["const_values"]) // 5, len(w["public_values"]) // 5, len(w["alu_values"]) // 20, n, 0, 0], np.uint32)
    return g, rc, w


@pytest.mark.parametrize("name", QUINTIC_FIXTURES)
def test_rust_quintic_layer_tables_and_proof(oracle, name):
    """The D = 5 tables against the reference: per-table main traces (quintic ALU incl. the packed-Horner columns,
    the 166-column permutation rows), the preprocessed commitment (62-column compact-D1 rows), acceptance, bytes -
    under the ordinary configuration and under koala_bear_quintic_params (challenge_degree = 5)."""
    import layer_lib
    import plonky3_recursion_amd as p3r
    g, rc, w = _quintic_layer(oracle, name=name)
    pk = g["packing"]
    fri = g["fri"]
    dc = int(g.get("challenge_degree", 4))
    prm = layer_lib.params(challenge_degree=dc, **fri)
    L = layer_lib.OracleLayer(oracle, "koala-bear", w, prm, rc=rc,
                              packing=dict(public_lanes=pk["public_lanes"], alu_lanes=pk["alu_lanes"],
                                           horner_packed_steps=pk["horner_packed_steps"], min_trace_height=pk["min_trace_height"],
                                           ext_degree=5))
    tables = {t["kind"]: t for t in L.tables()}
    for m in g["main_traces"]:
        kind = "poseidon2" if m["table"].startswith("poseidon2_perm/") else m["table"].lower()
        want = np.array(m["values"], np.uint32).reshape(-1, m["width"])
        assert np.array_equal(tables[kind]["main"], want), f"main trace of {m['table']}"
    outer = bytes.fromhex(g["batch_stark_proof_postcard_hex"])
    proof = p3r.BatchStarkProof.from_postcard(outer, "koala-bear", challenge_degree=dc)
    assert proof.ext_degree == 5 and proof.w_binomial is None and proof.alu_quintic_trinomial
    assert [e.op_type for e in proof.non_primitives] == ["poseidon2_perm/koala_bear_d1_w16"]
    assert np.array_equal(proof.preprocessed_commitment, L.prep_commit()), "compact-D1 preprocessed rows / commitment"
    cfg, keep = p3r.make_config("koala-bear", poseidon2_rc=rc, ext_degree=5, challenge_degree=dc, **fri)
    p3r.verify_all_tables(cfg, proof)
    assert L.prove() == proof.proof, "prove_batch bytes (D = 5)"


@pytest.mark.gpu
@pytest.mark.parametrize("name", QUINTIC_FIXTURES)
def test_hip_prover_reproduces_the_rust_quintic_proof(oracle, name):
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd import prover as pv
    import harness_adapters as wl
    g, rc, w = _quintic_layer(oracle, name=name)
    pk = g["packing"]
    fri = g["fri"]
    ctx = p3r.Context(field="koala-bear", poseidon2_rc=rc, ext_degree=5, challenge_degree=int(g.get("challenge_degree", 4)), **fri)
    tp = pv.TablePacking(public_lanes=pk["public_lanes"], alu_lanes=pk["alu_lanes"], horner_packed_steps=pk["horner_packed_steps"],
                         min_trace_height=pk["min_trace_height"])
    cache = pv.build_next_layer_prep(ctx, wl.circuit_prep_from_arrays(w, ext_degree=5), pv.FriRecursionBackend(),
                                     pv.ProveNextLayerParams(table_packing=tp))
    got = cache.prover.prove_all_tables(wl.traces_from_arrays(w, ext_degree=5), cache.circuit_prover_data)
    assert got.to_postcard() == bytes.fromhex(g["batch_stark_proof_postcard_hex"])
    cache.circuit_prover_data.free()
    ctx.close()


def test_quintic_fixture_mapping_inverts_the_table_builder(oracle):
    """No Rust file needed: a fixture of the same schema made from this repo's own D = 5 layer (the oracle's committed
    62-column rows and main traces) maps back to p3r_layer_desc fields that rebuild the same tables and commitment -
    so that, when the real fixture arrives, a failure of test_rust_quintic_layer_tables_and_proof is about the
    tables, not about the test's own unpacking."""
    import harness_lib
    import layer_lib
    import oracle_lib
    arrs = harness_lib.generate("koala-bear", 6, seed=77, flags=harness_lib.NO_RECOMPOSE, ext_degree=5, horner_chain_len=12,
                                sponge_chain_len=3, merkle_depth=4)
    prm = layer_lib.params(log_blowup=2, max_log_arity=2, log_final_poly_len=2, query_pow_bits=6, num_queries=8)
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=dict(ext_degree=5))
    tables = {t["kind"]: t for t in L.tables()}
    n = int(arrs["counts"][3])
    g = dict(rc=oracle_lib.default_rc("koala-bear").tolist(),
             preprocessed_columns=dict(primitive=[arrs["const_prep"].tolist(), arrs["public_prep"].tolist(), arrs["alu_prep13"].tolist()],
                                       non_primitive={"poseidon2_perm/koala_bear_d1_w16": tables["poseidon2"]["prep"][:n].reshape(-1).tolist()}),
             const_values=arrs["const_values"].tolist(), public_values=arrs["public_values"].tolist(),
             alu_values=arrs["alu_values"].tolist(),
             main_traces=[dict(table="poseidon2_perm/koala_bear_d1_w16", width=int(tables["poseidon2"]["main"].shape[1]),
                               values=tables["poseidon2"]["main"].reshape(-1).tolist())])
    _, rc, w = _quintic_layer(oracle, g)
    L2 = layer_lib.OracleLayer(oracle, "koala-bear", w, prm, packing=dict(ext_degree=5), rc=rc)
    for a, b in zip(L.tables(), L2.tables()):
        assert a["kind"] == b["kind"] and np.array_equal(a["main"], b["main"]) and np.array_equal(a["prep"], b["prep"]), a["kind"]
    assert np.array_equal(L.prep_commit(), L2.prep_commit())


# ---- the width-32 Poseidon2 table of the arity-4 MMCS (circuit-prover/tests/arity4_mmcs.rs as a fixture) ----
def _arity4_fixture():
    g = load("rust_arity4_layer_koala_bear.json")
    return g, np.array(g["w32_rc"], np.uint32), np.array(g["w32_diag"], np.uint32)


def test_rust_arity4_layer_constants_and_table(oracle):
    """With the DUMPED constants (round constants and internal diagonal of Poseidon2KoalaBear<32>): the harness permutation
    reproduces upstream's KATs; the oracle's width-32 trace rows equal the Rust prover's main trace of the table; the
    native verifier accepts the Rust proof.  A failure of the first assertion means the external layer (M4 blocks) or the
    round structure differs; of the second, the Poseidon2Cols<32> interior order or the arity-4 circuit columns."""
    import ctypes as C
    import layer_lib
    import plonky3_recursion_amd as p3r
    g, rc, diag = _arity4_fixture()
    assert len(rc) == 8 * 32 + 31 and len(diag) == 32
    lib = oracle.lib
    u32p = C.POINTER(C.c_uint32)
    lib.orc_p2w_permute.argtypes = [C.c_int, u32p, u32p, u32p, u32p, C.c_size_t]
    for kat in g["perm32_kats"]:
        x = np.array(kat["in"], np.uint32)
        y = np.empty(32, np.uint32)
        oracle._ck(lib.orc_p2w_permute(0, rc.ctypes.data_as(u32p), diag.ctypes.data_as(u32p), x.ctypes.data_as(u32p), y.ctypes.data_as(u32p), 1))
        assert y.tolist() == kat["out"], "Poseidon2KoalaBear<32>: external layer / round structure"
    rows = g["p2w_rows"]
    n = len(rows)
    main = next(m for m in g["main_traces"] if m["table"].endswith("_d4_w32"))
    want = np.array(main["values"], np.uint32).reshape(-1, main["width"])
    assert main["width"] == 32 + 8 * 32 + 31 + 4
    lib.orc_p2w_trace_rows.argtypes = [C.c_int, u32p, u32p, C.c_size_t, u32p, u32p, u32p, u32p]
    h = want.shape[0]
    inputs = np.zeros((h, 32), np.uint32)
    flags = np.zeros((h, 4), np.uint32)
    flags[:, 0] = 1
    sums = np.zeros(h, np.uint32)
    for r, row in enumerate(rows):
        inputs[r] = row["input_values"]
        flags[r] = [row["new_start"], row["merkle_path"], row["mmcs_bit"], row["mmcs_bit2"]]
        sums[r] = row["mmcs_index_sum"]
    got = np.empty_like(want)
    oracle._ck(lib.orc_p2w_trace_rows(0, rc.ctypes.data_as(u32p), diag.ctypes.data_as(u32p), h, inputs.ctypes.data_as(u32p),
                                      flags.ctypes.data_as(u32p), sums.ctypes.data_as(u32p), got.ctypes.data_as(u32p)))
    assert np.array_equal(got[:n], want[:n]), "Poseidon2Cols<32> interior order / arity-4 circuit columns"
    # the Rust proof under the native verifier (the table's constants passed as data)
    proof = bytes.fromhex(g["batch_stark_proof_postcard_hex"])
    bsp = p3r.BatchStarkProof.from_postcard(proof, "koala-bear")
    assert any(e.op_type == "poseidon2_perm/koala_bear_d4_w32" for e in bsp.non_primitives)
    cfg, keep = p3r.make_config("koala-bear", poseidon2_rc=np.array(g["rc"], np.uint32), poseidon2_w32_rc=rc, poseidon2_w32_diag=diag,
                                **layer_lib_fri_of_config_koala_bear())
    p3r.verify_all_tables(cfg, bsp)


def layer_lib_fri_of_config_koala_bear():
    """FRI parameters of circuit-prover's `config::koala_bear()` (config.rs:42-84, :126-136): what arity4_mmcs.rs proves under."""
    return dict(log_blowup=1, max_log_arity=1, cap_height=0, log_final_poly_len=0, commit_pow_bits=0, query_pow_bits=16,
                num_queries=100)


@pytest.mark.gpu
def test_hip_reproduces_the_rust_arity4_table(oracle):
    """The device's trace fill of the width-32 table on the Rust run's rows, with the dumped constants."""
    import plonky3_recursion_amd as p3r
    g, rc, diag = _arity4_fixture()
    main = next(m for m in g["main_traces"] if m["table"].endswith("_d4_w32"))
    want = np.array(main["values"], np.uint32).reshape(-1, main["width"])
    rows = g["p2w_rows"]
    ctx = p3r.Context(field="koala-bear", poseidon2_rc=np.array(g["rc"], np.uint32), poseidon2_w32_rc=rc, poseidon2_w32_diag=diag)
    got = ctx.generate_w32_trace_rows(np.array([r["input_values"] for r in rows], np.uint32),
                                      *[np.array([r[k] for r in rows], np.uint8) for k in ("new_start", "merkle_path", "mmcs_bit", "mmcs_bit2")],
                                      np.array([r["mmcs_index_sum"] for r in rows], np.uint32), height=want.shape[0])
    assert np.array_equal(got[:len(rows)], want[:len(rows)])
    ctx.close()


# ---- the NATIVE arity-4 MMCS (tools/rust_pin: arity4_mmcs()) - pins p3r_config.mmcs_arity = 4
def _arity4_mmcs_cases():
    g = load("rust_arity4_mmcs_koala_bear.json")
    _, rc, diag = _arity4_fixture()       # the width-32 constants come from the layer fixture of the same run
    from test_mmcs_reference_shapes import iota_matrix, mixed_height_matrices
    mats = {"single_height": [iota_matrix(1024, 4)], "wide_leaf_multi_chunk": [iota_matrix(1024, 40)],
            "odd_log2_height": [iota_matrix(512, 4)], "mixed_heights_with_injection": mixed_height_matrices()}
    for name, case in g.items():
        assert [list(m.shape) for m in mats[name]] == case["dims"], name
        yield name, mats[name], case, (rc, diag)


def test_rust_arity4_mmcs_oracle_tree(oracle):
    """The oracle's arity-4 tree (oracle/hash.hpp::commit4) under upstream's width-32 constants against the native
    MerkleTreeMmcs<.., 4, 8>: same root, same sibling list at every dumped index.  A root mismatch on `single_height`
    alone means the leaf sponge or the 4-to-1 compression; on `odd_log2_height` only, the content of the padded
    positions of a 2-node layer (this repo: zero digests) or a step-2 top level; on `mixed_heights_with_injection`
    only, the bridge / injection rule."""
    for name, mats, case, w32 in _arity4_mmcs_cases():
        cap, tree = oracle.commit4("koala-bear", mats, w32=w32)
        assert cap[0].tolist() == case["root"], name
        for o in case["openings"]:
            opened, proof = tree.open(o["index"])
            assert opened.tolist() == [v for row in o["opened_values"] for v in row], name
            assert proof.tolist() == o["opening_proof"], (name, o["index"])


@pytest.mark.gpu
def test_hip_reproduces_the_rust_arity4_mmcs(oracle):
    import plonky3_recursion_amd as p3r
    for name, mats, case, (rc, diag) in _arity4_mmcs_cases():
        ctx = p3r.Context(field="koala-bear", mmcs_arity=4, poseidon2_w32_rc=rc, poseidon2_w32_diag=diag)
        cap, tree = ctx.commit(mats)
        assert cap[0].tolist() == case["root"], name
        for o in case["openings"]:
            opened, proof = tree.open_batch(o["index"])
            assert proof.tolist() == o["opening_proof"], (name, o["index"])
            p3r.mmcs_verify(ctx.cfg, cap, [m.shape for m in mats], o["index"], opened, proof)
        tree.free()
        ctx.close()


# ---- ZK (HidingFriPcs): acceptance in both directions - a randomised proof has no byte parity to pin ------------------
@pytest.mark.parametrize("field,key", FIELDS)
def test_rust_zk_proof_is_accepted(oracle, field, key):
    """A NATIVE ZK proof (the reference's prove_all_tables under create_config_zk, recursion/examples/common/mod.rs:511-553)
    fed to this repo's verifiers: p3r_verify_batch under p3r_config.zk = 1 and the oracle's verify_batch must accept it,
    the BatchStarkProof wire format must round-trip under P3R_PROOF_ZK, and the non-ZK configuration must refuse it.
    A rejection names the verifier rule that differs from upstream's (csrc/verify_impl.h cites each one)."""
    import fib_lib
    import circuit_lib as cl
    import layer_lib
    import oracle_lib
    import plonky3_recursion_amd as p3r
    g = load(f"rust_fibonacci_zk_layer_{key}.json")
    rc = np.array(g["rc"], dtype=np.uint32)
    inner, outer = bytes.fromhex(g["batch_proof_postcard_hex"]), bytes.fromhex(g["batch_stark_proof_postcard_hex"])
    zk = dict(zk=1, num_random_codewords=g["zk"]["num_random_codewords"])
    proof = p3r.BatchStarkProof.from_postcard(outer, field, zk=True)
    assert proof.proof == inner and proof.to_postcard() == outer, "BatchStarkProof postcard layout of a hiding PCS's proof"
    assert list(proof.degree_bits) == g["degree_bits"], "extended degree bits in the preprocessed metadata"
    cfg, keep = p3r.make_config(field, poseidon2_rc=rc, **g["fri"], **zk)
    p3r.verify_all_tables(cfg, proof)
    cfg0, keep0 = p3r.make_config(field, poseidon2_rc=rc, **g["fri"])
    with pytest.raises(p3r.P3rError):
        p3r.verify_all_tables(cfg0, p3r.BatchStarkProof.from_postcard(outer, field, zk=True))
    # the oracle's verifier, against the statement alone (commitment and AIRs from the proof's own metadata)
    prm = layer_lib.params(**g["fri"], **zk)
    layer_lib.oracle_verify_statement(oracle, field, prm, proof.airs(), proof.preprocessed_commitment, inner, rc=rc)
    # how far the un-pinned prover-side choices are from upstream's, as information (no assertion): does upstream pad the
    # preprocessed round with zeros (commitment equality), are its quotient masks of the same shape (lengths only)
    circuit, inputs, fib = fib_lib.fibonacci_circuit(g["n"], oracle_lib.MODULUS[field])
    oc = cl.OracleCircuit(oracle, circuit).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, inputs, rc=rc)
    L = layer_lib.OracleLayer(oracle, field, oc.workload_arrays(), prm, packing=dict(g["packing"]), rc=rc)
    same_prep = np.array_equal(proof.preprocessed_commitment, L.prep_commit())
    print(f"{field}: upstream's ZK preprocessed commitment {'equals' if same_prep else 'differs from'} the zero-padded one here")


# ---- the hiding MMCS (MerkleTreeHidingMmcs: recursion/tests/zk_hiding_mmcs.rs) - pins p3r_config.mmcs_salt_elems ----------
@pytest.mark.parametrize("field,key", FIELDS)
def test_rust_hiding_mmcs_openings_are_accepted(oracle, field, key):
    """A NATIVE salted tree (tools/rust_pin: hiding_mmcs()): every opening `(salts, siblings)` must be accepted by
    p3r_mmcs_verify_salted and by the oracle's verify - leaf preimage `[row | salt]` per matrix of a height class in matrix
    order (recursion/src/pcs/mmcs.rs:315-413), opening-proof layout (:763-790) - and refused with one salt element or one
    opened value changed.  (The salts are the fixture's data: upstream draws them from the MMCS's own SmallRng.)"""
    import plonky3_recursion_amd as p3r
    g = load(f"rust_hiding_mmcs_{key}.json")
    rc = np.array(g["rc"], dtype=np.uint32)
    S = g["salt_elems"]
    cfg, keep = p3r.make_config(field, poseidon2_rc=rc, cap_height=g["cap_height"], mmcs_salt_elems=S, zk_seed=1)
    dims = [(m["height"], m["width"]) for m in g["matrices"]]
    cap = np.array(g["root"], dtype=np.uint32)
    for o in g["openings"]:
        opened = np.array([v for row in o["opened_values"] for v in row], dtype=np.uint32)
        salts = np.array(o["salts"], dtype=np.uint32).reshape(len(dims), S)
        proof = np.array(o["siblings"], dtype=np.uint32).reshape(-1, 8)
        p3r.mmcs_verify(cfg, cap, dims, o["index"], opened, proof, salts=salts)
        # the oracle: a hiding commitment of [M0, M1] with salts [S0, S1] is the plain one of [M0 | S0], [M1 | S1]
        wide = np.concatenate([np.concatenate([np.array(r, dtype=np.uint32), salts[k]]) for k, r in enumerate(o["opened_values"])])
        oracle.verify(field, cap, [(h, w + S) for h, w in dims], o["index"], wide, proof, rc=rc)
        bad = salts.copy()
        bad[1, S - 1] ^= 1
        with pytest.raises(p3r.P3rError):
            p3r.mmcs_verify(cfg, cap, dims, o["index"], opened, proof, salts=bad)
        bad = opened.copy()
        bad[0] ^= 1
        with pytest.raises(p3r.P3rError):
            p3r.mmcs_verify(cfg, cap, dims, o["index"], bad, proof, salts=salts)


@pytest.mark.parametrize("field,key", FIELDS)
def test_rust_hiding_mmcs_proof_is_accepted(oracle, field, key):
    """A NATIVE proof under HidingFriPcs WITH the hiding MMCS for the input and the commit-phase trees (tools/rust_pin:
    fibonacci_hiding_layer(), the configuration of recursion/tests/zk_hiding_mmcs.rs:41-58,120-131): the wire format must
    round-trip under P3R_PROOF_ZK | P3R_PROOF_SALTED, p3r_verify_batch under zk = 1, mmcs_salt_elems = 4 and the oracle's
    verify_batch must accept it, and the configuration without salts must refuse it."""
    import layer_lib
    import plonky3_recursion_amd as p3r
    g = load(f"rust_fibonacci_hiding_layer_{key}.json")
    rc = np.array(g["rc"], dtype=np.uint32)
    inner, outer = bytes.fromhex(g["batch_proof_postcard_hex"]), bytes.fromhex(g["batch_stark_proof_postcard_hex"])
    zk = dict(zk=1, num_random_codewords=g["zk"]["num_random_codewords"], mmcs_salt_elems=g["mmcs_salt_elems"])
    proof = p3r.BatchStarkProof.from_postcard(outer, field, zk=True, salted=True)
    assert proof.proof == inner and proof.to_postcard() == outer, "BatchStarkProof postcard layout under the hiding MMCS"
    assert list(proof.degree_bits) == g["degree_bits"]
    cfg, keep = p3r.make_config(field, poseidon2_rc=rc, zk_seed=1, **g["fri"], **zk)
    p3r.verify_all_tables(cfg, proof)
    cfg0, keep0 = p3r.make_config(field, poseidon2_rc=rc, zk=1, num_random_codewords=g["zk"]["num_random_codewords"], **g["fri"])
    with pytest.raises(p3r.P3rError):
        p3r.verify_all_tables(cfg0, p3r.BatchStarkProof.from_postcard(outer, field, zk=True, salted=True))
    prm = layer_lib.params(zk_key=(1, 0, 0, 0, 0, 0, 0, 0), **g["fri"], **zk)
    layer_lib.oracle_verify_statement(oracle, field, prm, proof.airs(), proof.preprocessed_commitment, inner, rc=rc)


@pytest.mark.parametrize("field,key", FIELDS)
def test_rust_accepts_our_zk_proof(field, key):
    """The other direction: `cargo run -- zk-accept` (tools/rust_pin) ran the reference's verify_all_tables on the ZK proof
    this repo made (tests/golden/zk_fibonacci_layer_for_rust_<field>.json, tools/gen_zk_fixture.py)."""
    import hashlib
    acc = load("rust_zk_acceptance.json")[key]
    fx = load(f"zk_fibonacci_layer_for_rust_{key}.json")
    assert hashlib.sha256(bytes.fromhex(fx["batch_stark_proof_postcard_hex"])).hexdigest() == fx["sha256"] == acc["fixture_sha256"], \
        "the acceptance record is about another fixture: re-run `cargo run -- zk-accept`"
    assert acc["fixture_round_constants_are_upstream"], "regenerate the fixture after rust_primitives.json exists (tools/gen_zk_fixture.py)"
    assert acc["deserialised"], acc["verdict"]
    assert acc["verdict"] == "accepted", acc["verdict"]


@pytest.mark.parametrize("field,key", FIELDS)
def test_our_zk_fixture_is_current(oracle, field, key):
    """The committed fixture is what the prover makes today and both verifiers here accept it (runs without cargo)."""
    import hashlib
    import sys
    import plonky3_recursion_amd as p3r
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_zk_fixture
    fx = load(f"zk_fibonacci_layer_for_rust_{key}.json")
    outer = bytes.fromhex(fx["batch_stark_proof_postcard_hex"])
    assert hashlib.sha256(outer).hexdigest() == fx["sha256"]
    if not os.path.exists(os.path.join(GOLDEN, "rust_primitives.json")):
        assert gen_zk_fixture.make(field, key)["sha256"] == fx["sha256"], "prover output drifted: python tools/gen_zk_fixture.py"
    proof = p3r.BatchStarkProof.from_postcard(outer, field, zk=True)
    assert len(proof.proof) == fx["batch_proof_len"]
    cfg, keep = p3r.make_config(field, poseidon2_rc=np.array(fx["rc"], dtype=np.uint32), zk=1, num_random_codewords=2, **fx["fri"])
    p3r.verify_all_tables(cfg, proof)
