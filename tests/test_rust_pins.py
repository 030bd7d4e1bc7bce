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
    L = layer_lib.OracleLayer(oracle, field, oc.workload_arrays(), prm, packing=dict(g["packing"]), rc=rc)
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
    tp = p3r.TablePacking(public_lanes=1, alu_lanes=1).with_fri_params(g["fri"]["log_final_poly_len"], g["fri"]["log_blowup"])
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
