"""CPU: the oracle's circuit-side restatement (oracle/circuit.hpp).

1. generate_preprocessed_columns and the runner against the literal expectations of the
   reference's own unit tests (tests/golden/reference_unit_tests.json, each case cites file:line);
2. preprocessing + run of the synthetic circuits against the harness's own, independently kept
   bookkeeping of the same workload (values, indices, signed multiplicities);
3. the runner's error paths (CircuitError variants of circuit/src/tables/runner.rs)."""
import json
import os

import numpy as np
import pytest

import circuit_lib as cl
import harness_lib
import layer_lib
import oracle_lib

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_unit_tests.json")))
KIND = dict(const=cl.OP_CONST, public=cl.OP_PUBLIC, add=cl.OP_ADD, mul=cl.OP_MUL, bool=cl.OP_BOOL,
            muladd=cl.OP_MULADD, horner=cl.OP_HORNER)
BABYBEAR = 0x78000001


def build(case):
    ops, ext = [], []
    for o in case["ops"]:
        e = list(o["ext"])
        if o["kind"] == "const":
            e = e + [0, 0, 0]   # a base-field constant embedded in the extension
        ops.append([KIND[o["kind"]], o["a"], o["b"], o["c"], o["out"], o["aux"], len(ext), len(e)])
        ext += e
    return cl.Circuit(case["witness_count"], ops, ext, case.get("public_rows", ()), case.get("private_rows", ()))


@pytest.mark.parametrize("case", GOLD["preprocessed_columns"], ids=lambda c: c["source"].split()[-1])
def test_preprocessed_columns_match_reference_unit_tests(oracle, case):
    oc = cl.OracleCircuit(oracle, build(case)).preprocess(BABYBEAR, case["D"])
    for name, want in case["expect"].items():
        assert oc.get(name).tolist() == want, name


@pytest.mark.parametrize("case", GOLD["runner"], ids=lambda c: c["source"].split()[1])
def test_runner_matches_reference_unit_tests(oracle, case):
    emb = lambda vals: [x for v in vals for x in (v, 0, 0, 0)]
    oc = cl.OracleCircuit(oracle, build(case))
    oc.run("baby-bear", cl.Inputs(emb(case["public_values"]), emb(case["private_values"])))
    e = case["expect"]
    for name in ("witness", "const_values", "public_values"):
        got = oc.get(name).reshape(-1, 4)
        assert got[:, 0].tolist() == e[name], name
        assert not got[:, 1:].any(), name
    alu = oc.get("alu_values").reshape(-1, 4, 4)
    assert alu[:, :, 0].tolist() == e["alu_values"]
    assert not alu[:, :, 1:].any()


SHAPES = [0, harness_lib.NO_POSEIDON2, harness_lib.NO_RECOMPOSE, harness_lib.RECOMPOSE_COEFF,
          harness_lib.RECOMPOSE_COEFF | harness_lib.NO_POSEIDON2, harness_lib.RECOMPOSE_BOTH,
          harness_lib.RECOMPOSE_BOTH | harness_lib.NO_POSEIDON2,
          harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE | harness_lib.SINGLE_PUBLIC,
          harness_lib.NO_POSEIDON2 | harness_lib.NO_ALU,
          # the width-32 Poseidon2 rows as ops of the circuit (arity-4 Merkle verification chains: kind 11)
          harness_lib.P2_W32_OPS]


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
@pytest.mark.parametrize("flags", SHAPES)
def test_circuit_path_reproduces_harness_bookkeeping(oracle, field, flags):
    a = harness_lib.generate(field, 8, seed=21, horner_chain_len=16, sponge_chain_len=3, merkle_depth=5, flags=flags)
    oc = cl.OracleCircuit(oracle, cl.Circuit.from_arrays(a)).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, cl.Inputs.from_arrays(a))
    w = oc.workload_arrays()
    for k in w:
        if k == "counts":   # counts[7]: rows of the width-32 Poseidon2 table, which is not part of the flattened circuit
            assert np.array_equal(w[k], a[k][:len(w[k])]) and not a[k][len(w[k]):].any(), k
            continue
        assert np.array_equal(w[k], a[k]), k
    if flags == harness_lib.P2_W32_OPS:
        # the generator keeps its own books (its rows follow the AIR, poseidon2-circuit-air/src/air.rs:1178-1342); the
        # oracle restates the executor: both reach the same 48-column rows, inputs, flags and multiplicities
        ops = a["ops"].reshape(-1, 8)
        w32 = ops[ops[:, 0] == cl.OP_P2W]
        fl = a["p2w_flags"].reshape(-1, 4)
        assert len(w32) == a["counts"][7] == len(fl) and len(a["pdw_op_ids"]) == int(fl[:, 1].sum())
        seeded = [r for r in range(1, len(fl)) if fl[r, 1] and not fl[r, 0] and not fl[r - 1, 1]]
        assert seeded, "a Merkle chain that continues a leaf sponge (update_chain_state's seed, executor.rs:462-491)"
        assert (fl[:, 1] & fl[:, 0]).any(), "a Merkle chain that starts from a CTL-loaded digest"
        prep = a["p2w_prep"].reshape(-1, 48)
        assert (prep[fl[:, 1] == 1][:, 1:32:4].sum(axis=1) >= 4).any(), "an injection / bridge level with CTL-loaded pads"
        assert not a["p2w_mmcs_index_sum"].any()
    if flags == 0:
        kinds = np.bincount(a["ops"].reshape(-1, 8)[:, 0], minlength=11)
        assert kinds.all()   # every op kind is exercised
        assert len(a["private_rows"]) and len(a["rewrite"]) and (a["p2_out_ctl"] == oracle_lib.MODULUS[field] - 1).any()


@pytest.mark.parametrize("flags", [0, harness_lib.RECOMPOSE_COEFF, harness_lib.RECOMPOSE_BOTH, harness_lib.P2_W32_OPS])
def test_circuit_path_proof_verifies(oracle, flags):
    """Traces + preprocessed columns derived from the circuit prove and verify (LogUp balanced
    under the reference's creator / reader multiplicity rules, every AIR satisfied).  RECOMPOSE_COEFF: the Recompose
    ops are the `recompose/coeff` kind - decomposition-hint outputs are created on the bus by the recompose rows and
    read by sponge inputs (circuit_builder.rs:1438-1463)."""
    field = "koala-bear"
    a = harness_lib.generate(field, 7, seed=4, horner_chain_len=10, sponge_chain_len=3, merkle_depth=4, flags=flags)
    oc = cl.OracleCircuit(oracle, cl.Circuit.from_arrays(a)).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, cl.Inputs.from_arrays(a))
    prm = layer_lib.params(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4)
    coeff = int(bool(flags & harness_lib.RECOMPOSE_COEFF))
    w = oc.workload_arrays()
    if coeff:
        rp = w["recompose_prep"].reshape(-1, 10)
        assert (rp[:, 3::2] > 0).any() and (rp[:, 1] == oracle_lib.MODULUS[field] - 1).any()   # owned coefficients that are read; outputs connected back
    if flags & harness_lib.RECOMPOSE_BOTH:
        # both Recompose tables (recompose_table_provers(lanes, true)): `recompose`, then `recompose/coeff`
        assert w["counts"][4] and w["counts"][6]
    L = layer_lib.OracleLayer(oracle, field, w, prm, packing=dict(recompose_coeff_lookups=coeff))
    if flags == harness_lib.P2_W32_OPS:
        # six tables: the width-32 one right after the width-16 one (W16 challenger rows, W32 MMCS rows)
        assert [x["kind"] for x in L.tables()] == ["const", "public", "alu", "poseidon2", "poseidon2_w32", "recompose"]
    if flags & harness_lib.RECOMPOSE_BOTH:
        t = L.tables()
        assert [x["kind"] for x in t[-2:]] == ["recompose", "recompose"] and t[-2]["prep"].shape[1] == 2 and t[-1]["prep"].shape[1] == 10
    L.verify(L.prove())


def small_circuit():
    # w0 = 0, w1 = 5 (const), w2 public, w3 = w2 + w1, w4 private
    ops = [[cl.OP_CONST, 0, 0, cl.NO_W, 0, cl.NO_W, 0, 4], [cl.OP_CONST, 0, 0, cl.NO_W, 1, cl.NO_W, 4, 4],
           [cl.OP_PUBLIC, 0, 0, cl.NO_W, 2, 0, 0, 0], [cl.OP_ADD, 2, 1, cl.NO_W, 3, cl.NO_W, 0, 0]]
    return ops, [0, 0, 0, 0, 5, 0, 0, 0]


def test_runner_error_paths(oracle):
    ops, ext = small_circuit()
    ok = cl.Circuit(4, ops, ext, [2])
    cl.OracleCircuit(oracle, ok).run("koala-bear", cl.Inputs([3, 0, 0, 0]))
    # WitnessConflict: out already holds a different value (runner.rs:473-510)
    bad = cl.Circuit(4, ops + [[cl.OP_ADD, 1, 1, cl.NO_W, 3, cl.NO_W, 0, 0]], ext, [2])
    with pytest.raises(RuntimeError, match="WitnessConflict"):
        cl.OracleCircuit(oracle, bad).run("koala-bear", cl.Inputs([3, 0, 0, 0]))
    # ... but an equal value is accepted
    same = cl.Circuit(4, ops + [[cl.OP_ADD, 1, 2, cl.NO_W, 3, cl.NO_W, 0, 0]], ext, [2])
    cl.OracleCircuit(oracle, same).run("koala-bear", cl.Inputs([3, 0, 0, 0]))
    # DivisionByZero: backward Mul with a = 0 (runner.rs:376-379)
    div = cl.Circuit(5, ops + [[cl.OP_MUL, 0, 4, cl.NO_W, 3, cl.NO_W, 0, 0]], ext, [2])
    with pytest.raises(RuntimeError, match="DivisionByZero"):
        cl.OracleCircuit(oracle, div).run("koala-bear", cl.Inputs([3, 0, 0, 0]))
    # WitnessNotSet (a operand) and WitnessNotSetForIndex (never written)
    with pytest.raises(RuntimeError, match="WitnessNotSet"):
        cl.OracleCircuit(oracle, cl.Circuit(6, ops + [[cl.OP_ADD, 5, 1, cl.NO_W, 4, cl.NO_W, 0, 0]], ext, [2])).run(
            "koala-bear", cl.Inputs([3, 0, 0, 0]))
    with pytest.raises(RuntimeError, match="WitnessNotSetForIndex"):
        cl.OracleCircuit(oracle, cl.Circuit(5, ops, ext, [2])).run("koala-bear", cl.Inputs([3, 0, 0, 0]))
    # UnclaimedPrivateInput at preprocessing (circuit.rs:497-503)
    with pytest.raises(RuntimeError, match="UnclaimedPrivateInput"):
        cl.OracleCircuit(oracle, cl.Circuit(5, ops, ext, [2], [4])).preprocess(0x7F000001)


@pytest.mark.parametrize("case", GOLD["table_matrices"], ids=lambda c: c["source"].split()[1])
def test_const_public_matrices_match_reference_unit_tests(oracle, case):
    """ConstAir / PublicAir trace_to_matrix + preprocessed_trace for D = 4 (reference literals)."""
    z32 = np.zeros(0, np.uint32)
    vals = np.array(case["values"], np.uint32).reshape(-1)
    prep = np.array(case["prep"], np.uint32).reshape(-1)
    n = len(case["values"])
    is_const = case["table"] == "const"
    one_const = (np.array([0, 0, 0, 0], np.uint32), np.array([0, 0], np.uint32))
    arrs = {k: z32 for k in harness_lib.ARRAYS}
    arrs.update(const_values=vals if is_const else one_const[0], const_prep=prep if is_const else one_const[1],
                public_values=z32 if is_const else vals, public_prep=z32 if is_const else prep,
                alu_values=np.zeros(16, np.uint32), alu_prep13=np.zeros(13, np.uint32),
                counts=np.array([n if is_const else 1, 0 if is_const else n, 1, 0, 0, 0], np.uint32))
    prm = layer_lib.params(log_blowup=1, max_log_arity=1, log_final_poly_len=0, query_pow_bits=0, num_queries=1)
    L = layer_lib.OracleLayer(oracle, "baby-bear", arrs, prm,
                              packing=dict(public_lanes=case["lanes"], min_trace_height=case["min_trace_height"]))
    t = {x["kind"]: x for x in L.tables()}[case["table"]]
    assert t["main"].tolist() == case["expect_main"]
    assert t["prep"].tolist() == case["expect_prep"]
