"""GPU: the width-32 Poseidon2 op at the CIRCUIT seam (P3R_OP_POSEIDON2_W32_PERM, ABI version 8) - the MMCS rows of a
verifier circuit built under `--arity4` (recursion/examples/recursive_aggregation.rs:902-1046: W16 challenger rows, W32
MMCS rows).  The device runner's width-32 rows (PoseidonPermExecutor::execute for is_arity4(), circuit/src/ops/
poseidon_perm/executor.rs:92-235,947-966), the 48-column preprocessed rows with their multiplicities (executor.rs:770-893,
batch_stark_prover.rs:97-246), the preprocessed commitment and the proof BYTES of p3r_prove_next_layer against the oracle's
sequential restatement - under the binary MMCS and under the prover's own arity-4 MMCS; both verifiers accept; the
CircuitError paths of the new op."""
import numpy as np
import pytest

import circuit_lib as cl
import harness_lib
import layer_lib
import oracle_lib

pytestmark = pytest.mark.gpu
GEN = dict(horner_chain_len=16, sponge_chain_len=3, merkle_depth=6)
SETS = [dict(log_blowup=1, max_log_arity=2, log_final_poly_len=1, query_pow_bits=3, num_queries=4),
        dict(log_blowup=2, max_log_arity=3, log_final_poly_len=2, cap_height=1, query_pow_bits=5, num_queries=6),
        dict(log_blowup=2, max_log_arity=2, log_final_poly_len=5, query_pow_bits=8, num_queries=8)]


def setup(oracle, field, log_h, kw, seed=None, **ctx_kw):
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    a = harness_lib.generate(field, log_h, seed=60 + log_h if seed is None else seed, flags=harness_lib.P2_W32_OPS, **GEN)
    oc = cl.OracleCircuit(oracle, cl.Circuit.from_arrays(a)).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, cl.Inputs.from_arrays(a))
    prm = layer_lib.params(**kw, **ctx_kw)
    ctx = p3r.Context(field=field, allow_unpinned_w32_defaults=True, **kw, **ctx_kw)
    tp = p3r.TablePacking().with_fri_params(kw["log_final_poly_len"], kw["log_blowup"])
    cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(a), p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=tp))
    return a, oc, prm, ctx, cache, wl.circuit_inputs_from_arrays(a)


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
@pytest.mark.parametrize("k,log_h", [(0, 7), (1, 9), (2, 10)])
def test_runner_prep_and_proof_bytes_equal_oracle(oracle, field, k, log_h):
    import plonky3_recursion_amd as p3r
    a, oc, prm, ctx, cache, inputs = setup(oracle, field, log_h, SETS[k])
    want = oc.workload_arrays()
    pc = cache.prepared_circuit
    cpd = pc.circuit_prover_data
    assert cpd.rows["poseidon2_w32"] == int(want["counts"][7]) > 0 and cpd.p2w_height > 0
    assert [cpd.rows[t] for t in ("const", "public", "alu", "poseidon2", "recompose")] == [int(x) for x in want["counts"][:5]]
    # the device runner: every table's rows, the width-32 ones included, against the sequential runner
    res = pc.run(inputs)
    assert np.array_equal(res.download("p2w_input_values").reshape(-1), want["p2w_inputs"])
    assert np.array_equal(res.download("p2w_flags").reshape(-1), want["p2w_flags"])
    assert np.array_equal(res.download("p2w_mmcs_index_sum").reshape(-1), want["p2w_mmcs_index_sum"])
    assert np.array_equal(res.download("p2_input_values").reshape(-1), want["p2_inputs"])
    assert np.array_equal(res.download("alu_values").reshape(-1), want["alu_values"])
    assert np.array_equal(res.download("recompose_values").reshape(-1), want["recompose_values"])
    assert np.array_equal(res.download("public_values").reshape(-1), want["public_values"])
    # ... and against the generator's own books (rows written from the AIR's point of view)
    assert np.array_equal(res.download("p2w_input_values").reshape(-1), a["p2w_inputs"])
    # the preparation: the commitment binds the 48-column rows and every multiplicity of the six tables
    L = layer_lib.OracleLayer(oracle, field, want, prm)
    assert np.array_equal(cpd.preprocessed_commitment, L.prep_commit())
    # the proof: prove_next_layer(circuit, inputs), from host inputs and from resident ones
    proof = L.prove()
    out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=inputs), ctx, p3r.FriRecursionBackend(),
                               p3r.ProveNextLayerParams(table_packing=pc.packing), prep=cache)
    assert out.proof.proof == proof
    rin = pc.upload_inputs(inputs)
    assert pc.prove(rin) == proof
    f = field.replace("-", "_")
    assert [e.op_type for e in out.proof.non_primitives][:2] == ["poseidon2_perm/%s_d4_w16" % f, "poseidon2_perm/%s_d4_w32" % f]
    assert [x["kind"] for x in out.proof.airs()] == [0, 1, 2, 3, 5, 4]
    cache.prover.verify_all_tables(out.proof)   # native verifier
    L.verify(proof)                             # the oracle's verifier
    back = p3r.BatchStarkProof.from_postcard(out.proof.to_postcard(), field)
    cache.prover.verify_all_tables(back)
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 4
    with pytest.raises(p3r.P3rError):
        cache.prover.verify_all_tables(p3r.BatchStarkProof(**{**out.proof.__dict__, "proof": bytes(bad)}))
    assert pc.levels >= 2
    rin.free(); res.free(); pc.free(); ctx.close()


@pytest.mark.parametrize("field", ["koala-bear", "baby-bear"])
def test_arity4_recursion_layer(oracle, field):
    """The layer `recursive_aggregation --arity4` proves: a circuit with width-32 MMCS rows under the prover's own arity-4
    MMCS (p3r_config.mmcs_arity = 4) - proof bytes equal the oracle's, both verifiers accept."""
    import plonky3_recursion_amd as p3r
    a, oc, prm, ctx, cache, inputs = setup(oracle, field, 9, SETS[2], mmcs_arity=4)
    want = oc.workload_arrays()
    pc = cache.prepared_circuit
    L = layer_lib.OracleLayer(oracle, field, want, prm)
    assert np.array_equal(pc.circuit_prover_data.preprocessed_commitment, L.prep_commit())
    proof = pc.prove(inputs)
    assert proof == L.prove()
    prover = p3r.BatchStarkProver(ctx)
    prover.verify_all_tables(prover.wrap_proof(proof, pc.circuit_prover_data))
    L.verify(proof)
    pc.free(); ctx.close()


def test_five_chained_layers_reuse_one_prep(oracle):
    """A cache hit: the width-32 circuit prepared once, five proofs from different inputs of the same shape."""
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    field, kw = "koala-bear", SETS[0]
    a, oc, prm, ctx, cache, inputs = setup(oracle, field, 8, kw)
    pc = cache.prepared_circuit
    first = pc.prove(inputs)
    for _ in range(4):
        assert pc.prove(inputs) == first      # same inputs, same bytes: no state leaks between runs
    pc.free(); ctx.close()


def _mini(field):
    """const 0, const 1, public x (2 limbs worth), a 2-row leaf sponge then one compression row."""
    P = oracle_lib.MODULUS[field]
    NO = cl.NO_W
    ops, ext = [], []

    def push(kind, a=0, b=0, c=NO, out=0, aux=NO, e=()):
        ops.append([kind, a, b, c, out, aux, len(ext), len(e)])
        ext.extend(e)
    push(cl.OP_CONST, out=0, e=[0, 0, 0, 0])
    push(cl.OP_CONST, out=1, e=[1, 0, 0, 0])
    for i in range(6):
        push(cl.OP_PUBLIC, out=2 + i, aux=i)
    # leaf sponge: new_start, six rate limbs from the bus; then a continuation absorbing two limbs
    push(cl.OP_P2W, a=0, aux=1, e=[2, 3, 4, 5, 6, 7, NO, NO, NO, NO, NO, 6] + [NO] * 6)
    push(cl.OP_P2W, a=1, aux=0, e=[2, 3, NO, NO, NO, NO, NO, NO, NO, NO, NO, 6] + [NO] * 6)
    # one 4-to-1 compression: continues the sponge (new_start = 0), pos = 1 + 2 * 0, outputs exposed
    push(cl.OP_P2W, a=2, aux=2, e=[NO] * 8 + [NO, 1, 0, 6, 8, 9, 10, 11, 12, 13])
    return ops, ext, P


def test_runner_errors_of_the_width32_op(oracle):
    """CircuitError paths of the arity-4 shape on the device: non-boolean direction bits (executor.rs:305-335), a missing
    bit, a chain without a previous state, private data for a sponge row, private data of the wrong width, an
    mmcs_index_sum witness (unsupported here), a duplicate op id."""
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    ops, ext, P = _mini(field)
    ctx = p3r.Context(field=field, allow_unpinned_w32_defaults=True, **SETS[0])
    tp = p3r.TablePacking().with_fri_params(SETS[0]["log_final_poly_len"], SETS[0]["log_blowup"])
    pub = np.arange(24, dtype=np.uint32).reshape(6, 4) + 5
    sib = np.arange(24, dtype=np.uint32) + 100

    def prepared(o=ops, e=ext):
        return p3r.PreparedCircuit(ctx, p3r.Circuit(14, np.array(o, dtype=np.uint32), np.array(e, dtype=np.uint32),
                                                    np.arange(2, 8, dtype=np.uint32)), tp)
    pc = prepared()
    assert pc.prepared_on_device         # (csrc/prep_device.hip covers the width-32 op; the flagged variants below go to the host path for their error)
    good = p3r.CircuitInputs(public_values=pub, private_data_w32_op_ids=np.array([2], np.uint32), private_data_w32_siblings=sib.reshape(1, 24))
    res = pc.run(good)
    # against the oracle's sequential runner
    oc = cl.OracleCircuit(oracle, cl.Circuit(14, ops, ext, list(range(2, 8))))
    oc.preprocess(P).run(field, cl.Inputs(pub, (), (), (), [2], sib))
    assert np.array_equal(res.download("p2w_input_values").reshape(-1), oc.get("p2w_inputs"))
    fl = res.download("p2w_flags")
    assert fl.tolist() == [[1, 0, 0, 0], [0, 0, 0, 0], [0, 1, 1, 0]]
    row = res.download("p2w_input_values")[2]
    assert row[0:8].tolist() == sib[0:8].tolist() and row[16:32].tolist() == sib[8:24].tolist()   # chunks 0, 2, 3: the siblings
    res.free()
    # a W16-list entry that names a width-32 op, and the other way round
    with pytest.raises(p3r.P3rError, match="other width"):
        pc.run(p3r.CircuitInputs(public_values=pub, private_data_op_ids=np.array([2], np.uint32), private_data_siblings=sib[:8].reshape(1, 8)))
    # private data on a sponge row
    with pytest.raises(p3r.P3rError, match="non-Merkle"):
        pc.run(p3r.CircuitInputs(public_values=pub, private_data_w32_op_ids=np.array([0], np.uint32), private_data_w32_siblings=sib.reshape(1, 24)))
    with pytest.raises(p3r.P3rError, match="already set"):
        pc.run(p3r.CircuitInputs(public_values=pub, private_data_w32_op_ids=np.array([2, 2], np.uint32),
                                 private_data_w32_siblings=np.stack([sib, sib])))
    pc.free()
    # a direction bit that is not boolean: witness 2 holds (5, 6, 7, 8)
    o2 = [list(r) for r in ops]
    e2 = list(ext)
    e2[o2[-1][6] + 9] = 2
    pc = prepared(o2, e2)
    with pytest.raises(p3r.P3rError, match="boolean mmcs_bit"):
        pc.run(good)
    pc.free()
    # static errors: a missing bit on a Merkle row, an accumulator witness, a chain with no previous state, a duplicate id
    e3 = list(ext); e3[ops[-1][6] + 10] = cl.NO_W
    with pytest.raises(p3r.P3rError, match="mmcs_bit2 must be provided"):
        prepared(ops, e3)
    e4 = list(ext); e4[ops[-1][6] + 8] = 1
    with pytest.raises(p3r.P3rError, match="mmcs_index_sum"):
        prepared(ops, e4)
    o5 = [list(r) for r in ops[:8]] + [list(ops[-1])]          # the compression row without its leaf sponge
    pc = prepared(o5, ext)
    with pytest.raises(p3r.P3rError, match="Poseidon2ChainMissingPreviousState"):
        pc.run(p3r.CircuitInputs(public_values=pub))
    pc.free()
    o6 = [list(r) for r in ops]; o6[-1][1] = 1
    with pytest.raises(p3r.P3rError, match="duplicate NonPrimitiveOpId"):
        prepared(o6, ext)
    # the op needs a D = 4 circuit
    ctx1 = p3r.Context(field=field, ext_degree=1, allow_unpinned_w32_defaults=True, **SETS[0])
    with pytest.raises(p3r.P3rError):
        p3r.PreparedCircuit(ctx1, p3r.Circuit(14, np.array(ops, dtype=np.uint32), np.array(ext, dtype=np.uint32), np.arange(2, 8, dtype=np.uint32)), tp)
    ctx1.close()
    ctx.close()


def test_width32_defaults_must_be_acknowledged():
    """The built-in width-32 constants are self-generated (P3R_EXT_UNPINNED_W32_DEFAULTS): a circuit that holds width-32
    ops, like the arity-4 MMCS, is refused on a context that did not ask for them."""
    import plonky3_recursion_amd as p3r
    field = "koala-bear"
    ops, ext, P = _mini(field)
    with pytest.raises(p3r.P3rError, match="UNPINNED|unpinned|self-generated"):
        p3r.Context(field=field, mmcs_arity=4, **SETS[0])
    ctx = p3r.Context(field=field, **SETS[0])
    tp = p3r.TablePacking().with_fri_params(SETS[0]["log_final_poly_len"], SETS[0]["log_blowup"])
    with pytest.raises(p3r.P3rError, match="UNPINNED|unpinned|self-generated"):
        p3r.PreparedCircuit(ctx, p3r.Circuit(14, np.array(ops, dtype=np.uint32), np.array(ext, dtype=np.uint32), np.arange(2, 8, dtype=np.uint32)), tp)
    ctx.close()
