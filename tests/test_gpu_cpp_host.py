"""GPU: the C++ host-side mirror (include/p3r.hpp) driven by examples/prove_next_layer.cpp -
build_next_layer_prep + prove_next_layer + verify_all_tables from compiled code - produces the same
proof bytes as the Python binding for the same circuit, and the oracle verifier accepts them."""
import os
import subprocess

import numpy as np
import pytest

import circuit_lib as cl
import harness_lib
import layer_lib
import oracle_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "prove_next_layer")


@pytest.mark.parametrize("field,log_h", [("koala-bear", 12), ("baby-bear", 11)])
def test_cpp_host_matches_python_binding(oracle, tmp_path, field, log_h):
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True)
    out_file = str(tmp_path / "proof.bin")
    r = subprocess.run([EXE, field, str(log_h), out_file, "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verify_all_tables ok" in r.stdout and r.stdout.strip().endswith("ok")
    got = open(out_file, "rb").read()
    # the same circuit through the Python binding
    a = harness_lib.generate(field, log_h, seed=0x5EED0000, horner_chain_len=64, sponge_chain_len=8, merkle_depth=20)
    fri = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15,
               num_queries=54)
    ctx = p3r.Context(field=field, **fri, allow_unpinned_w32_defaults=True)
    tp = p3r.TablePacking().with_fri_params(5, 2)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    assert pc.prove(wl.circuit_inputs_from_arrays(a)) == got
    # and the oracle's verifier, against the oracle's own preprocessing of that circuit
    oc = cl.OracleCircuit(oracle, cl.Circuit.from_arrays(a)).preprocess(oracle_lib.MODULUS[field])
    oc.run(field, cl.Inputs.from_arrays(a))
    L = layer_lib.OracleLayer(oracle, field, oc.workload_arrays(), layer_lib.params(**fri))
    L.verify(got, prep_cap=pc.circuit_prover_data.preprocessed_commitment)
    pc.free()
    ctx.close()


def test_cpp_host_quintic_layer(oracle, tmp_path):
    """`prove_next_layer <field> <log_h> <out> <layers> --quintic`: the compiled caller under koala_bear_quintic_params -
    a D = 5 verifier circuit with base-mode Poseidon2 permutations and both Recompose kinds, Challenge = the quintic
    field - gives the bytes of the Python binding; the oracle proves the same bytes from the generator's tables."""
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    out_file = str(tmp_path / "proof5.bin")
    r = subprocess.run([EXE, "koala-bear", "11", out_file, "2", "--quintic"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verify_all_tables ok" in r.stdout and "prove_aggregation_layer (cached prep)" in r.stdout
    got = open(out_file, "rb").read()
    a = harness_lib.generate("koala-bear", 11, seed=0x5EED0000, horner_chain_len=64, sponge_chain_len=8, merkle_depth=20,
                             flags=harness_lib.RECOMPOSE_BOTH, ext_degree=5)
    fri = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15,
               num_queries=54)
    ctx = p3r.Context(field="koala-bear", ext_degree=5, challenge_degree=5, **fri, allow_unpinned_w32_defaults=True)
    tp = p3r.TablePacking().with_fri_params(5, 2)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    assert pc.prove(wl.circuit_inputs_from_arrays(a, 5)) == got
    L = layer_lib.OracleLayer(oracle, "koala-bear", a, layer_lib.params(challenge_degree=5, **fri), packing=dict(ext_degree=5))
    assert np.array_equal(pc.circuit_prover_data.preprocessed_commitment, L.prep_commit())
    assert L.prove() == got
    pc.free()
    ctx.close()


def test_cpp_host_arity4_layer(oracle, tmp_path):
    """`prove_next_layer <field> <log_h> <out> <layers> --arity4`: the compiled caller with `FriParams::mmcs_arity = 4`
    (MyMmcsArity4: 4-to-1 trees over the width-32 permutation) gives the bytes of the Python binding and of the oracle."""
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    out_file = str(tmp_path / "proof4.bin")
    r = subprocess.run([EXE, "baby-bear", "10", out_file, "2", "--arity4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verify_all_tables ok" in r.stdout
    got = open(out_file, "rb").read()
    a = harness_lib.generate("baby-bear", 10, seed=0x5EED0000, horner_chain_len=64, sponge_chain_len=8, merkle_depth=20)
    fri = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15,
               num_queries=54, mmcs_arity=4)
    ctx = p3r.Context(field="baby-bear", **fri, allow_unpinned_w32_defaults=True)
    tp = p3r.TablePacking().with_fri_params(5, 2)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    assert pc.prove(wl.circuit_inputs_from_arrays(a)) == got
    L = layer_lib.OracleLayer(oracle, "baby-bear", a, layer_lib.params(**fri))
    assert np.array_equal(pc.circuit_prover_data.preprocessed_commitment, L.prep_commit())
    assert L.prove() == got
    pc.free()
    ctx.close()


def test_cpp_host_zk_layer(oracle, tmp_path):
    """`prove_next_layer <field> <log_h> <out> <layers> --zk`: the compiled caller with `FriParams::zk` (HidingFriPcs, two
    random codewords, seed 3).  The example checks that two proofs of one input differ and that `p3r_zk_set_nonce` replays
    one; the proof it writes is proof number 7, which the Python binding and the oracle reproduce byte for byte."""
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    out_file = str(tmp_path / "proofz.bin")
    r = subprocess.run([EXE, "koala-bear", "10", out_file, "2", "--zk"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verify_all_tables ok" in r.stdout and "postcard round trip ok" in r.stdout and r.stdout.strip().endswith("ok")
    got = open(out_file, "rb").read()
    a = harness_lib.generate("koala-bear", 10, seed=0x5EED0000, horner_chain_len=64, sponge_chain_len=8, merkle_depth=20)
    fri = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15,
               num_queries=54)
    ctx = p3r.Context(field="koala-bear", zk=1, num_random_codewords=2, zk_seed=3, **fri, allow_unpinned_w32_defaults=True)
    tp = p3r.TablePacking().with_fri_params(5, 2)
    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), tp)
    ctx.zk_nonce = 7
    assert pc.prove(wl.circuit_inputs_from_arrays(a)) == got
    L = layer_lib.OracleLayer(oracle, "koala-bear", a, layer_lib.params(zk=1, num_random_codewords=2, zk_seed=3, zk_nonce=7, **fri))
    assert np.array_equal(pc.circuit_prover_data.preprocessed_commitment, L.prep_commit())
    L.verify(got)
    assert L.prove() == got
    pc.free()
    ctx.close()


def test_fallback_paths_give_the_same_proof(tmp_path):
    """The tuning knobs exist in the `knobs` build of the library only (plonky3_recursion_amd/knobs/libp3r_hip.so,
    -DP3R_TUNING_KNOBS; the product build compiles them out).  The paths they select (copy-engine fetches instead
    of polled ones, one NTT launch per sub-transform size, 256-digest Merkle workgroups, 2^13-cell line tiles, the
    pre-round-2 NTT passes) must stay byte-identical to the product's proof; in the product build the same
    variables change nothing."""
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True)
    knobs_dir = os.path.join(ROOT, "plonky3_recursion_amd", "knobs")
    if not os.path.exists(os.path.join(knobs_dir, "libp3r_hip.so")):
        pytest.skip("knobs build of the library is absent (__graft_entry__.build() makes it)")

    def proof(name, knobs=True, **env):
        out_file = str(tmp_path / (name + ".bin"))
        e = {**os.environ, **env}
        if knobs:   # the example's RUNPATH finds the product library; LD_LIBRARY_PATH goes first
            e["LD_LIBRARY_PATH"] = knobs_dir + os.pathsep + e.get("LD_LIBRARY_PATH", "")
        r = subprocess.run([EXE, "koala-bear", "13", out_file, "1"], capture_output=True, text=True, timeout=300, env=e)
        assert r.returncode == 0, r.stdout + r.stderr
        return open(out_file, "rb").read()

    want = proof("product", knobs=False)
    assert proof("product_ignores_knobs", knobs=False, P3R_NTT_OLD="1", P3R_NO_POLLED_FETCH="1") == want
    assert proof("default") == want
    assert proof("unpolled", P3R_NO_POLLED_FETCH="1") == want
    assert proof("unmixed", P3R_NTT_NO_MIXED="1", P3R_NTT_LINE_LOG_TILE="13") == want
    assert proof("wide_subtrees", P3R_SUBTREE_NODES="256", P3R_COOP_MAX_NODES="32768",
                 P3R_COOP_MAX_LEAF_ROWS="32768") == want
    assert proof("old_ntt", P3R_NTT_OLD="1") == want


def test_both_width32_kernel_instances_give_the_same_proof(tmp_path):
    """The arity-4 MMCS launches the kernel instances that hold the built-in diagonal's lane forms at compile time when the
    configured diagonal is the built-in one (csrc/poseidon2_w32_f64.hip.h), the general instances (diagonal in scalar
    registers, six-instruction product) otherwise; the knobs build can force the general ones for the built-in diagonal
    (P3R_W32_GENERAL_DIAG): same proof bytes, large enough for the one-permutation-per-lane kernels to run."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    knobs_dir = os.path.join(ROOT, "plonky3_recursion_amd", "knobs")
    if not os.path.exists(os.path.join(knobs_dir, "libp3r_hip.so")):
        pytest.skip("knobs build of the library is absent (__graft_entry__.build() makes it)")
    got = {}
    for name, env in (("product", {}), ("knobs_builtin", {"LD_LIBRARY_PATH": knobs_dir}),
                      ("knobs_general", {"LD_LIBRARY_PATH": knobs_dir, "P3R_W32_GENERAL_DIAG": "1"})):
        out_file = str(tmp_path / (name + ".bin"))
        e = dict(os.environ)
        if "LD_LIBRARY_PATH" in env:
            e["LD_LIBRARY_PATH"] = env["LD_LIBRARY_PATH"] + os.pathsep + e.get("LD_LIBRARY_PATH", "")
        e.update({k: v for k, v in env.items() if k != "LD_LIBRARY_PATH"})
        for field, log_h in (("koala-bear", 16), ("baby-bear", 15)):
            r = subprocess.run([EXE, field, str(log_h), out_file, "1", "--arity4"], capture_output=True, text=True, timeout=300, env=e)
            assert r.returncode == 0, r.stdout + r.stderr
            got[(name, field)] = open(out_file, "rb").read()
    for field in ("koala-bear", "baby-bear"):
        assert got[("product", field)] == got[("knobs_builtin", field)] == got[("knobs_general", field)], field


def test_two_stream_commit_gives_the_same_proof(tmp_path):
    """The knobs build can hash one height class of each commit on a second stream while the main stream extends the next
    (prove_impl.hip.h::lde_and_commit, P3R_COMMIT_OVERLAP; measured, not the product's path: profiles/r05/
    commit_overlap_ab.txt): same bytes in every form, with and without ZK commitments."""
    import hashlib
    import sys
    knobs_lib = os.path.join(ROOT, "plonky3_recursion_amd", "knobs", "libp3r_hip.so")
    if not os.path.exists(knobs_lib):
        pytest.skip("knobs build of the library is absent (__graft_entry__.build() makes it)")
    script = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import harness_lib, harness_adapters as wl, plonky3_recursion_amd as p3r\n"
        "a = harness_lib.generate('koala-bear', 17, seed=5, horner_chain_len=64, sponge_chain_len=8, merkle_depth=20)\n"
        "for zk in (0, 1):\n"
        "    ctx = p3r.Context(field='koala-bear', zk=zk, zk_seed=3, allow_unpinned_w32_defaults=True)\n"
        "    pc = p3r.PreparedCircuit(ctx, wl.circuit_from_arrays(a), p3r.TablePacking().with_fri_params(5, 2))\n"
        "    print(hashlib.sha256(pc.prove(wl.circuit_inputs_from_arrays(a))).hexdigest())\n"
        "    pc.free(); ctx.close()\n" % (ROOT, os.path.join(ROOT, "tests")))

    def digests(**env):
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env={**os.environ, **env})
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout.split()

    want = digests()
    assert len(want) == 2 and want[0] != want[1]
    assert digests(P3R_LIB_PATH=knobs_lib) == want
    assert digests(P3R_LIB_PATH=knobs_lib, P3R_COMMIT_OVERLAP="1") == want
    assert digests(P3R_LIB_PATH=knobs_lib, P3R_COMMIT_OVERLAP="1", P3R_COMMIT_OVERLAP_MODE="2") == want
