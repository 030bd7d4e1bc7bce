"""Every literal of the reference's layout goldens (circuit-prover/src/air/shape_golden.rs:32-68 - "the only literal
goldens in the repo"): (main_width, preprocessed_width) of the Const, Public and ALU AIRs per extension
degree and lane count, checked on the tables the oracle builds for a synthetic layer of that degree.  CPU only; the GPU
suite compares the device's tables with the oracle's cell by cell."""
import pytest

import harness_lib
import layer_lib
import oracle_lib

# shape_golden.rs: const_air_shape_is_stable :32-37, public_air_shape_is_stable :39-44, alu_base_air_shape_is_stable :46-52,
# alu_binomial_air_shape_is_stable :54-62, alu_quintic_trinomial_air_shape_is_stable :64-68 (AluAir::new(0, lanes):
# the default horner pack of 2)
CONST = {1: (1, 2), 4: (4, 2), 5: (5, 2)}
PUBLIC = {(1, 1): (1, 2), (4, 1): (4, 2), (4, 2): (8, 4)}
ALU = {(1, 1): (7, 20), (1, 2): (11, 33), (2, 1): (14, 20), (4, 1): (28, 20), (4, 2): (44, 33), (5, 1): (35, 20)}


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.Oracle()


def tables(oracle, d, alu_lanes=1, public_lanes=1):
    flags = harness_lib.NO_POSEIDON2 | harness_lib.NO_RECOMPOSE
    arrs = harness_lib.generate("koala-bear", 5, seed=2, flags=flags, ext_degree=d, horner_chain_len=6, sponge_chain_len=2,
                                merkle_depth=3)
    prm = layer_lib.params(log_blowup=1, max_log_arity=1, log_final_poly_len=0, query_pow_bits=1, num_queries=2)
    packing = dict(alu_lanes=alu_lanes, public_lanes=public_lanes, horner_packed_steps=2, ext_degree=d)
    if d == 2:
        packing["ext_w"] = 3
    L = layer_lib.OracleLayer(oracle, "koala-bear", arrs, prm, packing=packing)
    return {t["kind"]: (t["main"].shape[1], t["prep"].shape[1]) for t in L.tables()}


@pytest.mark.parametrize("d", sorted(CONST))
def test_const_air_shape(oracle, d):
    assert tables(oracle, d)["const"] == CONST[d]


@pytest.mark.parametrize("d,lanes", sorted(PUBLIC))
def test_public_air_shape(oracle, d, lanes):
    assert tables(oracle, d, public_lanes=lanes)["public"] == PUBLIC[(d, lanes)]


@pytest.mark.parametrize("d,lanes", sorted(ALU))
def test_alu_air_shape(oracle, d, lanes):
    assert tables(oracle, d, alu_lanes=lanes)["alu"] == ALU[(d, lanes)]
