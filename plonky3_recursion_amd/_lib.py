"""ctypes loader for the C-ABI shared library (include/p3r.h).

The product has no CPU fallback: if the HIP library is missing this module raises, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# P3R_LIB_PATH: A/B runs of two builds of the library on one box (development only)
LIB_PATH = os.environ.get("P3R_LIB_PATH") or os.path.join(_HERE, "libp3r_hip.so")

P3R_ABI_VERSION = 8
P3R_EXT_LOOKUP_UNPACKED = 1
P3R_EXT_UNPINNED_W32_DEFAULTS = 2   # the built-in width-32 constants are self-generated: using them is an explicit choice (p3r.h)
P3R_EXT_ZK_DETERMINISTIC = 4        # ZK key taken as it is, p3r_zk_set_nonce allowed: reproducible proofs (tests, replay)
FIELD_KOALA_BEAR = 0
FIELD_BABY_BEAR = 1


class P3rConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("field", C.c_uint32),
        ("ext_degree", C.c_uint32),
        ("log_blowup", C.c_uint32),
        ("max_log_arity", C.c_uint32),
        ("cap_height", C.c_uint32),
        ("log_final_poly_len", C.c_uint32),
        ("commit_pow_bits", C.c_uint32),
        ("query_pow_bits", C.c_uint32),
        ("num_queries", C.c_uint32),
        ("device", C.c_int32),
        ("poseidon2_rc", C.POINTER(C.c_uint32)),
        ("poseidon2_rc_len", C.c_uint32),
        ("ext_choices", C.c_uint32),
        ("fri_log_arities", C.POINTER(C.c_uint8)),
        ("fri_log_arities_len", C.c_uint32),
        ("proof_layout", C.POINTER(C.c_uint8)),
        ("proof_layout_len", C.c_uint32), ("ext_w", C.c_uint32), ("challenge_degree", C.c_uint32),
        # ABI 6: constants of the width-32 permutation (NULL = the self-generated defaults)
        ("poseidon2_w32_rc", C.POINTER(C.c_uint32)), ("poseidon2_w32_rc_len", C.c_uint32),
        ("poseidon2_w32_diag", C.POINTER(C.c_uint32)),
        ("mmcs_arity", C.c_uint32),   # 0 / 2: binary MMCS over the width-16 permutation; 4: the arity-4 MMCS (width 32)
        # ABI 7: ZK = HidingFriPcs (create_config_zk, recursion/examples/common/mod.rs:511-553)
        ("zk", C.c_uint32), ("num_random_codewords", C.c_uint32), ("zk_key", C.c_uint32 * 8),
        ("mmcs_salt_elems", C.c_uint32),   # MerkleTreeHidingMmcs: salt elements per committed row (0: plain MMCS)
    ]


class P3rP2Rows(C.Structure):
    _fields_ = [
        ("n", C.c_size_t),
        ("input_values", C.POINTER(C.c_uint32)),
        ("new_start", C.POINTER(C.c_uint8)),
        ("merkle_path", C.POINTER(C.c_uint8)),
        ("mmcs_bit", C.POINTER(C.c_uint8)),
        ("mmcs_index_sum", C.POINTER(C.c_uint32)),
    ]


class P3rP2wRows(C.Structure):   # rows of the width-32 Poseidon2 table (ABI 6)
    _fields_ = [
        ("n", C.c_size_t),
        ("input_values", C.POINTER(C.c_uint32)),
        ("new_start", C.POINTER(C.c_uint8)),
        ("merkle_path", C.POINTER(C.c_uint8)),
        ("mmcs_bit", C.POINTER(C.c_uint8)),
        ("mmcs_bit2", C.POINTER(C.c_uint8)),
        ("mmcs_index_sum", C.POINTER(C.c_uint32)),
    ]


class P3rMatrix(C.Structure):
    _fields_ = [("values", C.POINTER(C.c_uint32)), ("height", C.c_size_t), ("width", C.c_size_t)]


class P3rAirDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("lanes", C.c_uint32), ("horner_packed_steps", C.c_uint32),
                ("coeff_lookups", C.c_uint32)]


class P3rLayerCounts(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("n_const", "n_public", "n_alu", "n_p2", "n_recompose", "n_recompose_coeff", "n_p2w")]


class P3rLayerDesc(C.Structure):
    _fields_ = [
        ("counts", P3rLayerCounts),
        ("public_lanes", C.c_uint32), ("alu_lanes", C.c_uint32), ("horner_packed_steps", C.c_uint32),
        ("recompose_lanes", C.c_uint32), ("min_trace_height", C.c_uint32),
        ("const_prep", C.POINTER(C.c_uint32)), ("public_prep", C.POINTER(C.c_uint32)),
        ("alu_prep13", C.POINTER(C.c_uint32)), ("recompose_prep", C.POINTER(C.c_uint32)),
        ("p2_new_start", C.POINTER(C.c_uint8)), ("p2_merkle_path", C.POINTER(C.c_uint8)),
        ("p2_mmcs_ctl_enabled", C.POINTER(C.c_uint8)), ("p2_in_ctl", C.POINTER(C.c_uint8)),
        ("p2_input_indices", C.POINTER(C.c_uint32)), ("p2_out_ctl", C.POINTER(C.c_uint32)),
        ("p2_output_indices", C.POINTER(C.c_uint32)), ("p2_mmcs_index_sum_idx", C.POINTER(C.c_uint32)),
        ("p2_absorb_len", C.POINTER(C.c_uint8)), ("recompose_coeff_lookups", C.c_uint32),
        ("recompose_coeff_prep", C.POINTER(C.c_uint32)),
        ("p2w_prep", C.POINTER(C.c_uint32)),
    ]


class P3rTraces(C.Structure):
    _fields_ = [
        ("n_const", C.c_size_t), ("const_values", C.POINTER(C.c_uint32)),
        ("n_public", C.c_size_t), ("public_values", C.POINTER(C.c_uint32)),
        ("n_alu", C.c_size_t), ("alu_values", C.POINTER(C.c_uint32)),
        ("p2", P3rP2Rows),
        ("n_recompose", C.c_size_t), ("recompose_values", C.POINTER(C.c_uint32)),
        ("n_recompose_coeff", C.c_size_t), ("recompose_coeff_values", C.POINTER(C.c_uint32)),
        ("p2w", P3rP2wRows),
    ]


class P3rOp(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("kind", "a", "b", "c", "out", "aux", "ext_off", "ext_len")]


class P3rCircuitDesc(C.Structure):
    _fields_ = [
        ("witness_count", C.c_uint32),
        ("n_ops", C.c_size_t), ("ops", C.POINTER(P3rOp)),
        ("n_ext", C.c_size_t), ("ext", C.POINTER(C.c_uint32)),
        ("n_public", C.c_size_t), ("public_rows", C.POINTER(C.c_uint32)),
        ("n_private", C.c_size_t), ("private_input_rows", C.POINTER(C.c_uint32)),
        ("n_rewrite", C.c_size_t), ("witness_rewrite", C.POINTER(C.c_uint32)),
        ("public_lanes", C.c_uint32), ("alu_lanes", C.c_uint32), ("horner_packed_steps", C.c_uint32),
        ("recompose_lanes", C.c_uint32), ("min_trace_height", C.c_uint32),
    ]


class P3rCircuitInputs(C.Structure):
    _fields_ = [
        ("public_values", C.POINTER(C.c_uint32)), ("private_values", C.POINTER(C.c_uint32)),
        ("n_private_data", C.c_size_t), ("private_data_op_ids", C.POINTER(C.c_uint32)),
        ("private_data_siblings", C.POINTER(C.c_uint32)),
        # ABI 8: private data of the width-32 Merkle rows (three sibling digests = 24 values per op id)
        ("n_private_data_w32", C.c_size_t), ("private_data_w32_op_ids", C.POINTER(C.c_uint32)),
        ("private_data_w32_siblings", C.POINTER(C.c_uint32)),
    ]


class P3rNpoTableEntry(C.Structure):
    _fields_ = [("op_type", C.c_char * 64), ("rows", C.c_uint64), ("lanes", C.c_uint32), ("air_variant", C.c_uint32),
                ("n_public_values", C.c_uint32), ("public_values", C.c_uint32 * 8)]


class P3rNpoLanes(C.Structure):
    _fields_ = [("op_type", C.c_char * 64), ("lanes", C.c_uint32)]


class P3rBatchStarkMeta(C.Structure):
    _fields_ = [
        ("proof_len", C.c_uint64), ("parse_ns", C.c_uint64),
        ("public_lanes", C.c_uint32), ("alu_lanes", C.c_uint32), ("min_trace_height", C.c_uint32),
        ("horner_packed_steps", C.c_uint32),
        ("n_npo_lanes", C.c_uint32), ("npo_lanes", P3rNpoLanes * 8),
        ("rows", C.c_uint64 * 3),
        ("alu_variant", C.c_uint32), ("ext_degree", C.c_uint32),
        ("has_w_binomial", C.c_uint32), ("w_binomial", C.c_uint32), ("alu_quintic_trinomial", C.c_uint32),
        ("n_non_primitives", C.c_uint32), ("non_primitives", P3rNpoTableEntry * 8),
        ("has_stark_common", C.c_uint32), ("cap_len", C.c_uint32), ("commitment", C.c_uint32 * (8 * 64)),
        ("n_instances", C.c_uint32), ("preprocessed_widths", C.c_uint32 * 16), ("degree_bits", C.c_uint32 * 16),
    ]


class P3rProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("total_ms", C.c_double), ("launches", C.c_uint64)]


u32p = C.POINTER(C.c_uint32)
vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/p3r.h declares must appear here
# (tests/test_abi.py checks header <-> table <-> .so agreement).
SIGNATURES = {
    "p3r_create": (vp, [C.POINTER(P3rConfig)]),
    "p3r_destroy": (None, [vp]),
    "p3r_last_error": (C.c_char_p, [vp]),
    "p3r_poseidon2_trace_width": (C.c_uint32, [vp]),
    "p3r_poseidon2_num_constants": (C.c_uint32, [vp]),
    "p3r_poseidon2_round_constants": (C.c_int, [vp, u32p]),
    "p3r_sync": (C.c_int, [vp]),
    "p3r_zk_nonce": (C.c_uint64, [vp]),
    "p3r_zk_set_nonce": (C.c_int, [vp, C.c_uint64]),
    "p3r_trim": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "p3r_dmat_upload": (vp, [vp, u32p, C.c_size_t, C.c_size_t]),
    "p3r_dmat_alloc": (vp, [vp, C.c_size_t, C.c_size_t]),
    "p3r_dmat_download": (C.c_int, [vp, vp, u32p]),
    "p3r_dmat_height": (C.c_size_t, [vp]),
    "p3r_dmat_width": (C.c_size_t, [vp]),
    "p3r_dmat_free": (None, [vp, vp]),
    "p3r_poseidon2_permute_batch": (C.c_int, [vp, u32p, u32p, C.c_size_t]),
    "p3r_poseidon2_permute_dmat": (C.c_int, [vp, vp]),
    "p3r_poseidon2_trace_fill": (C.c_int, [vp, C.POINTER(P3rP2Rows), u32p]),
    "p3r_poseidon2_trace_fill_dmat": (vp, [vp, C.POINTER(P3rP2Rows)]),
    "p3r_p2_rows_upload": (vp, [vp, C.POINTER(P3rP2Rows)]),
    "p3r_p2_rows_free": (None, [vp, vp]),
    "p3r_poseidon2_trace_fill_dev": (vp, [vp, vp]),
    "p3r_coset_lde": (C.c_int, [vp, u32p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, u32p]),
    "p3r_coset_lde_dmat": (vp, [vp, vp, C.c_uint32, C.c_uint32]),
    "p3r_mmcs_commit": (C.c_int, [vp, C.POINTER(P3rMatrix), C.c_size_t, u32p, C.POINTER(vp)]),
    "p3r_mmcs_commit_dmat": (C.c_int, [vp, C.POINTER(vp), C.c_size_t, u32p, C.POINTER(vp)]),
    "p3r_mmcs_open": (C.c_int, [vp, vp, C.c_size_t, u32p, u32p]),
    "p3r_tree_log_max_height": (C.c_size_t, [vp]),
    "p3r_tree_total_width": (C.c_size_t, [vp]),
    "p3r_tree_proof_len": (C.c_size_t, [vp]),
    "p3r_mmcs_verify": (C.c_int, [C.POINTER(P3rConfig), u32p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t,
                                  u32p, u32p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "p3r_mmcs_verify_salted": (C.c_int, [C.POINTER(P3rConfig), u32p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t,
                                         u32p, u32p, u32p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "p3r_tree_free": (None, [vp, vp]),
    "p3r_prep_create": (vp, [vp, C.POINTER(P3rAirDesc), C.POINTER(P3rMatrix), C.c_size_t, u32p]),
    "p3r_prep_free": (None, [vp, vp]),
    "p3r_prove_batch": (C.c_int, [vp, vp, C.POINTER(vp), C.c_size_t, C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t,
                                  C.POINTER(C.c_size_t)]),
    "p3r_prove_batch_host": (C.c_int, [vp, vp, C.POINTER(P3rMatrix), C.c_size_t, C.c_uint32, C.POINTER(C.c_uint8),
                                       C.c_size_t, C.POINTER(C.c_size_t)]),
    "p3r_layer_create": (vp, [vp, C.POINTER(P3rLayerDesc), u32p]),
    "p3r_layer_free": (None, [vp, vp]),
    "p3r_layer_table_heights": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "p3r_layer_recompose_coeff_height": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "p3r_layer_p2w_height": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "p3r_poseidon2_w32_permute_batch": (C.c_int, [vp, u32p, u32p, C.c_size_t]),
    "p3r_poseidon2_w32_trace_fill": (C.c_int, [vp, C.POINTER(P3rP2wRows), u32p]),
    "p3r_poseidon2_w32_trace_width": (C.c_uint32, [vp]),
    "p3r_layer_recompose_kind": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
    "p3r_layer_effective_lanes": (C.c_int, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "p3r_verify_batch": (C.c_int, [C.POINTER(P3rConfig), C.POINTER(P3rAirDesc), C.c_size_t, u32p, u32p,
                                   C.POINTER(C.c_uint8), C.c_size_t, C.c_uint32, C.c_char_p, C.c_size_t]),
    "p3r_batch_proof_len": (C.c_int, [C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint32, C.POINTER(C.c_size_t),
                                      C.c_char_p, C.c_size_t]),
    "p3r_batch_proof_len_layout": (C.c_int, [C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint32, C.POINTER(C.c_uint8),
                                             C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]),
    "p3r_batch_stark_proof_parse": (C.c_int, [C.c_uint32, C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint8),
                                              C.POINTER(P3rBatchStarkMeta), C.c_char_p, C.c_size_t]),
    "p3r_circuit_create": (vp, [vp, C.POINTER(P3rCircuitDesc), u32p]),
    "p3r_circuit_free": (None, [vp, vp]),
    "p3r_circuit_layer": (vp, [vp]),
    "p3r_circuit_counts": (C.c_int, [vp, C.POINTER(P3rLayerCounts)]),
    "p3r_circuit_levels": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "p3r_circuit_prepared_on_device": (C.c_int, [vp]),
    "p3r_circuit_run": (vp, [vp, vp, C.POINTER(P3rCircuitInputs)]),
    "p3r_prove_next_layer": (C.c_int, [vp, vp, C.POINTER(P3rCircuitInputs), C.c_uint32, C.POINTER(C.c_uint8),
                                       C.c_size_t, C.POINTER(C.c_size_t)]),
    "p3r_circuit_inputs_upload": (vp, [vp, vp, C.POINTER(P3rCircuitInputs)]),
    "p3r_circuit_inputs_free": (None, [vp, vp]),
    "p3r_circuit_run_resident": (vp, [vp, vp, vp]),
    "p3r_take_proof": (C.c_int, [vp, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t)]),
    "p3r_prove_next_layer_resident": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t,
                                                C.POINTER(C.c_size_t)]),
    "p3r_dtraces_get": (C.c_int, [vp, vp, vp, C.c_uint32, u32p, C.c_size_t]),
    "p3r_traces_upload": (vp, [vp, vp, C.POINTER(P3rTraces)]),
    "p3r_traces_free": (None, [vp, vp]),
    "p3r_prove_all_tables": (C.c_int, [vp, vp, C.POINTER(P3rTraces), C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t,
                                       C.POINTER(C.c_size_t)]),
    "p3r_prove_all_tables_resident": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t,
                                                C.POINTER(C.c_size_t)]),
    "p3r_layer_build_main_trace": (vp, [vp, vp, vp, C.c_uint32]),
    "p3r_time_permute_dmat": (C.c_int, [vp, vp, C.c_int, C.POINTER(C.c_double)]),
    "p3r_profile_enable": (C.c_int, [vp, C.c_int]),
    "p3r_profile_read": (C.c_int, [vp, C.POINTER(P3rProfileEntry), C.c_size_t, C.POINTER(C.c_size_t)]),
}

_lib = None


def load():
    """Load libp3r_hip.so (built by __graft_entry__.build()). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "plonky3_recursion_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
