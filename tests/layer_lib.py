"""Oracle-side recursion layer: table building, prove_batch, verify_batch (ctypes)."""
import ctypes as C

import numpy as np

import oracle_lib

u32p = C.POINTER(C.c_uint32)
KINDS = ["const", "public", "alu", "poseidon2", "recompose", "poseidon2_w32"]


class OrcWorkload(C.Structure):
    _fields_ = [
        ("n_const", C.c_size_t), ("const_values", u32p), ("const_prep", u32p),
        ("n_public", C.c_size_t), ("public_values", u32p), ("public_prep", u32p),
        ("n_alu", C.c_size_t), ("alu_values", u32p), ("alu_prep13", u32p),
        ("n_p2", C.c_size_t), ("p2_inputs", u32p), ("p2_flags", u32p), ("p2_mmcs_index_sum", u32p),
        ("p2_in_ctl", u32p), ("p2_input_indices", u32p), ("p2_out_ctl", u32p), ("p2_output_indices", u32p),
        ("p2_mmcs_index_sum_idx", u32p),
        ("n_recompose", C.c_size_t), ("recompose_values", u32p), ("recompose_prep", u32p),
        ("public_lanes", C.c_uint32), ("alu_lanes", C.c_uint32), ("horner_packed_steps", C.c_uint32),
        ("recompose_lanes", C.c_uint32), ("min_trace_height", C.c_uint32), ("ext_degree", C.c_uint32),
        ("p2_absorb_len", u32p), ("recompose_coeff_lookups", C.c_uint32), ("ext_w", C.c_uint32),
        ("n_recompose_coeff", C.c_size_t), ("recompose_coeff_values", u32p), ("recompose_coeff_prep", u32p),
        # the width-32 Poseidon2 table (arity-4 MMCS rows) and the constants of its permutation
        ("n_p2w", C.c_size_t), ("p2w_inputs", u32p), ("p2w_flags", u32p), ("p2w_mmcs_index_sum", u32p), ("p2w_prep", u32p),
        ("w32_rc", u32p), ("w32_diag", u32p),
    ]


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("log_blowup", "max_log_arity", "cap_height", "log_final_poly_len",
                                          "commit_pow_bits", "query_pow_bits", "num_queries", "ext_choices",
                                          "n_fri_log_arities")] + [("fri_log_arities", C.c_uint8 * 32),
                                                                   ("proof_layout", C.c_uint8 * 18),
                                                                   ("challenge_degree", C.c_uint32),
                                                                   ("mmcs_arity", C.c_uint32),
                                                                   # ZK (HidingFriPcs): twins of p3r_config.zk*
                                                                   ("zk", C.c_uint32), ("num_random_codewords", C.c_uint32),
                                                                   ("zk_key", C.c_uint32 * 8), ("zk_nonce", C.c_uint64),
                                                                   # forced proof-of-work witnesses (tools/resolve_pins.py)
                                                                   ("n_forced_pow", C.c_uint32), ("forced_pow", C.c_uint32 * 40),
                                                                   # MerkleTreeHidingMmcs: salt elements per committed row (0: plain MMCS)
                                                                   ("mmcs_salt_elems", C.c_uint32)]


def params(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0,
           query_pow_bits=15, num_queries=54, ext_choices=0, fri_log_arities=None, proof_layout=None, challenge_degree=4,
           mmcs_arity=2, zk=0, num_random_codewords=2, zk_seed=0, zk_nonce=0, forced_pow=None, zk_key=None,
           mmcs_salt_elems=0):
    p = OrcParams(log_blowup, max_log_arity, cap_height, log_final_poly_len, commit_pow_bits, query_pow_bits,
                  num_queries, ext_choices, 0)
    p.challenge_degree = challenge_degree
    p.mmcs_arity = mmcs_arity   # 4: the arity-4 MMCS over the width-32 permutation
    # create_config_zk (recursion/examples/common/mod.rs:511-553): HidingFriPcs, two random codewords, a seeded RNG;
    # zk_nonce = proofs already made under the configuration (its RNG state)
    # zk_seed: the harness's shorthand for the key (seed, 0, ..) - the same mapping as plonky3_recursion_amd.make_config
    p.zk, p.num_random_codewords, p.zk_nonce = zk, num_random_codewords, zk_nonce
    p.mmcs_salt_elems = mmcs_salt_elems
    key = list(zk_key) if zk_key is not None else [int(zk_seed) & 0xFFFFFFFF, (int(zk_seed) >> 32) & 0xFFFFFFFF, 0, 0, 0, 0, 0, 0]
    for i in range(8):
        p.zk_key[i] = int(key[i])
    if forced_pow:
        p.n_forced_pow = len(forced_pow)
        for i, w in enumerate(forced_pow):
            p.forced_pow[i] = int(w)
    if fri_log_arities is not None:
        p.n_fri_log_arities = len(fri_log_arities)
        for i, la in enumerate(fri_log_arities):
            p.fri_log_arities[i] = la
    if proof_layout is not None:
        assert len(proof_layout) == 18
        for i, v in enumerate(proof_layout):
            p.proof_layout[i] = v
    return p


def min_trace_height(p):
    # TablePacking::with_fri_params (packing.rs:100-106)
    return 1 << (p.log_final_poly_len + p.log_blowup + 1)


def fill_workload(wl_struct, arrays, packing, keep):
    def ptr(name):
        a = np.ascontiguousarray(arrays[name], dtype=np.uint32)
        keep.append(a)
        return a.ctypes.data_as(u32p)
    c = arrays["counts"]
    wl_struct.n_const, wl_struct.n_public, wl_struct.n_alu, wl_struct.n_p2, wl_struct.n_recompose = (int(x) for x in c[:5])
    for name in ("const_values", "const_prep", "public_values", "public_prep", "alu_values", "alu_prep13",
                 "p2_inputs", "p2_flags", "p2_mmcs_index_sum", "p2_in_ctl", "p2_input_indices", "p2_out_ctl",
                 "p2_output_indices", "p2_mmcs_index_sum_idx", "recompose_values", "recompose_prep"):
        setattr(wl_struct, name, ptr(name))
    wl_struct.public_lanes = packing.get("public_lanes", 1)
    wl_struct.alu_lanes = packing.get("alu_lanes", 3)
    wl_struct.horner_packed_steps = packing.get("horner_packed_steps", 4)
    wl_struct.recompose_lanes = packing.get("recompose_lanes", 1)
    wl_struct.min_trace_height = packing["min_trace_height"]
    wl_struct.ext_degree = packing.get("ext_degree", 4)   # 5: KoalaBear quintic circuits (compact-D1 Poseidon2 rows)
    wl_struct.recompose_coeff_lookups = packing.get("recompose_coeff_lookups", 0)
    wl_struct.ext_w = packing.get("ext_w", 0)
    if "p2_absorb_len" in arrays and len(arrays["p2_absorb_len"]):
        wl_struct.p2_absorb_len = ptr("p2_absorb_len")
    # the constants of the width-32 permutation: its table's (below) and the arity-4 MMCS's (params(mmcs_arity=4))
    w32 = packing.get("w32") or oracle_lib.default_w32(packing["field"])
    keep.extend(w32)
    wl_struct.w32_rc, wl_struct.w32_diag = w32[0].ctypes.data_as(u32p), w32[1].ctypes.data_as(u32p)
    if len(c) > 7 and int(c[7]):   # rows of the width-32 Poseidon2 table (harness flag P2_W32)
        wl_struct.n_p2w = int(c[7])
        for name in ("p2w_inputs", "p2w_flags", "p2w_mmcs_index_sum", "p2w_prep"):
            setattr(wl_struct, name, ptr(name))
    # a layer with both Recompose tables (harness flag RECOMPOSE_BOTH): the second one is `recompose/coeff`
    if len(c) > 6 and int(c[6]):
        wl_struct.n_recompose_coeff = int(c[6])
        wl_struct.recompose_coeff_values = ptr("recompose_coeff_values")
        wl_struct.recompose_coeff_prep = ptr("recompose_coeff_prep")


def oracle_verify_statement(orc, field, prm, airs, prep_cap, proof_bytes, rc=None, field_encoding=0, w32=None):
    """The oracle verifier from the statement alone: `airs` = dicts(kind, lanes, horner_packed_steps,
    coeff_lookups) in instance order (what `BatchStarkProof.airs()` rebuilds from the proof metadata).
    For layers whose tables the CPU would take minutes to rebuild.  Raises RuntimeError on rejection."""
    lib = orc.lib
    lib.orc_verify_batch_w32.argtypes = [C.c_int, u32p, u32p, u32p, C.POINTER(OrcParams), C.c_size_t, u32p, u32p,
                                         C.POINTER(C.c_uint8), C.c_size_t, C.c_int]
    rc = oracle_lib.default_rc(field) if rc is None else np.ascontiguousarray(rc, dtype=np.uint32)
    w32 = oracle_lib.default_w32(field) if w32 is None else w32   # of a width-32 table / the arity-4 MMCS
    a4 = np.array([[a["kind"], a.get("lanes", 1), a.get("horner_packed_steps", 2),
                    a.get("coeff_lookups", 0) | (a.get("ext_degree", 4) << 8) | (a.get("ext_w", 0) << 16)] for a in airs], dtype=np.uint32)
    cap = np.ascontiguousarray(prep_cap, dtype=np.uint32)
    b = (C.c_uint8 * len(proof_bytes)).from_buffer_copy(proof_bytes)
    orc._ck(lib.orc_verify_batch_w32(oracle_lib.FIELD_IDS[field], rc.ctypes.data_as(u32p), w32[0].ctypes.data_as(u32p),
                                     w32[1].ctypes.data_as(u32p), C.byref(prm), len(airs), a4.ctypes.data_as(u32p),
                                     cap.ctypes.data_as(u32p), b, len(proof_bytes), field_encoding))


class OracleLayer:
    """The five table instances of one recursion layer + prover data, on the CPU oracle."""

    def __init__(self, orc, field, arrays, prm, packing=None, rc=None):
        self.orc, self.field, self.prm = orc, field, prm
        lib = orc.lib
        lib.orc_layer_build.restype = C.c_void_p
        lib.orc_layer_build.argtypes = [C.c_int, u32p, C.POINTER(OrcWorkload)]
        lib.orc_layer_free.argtypes = [C.c_void_p]
        lib.orc_layer_num_tables.restype = C.c_size_t
        lib.orc_layer_num_tables.argtypes = [C.c_void_p]
        lib.orc_layer_table_info.argtypes = [C.c_void_p, C.c_size_t, u32p]
        lib.orc_layer_get_matrix.argtypes = [C.c_void_p, C.c_size_t, C.c_int, u32p]
        lib.orc_layer_prep_commit.argtypes = [C.c_void_p, C.POINTER(OrcParams), u32p]
        lib.orc_layer_prove.argtypes = [C.c_void_p, C.POINTER(OrcParams), C.c_int, C.POINTER(C.POINTER(C.c_uint8)),
                                        C.POINTER(C.c_size_t)]
        lib.orc_bytes_free.argtypes = [C.POINTER(C.c_uint8)]
        lib.orc_layer_verify.argtypes = [C.c_void_p, C.POINTER(OrcParams), u32p, C.POINTER(C.c_uint8), C.c_size_t,
                                         C.c_int]
        packing = dict(packing or {})
        packing.setdefault("min_trace_height", min_trace_height(prm))
        packing["field"] = field
        self.packing = packing
        wl = OrcWorkload()
        self._keep = []
        fill_workload(wl, arrays, packing, self._keep)
        self.rc = oracle_lib.default_rc(field) if rc is None else np.ascontiguousarray(rc, dtype=np.uint32)
        self.h = lib.orc_layer_build(oracle_lib.FIELD_IDS[field], self.rc.ctypes.data_as(u32p), C.byref(wl))
        if not self.h:
            raise RuntimeError("oracle: " + lib.orc_last_error().decode())

    def tables(self):
        out = []
        lib = self.orc.lib
        for i in range(lib.orc_layer_num_tables(self.h)):
            info = (C.c_uint32 * 6)()
            self.orc._ck(lib.orc_layer_table_info(self.h, i, info))
            kind, lanes, k, h, w, pw = (int(x) for x in info)
            main = np.empty((h, w), dtype=np.uint32)
            prep = np.empty((h, pw), dtype=np.uint32)
            self.orc._ck(lib.orc_layer_get_matrix(self.h, i, 0, main.ctypes.data_as(u32p)))
            self.orc._ck(lib.orc_layer_get_matrix(self.h, i, 1, prep.ctypes.data_as(u32p)))
            out.append(dict(kind=KINDS[kind], kind_id=kind, lanes=lanes, horner_k=k, main=main, prep=prep))
        return out

    def prep_commit(self):
        cap = np.empty((1 << self.prm.cap_height, 8), dtype=np.uint32)
        self.orc._ck(self.orc.lib.orc_layer_prep_commit(self.h, C.byref(self.prm), cap.ctypes.data_as(u32p)))
        return cap

    def prove(self, field_encoding=0):
        buf = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        self.orc._ck(self.orc.lib.orc_layer_prove(self.h, C.byref(self.prm), field_encoding, C.byref(buf), C.byref(n)))
        try:
            return bytes(C.string_at(buf, n.value))
        finally:
            self.orc.lib.orc_bytes_free(buf)

    def verify(self, proof_bytes, prep_cap=None, field_encoding=0):
        """Raises RuntimeError with the verifier's reason if the proof is rejected."""
        cap = self.prep_commit() if prep_cap is None else np.ascontiguousarray(prep_cap, dtype=np.uint32)
        b = (C.c_uint8 * len(proof_bytes)).from_buffer_copy(proof_bytes)
        self.orc._ck(self.orc.lib.orc_layer_verify(self.h, C.byref(self.prm), cap.ctypes.data_as(u32p), b,
                                                    len(proof_bytes), field_encoding))

    def __del__(self):
        try:
            self.orc.lib.orc_layer_free(self.h)
        except Exception:
            pass
