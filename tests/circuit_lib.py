"""Oracle-side circuit: preprocessing (generate_preprocessed_columns + get_airs_and_degrees_with_prep)
and the sequential CircuitRunner of oracle/circuit.hpp, through ctypes."""
import ctypes as C

import numpy as np

import oracle_lib

u32p = C.POINTER(C.c_uint32)
NO_W = 0xFFFFFFFF
(OP_CONST, OP_PUBLIC, OP_ADD, OP_MUL, OP_BOOL, OP_MULADD, OP_HORNER, OP_HINT_EXT, OP_HINT_BIN, OP_P2,
 OP_RECOMPOSE, OP_P2W) = range(12)

PREP_ARRAYS = ["const_prep", "public_prep", "alu_prep13", "recompose_prep", "recompose_coeff_prep", "p2_in_ctl", "p2_input_indices",
               "p2_out_ctl", "p2_output_indices", "p2_mmcs_index_sum_idx"]
RUN_ARRAYS = ["const_values", "public_values", "alu_values", "p2_inputs", "p2_mmcs_index_sum", "recompose_values",
              "recompose_coeff_values"]


class OrcCircuitDesc(C.Structure):
    _fields_ = [("witness_count", C.c_uint32), ("n_ops", C.c_size_t), ("ops", u32p), ("n_ext", C.c_size_t),
                ("ext", u32p), ("n_public", C.c_size_t), ("public_rows", u32p), ("n_private", C.c_size_t),
                ("private_input_rows", u32p), ("n_rewrite", C.c_size_t), ("witness_rewrite", u32p)]


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint32).reshape(-1))


class Circuit:
    """Flattened Circuit<EF> (circuit/src/circuit.rs:152-181): `ops` is (n, 8) uint32 rows
    [kind, a, b, c, out, aux, ext_off, ext_len] as in include/p3r.h."""

    def __init__(self, witness_count, ops, ext=(), public_rows=(), private_rows=(), rewrite=()):
        self.witness_count = int(witness_count)
        self.ops = _arr(ops).reshape(-1, 8)
        self.ext, self.public_rows = _arr(ext), _arr(public_rows)
        self.private_rows, self.rewrite = _arr(private_rows), _arr(rewrite)

    @classmethod
    def from_arrays(cls, a):
        return cls(int(a["counts"][5]), a["ops"], a["ext"], a["public_rows"], a["private_rows"], a["rewrite"])


class Inputs:
    def __init__(self, public_values=(), private_values=(), pd_op_ids=(), pd_siblings=(), pdw_op_ids=(), pdw_siblings=()):
        self.public_values, self.private_values = _arr(public_values), _arr(private_values)
        self.pd_op_ids, self.pd_siblings = _arr(pd_op_ids), _arr(pd_siblings)
        self.pdw_op_ids, self.pdw_siblings = _arr(pdw_op_ids), _arr(pdw_siblings)   # width-32 Merkle rows: 24 values per id

    @classmethod
    def from_arrays(cls, a):
        return cls(a["in_public_values"], a["in_private_values"], a["pd_op_ids"], a["pd_siblings"],
                   a.get("pdw_op_ids", ()), a.get("pdw_siblings", ()))


class OracleCircuit:
    def __init__(self, orc, circuit: Circuit):
        self.orc, self.circuit = orc, circuit
        lib = orc.lib
        lib.orc_circuit_new.restype = C.c_void_p
        lib.orc_circuit_new.argtypes = [C.POINTER(OrcCircuitDesc)]
        lib.orc_circuit_free.argtypes = [C.c_void_p]
        lib.orc_circuit_preprocess.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        lib.orc_circuit_run.argtypes = [C.c_void_p, C.c_int, u32p, u32p, u32p, C.c_size_t, u32p, u32p]
        lib.orc_circuit_run_w32.argtypes = [C.c_void_p, C.c_int, u32p, u32p, u32p, u32p, u32p, C.c_size_t, u32p, u32p, C.c_size_t, u32p, u32p]
        lib.orc_circuit_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(u32p), C.POINTER(C.c_size_t)]
        d = OrcCircuitDesc()
        c = circuit
        d.witness_count = c.witness_count
        d.n_ops, d.ops = len(c.ops), c.ops.ctypes.data_as(u32p)
        d.n_ext, d.ext = len(c.ext), c.ext.ctypes.data_as(u32p)
        d.n_public, d.public_rows = len(c.public_rows), c.public_rows.ctypes.data_as(u32p)
        d.n_private, d.private_input_rows = len(c.private_rows), c.private_rows.ctypes.data_as(u32p)
        d.n_rewrite, d.witness_rewrite = len(c.rewrite) // 2, c.rewrite.ctypes.data_as(u32p)
        self.h = lib.orc_circuit_new(C.byref(d))

    def get(self, name):
        p, n = u32p(), C.c_size_t()
        if self.orc.lib.orc_circuit_get(self.h, name.encode(), C.byref(p), C.byref(n)) != 0:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)

    def preprocess(self, modulus, d=4):
        self.orc._ck(self.orc.lib.orc_circuit_preprocess(self.h, modulus, d))
        return self

    def run(self, field, inputs: Inputs, rc=None, w32=None):
        rc = oracle_lib.default_rc(field) if rc is None else np.ascontiguousarray(rc, dtype=np.uint32)
        i = inputs
        if (self.circuit.ops[:, 0] == OP_P2W).any():
            w32 = oracle_lib.default_w32(field) if w32 is None else w32
            wrc, wdiag = (np.ascontiguousarray(x, dtype=np.uint32) for x in w32)
            self.orc._ck(self.orc.lib.orc_circuit_run_w32(
                self.h, oracle_lib.FIELD_IDS[field], rc.ctypes.data_as(u32p), wrc.ctypes.data_as(u32p), wdiag.ctypes.data_as(u32p),
                i.public_values.ctypes.data_as(u32p), i.private_values.ctypes.data_as(u32p), len(i.pd_op_ids),
                i.pd_op_ids.ctypes.data_as(u32p), i.pd_siblings.ctypes.data_as(u32p), len(i.pdw_op_ids),
                i.pdw_op_ids.ctypes.data_as(u32p), i.pdw_siblings.ctypes.data_as(u32p)))
            return self
        self.orc._ck(self.orc.lib.orc_circuit_run(
            self.h, oracle_lib.FIELD_IDS[field], rc.ctypes.data_as(u32p), i.public_values.ctypes.data_as(u32p),
            i.private_values.ctypes.data_as(u32p), len(i.pd_op_ids), i.pd_op_ids.ctypes.data_as(u32p),
            i.pd_siblings.ctypes.data_as(u32p)))
        return self

    def workload_arrays(self):
        """The arrays prove_all_tables consumes (same names as harness_lib.ARRAYS), from the
        preprocessing + the run."""
        out = {k: self.get(k) for k in PREP_ARRAYS + RUN_ARRAYS}
        pf = self.get("p2_prep_flags").reshape(-1, 3)   # new_start, merkle_path, mmcs_ctl_enabled (&& merkle)
        rf = self.get("p2_flags").reshape(-1, 4)        # new_start, merkle_path, mmcs_bit, mmcs_ctl_enabled
        assert np.array_equal(pf[:, :2], rf[:, :2])
        out["p2_flags"] = rf.reshape(-1)
        for k in ("p2w_inputs", "p2w_flags", "p2w_mmcs_index_sum", "p2w_prep"):   # the width-32 table (OP_P2W ops)
            out[k] = self.get(k)
        n = [len(out["const_values"]) // 4, len(out["public_values"]) // 4, len(out["alu_values"]) // 16,
             len(rf), len(out["recompose_values"]) // 4, self.circuit.witness_count, len(out["recompose_coeff_values"]) // 4,
             len(out["p2w_mmcs_index_sum"])]
        out["counts"] = np.array(n, dtype=np.uint32)
        return out

    def __del__(self):
        try:
            self.orc.lib.orc_circuit_free(self.h)
        except Exception:
            pass
