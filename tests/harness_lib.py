"""ctypes view of the synthetic-workload generator (harness/libp3r_synth.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDIR = os.path.join(ROOT, "harness")
LIB = os.path.join(HDIR, "libp3r_synth.so")
FIELD_IDS = {"koala-bear": 0, "baby-bear": 1}
u32p = C.POINTER(C.c_uint32)

ARRAYS = ["const_values", "const_prep", "public_values", "public_prep", "alu_values", "alu_prep13",
          "p2_inputs", "p2_flags", "p2_mmcs_index_sum", "p2_in_ctl", "p2_input_indices", "p2_out_ctl",
          "p2_output_indices", "p2_mmcs_index_sum_idx", "recompose_values", "recompose_prep", "counts",
          # the circuit the arrays above were derived from (flattened Circuit<EF>, include/p3r.h) and its inputs
          "ops", "ext", "public_rows", "in_public_values", "private_rows", "in_private_values", "pd_op_ids",
          "pd_siblings", "rewrite", "p2_absorb_len", "recompose_coeff_values", "recompose_coeff_prep",
          # the width-32 Poseidon2 table (flag P2_W32; counts[7] rows): inputs n x 32, flags n x 4, index sums, assembled prep n x 48
          "p2w_inputs", "p2w_flags", "p2w_mmcs_index_sum", "p2w_prep",
          # P2_W32_OPS: private data of the width-32 Merkle rows (op ids; 24 values = three sibling digests each)
          "pdw_op_ids", "pdw_siblings"]


def build():
    src = os.path.join(HDIR, "synth.cpp")
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return
    subprocess.run(["make", "-C", HDIR], check=True, capture_output=True)


NO_POSEIDON2, NO_RECOMPOSE, SINGLE_PUBLIC, NO_ALU, INDEPENDENT_SPONGES, RECOMPOSE_COEFF = 1, 2, 4, 8, 16, 32
# both Recompose tables in one layer: each op is `recompose` or `recompose/coeff` (arrays recompose_* and recompose_coeff_*;
# counts[6] = rows of the second table) - what a backend with coefficient lookups registers (batch_stark_prover.rs:1914-1932)
RECOMPOSE_BOTH = 64
# a width-32 Poseidon2 table next to the width-16 one: arity-4 Merkle chains and rate-24 sponges (Poseidon2Config::*_D4_W32);
# D = 4 only, at the prove_all_tables boundary (arrays p2w_*, counts[7])
P2_W32 = 128
# the width-32 rows as ops of the circuit (P3R_OP_POSEIDON2_W32_PERM) with the executor's exact row semantics, leaf sponges
# that seed Merkle chains, and their private data (arrays pdw_*): what a verifier circuit built under `--arity4` holds
P2_W32_OPS = 4096 | P2_W32


def generate(field, log_h, seed=0x5EED0000, horner_chain_len=64, sponge_chain_len=6, merkle_depth=20, rc=None,
             flags=0, ext_degree=4):
    """Returns dict name -> np.uint32 array (see harness/synth.cpp).  ext_degree=5: KoalaBear circuits over the
    quintic trinomial extension (Poseidon2 rows are then the compact-D1 ones).  flags | RECOMPOSE_COEFF: the Recompose
    table is the "recompose/coeff" variant (recompose_prep is n x (2 + 2 D))."""
    build()
    lib = C.CDLL(LIB)
    lib.syn_generate.restype = C.c_void_p
    lib.syn_generate.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, u32p, C.c_uint32]
    lib.syn_error.restype = C.c_char_p
    lib.syn_error.argtypes = [C.c_void_p]
    lib.syn_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(u32p), C.POINTER(C.c_size_t)]
    lib.syn_free.argtypes = [C.c_void_p]
    if rc is None:
        import oracle_lib
        rc = oracle_lib.default_rc(field)
    rc = np.ascontiguousarray(rc, dtype=np.uint32)
    if flags & P2_W32:
        import oracle_lib
        w32 = [np.ascontiguousarray(a, dtype=np.uint32) for a in oracle_lib.default_w32(field)]
        lib.syn_set_w32.argtypes = [u32p, u32p]
        lib.syn_set_w32(w32[0].ctypes.data_as(u32p), w32[1].ctypes.data_as(u32p))
    h = lib.syn_generate(FIELD_IDS[field], log_h, seed, horner_chain_len, sponge_chain_len, merkle_depth,
                         rc.ctypes.data_as(u32p), flags | (ext_degree << 8 if ext_degree != 4 else 0))
    try:
        err = lib.syn_error(h)
        if err:
            raise RuntimeError("synth: " + err.decode())
        out = {}
        for name in ARRAYS:
            p = u32p()
            n = C.c_size_t()
            if lib.syn_get(h, name.encode(), C.byref(p), C.byref(n)) != 0:
                out[name] = np.zeros(0, np.uint32)   # array never touched by this shape
                continue
            out[name] = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
        return out
    finally:
        lib.syn_free(h)
