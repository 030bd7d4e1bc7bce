"""ctypes view of the CPU oracle (oracle/libp3r_oracle.so). Test infrastructure only."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libp3r_oracle.so")

FIELD_NAMES = {"koala-bear": "koala_bear", "baby-bear": "baby_bear"}
FIELD_IDS = {"koala-bear": 0, "baby-bear": 1}
MODULUS = {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}
GENERATOR = {"koala-bear": 3, "baby-bear": 31}

u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(u32p)


def default_rc(field):
    with open(os.path.join(ROOT, "tests/golden/poseidon2_rc_default.json")) as fh:
        return np.array(json.load(fh)[FIELD_NAMES[field]], dtype=np.uint32)


def default_w32(field):
    """(round constants, internal diagonal) of the DEFAULT width-32 permutation (tools/gen_poseidon2_constants.py; unpinned)."""
    with open(os.path.join(ROOT, "tests/golden/poseidon2_w32_default.json")) as fh:
        d = json.load(fh)[FIELD_NAMES[field]]
    return np.array(d["rc"], dtype=np.uint32), np.array(d["diag"], dtype=np.uint32)


class Oracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(LIB)
        self.lib.orc_last_error.restype = C.c_char_p

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + self.lib.orc_last_error().decode())

    def trace_width(self, field):
        return self.lib.orc_p2_trace_width(FIELD_IDS[field])

    def permute(self, field, states, rc=None):
        rc = default_rc(field) if rc is None else rc
        a, p = _u32(states)
        out = np.empty_like(a)
        r, rp = _u32(rc)
        self._ck(self.lib.orc_p2_permute(FIELD_IDS[field], rp, p, out.ctypes.data_as(u32p), C.c_size_t(a.shape[0])))
        return out

    def trace_rows(self, field, inputs, new_start, merkle_path, mmcs_bit, index_sum, rc=None):
        rc = default_rc(field) if rc is None else rc
        a, p = _u32(inputs)
        n = a.shape[0]
        fl = [np.ascontiguousarray(x, dtype=np.uint8) for x in (new_start, merkle_path, mmcs_bit)]
        s, sp = _u32(index_sum)
        r, rp = _u32(rc)
        out = np.empty((n, self.trace_width(field)), dtype=np.uint32)
        self._ck(self.lib.orc_p2_trace_rows(FIELD_IDS[field], rp, C.c_size_t(n), p,
                                            fl[0].ctypes.data_as(u8p), fl[1].ctypes.data_as(u8p),
                                            fl[2].ctypes.data_as(u8p), sp, out.ctypes.data_as(u32p)))
        return out

    def coset_lde(self, field, evals, added_bits, shift):
        a, p = _u32(evals)
        out = np.empty((a.shape[0] << added_bits, a.shape[1]), dtype=np.uint32)
        self._ck(self.lib.orc_coset_lde(FIELD_IDS[field], p, C.c_size_t(a.shape[0]), C.c_size_t(a.shape[1]),
                                        C.c_uint32(added_bits), C.c_uint32(shift), out.ctypes.data_as(u32p)))
        return out

    def commit(self, field, mats, cap_height=0, rc=None):
        rc = default_rc(field) if rc is None else rc
        arrs = [np.ascontiguousarray(m, dtype=np.uint32) for m in mats]
        n = len(arrs)
        vals = (u32p * n)(*[a.ctypes.data_as(u32p) for a in arrs])
        hs = (C.c_size_t * n)(*[a.shape[0] for a in arrs])
        ws = (C.c_size_t * n)(*[a.shape[1] for a in arrs])
        cap = np.empty((1 << cap_height, 8), dtype=np.uint32)
        tree = C.c_void_p()
        r, rp = _u32(rc)
        self._ck(self.lib.orc_mmcs_commit(FIELD_IDS[field], rp, C.c_size_t(n), vals, hs, ws, cap_height,
                                          cap.ctypes.data_as(u32p), C.byref(tree)))
        return cap, OracleTree(self, tree, arrs, cap_height)

    def p2w_permute(self, field, states, w32=None):
        """The width-32 permutation on n x 32 canonical states (orc_p2w_permute)."""
        w32 = default_w32(field) if w32 is None else w32
        a, p = _u32(np.asarray(states, dtype=np.uint32).reshape(-1, 32))
        out = np.empty_like(a)
        self.lib.orc_p2w_permute.argtypes = [C.c_int, u32p, u32p, u32p, u32p, C.c_size_t]
        self._ck(self.lib.orc_p2w_permute(FIELD_IDS[field], w32[0].ctypes.data_as(u32p), w32[1].ctypes.data_as(u32p), p,
                                          out.ctypes.data_as(u32p), C.c_size_t(a.shape[0])))
        return out

    # ---- arity-4 MMCS over the width-32 permutation (oracle/hash.hpp: MerkleTree::commit4 / open4 / verify4)
    def commit4(self, field, mats, rc=None, w32=None):
        rc = default_rc(field) if rc is None else rc
        w32 = default_w32(field) if w32 is None else w32
        arrs = [np.ascontiguousarray(m, dtype=np.uint32) for m in mats]
        n = len(arrs)
        vals = (u32p * n)(*[a.ctypes.data_as(u32p) for a in arrs])
        hs = (C.c_size_t * n)(*[a.shape[0] for a in arrs])
        ws = (C.c_size_t * n)(*[a.shape[1] for a in arrs])
        cap = np.empty((1, 8), dtype=np.uint32)
        tree, plen = C.c_void_p(), C.c_size_t()
        r, rp = _u32(rc)
        self._ck(self.lib.orc_mmcs_commit4(FIELD_IDS[field], rp, w32[0].ctypes.data_as(u32p), w32[1].ctypes.data_as(u32p),
                                           C.c_size_t(n), vals, hs, ws, cap.ctypes.data_as(u32p), C.byref(plen), C.byref(tree)))
        t = OracleTree(self, tree, arrs, 0)
        t.proof_len = plen.value
        return cap, t

    def verify4(self, field, cap, dims, index, opened, proof, rc=None, w32=None):
        rc = default_rc(field) if rc is None else rc
        w32 = default_w32(field) if w32 is None else w32
        n = len(dims)
        hs = (C.c_size_t * n)(*[d[0] for d in dims])
        ws = (C.c_size_t * n)(*[d[1] for d in dims])
        c, cp = _u32(cap)
        o, op = _u32(opened)
        pf, pp = _u32(np.asarray(proof, dtype=np.uint32).reshape(-1, 8))
        r, rp = _u32(rc)
        ok = C.c_int()
        self._ck(self.lib.orc_mmcs_verify4(FIELD_IDS[field], rp, w32[0].ctypes.data_as(u32p), w32[1].ctypes.data_as(u32p), cp,
                                           C.c_size_t(n), hs, ws, C.c_size_t(index), op, pp, C.c_size_t(pf.shape[0]),
                                           C.byref(ok)))
        return bool(ok.value)

    def schedule4(self, heights):
        """[(step, height of the matrices injected after the level or 0)] of an arity-4 tree over `heights`."""
        n = len(heights)
        hs = (C.c_size_t * n)(*heights)
        steps = (C.c_uint32 * 64)()
        inj = (C.c_size_t * 64)()
        nl = C.c_size_t()
        self._ck(self.lib.orc_mmcs_schedule4(C.c_size_t(n), hs, steps, inj, C.c_size_t(64), C.byref(nl)))
        return [(int(steps[i]), int(inj[i])) for i in range(nl.value)]

    def verify(self, field, cap, dims, index, opened, proof, rc=None):
        rc = default_rc(field) if rc is None else rc
        n = len(dims)
        hs = (C.c_size_t * n)(*[d[0] for d in dims])
        ws = (C.c_size_t * n)(*[d[1] for d in dims])
        c, cp = _u32(cap)
        cap_height = int(np.log2(c.shape[0]))
        o, op = _u32(opened)
        pf, pp = _u32(proof)
        r, rp = _u32(rc)
        ok = C.c_int()
        self._ck(self.lib.orc_mmcs_verify(FIELD_IDS[field], rp, cp, cap_height, C.c_size_t(n), hs, ws,
                                          C.c_size_t(index), op, pp, C.c_size_t(pf.shape[0]), C.byref(ok)))
        return bool(ok.value)

    def challenger_script(self, field, ops, args, rc=None):
        rc = default_rc(field) if rc is None else rc
        o = np.ascontiguousarray(ops, dtype=np.int32)
        a, ap = _u32(args)
        out = np.empty(4 * len(ops) + 4, dtype=np.uint32)
        n = C.c_size_t()
        r, rp = _u32(rc)
        self._ck(self.lib.orc_challenger_script(FIELD_IDS[field], rp, o.ctypes.data_as(C.POINTER(C.c_int32)),
                                                C.c_size_t(len(ops)), ap, out.ctypes.data_as(u32p), C.byref(n)))
        return out[: n.value]

    def ext_ops(self, field, a, b):
        x, xp = _u32(a)
        y, yp = _u32(b)
        mul = np.empty(4, dtype=np.uint32)
        inv = np.empty(4, dtype=np.uint32)
        self._ck(self.lib.orc_ext_ops(FIELD_IDS[field], xp, yp, mul.ctypes.data_as(u32p), inv.ctypes.data_as(u32p)))
        return mul, inv


class OracleTree:
    def __init__(self, orc, handle, arrs, cap_height):
        self.orc, self.h, self.arrs, self.cap_height = orc, handle, arrs, cap_height
        self.log_max_h = int(np.log2(max(a.shape[0] for a in arrs)))

    def open(self, index):
        w = sum(a.shape[1] for a in self.arrs)
        opened = np.empty(w, dtype=np.uint32)
        proof = np.empty((getattr(self, "proof_len", self.log_max_h - self.cap_height), 8), dtype=np.uint32)
        self.orc._ck(self.orc.lib.orc_mmcs_open(self.h, C.c_size_t(index), opened.ctypes.data_as(u32p),
                                                proof.ctypes.data_as(u32p)))
        return opened, proof

    def __del__(self):
        try:
            self.orc.lib.orc_tree_free(self.h)
        except Exception:
            pass
