"""Thin object layer over the C ABI: context, device matrices, Poseidon2, LDE, MMCS.

Names follow the reference's traits where one exists:
  Context               StarkConfig built by circuit-prover/src/config.rs:92-137 /
                        recursion/examples/common/mod.rs:464-486 (FRI params + Poseidon2 perms)
  MerkleTree.commit/open_batch   p3_commit::Mmcs::{commit, open_batch}
  coset_lde_batch       p3_dft::TwoAdicSubgroupDft::coset_lde_batch (+ bit_reverse_rows)
  permute_batch         CryptographicPermutation<[F;16]>::permute over a batch
  generate_trace_rows   Poseidon2CircuitAir::generate_trace_rows (poseidon2-circuit-air/src/air.rs:280)
"""
import ctypes as C

import numpy as np

from . import _lib

FIELD_IDS = {"koala-bear": _lib.FIELD_KOALA_BEAR, "baby-bear": _lib.FIELD_BABY_BEAR}
MODULUS = {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}
GENERATOR = {"koala-bear": 3, "baby-bear": 31}


class P3rError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"p3r error {code}: {msg}")
        self.code = code


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(_lib.u32p)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def make_config(field="koala-bear", log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5,
                commit_pow_bits=0, query_pow_bits=15, num_queries=54, device=0, poseidon2_rc=None, ext_choices=0,
                fri_log_arities=None, proof_layout=None, ext_degree=4, ext_w=0, challenge_degree=4,
                poseidon2_w32_rc=None, poseidon2_w32_diag=None, mmcs_arity=2, zk=0, num_random_codewords=2, zk_seed=None,
                zk_key=None, zk_deterministic=None, allow_unpinned_w32_defaults=False, mmcs_salt_elems=0):
    """A `p3r_config` (+ the arrays it points into, which must stay alive with it).  `ext_choices` /
    `fri_log_arities`: the selectable protocol details of include/p3r.h.
    `ext_degree`: the circuit extension degree D of the traces - 4, or 5 for KoalaBear circuits over the quintic
    trinomial extension (primitive tables; the STARK itself stays on the D = 4 configuration)."""
    cfg = _lib.P3rConfig()
    cfg.abi_version = _lib.P3R_ABI_VERSION
    cfg.field = FIELD_IDS[field]
    cfg.ext_degree = ext_degree
    cfg.ext_w = ext_w     # W of x^D = W for ext_degree 2 / 6 / 8 (include/p3r.h)
    cfg.challenge_degree = challenge_degree   # 5: KoalaBear's quintic challenge field
    cfg.mmcs_arity = mmcs_arity               # 4: the arity-4 MMCS over the width-32 permutation (recursive_aggregation --arity4)
    # ZK: HidingFriPcs with `num_random_codewords` random codewords and a seeded RNG (create_config_zk)
    # `zk_key`: the 256-bit key of its generator (eight u32 words, or 32 bytes); the library mixes operating-system
    # entropy into it unless `zk_deterministic`.  `zk_seed` is the test harness's shorthand: key = (seed, 0, ..) taken
    # as it is - reproducible proofs, which is what a comparison with the CPU oracle needs and a deployment must not use.
    cfg.zk, cfg.num_random_codewords = int(zk), int(num_random_codewords) if zk else 0
    cfg.mmcs_salt_elems = int(mmcs_salt_elems)   # MerkleTreeHidingMmcs (recursion/tests/zk_hiding_mmcs.rs: 4); 0 = plain
    if zk_seed is not None and zk_key is not None:
        raise P3rError(-1, "pass zk_key or zk_seed, not both")
    if zk_seed is not None:
        zk_key = [int(zk_seed) & 0xFFFFFFFF, (int(zk_seed) >> 32) & 0xFFFFFFFF, 0, 0, 0, 0, 0, 0]
        zk_deterministic = True if zk_deterministic is None else zk_deterministic
    if zk_key is not None:
        words = np.frombuffer(bytes(zk_key), dtype="<u4") if isinstance(zk_key, (bytes, bytearray)) else np.asarray(zk_key, dtype=np.uint32)
        if words.size != 8:
            raise P3rError(-1, "zk_key must hold 256 bits (eight u32 words or 32 bytes)")
        for i in range(8):
            cfg.zk_key[i] = int(words[i])
    if zk_deterministic:
        ext_choices |= _lib.P3R_EXT_ZK_DETERMINISTIC
    cfg.log_blowup = log_blowup
    cfg.max_log_arity = max_log_arity
    cfg.cap_height = cap_height
    cfg.log_final_poly_len = log_final_poly_len
    cfg.commit_pow_bits = commit_pow_bits
    cfg.query_pow_bits = query_pow_bits
    cfg.num_queries = num_queries
    cfg.device = device
    rc = None
    if poseidon2_rc is not None:
        rc, ptr = _u32(poseidon2_rc)
        cfg.poseidon2_rc = ptr
        cfg.poseidon2_rc_len = rc.size
    # The width-32 permutation's constants live in un-vendored crates; the library's defaults for them are self-generated
    # and the C ABI wants that acknowledged (P3R_EXT_UNPINNED_W32_DEFAULTS): so does this mirror - a caller that uses
    # the arity-4 MMCS or the width-32 table without upstream's statics says `allow_unpinned_w32_defaults=True` (the
    # tests, bench.py's "unpinned" legs and the profiling tools do), as p3r.hpp's FriParams has it.
    if allow_unpinned_w32_defaults and (poseidon2_w32_rc is None or poseidon2_w32_diag is None):
        ext_choices |= _lib.P3R_EXT_UNPINNED_W32_DEFAULTS
    cfg.ext_choices = ext_choices
    ar = None
    if fri_log_arities is not None:
        ar, aptr = _u8(fri_log_arities)
        cfg.fri_log_arities = aptr
        cfg.fri_log_arities_len = ar.size
    pl = None
    if proof_layout is not None:
        pl, pptr = _u8(proof_layout)
        cfg.proof_layout = pptr
        cfg.proof_layout_len = pl.size
    # constants of the width-32 permutation (the table of the arity-4 MMCS rows); None = the library's defaults
    w_rc = w_dg = None
    if poseidon2_w32_rc is not None:
        w_rc, wptr = _u32(poseidon2_w32_rc)
        cfg.poseidon2_w32_rc = wptr
        cfg.poseidon2_w32_rc_len = w_rc.size
    if poseidon2_w32_diag is not None:
        w_dg, dptr = _u32(poseidon2_w32_diag)
        if w_dg.size != 32:
            raise P3rError(-1, "poseidon2_w32_diag must hold 32 values")
        cfg.poseidon2_w32_diag = dptr
    return cfg, (rc, ar, pl, w_rc, w_dg)


def verify_batch(cfg, airs, preprocessed_commitment, degree_bits, proof: bytes, canonical_field_encoding=False):
    """`verify_batch` behind `verify_all_tables` (batch_stark_prover.rs:1649-1727): host code in the
    C-ABI library, no GPU needed.  `cfg` is a `p3r_config` (e.g. `Context.cfg`), `airs` a list of
    dicts(kind, lanes, horner_packed_steps, coeff_lookups), `degree_bits` the verifier-side log2 trace
    height of every instance (the proof must declare the same).  Raises P3rError with the verifier's
    reason when the proof is rejected."""
    lib = _lib.load()
    arr = (_lib.P3rAirDesc * len(airs))()
    for i, a in enumerate(airs):
        arr[i].kind, arr[i].lanes = a["kind"], a.get("lanes", 1)
        arr[i].horner_packed_steps, arr[i].coeff_lookups = a.get("horner_packed_steps", 2), a.get("coeff_lookups", 0)
    cap, cap_p = _u32(preprocessed_commitment)
    if cap.size != 8 << cfg.cap_height:
        raise P3rError(-1, "preprocessed commitment must hold %d digests" % (1 << cfg.cap_height))
    if len(degree_bits) != len(airs):
        raise P3rError(-1, "degree_bits must have one entry per AIR (%d given, %d AIRs)" % (len(degree_bits), len(airs)))
    db, db_p = _u32(list(degree_bits) or [0])
    buf = (C.c_uint8 * max(len(proof), 1)).from_buffer_copy(proof if proof else b"\0")
    err = C.create_string_buffer(512)
    rc = lib.p3r_verify_batch(C.byref(cfg), arr, len(airs), cap_p, db_p, buf, len(proof), 1 if canonical_field_encoding else 0,
                              err, len(err))
    if rc != 0:
        raise P3rError(rc, err.value.decode())


def mmcs_verify(cfg, cap, dims, index, opened_values, proof, salts=None):
    """`Mmcs::verify_batch` on the host (p3r_mmcs_verify; no GPU): `dims` = (height, width) of the committed matrices in
    commit order, `opened_values` their opened rows concatenated, `proof` the sibling digests.  Honours cfg.mmcs_arity.
    `salts` (n_mats x cfg.mmcs_salt_elems): the hiding MMCS's opening (p3r_mmcs_verify_salted).
    Raises P3rError with the reason when the opening is rejected."""
    lib = _lib.load()
    c, cp = _u32(cap)
    o, op = _u32(opened_values)
    pf, pp = _u32(np.asarray(proof, dtype=np.uint32).reshape(-1, 8))
    n = len(dims)
    hs = (C.c_size_t * n)(*[int(d[0]) for d in dims])
    ws = (C.c_size_t * n)(*[int(d[1]) for d in dims])
    err = C.create_string_buffer(512)
    if salts is not None:
        sl, sp = _u32(np.asarray(salts, dtype=np.uint32).reshape(-1))
        if sl.size != n * cfg.mmcs_salt_elems:
            raise P3rError(-1, "salts must hold %d x %d values" % (n, cfg.mmcs_salt_elems))
        rc = lib.p3r_mmcs_verify_salted(C.byref(cfg), cp, n, hs, ws, int(index), op, sp, pp, pf.shape[0], err, len(err))
    else:
        rc = lib.p3r_mmcs_verify(C.byref(cfg), cp, n, hs, ws, int(index), op, pp, pf.shape[0], err, len(err))
    if rc != 0:
        raise P3rError(rc, err.value.decode())


class Context:
    """One per GPU; not thread-safe; one call in flight (include/p3r.h)."""

    def __init__(self, field="koala-bear", log_blowup=2, max_log_arity=2, cap_height=0,
                 log_final_poly_len=5, commit_pow_bits=0, query_pow_bits=15, num_queries=54,
                 device=0, poseidon2_rc=None, ext_choices=0, fri_log_arities=None, proof_layout=None, ext_degree=4, ext_w=0,
                 challenge_degree=4, poseidon2_w32_rc=None, poseidon2_w32_diag=None, mmcs_arity=2, zk=0,
                 num_random_codewords=2, zk_seed=None, zk_key=None, zk_deterministic=None, allow_unpinned_w32_defaults=False,
                 mmcs_salt_elems=0):
        self.lib = _lib.load()
        self.mmcs_arity = mmcs_arity
        self.zk = int(zk)
        self.field = field
        self.ext_degree = ext_degree
        self.ext_w = ext_w
        self.challenge_degree = challenge_degree
        self.p = MODULUS[field]
        cfg, self._rc_keep = make_config(field, log_blowup, max_log_arity, cap_height, log_final_poly_len,
                                         commit_pow_bits, query_pow_bits, num_queries, device, poseidon2_rc, ext_choices,
                                         fri_log_arities, proof_layout, ext_degree, ext_w, challenge_degree,
                                         poseidon2_w32_rc, poseidon2_w32_diag, mmcs_arity, zk, num_random_codewords, zk_seed,
                                         zk_key, zk_deterministic, allow_unpinned_w32_defaults, mmcs_salt_elems)
        self.cfg = cfg
        self.cap_height = cap_height
        self.log_blowup = log_blowup
        self.h = self.lib.p3r_create(C.byref(cfg))
        if not self.h:
            raise P3rError(-2, self.lib.p3r_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.p3r_destroy(self.h)
            self.h = None

    @property
    def zk_nonce(self):
        """Proofs made so far under a ZK configuration (the hiding PCS's RNG state: p3r_zk_nonce)."""
        return int(self.lib.p3r_zk_nonce(self.h))

    @zk_nonce.setter
    def zk_nonce(self, v):
        self.check(self.lib.p3r_zk_set_nonce(self.h, C.c_uint64(int(v))))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise P3rError(rc, self.lib.p3r_last_error(self.h).decode())

    def ptr(self, p):
        if not p:
            raise P3rError(-1, self.lib.p3r_last_error(self.h).decode())
        return p

    def sync(self):
        self.check(self.lib.p3r_sync(self.h))

    @property
    def poseidon2_trace_width(self):
        return self.lib.p3r_poseidon2_trace_width(self.h)

    # ---- matrices
    def upload(self, rowmajor):
        a, p = _u32(rowmajor)
        assert a.ndim == 2
        return DeviceMatrix(self, self.ptr(self.lib.p3r_dmat_upload(self.h, p, a.shape[0], a.shape[1])))

    # ---- Poseidon2
    def permute_batch(self, states):
        a, p = _u32(states)
        assert a.ndim == 2 and a.shape[1] == 16
        out = np.empty_like(a)
        self.check(self.lib.p3r_poseidon2_permute_batch(self.h, p, out.ctypes.data_as(_lib.u32p), a.shape[0]))
        return out

    def _p2_rows(self, inputs, new_start, merkle_path, mmcs_bit, mmcs_index_sum):
        keep = []
        rows = _lib.P3rP2Rows()
        a, rows.input_values = _u32(inputs)
        keep.append(a)
        rows.n = a.shape[0]
        for name, v in (("new_start", new_start), ("merkle_path", merkle_path), ("mmcs_bit", mmcs_bit)):
            b, p = _u8(v)
            keep.append(b)
            setattr(rows, name, p)
        b, rows.mmcs_index_sum = _u32(mmcs_index_sum)
        keep.append(b)
        return rows, keep

    def generate_trace_rows(self, inputs, new_start, merkle_path, mmcs_bit, mmcs_index_sum):
        rows, keep = self._p2_rows(inputs, new_start, merkle_path, mmcs_bit, mmcs_index_sum)
        out = np.empty((rows.n, self.poseidon2_trace_width), dtype=np.uint32)
        self.check(self.lib.p3r_poseidon2_trace_fill(self.h, C.byref(rows), out.ctypes.data_as(_lib.u32p)))
        return out

    def generate_trace_rows_device(self, inputs, new_start, merkle_path, mmcs_bit, mmcs_index_sum):
        rows, keep = self._p2_rows(inputs, new_start, merkle_path, mmcs_bit, mmcs_index_sum)
        return DeviceMatrix(self, self.ptr(self.lib.p3r_poseidon2_trace_fill_dmat(self.h, C.byref(rows))))

    # ---- the width-32 permutation (the arity-4 MMCS) and its table's trace rows (include/p3r.h, ABI 6)
    def poseidon2_w32_permute_batch(self, states):
        a, p = _u32(states)
        if a.ndim != 2 or a.shape[1] != 32:
            raise P3rError(-1, "states must have shape (n, 32)")
        out = np.empty_like(a)
        self.check(self.lib.p3r_poseidon2_w32_permute_batch(self.h, p, out.ctypes.data_as(_lib.u32p), a.shape[0]))
        return out

    def generate_w32_trace_rows(self, inputs, new_start, merkle_path, mmcs_bit, mmcs_bit2, mmcs_index_sum, height=None):
        """Poseidon2CircuitAir::generate_trace_rows of the arity-4 layout; rows are padded to `height` (a power of two)
        with filler rows (new_start = 1, zero state)."""
        a = np.ascontiguousarray(inputs, dtype=np.uint32).reshape(-1, 32)
        n = a.shape[0]
        h = height or (1 << max(n - 1, 0).bit_length())

        def pad(x, fill, dt):   # explicit shapes: n == 0 (a table of filler rows only) has no rows to take the width from
            x = np.asarray(x, dt)
            w = 32 if x.ndim == 2 else 1
            return np.concatenate([x.reshape(n, w), np.full((h - n, w), fill, dt)])
        rows = _lib.P3rP2wRows()
        keep = []

        def put(name, arr, ptr_of):
            x, p = ptr_of(np.ascontiguousarray(arr))
            keep.append(x)
            setattr(rows, name, p)
        rows.n = h
        put("input_values", pad(a, 0, np.uint32), _u32)
        put("new_start", pad(new_start, 1, np.uint8).reshape(-1), _u8)
        put("merkle_path", pad(merkle_path, 0, np.uint8).reshape(-1), _u8)
        put("mmcs_bit", pad(mmcs_bit, 0, np.uint8).reshape(-1), _u8)
        put("mmcs_bit2", pad(mmcs_bit2, 0, np.uint8).reshape(-1), _u8)
        put("mmcs_index_sum", pad(mmcs_index_sum, 0, np.uint32).reshape(-1), _u32)
        out = np.empty((h, int(self.lib.p3r_poseidon2_w32_trace_width(self.h))), dtype=np.uint32)
        self.check(self.lib.p3r_poseidon2_w32_trace_fill(self.h, C.byref(rows), out.ctypes.data_as(_lib.u32p)))
        return out

    def upload_p2_rows(self, inputs, new_start, merkle_path, mmcs_bit, mmcs_index_sum):
        rows, keep = self._p2_rows(inputs, new_start, merkle_path, mmcs_bit, mmcs_index_sum)
        return DeviceP2Rows(self, self.ptr(self.lib.p3r_p2_rows_upload(self.h, C.byref(rows))))

    def generate_trace_rows_resident(self, dev_rows):
        return DeviceMatrix(self, self.ptr(self.lib.p3r_poseidon2_trace_fill_dev(self.h, dev_rows.h)))

    # ---- LDE
    def coset_lde_batch(self, evals, added_bits, shift):
        a, p = _u32(evals)
        out = np.empty((a.shape[0] << added_bits, a.shape[1]), dtype=np.uint32)
        self.check(self.lib.p3r_coset_lde(self.h, p, a.shape[0], a.shape[1], added_bits, shift,
                                          out.ctypes.data_as(_lib.u32p)))
        return out

    def coset_lde_batch_device(self, dmat, added_bits, shift):
        return DeviceMatrix(self, self.ptr(self.lib.p3r_coset_lde_dmat(self.h, dmat.h, added_bits, shift)))

    # ---- MMCS
    def commit(self, mats):
        """Mmcs::commit over host matrices (list of 2-D uint32 arrays). Returns (cap, tree)."""
        keep = []
        arr = (_lib.P3rMatrix * len(mats))()
        for i, m in enumerate(mats):
            a, p = _u32(m)
            keep.append(a)
            arr[i].values, arr[i].height, arr[i].width = p, a.shape[0], a.shape[1]
        cap = np.empty((1 << self.cap_height, 8), dtype=np.uint32)
        tree = C.c_void_p()
        self.check(self.lib.p3r_mmcs_commit(self.h, arr, len(mats), cap.ctypes.data_as(_lib.u32p), C.byref(tree)))
        return cap, MerkleTree(self, tree.value, [])

    def commit_device(self, dmats):
        arr = (C.c_void_p * len(dmats))(*[d.h for d in dmats])
        cap = np.empty((1 << self.cap_height, 8), dtype=np.uint32)
        tree = C.c_void_p()
        self.check(self.lib.p3r_mmcs_commit_dmat(self.h, arr, len(dmats), cap.ctypes.data_as(_lib.u32p), C.byref(tree)))
        return cap, MerkleTree(self, tree.value, list(dmats))

    # ---- batch-STARK proving
    def prep_create(self, airs, prep_mats):
        """airs: list of dicts(kind, lanes, horner_packed_steps, coeff_lookups); prep_mats: 2-D arrays.
        Returns (commitment, ProverData)."""
        n = len(airs)
        descs = (_lib.P3rAirDesc * n)()
        arr = (_lib.P3rMatrix * n)()
        keep = []
        for i, (a, m) in enumerate(zip(airs, prep_mats)):
            descs[i].kind = a["kind"]
            descs[i].lanes = a.get("lanes", 1)
            descs[i].horner_packed_steps = a.get("horner_packed_steps", 2)
            descs[i].coeff_lookups = a.get("coeff_lookups", 0)
            x, p = _u32(m)
            keep.append(x)
            arr[i].values, arr[i].height, arr[i].width = p, x.shape[0], x.shape[1]
        cap = np.empty((1 << self.cap_height, 8), dtype=np.uint32)
        h = self.ptr(self.lib.p3r_prep_create(self.h, descs, arr, n, cap.ctypes.data_as(_lib.u32p)))
        return cap, ProverData(self, h)

    def _proof_call(self, fn, *args):
        buf = getattr(self, "_proof_buf", None)
        if buf is None:
            buf = self._proof_buf = C.create_string_buffer(1 << 20)
        n = C.c_size_t()
        rc = fn(*args, C.cast(buf, C.POINTER(C.c_uint8)), len(buf), C.byref(n))
        if rc == -6 and n.value > len(buf):  # P3R_EBUFFER: the library kept the proof (p3r_take_proof) - it is not made again
            buf = self._proof_buf = C.create_string_buffer(max(n.value + n.value // 4, 2 * len(buf)))
            rc = self.lib.p3r_take_proof(self.h, C.cast(buf, C.POINTER(C.c_uint8)), len(buf), C.byref(n))
        self.check(rc)
        return C.string_at(buf, n.value)   # one memcpy (slicing a ctypes array builds a list first)

    def prove_batch(self, prover_data, main_traces, canonical_field_encoding=False):
        """main_traces: DeviceMatrix list (resident) or 2-D uint32 arrays (host)."""
        flags = 1 if canonical_field_encoding else 0
        n = len(main_traces)
        if all(isinstance(m, DeviceMatrix) for m in main_traces):
            arr = (C.c_void_p * n)(*[m.h for m in main_traces])
            return self._proof_call(self.lib.p3r_prove_batch, self.h, prover_data.h, arr, n, flags)
        arr = (_lib.P3rMatrix * n)()
        keep = []
        for i, m in enumerate(main_traces):
            x, p = _u32(m)
            keep.append(x)
            arr[i].values, arr[i].height, arr[i].width = p, x.shape[0], x.shape[1]
        return self._proof_call(self.lib.p3r_prove_batch_host, self.h, prover_data.h, arr, n, flags)

    # ---- profiling
    def profile_enable(self, on=True):
        self.check(self.lib.p3r_profile_enable(self.h, 1 if on else 0))

    def trim(self):
        """Return this ctx's cached device memory to the driver; returns the number of bytes released."""
        n = C.c_uint64()
        self.check(self.lib.p3r_trim(self.h, C.byref(n)))
        return n.value

    def profile_read(self):
        buf = (_lib.P3rProfileEntry * 64)()
        n = C.c_size_t()
        self.check(self.lib.p3r_profile_read(self.h, buf, 64, C.byref(n)))
        return {buf[i].name.decode(): (buf[i].total_ms, buf[i].launches) for i in range(n.value)}

    def time_permute(self, dmat, iters):
        ms = C.c_double()
        self.check(self.lib.p3r_time_permute_dmat(self.h, dmat.h, iters, C.byref(ms)))
        return ms.value


class DeviceMatrix:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @property
    def shape(self):
        return (self.ctx.lib.p3r_dmat_height(self.h), self.ctx.lib.p3r_dmat_width(self.h))

    def download(self):
        out = np.empty(self.shape, dtype=np.uint32)
        self.ctx.check(self.ctx.lib.p3r_dmat_download(self.ctx.h, self.h, out.ctypes.data_as(_lib.u32p)))
        return out

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.p3r_dmat_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ProverData:
    """Device-resident preprocessed LDEs + commitment (ProverData::from_airs_and_degrees)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.p3r_prep_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceP2Rows:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.p3r_p2_rows_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MerkleTree:
    def __init__(self, ctx, handle, borrowed):
        self.ctx, self.h, self._borrowed = ctx, handle, borrowed

    @property
    def log_max_height(self):
        return self.ctx.lib.p3r_tree_log_max_height(self.h)

    def open_batch(self, index):
        """Returns (opened_values concatenated in commit order, proof[(depth, 8)])."""
        w = self.ctx.lib.p3r_tree_total_width(self.h)
        depth = self.ctx.lib.p3r_tree_proof_len(self.h)   # binary: log_max_height - cap_height; arity 4: sum of (step - 1)
        opened = np.empty(w, dtype=np.uint32)
        proof = np.empty((depth, 8), dtype=np.uint32)
        self.ctx.check(self.ctx.lib.p3r_mmcs_open(self.ctx.h, self.h, index, opened.ctypes.data_as(_lib.u32p),
                                                  proof.ctypes.data_as(_lib.u32p)))
        return opened, proof

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.p3r_tree_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
