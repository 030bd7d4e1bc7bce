"""Host-side mirror of the reference's interface for the `prove_next_layer` path.

Same names and argument meaning as the Rust API (the Rust toolchain is absent here, so the host
side above the C ABI is written in Python for tests/bench; the Rust shim a maintainer would add
is in INTEGRATION.md):

  TablePacking                circuit-prover/src/batch_stark_prover/packing.rs:10-27,100-106
  Traces                      circuit/src/tables/mod.rs:49-62 (flattened, D = 4)
  CircuitPrep                 output of get_airs_and_degrees_with_prep, circuit-prover/src/common.rs:127-390
  CircuitProverData           circuit-prover/src/batch_stark_prover.rs:314-341
  BatchStarkProver            circuit-prover/src/batch_stark_prover.rs:685-697, prove_all_tables :1203-1222
  BatchStarkProof             circuit-prover/src/batch_stark_prover.rs:610-636
  FriRecursionConfig/Backend  recursion/src/backend/fri.rs:41-128
  ProveNextLayerParams        recursion/src/recursion.rs:221-234
  NextLayerPrepCache          recursion/src/recursion.rs:295-298
  build_next_layer_prep       recursion/src/recursion.rs:342-394
  prove_next_layer            recursion/src/recursion.rs:401-502
  RecursionInput / Output     recursion/src/recursion.rs:96-139

What stays on the reference's (CPU, Rust) side and is NOT rebuilt here: building and running the
verifier circuit (`build_next_layer_circuit`, `CircuitRunner::run`).  `RecursionInput` therefore
carries the `Traces` that run produced; everything from there to the proof bytes is on the GPU.
"""
import ctypes as C
import dataclasses
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _lib
from .device import Context, DeviceMatrix, P3rError, _u32, _u8, make_config, verify_batch


@dataclass
class TablePacking:
    public_lanes: int = 1
    alu_lanes: int = 3
    horner_packed_steps: int = 4
    recompose_lanes: int = 1
    min_trace_height: int = 1

    def with_fri_params(self, log_final_poly_len: int, log_blowup: int) -> "TablePacking":
        # packing.rs:100-106: FRI needs log_height > log_final_poly_len + log_blowup
        self.min_trace_height = 1 << (log_final_poly_len + log_blowup + 1)
        return self

    def validate(self):
        # packing.rs:140-161
        if min(self.public_lanes, self.alu_lanes, self.recompose_lanes) < 1:
            raise ValueError("lane counts must be non-zero")
        if self.horner_packed_steps < 2:
            raise ValueError("horner_packed_steps must be at least 2")
        if self.min_trace_height & (self.min_trace_height - 1):
            raise ValueError("min_trace_height must be a power of two")


@dataclass
class Traces:
    """Flattened `Traces<EF>`: every array is canonical uint32."""
    const_values: np.ndarray        # (n_const, D)      D = the context's ext_degree (4, or 5 for quintic circuits)
    public_values: np.ndarray       # (n_public, D)
    alu_values: np.ndarray          # (n_alu, 4 D): [a, b, c, out] x D coefficients
    p2_input_values: np.ndarray     # (n_p2, 16)
    p2_new_start: np.ndarray        # (n_p2,) bool
    p2_merkle_path: np.ndarray
    p2_mmcs_bit: np.ndarray
    p2_mmcs_index_sum: np.ndarray   # (n_p2,)
    recompose_values: np.ndarray    # (n_recompose, D)
    # rows of the second Recompose table (`recompose/coeff` next to `recompose`; include/p3r.h, ABI version 5)
    recompose_coeff_values: Optional[np.ndarray] = None   # (n_recompose_coeff, D)
    # rows of the width-32 Poseidon2 table (arity-4 MMCS rows; include/p3r.h p3r_p2w_rows, ABI version 6)
    p2w_input_values: Optional[np.ndarray] = None         # (n_p2w, 32)
    p2w_new_start: Optional[np.ndarray] = None
    p2w_merkle_path: Optional[np.ndarray] = None
    p2w_mmcs_bit: Optional[np.ndarray] = None
    p2w_mmcs_bit2: Optional[np.ndarray] = None
    p2w_mmcs_index_sum: Optional[np.ndarray] = None


@dataclass
class CircuitPrep:
    """Per-op preprocessed data, as `get_airs_and_degrees_with_prep` produces it."""
    const_prep: np.ndarray          # (n_const, 2): [ext_mult, D*idx]
    public_prep: np.ndarray         # (n_public, 2)
    alu_prep13: np.ndarray          # (n_alu, 13)
    recompose_prep: np.ndarray      # (n_recompose, 2): [D*idx, mult]
    p2_new_start: np.ndarray
    p2_merkle_path: np.ndarray
    p2_mmcs_ctl_enabled: np.ndarray
    p2_in_ctl: np.ndarray           # (n_p2, IL)   IL x OL = 4 x 2 limbs (D = 4), 16 x 8 elements (the compact-D1
    p2_input_indices: np.ndarray    # (n_p2, IL)   table of a D = 5 circuit: include/p3r.h)
    p2_out_ctl: np.ndarray          # (n_p2, OL) multiplicities
    p2_output_indices: np.ndarray   # (n_p2, OL)
    p2_mmcs_index_sum_idx: np.ndarray
    p2_absorb_len: Optional[np.ndarray] = None   # (n_p2,) sponge length tags of compact-D1 rows (None: zeros)
    # the "recompose/coeff" table (per-coefficient bus tuples): recompose_prep is (n_recompose, 2 + 2 D)
    recompose_coeff_lookups: bool = False
    # a layer holding BOTH Recompose tables (recompose_table_provers(lanes, true), batch_stark_prover.rs:1914-1932):
    # recompose_prep is then the plain kind and this is the `recompose/coeff` table, (n, 2 + 2 D)
    recompose_coeff_prep: Optional[np.ndarray] = None
    # the width-32 Poseidon2 table: assembled preprocessed rows (n_p2w, 48), Poseidon2PreprocessedRow<8, 6>
    p2w_prep: Optional[np.ndarray] = None


class CircuitProverData:
    """Device-resident preprocessed LDEs + commitment + ALU schedule for one circuit shape."""

    def __init__(self, ctx: Context, prep: CircuitPrep, packing: TablePacking):
        packing.validate()
        self.ctx, self.packing = ctx, packing
        self._borrowed, self._owner = False, None
        d = _lib.P3rLayerDesc()
        keep = []

        def p32(a, cols=None):
            x, p = _u32(a)
            keep.append(x)
            return p

        def p8(a):
            x, p = _u8(a)
            keep.append(x)
            return p

        d.counts.n_const = len(prep.const_prep)
        d.counts.n_public = len(prep.public_prep)
        d.counts.n_alu = len(prep.alu_prep13)
        d.counts.n_p2 = len(prep.p2_new_start)
        rec_w = 2 + (2 * ctx.ext_degree if prep.recompose_coeff_lookups else 0)
        rec = np.asarray(prep.recompose_prep)
        if rec.size % rec_w:
            raise P3rError(-1, "recompose_prep must be (n, %d)" % rec_w)
        d.counts.n_recompose = rec.size // rec_w
        d.recompose_coeff_lookups = 1 if prep.recompose_coeff_lookups else 0
        rec2 = np.zeros(0, np.uint32) if prep.recompose_coeff_prep is None else np.asarray(prep.recompose_coeff_prep)
        if rec2.size % (2 + 2 * ctx.ext_degree):
            raise P3rError(-1, "recompose_coeff_prep must be (n, %d)" % (2 + 2 * ctx.ext_degree))
        d.counts.n_recompose_coeff = rec2.size // (2 + 2 * ctx.ext_degree)
        if d.counts.n_recompose_coeff:
            d.recompose_coeff_prep = p32(rec2)
        pw = np.zeros(0, np.uint32) if prep.p2w_prep is None else np.asarray(prep.p2w_prep)
        if pw.size % 48:
            raise P3rError(-1, "p2w_prep must be (n, 48)")
        d.counts.n_p2w = pw.size // 48
        if d.counts.n_p2w:
            d.p2w_prep = p32(pw)
        d.public_lanes, d.alu_lanes = packing.public_lanes, packing.alu_lanes
        d.horner_packed_steps, d.recompose_lanes = packing.horner_packed_steps, packing.recompose_lanes
        d.min_trace_height = packing.min_trace_height
        d.const_prep, d.public_prep = p32(prep.const_prep), p32(prep.public_prep)
        d.alu_prep13, d.recompose_prep = p32(prep.alu_prep13), p32(prep.recompose_prep)
        d.p2_new_start, d.p2_merkle_path = p8(prep.p2_new_start), p8(prep.p2_merkle_path)
        d.p2_mmcs_ctl_enabled, d.p2_in_ctl = p8(prep.p2_mmcs_ctl_enabled), p8(prep.p2_in_ctl)
        d.p2_input_indices, d.p2_out_ctl = p32(prep.p2_input_indices), p32(prep.p2_out_ctl)
        d.p2_output_indices, d.p2_mmcs_index_sum_idx = p32(prep.p2_output_indices), p32(prep.p2_mmcs_index_sum_idx)
        il, ol = (4, 2) if ctx.ext_degree == 4 else (16, 8)
        n_p2 = len(prep.p2_new_start)
        for name, w in (("p2_in_ctl", il), ("p2_input_indices", il), ("p2_out_ctl", ol), ("p2_output_indices", ol)):
            if np.asarray(getattr(prep, name)).size != n_p2 * w:
                raise P3rError(-1, "%s must hold %d x %d entries for ext_degree %d" % (name, n_p2, w, ctx.ext_degree))
        if prep.p2_absorb_len is not None and len(prep.p2_absorb_len):
            if len(prep.p2_absorb_len) != n_p2:
                raise P3rError(-1, "p2_absorb_len must hold one entry per Poseidon2 row")
            d.p2_absorb_len = p8(prep.p2_absorb_len)
        self.rows = dict(const=d.counts.n_const, public=d.counts.n_public, alu=d.counts.n_alu,
                         poseidon2=d.counts.n_p2, recompose=d.counts.n_recompose,
                         recompose_coeff=d.counts.n_recompose_coeff, poseidon2_w32=d.counts.n_p2w)
        self.preprocessed_commitment = np.empty((1 << ctx.cap_height, 8), dtype=np.uint32)
        self.h = ctx.ptr(ctx.lib.p3r_layer_create(ctx.h, C.byref(d),
                                                  self.preprocessed_commitment.ctypes.data_as(_lib.u32p)))
        self._read_shape()

    @classmethod
    def _borrow(cls, ctx: Context, handle, packing: TablePacking, rows: dict, commitment: np.ndarray, owner=None):
        """View of the CircuitProverData a prepared circuit owns (p3r_circuit_layer).  The view keeps its
        owner alive - the reference hands out an Rc<CircuitProverData> that outlives the cache slot it came
        from (recursion.rs:748-761) - and is invalidated only by an explicit PreparedCircuit.free()."""
        self = cls.__new__(cls)
        self.ctx, self.packing, self.h, self.rows = ctx, packing, handle, rows
        self.preprocessed_commitment, self._borrowed, self._owner = commitment, True, owner
        self._read_shape()
        return self

    def _read_shape(self):
        ctx, packing = self.ctx, self.packing
        hs = (C.c_size_t * 5)()
        ctx.check(ctx.lib.p3r_layer_table_heights(self.h, hs))
        self.table_heights = [int(x) for x in hs]   # 0 = table absent from the batch
        h5 = C.c_size_t()
        ctx.check(ctx.lib.p3r_layer_recompose_coeff_height(self.h, C.byref(h5)))
        self.recompose_coeff_height = int(h5.value)   # the second Recompose table (0 = absent)
        ctx.check(ctx.lib.p3r_layer_p2w_height(self.h, C.byref(h5)))
        self.p2w_height = int(h5.value)               # the width-32 Poseidon2 table (0 = absent)
        kind = C.c_uint32()
        ctx.check(ctx.lib.p3r_layer_recompose_kind(self.h, C.byref(kind)))
        self.recompose_coeff_lookups = bool(kind.value)   # the table at position 4 is `recompose/coeff`
        pl, al = C.c_uint32(), C.c_uint32()
        ctx.check(ctx.lib.p3r_layer_effective_lanes(self.h, C.byref(pl), C.byref(al)))
        # reduce_lanes_if_dummy (batch_stark_prover.rs:1305-1318): what the proof records (:1617-1622)
        self.effective_packing = dataclasses.replace(packing, public_lanes=pl.value, alu_lanes=al.value)

    def free(self):
        if self.h and self.ctx.h and not self._borrowed:
            self.ctx.lib.p3r_layer_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


TRACES_ARRAYS = dict(const_values=(0, "const", 4), public_values=(1, "public", 4), alu_values=(2, "alu", 16),
                     p2_input_values=(3, "poseidon2", 16), p2_flags=(4, "poseidon2", 3),
                     p2_mmcs_index_sum=(5, "poseidon2", 1), recompose_values=(6, "recompose", 4),
                     recompose_coeff_values=(7, "recompose_coeff", 4),
                     p2w_input_values=(8, "poseidon2_w32", 32), p2w_flags=(9, "poseidon2_w32", 4),
                     p2w_mmcs_index_sum=(10, "poseidon2_w32", 1))


class ResidentTraces:
    """`Traces` uploaded to HBM once (so a prove starts from device-resident inputs)."""

    def __init__(self, ctx: Context, cpd: CircuitProverData, traces: Traces):
        self.ctx, self.cpd = ctx, cpd
        t, self._keep = _traces_struct(traces, ctx.ext_degree)
        self.h = ctx.ptr(ctx.lib.p3r_traces_upload(ctx.h, cpd.h, C.byref(t)))

    @classmethod
    def _adopt(cls, ctx: Context, cpd: CircuitProverData, handle):
        self = cls.__new__(cls)
        self.ctx, self.cpd, self.h, self._keep = ctx, cpd, handle, None
        return self

    def download(self, name: str) -> np.ndarray:
        """One array of the device-resident Traces, canonical (see TRACES_ARRAYS)."""
        which, table, width = TRACES_ARRAYS[name]
        if table in ("const", "public", "alu", "recompose", "recompose_coeff"):
            width = width // 4 * self.ctx.ext_degree
        out = np.empty((self.cpd.rows.get(table, 0), width), dtype=np.uint32)
        self.ctx.check(self.ctx.lib.p3r_dtraces_get(self.ctx.h, self.cpd.h, self.h, which,
                                                    out.ctypes.data_as(_lib.u32p), out.size))
        return out

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.p3r_traces_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _traces_struct(tr: Traces, ext_degree=4):
    t = _lib.P3rTraces()
    keep = []
    for name, w in (("const_values", ext_degree), ("public_values", ext_degree), ("alu_values", 4 * ext_degree),
                    ("recompose_values", ext_degree)):
        a = np.asarray(getattr(tr, name))
        if a.ndim != 2 or a.shape[1] != w:
            raise P3rError(-1, "%s must have shape (n, %d) for ext_degree %d, got %r" % (name, w, ext_degree, a.shape))

    def p32(a):
        x, p = _u32(a)
        keep.append(x)
        return x, p

    x, t.const_values = p32(tr.const_values)
    t.n_const = x.shape[0]
    x, t.public_values = p32(tr.public_values)
    t.n_public = x.shape[0]
    x, t.alu_values = p32(tr.alu_values)
    t.n_alu = x.shape[0]
    x, t.p2.input_values = p32(tr.p2_input_values)
    t.p2.n = x.shape[0]
    for name, src in (("new_start", tr.p2_new_start), ("merkle_path", tr.p2_merkle_path), ("mmcs_bit", tr.p2_mmcs_bit)):
        b, p = _u8(src)
        keep.append(b)
        setattr(t.p2, name, p)
    x, t.p2.mmcs_index_sum = p32(tr.p2_mmcs_index_sum)
    x, t.recompose_values = p32(tr.recompose_values)
    t.n_recompose = x.shape[0]
    rc = getattr(tr, "recompose_coeff_values", None)
    if rc is not None and np.asarray(rc).size:
        rc = np.asarray(rc)
        if rc.ndim != 2 or rc.shape[1] != ext_degree:
            raise P3rError(-1, "recompose_coeff_values must have shape (n, %d), got %r" % (ext_degree, rc.shape))
        x, t.recompose_coeff_values = p32(rc)
        t.n_recompose_coeff = x.shape[0]
    pw = getattr(tr, "p2w_input_values", None)
    if pw is not None and np.asarray(pw).size:
        pw = np.asarray(pw)
        if pw.ndim != 2 or pw.shape[1] != 32:
            raise P3rError(-1, "p2w_input_values must have shape (n, 32), got %r" % (pw.shape,))
        x, t.p2w.input_values = p32(pw)
        t.p2w.n = x.shape[0]
        for name in ("new_start", "merkle_path", "mmcs_bit", "mmcs_bit2"):
            b, p = _u8(getattr(tr, "p2w_" + name))
            if b.shape[0] != t.p2w.n:
                raise P3rError(-1, "p2w_%s must hold one entry per row" % name)
            keep.append(b)
            setattr(t.p2w, name, p)
        # the C side copies n words from this array too: a missing or short one must not reach it
        idx = getattr(tr, "p2w_mmcs_index_sum", None)
        if idx is None or np.asarray(idx).reshape(-1).shape[0] != t.p2w.n:
            raise P3rError(-1, "p2w_mmcs_index_sum must hold one entry per row (%d)" % t.p2w.n)
        x, t.p2w.mmcs_index_sum = p32(np.asarray(idx).reshape(-1))
    return t, keep


def _varint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _pc_str(s: str) -> bytes:
    b = s.encode()
    return _varint(len(b)) + b


AIR_VARIANT_BASELINE, AIR_VARIANT_OPTIMIZED = 0, 1


@dataclass
class NonPrimitiveTableEntry:
    """circuit-prover/src/batch_stark_prover.rs:272-290."""
    op_type: str
    rows: int
    lanes: int = 1
    public_values: tuple = ()
    air_variant: int = AIR_VARIANT_BASELINE


@dataclass
class BatchStarkProof:
    """`proof` holds the postcard bytes of the inner `BatchProof<SC>` (what the GPU computes); the
    remaining fields are the metadata the reference stores next to it
    (circuit-prover/src/batch_stark_prover.rs:610-636, populated at :1598-1641)."""
    proof: bytes
    table_packing: TablePacking
    rows: tuple                      # RowCounts([const, public, alu]) - op counts, padded to >= 1
    alu_variant: int = AIR_VARIANT_OPTIMIZED   # BatchStarkProver::new default (:1106-1114)
    ext_degree: int = 4
    w_binomial: Optional[int] = None
    alu_quintic_trinomial: bool = False
    non_primitives: tuple = ()
    # stark_common: the preprocessed binding (serde_stark_common, :582-607)
    preprocessed_commitment: Optional[np.ndarray] = None
    preprocessed_widths: tuple = ()
    degree_bits: tuple = ()
    monty_r: int = 0                 # 2^32 mod p when field elements serialise in Montgomery form, else 0
    modulus: int = 0
    parse_ns: int = field(default=0, compare=False)   # from_postcard: time spent in the native parser (p3r_batch_stark_proof_parse)

    def _fe(self, x: int) -> bytes:
        return _varint((x << 32) % self.modulus if self.monty_r else x)

    def validate(self):
        """Structural invariants `#[derive(Deserialize)]` can bypass (batch_stark_prover.rs:666-681,
        RowCounts :459-480, packing.rs:140-161, NonPrimitiveTableEntry :272-300)."""
        if self.ext_degree not in (1, 2, 4, 5, 6, 8):
            raise P3rError(-1, "UnsupportedExtDegree(%d)" % self.ext_degree)
        if len(self.rows) != 3 or any(int(r) < 0 for r in self.rows):
            raise P3rError(-1, "bad RowCounts")
        tp = self.table_packing
        if tp.public_lanes == 0:
            raise P3rError(-1, 'ZeroLanes("public_lanes")')
        if tp.alu_lanes == 0:
            raise P3rError(-1, 'ZeroLanes("alu_lanes")')
        for e in self.non_primitives:
            if e.lanes == 0:
                raise P3rError(-1, "ZeroNpoLanes(%s)" % e.op_type)
        if tp.min_trace_height == 0 or tp.min_trace_height & (tp.min_trace_height - 1):
            raise P3rError(-1, "BadMinTraceHeight(%d)" % tp.min_trace_height)
        if tp.horner_packed_steps < 2:
            raise P3rError(-1, "BadHornerPackedSteps(%d)" % tp.horner_packed_steps)

    def airs(self):
        """The proved tables in instance order, from the metadata the proof carries."""
        tp = self.table_packing
        out = [dict(kind=0, lanes=1), dict(kind=1, lanes=tp.public_lanes),
               dict(kind=2, lanes=tp.alu_lanes, horner_packed_steps=tp.horner_packed_steps)]
        for e in self.non_primitives:
            if e.op_type.startswith("poseidon2_perm/") and e.op_type.endswith("_d4_w32") and self.ext_degree == 4:
                out.append(dict(kind=5, lanes=1))   # the width-32 table of the arity-4 MMCS (P3R_AIR_POSEIDON2_W32)
            elif e.op_type.startswith("poseidon2_perm/") and (self.ext_degree == 4 or e.op_type.endswith("_d1_w16")):
                out.append(dict(kind=3, lanes=1))
            elif e.op_type in ("recompose", "recompose/coeff"):
                out.append(dict(kind=4, lanes=e.lanes, coeff_lookups=1 if e.op_type == "recompose/coeff" else 0))
            else:
                raise P3rError(-5, "MissingTableProver(%s)" % e.op_type)
        return out

    @classmethod
    def from_postcard(cls, data: bytes, field: str, canonical_field_encoding=False, proof_layout=None,
                      challenge_degree=4, zk=False, salted=False) -> "BatchStarkProof":
        """Inverse of `to_postcard`: one pass of the C-ABI parser (p3r_batch_stark_proof_parse: framing of the
        inner `BatchProof`, the metadata that follows it, and the `validate()` rules) - a parent node of the
        aggregation tree runs this on each child, so it is native host code, not a Python loop.
        `proof_layout`: the 18 bytes of `p3r_config.proof_layout` when the proof was written with one.
        `challenge_degree`: 5 for a proof over KoalaBear's quintic challenge field (five words per extension
        element; P3R_PROOF_QUINTIC_CHALLENGE).  `zk`: the proof is a hiding PCS's (p3r_config.zk; P3R_PROOF_ZK) - its
        opening proof is the tuple (random opened values, FriProof), which the bytes do not announce.  `salted`: the MMCSs are
        hiding ones (p3r_config.mmcs_salt_elems > 0; P3R_PROOF_SALTED): every MMCS opening proof is (salts, siblings)."""
        from .device import FIELD_IDS, MODULUS
        lib = _lib.load()
        m, err = _lib.P3rBatchStarkMeta(), C.create_string_buffer(256)
        lay = None if proof_layout is None else (C.c_uint8 * 18)(*[int(v) for v in proof_layout])
        data = bytes(data)
        flags = (1 if canonical_field_encoding else 0) | (2 if challenge_degree == 5 else 0) | (4 if zk else 0) | (8 if salted else 0)
        rc = lib.p3r_batch_stark_proof_parse(FIELD_IDS[field], data, len(data), flags, lay, C.byref(m), err, len(err))
        if rc != 0:
            raise P3rError(rc, err.value.decode())
        entries = tuple(NonPrimitiveTableEntry(op_type=e.op_type.decode(), rows=int(e.rows), lanes=int(e.lanes),
                                               public_values=tuple(e.public_values[:e.n_public_values]),
                                               air_variant=int(e.air_variant))
                        for e in m.non_primitives[:m.n_non_primitives])
        npo_lanes = {e.op_type.decode(): int(e.lanes) for e in m.npo_lanes[:m.n_npo_lanes]}
        recompose_lanes = next((e.lanes for e in entries if e.op_type in ("recompose", "recompose/coeff")),
                               npo_lanes.get("recompose", npo_lanes.get("recompose/coeff", 1)))
        commitment = None
        if m.has_stark_common:
            commitment = np.frombuffer(m, dtype=np.uint32, count=8 * m.cap_len,
                                       offset=_lib.P3rBatchStarkMeta.commitment.offset).reshape(-1, 8).copy()
        return cls(proof=data[:m.proof_len],
                   table_packing=TablePacking(public_lanes=m.public_lanes, alu_lanes=m.alu_lanes,
                                              horner_packed_steps=m.horner_packed_steps, recompose_lanes=recompose_lanes,
                                              min_trace_height=m.min_trace_height),
                   rows=tuple(int(r) for r in m.rows), alu_variant=int(m.alu_variant), ext_degree=int(m.ext_degree),
                   w_binomial=int(m.w_binomial) if m.has_w_binomial else None,
                   alu_quintic_trinomial=bool(m.alu_quintic_trinomial), non_primitives=entries,
                   preprocessed_commitment=commitment,
                   preprocessed_widths=tuple(m.preprocessed_widths[:m.n_instances]) if m.has_stark_common else (),
                   degree_bits=tuple(m.degree_bits[:m.n_instances]) if m.has_stark_common else (),
                   monty_r=0 if canonical_field_encoding else 1, modulus=MODULUS[field], parse_ns=int(m.parse_ns))

    def to_postcard(self) -> bytes:
        """postcard bytes of the whole `BatchStarkProof<SC>`, field order = the serde derives of
        batch_stark_prover.rs:610-636, packing.rs:9-27, :459-460 (RowCounts), :272-290,
        :505-511 (SerializedStarkCommon), :495-500."""
        tp = self.table_packing
        out = bytearray(self.proof)
        # TablePacking { public_lanes, alu_lanes, npo_lanes: Vec<(NpoTypeId, usize)>, min_trace_height, horner_packed_steps }
        out += _varint(tp.public_lanes) + _varint(tp.alu_lanes)
        npo = [(e.op_type, e.lanes) for e in self.non_primitives if e.lanes != 1]
        out += _varint(len(npo))
        for name, lanes in npo:
            out += _pc_str(name) + _varint(lanes)
        out += _varint(tp.min_trace_height) + _varint(tp.horner_packed_steps)
        # RowCounts([usize; 3]) - fixed-size array: no length prefix
        for r in self.rows:
            out += _varint(max(int(r), 1))
        out += _varint(self.alu_variant)               # unit-variant enum -> variant index
        out += _varint(self.ext_degree)
        out += (b"\x01" + self._fe(self.w_binomial)) if self.w_binomial is not None else b"\x00"
        out += b"\x01" if self.alu_quintic_trinomial else b"\x00"
        out += _varint(len(self.non_primitives))
        for e in self.non_primitives:
            out += _pc_str(e.op_type) + _varint(e.rows) + _varint(e.lanes)
            out += _varint(len(e.public_values)) + b"".join(self._fe(int(v)) for v in e.public_values)
            out += _varint(e.air_variant)
        # stark_common: Option<SerializedStarkCommon { commitment, instances: Vec<Option<meta>>, matrix_to_instance }>
        if self.preprocessed_commitment is None:
            out += b"\x00"
        else:
            cap = np.asarray(self.preprocessed_commitment, dtype=np.uint64).reshape(-1, 8)
            out += b"\x01" + _varint(cap.shape[0])
            for d in cap:
                out += b"".join(self._fe(int(v)) for v in d)
            out += _varint(len(self.preprocessed_widths))
            for i, (w, db) in enumerate(zip(self.preprocessed_widths, self.degree_bits)):
                out += b"\x01" + _varint(i) + _varint(w) + _varint(db)
            out += _varint(len(self.preprocessed_widths))
            for i in range(len(self.preprocessed_widths)):
                out += _varint(i)
        return bytes(out)


W_BINOMIAL = {"koala-bear": 3, "baby-bear": 11}


def verify_all_tables(cfg, proof: BatchStarkProof, canonical_field_encoding=None, field=None):
    """Verify a `BatchStarkProof` against the preprocessed commitment and per-instance metadata it binds
    itself to (stark_common); `cfg` is the `p3r_config` the proof was made with.  No GPU needed.
    The proof's extension metadata must be the verifier's (batch_stark_prover.rs:1245-1263:
    ExtDegreeMismatch, BinomialWMismatch, QuinticReductionMismatch)."""
    if proof.preprocessed_commitment is None:
        raise P3rError(-1, "proof carries no preprocessed commitment (stark_common)")
    want_d = int(cfg.ext_degree)   # the verifier's expected trace element field EF
    if proof.ext_degree != want_d:
        raise P3rError(-1, "ExtDegreeMismatch: proof has ext_degree %d, the verifier expects %d" % (proof.ext_degree, want_d))
    field = field or {0: "koala-bear", 1: "baby-bear"}[int(cfg.field)]
    # EF = BinomialExtensionField<F, 4>: W = the field's; EF = QuinticTrinomialExtensionField<F>: no binomial W
    # and the trinomial reduction flag (field_params.rs:54-66)
    # D = 1: the base field has no W either; D = 2 / 6 / 8: the verifier's own W (p3r_config.ext_w)
    want_w = W_BINOMIAL[field] if want_d == 4 else int(cfg.ext_w) if want_d in (2, 6, 8) else None
    if proof.w_binomial != want_w:
        raise P3rError(-1, "BinomialWMismatch: proof has W = %r, the verifier expects %r" % (proof.w_binomial, want_w))
    if bool(proof.alu_quintic_trinomial) != (want_d == 5):
        raise P3rError(-1, "QuinticReductionMismatch: proof quintic-trinomial flag %s does not match the verifier's expected "
                       "trace field (%s)" % (bool(proof.alu_quintic_trinomial), want_d == 5))
    airs = proof.airs()
    if len(proof.preprocessed_widths) != len(airs) or len(proof.degree_bits) != len(airs):
        raise P3rError(-1, "InvalidProofShape: %d AIRs, %d preprocessed widths, %d degree_bits"
                       % (len(airs), len(proof.preprocessed_widths), len(proof.degree_bits)))
    canonical = (proof.monty_r == 0) if canonical_field_encoding is None else canonical_field_encoding
    verify_batch(cfg, airs, proof.preprocessed_commitment, proof.degree_bits, proof.proof, canonical)


class BatchStarkProver:
    def __init__(self, ctx: Context, table_packing: Optional[TablePacking] = None):
        self.ctx = ctx
        self.table_packing = table_packing or TablePacking().with_fri_params(
            ctx.cfg.log_final_poly_len, ctx.cfg.log_blowup)

    def _call(self, fn, *args):
        return self.ctx._proof_call(fn, *args)

    def prove_all_tables(self, traces, circuit_prover_data: CircuitProverData,
                         canonical_field_encoding=False) -> BatchStarkProof:
        """traces: `Traces` (host) or `ResidentTraces` (already in HBM)."""
        flags = 1 if canonical_field_encoding else 0
        ctx = self.ctx
        if isinstance(traces, ResidentTraces):
            raw = self._call(ctx.lib.p3r_prove_all_tables_resident, ctx.h, circuit_prover_data.h, traces.h, flags)
        else:
            t, keep = _traces_struct(traces, ctx.ext_degree)
            raw = self._call(ctx.lib.p3r_prove_all_tables, ctx.h, circuit_prover_data.h, C.byref(t), flags)
        return self.wrap_proof(raw, circuit_prover_data, canonical_field_encoding)

    def wrap_proof(self, raw: bytes, cpd: CircuitProverData, canonical_field_encoding=False) -> BatchStarkProof:
        """The `BatchStarkProof` around the inner `BatchProof` bytes of a layer: the metadata fields the
        reference fills next to `proof` (batch_stark_prover.rs:610-636, 1597-1641)."""
        ctx = self.ctx
        tp = cpd.effective_packing
        # circuit/src/ops/npo.rs:38, poseidon2_perm/config.rs:413-427: the D4 table, or the compact-D1 one of a D = 5 circuit
        p2_name = "poseidon2_perm/%s_%s_w16" % (ctx.field.replace("-", "_"), "d4" if ctx.ext_degree == 4 else "d1")   # D = 1, 5: D1
        k = tp.horner_packed_steps
        # proof order: [Const, Public, Alu, Poseidon2, Poseidon2-W32, Recompose, Recompose/coeff]
        heights = list(cpd.table_heights[:4]) + [getattr(cpd, "p2w_height", 0)] + [cpd.table_heights[4]] + \
            [getattr(cpd, "recompose_coeff_height", 0)]
        present = [h > 0 for h in heights]
        coeff = getattr(cpd, "recompose_coeff_lookups", False)
        prep_widths = (2, 2 * tp.public_lanes, 13 * tp.alu_lanes + 7 * (k - 1), 24 if ctx.ext_degree == 4 else 62, 48,
                       (2 + (2 * ctx.ext_degree if coeff else 0)) * tp.recompose_lanes,
                       (2 + 2 * ctx.ext_degree) * tp.recompose_lanes)
        # non-primitive tables without rows are not proved (poseidon2.rs:1089-1092, recompose.rs:77-80)
        npo = []
        if present[3]:
            # Poseidon2Prover reports the PADDED row count (poseidon2.rs:1449)
            npo.append(NonPrimitiveTableEntry(op_type=p2_name, rows=cpd.table_heights[3], lanes=1))
        if present[4]:
            # the width-32 table of the arity-4 MMCS (Poseidon2Config::*_D4_W32; NpoTypeId poseidon2_perm/<field>_d4_w32)
            npo.append(NonPrimitiveTableEntry(op_type="poseidon2_perm/%s_d4_w32" % ctx.field.replace("-", "_"),
                                              rows=cpd.p2w_height, lanes=1))
        if present[5]:
            # RecomposeProver reports the op count (recompose.rs:125)
            # circuit/src/ops/npo.rs:48-60: "recompose", or "recompose/coeff" for the per-coefficient variant
            npo.append(NonPrimitiveTableEntry(op_type="recompose/coeff" if coeff else "recompose",
                                              rows=cpd.rows["recompose"], lanes=tp.recompose_lanes))
        if present[6]:
            # both table provers are registered: `recompose`, then `recompose/coeff` (batch_stark_prover.rs:1924-1928)
            npo.append(NonPrimitiveTableEntry(op_type="recompose/coeff", rows=cpd.rows["recompose_coeff"],
                                              lanes=tp.recompose_lanes))
        return BatchStarkProof(
            proof=raw, table_packing=tp,
            rows=(cpd.rows["const"], cpd.rows["public"], cpd.rows["alu"]),
            ext_degree=ctx.ext_degree,
            w_binomial=W_BINOMIAL[ctx.field] if ctx.ext_degree == 4 else ctx.ext_w if ctx.ext_degree in (2, 6, 8) else None,
            alu_quintic_trinomial=ctx.ext_degree == 5,
            non_primitives=tuple(npo),
            preprocessed_commitment=cpd.preprocessed_commitment,
            preprocessed_widths=tuple(w for w, ok in zip(prep_widths, present) if ok),
            # ZK: the preprocessed metadata holds the EXTENDED degree bits (recursion.rs:374: `d + config.is_zk()`)
            degree_bits=tuple(int(h).bit_length() - 1 + int(getattr(ctx, "zk", 0)) for h in heights if h > 0),
            monty_r=0 if canonical_field_encoding else 1, modulus=ctx.p)

    def verify_all_tables(self, proof: BatchStarkProof, canonical_field_encoding=None):
        """circuit-prover/src/batch_stark_prover.rs:1230-1268: re-check the proof metadata, rebuild the
        AIR list from it and run `verify_batch` (native host code, include/p3r.h).  Raises P3rError
        with the reason on rejection."""
        proof.validate()
        verify_all_tables(self.ctx.cfg, proof, canonical_field_encoding)

    def build_main_trace(self, resident: ResidentTraces, cpd: CircuitProverData, table: int) -> DeviceMatrix:
        return DeviceMatrix(self.ctx, self.ctx.ptr(
            self.ctx.lib.p3r_layer_build_main_trace(self.ctx.h, cpd.h, resident.h, table)))


# ----------------------------------------------------------------------------- circuit boundary
NO_WITNESS = 0xFFFFFFFF
(OP_CONST, OP_PUBLIC, OP_ALU_ADD, OP_ALU_MUL, OP_ALU_BOOL_CHECK, OP_ALU_MUL_ADD, OP_ALU_HORNER_ACC,
 OP_HINT_EXT_DECOMPOSITION, OP_HINT_BINARY_DECOMPOSITION, OP_POSEIDON2_PERM, OP_RECOMPOSE) = range(11)


@dataclass
class Circuit:
    """Flattened `Circuit<EF>` (circuit/src/circuit.rs:152-181).  `ops` is (n, 8) uint32 in execution
    order: [kind, a, b, c, out, aux, ext_off, ext_len], see `p3r_op_kind` in include/p3r.h."""
    witness_count: int
    ops: np.ndarray
    ext: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    public_rows: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    private_input_rows: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    witness_rewrite: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.uint32))

    @property
    def public_flat_len(self):
        return len(np.asarray(self.public_rows).reshape(-1))

    @property
    def private_flat_len(self):
        return len(np.asarray(self.private_input_rows).reshape(-1))

    def runner(self, prepared: "PreparedCircuit") -> "CircuitRunner":
        return CircuitRunner(prepared)


@dataclass
class CircuitInputs:
    public_values: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.uint32))
    private_values: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.uint32))
    private_data_op_ids: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    private_data_siblings: np.ndarray = field(default_factory=lambda: np.zeros((0, 8), np.uint32))
    # private data of width-32 Merkle rows (P3R_OP_POSEIDON2_W32_PERM): the three sibling digests of an arity-4 level
    private_data_w32_op_ids: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    private_data_w32_siblings: np.ndarray = field(default_factory=lambda: np.zeros((0, 24), np.uint32))


def _flat32(a):
    x = np.ascontiguousarray(np.asarray(a, dtype=np.uint32).reshape(-1))
    return x, x.ctypes.data_as(_lib.u32p)


class PreparedCircuit:
    """A circuit prepared for repeated proving: preprocessed columns (generate_preprocessed_columns +
    get_airs_and_degrees_with_prep), their commitment, and the levelised execution schedule."""

    def __init__(self, ctx: Context, circuit: Circuit, packing: TablePacking):
        packing.validate()
        self.ctx, self.circuit, self.packing = ctx, circuit, packing
        d = _lib.P3rCircuitDesc()
        ops, _ = _flat32(circuit.ops)
        if ops.size % 8:
            raise ValueError("ops must be (n, 8)")
        keep = [ops]
        d.witness_count = circuit.witness_count
        d.n_ops, d.ops = ops.size // 8, C.cast(ops.ctypes.data_as(_lib.u32p), C.POINTER(_lib.P3rOp))
        for name, attr in (("ext", "ext"), ("public", "public_rows"), ("private", "private_input_rows"),
                           ("rewrite", "witness_rewrite")):
            x, ptr = _flat32(getattr(circuit, attr))
            keep.append(x)
            n = x.size // 2 if name == "rewrite" else x.size
            setattr(d, "n_" + name, n)
            setattr(d, {"ext": "ext", "public": "public_rows", "private": "private_input_rows",
                        "rewrite": "witness_rewrite"}[name], ptr)
        d.public_lanes, d.alu_lanes = packing.public_lanes, packing.alu_lanes
        d.horner_packed_steps, d.recompose_lanes = packing.horner_packed_steps, packing.recompose_lanes
        d.min_trace_height = packing.min_trace_height
        commit = np.empty((1 << ctx.cap_height, 8), dtype=np.uint32)
        self.h = ctx.ptr(ctx.lib.p3r_circuit_create(ctx.h, C.byref(d), commit.ctypes.data_as(_lib.u32p)))
        cn = _lib.P3rLayerCounts()
        ctx.check(ctx.lib.p3r_circuit_counts(self.h, C.byref(cn)))
        rows = dict(const=cn.n_const, public=cn.n_public, alu=cn.n_alu, poseidon2=cn.n_p2, recompose=cn.n_recompose,
                    recompose_coeff=cn.n_recompose_coeff, poseidon2_w32=cn.n_p2w)
        lv = C.c_size_t()
        ctx.check(ctx.lib.p3r_circuit_levels(self.h, C.byref(lv)))
        self.levels = lv.value
        self.prepared_on_device = bool(ctx.lib.p3r_circuit_prepared_on_device(self.h))
        self._cpd_args = (ctx.lib.p3r_circuit_layer(self.h), packing, rows, commit)
        self._cpd_view = None


    @property
    def circuit_prover_data(self) -> CircuitProverData:
        """The CircuitProverData this circuit was prepared into.  Views hold a reference to the prepared circuit
        (not the other way round: no reference cycle), so the device data lives exactly as long as the circuit,
        a cache slot or any RecursionOutput / CircuitProverData handed out still refers to it."""
        import weakref
        view = self._cpd_view() if self._cpd_view is not None else None
        if view is None:
            if not self.h:
                raise P3rError(-1, "the prepared circuit has been freed")
            view = CircuitProverData._borrow(self.ctx, *self._cpd_args, owner=self)
            self._cpd_view = weakref.ref(view)
        return view

    @staticmethod
    def _inputs_struct(circuit: Circuit, inputs: CircuitInputs, d: int = 4):
        t = _lib.P3rCircuitInputs()
        pub, t.public_values = _flat32(inputs.public_values)
        prv, t.private_values = _flat32(inputs.private_values)
        # set_public_inputs / set_private_inputs length checks (runner.rs:84-90,107-113); d coefficients per input
        if pub.size != d * circuit.public_flat_len:
            raise P3rError(-1, "PublicInputLengthMismatch { expected: %d, got: %d }" % (circuit.public_flat_len, pub.size // d))
        if prv.size != d * circuit.private_flat_len:
            raise P3rError(-1, "PrivateInputLengthMismatch { expected: %d, got: %d }" % (circuit.private_flat_len, prv.size // d))
        ids, t.private_data_op_ids = _flat32(inputs.private_data_op_ids)
        sib, t.private_data_siblings = _flat32(inputs.private_data_siblings)
        if sib.size != 8 * ids.size:
            raise ValueError("private_data_siblings must hold two extension limbs (8 values) per op id")
        t.n_private_data = ids.size
        idw, t.private_data_w32_op_ids = _flat32(inputs.private_data_w32_op_ids)
        sibw, t.private_data_w32_siblings = _flat32(inputs.private_data_w32_siblings)
        if sibw.size != 24 * idw.size:
            raise ValueError("private_data_w32_siblings must hold three digests of two extension limbs (24 values) per op id")
        t.n_private_data_w32 = idw.size
        return t, (pub, prv, ids, sib, idw, sibw)

    def upload_inputs(self, inputs: CircuitInputs) -> "ResidentInputs":
        return ResidentInputs(self, inputs)

    def run(self, inputs) -> ResidentTraces:
        """`CircuitRunner::run` on the device (circuit/src/tables/runner.rs:195-253).
        inputs: `CircuitInputs` (host) or `ResidentInputs` (already in HBM)."""
        if isinstance(inputs, ResidentInputs):
            h = self.ctx.ptr(self.ctx.lib.p3r_circuit_run_resident(self.ctx.h, self.h, inputs.h))
        else:
            t, keep = self._inputs_struct(self.circuit, inputs, self.ctx.ext_degree)
            h = self.ctx.ptr(self.ctx.lib.p3r_circuit_run(self.ctx.h, self.h, C.byref(t)))
        return ResidentTraces._adopt(self.ctx, self.circuit_prover_data, h)

    def prove(self, inputs, canonical_field_encoding=False) -> bytes:
        """Run + prove_all_tables in one call (the inner BatchProof bytes)."""
        flags = 1 if canonical_field_encoding else 0
        if isinstance(inputs, ResidentInputs):
            return self.ctx._proof_call(self.ctx.lib.p3r_prove_next_layer_resident, self.ctx.h, self.h, inputs.h, flags)
        t, keep = self._inputs_struct(self.circuit, inputs, self.ctx.ext_degree)
        return self.ctx._proof_call(self.ctx.lib.p3r_prove_next_layer, self.ctx.h, self.h, C.byref(t), flags)

    def free(self):
        """Explicit release (device memory is returned now): outstanding CircuitProverData views become invalid."""
        view = self._cpd_view() if getattr(self, "_cpd_view", None) is not None else None
        if view is not None:
            view.h = None
        if self.h and self.ctx.h:
            self.ctx.lib.p3r_circuit_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ResidentInputs:
    """`CircuitInputs` uploaded to HBM once."""

    def __init__(self, prepared: PreparedCircuit, inputs: CircuitInputs):
        self.ctx = prepared.ctx
        t, keep = PreparedCircuit._inputs_struct(prepared.circuit, inputs, prepared.ctx.ext_degree)
        self.h = self.ctx.ptr(self.ctx.lib.p3r_circuit_inputs_upload(self.ctx.h, prepared.h, C.byref(t)))

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.p3r_circuit_inputs_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CircuitRunner:
    """The reference's runner surface (circuit/src/tables/runner.rs:22-253) over a prepared circuit."""

    def __init__(self, prepared: PreparedCircuit):
        self.prepared = prepared
        self._inputs = CircuitInputs()
        self._pd = {}

    def set_public_inputs(self, values):
        self._inputs.public_values = np.asarray(values, dtype=np.uint32).reshape(-1, self.prepared.ctx.ext_degree)

    def set_private_inputs(self, values):
        self._inputs.private_values = np.asarray(values, dtype=np.uint32).reshape(-1, self.prepared.ctx.ext_degree)

    def set_private_data(self, op_id: int, sibling):
        if op_id in self._pd:
            raise P3rError(-1, "IncorrectNonPrimitiveOpPrivateData: private data already set for NonPrimitiveOpId(%d)" % op_id)
        self._pd[op_id] = np.asarray(sibling, dtype=np.uint32).reshape(8)

    def run(self) -> ResidentTraces:
        ids = np.array(list(self._pd.keys()), dtype=np.uint32)
        self._inputs.private_data_op_ids = ids
        self._inputs.private_data_siblings = (np.stack([self._pd[int(i)] for i in ids]) if len(ids)
                                              else np.zeros((0, 8), np.uint32))
        return self.prepared.run(self._inputs)


# ----------------------------------------------------------------------------- recursion API
@dataclass
class FriRecursionConfig:
    """recursion/src/backend/fri.rs:41-106: which NPO tables a layer uses."""
    poseidon2: bool = True
    recompose: bool = True
    recompose_lanes: int = 1


class FriRecursionBackend:
    """recursion/src/backend/fri.rs:113-128.  The table provers it registers for D = 4 are the
    Poseidon2 and Recompose tables (:693-721); both are built in to this prover."""

    def __init__(self, config: Optional[FriRecursionConfig] = None):
        self.config = config or FriRecursionConfig()

    def non_primitive_provers(self, ext_degree: int):
        # the D = 4 backend registers its table provers for D = 4 circuits and none for any other degree
        # (fri.rs:693-721: `else { Vec::new() }`): a D = 5 layer is proved from its primitive tables
        return ["poseidon2_perm", "recompose"] if ext_degree == 4 else []


class FriRecursionBackendD5(FriRecursionBackend):
    """recursion/src/backend/fri.rs:741-852: the backend of KoalaBear quintic circuits.  With the D1 permutation
    (`challenger_perm_config.extension_degree() != 5`, hence `cl = true`) it registers the compact-D1 Poseidon2 table and
    BOTH Recompose tables for D = 5 circuits (:816-835), and nothing for any other degree; all three are built in to this
    prover, so a context with ext_degree = 5 proves whatever tables the circuit fills (five or six)."""

    def non_primitive_provers(self, ext_degree: int):
        return ["poseidon2_perm/koala_bear_d1_w16", "recompose", "recompose/coeff"] if ext_degree == 5 else []


def _lib_code(name):
    return {"UNSUPPORTED": -5}[name]


@dataclass
class ProveNextLayerParams:
    table_packing: TablePacking = field(default_factory=TablePacking)


@dataclass
class RecursionInput:
    """What `prove_next_layer` proves.  Either the inputs of the verifier circuit (public inputs +
    the Merkle siblings `set_fri_private_data` extracts from `prev`, recursion.rs:455-476), in which
    case the circuit is run on the device, or already-computed `Traces`."""
    traces: Optional[Traces] = None
    prev_proof: Optional[BatchStarkProof] = None
    circuit_inputs: Optional[CircuitInputs] = None


@dataclass
class RecursionOutput:
    proof: BatchStarkProof
    circuit_prover_data: CircuitProverData

    def into_recursion_input(self, next_traces: Traces) -> RecursionInput:
        return RecursionInput(traces=next_traces, prev_proof=self.proof)


@dataclass
class NextLayerPrepCache:
    prover: BatchStarkProver
    circuit_prover_data: CircuitProverData
    prepared_circuit: Optional[PreparedCircuit] = None


def build_next_layer_prep(ctx: Context, circuit, backend: FriRecursionBackend,
                          params: ProveNextLayerParams) -> NextLayerPrepCache:
    """recursion.rs:342-394.  `circuit` is a `Circuit` (preprocessed columns are derived here, as
    get_airs_and_degrees_with_prep does) or an already-flattened `CircuitPrep`."""
    backend.non_primitive_provers(ctx.ext_degree)
    prover = BatchStarkProver(ctx, params.table_packing)
    if isinstance(circuit, Circuit):
        pc = PreparedCircuit(ctx, circuit, params.table_packing)
        return NextLayerPrepCache(prover=prover, circuit_prover_data=pc.circuit_prover_data, prepared_circuit=pc)
    cpd = CircuitProverData(ctx, circuit, params.table_packing)
    return NextLayerPrepCache(prover=prover, circuit_prover_data=cpd)


def prove_next_layer(inp: RecursionInput, ctx: Context, backend: FriRecursionBackend, params: ProveNextLayerParams,
                     prep: Optional[NextLayerPrepCache] = None,
                     circuit_prep: Optional[CircuitPrep] = None) -> RecursionOutput:
    """One recursion layer.  With `prep` (the fast path, recursion.rs:426-450) only the proof is
    computed; without it the preprocessed commitment is rebuilt first (recursion.rs:452-501)."""
    if prep is None:
        if circuit_prep is None:
            raise ValueError("prove_next_layer needs either a NextLayerPrepCache or the circuit to build one")
        prep = build_next_layer_prep(ctx, circuit_prep, backend, params)
    if inp.traces is not None:
        traces = inp.traces
    else:
        if prep.prepared_circuit is None or inp.circuit_inputs is None:
            raise ValueError("without Traces, prove_next_layer needs a prepared Circuit and its inputs")
        # runner.run() (recursion.rs:478) + prove_all_tables as ONE call (p3r_prove_next_layer): the
        # prover is enqueued behind the circuit run without waiting for it
        raw = prep.prepared_circuit.prove(inp.circuit_inputs)
        return RecursionOutput(proof=prep.prover.wrap_proof(raw, prep.circuit_prover_data),
                               circuit_prover_data=prep.circuit_prover_data)
    proof = prep.prover.prove_all_tables(traces, prep.circuit_prover_data)
    return RecursionOutput(proof=proof, circuit_prover_data=prep.circuit_prover_data)



# ----------------------------------------------------------------------------- 2-to-1 aggregation
@dataclass(frozen=True)
class AggregationCircuitFingerprint:
    """recursion.rs:72-93: rejects `AggregationPrepCache` hits when a later aggregation step compiles to
    a different verification circuit even though params and config match."""
    witness_count: int
    public_flat_len: int
    private_flat_len: int
    ops_len: int
    # The reference keys the slot by the four lengths and then runs the NEW circuit against the cached prover
    # data (a same-shape, different-content circuit fails to verify there).  Here a hit also reuses the cached
    # execution schedule, so the key additionally binds the circuit's content.
    content_digest: bytes = b""


def _circuit_content_digest(circuit: Circuit) -> bytes:
    """128-bit digest of ops / ext / public_rows / private_input_rows / witness_rewrite (cached on the object)."""
    cached = getattr(circuit, "_content_digest", None)
    arrays = [np.ascontiguousarray(np.asarray(getattr(circuit, n), dtype=np.uint32).reshape(-1))
              for n in ("ops", "ext", "public_rows", "private_input_rows", "witness_rewrite")]
    key = tuple((a.__array_interface__["data"][0], a.size) for a in arrays)
    if cached is not None and cached[0] == key:
        return cached[1]
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except ImportError:  # pragma: no cover - xxhash ships with the image
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        h.update(np.uint64(a.size).tobytes())
        h.update(memoryview(a).cast("B"))
    d = h.digest()
    try:
        object.__setattr__(circuit, "_content_digest", (key, d))
    except Exception:
        pass
    return d


def aggregation_circuit_fingerprint(circuit: Circuit) -> AggregationCircuitFingerprint:
    return AggregationCircuitFingerprint(
        witness_count=int(circuit.witness_count),
        public_flat_len=int(np.asarray(circuit.public_rows).size),
        private_flat_len=int(np.asarray(circuit.private_input_rows).size),
        ops_len=int(np.asarray(circuit.ops).reshape(-1, 8).shape[0]),
        content_digest=_circuit_content_digest(circuit))


@dataclass
class AggregationPrepCache:
    """recursion.rs:95-99 (+ the prepared circuit: the levelised schedule of the device runner)."""
    circuit_fingerprint: AggregationCircuitFingerprint
    circuit_prover_data: CircuitProverData
    prover: BatchStarkProver
    prepared_circuit: PreparedCircuit


def pack_aggregation_inputs(left: CircuitInputs, right: CircuitInputs, left_non_primitive_ops: int) -> CircuitInputs:
    """The inputs of the aggregation circuit from the two halves it verifies: the left proof's public /
    private values and Merkle siblings first, then the right proof's, whose non-primitive op ids are
    offset by the number of non-primitive ops the left verifier contributes - the packing
    `run_aggregation_verification_circuit` does for the two `VerifierResult`s (recursion.rs:596-640)."""
    def cat(a, b, w):
        return np.concatenate([np.asarray(a, np.uint32).reshape(-1, w), np.asarray(b, np.uint32).reshape(-1, w)])
    ids = np.concatenate([np.asarray(left.private_data_op_ids, np.uint32).reshape(-1),
                          np.asarray(right.private_data_op_ids, np.uint32).reshape(-1) + np.uint32(left_non_primitive_ops)])
    return CircuitInputs(public_values=cat(left.public_values, right.public_values, 4),
                         private_values=cat(left.private_values, right.private_values, 4),
                         private_data_op_ids=ids,
                         private_data_siblings=cat(left.private_data_siblings, right.private_data_siblings, 8),
                         private_data_w32_op_ids=np.concatenate([
                             np.asarray(left.private_data_w32_op_ids, np.uint32).reshape(-1),
                             np.asarray(right.private_data_w32_op_ids, np.uint32).reshape(-1) + np.uint32(left_non_primitive_ops)]),
                         private_data_w32_siblings=cat(left.private_data_w32_siblings, right.private_data_w32_siblings, 24))


def prove_aggregation_layer(left: RecursionInput, right: RecursionInput, verification_circuit: Circuit, ctx: Context,
                            backend: FriRecursionBackend, params: ProveNextLayerParams,
                            prep_cache: Optional[list] = None, left_non_primitive_ops: int = 0) -> RecursionOutput:
    """recursion.rs:656-762: one node of the 2-to-1 tree.  `verification_circuit` verifies both inputs;
    `left.circuit_inputs` / `right.circuit_inputs` are each side's share of its inputs (what the
    verifier results pack from the two proofs).  `prep_cache` is the reference's
    `Option<&mut Option<AggregationPrepCache>>` as a one-element list: pass `[None]` on the first call,
    the slot is filled and reused while the circuit fingerprint stays the same, and ignored (then
    replaced) when it changes."""
    if left.circuit_inputs is None or right.circuit_inputs is None:
        raise ValueError("prove_aggregation_layer needs the circuit inputs of both sides")
    fp = aggregation_circuit_fingerprint(verification_circuit)
    inputs = pack_aggregation_inputs(left.circuit_inputs, right.circuit_inputs, left_non_primitive_ops)
    cached = prep_cache[0] if prep_cache else None
    if cached is not None and cached.circuit_fingerprint == fp:
        # run + prove_all_tables as one call: the prover does not wait for the circuit run
        proof = cached.prover.wrap_proof(cached.prepared_circuit.prove(inputs), cached.circuit_prover_data)
        return RecursionOutput(proof=proof, circuit_prover_data=cached.circuit_prover_data)
    backend.non_primitive_provers(4)
    prover = BatchStarkProver(ctx, params.table_packing)
    pc = PreparedCircuit(ctx, verification_circuit, params.table_packing)   # get_airs_and_degrees_with_prep + ProverData
    proof = prover.wrap_proof(pc.prove(inputs), pc.circuit_prover_data)
    cpd = pc.circuit_prover_data
    if prep_cache is not None:
        if not prep_cache:
            prep_cache.append(None)
        # the replaced entry is NOT freed here: RecursionOutputs returned earlier may still hold its
        # CircuitProverData (an Rc in the reference); it is released when the last of them goes away
        prep_cache[0] = AggregationPrepCache(fp, cpd, prover, pc)
    return RecursionOutput(proof=proof, circuit_prover_data=cpd)


# ----------------------------------------------------------------------------- tracing spans
# The reference instruments its hot path with `tracing` spans named after the functions
# (`#[instrument(skip_all)]`: recursion.rs:400 `prove_next_layer`, batch_stark_prover.rs:1202
# `prove_all_tables`, tables/runner.rs:194 `run`, and the per-AIR trace builders alu_air.rs:496,
# const_air.rs:90, public_air.rs:127, recompose_air.rs:94, poseidon2-circuit-air air.rs:279); the
# examples print them with tracing_forest as `name [ 109ms | 100.00% ]` and scripts/benchmark.sh:87-101
# parses those lines.  span_report() renders the ctx's stage / kernel timers in the same shape under
# the same names; the stages inside prove_batch (an un-vendored crate) keep descriptive names.
_SPAN_TREE = (
    ("run", ("stage", "run_circuit"), ()),
    ("prove_all_tables", None, (
        ("AluAir::trace_to_matrix", ("kernel", "alu_trace"), ()),
        ("ConstAir::build_trace + WitnessSendAir::build_trace + RecomposeAir::build_trace", ("kernel", "trace_to_matrix"), ()),
        ("Poseidon2CircuitAir::build_trace", ("kernel", "p2_acc_scan", "p2_trace_fill"), ()),
        ("prove_batch", None, (
            ("commit to main traces", ("stage", "main_lde_commit"), ()),
            ("observe instance shapes and commitments", ("stage", "transcript_head"), ()),
            ("LogUp permutation traces and commit", ("stage", "logup_aux_commit"), ()),
            ("compute and commit quotient", ("stage", "quotient_commit"), ()),
            ("open", ("stage", "openings"), ()),
            ("FRI reduced openings", ("stage", "fri_reduce"), ()),
            ("FRI commit phase", ("stage", "fri_commit_phase"), ()),
            ("FRI query phase", ("stage", "queries"), ()),
            ("serialize proof", ("stage", "serialize"), ()),
        )),
    )),
)


def span_report(profile: dict, steps: int = 1, root: str = "prove_next_layer") -> str:
    """`profile` = Context.profile_read() after `steps` profiled prove_next_layer calls."""
    def own(src):
        if src is None:
            return None
        kind, names = src[0], src[1:]
        return sum(profile.get(("stage:" + n) if kind == "stage" else n, (0.0, 0))[0] for n in names) / steps

    def total(node):
        name, src, kids = node
        t = own(src)
        return t if t is not None else sum(total(k) for k in kids)

    whole = sum(total(n) for n in _SPAN_TREE) or 1e-12
    lines = ["%s [ %s | 100.00%% ]" % (root, _fmt_ms(whole))]

    def emit(nodes, prefix):
        for i, node in enumerate(nodes):
            last = i == len(nodes) - 1
            t = total(node)
            lines.append("%s%s %s [ %s | %.2f%% ]" % (prefix, "\u2515\u2501" if last else "\u251d\u2501", node[0], _fmt_ms(t),
                                                   100.0 * t / whole))
            emit(node[2], prefix + ("   " if last else "\u2502  "))
    emit(_SPAN_TREE, "")
    return "\n".join(lines)


def _fmt_ms(ms: float) -> str:
    if ms >= 1000.0:
        return "%.2fs" % (ms / 1000.0)
    if ms >= 1.0:
        return "%.2fms" % ms
    return "%.0f\u00b5s" % (ms * 1000.0)
