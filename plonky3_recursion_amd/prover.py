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
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _lib
from .device import Context, DeviceMatrix, P3rError, _u32, _u8


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
    const_values: np.ndarray        # (n_const, 4)
    public_values: np.ndarray       # (n_public, 4)
    alu_values: np.ndarray          # (n_alu, 16): [a, b, c, out] x 4 coefficients
    p2_input_values: np.ndarray     # (n_p2, 16)
    p2_new_start: np.ndarray        # (n_p2,) bool
    p2_merkle_path: np.ndarray
    p2_mmcs_bit: np.ndarray
    p2_mmcs_index_sum: np.ndarray   # (n_p2,)
    recompose_values: np.ndarray    # (n_recompose, 4)


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
    p2_in_ctl: np.ndarray           # (n_p2, 4)
    p2_input_indices: np.ndarray    # (n_p2, 4)
    p2_out_ctl: np.ndarray          # (n_p2, 2) multiplicities
    p2_output_indices: np.ndarray   # (n_p2, 2)
    p2_mmcs_index_sum_idx: np.ndarray


class CircuitProverData:
    """Device-resident preprocessed LDEs + commitment + ALU schedule for one circuit shape."""

    def __init__(self, ctx: Context, prep: CircuitPrep, packing: TablePacking):
        packing.validate()
        self.ctx, self.packing = ctx, packing
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
        d.counts.n_recompose = len(prep.recompose_prep)
        d.public_lanes, d.alu_lanes = packing.public_lanes, packing.alu_lanes
        d.horner_packed_steps, d.recompose_lanes = packing.horner_packed_steps, packing.recompose_lanes
        d.min_trace_height = packing.min_trace_height
        d.const_prep, d.public_prep = p32(prep.const_prep), p32(prep.public_prep)
        d.alu_prep13, d.recompose_prep = p32(prep.alu_prep13), p32(prep.recompose_prep)
        d.p2_new_start, d.p2_merkle_path = p8(prep.p2_new_start), p8(prep.p2_merkle_path)
        d.p2_mmcs_ctl_enabled, d.p2_in_ctl = p8(prep.p2_mmcs_ctl_enabled), p8(prep.p2_in_ctl)
        d.p2_input_indices, d.p2_out_ctl = p32(prep.p2_input_indices), p32(prep.p2_out_ctl)
        d.p2_output_indices, d.p2_mmcs_index_sum_idx = p32(prep.p2_output_indices), p32(prep.p2_mmcs_index_sum_idx)
        self.rows = dict(const=d.counts.n_const, public=d.counts.n_public, alu=d.counts.n_alu,
                         poseidon2=d.counts.n_p2, recompose=d.counts.n_recompose)
        self.preprocessed_commitment = np.empty((1 << ctx.cap_height, 8), dtype=np.uint32)
        self.h = ctx.ptr(ctx.lib.p3r_layer_create(ctx.h, C.byref(d),
                                                  self.preprocessed_commitment.ctypes.data_as(_lib.u32p)))
        hs = (C.c_size_t * 5)()
        ctx.check(ctx.lib.p3r_layer_table_heights(self.h, hs))
        self.table_heights = [int(x) for x in hs]

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.p3r_layer_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ResidentTraces:
    """`Traces` uploaded to HBM once (so a prove starts from device-resident inputs)."""

    def __init__(self, ctx: Context, cpd: CircuitProverData, traces: Traces):
        self.ctx = ctx
        t, self._keep = _traces_struct(traces)
        self.h = ctx.ptr(ctx.lib.p3r_traces_upload(ctx.h, cpd.h, C.byref(t)))

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.p3r_traces_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _traces_struct(tr: Traces):
    t = _lib.P3rTraces()
    keep = []

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
    return t, keep


@dataclass
class BatchStarkProof:
    """`proof` holds the postcard bytes of the inner `BatchProof<SC>`; the remaining fields are the
    metadata the reference stores next to it (batch_stark_prover.rs:610-636)."""
    proof: bytes
    table_packing: TablePacking
    rows: dict
    ext_degree: int = 4
    w_binomial: Optional[int] = None
    alu_quintic_trinomial: bool = False
    non_primitives: tuple = ("poseidon2_perm", "recompose")
    preprocessed_commitment: Optional[np.ndarray] = None


W_BINOMIAL = {"koala-bear": 3, "baby-bear": 11}


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
            t, keep = _traces_struct(traces)
            raw = self._call(ctx.lib.p3r_prove_all_tables, ctx.h, circuit_prover_data.h, C.byref(t), flags)
        return BatchStarkProof(proof=raw, table_packing=circuit_prover_data.packing, rows=dict(circuit_prover_data.rows),
                               w_binomial=W_BINOMIAL[ctx.field],
                               preprocessed_commitment=circuit_prover_data.preprocessed_commitment)

    def build_main_trace(self, resident: ResidentTraces, cpd: CircuitProverData, table: int) -> DeviceMatrix:
        return DeviceMatrix(self.ctx, self.ctx.ptr(
            self.ctx.lib.p3r_layer_build_main_trace(self.ctx.h, cpd.h, resident.h, table)))


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
        if ext_degree != 4:
            raise P3rError(_lib_code("UNSUPPORTED"), f"UnsupportedDegree({ext_degree})")
        return ["poseidon2_perm", "recompose"]


def _lib_code(name):
    return {"UNSUPPORTED": -5}[name]


@dataclass
class ProveNextLayerParams:
    table_packing: TablePacking = field(default_factory=TablePacking)


@dataclass
class RecursionInput:
    """What `prove_next_layer` proves: the traces of the verifier circuit run over `prev`.
    (`prev` itself - a UniStark or BatchStark proof - is consumed by the Rust-side circuit runner.)"""
    traces: Traces
    prev_proof: Optional[BatchStarkProof] = None


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


def build_next_layer_prep(ctx: Context, circuit_prep: CircuitPrep, backend: FriRecursionBackend,
                          params: ProveNextLayerParams) -> NextLayerPrepCache:
    backend.non_primitive_provers(4)
    cpd = CircuitProverData(ctx, circuit_prep, params.table_packing)
    return NextLayerPrepCache(prover=BatchStarkProver(ctx, params.table_packing), circuit_prover_data=cpd)


def prove_next_layer(inp: RecursionInput, ctx: Context, backend: FriRecursionBackend, params: ProveNextLayerParams,
                     prep: Optional[NextLayerPrepCache] = None,
                     circuit_prep: Optional[CircuitPrep] = None) -> RecursionOutput:
    """One recursion layer.  With `prep` (the fast path, recursion.rs:426-450) only the proof is
    computed; without it the preprocessed commitment is rebuilt first (recursion.rs:452-501)."""
    if prep is None:
        if circuit_prep is None:
            raise ValueError("prove_next_layer needs either a NextLayerPrepCache or the CircuitPrep to build one")
        prep = build_next_layer_prep(ctx, circuit_prep, backend, params)
    proof = prep.prover.prove_all_tables(inp.traces, prep.circuit_prover_data)
    return RecursionOutput(proof=proof, circuit_prover_data=prep.circuit_prover_data)
