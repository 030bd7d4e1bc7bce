"""MI355X-native batch-STARK prover for the p3-recursion `prove_next_layer` hot path.

Host-side mirror of the reference's interface for this path over the C ABI in include/p3r.h.
The HIP library is mandatory: importing the package is cheap, but any compute entry point
raises if `libp3r_hip.so` has not been built (no CPU fallback).
"""
from .device import Context, DeviceMatrix, MerkleTree, P3rError, make_config, mmcs_verify, verify_batch  # noqa: F401
from .prover import (AggregationCircuitFingerprint, AggregationPrepCache, BatchStarkProof, BatchStarkProver, Circuit, CircuitInputs, CircuitPrep,  # noqa: F401
                     CircuitProverData, CircuitRunner, PreparedCircuit, FriRecursionBackend, FriRecursionBackendD5, FriRecursionConfig, NextLayerPrepCache, ProveNextLayerParams,
                     RecursionInput, RecursionOutput, ResidentTraces, TablePacking, Traces,
                     aggregation_circuit_fingerprint, build_next_layer_prep, pack_aggregation_inputs, prove_aggregation_layer,
                     prove_next_layer, span_report, verify_all_tables)

__all__ = ["Context", "DeviceMatrix", "MerkleTree", "P3rError"]
