// Sanitizer build of the HOST-ONLY code of libp3r_hip.so (tests/san/Makefile: hipcc --cuda-host-only
// -fsanitize=address,undefined, no device code is generated or run): the proof parsers, the native verifier,
// Mmcs::verify_batch (csrc/host_abi.h -> verify_impl.h) and the host side of the circuit boundary (csrc/circuit_host.h:
// validate_circuit, the preprocessed columns, the execution schedule).  These run on bytes and circuits that reach a
// parent node of an aggregation tree from other ranks (plonky3_recursion_amd/aggregation.py); the rules they enforce
// are circuit-prover/src/batch_stark_prover.rs:459-488,666-681 and packing.rs:140-161.  Test infrastructure.
#include "../../plonky3_recursion_amd/csrc/host_abi.h"
#include "../../plonky3_recursion_amd/csrc/circuit_host.h"

extern "C" {

// validate_circuit + circuit_tables + build_schedule of one flattened circuit (what p3r_circuit_create does on the host
// before anything touches the device).  0: prepared; 1: refused with a reason (the expected outcome for a malformed
// circuit); anything the sanitizers object to aborts the process.
int san_circuit_host_prep(const p3r_circuit_desc* d, uint32_t field, uint32_t ext_degree, char* err, size_t err_cap) {
  try {
    if (!d) throw std::runtime_error("NULL desc");
    auto need = [&](const void* p, size_t n, const char* what) { if (n && !p) fail(P3R_EINVAL, "%s is NULL", what); };
    need(d->ops, d->n_ops, "ops"); need(d->ext, d->n_ext, "ext"); need(d->public_rows, d->n_public, "public_rows");
    need(d->private_input_rows, d->n_private, "private_input_rows"); need(d->witness_rewrite, d->n_rewrite, "witness_rewrite");
    check_circuit_sizes(*d);
    if (!d->public_lanes || !d->alu_lanes || !d->recompose_lanes) fail(P3R_EINVAL, "lane counts must be positive");
    if (d->horner_packed_steps < 2 || d->horner_packed_steps > 8) fail(P3R_EINVAL, "horner_packed_steps must be in 2..8");
    if (ext_degree != 1 && ext_degree != 4 && ext_degree != 5) fail(P3R_EUNSUPPORTED, "circuit degree %u", ext_degree);
    HostCircuit h;
    h.witness_count = d->witness_count;
    h.ops.assign(d->ops, d->ops + d->n_ops);
    h.ext.assign(d->ext, d->ext + d->n_ext);
    h.public_rows.assign(d->public_rows, d->public_rows + d->n_public);
    h.private_rows.assign(d->private_input_rows, d->private_input_rows + d->n_private);
    h.rewrite.assign(d->witness_rewrite, d->witness_rewrite + 2 * d->n_rewrite);
    validate_circuit(h, ext_degree);
    const RunSchedule S = build_schedule(h, ext_degree);
    if (!S.deferred_error.empty()) throw std::runtime_error(S.deferred_error);
    if (field == P3R_FIELD_KOALA_BEAR) (void)circuit_tables<KoalaBearParams>(h, ext_degree);
    else (void)circuit_tables<BabyBearParams>(h, ext_degree);
    return 0;
  } catch (const std::exception& e) {
    if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
    return 1;
  }
}

}  // extern "C"
