#!/usr/bin/env python3
"""bench.py - one JSON line for the driver (DESIGN.md "Measurement").

A "step" is one `prove_next_layer` (recursion/src/recursion.rs:401-502 with a NextLayerPrepCache):
the verifier circuit is RUN on the device (CircuitRunner::run, levelised) and its tables are proved
(`prove_all_tables`), over the synthetic KoalaBear 2^20-row recursion layer of SURVEY.md section 8d,
with the circuit inputs (public values, Merkle siblings) and the cached preprocessed data already
resident in HBM.  It covers the witness fill, K1-K3 trace building, main/permutation/quotient
LDE + MMCS commits, LogUp, quotient evaluation, openings, FRI commit/fold/grind/queries and proof
serialisation.

Multi-GPU: independent proofs (aggregation-tree nodes) shard one per rank with no data-path
collective - weak scaling; only the barrier / max-over-ranks timing
crosses ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec

FRI = dict(log_blowup=2, max_log_arity=2, cap_height=0, log_final_poly_len=5, commit_pow_bits=0,
           query_pow_bits=15, num_queries=54)  # recursive_fibonacci.rs:71-147, examples/common/mod.rs:472
GEN_KNOBS = dict(horner_chain_len=64, sponge_chain_len=8, merkle_depth=20)
# BASELINE config 2 (layer 1 over a uni-stark Keccak proof): same table mix,
# Horner chains of ~2600 steps and sponge chains of ~330 permutations
CONFIG2_KNOBS = dict(horner_chain_len=2600, sponge_chain_len=330, merkle_depth=20)


def lookup_aux_widths(alu_lanes, horner_k):
    """(aux EF columns, quotient chunks) per table, DESIGN.md 'LogUp' packing rule."""
    alu_lookups = 4 * alu_lanes + 2 * (horner_k - 1)
    return {"const": (2, 1), "public": (2, 1), "alu": ((alu_lookups + 1) // 2 + 1, 2), "poseidon2": (5, 2),
            "recompose": (2, 1)}


def workload_model(field, heights, widths, packing):
    """Analytic per-prove counts: Poseidon2 permutations and algorithmic bytes of the hash kernel."""
    names = ["const", "public", "alu", "poseidon2", "recompose"]
    aux = lookup_aux_widths(packing.alu_lanes, packing.horner_packed_steps)
    B = 1 << FRI["log_blowup"]
    perms = heights[3]  # K3: one per Poseidon2-table row
    hash_cells = 0      # k_mmcs_hash_rows only (the committed LDEs); FRI leaves use the strided variant
    hash_rows = 0
    hash_perms = 0
    hash_launches = 0

    def commit(mats):
        nonlocal perms, hash_cells, hash_rows, hash_perms, hash_launches
        by_h = {}
        for h, w in mats:
            by_h[h] = by_h.get(h, 0) + w
        hmax = max(by_h)
        for h, w in by_h.items():
            perms += h * ((w + 7) // 8)
            hash_perms += h * ((w + 7) // 8)
            hash_cells += h * w
            hash_rows += h
            if h != hmax:
                perms += h
        hash_launches += 1  # one job-list launch per commit covers every height class
        perms += hmax - 1

    commit([(heights[i] * B, widths[i]) for i in range(5)])                       # main
    commit([(heights[i] * B, aux[n][0] * 4) for i, n in enumerate(names)])        # permutation
    commit([(heights[i] * B, 4) for i, n in enumerate(names) for _ in range(aux[n][1])])  # quotient chunks
    # FRI commit phase: arity-4 folds from the tallest LDE down to 2^(log_final_poly_len+log_blowup)
    h = max(heights) * B
    final = 1 << (FRI["log_final_poly_len"] + FRI["log_blowup"])
    while h > final:
        la = min(FRI["max_log_arity"], (h // final).bit_length() - 1)
        rows = h >> la
        perms += rows * (((4 << la) + 7) // 8) + rows - 1
        h = rows
    return perms, hash_perms, 4 * hash_cells + 32 * hash_rows, hash_launches


def hbm_families(heights, widths, packing, kernel_ms):
    """Achieved HBM rate of the streaming kernel families: algorithmic bytes / measured family time.
    NTT (K5): 16 B per cell for the inverse transform, 4 + 4B + 8B = 52 B per cell for the four-coset
    forward one (B = 4). Reduced openings (K10): every committed LDE cell once (4 B) plus, per LDE row,
    the 16-B inverse vector of each opening point and the 16-B accumulator. Openings (K9): every trace
    cell once plus a 16-B weight per row, point and group of 8 columns."""
    names = ["const", "public", "alu", "poseidon2", "recompose"]
    aux = lookup_aux_widths(packing.alu_lanes, packing.horner_packed_steps)
    B = 1 << FRI["log_blowup"]
    prep_w = [2, 2 * packing.public_lanes, 13 * packing.alu_lanes + 7 * (packing.horner_packed_steps - 1), 24,
              2 * packing.recompose_lanes]
    ntt = fri = opn = 0
    for i, n in enumerate(names):
        h = heights[i]
        if not h:
            continue
        w_main, w_aux, w_q = widths[i], aux[n][0] * 4, aux[n][1] * 4
        ntt += h * (w_main + w_aux + w_q) * (16 + 4 + 4 * B + 8 * B)
        for w, pts in ((w_main, 2 if n in ("alu", "poseidon2") else 1), (prep_w[i], 2), (w_aux, 2), (w_q, 1)):
            fri += h * B * (4 * w + 16 * pts)
            opn += h * (4 * w + 16 * pts * ((w + 7) // 8))
        fri += h * B * 16
    ntt_min = ntt * (4 + 4 * B) // (16 + 4 + 4 * B + 8 * B)   # one read of the trace + one write of the LDE: 20 B per cell
    out = {}
    for fam, nbytes, keys in (("ntt", ntt, ("ntt_inverse_1", "ntt_inverse_2", "ntt_forward_1", "ntt_forward_2")), ("fri_reduced_openings", fri, ("fri_reduce",)),
                              ("openings", opn, ("open_dot",))):
        ms = sum(kernel_ms.get(k, 0.0) for k in keys)
        if ms:
            gbs = nbytes / (ms * 1e-3) / 1e9
            out[fam] = {"algorithmic_bytes": nbytes, "ms": ms, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS}
            if fam == "ntt":   # the four-pass model above (68 B per cell) next to the one-read-one-write minimum
                bf, prod = ntt_arithmetic(heights, widths, packing)
                arith_ms = (bf / NTT_BUTTERFLY_RATE + prod / MONT_PRODUCT_RATE) * 1e3
                out[fam].update({"model_bytes_per_cell": 16 + 4 + 4 * B + 8 * B, "min_bytes_per_cell": 4 + 4 * B,
                                 "min_bytes": ntt_min, "achieved_vs_min": ntt_min / (ms * 1e-3) / 1e9,
                                 "frac_vs_min": ntt_min / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "arithmetic": {"butterflies": bf, "products": prod, "butterfly_rate": NTT_BUTTERFLY_RATE,
                                                "product_rate": MONT_PRODUCT_RATE, "floor_ms": arith_ms, "frac": arith_ms / ms,
                                                "source": "profiles/r04/ntt_bound.txt: the passes are bound by the butterflies "
                                                          "they issue plus their memory phases, not by HBM bytes"}})
    return out


PROFILE_ROUND = "r06"   # profiles/<round>/ holds the rocprofv3 summaries the roofline numbers refer to


def profile_file(name):
    """Path of a committed summary: this round's, else the newest earlier round's (the line names what it read)."""
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(p):
            return p, f"profiles/{rnd}/{name}"
    return None, None
FP64_FMA_SPEC = 39.3e12  # /opt/skills/guides/MI355X_MICROARCH.md: 78.6 TFLOP/s FP64 vector = 39.3 T FMA lane-ops/s
PUBLISHED_CPU_MS = 109.0  # prove_next_layer of a real (~2^15-row) verifier circuit, Apple M4 Pro, 14 cores


HASH_KERNEL_SOURCES = ("poseidon2_f64.hip.h", "kernels.hip.h")   # what k_mmcs_hash_rows is compiled from


def kernel_source_digest(names=HASH_KERNEL_SOURCES):
    """sha256 of the CODE of the named kernel sources: comments and blank space are stripped first, so that editing a comment
    does not invalidate an instruction count (and nobody is tempted to re-stamp a measurement file by hand - the count is
    a property of the code the compiler saw)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "plonky3_recursion_amd", "csrc", n), "r") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)      # block comments
        text = re.sub(r"//[^\n]*", "", text)                    # line comments (no string literal of these files holds //)
        text = "\n".join(ln.strip() for ln in text.split("\n") if ln.strip())
        h.update(n.encode() + b"\0" + text.encode() + b"\0")
    return h.hexdigest()


W32_KERNEL_SOURCES = ("poseidon2_f64.hip.h", "poseidon2_w32_f64.hip.h", "kernels_mmcs4.hip.h")   # k_mmcs4_hash_rows


def w32_leaf_roofline(field, heights, widths, packing, hash_rows_ms):
    """The arity-4 MMCS's leaf kernel against the FP64 issue peak: width-32 permutations per proof (rate-24 sponge over
    the concatenated row of every height class of the three commits) x instructions per permutation of the built-in
    diagonal's kernel instance (profiles/<round>/pmc_hash_rows.json::width32, measured on KoalaBear; refused when the
    kernel's sources are not the ones it was measured on) / the measured time of mmcs_hash_rows."""
    names = ["const", "public", "alu", "poseidon2", "recompose"]
    aux = lookup_aux_widths(packing.alu_lanes, packing.horner_packed_steps)
    B = 1 << FRI["log_blowup"]
    perms = 0
    for mats in ([(heights[i] * B, widths[i]) for i in range(5)], [(heights[i] * B, aux[n][0] * 4) for i, n in enumerate(names)],
                 [(heights[i] * B, 4) for i, n in enumerate(names) for _ in range(aux[n][1])]):
        by_h = {}
        for h, w in mats:
            by_h[h] = by_h.get(h, 0) + w
        perms += sum(h * ((w + 23) // 24) for h, w in by_h.items())
    out = {"kernel": "k_mmcs4_hash_rows<PP, BUILTIN = true>", "bound": "valu-issue", "width32_perms_per_step_in_kernel": perms,
           "peak": VALU_ISSUE_SPEC / 1e12, "unit": "T FP64 lane-ops/s", "valu_insts_per_perm": None, "achieved": None, "frac": None}
    try:
        path, src = profile_file("pmc_hash_rows.json")
        with open(path) as fh:
            rec = json.load(fh)
        have = kernel_source_digest(W32_KERNEL_SOURCES)
        if rec.get("width32_sources_sha256") != have:
            out["frac_null_reason"] = f"{src}::width32 was measured on other kernel sources: re-run tools/profile_round.sh"
        elif field != "koala-bear":
            out["frac_null_reason"] = "the width-32 instruction count is measured on KoalaBear only"
        else:
            insts = float(rec["width32"]["builtin"]["valu_insts_per_perm"])
            out["valu_insts_per_perm"] = insts
            out["source"] = f"{src}::width32.builtin (general instance: {rec['width32']['general']['valu_insts_per_perm']:.0f})"
            if hash_rows_ms:
                out["achieved"] = perms * insts / (hash_rows_ms * 1e-3) / 1e12
                out["frac"] = out["achieved"] / out["peak"]
    except Exception as e:
        out["frac_null_reason"] = f"no committed width-32 instruction count ({e!r})"
    return out


def measured_leaf_roofline(field, prof, steps=1):
    """Leaf hashing of ANY configuration against the FP64 issue peak, from what the launches actually absorbed: the library
    counts the permutations of its k_mmcs_hash_rows launches while profiling is on (csrc/profile.h::prof_count, entry
    `stage:count:hash_rows_perms`), so no table-mix model is needed - used for the ZK leg (twice the rows, codeword columns,
    a random round, eight masked chunks).  Instructions per permutation: the committed count of the same kernel
    (committed_valu_model).  prof: Context.profile_read() of `steps` proofs."""
    perms = prof.get("stage:count:hash_rows_perms", (None,))[0]
    ms = prof.get("mmcs_hash_rows", (0.0, 0))[0]
    insts, fma_rate, vsrc = committed_valu_model(field)
    out = {"kernel": "k_mmcs_hash_rows", "bound": "valu-issue", "perms_per_step_in_kernel": (perms / steps) if perms else None,
           "ms_per_step": ms / steps if ms else None, "launches_per_step": prof.get("mmcs_hash_rows", (0.0, 0))[1] / steps,
           "valu_insts_per_perm": insts, "peak": FP64_FMA_SPEC / 1e12, "unit": "T FP64 lane-ops/s", "achieved": None, "frac": None,
           "frac_null_reason": vsrc.get("refused")}
    if perms and ms and insts:
        out["perms_per_s"] = perms / (ms * 1e-3)
        out["achieved"] = out["perms_per_s"] * insts / 1e12
        out["frac"] = out["achieved"] / out["peak"]
        out["frac_measured_fma_rate"] = (out["achieved"] * 1e12 / fma_rate) if fma_rate else None
    elif not perms:
        out["frac_null_reason"] = out["frac_null_reason"] or "the library did not report stage:count:hash_rows_perms"
    return out


def committed_valu_model(field):
    """FP64 instructions per Poseidon2 permutation of k_mmcs_hash_rows and the measured v_fma_f64 rate, both read
    from files under profiles/<round>/ so that every number of `valu_roofline` can be recomputed:
      pmc_hash_rows.json      rocprofv3 --pmc SQ_INSTS_VALU over tools/pmc_hash_rows.py (N commits of one matrix of
                              known shape): valu_insts_per_perm = SQ_INSTS_VALU x 64 lanes / permutations
      microbench_int_rates.txt  the `v_fma_f64` line of tools/microbench/int_rates on the same box."""
    insts = rate = None
    src = {}
    try:
        p, src["valu_insts_per_perm"] = profile_file("pmc_hash_rows.json")
        with open(p) as fh:
            rec = json.load(fh)
        # the count is a property of the kernel's code: a file measured on other sources is refused, not quoted
        want, have = rec.get("kernel_sources_sha256"), kernel_source_digest()
        if want != have:
            src["refused"] = (f"{src['valu_insts_per_perm']} was measured on other kernel sources (sha256 of "
                              f"{' + '.join(HASH_KERNEL_SOURCES)}: file {str(want)[:12]}, built {have[:12]}): re-run tools/profile_round.sh")
        else:
            insts = float(rec["fields"][field]["valu_insts_per_perm"])
    except Exception:
        pass
    try:
        p, src["peak_measured_lane_ops_per_s"] = profile_file("microbench_int_rates.txt")
        with open(p) as fh:
            for ln in fh:
                if ln.startswith("v_fma_f64"):
                    rate = float(ln.split()[1]) * 1e12
    except Exception:
        pass
    return insts, rate, src


# Arithmetic of the LDE passes (profiles/r04/ntt_bound.txt, tools/microbench/ntt_valu on one MI355X): the radix-2
# butterfly of csrc/kernels_ntt2.hip.h (ten integer instructions, two of them v_mad_u64_u32) issues at 3.9 T/s with no
# memory in the loop, a Montgomery product at 7.7 T/s (profiles/r03/microbench_int_rates.txt).
NTT_BUTTERFLY_RATE = 3.9e12
MONT_PRODUCT_RATE = 7.68e12
MONT_PRODUCT_RATE_PLAIN = 5.82e12   # "Montgomery product" line of microbench_int_rates.txt (mul_lo / mul_hi / borrow form)
VALU_ISSUE_SPEC = 39.3e12           # full-rate VALU lane-instructions/s: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (the guide's FP64 FMA figure)

# kernels (names as rocprofv3 reports them, tools/collect_profiles.py::kname) behind the families of kernel_ms_per_step
VALU_FAMILY_KERNELS = {
    "ntt": (("ntt_inverse_1", "ntt_inverse_2", "ntt_forward_1", "ntt_forward_2"),
            ("k_ntt_col", "k_ntt_col_mixed", "k_ntt_fwd_line", "k_ntt_fwd_line_mixed", "k_ntt_tile")),
    "quotient": (("quotient",), ("k_quotient",)),
    "mmcs_compress": (("mmcs_compress",), ("k_mmcs_compress", "k_mmcs_compress_f64", "k_mmcs_subtree", "k_mmcs_top")),
    "run_levels": (("run_levels",), ("k_run_level", "k_run_levels_narrow", "k_run_chains_block", "k_run_chains_wave")),
    "logup_aux": (("logup_aux",), ("k_logup_aux", "k_ef_scan")),
    "fri_reduce": (("fri_reduce",), ("k_fri_reduce_pre", "k_fri_vsum")),
    "open_dot": (("open_dot",), ("k_open_dot", "k_open_reduce")),
}


FP64_FAMILIES = ("mmcs_compress",)   # (and the dominant kernel, priced in `roofline`): FP64 arithmetic, FP64 issue rate


def valu_families(kernel_ms):
    """What binds the families that are not HBM-bound: VALU lane-instructions per proof (SQ_INSTS_VALU x 64 from the
    committed PMC pass, profiles/<round>/pmc_sq.json, divided by the proofs of that run) against an issue floor that
    follows the family's INSTRUCTION MIX: floor_ms = N x sum_opcode share / rate, shares from the disassembly of the
    launched kernel instances and measured per-opcode rates (tools/valu_mix.py -> profiles/<round>/valu_mix.json: a gfx950
    SIMD retires 32-bit adds faster than the FP64 FMA rate and v_mad_u64_u32 / v_mul_hi_u32 at a half / a third of it, so
    one divisor for all families - the 39.3 T/s of round 5 - was wrong for the integer ones in both directions).
    frac = floor_ms / ms.  The FP64 families keep the FP64 issue rate.  `frac_fp64_rate` is the old reading, for
    comparison across rounds."""
    p, name = profile_file("pmc_sq.json")
    if not p:
        return None
    try:
        with open(p) as fh:
            rec = json.load(fh)
        ks = rec["kernels"]
        proofs = rec.get("proofs_in_run") or ks.get("k_quotient", {}).get("launches")   # one quotient launch per proof
        mix, mix_name = {}, None
        mp, mix_name = profile_file("valu_mix.json")
        if mp:
            with open(mp) as fh:
                mix = json.load(fh).get("families", {})
        out = {"source": f"{name}: SQ_INSTS_VALU x 64 lanes / {proofs} proofs of that run; times: this run's kernel_ms_per_step; "
                         f"mix and rates: {mix_name}",
               "fp64_lane_insts_per_s": VALU_ISSUE_SPEC}
        for fam, (time_keys, kernels) in VALU_FAMILY_KERNELS.items():
            insts = sum(ks[k]["SQ_INSTS_VALU"] for k in kernels if k in ks and "SQ_INSTS_VALU" in ks[k]) * 64.0 / proofs
            ms = sum(kernel_ms.get(k, 0.0) for k in time_keys)
            if insts and ms:
                rate = insts / (ms * 1e-3)
                eff = VALU_ISSUE_SPEC if fam in FP64_FAMILIES or fam not in mix else mix[fam]["effective_rate_T_per_s"] * 1e12
                floor_ms = insts / eff * 1e3
                out[fam] = {"valu_lane_insts_per_step": insts, "ms": ms, "achieved_lane_insts_per_s": rate,
                            "mix_rate_lane_insts_per_s": eff, "floor_ms": floor_ms, "frac": floor_ms / ms,
                            "frac_fp64_rate": rate / VALU_ISSUE_SPEC,
                            "mix": None if fam in FP64_FAMILIES or fam not in mix else mix[fam]["share_by_priced_opcode"],
                            "bound": "valu-issue" if floor_ms / ms >= 0.6 else "latency / memory phases",
                            "kernels": [k for k in kernels if k in ks]}
        return out
    except Exception as e:
        return {"error": repr(e)}


def ntt_arithmetic(heights, widths, packing):
    """Butterflies and twiddle / shift products of a proof's coset LDEs: per input cell of a 2^n-row column,
    n/2 butterflies for the inverse transform and 4 x n/2 for the four cosets; 2 products for the four-step twiddle of
    the inverse transform and 3 per coset (coset shift power, four-step twiddle chain) for the forward one."""
    names = ["const", "public", "alu", "poseidon2", "recompose"]
    aux = lookup_aux_widths(packing.alu_lanes, packing.horner_packed_steps)
    B = 1 << FRI["log_blowup"]
    bf = prod = 0
    for i, n in enumerate(names):
        h = heights[i]
        if not h:
            continue
        cells = h * (widths[i] + aux[n][0] * 4 + aux[n][1] * 4)
        bf += cells * (h.bit_length() - 1) / 2 * (1 + B)
        prod += cells * (2 + 3 * B)
    return bf, prod


def proof_roofline(field, heights, widths, packing, perms, insts_per_perm, hash_bytes, ms_per_step, fams):
    """The whole proof against its two roofs (reported separately and summed): every Poseidon2
    permutation of the proof (workload_model: trace fill, leaf hashing, compressions, FRI leaves) at the dominant
    kernel's instructions per permutation against the FP64 vector peak; every streaming kernel at its MINIMUM
    algorithmic bytes against the HBM peak - the LDE as one read of the trace and one write of the four cosets
    (20 B per cell, not the 68 B the four passes move), the hash kernel's read of the committed LDEs, the quotient's
    read of main / preprocessed / aux LDEs on the quotient domain (local + next row), openings and reduced openings.
    frac = floor_ms / ms_per_step."""
    names = ["const", "public", "alu", "poseidon2", "recompose"]
    aux = lookup_aux_widths(packing.alu_lanes, packing.horner_packed_steps)
    prep_w = [2, 2 * packing.public_lanes, 13 * packing.alu_lanes + 7 * (packing.horner_packed_steps - 1), 24,
              2 * packing.recompose_lanes]
    quot = 0
    for i, n in enumerate(names):
        if heights[i]:   # quotient domain = chunks x h rows; local + next of every column; 16 B per chunk row written
            c = aux[n][1]
            quot += heights[i] * c * (2 * 4 * (widths[i] + prep_w[i] + 4 * aux[n][0]) + 16)
    hbm = {"lde_min": fams.get("ntt", {}).get("min_bytes", 0), "leaf_hash_read": hash_bytes, "quotient": quot,
           "reduced_openings": fams.get("fri_reduced_openings", {}).get("algorithmic_bytes", 0),
           "openings": fams.get("openings", {}).get("algorithmic_bytes", 0)}
    hbm_bytes = sum(hbm.values())
    valu_ms = perms * insts_per_perm / FP64_FMA_SPEC * 1e3 if insts_per_perm else None
    hbm_ms = hbm_bytes / (HBM_PEAK_GBS * 1e9) * 1e3
    floor = (valu_ms or 0.0) + hbm_ms
    ntt_arith = fams.get("ntt", {}).get("arithmetic", {}).get("floor_ms")
    out = {"valu_floor_ms": valu_ms, "hbm_floor_ms": hbm_ms, "floor_ms": floor, "frac": floor / ms_per_step,
           "poseidon2_perms": perms, "valu_insts_per_perm": insts_per_perm, "valu_peak_lane_ops_per_s": FP64_FMA_SPEC,
           "hbm_min_bytes": hbm_bytes, "hbm_min_bytes_by_family": hbm, "hbm_peak_GBps": HBM_PEAK_GBS}
    if ntt_arith and valu_ms:
        # the LDE passes priced by their butterflies (what binds them, profiles/r04/ntt_bound.txt) instead of by 20 B per cell
        f2 = valu_ms + ntt_arith + (hbm_bytes - hbm["lde_min"]) / (HBM_PEAK_GBS * 1e9) * 1e3
        out["with_lde_arithmetic"] = {"lde_arithmetic_floor_ms": ntt_arith, "floor_ms": f2, "frac": f2 / ms_per_step}
    return out


def pmc_traffic_bytes(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (collected with rocprofv3 in
    separate --pmc passes by tools/profile_round.sh; it cannot be collected from inside this process).
    The kernel's HBM traffic is its input cells and output digests whatever arithmetic hashes them, so
    the figure only goes stale when the table mix changes - `traffic_source` names the file."""
    try:
        with open(profile_file("pmc_traffic.json")[0]) as fh:
            k = json.load(fh)["kernels"][kernel]
        # gfx950: FETCH_SIZE tallies the 128-B requests of a coalesced streaming read at 64 B
        # (/opt/skills/guides/MI355X_MICROARCH.md, HBM section) - doubled before comparing with bytes.
        # Calibration on this access pattern (4 B per lane, every LDE cell read exactly once):
        # reported 161 MB against 332 MB that must be read per launch = 0.49.
        return (2.0 * k["fetch_kb_per_launch"] + k["write_kb_per_launch"]) * 1024.0
    except Exception:
        return None


def cpu_baseline(field, log_h):
    """The CPU oracle (kind 'port', OpenMP over the host cores: rows of a commit, columns of an LDE,
    rows of the LogUp / quotient / reduced-opening passes): run the circuit, prove its tables, same
    table mix at a bounded size.  Returns (seconds, circuit-run seconds, threads)."""
    import circuit_lib
    import harness_lib
    import layer_lib
    import oracle_lib
    orc = oracle_lib.Oracle()
    arrs = harness_lib.generate(field, log_h, seed=1, **GEN_KNOBS)
    prm = layer_lib.params(**FRI)
    oc = circuit_lib.OracleCircuit(orc, circuit_lib.Circuit.from_arrays(arrs)).preprocess(oracle_lib.MODULUS[field])
    inputs = circuit_lib.Inputs.from_arrays(arrs)
    t0 = time.perf_counter()
    oc.run(field, inputs)
    run_s = time.perf_counter() - t0
    L = layer_lib.OracleLayer(orc, field, oc.workload_arrays(), prm)
    L.prep_commit()  # preprocessed commitment is cached in the reference too (NextLayerPrepCache)
    t0 = time.perf_counter()
    L.prove()
    return run_s + (time.perf_counter() - t0), run_s, int(orc.lib.orc_num_threads())


def small_layers(ctx, p3r, wl, packing, field, sizes=(14, 15, 16), steps=20):
    """The reference's real verifier circuits have 2^14..2^16 rows:
    the same step at those sizes, each proof verified natively."""
    import harness_lib
    out = {}
    for lh in sizes:
        arrs = harness_lib.generate(field, lh, seed=0x5EED0000, **GEN_KNOBS)
        circ, hin = wl.circuit_from_arrays(arrs), wl.circuit_inputs_from_arrays(arrs)
        params = p3r.ProveNextLayerParams(table_packing=packing)
        warm = p3r.build_next_layer_prep(ctx, circ, p3r.FriRecursionBackend(), params)   # pools, job tables
        warm.prepared_circuit.prove(hin)
        warm.prepared_circuit.free()
        ctx.sync()
        # prove_next_layer with prep = None (recursion.rs:452-501): preparation + proof from host inputs
        m0 = time.perf_counter()
        cache = p3r.build_next_layer_prep(ctx, circ, p3r.FriRecursionBackend(), params)
        pc = cache.prepared_circuit
        miss_proof = pc.prove(hin)
        ctx.sync()
        miss_ms = (time.perf_counter() - m0) * 1e3
        res = pc.upload_inputs(hin)
        proof = pc.prove(res)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            proof = pc.prove(res)
        ctx.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        try:
            cache.prover.verify_all_tables(cache.prover.wrap_proof(proof, pc.circuit_prover_data))
            ok = True
        except Exception as e:
            print(f"bench: small layer 2^{lh}: proof rejected: {e}", file=sys.stderr)
            ok = False
        out[str(lh)] = {"ms_per_step": ms, "steps": steps, "proof_bytes": len(proof), "proof_verified": ok and miss_proof == proof,
                        "prep_miss_ms": miss_ms, "table_heights": pc.circuit_prover_data.table_heights}
        res.free()
        pc.free()
    return out


def small_layer_throughput(p3r, wl, packing, field, log_h=15, provers=8, reps=40):
    """Independent proofs of production-size layers on ONE GPU: `provers` host threads, each with its own
    p3r context (HIP stream + memory pool) - what a recursion service runs, since one 2^15-row proof does
    not fill the chip (the Merkle tops and the circuit-run chains are latency-bound).  Every proof must be
    the same bytes as the single-prover one.  Eight provers saturate the GPU at this size (tools/concurrent_small.py:
    4 -> 728, 8 -> 768, 16 -> 783 proofs/s; three processes of four provers together give the same 805)."""
    import threading
    import harness_lib
    arrs = harness_lib.generate(field, log_h, seed=0x5EED0000, **GEN_KNOBS)
    workers = []
    for _ in range(provers):
        c = p3r.Context(field=field, **FRI)
        pc = p3r.PreparedCircuit(c, wl.circuit_from_arrays(arrs), packing)
        res = pc.upload_inputs(wl.circuit_inputs_from_arrays(arrs))
        workers.append((c, pc, res, pc.prove(res)))
    ok = all(w[3] == workers[0][3] for w in workers)
    bad = []
    for w in workers:          # a second proof per prover: pools and job tables are warm when the clock starts
        w[1].prove(w[2])

    def run(w):
        _, pc, res, ref = w
        for _ in range(reps):
            if pc.prove(res) != ref:
                bad.append(1)

    ts = [threading.Thread(target=run, args=(w,)) for w in workers]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    for c, pc, res, _ in workers:
        res.free()
        pc.free()
        c.close()
    return {"log_height": log_h, "provers": provers, "proofs": provers * reps, "proofs_per_s": provers * reps / dt,
            "ms_per_proof_amortised": dt / (provers * reps) * 1e3, "all_proofs_identical": ok and not bad}


def run_tree(args, torch, dist, rank, world, local_rank, coll_device, backend_name):
    """BASELINE config 4: 2-to-1 aggregation trees of `--tree-leaves` leaf proofs -> 1 root over the ranks
    (one GPU each).  Leaves are prove_next_layer over the synthetic layer at 2^leaf_log_height rows, every
    aggregation node is prove_aggregation_layer (recursion.rs:656-762) over the synthetic layer at TWICE
    the Poseidon2 / ALU counts, with one AggregationPrepCache per prover.  The
    scheduler is dependency-driven (aggregation.run_aggregation_forest): a parent starts when ITS two
    children are on its rank (it sits where its left child was; the right child's bytes move by send/recv),
    parses them natively (p3r_batch_stark_proof_parse: framing + metadata rules, `child_parse_ms`) and -
    with --tree-verify-children - verifies them first.  --trees K keeps K independent trees in flight,
    tree t placed with rank offset t: every rank proves the same number of nodes, which is the form of the
    workload that can scale linearly (one tree has a critical path of log2(leaves) + 1 dependent proofs).
    The synthetic node circuit's VALUES do not depend on the children (building verifier circuits is the
    reference's CPU front end, out of scope); its SCHEDULE does.  --tree-level-barriers selects the
    level-synchronous scheduler (per-level wall times)."""
    import harness_lib
    import harness_adapters as wl
    import plonky3_recursion_amd as p3r
    from plonky3_recursion_amd.aggregation import TreePlan, predict_forest_wall_ms, run_aggregation_forest, run_aggregation_tree
    field, lh = args.field, args.leaf_log_height
    import queue
    packing = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    params = p3r.ProveNextLayerParams(table_packing=packing)
    backend = p3r.FriRecursionBackend()
    n_trees = args.trees if args.trees > 0 else world
    if args.tree_level_barriers and n_trees != 1:
        print("bench: --tree-level-barriers measures one tree", file=sys.stderr)
        sys.exit(2)
    plans = [TreePlan(args.tree_leaves, world, offset=t) for t in range(n_trees)]
    la = harness_lib.generate(field, lh, seed=0x5EED0000, **GEN_KNOBS)
    leaf_circuit, leaf_host_inputs = wl.circuit_from_arrays(la), wl.circuit_inputs_from_arrays(la)
    del la
    na = harness_lib.generate(field, lh + 1, seed=0x5EED0001, **GEN_KNOBS)   # aggregation node: twice the counts
    node_circuit = wl.circuit_from_arrays(na)
    n_left, n_right, left_ops = wl.split_aggregation_inputs(wl.circuit_inputs_from_arrays(na))
    del na
    # one worker = one p3r_ctx (HIP stream + memory pool) with its own NextLayerPrepCache /
    # AggregationPrepCache; --tree-workers > 1 proves the nodes a rank holds concurrently
    workers = queue.Queue()
    all_workers = []
    for _ in range(max(1, args.tree_workers)):
        # --zk: `config_with_fri_params_zk(.., rng_seed)` per prover (recursive_aggregation.rs:711-721); distinct seeds, as
        # distinct PCS objects have distinct RNG states
        zk_kw = dict(zk=1, num_random_codewords=2) if args.zk else {}   # (the library keys every context from the operating system)
        wctx = p3r.Context(field=field, device=local_rank, **FRI, **zk_kw)
        lc = p3r.build_next_layer_prep(wctx, leaf_circuit, backend, params)
        wk = dict(ctx=wctx, leaf_cache=lc, leaf_inputs=lc.prepared_circuit.upload_inputs(leaf_host_inputs), agg_cache=[None])
        workers.put(wk)
        all_workers.append(wk)
    ctx = all_workers[0]["ctx"]
    import threading
    stats_lock = threading.Lock()
    stats = {"leaf_ms": [], "node_ms": [], "child_parse_ms": [], "child_parse_native_ms": [], "child_verify_ms": []}

    def note(key, ms):
        with stats_lock:
            stats[key].append(ms)

    def prove_leaf(_tree, i):
        wk = workers.get()
        try:
            t0 = time.perf_counter()
            out = p3r.prove_next_layer(p3r.RecursionInput(circuit_inputs=wk["leaf_inputs"]), wk["ctx"], backend, params,
                                       prep=wk["leaf_cache"])
            note("leaf_ms", (time.perf_counter() - t0) * 1e3)
            return out.proof
        finally:
            workers.put(wk)

    def decode(data):
        """A child proof off the wire (or, in the level-synchronous mode, any child): native parse + metadata rules."""
        t0 = time.perf_counter()
        proof = p3r.BatchStarkProof.from_postcard(data, field, zk=args.zk)
        note("child_parse_ms", (time.perf_counter() - t0) * 1e3)
        note("child_parse_native_ms", proof.parse_ns * 1e-6)
        return proof

    def prove_parent(_tree, level, node, left, right):
        wk = workers.get()
        try:
            # children proved on this rank arrive as the BatchStarkProof their prover returned; only proofs that changed
            # rank were serialised, and those were parsed on receipt by the communication thread (decode)
            children = [c if isinstance(c, p3r.BatchStarkProof) else decode(c) for c in (left, right)]
            if args.tree_verify_children:
                t1 = time.perf_counter()
                for c in children:
                    p3r.verify_all_tables(wk["ctx"].cfg, c)
                note("child_verify_ms", (time.perf_counter() - t1) * 1e3)
            t1 = time.perf_counter()
            out = p3r.prove_aggregation_layer(
                p3r.RecursionInput(prev_proof=children[0], circuit_inputs=n_left),
                p3r.RecursionInput(prev_proof=children[1], circuit_inputs=n_right),
                node_circuit, wk["ctx"], backend, params, prep_cache=wk["agg_cache"], left_non_primitive_ops=left_ops)
            note("node_ms", (time.perf_counter() - t1) * 1e3)
            return out.proof
        finally:
            workers.put(wk)

    def barrier():
        for wk in all_workers:
            wk["ctx"].sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    level_ms = []
    node_done = []

    def on_level(level, seconds):
        level_ms.append(seconds * 1e3)

    # warm-up: one leaf and one node per worker (fills the AggregationPrepCache, the pools, the job tables)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(all_workers)) as ex:
        warm = list(ex.map(lambda i: prove_leaf(0, i), range(len(all_workers))))
        list(ex.map(lambda w: prove_parent(0, 1, 0, w, w), warm))
        decode(warm[0].to_postcard())
    # ---- the inputs of the wall-time prediction (aggregation.predict_forest_wall_ms), measured on an idle GPU before the
    # timed region: SOLO latency of a leaf and of a node (one prover, nothing else in flight), the GPU's capacity for
    # concurrent proofs (this rank's provers all proving leaves at once), and what a child proof costs when it changes
    # rank (serialise + native parse here; + one send/recv of that size when there is a peer)
    def solo(fn, reps=3):
        best = None
        for _ in range(reps):
            barrier_local()
            t1 = time.perf_counter()
            fn()
            barrier_local()
            best = min(best or 1e9, (time.perf_counter() - t1) * 1e3)
        return best

    def barrier_local():
        for wk in all_workers:
            wk["ctx"].sync()

    # ranks that share a GPU (the gloo test runs) take their SOLO latencies one after the other ...
    local_devices = [local_rank] * world
    if dist is not None and world > 1:
        t = torch.tensor([local_rank], dtype=torch.int64, device=coll_device)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        local_devices = [int(g.item()) for g in got]
    solo_leaf_ms = solo_node_ms = None
    for r in range(world):
        if r == rank:
            solo_leaf_ms = solo(lambda: prove_leaf(0, 0))
            solo_node_ms = solo(lambda: prove_parent(0, 1, 0, warm[0], warm[0]))
        if dist is not None:
            dist.barrier()
    # ... and measure the GPU's capacity TOGETHER: every prover of every rank proves twelve leaves (then twelve nodes) at
    # once; capacity = solo-milliseconds of work done on this rank's GPU per millisecond of wall time.  (Provers of two
    # processes share a GPU less well than provers of one: the measurement has to be made in the run's own form.)
    sharers = max(1, sum(1 for d in local_devices if d == local_devices[rank])) if backend_name != "nccl" else 1
    gpu_capacity = [1.0, 1.0]
    if len(all_workers) * sharers > 1:
        with ThreadPoolExecutor(max_workers=len(all_workers)) as ex:
            reps = 12   # proofs per prover and sample; the better of two samples (a 30 ms sample of four proofs each
            #             varied between 1.5 and 2.0 from run to run and took the four-tree prediction with it)
            for which, fn, alone in ((0, lambda i: [prove_leaf(0, i) for _ in range(reps)], solo_leaf_ms),
                                     (1, lambda i: [prove_parent(0, 1, 0, warm[0], warm[0]) for _ in range(reps)], solo_node_ms)):
                best = 0.0
                for _ in range(2):
                    barrier()
                    t1 = time.perf_counter()
                    list(ex.map(fn, range(len(all_workers))))
                    barrier()
                    best = max(best, reps * len(all_workers) * sharers * alone / ((time.perf_counter() - t1) * 1e3))
                gpu_capacity[which] = min(float(len(all_workers) * sharers), best)
    if dist is not None and world > 1:   # rank 0's measurements are the ones the prediction uses
        t = torch.tensor([solo_leaf_ms, solo_node_ms] + gpu_capacity, dtype=torch.float64, device=coll_device)
        dist.broadcast(t, 0)
        solo_leaf_ms, solo_node_ms, gpu_capacity = float(t[0]), float(t[1]), [float(t[2]), float(t[3])]
    wire = warm[0].to_postcard()
    t1 = time.perf_counter()
    for _ in range(5):
        decode(warm[0].to_postcard())
    codec_ms = (time.perf_counter() - t1) / 5 * 1e3
    link_ms = 0.0
    if dist is not None and world > 1:   # one proof-sized message rank 0 -> rank 1 -> rank 0 (the collective layer's own latency)
        from plonky3_recursion_amd.aggregation import _recv_bytes, _send_bytes
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            if rank == 0:
                _send_bytes(dist, wire, 1, coll_device)
                _recv_bytes(dist, 1, coll_device)
            elif rank == 1:
                _send_bytes(dist, _recv_bytes(dist, 0, coll_device), 0, coll_device)
        link_ms = (time.perf_counter() - t1) / 6 * 1e3
        t = torch.tensor([link_ms], dtype=torch.float64, device=coll_device)
        dist.broadcast(t, 0)
        link_ms = float(t.item())
    for v in stats.values():
        v.clear()
    times = []
    roots = None
    for _ in range(args.warmup + args.steps):
        barrier()
        level_ms.clear()
        node_done.clear()
        t0 = time.perf_counter()
        if args.tree_level_barriers:
            r = run_aggregation_tree(plans[0], rank, lambda i: prove_leaf(0, i).to_postcard(),
                                     lambda lv, nd, lb, rb: prove_parent(0, lv, nd, lb, rb).to_postcard(),
                                     dist=dist, device=coll_device, on_level=on_level, level_barrier=barrier,
                                     workers=len(all_workers))
            roots = [r] if rank == 0 else None
        else:
            roots = run_aggregation_forest(plans, rank, prove_leaf, prove_parent, dist=dist, device=coll_device,
                                           workers=len(all_workers), encode=lambda pr: pr.to_postcard(), decode=decode,
                                           on_node=lambda t, lv, nd, sec: node_done.append((lv, sec * 1e3)))
        barrier()
        times.append(time.perf_counter() - t0)
    if not args.tree_level_barriers:   # rank 0: when the last node of each level (of any tree) was proved here
        level_ms = [max((ms for lv, ms in node_done if lv == l), default=None) for l in range(plans[0].levels)]
    times = times[args.warmup:]
    dt = sum(times)
    own_dt = dt
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ranks = rank_report(torch, dist, world, backend_name, local_rank, coll_device, rank_ms=own_dt / max(len(times), 1) * 1e3)
    ok = True
    if rank == 0:
        import hashlib
        roots = [r.to_postcard() if isinstance(r, p3r.BatchStarkProof) else r for r in roots]
        for root in roots:
            rp = p3r.BatchStarkProof.from_postcard(root, field, zk=args.zk)    # the root is checked from its wire form
            try:
                p3r.verify_all_tables(ctx.cfg, rp)
            except Exception as e:
                print(f"bench: a ROOT proof was rejected: {e}", file=sys.stderr)
                ok = False
        root = roots[0]
        n_nodes = (2 * args.tree_leaves - 1) * n_trees
        ms_tree = dt / args.steps * 1e3
        mean = lambda v: (sum(v) / len(v)) if v else None
        # prediction: the scheduler's own event model on the solo measurements above - for THIS run (ranks that share a GPU
        # share its capacity) and for one GPU per rank at 1 / 2 / 4 / 8 ranks, which is what a multi-GPU run is compared with
        message_ms = codec_ms + link_ms
        gpu_ids = {p["rank"]: (p["device"], p["pci"]) for p in ranks["per_rank"]}
        here = predict_forest_wall_ms(plans, solo_leaf_ms, solo_node_ms, message_ms, workers=len(all_workers), gpu_of_rank=gpu_ids,
                                      gpu_capacity=gpu_capacity)
        worlds = {}
        for w in (1, 2, 4, 8):
            pl = [TreePlan(args.tree_leaves, w, offset=t) for t in range(args.trees if args.trees > 0 else w)]
            pr = predict_forest_wall_ms(pl, solo_leaf_ms, solo_node_ms, message_ms, workers=len(all_workers), gpu_capacity=gpu_capacity)
            worlds[str(w)] = {"predicted_wall_ms": pr["wall_ms"], "critical_path_ms": pr["critical_path_ms"], "trees": len(pl),
                              "proofs_per_s": pr["nodes"] / (pr["wall_ms"] * 1e-3)}
        prediction = {
            "model": "aggregation.predict_forest_wall_ms: dependency-driven schedule replayed on solo times; proofs in flight on one "
                     "GPU share it above `gpu_capacity`; one communication thread per rank",
            "inputs": {"solo_leaf_ms": solo_leaf_ms, "solo_node_ms": solo_node_ms,
                       "gpu_capacity_concurrent_proofs": {"leaves": gpu_capacity[0], "nodes": gpu_capacity[1]},
                       "message_ms": message_ms, "serialise_plus_parse_ms": codec_ms, "link_ms": link_ms, "proof_bytes": len(wire),
                       "workers_per_rank": len(all_workers)},
            "this_run": {"predicted_wall_ms": here["wall_ms"], "measured_wall_ms": ms_tree, "measured_over_predicted": ms_tree / here["wall_ms"],
                         "critical_path_ms": here["critical_path_ms"], "busy_ms_per_rank": here["busy_ms_per_rank"]},
            "one_gpu_per_rank": worlds,
            "note": "a tree is log2(leaves) + 1 dependent proofs deep: beyond `leaves` / 2 GPUs its wall time is the critical path "
                    "(solo latencies + one message per level), whatever the GPU count; --trees 0 is the form that scales",
        }
        what = f"{n_trees} independent 2-to-1 aggregation trees in flight" if n_trees > 1 else "2-to-1 aggregation tree"
        emit({
            "metric": f"aggregation tree wall ms ({args.tree_leaves} leaf proofs -> 1 root, prove_aggregation_layer per node), KoalaBear",
            "value": ms_tree, "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_tree, "higher_is_better": False,
            "scaling": "weak" if args.trees == 0 else "strong", "vs_baseline": None,
            "dtype": "u32 Montgomery (31-bit prime field, degree-4 extension); Poseidon2 hashing as exact integers in f64", "data": "synthetic",
            "config": {"workload": f"{what}, {args.tree_leaves} leaves (prove_next_layer, synthetic {field} "
                                   f"2^{lh}-row layer) -> {args.tree_leaves - 1} nodes (prove_aggregation_layer, 2^{lh + 1}-row "
                                   f"layer = twice the Poseidon2 / ALU counts), one rank per GPU, parent on its left child's rank, "
                                   f"tree t placed with rank offset t",
                       "zk": bool(args.zk),
                       "field": field, "leaf_log_height": lh, "node_log_height": lh + 1, "leaves": args.tree_leaves,
                       "trees": n_trees, "nodes": n_nodes, "fri": FRI, "workers_per_rank": len(all_workers),
                       "scheduler": "level-synchronous" if args.tree_level_barriers else "dependency-driven",
                       "parallelism": f"tree nodes over {world} ranks ({len(all_workers)} concurrent provers = HIP streams per rank), "
                                      f"send/recv of child proofs only"},
            "ranks": ranks,
            "proofs_per_s": n_nodes / (ms_tree * 1e-3), "trees_per_s": n_trees / (ms_tree * 1e-3),
            "critical_path_ms": here["critical_path_ms"],
            "prediction": prediction,
            "root_verified": ok, "roots_verified": len(roots) if ok else 0,
            "root_sha256": hashlib.sha256(root).hexdigest(), "root_bytes": len(root),
            "rank0": {"leaf_ms": mean(stats["leaf_ms"]), "node_ms": mean(stats["node_ms"]),
                      "child_parse_ms": mean(stats["child_parse_ms"]), "child_parse_native_ms": mean(stats["child_parse_native_ms"]),
                      "child_verify_ms": mean(stats["child_verify_ms"]),
                      "level_wall_ms_last_step": list(level_ms)},
            "proof_verified": ok, "proof_sha256": hashlib.sha256(root).hexdigest(),
        }, args)
    for wk in all_workers:
        wk["leaf_inputs"].free()
        wk["leaf_cache"].prepared_circuit.free()
        if wk["agg_cache"][0] is not None:
            wk["agg_cache"][0].prepared_circuit.free()
        wk["ctx"].close()
    if dist is not None:
        dist.destroy_process_group()
    flush_final_line()
    if not ok:
        sys.exit(3)


def self_launch(n):
    """One rank per GPU over RCCL: python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <same flags>."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    # the ranks set up side by side (workload generation, the host halves of preparation and verification): their
    # OpenMP teams share the host's cores instead of each taking all of them
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def weak_scaling_prediction(solo_ms, sigma=0.015):
    """The plain multi-GPU entry shards nothing: every rank proves its own layer on its own GPU, inputs resident, and the
    only collectives are the barrier and the MAX of the timings outside the timed steps.  So the step of N ranks is the
    slowest rank's solo step: solo x (1 + sigma x E[max of N standard normals]), sigma = the GPU-to-GPU spread seen across
    the boxes of profiles/r03 - r05 (1.5 %); the aggregate rate is N proofs per that step.  What the run cannot show on
    one GPU and a shared node may add: host cores (each rank's launch loop and transcript take one core; the workload
    generator before the timed region is single-threaded)."""
    emax = {1: 0.0, 2: 0.5642, 4: 1.0294, 8: 1.4236}
    return {"model": "independent proofs, one per GPU: step(N) = slowest rank's solo step = solo x (1 + 0.015 x E[max of N normals]); "
                     "no data-path collective", "solo_ms": solo_ms,
            "one_gpu_per_rank": {str(n): {"predicted_ms_per_step": solo_ms * (1 + sigma * e), "predicted_proofs_per_s": n / (solo_ms * (1 + sigma * e)) * 1e3}
                                 for n, e in emax.items()}}


def rank_report(torch, dist, world, backend, local_rank, coll_device, rank_ms=None):
    """What the collective layer saw: world size and backend as it reports them, and per rank the device ordinal, its PCI
    address (two ranks on one GPU show the same one) and the rank's own time for the timed region."""
    def pci(i):
        try:
            p = torch.cuda.get_device_properties(i)
            return [int(getattr(p, "pci_domain_id", 0)), int(getattr(p, "pci_bus_id", -1)), int(getattr(p, "pci_device_id", -1))]
        except Exception:
            return [0, -1, -1]
    mine = [local_rank] + pci(local_rank) + [int(round((rank_ms or 0.0) * 1000.0))]
    if dist is None:
        rows, ws, be = [mine], 1, None
    else:
        t = torch.tensor(mine, dtype=torch.int64, device=coll_device)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        rows, ws, be = [[int(v) for v in g.tolist()] for g in got], dist.get_world_size(), dist.get_backend()
    per_rank = [{"rank": r, "device": row[0], "pci": "%04x:%02x:%02x" % (row[1], row[2] & 0xFF, row[3] & 0xFF) if row[2] >= 0 else None,
                 "ms": row[4] / 1000.0 if rank_ms is not None else None} for r, row in enumerate(rows)]
    return {"world_size": ws, "backend": be, "devices": [row[0] for row in rows], "per_rank": per_rank,
            "distinct_gpus": len({(p["device"], p["pci"]) for p in per_rank}),
            "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "host_cpus": os.cpu_count()}


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "proof_verified", "proof_sha256")
CONTRACT_MAX_CHARS = 6000
SUSTAINED_S = 6.0   # seconds of back-to-back headline proofs after the timed region (default run, one GPU)   # the driver keeps an ~8 KB tail of stdout: the final line must fit whole inside it


def _round_floats(x, digits=6):
    """Floats to `digits` significant figures (the final line is read by people and by a size-limited record)."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _round_floats(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round_floats(v, digits) for v in x]
    return x


def contract_line(full, detail_path=None):
    """The ONE line the driver parses: exactly the contract fields of a full result dict (everything else - kernel
    families, secondary legs, predictions - lives in the detail file).  Pure function of `full`; tests/test_bench_line_contract.py
    holds it to CONTRACT_MAX_CHARS on canned results."""
    cfg = full.get("config") or {}
    out = {k: full.get(k) for k in CONTRACT_KEYS if k not in ("config", "roofline", "cpu_baseline")}
    out["config"] = {k: cfg.get(k) for k in ("workload", "field", "log_height", "table_heights", "table_widths", "fri",
                                            "independent_proofs", "proof_bytes", "parallelism", "tree", "leaves", "trees",
                                            "leaf_log_height", "workers_per_rank", "zk") if k in cfg}
    r = full.get("roofline")
    if r:
        hbm = r.get("hbm") or {}
        out["roofline"] = {
            "kernel": r.get("kernel"), "bound": r.get("bound"), "achieved": r.get("achieved"), "peak": r.get("peak"),
            "unit": "T FP64 lane-ops/s" if r.get("bound") == "valu-issue" else r.get("unit"),
            "frac": r.get("frac"), "traffic": r.get("traffic"), "avg_launch_ms": r.get("avg_launch_ms"),
            "perms_per_s": r.get("perms_per_s"), "valu_insts_per_perm": r.get("valu_insts_per_perm"),
            "frac_measured_fma_rate": r.get("frac_measured"), "vs_survey_integer_roof": r.get("vs_survey_integer_roof"),
            "frac_null_reason": r.get("frac_null_reason"),
            "hbm": {k: hbm.get(k) for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch",
                                            "traffic_vs_algorithmic")},
            "sources": r.get("source_files"),
        }
    else:
        out["roofline"] = None
    c = full.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = {k: c.get(k) for k in ("value", "unit", "cores", "kind", "sample", "circuit_run_ms",
                                                     "gpu_ms_same_sample", "published_reference_ms")}
    else:
        out["cpu_baseline"] = None
    for k in ("poseidon2_perms_per_s", "dominant_kernel_family", "value_incl_h2d_ms", "prep_miss_ms", "proof_verify_ms",
              "root_handoff_ms", "proofs_per_s", "trees_per_s", "critical_path_ms", "roots_verified"):
        if full.get(k) is not None:
            out[k] = full[k]
    pr = full.get("proof_roofline")
    if pr and pr.get("frac") is not None:
        out["proof_roofline_frac"] = pr["frac"]
    if full.get("ranks"):
        rk = full["ranks"]
        out["ranks"] = {k: rk.get(k) for k in ("backend", "world_size", "distinct_gpus") if k in rk}
    if detail_path:
        out["detail"] = detail_path
    out = _round_floats(out)
    text = json.dumps(out, separators=(", ", ": "), allow_nan=False)
    if len(text) > CONTRACT_MAX_CHARS:
        # never let prose push the line over the record's tail: drop the optional extras, then shorten the strings
        for k in ("ranks", "proof_roofline_frac", "root_handoff_ms", "proof_verify_ms", "prep_miss_ms", "value_incl_h2d_ms"):
            out.pop(k, None)
        if out.get("cpu_baseline"):
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:200]
        out["config"]["workload"] = out["config"]["workload"][:300]
        text = json.dumps(out, separators=(", ", ": "), allow_nan=False)
    assert len(text) <= CONTRACT_MAX_CHARS, f"final bench line is {len(text)} characters"
    return text


def write_detail(full, path):
    """Everything measured, next to the final line: <path> (and gpurun_out/ when that scratch directory exists, so a
    gpurun call brings it back)."""
    text = json.dumps(full, indent=1)
    paths = [path]
    scratch = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(scratch) and os.path.abspath(path) == os.path.join(ROOT, "bench_detail.json"):
        paths.append(os.path.join(scratch, os.path.basename(path)))
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(text + "\n")
        except OSError as e:   # a read-only checkout must not cost the run its number
            print(f"bench: could not write {p}: {e}", file=sys.stderr)


_FINAL_LINE = []


def emit(full, args):
    """Detail to its file now; the one contract line is printed by flush_final_line() as the process's LAST output - after
    the contexts are closed and the process group is destroyed, so that nothing a library prints on the way out follows it."""
    write_detail(full, args.detail_out)
    _FINAL_LINE.append(contract_line(full, os.path.basename(args.detail_out)))


def flush_final_line():
    """The contract line as the LAST line of stdout.  Libraries write to the C-level stdout too and, behind a pipe, only
    when their buffer is flushed at exit: RCCL's version banner ("RCCL version : .. Librccl path : ..", five lines) came out
    AFTER the JSON line on the first run under nccl.  So: flush every C stream first, print the line, then point file
    descriptor 1 at stderr - whatever is written to stdout from here on cannot follow the line."""
    sys.stderr.flush()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    for text in _FINAL_LINE:
        print(text, flush=True)
    if _FINAL_LINE:
        try:
            os.dup2(2, 1)
        except OSError:
            pass
    _FINAL_LINE.clear()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-height", type=int, default=20)
    ap.add_argument("--field", default="koala-bear")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-log-height", type=int, default=14)
    ap.add_argument("--tree", action="store_true",
                    help="BASELINE config 4: prove a 2-to-1 aggregation tree (leaves -> root) over the ranks instead of "
                         "independent layers")
    ap.add_argument("--tree-leaves", type=int, default=8)
    ap.add_argument("--leaf-log-height", type=int, default=15,
                    help="rows of a leaf layer (nodes have twice as many); the reference's real verifier circuits have 2^14..2^16")
    ap.add_argument("--tree-workers", type=int, default=1,
                    help="concurrent provers (one p3r_ctx = one HIP stream each) per rank: layers of this size do not fill an "
                         "MI355X on their own")
    ap.add_argument("--tree-verify-children", action="store_true", help="verify both children natively before proving a node")
    ap.add_argument("--tree-level-barriers", action="store_true", help="barrier between levels (per-level wall times)")
    ap.add_argument("--spans", action="store_true",
                    help="print the per-stage timers as tracing-forest spans under the reference's span names (stderr)")
    ap.add_argument("--full", action="store_true",
                    help="also run the secondary legs (config 2 knobs, config 0 base layer, D = 5, width-32 table, arity-4 MMCS, "
                         "ZK) and the second, 4x larger CPU sample: the profile rounds' form; the default run is the headline "
                         "with its roofline and cpu_baseline and finishes in well under a minute")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where everything that is not a contract field goes (the final stdout line stays under "
                         f"{CONTRACT_MAX_CHARS} characters)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group (nccl = RCCL unless P3R_BENCH_BACKEND says otherwise) also at world size 1: "
                         "barrier, all_reduce(MAX) of the timings and the root hand-off run through the collective library on one GPU")
    ap.add_argument("--no-small-layers", action="store_true", help="skip the 2^14/2^15/2^16-row layers")
    ap.add_argument("--no-sustained", action="store_true", help="skip the sustained leg (SUSTAINED_S seconds of back-to-back proofs after the timed region)")
    ap.add_argument("--no-quintic", action="store_true",
                    help="skip the D = 5 layer (KoalaBear quintic circuits: ALU, compact-D1 Poseidon2, recompose/coeff)")
    ap.add_argument("--no-config2", action="store_true",
                    help="skip the secondary measurement with BASELINE config 2's chain-length knobs")
    ap.add_argument("--zk", action="store_true",
                    help="--tree: every proof of the tree under the ZK configuration (HidingFriPcs: `recursive_aggregation --zk`)")
    ap.add_argument("--trees", type=int, default=1,
                    help="--tree: independent trees in flight (0 = one per rank: the throughput form, weak scaling); tree t "
                         "is placed with rank offset t, so every rank proves the same number of nodes")
    args = ap.parse_args()
    if not args.full:
        args.no_quintic = args.no_config2 = True

    # `python bench.py --gpus N` on its own starts the N ranks itself: the parent spawns
    # `python -m torch.distributed.run` BEFORE it has imported torch or touched a device (a process that has
    # initialised the GPU must never exec or fork workers), forwards the ranks' output and exits with their
    # status.  Under a launcher (WORLD_SIZE set) --gpus must agree with it.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return self_launch(args.gpus)
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks", file=sys.stderr)
        sys.exit(2)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank != 0:
        # only rank 0 speaks on stdout (one JSON line): whatever a library of another rank prints there goes to stderr
        sys.stdout.flush()
        os.dup2(2, 1)
    if world > 1:
        # the ranks set up side by side: cap their OpenMP teams BEFORE torch is imported - torch loads an OpenMP runtime,
        # and libgomp reads OMP_NUM_THREADS when it is loaded, not when a team starts
        cap = max(1, (os.cpu_count() or world) // world)
        cur = os.environ.get("OMP_NUM_THREADS")
        if cur is None or not cur.isdigit() or int(cur) > cap:
            os.environ["OMP_NUM_THREADS"] = str(cap)

    import torch
    import harness_lib
    import plonky3_recursion_amd as p3r
    import harness_adapters as wl

    dist = None
    # P3R_BENCH_BACKEND=gloo exercises the multi-rank path where the ranks cannot have a GPU each
    # (several ranks share device 0); the driver's runs use nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("P3R_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
    if ndev == 0:
        print("bench: no GPU visible to this process (torch.cuda.device_count() == 0); this benchmark has no CPU path",
              file=sys.stderr)
        sys.exit(2)
    if backend == "nccl" and world > 1 and ndev < world:
        # one rank per GPU over RCCL: a rank without a device of its own would share one and deadlock or fail inside
        # ncclCommInitRank with an error that does not say why
        print(f"bench: --gpus {world} under the nccl backend needs {world} visible GPUs, this node shows {ndev} "
              f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); P3R_BENCH_BACKEND=gloo runs the ranks on shared devices",
              file=sys.stderr)
        sys.exit(2)
    if backend != "nccl":
        local_rank = local_rank % max(ndev, 1)
    coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    if world > 1 or args.force_dist:
        if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:   # --force-dist without a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        import datetime
        limit = datetime.timedelta(seconds=300)   # a lost rank aborts the run instead of hanging it
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)

    if args.tree:
        return run_tree(args, torch, dist, rank, world, local_rank, coll_device, backend)

    field, log_h = args.field, args.log_height
    packing = p3r.TablePacking().with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
    # production-size layers first, on a quiet GPU: eight concurrent provers (their own contexts); measured after the
    # 2^20-row work of this process the same call reads 15 % lower
    small_tput = None
    if rank == 0 and world == 1 and not args.no_small_layers:
        small_tput = small_layer_throughput(p3r, wl, packing, field)
    ctx = p3r.Context(field=field, device=local_rank, **FRI)
    arrs = harness_lib.generate(field, log_h, seed=0x5EED0000 + rank, **GEN_KNOBS)
    cache = p3r.build_next_layer_prep(ctx, wl.circuit_from_arrays(arrs), p3r.FriRecursionBackend(),
                                      p3r.ProveNextLayerParams(table_packing=packing))
    cpd, pc = cache.circuit_prover_data, cache.prepared_circuit
    resident = pc.upload_inputs(wl.circuit_inputs_from_arrays(arrs))
    counts = [int(x) for x in arrs["counts"]] + [len(arrs["ops"]) // 8]

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    proof_len = 0
    last_proof = b""
    for _ in range(args.warmup):
        proof_len = len(pc.prove(resident))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last_proof = pc.prove(resident)
        proof_len = len(last_proof)
    barrier()
    dt = time.perf_counter() - t0
    # final hand-off of the finished proofs to rank 0 over RCCL/xGMI (outside the timed region:
    # it is one ~0.6 MB message per rank and has no counterpart in the reference)
    handoff_ms = None
    if dist is not None:
        try:
            from plonky3_recursion_amd.aggregation import gather_proofs_to_root
            h0 = time.perf_counter()
            got = gather_proofs_to_root(last_proof, dist, rank, world, device=coll_device)
            torch.cuda.synchronize()
            handoff_ms = (time.perf_counter() - h0) * 1e3
            if rank == 0:
                assert len(got) == world
        except Exception as e:  # never let the hand-off demo break the measurement
            handoff_ms = f"failed: {e}"
    # What was timed is checked: the last proof of the timed region goes through verify_all_tables
    # (batch_stark_prover.rs:1230-1268; native host verifier of the C-ABI library), as every layer
    # does in recursion/examples/recursive_fibonacci.rs:444-454.  A rejected proof fails the run.
    import hashlib
    proof_sha256 = hashlib.sha256(last_proof).hexdigest()
    v0 = time.perf_counter()
    try:
        cache.prover.verify_all_tables(cache.prover.wrap_proof(last_proof, cpd))
        proof_verified = True
    except Exception as e:
        print(f"bench: rank {rank}: the timed proof was REJECTED by verify_all_tables: {e}", file=sys.stderr)
        proof_verified = False
    verify_ms = (time.perf_counter() - v0) * 1e3
    # the same step with the circuit inputs handed over from host memory (public values + Merkle
    # siblings cross PCIe inside the call) - not `value`, reported next to it
    host_inputs = wl.circuit_inputs_from_arrays(arrs)
    same = pc.prove(host_inputs) == last_proof
    ctx.sync()
    h0 = time.perf_counter()
    for _ in range(3):
        pc.prove(host_inputs)
    ctx.sync()
    incl_h2d_ms = (time.perf_counter() - h0) / 3 * 1e3
    if not same:
        print("bench: host-input proof differs from the resident-input proof", file=sys.stderr)
        proof_verified = False
    del host_inputs
    # the same step back to back for SUSTAINED_S seconds (rank 0 of a one-GPU run): the timed region above is K steps -
    # half a second at the driver's K = 20 - and a utilisation sampler that looks at the GPU every five seconds can miss
    # it altogether; this leg is long enough to be seen, and says whether the rate holds once clocks and HBM have settled
    sustained = None
    if rank == 0 and world == 1 and not args.no_sustained and not args.no_small_layers:   # (profiling runs pass --no-small-layers)
        s0 = time.perf_counter()
        n_sus, all_same = 0, True
        while time.perf_counter() - s0 < SUSTAINED_S:
            all_same = (pc.prove(resident) == last_proof) and all_same
            n_sus += 1
        sus_s = time.perf_counter() - s0
        sustained = {"seconds": sus_s, "proofs": n_sus, "ms_per_step": sus_s / n_sus * 1e3, "vs_timed_region": (sus_s / n_sus) / (dt / args.steps),
                     "every_proof_identical_to_the_timed_one": all_same}
        if not all_same:
            print("bench: a proof of the sustained leg differs from the timed region's", file=sys.stderr)
            proof_verified = False
    # per-kernel-family times come from two EXTRA steps with HIP-event bracketing switched on, so
    # the event overhead is not inside `value`
    prof_steps = 2
    ctx.profile_enable(True)
    for _ in range(prof_steps):
        pc.prove(resident)
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    if args.spans and rank == 0:
        # the reference's tracing-forest view of the same step (scripts/benchmark.sh:87-101 parses it)
        print(p3r.span_report(prof, prof_steps), file=sys.stderr)

    own_dt = dt
    if dist is not None:
        t = torch.tensor([dt, 0.0 if proof_verified else 1.0], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        proof_verified = proof_verified and float(t[1].item()) == 0.0
    ranks = rank_report(torch, dist, world, backend, local_rank, coll_device, rank_ms=own_dt / args.steps * 1e3)
    ms_per_step = dt / args.steps * 1e3

    # prep-cache miss (recursion.rs:452-501, prep=None): preprocessed columns + LDE + commitment are
    # rebuilt from the circuit before the proof; once, on rank 0 at N = 1
    prep_miss_ms = None
    prep_miss_runs = None
    prep_breakdown = None
    small = {}
    if rank == 0 and world == 1:
        circ = wl.circuit_from_arrays(arrs)
        hin = wl.circuit_inputs_from_arrays(arrs)
        # two complete misses (nothing of the first survives into the second but the allocator's pool); the line
        # carries both and quotes the smaller: a single sample is exposed to whatever else the host does in those 50 ms
        prep_miss_runs = []
        cache2 = None
        for _ in range(2):
            if cache2 is not None:
                cache2.prepared_circuit.free()
            ctx.sync()
            m0 = time.perf_counter()
            cache2 = p3r.build_next_layer_prep(ctx, circ, p3r.FriRecursionBackend(),
                                               p3r.ProveNextLayerParams(table_packing=packing))
            miss_proof = cache2.prepared_circuit.prove(hin)
            ctx.sync()
            prep_miss_runs.append((time.perf_counter() - m0) * 1e3)
        prep_miss_ms = min(prep_miss_runs)
        # where the preparation spends it (a second, bracketed build; not part of prep_miss_ms)
        ctx.profile_enable(True)
        cache3 = p3r.build_next_layer_prep(ctx, circ, p3r.FriRecursionBackend(),
                                           p3r.ProveNextLayerParams(table_packing=packing))
        prep_breakdown = {k[6:]: v[0] for k, v in ctx.profile_read().items() if k.startswith("stage:prep_")}
        ctx.profile_enable(False)
        cache3.prepared_circuit.free()
        del cache3
        if miss_proof != last_proof:
            print("bench: prep-miss proof differs from the cached-prep proof", file=sys.stderr)
            proof_verified = False
        cache2.prepared_circuit.free()
        del circ, hin, cache2
        if not args.no_small_layers:
            small = small_layers(ctx, p3r, wl, packing, field)
    del arrs

    if rank == 0:
        p2w = ctx.poseidon2_trace_width
        k = packing.horner_packed_steps
        widths = [4, 4 * packing.public_lanes, 16 * packing.alu_lanes + ((k - 1) // 2 + 2 * (k - 1) + 1) * 4, p2w,
                  4 * packing.recompose_lanes]
        perms, hash_perms, hash_bytes, model_launches = workload_model(field, cpd.table_heights, widths, packing)
        kernel_ms = {kk: v[0] / prof_steps for kk, v in prof.items() if not kk.startswith("stage:")}
        stage_ms = {kk[6:]: v[0] / prof_steps for kk, v in prof.items() if kk.startswith("stage:") and not kk.startswith("stage:count:")}
        counted_perms = prof.get("stage:count:hash_rows_perms", (None,))[0]   # what the launches absorbed (csrc/profile.h::prof_count)
        if counted_perms is not None and counted_perms / prof_steps != workload_model(field, cpd.table_heights, widths, packing)[1]:
            print(f"bench: the leaf-hash launches absorbed {counted_perms / prof_steps:.0f} permutations per step, the model says "
                  f"{workload_model(field, cpd.table_heights, widths, packing)[1]}", file=sys.stderr)
        dominant = max(kernel_ms, key=kernel_ms.get) if kernel_ms else None
        hash_ms, hash_launches = prof.get("mmcs_hash_rows", (0.0, 0))
        launches_per_step = hash_launches / prof_steps
        if launches_per_step != model_launches:
            print(f"bench: {launches_per_step} k_mmcs_hash_rows launches per step, the model expects {model_launches}",
                  file=sys.stderr)
        avg_launch_ms = hash_ms / hash_launches if hash_launches else float("nan")
        achieved = (hash_bytes / launches_per_step) / (avg_launch_ms * 1e-3) / 1e9 if hash_launches else None
        line = {
            "metric": "prove_next_layer ms + Poseidon2 perms/s, KoalaBear 2^20-row circuit, 1/8 GPU",
            "value": ms_per_step,
            "unit": "ms",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": False,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 Montgomery (31-bit prime field, degree-4 extension); Poseidon2 hashing as exact integers in f64",
            "data": "synthetic",
            "config": {
                "workload": f"prove_next_layer (verifier-circuit run on the device + prove_all_tables) of the synthetic "
                            f"{field} 2^{log_h}-row recursion layer: tables const/public/alu/poseidon2/recompose, "
                            f"FRI blowup 4, arity<=4, 54 queries, 15-bit PoW; circuit inputs + NextLayerPrepCache "
                            f"resident in HBM",
                "circuit_levels": pc.levels,
                "field": field, "log_height": log_h, "table_heights": cpd.table_heights, "table_widths": widths,
                "ops": dict(zip(["const", "public", "alu", "poseidon2", "recompose", "witnesses", "circuit_ops"], counts)),
                "fri": FRI, "independent_proofs": world, "proof_bytes": proof_len,
                "parallelism": f"{world} independent proofs, one per GPU, no data-path collective",
            },
            "ranks": ranks,
            # what an N-GPU run of this entry is to be compared with (the tree entry has its own: run_tree)
            # (ranks that share a GPU - the gloo run on a one-GPU box - split it: solo = step x GPUs / ranks)
            "scaling_prediction": weak_scaling_prediction(ms_per_step * max(1, ranks.get("distinct_gpus") or world) / world),
            "proof_verified": proof_verified,
            "proof_sha256": proof_sha256,
            "proof_verify_ms": verify_ms,
            "value_incl_h2d_ms": incl_h2d_ms,
            "prep_miss_ms": prep_miss_ms,
            "prep_miss_ms_runs": prep_miss_runs,
            "prep_miss_breakdown_ms": prep_breakdown,
            "small_layers": small or None,
            "small_layer_throughput": small_tput,
            "sustained": sustained,
            "root_handoff_ms": handoff_ms,
            "poseidon2_perms_per_s": perms * world / (ms_per_step * 1e-3),
            "poseidon2_perms_per_step": perms,
            "kernel_ms_per_step": kernel_ms,
            "stage_wall_ms_per_step": stage_ms,
            "dominant_kernel_family": dominant,
        }
        # `roofline`: the dominant kernel against the roof that binds it.  MMCS leaf hashing is one Poseidon2 permutation
        # per 32 B absorbed: it is bound by VALU issue - the permutation runs in FP64 (exact integer arithmetic in
        # doubles, csrc/poseidon2_f64.hip.h), so its price is FP64 instructions per permutation x the FP64 issue rate
        # (guide: 78.6 TFLOP/s FP64 vector = 39.3 T FMA lane-ops/s).  Instructions per permutation and the measured
        # v_fma_f64 rate come from named files of profiles/<round>/ (committed_valu_model); the kernel's HBM side
        # (algorithmic bytes, PMC traffic) is the sub-object `hbm`.
        hash_total_ms = kernel_ms.get("mmcs_hash_rows", 0.0)
        insts, fma_rate, vsrc = committed_valu_model(field)
        ach = hash_perms / (hash_total_ms * 1e-3) if hash_total_ms else None
        traffic = pmc_traffic_bytes("k_mmcs_hash_rows") if (field, log_h) == ("koala-bear", 20) else None
        line["roofline"] = {
            "kernel": "k_mmcs_hash_rows",
            "bound": "valu-issue",
            "achieved": (ach * insts / 1e12) if ach and insts else None,
            "peak": FP64_FMA_SPEC / 1e12,
            "unit": "T FP64 lane-ops/s (MI355X_MICROARCH.md: 78.6 TFLOP/s FP64 vector = 39.3 T FMA lane-ops/s)",
            "frac": (ach * insts / FP64_FMA_SPEC) if ach and insts else None,
            "traffic": traffic,
            "avg_launch_ms": avg_launch_ms,
            "perms_per_s": ach,
            "perms_per_step_in_kernel": hash_perms,
            "perms_per_step_counted_by_the_library": (counted_perms / prof_steps) if counted_perms is not None else None,
            "valu_insts_per_perm": insts,
            "peak_perms_per_s": (FP64_FMA_SPEC / insts) if insts else None,
            "peak_measured_lane_ops_per_s": fma_rate,
            "frac_measured": (ach * insts / fma_rate) if ach and insts and fma_rate else None,
            "frac_null_reason": vsrc.get("refused"),
            # SURVEY 8d's own yardstick for this kernel: 616 modular products per permutation at the chip's measured
            # Montgomery rate (mul_lo / mul_hi form, microbench_int_rates.txt) - the FP64 formulation is measured against
            # the INTEGER roof the survey proposed
            "vs_survey_integer_roof": (ach / (MONT_PRODUCT_RATE_PLAIN / 616.0)) if ach else None,
            "survey_integer_roof_perms_per_s": MONT_PRODUCT_RATE_PLAIN / 616.0,
            # how each was obtained: tools/profile_round.sh (rocprofv3 --pmc SQ_INSTS_VALU over tools/pmc_hash_rows.py;
            # the v_fma_f64 line of tools/microbench/int_rates; --pmc FETCH_SIZE / WRITE_SIZE passes of this command with
            # bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction)
            "source_files": {"valu_insts_per_perm": vsrc.get("valu_insts_per_perm"),
                             "peak_measured_lane_ops_per_s": vsrc.get("peak_measured_lane_ops_per_s"),
                             "traffic": profile_file("pmc_traffic.json")[1],
                             "kernel_stats": profile_file("prove_next_layer_final_kernel_stats.csv")[1]},
            "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                    "algorithmic_bytes_per_launch": hash_bytes / launches_per_step if launches_per_step else None,
                    "traffic_vs_algorithmic": (traffic / (hash_bytes / launches_per_step)) if traffic and launches_per_step else None,
                    "note": "4 w + 32 B per LDE row; not the binding roof of this kernel"},
        }
        # The streaming families against HBM, from the same algorithmic byte counts as DESIGN.md §3/§7.
        line["hbm_families"] = hbm_families(cpd.table_heights, widths, packing, kernel_ms)
        # ... and against VALU issue: what binds the quotient, the NTT passes, the Merkle levels and the circuit run
        line["valu_families"] = valu_families(kernel_ms) if (field, log_h) == ("koala-bear", 20) else None
        line["proof_roofline"] = proof_roofline(field, cpd.table_heights, widths, packing, perms, insts, hash_bytes, ms_per_step,
                                                line["hbm_families"])
        if not args.no_cpu_baseline and world == 1:
            lh = args.cpu_baseline_log_height
            cdt, crun, cores = cpu_baseline(field, lh)
            # a second, four times larger sample: how the port scales (the line's own comparator stays the first)
            cdt2, crun2 = (cpu_baseline(field, lh + 2)[:2]) if args.full else (None, None)
            line["cpu_baseline"] = {
                "value": cdt * 1e3, "unit": "ms", "cores": cores, "kind": "port",
                "sample": f"same prove_next_layer (circuit run + prove, same table mix, same FRI parameters, same "
                          f"synthetic generator) at 2^{lh} rows = 1/{1 << (log_h - lh)} of the workload, oracle/ C++ "
                          f"restatement, OpenMP on {cores} threads (the circuit run is sequential, as in the reference)",
                "circuit_run_ms": crun * 1e3,
                "gpu_ms_same_sample": small.get(str(lh), {}).get("ms_per_step") if small else None,
                "samples": [{"log_height": lh, "ms": cdt * 1e3, "circuit_run_ms": crun * 1e3,
                             "gpu_ms": small.get(str(lh), {}).get("ms_per_step") if small else None}] +
                           ([{"log_height": lh + 2, "ms": cdt2 * 1e3, "circuit_run_ms": crun2 * 1e3,
                              "gpu_ms": small.get(str(lh + 2), {}).get("ms_per_step") if small else None}] if cdt2 else []),
                "scaling_4x_rows": (cdt2 / cdt) if cdt2 else None,
                "note": "a label, not a comparator: the oracle is a deliberately plain restatement (u64 % arithmetic, textbook "
                        "NTT).  The reference's own published figure is `published_reference_ms`.",
                "published_reference_ms": PUBLISHED_CPU_MS,
                "published_reference": "BASELINE.md: prove_next_layer of a real verifier circuit (~2^15 rows, steady state), "
                                       "Apple M4 Pro 14 cores; compare with small_layers['15']",
            }
        if not args.no_config2 and world == 1:
            # BASELINE config 2 (recursive_keccak.rs:386-399: layer 1 over a uni-stark Keccak proof), the way the
            # reference runs it: build_and_prove_next_layer = prove_next_layer with prep = None (recursion.rs:452-539),
            # i.e. preparation + circuit run + proof, from host inputs.  Config 2's chain-length knobs (Horner chains of
            # ~2600 steps, sponge chains of ~330 permutations) with INDEPENDENT sponge chains (one leaf hash per opened
            # row; Merkle paths may hang off them).  The proof is verified.  On rank 0 at N = 1 only.
            resident.free()
            pc.free()
            arrs2 = harness_lib.generate(field, log_h, seed=0x5EED0000, flags=harness_lib.INDEPENDENT_SPONGES, **CONFIG2_KNOBS)
            circ2, hin2 = wl.circuit_from_arrays(arrs2), wl.circuit_inputs_from_arrays(arrs2)
            n_ops2 = len(arrs2["ops"]) // 8
            del arrs2
            params2 = p3r.ProveNextLayerParams(table_packing=packing)
            miss = []
            for _ in range(3):
                ctx.sync()
                t2 = time.perf_counter()
                cache2 = p3r.build_next_layer_prep(ctx, circ2, p3r.FriRecursionBackend(), params2)
                proof2 = cache2.prepared_circuit.prove(hin2)
                ctx.sync()
                miss.append((time.perf_counter() - t2) * 1e3)
                if len(miss) < 3:
                    cache2.prepared_circuit.free()
            pc2 = cache2.prepared_circuit
            try:
                cache2.prover.verify_all_tables(cache2.prover.wrap_proof(proof2, pc2.circuit_prover_data))
                ok2 = True
            except Exception as e:
                print(f"bench: config 2: proof rejected: {e}", file=sys.stderr)
                ok2 = False
            res2 = pc2.upload_inputs(hin2)
            pc2.prove(res2)
            ctx.sync()
            t2 = time.perf_counter()
            for _ in range(3):
                pc2.prove(res2)
            ctx.sync()
            ms2 = (time.perf_counter() - t2) / 3 * 1e3
            ctx.profile_enable(True)
            pc2.prove(res2)
            prof2 = ctx.profile_read()
            ctx.profile_enable(False)
            line["config2_keccak_layer1_knobs"] = {
                "build_and_prove_ms": min(miss[1:]), "build_and_prove_ms_runs": miss, "proof_verified": ok2,
                "ms_per_step_cached_prep": ms2, "steps": 3, "circuit_ops": n_ops2, "circuit_levels": pc2.levels,
                "run_circuit_ms": prof2.get("stage:run_circuit", (None,))[0],
                "workload": f"prove_next_layer with prep = None (preparation + run + proof, host inputs) of the synthetic {field} "
                            f"2^{log_h}-row layer with Horner chains up to 2600 steps and independent sponge chains of 330 "
                            f"permutations (BASELINE config 2); ms_per_step_cached_prep = the same layer with a NextLayerPrepCache"}
            proof_verified = proof_verified and ok2
            line["proof_verified"] = proof_verified
            res2.free()
            pc2.free()
            resident = pc = None
        if not args.no_quintic and world == 1:
            # BASELINE config 0's base layer (recursive_fibonacci --n 1000: CircuitBuilder<F>, D = 1 traces,
            # TablePacking::new(1, 1)): prepare + run + prove on the device under ext_degree = 1, proof verified
            if resident is not None:
                resident.free()
                pc.free()
                resident = pc = None
            P = p3r.MODULUS[field] if hasattr(p3r, "MODULUS") else {"koala-bear": 0x7F000001, "baby-bear": 0x78000001}[field]
            NOW, n_fib = 0xFFFFFFFF, 1000
            ops0, ext0 = [[0, 0, 0, NOW, 0, NOW, 0, 1], [1, 0, 0, NOW, 1, 0, 0, 0], [0, 0, 0, NOW, 2, NOW, 1, 1]], [0, 1]
            fa, fb, a_w, b_w, nxt = 0, 1, 0, 2, 3
            for i in range(2, n_fib + 1):
                out_w = 1 if i == n_fib else nxt
                ops0.append([2, a_w, b_w, NOW, out_w, NOW, 0, 0])
                a_w, b_w, fa, fb, nxt = b_w, out_w, fb, (fa + fb) % P, nxt + 1
            ctx0 = p3r.Context(field=field, ext_degree=1, **FRI)
            tp0 = p3r.TablePacking(public_lanes=1, alu_lanes=1, horner_packed_steps=2).with_fri_params(FRI["log_final_poly_len"], FRI["log_blowup"])
            circ0 = p3r.Circuit(nxt - 1, np.array(ops0, dtype=np.uint32), np.array(ext0, dtype=np.uint32), np.array([1], dtype=np.uint32))
            t0 = time.perf_counter()
            cache0 = p3r.build_next_layer_prep(ctx0, circ0, p3r.FriRecursionBackend(), p3r.ProveNextLayerParams(table_packing=tp0))
            in0 = p3r.CircuitInputs(public_values=np.array([[fb]], dtype=np.uint32))
            proof0 = cache0.prepared_circuit.prove(in0)
            ctx0.sync()
            first_ms = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            for _ in range(10):
                cache0.prepared_circuit.prove(in0)
            ctx0.sync()
            ms0 = (time.perf_counter() - t0) / 10 * 1e3
            try:
                cache0.prover.verify_all_tables(cache0.prover.wrap_proof(proof0, cache0.prepared_circuit.circuit_prover_data))
                ok0 = True
            except Exception as e:
                print(f"bench: config 0 base layer: proof rejected: {e}", file=sys.stderr)
                ok0 = False
            line["config0_fibonacci_base_layer"] = {
                "ms_per_step": ms0, "steps": 10, "build_and_prove_ms": first_ms, "proof_verified": ok0, "proof_bytes": len(proof0),
                "ext_degree": 1, "table_heights": cache0.prepared_circuit.circuit_prover_data.table_heights,
                "workload": f"BASELINE config 0's base layer: the Fibonacci(n = {n_fib}) circuit over the base field (CircuitBuilder<F>, "
                            f"Const 2 / Public 1 / ALU 999 Adds, TablePacking::new(1, 1)) prepared, run and proved on the device "
                            f"(prove_next_layer, ext_degree = 1), {field}, the headline FRI parameters"}
            proof_verified = proof_verified and ok0
            line["proof_verified"] = proof_verified
            cache0.prepared_circuit.free()
            ctx0.close()
        if not args.no_quintic and world == 1 and field == "koala-bear":
            # SURVEY 8(f).4, D = 5: the table mix FriRecursionBackendD5 registers (backend/fri.rs:741-852) - Const, Public,
            # ALU over the quintic trinomial extension, compact-D1 Poseidon2, Recompose and Recompose with coefficient lookups - at the
            # prove_all_tables boundary with Traces resident in HBM, same height, same FRI parameters, proof verified.
            if resident is not None:
                resident.free()
                pc.free()
                resident = pc = None
            arrs5 = harness_lib.generate(field, log_h, seed=0x5EED0005, flags=harness_lib.RECOMPOSE_BOTH, ext_degree=5, **GEN_KNOBS)
            counts5 = [int(x) for x in arrs5["counts"]]
            prep5 = wl.circuit_prep_from_arrays(arrs5, ext_degree=5)   # both Recompose tables: `recompose`, `recompose/coeff`
            traces5 = wl.traces_from_arrays(arrs5, ext_degree=5)
            circ5, hin5 = wl.circuit_from_arrays(arrs5), wl.circuit_inputs_from_arrays(arrs5, 5)
            del arrs5
            # challenge degree 4: the D = 4 STARK configuration the reference's D = 5 unit tests prove under
            # (batch_stark_prover/tests.rs:844-1029); 5: koala_bear_quintic_params, the configuration of
            # recursive_fibonacci --quintic (LogUp, quotient, openings, FRI and transcript over five-word elements)
            for dc, key in ((4, "quintic_backend_layer"), (5, "quintic_challenge_layer")):
                ctx5 = p3r.Context(field=field, ext_degree=5, challenge_degree=dc, **FRI)
                cache5 = p3r.build_next_layer_prep(ctx5, prep5, p3r.FriRecursionBackendD5(),
                                                   p3r.ProveNextLayerParams(table_packing=packing))
                cpd5 = cache5.circuit_prover_data
                res5 = p3r.ResidentTraces(ctx5, cpd5, traces5)
                proof5 = cache5.prover.prove_all_tables(res5, cpd5)
                ctx5.sync()
                t5 = time.perf_counter()
                for _ in range(3):
                    cache5.prover.prove_all_tables(res5, cpd5)
                ctx5.sync()
                ms5 = (time.perf_counter() - t5) / 3 * 1e3
                try:
                    cache5.prover.verify_all_tables(proof5)
                    ok5 = True
                except Exception as e:
                    print(f"bench: D = 5 layer, challenge degree {dc}: proof rejected: {e}", file=sys.stderr)
                    ok5 = False
                ctx5.profile_enable(True)
                cache5.prover.prove_all_tables(res5, cpd5)
                prof5 = ctx5.profile_read()
                ctx5.profile_enable(False)
                line[key] = {
                    "ms_per_step": ms5, "steps": 3, "proof_verified": ok5, "proof_bytes": len(proof5.proof),
                    "ext_degree": 5, "challenge_degree": dc, "tables": [e.op_type for e in proof5.non_primitives],
                    "table_heights": cpd5.table_heights + [cpd5.recompose_coeff_height],
                    "ops": dict(zip(["const", "public", "alu", "poseidon2", "recompose", "witnesses", "recompose/coeff"], counts5)),
                    "kernel_ms": {k: v[0] for k, v in prof5.items() if not k.startswith("stage:")},
                    "workload": f"prove_all_tables (Traces resident in HBM) of the synthetic {field} 2^{log_h}-row D = 5 layer: "
                                f"const / public / alu over F[x]/(x^5 + x^2 - 1) / compact-D1 poseidon2 / recompose / recompose with "
                                f"coefficient lookups (the six tables a verifier circuit of FriRecursionBackendD5 fills), " + ("the D = 4 STARK configuration" if dc == 4 else
                                "the quintic STARK configuration (Challenge = F[x]/(x^5 + x^2 - 1))") + ", same FRI parameters"}
                proof_verified = proof_verified and ok5
                line["proof_verified"] = proof_verified
                res5.free()
                cpd5.free()
                if dc == 5:
                    # the same layer from the circuit boundary: the D = 5 verifier circuit (quintic ALU ops, base-mode
                    # Poseidon2 permutations, recompose/coeff ops fed by decomposition hints) prepared once, then run on
                    # the device and proved per step - prove_next_layer of a `--quintic` recursion layer
                    t5 = time.perf_counter()
                    pc5 = p3r.PreparedCircuit(ctx5, circ5, packing)
                    prep_ms5 = (time.perf_counter() - t5) * 1e3
                    rin5 = pc5.upload_inputs(hin5)
                    same5 = pc5.prove(rin5) == proof5.proof
                    ctx5.sync()
                    t5 = time.perf_counter()
                    for _ in range(3):
                        pc5.prove(rin5)
                    ctx5.sync()
                    line[key]["prove_next_layer_ms"] = (time.perf_counter() - t5) / 3 * 1e3
                    line[key]["circuit_prep_ms"] = prep_ms5
                    line[key]["circuit_levels"] = pc5.levels
                    line[key]["same_proof_as_from_traces"] = same5
                    proof_verified = proof_verified and same5
                    line["proof_verified"] = proof_verified
                    rin5.free()
                    pc5.free()
                    # the size of a real verifier circuit (2^15 rows) under the same configuration: what one steady-state
                    # layer of `recursive_fibonacci --quintic` costs here
                    a15 = harness_lib.generate(field, 15, seed=0x5EED0015, flags=harness_lib.RECOMPOSE_BOTH, ext_degree=5, **GEN_KNOBS)
                    c15 = wl.circuit_from_arrays(a15)
                    p3r.PreparedCircuit(ctx5, c15, packing).free()   # first preparation of this size: allocator pool, table caches
                    ctx5.sync()
                    t5 = time.perf_counter()
                    pc15 = p3r.PreparedCircuit(ctx5, c15, packing)
                    ctx5.sync()
                    prep15 = (time.perf_counter() - t5) * 1e3
                    rin15 = pc15.upload_inputs(wl.circuit_inputs_from_arrays(a15, 5))
                    raw15 = pc15.prove(rin15)
                    ctx5.sync()
                    t5 = time.perf_counter()
                    for _ in range(10):
                        pc15.prove(rin15)
                    ctx5.sync()
                    ms15 = (time.perf_counter() - t5) / 10 * 1e3
                    try:
                        cache15 = pc15.circuit_prover_data
                        p3r.BatchStarkProver(ctx5).verify_all_tables(p3r.BatchStarkProver(ctx5).wrap_proof(raw15, cache15))
                        ok15 = True
                    except Exception as e:
                        print(f"bench: 2^15-row quintic layer: proof rejected: {e}", file=sys.stderr)
                        ok15 = False
                    line[key]["small_layer_2p15"] = {"prove_next_layer_ms": ms15, "steps": 10, "proof_bytes": len(raw15),
                                                     "proof_verified": ok15, "circuit_prep_ms": prep15}
                    proof_verified = proof_verified and ok15
                    line["proof_verified"] = proof_verified
                    rin15.free()
                    pc15.free()
                ctx5.close()
        if not args.no_quintic and world == 1:
            # SURVEY 8(f).4, arity-4 width-32: a layer that holds the width-32 Poseidon2 table (the arity-4 MMCS rows of a
            # mixed-config verifier circuit: W16 challenger + W32 MMCS, recursive_aggregation.rs:902-1000) next to the five
            # tables of the headline layer, at the prove_all_tables boundary with Traces resident in HBM; proof verified
            if resident is not None:
                resident.free()
                pc.free()
                resident = pc = None
            arrsw = harness_lib.generate(field, log_h, seed=0x5EED0032, flags=harness_lib.P2_W32, **GEN_KNOBS)
            countsw = [int(x) for x in arrsw["counts"]]
            ctxw = p3r.Context(field=field, **FRI, allow_unpinned_w32_defaults=True)
            cpdw = p3r.CircuitProverData(ctxw, wl.circuit_prep_from_arrays(arrsw), packing)
            resw = p3r.ResidentTraces(ctxw, cpdw, wl.traces_from_arrays(arrsw))
            del arrsw
            proverw = p3r.BatchStarkProver(ctxw)
            proofw = proverw.prove_all_tables(resw, cpdw)
            ctxw.sync()
            tw = time.perf_counter()
            for _ in range(3):
                proverw.prove_all_tables(resw, cpdw)
            ctxw.sync()
            msw = (time.perf_counter() - tw) / 3 * 1e3
            try:
                proverw.verify_all_tables(proofw)
                okw = True
            except Exception as e:
                print(f"bench: width-32 layer: proof rejected: {e}", file=sys.stderr)
                okw = False
            ctxw.profile_enable(True)
            proverw.prove_all_tables(resw, cpdw)
            profw = ctxw.profile_read()
            ctxw.profile_enable(False)
            line["width32_table_layer"] = {
                "ms_per_step": msw, "steps": 3, "proof_verified": okw, "proof_bytes": len(proofw.proof),
                "tables": [e.op_type for e in proofw.non_primitives],
                "table_heights": cpdw.table_heights + [cpdw.p2w_height],
                "ops": dict(zip(["const", "public", "alu", "poseidon2", "recompose", "witnesses", "recompose/coeff", "poseidon2_w32"], countsw)),
                "width32_table": {"main_columns": (319 if field == "koala-bear" else 604) + 4, "preprocessed_columns": 48, "bus_interactions": 16,
                                  "constants": "self-generated defaults (p3r_config.poseidon2_w32_rc / _diag = NULL): unpinned"},
                "kernel_ms": {k: v[0] for k, v in profw.items() if not k.startswith("stage:")},
                "workload": f"prove_all_tables (Traces resident in HBM) of the synthetic {field} 2^{log_h}-row layer with a sixth table: "
                            f"const / public / alu / poseidon2 (width 16) / poseidon2 width 32 (arity-4 Merkle chains with injection and "
                            f"bridge levels, rate-24 sponge chains: 2^{log_h - 2} rows) / recompose, same FRI parameters"}
            # the width-32 legs run on SELF-GENERATED constants (P3R_EXT_UNPINNED_W32_DEFAULTS): their self-verification says
            # prover and verifier agree with each other, not with upstream - it is reported here and does not feed the
            # headline `proof_verified`
            line.setdefault("unpinned_legs_self_verified", {})["width32_table_layer"] = okw
            resw.free()
            cpdw.free()
            ctxw.close()
            # SURVEY 8(f).4, arity-4: the headline layer proved under the prover's OWN arity-4 MMCS (p3r_config.mmcs_arity = 4:
            # MyMmcsArity4 of `recursive_aggregation --arity4` - width-32 sponge leaves of rate 24, 4-to-1 levels, W16
            # challenger): same circuit, same inputs, same FRI parameters as the headline; prove_next_layer with a cache
            arrs4 = harness_lib.generate(field, log_h, seed=0x5EED0000, **GEN_KNOBS)
            ctx4 = p3r.Context(field=field, mmcs_arity=4, **FRI, allow_unpinned_w32_defaults=True)
            pc4 = p3r.PreparedCircuit(ctx4, wl.circuit_from_arrays(arrs4), packing)
            rin4 = pc4.upload_inputs(wl.circuit_inputs_from_arrays(arrs4))
            del arrs4
            raw4 = pc4.prove(rin4)
            ctx4.sync()
            t4 = time.perf_counter()
            for _ in range(5):
                pc4.prove(rin4)
            ctx4.sync()
            ms4 = (time.perf_counter() - t4) / 5 * 1e3
            try:
                prover4 = p3r.BatchStarkProver(ctx4)
                prover4.verify_all_tables(prover4.wrap_proof(raw4, pc4.circuit_prover_data))
                ok4 = True
            except Exception as e:
                print(f"bench: arity-4 MMCS layer: proof rejected: {e}", file=sys.stderr)
                ok4 = False
            ctx4.profile_enable(True)
            pc4.prove(rin4)
            prof4 = ctx4.profile_read()
            ctx4.profile_enable(False)
            line["arity4_mmcs_layer"] = {
                "ms_per_step": ms4, "steps": 5, "proof_verified": ok4, "proof_bytes": len(raw4),
                "kernel_ms": {k: v[0] for k, v in prof4.items() if not k.startswith("stage:")},
                "constants": "self-generated defaults (p3r_config.poseidon2_w32_rc / _diag = NULL): unpinned",
                "leaf_roofline": w32_leaf_roofline(field, cpd.table_heights, widths, packing, prof4.get("mmcs_hash_rows", (0.0,))[0]),
                "workload": f"the headline prove_next_layer (same circuit, inputs, tables and FRI parameters) with every commitment - "
                            f"traces, LogUp columns, quotient chunks, FRI commit phases - under the arity-4 MMCS over the width-32 "
                            f"permutation; challenger on the width-16 permutation.  The circuit itself holds width-16 rows only: "
                            f"arity4_recursion_layer is the layer whose verifier circuit fills the width-32 table"}
            line.setdefault("unpinned_legs_self_verified", {})["arity4_mmcs_layer"] = ok4
            rin4.free()
            pc4.free()
            # ... and the layer `recursive_aggregation --arity4` proves (recursive_aggregation.rs:902-1046): the verifier circuit
            # of an arity-4 proof - width-16 challenger rows AND width-32 MMCS rows (P3R_OP_POSEIDON2_W32_PERM: leaf sponges of
            # rate 24 that seed 4-to-1 compression chains with injection and bridge levels) - run on the device and proved
            # under the arity-4 MMCS, from the circuit boundary with a cache
            arrsr = harness_lib.generate(field, log_h, seed=0x5EED0032, flags=harness_lib.P2_W32_OPS, **GEN_KNOBS)
            countsr = [int(x) for x in arrsr["counts"]]
            tr = time.perf_counter()
            pcr = p3r.PreparedCircuit(ctx4, wl.circuit_from_arrays(arrsr), packing)
            ctx4.sync()
            prepr_ms = (time.perf_counter() - tr) * 1e3
            rinr = pcr.upload_inputs(wl.circuit_inputs_from_arrays(arrsr))
            del arrsr
            rawr = pcr.prove(rinr)
            ctx4.sync()
            tr = time.perf_counter()
            for _ in range(5):
                pcr.prove(rinr)
            ctx4.sync()
            msr = (time.perf_counter() - tr) / 5 * 1e3
            try:
                prover4.verify_all_tables(prover4.wrap_proof(rawr, pcr.circuit_prover_data))
                okr = True
            except Exception as e:
                print(f"bench: arity-4 recursion layer: proof rejected: {e}", file=sys.stderr)
                okr = False
            ctx4.profile_enable(True)
            pcr.prove(rinr)
            profr = ctx4.profile_read()
            ctx4.profile_enable(False)
            line["arity4_recursion_layer"] = {
                "ms_per_step": msr, "steps": 5, "proof_verified": okr, "proof_bytes": len(rawr), "circuit_prep_ms": prepr_ms,
                "prepared_on_device": pcr.prepared_on_device, "circuit_levels": pcr.levels,
                "table_heights": pcr.circuit_prover_data.table_heights + [pcr.circuit_prover_data.p2w_height],
                "ops": dict(zip(["const", "public", "alu", "poseidon2", "recompose", "witnesses", "recompose/coeff", "poseidon2_w32"], countsr)),
                "kernel_ms": {k: v[0] for k, v in profr.items() if not k.startswith("stage:")},
                "run_circuit_ms": profr.get("stage:run_circuit", (None,))[0],
                "constants": "self-generated defaults (p3r_config.poseidon2_w32_rc / _diag = NULL): unpinned",
                "workload": f"prove_next_layer (verifier-circuit run on the device + prove_all_tables, cached preparation) of the synthetic "
                            f"{field} 2^{log_h}-row layer of an arity-4 recursion: six tables - const / public / alu / poseidon2 width 16 "
                            f"(challenger) / poseidon2 width 32 (MMCS: 2^{log_h - 2} rows run from P3R_OP_POSEIDON2_W32_PERM ops) / recompose - "
                            f"every commitment under the arity-4 MMCS, same FRI parameters"}
            line.setdefault("unpinned_legs_self_verified", {})["arity4_recursion_layer"] = okr
            rinr.free()
            pcr.free()
            ctx4.close()
            # SURVEY 8(f).4, ZK: the headline layer under HidingFriPcs (p3r_config.zk = 1: create_config_zk of
            # recursion/examples/common/mod.rs:511-553 - two random codewords, seeded RNG): every commitment over the
            # extended trace domain (twice the rows, two more columns), a random round, eight masked quotient chunks per
            # constrained table instead of two, LogUp packed in triples.  Same circuit, inputs and FRI parameters;
            # prove_next_layer with a cache; every timed proof is a different proof, the first and the last are verified
            arrsz = harness_lib.generate(field, log_h, seed=0x5EED0000, **GEN_KNOBS)
            ctxz = p3r.Context(field=field, zk=1, num_random_codewords=2, **FRI)   # keyed from the operating system: the production form
            pcz = p3r.PreparedCircuit(ctxz, wl.circuit_from_arrays(arrsz), packing)
            rinz = pcz.upload_inputs(wl.circuit_inputs_from_arrays(arrsz))
            del arrsz
            rawz = pcz.prove(rinz)
            ctxz.sync()
            tz = time.perf_counter()
            for _ in range(5):
                lastz = pcz.prove(rinz)
            ctxz.sync()
            msz = (time.perf_counter() - tz) / 5 * 1e3
            try:
                proverz = p3r.BatchStarkProver(ctxz)
                for rz in (rawz, lastz):
                    proverz.verify_all_tables(proverz.wrap_proof(rz, pcz.circuit_prover_data))
                okz = rawz != lastz
                if not okz:
                    print("bench: ZK layer: two proofs of one input are identical", file=sys.stderr)
            except Exception as e:
                print(f"bench: ZK layer: proof rejected: {e}", file=sys.stderr)
                okz = False
            ctxz.profile_enable(True)
            pcz.prove(rinz)
            profz = ctxz.profile_read()
            ctxz.profile_enable(False)
            line["zk_layer"] = {
                "ms_per_step": msz, "steps": 5, "proof_verified": okz, "proof_bytes": len(rawz),
                "vs_headline": msz / ms_per_step,
                "num_random_codewords": 2, "proofs_made": ctxz.zk_nonce,
                "kernel_ms": {k: v[0] for k, v in profz.items() if not k.startswith("stage:")},
                "leaf_roofline": measured_leaf_roofline(field, profz),
                "parity": "byte parity with upstream is undefined for a randomised proof (the prover side of HidingFriPcs is un-vendored): "
                          "accepted by the native verifier here, by both verifiers and byte-identical to the oracle under the shared "
                          "counter-based randomness in tests/test_gpu_zk.py",
                "workload": f"the headline prove_next_layer (same circuit, inputs, tables and FRI parameters) under the ZK configuration: "
                            f"traces, LogUp columns and preprocessed columns committed over 2^{log_h + 1}-row extended domains with two "
                            f"random codeword columns each, a random round, 2^(log_qd + 1) masked quotient chunks"}
            proof_verified = proof_verified and okz
            line["proof_verified"] = proof_verified
            rinz.free()
            pcz.free()
            ctxz.close()
            # ... and with the HIDING MMCS on top (p3r_config.mmcs_salt_elems = 4: MerkleTreeHidingMmcs for the input trees and
            # the FRI commit-phase trees, the configuration of recursion/tests/zk_hiding_mmcs.rs - "the upstream-recommended ZK
            # setup"): one 4-column salt matrix per committed LDE, hashed with its rows and opened with them
            arrsh = harness_lib.generate(field, log_h, seed=0x5EED0000, **GEN_KNOBS)
            ctxh = p3r.Context(field=field, zk=1, num_random_codewords=2, mmcs_salt_elems=4, **FRI)
            pch = p3r.PreparedCircuit(ctxh, wl.circuit_from_arrays(arrsh), packing)
            rinh = pch.upload_inputs(wl.circuit_inputs_from_arrays(arrsh))
            del arrsh
            rawh = pch.prove(rinh)
            ctxh.sync()
            th = time.perf_counter()
            for _ in range(3):
                lasth = pch.prove(rinh)
            ctxh.sync()
            msh = (time.perf_counter() - th) / 3 * 1e3
            try:
                proverh = p3r.BatchStarkProver(ctxh)
                for rh in (rawh, lasth):
                    proverh.verify_all_tables(proverh.wrap_proof(rh, pch.circuit_prover_data))
                okh = rawh != lasth
            except Exception as e:
                print(f"bench: ZK layer under the hiding MMCS: proof rejected: {e}", file=sys.stderr)
                okh = False
            ctxh.profile_enable(True)
            pch.prove(rinh)
            profh = ctxh.profile_read()
            ctxh.profile_enable(False)
            line["zk_hiding_mmcs_layer"] = {
                "ms_per_step": msh, "steps": 3, "proof_verified": okh, "proof_bytes": len(rawh), "vs_headline": msh / ms_per_step,
                "vs_zk_layer": msh / msz, "mmcs_salt_elems": 4,
                "kernel_ms": {k: v[0] for k, v in profh.items() if not k.startswith("stage:")},
                "workload": "the zk_layer leg with every input tree and every FRI commit-phase tree salted (four elements per leaf "
                            "and matrix); byte-identical to the oracle under shared randomness in tests/test_gpu_hiding_mmcs.py"}
            proof_verified = proof_verified and okh
            line["proof_verified"] = proof_verified
            rinh.free()
            pch.free()
            ctxh.close()
        emit(line, args)
    if resident is not None:
        resident.free()
        pc.free()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    flush_final_line()
    if not proof_verified or (rank == 0 and not all(line.get("unpinned_legs_self_verified", {}).values())):
        sys.exit(3)


if __name__ == "__main__":
    main()
