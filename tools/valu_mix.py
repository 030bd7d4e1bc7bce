#!/usr/bin/env python3
"""Instruction-class mix of the integer kernel families, and the issue floor it implies -> profiles/<round>/valu_mix.json.

bench.py's `valu_families` used to divide every family's VALU lane-instructions by ONE rate, 39.3 T/s - the FP64 FMA rate.
That is the right price for the two Poseidon2 families (FP64 arithmetic) and the wrong one for integer kernels: a gfx950
SIMD retires 32-bit adds at up to twice that rate and v_mul_hi_u32 / v_mad_u64_u32 at a half / a third of it
(tools/microbench/valu_classes.hip measures one opcode per kernel).  This tool prices a family by its MIX:

    floor_ms(family) = N_valu(family) x sum_opcode share(opcode) / rate(opcode)

  N_valu      dynamic: SQ_INSTS_VALU x 64 of the family's kernels per proof (profiles/<round>/pmc_sq.json)
  share       static: the opcode histogram of the disassembled kernel INSTANCES the run launched (the code objects inside
              plonky3_recursion_amd/libp3r_hip.so, llvm-objdump), instances weighted by their share of the family's
              time in the rocprofv3 kernel stats.  Loops count once; the hot kernels here are unrolled straight-line code
              (butterflies, constraint evaluation), so the static mix is the dynamic mix to within their small loops.
  rate        measured lane-instructions/s of that opcode (profiles/<round>/microbench_valu_classes.txt); an opcode that was
              not measured takes the rate of its class representative (CLASS_OF below) - the FASTER choice where in doubt, so
              that the floor stays a lower bound and the fraction is not flattered.

usage: python tools/valu_mix.py <round> [--lib path/to/libp3r_hip.so]     (no GPU needed: reads committed profile files)
"""
import collections
import csv
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

# opcode -> the measured opcode that stands for it when it was not measured itself
CLASS_OF = [
    (r"^v_(fma|add|mul|min|max|rndne|fract|floor|trunc|ldexp|cvt)_f64|^v_cvt_f64|^v_cvt_[ui]32_f64", "v_fma_f64"),
    (r"^v_mad_[ui]64_[ui]32", "v_mad_u64_u32"),
    (r"^v_mul_hi_[ui]32", "v_mul_hi_u32+v_or_b32"),
    (r"^v_mul_lo_[ui]32", "v_mul_lo_u32"),
    (r"^v_(mul|mad)_[ui]32_[ui]24|^v_mul_hi_[ui]32_[ui]24", "v_mad_u32_u24"),
    (r"^v_(add3|lshl_add|add_lshl|lshl_or|and_or|or3|xad|bfe|bfi|alignbit|alignbyte|perm|med3|min3|max3|sad)_", "v_add3_u32"),
    (r"^v_(addc|subb|subbrev)_co", "v_addc_co_u32"),
    (r"^v_(add|sub|subrev)_co", "v_add_co_u32"),
    (r"^v_cmp|^v_cmpx", "v_cmp_lt_u32"),
    (r"^v_cndmask", "v_cmp_lt_u32"),
    (r"_dpp$", "v_add_u32_dpp"),
    (r"^v_(readlane|readfirstlane|writelane|permlane)", "v_mov_b32"),
    (r"^v_lshlrev_b64|^v_lshrrev_b64|^v_ashrrev_i64", "v_mad_u64_u32"),
    (r"^v_", "v_add_u32"),
]
# rates to fall back on until microbench_valu_classes.txt exists for the round (microbench_int_rates.txt of round 5)
FALLBACK_RATES = {"v_add_u32": 56.43, "v_mul_lo_u32": 36.11, "v_mul_hi_u32+v_or_b32": 19.51, "v_mad_u64_u32": 17.0,
                  "v_add3_u32": 21.61, "v_fma_f64": 36.62}


def profile_file(rnd, name):
    for r in [rnd] + ["r%02d" % k for k in range(int(rnd[1:]) - 1, 1, -1)]:
        p = os.path.join(ROOT, "profiles", r, name)
        if os.path.exists(p):
            return p, "profiles/%s/%s" % (r, name)
    return None, None


def load_rates(rnd):
    p, src = profile_file(rnd, "microbench_valu_classes.txt")
    rates = {}
    if p:
        for ln in open(p):
            f = ln.split()
            if len(f) >= 2 and not ln.startswith("#"):
                try:
                    v = float(f[1])
                except ValueError:
                    continue
                # v_cndmask_b32 reads 6.7 T/s in the one-opcode loop (sixteen back-to-back reads of VCC as a mask: a
                # property of that loop, not of a select behind its compare): not used as a price; the opcode takes the
                # compare's rate (CLASS_OF)
                if v >= 10.0:
                    rates[f[0]] = v
    if not rates:
        rates, src = dict(FALLBACK_RATES), "tools/valu_mix.py::FALLBACK_RATES (profiles/r05/microbench_int_rates.txt)"
    return rates, src


def rate_of(op, rates):
    """(T lane-instructions/s, the measured opcode it was taken from)"""
    base = re.sub(r"_(e32|e64|sdwa)$", "", op)
    if base in rates:
        return rates[base], base
    for pat, rep in CLASS_OF:
        if re.search(pat, base):
            if rep in rates:
                return rates[rep], rep
            if rep == "v_mul_hi_u32+v_or_b32" and "v_mul_hi_u32" in rates:
                return rates["v_mul_hi_u32"], "v_mul_hi_u32"
    return rates.get("v_add_u32", 56.43), "v_add_u32"


def disassemble(lib):
    """demangled kernel name -> Counter(opcode) over the gfx950 code objects bundled in `lib`."""
    tmp = tempfile.mkdtemp(prefix="valu_mix_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", local], check=True, capture_output=True, cwd=tmp)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True,
                                 capture_output=True, text=True).stdout
            cur = None
            for ln in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    cur = out.setdefault(m.group(1), collections.Counter())
                    continue
                if cur is None:
                    continue
                ln = ln.strip()
                if not ln or ln.startswith((";", "//")):
                    continue
                op = ln.split()[0]
                if op.startswith(("v_", "s_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
                    cur[op] += 1
        names = list(out)
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return {d: out[n] for n, d in zip(names, dem)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r06"
    lib = os.path.join(ROOT, "plonky3_recursion_amd", "libp3r_hip.so")
    if "--lib" in sys.argv:
        lib = sys.argv[sys.argv.index("--lib") + 1]
    sys.path.insert(0, ROOT)
    import bench
    rates, rate_src = load_rates(rnd)
    stats_p, stats_src = profile_file(rnd, "prove_next_layer_final_kernel_stats.csv")
    inst_ns = {}
    for r in csv.DictReader(open(stats_p)):
        inst_ns[re.sub(r"^void ", "", r["Name"])] = float(r["TotalDurationNs"])
    code = {re.sub(r"^void ", "", k): v for k, v in disassemble(lib).items()}
    fams = {}
    for fam, (_, kernels) in bench.VALU_FAMILY_KERNELS.items():
        mix = collections.Counter()
        used, total_ns = [], 0.0
        for name, ns in inst_ns.items():
            m = re.search(r"(k_[a-z0-9_]+)", name)
            if not m or m.group(1) not in kernels or name not in code:
                continue
            c = code[name]
            n_valu = sum(v for k, v in c.items() if k.startswith("v_"))
            if not n_valu:
                continue
            for k, v in c.items():
                if k.startswith("v_"):
                    mix[k] += ns * v / n_valu
            used.append({"instance": name, "ms_in_run": ns / 1e6, "static_valu": n_valu,
                         "static_other": {p: sum(v for k, v in c.items() if k.startswith(p)) for p in ("s_", "ds_", "global_", "buffer_")}})
            total_ns += ns
        if not total_ns:
            continue
        share = {k: v / total_ns for k, v in mix.items()}
        # seconds per lane-instruction of the mix, in units of 1e-12 (1 / T per s)
        cost = sum(s / rate_of(k, rates)[0] for k, s in share.items())
        by_rate = collections.Counter()
        for k, s in share.items():
            by_rate[rate_of(k, rates)[1]] += s
        fams[fam] = {
            "effective_rate_T_per_s": 1.0 / cost,
            "share_by_priced_opcode": dict(sorted(by_rate.items(), key=lambda kv: -kv[1])),
            "top_opcodes": dict(sorted(share.items(), key=lambda kv: -kv[1])[:16]),
            "instances": sorted(used, key=lambda u: -u["ms_in_run"])[:12],
        }
    out = {
        "provenance": "tools/valu_mix.py: static opcode histogram of the kernel instances of " + (stats_src or "?") +
                      " (disassembly of the built libp3r_hip.so, instances weighted by their time), priced with " + rate_src,
        "kernel_sources_sha256": bench.kernel_source_digest(("kernels_ntt2.hip.h", "kernels_ntt.hip.h", "kernels_stark.hip.h", "air_device.hip.h")),
        "rates_T_per_s": rates,
        "families": fams,
    }
    os.makedirs(os.path.join(ROOT, "profiles", rnd), exist_ok=True)
    path = os.path.join(ROOT, "profiles", rnd, "valu_mix.json")
    json.dump(out, open(path, "w"), indent=1)
    for fam, f in fams.items():
        print("%-14s effective %.1f T/s   %s" % (fam, f["effective_rate_T_per_s"],
                                                 ", ".join("%s %.0f%%" % (k, 100 * v) for k, v in list(f["share_by_priced_opcode"].items())[:6])))
    print("wrote", path)


if __name__ == "__main__":
    main()
