#!/usr/bin/env python3
"""First contact with cargo as a configuration search, not a debugging session.

Given a layer fixture of tools/rust_pin (tests/golden/rust_fibonacci_layer_<field>.json, rust_npo_layer_<field>.json,
rust_fibonacci_base_layer_<field>.json: proof bytes made by the REFERENCE), walk the protocol details the in-tree sources
do not pin - the `[EXT]` switches of DESIGN.md section 4, each a `p3r_config` field - through the CPU oracle and print
the configuration under which this repo reproduces the reference's bytes, or the first structure of the proof no
switch reproduces:

    round constants            taken from the fixture                      -> p3r_config.poseidon2_rc
    field-element encoding     Montgomery word | canonical                 -> P3R_PROVE_CANONICAL_FIELD_ENCODING
    LogUp packing              greedy same-bus packing | one column each   -> ext_choices & P3R_EXT_LOOKUP_UNPACKED
    FRI folding schedule       read off the reference's proof              -> fri_log_arities (None when the rule gives it)
    proof-of-work witnesses    smallest | whatever the reference found     (diagnosis: the library always takes the smallest)
    struct field order         located block by block in the bytes         -> proof_layout (18 bytes: batch | fri | opened)

The search needs no agreement on field order to start: a candidate (encoding, packing, schedule) is proved by the oracle
under the identity layout, the proof is cut into the serialised FIELDS of its three structs, and each field is looked up
in the reference's bytes as a contiguous block.  All blocks found = the values agree; their positions give the layout.
A block that is never found is reported with its place in transcript order - the first such block names the stage
where prover and reference part ways (circuit-prover/src/batch_stark_prover.rs:1203-1222 is the call being matched).

usage: python tools/resolve_pins.py tests/golden/rust_fibonacci_layer_koala_bear.json [...]
CPU only.  Test infrastructure / tooling: drives the oracle, never the product."""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import circuit_lib as cl      # noqa: E402
import fib_lib                # noqa: E402
import layer_lib              # noqa: E402
import oracle_lib             # noqa: E402
import proof_codec as pc      # noqa: E402

FIELD_OF = {"koala_bear": "koala-bear", "baby_bear": "baby-bear"}
BATCH_FIELDS = ["commitments", "opened_values", "opening_proof", "global_lookup_data", "degree_bits"]
FRI_FIELDS = ["commit_phase_commits", "commit_pow_witnesses", "query_proofs", "final_poly", "query_pow_witness"]
OPENED_FIELDS = ["trace_local", "trace_next", "preprocessed_local", "preprocessed_next", "quotient_chunks", "random",
                 "permutation_local", "permutation_next"]


# ---- serialised blocks of one proof (identity layout): what must occur, contiguously, in the other prover's bytes
def _v(x):
    return pc._varint(x)


def _fes(xs):
    return b"".join(_v(x) for x in xs)


def _vec(xs, item):
    return _v(len(xs)) + b"".join(item(x) for x in xs)


def _opt(x, item):
    return b"\x00" if x is None else b"\x01" + item(x)


def _vec_ef(v):
    return _vec(v, _fes)


def _cap(c):
    return _vec(c, _fes)


def opened_blocks(o):
    return [_vec_ef(o["trace_local"]), _opt(o["trace_next"], _vec_ef), _opt(o["preprocessed_local"], _vec_ef),
            _opt(o["preprocessed_next"], _vec_ef), _vec(o["quotient_chunks"], _vec_ef), _opt(o["random"], _vec_ef),
            _vec_ef(o["permutation_local"]), _vec_ef(o["permutation_next"])]


def fri_blocks(f):
    def query(q):
        return (_vec(q["input_proof"], lambda b: _vec(b["opened_values"], lambda r: _vec(r, _v)) + _vec(b["opening_proof"], _fes)) +
                _vec(q["commit_phase_openings"], lambda s: bytes([s["log_arity"]]) + _vec_ef(s["sibling_values"]) + _vec(s["opening_proof"], _fes)))
    return [_vec(f["commit_phase_commits"], _cap), _vec(f["commit_pow_witnesses"], _v), _vec(f["query_proofs"], query),
            _vec_ef(f["final_poly"]), _v(f["query_pow_witness"])]


def batch_blocks(p):
    c = p["commitments"]
    commitments = _cap(c["main"]) + _opt(c["permutation"], _cap) + _cap(c["quotient"]) + _opt(c["random"], _cap)
    opened = _v(len(p["opened"])) + b"".join(b"".join(opened_blocks(o)) for o in p["opened"])
    fri = b"".join(fri_blocks(p["opening_proof"]))
    return [commitments, opened, fri, _vec(p["lookup_terminals"], lambda t: _opt(t, _fes)), _vec(p["degree_bits"], _v)]


def order_of(blocks, hay, start=0, names=None):
    """Positions of `blocks` inside `hay` as a tiling from `start`: returns (permutation, end) or (None, name of the first
    block - in the given order - that occurs nowhere).  Empty / duplicate blocks are placed greedily."""
    n = len(blocks)
    used, perm, at = [False] * n, [], start
    for _ in range(n):
        nxt = next((i for i in range(n) if not used[i] and hay.startswith(blocks[i], at)), None)
        if nxt is None:
            missing = next((i for i in range(n) if not used[i] and hay.find(blocks[i]) < 0), None)
            return None, (names[missing] if names and missing is not None else missing)
        used[nxt] = True
        perm.append(nxt)
        at += len(blocks[nxt])
    return perm, at


def first_missing(ours, theirs):
    """The first structure, in TRANSCRIPT order, whose serialised value occurs nowhere in `theirs`: where the two provers
    part ways (everything after it hangs off a different challenger state and differs too)."""
    c = ours["commitments"]
    probes = [("commitments.main", _cap(c["main"]))]
    if c["permutation"] is not None:
        probes.append(("commitments.permutation", _cap(c["permutation"])))
    for i, t in enumerate(ours["lookup_terminals"]):
        if t is not None:
            probes.append(("global_lookup_data[%d]" % i, _fes(t)))
    probes.append(("commitments.quotient_chunks", _cap(c["quotient"])))
    for k, o in enumerate(ours["opened"]):
        for i, b in enumerate(opened_blocks(o)):
            if len(b) > 1:
                probes.append(("opened_values.instances[%d].%s" % (k, OPENED_FIELDS[i]), b))
    fb = fri_blocks(ours["opening_proof"])
    for i in (0, 1, 3, 4, 2):
        probes.append(("opening_proof." + FRI_FIELDS[i], fb[i]))
    probes.append(("degree_bits", _vec(ours["degree_bits"], _v)))
    for name, blk in probes:
        if theirs.find(blk) < 0:
            return name
    return None


def locate_layout(ours, theirs):
    """The proof_layout (batch[5] | fri[5] | opened[8]) under which the proof `ours` (decoded, made under the identity
    layout) serialises to the bytes `theirs`; or (None, diagnosis)."""
    miss = first_missing(ours, theirs)
    if miss is not None:
        return None, miss
    bb = batch_blocks(ours)
    # the nested orders first: inside `theirs` the opening_proof / the instances are contiguous whatever the outer order
    fb = fri_blocks(ours["opening_proof"])
    fri_perm = None
    for cand in itertools.permutations(range(5)):
        if theirs.find(b"".join(fb[i] for i in cand)) >= 0:
            fri_perm = list(cand)
            break
    if fri_perm is None:
        # which FRI field's VALUE is absent (transcript order: commits, commit PoW, final poly, query PoW, queries)
        for i in (0, 1, 3, 4, 2):
            if theirs.find(fb[i]) < 0:
                return None, "opening_proof." + FRI_FIELDS[i]
        return None, "opening_proof (every field occurs, but not adjacent in any order)"
    # eight fields per instance: anchor on the largest block of the richest instance and tile outwards (both directions,
    # with backtracking: one-byte blocks - absent options - match in many places); a tiling is kept if the WHOLE
    # opened_values vector, re-serialised in that order, occurs in the bytes
    rich = max(range(len(ours["opened"])), key=lambda k: sum(len(b) for b in opened_blocks(ours["opened"][k])))
    ob = opened_blocks(ours["opened"][rich])
    anchor = max(range(8), key=lambda k: len(ob[k]))

    def tilings(lo, hi, rem, left, right):
        if not rem:
            yield left[::-1] + [anchor] + right
            return
        for b in sorted(rem):
            n = len(ob[b])
            if theirs.startswith(ob[b], hi):
                yield from tilings(lo, hi + n, rem - {b}, left, right + [b])
            if lo >= n and theirs[lo - n:lo] == ob[b]:
                yield from tilings(lo - n, hi, rem - {b}, left + [b], right)

    def whole(perm):
        return _v(len(ours["opened"])) + b"".join(b"".join(opened_blocks(o)[i] for i in perm) for o in ours["opened"])

    opened_perm, seen = None, set()
    pos = theirs.find(ob[anchor])
    while pos >= 0 and opened_perm is None:
        for perm in tilings(pos, pos + len(ob[anchor]), frozenset(range(8)) - {anchor}, [], []):
            if tuple(perm) in seen:
                continue
            seen.add(tuple(perm))
            if theirs.find(whole(perm)) >= 0:
                opened_perm = perm
                break
        pos = theirs.find(ob[anchor], pos + 1)
    if opened_perm is None:
        for k, o in enumerate(ours["opened"]):
            for i, b in enumerate(opened_blocks(o)):
                if len(b) > 1 and theirs.find(b) < 0:
                    return None, "opened_values.instances[%d].%s" % (k, OPENED_FIELDS[i])
        return None, "opened_values (every field occurs, but the instances do not tile in one field order)"
    # outer order, with the nested blocks re-serialised in the orders just found
    bb[1] = whole(opened_perm)
    bb[2] = b"".join(fb[i] for i in fri_perm)
    perm, end = order_of(bb, theirs, 0, BATCH_FIELDS)
    if perm is None:
        return None, str(end)
    if end != len(theirs):
        return None, "%d trailing bytes after the last field" % (len(theirs) - end)
    return perm + fri_perm + opened_perm, None


def legal_schedules(log_heights, log_final, max_log_arity, limit=256):
    """Every FRI folding schedule that reaches each input height and the final height exactly with steps of at most
    max_log_arity bits (the verifier's conditions, recursion/src/pcs/fri/verifier.rs:587-781): the candidates when the
    reference's own schedule cannot be read off its bytes (unknown field order).  Small for fixture-sized proofs."""
    stops = sorted(set(h for h in log_heights if h > log_final), reverse=True) + [log_final]
    out = [[]]
    for hi, lo in zip(stops, stops[1:]):
        gap, parts = hi - lo, []

        def comp(rest, acc):
            if rest == 0:
                parts.append(acc)
                return
            for k in range(1, min(max_log_arity, rest) + 1):
                comp(rest - k, acc + [k])
        comp(gap, [])
        out = [o + q for o in out for q in parts]
        if len(out) > limit:
            return out[:limit]
    return out


# ---- the layer of a fixture on the oracle
def layer_of(orc, fx, prm, field, rc):
    if "circuit" in fx:        # a lowered circuit in this repo's op format (rust_npo_layer_*)
        c = fx["circuit"]
        circuit = cl.Circuit(c["witness_count"], np.array(c["ops"], dtype=np.uint32), c["ext"], c["public_rows"], c["private_rows"], c["rewrite"])
        pd = fx["inputs"]["private_data"]
        inputs = cl.Inputs(np.array(fx["inputs"]["public_values"], dtype=np.uint32).reshape(-1), (), [d["op_id"] for d in pd],
                           np.array([d["sibling"] for d in pd], dtype=np.uint32).reshape(-1))
        d = fx.get("ext_degree", 4)
    else:                      # the Fibonacci circuit (recursive_fibonacci.rs:315-337)
        d = fx.get("ext_degree", 4)
        circuit, inputs, fib = fib_lib.fibonacci_circuit(fx["n"], oracle_lib.MODULUS[field], ext_degree=d)
        assert fib == fx["fib"], "the Fibonacci value of the fixture"
    oc = cl.OracleCircuit(orc, circuit).preprocess(oracle_lib.MODULUS[field], d)
    oc.run(field, inputs, rc=rc)
    packing = dict(fx["packing"])
    packing.setdefault("horner_packed_steps", 2)
    if d != 4:
        packing["ext_degree"] = d
    return layer_lib.OracleLayer(orc, field, oc.workload_arrays(), prm, packing=packing, rc=rc)


def resolve(fx, orc=None, log=print):
    """-> dict(resolved=bool, config=..., diagnosis=...).  `fx`: a fixture dict in tools/rust_pin's schema."""
    orc = orc or oracle_lib.Oracle()
    field = FIELD_OF[fx["field"]]
    rc = np.array(fx["rc"], dtype=np.uint32)
    theirs = bytes.fromhex(fx["batch_proof_postcard_hex"])
    zk = bool(fx.get("zk"))
    if zk:
        return dict(resolved=False, diagnosis="a ZK proof is randomised: there are no bytes to reproduce (acceptance is what "
                                              "tests/test_rust_pins.py::test_rust_zk_proof_is_accepted checks)")
    notes = []
    if not np.array_equal(rc, oracle_lib.default_rc(field)):
        notes.append("round constants differ from the built-in table: pass the fixture's as p3r_config.poseidon2_rc "
                     "(and regenerate csrc/poseidon2_rc_default.inc from them)")
    # the reference's own folding schedule and proof-of-work witnesses, if its bytes decode under some layout guess: the
    # identity layout first (the common case); they only seed candidates, a wrong guess costs nothing
    guesses = []
    try:
        d = pc.decode(theirs)
        if d["_consumed"] == len(theirs):
            arities = [s["log_arity"] for s in d["opening_proof"]["query_proofs"][0]["commit_phase_openings"]]
            guesses.append((arities, d["opening_proof"]["commit_pow_witnesses"], d["opening_proof"]["query_pow_witness"]))
    except Exception:
        pass
    P = oracle_lib.MODULUS[field]
    # ... and, for when they do not decode (another field order), every legal schedule of a proof of this shape
    schedules = [g[0] for g in guesses]
    try:
        L0 = layer_of(orc, fx, layer_lib.params(**fx["fri"]), field, rc)
        lb = fx["fri"]["log_blowup"]
        heights = [db + lb for db in pc.decode(L0.prove())["degree_bits"]]
        for sch in legal_schedules(heights, fx["fri"]["log_final_poly_len"] + lb, fx["fri"]["max_log_arity"]):
            if sch not in schedules:
                schedules.append(sch)
    except RuntimeError:
        pass
    candidates = []
    for enc, unpacked in itertools.product((0, 1), (0, 1)):   # the rule first, then explicit schedules
        candidates.append(dict(enc=enc, unpacked=unpacked, arities=None, forced=None))
    for enc, unpacked in itertools.product((0, 1), (0, 1)):
        for arities in schedules:
            candidates.append(dict(enc=enc, unpacked=unpacked, arities=arities, forced=None))
    best = None
    for cand in candidates:
        kw = dict(fx["fri"], ext_choices=cand["unpacked"], fri_log_arities=cand["arities"])
        try:
            L = layer_of(orc, fx, layer_lib.params(**kw), field, rc)
            ours_bytes = L.prove(field_encoding=cand["enc"])
        except RuntimeError as e:
            log("  candidate %s: the oracle refuses it (%s)" % (cand, e))
            continue
        ours = pc.decode(ours_bytes)
        layout, why = locate_layout(ours, theirs)
        if layout is not None:
            rule = None
            if cand["arities"] is not None:   # is the explicit schedule just what the rule gives?
                L0 = layer_of(orc, fx, layer_lib.params(**dict(kw, fri_log_arities=None)), field, rc)
                if L0.prove(field_encoding=cand["enc"]) == ours_bytes:
                    rule = "rule"
            cfg = dict(field=field, poseidon2_rc="fixture" if notes else "built-in", canonical_field_encoding=bool(cand["enc"]),
                       ext_choices=cand["unpacked"], fri_log_arities=None if (cand["arities"] is None or rule) else cand["arities"],
                       proof_layout=None if layout == list(range(5)) + list(range(5)) + list(range(8)) else layout, **fx["fri"])
            return dict(resolved=True, config=cfg, notes=notes)
        # rank the near misses by how far into the transcript they got
        depth = ["commitments.main", "commitments.permutation", "global_lookup_data", "commitments.quotient_chunks", "opened_values",
                 "opening_proof.commit_phase_commits", "opening_proof.commit_pow_witnesses",
                 "opening_proof.final_poly", "opening_proof.query_pow_witness", "opening_proof.query_proofs"]
        score = next((i for i, name in enumerate(depth) if why and why.startswith(name)), len(depth))
        if best is None or score > best[0]:
            best = (score, cand, why, ours)
    # no candidate reproduces the bytes.  One more question before giving up: would the reference's OWN proof-of-work
    # witnesses (any valid witness is a valid proof; upstream searches in parallel) make the rest agree?
    diagnosis = "no combination of switches reproduces the reference's bytes"
    if best is not None:
        score, cand, why, ours = best
        diagnosis += "; closest: encoding=%s, lookup_unpacked=%d, fri_log_arities=%s - first structure that differs: %s" % (
            "canonical" if cand["enc"] else "montgomery", cand["unpacked"], cand["arities"], why)
        if why and ("pow_witness" in why or "query_proofs" in why or "final_poly" in why) and guesses:
            arities, cw, qw = guesses[0]
            conv = (lambda w: w) if cand["enc"] else (lambda w: (w * pow(1 << 32, -1, P)) % P)
            forced = [conv(w) for w, bits in zip(cw, itertools.repeat(fx["fri"].get("commit_pow_bits", 0))) if bits]
            if fx["fri"].get("query_pow_bits", 0):
                forced.append(conv(qw))
            try:
                kw = dict(fx["fri"], ext_choices=cand["unpacked"], fri_log_arities=cand["arities"], forced_pow=forced)
                L = layer_of(orc, fx, layer_lib.params(**kw), field, rc)
                layout, why2 = locate_layout(pc.decode(L.prove(field_encoding=cand["enc"])), theirs)
                if layout is not None:
                    diagnosis += ("; WITH the reference's proof-of-work witnesses forced the bytes agree: the only difference is "
                                  "the PoW witness rule (this library returns the smallest witness, the reference another valid "
                                  "one) - proofs verify both ways, byte equality needs the reference's search order")
            except RuntimeError as e:
                diagnosis += "; forcing the reference's proof-of-work witnesses fails: %s" % e
    return dict(resolved=False, diagnosis=diagnosis, notes=notes)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        print(__doc__)
        sys.exit(2)
    rc_all = 0
    for path in sys.argv[1:]:
        with open(path) as fh:
            fx = json.load(fh)
        print(path)
        out = resolve(fx)
        print(json.dumps(out, indent=1))
        rc_all |= 0 if out["resolved"] else 1
    sys.exit(rc_all)
