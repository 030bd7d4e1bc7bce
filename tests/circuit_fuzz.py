"""Random circuits for the runner / preprocessing parity tests: arbitrary dependency structure
(random DAGs, Horner chains that break in the middle, sponge / Merkle chains fed by recent outputs,
backward ops, repeated ops that re-write the same value, hints, private inputs, rewrites).  Values
are never computed here: the oracle's sequential runner is the ground truth."""
import random

import numpy as np

import circuit_lib as cl

N = cl.NO_W


def random_circuit(seed, n_ops=300, modulus=0x7F000001, w32=False):
    """w32: a fifth of the permutation chains are chains of the WIDTH-32 table (cl.OP_P2W: sponge and Merkle rows freely
    mixed - a Merkle row continues whatever width-32 row came last, a sponge row the last SPONGE row, as the executor's two
    chain states have it; named limbs anywhere, bits from the constants, three sibling digests as private data on most
    Merkle rows).  Off by default: the draws of the existing seeds do not change."""
    rng = random.Random(seed)
    ops, ext = [], []
    ws = []            # witnesses that are set when the next op runs
    base_ws = []       # ... known to hold base-field values (coefficients 1..3 zero)
    n_w = 0
    public_rows, public_values, private_rows, private_values = [], [], [], []
    pd_ids, pd_sibs = [], []
    pdw_ids, pdw_sibs = [], []
    have_w = {False: False, True: False}   # width-32 op type: normal / Merkle chain state exists
    npo_id = 0
    rewrite = []

    def fresh():
        nonlocal n_w
        n_w += 1
        return n_w - 1

    def rnd_ef(base=False):
        v = [rng.randrange(modulus)] + ([0, 0, 0] if base else [rng.randrange(modulus) for _ in range(3)])
        return v

    def op(kind, a=0, b=0, c=N, out=0, aux=N, e=()):
        ops.append([kind, a, b, c, out, aux, len(ext), len(e)])
        ext.extend(e)

    def const(v):
        w = fresh()
        op(cl.OP_CONST, out=w, e=v)
        ws.append(w)
        if not any(v[1:]):
            base_ws.append(w)
        return w

    zero, one = const([0, 0, 0, 0]), const([1, 0, 0, 0])
    nonzero = [one] + [const(rnd_ef()) for _ in range(3)]
    for _ in range(6):
        const(rnd_ef(base=rng.random() < 0.6))
    for i in range(5):
        w = fresh()
        base = rng.random() < 0.5
        v = rnd_ef(base)
        public_rows.append(w); public_values.append(v)
        op(cl.OP_PUBLIC, out=w, aux=i)
        ws.append(w)
        if base:
            base_ws.append(w)
    unclaimed = []
    for _ in range(4):
        w = fresh()
        private_rows.append(w); private_values.append(rnd_ef())
        unclaimed.append(w)
    pick = lambda: rng.choice(ws)
    last_alu = None
    have_chain = {False: False, True: False}   # sponge / merkle chain started
    while len(ops) < n_ops:
        k = rng.random()
        if unclaimed and k < 0.06:
            # a private input is set from the start; its first ALU use claims it on the bus
            pw = unclaimed.pop()
            o = fresh()
            if rng.random() < 0.5:
                op(cl.OP_ADD, a=pw, b=pick(), out=o)
            else:
                op(cl.OP_MULADD, a=pick(), b=pick(), c=pw, out=o)
            ws += [pw, o]
        elif k < 0.30:
            kind = rng.choice([cl.OP_ADD, cl.OP_MUL])
            a, b, o = pick(), pick(), fresh()
            op(kind, a=a, b=b, out=o)
            ws.append(o)
            last_alu = (kind, a, b, o)
        elif k < 0.34 and last_alu:
            kind, a, b, o = last_alu                     # the same op again: `out` is re-written with an equal value
            op(kind, a=a, b=b, out=o)
        elif k < 0.42:
            kind = rng.choice([cl.OP_ADD, cl.OP_MUL])    # backward: a and out known, b solved for
            a = rng.choice(nonzero) if kind == cl.OP_MUL else pick()
            b = fresh()
            op(kind, a=a, b=b, out=pick())
            ws.append(b)
        elif k < 0.47:
            a, o = rng.choice([zero, one]), fresh()
            op(cl.OP_BOOL, a=a, b=zero, c=a, out=o)
            ws.append(o)
        elif k < 0.57:
            io = fresh() if rng.random() < 0.3 else N
            o = fresh()
            op(cl.OP_MULADD, a=pick(), b=pick(), c=pick() if rng.random() < 0.8 else N, out=o, aux=io)
            ws.append(o)
            if io != N:
                ws.append(io)
        elif k < 0.70:
            b, acc = pick(), rng.choice([zero, pick()])
            for _ in range(rng.randint(1, 14)):        # operands may be outputs of this very chain
                o = fresh()
                op(cl.OP_HORNER, a=pick(), b=b, c=pick(), out=o, aux=acc)
                ws.append(o)
                acc = o
                if rng.random() < 0.1:
                    b = pick()                           # a different multiplier splits the chain
        elif k < 0.74:
            outs = [fresh() for _ in range(4)]
            op(cl.OP_HINT_EXT, a=pick(), e=outs)
            ws += outs; base_ws += outs
        elif k < 0.77:
            outs = [fresh() for _ in range(rng.randint(1, 40))]
            op(cl.OP_HINT_BIN, a=pick(), e=outs)
            ws += outs; base_ws += outs
        elif k < 0.82:
            o = fresh()
            op(cl.OP_RECOMPOSE, a=npo_id, out=o, e=[rng.choice(base_ws) for _ in range(4)])
            npo_id += 1
            ws.append(o)
        elif w32 and k < 0.86:
            for j in range(rng.randint(1, 10)):
                merkle = rng.random() < 0.6
                new_start = not have_w[merkle] or rng.random() < 0.15
                e = [pick() if rng.random() < (0.5 if new_start else 0.2) else N for _ in range(8)]
                e.append(N)                                                             # no mmcs_index_sum on this table
                e += [rng.choice([zero, one]), rng.choice([zero, one])] if merkle else [N, N]
                n_out = rng.choice([6, 8])
                outs = [fresh() if rng.random() < 0.5 else N for _ in range(n_out)]
                op(cl.OP_P2W, a=npo_id, aux=(1 if new_start else 0) | (2 if merkle else 0), e=e + [n_out] + outs)
                if merkle and rng.random() < 0.8:
                    pdw_ids.append(npo_id)
                    pdw_sibs.append([rng.randrange(modulus) for _ in range(24)])
                npo_id += 1
                have_w[merkle] = True
                if not merkle:
                    have_w[True] = True        # a sponge row seeds the Merkle state (executor.rs:462-491)
                ws += [o for o in outs if o != N]
        else:
            merkle = rng.random() < 0.5
            for j in range(rng.randint(1, 9)):
                new_start = j == 0 or not have_chain[merkle]
                e = [pick() if rng.random() < (0.6 if new_start else 0.25) else N for _ in range(4)]
                e.append(rng.choice(base_ws) if merkle and rng.random() < 0.3 else N)   # mmcs_index_sum
                e.append(rng.choice([zero, one]) if merkle else N)                       # mmcs_bit
                n_out = rng.choice([2, 4])
                outs = [fresh() if rng.random() < 0.6 else N for _ in range(n_out)]
                op(cl.OP_P2, a=npo_id, aux=(1 if new_start else 0) | (2 if merkle else 0), e=e + [n_out] + outs)
                if merkle and rng.random() < 0.8:
                    pd_ids.append(npo_id)
                    pd_sibs.append([rng.randrange(modulus) for _ in range(8)])
                npo_id += 1
                have_chain[merkle] = True
                ws += [o for o in outs if o != N]
    for pw in unclaimed:
        o = fresh()
        op(cl.OP_ADD, a=pw, b=pick(), out=o)
    for _ in range(3):
        rewrite += [fresh(), pick()]
    circuit = cl.Circuit(n_w, ops, ext, public_rows, private_rows, rewrite)
    inputs = cl.Inputs(np.array(public_values, np.uint32), np.array(private_values, np.uint32), pd_ids,
                       np.array(pd_sibs, np.uint32) if pd_sibs else (), pdw_ids, np.array(pdw_sibs, np.uint32) if pdw_sibs else ())
    return circuit, inputs
