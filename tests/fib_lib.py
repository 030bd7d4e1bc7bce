"""The base circuit of the reference's recursive_fibonacci example
(recursion/examples/recursive_fibonacci.rs:315-337): one public input `expected_result`, the
constants F(0) = 0 and F(1) = 1, n - 1 additions, and `connect(b, expected_result)` - which merges
the last sum with the public witness, so the last Add writes onto an already-defined witness.
Embedded in the degree-4 extension (values (v, 0, 0, 0)) for the D = 4 tables."""
import numpy as np

import circuit_lib as cl

N = cl.NO_W


def fibonacci_circuit(n, modulus, ext_degree=4):
    """ext_degree = 1: the circuit as the example builds it, `CircuitBuilder<F>` (constants and inputs are single
    base-field elements)."""
    d = ext_degree
    ops, ext = [], []
    ops.append([cl.OP_CONST, 0, 0, N, 0, N, 0, d]); ext += [0] * d                 # w0 = F(0) = ExprId::ZERO
    ops.append([cl.OP_PUBLIC, 0, 0, N, 1, 0, 0, 0])                               # w1 = expected_result
    ops.append([cl.OP_CONST, 0, 0, N, 2, N, d, d]); ext += [1] + [0] * (d - 1)     # w2 = F(1)
    a, b, nxt = 0, 2, 3
    fa, fb = 0, 1
    for i in range(2, n + 1):
        out = 1 if i == n else nxt          # connect(b, expected_result)
        ops.append([cl.OP_ADD, a, b, N, out, N, 0, 0])
        a, b = b, out
        fa, fb = fb, (fa + fb) % modulus
        nxt += 1
    witness_count = nxt - 1 if n >= 2 else 3
    return cl.Circuit(witness_count, ops, ext, public_rows=[1]), cl.Inputs(public_values=[fb] + [0] * (d - 1)), fb
