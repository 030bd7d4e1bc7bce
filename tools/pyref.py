"""Third, pure-Python (big-int) restatement of the small primitives, used only to GENERATE the
self-made golden fixtures under tests/golden/ (tools/gen_golden.py) and never at run time.

Independent of both oracle/ (C++) and the HIP product: plain Python ints, matrices applied as
explicit matrix products, naive O(n^2) DFT.  Same reference anchors as oracle/hash.hpp.
PARITY UNPINNED: no upstream golden vectors exist.
"""

FIELDS = {
    "koala_bear": dict(id=0, p=0x7F000001, gen=3, two_adicity=24, w=3, sbox=3, regs=0, partial=20),
    "baby_bear": dict(id=1, p=0x78000001, gen=31, two_adicity=27, w=11, sbox=7, regs=1, partial=13),
}

M4 = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]


def inv(x, p):
    return pow(x, p - 2, p)


def internal_diag(f):
    p = f["p"]
    h = lambda k: inv(pow(2, k, p), p)
    if f["id"] == 0:
        d = [-2, 1, 2, h(1), 3, 4, -h(1), -3, -4, h(8), h(3), h(24), -h(8), -h(3), -h(4), -h(24)]
    else:
        d = [-2, 1, 2, h(1), 3, 4, -h(1), -3, -4, h(8), h(2), h(3), h(27), -h(8), -h(4), -h(27)]
    return [x % p for x in d]


def external(s, p):
    out = []
    for i in range(16):
        acc = 0
        for j in range(16):
            acc += M4[i % 4][j % 4] * (2 if i // 4 == j // 4 else 1) * s[j]
        out.append(acc % p)
    return out


def internal(s, f):
    p = f["p"]
    d = internal_diag(f)
    tot = sum(s) % p
    return [(s[i] * d[i] + tot) % p for i in range(16)]


def permute(state, rc, f, cells=None):
    p, deg, regs, partial = f["p"], f["sbox"], f["regs"], f["partial"]
    s = list(state)
    if cells is not None:
        cells.extend(s)
    s = external(s, p)
    k = 0

    def full(s, k):
        t = []
        for i in range(16):
            x = (s[i] + rc[k + i]) % p
            if regs == 1 and cells is not None:
                cells.append(pow(x, 3, p))
            t.append(pow(x, deg, p))
        t = external(t, p)
        if cells is not None:
            cells.extend(t)
        return t, k + 16

    for _ in range(4):
        s, k = full(s, k)
    for _ in range(partial):
        x = (s[0] + rc[k]) % p
        k += 1
        if regs == 1 and cells is not None:
            cells.append(pow(x, 3, p))
        s[0] = pow(x, deg, p)
        if cells is not None:
            cells.append(s[0])
        s = internal(s, f)
    for _ in range(4):
        s, k = full(s, k)
    return s


def sponge_hash(vals, rc, f):
    s = [0] * 16
    i = 0
    while i < len(vals):
        chunk = vals[i:i + 8]
        for j, v in enumerate(chunk):
            s[j] = v
        s = permute(s, rc, f)
        i += 8
    return s[:8]


def compress(l, r, rc, f):
    return permute(list(l) + list(r), rc, f)[:8]


class Challenger:
    def __init__(self, rc, f):
        self.rc, self.f = rc, f
        self.state = [0] * 16
        self.inb, self.outb = [], []

    def duplex(self):
        n = len(self.inb)
        for i, v in enumerate(self.inb):
            self.state[i] = v
        self.inb = []
        if n > 0:
            for i in range(n, 8):
                self.state[i] = 0
            self.state[8] = (self.state[8] + n) % self.f["p"]
        self.state = permute(self.state, self.rc, self.f)
        self.outb = list(self.state[:8])

    def observe(self, v):
        self.outb = []
        self.inb.append(v)
        if len(self.inb) == 8:
            self.duplex()

    def sample(self):
        if self.inb or not self.outb:
            self.duplex()
        return self.outb.pop()

    def sample_bits(self, b):
        return self.sample() & ((1 << b) - 1)


def ext_mul(a, b, f):
    p, w = f["p"], f["w"]
    r = [0] * 4
    for i in range(4):
        for j in range(4):
            t = a[i] * b[j]
            if i + j >= 4:
                r[i + j - 4] += w * t
            else:
                r[i + j] += t
    return [x % p for x in r]


def ext_pow(a, e, f):
    r = [1, 0, 0, 0]
    b = list(a)
    while e:
        if e & 1:
            r = ext_mul(r, b, f)
        b = ext_mul(b, b, f)
        e >>= 1
    return r


def ext_inv(a, f):
    return ext_pow(a, f["p"] ** 4 - 2, f)


def two_adic_generator(bits, f):
    return pow(f["gen"], (f["p"] - 1) >> bits, f["p"])


def bitrev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def coset_lde_bitrev(col, added_bits, shift, f):
    """Naive: interpolate `col` (evaluations over the subgroup, natural order) and evaluate on
    shift*<w_m>; return rows in bit-reversed order."""
    p = f["p"]
    h = len(col)
    lh = h.bit_length() - 1
    g = two_adic_generator(lh, f)
    ginv = inv(g, p)
    hinv = inv(h, p)
    coeffs = [sum(col[n] * pow(ginv, n * k, p) for n in range(h)) * hinv % p for k in range(h)]
    m = h << added_bits
    lm = lh + added_bits
    wm = two_adic_generator(lm, f)
    out = []
    for i in range(m):
        x = shift * pow(wm, bitrev(i, lm), p) % p
        out.append(sum(c * pow(x, k, p) for k, c in enumerate(coeffs)) % p)
    return out
