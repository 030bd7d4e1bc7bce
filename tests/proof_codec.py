"""postcard codec of the inner `BatchProof` for TESTS: bytes <-> a nested Python structure, so that negative cases
and the structure-aware mutator (tests/san/) can edit a proof field by field and re-serialise it.  Field elements
stay the varint words the proof holds (Montgomery or canonical: the codec does not care).  Identity field order
(recursion/src/types/proof.rs:403-409,452-457,527-534,585-589, pcs/fri/targets.rs:104-110); `zk`: the opening
proof is HidingFriPcs's tuple (OpenedValues<Challenge>, FriProof) (pcs/fri/targets.rs:1007-1057)."""


class Reader:
    def __init__(self, b):
        self.b, self.i = b, 0

    def byte(self):
        v = self.b[self.i]
        self.i += 1
        return v

    def varint(self):
        v = shift = 0
        while True:
            x = self.byte()
            v |= (x & 0x7F) << shift
            if not x & 0x80:
                return v
            shift += 7

    def vec(self, item):
        return [item() for _ in range(self.varint())]

    def opt(self, item):
        return item() if self.byte() else None


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def decode(data, dc=4, zk=False, salted=False):
    """`salted`: the MMCSs are hiding ones - every MMCS opening proof is the tuple (salts: Vec<Vec<F>>, siblings)
    (SaltedMmcsProof, recursion/src/pcs/mmcs.rs:763-768); decoded as `salts` next to `opening_proof`."""
    r = Reader(bytes(data))
    fe = r.varint
    ef = lambda: [fe() for _ in range(dc)]                 # noqa: E731
    vec_ef = lambda: r.vec(ef)                             # noqa: E731
    digest = lambda: [fe() for _ in range(8)]              # noqa: E731
    cap = lambda: r.vec(digest)                            # noqa: E731
    p = {}
    p["commitments"] = dict(main=cap(), permutation=r.opt(cap), quotient=cap(), random=r.opt(cap))
    p["opened"] = r.vec(lambda: dict(trace_local=vec_ef(), trace_next=r.opt(vec_ef), preprocessed_local=r.opt(vec_ef),
                                     preprocessed_next=r.opt(vec_ef), quotient_chunks=r.vec(vec_ef), random=r.opt(vec_ef),
                                     permutation_local=vec_ef(), permutation_next=vec_ef()))
    fri = {}
    if zk:
        fri["random_opened_values"] = r.vec(lambda: r.vec(lambda: r.vec(vec_ef)))   # rounds -> matrices -> points -> values
    fri["commit_phase_commits"] = r.vec(cap)
    fri["commit_pow_witnesses"] = r.vec(fe)
    fri["query_proofs"] = r.vec(lambda: dict(
        input_proof=r.vec(lambda: dict(opened_values=r.vec(lambda: r.vec(fe)),
                                       **(dict(salts=r.vec(lambda: r.vec(fe))) if salted else {}), opening_proof=r.vec(digest))),
        commit_phase_openings=r.vec(lambda: dict(log_arity=r.byte(), sibling_values=vec_ef(),
                                                 **(dict(salts=r.vec(lambda: r.vec(fe))) if salted else {}),
                                                 opening_proof=r.vec(digest)))))
    fri["final_poly"] = vec_ef()
    fri["query_pow_witness"] = fe()
    p["opening_proof"] = fri
    p["lookup_terminals"] = r.vec(lambda: r.opt(ef))
    p["degree_bits"] = r.vec(r.varint)
    p["_consumed"] = r.i
    return p


def encode(p, zk=None):
    out = bytearray()
    V = lambda v: out.extend(_varint(v))                   # noqa: E731

    def fes(xs):
        for x in xs:
            V(x)

    def vec(xs, item):
        V(len(xs))
        for x in xs:
            item(x)

    def opt(x, item):
        if x is None:
            out.append(0)
        else:
            out.append(1)
            item(x)

    vec_ef = lambda v: vec(v, fes)                         # noqa: E731
    cap = lambda c: vec(c, fes)                            # noqa: E731
    c = p["commitments"]
    cap(c["main"]); opt(c["permutation"], cap); cap(c["quotient"]); opt(c["random"], cap)

    def inst(o):
        vec_ef(o["trace_local"]); opt(o["trace_next"], vec_ef); opt(o["preprocessed_local"], vec_ef)
        opt(o["preprocessed_next"], vec_ef); vec(o["quotient_chunks"], vec_ef); opt(o["random"], vec_ef)
        vec_ef(o["permutation_local"]); vec_ef(o["permutation_next"])
    vec(p["opened"], inst)
    f = p["opening_proof"]
    if zk if zk is not None else "random_opened_values" in f:
        vec(f["random_opened_values"], lambda rd: vec(rd, lambda m: vec(m, vec_ef)))
    vec(f["commit_phase_commits"], cap)
    vec(f["commit_pow_witnesses"], V)

    def query(q):
        def batch(b):
            vec(b["opened_values"], lambda row: vec(row, V))
            if "salts" in b:
                vec(b["salts"], lambda row: vec(row, V))
            vec(b["opening_proof"], fes)
        vec(q["input_proof"], batch)

        def step(s):
            out.append(s["log_arity"])
            vec_ef(s["sibling_values"])
            if "salts" in s:
                vec(s["salts"], lambda row: vec(row, V))
            vec(s["opening_proof"], fes)
        vec(q["commit_phase_openings"], step)
    vec(f["query_proofs"], query)
    vec_ef(f["final_poly"])
    V(f["query_pow_witness"])
    vec(p["lookup_terminals"], lambda t: opt(t, fes))
    vec(p["degree_bits"], V)
    return bytes(out)
