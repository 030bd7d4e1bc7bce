"""CPU: the Rust FFI shim of INTEGRATION.md is source that no compiler in this image has ever seen (no cargo), so its
`#[repr(C)]` blocks are checked against include/p3r.h at the text level: same structs, same fields in the same order,
matching types, and every `extern "C"` function it declares is one the header declares with the same number of
parameters.  A field added to the C ABI without the shim following (or the reverse) fails here."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_c_comments(t):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", t, flags=re.S))


def c_structs(header):
    """name -> [(field, normalised type)] for every `typedef struct NAME { ... } NAME;` with a body."""
    t = strip_c_comments(header)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", t, flags=re.S):
        name, body = m.group(1), m.group(2)
        # anonymous nested struct members: `struct { char op_type[64]; uint32_t lanes; } npo_lanes[N];`
        body = re.sub(r"struct\s*\{[^{}]*\}\s*(\w+)\s*(\[[^\]]*\])?\s*;", lambda a: "__anon__ %s%s;" % (a.group(1), a.group(2) or ""), body)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"(const\s+)?(\w+)\s*(\*)?\s*(.*)$", decl)   # every base type of the header is one word
            assert mm, (name, decl)
            const, base, ptr, rest = mm.group(1), mm.group(2), mm.group(3), mm.group(4)
            depth, cur, names = 0, "", []
            for ch in rest:
                depth += ch == "["
                depth -= ch == "]"
                if ch == "," and depth == 0:
                    names.append(cur)
                    cur = ""
                else:
                    cur += ch
            names.append(cur)
            for nm in [x.strip() for x in names if x.strip()]:
                am = re.match(r"(\w+)\s*(?:\[(.*)\])?$", nm)
                assert am, (name, decl, nm)
                fields.append((am.group(1), norm_c(base, bool(ptr), bool(const), am.group(2))))
        out[name] = fields
    return out


C_SCALARS = {"uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "size_t": "usize", "uint8_t": "u8", "int": "i32", "char": "c_char",
             "double": "f64"}


def norm_dim(d):
    return re.sub(r"\s+", "", d)


def norm_c(base, ptr, const, dim):
    t = C_SCALARS.get(base, base)
    if ptr:
        t = ("*const " if const else "*mut ") + t
    if dim is not None:
        t = "[%s; %s]" % (t, norm_dim(dim))
    return t


def rust_structs(md):
    out = {}
    md = strip_c_comments(md)   # comments may hold braces (type names like Poseidon2{Koala,Baby}Bear<32>)
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{(.*?)\}", md, flags=re.S):
        body = m.group(2)
        fields = []
        # split on commas that are not inside brackets
        depth, cur, parts = 0, "", []
        for ch in body:
            if ch in "[(<":
                depth += 1
            elif ch in "])>":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        parts.append(cur)
        for part in parts:
            part = " ".join(part.split())
            if not part:
                continue
            fm = re.match(r"pub (\w+)\s*:\s*(.+)$", part)
            assert fm, (m.group(1), part)
            ty = fm.group(2).strip()
            am = re.match(r"\[(.+);\s*(.+)\]$", ty)
            if am:
                ty = "[%s; %s]" % (am.group(1).strip(), norm_dim(am.group(2)))
            fields.append((fm.group(1), ty))
        out[m.group(1)] = fields
    return out


def test_repr_c_blocks_follow_the_header():
    header = open(os.path.join(ROOT, "include", "p3r.h")).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    cs, rs = c_structs(header), rust_structs(md)
    # the shim must at least cover the structs of the drop-in boundary
    for need in ("p3r_config", "p3r_p2_rows", "p3r_layer_desc_counts", "p3r_layer_desc", "p3r_traces", "p3r_op", "p3r_circuit_desc",
                 "p3r_circuit_inputs", "p3r_npo_table_entry", "p3r_batch_stark_meta"):
        assert need in rs, "INTEGRATION.md has no #[repr(C)] block for %s" % need
    anon = {"p3r_npo_lanes_entry"}   # stands for an anonymous member struct of the header
    for name, rf in rs.items():
        if name in anon:
            continue
        assert name in cs, "#[repr(C)] struct %s has no counterpart in include/p3r.h" % name
        cf = cs[name]
        assert [f for f, _ in rf] == [f for f, _ in cf], (name, [f for f, _ in rf], [f for f, _ in cf])
        for (fn, rt), (_, ct) in zip(rf, cf):
            if ct.startswith("__anon__") or "__anon__" in ct:
                assert "p3r_npo_lanes_entry" in rt, (name, fn, rt, ct)
                assert rt.split(";")[1:] == ct.split(";")[1:], (name, fn, rt, ct)   # same array length
                continue
            assert rt == ct, "%s.%s: Rust `%s` against C `%s`" % (name, fn, rt, ct)
    # the anonymous member itself
    assert rs["p3r_npo_lanes_entry"] == [("op_type", "[c_char; 64]"), ("lanes", "u32")]
    assert "struct { char op_type[64]; uint32_t lanes; } npo_lanes[P3R_META_MAX_NPO];" in header
    # constants the array lengths refer to
    for const in ("P3R_META_MAX_NPO", "P3R_META_MAX_INSTANCES", "P3R_META_MAX_CAP"):
        cv = re.search(r"#define\s+%s\s+(\d+)" % const, header).group(1)
        rv = re.search(r"pub const %s: usize = (\d+);" % const, md).group(1)
        assert cv == rv, (const, cv, rv)


def test_extern_functions_of_the_shim_exist_with_the_same_arity():
    header = strip_c_comments(open(os.path.join(ROOT, "include", "p3r.h")).read())
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    cfn = {}
    for m in re.finditer(r"\b(p3r_\w+)\s*\(([^;{}]*?)\)\s*;", header, flags=re.S):
        args = m.group(2).strip()
        cfn[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    seen = 0
    for m in re.finditer(r"pub fn (p3r_\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", md, flags=re.S):
        name, args = m.group(1), strip_c_comments(m.group(2))
        n = len([a for a in args.split(",") if a.strip()])
        assert name in cfn, "INTEGRATION.md declares %s, include/p3r.h does not" % name
        assert cfn[name] == n, "%s: %d parameters in the shim, %d in the header" % (name, n, cfn[name])
        seen += 1
    assert seen >= 8


def test_shim_literals_name_the_headers_abi_version_and_every_config_field():
    """The `p3r_config { .. }` literals of the shim text: `abi_version: N` is p3r.h's P3R_ABI_VERSION, and a literal
    that is not built with `..` names every field of the struct (a Rust struct literal with a missing field does not
    compile - the round-3 literal silently lacked the ABI-6 fields)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    hdr = open(os.path.join(root, "include", "p3r.h")).read()
    abi = int(re.search(r"#define\s+P3R_ABI_VERSION\s+(\d+)", hdr).group(1))
    for m in re.finditer(r"abi_version:\s*(\d+)", md):
        assert int(m.group(1)) == abi, "INTEGRATION.md names ABI version %s, p3r.h has %d" % (m.group(1), abi)
    body = re.search(r"typedef struct p3r_config \{(.*?)\} p3r_config;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [re.sub(r"\[.*\]$", "", re.split(r"[\s\*]+", d.strip())[-1]) for d in body.split(";") if d.strip()]   # (zk_key[8] -> zk_key)
    lits = re.findall(r"let cfg = p3r_config \{(.*?)\};", md, re.S)
    assert lits, "no p3r_config literal found in INTEGRATION.md"
    for lit in lits:
        lit = re.sub(r"//[^\n]*", "", lit)
        if ".." in lit:
            continue
        named = set(re.findall(r"(\w+)\s*:", lit)) | set(re.findall(r"[{,]\s*(\w+)\s*(?=[,}])", "{" + lit + "}"))
        missing = [f for f in fields if f not in named]
        assert not missing, "p3r_config literal in INTEGRATION.md lacks %s" % missing
