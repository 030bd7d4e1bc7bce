"""Static per-kernel instruction histogram of a gfx950 assembly file (hipcc -save-temps).

usage: python tools/isa_histogram.py file.s [kernel-name-substring ...]
Loops are counted once (static count); the dynamic count per wave comes from the SQ_INSTS_VALU /
SQ_WAVES counters (tools/profile_round.sh).  Prints registers, code bytes and the opcode mix.
"""
import collections
import re
import sys


def kernels(text):
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', text, re.S | re.M):
        yield m.group(1), m.group(2)


def meta(text):
    out = {}
    for blk in re.finditer(r'\.name:\s+(\S+)(.*?)(?=\n\s+- \.|\n\.\.\.|\Z)', text, re.S):
        d = {}
        for key in ('vgpr_count', 'sgpr_count', 'vgpr_spill_count'):
            mm = re.search(r'\.%s:\s+(\d+)' % key, blk.group(0))
            if mm:
                d[key] = int(mm.group(1))
        out[blk.group(1)] = d
    return out


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2:]
    md = meta(text)
    for name, body in kernels(text):
        if want and not any(w in name for w in want):
            continue
        c = collections.Counter()
        for line in body.split('\n'):
            line = line.strip()
            if not line or line.startswith(('.', ';')) or line.endswith(':'):
                continue
            c[line.split()[0]] += 1
        tot = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        print('%s: %d instructions (%d VALU) %s' % (name, tot, valu, md.get(name, {})))
        for k, v in c.most_common(24):
            print('    %-28s %6d' % (k, v))


if __name__ == '__main__':
    main()
