#!/usr/bin/env python3
"""One-letter-per-instruction stream of the basic blocks of a kernel that hold the most MFMAs (from a `hipcc -S` listing): shows whether the
scheduler interleaved what the source asked for.   python tools/isa_stream.py /tmp/x.s <mangled-name-regex> [n_blocks]
M mfma, e v_exp, a v_pk_add, c v_cvt_pk, r ds_read, w ds_write, G global/buffer load, S global store, | s_waitcnt, B s_barrier, n s_nop, v other VALU, s other SALU"""
import re
import sys


def main(path, pat, n_blocks=2):
    s = open(path).read()
    m = re.search(r'^(' + pat + r'):.*\n', s, re.M)
    if not m:
        sys.exit(f"no function matching {pat}")
    body = s[m.end():s.index('.Lfunc_end', m.end())].split('\n')
    blocks, cur = [], []
    for ln in body:
        if re.match(r'^\.LBB', ln):
            blocks.append(cur)
            cur = [ln]
        else:
            cur.append(ln)
    blocks.append(cur)
    ops = lambda b: [l.split()[0] for l in b if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]   # noqa: E731
    code = lambda x: ('M' if 'mfma' in x else 'e' if x.startswith('v_exp') else 'a' if x.startswith('v_pk_add') else 'c' if x.startswith('v_cvt_pk')   # noqa: E731
                      else 'r' if x.startswith('ds_read') else 'w' if x.startswith('ds_write') else 'G' if x.startswith(('global_load', 'buffer_load'))
                      else 'S' if x.startswith(('global_store', 'buffer_store')) else '|' if x.startswith('s_waitcnt') else 'B' if x.startswith('s_barrier')
                      else 'n' if x.startswith('s_nop') else 'v' if x.startswith('v_') else 's' if x.startswith('s_') else '?')
    for b in sorted(blocks, key=lambda b: -sum('mfma' in o for o in ops(b)))[:n_blocks]:
        o = ops(b)
        print(b[0] if b and b[0].startswith('.LBB') else '(entry)', len(o), 'instructions,', sum('mfma' in x for x in o), 'mfma,', sum(x.startswith('v_exp') for x in o), 'exp')
        print(''.join(code(x) for x in o))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2)
