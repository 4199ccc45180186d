#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing (diagnostics for the MFMA / VALU balance of the hot loops).

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only x.hip -o x.s
    python tools/asm_blocks.py x.s <substring of the mangled kernel name> [min MFMAs per block]
"""
import re
import sys
from collections import Counter


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    s = open(path).read()
    for m in re.finditer(r"^(_Z\w*" + re.escape(key) + r"\w*):\s*;", s, re.M):
        name = m.group(1)
        body = s[m.end():s.index(".Lfunc_end", m.end())]
        blocks, cur, label = [], [], "entry"
        for line in body.split("\n"):
            line = line.strip()
            if re.match(r"^\.LBB\d+_\d+:", line):
                blocks.append((label, cur))
                cur, label = [], line.split(":")[0]
            elif line and not line.startswith(";") and not line.startswith("."):
                cur.append(line.split()[0])
        blocks.append((label, cur))
        print(name)
        for lab, b in blocks:
            c = Counter("v_mfma" if x.startswith("v_mfma") else x for x in b)
            nexp = sum(v for k, v in c.items() if k.startswith("v_exp"))
            if c["v_mfma"] < min_mfma and not nexp:
                continue
            valu = sum(v for k, v in c.items() if k.startswith("v_") and k != "v_mfma")
            pick = lambda *pre: sum(v for k, v in c.items() if any(p in k for p in pre))      # noqa: E731
            print(f"  {lab:10s} insts {len(b):4d}  mfma {c['v_mfma']:3d}  valu {valu:4d} (exp {nexp}, sub/add {pick('v_sub_f32', 'v_add_f32', 'v_fma_f32', 'v_mul_f32')}, pk {pick('v_pk_')}, "
                  f"cvt {pick('cvt')}, max {pick('max')}, mov {pick('v_mov', 'accvgpr')}, cmp {pick('v_cmp')}, perm/dpp {pick('permlane', 'dpp', 'bpermute', 'swizzle')})  "
                  f"ds {pick('ds_')}  vmem {pick('global_', 'buffer_')}  salu {sum(v for k, v in c.items() if k.startswith('s_'))} (nop {c['s_nop']}, wait {c['s_waitcnt']}, barrier {c['s_barrier']})")


if __name__ == "__main__":
    main()
