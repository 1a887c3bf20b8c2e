#!/usr/bin/env python3
"""Instruction-class sequence of the hottest basic block (most v_mfma) of one kernel in the library's ISA listing:
    make -C evo_amd/csrc asm;  python tools/isa_loop.py gemm_tn128_gkIdE [/tmp/evo_amd-hip-amdgcn-amd-amdhsa-gfx950.s]
M = v_mfma, r / w = LDS read / write, G = global load, v = other vector, s = scalar, B = s_barrier, |..| = s_waitcnt."""
import re
import sys

pat = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/evo_amd-hip-amdgcn-amd-amdhsa-gfx950.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % pat, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur, name = [], [], "entry"
for ln in lines[start + 1:end]:
    if re.match(r"^\.LBB\d+_\d+:", ln):
        blocks.append((name, cur))
        name, cur = ln.split(":")[0], []
    else:
        cur.append(ln)
blocks.append((name, cur))
for name, body in sorted(blocks, key=lambda b: -sum("v_mfma" in l for l in b[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    seq = []
    for l in body:
        l = l.strip()
        if not l or l[0] in ";.":
            continue
        op = l.split()[0]
        seq.append("M" if op.startswith("v_mfma") else "r" if op.startswith(("ds_read", "ds_load")) else
                   "w" if op.startswith(("ds_write", "ds_store")) else "G" if op.startswith(("global_load", "buffer_load")) else
                   "|" + l.split(None, 1)[1].replace(" ", "") + "|" if op.startswith("s_waitcnt") else
                   "B" if op.startswith("s_barrier") else "v" if op.startswith("v_") else "s" if op.startswith("s_") else "?")
    out = "".join(seq)
    print("%s: %d MFMA, %d vector, %d LDS, %d global loads" % (name, out.count("M"), out.count("v"), out.count("r") + out.count("w"), out.count("G")))
    print(out)
