#!/usr/bin/env python3
"""Compact trace of the main loop of one kernel from hipcc -S output: one token per instruction
(M = mfma, X = scaled mfma, r = ds_read, w = ds_write, g = global_load, v = VALU, s = SALU,
W(...) = s_waitcnt, B = s_barrier).  Development tool: shows how loads, waits and MFMAs interleave."""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
s = open(path).read()
names = [n for n in re.findall(r'^(\w+):', s, re.M) if re.search(pat, n)]
for name in names:
    i = s.index(name + ":"); j = s.index("s_endpgm", i)
    blocks = re.split(r'\n(\.LBB\d+_\d+):', s[i:j])
    print("==", name)
    for bi in range(1, len(blocks), 2):
        lab, b = blocks[bi], blocks[bi + 1]
        if b.count("v_mfma") < 8:
            continue
        out = []
        for l in b.split("\n"):
            l = l.strip()
            if not l or l[0] in ";.":
                continue
            op = l.split()[0]
            if op.startswith("v_mfma_scale"): out.append("X")
            elif op.startswith("v_mfma"): out.append("M")
            elif op.startswith("ds_read"): out.append("r")
            elif op.startswith("ds_write"): out.append("w")
            elif op.startswith("global_load") or op.startswith("buffer_load"): out.append("g")
            elif op.startswith("global_store"): out.append("S")
            elif op == "s_waitcnt": out.append(" W(" + l.split(None, 1)[1].replace("cnt", "") + ") ")
            elif op == "s_barrier": out.append(" B ")
            elif op.startswith("v_"): out.append("v")
            elif op.startswith("s_"): out.append("s")
            else: out.append("?")
        print(lab, "".join(out))
