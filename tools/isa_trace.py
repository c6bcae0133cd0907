#!/usr/bin/env python3
"""Compact instruction-class trace of one kernel in a hipcc -S dump (diagnostic: shows MFMA / VALU / LDS interleave).
usage: isa_trace.py k.s <mangled-name-substring> [width]
M mfma, v valu, t transcendental, d ds_read, D ds_write, g global load, G global store, s salu, w s_waitcnt, b branch/barrier, x scratch"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
width = int(sys.argv[3]) if len(sys.argv) > 3 else 128
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(key.split("$")[-1]) or (l.startswith("_Z") and key in l.split(":")[0]))
out = []
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith(".Lfunc_end"): break
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"\.LBB\d+_\d+:", s): out.append("|")
        continue
    op = s.split()[0]
    if "mfma" in op: c = "M"
    elif op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")): c = "t"
    elif op.startswith("v_"): c = "v"
    elif op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"): c = "d"
    elif op.startswith("ds_"): c = "D"
    elif op.startswith("scratch_"): c = "x"
    elif op.startswith(("global_load", "buffer_load", "flat_load")): c = "g"
    elif op.startswith(("global_", "buffer_", "flat_")): c = "G"
    elif op.startswith("s_waitcnt"): c = "w"
    elif op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_nop")): c = "b" if not op.startswith("s_nop") else "n"
    elif op.startswith("s_"): c = "s"
    else: c = "?"
    out.append(c)
t = "".join(out)
for i in range(0, len(t), width): print(f"{i:6d} {t[i:i+width]}")
from collections import Counter
print(Counter(t))
