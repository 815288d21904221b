#!/usr/bin/env python3
"""Basic blocks of one kernel in build/spacecarve.s with instruction counts (VALU / SALU / memory).
    python tools/isa_blocks.py carve_list_kernelILb1 [--dump LBB12_7]"""
import re
import sys

s = open("build/spacecarve.s").read()
key = sys.argv[1]
m = re.search(r"^(_ZN\S*" + re.escape(key) + r"\S*):[^\n]*\n(.*?)\n\.Lfunc_end", s, re.S | re.M)
print(m.group(1))
lines = m.group(2).split("\n")
bb = [["entry", []]]
for l in lines:
    if re.match(r"^\.LBB\d+_\d+:", l):
        bb.append([l.split(":")[0], []])
    elif l.strip() and not l.strip().startswith(";") and not l.strip().startswith("."):
        bb[-1][1].append(l.strip())
dump = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--dump" else None
for name, ins in bb:
    v = sum(1 for i in ins if i.startswith("v_"))
    sa = sum(1 for i in ins if i.startswith("s_"))
    g = sum(1 for i in ins if i.startswith(("global_", "flat_", "buffer_", "ds_")))
    print(f"{name:12s} {len(ins):4d}  valu {v:4d}  salu {sa:4d}  mem {g:3d}")
    if dump and name.endswith(dump):
        print("\n".join("    " + i for i in ins))
