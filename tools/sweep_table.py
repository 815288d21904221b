#!/usr/bin/env python3
"""Table of bench JSON lines: python tools/sweep_table.py gpurun_out/r02/sw1/*.json"""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as ex:
        print(f, "unreadable", ex)
        continue
    b = d["kernels_breakdown_pass"]
    print("%-40s step %.4f  event %.4f  frac %.3f | %s" % (
        f.split("/")[-1], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"],
        " ".join("%s %.1f" % (k, b[k]["avg_ms"] * 1e3) for k in ("pack", "flags", "carve", "list", "step"))))
