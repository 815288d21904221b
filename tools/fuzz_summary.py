#!/usr/bin/env python3
"""Collects the round's fuzz runs (gpurun_out/<round>/fuzz_*.log) into profiles/<round>_fuzz_summary.json: python tools/fuzz_summary.py r05 [notes...]."""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r05"
runs = []
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", ROUND, "fuzz_*.log"))):
    txt = open(f).read()
    cases = len(re.findall(r"^case \d+:", txt, re.M))
    mism = len(re.findall(r"MISMATCH", txt))
    m = re.search(r"(\d+) cases \((\d+) with the average kernel too\), (\d+) with mismatches; (\d+) certified views, (\d+) not", txt)
    ent = {"log": os.path.basename(f), "cases_started": cases, "mismatch_lines": mism, "gpu_fault": "Memory access fault" in txt}
    mr = re.search(r"(\d+) cases also as the ranks of an N > 1 run", txt)
    if mr:
        ent["cases_also_as_ranks_through_the_sparse_wire"] = int(mr.group(1))
    if m:
        ent.update(cases_completed=int(m.group(1)), with_average=int(m.group(2)), cases_with_mismatches=int(m.group(3)),
                   certified_views=int(m.group(4)), uncertified_views=int(m.group(5)))
    runs.append(ent)
out = {"tool": "tools/fuzz_carve.py <cases> <seed> on one MI355X (random grids, scenes, rigs, default values, knob sets; "
               "host masks or a device batch; fresh volume + a second batch; every third case also the average kernel)",
       "runs": runs, "notes": sys.argv[2:]}
json.dump(out, open(os.path.join(ROOT, "profiles", ROUND + "_fuzz_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
