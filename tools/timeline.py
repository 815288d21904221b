"""Kernel timeline of one fused step from a rocprofv3 --kernel-trace CSV (diagnostic)."""
import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_y"
which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n[n.find("::") + 2:] if "::" in n else n
    return n[:26]
idx = [i for i, r in enumerate(rows) if "pack16" in r["Kernel_Name"]]
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]["Start_Timestamp"]); prev = None
for r in rows[i0:i1 + 1]:
    s = int(r["Start_Timestamp"]) - t0; e = int(r["End_Timestamp"]) - t0
    print(f"{short(r['Kernel_Name']):28s} start {s/1000:8.1f} dur {(e-s)/1000:7.1f} gap {((s-prev) if prev is not None else 0)/1000:6.1f}")
    prev = e
