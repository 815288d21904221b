#!/usr/bin/env python3
"""profiles/<tag>_clock_ramp.txt from the two per-dispatch traces tools/final_profiles.sh leaves under gpurun_out/<tag>/
(prof_ramp: 400 + 20 batches back to back; prof_cadence: the driver's own command): kernel sums per fused batch.

    python tools/ramp_summary.py r04 > profiles/r04_clock_ramp.txt
"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUSED = ("pack_band_kernel", "brick_flags_kernel", "carve_brick_kernel", "brick_confirm_kernel", "carve_special_kernel", "carve_list_kernel")


def batches(tag, sub):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", tag, sub, "**", "*_kernel_trace.csv"), recursive=True)
    rows = []
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
        n = r["Kernel_Name"]
        if any(k in n for k in FUSED):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    out = []
    for s, e, n in rows:
        if "pack_band_kernel" in n:
            out.append({"t": s, "sum": 0.0, "final": 0.0, "per": {}})
        if not out:
            continue
        b = out[-1]
        b["sum"] += (e - s) / 1e3
        key = next(k for k in FUSED if k in n) + ("<true" if "carve_list_kernel<true" in n else "<false" if "carve_list_kernel<false" in n else "")
        b["per"][key] = b["per"].get(key, 0.0) + (e - s) / 1e3
        if "carve_list_kernel<true" in n:
            b["final"] = (e - s) / 1e3
    return [b for b in out if len(b["per"]) >= 6]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    ramp = batches(tag, "prof_ramp")
    t0 = ramp[0]["t"]
    print(f"rocprofv3 --kernel-trace of `python3 bench.py --gpus 1 --steps 400 --warmup 20 --skip-other-path ...` (tools/final_profiles.sh {tag}):")
    print(f"{len(ramp)} fused batches of the plant scene back to back after the host-side set-up (device idle before batch 0);")
    print("per group of 35 batches: time of the group's first batch since batch 0, mean sum of the seven kernels' durations, mean of the final stage.\n")
    print("batch   t[ms]  kernel-sum[us]  final-stage[us]")
    for i in range(0, len(ramp), 35):
        g = ramp[i:i + 35]
        print(f"{i:5d} {(g[0]['t'] - t0) / 1e6:7.1f} {sum(b['sum'] for b in g) / len(g):15.1f} {sum(b['final'] for b in g) / len(g):16.1f}")
    cad = batches(tag, "prof_cadence")
    t0 = min(b["t"] for b in cad)
    print(f"\nThe driver's own command under rocprofv3 (`--steps 20 --warmup 5`, the stream and per_view legs in front): fused batches by start time")
    print("batch   t[ms]  kernel-sum[us]")
    for i, b in enumerate(cad):
        print(f"{i:5d} {(b['t'] - t0) / 1e6:7.1f} {b['sum']:15.1f}")
    timed = cad[6:26] if len(cad) >= 26 else cad
    keys = sorted({k for b in timed for k in b["per"]})
    print(f"\nmeans over batches 6..25 (the 20 timed ones: 1 set-up + 5 warm-up in front): " +
          " + ".join(f"{k} {sum(b['per'].get(k, 0.0) for b in timed) / len(timed):.1f}" for k in keys) +
          f" = {sum(b['sum'] for b in timed) / len(timed):.1f} us")


if __name__ == "__main__":
    main()
