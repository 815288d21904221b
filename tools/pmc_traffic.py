#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/profile_gpu.sh into the summaries kept under profiles/.

    python tools/pmc_traffic.py <tag> [--scene plant --n 512 --views 72]

Reads  gpurun_out/prof_<tag>_{stats,fetch,write}/**.csv
Writes profiles/<tag>_kernel_stats.csv      (rocprofv3 --kernel-trace --stats summary, verbatim)
       profiles/<tag>_pmc.json              (per kernel: mean FETCH_SIZE / WRITE_SIZE, corrected bytes)
       profiles/pmc_traffic.json            (what bench.py puts in roofline.traffic)

HBM-byte rule (/opt/skills/guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming
read (16 B/lane), so the state-streaming kernel's read side is doubled; WRITE_SIZE is exact
for 16 B/lane stores.  The fused kernel's reads are 4-byte gathers (uncalibrated width): its
FETCH_SIZE is kept raw and flagged.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(tag, which):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{which}", "**", "*_counter_collection.csv"),
                      recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "n": len(v)} for k, v in agg.items()}


def short(name):
    for key in ("carve_kernel_1<false", "carve_kernel_1<true", "carve_kernel<true", "carve_kernel<false",
                "carve_brick_kernel<true", "carve_brick_kernel<false", "carve_brick_light_kernel<true",
                "carve_brick_light_kernel<false", "brick_flags_kernel", "brick_confirm_kernel",
                "carve_list_kernel<true", "carve_list_kernel<false", "carve_special_kernel", "bits_tiles_kernel", "carve_resume_kernel<true",
                "carve_resume_kernel<false", "average_kernel", "pack16_kernel", "pack_band_kernel", "pack_kernel", "fill_kernel"):
        if key in name:
            return key + (">" if "<" in key else "")
    return name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--scene", default="plant")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--views", type=int, default=72)
    a = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{a.tag}_stats", "**", "*_kernel_stats.csv"),
                      recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{a.tag}_kernel_stats.csv"))
    fetch = mean_counter(a.tag, "fetch")
    write = mean_counter(a.tag, "write")
    per_kernel = {}
    for name in sorted(set(fetch) | set(write)):
        s = short(name)
        f = fetch.get(name, {}).get("mean")
        w = write.get(name, {}).get("mean")
        wide = s.startswith("carve_kernel_1<false") or s.startswith("carve_kernel<false") or s.startswith("pack16_kernel") or s.startswith("pack_band_kernel")
        ent = {"kernel": name, "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w,
               "launches_fetch_pass": fetch.get(name, {}).get("n"),
               "read_correction": 2.0 if wide else 1.0,
               "read_note": ("16 B/lane streaming read: FETCH_SIZE x2 (gfx950)" if wide else
                             "reads are not 16 B/lane streams: FETCH_SIZE raw, uncalibrated")}
        if f is not None and w is not None:
            ent["hbm_bytes_per_launch"] = (f * ent["read_correction"] + w) * 1024.0
        per_kernel[s] = ent
    json.dump(per_kernel, open(os.path.join(ROOT, "profiles", f"{a.tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    # stream: the per-view kernel.  fused: one batch = one launch of each kernel of the sequence.
    key = f"stream_{a.scene}_{a.n}_{a.views}"
    for s, ent in per_kernel.items():
        if s.startswith("carve_kernel_1<false") and "hbm_bytes_per_launch" in ent:
            traffic[key] = {"hbm_bytes_per_launch": ent["hbm_bytes_per_launch"], "source": f"profiles/{a.tag}_pmc.json",
                            "kernel": ent["kernel"], "read_correction": ent["read_correction"]}
    seq = ("pack16_kernel", "pack_band_kernel", "brick_flags_kernel", "carve_brick_kernel<true>", "brick_confirm_kernel", "carve_kernel<true>",
           "carve_special_kernel", "carve_list_kernel<false>", "carve_list_kernel<true>", "carve_resume_kernel<true>")
    parts = {s: per_kernel[s]["hbm_bytes_per_launch"] for s in seq
             if s in per_kernel and "hbm_bytes_per_launch" in per_kernel[s]}
    if "carve_list_kernel<true>" in parts:  # a run of the fused schedule (a stream run has pack and flags launches too)
        traffic[f"fused_{a.scene}_{a.n}_{a.views}"] = {
            "hbm_bytes_per_launch": sum(parts.values()), "source": f"profiles/{a.tag}_pmc.json",
            "kernel": "fused batch = one launch each of: " + ", ".join(parts), "per_kernel": parts,
            "read_correction": "the pack kernel x2 (16 B/lane streaming reads); the others raw -- carve_brick_kernel also "
                               "carries the packing riders' wide reads of the remaining masks, which FETCH_SIZE "
                               "reports at half their bytes: the sum understates the batch by up to 50 MB"}
    json.dump(traffic, open(traffic_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(per_kernel, indent=1))


if __name__ == "__main__":
    main()
