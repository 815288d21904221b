#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in boundary: 72 uint8 masks in HOST memory (decoded arrays,
what Backprojection.process_view receives, cl.py:190) -> carved volume, 512^3.

  in    host masks -> labels in HBM      (sc_process_view x V + sc_synchronize)
  out   ... -> labels in host memory     (+ sc_get_values: 512 MiB device -> host)

Prints one JSON line.  Not the headline metric (bench.py keeps the masks resident)."""
import argparse, json, os, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--views", type=int, default=72)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from plant3dvision_amd import _native as nat, scenes
    shape, origin, vs, views = scenes.make_scene((a.n,) * 3, a.views, "plant")
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    V = len(views)
    N = int(np.prod(shape))
    mask_bytes = sum(m.nbytes for _, _, _, m in views)

    def run(read_back):
        e.clear()
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        if read_back:
            return e.get_values()
        e.flush()
        e.synchronize()

    run(True)  # warm: allocations, pinned ring
    t_in, t_out = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter(); run(False); t_in.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); run(True); t_out.append(time.perf_counter() - t0)
    ti, to = min(t_in), min(t_out)
    # opt-in: a page-locked destination kept across read-backs (sc_host_alloc)
    t0 = time.perf_counter(); pinned = nat.pinned_empty(shape, np.int32); t_pin = time.perf_counter() - t0
    t_rb = []
    for _ in range(a.reps):
        run(False)
        t0 = time.perf_counter(); e.get_values(pinned); t_rb.append(time.perf_counter() - t0)
    t_pg = []
    pageable = np.empty(shape, np.int32)
    for _ in range(a.reps):
        run(False)
        t0 = time.perf_counter(); e.get_values(pageable); t_pg.append(time.perf_counter() - t0)
    print(json.dumps({
        "read_back_pageable_ms": min(t_pg) * 1e3, "read_back_page_locked_ms": min(t_rb) * 1e3,
        "page_locking_512MiB_ms": t_pin * 1e3,
        "workload": f"{a.n}^3 x {V} host uint8 masks {views[0][3].shape[1]}x{views[0][3].shape[0]} ({mask_bytes / 1e6:.0f} MB)",
        "host_masks_to_labels_in_hbm_ms": ti * 1e3, "rate_in_Mvoxel_views_per_s": N * V / ti / 1e6,
        "host_masks_to_labels_in_host_memory_ms": to * 1e3, "rate_out_Mvoxel_views_per_s": N * V / to / 1e6,
        "mask_upload_GBps": mask_bytes / ti / 1e9}))
    e.close()


if __name__ == "__main__":
    main()
