#!/usr/bin/env python3
"""Timing of the Voxels -> PointCloud hand-over: vol2pcd on the device-resident
carve volume vs. the reference algorithm (SciPy/NumPy, oracle/vol2pcd_oracle.py) on the host."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--cpu-n", type=int, default=192)
    ap.add_argument("--views", type=int, default=72)
    a = ap.parse_args()
    from plant3dvision_amd import proc3d, scenes
    from plant3dvision_amd.cl import Backprojection
    from oracle import vol2pcd_oracle
    out = {}
    for n, do_cpu in ((a.cpu_n, True), (a.n, False)):
        shape, origin, vs, views = scenes.make_scene(n, a.views, "plant")
        bp = Backprojection(shape, origin, vs)
        for K, R, t, m in views:
            bp.process_view(K, R, t, m)
        bp.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            pc = proc3d.vol2pcd(bp, np.array(origin), vs, 1.0, as_open3d=False)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        key = f"{n}^3"
        out[key] = {"gpu_s": best, "points": len(pc), "Mvoxels_per_s_gpu": n ** 3 / best / 1e6}
        if do_cpu:
            vol = bp.get_values().copy()
            t0 = time.perf_counter()
            pts, *_ = vol2pcd_oracle.vol2pcd(vol, np.array(origin), vs, 1.0)
            out[key]["scipy_s"] = time.perf_counter() - t0
            out[key]["speedup"] = out[key]["scipy_s"] / best
            assert len(pts) == len(pc)
        bp.close()
    print(json.dumps(out))

if __name__ == "__main__":
    main()
