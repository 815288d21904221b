#!/usr/bin/env python3
"""Host masks -> host labels (the bench's e2e leg) under a pool size / host-pack setting: all repetitions printed."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes
from plant3dvision_amd.cl import Backprojection
threads, host_pack, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
shape, origin, vs, views = scenes.make_scene((512, 512, 512), 72, "plant")
bp = Backprojection(list(shape), origin, vs, device=0)
bp._engine.set_option(nat.SC_OPT_HOST_THREADS, threads)
bp._engine.set_option(nat.SC_OPT_HOST_PACK, host_pack)
def run():
    bp.clear()
    t0 = time.perf_counter()
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    t1 = time.perf_counter()
    bp._engine.synchronize()
    t2 = time.perf_counter()
    v = bp.get_values()
    t3 = time.perf_counter()
    return v, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3
vol, *_ = run()
rows = []
for _ in range(reps):
    bp.recycle(vol); del vol
    t0 = time.perf_counter()
    vol, a, b, c = run()
    rows.append([round((time.perf_counter() - t0) * 1e3, 2), round(a, 2), round(b, 2), round(c, 2)])
print(json.dumps({"threads": threads, "host_pack": host_pack, "median_total": float(np.median([r[0] for r in rows])),
                  "median_submit": float(np.median([r[1] for r in rows])), "median_sync": float(np.median([r[2] for r in rows])),
                  "median_readback": float(np.median([r[3] for r in rows])), "totals": [r[0] for r in rows]}))
