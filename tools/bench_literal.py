#!/usr/bin/env python3
"""The reference's literal configuration (configs/test_geom_pipe_real.toml:27-36 -> 301 x 301 x 561 voxels,
60 views; nz is not a multiple of 4, so label rows are not 16-byte aligned) and a few other odd shapes:
fused carve, device batch, ms per batch.  Diagnostic; prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import scenes, _native as nat

def run(shape, origin, vs, views, steps=20):
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = eng.dev_alloc(stack.nbytes); eng.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    def go(n):
        for _ in range(n):
            eng.clear(); eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8); eng.flush()
    go(3); eng.synchronize()
    eng.span_begin(); go(steps); ms = eng.span_end() / steps
    live, s0, s1, ovf = eng.fused_counts()
    late = eng.fused_counts_ex()["late_bricks"]
    eng.dev_free(ptr); eng.close()
    n = int(np.prod(shape))
    return {"shape": list(shape), "views": V, "ms_per_batch": ms, "Mvoxel_views_per_s": n * V / ms / 1e3,
            "label_write_GBps": 4.0 * n / ms / 1e6, "live_bricks": live, "late_bricks": late, "survivors": [s0, s1], "overflow": ovf}

out = {}
shapes = [(300, 300, 560), (304, 304, 576), (512, 512, 512), (500, 500, 500), (511, 513, 509)]
if len(sys.argv) > 1:  # e.g. 500x500x500 literal
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if a not in ("literal", "dense")]
if len(sys.argv) == 1 or "literal" in sys.argv[1:]:
    shape, origin, vs, views = scenes.literal_real_plant_scene(60, "plant")
    out["literal_301x301x561_60_views"] = run(shape, origin, vs, views)
if len(sys.argv) == 1 or "dense" in sys.argv[1:]:
    shape, origin, vs, views = scenes.make_scene(512, 72, "dense")
    out["dense_512_72_views"] = run(shape, origin, vs, views)
shapes = [s_ for s_ in shapes if s_]
for shp in shapes:
    shape, origin, vs, views = scenes.make_scene(shp, 72, "plant")
    out["x".join(map(str, shp)) + "_72_views"] = run(shape, origin, vs, views)
print(json.dumps(out))
