#!/usr/bin/env python3
"""Two engines (two streams, two label volumes) carving independent batches side by side on one GPU
against one engine doing them one after the other (diagnostic): batches per second."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import scenes, _native as nat
shape, origin, vs, views = scenes.make_scene(512, 72, "plant")
stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
V, H, W = stack.shape
engs = []
for _ in range(3):
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    p = e.dev_alloc(stack.nbytes); e.dev_upload(p, stack)
    engs.append((e, p))
def batch(e, p):
    e.clear(); e.process_views_device(K, R, t, p, V, H, W, nat.SC_MASK_U8); e.flush()
T = time.perf_counter
for n_eng in (1, 2, 3):
    use = engs[:n_eng]
    for e, p in use:
        batch(e, p)
    for e, p in use:
        e.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = T()
        for i in range(120):
            e, p = use[i % n_eng]
            batch(e, p)
        for e, p in use:
            e.synchronize()
        best = min(best, T() - t0)
    print(f"{n_eng} engine(s): {1e3*best/120:.4f} ms per batch", flush=True)
