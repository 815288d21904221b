#!/usr/bin/env python3
"""Fixed cost of a timed region of K fused batches (diagnostic for bench.py's --steps): time(K) = a + b K."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import scenes, _native as nat
shape, origin, vs, views = scenes.make_scene(512, 72, "plant")
eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
ptr = eng.dev_alloc(stack.nbytes); eng.dev_upload(ptr, stack)
K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
V, H, W = stack.shape
def steps(n):
    for _ in range(n):
        eng.clear()
        eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
        eng.flush()
steps(20); eng.synchronize()
T = time.perf_counter
res = {}
for n in (1, 2, 5, 10, 20, 50, 100, 200):
    best = 1e9
    for rep in range(5):
        eng.synchronize()
        t0 = T(); steps(n); t1 = T(); eng.synchronize(); t2 = T()
        if t2 - t0 < best:
            best, enq = t2 - t0, t1 - t0
    res[n] = best
    print(f"K {n:4d}: {1e3*best:.4f} ms total, {1e3*best/n:.4f} per step, host enqueue {1e3*enq:.4f} ms", flush=True)
ns = np.array(list(res)); ts = np.array([res[n] for n in ns])
b, a = np.polyfit(ns, ts, 1)
print(f"fit: {1e3*a:.4f} ms fixed + {1e3*b:.4f} ms per step")
