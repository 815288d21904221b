#!/usr/bin/env python3
"""Ad-hoc kernel timings on the GPU box (HIP events via SC_OPT_TIME_KERNELS)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes

def run(kind, n, V, opts=(), reps=5, vpl=0, label=""):
    shape, origin, vs, views = scenes.make_scene(n, V, kind)
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    for k, v in opts: e.set_option(k, v)
    e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = e.dev_alloc(stack.nbytes); e.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    Vn, H, W = stack.shape
    for it in range(reps + 2):
        if it == 2:
            e.set_option(nat.SC_OPT_TIME_KERNELS, 1); e.reset_kernel_stats()
        e.clear(); e.process_views_device(K, R, t, ptr, Vn, H, W, nat.SC_MASK_U8); e.flush()
    e.synchronize()
    out = {}
    for name, kid in (("carve", 0), ("list", 4), ("pack", 2), ("fill", 3)):
        c, ms = e.kernel_stats(kid)
        if c: out[name] = round(ms / c * 1e3, 1)
    print(f"{label or kind:28s} n={n} V={V} vpl={vpl} opts={list(opts)} us/launch: {out}")
    e.dev_free(ptr); e.close()

if __name__ == "__main__":
    run("empty", 512, 6, label="empty masks: 1 dense view")
    run("plant", 512, 6, label="plant 6 views")
    run("plant", 512, 6, opts=[(5, 0)], label="plant 6 views no compaction")
    run("plant", 512, 1, vpl=1, label="plant 1 view stream fresh")
    run("plant", 512, 72, label="plant 72")
    run("plant", 512, 72, opts=[(6, 1), (7, 1)], label="plant 72 dense=1 (overflow?)")
    run("solid", 512, 6, label="solid 6 (overflow -> the special kernel's dense pass)")
    shape, origin, vs, _ = scenes.make_scene(512, 1, "empty")
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    e.set_option(nat.SC_OPT_TIME_KERNELS, 1)
    for _ in range(4):
        e.clear(); e.values_device_ptr()
    e.synchronize(); print("fill", e.kernel_stats(3))
