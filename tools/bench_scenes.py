#!/usr/bin/env python3
"""Fused carve time per scene (GPU box): plant / dense / noise / solid at n^3 x V and the reference's literal
301 x 301 x 561 x 60 configuration, with engine options from the command line -- the A/B tool for changes to
the survivor stages.  SPACECARVE_LIB=<other build> runs the same thing on another build of the library.

    python tools/bench_scenes.py [--n 512] [--views 72] [--steps 10] [--scenes plant,dense,literal,noise]
                                 [--opt KEY=VALUE ...] [--tag name]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--views", type=int, default=72)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--scenes", default="plant,dense,literal,noise,solid")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--tag", default="")
    ap.add_argument("--kernels", action="store_true", help="per-kernel HIP-event breakdown (a separate pass)")
    ap.add_argument("--trace-steps", action="store_true", help="host time of every call of the timed steps; reports the slowest step")
    a = ap.parse_args()
    opts = []
    for kv in a.opt:
        k, v = kv.split("=")
        opts.append((getattr(nat, k) if not k.isdigit() else int(k), int(v)))
    out = {"tag": a.tag, "lib": nat.LIB_PATH, "opts": a.opt}
    for kind in a.scenes.split(","):
        if kind == "literal":
            shape, origin, vs, views = scenes.literal_real_plant_scene(60, "plant")
        elif kind == "empty":  # the plant's rig over pictures without foreground: every brick is carved whole (the fill alone)
            shape, origin, vs, views = scenes.make_scene(a.n, a.views, "plant")
            views = [(K, R, t, np.zeros_like(m)) for K, R, t, m in views]
        else:
            shape, origin, vs, views = scenes.make_scene(a.n, a.views, kind)
        eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, device=0)
        for k, v in opts:
            eng.set_option(k, v)
        stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
        ptr = eng.dev_alloc(stack.nbytes)
        eng.dev_upload(ptr, stack)
        K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
        V, H, W = stack.shape

        def step():
            eng.clear()
            eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
            eng.flush()

        for _ in range(3):
            step()
        eng.synchronize()
        eng.span_begin()
        t0 = time.perf_counter()
        worst = (0.0, -1, "")
        for i in range(a.steps):
            if a.trace_steps:  # where does a slow step spend its host time?
                t1 = time.perf_counter(); eng.clear()
                t2 = time.perf_counter(); eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
                t3 = time.perf_counter(); eng.flush()
                t4 = time.perf_counter()
                if t4 - t1 > worst[0]:
                    worst = (t4 - t1, i, f"clear {1e3 * (t2 - t1):.3f} enqueue {1e3 * (t3 - t2):.3f} flush {1e3 * (t4 - t3):.3f} ms")
            else:
                step()
        ms_dev = eng.span_end() / a.steps
        eng.synchronize()
        ms_host = (time.perf_counter() - t0) / a.steps * 1e3
        ent = {"ms": round(ms_dev, 4), "ms_host": round(ms_host, 4)}
        if a.trace_steps:
            ent["slowest_step"] = {"step": worst[1], "host_ms": round(1e3 * worst[0], 3), "parts": worst[2]}
        try:
            ent["counts"] = eng.fused_counts_ex()
        except Exception as exc:  # an older build of the library
            ent["counts"] = str(exc)
        if a.kernels:
            eng.set_option(nat.SC_OPT_TIME_KERNELS, 1)
            eng.reset_kernel_stats()
            for _ in range(3):
                step()
            eng.synchronize()
            ks = {}
            for name, kid in (("pack", nat.SC_KERNEL_PACK), ("flags", nat.SC_KERNEL_FLAGS), ("dense", nat.SC_KERNEL_CARVE),
                              ("lists", nat.SC_KERNEL_LIST)):
                n, ms = eng.kernel_stats(kid)
                ks[name] = round(ms / 3, 4)
            ent["kernels_ms"] = ks
            eng.set_option(nat.SC_OPT_TIME_KERNELS, 0)
        out[kind] = ent
        eng.dev_free(ptr)
        eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
