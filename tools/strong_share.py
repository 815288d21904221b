#!/usr/bin/env python3
"""What one rank of a strong-scaled run does (GPU box): the 512^3 x 72 plant scene with the x-planes dealt
cyclically over W ranks, rank 0's share carved on this GPU, for W = 1, 2, 4, 8 -- device time per batch and the
kernel breakdown.  The collective is not here (one GPU); the driver's SCALE run has it.

    python tools/strong_share.py > out.json
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes  # noqa: E402


def main():
    shape, origin, vs, views = scenes.make_scene(512, 72, "plant")
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    out = {}
    for world in (1, 2, 4, 8):
        eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, device=0, cyclic=(0, world))
        ptr = eng.dev_alloc(stack.nbytes)
        eng.dev_upload(ptr, stack)

        def step():
            eng.clear()
            eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
            eng.flush()

        for _ in range(4):
            step()
        eng.synchronize()
        best = 1e9
        for _ in range(3):
            eng.span_begin()
            for _ in range(20):
                step()
            best = min(best, eng.span_end() / 20)
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
        out[f"world_{world}"] = {"planes": 512 // world, "ms_per_batch": round(best, 4), "kernels_ms": ks,
                                 "speedup_over_1": None}
        eng.dev_free(ptr)
        eng.close()
    base = out["world_1"]["ms_per_batch"]
    for k, v in out.items():
        v["speedup_over_1"] = round(base / v["ms_per_batch"], 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
