#!/usr/bin/env python3
"""One-knob-at-a-time sweep of the fused carve on a scene (plant, dense, noise, solid, literal; GPU box): ms per batch by option value, three
repeats each, the best kept.  usage: python tools/sweep_knobs.py [scene] [sets.json] > out.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes  # noqa: E402

SWEEP = [
    (),
    (("SC_OPT_STAGE1_STORE_SHARE", 0), ("SC_OPT_STAGE1_LIST_BLOCKS", 2048)),
    (("SC_OPT_STAGE1_STORE_SHARE", 0), ("SC_OPT_STAGE1_LIST_BLOCKS", 1792)),
    (("SC_OPT_STAGE1_STORE_SHARE", 0), ("SC_OPT_STAGE1_LIST_BLOCKS", 1536)),
    (("SC_OPT_STAGE1_STORE_SHARE", 0), ("SC_OPT_STAGE1_LIST_BLOCKS", 1280)),
    (("SC_OPT_STAGE1_STORE_SHARE", 2), ("SC_OPT_STAGE1_LIST_BLOCKS", 1792)),
    (("SC_OPT_STAGE1_STORE_SHARE", 2), ("SC_OPT_STAGE1_LIST_BLOCKS", 1536)),
    (("SC_OPT_STAGE1_STORE_SHARE", 4), ("SC_OPT_STAGE1_LIST_BLOCKS", 1792), ("SC_OPT_FILL_BLOCKS", 256)),
    (("SC_OPT_STAGE1_VOXELS", 1), ("SC_OPT_STAGE1_LIST_BLOCKS", 1792), ("SC_OPT_STAGE1_STORE_SHARE", 0)),
    (("SC_OPT_STAGE1_VOXELS", 1),),
    (("SC_OPT_FINAL_VOXELS", 1),), (("SC_OPT_FINAL_VOXELS", 4),),
    (("SC_OPT_DEFER_SHARE", 14),), (("SC_OPT_DEFER_SHARE", 12),),
    (),
]


def main():
    global SWEEP
    kind = sys.argv[1] if len(sys.argv) > 1 else "plant"
    if len(sys.argv) > 2:  # a sweep of one's own: a JSON list of {option: value} sets
        SWEEP = [tuple(d.items()) for d in json.load(open(sys.argv[2]))]
    if kind == "literal":  # the reference's own configuration: 301 x 301 x 561 voxels, 60 views
        shape, origin, vs, views = scenes.literal_real_plant_scene(60, "plant")
    else:
        shape, origin, vs, views = scenes.make_scene(512, 72, kind)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    out = []
    for opts in SWEEP:
        eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, device=0)
        for k, v in opts:
            eng.set_option(getattr(nat, k), v)
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
        out.append({"opts": dict(opts), "ms": round(best, 4)})
        print(out[-1], file=sys.stderr, flush=True)
        eng.dev_free(ptr)
        eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
