#!/usr/bin/env python3
"""The averaging launch on the bench's four mask forms (GPU box): 512^3 x 72 views, uint8 binary / uint8 grey (bytes +
table) and float32 binary / grey, HIP events around the averaging kernel.  SPACECARVE_LIB=<other build> for an A/B.
    python tools/bench_avg.py [--reps 5] [--forms u8_grey,f32_grey,u8_binary,f32] [--tag name]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes  # noqa: E402
from plant3dvision_amd.cl import averaging_table, img_as_float32  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--forms", default="u8_grey,f32_grey,u8_binary,f32")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    shape, origin, vs, views = scenes.make_scene(512, 72, "plant")
    binary = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    grey = np.random.default_rng(4321).integers(0, 256, binary.shape, dtype=np.uint8)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = binary.shape
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE, device=0)
    eng.set_lut(averaging_table(False))
    buf = eng.dev_alloc(binary.size * 4)
    out = {"tag": a.tag, "lib": nat.LIB_PATH}
    forms = {"u8_binary": (binary, nat.SC_MASK_U8_LUT), "u8_grey": (grey, nat.SC_MASK_U8_LUT),
             "f32": (None, nat.SC_MASK_F32), "f32_grey": (None, nat.SC_MASK_F32)}
    for name in a.forms.split(","):
        data, code = forms[name]
        if data is None:
            data = img_as_float32(binary if name == "f32" else grey)
        eng.dev_upload(buf, data)
        for it in range(a.reps + 2):
            if it == 2:
                eng.set_option(nat.SC_OPT_TIME_KERNELS, 1)
                eng.reset_kernel_stats()
            eng.clear()
            eng.process_views_device(K, R, t, buf, V, H, W, code)
            eng.flush()
        eng.synchronize()
        _, ms = eng.kernel_stats(nat.SC_KERNEL_AVERAGE)
        eng.set_option(nat.SC_OPT_TIME_KERNELS, 0)
        out[name] = {"ms": round(ms / a.reps, 4)}
    eng.dev_free(buf)
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
