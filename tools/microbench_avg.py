#!/usr/bin/env python3
"""Ad-hoc timing of the averaging kernels (HIP events) on the GPU box."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes

def run(n, V, W, H, vpl, reps=3, u8=False, binary=False, brick=1):
    shape, origin, vs, views = scenes.make_scene(n, V, "solid", width=W, height=H, fx=FXS * W, fy=FXS * W, cx=W / 2, cy=H / 2)
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE)
    e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
    e.set_option(nat.SC_OPT_AVG_BRICK, brick)
    rng = np.random.default_rng(0)
    if u8 and binary:  # what Segmentation2D writes: the plant's silhouettes, 0 / 255
        _, _, _, pv = scenes.make_scene(n, V, "plant", width=W, height=H, fx=FXS * W, fy=FXS * W, cx=W / 2, cy=H / 2)
        stack = np.ascontiguousarray(np.stack([m for _, _, _, m in pv]))
        e.set_lut(np.log(np.float32(1e-10) + np.arange(256, dtype=np.float32) / np.float32(255)))
    elif u8:
        stack = rng.integers(0, 256, (V, H, W), dtype=np.uint8)
        e.set_lut(np.arange(256, dtype=np.float32) / np.float32(255))
    else:
        stack = rng.random((V, H, W), dtype=np.float32)
    code = nat.SC_MASK_U8_LUT if u8 else nat.SC_MASK_F32
    ptr = e.dev_alloc(stack.nbytes); e.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    for it in range(reps + 1):
        if it == 1:
            e.set_option(nat.SC_OPT_TIME_KERNELS, 1); e.reset_kernel_stats()
        e.clear(); e.process_views_device(K, R, t, ptr, V, H, W, code); e.flush()
    e.synchronize()
    c, ms = e.kernel_stats(nat.SC_KERNEL_AVERAGE)
    nvv = n ** 3 * V
    per_step = ms / reps
    print(f"average {'u8+table' if u8 else 'float32 '}{' binary' if binary else ''} brick={brick} n={n} V={V} {W}x{H} vpl={vpl}: {per_step:.3f} ms/step, {c // reps} launches, "
          f"{nvv / per_step / 1e3:.4g} Mvoxel*views/s, {8.0 * n**3 * (c // reps) / per_step / 1e6:.0f} GB/s state traffic")
    e.dev_free(ptr); e.close()

FXS = 1163.6854 / 1440
if __name__ == "__main__":
    run(512, 72, 1440, 1080, 0)
    run(512, 72, 1440, 1080, 1)
    run(512, 18, 896, 896, 0)
    run(512, 18, 896, 896, 1)
    run(512, 72, 1440, 1080, 0, u8=True)
    run(512, 72, 1440, 1080, 0, u8=True, brick=0)
    run(512, 72, 1440, 1080, 1, u8=True)
    run(512, 72, 1440, 1080, 0, u8=True, binary=True)
    run(512, 72, 1440, 1080, 0, u8=True, binary=True, brick=0)
    run(512, 18, 896, 896, 0, u8=True)
    run(512, 18, 896, 896, 1, u8=True)
