#!/usr/bin/env python3
"""Probe (GPU box): what does the bit-packing kernel's time depend on -- the bytes, or how a block's reads lie
in memory?  Packs V masks of the same number of bytes in three shapes (all ahead of the dense stage,
SC_OPT_PACK_RIDE 0) on a tiny grid and prints the mean time of the pack kernel per batch:
  1440 x 1080   a block reads 128 rows x 128 B, 1440 B apart (the bench's masks);
  128 x 12150   the same bytes, a block's 16 KB contiguous;
  2880 x 540    rows twice as long.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plant3dvision_amd import _native as nat  # noqa: E402


def main():
    V = 72
    out = {}
    rng = np.random.default_rng(1)
    for W, H in ((1440, 1080), (128, 12150), (2880, 540), (1440, 1080)):
        eng = nat.Engine((16, 32, 64), (0.0, 0.0, 0.0), 1.0, nat.SC_MODE_CARVE, device=0)
        eng.set_option(nat.SC_OPT_PACK_RIDE, 0)
        stack = (rng.random((V, H, W)) < 0.3).astype(np.uint8) * 255
        ptr = eng.dev_alloc(stack.nbytes)
        eng.dev_upload(ptr, stack)
        K = np.tile(np.array([1000.0, 1000.0, W / 2, H / 2], dtype=np.float32), (V, 1))
        R = np.tile(np.eye(3, dtype=np.float32).reshape(9), (V, 1))
        t = np.tile(np.array([-8.0, -16.0, 500.0], dtype=np.float32), (V, 1))

        def step():
            eng.clear()
            eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
            eng.flush()

        for _ in range(3):
            step()
        eng.synchronize()
        eng.set_option(nat.SC_OPT_TIME_KERNELS, 1)
        eng.reset_kernel_stats()
        for _ in range(5):
            step()
        eng.synchronize()
        n, ms = eng.kernel_stats(nat.SC_KERNEL_PACK)
        out[f"{W}x{H}"] = {"pack_ms_per_batch": round(ms / 5, 4), "launches": n, "MB": round(stack.nbytes / 1e6, 1),
                           "TB_per_s": round(stack.nbytes / (ms / 5 * 1e-3) / 1e12, 2)}
        eng.dev_free(ptr)
        eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
