#!/usr/bin/env python3
"""Phases of masks2d.voxels_from_masks for three averaging labels at 512^3 (GPU box; diagnostics for tools/bench_ml.py)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from plant3dvision_amd import masks2d, scenes
S = 896
shape, origin, vs, views = scenes.make_scene(512, 72, "solid", width=S, height=S, fx=0.8 * S, fy=0.8 * S, cx=S / 2, cy=S / 2)
cams = [scenes.camera_dict(K, R, t) for K, R, t, _ in views]
torch.manual_seed(0)
masks = {}
for name in ("background", "flower", "fruit"):
    m = (torch.rand(72, 14, 14, device="cuda") > 0.5).to(torch.uint8) * 255
    masks[name] = torch.nn.functional.interpolate(m[:, None].float(), size=(S, S), mode="nearest")[:, 0].to(torch.uint8).contiguous()
torch.cuda.synchronize()
for rep in range(4):
    tm = {}
    vols = None
    t0 = time.perf_counter()
    vols = masks2d.voxels_from_masks(masks, cams, shape, origin, vs, type="averaging", log=True, timing=tm)
    t1 = time.perf_counter()
    vols = None
    t2 = time.perf_counter()
    print(json.dumps({"total_ms": round((t1 - t0) * 1e3, 1), "free_ms": round((t2 - t1) * 1e3, 1), **{k: round(v, 1) for k, v in tm.items()}}))
