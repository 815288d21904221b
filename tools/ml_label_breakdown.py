#!/usr/bin/env python3
"""One label of the ML feeder, phase by phase (diagnostic for tools/bench_ml.py): device work, read-back, exp / clip."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plant3dvision_amd import masks2d, scenes, _native as nat
from plant3dvision_amd.cl import averaging_table
from plant3dvision_amd.tasks.cl import _exp_clip
S = 896
shape, origin, vs, views = scenes.make_scene(512, 72, "solid", width=S, height=S, fx=0.8 * S, fy=0.8 * S, cx=S / 2, cy=S / 2)
K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
torch.manual_seed(0)
m = (torch.rand(72, 14, 14, device="cuda") > 0.5).to(torch.uint8) * 255
m = torch.nn.functional.interpolate(m[:, None].float(), size=(S, S), mode="nearest")[:, 0].to(torch.uint8).contiguous()
torch.cuda.synchronize()
T = time.perf_counter
for mode, code in ((nat.SC_MODE_AVERAGE, nat.SC_MASK_U8_LUT), (nat.SC_MODE_CARVE, nat.SC_MASK_U8)):
    eng = nat.Engine(shape, origin, vs, mode)
    if mode == nat.SC_MODE_AVERAGE:
        eng.set_lut(averaging_table(True))
    for rep in range(3):
        vol = out = dest = h = None  # released outside the timed phases
        eng.clear()
        t0 = T()
        h = nat.TouchedEmpty(tuple(shape), np.float32 if mode == nat.SC_MODE_AVERAGE else np.int32)
        eng.process_views_device(K, R, t, m.data_ptr(), 72, S, S, code)
        eng.flush()
        t1 = T()
        eng.synchronize()
        t2 = T()
        dest = h.result()
        t3 = T()
        vol = eng.get_values(dest)
        t4 = T()
        msg = ""
        if mode == nat.SC_MODE_AVERAGE:
            out = _exp_clip(vol, inplace=True)
            t5 = T()
            o1 = np.exp(vol[:64])
            t6 = T()
            msg = f"  exp/clip {1e3*(t5-t4):.1f} (one thread would take {1e3*(t6-t5)*8:.0f})  min/max {float(out.min()):.3g}/{float(out.max()):.3g}"
        if mode == nat.SC_MODE_AVERAGE:
            # the same label again with the pieces' exp / clip beside the copy (Engine.get_values_pipelined)
            def piece(v):
                np.exp(v, out=v); np.minimum(v, np.float32(1.0), out=v)
            for pb in (8 << 20, 32 << 20, 128 << 20):
                t7 = T()
                eng.get_values_pipelined(dest, piece, piece_bytes=pb)
                msg += f"  pipelined({pb >> 20} MiB pieces) {1e3*(T()-t7):.1f}"
        print(f"{'average' if mode == nat.SC_MODE_AVERAGE else 'carve'}: enqueue {1e3*(t1-t0):.1f}  device {1e3*(t2-t1):.1f}  wait pages {1e3*(t3-t2):.1f}  read-back {1e3*(t4-t3):.1f}{msg} ms", flush=True)
    eng.close()
