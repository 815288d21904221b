#!/usr/bin/env python3
"""Read-back over the 2-bit wire alone: ms per call into a touched int32 array (GPU box)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 0
shape, origin, vs, views = scenes.make_scene((512, 512, 512), 12, "plant")
e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
if threads:
    e.set_option(nat.SC_OPT_HOST_THREADS, threads)
for K, R, t, m in views:
    e.process_view(K, R, t, m, nat.SC_MASK_U8)
e.synchronize()
out = np.zeros(512 ** 3, dtype=np.int32)
stg = np.zeros(512 ** 3 // 4, dtype=np.uint8)
ts = []
for _ in range(8):
    t0 = time.perf_counter()
    e.get_values_wire2(out, stg)
    ts.append((time.perf_counter() - t0) * 1e3)
# the widening alone
packed = np.ascontiguousarray(e.get_values_packed(2)).view(np.uint32).reshape(-1)[: 512 ** 3 // 16]
tw = []
for _ in range(5):
    t0 = time.perf_counter()
    nat.widen_labels2(packed, 512 ** 3, out=out)
    tw.append((time.perf_counter() - t0) * 1e3)
print(json.dumps({"threads": threads, "nt": os.environ.get("SC_WIDEN_NT", "1"), "wire2_ms": [round(x, 2) for x in ts], "widen_only_ms": [round(x, 2) for x in tw]}))
