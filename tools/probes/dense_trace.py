#!/usr/bin/env python3
"""Probe (GPU box, a library built with -DSC_TRACE_DENSE): what the dense stage's walker wavefronts did in the last
batch of a scene -- start and end on the 100 MHz clock, bricks asked about, units projected.
    hipcc ... -DSC_TRACE_DENSE -o build/variants/lib_trace.so ...;  SPACECARVE_LIB=build/variants/lib_trace.so python tools/probes/dense_trace.py plant"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plant3dvision_amd import _native as nat, scenes  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "plant"
    shape, origin, vs, views = scenes.make_scene(512, 72, kind)
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, device=0)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        eng.set_option(getattr(nat, k), int(v))
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = eng.dev_alloc(stack.nbytes)
    eng.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    for _ in range(60):  # (the sustained clock state: the first ~15 ms of device activity run slower)
        eng.clear()
        eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
        eng.flush()
    eng.synchronize()
    lib = ctypes.CDLL(nat.LIB_PATH)
    tr = np.zeros((8192, 8), dtype=np.uint32)
    rc = lib.sc_debug_dense_trace(ctypes.c_void_p(tr.ctypes.data))
    assert rc == 0, rc
    tr_all = tr
    tr = tr[:4096][tr[:4096, 1] != 0]
    w = tr[:4096] if len(tr) >= 4096 else tr
    t0 = w[:, 0].astype(np.int64); t1 = w[:, 1].astype(np.int64)
    base = t0.min()
    dur = (t1 - t0) * 0.01  # us
    out = {"wavefronts": int(len(w)), "start_us_pct": np.percentile((t0 - base) * 0.01, [0, 50, 90, 99, 100]).round(2).tolist(),
           "end_us_pct": np.percentile((t1 - base) * 0.01, [0, 10, 50, 90, 99, 100]).round(2).tolist(),
           "busy_us_pct": np.percentile(dur, [0, 50, 90, 99, 100]).round(2).tolist(),
           "bricks_per_wf_hist": np.bincount(w[:, 2].astype(np.int64), minlength=8)[:12].tolist(),
           "units_per_wf_hist": np.bincount(w[:, 3].astype(np.int64), minlength=8)[:20].tolist(),
           "bricks": int(w[:, 2].sum()), "units": int(w[:, 3].sum()), "mean_busy_us": round(float(dur.mean()), 2),
           "us_per_ticket": round(float(w[:, 6].sum()) * 0.01 / max(1, int(w[:, 2].sum())), 3),
           "us_per_verdict": round(float(w[:, 4].sum()) * 0.01 / max(1, int(w[:, 2].sum())), 3),
           "us_per_unit": round(float(w[:, 5].sum()) * 0.01 / max(1, int(w[:, 3].sum())), 3)}
    order = np.argsort(t1)[-5:]
    out["last_finishers"] = [{"bricks": int(w[i, 2]), "units": int(w[i, 3]), "start": round((t0[i] - base) * 0.01, 2),
                              "end": round((t1[i] - base) * 0.01, 2)} for i in order]
    if tr_all[8191, 2]:
        out["riders"] = {"blocks": int(tr_all[8191, 2]), "last_end_us": round((int(tr_all[8191, 1]) - int(base)) * 0.01, 2)}
    u = tr_all[4096:8191]
    u = u[u[:, 1] != 0]
    if len(u):
        t0 = u[:, 0].astype(np.int64); t1 = u[:, 1].astype(np.int64)
        b = t0.min()
        nun = int(u[:, 2].sum())
        out["unit_verdict_kernel"] = {"wavefronts": int(len(u)), "units": nun,
                                      "end_us_pct": np.percentile((t1 - b) * 0.01, [0, 50, 90, 100]).round(2).tolist(),
                                      "units_per_wf_pct": np.percentile(u[:, 2], [0, 50, 100]).tolist(),
                                      "us_fetching_the_unit_id": round(float(u[:, 4].sum()) * 0.01 / max(1, nun), 3),
                                      "us_per_unit_after_that": round(float(u[:, 5].sum()) * 0.01 / max(1, nun), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
