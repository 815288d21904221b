#!/usr/bin/env python3
"""Per-rank fused carve time of the N-GPU weak-scaling workload, measured one slab after the
other on ONE GPU (what the slowest rank of an N-GPU run would take).  Diagnostic."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes
import bench

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mode = sys.argv[2] if len(sys.argv) > 2 else "slab"
    shape = bench.global_shape(512, N)
    gshape, origin, vs, views = scenes.make_scene(tuple(shape), 72, "plant")
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    out = []
    for r in range(N):
        nx = gshape[0]
        i0, i1 = nx * r // N, nx * (r + 1) // N
        if mode == "slab":
            e = nat.Engine(gshape, origin, vs, nat.SC_MODE_CARVE, slab=(i0, i1))
        else:
            e = nat.Engine(gshape, origin, vs, nat.SC_MODE_CARVE, cyclic=(r, N))
        ptr = e.dev_alloc(stack.nbytes); e.dev_upload(ptr, stack)
        for it in range(6):
            if it == 2:
                e.synchronize(); t0 = time.perf_counter()
            e.clear(); e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8); e.flush()
        e.synchronize(); dt = (time.perf_counter() - t0) / 4
        kept = int((e.get_values() == 1).sum())
        out.append((r, round(dt * 1e3, 4), kept))
        e.dev_free(ptr); e.close()
    ts = [o[1] for o in out]
    print(json.dumps({"N": N, "grid": gshape, "per_rank_ms": out, "max_ms": max(ts), "mean_ms": sum(ts) / len(ts)}))

if __name__ == "__main__":
    main()
