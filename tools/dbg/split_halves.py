"""One batch as TWO half-grid engines on two streams of the same GPU (x-plane slabs or the plane-cyclic deal): do the halves
fill each other's kernel boundaries?  Host clock around K steps, everything waited for."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from plant3dvision_amd import _native as nat, scenes
shape, origin, vs, views = scenes.make_scene(512, 72, os.environ.get("SCENE", "plant"))
stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
V, H, W = stack.shape
def make(kw):
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, **kw)
    return e
full = make({})
ptr = full.dev_alloc(stack.nbytes); full.dev_upload(ptr, stack)
def run(name, engines, steps=60):
    def step():
        for e in engines:
            e.clear(); e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8); e.flush()
    for _ in range(5): step()
    for e in engines: e.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    for e in engines: e.synchronize()
    print(f"{name:44s} {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step", flush=True)
run("one engine, the whole grid", [full])
for parts in (2, 4):
    sl = [make({"slab": (512 * r // parts, 512 * (r + 1) // parts)}) for r in range(parts)]
    run(f"{parts} engines, slabs", sl)
    for e in sl: e.close()
    cy = [make({"cyclic": (r, parts)}) for r in range(parts)]
    run(f"{parts} engines, planes dealt cyclically", cy)
    got = np.empty(shape, dtype=np.int32)
    for r, e in enumerate(cy): got[r::parts] = e.get_values()
    print("   equal to the one engine's labels:", bool(np.array_equal(got, full.get_values())))
    for e in cy: e.close()
run("one engine again", [full])
