import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from plant3dvision_amd import _native as nat, scenes
shape, origin, vs, views = scenes.make_scene(512, 72, os.environ.get("SCENE", "plant"))
eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
ptr = eng.dev_alloc(stack.nbytes); eng.dev_upload(ptr, stack)
K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
V, H, W = stack.shape
for i in range(3):
    eng.clear(); eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
    buf = eng.get_values_sparse()
    print("header", buf[:64].view(np.uint32).tolist(), "counts", list(eng.fused_counts_ex()), flush=True)
