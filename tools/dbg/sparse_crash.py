import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from plant3dvision_amd import _native as nat, scenes
shape = (24, 48, 192)
_, origin, vs, views = scenes.make_scene(shape, 10, "plant")
eng = nat.Engine(list(shape), origin, vs, nat.SC_MODE_CARVE)
for K, R, t, m in views:
    eng.process_view(K, R, t, m, nat.SC_MASK_U8)
lab = eng.get_values()
print("labels ok", np.unique(lab, return_counts=True), flush=True)
nb = nat.sparse_bricks(*shape)
ptr, nbytes = eng.values_sparse(nb)
print("values_sparse", hex(ptr), nbytes, nat.sparse_rank_bytes(nb, nb), flush=True)
eng.synchronize()
out = np.zeros(nbytes, dtype=np.uint8)
eng.dev_download(out, ptr)
print("download ok", out[:64].view(np.uint32), flush=True)
buf = eng.get_values_sparse(nb)
print("sparse ok", buf[:64].view(np.uint32), flush=True)
