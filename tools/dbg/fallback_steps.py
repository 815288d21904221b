"""Where a step of the STAGED fallback (torch.distributed gloo, through the hosts) spends its time: one GPU, gloo group of one."""
import sys, time, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch, torch.distributed as dist
from plant3dvision_amd import _native as nat, scenes
from plant3dvision_amd.sharded import ShardedBackprojection
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
shape, origin, vs, views = scenes.make_scene(512, 72, "plant")
sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, device=0)
sb.force_collective = True
eng = sb.engine
stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
ptr = eng.dev_alloc(stack.nbytes); eng.dev_upload(ptr, stack)
K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
V, H, W = stack.shape
def step():
    eng.clear(); eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
    return sb.all_gather(compress="sparse", unpack=False, overlap=True, check=False)
for i in range(3): step().verify()
t0 = time.perf_counter()
for i in range(10): g = step(); g.verify()
print("ms/step", (time.perf_counter() - t0) / 10 * 1e3, flush=True)
pr = cProfile.Profile(); pr.enable()
for i in range(10): g = step(); g.verify()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
