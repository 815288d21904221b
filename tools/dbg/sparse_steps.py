"""Where a carve + sparse-assembly step spends its time (one GPU, RCCL group of one): variants A..E, host clock."""
import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from plant3dvision_amd import _native as nat, scenes
from plant3dvision_amd.sharded import ShardedBackprojection
if os.environ.get("TORCHDIST"):
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    if os.environ["TORCHDIST"] == "2":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        dist.barrier()
n = int(os.environ.get("N", "512"))
shape, origin, vs, views = scenes.make_scene(n, 72, os.environ.get("SCENE", "plant"))
sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, device=0)
sb.force_collective = True
comm = sb.init_comm(nat.Comm.unique_id())
eng = sb.engine
stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
ptr = eng.dev_alloc(stack.nbytes); eng.dev_upload(ptr, stack)
K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
V, H, W = stack.shape
def batch():
    eng.clear(); eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
def run(name, step, fin, steps=40):
    for i in range(5): step(i)
    fin(); 
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    t1 = time.perf_counter()
    fin()
    t2 = time.perf_counter()
    print(f"{name:40s} {(t2-t0)/steps*1e3:.4f} ms/step   (enqueue {(t1-t0)/steps*1e3:.4f})", flush=True)
def fin():
    eng.synchronize(); comm.synchronize()
batch(); sb.all_gather(compress="sparse", unpack=False)
print("cap", sb._sparse_cap, "stride", sb.sparse_rank_bytes(), flush=True)
VAR = os.environ.get("VARIANTS", "ABCDEFGH")
if "A" in VAR: run("A carve only", lambda i: (batch(), eng.flush()), fin)
if "B" in VAR: run("B carve + values_sparse (pack)", lambda i: (batch(), eng.values_sparse(sb._sparse_cap)), fin)
def stepC(i, ov=False, ver=False, st={}):
    batch()
    g = sb.all_gather(compress="sparse", unpack=False, overlap=ov, check=False)
    if ver and st.get("p") is not None: st["p"].verify()
    st["p"] = g
if "C" in VAR: run("C + all_gather serial, no verify", lambda i: stepC(i), fin)
if "D" in VAR: run("D + all_gather overlap, no verify", lambda i: stepC(i, True), fin)
if "E" in VAR: run("E + all_gather overlap, verify prev", lambda i: stepC(i, True, True), fin)
if "F" in VAR: run("F + all_gather serial, verify prev", lambda i: stepC(i, False, True), fin)
# raw library calls without the Python class
stride = sb.sparse_rank_bytes(); cap = sb._sparse_cap
from plant3dvision_amd.sharded import DevMem
rb = [DevMem(eng, stride), DevMem(eng, stride)]
if "G" in VAR: run("G raw eng.all_gather_sparse overlap", lambda i: (batch(), eng.all_gather_sparse(comm, cap, rb[i & 1].ptr, stride, overlap=True)), fin)
if "H" in VAR: run("H raw comm.all_gather only (4 MB)", lambda i: comm.all_gather(rb[0].ptr, rb[1].ptr, stride), fin)
if "T" in VAR:
    # two engines take turns (double buffering at the engine's grain): engine B carves step k + 1 while engine A's labels
    # of step k are packed and gathered -- the pack is on A's stream, nobody's next carve waits for it
    sb2 = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, device=0)
    sb2.force_collective = True
    sb2.comm = comm
    sbs = (sb, sb2)
    def batch_on(e):
        e.clear(); e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
    for q in sbs:
        batch_on(q.engine); q.all_gather(compress="sparse", unpack=False)
    stT = {}
    def stepT(i, ver=True):
        q = sbs[i & 1]
        batch_on(q.engine)
        g = q.all_gather(compress="sparse", unpack=False, overlap=True, check=False)
        if ver and stT.get("p") is not None: stT["p"].verify()
        stT["p"] = g
    def finT():
        for q in sbs: q.engine.synchronize()
        comm.synchronize()
    run("T two engines in turn, overlap, verify prev", stepT, finT)
    run("T' the same without verify", lambda i: stepT(i, False), finT)
    def stepA2(i):
        q = sbs[i & 1]; batch_on(q.engine); q.engine.flush()
    run("A2 carve only, two engines in turn", stepA2, finT)
    if "C" in VAR: run("C again, twin alive", lambda i: stepC(i), fin)
    sb2.comm = None
    sb2.close()
    if "C" in VAR: run("C again, after the twin is closed", lambda i: stepC(i), fin)
    if "F" in VAR: run("F again, after the twin is closed", lambda i: stepC(i, False, True), fin)
    if "Q" in VAR:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for i in range(100): stepC(i)
        pr.disable(); fin()
        pstats.Stats(pr).sort_stats("tottime").print_stats(8)
        import ctypes
        t0 = time.perf_counter()
        for i in range(100):
            batch(); eng.flush()
        t1 = time.perf_counter(); fin()
        print("carve only enqueue after twin", (t1 - t0) / 100 * 1e3)
        for name in ("values_sparse",):
            t0 = time.perf_counter()
            for i in range(100):
                batch(); eng.values_sparse(sb._sparse_cap)
            t1 = time.perf_counter(); fin()
            print("carve + values_sparse enqueue", (t1 - t0) / 100 * 1e3, "total", (time.perf_counter() - t0) / 100 * 1e3)
if "P" in VAR:
    import cProfile, pstats
    st = {}
    pr = cProfile.Profile()
    for i in range(5): stepC(i, True, True, st)
    fin()
    pr.enable()
    for i in range(200): stepC(i, True, True, st)
    pr.disable()
    fin()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
