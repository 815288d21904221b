"""N>1 path: slab partition + grid assembly.  CPU: world_size-2/3 gloo process groups with
the oracle standing in for the device (host logic only).  GPU: the same class over the HIP
engine in one process (world_size 1) -- multi-GPU runs belong to the driver."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_c
from plant3dvision_amd.sharded import ShardedBackprojection, rank_planes, slab_bounds
from tests.helpers import OracleEngine, scene, unpack_labels_np


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_slab_bounds_cover_the_axis_exactly():
    for nx, w in ((512, 8), (301, 8), (7, 7), (10, 3), (1024, 8)):
        b = [slab_bounds(nx, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == nx
        assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))
        sizes = [i1 - i0 for i0, i1 in b]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    with pytest.raises(ValueError):
        slab_bounds(3, 4, 0)


def test_rank_planes_partition_the_axis():
    for part in ("cyclic", "slab"):
        for nx, w in ((512, 8), (301, 8), (7, 7), (10, 3)):
            allp = sorted(i for r in range(w) for i in rank_planes(nx, w, r, part))
            assert allp == list(range(nx)), (part, nx, w)
            sizes = [len(rank_planes(nx, w, r, part)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
    assert list(rank_planes(10, 3, 1, "cyclic")) == [1, 4, 7]
    with pytest.raises(ValueError):
        rank_planes(10, 3, 0, "blocks")


def _worker(rank, world, port, shape, mode, q, partition="cyclic"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, origin, vs, views = scene(tuple(shape), 5, "plant")
        if mode == "averaging":
            rng = np.random.default_rng(1)
            views = [(K, R, t, rng.random(m.shape, dtype=np.float32)) for K, R, t, m in views]
        sb = ShardedBackprojection(shape, origin, vs, type=mode, engine_factory=OracleEngine,
                                   partition=partition, unpack_fn=unpack_labels_np)
        assert (sb.rank, sb.world_size) == (rank, world)
        for K, R, t, m in views:
            sb.process_view(K, R, t, m)
        full_ag = sb.all_gather().numpy()
        full_ar = sb.all_reduce().numpy()
        host = sb.gather_to_host(dst=0)
        res = {"rank": rank, "planes": list(sb.planes), "ag": full_ag, "ar": full_ar, "host": host}
        if mode == "carving":
            res["ag8"] = sb.all_gather(compress=True).numpy()
            narrow = sb.all_gather(compress=True, widen=False)
            assert narrow.dtype == torch.int8
            res["ag8n"] = narrow.numpy()
            # reusable buffers: the second call lands in the same memory
            pad = sb._planes_max() * shape[1] * shape[2]
            recv = torch.empty(pad * world, dtype=torch.int32)
            out = torch.empty(pad * world, dtype=torch.int32)
            a = sb.all_gather(recv=recv, out=out)
            b = sb.all_gather(recv=recv, out=out)
            assert a.data_ptr() == b.data_ptr() and torch.equal(a, b)
            res["host32"] = sb.gather_to_host(dst=world - 1, compress=False)
            # labels at 2 bits over the wire (1 bit: the occupancy), unpacked into global order
            p2 = sb.all_gather(compress="2bit")
            assert p2.dtype == torch.int32
            res["ag2"] = p2.numpy()
            p2n = sb.all_gather(compress="2bit", widen=False)
            assert p2n.dtype == torch.int8
            res["ag2n"] = p2n.numpy()
            res["ag1"] = sb.all_gather(compress="1bit", widen=False).numpy()
            rb = sb.packed_rank_bytes(2)
            recv2 = torch.empty(rb * world, dtype=torch.uint8)
            out2 = torch.empty(int(np.prod(shape)), dtype=torch.int8)
            a2 = sb.all_gather(compress="2bit", widen=False, recv=recv2, out=out2)
            assert a2.data_ptr() == out2.data_ptr()
            # the brick-sparse form (round 6): codes + the mixed bricks' labels only; bit-equal to the dense 2-bit form
            ps = sb.all_gather(compress="sparse")
            assert ps.dtype == torch.int32
            res["ags"] = ps.numpy()
            res["agsn"] = sb.all_gather(compress="sparse", widen=False).numpy()
            res["hosts"] = sb.gather_to_host(dst=0, compress="sparse")
            # a capacity too small for some rank: every rank sees the same headers and gathers again with more slots
            sb._sparse_cap = 16
            res["ags_small_cap"] = sb.all_gather(compress="sparse").numpy()
            res["cap_after"] = sb._sparse_cap
            # a twin engine on the same planes (the double buffering of a pipeline of scans): the two take gathers in
            # turn, on every rank alike, and each assembles the whole grid
            tw = sb.twin()
            assert tw.planes == sb.planes and tw.comm is sb.comm and tw._sparse_cap == sb._sparse_cap
            for K, R, t, m in views:
                tw.process_view(K, R, t, m)
            first = sb.all_gather(compress="sparse").numpy().copy()
            res["ags_twin"] = tw.all_gather(compress="sparse").numpy().copy()
            assert np.array_equal(first, sb.all_gather(compress="sparse").numpy())
            tw.close()
        q.put(res)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,mode,partition", [(2, (16, 10, 12), "carving", "cyclic"),
                                                        (3, (17, 9, 8), "carving", "cyclic"),
                                                        (3, (17, 9, 8), "carving", "slab"),
                                                        (2, (16, 10, 12), "carving", "slab"),
                                                        (3, (11, 8, 16), "carving", "cyclic"),   # planes of whole packed words
                                                        (2, (10, 16, 32), "carving", "slab"),
                                                        (2, (9, 8, 12), "averaging", "cyclic")])
def test_gloo_sharded_equals_single(world, shape, mode, partition):
    _, origin, vs, views = scene(tuple(shape), 5, "plant")
    if mode == "carving":
        want = oracle_c.carve(list(shape), origin, vs, views)
    else:
        rng = np.random.default_rng(1)
        fviews = [(K, R, t, rng.random(m.shape, dtype=np.float32)) for K, R, t, m in views]
        want = oracle_c.average(list(shape), origin, vs, fviews)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, list(shape), mode, q, partition)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for res in results:
        assert np.array_equal(res["ag"], want)
        assert np.array_equal(res["ar"], want)
        if mode == "carving":
            assert np.array_equal(res["ag8"], want)
            assert res["ag8n"].dtype == np.int8 and np.array_equal(res["ag8n"], want)
            assert np.array_equal(res["ag2"], want) and np.array_equal(res["ag2n"], want)
            assert np.array_equal(res["ag1"], (want == 1).astype(np.int8))
            assert np.array_equal(res["ags"], want) and res["agsn"].dtype == np.int8 and np.array_equal(res["agsn"], want)
            assert np.array_equal(res["ags_small_cap"], want) and res["cap_after"] >= 16
            assert np.array_equal(res["ags_twin"], want)
            if res["rank"] == 0:
                assert res["hosts"].dtype == np.int32 and np.array_equal(res["hosts"], want)
            else:
                assert res["hosts"] is None
            if res["rank"] == world - 1:
                assert res["host32"].dtype == np.int32 and np.array_equal(res["host32"], want)
            else:
                assert res["host32"] is None
        if res["rank"] == 0:
            assert res["host"].dtype == want.dtype and np.array_equal(res["host"], want)
        else:
            assert res["host"] is None


@pytest.mark.gpu
def test_sharded_class_over_hip_engine_single_rank(gpu_device):
    shape, origin, vs, views = scene((40, 24, 32), 6, "plant")
    want = oracle_c.carve(shape, origin, vs, views)
    sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, device=0)
    for K, R, t, m in views:
        sb.process_view(K, R, t, m)
    assert np.array_equal(sb.get_local(), want)
    assert np.array_equal(sb.gather_to_host(), want)
    # zero-copy view of the engine's device memory through torch (what RCCL would send)
    assert np.array_equal(sb.all_gather().cpu().numpy(), want)
    assert np.array_equal(sb.all_reduce().cpu().numpy(), want)
    sb.close()


def _gpu_worker(rank, world, port, shape, partition, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, origin, vs, views = scene(tuple(shape), 8, "plant")
        sb = ShardedBackprojection(shape, origin, vs, rank=rank, world_size=world, device=0, partition=partition)
        for K, R, t, m in views:
            sb.process_view(K, R, t, m)
        full = sb.all_gather()
        assert full.is_cuda and full.dtype == torch.int32
        narrow = sb.all_gather(compress=True, widen=False)
        assert narrow.is_cuda and narrow.dtype == torch.int8
        p2 = sb.all_gather(compress="2bit", widen=False)
        assert p2.is_cuda and p2.dtype == torch.int8
        p2w = sb.all_gather(compress="2bit")
        p1 = sb.all_gather(compress="1bit", widen=False)
        # the assembled grid in its packed form, read by vol2pcd as it is (no full-size grid is written)
        from plant3dvision_amd import proc3d
        clouds = {}
        for comp in ("1bit", "2bit"):
            pg = sb.all_gather(compress=comp, unpack=False)
            assert pg.recv.is_cuda and pg.bits == (1 if comp == "1bit" else 2)
            assert np.array_equal(pg.unpack().cpu().numpy(), p1.cpu().numpy() if comp == "1bit" else p2.cpu().numpy())
            pc = proc3d.vol2pcd(pg, origin, vs, 0.0, as_open3d=False)
            clouds[comp] = (np.asarray(pc.points), np.asarray(pc.normals))
        # the pipeline bench.py times at N > 1: batches back to back, the collective of one beside the carve of the
        # next (overlap=True: the engine waits for it only before it packs again), alternating receive buffers
        recv = [torch.empty(sb.packed_rank_bytes(2) * world, dtype=torch.uint8, device=full.device) for _ in range(2)]
        grids = []
        for i in range(3):
            sb.clear()
            for K, R, t, m in (views if i != 1 else views[:5]):  # the middle batch is a different volume
                sb.process_view(K, R, t, m)
            grids.append(sb.all_gather(compress="2bit", recv=recv[i & 1], unpack=False, overlap=True))
        sb.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(grids[2].unpack().cpu().numpy(), p2.cpu().numpy())          # the last batch: all the views again
        assert np.array_equal(grids[0].unpack().cpu().numpy(), p2.cpu().numpy())          # recv[0] was rewritten by batch 2: same grid
        assert not np.array_equal(grids[1].unpack().cpu().numpy(), p2.cpu().numpy())      # recv[1] holds the 5-view volume
        ref = proc3d.vol2pcd((full.cpu().numpy() == 1).astype(np.uint8), origin, vs, 0.0, as_open3d=False)
        for comp, (pts, nrm) in clouds.items():
            assert np.array_equal(pts, np.asarray(ref.points)) and np.array_equal(nrm, np.asarray(ref.normals)), comp
        assert len(ref.points) > 0
        q.put({"rank": rank, "ag": full.cpu().numpy(), "ag8n": narrow.cpu().numpy(),
               "ag2n": p2.cpu().numpy(), "ag2": p2w.cpu().numpy(), "ag1": p1.cpu().numpy(),
               "ar": sb.all_reduce().cpu().numpy(), "host": sb.gather_to_host(dst=0)})
        sb.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("partition,shape", [("cyclic", (37, 48, 64)), ("slab", (36, 32, 128)), ("cyclic", (21, 30, 75))])
def test_two_ranks_over_hip_engines_sharing_the_gpu(gpu_device, partition, shape):
    """world_size 2 with the HIP engine behind both ranks (one GPU, gloo as the transport): the
    device branches of all_gather / all_reduce / gather_to_host that RCCL would run on 8 GPUs."""
    _, origin, vs, views = scene(tuple(shape), 8, "plant")
    want = oracle_c.carve(list(shape), origin, vs, views)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, list(shape), partition, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for res in results:
        assert np.array_equal(res["ag"], want) and np.array_equal(res["ar"], want)
        assert res["ag8n"].dtype == np.int8 and np.array_equal(res["ag8n"], want)
        assert np.array_equal(res["ag2n"], want) and res["ag2"].dtype == np.int32 and np.array_equal(res["ag2"], want)
        assert np.array_equal(res["ag1"], (want == 1).astype(np.int8))
        if res["rank"] == 0:
            assert res["host"].dtype == np.int32 and np.array_equal(res["host"], want)
        else:
            assert res["host"] is None


@pytest.mark.gpu
@pytest.mark.parametrize("nz", [128, 70])
@pytest.mark.parametrize("partition,ndev", [("cyclic", 3), ("slab", 2), ("cyclic", 1)])
def test_several_engines_from_one_process(gpu_device, partition, ndev, nz):
    """sc_create_sharded: one engine per listed device (here the one GPU, several times), x-planes dealt
    as the ranks of the multi-process path get them, the grid read back whole in global order."""
    from plant3dvision_amd import _native as nat
    from plant3dvision_amd.cl import Backprojection, img_as_float32
    shape, origin, vs, views = scene((37, 48, nz), 9, "plant")  # nz = 70: padded rows on the device
    want = oracle_c.carve(list(shape), origin, vs, views, nthreads=4)
    g = nat.EngineGroup(shape, origin, vs, nat.SC_MODE_CARVE, [0] * ndev, partition=partition)
    for vpl in (0, 1):
        g.clear()
        g.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
        for K, R, t, m in views:
            g.process_view(K, R, t, m, nat.SC_MASK_U8)
        assert np.array_equal(g.get_values(), want), (partition, ndev, vpl)
    g.close()
    # the class with a list of devices; averaging through the table path
    bp = Backprojection(shape, origin, vs, device=[0] * ndev)
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    assert np.array_equal(bp.get_values(), want)
    bp.close()
    wantf = oracle_c.average(list(shape), origin, vs, [(K, R, t, img_as_float32(m)) for K, R, t, m in views])
    bp = Backprojection(shape, origin, vs, type="averaging", device=[0] * ndev)
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    assert np.array_equal(bp.get_values().view(np.uint32), wantf.view(np.uint32))
    bp.close()
    with pytest.raises(ValueError):
        nat.EngineGroup(shape, origin, vs, nat.SC_MODE_CARVE, [])


@pytest.mark.gpu
def test_collectives_through_rccl_with_a_process_group_of_one(gpu_device):
    """The calls an 8-GPU run makes -- all_gather_into_tensor / all_reduce / gather on device tensors that
    alias the engine's memory, RCCL as the backend -- on the one GPU there is, with a group of one rank:
    same buffers, same stream ordering, nothing moved over xGMI."""
    port = _free_port()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        for partition, shape in (("cyclic", (37, 48, 64)), ("slab", (36, 32, 128))):
            _, origin, vs, views = scene(tuple(shape), 8, "plant")
            want = oracle_c.carve(list(shape), origin, vs, views)
            sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, device=0, partition=partition)
            sb.force_collective = True
            for K, R, t, m in views:
                sb.process_view(K, R, t, m)
            full = sb.all_gather()
            assert full.is_cuda and full.dtype == torch.int32 and np.array_equal(full.cpu().numpy(), want)
            narrow = sb.all_gather(compress=True, widen=False)
            assert narrow.dtype == torch.int8 and np.array_equal(narrow.cpu().numpy(), want)
            p2 = sb.all_gather(compress="2bit", widen=False)
            assert p2.is_cuda and p2.dtype == torch.int8 and np.array_equal(p2.cpu().numpy(), want)
            p2w = sb.all_gather(compress="2bit")
            assert p2w.dtype == torch.int32 and np.array_equal(p2w.cpu().numpy(), want)
            p1 = sb.all_gather(compress="1bit", widen=False)
            assert np.array_equal(p1.cpu().numpy(), (want == 1).astype(np.int8))
            assert np.array_equal(sb.all_reduce().cpu().numpy(), want)
            host = sb.gather_to_host(dst=0)  # the 2-bit wire + sc_widen_labels2_ranks
            assert host.dtype == np.int32 and np.array_equal(host, want)
            host8 = sb.gather_to_host(dst=0, compress=True)  # the int8 route of rounds 2-3
            assert host8.dtype == np.int32 and np.array_equal(host8, want)
            from plant3dvision_amd import proc3d
            pg = sb.all_gather(compress="1bit", unpack=False)
            pc = proc3d.vol2pcd(pg, origin, vs, 0.0, as_open3d=False)
            ref = proc3d.vol2pcd((want == 1).astype(np.uint8), origin, vs, 0.0, as_open3d=False)
            assert len(ref.points) > 0 and np.array_equal(np.asarray(pc.points), np.asarray(ref.points))
            assert np.array_equal(np.asarray(pc.normals), np.asarray(ref.normals))
            dist.barrier()
            sb.close()
    finally:
        dist.destroy_process_group()
