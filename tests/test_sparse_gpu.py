"""GPU: the brick-sparse transport form of carve labels (round 6; include/spacecarve.h, csrc/sc_sparse.h) -- packed from
a batch's verdict bytes and live list, bit-equal to the labels (and so to the dense 2-bit form) whatever the labels'
history; the library's own RCCL communicator (a group of one on the one-GPU box: the calls, buffers and stream order of
an 8-GPU run)."""
import numpy as np
import pytest

from oracle import oracle_c
from plant3dvision_amd import _native as nat
from plant3dvision_amd import proc3d, scenes
from plant3dvision_amd.sharded import ShardedBackprojection, SparseGrid, SparseOverflow
from tests.helpers import scene, sparse_header_np, unpack_sparse_np

pytestmark = pytest.mark.gpu


def _poses(views):
    return (np.stack([v[0] for v in views]), np.stack([v[1] for v in views]), np.stack([v[2] for v in views]))


def _sparse_equals_labels(eng, shape, first=0, stride=1, cap=0):
    lab = eng.get_values()
    buf = eng.get_values_sparse(cap)
    h = sparse_header_np(buf)
    assert h["magic"] == 0x50534353 and h["planes"] == lab.shape[0] and (h["first"], h["stride"]) == (first, stride)
    assert h["nmixed"] <= h["cap"], h
    full_shape = (first + (lab.shape[0] - 1) * stride + 1, lab.shape[1], lab.shape[2])
    got = unpack_sparse_np(buf, buf.size, 1, full_shape)
    assert np.array_equal(got[first::stride], lab)
    return h, lab


@pytest.mark.parametrize("kind", ["plant", "dense", "solid", "noise"])
@pytest.mark.parametrize("shape", [(24, 48, 192), (13, 37, 150), (5, 16, 64)])
def test_sparse_form_equals_the_labels_whatever_their_history(gpu_device, kind, shape):
    _, origin, vs, views = scenes.make_scene(shape, 10, kind)
    want = oracle_c.carve(list(shape), origin, vs, views, nthreads=4)
    K, R, t = _poses(views)
    eng = nat.Engine(list(shape), origin, vs, nat.SC_MODE_CARVE)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = eng.dev_alloc(stack.nbytes)
    eng.dev_upload(ptr, stack)
    V, H, W = stack.shape
    nbricks = nat.sparse_bricks(*shape)
    # (1) ONE fused batch on a cleared volume: verdict bytes + live list (one launch)
    eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
    h, lab = _sparse_equals_labels(eng, shape, cap=nbricks)
    assert np.array_equal(lab, want)
    assert h["nmixed"] <= nbricks
    if kind == "solid":
        assert h["nmixed"] == 0  # every brick kept whole or untouched: codes only
    # twice in a row (the two send buffers alternate), the same answer
    _sparse_equals_labels(eng, shape, cap=nbricks)
    # (2) one launch per view (the reference's cadence): dead bytes + the rest read
    eng.clear()
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 1)
    eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
    _, lab = _sparse_equals_labels(eng, shape, cap=nbricks)
    assert np.array_equal(lab, want)
    # (3) two batches one after the other (the second on a volume that is not fresh), and the streaming kernel
    eng.clear()
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 0)
    eng.process_views_device(K[:6], R[:6], t[:6], ptr, 6, H, W, nat.SC_MASK_U8)
    eng.flush()
    eng.process_views_device(K[6:], R[6:], t[6:], ptr + 6 * H * W, V - 6, H, W, nat.SC_MASK_U8)
    _, lab = _sparse_equals_labels(eng, shape, cap=nbricks)
    assert np.array_equal(lab, want)
    eng.clear()
    eng.set_option(nat.SC_OPT_VIEW_BRICK, 0)
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 1)
    eng.process_views_device(K[:3], R[:3], t[:3], ptr, 3, H, W, nat.SC_MASK_U8)
    _sparse_equals_labels(eng, shape, cap=nbricks)
    # (4) a volume no view has touched
    eng.clear()
    h, lab = _sparse_equals_labels(eng, shape, cap=nbricks)
    assert h["nmixed"] == 0 and (lab == 0).all()
    eng.dev_free(ptr)
    eng.close()


@pytest.mark.parametrize("default", [-1, 0, 1])
def test_sparse_form_other_default_values_and_rank_planes(gpu_device, default):
    shape, origin, vs, views = scene((21, 40, 130), 9, "plant")
    for kw, first, stride in (({"cyclic": (1, 3)}, 1, 3), ({"slab": (4, 15)}, 4, 1)):
        eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=float(default), **kw)
        for K, R, t, m in views:
            eng.process_view(K, R, t, m, nat.SC_MASK_U8)
        h, lab = _sparse_equals_labels(eng, shape, first=first, stride=stride, cap=4096)
        planes = list(range(first, shape[0], stride)) if stride > 1 else list(range(4, 15))
        assert np.array_equal(lab, oracle_c.carve(shape, origin, vs, views, default_value=default, nthreads=4)[planes])
        eng.close()
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=5.0)
    with pytest.raises(nat.SpaceCarveError):
        eng.get_values_sparse()
    eng.close()


def test_sparse_form_says_when_slots_ran_out_and_codes_stay_right(gpu_device):
    shape, origin, vs, views = scene((16, 64, 256), 8, "noise")
    # noise after few views: many mixed bricks
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    for K, R, t, m in views[:2]:
        eng.process_view(K, R, t, m, nat.SC_MASK_U8)
    lab = eng.get_values()
    full = eng.get_values_sparse(nat.sparse_bricks(*shape))
    hf = sparse_header_np(full)
    assert hf["nmixed"] > 32
    small = eng.get_values_sparse(16)
    hs = sparse_header_np(small)
    assert hs["cap"] == 16 and hs["nmixed"] == hf["nmixed"] > hs["cap"]
    nb = hf["nbricks"]
    assert np.array_equal(small[64:64 + nb], full[64:64 + nb])  # the codes are complete either way
    with pytest.raises(nat.SpaceCarveError):
        nat.widen_sparse_ranks(small, small.size, 1, shape)
    assert np.array_equal(nat.widen_sparse_ranks(full, full.size, 1, shape), lab)
    eng.close()


@pytest.mark.parametrize("kind", ["plant", "dense"])
def test_sparse_unpack_on_the_device_three_output_kinds(gpu_device, kind):
    import torch
    shape, origin, vs, views = scene((20, 50, 140), 10, kind)
    want = oracle_c.carve(shape, origin, vs, views, nthreads=4)
    sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, device=0)
    for K, R, t, m in views:
        sb.process_view(K, R, t, m)
    g32 = sb.all_gather(compress="sparse")
    assert g32.is_cuda and g32.dtype == torch.int32 and np.array_equal(g32.cpu().numpy(), want)
    g8 = sb.all_gather(compress="sparse", widen=False)
    assert g8.dtype == torch.int8 and np.array_equal(g8.cpu().numpy(), want)
    assert np.array_equal(g8.cpu().numpy(), sb.all_gather(compress="2bit", widen=False).cpu().numpy())
    grid = sb.all_gather(compress="sparse", unpack=False)
    assert isinstance(grid, SparseGrid)  # (its size against the dense form: the 512^3 test below)
    occ = grid.unpack(kind=0)
    assert occ.dtype == torch.uint8 and np.array_equal(occ.cpu().numpy(), (want == 1).astype(np.uint8))
    assert np.array_equal(grid.to_host(), want)
    assert np.array_equal(sb.gather_to_host(compress="sparse"), want)
    # vol2pcd reads the sparse grid's occupancy in place: the same cloud as from the labels
    pc = proc3d.vol2pcd(grid, origin, vs, 0.0, as_open3d=False)
    ref = proc3d.vol2pcd((want == 1).astype(np.uint8), origin, vs, 0.0, as_open3d=False)
    assert len(ref.points) > 0 and np.array_equal(np.asarray(pc.points), np.asarray(ref.points))
    assert np.array_equal(np.asarray(pc.normals), np.asarray(ref.normals))
    sb.close()


def test_library_rccl_group_of_one_sparse_and_dense_with_overlap(gpu_device):
    """RCCL bound behind the C ABI (sc_comm_*): no torch in the data path.  A communicator of ONE rank on the one-GPU box
    goes through ncclCommInitRank / ncclAllGather like eight would."""
    shape, origin, vs, views = scene((32, 64, 192), 10, "plant")
    want = oracle_c.carve(shape, origin, vs, views, nthreads=4)
    sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, device=0)
    sb.force_collective = True
    comm = sb.init_comm(nat.Comm.unique_id())
    assert comm.world_size == 1 and sb.init_comm() is comm
    eng = sb.engine
    for K, R, t, m in views:
        sb.process_view(K, R, t, m)
    grid = sb.all_gather(compress="sparse", unpack=False)
    assert isinstance(grid, SparseGrid) and not hasattr(grid.recv, "is_cuda")  # a DevMem, not a tensor
    assert np.array_equal(grid.to_host(), want)
    assert np.array_equal(sb.gather_to_host(compress="sparse"), want)
    # a capacity that is too small: found from the headers, gathered again
    sb._sparse_cap = 16
    assert np.array_equal(sb.all_gather(compress="sparse", unpack=False).to_host(), want)
    assert sb._sparse_cap > 16
    # the pipeline bench.py times at N > 1: the collective of batch k beside the carve of batch k + 1
    grids = []
    for i in range(4):
        sb.clear()
        for K, R, t, m in (views if i != 2 else views[:4]):
            sb.process_view(K, R, t, m)
        grids.append(sb.all_gather(compress="sparse", unpack=False, overlap=True))
        if i >= 1:
            grids[i - 1].verify()  # the previous batch's headers, while this batch runs
            if i - 1 == 2:
                part = oracle_c.carve(shape, origin, vs, views[:4], nthreads=4)
                assert np.array_equal(grids[i - 1].to_host(), part)
    assert np.array_equal(grids[3].verify().to_host(), want)
    # the dense 2-bit form through the library's collective
    stride = sb.packed_rank_bytes(2)
    from plant3dvision_amd.sharded import DevMem
    recv = DevMem(eng, stride)
    eng.all_gather_packed(comm, 2, recv.ptr, stride, overlap=True)
    comm.synchronize()
    host = np.empty(stride, dtype=np.uint8)
    eng.dev_download(host, recv.ptr)
    assert np.array_equal(nat.widen_labels2(host.view(np.uint32), int(np.prod(shape))).reshape(shape), want)
    eng.all_gather_packed(comm, 2, recv.ptr, stride, overlap=False)
    eng.synchronize()
    eng.dev_download(host, recv.ptr)
    assert np.array_equal(nat.widen_labels2(host.view(np.uint32), int(np.prod(shape))).reshape(shape), want)
    recv.free()
    with pytest.raises(ValueError):
        eng.all_gather_sparse(comm, 0, 1, 1)  # every rank names the capacity
    sb.close()


def test_twin_engines_take_the_scans_in_turn_over_one_communicator(gpu_device):
    """ShardedBackprojection.twin(): a second engine on the rank's planes sharing the library's communicator -- scan k's
    pack and collective beside scan k + 1's carve on the other engine (what bench.py's `value` at N > 1 runs).  Different
    scans alternate here, so a grid that came from the wrong engine or buffer would show."""
    shape, origin, vs, views = scene((32, 64, 192), 10, "plant")
    wants = [oracle_c.carve(shape, origin, vs, views[:n], nthreads=4) for n in (10, 3, 6, 1, 8)]
    sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, device=0)
    sb.force_collective = True
    comm = sb.init_comm(nat.Comm.unique_id())
    tw = sb.twin()
    assert tw.comm is comm and tw.force_collective and tw.engine is not sb.engine
    pair = (sb, tw)
    grids = []
    for i, n in enumerate((10, 3, 6, 1, 8)):
        q = pair[i & 1]
        q.clear()
        for K, R, t, m in views[:n]:
            q.process_view(K, R, t, m)
        grids.append(q.all_gather(compress="sparse", unpack=False, overlap=True, check=False))
        if i >= 1:
            assert np.array_equal(grids[i - 1].verify().to_host(), wants[i - 1]), i - 1  # while scan i runs
    assert np.array_equal(grids[-1].verify().to_host(), wants[-1])
    assert np.array_equal(tw.get_local(), wants[3]) and np.array_equal(sb.get_local(), wants[4])
    tw.close()
    assert np.array_equal(sb.all_gather(compress="sparse", unpack=False).to_host(), wants[4])  # the parent's communicator lives on
    sb.close()


def test_sparse_form_at_the_benchmarked_size(gpu_device):
    """512^3 x 72, the plant: the sparse buffer of the fused batch decodes to the oracle's labels; its size."""
    shape, origin, vs, views = scene(512, 72, "plant")
    want = oracle_c.carve(shape, origin, vs, views, nthreads=16)
    K, R, t = _poses(views)
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = eng.dev_alloc(stack.nbytes)
    eng.dev_upload(ptr, stack)
    eng.process_views_device(K, R, t, ptr, *stack.shape, nat.SC_MASK_U8)
    buf = eng.get_values_sparse()
    h = sparse_header_np(buf)
    assert h["nbricks"] == 131072 and h["nmixed"] <= h["cap"] == 16384
    assert h["nmixed"] < 8000  # the live bricks that are not uniform after all 72 views
    assert buf.size < (5 << 20) and np.array_equal(nat.widen_sparse_ranks(buf, buf.size, 1, shape), want)
    eng.dev_free(ptr)
    eng.close()


def test_sparse_form_of_a_rank_of_cfg4(gpu_device):
    """BASELINE cfg 4: rank 3 of 8 of the 1024^3 x 72 grid (128 planes, plane-cyclic).  Its sparse buffer against its
    labels without building anything of the grid's size on the host: the code of every brick from the labels'
    per-brick minimum and maximum, the payload of the mixed bricks word by word."""
    shape, origin, vs, views = scene(1024, 72, "plant")
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, cyclic=(3, 8))
    K, R, t = _poses(views)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = eng.dev_alloc(stack.nbytes)
    eng.dev_upload(ptr, stack)
    eng.process_views_device(K, R, t, ptr, *stack.shape, nat.SC_MASK_U8)
    lab = eng.get_values()                      # [128][1024][1024] int32
    buf = eng.get_values_sparse()
    h = sparse_header_np(buf)
    assert (h["planes"], h["first"], h["stride"], h["bricks_y"], h["bricks_z"]) == (128, 3, 8, 64, 16)
    assert h["nbricks"] == 131072 and h["nmixed"] <= h["cap"]
    bricks = lab.reshape(128, 64, 16, 16, 64).transpose(0, 1, 3, 2, 4).reshape(131072, 1024)  # (plane, by, bz) x (jl, kl)
    lo, hi = bricks.min(axis=1), bricks.max(axis=1)
    want_code = np.where(lo != hi, 2, np.where(lo == -1, 3, lo)).astype(np.uint8)
    codes = buf[64:64 + 131072]
    assert np.array_equal(codes, want_code)
    assert h["nmixed"] == int((want_code == 2).sum()) and h["nread"] >= h["nmixed"]
    o_ids = 64 + 131072
    o_pay = o_ids + ((h["cap"] * 4 + 63) & ~63)
    ids = buf[o_ids:o_ids + 4 * h["nmixed"]].view(np.uint32)
    assert np.array_equal(np.sort(ids), np.nonzero(want_code == 2)[0])
    words = buf[o_pay:o_pay + 256 * h["nmixed"]].view(np.uint32).reshape(-1, 64).astype(np.int64)
    two = ((words[:, :, None] >> (np.arange(16, dtype=np.int64) * 2)[None, None, :]) & 3).reshape(-1, 1024)
    assert np.array_equal(np.where(two == 3, -1, two), bricks[ids])
    eng.dev_free(ptr)
    eng.close()


def test_sparse_form_without_brick_verdicts_and_on_other_engines(gpu_device):
    """A column too long for the brick form (nz > 4096: no verdict bytes, no live list -- every brick is read), and an
    averaging engine (no labels to pack: SC_ERR_STATE)."""
    shape, origin, vs, views = scenes.make_scene((2, 20, 4200), 7, "plant")
    want = oracle_c.carve(list(shape), origin, vs, views, nthreads=4)
    eng = nat.Engine(list(shape), origin, vs, nat.SC_MODE_CARVE)
    for K, R, t, m in views:
        eng.process_view(K, R, t, m, nat.SC_MASK_U8)
    h, lab = _sparse_equals_labels(eng, shape, cap=nat.sparse_bricks(*shape))
    assert np.array_equal(lab, want) and h["bricks_z"] == 66 and h["nread"] == h["nbricks"]
    eng.close()
    avg = nat.Engine([8, 16, 64], [0.0, 0.0, 0.0], 1.0, nat.SC_MODE_AVERAGE)
    with pytest.raises(nat.SpaceCarveError):
        avg.values_sparse()
    avg.close()


def test_a_carve_rank_assembles_without_torch_and_the_id_travels_over_tcp(gpu_device):
    """VERDICT r05 missing 3: the multi-GPU path needs no torch on a carve rank.  (a) A fresh interpreter shards, gathers
    through the library's communicator and never imports torch; (b) the 128-byte RCCL id goes from rank 0 to rank 1 over
    the standard library's TCP rendezvous (two processes; the communicator itself cannot be made with two ranks on one
    device, so the exchange is what is checked)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code_a = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from plant3dvision_amd import scenes\n"
        "from plant3dvision_amd.sharded import ShardedBackprojection\n"
        "shape, origin, vs, views = scenes.make_scene((16, 32, 128), 8, 'plant')\n"
        "sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, device=0)\n"
        "sb.force_collective = True\n"
        "sb.init_comm()\n"          # (no process group: the id comes from exchange_unique_id, a world of one)
        "for K, R, t, m in views: sb.process_view(K, R, t, m)\n"
        "g = sb.all_gather(compress='sparse', unpack=False)\n"
        "vol = g.to_host(); full = sb.gather_to_host()\n"
        "assert np.array_equal(vol, full) and np.array_equal(vol, sb.get_local())\n"
        "sb.close()\n"
        "assert 'torch' not in sys.modules, 'torch was imported'\n"
        "print('no-torch ok', int((vol == 1).sum()))\n") % root
    r = subprocess.run([sys.executable, "-c", code_a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"no-torch ok" in r.stdout, r.stderr.decode(errors="replace")[-2000:]
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    code_b = (
        "import sys; sys.path.insert(0, %r)\n"
        "from plant3dvision_amd.sharded import exchange_unique_id\n"
        "uid = exchange_unique_id(int(sys.argv[1]), 2, '127.0.0.1', %d)\n"
        "assert len(uid) == 128 and 'torch' not in sys.modules\n"
        "sys.stdout.write(bytes(uid).hex())\n") % (root, port)
    procs = [subprocess.Popen([sys.executable, "-c", code_b, str(rank)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for rank in (0, 1)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode(errors="replace")[-800:] for o in outs]
    ids = [o[0].decode().strip().splitlines()[-1] for o in outs]
    assert len(ids[0]) == 256 and ids[0] == ids[1] and set(ids[0]) != {"0"}


@pytest.mark.parametrize("world,partition", [(2, "cyclic"), (3, "cyclic"), (8, "cyclic"), (3, "slab"), (8, "slab")])
def test_several_ranks_buffers_unpack_into_one_grid_on_the_device(gpu_device, world, partition):
    """What an N > 1 assembly does behind its collective, with the transport taken out: every rank of a `world`-rank
    partition (engines side by side on the one GPU; nx is no multiple of world, so the ranks differ by a plane) packs
    its planes, the buffers land rank-major `rank_bytes` apart as an all-gather leaves them, and sc_sparse_headers /
    sc_unpack_sparse / sc_widen_sparse_ranks turn them into ONE grid in global order == the oracle's."""
    shape, origin, vs, views = scene((21, 40, 130), 9, "plant")
    want = oracle_c.carve(shape, origin, vs, views, nthreads=4)
    ranks = [ShardedBackprojection(shape, origin, vs, rank=r, world_size=world, device=0, partition=partition)
             for r in range(world)]
    cap = 64
    stride = ranks[0].sparse_rank_bytes(cap)
    assert all(sb.sparse_rank_bytes(cap) == stride for sb in ranks)  # the rank with the most planes sets it for all
    wire = np.zeros(stride * world, dtype=np.uint8)
    mixed, caps = [], []
    for r, sb in enumerate(ranks):
        for K, R, t, m in views:
            sb.process_view(K, R, t, m)
        buf = sb.engine.get_values_sparse(cap)
        assert buf.size <= stride
        wire[r * stride:r * stride + buf.size] = buf
        h = sparse_header_np(buf)
        assert h["planes"] == len(sb.planes) and h["first"] == sb.planes[0]
        assert h["stride"] == (world if partition == "cyclic" else 1)
        mixed.append(h["nmixed"])
        caps.append((min(cap, h["nbricks"]) + 15) & ~15)  # a rank never carries more slots than it has bricks
    eng = ranks[0].engine
    recv = eng.dev_alloc(wire.nbytes)
    eng.dev_upload(recv, wire)
    nm, cp = nat.sparse_headers(0, eng.stream(), recv, stride, world)
    assert list(nm) == mixed and list(cp) == caps
    n = int(np.prod(shape))
    for kind, dt, ref in ((4, np.int32, want), (1, np.int8, want.astype(np.int8)), (0, np.uint8, (want == 1).astype(np.uint8))):
        out = eng.dev_alloc(n * np.dtype(dt).itemsize)
        nat.unpack_sparse(0, eng.stream(), recv, stride, world, shape, out, kind)
        eng.synchronize()
        got = np.empty(n, dtype=dt)
        eng.dev_download(got, out)
        assert np.array_equal(got.reshape(shape), ref), (kind, world, partition)
        eng.dev_free(out)
    assert np.array_equal(nat.widen_sparse_ranks(wire, stride, world, shape), want)
    assert np.array_equal(unpack_sparse_np(wire, stride, world, shape), want)
    eng.dev_free(recv)
    for sb in ranks:
        sb.close()
