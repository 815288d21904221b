"""GPU: BASELINE.json's configurations at their full sizes (cfg 3 lives in test_parity_gpu.py).

* the reference's literal ``configs/test_geom_pipe_real.toml`` grid (301 x 301 x 561, 60 views): the
  whole grid against the threaded oracle;
* cfg 4 (1024^3 x 72 over 8 ranks): ranks 0, 3 and 7 of 8 built one after the other on one device,
  plane-cyclic and slab partitions -- every voxel of a rank's 128 planes against the oracle over those planes
  (coordinates from the GLOBAL plane index) + the committed digests + properties;
* cfg 5 (Masks2D feeding a 512^3 volume): 72 stand-in predictions of 896 x 896 -> per-label masks on
  the device -> 512^3 averaging and carving volumes -- properties + a voxel sample;
* the bench's extra scenes (dense, solid, noise) at 512^3: the whole grid against the oracle, fused == per view.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle_c
from plant3dvision_amd import _native as nat
from plant3dvision_amd import masks2d, scenes
from plant3dvision_amd.cl import EPS, averaging_table
from plant3dvision_amd.sharded import rank_planes
from tests.helpers import histogram3, scene, sha256

pytestmark = pytest.mark.gpu
THREADS = min(32, os.cpu_count() or 8)
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synthetic_digests.json")))


def _poses(views):
    return (np.stack([v[0] for v in views]), np.stack([v[1] for v in views]), np.stack([v[2] for v in views]))


def _carve_sample(ijk, origin, vs, views, masks=None, default=0):
    """Closed form of the carve on explicit voxels via the oracle's projection."""
    n = len(ijk)
    carved = np.zeros(n, dtype=bool)
    seen = np.zeros(n, dtype=bool)
    for q, (K, R, t, m) in enumerate(views):
        if masks is not None:
            m = masks[q]
        u, v, ok = oracle_c.project(ijk, origin, vs, K, R, t, m.shape[1], m.shape[0])
        ok = ok.astype(bool)
        hit = np.zeros(n, dtype=bool)
        hit[ok] = m[v[ok], u[ok]] != 0
        carved |= ok & ~hit
        seen |= ok
    return np.where(carved, -1, np.where(seen & (default == 0), 1, default)).astype(np.int32)


def _average_sample(ijk, origin, vs, views, masks, table):
    """The float32 sum of backprojection.c:54 in view order on explicit voxels."""
    acc = np.zeros(len(ijk), dtype=np.float32)
    for q, (K, R, t, _) in enumerate(views):
        m = masks[q]
        u, v, ok = oracle_c.project(ijk, origin, vs, K, R, t, m.shape[1], m.shape[0])
        ok = ok.astype(bool)
        acc[ok] = acc[ok] + table[m[v[ok], u[ok]]]
    return acc


def _device_masks(eng, views):
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = eng.dev_alloc(stack.nbytes)
    eng.dev_upload(ptr, stack)
    return ptr, stack.shape


def _run(eng, K, R, t, ptr, dims, vpl, order=None, code=nat.SC_MASK_U8):
    V, H, W = dims
    eng.clear()
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
    if order is None:
        eng.process_views_device(K, R, t, ptr, V, H, W, code)
    else:
        for q in order:
            eng.process_views_device(K[q:q + 1], R[q:q + 1], t[q:q + 1], ptr + int(q) * H * W, 1, H, W, code)
    return eng.get_values()


# -- the reference's literal configuration ------------------------------------------------------------
@pytest.mark.parametrize("kind", ["plant", "dense"])
def test_literal_test_geom_pipe_real_grid_whole_grid_vs_oracle(gpu_device, kind):
    """configs/test_geom_pipe_real.toml:27-36 -> 301 x 301 x 561 voxels (tasks/cl.py:143-145), the scan
    path of tests/testdata/real_plant/scan.toml (60 views): every voxel against the oracle."""
    shape, origin, vs, views = scenes.literal_real_plant_scene(60, kind)
    assert shape == [301, 301, 561]
    want = oracle_c.carve(shape, origin, vs, views, nthreads=THREADS)
    h = histogram3(want)
    assert h[1] > 0.2 * want.size and h[2] > 0  # the lower part of the box is seen by no view
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    K, R, t = _poses(views)
    ptr, dims = _device_masks(eng, views)
    for vpl in (0, 1, 7):
        got = _run(eng, K, R, t, ptr, dims, vpl)
        assert np.array_equal(got, want), (kind, vpl, histogram3(got), h)
    eng.dev_free(ptr)
    eng.close()
    # the class, fed host masks one by one like the reference's loop (cl.py:282-303)
    from plant3dvision_amd.cl import Backprojection
    bp = Backprojection(shape, origin, vs)
    for Kq, Rq, tq, m in views:
        bp.process_view(Kq, Rq, tq, m)
    assert np.array_equal(bp.get_values(), want)
    bp.close()


# -- cfg 4: 1024^3 x 72, ranks of 8 ---------------------------------------------------------------------
@pytest.mark.parametrize("partition", ["cyclic", "slab"])
def test_cfg4_1024_cubed_ranks_of_8(gpu_device, partition):
    """BASELINE cfg 4: what ranks 0, 3 and 7 of 8 compute of the 1024^3 grid (128 planes = 512 MiB
    each): every voxel against the oracle run over the same planes with GLOBAL indices (and its committed
    digest); fused == one launch per view == permuted order (sha)."""
    shape, origin, vs, views = scene(1024, 72, "plant")
    K, R, t = _poses(views)
    rng = np.random.default_rng(11)
    perm = rng.permutation(len(views))
    alive = 0
    for rank in (0, 3, 7):
        planes = rank_planes(shape[0], 8, rank, partition)
        kw = {"cyclic": (rank, 8)} if partition == "cyclic" else {"slab": (planes.start, planes.stop)}
        eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, **kw)
        assert eng.slab_shape == (128, 1024, 1024)
        ptr, dims = _device_masks(eng, views)
        fused = _run(eng, K, R, t, ptr, dims, 0).copy()
        want = oracle_c.carve_planes(shape, origin, vs, views, planes.start, planes.step, len(planes), nthreads=THREADS)
        assert np.array_equal(fused, want), (partition, rank, histogram3(fused), histogram3(want))
        del want
        dig = sha256(fused)
        gold = GOLD[f"plant_1024_72_{partition}_rank{rank}of8"]
        assert dig == gold["sha256_int32"] and histogram3(fused) == gold["hist_m1_0_p1"]
        assert sha256(_run(eng, K, R, t, ptr, dims, 1)) == dig
        assert sha256(_run(eng, K, R, t, ptr, dims, 5, order=perm)) == dig
        alive += int((fused == 1).sum())
        eng.dev_free(ptr)
        eng.close()
    assert alive > 0  # the object is in there (cyclic: in every rank; slab: in rank 3)


@pytest.mark.parametrize("partition", ["cyclic", "slab"])
def test_cfg4_every_rank_of_8_equals_the_oracles_digest(gpu_device, partition):
    """BASELINE cfg 4, the WHOLE 1024^3 x 72 grid: the fused labels of each of the 8 ranks against the oracle's committed
    SHA-256 and histogram over the same planes (tests/golden/make_golden.py ranks; ranks 0, 3 and 7 are compared
    element by element above), and the ranks' histograms add up to the whole grid's."""
    shape, origin, vs, views = scene(1024, 72, "plant")
    K, R, t = _poses(views)
    total = [0, 0, 0]
    for rank in range(8):
        planes = rank_planes(shape[0], 8, rank, partition)
        kw = {"cyclic": (rank, 8)} if partition == "cyclic" else {"slab": (planes.start, planes.stop)}
        eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, **kw)
        ptr, dims = _device_masks(eng, views)
        fused = _run(eng, K, R, t, ptr, dims, 0)
        gold = GOLD[f"plant_1024_72_{partition}_rank{rank}of8"]
        h = histogram3(fused)
        assert h == gold["hist_m1_0_p1"], (partition, rank, h)
        assert sha256(fused) == gold["sha256_int32"], (partition, rank)
        total = [a + b for a, b in zip(total, h)]
        eng.dev_free(ptr)
        eng.close()
    assert total == GOLD["plant_1024_72_whole_grid"]["hist_m1_0_p1"] and sum(total) == 1024 ** 3


# -- the bench's other scenes at cfg 3's size ---------------------------------------------------------
@pytest.mark.parametrize("kind", ["dense", "solid", "noise"])
def test_cfg3_other_scenes_whole_grid_vs_oracle(gpu_device, kind):
    """The bench's other scenes at 512^3 x 72: every voxel against the oracle (a few seconds of host threads) and
    its committed digest, then fused == one launch per view == permuted order."""
    shape, origin, vs, views = scene(512, 72, kind)
    K, R, t = _poses(views)
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    ptr, dims = _device_masks(eng, views)
    fused = _run(eng, K, R, t, ptr, dims, 0).copy()
    want = oracle_c.carve(shape, origin, vs, views, nthreads=THREADS)
    assert np.array_equal(fused, want), (kind, histogram3(fused), histogram3(want))
    del want
    dig = sha256(fused)
    gold = GOLD[f"{kind}_512_72"]
    assert dig == gold["sha256_int32"] and histogram3(fused) == gold["hist_m1_0_p1"]
    assert sha256(_run(eng, K, R, t, ptr, dims, 1)) == dig
    assert sha256(_run(eng, K, R, t, ptr, dims, 9, order=np.random.default_rng(3).permutation(72))) == dig
    h = histogram3(fused)
    if kind == "dense":
        assert 0.19 * fused.size < h[2] < 0.30 * fused.size  # the visual hull of a 20 % object
    eng.dev_free(ptr)
    eng.close()


# -- cfg 5: Masks2D feeding a 512^3 volume -------------------------------------------------------------
def test_cfg5_masks2d_896_feeding_512_cubed(gpu_device):
    """ml_pipe_real.toml's shape of work: 72 predictions of 896 x 896 (a seeded stand-in for the
    unvendored romiseg network) -> per-label uint8 masks on the device -> 512^3 volumes, averaging
    (log, as the config) and carving, no host round trip and NO synchronisation between the torch ops
    that make the masks and the engine that reads them."""
    import torch
    labels = ["background", "flower", "stem"]
    shape, origin, vs, views = scenes.make_scene(512, 72, "empty", width=896, height=896, fx=371.2 * 2, fy=371.2 * 2,
                                                 cx=448.0, cy=448.0)
    cams = [scenes.camera_dict(K, R, t) for K, R, t, _ in views]
    torch.manual_seed(0)
    coarse = torch.rand(len(views), 3, 14, 14, device="cuda")
    images = torch.nn.functional.interpolate(coarse, size=(896, 896), mode="bilinear", align_corners=False)
    net = masks2d.StandInSegmenter(labels, seed=1)
    pred = net(images)
    del images
    thr = float(pred[:, 1].median())
    # enough queued work on torch's stream that the masks are NOT ready when the engine is called
    busy = torch.rand(4096, 4096, device="cuda")
    for _ in range(20):
        busy = busy @ busy * 1e-3
    masks = masks2d.masks_from_predictions(pred, labels, threshold=thr, dilation=1)
    vols = masks2d.voxels_from_masks(masks, cams, shape, origin, vs, type="averaging", log=True)
    host = {name: masks[name].cpu().numpy() for name in labels}
    fills = [float((host[name] != 0).mean()) for name in labels]
    assert any(0.05 < f < 0.95 for f in fills), fills
    table = averaging_table(True)
    rng = np.random.default_rng(2)
    ijk = np.stack([rng.integers(0, s, 20000) for s in shape], axis=1).astype(np.int32)
    for name in labels:
        with np.errstate(over="ignore"):
            want = np.exp(_average_sample(ijk, origin, vs, views, host[name], table))
        want[want > 1] = 1.0  # tasks/cl.py:172-174
        got = vols[name][ijk[:, 0], ijk[:, 1], ijk[:, 2]]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    del vols
    # the same volumes by the schedules the engine offers (bitwise: the sum keeps the view order)
    K, R, t = _poses(views)
    name = "flower"
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE)
    eng.set_lut(table)
    m = masks[name]
    torch.cuda.synchronize()
    ref = _run(eng, K, R, t, m.data_ptr(), tuple(m.shape), 0, code=nat.SC_MASK_U8_LUT).copy()
    dig = sha256(ref)
    want = _average_sample(ijk, origin, vs, views, host[name], table)
    assert np.array_equal(ref[ijk[:, 0], ijk[:, 1], ijk[:, 2]].view(np.uint32), want.view(np.uint32))
    assert sha256(_run(eng, K, R, t, m.data_ptr(), tuple(m.shape), 9, code=nat.SC_MASK_U8_LUT)) == dig
    eng.set_option(nat.SC_OPT_AVG_BRICK, 0)
    assert sha256(_run(eng, K, R, t, m.data_ptr(), tuple(m.shape), 0, code=nat.SC_MASK_U8_LUT)) == dig
    eng.close()
    del ref
    # carving from the same device masks ("flower" as the object)
    vol = masks2d.voxels_from_masks({name: masks[name]}, cams, shape, origin, vs, type="carving")[name]
    want = _carve_sample(ijk, origin, vs, views, masks=host[name])
    assert np.array_equal(vol[ijk[:, 0], ijk[:, 1], ijk[:, 2]], want)
