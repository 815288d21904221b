"""vol2pcd on the GPU against the reference algorithm run with SciPy/NumPy."""
import numpy as np
import pytest

from oracle import oracle_c, vol2pcd_oracle
from plant3dvision_amd import proc3d
from plant3dvision_amd.cl import Backprojection
from tests.helpers import scene


def test_gaussian_weights_are_scipys():
    from scipy.ndimage import _filters
    w = _filters._gaussian_kernel1d(1.0, 0, 4)
    assert np.array_equal(proc3d.gaussian_weights(1.0), w[4:])


def test_reference_unit_test_ball_oracle():
    # reference tests/unit/test_proc3d.py:64-69: a ball of radius 20 yields points
    vol = np.zeros((60, 60, 60))
    x, y, z = np.meshgrid(range(-30, 30), range(-30, 30), range(-30, 30))
    vol[x * x + y * y + z * z < 20 * 20] = 1.0
    pts, normals, *_ = vol2pcd_oracle.vol2pcd(vol, np.array([-30, -30, -30]), 1.0)
    assert len(pts) > 0
    r = np.linalg.norm(pts, axis=1)
    assert abs(np.median(r) - 19.5) < 1.0  # points sit on the sphere


def _check(vol, origin, vs, lsv, got):
    pts, normals, dist, grads, idx = vol2pcd_oracle.vol2pcd(vol, origin, vs, lsv)
    assert len(got.points) == len(pts), (len(got.points), len(pts))
    if len(pts) == 0:
        return
    # same voxels, same order: the rounded voxel index of every point's source must agree
    np.testing.assert_allclose(got.points, pts, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(got.normals, normals, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("lsv", [0.0, 0.5, 1.0])
def test_ball_like_reference_unit_test(gpu_device, lsv):
    vol = np.zeros((64, 50, 70))
    x, y, z = np.meshgrid(range(-32, 32), range(-25, 25), range(-35, 35), indexing="ij")
    vol[x * x + y * y + z * z < 20 * 20] = 1.0
    origin = np.array([-32.0, -25.0, -35.0])
    got = proc3d.vol2pcd(vol, origin, 1.0, lsv, as_open3d=False)
    assert len(got.points) > 0
    _check(vol, origin, 1.0, lsv, got)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.int32, np.float32, np.uint8, np.float64])
def test_random_blobs_touching_the_border(gpu_device, dtype):
    rng = np.random.default_rng(3)
    shape = (37, 41, 29)
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), axis=-1)
    vol = np.zeros(shape)
    for _ in range(9):
        c = rng.uniform(0, 1, 3) * np.array(shape)
        r = rng.uniform(2, 9)
        vol[((g - c) ** 2).sum(-1) < r * r] = 1
    origin = np.array([3.5, -2.25, 10.0])
    got = proc3d.vol2pcd(vol.astype(dtype), origin, 0.5, 0.5, as_open3d=False)
    _check(vol, origin, 0.5, 0.5, got)


@pytest.mark.gpu
def test_carve_volume_consumed_on_device(gpu_device):
    """Voxels -> PointCloud without the read-back: the Backprojection's device state goes in."""
    shape, origin, vs, views = scene((72, 64, 96), 16, "plant")
    bp = Backprojection(shape, origin, vs)
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    got = proc3d.vol2pcd(bp, np.array(origin), vs, 1.0, as_open3d=False)
    labels = oracle_c.carve(shape, origin, vs, views, nthreads=4)
    assert (labels == 1).sum() > 50
    _check(labels, np.array(origin), vs, 1.0, got)
    host = proc3d.vol2pcd(bp.get_values(), np.array(origin), vs, 1.0, as_open3d=False)
    assert np.array_equal(host.points, got.points) and np.array_equal(host.normals, got.normals)


@pytest.mark.gpu
def test_empty_and_full_volumes(gpu_device):
    z = np.zeros((8, 9, 10), dtype=np.int32)
    assert len(proc3d.vol2pcd(z, np.zeros(3), 1.0, as_open3d=False)) == 0
    with pytest.raises(ValueError):
        proc3d.vol2pcd(np.zeros((1, 4, 4)), np.zeros(3), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("lsv", [0.0, 1.0])
def test_slabs_give_the_points_of_the_whole_volume(gpu_device, lsv):
    """A volume whose work buffers exceed the limit goes through in x-slabs with a halo of the pipeline's reach:
    the same points and normals, bit for bit and in the same order, as the whole volume in one piece -- blobs
    across slab borders, at the volume's ends, thin sheets along x, host and device-resident volumes."""
    rng = np.random.default_rng(5)
    shape = (190, 40, 36)
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), axis=-1)
    vol = np.zeros(shape, dtype=np.uint8)
    for _ in range(30):
        c = rng.uniform(0, 1, 3) * np.array(shape)
        r = rng.uniform(2, 11)
        vol[((g - c) ** 2).sum(-1) < r * r] = 1
    vol[60:140, 10:12, 5:30] = 1  # a sheet along x
    origin = np.array([1.5, -2.0, 7.0])
    try:
        proc3d.set_scratch_limit(0)
        whole = proc3d.vol2pcd(vol, origin, 0.5, lsv, as_open3d=False)
        assert len(whole.points) > 1000
        for limit in (1, 3 << 20, 6 << 20):  # the smallest slabs the halo allows, then larger ones
            proc3d.set_scratch_limit(limit)
            got = proc3d.vol2pcd(vol, origin, 0.5, lsv, as_open3d=False)
            assert np.array_equal(got.points.view(np.uint64), whole.points.view(np.uint64)), limit
            assert np.array_equal(got.normals.view(np.uint64), whole.normals.view(np.uint64)), limit
            got32 = proc3d.vol2pcd(vol.astype(np.int32), origin, 0.5, lsv, as_open3d=False)
            assert np.array_equal(got32.points.view(np.uint64), whole.points.view(np.uint64)), limit
    finally:
        proc3d.set_scratch_limit(8 << 30)
    _check(vol.astype(np.float64), origin, 0.5, lsv, whole)


@pytest.mark.gpu
def test_ball_at_1024_cubed_in_slabs(gpu_device):
    """The reference's unit-test ball (tests/unit/test_proc3d.py:64-69) at the size of BASELINE cfg 4's assembled
    grid, 1024^3 labels resident on the device (1 GiB as uint8): in slabs of at most 1 GiB of work buffers, of
    8 GiB (the default), and in one piece (52 GB) -- the same points in the same order; points on the sphere."""
    import torch
    from plant3dvision_amd import _native as nat
    n, r = 1024, 300
    ax = torch.arange(n, device="cuda", dtype=torch.float32) - n / 2
    vol = torch.zeros((n, n, n), dtype=torch.uint8, device="cuda")
    for i0 in range(0, n, 64):  # in pieces: the distance field as float32 would be 4 GiB
        d2 = ax[i0:i0 + 64, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2
        vol[i0:i0 + 64] = (d2 < r * r).to(torch.uint8)
    torch.cuda.synchronize()
    origin = np.array([-n / 2.0] * 3)

    class Resident:  # what proc3d.vol2pcd takes in place of an array: a volume on the device
        def __init__(self):
            self.shape, self.dtype, self.device = [n, n, n], np.uint8, 0
            self._engine = self

        def values_device_ptr(self):
            return vol.data_ptr()

        def synchronize(self):
            torch.cuda.synchronize()

    res = {}
    try:
        for name, limit in (("1GiB", 1 << 30), ("8GiB", 8 << 30), ("whole", 0)):
            proc3d.set_scratch_limit(limit)
            res[name] = proc3d.vol2pcd(Resident(), origin, 1.0, 0.0, as_open3d=False)
    finally:
        proc3d.set_scratch_limit(8 << 30)
        proc3d.release_device_buffers()
    whole = res["whole"]
    assert len(whole.points) > 1_000_000
    for name in ("1GiB", "8GiB"):
        assert np.array_equal(res[name].points.view(np.uint64), whole.points.view(np.uint64)), name
        assert np.array_equal(res[name].normals.view(np.uint64), whole.normals.view(np.uint64)), name
    rad = np.linalg.norm(whole.points, axis=1)
    assert abs(np.median(rad) - (r - 0.5)) < 1.0
