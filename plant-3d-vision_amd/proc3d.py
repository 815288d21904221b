"""Voxel <-> world convention of the reference (``plant3dvision/proc3d.py:28-65``), pinned
by its ``tests/unit/test_proc3d.py:12-30``: voxel centres sit at ``origin + index *
voxel_size`` -- the same rule the kernels use (``kernels/backprojection.c:71-73``)."""
import numpy as np


def index2point(indexes, origin, voxel_size):
    """Nxd indices -> Nxd points (proc3d.py:28-45)."""
    return voxel_size * np.asarray(indexes) + np.asarray(origin)[np.newaxis, :]


def point2index(points, origin, voxel_size):
    """Nxd points -> Nxd integer indices, rounded to nearest (proc3d.py:48-65)."""
    return np.array(np.round((np.asarray(points) - np.asarray(origin)[np.newaxis, :]) / voxel_size),
                    dtype=int)


class PointCloud:
    """What ``vol2pcd`` returns where open3d is absent: ``points`` and ``normals`` as float64
    ``[n, 3]`` arrays (the two attributes the reference's callers read from the open3d object)."""

    def __init__(self, points, normals):
        self.points = points
        self.normals = normals

    def __len__(self):
        return len(self.points)


def gaussian_weights(sigma=1.0, truncate=4.0):
    """The weights ``scipy.ndimage.gaussian_filter`` uses (``_gaussian_kernel1d``): radius
    ``int(truncate*sigma + 0.5)``, ``exp(-0.5/sigma^2 * x^2)`` normalised by its sum.  Returns
    the distinct half, centre first."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:], dtype=np.float64)


def release_device_buffers():
    """Give back the device work buffers ``vol2pcd`` keeps between calls (``sc_vol2pcd_release``; it keeps them
    only while they are at most 1 GiB).  ``Backprojection.close`` calls this too."""
    from . import _native as nat
    nat.backend().call("sc_vol2pcd_release")


def set_scratch_limit(nbytes):
    """Largest device work buffers a ``vol2pcd`` call may take (default 8 GiB; 0 = no limit): a volume that needs
    more -- 49 bytes per voxel -- goes through in x-slabs with a halo, same points in the same order."""
    from . import _native as nat
    nat.backend().call("sc_vol2pcd_set_scratch_limit", int(nbytes))


def vol2pcd(volume, origin, voxel_size, level_set_value=0, device=0, as_open3d=True):
    """Converts a volume into a point-cloud with normals, on the GPU
    (``plant3dvision/proc3d.py:490-570``; same signature, ``device`` / ``as_open3d`` added).

    ``volume`` may be a NumPy array (int32 / float32 / float64 / uint8, C-order), a
    ``Backprojection`` whose device-resident volume is used in place -- the 4N-byte grid then
    never crosses PCIe, only the shell's points and normals come back -- or the ``PackedGrid`` of a
    sharded run (``ShardedBackprojection.all_gather(compress="1bit" | "2bit", unpack=False)``): the
    ranks' packed planes are read as they are and the full-size grid is never written; or its ``SparseGrid``
    (``compress="sparse"``): the uint8 occupancy ``label == 1`` is written on the device from the codes and the mixed
    bricks (1 byte per voxel) and read in place.
    Returns an ``open3d.geometry.PointCloud`` when open3d is importable (and ``as_open3d``),
    else a :class:`PointCloud` with the same ``points`` / ``normals``.
    """
    import ctypes

    from . import _native as nat

    b = nat.backend()
    codes = {np.dtype(np.int32): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.uint8): 3}
    keep = None
    origin64 = np.ascontiguousarray(np.asarray(origin, dtype=np.float64).reshape(3))
    gw = gaussian_weights(1.0)
    assert gw.size == 5
    out = np.zeros(2, dtype=np.uintp)
    cnt = np.zeros(1, dtype=np.int64)
    if hasattr(volume, "occupancy_device"):  # a sharded run's SparseGrid: its occupancy (label == 1) on the device
        sg = volume
        ptr, keep = sg.occupancy_device()
        rc = b.call("sc_vol2pcd", ptr, 1, codes[np.dtype(np.uint8)], sg.shape[0], sg.shape[1], sg.shape[2], nat.addr(origin64),
                    float(voxel_size), float(level_set_value), nat.addr(gw), int(sg.device), nat.addr(out),
                    nat.addr(out) + 8, nat.addr(cnt))
    elif hasattr(volume, "recv") and hasattr(volume, "rank_bytes"):  # a sharded run's PackedGrid: read as it is
        import torch
        pg = volume
        torch.cuda.synchronize(pg.recv.device)  # the collective that filled it
        rc = b.call("sc_vol2pcd_packed", int(pg.recv.data_ptr()), int(pg.rank_bytes), int(pg.world),
                    0 if pg.partition == "cyclic" else 1, int(pg.bits), pg.shape[0], pg.shape[1], pg.shape[2],
                    nat.addr(origin64), float(voxel_size), float(level_set_value), nat.addr(gw), int(pg.device),
                    nat.addr(out), nat.addr(out) + 8, nat.addr(cnt))
    else:
        if hasattr(volume, "_engine") and hasattr(volume, "shape"):  # a Backprojection: use its state in place
            bp = volume
            ptr = bp._engine.values_device_ptr()
            bp._engine.synchronize()
            shape = [int(s) for s in bp.shape]
            code, on_device, device = codes[np.dtype(bp.dtype)], 1, bp.device
        else:
            vol = np.asarray(volume)
            if vol.ndim != 3:
                raise ValueError("volume must be 3-D")
            if vol.dtype == np.bool_:
                vol = vol.view(np.uint8)
            if vol.dtype not in codes:
                vol = vol.astype(np.float64)
            keep = np.ascontiguousarray(vol)
            ptr, shape, code, on_device = nat.addr(keep), list(keep.shape), codes[keep.dtype], 0
        rc = b.call("sc_vol2pcd", ptr, on_device, code, shape[0], shape[1], shape[2], nat.addr(origin64),
                    float(voxel_size), float(level_set_value), nat.addr(gw), int(device), nat.addr(out),
                    nat.addr(out) + 8, nat.addr(cnt))
    if rc != 0:
        msg = b.string(b.call("sc_vol2pcd_last_error"))
        if rc == nat.SC_ERR_INVALID:
            raise ValueError(f"sc_vol2pcd: {msg}")
        raise nat.SpaceCarveError(f"sc_vol2pcd: {msg} (code {rc})")
    n = int(cnt[0])
    if n:
        # the library's buffers become the arrays (no copy); they are released with the last view
        import weakref

        def adopt(address):
            owner = (ctypes.c_double * (3 * n)).from_address(address)
            weakref.finalize(owner, b.call, "sc_free_host", address)
            return np.frombuffer(owner, dtype=np.float64).reshape(n, 3)

        pts, nrm = adopt(int(out[0])), adopt(int(out[1]))
    else:
        pts = np.zeros((0, 3))
        nrm = np.zeros((0, 3))
    ok = ~np.isnan(nrm).any(axis=1)  # proc3d.py:559-561: keep points with a positive gradient norm
    if not ok.all():
        pts, nrm = pts[ok], nrm[ok]
    if as_open3d:
        try:
            import open3d as o3d  # type: ignore
        except ImportError:
            o3d = None
        if o3d is not None:
            pcd = o3d.geometry.PointCloud()
            pcd.points = o3d.utility.Vector3dVector(pts)
            pcd.normals = o3d.utility.Vector3dVector(nrm)
            return pcd
    return PointCloud(pts, nrm)


def backproject_points(points, K, rot, tvec):
    """``plant3dvision/proc3d.py:655-659`` (host, NumPy): pixel coordinates of 3-D points."""
    x = rot @ points.transpose() + tvec[:, np.newaxis]
    x = K @ x
    x = x / x[2, :][np.newaxis, :]
    return x[:2, :].transpose()


def label_points(points, cameras, masks, device=0):
    """The scoring loop of ``SegmentedPointCloud.run`` (``tasks/proc3d.py:203-232``) on the GPU.

    points  : ``[P, 3]`` float64 (``np.asarray(pcd.points)``)
    cameras : list of V camera dicts (``colmap_camera`` / ``camera`` metadata schema)
    masks   : uint8 ``[L, V, H, W]`` -- NumPy array, or a CUDA torch tensor (e.g. stacked
              ``masks2d.masks_from_predictions`` output) used in place
    Returns ``(labels int32 [P], scores float64 [L, P])``; ``labels[i]`` indexes the L labels in
    the order given (the reference iterates a Python ``set``; fix the order yourself).
    """
    from . import _native as nat

    b = nat.backend()
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
    P = pts.shape[0]
    V = len(cameras)
    K = np.ascontiguousarray(np.array([c["camera_model"]["params"][0:4] for c in cameras], dtype=np.float64).reshape(V, 4))
    R = np.ascontiguousarray(np.array([c["rotmat"] for c in cameras], dtype=np.float64).reshape(V, 9))
    t = np.ascontiguousarray(np.array([c["tvec"] for c in cameras], dtype=np.float64).reshape(V, 3))
    if hasattr(masks, "data_ptr"):  # torch tensor on the device
        import torch
        if masks.dtype != torch.uint8 or masks.dim() != 4 or not masks.is_contiguous() or not masks.is_cuda:
            raise ValueError("device masks must be a contiguous uint8 CUDA tensor [L, V, H, W]")
        torch.cuda.current_stream(masks.device.index).synchronize()
        L, Vm, H, W = (int(s) for s in masks.shape)
        mptr, on_dev, device = masks.data_ptr(), 1, masks.device.index
    else:
        m = np.ascontiguousarray(np.asarray(masks))
        if m.dtype != np.uint8 or m.ndim != 4:
            raise ValueError("masks must be uint8 [L, V, H, W]")
        L, Vm, H, W = m.shape
        mptr, on_dev = nat.addr(m), 0
    if Vm != V:
        raise ValueError("one camera per view")
    scores = np.zeros((L, P), dtype=np.float64)
    labels = np.zeros(P, dtype=np.int32)
    rc = b.call("sc_label_points", nat.addr(pts), P, L, V, nat.addr(K), nat.addr(R), nat.addr(t), mptr, on_dev,
                H, W, int(device), nat.addr(scores), nat.addr(labels))
    if rc != 0:
        msg = b.string(b.call("sc_label_points_last_error"))
        if rc == nat.SC_ERR_INVALID:
            raise ValueError(f"sc_label_points: {msg}")
        raise nat.SpaceCarveError(f"sc_label_points: {msg} (code {rc})")
    return labels, scores
