"""CPU oracle for ``vol2pcd``.  TEST INFRASTRUCTURE ONLY.

The reference's own algorithm (``plant3dvision/proc3d.py:490-570``) with the reference's own
third-party calls -- ``scipy.ndimage.distance_transform_edt`` / ``gaussian_filter`` and
``numpy.gradient`` (scipy and numpy are installed here; the reference pins neither) -- minus the
open3d container (absent) and with the per-point loop (:539-555, joblib) vectorised.  The vector
norm is taken the way NumPy takes it per point (``np.linalg.norm`` -> BLAS), so the last bits of
points and normals are machine-dependent in the reference itself; tests compare with a few ulp of
slack there and exactly on which voxels are selected and in what order.
"""
import numpy as np
from scipy.ndimage import distance_transform_edt, gaussian_filter


def vol2pcd(volume, origin, voxel_size, level_set_value=0):
    volume = 1.0 * (np.asarray(volume) > 0.5)  # :515
    dist = distance_transform_edt(volume)  # :518
    mdist = distance_transform_edt(1 - volume)  # :519
    dist = np.where(dist > 0.5, dist - 0.5, -mdist + 0.5)  # :522
    gx, gy, gz = np.gradient(dist)  # :525
    gx = gaussian_filter(gx, 1)  # :528-530
    gy = gaussian_filter(gy, 1)
    gz = gaussian_filter(gz, 1)
    on_edge = (dist > -level_set_value) * (dist <= -level_set_value + np.sqrt(3))  # :533
    x, y, z = np.nonzero(on_edge)
    grad = np.stack([gx[x, y, z], gy[x, y, z], gz[x, y, z]], axis=1)
    norm = np.array([np.linalg.norm(g) for g in grad]) if len(grad) < 20000 else np.sqrt((grad ** 2).sum(axis=1))
    keep = norm > 0  # :544, :559-561
    gn = grad[keep] / norm[keep, None]
    val = dist[x, y, z][keep] + level_set_value - np.sqrt(3) / 2  # :546
    idx = np.stack([x, y, z], axis=1)[keep].astype(np.float64)
    pts = idx - gn * val[:, None]  # :547-549
    normals = -gn  # :550-552
    pts = voxel_size * pts + np.asarray(origin, dtype=np.float64)[np.newaxis, :]  # index2point, :563
    normals = normals / np.linalg.norm(normals, axis=1)[:, None]  # open3d normalize_normals, :567
    return pts, normals, dist, (gx, gy, gz), np.stack([x, y, z], axis=1)[keep]
