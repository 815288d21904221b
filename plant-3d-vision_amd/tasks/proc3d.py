"""The ``PointCloud`` task logic of the reference (``plant3dvision/tasks/proc3d.py:20-136``)
over the GPU ``vol2pcd``: single volume -> one cloud; labelled volumes -> arg-max over labels,
contrast / score gates, one cloud per non-background label.  luigi / plantdb / open3d are not
needed: volumes in, ``(points, normals, point_labels)`` out."""
import logging

import numpy as np

logger = logging.getLogger(__name__)

#: parameter defaults of the reference task (tasks/proc3d.py:59-63)
POINTCLOUD_DEFAULTS = dict(level_set_value=1.0, background_prior=1.0, min_contrast=10.0, min_score=0.2)


def point_cloud_run(voxels, origin, voxel_size, level_set_value=1.0, background_prior=1.0,
                    min_contrast=10.0, min_score=0.2, vol2pcd=None):
    """``PointCloud.run`` (tasks/proc3d.py:65-136).

    voxels : a single 3-D volume (NumPy array or a ``Backprojection``, consumed on the device),
        or a dict ``{label: volume}`` as ``Voxels`` writes for labelled masks.  A one-entry dict
        counts as a single volume (:70-72).
    Returns ``(points [n,3], normals [n,3], point_labels)``; ``point_labels`` is ``None`` for a
    single volume and the per-point label list for the multiclass case (:121-125).
    """
    if vol2pcd is None:
        from ..proc3d import vol2pcd as _v

        def vol2pcd(v, o, s, l):
            return _v(v, o, s, l, as_open3d=False)
    origin = np.array(origin)
    voxel_size = float(voxel_size)
    if isinstance(voxels, dict) and len(voxels) == 1:
        voxels = voxels[list(voxels.keys())[0]]
    if not isinstance(voxels, dict):
        out = vol2pcd(voxels, origin, voxel_size, level_set_value)  # :134
        return np.asarray(out.points), np.asarray(out.normals), None

    l = list(voxels.keys())
    res = np.zeros((*voxels[l[0]].shape, len(l)))  # :82
    for i in range(len(l)):
        res[:, :, :, i] = voxels[l[i]]
    for i in range(len(l)):
        if l[i] == 'background':
            res[:, :, :, i] *= background_prior  # :85-87
    res_idx = np.argmax(res, axis=3)  # :91
    pts, nrm, point_labels = [], [], []
    for i in range(len(l)):
        logger.debug(f"label = {l[i]}")
        if l[i] != 'background':
            pred_no_c = np.max(np.delete(res, i, axis=3), axis=3)  # :106
            pred_c = (res_idx == i)  # :108
            if min_contrast > 1.0:
                pred_c *= (pred_c > (min_contrast * pred_no_c))  # :110
            pred_c *= (pred_c > min_score)  # :111
            out = vol2pcd(pred_c, origin, voxel_size, level_set_value)  # :113
            pts.append(np.asarray(out.points))
            nrm.append(np.asarray(out.normals))
            point_labels = point_labels + [l[i]] * len(out.points)  # :121
    if pts:
        return np.concatenate(pts, axis=0), np.concatenate(nrm, axis=0), point_labels
    return np.zeros((0, 3)), np.zeros((0, 3)), point_labels
