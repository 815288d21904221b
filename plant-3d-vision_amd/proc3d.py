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
