#!/usr/bin/env python3
"""Voxels -> PointCloud on the reference's own ``virtual_plant`` test data (18 views, exact
``camera`` metadata), the way ``configs/test_geom_pipe_virtual.toml`` drives it: channel
``stem`` carved at voxel_size 0.5 from the bounding box of ``metadata/images.json``, then
``vol2pcd`` on the device-resident volume.  Needs an MI355X and the built library.

    python examples/virtual_plant_voxels.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from plant3dvision_amd.cl import Backprojection  # noqa: E402
from plant3dvision_amd.proc3d import vol2pcd  # noqa: E402
from plant3dvision_amd.tasks.cl import grid_from_bounding_box  # noqa: E402


class MaskFile:
    """Minimal stand-in for a plantdb ``File`` (id, metadata, pixels)."""

    def __init__(self, fid, array, camera, channel):
        self.id, self.array = fid, array
        self._md = {"camera": camera, "channel": channel}

    def get_metadata(self, key=None, default=None):
        return self._md if key is None else self._md.get(key, default)


def main():
    data = np.load(os.path.join(ROOT, "tests", "golden", "virtual_plant_inputs.npz"))
    bbox = {a: list(data["bbox"][i]) for i, a in enumerate("xyz")}
    voxel_size = 0.5
    shape, origin = grid_from_bounding_box(bbox, voxel_size)  # tasks/cl.py:143-147
    files = []
    for q in range(data["masks_stem"].shape[0]):
        cam = {"camera_model": {"params": data["K_stem"][q].tolist()},
               "rotmat": data["R_stem"][q].tolist(), "tvec": data["t_stem"][q].tolist()}
        files.append(MaskFile(f"{q:05d}_stem", data["masks_stem"][q], cam, "stem"))
    bp = Backprojection(shape, origin, voxel_size, type="carving")
    vol = bp.process_fileset(files, "camera")
    labels, counts = np.unique(vol, return_counts=True)
    print(f"grid {shape}, origin {origin}: labels {dict(zip(labels.tolist(), counts.tolist()))}")
    pcd = vol2pcd(bp, np.array(origin), voxel_size, level_set_value=0.0, as_open3d=False)
    print(f"point cloud: {len(pcd)} points, z range {pcd.points[:, 2].min():.1f} .. {pcd.points[:, 2].max():.1f}")
    bp.close()


if __name__ == "__main__":
    main()
