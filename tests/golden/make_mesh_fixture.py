#!/usr/bin/env python3
"""Turns the reference's ground-truth mesh of the virtual plant into a small fixture (build container only:
/root/reference does not exist on the GPU box):

    python tests/golden/make_mesh_fixture.py     ->  tests/golden/virtual_plant_mesh.npz

Input : ``tests/testdata/virtual_plant/VirtualPlant_*/VirtualPlant.obj`` -- the mesh the 18 views of
        ``tests/testdata/virtual_plant/images`` were rendered from.  DATA (vertex coordinates), never source text.
Axes  : the .obj is Blender's y-up export; the reference brings its voxelisation into the scan's frame with
        ``arr = np.swapaxes(arr, 2, 1); arr = np.flip(arr, 1)`` (plant3dvision/tasks/evaluation.py:121-122), i.e.
        world (x, y, z) = obj (x, -z, y).  Checked here against the scan's own ``bounding_box``
        (metadata/images.json): the vertices' extent under that rule is the bounding box to 1e-4.
Output: ``vertices`` float32 [31411][3] in the scan's frame and ``triangles`` int32 [58439][3] (the .obj's faces,
        quads split along their first vertex's diagonals).  This is the one reference-held artefact on the carve
        path that does not pass through this repo's reading of the kernel: "the visual hull carved from the
        reference's own masks and poses contains the reference's own object" (tests/test_virtual_plant_mesh.py).
"""
import glob
import json
import os

import numpy as np

REF = "/root/reference/tests/testdata/virtual_plant"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    (obj,) = glob.glob(os.path.join(REF, "VirtualPlant_*", "VirtualPlant.obj"))
    v = np.array([[float(x) for x in ln.split()[1:4]] for ln in open(obj) if ln.startswith("v ")], dtype=np.float64)
    world = np.stack([v[:, 0], -v[:, 2], v[:, 1]], axis=1)  # evaluation.py:121-122 on coordinates
    bbox = json.load(open(os.path.join(REF, "metadata", "images.json")))["bounding_box"]
    want = np.array([bbox["x"], bbox["y"], bbox["z"]], dtype=np.float64)
    got = np.stack([world.min(axis=0), world.max(axis=0)], axis=1)
    assert np.abs(got - want).max() < 1e-4, (got, want)
    tri = []
    for ln in open(obj):
        if ln.startswith("f "):
            p = [int(x.split("/")[0]) - 1 for x in ln.split()[1:]]
            for k in range(1, len(p) - 1):
                tri.append((p[0], p[k], p[k + 1]))
    tri = np.array(tri, dtype=np.int32)
    assert tri.min() >= 0 and tri.max() < len(world)
    np.savez_compressed(os.path.join(OUT, "virtual_plant_mesh.npz"), vertices=world.astype(np.float32), triangles=tri)
    print("wrote virtual_plant_mesh.npz:", world.shape, tri.shape, "extent", got.tolist())


if __name__ == "__main__":
    main()
