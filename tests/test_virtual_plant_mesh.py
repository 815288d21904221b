"""The one semantic check on the carve path that does not pass through this repo's reading of the kernel: the
reference's test data holds the MESH its 18 virtual-plant views were rendered from
(``tests/testdata/virtual_plant/VirtualPlant_*/VirtualPlant.obj``; vertices as a fixture, axis rule of
``plant3dvision/tasks/evaluation.py:121-122``: tests/golden/make_mesh_fixture.py).  A visual hull carved from the
reference's own ``background`` masks (inverted, cl.py:300-301) and poses must contain the reference's own object.

It cannot pin bits (the oracle stays "parity unpinned"); it pins the conventions a wrong reading would
break: world -> camera as ``R X + t`` with ``rotmat`` row-major, ``u`` = column / ``v`` = row, OPENCV intrinsics
order, voxel centre = origin + index * voxel_size, foreground = ``inverted != 0``.

Tolerances, and why they are not 100 %: the scanner renders with ``add_leaf_displacement = true`` (scan.toml), so
a few leaves in the pictures are not where the mesh has them (2.1 % of the vertices miss the silhouette of at least
one of the 18 views); and a voxel is labelled by the pixel its CENTRE projects to (backprojection.c:71-79), while a
stem is 0.09 units thick (scan.toml STEM_DIAMETER) = half a pixel -- the vertex's own voxel is kept only if its
centre happens to project onto the plant in every view: 93.9 % at voxel size 0.25, 79 % at 0.5, 56 % at 1.0, which
is why the test carves at 0.25.  Every vertex has a kept voxel among the 27 around it at that size (measured 100 %).
The other direction uses the mesh's SURFACE (its triangles sampled every 0.1 units: 6 207 voxels): 99.7 % of those
voxels have a hull voxel within one voxel, 98.8 % of the hull's 8 260 voxels lie within three voxels of the surface
(a visual hull from 18 views on a circle is fatter than the object where leaves shade each other), and the hull has
1.33 x as many voxels as the surface touches -- bounds below: 99 %, 97 %, [1, 2] x.
"""
import os

import numpy as np
import pytest

from oracle import oracle_c
from plant3dvision_amd.tasks.cl import grid_from_bounding_box

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VS = 0.25


def _surface_samples(verts, tri, step=0.1):
    """The vertices plus a barycentric grid on every triangle, fine enough that neighbours are below `step` apart."""
    a, b, c = verts[tri[:, 0]], verts[tri[:, 1]], verts[tri[:, 2]]
    edge = np.maximum.reduce([np.linalg.norm(b - a, axis=1), np.linalg.norm(c - a, axis=1), np.linalg.norm(c - b, axis=1)])
    n = np.maximum(1, np.ceil(edge / step).astype(int))
    out = [verts]
    for nn in np.unique(n):
        sel = n == nn
        if nn == 1:
            out.append((a[sel] + b[sel] + c[sel]) / 3)
            continue
        i, j = np.meshgrid(np.arange(nn + 1), np.arange(nn + 1), indexing="ij")
        m = (i + j) <= nn
        u, w = (i[m] / nn)[:, None, None], (j[m] / nn)[:, None, None]
        out.append((a[sel][None] * (1 - u - w) + b[sel][None] * u + c[sel][None] * w).reshape(-1, 3))
    return np.concatenate(out)


def _data():
    d = np.load(os.path.join(GOLD, "virtual_plant_inputs.npz"))
    mesh = np.load(os.path.join(GOLD, "virtual_plant_mesh.npz"))
    verts = mesh["vertices"].astype(np.float64)
    _data.surface = _surface_samples(verts, mesh["triangles"])
    bbox = {"x": list(d["bbox"][0]), "y": list(d["bbox"][1]), "z": list(d["bbox"][2])}
    views = [(d["K_background"][q].astype(np.float32), d["R_background"][q].reshape(9).astype(np.float32),
              d["t_background"][q].astype(np.float32), d["masks_background"][q]) for q in range(d["masks_background"].shape[0])]
    return verts, bbox, views


def _vertex_stats(lab, verts, shape, origin):
    idx = np.round((verts - np.asarray(origin, dtype=np.float64)) / VS).astype(int)  # proc3d.point2index's rule
    inb = np.all((idx >= 0) & (idx < np.asarray(shape)), axis=1)
    own = np.zeros(len(verts), dtype=bool)
    own[inb] = lab[idx[inb, 0], idx[inb, 1], idx[inb, 2]] == 1
    near = np.zeros(len(verts), dtype=bool)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                j = idx + [dx, dy, dz]
                ok = np.all((j >= 0) & (j < np.asarray(shape)), axis=1)
                near[ok] |= lab[j[ok, 0], j[ok, 1], j[ok, 2]] == 1
    # the other direction, on the mesh's surface: the voxels it touches against the hull's
    sidx = np.round((_data.surface - np.asarray(origin, dtype=np.float64)) / VS).astype(int)
    sin = np.all((sidx >= 0) & (sidx < np.asarray(shape)), axis=1)
    occ = np.zeros(shape, dtype=bool)
    occ[sidx[sin, 0], sidx[sin, 1], sidx[sin, 2]] = True
    from scipy.ndimage import binary_dilation
    hull = lab == 1
    mesh_near_hull = float((occ & binary_dilation(hull)).sum() / occ.sum())
    hull_near_mesh = float((hull & binary_dilation(occ, iterations=3)).sum() / max(1, hull.sum()))
    return inb.mean(), own.mean(), near.mean(), int(hull.sum()), int(occ.sum()), hull_near_mesh, mesh_near_hull


def _check(lab, verts, shape, origin):
    inb, own, near, nhull, nocc, hull_near_mesh, mesh_near_hull = _vertex_stats(lab, verts, shape, origin)
    assert inb > 0.999                      # the grid of tasks/cl.py:95-113 holds the object
    assert own >= 0.92, own                 # measured 0.9389 (see the module text for what is missing)
    assert near >= 0.999, near              # measured 1.0
    # the hull holds the object's surface (measured 0.9974 of its voxels within one voxel of the hull) and is the
    # object and little else: 0.9875 of its voxels within three voxels of the surface, 1.33 x the surface's voxels
    assert mesh_near_hull >= 0.99, mesh_near_hull
    assert hull_near_mesh >= 0.97, hull_near_mesh
    assert nocc <= nhull <= 2 * nocc, (nhull, nocc)
    assert (lab == 0).sum() == 0            # every voxel of the bounding box is seen by some view
    return own, near, nhull, nocc, hull_near_mesh


def test_mesh_vertices_land_on_the_plant_in_the_reference_s_own_pictures():
    """Projection and pose conventions alone (oracle_c.project is backprojection.c:3-34 on points): each vertex of
    the mesh, in each of the 18 views, lands inside the picture on a pixel that is not pure background."""
    verts, _, views = _data()
    fracs = []
    in_all = np.ones(len(verts), dtype=bool)
    for K, R, t, m in views:
        # voxel "indices" = the vertices themselves: origin 0, voxel size 1 gives X = (float)i only for integers, so
        # project in float64 here -- what is pinned is the convention, not the rounding
        pc = verts @ R.reshape(3, 3).astype(np.float64).T + t.astype(np.float64)
        u = (pc[:, 0] / pc[:, 2] * K[0] + K[2]).astype(int)
        v = (pc[:, 1] / pc[:, 2] * K[1] + K[3]).astype(int)
        ok = (pc[:, 2] > 0) & (u >= 0) & (u < m.shape[1]) & (v >= 0) & (v < m.shape[0])
        fg = np.zeros(len(verts), dtype=bool)
        fg[ok] = np.invert(m)[v[ok], u[ok]] != 0   # cl.py:300-301, backprojection.c:79
        fracs.append(fg.mean())
        in_all &= fg | ~ok
    assert min(fracs) >= 0.99, fracs        # measured 0.9942 .. 0.9998
    assert in_all.mean() >= 0.97            # measured 0.979: displaced leaves
    # the check can fail: the .obj's own axes (y up) put half of the vertices off the plant
    raw = np.stack([verts[:, 0], verts[:, 2], -verts[:, 1]], axis=1)
    K, R, t, m = views[0]
    pc = raw @ R.reshape(3, 3).astype(np.float64).T + t.astype(np.float64)
    u = (pc[:, 0] / pc[:, 2] * K[0] + K[2]).astype(int)
    v = (pc[:, 1] / pc[:, 2] * K[1] + K[3]).astype(int)
    ok = (pc[:, 2] > 0) & (u >= 0) & (u < m.shape[1]) & (v >= 0) & (v < m.shape[0])
    assert (np.invert(m)[v[ok], u[ok]] != 0).sum() / len(raw) < 0.7


def test_oracle_hull_contains_the_reference_s_mesh():
    verts, bbox, views = _data()
    shape, origin = grid_from_bounding_box(bbox, VS)
    lab = oracle_c.carve(list(shape), origin, VS, [(K, R, t, np.invert(m)) for K, R, t, m in views], nthreads=8)
    _check(lab, verts, shape, origin)


@pytest.mark.gpu
def test_hip_hull_contains_the_reference_s_mesh(gpu_device):
    """The product path, through the drop-in class and its fileset-loop inversion (cl.py:300-301), fused and in
    the reference's cadence of one launch per view."""
    from plant3dvision_amd.cl import Backprojection
    verts, bbox, views = _data()
    shape, origin = grid_from_bounding_box(bbox, VS)
    for vpl in (0, 1):
        bp = Backprojection(list(shape), origin, VS, device=gpu_device, views_per_launch=vpl)
        for K, R, t, m in views:
            bp._submit_view(K, R, t, m, invert=True)
        lab = bp.get_values().copy()
        bp.close()
        _check(lab, verts, shape, origin)
    # and bit for bit the oracle's hull
    want = oracle_c.carve(list(shape), origin, VS, [(K, R, t, np.invert(m)) for K, R, t, m in views], nthreads=8)
    assert np.array_equal(lab, want)
