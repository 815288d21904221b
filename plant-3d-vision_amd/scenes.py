"""Deterministic synthetic carving scenes for bench.py and the tests.

Host-side NumPy only.  A scene is ``(shape, origin, voxel_size, views)`` with
``views = [(K float32[4], R float32[9], t float32[3], mask uint8[H, W]), ...]`` in the
reference's conventions: ``x_cam = R . X + t`` with ``R`` the row-major flattening of
``rotmat`` (``plant3dvision/cl.py:293-296``), ``K = [fx, fy, cx, cy]``, and masks as
``plantdb.io.read_image`` would return them (H x W uint8, 0 = background).

Scenes
------
``plant``  S1: seeded phantom (vertical stem + 12 leaf ellipsoids, ``default_rng(1234)``)
           splatted into every view and dilated by 2 px -- a few % foreground; headline.
``solid``  S2: all-255 masks: nothing is ever carved, every voxel does every view.
``noise``  S3: Bernoulli(0.5) pixels, ``default_rng(5678)``: worst gather incoherence.
``dense``  a solid tri-axial ellipsoid filling 20 % of the grid's bounding box, exact silhouettes
           (ray-cast per pixel), cameras close enough (``radius_factor`` 1.2) for ~30 % foreground:
           what a tight bounding box around a bulky object looks like -- no shortcut for empty or
           all-white pictures applies to most of it.
``literal_real_plant_scene()``: the grid of ``configs/test_geom_pipe_real.toml:27-36`` with the
           scan path of ``tests/testdata/real_plant/scan.toml`` (60 views; cameras level with the top of
           the box, so its lower part is seen by no view).
"""
import math

import numpy as np

# intrinsics of the real_plant scanner camera (plant3dvision/colmap.py:78-82) on 1440x1080
FX = FY = 1163.6854
CX, CY = 720.0, 540.0
WIDTH, HEIGHT = 1440, 1080
CENTER = (375.0, 375.0, -35.0)
VOXEL_SIZE = 0.5


def grid_for(n, voxel_size=VOXEL_SIZE, center=CENTER):
    """Cubic grid of n^3 voxels centred on ``center`` -> (shape, origin)."""
    if isinstance(n, int):
        n = (n, n, n)
    shape = [int(s) for s in n]
    origin = [float(c - (s - 1) * voxel_size / 2.0) for c, s in zip(center, shape)]
    return shape, origin


def ring_cameras(n_views, center, radius, height=None, tilt_deg=0.0, fx=FX, fy=FY, cx=CX,
                 cy=CY, phase_deg=0.0):
    """Cameras equally spaced on a circle around ``center`` looking at it.

    Rows of R are (right, down, forward) in world coordinates; ``t = -R.C``.  Computed in
    float64, returned as the float32 triples the reference builds (cl.py:293-296).
    """
    c = np.asarray(center, dtype=np.float64)
    poses = []
    for q in range(n_views):
        th = math.radians(phase_deg) + 2.0 * math.pi * q / n_views
        C = c + np.array([radius * math.cos(th), radius * math.sin(th), 0.0])
        if height is not None:
            C[2] = height
        target = c.copy()
        if tilt_deg:
            # tilt the optical axis downwards by tilt_deg about the camera's right axis
            target[2] = C[2] - math.tan(math.radians(tilt_deg)) * radius
        fwd = target - C
        fwd /= np.linalg.norm(fwd)
        right = np.cross(np.array([0.0, 0.0, -1.0]), fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])
        t = -R @ C
        K = np.array([fx, fy, cx, cy], dtype=np.float32)
        poses.append((K, R.reshape(9).astype(np.float32), t.astype(np.float32)))
    return poses


def _dilate(mask, r):
    """Binary dilation by a (2r+1)^2 square, separable."""
    if r <= 0:
        return mask
    out = mask.copy()
    for s in range(1, r + 1):
        out[:, s:] |= mask[:, :-s]
        out[:, :-s] |= mask[:, s:]
    tmp = out.copy()
    for s in range(1, r + 1):
        out[s:, :] |= tmp[:-s, :]
        out[:-s, :] |= tmp[s:, :]
    return out


def phantom_points(extent, center, spacing, seed=1234, n_leaves=12, stem_radius=None):
    """Lattice sample of the S1 phantom (float64 world points, shape [P, 3]).

    extent: edge length of the carved volume (n * voxel_size); stem_radius defaults to
    3 voxels' worth expressed by the caller.
    """
    rng = np.random.default_rng(seed)
    c = np.asarray(center, dtype=np.float64)
    L = float(extent)
    pts = []
    # stem: vertical cylinder through the centre, 80 % of the height
    r = stem_radius if stem_radius is not None else 0.006 * L
    g = np.arange(-r, r + spacing, spacing)
    zz = np.arange(-0.4 * L, 0.4 * L + spacing, spacing)
    X, Y, Z = np.meshgrid(g, g, zz, indexing="ij")
    keep = X * X + Y * Y <= r * r
    pts.append(np.stack([X[keep], Y[keep], Z[keep]], axis=1) + c)
    # leaves: flat ellipsoids, random pose, attached around the stem
    for _ in range(n_leaves):
        a = rng.uniform(0.06, 0.12) * L
        b = rng.uniform(0.02, 0.05) * L
        h = rng.uniform(0.005, 0.012) * L
        az = rng.uniform(0, 2 * math.pi)
        el = rng.uniform(-0.6, 0.6)
        roll = rng.uniform(-0.5, 0.5)
        zc = rng.uniform(-0.33, 0.33) * L
        ca, sa, ce, se, cr, sr = (math.cos(az), math.sin(az), math.cos(el), math.sin(el),
                                  math.cos(roll), math.sin(roll))
        Rz = np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1]])
        Ry = np.array([[ce, 0, se], [0, 1, 0], [-se, 0, ce]])
        Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        Rm = Rz @ Ry @ Rx
        ga = np.arange(-a, a + spacing, spacing)
        gb = np.arange(-b, b + spacing, spacing)
        gh = np.arange(-h, h + spacing, spacing)
        A, B, Hh = np.meshgrid(ga, gb, gh, indexing="ij")
        keep = (A / a) ** 2 + (B / b) ** 2 + (Hh / h) ** 2 <= 1.0
        local = np.stack([A[keep] + a, B[keep], Hh[keep]], axis=1)  # root at the stem
        pts.append(local @ Rm.T + c + np.array([0.0, 0.0, zc]))
    return np.concatenate(pts, axis=0)


def splat_mask(points, K, R, t, width, height, dilate=2):
    """Forward-project float64 points with the float32 pose; returns uint8 {0,255} mask."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    t = np.asarray(t, dtype=np.float64)
    cam = points @ R.T + t
    z = cam[:, 2]
    front = z > 1e-6
    u = np.floor(cam[front, 0] / z[front] * float(K[0]) + float(K[2])).astype(np.int64)
    v = np.floor(cam[front, 1] / z[front] * float(K[1]) + float(K[3])).astype(np.int64)
    inside = (u >= 0) & (u < width) & (v >= 0) & (v < height)
    m = np.zeros((height, width), dtype=bool)
    m[v[inside], u[inside]] = True
    m = _dilate(m, dilate)
    return (m.astype(np.uint8)) * np.uint8(255)


def ellipsoid_silhouette(K, R, t, width, height, centre, semi_axes):
    """Exact silhouette (uint8 {0,255}) of the axis-aligned ellipsoid ``sum(((X - centre) / a)^2) <= 1``:
    a pixel is foreground when the ray through its centre meets the ellipsoid in front of the camera."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    t = np.asarray(t, dtype=np.float64)
    C = -R.T @ t
    u = (np.arange(width, dtype=np.float64) + 0.5 - float(K[2])) / float(K[0])
    v = (np.arange(height, dtype=np.float64) + 0.5 - float(K[3])) / float(K[1])
    a = np.asarray(semi_axes, dtype=np.float64)
    o = (C - np.asarray(centre, dtype=np.float64)) / a
    # ray direction in world coordinates, scaled by the semi-axes: d = R^T (u, v, 1) / a
    dx = (R[0, 0] * u[None, :] + R[1, 0] * v[:, None] + R[2, 0]) / a[0]
    dy = (R[0, 1] * u[None, :] + R[1, 1] * v[:, None] + R[2, 1]) / a[1]
    dz = (R[0, 2] * u[None, :] + R[1, 2] * v[:, None] + R[2, 2]) / a[2]
    A = dx * dx + dy * dy + dz * dz
    B = dx * o[0] + dy * o[1] + dz * o[2]
    Cc = float(o @ o) - 1.0
    hit = (B * B - A * Cc >= 0.0) & (B < 0.0)
    return hit.astype(np.uint8) * np.uint8(255)


#: semi-axes of the "dense" object as fractions of the grid's longest edge: 4/3 pi abc = 0.2006
DENSE_SEMI_AXES = (0.38, 0.28, 0.45)


def literal_real_plant_scene(n_views=60, kind="plant", width=WIDTH, height=HEIGHT):
    """The reference's literal ``Voxels`` configuration (``configs/test_geom_pipe_real.toml:27-36``):
    bounding box x[300,450] y[300,450] z[-175,105] at voxel_size 0.5 -> 301 x 301 x 561 voxels
    (``tasks/cl.py:143-145``), seen by the scan path of ``tests/testdata/real_plant/scan.toml``
    (circle centre (375, 375), radius 300, z = 80, tilt 0, 60 points) through the scanner's
    camera model (``colmap.py:78-82``).  The pictures are synthetic (the S1 phantom standing in the
    box); poses are the nominal ones of the scan path, not COLMAP's."""
    bbox = {"x": [300, 450], "y": [300, 450], "z": [-175, 105]}
    vs = 0.5
    shape = [int((bbox[a][1] - bbox[a][0]) / vs) + 1 for a in "xyz"]  # tasks/cl.py:143-145
    origin = [float(bbox[a][0]) for a in "xyz"]
    centre = [origin[a] + (shape[a] - 1) * vs / 2.0 for a in range(3)]
    poses = ring_cameras(n_views, (375.0, 375.0, 80.0), 300.0)  # level cameras at z = 80
    extent = max(shape) * vs
    if kind == "plant":
        pts = phantom_points(extent, centre, max(1.5 * (300.0 - 0.4 * extent) / FX, extent / 800.0))
        masks = [splat_mask(pts, K, R, t, width, height, dilate=2) for K, R, t in poses]
    elif kind == "dense":
        masks = [ellipsoid_silhouette(K, R, t, width, height, centre, [f * extent for f in (0.2, 0.15, 0.45)])
                 for K, R, t in poses]
    else:
        raise ValueError(f"unknown scene kind {kind!r}")
    return shape, origin, vs, [(K, R, t, m) for (K, R, t), m in zip(poses, masks)]


def scene_poses(n, n_views, kind="plant", voxel_size=VOXEL_SIZE, center=CENTER, radius_factor=2.0, tilt_deg=0.0,
                fx=FX, fy=FY, cx=CX, cy=CY):
    """Grid and camera ring of ``make_scene`` without the masks: (shape, origin, voxel_size, poses, radius, extent)."""
    shape, origin = grid_for(n, voxel_size, center)
    extent = max(shape) * voxel_size
    if kind == "dense" and radius_factor == 2.0:
        radius_factor = 1.2  # close cameras: the object fills ~30 % of every picture
    radius = radius_factor * extent
    poses = ring_cameras(n_views, center, radius, tilt_deg=tilt_deg, fx=fx, fy=fy, cx=cx, cy=cy)
    return shape, origin, float(voxel_size), poses, radius, extent


def make_scene(n, n_views, kind="plant", width=WIDTH, height=HEIGHT, voxel_size=VOXEL_SIZE,
               center=CENTER, radius_factor=2.0, tilt_deg=0.0, fx=FX, fy=FY, cx=CX, cy=CY,
               seed=None):
    """Build scene ``kind`` for an n^3 (or (nx,ny,nz)) grid and ``n_views`` cameras.

    Camera ring radius = radius_factor * max(n) * voxel_size at the height of the centre
    (every voxel then projects inside every 1440x1080 image: worst-case work).
    """
    shape, origin, voxel_size, poses, radius, extent = scene_poses(n, n_views, kind, voxel_size, center, radius_factor,
                                                                   tilt_deg, fx, fy, cx, cy)
    masks = []
    if kind == "plant":
        # lattice spacing ~1.5 px at the nearest depth -- but never finer than extent / 800 (the
        # default ring gives extent / 620): a ring close to or inside the volume would otherwise
        # ask for a lattice of billions of points (or a negative spacing, i.e. none at all); its
        # masks are then dotted rather than solid near the camera, which the parity tests do not mind
        spacing = max(1.5 * (radius - 0.75 * extent) / fx, extent / 800.0)
        # stem radius 0.006 * extent = 3 voxels at 512^3; the scene is
        # scale-invariant in pixel space, so every n sees the same masks
        pts = phantom_points(extent, center, spacing, seed=1234 if seed is None else seed)
        for K, R, t in poses:
            masks.append(splat_mask(pts, K, R, t, width, height, dilate=2))
    elif kind == "solid":
        full = np.full((height, width), 255, dtype=np.uint8)
        masks = [full for _ in poses]
    elif kind == "noise":
        rng = np.random.default_rng(5678 if seed is None else seed)
        for _ in poses:
            masks.append((rng.random((height, width)) < 0.5).astype(np.uint8) * np.uint8(255))
    elif kind == "dense":
        semi = [f * extent for f in DENSE_SEMI_AXES]
        masks = [ellipsoid_silhouette(K, R, t, width, height, center, semi) for K, R, t in poses]
    elif kind == "empty":
        zero = np.zeros((height, width), dtype=np.uint8)
        masks = [zero for _ in poses]
    else:
        raise ValueError(f"unknown scene kind {kind!r}")
    views = [(K, R, t, m) for (K, R, t), m in zip(poses, masks)]
    return shape, origin, float(voxel_size), views


def camera_dict(K, R, t):
    """The metadata dict ``process_label`` reads (cl.py:293-296)."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    return {"camera_model": {"params": [float(k) for k in K]},
            "rotmat": [[float(x) for x in row] for row in R],
            "tvec": [float(x) for x in np.asarray(t).reshape(3)]}
