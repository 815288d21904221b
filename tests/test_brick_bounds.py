"""CPU check of the bound behind the fused carve's brick verdicts (csrc/sc_verdicts.h,
``brick_verdict``): the image of a brick's four corners, widened by the slack the kernel computes,
must contain the float32 pixel coordinates the reference arithmetic (backprojection.c:3-34) gives
EVERY voxel of the brick, and a brick judged in front of the camera must have every voxel in front.
The kernel's test is restated here in NumPy float32, operation by operation; its quotient is an
estimate (v_rcp_f32, 1 ulp, times the numerator), modelled as the exact quotient with the bound on
that error taken OFF the slack."""
import numpy as np
import pytest

from tests.helpers import scene

F = np.float32
BY, BZ = 16, 64


def voxel_pixels(shape, origin, vs, K, R, t):
    """float32 (p_z, u_f, v_f) of every voxel, the reference's operations in order."""
    nx, ny, nz = shape
    x = (F(origin[0]) + np.arange(nx, dtype=np.int32).astype(F) * F(vs)).astype(F)[:, None, None]
    y = (F(origin[1]) + np.arange(ny, dtype=np.int32).astype(F) * F(vs)).astype(F)[None, :, None]
    z = (F(origin[2]) + np.arange(nz, dtype=np.int32).astype(F) * F(vs)).astype(F)[None, None, :]
    R = np.asarray(R, dtype=F).reshape(9); t = np.asarray(t, dtype=F).reshape(3); K = np.asarray(K, dtype=F).reshape(4)
    row = lambda a, b, c, d: (((a * x) + (b * y)) + (c * z)) + d
    with np.errstate(all="ignore"):
        pz = row(R[6], R[7], R[8], t[2]); px = row(R[0], R[1], R[2], t[0]); py = row(R[3], R[4], R[5], t[1])
        return pz, ((px / pz) * K[0]) + K[2], ((py / pz) * K[1]) + K[3]


def brick_box(origin, vs, K, R, t, i, j0, k0, info=None, nj=BY, nk=BZ):
    """The kernel's bounding box (``rect_box``) of the rectangle of voxels (plane i, columns j0 .. j0 + nj - 1,
    voxels k0 .. k0 + nk - 1; a brick by default): None if it gives up, else (umin, umax, vmin, vmax) AFTER
    widening, with the estimate-error allowance removed.
    `info` (a dict) receives `behind`: every corner deeper than 4 ez behind the camera."""
    R = np.asarray(R, dtype=F).reshape(9); t = np.asarray(t, dtype=F).reshape(3); K = np.asarray(K, dtype=F).reshape(4)
    x = F(F(origin[0]) + F(i) * F(vs))
    ez = ex = ey = qxm = qym = F(0)
    pzmin, umin, umax, vmin, vmax = F(np.inf), F(np.inf), F(-np.inf), F(np.inf), F(-np.inf)
    pzmax = F(-np.inf)
    with np.errstate(all="ignore"):
        for c in range(4):
            y = F(F(origin[1]) + F(j0 + (nj - 1 if c >> 1 else 0)) * F(vs))
            z = F(F(origin[2]) + F(k0 + (nk - 1 if c & 1 else 0)) * F(vs))
            rzx, rzy, rzz = R[6] * x, R[7] * y, R[8] * z
            rxx, rxy, rxz = R[0] * x, R[1] * y, R[2] * z
            ryx, ryy, ryz = R[3] * x, R[4] * y, R[5] * z
            pz = ((rzx + rzy) + rzz) + t[2]; px = ((rxx + rxy) + rxz) + t[0]; py = ((ryx + ryy) + ryz) + t[1]
            ez = max(ez, (abs(rzx) + abs(rzy) + abs(rzz) + abs(t[2])) * F(2.0 ** -19))
            ex = max(ex, (abs(rxx) + abs(rxy) + abs(rxz) + abs(t[0])) * F(2.0 ** -19))
            ey = max(ey, (abs(ryx) + abs(ryy) + abs(ryz) + abs(t[1])) * F(2.0 ** -19))
            qx, qy = px / pz, py / pz
            u, v = qx * K[0] + K[2], qy * K[1] + K[3]
            if np.isnan(u) or np.isnan(v) or np.isnan(pz):
                return None
            pzmin = min(pzmin, pz); pzmax = max(pzmax, pz); qxm = max(qxm, abs(qx)); qym = max(qym, abs(qy))
            umin, umax, vmin, vmax = min(umin, u), max(umax, u), min(vmin, v), max(vmax, v)
        if info is not None:
            info["behind"] = bool(pzmax < F(-4) * ez)
        if not pzmin > F(4) * ez:
            return None
        inv = F(2) / pzmin
        mu = F(2) + abs(K[0]) * (ex + qxm * ez) * inv + (abs(K[0]) * qxm + abs(K[2]) + max(abs(umin), abs(umax))) * F(2.0 ** -20)
        mv = F(2) + abs(K[1]) * (ey + qym * ez) * inv + (abs(K[1]) * qym + abs(K[3]) + max(abs(vmin), abs(vmax))) * F(2.0 ** -20)
        # the kernel's quotients are estimates: up to 1.5 ulp of q off, i.e. |K q| 2^-22 px, and the
        # box it widens may be that much narrower than the one computed here
        mu = mu - abs(K[0]) * qxm * F(2.0 ** -22)
        mv = mv - abs(K[1]) * qym * F(2.0 ** -22)
    return float(umin - mu), float(umax + mu), float(vmin - mv), float(vmax + mv)


@pytest.mark.parametrize("shape,kw", [
    ((3, 40, 150), dict(radius_factor=1.5)),
    ((2, 33, 130), dict(radius_factor=0.35)),                     # cameras among the bricks
    ((2, 48, 192), dict(radius_factor=0.8, tilt_deg=40.0)),
    ((2, 32, 128), dict(radius_factor=3.0, width=2000, height=1500, fx=3000.0, fy=3000.0, cx=1000.0, cy=750.0)),
    ((2, 20, 70), dict(radius_factor=1.0, width=200, height=90, fx=150.0, fy=150.0, cx=100.0, cy=45.0)),
    ((2, 32, 64), dict(radius_factor=2.0, width=640, height=480, fx=900.0, fy=900.0, cx=4000.0, cy=-2500.0)),  # principal point far outside
])
def test_corner_box_with_slack_holds_every_voxel(shape, kw):
    sh, origin, vs, views = scene(shape, 7, "plant", **kw)
    checked = 0
    for K, R, t, _ in views:
        pz, uf, vf = voxel_pixels(sh, origin, vs, K, R, t)
        for i in range(sh[0]):
            for j0 in range(0, sh[1], BY):
                for k0 in range(0, sh[2], BZ):
                    box = brick_box(origin, vs, K, R, t, i, j0, k0)
                    if box is None:
                        continue
                    sl = (i, slice(j0, min(sh[1], j0 + BY)), slice(k0, min(sh[2], k0 + BZ)))
                    assert (pz[sl] > 0).all(), "a brick judged in front has a voxel behind"
                    u, v = uf[sl], vf[sl]
                    assert (u >= box[0]).all() and (u <= box[1]).all(), (i, j0, k0, box, float(u.min()), float(u.max()))
                    assert (v >= box[2]).all() and (v <= box[3]).all(), (i, j0, k0, box, float(v.min()), float(v.max()))
                    checked += 1
    assert checked > 0


def _random_pose(rng, center, extent):
    """A camera somewhere around (or inside) the volume, looking roughly at it, random roll."""
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    dist = extent * rng.choice([0.2, 0.6, 1.0, 2.5, 8.0])
    C = np.asarray(center) + d * dist
    aim = np.asarray(center) + rng.normal(size=3) * extent * rng.choice([0.0, 0.2, 0.6])
    fwd = aim - C; fwd /= np.linalg.norm(fwd)
    up = rng.normal(size=3); up -= fwd * (up @ fwd); up /= np.linalg.norm(up)
    right = np.cross(up, fwd)
    R = np.stack([right, np.cross(fwd, right), fwd])
    return R.reshape(9).astype(F), (-R @ C).astype(F)


@pytest.mark.parametrize("seed", range(12))
def test_corner_box_random_cameras(seed):
    """Random poses (far, near, inside the grid; any roll), focal lengths from fisheye-wide to
    telephoto, principal points in and far out of the picture: whenever the kernel's test accepts a
    brick's box, every voxel of the brick must be in front and project inside it."""
    rng = np.random.default_rng(1000 + seed)
    shape = (2, int(rng.integers(5, 40)), int(rng.integers(10, 150)))
    vs = float(rng.choice([0.25, 1.0, 3.0]))
    origin = (rng.normal(size=3) * 50.0).astype(F)
    extent = max(shape) * vs
    center = origin + np.array(shape) * vs / 2.0
    accepted = 0
    for _ in range(10):
        R, t = _random_pose(rng, center, extent)
        w, h = int(rng.integers(40, 2000)), int(rng.integers(40, 1500))
        f = float(w * rng.choice([0.2, 0.8, 2.0, 10.0]))
        cx = float(w * rng.choice([0.5, 0.3, -2.0, 6.0])); cy = float(h * rng.choice([0.5, 0.7, 4.0]))
        K = np.array([f, f * rng.uniform(0.8, 1.25), cx, cy], dtype=F)
        pz, uf, vf = voxel_pixels(shape, origin, vs, K, R, t)
        for i in range(shape[0]):
            for j0 in range(0, shape[1], BY):
                for k0 in range(0, shape[2], BZ):
                    box = brick_box(origin, vs, K, R, t, i, j0, k0)
                    if box is None:
                        continue
                    sl = (i, slice(j0, min(shape[1], j0 + BY)), slice(k0, min(shape[2], k0 + BZ)))
                    assert (pz[sl] > 0).all()
                    u, v = uf[sl], vf[sl]
                    assert (u >= box[0]).all() and (u <= box[1]).all(), (seed, box, float(u.min()), float(u.max()))
                    assert (v >= box[2]).all() and (v <= box[3]).all(), (seed, box, float(v.min()), float(v.max()))
                    accepted += 1
    assert accepted > 0


def brick_outside(origin, vs, K, R, t, W, H, i, j0, k0, nj=BY, nk=BZ):
    """The kernel's OUTSIDE verdict (brick_footprint): every corner well behind the camera, or the brick
    in front and its widened box entirely left of -1, right of W, above -1 or below H."""
    info = {}
    box = brick_box(origin, vs, K, R, t, i, j0, k0, info, nj=nj, nk=nk)
    if info.get("behind"):
        return True
    if box is None:
        return False
    # brick_box took the estimate allowance OFF the slack (the smallest box the kernel may use); the
    # verdict needs the whole box out of the picture, so the smallest box is the one to test
    return box[1] <= -1.0 or box[0] >= float(W) or box[3] <= -1.0 or box[2] >= float(H)


@pytest.mark.parametrize("seed", range(10))
def test_outside_verdict_means_no_voxel_is_touched(seed):
    """Whenever the restated test calls a brick OUTSIDE for a view, the reference arithmetic rejects every
    voxel of it (p_z < 0, or the truncated pixel outside the picture: backprojection.c:13,23-31)."""
    rng = np.random.default_rng(5000 + seed)
    shape = (2, int(rng.integers(5, 50)), int(rng.integers(10, 200)))
    vs = float(rng.choice([0.25, 1.0, 3.0]))
    origin = (rng.normal(size=3) * 50.0).astype(F)
    extent = max(shape) * vs
    center = origin + np.array(shape) * vs / 2.0
    outside = inside_some = 0
    for _ in range(12):
        R, t = _random_pose(rng, center, extent)
        w, h = int(rng.integers(20, 400)), int(rng.integers(20, 300))  # small pictures: much falls outside
        f = float(w * rng.choice([0.5, 2.0, 10.0]))
        K = np.array([f, f * rng.uniform(0.8, 1.25), w * rng.choice([0.5, 0.2, 1.5]), h * rng.choice([0.5, 0.8, -0.5])], dtype=F)
        pz, uf, vf = voxel_pixels(shape, origin, vs, K, R, t)
        with np.errstate(invalid="ignore"):
            touched = ~(pz < 0) & (uf > -1) & (uf < w) & (vf > -1) & (vf < h)
        for i in range(shape[0]):
            for j0 in range(0, shape[1], BY):
                for k0 in range(0, shape[2], BZ):
                    sl = (i, slice(j0, min(shape[1], j0 + BY)), slice(k0, min(shape[2], k0 + BZ)))
                    if brick_outside(origin, vs, K, R, t, w, h, i, j0, k0):
                        assert not touched[sl].any(), (seed, i, j0, k0)
                        outside += 1
                    elif touched[sl].any():
                        inside_some += 1
    assert outside > 0 and inside_some > 0, (outside, inside_some)


# ---- units (a wavefront's share of a brick: 16 columns x 16 voxels) and the cell level ---------------------

UJ, UK = 16, 16


def cell_masks(mask):
    """Restatement of what ``pack16_block`` records per 32x32 tile: a 4x4 map of its 8x8-pixel cells, bit
    cy * 4 + cx of `fg` = the cell holds some foreground, of `bg` = some background (padding is background)."""
    H, W = mask.shape
    ty, tx = (H + 31) // 32, (W + 31) // 32
    pad = np.zeros((ty * 32, tx * 32), dtype=bool)
    pad[:H, :W] = mask != 0
    cells = pad.reshape(ty * 4, 8, tx * 4, 8)
    return cells.any(axis=(1, 3)), (~cells).any(axis=(1, 3))  # [cells_y][cells_x] each


def unit_verdict(origin, vs, K, R, t, W, H, fg, bg, i, j0, k0):
    """``rect_verdict_cells`` on a unit: 4 OUTSIDE, 1 EMPTY, 2 FULL, 0 undecided."""
    if brick_outside(origin, vs, K, R, t, W, H, i, j0, k0, nj=UJ, nk=UK):
        return 4
    box = brick_box(origin, vs, K, R, t, i, j0, k0, nj=UJ, nk=UK)
    if box is None:
        return 0
    # the kernel widens by MORE than `box` (the estimate allowance was taken off): test the larger box the
    # other way round -- inside the picture with the full slack -- by asking for the smaller one plus 1 px
    umin, umax, vmin, vmax = box
    if not (umin - 1 >= 0 and umax + 1 <= W - 1 and vmin - 1 >= 0 and vmax + 1 <= H - 1):
        return 0
    cx0, cx1, cy0, cy1 = int(umin - 1) >> 3, int(umax + 1) >> 3, int(vmin - 1) >> 3, int(vmax + 1) >> 3
    if not fg[cy0:cy1 + 1, cx0:cx1 + 1].any():
        return 1
    if not bg[cy0:cy1 + 1, cx0:cx1 + 1].any():
        return 2
    return 0


@pytest.mark.parametrize("seed", range(8))
def test_unit_verdicts_at_the_cell_level_hold_for_every_voxel(seed):
    """Whenever the restated cell-level verdict calls a 16 x 16-voxel unit EMPTY / FULL / OUTSIDE for a view, the
    reference arithmetic lands every voxel of it in-image on a zero pixel / on a non-zero pixel / nowhere
    (backprojection.c:13,23-31,79-83) -- discs and bars of foreground at all scales, cameras near and far."""
    rng = np.random.default_rng(9000 + seed)
    shape = (2, int(rng.integers(10, 60)), int(rng.integers(10, 120)))
    vs = float(rng.choice([0.25, 1.0, 3.0]))
    origin = (rng.normal(size=3) * 50.0).astype(F)
    extent = max(shape) * vs
    center = origin + np.array(shape) * vs / 2.0
    seen = {1: 0, 2: 0, 4: 0, 0: 0}
    for _ in range(10):
        R, t = _random_pose(rng, center, extent)
        w, h = int(rng.integers(40, 700)), int(rng.integers(40, 500))
        f = float(w * rng.choice([0.5, 1.0, 3.0]))
        K = np.array([f, f * rng.uniform(0.8, 1.25), w * rng.choice([0.5, 0.3, 0.9]), h * rng.choice([0.5, 0.7])], dtype=F)
        yy, xx = np.mgrid[0:h, 0:w]
        kind = int(rng.integers(0, 4))
        if kind == 0:
            mask = ((yy - h * rng.uniform(0.2, 0.8)) ** 2 + (xx - w * rng.uniform(0.2, 0.8)) ** 2) < (rng.uniform(0.1, 0.6) * min(w, h)) ** 2
        elif kind == 1:
            mask = (xx > w * rng.uniform(0.2, 0.6)) & (yy < h * rng.uniform(0.4, 0.9))
        elif kind == 2:
            mask = np.ones((h, w), dtype=bool)
        else:
            mask = np.zeros((h, w), dtype=bool)
        mask = mask.astype(np.uint8) * np.uint8(255)
        fg, bg = cell_masks(mask)
        pz, uf, vf = voxel_pixels(shape, origin, vs, K, R, t)
        with np.errstate(invalid="ignore"):
            touched = ~(pz < 0) & (uf > -1) & (uf < w) & (vf > -1) & (vf < h)
        ui = np.where(touched, uf, 0).astype(np.int64).clip(0, w - 1)  # (-1, 0) truncates to 0
        vi = np.where(touched, vf, 0).astype(np.int64).clip(0, h - 1)
        pix = mask[vi, ui]
        for i in range(shape[0]):
            for j0 in range(0, shape[1], UJ):
                for k0 in range(0, shape[2], UK):
                    sl = (i, slice(j0, min(shape[1], j0 + UJ)), slice(k0, min(shape[2], k0 + UK)))
                    v = unit_verdict(origin, vs, K, R, t, w, h, fg, bg, i, j0, k0)
                    seen[v] += 1
                    if v == 4:
                        assert not touched[sl].any(), (seed, "outside", i, j0, k0)
                    elif v == 1:
                        assert touched[sl].all() and (pix[sl] == 0).all(), (seed, "empty", i, j0, k0)
                    elif v == 2:
                        assert touched[sl].all() and (pix[sl] != 0).all(), (seed, "full", i, j0, k0)
    assert seen[1] + seen[2] + seen[4] > 0
