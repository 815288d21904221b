"""NumPy float32 restatement of the reference kernels.  TEST INFRASTRUCTURE ONLY.

Independent of ``spacecarve_oracle.c``: every arithmetic step is a separate float32 ufunc
(NumPy never contracts a*b+c), division is IEEE, and the ``(int)`` cast's out-of-range
behaviour is written out.  Follows ``plant3dvision/kernels/backprojection.c`` line by
line (carve :57-84, average :36-55, backproject_point :3-34); voxel order is C-order
``[nx][ny][nz]`` as ``common.h:1-12`` unravels it.  Meant for grids up to ~128^3.
"""
import numpy as np

F = np.float32
INT_MIN = np.int32(-2 ** 31)


def _cvt_trunc(f):
    """(int)f with x86 cvttss2si semantics: INT_MIN for NaN/inf/out of int32 range."""
    good = (f > F(-2147483904.0)) & (f < F(2147483648.0))
    out = np.full(f.shape, INT_MIN, dtype=np.int32)
    out[good] = np.trunc(f[good]).astype(np.int32)
    return out


def backproject(shape, origin, voxel_size, K, R, t, W, H):
    """Project every voxel centre.  Returns (ok[nx,ny,nz] bool, u, v int32)."""
    nx, ny, nz = (int(s) for s in shape)
    vi = np.array([*origin, voxel_size], dtype=F)  # cl.py:182
    K = np.asarray(K, dtype=F).reshape(4)
    R = np.asarray(R, dtype=F).reshape(9)
    t = np.asarray(t, dtype=F).reshape(3)
    # backprojection.c:71-73: volinfo[a] + (float)index * volinfo[3]
    x = (vi[0] + np.arange(nx, dtype=np.int32).astype(F) * vi[3]).astype(F)[:, None, None]
    y = (vi[1] + np.arange(ny, dtype=np.int32).astype(F) * vi[3]).astype(F)[None, :, None]
    z = (vi[2] + np.arange(nz, dtype=np.int32).astype(F) * vi[3]).astype(F)[None, None, :]

    def row(a, b, c, d):  # ((a*x + b*y) + c*z) + d, left to right (:11,17,18)
        return (((a * x) + (b * y)) + (c * z)) + d

    with np.errstate(all="ignore"):
        p_z = row(R[6], R[7], R[8], t[2])
        p_x = row(R[0], R[1], R[2], t[0])
        p_y = row(R[3], R[4], R[5], t[1])
        assert p_z.dtype == F and p_x.dtype == F
        front = ~(p_z < 0)  # :13 -- NaN is not rejected here
        px = ((p_x / p_z) * K[0]) + K[2]  # :20
        py = ((p_y / p_z) * K[1]) + K[3]  # :21
        assert px.dtype == F
    u = _cvt_trunc(px)
    v = _cvt_trunc(py)
    ok = front & ~((u < 0) | (u > W - 1)) & ~((v < 0) | (v > H - 1))  # :26-31
    return ok, u, v


def carve_view(labels, origin, voxel_size, K, R, t, mask):
    """One ``carve`` launch, in place on int32 ``labels[nx,ny,nz]``."""
    mask = np.ascontiguousarray(mask, dtype=np.int32)  # cl.py:215
    H, W = mask.shape
    ok, u, v = backproject(labels.shape, origin, voxel_size, K, R, t, W, H)
    live = (labels != -1) & ok  # :67, :76
    m = np.zeros(labels.shape, dtype=np.int32)
    m[live] = mask[v[live], u[live]]
    zero = live & (m == 0)
    seen = live & (m != 0) & (labels == 0)
    labels[zero] = -1  # :79-80
    labels[seen] = 1  # :81-83
    return labels


def average_view(values, origin, voxel_size, K, R, t, mask):
    """One ``average`` launch, in place on float32 ``values[nx,ny,nz]``."""
    mask = np.ascontiguousarray(mask, dtype=F)
    H, W = mask.shape
    ok, u, v = backproject(values.shape, origin, voxel_size, K, R, t, W, H)
    values[ok] = values[ok] + mask[v[ok], u[ok]]  # :54
    return values


def carve(shape, origin, voxel_size, views, default_value=0):
    labels = np.ascontiguousarray(default_value * np.ones(shape, dtype=np.int32), dtype=np.int32)
    for K, R, t, mask in views:
        carve_view(labels, origin, voxel_size, K, R, t, mask)
    return labels


def average(shape, origin, voxel_size, views, default_value=0):
    values = np.ascontiguousarray(default_value * np.ones(shape, dtype=F), dtype=F)
    for K, R, t, mask in views:
        average_view(values, origin, voxel_size, K, R, t, mask)
    return values


def carve_closed_form(shape, origin, voxel_size, views, default_value=0):
    """Order-independent closed form of the carve state, used to justify
    view fusion/re-ordering: -1 if any in-image hit on a zero pixel; else 1 if seen and
    default is 0; else default."""
    carved = np.zeros(shape, dtype=bool)
    seen = np.zeros(shape, dtype=bool)
    for K, R, t, mask in views:
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        H, W = mask.shape
        ok, u, v = backproject(shape, origin, voxel_size, K, R, t, W, H)
        m = np.zeros(shape, dtype=np.int32)
        m[ok] = mask[v[ok], u[ok]]
        carved |= ok & (m == 0)
        seen |= ok
    d = np.int32(default_value)
    out = np.full(shape, d, dtype=np.int32)
    if d == 0:
        out[seen] = 1
    if d != -1:
        out[carved] = -1
    return out
