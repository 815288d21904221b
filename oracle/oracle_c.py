"""ctypes binding of ``oracle/spacecarve_oracle.c``.  TEST INFRASTRUCTURE ONLY.

Mirrors how ``plant3dvision/cl.py`` feeds the reference kernels:
``volinfo = float32[ox, oy, oz, voxel_size]`` (cl.py:181-183), ``shape = int32[3]``
(cl.py:185-187), mask cast to int32 for carve / float32 for average (cl.py:215).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libspacecarve_oracle.so")
_lib = None


def build(force=False):
    """Compile the C oracle in place (gcc, strict IEEE flags from oracle/Makefile)."""
    src = os.path.join(_HERE, "spacecarve_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libspacecarve_oracle.so"])
    return _LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    # SPACECARVE_ORACLE_LIB: another build of the same source, e.g. `make -C oracle sanitize`
    lib = ctypes.CDLL(os.environ.get("SPACECARVE_ORACLE_LIB") or _LIB_PATH)
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int32)
    common_tail = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    lib.oracle_carve_view.argtypes = [ip, ip, fp, fp, fp, fp, ip] + common_tail
    lib.oracle_carve_view.restype = ctypes.c_int
    lib.oracle_carve_view_planes.argtypes = [ip, ip, fp, fp, fp, fp, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    lib.oracle_carve_view_planes.restype = ctypes.c_int
    lib.oracle_average_view.argtypes = [fp, ip, fp, fp, fp, fp, fp] + common_tail
    lib.oracle_average_view.restype = ctypes.c_int
    lib.oracle_project.argtypes = [ip, ctypes.c_int64, fp, fp, fp, fp, ctypes.c_int,
                                   ctypes.c_int, ip, ip, ip]
    lib.oracle_project.restype = ctypes.c_int
    lib.oracle_selftest_project.argtypes = [ctypes.c_int64, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int]
    lib.oracle_selftest_project.restype = ctypes.c_int
    _lib = lib
    return lib


def _f32(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
    assert a.size == n, (a.size, n)
    return a


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


class OracleVolume:
    """State + per-view update, the way ``Backprojection`` drives the kernels."""

    def __init__(self, shape, origin, voxel_size, type="carving", default_value=0):
        self.shape = [int(s) for s in shape]
        self.type = type
        if type == "carving":
            self.dtype = np.int32
        elif type == "averaging":
            self.dtype = np.float32
        else:
            raise ValueError(type)
        self.default_value = default_value
        self.shape_h = np.array(self.shape, dtype=np.int32)
        self.volinfo_h = np.array([*origin, voxel_size], dtype=np.float32)  # cl.py:182
        self.values = np.ascontiguousarray(default_value * np.ones(self.shape, dtype=self.dtype),
                                           dtype=self.dtype)

    def clear(self):
        self.values[...] = self.default_value

    def process_view(self, intrinsics, rot, tvec, mask, nthreads=1, begin=-1, end=-1):
        lib = load()
        K = _f32(intrinsics, 4)
        R = _f32(rot, 9)
        t = _f32(tvec, 3)
        mask = np.asarray(mask)
        assert mask.ndim == 2
        H, W = mask.shape
        mask_h = np.ascontiguousarray(mask, dtype=self.dtype)  # cl.py:215
        if self.dtype == np.int32:
            rc = lib.oracle_carve_view(_ip(self.values), _ip(self.shape_h), _fp(self.volinfo_h),
                                       _fp(K), _fp(R), _fp(t), _ip(mask_h), W, H, begin, end,
                                       nthreads)
        else:
            rc = lib.oracle_average_view(_fp(self.values), _ip(self.shape_h),
                                         _fp(self.volinfo_h), _fp(K), _fp(R), _fp(t),
                                         _fp(mask_h), W, H, begin, end, nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle returned {rc}")

    def get_values(self):
        return self.values


def carve(shape, origin, voxel_size, views, default_value=0, nthreads=1):
    """views: iterable of (K[4], R[9], t[3], mask[H,W]).  Returns int32 labels."""
    vol = OracleVolume(shape, origin, voxel_size, "carving", default_value)
    for K, R, t, mask in views:
        vol.process_view(K, R, t, mask, nthreads=nthreads)
    return vol.values


def carve_planes(shape, origin, voxel_size, views, first, stride, nplanes, default_value=0, nthreads=1):
    """A rank's share of the carve: planes first, first + stride, ... (nplanes of them) of the
    [nx][ny][nz] grid, coordinates from the GLOBAL plane index.  Returns int32 [nplanes][ny][nz]."""
    lib = load()
    shape_h = np.array([int(s) for s in shape], dtype=np.int32)
    volinfo = np.array([*origin, voxel_size], dtype=np.float32)
    labels = np.full((int(nplanes), int(shape[1]), int(shape[2])), default_value, dtype=np.int32)
    for K, R, t, mask in views:
        K, R, t = _f32(K, 4), _f32(R, 9), _f32(t, 3)
        H, W = mask.shape
        mask_h = np.ascontiguousarray(mask, dtype=np.int32)  # cl.py:215
        rc = lib.oracle_carve_view_planes(_ip(labels), _ip(shape_h), _fp(volinfo), _fp(K), _fp(R), _fp(t),
                                          _ip(mask_h), W, H, int(first), int(stride), int(nplanes), int(nthreads))
        if rc != 0:
            raise RuntimeError(f"oracle returned {rc}")
    return labels


def average(shape, origin, voxel_size, views, default_value=0, nthreads=1):
    """views: iterable of (K, R, t, float32 mask[H,W]), summed in the order given."""
    vol = OracleVolume(shape, origin, voxel_size, "averaging", default_value)
    for K, R, t, mask in views:
        vol.process_view(K, R, t, mask, nthreads=nthreads)
    return vol.values


def project(ijk, origin, voxel_size, K, R, t, W, H):
    """Reference projection of explicit voxel indices -> (u, v, ok) int32 arrays."""
    lib = load()
    ijk = np.ascontiguousarray(np.asarray(ijk, dtype=np.int32).reshape(-1, 3))
    n = ijk.shape[0]
    volinfo = np.array([*origin, voxel_size], dtype=np.float32)
    u = np.empty(n, dtype=np.int32)
    v = np.empty(n, dtype=np.int32)
    ok = np.empty(n, dtype=np.int32)
    K = _f32(K, 4)
    R = _f32(R, 9)
    t = _f32(t, 3)
    lib.oracle_project(_ip(ijk), n, _fp(volinfo), _fp(K), _fp(R), _fp(t), int(W), int(H),
                       _ip(u), _ip(v), _ip(ok))
    return u, v, ok


def selftest_project(poses, count=None, seed=1, ijk=None, pose_idx=None, words=True, digests=False,
                     nthreads=1):
    """The reference projection on the samples of the engine's ``sc_selftest_project`` (same
    generator, same result words / digests); ``poses`` is the same ``[n, 28]`` record array."""
    lib = load()
    poses = np.ascontiguousarray(poses)
    assert poses.dtype.itemsize == 4 and poses.ndim == 2 and poses.shape[1] == 28
    jk = pi = None
    if ijk is not None:
        jk = np.ascontiguousarray(np.asarray(ijk, dtype=np.int32).reshape(-1, 3))
        count = jk.shape[0]
        if pose_idx is not None:
            pi = np.ascontiguousarray(np.asarray(pose_idx, dtype=np.int32).reshape(-1))
            assert pi.size == count
    count = int(count)
    w = np.empty(count, dtype=np.uint32) if words else None
    d = np.empty((count + 65535) >> 16, dtype=np.uint64) if digests else None
    rc = lib.oracle_selftest_project(count, int(seed), int(poses.shape[0]), poses.ctypes.data,
                                     jk.ctypes.data if jk is not None else None,
                                     pi.ctypes.data if pi is not None else None,
                                     w.ctypes.data if w is not None else None,
                                     d.ctypes.data if d is not None else None, int(nthreads))
    if rc != 0:
        raise RuntimeError(f"oracle returned {rc}")
    return w, d
