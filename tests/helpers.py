"""Test-only helpers: oracle-backed stand-ins so that host logic (label loops, sharding,
Voxels post-processing) runs on CPU.  Never imported by the product."""
import hashlib

import numpy as np

from oracle import oracle_c
from plant3dvision_amd import _native as nat
from plant3dvision_amd.cl import Backprojection


class OracleEngine:
    """Same surface as ``_native.Engine``, computed by the C oracle (whole grid or slab)."""

    def __init__(self, shape, origin, voxel_size, mode, default_value=0.0, device=0, slab=None,
                 cyclic=None):
        self.shape = tuple(int(s) for s in shape)
        if cyclic is not None:
            self.planes = range(int(cyclic[0]), self.shape[0], int(cyclic[1]))
            self.slab = (0, self.shape[0])  # the oracle updates the whole grid, we return the planes
        else:
            self.slab = (0, self.shape[0]) if slab is None else (int(slab[0]), int(slab[1]))
            self.planes = range(self.slab[0], self.slab[1])
        self.slab_shape = (len(self.planes), self.shape[1], self.shape[2])
        self.mode = mode
        self.dtype = np.int32 if mode == nat.SC_MODE_CARVE else np.float32
        self.device = device
        kind = "carving" if mode == nat.SC_MODE_CARVE else "averaging"
        dv = int(default_value) if mode == nat.SC_MODE_CARVE else np.float32(default_value)
        # the oracle keeps the whole grid; only the slab's flat range is ever updated
        self._vol = oracle_c.OracleVolume(self.shape, origin, voxel_size, kind, dv)
        plane = self.shape[1] * self.shape[2]
        self._begin, self._end = self.slab[0] * plane, self.slab[1] * plane
        self.views_seen = 0

    def set_option(self, key, value):
        pass

    def set_stream(self, ptr):
        pass

    def order_after(self, ptr):
        pass

    def clear(self):
        self._vol.clear()

    def set_lut(self, lut):
        self._lut = np.asarray(lut, dtype=np.float32)

    def process_view(self, K, R, t, mask, mask_dtype):
        if mask_dtype == nat.SC_MASK_U8_LUT:
            mask = self._lut[np.asarray(mask, dtype=np.uint8)]
        if mask_dtype == nat.SC_MASK_U8_INV:      # what the device folds into its bit packing
            mask = np.invert(np.asarray(mask, dtype=np.uint8))
        elif mask_dtype == nat.SC_MASK_BOOL_INV:
            mask = np.invert(np.asarray(mask).astype(np.bool_))
        self._vol.process_view(K, R, t, mask, begin=self._begin, end=self._end)
        self.views_seen += 1

    def flush(self):
        pass

    def synchronize(self):
        pass

    def get_values(self, out=None):
        pl = self.planes
        vals = self._vol.values[pl.start:pl.stop:pl.step]
        if out is None:
            return vals.copy()
        out[...] = vals.reshape(out.shape)
        return out

    def values_device_ptr(self):
        return 0

    def get_values_packed(self, bits=2):
        """NumPy restatement of ``pack_labels_kernel``: label & 3 at 2 bits, label == 1 at 1 bit, voxel v at bit
        bits * (v % (32 / bits)) of word v / (32 / bits)."""
        return pack_labels_np(self.get_values(), bits)

    def num_voxels(self):
        return int(np.prod(self.slab_shape))

    def close(self):
        pass


def pack_labels_np(values, bits):
    v = np.asarray(values, dtype=np.int64).reshape(-1)
    per = 32 // bits
    code = (v & 3) if bits == 2 else (v == 1).astype(np.int64)
    pad = (-code.size) % per
    code = np.concatenate([code, np.zeros(pad, dtype=np.int64)]).reshape(-1, per)
    shifts = (np.arange(per, dtype=np.int64) * bits)[None, :]
    return (code << shifts).sum(axis=1).astype(np.uint32)


def unpack_labels_np(recv_bytes, rank_bytes, world, partition, shape, bits, dtype):
    """NumPy restatement of ``unpack_labels_kernel``: the ranks' packed planes -> one grid in global order."""
    from plant3dvision_amd.sharded import rank_planes
    nx, ny, nz = shape
    plane = ny * nz
    per = 32 // bits
    full = np.empty((nx, plane), dtype=dtype)
    buf = np.ascontiguousarray(recv_bytes).view(np.uint8)
    for r in range(world):
        words = buf[r * rank_bytes:(r + 1) * rank_bytes].view(np.uint32).astype(np.int64)
        pl = rank_planes(nx, world, r, partition)
        nv = len(pl) * plane
        idx = np.arange(nv, dtype=np.int64)
        code = (words[idx // per] >> ((idx % per) * bits)) & ((1 << bits) - 1)
        lab = np.where(code == 3, -1, code) if bits == 2 else code
        full[pl.start:pl.stop:pl.step] = lab.reshape(len(pl), plane).astype(dtype)
    return full.reshape(nx, ny, nz)


class OracleBackprojection(Backprojection):
    """``Backprojection`` host logic over the oracle engine (CPU tests only)."""

    def init_buffers(self):
        self._engine = OracleEngine(self.shape, self.origin, self.voxel_size, self._mode,
                                    default_value=float(self.default_value))
        self._lut = None
        self.values_h = np.ascontiguousarray(
            self.default_value * np.ones(self.shape, dtype=self.dtype), dtype=self.dtype)


class FakeFile:
    """Duck-type of a plantdb ``File`` as ``process_label`` uses it (cl.py:282-298)."""

    def __init__(self, fid, array, metadata):
        self.id = fid
        self.array = array
        self._md = metadata

    def get_metadata(self, key=None, default=None):
        if key is None:
            return self._md
        return self._md.get(key, default)


class FakeFileset:
    def __init__(self, files, metadata=None):
        self._files = files
        self._md = metadata or {}

    def get_files(self, query=None):
        return list(self._files)

    def get_metadata(self, key=None, default=None):
        if key is None:
            return self._md
        return self._md.get(key, default)


def files_from_views(views, camera_key="colmap_camera", channel=None):
    from plant3dvision_amd.scenes import camera_dict
    files = []
    for q, (K, R, t, mask) in enumerate(views):
        md = {camera_key: camera_dict(K, R, t)}
        if channel is not None:
            md["channel"] = channel
        files.append(FakeFile(f"{q:05d}_{channel or 'mask'}", mask, md))
    return files


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def histogram3(labels):
    return [int((labels == -1).sum()), int((labels == 0).sum()), int((labels == 1).sum())]


import functools  # noqa: E402

from plant3dvision_amd import scenes as _scenes  # noqa: E402


@functools.lru_cache(maxsize=24)
def scene(n, n_views, kind="plant", **kw):
    """Memoised ``scenes.make_scene`` (the masks are read-only by convention)."""
    shape, origin, vs, views = _scenes.make_scene(n, n_views, kind, **kw)
    for _, _, _, m in views:
        m.setflags(write=False)
    return shape, origin, vs, views
