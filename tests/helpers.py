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

    def get_values_sparse(self, cap=0):
        """NumPy restatement of ``sparse_pack_kernel`` (the brick-sparse form of include/spacecarve.h)."""
        pl = self.planes
        return pack_sparse_np(self.get_values(), pl.start, pl.step if len(pl) > 1 else 1, cap)

    def num_voxels(self):
        return int(np.prod(self.slab_shape))

    def close(self):
        pass


SPARSE_MAGIC = 0x50534353


def sparse_layout_np(nbricks, cap):
    codes = 64
    ids = codes + ((nbricks + 63) & ~63)
    payload = ids + ((cap * 4 + 63) & ~63)
    return codes, ids, payload, payload + cap * 256


def sparse_cap_np(cap, nbricks):
    cap = min(max(int(cap), 16), nbricks)
    return (cap + 15) & ~15


def pack_sparse_np(values, first, stride, cap=0):
    """One rank's planes ``values[P][ny][nz]`` (int32 labels in {-1, 0, 1}) -> its sparse buffer (uint8 array): header,
    one code per 16 x 64 brick (0 / 1 / 3 uniform, 2 mixed), ids and 256-byte payloads of the mixed bricks in brick
    order (the device hands slots out in any order; a reader goes by the ids)."""
    v = np.asarray(values, dtype=np.int32)
    P, ny, nz = v.shape
    bys, bzs = (ny + 15) // 16, (nz + 63) // 64
    nbricks = P * bys * bzs
    cap = sparse_cap_np(cap if cap else max(1024, nbricks // 8), nbricks)
    o_codes, o_ids, o_pay, total = sparse_layout_np(nbricks, cap)
    buf = np.zeros(total, dtype=np.uint8)
    padded = np.zeros((P, bys * 16, bzs * 64), dtype=np.int64)
    valid = np.zeros((P, bys * 16, bzs * 64), dtype=bool)
    padded[:, :ny, :nz] = v & 3
    valid[:, :ny, :nz] = True
    bricks = padded.reshape(P, bys, 16, bzs, 64).transpose(0, 1, 3, 2, 4).reshape(nbricks, 1024)
    bvalid = valid.reshape(P, bys, 16, bzs, 64).transpose(0, 1, 3, 2, 4).reshape(nbricks, 1024)
    codes = np.full(nbricks, 2, dtype=np.uint8)
    for c in (0, 1, 3):  # (the kernel's priority: all -1, then all 1, then all 0 -- they exclude each other on valid voxels)
        codes[np.all((bricks == c) | ~bvalid, axis=1)] = c
    mixed = np.nonzero(codes == 2)[0]
    buf[o_codes:o_codes + nbricks] = codes
    shifts = (np.arange(16, dtype=np.int64) * 2)[None, None, :]
    nput = min(len(mixed), cap)
    if nput:
        words = (np.where(bvalid[mixed[:nput]], bricks[mixed[:nput]], 0).reshape(nput, 64, 16) << shifts).sum(axis=2).astype(np.uint32)
        buf[o_ids:o_ids + 4 * nput] = mixed[:nput].astype(np.uint32).view(np.uint8)
        buf[o_pay:o_pay + 256 * nput] = words.reshape(-1).view(np.uint8)
    hdr = np.array([SPARSE_MAGIC, 1, 2, nbricks, cap, len(mixed), P, ny, nz, bys, bzs, first, stride, 0, 0, 0], dtype=np.uint32)
    buf[:64] = hdr.view(np.uint8)
    return buf


def sparse_header_np(buf):
    h = np.ascontiguousarray(buf[:64]).view(np.uint32)
    keys = ("magic", "version", "bits", "nbricks", "cap", "nmixed", "planes", "ny", "nz", "bricks_y", "bricks_z", "first", "stride", "nread")
    return {k: int(h[i]) for i, k in enumerate(keys)}


def unpack_sparse_np(recv, rank_bytes, world, shape, dtype=np.int32):
    """NumPy restatement of ``sparse_unpack_kernel`` / ``sc_widen_sparse_ranks``: the ranks' sparse buffers -> one grid."""
    nx, ny, nz = shape
    full = np.full((nx, ny, nz), 99, dtype=dtype)
    recv = np.ascontiguousarray(recv).view(np.uint8).reshape(-1)
    for r in range(world):
        buf = recv[r * rank_bytes:(r + 1) * rank_bytes]
        h = sparse_header_np(buf)
        assert h["magic"] == SPARSE_MAGIC and h["nmixed"] <= h["cap"], h
        o_codes, o_ids, o_pay, _ = sparse_layout_np(h["nbricks"], h["cap"])
        codes = buf[o_codes:o_codes + h["nbricks"]]
        ids = buf[o_ids:o_ids + 4 * h["nmixed"]].view(np.uint32)
        words = buf[o_pay:o_pay + 256 * h["nmixed"]].view(np.uint32).reshape(-1, 64).astype(np.int64)
        bys, bzs = h["bricks_y"], h["bricks_z"]
        lab = np.where(codes == 3, -1, codes.astype(np.int64))[:, None] * np.ones((1, 1024), dtype=np.int64)
        if len(ids):
            two = (words[:, :, None] >> (np.arange(16, dtype=np.int64) * 2)[None, None, :]) & 3
            lab[ids] = np.where(two == 3, -1, two).reshape(-1, 1024)
        planes = lab.reshape(h["planes"], bys, bzs, 16, 64).transpose(0, 1, 3, 2, 4).reshape(h["planes"], bys * 16, bzs * 64)
        gi = h["first"] + np.arange(h["planes"]) * h["stride"]
        full[gi] = planes[:, :ny, :nz].astype(dtype)
    return full


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
