"""Multi-GPU space carving: one process per GPU, the grid sharded by X-planes.

The reference is single-device (one module-global OpenCL queue, ``plant3dvision/cl.py:29-30``);
this module is the MI355X-native addition.  Voxels are independent
(``kernels/backprojection.c:64-84``: one owner per voxel), so:

* rank r owns a set of x-planes of the C-order ``[nx][ny][nz]`` grid, computed from GLOBAL
  indices, hence bit-identical to the single-GPU result.  Two partitions:
  ``"cyclic"`` (default): planes ``r, r+W, r+2W, ...`` (``sc_create_cyclic``).  The object sits
  in the middle of the grid, so contiguous slabs give the middle ranks all the surviving voxels
  (measured at 1024^3 / 8 ranks: 6.3 ms on ranks 3-4 against 0.34 ms on the others); dealing the
  planes round-robin gives every rank the same mix.
  ``"slab"``: ``i in [nx*r//W, nx*(r+1)//W)`` (``sc_create_slab``), a contiguous block of the
  output, for consumers that want one;
* every rank applies every view to its slab: the data path needs NO collective;
* assembling the full grid is a separate, optional step: ``gather_to_host`` (what a
  ``Voxels`` run needs: the volume in host memory of one process), ``all_gather`` (RCCL
  all-gather over xGMI, every GPU ends with the full grid) or ``all_reduce`` (the
  zero-padded sum the north star words; ring-bound).

``torch.distributed`` is plumbing only (process group, RCCL/gloo collectives); the carve runs
in the HIP engine.  ``engine_factory`` exists so the host logic can be exercised on CPU with
gloo in tests; the default is the HIP engine and there is no CPU fallback.
"""
import numpy as np

from . import _native as nat


def rank_planes(nx, world_size, rank, partition="cyclic"):
    """The global x indices rank ``rank`` owns, as a ``range``."""
    if partition == "cyclic":
        if not 0 <= rank < world_size:
            raise ValueError("rank out of range")
        if world_size > nx:
            raise ValueError(f"cannot shard {nx} planes over {world_size} ranks")
        return range(rank, nx, world_size)
    if partition == "slab":
        i0, i1 = slab_bounds(nx, world_size, rank)
        return range(i0, i1)
    raise ValueError("partition must be 'cyclic' or 'slab'")


def slab_bounds(nx, world_size, rank):
    """X range ``[i0, i1)`` of ``rank``; slabs differ by at most one plane."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    if world_size > nx:
        raise ValueError(f"cannot shard {nx} planes over {world_size} ranks")
    return nx * rank // world_size, nx * (rank + 1) // world_size


class _DeviceBuffer:
    """Exposes an engine-owned device allocation to torch (zero copy)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


class ShardedBackprojection:
    """Slab-sharded ``Backprojection``: same per-view interface, one slab per rank."""

    def __init__(self, shape, origin, voxel_size, type="carving", default_value=0, rank=None,
                 world_size=None, device=None, engine_factory=None, views_per_launch=0,
                 partition="cyclic", log=False):
        if rank is None or world_size is None:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_initialized() else 0
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank, self.world_size = int(rank), int(world_size)
        self.shape = [int(s) for s in shape]
        self.origin = origin
        self.voxel_size = voxel_size
        self.default_value = default_value
        self.log = log
        self._lut = None
        if type == "carving":
            self.dtype, self._mode = np.int32, nat.SC_MODE_CARVE
        elif type == "averaging":
            self.dtype, self._mode = np.float32, nat.SC_MODE_AVERAGE
        else:
            raise ValueError(f"Unknown kernel type {type}, valid values are 'averaging' or 'carving'!")
        self.partition = partition
        self.planes = rank_planes(self.shape[0], self.world_size, self.rank, partition)
        self.device = self.rank if device is None else int(device)
        factory = engine_factory or nat.Engine
        if partition == "cyclic":
            self._engine = factory(self.shape, origin, voxel_size, self._mode,
                                   default_value=float(default_value), device=self.device,
                                   cyclic=(self.rank, self.world_size))
        else:
            self._engine = factory(self.shape, origin, voxel_size, self._mode,
                                   default_value=float(default_value), device=self.device,
                                   slab=(self.planes.start, self.planes.stop))
        if views_per_launch:
            self._engine.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, int(views_per_launch))
        self._on_gpu = engine_factory is None

    @property
    def engine(self):
        return self._engine

    @property
    def slab_shape(self):
        return (len(self.planes), self.shape[1], self.shape[2])

    def process_view(self, intrinsics, rot, tvec, mask, mask_dtype=None):
        """``Backprojection.process_view`` on this rank's planes: the same host conversions
        (cl.py:205-215: ``img_as_float32``, the ``log`` step, the dtype casts, uint8 + table for
        averaging).  ``mask_dtype`` hands an already converted mask straight to the engine."""
        if mask_dtype is not None:
            self._engine.process_view(intrinsics, rot, tvec, np.ascontiguousarray(mask), mask_dtype)
            return
        from .cl import submit_view
        self._lut = submit_view(self._engine, self.dtype, self.log, self._lut, intrinsics, rot, tvec,
                                mask, invert=False)

    def clear(self):
        self._engine.clear()

    def flush(self):
        self._engine.flush()

    def synchronize(self):
        self._engine.synchronize()

    def get_local(self):
        """This rank's planes as a host array ``[len(planes), ny, nz]`` (in ``self.planes`` order)."""
        return self._engine.get_values()

    # -- assembling the grid ----------------------------------------------------------------
    def _slab_tensor(self):
        import torch
        if self._on_gpu:
            ptr = self._engine.values_device_ptr()
            self._engine.synchronize()
            typestr = "<i4" if self.dtype == np.int32 else "<f4"
            buf = _DeviceBuffer(ptr, self._engine.num_voxels(), typestr)
            return torch.as_tensor(buf, device=f"cuda:{self.device}")
        return torch.from_numpy(np.ascontiguousarray(self.get_local()).reshape(-1))

    def _max_slab_voxels(self):
        planes = max(len(rank_planes(self.shape[0], self.world_size, r, self.partition))
                     for r in range(self.world_size))
        return planes * self.shape[1] * self.shape[2]

    def _assemble(self, per_rank):
        """Full grid (torch tensor ``[nx, ny, nz]``) from one flat, possibly padded, tensor per
        rank, placing every rank's planes at their global x indices."""
        import torch
        plane = self.shape[1] * self.shape[2]
        first = per_rank[0]
        full = torch.empty((self.shape[0], self.shape[1], self.shape[2]), dtype=first.dtype,
                           device=first.device)
        for r, flat in enumerate(per_rank):
            pl = rank_planes(self.shape[0], self.world_size, r, self.partition)
            full[pl.start:pl.stop:pl.step] = flat[: len(pl) * plane].reshape(len(pl), *self.shape[1:])
        return full

    def all_gather(self, compress=False):
        """Full grid on every rank (torch tensor on the slab's device), by all-gather.

        compress=True sends carve labels as int8 (labels are in {-1, 0, 1} when
        default_value is): 4x less xGMI traffic, widened back after the collective.
        """
        import torch
        import torch.distributed as dist
        local = self._slab_tensor()
        if compress:
            if self.dtype != np.int32:
                raise ValueError("compression is for carve labels only")
            local = local.to(torch.int8)
        if self.world_size == 1:
            return local.to(torch.int32 if compress else local.dtype).reshape(self.shape)
        pad = self._max_slab_voxels()
        send = local
        if local.numel() != pad:
            send = torch.zeros(pad, dtype=local.dtype, device=local.device)
            send[: local.numel()] = local
        recv = torch.empty(pad * self.world_size, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(recv, send)
        full = self._assemble([recv[r * pad:(r + 1) * pad] for r in range(self.world_size)])
        if compress:
            full = full.to(torch.int32)
        return full

    def all_reduce(self):
        """Full grid on every rank by summing zero-padded full-size buffers (the north
        star's wording).  Exact: every voxel is non-zero on exactly one rank."""
        import torch
        import torch.distributed as dist
        local = self._slab_tensor()
        full = torch.zeros((self.shape[0], self.shape[1], self.shape[2]), dtype=local.dtype,
                           device=local.device)
        pl = self.planes
        full[pl.start:pl.stop:pl.step] = local.reshape(len(pl), *self.shape[1:])
        if self.world_size > 1:
            dist.all_reduce(full, op=dist.ReduceOp.SUM)
        return full

    def gather_to_host(self, dst=0):
        """Full grid as a NumPy array on rank ``dst`` (None elsewhere): each rank copies its
        own slab device->host over its own PCIe link; only host memory is exchanged."""
        import torch
        import torch.distributed as dist
        local = np.ascontiguousarray(self.get_local())
        if self.world_size == 1:
            return local.reshape(self.shape)
        if dist.get_backend() == "gloo":
            t = torch.from_numpy(local.reshape(-1))
            pad = self._max_slab_voxels()
            send = torch.zeros(pad, dtype=t.dtype)
            send[: t.numel()] = t
            bufs = [torch.empty_like(send) for _ in range(self.world_size)] if self.rank == dst else None
            dist.gather(send, bufs, dst=dst)
            if self.rank != dst:
                return None
            return self._assemble(bufs).numpy()
        objs = [None] * self.world_size if self.rank == dst else None
        dist.gather_object(local, objs, dst=dst)
        if self.rank != dst:
            return None
        out = np.empty(self.shape, dtype=self.dtype)
        for r, part in enumerate(objs):
            pl = rank_planes(self.shape[0], self.world_size, r, self.partition)
            out[pl.start:pl.stop:pl.step] = part
        return out

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
