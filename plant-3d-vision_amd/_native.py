"""Loader and thin binding of ``libspacecarve.so`` (the C ABI in ``include/spacecarve.h``).

The north star asks for cffi; cffi (ABI mode, ``ffi.dlopen``) is used when it is importable
and stdlib ``ctypes`` otherwise (``SPACECARVE_FFI=cffi|ctypes`` forces one).  Both paths
pass every pointer as a plain address, so the rest of the package never sees a backend
type.  There is no CPU fallback: if the library is missing or no gfx950 device is
visible, calls raise.
"""
import ctypes
import importlib.util
import os
import subprocess
import sys
import threading

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libspacecarve.so")
# A/B measurements only (tools/, bench.py --opt runs): another build of the same library, e.g. the previous
# round's; entry points it lacks are simply not bound
_ALT_LIB = os.environ.get("SPACECARVE_LIB", "")
if _ALT_LIB:
    LIB_PATH = os.path.abspath(_ALT_LIB)
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), "include", "spacecarve.h")

# constants of include/spacecarve.h
SC_OK = 0
SC_ERR_INVALID, SC_ERR_DEVICE, SC_ERR_NOMEM, SC_ERR_STATE = -1, -2, -3, -4
SC_MODE_CARVE, SC_MODE_AVERAGE = 0, 1
SC_CREATE_DEFERRED = 1
SC_MASK_U8, SC_MASK_I32, SC_MASK_F32, SC_MASK_U8_INV, SC_MASK_BOOL_INV = 0, 1, 2, 3, 4
SC_MASK_U8_LUT = 5
SC_OPT_VIEWS_PER_LAUNCH, SC_OPT_VIEW_ORDER, SC_OPT_TIME_KERNELS, SC_OPT_MAX_PENDING = 1, 2, 3, 4
SC_OPT_COMPACT, SC_OPT_DENSE_VIEWS, SC_OPT_STAGE1_VIEWS, SC_OPT_LIST_BLOCKS = 5, 6, 7, 8
SC_OPT_VIEW_GROUP, SC_OPT_BRICK, SC_OPT_FLAG_VIEWS, SC_OPT_STAGE2_VIEWS = 9, 10, 11, 12
SC_KERNEL_CARVE, SC_KERNEL_AVERAGE, SC_KERNEL_PACK, SC_KERNEL_FILL, SC_KERNEL_LIST = 0, 1, 2, 3, 4
SC_KERNEL_FLAGS, SC_KERNEL_STEP = 5, 6
SC_OPT_PACK_ROWS, SC_OPT_DEFER_STORES, SC_OPT_DEFER_SHARE = 13, 14, 15
SC_OPT_FULL_BRICKS, SC_OPT_AVG_BRICK = 19, 20
SC_OPT_STAGE1_STORE_SHARE, SC_OPT_STAGE1_LIST_BLOCKS = 17, 21
SC_OPT_PACK_RIDE, SC_OPT_BRICK_WALKERS, SC_OPT_FILL_BLOCKS, SC_OPT_FINAL_VOXELS = 22, 23, 25, 24
SC_OPT_VIEW_BRICK, SC_OPT_AVG_TILE_F32, SC_OPT_STAGE1_VOXELS, SC_OPT_RESERVE_EVENTS = 26, 29, 30, 31
SC_OPT_BULK_MIN, SC_OPT_ITEM_BIAS, SC_OPT_UNIT_BLOCKS, SC_OPT_BULK_FLOOR = 32, 33, 34, 35
SC_OPT_BULK_ADAPT = SC_OPT_BULK_FLOOR  # deprecated name of key 35 (0 still means: the units are always asked)
SC_OPT_UNIT_CULL, SC_OPT_LIST_CAP, SC_OPT_HOST_PACK, SC_OPT_HOST_THREADS, SC_OPT_LDS_TILES, SC_OPT_BULK_LIVE, SC_OPT_SAFE_KERNELS = 37, 38, 39, 40, 41, 42, 43
SC_OPT_DENSE_EXTRA = 44
SC_OPT_SPEC_SHARE, SC_OPT_SPEC_BLOCKS, SC_OPT_LATE_ROAD = 45, 46, 47

# name -> (restype, [argtypes]); 'p' pointer, 'i' int, 'q' int64, 'f' float, 's' const char*
_SIGNATURES = {
    "sc_abi_version": ("i", []),
    "sc_last_error": ("s", []),
    "sc_device_count": ("i", ["p"]),
    "sc_create": ("i", ["p", "q", "q", "q", "p", "f", "i", "f", "i"]),
    "sc_create_slab": ("i", ["p", "q", "q", "q", "q", "q", "p", "f", "i", "f", "i"]),
    "sc_create_cyclic": ("i", ["p", "q", "q", "q", "q", "q", "p", "f", "i", "f", "i"]),
    "sc_prewarm": ("i", ["i"]),
    "sc_prewarm_wait": ("v", []),
    "sc_create_ex": ("i", ["p", "q", "q", "q", "q", "q", "q", "p", "f", "i", "f", "i", "i"]),
    "sc_setup_times": ("i", ["p", "p"]),
    "sc_destroy": ("v", ["p"]),
    "sc_clear": ("i", ["p"]),
    "sc_set_option": ("i", ["p", "i", "q"]),
    "sc_set_stream": ("i", ["p", "p"]),
    "sc_order_after": ("i", ["p", "p"]),
    "sc_order_before": ("i", ["p", "p"]),
    "sc_set_lut": ("i", ["p", "p"]),
    "sc_process_view": ("i", ["p", "p", "p", "p", "p", "i", "i", "i", "q"]),
    "sc_process_views": ("i", ["p", "i", "p", "p", "p", "p", "i", "i", "i", "q"]),
    "sc_process_views_device": ("i", ["p", "i", "p", "p", "p", "p", "i", "i", "i"]),
    "sc_process_png_views": ("i", ["p", "i", "p", "p", "p", "p", "p", "i", "i"]),
    "sc_average_labels": ("i", ["p", "i", "i", "p", "p", "p", "p", "i", "i"]),
    "sc_average_labels_fused_count": ("q", []),
    "sc_flush": ("i", ["p"]),
    "sc_synchronize": ("i", ["p"]),
    "sc_get_values": ("i", ["p", "p"]),
    "sc_get_values_i8": ("i", ["p", "p"]),
    "sc_values_device_ptr": ("i", ["p", "p"]),
    "sc_num_voxels": ("q", ["p"]),
    "sc_packed_bytes": ("q", ["q", "i"]),
    "sc_values_packed": ("i", ["p", "i", "p", "p"]),
    "sc_get_values_packed": ("i", ["p", "i", "p"]),
    "sc_get_values_wire2": ("i", ["p", "p", "p", "q", "i"]),
    "sc_widen_labels2": ("i", ["p", "q", "p", "i"]),
    "sc_hostpack_bits": ("i", ["p", "i", "i", "i", "q", "p"]),
    "sc_widen_labels2_ranks": ("i", ["p", "q", "i", "i", "q", "q", "q", "p"]),
    "sc_vol2pcd_packed": ("i", ["p", "q", "i", "i", "i", "q", "q", "q", "p", "d", "d", "p", "i", "p", "p", "p"]),
    "sc_unpack_labels": ("i", ["i", "p", "p", "q", "i", "i", "q", "q", "q", "i", "p", "i"]),
    "sc_sparse_bricks": ("q", ["q", "q", "q"]),
    "sc_sparse_rank_bytes": ("q", ["q", "q"]),
    "sc_values_sparse": ("i", ["p", "q", "p", "p"]),
    "sc_get_values_sparse": ("i", ["p", "q", "p", "q"]),
    "sc_sparse_headers": ("i", ["i", "p", "p", "p", "q", "i", "p", "p"]),
    "sc_unpack_sparse": ("i", ["i", "p", "p", "q", "i", "q", "q", "q", "p", "i"]),
    "sc_widen_sparse_ranks": ("i", ["p", "q", "i", "q", "q", "q", "p"]),
    "sc_comm_available": ("i", []),
    "sc_comm_unique_id": ("i", ["p", "q"]),
    "sc_comm_create": ("i", ["p", "p", "i", "i", "i"]),
    "sc_comm_destroy": ("v", ["p"]),
    "sc_comm_size": ("i", ["p"]),
    "sc_comm_rank": ("i", ["p"]),
    "sc_comm_stream": ("i", ["p", "p"]),
    "sc_comm_synchronize": ("i", ["p"]),
    "sc_comm_barrier": ("i", ["p"]),
    "sc_comm_all_gather": ("i", ["p", "p", "p", "q", "p"]),
    "sc_engine_stream": ("i", ["p", "p"]),
    "sc_all_gather_sparse": ("i", ["p", "p", "q", "p", "q", "i", "p", "p"]),
    "sc_sparse_wait_headers": ("i", ["p", "p", "q", "i", "p", "p"]),
    "sc_all_gather_packed": ("i", ["p", "p", "i", "p", "q", "i"]),
    "sc_kernel_stats": ("i", ["p", "i", "p", "p"]),
    "sc_reset_kernel_stats": ("i", ["p"]),
    "sc_fused_counts": ("i", ["p", "p"]),
    "sc_fused_counts_ex": ("i", ["p", "p"]),
    "sc_span_begin": ("i", ["p"]),
    "sc_span_end": ("i", ["p", "p"]),
    "sc_selftest_division": ("i", ["p", "q", "I", "i", "p", "p"]),
    "sc_selftest_project": ("i", ["p", "q", "I", "i", "p", "p", "p", "p", "p"]),
    "sc_view_certified": ("i", ["p", "f", "q", "q", "q", "p", "p", "p", "p"]),
    "sc_vol2pcd": ("i", ["p", "i", "i", "q", "q", "q", "p", "d", "d", "p", "i", "p", "p", "p"]),
    "sc_vol2pcd_last_error": ("s", []),
    "sc_vol2pcd_release": ("v", []),
    "sc_vol2pcd_set_scratch_limit": ("v", ["q"]),
    "sc_free_host": ("v", ["p"]),
    "sc_label_points": ("i", ["p", "q", "i", "i", "p", "p", "p", "p", "i", "i", "i", "i", "p", "p"]),
    "sc_label_points_last_error": ("s", []),
    "sc_create_sharded": ("i", ["p", "q", "q", "q", "p", "f", "i", "f", "p", "i", "i"]),
    "sc_group_destroy": ("v", ["p"]),
    "sc_group_size": ("i", ["p"]),
    "sc_group_engine": ("p", ["p", "i"]),
    "sc_group_clear": ("i", ["p"]),
    "sc_group_set_option": ("i", ["p", "i", "q"]),
    "sc_group_set_lut": ("i", ["p", "p"]),
    "sc_group_process_view": ("i", ["p", "p", "p", "p", "p", "i", "i", "i", "q"]),
    "sc_group_flush": ("i", ["p"]),
    "sc_group_synchronize": ("i", ["p"]),
    "sc_group_get_values": ("i", ["p", "p"]),
    "sc_png_info": ("i", ["p", "q", "p", "p"]),
    "sc_png_decode_gray8": ("i", ["p", "q", "p", "i", "i"]),
    "sc_png_last_error": ("s", []),
    "sc_host_alloc": ("i", ["i", "q", "p"]),
    "sc_host_free": ("v", ["p"]),
    "sc_dev_alloc": ("i", ["p", "q", "p"]),
    "sc_dev_free": ("i", ["p", "p"]),
    "sc_dev_upload": ("i", ["p", "p", "p", "q"]),
    "sc_dev_download": ("i", ["p", "p", "p", "q"]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "q": ctypes.c_int64, "f": ctypes.c_float,
       "I": ctypes.c_uint32, "d": ctypes.c_double,
       "s": ctypes.c_char_p, "v": None}
_CDEF = {"p": "void *", "i": "int", "q": "int64_t", "f": "float", "s": "const char *", "I": "uint32_t", "d": "double",
         "v": "void"}


class SpaceCarveError(RuntimeError):
    """A C-ABI call failed (device / memory / state)."""


def build(force=False):
    """Compile ``csrc/spacecarve.hip`` for gfx950 into ``libspacecarve.so`` (in-tree)."""
    csrc = os.path.join(_PKG_DIR, "csrc")
    deps = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile"]
    deps.append(HEADER_PATH)
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", os.path.join(_PKG_DIR, "csrc"), "-s", "-B", "all"])
    return LIB_PATH


class _CtypesBackend:
    name = "ctypes"

    def __init__(self, path):
        self.lib = ctypes.CDLL(path)
        self.fn = {}
        for name, (res, args) in _SIGNATURES.items():
            try:
                f = getattr(self.lib, name)
            except AttributeError:
                if _ALT_LIB:
                    continue
                raise
            f.restype = _CT[res]
            f.argtypes = [_CT[a] for a in args]
            self.fn[name] = f

    def call(self, name, *args):
        return self.fn[name](*args)

    @staticmethod
    def string(ret):
        return (ret or b"").decode("utf-8", "replace")


class _CffiBackend:
    name = "cffi"

    def __init__(self, path):
        import cffi  # noqa: F401  (ImportError -> caller falls back to ctypes)

        self.ffi = cffi.FFI()
        decls = []
        for name, (res, args) in _SIGNATURES.items():
            decls.append(f"{_CDEF[res]} {name}({', '.join(_CDEF[a] for a in args) or 'void'});")
        self.ffi.cdef("\n".join(decls))
        self.lib = self.ffi.dlopen(path)
        self.sig = _SIGNATURES

    def call(self, name, *args):
        kinds = self.sig[name][1]
        conv = [self.ffi.cast("void *", int(a or 0)) if k == "p" else a for k, a in zip(kinds, args)]
        return getattr(self.lib, name)(*conv)

    def string(self, ret):
        return self.ffi.string(ret).decode("utf-8", "replace") if ret else ""


_backend = None


_hip_runtime = None


def hip_runtime():
    """The HIP runtime this process uses, as a ``ctypes.CDLL`` (the one ``_preload_hip_runtime`` chose, else the one
    libspacecarve.so brought in): for diagnostics that time the runtime's own initialisation (bench.py's cold-process leg)."""
    backend()
    if _hip_runtime is not None:
        return _hip_runtime
    if "torch" in sys.modules:
        import torch
        return ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
    return ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own ``libamdhip64.so`` (same SONAME as
    /opt/rocm's); whichever is loaded first serves everybody, and torch fails if that is not its
    own.  So when torch is installed but not imported yet, load ITS runtime before our library:
    device pointers, streams and RCCL then work across torch and the engine in either import
    order.  Without torch the system ROCm runtime is used."""
    global _hip_runtime
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    lib = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(lib):
        try:
            _hip_runtime = ctypes.CDLL(lib, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def backend():
    """Load the library once.  Raises ``SpaceCarveError`` if it is not built."""
    global _backend
    if _backend is not None:
        return _backend
    if not os.path.exists(LIB_PATH):
        raise SpaceCarveError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C plant-3d-vision_amd/csrc` (needs hipcc, gfx950)")
    _preload_hip_runtime()
    want = os.environ.get("SPACECARVE_FFI", "").lower()
    if want not in ("", "cffi", "ctypes"):
        raise ValueError("SPACECARVE_FFI must be 'cffi' or 'ctypes'")
    if want != "ctypes":
        try:
            _backend = _CffiBackend(LIB_PATH)
            return _backend
        except ImportError:
            if want == "cffi":
                raise
    _backend = _CtypesBackend(LIB_PATH)
    return _backend


def last_error():
    b = backend()
    return b.string(b.call("sc_last_error"))


def check(rc, what):
    if rc == SC_OK:
        return
    msg = f"{what}: {last_error()} (code {rc})"
    if rc == SC_ERR_INVALID:
        raise ValueError(msg)
    if rc == SC_ERR_NOMEM:
        raise MemoryError(msg)
    raise SpaceCarveError(msg)


def addr(a):
    """Address of a C-contiguous ndarray's data."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def pinned_empty(shape, dtype, device=0):
    """``np.empty(shape, dtype)`` in page-locked host memory (``sc_host_alloc``): device-to-host
    copies into it run at the link's rate.  The memory is released when the last view of the
    array is collected.  Takes ~0.1 s for 512 MiB: call it off the critical path."""
    import weakref
    b = backend()
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    if nbytes <= 0:
        return np.empty(shape, dtype)
    out = np.zeros(1, dtype=np.uintp)
    check(b.call("sc_host_alloc", int(device), nbytes, addr(out)), "sc_host_alloc")
    ptr = int(out[0])
    owner = (ctypes.c_ubyte * nbytes).from_address(ptr)
    weakref.finalize(owner, b.call, "sc_host_free", ptr)
    return np.frombuffer(owner, dtype=dtype).reshape(shape)


_staging = {}
_staging_lock = threading.Lock()


def staging_ring(piece_bytes, slots=4, device=0):
    """(ring, state) -- ``slots`` page-locked pieces (one allocation per process and size, kept for its life): the
    landing place of ``Engine.get_values_staged``, and which slots still hold pieces in flight.  ~0.03 s for
    128 MiB the first time."""
    key = (int(piece_bytes), int(slots), int(device))
    with _staging_lock:
        ent = _staging.get(key)
        if ent is None:
            ring = pinned_empty((int(slots), int(piece_bytes)), np.uint8, device=device)
            ent = _staging[key] = (ring, {"next": 0, "busy": [[] for _ in range(int(slots))], "lock": threading.Lock()})
    return ent


def host_workers(limit=8):
    """Threads for host-side helpers (decode-ahead, widening): the CPUs this process may run on, at
    most ``limit``.  Measured on the GPU box (256 hardware threads visible): PNG decoding through PIL
    stops scaling at ~3x whatever the thread count (32 ms for 72 masks with 8, 16 or 32 threads), and
    16 decoders beside the page-touching threads made the whole read slower than 8."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(int(limit), avail))


def widen_i8(dst, src, workers=None):
    """``dst[...] = src`` (int8 -> int32, same shape) slab by slab on a few threads."""
    import threading
    flat_d, flat_s = dst.reshape(-1), src.reshape(-1)
    n = flat_d.size
    workers = workers or host_workers()
    parts = max(1, min(workers, n // (1 << 20) or 1))
    bounds = [n * q // parts for q in range(parts + 1)]

    def work(a, b):
        np.copyto(flat_d[a:b], flat_s[a:b], casting="unsafe")  # releases the GIL

    ths = [threading.Thread(target=work, args=(bounds[q], bounds[q + 1])) for q in range(parts)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()


class TouchedEmpty:
    """``np.empty(shape, dtype)`` whose pages are being touched on a few host threads (no HIP calls
    on them): a 512 MiB read-back into a fresh array spends 40 ms in first-touch page faults, into
    touched pages 10 ms.  Start it before the device work, call ``result()`` when the array is
    needed."""

    def __init__(self, shape, dtype, threads=None):
        import threading
        threads = threads or min(4, host_workers())  # first-touch faults do not scale past a few threads
        self._arr = np.empty(shape, dtype=dtype)
        flat = self._arr.reshape(-1)
        step = max(1, 4096 // self._arr.itemsize)
        n = flat.size
        parts = max(1, min(int(threads), n // (step * 256) or 1))
        bounds = [(n * q // parts) // step * step for q in range(parts)] + [n]

        def work(a, b):
            flat[a:b:step] = 0  # one write per page; NumPy releases the GIL

        self._threads = [threading.Thread(target=work, args=(bounds[q], bounds[q + 1]), daemon=True)
                         for q in range(parts)]
        for th in self._threads:
            th.start()

    def result(self):
        for th in self._threads:
            th.join()
        self._threads = []
        return self._arr


def average_labels(engines, K, R, t, mask_ptrs, n_views, H, W):
    """``sc_average_labels``: the labels of one scan in one launch -- ``engines[l]`` (averaging, its table set)
    takes ``mask_ptrs[l]`` (uint8 ``[n_views][H][W]`` in device memory); one set of poses."""
    K, R, t = Engine._pose(K, R, t)
    if K.size != 4 * n_views or R.size != 9 * n_views or t.size != 3 * n_views:
        raise ValueError("pose arrays do not match the view count")
    if len(engines) != len(mask_ptrs) or not engines:
        raise ValueError("one mask stack per engine")
    handles = np.array([int(e._h) for e in engines], dtype=np.uintp)
    ptrs = np.array([int(p) for p in mask_ptrs], dtype=np.uintp)
    check(backend().call("sc_average_labels", addr(handles), len(engines), int(n_views), addr(K), addr(R), addr(t),
                         addr(ptrs), int(H), int(W)), "sc_average_labels")


def packed_bytes(voxels, bits):
    """Bytes of ``voxels`` labels packed at ``bits`` bits each (whole 16-byte groups)."""
    return int(backend().call("sc_packed_bytes", int(voxels), int(bits)))


def widen_labels2(packed, voxels, out=None, threads=None):
    """``sc_widen_labels2``: ``voxels`` labels at 2 bits each (uint32 words, 16 per word) -> int32, on host threads
    inside the library (no device needed)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint32).reshape(-1)
    if packed.size < (int(voxels) + 15) // 16:
        raise ValueError("too few packed words")
    if out is None:
        out = np.empty(int(voxels), dtype=np.int32)
    if out.dtype != np.int32 or out.size != int(voxels) or not out.flags["C_CONTIGUOUS"]:
        raise ValueError("output buffer has the wrong dtype/size/layout")
    check(backend().call("sc_widen_labels2", addr(packed), int(voxels), addr(out),
                         int(threads if threads is not None else host_workers(16))), "sc_widen_labels2")
    return out


def unpack_labels(device, stream_ptr, recv_ptr, rank_bytes, world, partition, shape, bits, out_ptr, out_bytes):
    """``sc_unpack_labels``: the ranks' packed planes (device memory) -> one grid in global order."""
    check(backend().call("sc_unpack_labels", int(device), int(stream_ptr or 0), int(recv_ptr), int(rank_bytes), int(world),
                         0 if partition == "cyclic" else 1, int(shape[0]), int(shape[1]), int(shape[2]), int(bits),
                         int(out_ptr), int(out_bytes)), "sc_unpack_labels")


def sparse_bricks(planes, ny, nz):
    """Bricks (16 columns x 64 voxels of one x-plane) of ``planes`` planes: the units of the sparse label form."""
    n = backend().call("sc_sparse_bricks", int(planes), int(ny), int(nz))
    if n < 0:
        raise ValueError("bad shape for the sparse label form")
    return int(n)


def sparse_rank_bytes(nbricks, cap):
    """Bytes of one rank's sparse label buffer: header + codes + ``cap`` payload slots (``include/spacecarve.h``)."""
    n = backend().call("sc_sparse_rank_bytes", int(nbricks), int(cap))
    if n < 0:
        raise ValueError("bad brick count / capacity")
    return int(n)


def sparse_headers(device, stream_ptr, recv_ptr, rank_bytes, world, done_event=0):
    """(nmixed[world], cap[world]) of gathered sparse buffers in device memory; waits for ``done_event`` (what
    ``Engine.all_gather_sparse`` returned) or, without one, for ``stream_ptr``."""
    nm = np.zeros(int(world), dtype=np.uint32)
    cp = np.zeros(int(world), dtype=np.uint32)
    check(backend().call("sc_sparse_headers", int(device), int(stream_ptr or 0), int(done_event or 0), int(recv_ptr), int(rank_bytes),
                         int(world), addr(nm), addr(cp)), "sc_sparse_headers")
    return nm, cp


def sparse_wait_headers(done_event, headers_host, rank_bytes, world):
    """(nmixed[world], cap[world]) of the gather ``Engine.all_gather_sparse`` returned (event, host headers) for."""
    nm = np.zeros(int(world), dtype=np.uint32)
    cp = np.zeros(int(world), dtype=np.uint32)
    check(backend().call("sc_sparse_wait_headers", int(done_event), int(headers_host), int(rank_bytes), int(world), addr(nm), addr(cp)),
          "sc_sparse_wait_headers")
    return nm, cp


def unpack_sparse(device, stream_ptr, recv_ptr, rank_bytes, world, shape, out_ptr, out_kind):
    """``sc_unpack_sparse``: the ranks' gathered sparse buffers -> ONE grid in global order on the device;
    ``out_kind`` 4 int32 labels, 1 int8 labels, 0 uint8 occupancy (label == 1)."""
    check(backend().call("sc_unpack_sparse", int(device), int(stream_ptr or 0), int(recv_ptr), int(rank_bytes), int(world),
                         int(shape[0]), int(shape[1]), int(shape[2]), int(out_ptr), int(out_kind)), "sc_unpack_sparse")


def widen_sparse_ranks(packed, rank_bytes, world, shape, out=None):
    """Host code only: the ranks' sparse buffers (a uint8 array, ``rank_bytes`` apart) -> the int32 grid in global
    order -- the host end of ``ShardedBackprojection.gather_to_host`` (cl.py:229-232 of a sharded run)."""
    packed = np.ascontiguousarray(packed).view(np.uint8).reshape(-1)
    if packed.size < int(rank_bytes) * int(world):
        raise ValueError("buffer smaller than world x rank_bytes")
    n = int(np.prod(shape))
    if out is None or out.dtype != np.int32 or out.size != n or not out.flags["C_CONTIGUOUS"]:
        out = np.empty(n, dtype=np.int32)
    check(backend().call("sc_widen_sparse_ranks", addr(packed), int(rank_bytes), int(world), int(shape[0]), int(shape[1]),
                         int(shape[2]), addr(out.reshape(-1))), "sc_widen_sparse_ranks")
    return out.reshape(tuple(int(s) for s in shape))


class Comm:
    """One RCCL communicator of this process (``sc_comm_*``): the library enqueues the collectives itself, no torch.
    ``unique_id()`` on rank 0, the 128 bytes to every rank by the host's own means, then ``Comm(id, world, rank, device)``
    on every rank (collective)."""

    ID_BYTES = 128

    @staticmethod
    def available():
        """librccl can be opened (nothing else is done; raises with the loader's message otherwise)."""
        check(backend().call("sc_comm_available"), "sc_comm_available")
        return True

    @staticmethod
    def unique_id():
        buf = np.zeros(Comm.ID_BYTES, dtype=np.uint8)
        check(backend().call("sc_comm_unique_id", addr(buf), Comm.ID_BYTES), "sc_comm_unique_id")
        return buf.tobytes()

    def __init__(self, unique_id, world_size, rank, device=0):
        self._b = backend()
        uid = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        if uid.size != Comm.ID_BYTES:
            raise ValueError("the id is %d bytes" % Comm.ID_BYTES)
        out = np.zeros(1, dtype=np.uintp)
        check(self._b.call("sc_comm_create", addr(out), addr(uid), int(world_size), int(rank), int(device)), "sc_comm_create")
        self._h = int(out[0])
        self.world_size, self.rank, self.device = int(world_size), int(rank), int(device)

    @property
    def handle(self):
        return self._h

    def stream(self):
        out = np.zeros(1, dtype=np.uintp)
        check(self._b.call("sc_comm_stream", self._h, addr(out)), "sc_comm_stream")
        return int(out[0])

    def synchronize(self):
        check(self._b.call("sc_comm_synchronize", self._h), "sc_comm_synchronize")

    def barrier(self):
        """Every rank has called this when it returns (a 16-byte all-gather, waited for on the host)."""
        check(self._b.call("sc_comm_barrier", self._h), "sc_comm_barrier")

    def all_gather(self, send_ptr, recv_ptr, bytes_per_rank, stream_ptr=0):
        check(self._b.call("sc_comm_all_gather", self._h, int(send_ptr), int(recv_ptr), int(bytes_per_rank), int(stream_ptr or 0)),
              "sc_comm_all_gather")

    def close(self):
        if getattr(self, "_h", 0):
            self._b.call("sc_comm_destroy", self._h)
            self._h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def view_certified(shape, origin, voxel_size, K, R, t):
    """Does a view with this pose take the kernels' certified (cheaper, same results) projection path on
    this grid?  Host arithmetic only (``sc_view_certified``)."""
    o = np.ascontiguousarray(np.asarray(origin, dtype=np.float32).reshape(3))
    K = np.ascontiguousarray(np.asarray(K, dtype=np.float32).reshape(4))
    R = np.ascontiguousarray(np.asarray(R, dtype=np.float32).reshape(9))
    t = np.ascontiguousarray(np.asarray(t, dtype=np.float32).reshape(3))
    out = np.zeros(1, dtype=np.int32)
    check(backend().call("sc_view_certified", addr(o), float(voxel_size), int(shape[0]), int(shape[1]), int(shape[2]),
                         addr(K), addr(R), addr(t), addr(out)), "sc_view_certified")
    return bool(out[0])


def pose_records(entries):
    """``[n, 28]`` uint32 array of ``sc_selftest_project`` pose records from an iterable of
    ``(K[4], R[9], t[3], origin[3], voxel_size, W, H, shape[3] or None)``."""
    entries = list(entries)
    rec = np.zeros((len(entries), 28), dtype=np.uint32)
    f = rec.view(np.float32)
    i = rec.view(np.int32)
    for q, (K, R, t, origin, vs, W, H, shape) in enumerate(entries):
        f[q, 0:4] = np.asarray(K, dtype=np.float32).reshape(4)
        f[q, 4:13] = np.asarray(R, dtype=np.float32).reshape(9)
        f[q, 13:16] = np.asarray(t, dtype=np.float32).reshape(3)
        f[q, 16:19] = np.asarray(origin, dtype=np.float32).reshape(3)
        f[q, 19] = np.float32(vs)
        i[q, 20], i[q, 21] = int(W), int(H)
        if shape is not None:
            i[q, 22:25] = [int(x) for x in shape]
    return rec


def widen_labels2_ranks(packed, rank_bytes, world, partition, shape, out=None):
    """``sc_widen_labels2_ranks``: ``world`` ranks' planes at 2 bits per label, rank-major and ``rank_bytes`` apart
    (a gather of the ranks' ``values_packed(2)`` buffers), into one int32 grid in global plane order (host code)."""
    packed = np.ascontiguousarray(packed).view(np.uint32).reshape(-1)
    shape = [int(s) for s in shape]
    if packed.size * 4 < int(rank_bytes) * int(world):
        raise ValueError("packed buffer smaller than world * rank_bytes")
    if out is None:
        out = np.empty(shape, dtype=np.int32)
    if out.dtype != np.int32 or out.size != int(np.prod(shape)) or not out.flags["C_CONTIGUOUS"]:
        raise ValueError("output buffer has the wrong dtype/size/layout")
    check(backend().call("sc_widen_labels2_ranks", addr(packed), int(rank_bytes), int(world),
                         0 if partition == "cyclic" else 1, shape[0], shape[1], shape[2], addr(out)), "sc_widen_labels2_ranks")
    return out.reshape(shape)


def hostpack_bits(mask, dtype_code=None):
    """``sc_hostpack_bits``: the bit form in which a carve mask handed to ``process_view`` crosses PCIe -- uint32
    ``[H][(W + 31) // 32]``, pixel ``u`` of a row at bit ``u & 31`` of word ``u >> 5`` (host code, no device needed)."""
    mask = np.asarray(mask)
    if mask.ndim != 2:
        raise ValueError("mask must be 2-D")
    if dtype_code is None:
        dtype_code = SC_MASK_I32 if mask.dtype == np.int32 else SC_MASK_U8
    want = np.int32 if dtype_code == SC_MASK_I32 else np.uint8
    if mask.dtype == np.bool_ and want is np.uint8:
        mask = mask.view(np.uint8)
    if mask.dtype != want or mask.strides[1] != mask.itemsize:
        mask = np.ascontiguousarray(mask, dtype=want)
    H, W = mask.shape
    out = np.empty((H, (W + 31) // 32), dtype=np.uint32)
    # (rows may be padded -- a view of a wider array: the row stride goes along)
    check(backend().call("sc_hostpack_bits", int(mask.ctypes.data), H, W, int(dtype_code), int(mask.strides[0]), addr(out)),
          "sc_hostpack_bits")
    return out


def png_decode_gray8(raw):
    """``(H, W)`` uint8 array of an 8-bit greyscale, non-interlaced PNG given as bytes, or ``None`` when
    the file is anything else (the caller then uses its usual reader).  The foreign call releases the
    interpreter lock: decode-ahead threads really run side by side."""
    b = backend()
    buf = np.frombuffer(raw, dtype=np.uint8)
    wh = np.zeros(2, dtype=np.int32)
    if b.call("sc_png_info", addr(buf), int(buf.size), addr(wh), addr(wh) + 4) != SC_OK:
        return None
    W, H = int(wh[0]), int(wh[1])
    try:
        out = np.empty((H, W), dtype=np.uint8)
    except MemoryError:  # a header that lies about the size: the usual reader gets the file (and its error)
        return None
    if b.call("sc_png_decode_gray8", addr(buf), int(buf.size), addr(out), W, H) != SC_OK:
        return None
    return out


_prewarm_registered = False


def prewarm(device=None):
    """``sc_prewarm``: bring the HIP runtime up and make the device's first stream on a thread of the library's (what the
    reference's module-global context and queue cost at import, cl.py:29-30, without blocking the importer).  Nothing
    happens without a ROCm device node, with ``SC_PREWARM=0``, or when the library is not built."""
    if os.environ.get("SC_PREWARM", "1") == "0" or not os.path.exists("/dev/kfd") or not os.path.exists(LIB_PATH):
        return False
    if device is None:
        device = int(os.environ.get("SC_DEVICE", os.environ.get("LOCAL_RANK", "0")) or 0)
    try:
        b = backend()
        ok = b.call("sc_prewarm", int(device)) == SC_OK
        global _prewarm_registered
        if ok and not _prewarm_registered:
            # the interpreter must not go down (exit handlers, the runtime's among them) while that thread is inside hipInit
            import atexit
            atexit.register(b.call, "sc_prewarm_wait")
            _prewarm_registered = True
        return ok
    except Exception:  # noqa: BLE001  (an optimisation only)
        return False


def device_count():
    out = np.zeros(1, dtype=np.int32)
    check(backend().call("sc_device_count", addr(out)), "sc_device_count")
    return int(out[0])


class Engine:
    """Owning handle of one ``sc_engine`` (whole grid or an X-slab of it)."""

    def __init__(self, shape, origin, voxel_size, mode, default_value=0.0, device=0, slab=None,
                 cyclic=None, deferred=False):
        """slab=(i0, i1): the engine owns x-planes [i0, i1); cyclic=(first, stride): planes
        first, first+stride, ... ; neither: the whole grid.  deferred: ``SC_CREATE_DEFERRED`` -- the arguments are
        judged now, the device half of the set-up runs on a thread of the library's and the first call that needs
        the device joins it (and raises what it failed with)."""
        self._b = backend()
        self._h = 0
        nx, ny, nz = (int(s) for s in shape)
        origin32 = np.ascontiguousarray(np.asarray(origin, dtype=np.float32).reshape(3))
        out = np.zeros(1, dtype=np.uintp)
        self.planes = None
        if slab is not None and cyclic is not None:
            raise ValueError("slab and cyclic are exclusive")
        if deferred and cyclic is None:
            i0, i1 = (0, nx) if slab is None else (int(slab[0]), int(slab[1]))
            rc = self._b.call("sc_create_ex", addr(out), nx, ny, nz, i0, 1, i1 - i0, addr(origin32),
                              float(np.float32(voxel_size)), int(mode), float(default_value), int(device), SC_CREATE_DEFERRED)
            self.slab = (i0, i1)
        elif cyclic is not None:
            first, stride = int(cyclic[0]), int(cyclic[1])
            rc = self._b.call("sc_create_cyclic", addr(out), nx, ny, nz, first, stride, addr(origin32),
                              float(np.float32(voxel_size)), int(mode), float(default_value),
                              int(device))
            self.planes = range(first, nx, stride)
            self.slab = (first, nx)
        elif slab is None:
            rc = self._b.call("sc_create", addr(out), nx, ny, nz, addr(origin32),
                              float(np.float32(voxel_size)), int(mode), float(default_value),
                              int(device))
            self.slab = (0, nx)
        else:
            i0, i1 = int(slab[0]), int(slab[1])
            rc = self._b.call("sc_create_slab", addr(out), nx, ny, nz, i0, i1, addr(origin32),
                              float(np.float32(voxel_size)), int(mode), float(default_value),
                              int(device))
            self.slab = (i0, i1)
        check(rc, "sc_create")
        self._h = int(out[0])
        self.mode = int(mode)
        self.shape = (nx, ny, nz)
        if self.planes is None:
            self.planes = range(self.slab[0], self.slab[1])
        self.slab_shape = (len(self.planes), ny, nz)
        self.dtype = np.int32 if mode == SC_MODE_CARVE else np.float32
        self.device = int(device)

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if self._h:
            self._b.call("sc_destroy", self._h)
            self._h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        if not self._h:
            raise SpaceCarveError("engine is closed")
        check(self._b.call(name, self._h, *args), name)

    # -- options ----------------------------------------------------------------------
    def set_option(self, key, value):
        self._call("sc_set_option", int(key), int(value))

    def set_lut(self, lut):
        lut = np.ascontiguousarray(np.asarray(lut, dtype=np.float32).reshape(-1))
        if lut.size != 256:
            raise ValueError("the table has 256 entries")
        self._call("sc_set_lut", addr(lut))

    def set_stream(self, stream_ptr):
        """Adopt a caller's (non-default) HIP stream; 0 restores the engine's own."""
        self._call("sc_set_stream", int(stream_ptr or 0))

    def order_after(self, stream_ptr):
        """Everything enqueued from now on runs after what ``stream_ptr`` holds so far
        (0 = the legacy default stream, e.g. torch's default stream)."""
        self._call("sc_order_after", int(stream_ptr or 0))

    def order_before(self, stream_ptr):
        """What ``stream_ptr`` (0 = the legacy default stream) is given from now on runs after everything the
        engine has enqueued so far."""
        self._call("sc_order_before", int(stream_ptr or 0))

    # -- work -------------------------------------------------------------------------
    def clear(self):
        self._call("sc_clear")

    @staticmethod
    def _pose(K, R, t):
        K = np.ascontiguousarray(np.asarray(K, dtype=np.float32).reshape(-1))
        R = np.ascontiguousarray(np.asarray(R, dtype=np.float32).reshape(-1))
        t = np.ascontiguousarray(np.asarray(t, dtype=np.float32).reshape(-1))
        return K, R, t

    def process_view(self, K, R, t, mask, mask_dtype):
        K, R, t = self._pose(K, R, t)
        if K.size != 4 or R.size != 9 or t.size != 3:
            raise ValueError("need intrinsics[4], rot[9], tvec[3]")
        if mask.ndim != 2:
            raise ValueError("mask must be 2-D (H, W)")
        mask = np.ascontiguousarray(mask)
        H, W = mask.shape
        self._call("sc_process_view", addr(K), addr(R), addr(t), addr(mask), H, W,
                   int(mask_dtype), 0)

    def process_png_views(self, K, R, t, files, invert=False, threads=0):
        """``sc_process_png_views``: ``files`` is a list of bytes-like objects holding 8-bit greyscale PNG masks;
        decoded and reduced to bits on host threads inside the library, enqueued in the order given (carve engines).
        Raises ``ValueError`` when a file is not such a PNG (nothing is enqueued then)."""
        n = len(files)
        K, R, t = self._pose(K, R, t)
        if K.size != 4 * n or R.size != 9 * n or t.size != 3 * n:
            raise ValueError("pose arrays do not match the file count")
        bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
        ptrs = np.array([b.ctypes.data for b in bufs], dtype=np.uintp)
        sizes = np.array([b.size for b in bufs], dtype=np.int64)
        self._call("sc_process_png_views", n, addr(K), addr(R), addr(t), addr(ptrs), addr(sizes), 1 if invert else 0, int(threads))

    def process_views_device(self, K, R, t, masks_dev, n_views, H, W, mask_dtype):
        K, R, t = self._pose(K, R, t)
        if K.size != 4 * n_views or R.size != 9 * n_views or t.size != 3 * n_views:
            raise ValueError("pose arrays do not match the view count")
        self._call("sc_process_views_device", int(n_views), addr(K), addr(R), addr(t),
                   int(masks_dev), int(H), int(W), int(mask_dtype))

    def flush(self):
        self._call("sc_flush")

    def synchronize(self):
        self._call("sc_synchronize")

    def get_values(self, out=None):
        if out is None:
            out = np.empty(self.slab_shape, dtype=self.dtype)
        if out.dtype != self.dtype or out.size != int(np.prod(self.slab_shape)) \
                or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("output buffer has the wrong dtype/size/layout")
        self._call("sc_get_values", addr(out))
        return out

    def get_values_i8(self, out=None):
        """Carve labels as int8 (a quarter of the bytes over PCIe)."""
        if out is None:
            out = np.empty(self.slab_shape, dtype=np.int8)
        if out.dtype != np.int8 or out.size != int(np.prod(self.slab_shape)) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("output buffer has the wrong dtype/size/layout")
        self._call("sc_get_values_i8", addr(out))
        return out

    def get_values_wire2(self, out, staging=None, threads=None):
        """Carve labels into the int32 array ``out``: 2 bits each over PCIe in pieces (into a page-locked buffer the
        engine keeps), widened by the library's host pool as the pieces land (``sc_get_values_wire2``).  ``staging``
        and ``threads`` are ignored (round 3 arguments)."""
        if out.dtype != np.int32 or out.size != int(np.prod(self.slab_shape)) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("output buffer has the wrong dtype/size/layout")
        self._call("sc_get_values_wire2", addr(out), 0, 0, 0)
        return out

    def values_device_ptr(self):
        out = np.zeros(1, dtype=np.uintp)
        self._call("sc_values_device_ptr", addr(out))
        return int(out[0])

    def values_packed(self, bits=2):
        """(device pointer, bytes) of the carve labels at ``bits`` (2 or 1) bits each (``sc_values_packed``)."""
        ptr = np.zeros(1, dtype=np.uintp)
        nbytes = np.zeros(1, dtype=np.int64)
        self._call("sc_values_packed", int(bits), addr(ptr), addr(nbytes))
        return int(ptr[0]), int(nbytes[0])

    def get_values_packed(self, bits=2):
        """The packed labels as a host array of 32-bit words."""
        per = 32 // int(bits)
        out = np.empty((self.num_voxels() + per - 1) // per, dtype=np.uint32)
        self._call("sc_get_values_packed", int(bits), addr(out))
        return out

    def setup_times(self):
        """(ms the device half of the set-up took, ms the first call that needed the device waited for it)."""
        out = np.zeros(2, dtype=np.float64)
        self._call("sc_setup_times", addr(out))
        return float(out[0]), float(out[1])

    def values_sparse(self, cap=0):
        """(device pointer, bytes) of the labels in the brick-sparse form (``sc_values_sparse``); work may still run."""
        out = np.zeros(1, dtype=np.uintp)
        nb = np.zeros(1, dtype=np.int64)
        self._call("sc_values_sparse", int(cap), addr(out), addr(nb))
        return int(out[0]), int(nb[0])

    def get_values_sparse(self, cap=0):
        """The sparse buffer in host memory (uint8 array); ``header`` fields via ``sparse_header``."""
        nbricks = sparse_bricks(self.slab_shape[0], self.slab_shape[1], self.slab_shape[2])
        cap = int(cap) if cap else max(1024, nbricks // 8)
        out = np.zeros(sparse_rank_bytes(nbricks, cap), dtype=np.uint8)
        self._call("sc_get_values_sparse", cap, addr(out), out.size)
        return out

    def stream(self):
        out = np.zeros(1, dtype=np.uintp)
        self._call("sc_engine_stream", addr(out))
        return int(out[0])

    def all_gather_sparse(self, comm, cap, recv_ptr, rank_stride, overlap=False):
        """Pack + all-gather through the library's communicator; returns the event recorded behind the collective."""
        ev = np.zeros(2, dtype=np.uintp)
        self._call("sc_all_gather_sparse", comm.handle, int(cap), int(recv_ptr), int(rank_stride), 1 if overlap else 0, addr(ev),
                   addr(ev) + 8)
        return int(ev[0]), int(ev[1])

    def all_gather_packed(self, comm, bits, recv_ptr, rank_stride, overlap=False):
        self._call("sc_all_gather_packed", comm.handle, int(bits), int(recv_ptr), int(rank_stride), 1 if overlap else 0)

    def num_voxels(self):
        return int(self._b.call("sc_num_voxels", self._h))

    # -- timing -----------------------------------------------------------------------
    def kernel_stats(self, kernel_id):
        n = np.zeros(1, dtype=np.int64)
        ms = np.zeros(1, dtype=np.float64)
        self._call("sc_kernel_stats", int(kernel_id), addr(n), addr(ms))
        return int(n[0]), float(ms[0])

    def reset_kernel_stats(self):
        self._call("sc_reset_kernel_stats")

    def span_begin(self):
        """First event of a pair on the engine's stream (see ``span_end``)."""
        self._call("sc_span_begin")

    def span_end(self):
        """Second event of the pair: waits for it, returns the milliseconds since ``span_begin``."""
        out = np.zeros(1, dtype=np.float64)
        self._call("sc_span_end", addr(out))
        return float(out[0])

    def fused_counts(self):
        """(live bricks, voxels alive after the dense stage, after the first list stage, overflow)
        of the last fused carve launch."""
        out = np.zeros(4, dtype=np.int64)
        self._call("sc_fused_counts", addr(out))
        return tuple(int(x) for x in out)

    def fused_counts_ex(self):
        """``fused_counts()`` as a dict, plus ``late_bricks`` and ``bulk_units``."""
        out = np.zeros(8, dtype=np.int64)
        self._call("sc_fused_counts_ex", addr(out))
        return {"live_bricks": int(out[0]), "alive_after_dense_stage": int(out[1]),
                "alive_after_first_list_stage": int(out[2]), "list_overflow": int(out[3]), "late_bricks": int(out[4]),
                "bulk_units": int(out[5]), "unit_items": int(out[6])}

    def selftest_division(self, count, seed=1, mode=1):
        """(mismatches, fast_pairs) of the shared-reciprocal division vs hipcc's IEEE division."""
        out = np.zeros(2, dtype=np.uint64)
        self._call("sc_selftest_division", int(count), int(seed), int(mode), addr(out), addr(out) + 8)
        return int(out[0]), int(out[1])

    def selftest_project(self, poses, count=None, seed=1, ijk=None, pose_idx=None, words=True,
                         digests=False):
        """Result words (and/or digests per 65536 samples) of the kernels' projection on explicit
        (``ijk`` [+ ``pose_idx``]) or hashed (``count``, ``seed``) samples; ``poses`` is an array
        ``[n, 28]`` of 32-bit words (``pose_records``).  See include/spacecarve.h."""
        poses = np.ascontiguousarray(poses)
        if poses.dtype.itemsize != 4 or poses.ndim != 2 or poses.shape[1] != 28:
            raise ValueError("poses must be [n, 28] 32-bit words")
        pi = jk = None
        if ijk is not None:
            jk = np.ascontiguousarray(np.asarray(ijk, dtype=np.int32).reshape(-1, 3))
            count = jk.shape[0]
            if pose_idx is not None:
                pi = np.ascontiguousarray(np.asarray(pose_idx, dtype=np.int32).reshape(-1))
                if pi.size != count:
                    raise ValueError("one pose index per sample")
        count = int(count)
        w = np.empty(count, dtype=np.uint32) if words else None
        d = np.empty((count + 65535) >> 16, dtype=np.uint64) if digests else None
        self._call("sc_selftest_project", count, int(seed), int(poses.shape[0]), addr(poses),
                   addr(jk) if jk is not None else 0, addr(pi) if pi is not None else 0,
                   addr(w) if w is not None else 0, addr(d) if d is not None else 0)
        return w, d

    # -- device memory helpers --------------------------------------------------------
    def dev_alloc(self, nbytes):
        out = np.zeros(1, dtype=np.uintp)
        self._call("sc_dev_alloc", int(nbytes), addr(out))
        return int(out[0])

    def dev_free(self, ptr):
        self._call("sc_dev_free", int(ptr))

    def dev_upload(self, dst_dev, src):
        src = np.ascontiguousarray(src)
        self._call("sc_dev_upload", int(dst_dev), addr(src), int(src.nbytes))

    def dev_download(self, dst, src_dev):
        assert dst.flags["C_CONTIGUOUS"]
        self._call("sc_dev_download", addr(dst), int(src_dev), int(dst.nbytes))

    def get_values_staged(self, out, fn, piece_bytes=32 << 20, workers=None, pool=None, slots=4):
        """The volume through a page-locked ring into ``out``: a piece crosses PCIe into the ring by DMA (no staging
        copy inside the runtime: a 512 MiB float volume crosses in ~10 ms instead of ~12 into pageable memory) and
        ``fn(src_piece, dst_piece)`` -- e.g. ``np.exp(src, out=dst)`` + clip (tasks/cl.py:172-174) -- takes it from
        there to its place in ``out`` on host threads while the next pieces cross: ONE pass over host memory where a
        copy and a transform in place were two.  ``pool``: a caller's ``ThreadPoolExecutor`` -- the pieces' futures
        are returned instead of waited for, so the next volume's copy follows this one's at once (the ring remembers
        which of its slots still hold pieces in flight, across calls; one staged read-back at a time per process)."""
        from concurrent.futures import ThreadPoolExecutor
        # (`out` may be of another dtype than the volume -- the reference's float64 label array: `fn` converts)
        if out.size != int(np.prod(self.slab_shape)) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("output buffer has the wrong size/layout")
        flat = out.reshape(-1)
        item = np.dtype(self.dtype).itemsize
        self.flush()
        src = self.values_device_ptr()  # (a snapshot without the row padding when the grid has some)
        self.synchronize()
        step = max(1, int(piece_bytes) // item)
        ring, state = staging_ring(step * item, slots, self.device)
        ring = ring.view(self.dtype)  # [slots][step]
        nw = int(workers or host_workers())
        sub = max(1, step // nw)
        own = pool is None
        if own:
            pool = ThreadPoolExecutor(max_workers=nw)
        futs = []
        try:
            with state["lock"]:
                for a in range(0, flat.size, step):
                    b = min(flat.size, a + step)
                    s = state["next"] % slots
                    state["next"] += 1
                    # the slot's previous piece has left the ring -- or failed: either way the slot is free afterwards
                    # (a `fn` that raised once must not poison the ring for every later read-back)
                    prev, state["busy"][s] = state["busy"][s], []
                    err = None
                    for f in prev:
                        try:
                            f.result()
                        except BaseException as ex:  # noqa: BLE001
                            err = err or ex
                    if err is not None:
                        raise err
                    land = ring[s][: b - a]
                    self.dev_download(land, src + a * item)
                    state["busy"][s] = [pool.submit(fn, land[c - a:min(b, c + sub) - a], flat[c:min(b, c + sub)])
                                        for c in range(a, b, sub)]
                    futs += state["busy"][s]
            if not own:
                return futs
            for f in futs:
                f.result()
        except BaseException:
            # nothing of this call may stay in flight into the caller's `out`, and no failed piece may stay on a slot
            with state["lock"]:
                for slot in state["busy"]:
                    for f in slot:
                        try:
                            f.result()
                        except BaseException:  # noqa: BLE001
                            pass
                    slot.clear()
            raise
        finally:
            if own:
                pool.shutdown(wait=True)
        return out

    def get_values_pipelined(self, out, on_piece, piece_bytes=32 << 20, workers=None, pool=None):
        """The volume into ``out`` (the reference's dtype, C order) in pieces of ``piece_bytes``, and ``on_piece(view)``
        -- a host function over the piece that has just landed, e.g. ``np.exp`` + clip (tasks/cl.py:172-174) -- on a
        few host threads WHILE the next pieces cross PCIe.  A 512 MiB float volume takes ~11 ms to cross and ~10 ms
        of ``exp`` on 8 threads: one after the other they are 21 ms per label, this way the copy alone.
        ``on_piece`` must release the interpreter lock for the overlap to happen (NumPy's loops do).
        ``pool``: a caller's ``ThreadPoolExecutor`` -- the pieces' futures are returned instead of waited for, so that
        the next volume's copy follows this one's at once (several labels: one PCIe link, no gap between them)."""
        from concurrent.futures import ThreadPoolExecutor
        if out.dtype != self.dtype or out.size != int(np.prod(self.slab_shape)) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("output buffer has the wrong dtype/size/layout")
        flat = out.reshape(-1)
        self.flush()
        src = self.values_device_ptr()  # (a snapshot without the row padding when the grid has some)
        self.synchronize()
        step = max(1, int(piece_bytes) // flat.itemsize)
        nw = int(workers or host_workers())
        sub = max(1, step // nw)
        own = pool is None
        if own:
            pool = ThreadPoolExecutor(max_workers=nw)
        try:
            futs = []
            for a in range(0, flat.size, step):
                b = min(flat.size, a + step)
                self.dev_download(flat[a:b], src + a * flat.itemsize)  # (blocks this thread, not the pool's)
                for c in range(a, b, sub):
                    futs.append(pool.submit(on_piece, flat[c:min(b, c + sub)]))
            if not own:
                return futs
            for f in futs:
                f.result()
        except BaseException:
            # nothing of this call may stay in flight into the caller's `out`, and no failed piece may stay on a slot
            with state["lock"]:
                for slot in state["busy"]:
                    for f in slot:
                        try:
                            f.result()
                        except BaseException:  # noqa: BLE001
                            pass
                    slot.clear()
            raise
        finally:
            if own:
                pool.shutdown(wait=True)
        return out


class EngineGroup:
    """Several devices driven from this process (``sc_create_sharded``): same surface as ``Engine``
    for what ``Backprojection`` needs; the grid comes back whole, in global order."""

    def __init__(self, shape, origin, voxel_size, mode, devices, default_value=0.0, partition="cyclic"):
        self._b = backend()
        self._h = 0
        nx, ny, nz = (int(s) for s in shape)
        devs = np.ascontiguousarray(np.asarray(list(devices), dtype=np.int32))
        origin32 = np.ascontiguousarray(np.asarray(origin, dtype=np.float32).reshape(3))
        out = np.zeros(1, dtype=np.uintp)
        if partition not in ("cyclic", "slab"):
            raise ValueError("partition must be 'cyclic' or 'slab'")
        check(self._b.call("sc_create_sharded", addr(out), nx, ny, nz, addr(origin32), float(np.float32(voxel_size)),
                           int(mode), float(default_value), addr(devs), int(devs.size),
                           0 if partition == "cyclic" else 1), "sc_create_sharded")
        self._h = int(out[0])
        self.mode = int(mode)
        self.shape = self.slab_shape = (nx, ny, nz)
        self.dtype = np.int32 if mode == SC_MODE_CARVE else np.float32
        self.devices = [int(d) for d in devs]

    def close(self):
        if self._h:
            self._b.call("sc_group_destroy", self._h)
            self._h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        if not self._h:
            raise SpaceCarveError("engine group is closed")
        check(self._b.call(name, self._h, *args), name)

    def set_option(self, key, value):
        self._call("sc_group_set_option", int(key), int(value))

    def set_lut(self, lut):
        lut = np.ascontiguousarray(np.asarray(lut, dtype=np.float32).reshape(-1))
        if lut.size != 256:
            raise ValueError("the table has 256 entries")
        self._call("sc_group_set_lut", addr(lut))

    def clear(self):
        self._call("sc_group_clear")

    def process_view(self, K, R, t, mask, mask_dtype):
        K, R, t = Engine._pose(K, R, t)
        if K.size != 4 or R.size != 9 or t.size != 3:
            raise ValueError("need intrinsics[4], rot[9], tvec[3]")
        if mask.ndim != 2:
            raise ValueError("mask must be 2-D (H, W)")
        mask = np.ascontiguousarray(mask)
        H, W = mask.shape
        self._call("sc_group_process_view", addr(K), addr(R), addr(t), addr(mask), H, W, int(mask_dtype), 0)

    def flush(self):
        self._call("sc_group_flush")

    def synchronize(self):
        self._call("sc_group_synchronize")

    def get_values(self, out=None):
        if out is None:
            out = np.empty(self.shape, dtype=self.dtype)
        if out.dtype != self.dtype or out.size != int(np.prod(self.shape)) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("output buffer has the wrong dtype/size/layout")
        self._call("sc_group_get_values", addr(out))
        return out

    def values_device_ptr(self):
        return None  # the grid lives on several devices

    def num_voxels(self):
        return int(np.prod(self.shape))
