"""Host-side mirror of ``plant3dvision.cl.Backprojection`` over the MI355X engine.

Same class name, constructor, attributes, methods and error behaviour as the reference
(``plant3dvision/cl.py:47-311``); the PyOpenCL context/queue/kernels are replaced by the
HIP engine behind ``include/spacecarve.h``.  Differences a caller can observe:

* ``process_view`` only *enqueues* the view (the mask is consumed before it returns);
  the carve itself runs when values are requested, when ``flush()`` is called, or every
  ``views_per_launch`` views.  The reference's per-view ``queue.finish()`` (cl.py:226) is
  ``synchronize()``.  Results are identical because nothing is observable in between.
* ``values_d`` & co. are not OpenCL buffers (``values_d`` is the device address, the
  small pose buffers do not exist).
* There is no CPU path: constructing an instance without the built library or without a
  gfx950 GPU raises.
"""
import collections
import logging
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native as nat

logger = logging.getLogger(__name__)

EPS = 1e-10  # cl.py:35


def img_as_float32(image):
    """What ``skimage.util.img_as_float32`` (scikit-image, unpinned in the reference's
    requirements.txt:19; call site cl.py:206) does for the mask dtypes that occur:
    unsigned ints are MULTIPLIED by float32(1/max) (skimage.util.dtype._convert), bool
    becomes {0, 1}, floats are cast.  Parity of this third-party step is unpinned."""
    image = np.asarray(image)
    if image.dtype == np.float32:
        return image
    if image.dtype == np.bool_:
        return image.astype(np.float32)
    if image.dtype.kind == "u":
        imax = np.iinfo(image.dtype).max
        comp = np.float32 if image.dtype.itemsize <= 2 else np.float64
        return np.multiply(image, 1.0 / imax, dtype=comp).astype(np.float32, copy=False)
    if image.dtype.kind == "i":
        info = np.iinfo(image.dtype)
        comp = np.float32 if image.dtype.itemsize <= 2 else np.float64
        out = np.add(image, 0.5, dtype=comp)
        out *= 2 / (float(info.max) - float(info.min))
        return out.astype(np.float32, copy=False)
    if image.dtype.kind == "f":
        return image.astype(np.float32)
    raise ValueError(f"cannot convert mask of dtype {image.dtype} to float32")


def read_image(fi):
    """``plantdb.io.read_image`` (cl.py:298; plantdb is an unvendored submodule) when it is
    importable; otherwise files that carry their pixels (``read_image()`` / ``array``).

    Files that hand out their bytes (plantdb's ``File.read_raw()``, which is what ``io.read_image``
    itself decodes) go through the engine's own PNG decoder when they are plain 8-bit greyscale --
    every mask a ``Masks`` / ``Segmentation2D`` fileset holds: same pixels (PNG is lossless), decoded
    with the interpreter lock released, so the decode-ahead threads scale.  Anything else takes the
    usual reader."""
    raw = None
    if hasattr(fi, "read_raw"):
        try:
            raw = fi.read_raw()
        except Exception:
            raw = None
    if isinstance(raw, (bytes, bytearray, memoryview)) and len(raw) > 33:
        arr = nat.png_decode_gray8(raw)
        if arr is not None:
            return arr
    try:
        from plantdb import io  # type: ignore
    except ImportError:
        io = None
    if io is not None and not hasattr(fi, "array") and not hasattr(fi, "read_image"):
        return io.read_image(fi)
    if hasattr(fi, "read_image"):
        return fi.read_image()
    if hasattr(fi, "array"):
        return fi.array
    raise TypeError(f"cannot read an image from {fi!r}: plantdb is not installed and the "
                    f"object has neither read_image() nor .array")


def auto_format_bytes(size_bytes, unit="octets"):
    """plant3dvision/utils.py, used for the log line at cl.py:157."""
    if unit.lower() in ("octets", "o"):
        base, names = 1024, ["o", "Ko", "Mo", "Go", "To"]
    else:
        base, names = 1024, ["B", "KB", "MB", "GB", "TB"]
    v = float(size_bytes)
    for nm in names:
        if v < base or nm == names[-1]:
            return f"{v:.1f} {nm}"
        v /= base


def _assign_slabs(dst, src, workers=None):
    """``dst[...] = src`` (same shape, casting as NumPy does) slab by slab on a few threads: the
    float64 stack of ``process_fileset`` (cl.py:249,254) is 1 GiB per label at 512^3, and the
    single-threaded copy-with-cast costs more than the whole carve."""
    if src.ndim == 0 or src.shape[0] < 2 or src.size < (1 << 22):
        dst[...] = src
        return
    workers = workers or min(8, nat.host_workers())
    n0 = src.shape[0]
    bounds = [n0 * q // workers for q in range(workers + 1)]

    def work(q):
        a, b = bounds[q], bounds[q + 1]
        if b > a:
            dst[a:b] = src[a:b]

    with ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(work, range(workers)))


def averaging_table(log):
    """What the host conversion of cl.py:205-208 makes of each of the 256 byte values: the
    conversions are elementwise, so ``table[mask] == convert(mask)`` bit for bit."""
    lut = img_as_float32(np.arange(256, dtype=np.uint8))  # cl.py:205-206
    if log:
        with np.errstate(divide="ignore", invalid="ignore"):
            lut = np.log(EPS + lut)  # cl.py:207-208
    return np.ascontiguousarray(lut, dtype=np.float32)


def submit_view(engine, dtype, log, lut, intrinsics, rot, tvec, mask, invert=False):
    """The host half of ``process_view`` (cl.py:205-215) for an engine of state type ``dtype``
    (int32 carve / float32 average), shared by ``Backprojection`` and ``ShardedBackprojection``.
    Returns the averaging table in use (``lut``, built and handed to the engine on first need).

    For 1-byte carving masks the fileset loop's ``np.invert`` (cl.py:300-301) is folded into the
    device-side bit packing (``SC_MASK_U8_INV`` / ``SC_MASK_BOOL_INV``); every other case inverts
    on the host exactly as the reference does."""
    mask = np.asarray(mask)
    fold = invert and dtype == np.int32 and mask.dtype in (np.uint8, np.bool_)
    if invert and not fold:
        mask = np.invert(mask)
    if dtype == np.float32 and mask.dtype == np.uint8:
        # averaging of a uint8 mask: ship the bytes and let the device look the float up in a
        # 256-entry table built by the SAME host operations on the 256 byte values
        if lut is None:
            lut = averaging_table(log)
            engine.set_lut(lut)
        engine.process_view(intrinsics, rot, tvec, np.ascontiguousarray(mask), nat.SC_MASK_U8_LUT)
        return lut
    if dtype == np.float32 and mask.dtype != np.float32:
        mask = img_as_float32(mask)  # cl.py:205-206
    if log and dtype == np.float32:
        with np.errstate(divide="ignore", invalid="ignore"):
            mask = np.log(EPS + mask)  # cl.py:207-208

    if dtype == np.int32:
        # cl.py:215 casts to int32 and the kernel tests == 0 (backprojection.c:79);
        # for 1-byte masks that test is done on the bytes themselves
        if mask.dtype == np.bool_:
            mask_h = np.ascontiguousarray(mask).view(np.uint8)
            code = nat.SC_MASK_BOOL_INV if fold else nat.SC_MASK_U8
        elif mask.dtype == np.uint8:
            mask_h = np.ascontiguousarray(mask)
            code = nat.SC_MASK_U8_INV if fold else nat.SC_MASK_U8
        else:
            mask_h, code = np.ascontiguousarray(mask, dtype=np.int32), nat.SC_MASK_I32
    else:
        mask_h, code = np.ascontiguousarray(mask, dtype=np.float32), nat.SC_MASK_F32
    engine.process_view(intrinsics, rot, tvec, mask_h, code)
    return lut


def _assign_and_return(dst, src):
    _assign_slabs(dst, src)
    return src


# The reference creates its OpenCL context and queue when the module is imported (cl.py:29-30).  Here the runtime comes
# up on a thread of the library's from this import on (nat.prewarm), beside whatever the importer does before it builds
# its first Backprojection; nothing blocks, and nothing happens on a box without a ROCm device (SC_PREWARM=0: never).
nat.prewarm()


class Backprojection(object):
    """Back-projection onto a voxel volume (drop-in for ``plant3dvision.cl.Backprojection``).

    Parameters are the reference's (cl.py:118); ``device`` (an ordinal, or a list of ordinals to
    shard the grid's x-planes over several GPUs from this one process) and ``views_per_launch`` are
    additions with neutral defaults.

    Attributes
    ----------
    shape, origin, voxel_size, default_value, log, labels : as given
    dtype : numpy.int32 ("carving") or numpy.float32 ("averaging")   (cl.py:145-150)
    kernel : str, "carve" or "average" -- the HIP kernel that will run
    values_h : numpy.ndarray, host copy of the volume (refreshed by ``get_values``)
    values_d : int, device address of the volume (``device_values()``; fetched on demand, dropped whenever the
        state changes, so a read is never a stale snapshot)
    """

    def __init__(self, shape, origin, voxel_size, type="carving", default_value=0, labels=None,
                 log=False, device=0, views_per_launch=0, decode_workers=None):
        self.shape = shape
        self.origin = origin
        self.voxel_size = voxel_size
        self.default_value = default_value
        self.log = log
        self.labels = labels
        if type == "carving":
            self.dtype = np.int32
            self.kernel = "carve"
            self._mode = nat.SC_MODE_CARVE
        elif type == "averaging":
            self.dtype = np.float32
            self.kernel = "average"
            self._mode = nat.SC_MODE_AVERAGE
        else:
            raise ValueError(f"Unknown kernel type {type}, valid values are 'averaging' or 'carving'!")

        buff_size = int(np.prod([int(s) for s in self.shape])) * np.dtype(self.dtype).itemsize
        logger.info(f"Buffer shape is {self.shape}")
        logger.info(f"Required memory for buffer is {auto_format_bytes(buff_size)}!")

        self.device = device
        self.views_per_launch = views_per_launch
        # image files are decoded ahead of the device on this many threads (1 = the reference's
        # strictly serial read -> process loop, cl.py:282-303); results do not depend on it
        self.decode_workers = (nat.host_workers(10) if decode_workers is None  # measured: 6 / 8 / 10 / 12 / 16 -> 24 / 20.4 / 18.4 / 19.5 / 20.9 ms
                               else max(1, int(decode_workers)))
        self._values_h = None
        self._spare = None
        self._prefault = None
        self._narrow_h = None
        self.intrinsics_d = None
        self.rot_d = None
        self.tvec_d = None
        self.volinfo_d = None
        self.shape_d = None
        self._engine = None
        self.init_buffers()

    # ---------------------------------------------------------------------------------
    def init_buffers(self):
        """cl.py:171-188.  The host array is built lazily in ``get_values``/``values_h``
        consumers; the device state starts as ``default_value`` everywhere."""
        if self._engine is not None:
            self._engine.close()
        if isinstance(self.device, (list, tuple)):
            # several GPUs from this process: x-planes dealt round-robin (sc_create_sharded)
            self._engine = nat.EngineGroup(self.shape, self.origin, self.voxel_size, self._mode, self.device,
                                           default_value=float(self.default_value))
        else:
            # The device half of the engine's set-up runs beside what the caller does next (SC_CREATE_DEFERRED): the
            # reference pays its context and queue at import (cl.py:29-30), a fresh process here pays 130-240 ms of
            # runtime set-up -- behind which the file loop's reads and decodes now hide (process_label).  "No device at
            # all" is still this constructor's error (the device node is looked at, not the runtime); a device that
            # is there and unusable (not a gfx950, out of memory) is the first call's.  SC_ASYNC_CREATE=0: all here.
            deferred = os.environ.get("SC_ASYNC_CREATE", "1") != "0"
            if deferred and not os.path.exists("/dev/kfd"):
                raise nat.SpaceCarveError("no ROCm device: /dev/kfd is missing (this engine has no CPU path)")
            self._engine = nat.Engine(self.shape, self.origin, self.voxel_size, self._mode,
                                      default_value=float(self.default_value), device=self.device, deferred=deferred)
        self._lut = None
        self._values_d = None
        if self.views_per_launch:
            self._engine.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, int(self.views_per_launch))
        self._values_h = None  # cl.py:173 builds default * ones here; see the values_h property
        self._spare = None
        self._start_prefault()
        self.volinfo_d = np.array([*self.origin, self.voxel_size], dtype=np.float32)  # cl.py:182
        self.shape_d = np.array(self.shape, dtype=np.int32)  # cl.py:186
        return

    def process_view(self, intrinsics, rot, tvec, mask):
        """Process a new view (cl.py:190-227).

        intrinsics: [f_x, f_y, c_x, c_y]; rot: rotation matrix (flattened row-major or 3x3);
        tvec: translation; mask: 2-D array (uint8/bool/int for carving, anything
        ``img_as_float32`` accepts for averaging).
        """
        self._submit_view(intrinsics, rot, tvec, mask, invert=False)
        return

    def _submit_view(self, intrinsics, rot, tvec, mask, invert):
        """``process_view`` plus the fileset loop's optional ``np.invert`` (cl.py:300-301)."""
        self._values_d = None  # the state is about to change: a cached device address may be a stale snapshot
        self._lut = submit_view(self._engine, self.dtype, self.log, self._lut, intrinsics, rot, tvec,
                                mask, invert)
        return

    def flush(self):
        """Launch every view enqueued so far (asynchronous)."""
        self._engine.flush()

    def synchronize(self):
        """Flush and wait: the reference's ``queue.finish()`` (cl.py:226)."""
        self._engine.synchronize()

    # -- host copy of the volume ---------------------------------------------------------------
    # The reference keeps ``values_h = default_value * ones(shape)`` from construction / clear()
    # on (cl.py:173,309) and copies the device buffer into it in get_values.  Building that array
    # costs 0.1-0.5 s of host time at 512^3, at construction and again at every clear(); but a
    # FRESH array is no better: its first-touch page faults make the 512 MiB read-back take 50 ms
    # instead of 10 (measured; page-locking does not beat already-touched pageable memory here,
    # tools/bench_host_masks.py).  So: the destination of the next read-back is allocated and
    # touched on a host thread while the device works (no HIP calls on that thread), the default-
    # valued contents are only written if somebody reads ``values_h`` before a read-back, and
    # clear() hands out a new buffer unless nobody else holds the old one (arrays returned by
    # get_values keep their contents across clear(), as in the reference).
    @property
    def values_h(self):
        if self._values_h is None:
            buf = self._take_buffer()
            buf[...] = self.default_value
            self._values_h = buf
        return self._values_h

    @values_h.setter
    def values_h(self, value):
        self._values_h = value

    def _narrow_ok(self):
        """Carve labels of a volume large enough to pay cross PCIe as int8 (``sc_get_values_i8``)."""
        return (self.dtype == np.int32 and int(np.prod([int(s) for s in self.shape])) >= (1 << 24)
                and -128 <= int(self.default_value) <= 127 and hasattr(self._engine, "get_values_i8"))

    def _wire2_ok(self):
        """Three-state labels: the 2-bit wire (``sc_get_values_wire2``), which needs no staging array of ours."""
        return (self._narrow_ok() and float(self.default_value) in (-1.0, 0.0, 1.0)
                and hasattr(self._engine, "get_values_wire2"))

    def _start_prefault(self):
        # the buffers the next read-back lands in, touched on host threads while the device works: the
        # volume-sized array, and the int8 staging buffer where labels travel as bytes
        shape = tuple(int(s) for s in self.shape)
        if self._narrow_ok() and not self._wire2_ok() and not isinstance(self._narrow_h, np.ndarray):
            self._narrow_h = nat.TouchedEmpty(shape, np.int8, threads=2)
        self._prefault = nat.TouchedEmpty(shape, self.dtype)

    def _take_buffer(self):
        """An array of the volume's shape (its pages already touched where that was prepared)."""
        shape = tuple(int(s) for s in self.shape)
        buf, self._spare = self._spare, None
        if buf is None and self._prefault is not None:
            buf = self._prefault.result()
            self._prefault = None
        if buf is None or buf.dtype != self.dtype or buf.shape != shape:
            buf = np.empty(shape, dtype=self.dtype)
        return buf

    def get_values(self):
        """Gets computed values from the device (cl.py:229-232); the returned array
        aliases ``values_h`` like the reference's.

        Carve labels of volumes of 2^24 voxels and more cross PCIe narrow, into a staging buffer whose pages
        were touched on host threads while the device worked, and are widened into the int32 array on host
        threads: at 2 bits each when ``default_value`` is -1, 0 or 1 (``sc_get_values_wire2``: the pieces are
        widened inside the library as they land), else as int8 (``sc_get_values_i8``: measured at 512^3 2.8 ms
        + 4.0 ms against 10.6 ms for an int32 copy into touched pages and 31 ms into a fresh array).  Smaller
        volumes, default values that do not fit a byte and averaging volumes are copied as they are."""
        if self._values_h is None:
            self._values_h = self._take_buffer()
        if self._wire2_ok():
            # three-state labels: 2 bits each over PCIe, widened inside the library as the pieces land
            self._engine.get_values_wire2(self._values_h.reshape(-1))
        elif self._narrow_ok():
            if isinstance(self._narrow_h, nat.TouchedEmpty):
                self._narrow_h = self._narrow_h.result()
            if self._narrow_h is None or self._narrow_h.size != self._values_h.size:
                self._narrow_h = np.empty(self._values_h.shape, dtype=np.int8)
            self._engine.get_values_i8(self._narrow_h)
            nat.widen_i8(self._values_h, self._narrow_h)
        else:
            self._engine.get_values(self._values_h)
        return self._values_h.reshape(self.shape)

    @property
    def values_d(self):
        """The reference's ``values_d`` (cl.py:175) is a device buffer, valid from ``init_buffers`` on; here: the
        device address of the volume.  It is fetched by ``device_values()`` when nothing is cached (which launches
        pending views and, on a padded grid, makes the snapshot) and the cache is dropped by everything that changes
        the state -- ``clear``, ``process_view`` & co., ``init_buffers`` -- so a read never yields a stale address
        (ADVICE r04)."""
        cached = getattr(self, "_values_d", None)
        if cached is None and getattr(self, "_engine", None) is not None:
            cached = self.device_values()
        return cached

    def device_values(self):
        """Device address of the volume as ``nx * ny * nz`` contiguous elements in the reference's order: pending
        views are launched; on a grid whose rows are padded on the device (nz not a multiple of 64) this call makes
        a snapshot without the padding (``sc_values_device_ptr``), valid until the state changes."""
        if getattr(self, "_engine", None) is None:
            return None
        self._values_d = self._engine.values_device_ptr()
        return self._values_d

    def process_fileset(self, fs, camera_metadata, invert=False):
        """Processes a whole fileset (cl.py:234-257): one volume, or with ``labels`` a
        float64 array ``[len(labels), *shape]``; the first label is not cleared."""
        if self.labels is not None:
            result = np.zeros((len(self.labels), *self.shape))
            if self._can_stage_labels():
                return self._process_labels_staged(fs, camera_metadata, invert, result)
            # result[i, :] = volume (cl.py:254) is a 1 GiB float64 write per label at 512^3 and costs more
            # than the label's whole carve: it runs on a helper thread while the next label is decoded,
            # carved and read back into another buffer
            pending = None
            with ThreadPoolExecutor(max_workers=1) as bg:
                for i, label in enumerate(self.labels):
                    logger.info(f"Processing label '{label}'...")
                    if i != 0:
                        self.clear()
                    vol = self.process_label(fs, camera_metadata, label, invert)
                    if pending is not None:
                        done = pending.result()
                        self.recycle(done)  # ours alone: a later label reads back into the same pages
                    pending = bg.submit(_assign_and_return, result[i], vol)
                    self._values_h = None  # `vol` belongs to the helper now
                    del vol
                if pending is not None:
                    pending.result()
            return result
        else:
            return self.process_label(fs, camera_metadata, None, invert=invert)

    #: set by ``tasks.cl.voxels_run`` (ours, not the reference's) before ``process_fileset`` when the task will apply
    #: ``np.exp`` + clip to the labelled result anyway (tasks/cl.py:172-174): the staged read-back below then applies
    #: them piece by piece on its way into the float64 array, and ``_label_post_applied`` tells the task so
    _label_post = None
    _label_post_applied = False
    _label_single_valued = None

    def _can_stage_labels(self):
        """One HIP engine (not a group of devices), a volume worth the pieces (``SC_LABELS_STAGED=0`` in the environment:
        never -- the two-pass route of rounds 3-4, for A/B measurements)."""
        return (isinstance(self._engine, nat.Engine) and hasattr(self._engine, "get_values_staged")
                and int(np.prod(self.shape)) >= (1 << 24) and os.environ.get("SC_LABELS_STAGED", "1") != "0")

    def _process_labels_staged(self, fs, camera_metadata, invert, result):
        """The label loop of cl.py:248-255 with ONE pass over host memory per label (round 5): the label's volume
        crosses PCIe in pieces into a page-locked ring, and host threads take each piece from there to its place in
        the float64 ``result[i]`` -- the widening of ``result[i, :] = volume`` (cl.py:254) and, when the task has
        announced them (``_label_post``), its ``np.exp`` + clip (tasks/cl.py:172-174: float64, on the widened values,
        as the reference computes them) -- while the next pieces cross and, behind the last piece, while the next
        label's masks are decoded and carved.  Same values as the two-pass route; the float32 host copy of a label
        (``values_h``) is not made."""
        post = self._label_post == "exp_clip"
        state = {"first": None, "single": True}

        def land(src, dst):
            # (the reference looks at the result BEFORE the exponential for its "one value only" warning, tasks/cl.py:168)
            if state["single"]:
                if state["first"] is None:
                    state["first"] = src[0]
                f = state["first"]
                same = bool(np.isnan(src[:4096]).all() and np.isnan(src).all()) if f != f else \
                    bool((src[:4096] == f).all() and (src == f).all())
                if not same:
                    state["single"] = False
            np.copyto(dst, src)  # float32 / int32 -> float64, exact
            if post:
                np.exp(dst, out=dst)
                np.minimum(dst, 1.0, out=dst)  # vol[vol > 1] = 1 (NaN stays NaN)

        futs = []
        with ThreadPoolExecutor(max_workers=nat.host_workers()) as pool:
            for i, label in enumerate(self.labels):
                logger.info(f"Processing label '{label}'...")
                if i != 0:
                    self.clear()
                self._submit_label(fs, camera_metadata, label, invert)
                futs += self._engine.get_values_staged(result[i], land, pool=pool)
                self._values_h = None
            for f in futs:
                f.result()
        self._label_post_applied = post
        self._label_single_valued = state["single"]
        return result

    def process_label(self, fs, camera_metadata, label=None, invert=False):
        """Processes a whole fileset for a given label (cl.py:259-305)."""
        self._submit_label(fs, camera_metadata, label, invert)
        return self.get_values()

    def _submit_label(self, fs, camera_metadata, label=None, invert=False):
        """The file loop of ``process_label`` (cl.py:279-303): every selected view handed to the engine, nothing read back."""
        if hasattr(fs, "get_files") and not isinstance(fs, (list, tuple)):
            fs = fs.get_files()  # cl.py:279-280

        selected = []
        for fi in fs:
            if label is not None and fi.get_metadata("channel") != label:  # cl.py:284
                continue
            logger.debug("processing file %s" % fi.id)
            cam = fi.get_metadata(camera_metadata, default=None)
            if cam is None:
                logger.warning(
                    f"Could not get camera params from '{camera_metadata}' for {fi.id}, skipping...")
                continue
            selected.append((fi, cam))

        def submit(cam, mask):
            intrinsics = np.array(cam["camera_model"]['params'][0:4], dtype=np.float32)  # :293
            rot = np.array(sum(cam['rotmat'], []), dtype=np.float32)  # :295
            tvec = np.array(cam['tvec'], dtype=np.float32)  # :296
            self._submit_view(intrinsics, rot, tvec, mask, invert)  # :300-303

        if self._submit_encoded(selected, invert):
            return
        if self.decode_workers <= 1 or len(selected) <= 1:
            for fi, cam in selected:
                submit(cam, read_image(fi))  # :298
        else:
            # decode ahead on a few threads, submit strictly in file order (the float sum of the
            # averaging kernel depends on it); the device packs/carves while the host decodes
            window = 2 * self.decode_workers
            with ThreadPoolExecutor(max_workers=self.decode_workers) as pool:
                pending = collections.deque()
                it = iter(selected)
                for fi, cam in it:
                    pending.append((cam, pool.submit(read_image, fi)))
                    if len(pending) >= window:
                        break
                while pending:
                    cam, fut = pending.popleft()
                    mask = fut.result()
                    nxt = next(it, None)
                    if nxt is not None:
                        pending.append((nxt[1], pool.submit(read_image, nxt[0])))
                    submit(cam, mask)

    def _submit_encoded(self, selected, invert):
        """The file loop (cl.py:282-303) handed to the library in one call when it can take it: a carve volume on one
        engine, every file handing out its bytes (``read_raw``: plantdb's ``File``) and holding an 8-bit greyscale PNG
        -- what a ``Masks`` / ``Segmentation2D`` fileset holds.  The files are decoded on the library's own threads and
        each mask is reduced to bits as it comes out of the decoder (``sc_process_png_views``); same pixels, same order,
        same ``np.invert``.  Returns False (and does nothing) otherwise: the decode-ahead loop below takes over."""
        if self.dtype != np.int32 or self.decode_workers <= 1 or len(selected) < 2:
            return False
        if not hasattr(self._engine, "process_png_views") or not all(hasattr(fi, "read_raw") for fi, _ in selected):
            return False
        try:
            raws = [fi.read_raw() for fi, _ in selected]
        except Exception:
            return False
        if not all(isinstance(r, (bytes, bytearray, memoryview)) and len(r) > 33 for r in raws):
            return False
        K = np.array([cam["camera_model"]['params'][0:4] for _, cam in selected], dtype=np.float32)  # :293
        R = np.array([sum(cam['rotmat'], []) for _, cam in selected], dtype=np.float32)  # :295
        t = np.array([cam['tvec'] for _, cam in selected], dtype=np.float32)  # :296
        self._values_d = None
        try:
            self._engine.process_png_views(K, R, t, raws, invert=invert, threads=max(self.decode_workers, 16))
        except ValueError:  # some file is not a grey8 PNG: nothing was enqueued
            return False
        return True

    def clear(self):
        """Clear computed values (cl.py:307-311)."""
        # values_h becomes a fresh default-valued array (cl.py:309), built when read.  The old one
        # may be held by the caller (get_values returned it) and keeps its contents: it is never
        # reused here unless the caller hands it back with ``recycle``.
        self._values_h = None
        self._values_d = None
        if self._spare is None and self._prefault is None:
            self._start_prefault()
        self._engine.clear()
        return

    def recycle(self, array):
        """Hand an array ``get_values`` returned back as the destination of a later read-back
        (its pages are touched already: a 512 MiB read-back into it takes 10 ms instead of 50).
        The caller promises not to use it afterwards.  No reference counterpart."""
        base = array
        while getattr(base, "base", None) is not None and isinstance(base.base, np.ndarray):
            base = base.base
        shape = tuple(int(s) for s in self.shape)
        if (isinstance(base, np.ndarray) and base.dtype == self.dtype and base.size == int(np.prod(shape))
                and base.flags["C_CONTIGUOUS"] and base.flags["WRITEABLE"]):
            if self._values_h is base:
                self._values_h = None
            self._spare = base.reshape(shape)

    def close(self):
        """Release device memory now (otherwise at garbage collection)."""
        if self._engine is not None:
            self._engine.close()
            self._engine = None
            try:  # the consumer's cached device buffers go with the volume they were sized for
                nat.backend().call("sc_vol2pcd_release")
            except Exception:
                pass
