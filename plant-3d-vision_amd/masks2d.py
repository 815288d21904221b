"""The ML feeder of the carve: segmentation predictions -> per-label masks -> volumes, on device.

Mirror of what ``Segmentation2D.run`` does to the network output before ``Voxels`` reads it
(``plant3dvision/tasks/proc2d.py:351-391``), but the masks never leave the GPU: the prediction
tensor ``[n_img, n_labels, Sy, Sx]`` (``:351,365``) is post-processed with torch ops and handed to
the engine by device pointer (``sc_process_views_device``) on torch's own HIP stream -- no PNG
round trip, no host copy (SURVEY 8f row 3 / BASELINE cfg 5).

PyTorch is plumbing here (the network and a few elementwise ops).  ``romiseg`` and its trained
weights are not vendored in the reference (empty submodule, weights downloaded by
``get_model.sh``), so ``StandInSegmenter`` -- a small seeded conv net with the same I/O contract --
stands in for it in tests and benchmarks; swap in the real model where it exists.
"""
import numpy as np

from . import _native as nat


#: 3x3 footprints of skimage's series decomposition of a disk (skimage/morphology/footprints.py,
#: ``_nsphere_series_decomposition`` after Park & Chin 1995): the four T-shaped elements, the
#: "diamond" (3x3 cross) and the square, as lists of (dy, dx) offsets of their set pixels.
_T0 = ((-1, -1), (-1, 0), (-1, 1), (0, 0), (1, 0))
_FOOTPRINTS = {
    "t0": _T0,
    "t90": tuple((-dx, dy) for dy, dx in _T0),    # np.rot90(t0, 1)
    "t180": tuple((-dy, -dx) for dy, dx in _T0),  # np.rot90(t0, 2)
    "t270": tuple((dx, -dy) for dy, dx in _T0),   # np.rot90(t0, 3)
    "diamond": ((-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)),
    "square": tuple((dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)),
}


def _compose(counts):
    """The footprint a series (n_t_series, n_diamond, n_square) amounts to, as a boolean array."""
    import numpy as np
    a, b, c = counts
    r = 4 * a + b + c
    img = np.zeros((2 * r + 1, 2 * r + 1), dtype=bool)
    img[r, r] = True
    for name, reps in (("t0", a), ("t90", a), ("t180", a), ("t270", a), ("diamond", b), ("square", c)):
        for _ in range(reps):
            out = np.zeros_like(img)
            for dy, dx in _FOOTPRINTS[name]:
                out[max(dy, 0):img.shape[0] + min(dy, 0), max(dx, 0):img.shape[1] + min(dx, 0)] |= \
                    img[max(-dy, 0):img.shape[0] + min(-dy, 0), max(-dx, 0):img.shape[1] + min(-dx, 0)]
            img = out
    return img


def disk_series(n):
    """``skimage.morphology.disk(n, decomposition='sequence')`` as (footprint name, repetitions)
    pairs: for n = 1 the 3x3 cross; for n > 1 a series of T-shaped, diamond and square 3x3 elements
    whose composition is closest to ``disk(n, strict_radius=False)``.

    scikit-image (unpinned in the reference's requirements.txt:17) ships the series counts as a
    precomputed table (``disk_decompositions.npy``) that is not available in this image; the counts
    are regenerated here by the procedure that table was made with (exhaustive search over
    (n_t_series, n_diamond, n_square) with 4 a + b + c = n, least sum of absolute differences, first
    minimum in that order).  PARITY UNPINNED for n > 1: identical footprints by construction of the
    same search, not by comparison with the table.  Every shipped ML configuration uses n = 1."""
    import numpy as np
    n = int(n)
    if n < 1:
        return ()
    if n == 1:
        return (("diamond", 1),)
    if n > 32:
        raise ValueError("dilation radius beyond 32 is not supported")
    yy, xx = np.mgrid[-n:n + 1, -n:n + 1]
    desired = (xx * xx + yy * yy) <= (n + 0.5) ** 2  # disk(n, strict_radius=False)
    best, best_err = None, None
    for a in range(n // 4 + 1):
        for b in range(n - 4 * a + 1):
            c = n - 4 * a - b
            err = int(np.sum(desired != _compose((a, b, c))))
            if best_err is None or err < best_err:
                best, best_err = (a, b, c), err
    a, b, c = best
    seq = []
    if a:
        seq += [("t0", a), ("t90", a), ("t180", a), ("t270", a)]
    if b:
        seq.append(("diamond", b))
    if c:
        seq.append(("square", c))
    return tuple(seq)


def dilate3x3(mask, offsets):
    """``scipy.ndimage.binary_dilation(mask, structure)`` for a 3x3 structure given by the (dy, dx)
    offsets of its set pixels, on bool tensors ``[..., H, W]``: out[p] = OR_o mask[p - o]; pixels
    beyond the border count as background (nothing wraps)."""
    import torch
    H, W = mask.shape[-2], mask.shape[-1]
    out = torch.zeros_like(mask)
    for dy, dx in offsets:
        out[..., max(dy, 0):H + min(dy, 0), max(dx, 0):W + min(dx, 0)] |= \
            mask[..., max(-dy, 0):H + min(-dy, 0), max(-dx, 0):W + min(-dx, 0)]
    return out


def dilate_cross(mask, n):
    """``proc2d.dilation(img, n)`` (``plant3dvision/proc2d.py:172-220``) for bool tensors
    ``[..., H, W]``: ``binary_dilation`` with ``disk(n, decomposition='sequence')``, i.e. one 3x3
    footprint after the other, each repeated as the series says (``disk_series``).  n = 1 -- every
    shipped ML configuration -- is the 3x3 cross."""
    for name, reps in disk_series(n):
        for _ in range(reps):
            mask = dilate3x3(mask, _FOOTPRINTS[name])
    return mask


def masks_from_predictions(pred, label_names, labels=None, inverted_labels=("background",),
                           binarize=True, threshold=0.01, dilation=1):
    """Per-label uint8 masks from a prediction tensor, exactly the per-image chain of
    ``tasks/proc2d.py:365-380``: optional ``1 - im``, ``im > threshold``, dilation,
    ``(im * 255).astype(uint8)``, optional ``255 - im``.

    pred : torch.Tensor ``[n_img, n_labels, Sy, Sx]`` float32 (any device).
    Returns ``{label: uint8 tensor [n_img, Sy, Sx]}`` on the same device, in label order.
    """
    import torch
    if pred.dim() != 4 or pred.shape[1] != len(label_names):
        raise ValueError("pred must be [n_img, n_labels, Sy, Sx] with one channel per label name")
    use = list(labels) if labels else list(label_names)  # proc2d.py:337-343
    out = {}
    for name in use:
        im = pred[:, label_names.index(name)].to(torch.float32)
        inv = name in inverted_labels
        if inv:
            im = 1.0 - im  # :367-368
        if binarize:
            b = im > threshold  # :371
            if dilation > 0:
                b = dilate_cross(b, dilation)  # :373-374
            m = b.to(torch.uint8) * 255  # :376
        else:
            # :376 `(im * 255).astype(np.uint8)`: truncation toward zero, and outside [0, 255]
            # what x86 NumPy does (float -> int32 -> low byte; torch's own float -> uint8 cast is
            # undefined there and differs between devices)
            m = (im * 255.0).to(torch.int32).bitwise_and(255).to(torch.uint8)
        if inv:
            m = 255 - m  # :378-379
        out[name] = m.contiguous()
    return out


def mask_files_layout(label_names, images_metadata, id_im, labels=None):
    """The files ``Segmentation2D.run`` writes and their metadata (``tasks/proc2d.py:362-393``), without
    writing them: for every image, for every label of the filtered list, the file id
    ``'%03d_%s' % (img_id, label)`` (:366) with metadata ``{'image_id': id_im[img_id][0], **original,
    'channel': label}`` (:383-390); and the fileset metadata ``{'label_names': [...]}`` (:392-393).
    What ``Voxels`` later reads from a mask file -- ``channel`` (cl.py:284) and the camera entry copied
    from the image's own metadata (cl.py:286-296) -- comes from here.

    Returns ``(files, fileset_metadata)`` with ``files = [(file_id, img_id, label, metadata), ...]``."""
    if labels:
        label_range = [label_names.index(x) for x in labels]  # :342-343 (ValueError for an unknown label)
    else:
        label_range = range(len(label_names))
    files = []
    for img_id in range(len(images_metadata)):
        for label_id in label_range:
            name = label_names[label_id]
            md = {"image_id": id_im[img_id][0], **images_metadata[img_id]}
            md["channel"] = name
            files.append(("%03d_%s" % (img_id, name), img_id, name, md))
    return files, {"label_names": [label_names[j] for j in label_range]}


def cameras_for_label(files, label, camera_metadata="colmap_camera"):
    """The camera dictionaries ``Backprojection.process_label`` would use for ``label`` (cl.py:282-296):
    files of that channel, in order, skipping those without the camera entry."""
    cams, img_ids = [], []
    for _, img_id, name, md in files:
        if name != label:
            continue
        cam = md.get(camera_metadata)
        if cam is None:
            continue
        cams.append(cam)
        img_ids.append(img_id)
    return cams, img_ids


def voxels_from_masks(masks, cameras, shape, origin, voxel_size, type="averaging", log=True,
                      invert=False, device=None, overlap=True, timing=None):
    """``Voxels`` on device-resident masks: one volume per label.

    masks   : ``{label: uint8 cuda tensor [n_img, Sy, Sx]}`` (``masks_from_predictions``)
    cameras : list of ``n_img`` camera dicts (the ``colmap_camera`` metadata schema, cl.py:293-296)
    overlap : with several labels, two engines (two device volumes) take the labels in turn: while one
        label's volume crosses PCIe and gets its ``exp`` / clip on a helper thread, the device already
        works on the next label.  ``False`` = one engine, one label after the other (cl.py:248-255).
        Same volumes either way.
    timing  : a dict to receive the phases' host-clock milliseconds (shared-launch path; diagnostics).
    Returns ``{label: ndarray}`` -- float32 for "averaging" (after ``exp`` / clip when ``log``,
    tasks/cl.py:172-174), int32 for "carving".
    """
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from .cl import EPS, img_as_float32
    first = next(iter(masks.values()))
    if not first.is_cuda:
        raise ValueError("masks must live on the GPU (there is no CPU path)")
    dev = first.device.index if device is None else int(device)
    n_img, H, W = first.shape
    if len(cameras) != n_img:
        raise ValueError("one camera per image")
    K = np.array([c["camera_model"]["params"][0:4] for c in cameras], dtype=np.float32)
    R = np.array([sum(c["rotmat"], []) for c in cameras], dtype=np.float32)
    t = np.array([c["tvec"] for c in cameras], dtype=np.float32)
    mode = nat.SC_MODE_AVERAGE if type == "averaging" else nat.SC_MODE_CARVE
    if type not in ("averaging", "carving"):
        raise ValueError(f"Unknown kernel type {type}, valid values are 'averaging' or 'carving'!")
    for m in masks.values():
        if m.dtype != torch.uint8 or tuple(m.shape) != (n_img, H, W) or not m.is_contiguous():
            raise ValueError("masks must be contiguous uint8 [n_img, Sy, Sx]")
    lut = None
    if mode == nat.SC_MODE_AVERAGE:
        lut = img_as_float32(np.arange(256, dtype=np.uint8))
        if log:
            with np.errstate(divide="ignore"):
                lut = np.log(EPS + lut)
    # The engines keep their own (non-blocking) streams and are ordered explicitly behind torch's:
    # torch's default stream has handle 0, which is NOT a stream the engine could adopt (0 means
    # "own stream" to sc_set_stream) and which a non-blocking stream does not synchronise with.
    producer = torch.cuda.current_stream(dev).cuda_stream
    vol_dtype = np.float32 if mode == nat.SC_MODE_AVERAGE else np.int32
    vol_shape = tuple(int(s) for s in shape)
    engines = []

    narrow = mode == nat.SC_MODE_CARVE and int(np.prod(vol_shape)) >= (1 << 24)

    def exp_clip_piece(piece):
        # np.exp, then vol[vol > 1] = 1 (tasks/cl.py:172-174) as np.minimum: same values, elementwise -- a piece at a time
        np.exp(piece, out=piece)
        np.minimum(piece, piece.dtype.type(1.0), out=piece)

    def exp_clip_from(src_piece, dst_piece):
        # the same from a piece that has landed in the page-locked ring to its place in the volume: one pass
        np.exp(src_piece, out=dst_piece)
        np.minimum(dst_piece, dst_piece.dtype.type(1.0), out=dst_piece)

    def finish(eng, dest, dest8, src):
        if dest8 is None and mode == nat.SC_MODE_AVERAGE and log and int(np.prod(vol_shape)) >= (1 << 24):
            # round 5: the exp / clip of a piece runs on host threads while the next pieces cross PCIe
            vol = eng.get_values_staged(dest.result(), exp_clip_from)
            del src
            return vol
        if dest8 is not None:
            # labels cross PCIe as bytes and are widened on host threads (as Backprojection.get_values)
            small = eng.get_values_i8(dest8.result())
            vol = np.empty(vol_shape, vol_dtype)  # its pages are faulted in by the widening threads
            nat.widen_i8(vol, small)
        else:
            vol = eng.get_values(dest.result())  # flushes and waits: `src` may go now
        del src
        if mode == nat.SC_MODE_AVERAGE and log:
            from .tasks.cl import _exp_clip
            # np.exp, then vol[vol > 1] = 1 (tasks/cl.py:172-174), over the array just read back
            vol = _exp_clip(vol, inplace=True)
        return vol

    out = {}
    if mode == nat.SC_MODE_AVERAGE and 1 < len(masks) <= 4 and overlap:
        # The labels share their cameras: ONE launch projects every voxel once per view and reads all the
        # labels' masks at that pixel (``sc_average_labels``; each label's sum is the same additions in the same
        # order as its own launch would make), one engine -- one device volume -- per label.  The volumes then
        # cross PCIe one after the other, the ``exp`` / clip of one on a helper thread beside the next copy.
        # (Up to 4 labels: with 6 the shared launch lost to six of their own, 16.5 against 14.3 ms on the device.)
        try:
            import threading
            import time
            tm = [("start", time.perf_counter())]
            srcs = []
            nlab = len(masks)
            dests = [None] * nlab
            ready = [threading.Event() for _ in range(nlab)]
            touch_error = []

            def touch_in_turn():
                # the labels' host arrays, pages touched in the order the copies need them (all at once they fight
                # over the same page-fault path: the first label's pages are what the first copy waits for)
                # (an allocation that fails here must reach the waiting thread, not leave it waiting)
                try:
                    for i in range(nlab):
                        dests[i] = nat.TouchedEmpty(vol_shape, vol_dtype, threads=nat.host_workers()).result()
                        ready[i].set()
                except BaseException as ex:  # noqa: BLE001
                    touch_error.append(ex)
                finally:
                    for ev in ready:
                        ev.set()

            toucher = threading.Thread(target=touch_in_turn, daemon=True)
            toucher.start()
            for label, m in masks.items():
                eng = nat.Engine(shape, origin, voxel_size, mode, device=dev)
                engines.append(eng)
                eng.set_lut(lut)
                eng.order_after(producer)  # the masks are complete before the engines read them
                srcs.append((255 - m) if invert else m)
            tm.append(("engines", time.perf_counter()))
            nat.average_labels(engines, K, R, t, [x.data_ptr() for x in srcs], n_img, H, W)
            tm.append(("enqueue", time.perf_counter()))
            # ONE pipeline over all the labels (round 5): the volumes cross the one PCIe link back to back, in pieces,
            # and every piece's exp / clip (tasks/cl.py:172-174) runs on host threads beside the copies that follow
            pipelined = log and int(np.prod(vol_shape)) >= (1 << 24)
            with ThreadPoolExecutor(max_workers=nat.host_workers()) as pool:
                futs = []
                for i, (label, eng) in enumerate(zip(masks, engines)):
                    ready[i].wait()
                    if touch_error:
                        raise touch_error[0]
                    tm.append((f"pages{i}", time.perf_counter()))
                    if pipelined:
                        # (the ring is shared by the labels: a slot is written again only when its piece has left it)
                        futs += eng.get_values_staged(dests[i], exp_clip_from, pool=pool)
                        out[label] = dests[i]
                    else:
                        out[label] = finish(eng, _Ready(dests[i]), None, None)
                    tm.append((f"copied{i}", time.perf_counter()))
                for f in futs:
                    f.result()
                tm.append(("exp_tail", time.perf_counter()))
            toucher.join()
            # the device volumes go back to the driver on a thread of their own: three hipFree of 512 MiB are
            # milliseconds the caller need not wait for (joined at interpreter exit; a failure there cannot be reported
            # to anybody, and there is nothing the caller could do about one)
            closing, engines = engines, []
            _close_later(closing)
            tm.append(("handed_back", time.perf_counter()))
            if timing is not None:
                timing.update({name: (b - a) * 1e3 for (_, a), (name, b) in zip(tm[:-1], tm[1:])})
            del srcs
        finally:
            for eng in engines:
                eng.close()
        return out
    try:
        for _ in range(2 if overlap and len(masks) > 1 else 1):
            eng = nat.Engine(shape, origin, voxel_size, mode, device=dev)
            engines.append(eng)
            if lut is not None:
                eng.set_lut(lut)
        pending = [None] * len(engines)  # per engine: (label, future of its volume)
        with ThreadPoolExecutor(max_workers=1) as helper:
            for q, (label, m) in enumerate(masks.items()):
                k = q % len(engines)
                eng = engines[k]
                if pending[k] is not None:
                    out[pending[k][0]] = pending[k][1].result()
                if q >= len(engines):
                    eng.clear()  # cl.py:252-253
                # the label's host array: pages touched on host threads while the device works
                dest = None if narrow else nat.TouchedEmpty(vol_shape, vol_dtype)
                dest8 = nat.TouchedEmpty(vol_shape, np.int8) if narrow else None
                if mode == nat.SC_MODE_AVERAGE:
                    src = (255 - m) if invert else m
                    code = nat.SC_MASK_U8_LUT
                else:
                    src = m
                    code = nat.SC_MASK_U8_INV if invert else nat.SC_MASK_U8
                eng.order_after(producer)  # `m` / `src` are complete before the engine reads them
                eng.process_views_device(K, R, t, src.data_ptr(), n_img, H, W, code)
                if len(engines) > 1:
                    eng.flush()  # the label's kernels are queued now, behind nothing of the other engine
                    pending[k] = (label, helper.submit(finish, eng, dest, dest8, src))
                else:
                    out[label] = finish(eng, dest, dest8, src)
                del src
            for entry in pending:
                if entry is not None:
                    out[entry[0]] = entry[1].result()
        out = {label: out[label] for label in masks}  # in label order
    finally:
        for eng in engines:
            eng.close()
    return out


_closers = []


def _close_later(engines):
    """``close()`` of engines whose results have been read back, off the caller's path."""
    import atexit
    import threading

    def work():
        for eng in engines:
            try:
                eng.close()
            except Exception:  # noqa: BLE001
                pass

    th = threading.Thread(target=work, daemon=True)
    if not _closers:
        atexit.register(lambda: [t.join() for t in list(_closers)])
    _closers[:] = [t for t in _closers if t.is_alive()]
    _closers.append(th)
    th.start()


class _Ready:
    """An array that is there already, with ``TouchedEmpty``'s ``result()``."""

    def __init__(self, arr):
        self._arr = arr

    def result(self):
        return self._arr


class StandInSegmenter:
    """Seeded stand-in for ``romiseg.Segmentation2D.segmentation`` (unvendored): a small conv
    net mapping RGB images ``[n, 3, Sy, Sx]`` to per-label probabilities ``[n, L, Sy, Sx]``
    (softmax over labels).  Random weights: it produces structured masks, not plant parts."""

    def __init__(self, label_names, seed=0, device="cuda"):
        import torch
        self.label_names = list(label_names)
        g = torch.Generator().manual_seed(seed)
        L = len(self.label_names)
        self.w1 = (torch.randn(16, 3, 5, 5, generator=g) * 0.2).to(device)
        self.w2 = (torch.randn(16, 16, 5, 5, generator=g) * 0.1).to(device)
        self.w3 = (torch.randn(L, 16, 1, 1, generator=g) * 0.5).to(device)

    def __call__(self, images):
        import torch
        import torch.nn.functional as F
        with torch.no_grad():
            x = F.relu(F.conv2d(images, self.w1, padding=2))
            x = F.relu(F.conv2d(x, self.w2, padding=2))
            return torch.softmax(F.conv2d(x, self.w3) * 4.0, dim=1)
