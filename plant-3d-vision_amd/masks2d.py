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


def dilate_cross(mask, n):
    """``proc2d.dilation(img, n)`` (``plant3dvision/proc2d.py:172-220``) for bool tensors
    ``[..., H, W]``: ``binary_dilation`` with ``disk(n, decomposition='sequence')``.  For n = 1
    (every shipped config) that footprint is the 3x3 cross; larger radii use skimage's series
    table, which is not reproduced here."""
    import torch
    if n == 0:
        return mask
    if n != 1:
        raise NotImplementedError("only dilation 0 or 1 (3x3 cross) is mirrored exactly")
    out = mask.clone()
    out[..., 1:, :] |= mask[..., :-1, :]
    out[..., :-1, :] |= mask[..., 1:, :]
    out[..., :, 1:] |= mask[..., :, :-1]
    out[..., :, :-1] |= mask[..., :, 1:]
    return out


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


def voxels_from_masks(masks, cameras, shape, origin, voxel_size, type="averaging", log=True,
                      invert=False, device=None):
    """``Voxels`` on device-resident masks: one volume per label.

    masks   : ``{label: uint8 cuda tensor [n_img, Sy, Sx]}`` (``masks_from_predictions``)
    cameras : list of ``n_img`` camera dicts (the ``colmap_camera`` metadata schema, cl.py:293-296)
    Returns ``{label: ndarray}`` -- float32 for "averaging" (after ``exp`` / clip when ``log``,
    tasks/cl.py:172-174), int32 for "carving".
    """
    import torch
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
    eng = nat.Engine(shape, origin, voxel_size, mode, device=dev)
    # The engine keeps its own (non-blocking) stream and is ordered explicitly behind torch's:
    # torch's default stream has handle 0, which is NOT a stream the engine could adopt (0 means
    # "own stream" to sc_set_stream) and which a non-blocking stream does not synchronise with.
    producer = torch.cuda.current_stream(dev).cuda_stream
    if mode == nat.SC_MODE_AVERAGE:
        lut = img_as_float32(np.arange(256, dtype=np.uint8))
        if log:
            with np.errstate(divide="ignore"):
                lut = np.log(EPS + lut)
        eng.set_lut(lut)
    out = {}
    try:
        for q, (label, m) in enumerate(masks.items()):
            if m.dtype != torch.uint8 or tuple(m.shape) != (n_img, H, W) or not m.is_contiguous():
                raise ValueError("masks must be contiguous uint8 [n_img, Sy, Sx]")
            if q:
                eng.clear()  # cl.py:252-253
            # the label's host array: pages touched on host threads while the device works
            dest = nat.TouchedEmpty(tuple(int(s) for s in shape),
                                    np.float32 if mode == nat.SC_MODE_AVERAGE else np.int32)
            if mode == nat.SC_MODE_AVERAGE:
                src = (255 - m) if invert else m
                eng.order_after(producer)  # `m` / `src` are complete before the engine reads them
                eng.process_views_device(K, R, t, src.data_ptr(), n_img, H, W, nat.SC_MASK_U8_LUT)
            else:
                code = nat.SC_MASK_U8_INV if invert else nat.SC_MASK_U8
                src = m
                eng.order_after(producer)
                eng.process_views_device(K, R, t, src.data_ptr(), n_img, H, W, code)
            vol = eng.get_values(dest.result())  # flushes and waits: `src` may go now
            del src
            if mode == nat.SC_MODE_AVERAGE and log:
                from .tasks.cl import _exp_clip
                vol = _exp_clip(vol)  # np.exp, then vol[vol > 1] = 1 (tasks/cl.py:172-174)
            out[label] = vol
    finally:
        eng.close()
    return out


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
