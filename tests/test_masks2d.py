"""The ML feeder: prediction tensor -> per-label masks -> volumes.
CPU: the mask chain against a NumPy/SciPy restatement of tasks/proc2d.py:365-380.
GPU: device-resident masks through the engine against the oracle."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import oracle_c
from plant3dvision_amd import masks2d, scenes
from plant3dvision_amd.cl import EPS, img_as_float32

LABELS = ["background", "flower", "stem"]
CROSS = ndimage.generate_binary_structure(2, 1)  # == skimage disk(1): the 3x3 cross


def _reference_chain(im, inverted, binarize, threshold, dilation):
    """tasks/proc2d.py:365-380 for one image and label, in NumPy."""
    if inverted:
        im = 1.0 - im
    if binarize:
        im = im > threshold
        if dilation > 0:
            im = ndimage.binary_dilation(im, structure=CROSS, iterations=1)
    im = (im * 255).astype(np.uint8)
    if inverted:
        im = 255 - im
    return im


@pytest.mark.parametrize("binarize,dilation", [(True, 1), (True, 0), (False, 0)])
def test_masks_from_predictions_matches_reference_chain(binarize, dilation):
    rng = np.random.default_rng(0)
    pred = rng.random((3, len(LABELS), 40, 56), dtype=np.float32)
    pred[:, :, :3, :] = 0.0  # structure at the border (dilation must not wrap)
    out = masks2d.masks_from_predictions(torch.from_numpy(pred), LABELS, binarize=binarize,
                                         threshold=0.6, dilation=dilation)
    assert list(out) == LABELS
    for li, name in enumerate(LABELS):
        for i in range(pred.shape[0]):
            want = _reference_chain(pred[i, li], name == "background", binarize, 0.6, dilation)
            assert np.array_equal(out[name][i].numpy(), want), (name, i)
    sub = masks2d.masks_from_predictions(torch.from_numpy(pred), LABELS, labels=["stem"])
    assert list(sub) == ["stem"]


def test_unsupported_dilation_radius_is_loud():
    with pytest.raises(NotImplementedError):
        masks2d.dilate_cross(torch.zeros(4, 4, dtype=torch.bool), 2)


@pytest.mark.gpu
@pytest.mark.parametrize("type_,log", [("averaging", True), ("averaging", False), ("carving", False)])
def test_device_resident_masks_to_volumes(gpu_device, type_, log):
    """cfg 5 in miniature: stand-in network on the GPU -> masks on the GPU -> volumes; the
    same masks (downloaded) through the oracle must give the same volumes bit for bit."""
    shape, origin, vs, views = scenes.make_scene(28, 6, "plant", width=160, height=128, fx=130.0, fy=130.0,
                                                 cx=80.0, cy=64.0)
    cams = [scenes.camera_dict(K, R, t) for K, R, t, _ in views]
    torch.manual_seed(0)
    coarse = torch.rand(len(views), 3, 8, 10, device="cuda")  # low-frequency "images"
    images = torch.nn.functional.interpolate(coarse, size=(128, 160), mode="bilinear", align_corners=False)
    net = masks2d.StandInSegmenter(LABELS, seed=1)
    pred = net(images)
    assert pred.shape == (len(views), len(LABELS), 128, 160)
    thr = float(pred[:, 1].median())  # about half of the "flower" pixels pass
    masks = masks2d.masks_from_predictions(pred, LABELS, threshold=thr, dilation=1)
    vols = masks2d.voxels_from_masks(masks, cams, shape, origin, vs, type=type_, log=log)
    assert list(vols) == LABELS
    fill = {name: float((masks[name] != 0).float().mean()) for name in LABELS}
    assert any(0.02 < f < 0.98 for f in fill.values()), fill  # the stand-in produces structure
    for name in LABELS:
        host = masks[name].cpu().numpy()
        if type_ == "carving":
            want = oracle_c.carve(shape, origin, vs, [(K, R, t, host[q]) for q, (K, R, t, _) in enumerate(views)])
        else:
            conv = (lambda m: np.log(EPS + img_as_float32(m))) if log else img_as_float32
            with np.errstate(divide="ignore"):
                want = oracle_c.average(shape, origin, vs, [(K, R, t, conv(host[q])) for q, (K, R, t, _) in enumerate(views)])
            if log:
                want = np.exp(want)
                want[want > 1] = 1.0
        assert np.array_equal(vols[name], want), name
