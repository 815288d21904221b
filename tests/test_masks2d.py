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


def test_mask_files_layout_carries_the_image_metadata_and_the_label_names():
    """tasks/proc2d.py:362-393: file ids, per-file metadata (image_id, the image's own metadata, the
    channel) and the fileset's label_names, in the reference's loop order."""
    names = ["background", "flower", "stem"]
    imd = [{"colmap_camera": {"rotmat": q}, "shot_id": f"{q:03d}"} for q in range(2)]
    id_im = [("00000_rgb", 0), ("00001_rgb", 1)]
    files, fsmd = masks2d.mask_files_layout(names, imd, id_im)
    assert [f[0] for f in files] == ["000_background", "000_flower", "000_stem", "001_background", "001_flower", "001_stem"]
    assert files[4][3] == {"image_id": "00001_rgb", "colmap_camera": {"rotmat": 1}, "shot_id": "001", "channel": "flower"}
    assert fsmd == {"label_names": names}
    files, fsmd = masks2d.mask_files_layout(names, imd, id_im, labels=["stem", "flower"])  # the order of `labels`
    assert [f[0] for f in files] == ["000_stem", "000_flower", "001_stem", "001_flower"]
    assert fsmd == {"label_names": ["stem", "flower"]}
    cams, ids = masks2d.cameras_for_label(files, "flower")
    assert cams == [{"rotmat": 0}, {"rotmat": 1}] and ids == [0, 1]
    imd[1].pop("colmap_camera")  # a file without camera metadata is skipped (cl.py:288-291)
    files, _ = masks2d.mask_files_layout(names, imd, id_im)
    assert masks2d.cameras_for_label(files, "stem")[1] == [0]
    with pytest.raises(ValueError):
        masks2d.mask_files_layout(names, imd, id_im, labels=["leaf"])


T0 = np.array([[1, 1, 1], [0, 1, 0], [0, 1, 0]], dtype=bool)  # skimage _t_shaped_element_series, 2-D
STRUCTURES = {"t0": T0, "t90": np.rot90(T0, 1), "t180": np.rot90(T0, 2), "t270": np.rot90(T0, 3),
              "diamond": CROSS, "square": np.ones((3, 3), dtype=bool)}


def test_3x3_footprint_dilation_is_scipy_binary_dilation():
    """Every element of the series, orientation of the T shapes included, against SciPy -- which is
    what skimage.morphology.binary_dilation calls per footprint of a sequence."""
    rng = np.random.default_rng(1)
    img = rng.random((37, 53)) < 0.08
    img[0, :5] = True; img[-1, -3:] = True; img[:4, 0] = True  # structure at the borders
    for name, st in STRUCTURES.items():
        offs = masks2d._FOOTPRINTS[name]
        assert sorted(offs) == sorted((int(y) - 1, int(x) - 1) for y, x in zip(*np.nonzero(st))), name
        got = masks2d.dilate3x3(torch.from_numpy(img), offs).numpy()
        assert np.array_equal(got, ndimage.binary_dilation(img, structure=st)), name


@pytest.mark.parametrize("n", [2, 3, 5, 6, 8])
def test_dilation_by_a_disk_series(n):
    """proc2d.dilation(img, n > 1): the footprints of the series one after the other, each as often as
    the series says (what binary_dilation does with a footprint sequence); the series itself is the
    one closest to disk(n, strict_radius=False) among 4a + b + c = n."""
    series = masks2d.disk_series(n)
    rng = np.random.default_rng(n)
    img = rng.random((64, 80)) < 0.01
    want = img
    for name, reps in series:
        for _ in range(reps):
            want = ndimage.binary_dilation(want, structure=STRUCTURES[name])
    got = masks2d.dilate_cross(torch.from_numpy(img), n).numpy()
    assert np.array_equal(got, want)
    # a single pixel grows into the composed footprint: radius n along the axes, a disk within 1 px
    one = np.zeros((2 * n + 5, 2 * n + 5), dtype=bool)
    one[n + 2, n + 2] = True
    fp = masks2d.dilate_cross(torch.from_numpy(one), n).numpy()
    yy, xx = np.mgrid[-n - 2:n + 3, -n - 2:n + 3]
    disk = xx * xx + yy * yy <= (n + 0.5) ** 2
    assert fp[n + 2, 2] and fp[2, n + 2] and not fp[n + 2, 1] and not fp[1, n + 2]
    assert (fp != disk).sum() <= 0.12 * disk.sum()
    assert sum(r * (4 if k == "t0" else 0 if k.startswith("t") else 1) for k, r in series) == n


def test_out_of_range_predictions_follow_numpy_astype_uint8():
    """`(im * 255).astype(np.uint8)` (tasks/proc2d.py:376) outside [0, 1]: x86 NumPy wraps through
    int32; the device chain does the same instead of torch's undefined float -> uint8 cast."""
    pred = torch.tensor([[[[-0.5, 0.0, 0.25, 1.0, 1.5, 2.0, -1.0, 0.999]]]], dtype=torch.float32)
    out = masks2d.masks_from_predictions(pred, ["stem"], binarize=False)["stem"].numpy().ravel()
    with np.errstate(invalid="ignore"):
        want = (pred.numpy().ravel() * 255).astype(np.int32).astype(np.uint8)  # wrap, as the reference's CPU does
    assert np.array_equal(out, want)


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
    # one engine, label after label (the reference's sequence) == two engines taking turns
    serial = masks2d.voxels_from_masks(masks, cams, shape, origin, vs, type=type_, log=log, overlap=False)
    assert list(serial) == LABELS
    for name in LABELS:
        assert np.array_equal(serial[name].view(np.uint32), vols[name].view(np.uint32)), name
    five = {f"{name}{k}": masks[name] for k in range(2) for name in LABELS[:3]}  # more labels than engines
    got = masks2d.voxels_from_masks(five, cams, shape, origin, vs, type=type_, log=log)
    assert list(got) == list(five)
    for key in five:
        assert np.array_equal(got[key].view(np.uint32), vols[key[:-1]].view(np.uint32)), key
