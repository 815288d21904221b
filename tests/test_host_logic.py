"""CPU: the host-side mirror of the reference interface (constructor contract, label loop,
Voxels grid math and post-processing), with the oracle standing in for the device."""
import os

import numpy as np
import pytest

from oracle import oracle_c
from plant3dvision_amd import scenes
from plant3dvision_amd.cl import EPS, Backprojection, img_as_float32
from plant3dvision_amd.tasks import cl as tasks_cl
from tests.helpers import FakeFile, FakeFileset, OracleBackprojection, files_from_views, scene


def test_ctor_contract_like_reference_unit_test():
    # reference tests/unit/test_cl.py:5-9
    bp = OracleBackprojection([10, 10, 10], [0.0, 0.0, 0.0], 1.0)
    assert bp.dtype == np.int32 and bp.kernel == "carve"
    assert bp.get_values().shape == (10, 10, 10) and (bp.get_values() == 0).all()
    bp = OracleBackprojection([10, 10, 10], [0.0, 0.0, 0.0], 1.0, 'averaging')
    assert bp.dtype == np.float32 and bp.kernel == "average"
    for attr in ("shape", "origin", "voxel_size", "default_value", "log", "labels", "values_h",
                 "values_d", "intrinsics_d", "rot_d", "tvec_d", "volinfo_d", "shape_d"):
        assert hasattr(bp, attr)


def test_unknown_type_raises_value_error():
    with pytest.raises(ValueError, match="Unknown kernel type"):
        Backprojection([4, 4, 4], [0, 0, 0], 1.0, type="median")  # cl.py:152


def test_img_as_float32_matches_skimage_rule():
    m = np.arange(256, dtype=np.uint8).reshape(16, 16)
    f = img_as_float32(m)
    assert f.dtype == np.float32
    assert np.array_equal(f, m.astype(np.float32) * np.float32(1.0 / 255))
    assert img_as_float32(np.array([[True, False]])).tolist() == [[1.0, 0.0]]
    g = np.linspace(0, 1, 7, dtype=np.float32).reshape(1, 7)
    assert img_as_float32(g) is g


def test_process_label_filters_channel_and_missing_camera():
    shape, origin, vs, views = scene(16, 4, "plant")
    files = files_from_views(views, "colmap_camera", channel="stem")
    files += files_from_views(views[:2], "colmap_camera", channel="leaf")
    files.append(FakeFile("nocam", views[0][3], {"channel": "stem"}))  # skipped with a warning
    bp = OracleBackprojection(shape, origin, vs)
    vol = bp.process_label(FakeFileset(files), "colmap_camera", "stem")
    assert bp._engine.views_seen == 4
    assert np.array_equal(vol, oracle_c.carve(shape, origin, vs, views))


def test_process_fileset_labels_float64_and_clear_sequence():
    shape, origin, vs, views = scene(12, 3, "plant")
    stem = files_from_views(views, "camera", channel="stem")
    bg = [FakeFile(f.id.replace("stem", "background"), np.invert(f.array),
                   {"camera": f.get_metadata("camera"), "channel": "background"}) for f in stem]
    bp = OracleBackprojection(shape, origin, vs, labels=["stem", "background"])
    res = bp.process_fileset(stem + bg, "camera")
    assert res.dtype == np.float64 and res.shape == (2, *shape)  # cl.py:249
    assert np.array_equal(res[0], oracle_c.carve(shape, origin, vs, views))
    inv = [(K, R, t, np.invert(m)) for K, R, t, m in views]
    assert np.array_equal(res[1], oracle_c.carve(shape, origin, vs, inv))


def test_invert_is_numpy_invert_on_raw_dtype():
    shape, origin, vs, views = scene(12, 3, "plant")
    files = files_from_views(views, "colmap_camera")
    bp = OracleBackprojection(shape, origin, vs)
    vol = bp.process_fileset(files, "colmap_camera", invert=True)
    inv = [(K, R, t, np.invert(m)) for K, R, t, m in views]
    assert np.array_equal(vol, oracle_c.carve(shape, origin, vs, inv))


def test_invert_bool_and_int_masks():
    shape, origin, vs, views = scene(12, 3, "plant")
    for conv in (lambda m: m != 0, lambda m: m.astype(np.int32)):
        vv = [(K, R, t, conv(m)) for K, R, t, m in views]
        bp = OracleBackprojection(shape, origin, vs, decode_workers=3)
        vol = bp.process_fileset(files_from_views(vv, "colmap_camera"), "colmap_camera", invert=True)
        inv = [(K, R, t, np.invert(m)) for K, R, t, m in vv]
        assert np.array_equal(vol, oracle_c.carve(shape, origin, vs, inv))


def test_averaging_log_path_matches_reference_ops():
    shape, origin, vs, views = scene(10, 3, "plant")
    bp = OracleBackprojection(shape, origin, vs, type="averaging", log=True)
    vol = bp.process_fileset(files_from_views(views, "colmap_camera"), "colmap_camera")
    fviews = [(K, R, t, np.log(EPS + img_as_float32(m))) for K, R, t, m in views]
    assert vol.dtype == np.float32
    assert np.array_equal(vol, oracle_c.average(shape, origin, vs, fviews))


def test_grid_from_bounding_box_literal_config():
    # configs/test_geom_pipe_real.toml:31-36 with voxel_size 0.5 -> 301 x 301 x 561
    bbox = {"x": [300, 450], "y": [300, 450], "z": [-175, 105]}
    shape, origin = tasks_cl.grid_from_bounding_box(bbox, 0.5)
    assert shape == [301, 301, 561] and origin == [300, 300, -175]
    shape, origin = tasks_cl.grid_from_bounding_box(bbox, 0.5, {"dx": 1.5, "dy": -2, "dz": 0})
    assert shape == [301, 301, 561] and origin == [301.5, 298, -175]


def test_voxels_run_carving_and_metadata():
    shape, origin, vs, views = scene(16, 4, "plant")
    bbox = {a: [o, o + (n - 1) * vs] for a, o, n in zip("xyz", origin, shape)}
    vol, labels, md = tasks_cl.voxels_run(files_from_views(views), bbox, voxel_size=vs,
                                          backprojection_cls=OracleBackprojection)
    assert labels is None and md == {"voxel_size": vs, "origin": origin}
    assert np.array_equal(vol, oracle_c.carve(shape, origin, vs, views))


def test_voxels_run_averaging_exp_clip_and_labels():
    shape, origin, vs, views = scene(10, 3, "plant")
    bbox = {a: [o, o + (n - 1) * vs] for a, o, n in zip("xyz", origin, shape)}
    files = files_from_views(views, channel="stem")
    vol, labels, md = tasks_cl.voxels_run(files, bbox, voxel_size=vs, type="averaging", log=True,
                                          fileset_label_names=["stem"],
                                          backprojection_cls=OracleBackprojection)
    assert labels == ["stem"] and set(vol) == {"stem"}
    fviews = [(K, R, t, np.log(EPS + img_as_float32(m))) for K, R, t, m in views]
    want = np.exp(oracle_c.average(shape, origin, vs, fviews).astype(np.float64))
    want[want > 1] = 1.0
    assert np.array_equal(vol["stem"], want)  # tasks/cl.py:172-174


def test_voxels_run_without_bounding_box_exits():
    with pytest.raises(SystemExit):
        tasks_cl.voxels_run([], None, backprojection_cls=OracleBackprojection)  # tasks/cl.py:120-122


def test_voxels_run_post_processing_helpers_match_numpy_expressions():
    """tasks/cl.py:168,172-174: `len(np.unique(vol)) == 1` and `np.exp(vol); vol[vol > 1] = 1`
    are restated (no sort; np.minimum instead of a boolean-mask store) -- the results must be the same arrays."""
    from plant3dvision_amd.tasks.cl import _exp_clip, _single_valued
    rng = np.random.default_rng(5)
    for dt in (np.float32, np.float64):
        v = (rng.standard_normal((33, 200, 180)) * 3).astype(dt)
        v[3, 4, 5] = np.nan
        v[1, 2, 3] = -np.inf
        v[0, 0, 1] = np.inf
        want = np.exp(v)
        want[want > 1] = 1.0
        got = _exp_clip(v)
        assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True)
        small = v[:2, :3, :4]
        w2 = np.exp(small); w2[w2 > 1] = 1.0
        assert np.array_equal(_exp_clip(small), w2, equal_nan=True)
    # big volumes go slab by slab over host threads: the same bits as one call
    big = (rng.standard_normal((1 << 22) + 12345) * 4).astype(np.float32)
    big[[0, 77, -1]] = [np.nan, np.inf, -np.inf]
    want = np.exp(big)
    want[want > 1] = 1.0
    for workers in (None, 1, 3, 7):
        got = _exp_clip(big, workers=workers)
        assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert not np.shares_memory(_exp_clip(big), big)
    for arr in (big.copy(), big[:1000].copy()):
        w = np.exp(arr); w[w > 1] = 1.0
        got = _exp_clip(arr, inplace=True)
        assert got is arr and np.array_equal(got.view(np.uint32), w.view(np.uint32))
    last = np.zeros(10000); last[-1] = 1
    for arr in (np.zeros((5, 5)), np.arange(9.0), np.full((3, 3), np.nan), np.array([np.nan, 1.0]),
                np.array([1.0, np.nan]), np.zeros(10000, dtype=np.int32), last, np.full((2, 2), -1, dtype=np.int32)):
        assert _single_valued(arr) == (len(np.unique(arr)) == 1), arr


def test_touched_empty_is_a_plain_array_with_resident_pages():
    """``_native.TouchedEmpty``: the read-back buffers whose pages are touched on host threads."""
    from plant3dvision_amd._native import TouchedEmpty
    for shape, dt in (((3, 5, 7), np.int32), ((64, 64, 300), np.float32), ((0, 4), np.int32)):
        arr = TouchedEmpty(shape, dt, threads=3).result()
        assert arr.shape == shape and arr.dtype == dt and arr.flags["C_CONTIGUOUS"] and arr.flags["WRITEABLE"]
        arr[...] = 7
        assert (arr == 7).all()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("radius_factor", [0.3, 0.76, 0.8, 1.0, 2.0])
def test_synthetic_scene_stays_small_for_any_camera_ring(radius_factor):
    """The plant phantom is sampled on a lattice whose spacing follows the nearest camera depth; a
    ring just outside 0.75 x the extent once asked for ~10^10 points (host memory of the GPU box)."""
    import resource
    from plant3dvision_amd import scenes
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    shape, origin, vs, views = scenes.make_scene((8, 24, 40), 3, "plant", radius_factor=radius_factor)
    assert len(views) == 3 and views[0][3].dtype == np.uint8
    grown_mb = (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before) / 1024.0
    assert grown_mb < 1500, grown_mb


def test_sharded_process_view_does_the_reference_conversions():
    """ShardedBackprojection.process_view = Backprojection.process_view (cl.py:205-215): a uint8
    mask in averaging mode goes through img_as_float32 (and log), not a bare float cast."""
    from plant3dvision_amd.sharded import ShardedBackprojection
    from tests.helpers import OracleEngine
    shape, origin, vs, views = scene(12, 3, "plant")
    rng = np.random.default_rng(3)
    grey = [(K, R, t, rng.integers(0, 256, m.shape, dtype=np.uint8)) for K, R, t, m in views]
    for log in (False, True):
        conv = (lambda m: np.log(EPS + img_as_float32(m))) if log else img_as_float32
        with np.errstate(divide="ignore"):
            want = oracle_c.average(shape, origin, vs, [(K, R, t, conv(m)) for K, R, t, m in grey])
        for world, part in ((1, "cyclic"), (3, "cyclic"), (2, "slab")):
            parts = {}
            for r in range(world):
                sb = ShardedBackprojection(shape, origin, vs, type="averaging", rank=r, world_size=world,
                                           engine_factory=OracleEngine, partition=part, log=log)
                for K, R, t, m in grey:
                    sb.process_view(K, R, t, m)
                for i, plane in zip(sb.planes, sb.get_local()):
                    parts[i] = plane
            got = np.stack([parts[i] for i in range(shape[0])])
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (log, world, part)
    # bool masks in carving mode, float masks in averaging mode: as the class does
    sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, engine_factory=OracleEngine)
    for K, R, t, m in views:
        sb.process_view(K, R, t, m != 0)
    assert np.array_equal(sb.get_local(), oracle_c.carve(shape, origin, vs, views))


def test_clear_never_reuses_an_array_the_caller_holds_unless_recycled():
    shape, origin, vs, views = scene(10, 3, "plant")
    bp = OracleBackprojection(shape, origin, vs)
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    first = bp.get_values()
    keep = first.copy()
    alias = first  # a second reference changes nothing: ownership is explicit, not counted
    bp.clear()
    assert (bp.get_values() == 0).all()          # cl.py:307-311
    assert np.array_equal(first, keep) and alias is first  # the array handed out kept its contents
    second = bp.get_values()
    assert not np.shares_memory(first, second)
    bp.recycle(first)                             # handed back: the next read-back may land in it
    bp.clear()
    third = bp.get_values()
    assert np.shares_memory(third, first)
    bp.recycle(np.zeros(3, dtype=np.int32))       # wrong size: ignored
    bp.recycle(np.zeros(shape, dtype=np.float32)) # wrong dtype: ignored
    bp.clear()
    assert bp.get_values().dtype == np.int32


def test_native_png_decoder_matches_pil_and_refuses_what_it_does_not_know():
    """The ingest decoder (sc_png_decode_gray8) on 8-bit greyscale PNGs of every zlib level and size
    parity, against the pixels they were written from; other kinds of PNG are refused (None) so that
    the caller's usual reader takes them; read_image() prefers it for files that hand out bytes."""
    import io
    from PIL import Image
    from plant3dvision_amd import _native as nat
    from plant3dvision_amd.cl import read_image
    rng = np.random.default_rng(0)
    pics = [rng.integers(0, 256, (37, 53), dtype=np.uint8), (rng.random((128, 160)) < 0.1).astype(np.uint8) * 255,
            np.zeros((1, 1), np.uint8), np.tile(np.arange(256, dtype=np.uint8), (300, 5))]
    for m in pics:
        for lvl in (0, 1, 6, 9):
            bio = io.BytesIO()
            Image.fromarray(m).save(bio, format="PNG", compress_level=lvl)
            got = nat.png_decode_gray8(bio.getvalue())
            assert got is not None and got.dtype == np.uint8 and np.array_equal(got, m), (m.shape, lvl)
    m = pics[1]
    for convert in ("RGB", "LA", "P", "1"):
        bio = io.BytesIO()
        Image.fromarray(m).convert(convert).save(bio, format="PNG")
        assert nat.png_decode_gray8(bio.getvalue()) is None, convert
    bio = io.BytesIO()
    Image.fromarray(m.astype(np.uint16) * 257).save(bio, format="PNG")  # 16-bit grey
    assert nat.png_decode_gray8(bio.getvalue()) is None
    bio = io.BytesIO()
    Image.fromarray(m).save(bio, format="PNG")
    good = bio.getvalue()
    assert nat.png_decode_gray8(good[:60]) is None and nat.png_decode_gray8(b"x" * 100) is None
    bad = bytearray(good); bad[len(bad) // 2] ^= 0xff  # corrupted stream: refused, not decoded wrongly
    assert nat.png_decode_gray8(bytes(bad)) is None

    class RawFile:
        def __init__(self, raw, array):
            self._raw, self.array = raw, array

        def read_raw(self):
            return self._raw

    assert np.array_equal(read_image(RawFile(good, None)), m)           # the native decoder
    rgb = io.BytesIO(); Image.fromarray(m).convert("RGB").save(rgb, format="PNG")
    assert read_image(RawFile(rgb.getvalue(), m)) is m                   # refused -> the usual reader


def test_png_decoder_survives_mutated_files_under_the_sanitizers(tmp_path):
    """csrc/pngdec.cpp reads files from disk: built for the CPU with ASan + UBSan, it must decode or refuse
    truncated / corrupted / foreign PNG files without touching memory it does not own (tools/png_fuzz.cpp)."""
    import shutil
    import subprocess
    from PIL import Image
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "png_fuzz")
    built = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-I" + os.path.join(root, "include"), "-o", exe, os.path.join(root, "tools", "png_fuzz.cpp"),
                            os.path.join(root, "plant-3d-vision_amd", "csrc", "pngdec.cpp"), "-lz"],
                           capture_output=True, text=True)
    if built.returncode != 0:
        pytest.skip("no sanitizer runtime for the host compiler here: " + built.stderr[-200:])
    rng = np.random.default_rng(0)
    seeds = []
    for i, (w, h) in enumerate([(64, 48), (1, 1), (33, 7), (257, 129)]):
        for name, arr, kw in (("b", (rng.random((h, w)) > 0.5).astype(np.uint8) * 255, dict(compress_level=1)),
                              ("g", rng.integers(0, 256, (h, w), dtype=np.uint8), dict(compress_level=9, optimize=True)),
                              ("s", (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8), dict())):
            path = str(tmp_path / f"{name}{i}.png")
            Image.fromarray(arr).save(path, **kw)
            seeds.append(path)
    Image.fromarray(rng.integers(0, 256, (20, 20, 3), dtype=np.uint8)).save(str(tmp_path / "rgb.png"))
    Image.fromarray(rng.integers(0, 65536, (20, 20)).astype(np.uint16)).save(str(tmp_path / "g16.png"))
    seeds += [str(tmp_path / "rgb.png"), str(tmp_path / "g16.png")]
    out = subprocess.run([exe, "1500"] + seeds, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "accepted" in out.stdout and "refused" in out.stdout
