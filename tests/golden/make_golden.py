#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ (run from the repo root, in the build
container where /root/reference exists):

    python tests/golden/make_golden.py

Inputs  : the reference's own test data ``tests/testdata/virtual_plant`` (18 views x
          channels, exact ``camera`` metadata per file) -- copied here as DATA (decoded
          pixels + pose numbers), never as source text.
Outputs : expected volumes computed by the CPU oracle (oracle/spacecarve_oracle.c), each
          asserted equal to the independent NumPy restatement (oracle/oracle_np.py) before
          it is written.  They are ORACLE outputs, not outputs of the reference's OpenCL
          run (see DESIGN.md "Oracle" for the pin status: parity unpinned).
"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_c, oracle_np  # noqa: E402
from plant3dvision_amd import scenes  # noqa: E402
from plant3dvision_amd.cl import img_as_float32  # noqa: E402
from plant3dvision_amd.tasks.cl import grid_from_bounding_box  # noqa: E402
from tests.helpers import histogram3, sha256  # noqa: E402

REF = "/root/reference/tests/testdata/virtual_plant"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_virtual_plant():
    from PIL import Image
    bbox = json.load(open(os.path.join(REF, "metadata", "images.json")))["bounding_box"]
    data = {"bbox": np.array([bbox["x"], bbox["y"], bbox["z"]], dtype=np.float64)}
    for channel in ("stem", "background"):
        masks, K, R, t = [], [], [], []
        for f in sorted(glob.glob(os.path.join(REF, "images", f"*_{channel}.png"))):
            stem = os.path.splitext(os.path.basename(f))[0]
            md = json.load(open(os.path.join(REF, "metadata", "images", stem + ".json")))
            assert md["channel"] == channel
            cam = md["camera"]
            masks.append(np.array(Image.open(f)))
            K.append(cam["camera_model"]["params"][0:4])
            R.append(cam["rotmat"])
            t.append(cam["tvec"])
        data[f"masks_{channel}"] = np.stack(masks).astype(np.uint8)
        data[f"K_{channel}"] = np.array(K, dtype=np.float64)
        data[f"R_{channel}"] = np.array(R, dtype=np.float64)
        data[f"t_{channel}"] = np.array(t, dtype=np.float64)
    return data


def vp_views(data, channel, invert=False):
    views = []
    for q in range(data[f"masks_{channel}"].shape[0]):
        m = data[f"masks_{channel}"][q]
        if invert:
            m = np.invert(m)  # cl.py:300-301
        views.append((data[f"K_{channel}"][q].astype(np.float32),
                      data[f"R_{channel}"][q].reshape(9).astype(np.float32),
                      data[f"t_{channel}"][q].astype(np.float32), m))
    return views


def both_carve(shape, origin, vs, views, default_value=0):
    a = oracle_c.carve(shape, origin, vs, views, default_value)
    if np.prod(shape) <= 160 ** 3:
        b = oracle_np.carve(shape, origin, vs, views, default_value)
        assert np.array_equal(a, b), "C and NumPy oracles disagree"
    return a


def main():
    data = load_virtual_plant()
    np.savez_compressed(os.path.join(OUT, "virtual_plant_inputs.npz"), **data)
    bbox = {"x": list(data["bbox"][0]), "y": list(data["bbox"][1]), "z": list(data["bbox"][2])}
    exp = {}
    for vs in (1.0, 0.5):
        shape, origin = grid_from_bounding_box(bbox, vs)
        tag = f"vs{str(vs).replace('.', '')}"
        exp[f"shape_{tag}"] = np.array(shape)
        exp[f"origin_{tag}"] = np.array(origin)
        lab = both_carve(shape, origin, vs, vp_views(data, "stem"))
        exp[f"carve_stem_{tag}"] = lab.astype(np.int8)
        print("stem", tag, shape, histogram3(lab))
        lab = both_carve(shape, origin, vs, vp_views(data, "background", invert=True))
        exp[f"carve_background_invert_{tag}"] = lab.astype(np.int8)
        print("background/invert", tag, shape, histogram3(lab))
    shape, origin = grid_from_bounding_box(bbox, 1.0)
    fviews = [(K, R, t, img_as_float32(m)) for K, R, t, m in vp_views(data, "stem")]
    avg = oracle_c.average(shape, origin, 1.0, fviews)
    assert np.array_equal(avg, oracle_np.average(shape, origin, 1.0, fviews))
    exp["average_stem_nolog_vs10"] = avg
    np.savez_compressed(os.path.join(OUT, "virtual_plant_expected.npz"), **exp)

    syn = {}
    for n, v, kind in ((32, 6, "plant"), (64, 12, "plant"), (48, 5, "noise")):
        shape, origin, vs, views = scenes.make_scene(n, v, kind)
        lab = both_carve(shape, origin, vs, views)
        syn[f"{kind}_{n}_{v}"] = lab.astype(np.int8)
        print(kind, n, v, histogram3(lab))
    digests = {}
    for n, v, kind in ((128, 12, "plant"), (128, 12, "noise"), (128, 12, "solid")):
        shape, origin, vs, views = scenes.make_scene(n, v, kind)
        lab = both_carve(shape, origin, vs, views)
        digests[f"{kind}_{n}_{v}"] = {"sha256_int32": sha256(lab.astype(np.int32)),
                                      "hist_m1_0_p1": histogram3(lab)}
        print(kind, n, v, digests[f"{kind}_{n}_{v}"])
    # non-cubic slice of the literal test_geom_pipe_real.toml grid (301x301x561, vs 0.5)
    shape, origin, vs, views = scenes.make_scene((61, 45, 113), 8, "plant")
    lab = both_carve(shape, origin, vs, views)
    syn["plant_61x45x113_8"] = lab.astype(np.int8)
    np.savez_compressed(os.path.join(OUT, "synthetic_expected.npz"), **syn)
    path = os.path.join(OUT, "synthetic_digests.json")
    if os.path.exists(path):  # keep the full-size entries of big_digests()
        digests = {**json.load(open(path)), **digests}
    json.dump(digests, open(path, "w"), indent=1, sort_keys=True)


def big_digests(nthreads=None):
    """BASELINE cfg 3 (512^3 x 72: the benchmarked scene and the bench's other three) and cfg 4 (1024^3 x 72, the
    planes of every rank of 8, both partitions: rank_digests): SHA-256 + histogram of the ORACLE's int32 labels over the
    whole grid / the rank's planes.  `python tests/golden/make_golden.py big` (about ten minutes on 8 cores; the
    sizes at which tests and bench.py's parity_check compare the HIP path with these)."""
    from plant3dvision_amd.sharded import rank_planes
    nthreads = nthreads or min(32, os.cpu_count() or 8)
    path = os.path.join(OUT, "synthetic_digests.json")
    digests = json.load(open(path))
    for kind in ("plant", "solid", "dense", "noise"):
        shape, origin, vs, views = scenes.make_scene(512, 72, kind)
        lab = oracle_c.carve(shape, origin, vs, views, nthreads=nthreads)
        digests[f"{kind}_512_72"] = {"sha256_int32": sha256(lab), "hist_m1_0_p1": histogram3(lab)}
        print(kind, 512, 72, digests[f"{kind}_512_72"], flush=True)
        del lab
    json.dump(digests, open(path, "w"), indent=1, sort_keys=True)
    rank_digests(nthreads)


def rank_digests(nthreads=None, only_missing=False):
    """BASELINE cfg 4: EVERY rank of 8 of the 1024^3 x 72 grid, both partitions (16 x 128 planes), and the histogram of
    the whole grid as the sum of a partition's ranks (the two partitions must agree: a checksum of checksums).
    `python tests/golden/make_golden.py ranks` computes only the keys that are not there yet."""
    from plant3dvision_amd.sharded import rank_planes
    nthreads = nthreads or min(32, os.cpu_count() or 8)
    path = os.path.join(OUT, "synthetic_digests.json")
    digests = json.load(open(path))
    shape, origin, vs, views = scenes.make_scene(1024, 72, "plant")
    for partition in ("cyclic", "slab"):
        for rank in range(8):
            key = f"plant_1024_72_{partition}_rank{rank}of8"
            if only_missing and key in digests:
                continue
            pl = rank_planes(shape[0], 8, rank, partition)
            lab = oracle_c.carve_planes(shape, origin, vs, views, pl.start, pl.step, len(pl), nthreads=nthreads)
            digests[key] = {"sha256_int32": sha256(lab), "hist_m1_0_p1": histogram3(lab)}
            print(key, digests[key], flush=True)
            del lab
    # the weak-scaling grids of bench.py at 2 and 4 GPUs (bench.py GRIDS_512): what rank 0 compares its digest with
    for world, gshape in ((2, (640, 640, 640)), (4, (808, 800, 832))):
        key = f"plant_{gshape[0]}_72_cyclic_rank0of{world}"
        if only_missing and key in digests:
            continue
        sh, og, vz, vw = scenes.make_scene(gshape, 72, "plant")
        pl = rank_planes(sh[0], world, 0, "cyclic")
        lab = oracle_c.carve_planes(sh, og, vz, vw, pl.start, pl.step, len(pl), nthreads=nthreads)
        digests[key] = {"sha256_int32": sha256(lab), "hist_m1_0_p1": histogram3(lab), "global_grid": list(gshape)}
        print(key, digests[key], flush=True)
        del lab
    sums = {}
    for partition in ("cyclic", "slab"):
        hs = [digests[f"plant_1024_72_{partition}_rank{r}of8"]["hist_m1_0_p1"] for r in range(8)]
        sums[partition] = [int(sum(h[q] for h in hs)) for q in range(3)]
    assert sums["cyclic"] == sums["slab"] and sum(sums["cyclic"]) == 1024 ** 3, sums
    digests["plant_1024_72_whole_grid"] = {"hist_m1_0_p1": sums["cyclic"]}
    print("plant_1024_72_whole_grid", sums["cyclic"], flush=True)
    json.dump(digests, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        big_digests()
    elif len(sys.argv) > 1 and sys.argv[1] == "ranks":
        rank_digests(only_missing=True)
    else:
        main()
