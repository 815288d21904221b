#!/usr/bin/env python3
"""cProfile of Backprojection.process_fileset on 72 PNG masks (diagnostic for tools/bench_e2e.py)."""
import cProfile, pstats, os, sys, tempfile, time, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_e2e import PngFile
from PIL import Image
from plant3dvision_amd import scenes
from plant3dvision_amd.cl import Backprojection

shape, origin, vs, views = scenes.make_scene(512, 72, "plant")
tmp = tempfile.mkdtemp(prefix="sc_e2e_")
files = []
for q, (K, R, t, m) in enumerate(views):
    path = os.path.join(tmp, f"{q:05d}_mask.png")
    Image.fromarray(m).save(path, compress_level=1)
    files.append(PngFile(f"{q:05d}_mask", path, {"colmap_camera": scenes.camera_dict(K, R, t)}))
for rep in range(3):
    bp = Backprojection(shape, origin, vs)
    time.sleep(0.2)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    vol = bp.process_fileset(files, "colmap_camera")
    pr.disable()
    dt = time.perf_counter() - t0
    bp.close()
print("last rep: %.1f ms" % (dt * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
print(s.getvalue()[:4000])
