#!/usr/bin/env python3
"""Where the 72 PNG -> 512^3 volume time goes (diagnostic): phases of Backprojection.process_label timed apart."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_e2e import PngFile
from PIL import Image
from plant3dvision_amd import scenes, _native as nat
from plant3dvision_amd.cl import Backprojection, read_image
from concurrent.futures import ThreadPoolExecutor

shape, origin, vs, views = scenes.make_scene(512, 72, "plant")
tmp = tempfile.mkdtemp(prefix="sc_e2e_")
files = []
for q, (K, R, t, m) in enumerate(views):
    path = os.path.join(tmp, f"{q:05d}_mask.png")
    Image.fromarray(m).save(path, compress_level=1)
    files.append(PngFile(f"{q:05d}_mask", path, {"colmap_camera": scenes.camera_dict(K, R, t)}))
T = time.perf_counter
for workers, decoder in ((8, 'native'), (16, 'native'), (8, 'pil'), (16, 'pil')):
    for rep in range(3):
        vol = buf = masks = None  # freed outside the timed phases (munmap of a 512 MiB array: 25 ms)
        t0 = T()
        bp = Backprojection(shape, origin, vs, decode_workers=workers)
        t1 = T()
        with ThreadPoolExecutor(max_workers=workers) as pool:
            futs = [pool.submit(read_image if decoder == 'native' else PngFile.read_image, f) for f in files]
            masks = [f.result() for f in futs]
        t2 = T()
        for (K, R, t, _), m in zip(views, masks):
            bp.process_view(K, R, t, m)
        t3 = T()
        bp.synchronize()
        t4 = T()
        buf = bp._take_buffer()
        t5 = T()
        bp._values_h = buf
        vol = bp.get_values()
        t6 = T()
        bp.close()
    print(f"workers {workers} ({decoder} decoder): ctor {1e3*(t1-t0):.1f}  decode-all {1e3*(t2-t1):.1f}  submit {1e3*(t3-t2):.1f}  sync {1e3*(t4-t3):.1f}  "
          f"take_buffer {1e3*(t5-t4):.1f}  get_values {1e3*(t6-t5):.1f} ms")
# read-back variants alone
bp = Backprojection(shape, origin, vs)
for (K, R, t, m) in views:
    bp.process_view(K, R, t, m)
bp.synchronize()
e = bp._engine
out32 = np.empty(shape, np.int32); out32[...] = 0
out8 = np.empty(shape, np.int8); out8[...] = 0
for rep in range(3):
    t0 = T(); e.get_values(out32); t1 = T(); e.get_values_i8(out8); t2 = T(); nat.widen_i8(out32, out8); t3 = T()
    fresh = np.empty(shape, np.int32); t4 = T(); nat.widen_i8(fresh, out8); t5 = T()
    t6 = T(); te = nat.TouchedEmpty(tuple(shape), np.int32).result(); t7 = T()
print(f"D2H int32 touched {1e3*(t1-t0):.1f}  D2H int8 {1e3*(t2-t1):.1f}  widen touched {1e3*(t3-t2):.1f}  widen fresh {1e3*(t5-t4):.1f}  TouchedEmpty(16) {1e3*(t7-t6):.1f} ms")
bp.close()
for f in files:
    os.remove(f.path)
os.rmdir(tmp)
