#!/usr/bin/env python3
"""End-to-end ``process_fileset`` timing (mask ingest -> device).

Writes the S1 scene's masks as PNG files (what a ``Masks`` fileset holds), then times
``Backprojection.process_fileset`` from PNG files on disk to the volume in host memory:

  serial   decode_workers=1, views_per_launch=1 + a synchronize per view: the reference's
           schedule (read -> upload -> launch -> queue.finish per view, cl.py:282-303,226)
  default  decode-ahead threads + deferred fused carve (what the drop-in does)

Both are checked equal.  Prints one JSON line.  Not the headline metric (that is bench.py).
"""
import argparse, json, os, sys, tempfile, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class PngFile:
    """Duck-type of a plantdb File whose pixels are decoded from disk on every read."""

    def __init__(self, fid, path, md):
        self.id, self.path, self._md = fid, path, md

    def get_metadata(self, key=None, default=None):
        return self._md if key is None else self._md.get(key, default)

    def read_raw(self):  # plantdb's File.read_raw(): the bytes io.read_image decodes
        with open(self.path, "rb") as f:
            return f.read()

    def read_image(self):
        from PIL import Image
        with Image.open(self.path) as im:
            return np.array(im)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--views", type=int, default=72)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--workers", type=int, default=0, help="decode threads of the default run (0: the class's own choice)")
    a = ap.parse_args()
    from PIL import Image
    from plant3dvision_amd import scenes
    from plant3dvision_amd.cl import Backprojection

    shape, origin, vs, views = scenes.make_scene(a.n, a.views, "plant")
    tmp = tempfile.mkdtemp(prefix="sc_e2e_")
    files = []
    for q, (K, R, t, m) in enumerate(views):
        path = os.path.join(tmp, f"{q:05d}_mask.png")
        Image.fromarray(m).save(path, compress_level=1)
        files.append(PngFile(f"{q:05d}_mask", path, {"colmap_camera": scenes.camera_dict(K, R, t)}))
    t0 = time.perf_counter()
    for f in files:
        f.read_image()
    t_decode = time.perf_counter() - t0

    class SerialBackprojection(Backprojection):
        def _submit_view(self, *args, **kw):
            super()._submit_view(*args, **kw)
            self.synchronize()  # queue.finish() after every view, cl.py:226

    res = {}
    vols = {}
    class PilFile(PngFile):  # a file without read_raw: the Python decoder, as in round 1
        read_raw = None

    pil_files = [PilFile(f.id, f.path, f._md) for f in files]
    for name, cls, kw in (("serial", SerialBackprojection, dict(decode_workers=1, views_per_launch=1)),
                          ("pil", Backprojection, dict()),
                          ("default", Backprojection, dict(decode_workers=a.workers) if a.workers else dict())):
        best, runs = None, []
        for _ in range(a.reps):
            bp = cls(shape, origin, vs, **kw)
            t_ctor = time.perf_counter()
            vol = None  # the previous volume goes back to the OS outside the timed region (munmap of 512 MiB: 25 ms)
            t0 = time.perf_counter()
            vol = bp.process_fileset(pil_files if name != "default" else files, "colmap_camera")
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            runs.append(round(dt, 5))
            vols[name] = vol if name == "default" else vol.copy()
            bp.close()
        vol = None
        res[name] = best
        res[name + "_runs"] = runs
    assert np.array_equal(vols["serial"], vols["default"]) and np.array_equal(vols["pil"], vols["default"])
    n_vv = int(np.prod(shape)) * len(views)
    print(json.dumps({"workload": f"{a.n}^3 x {len(views)} PNG masks {views[0][3].shape[1]}x{views[0][3].shape[0]} -> volume in host memory",
                      "decode_only_s": t_decode, "serial_s": res["serial"], "python_decoder_s": res["pil"], "default_s": res["default"],
                      "default_runs_s": res["default_runs"], "speedup": res["serial"] / res["default"],
                      "default_Mvoxel_views_per_s": n_vv / res["default"] / 1e6,
                      "serial_Mvoxel_views_per_s": n_vv / res["serial"] / 1e6,
                      "host_threads": os.cpu_count()}))
    for f in files:
        os.remove(f.path)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
