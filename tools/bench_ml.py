#!/usr/bin/env python3
"""BASELINE cfg 5 in outline: stand-in segmentation network (PyTorch-ROCm) -> per-label masks ->
512^3 volume, everything on one MI355X, masks never leaving HBM.  Prints one JSON line.
The network is a seeded stand-in (romiseg and its weights are not vendored): the number that
matters is the mask -> volume part."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--views", type=int, default=72)
    ap.add_argument("--size", type=int, default=896)
    ap.add_argument("--type", default="averaging")
    ap.add_argument("--labels", type=int, default=3, help="labels turned into volumes (1..6)")
    a = ap.parse_args()
    import torch
    from plant3dvision_amd import masks2d, scenes
    labels = ["background", "flower", "fruit", "leaf", "pedicel", "stem"]
    S = a.size
    shape, origin, vs, views = scenes.make_scene(a.n, a.views, "solid", width=S, height=S, fx=0.8 * S, fy=0.8 * S, cx=S / 2, cy=S / 2)
    cams = [scenes.camera_dict(K, R, t) for K, R, t, _ in views]
    torch.manual_seed(0)
    coarse = torch.rand(a.views, 3, 14, 14, device="cuda")
    images = torch.nn.functional.interpolate(coarse, size=(S, S), mode="bilinear", align_corners=False)
    net = masks2d.StandInSegmenter(labels, seed=1)
    def sync(): torch.cuda.synchronize()
    res = {}
    use = labels[:max(1, min(a.labels, len(labels)))]
    vols = None
    for it in range(3):  # first pass warms up
        sync(); t0 = time.perf_counter()
        pred = torch.cat([net(images[i:i + 8]) for i in range(0, a.views, 8)])
        sync(); t1 = time.perf_counter()
        thr = 0.3
        masks = masks2d.masks_from_predictions(pred, labels, labels=use, threshold=thr, dilation=1)
        sync(); t2 = time.perf_counter()
        res = {"network_s": t1 - t0, "masks_s": t2 - t1}
        for name, overlap in (("one_engine", False), ("two_engines", True)):
            vols = None  # the previous volumes go back to the OS outside the timed region
            t2 = time.perf_counter()
            vols = masks2d.voxels_from_masks(masks, cams, shape, origin, vs, type=a.type, log=a.type == "averaging",
                                             overlap=overlap)
            res[f"volumes_incl_readback_{name}_s"] = time.perf_counter() - t2
        vols1 = None
        t2 = time.perf_counter()
        vols1 = masks2d.voxels_from_masks({use[0]: masks[use[0]]}, cams, shape, origin, vs, type=a.type,
                                          log=a.type == "averaging")
        res["volume_incl_readback_s"] = time.perf_counter() - t2
        vols1 = None
    # device time alone: the labels in one launch (sc_average_labels) against one launch per label
    if a.type == "averaging" and len(use) > 1:
        from plant3dvision_amd import _native as nat
        from plant3dvision_amd.cl import averaging_table
        K = np.array([c["camera_model"]["params"][0:4] for c in cams], dtype=np.float32)
        R = np.array([sum(c["rotmat"], []) for c in cams], dtype=np.float32)
        t = np.array([c["tvec"] for c in cams], dtype=np.float32)
        engs = [nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE) for _ in use]
        for e in engs:
            e.set_lut(averaging_table(True))
        ptrs = [masks[name].data_ptr() for name in use]
        sync()

        def shared():
            for e in engs:
                e.clear()
            nat.average_labels(engs, K, R, t, ptrs, a.views, S, S)
            for e in engs:
                e.synchronize()

        def one_by_one():
            for e, p in zip(engs, ptrs):
                e.clear()
                e.process_views_device(K, R, t, p, a.views, S, S, nat.SC_MASK_U8_LUT)
                e.flush()
            for e in engs:
                e.synchronize()

        for fn, key in ((shared, "device_ms_labels_in_one_launch"), (one_by_one, "device_ms_label_by_label")):
            fn()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            res[key] = (time.perf_counter() - t0) / 5 * 1e3
        for e in engs:
            e.close()
    v = vols["background"]
    nvv = int(np.prod(shape)) * a.views
    res.update({"workload": f"{a.views} images {S}x{S} -> stand-in net (6 labels) -> {len(use)} label(s) {use} -> {a.n}^3 {a.type} "
                            f"volumes in host memory (exp / clip included); volume_incl_readback_s = the first label alone",
                "labels": len(use),
                "mask_to_volume_Mvoxel_views_per_s": nvv / res["volume_incl_readback_s"] / 1e6,
                "volume_min_max": [float(v.min()), float(v.max())]})
    print(json.dumps(res))

if __name__ == "__main__":
    main()
