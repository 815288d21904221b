#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random grid shapes, scenes, camera set-ups (rings inside the grid,
principal points thousands of pixels off the picture, focal lengths up to 1e5, voxel sizes from 1e-3
to 1e3), default values and pipeline knobs; every volume must equal the oracle's, fresh and on a
second batch, through host masks or a device batch.
Usage: timeout -k 10 900 python tools/fuzz_carve.py [cases] [seed] [summary.json] [share of big cases, default 0]
(big cases: grids of up to 48 x 160 x 320 voxels and up to 40 views -- tens of blocks in the flags kernel, several per
sub-list of the candidate list, riders in most cases)
FUZZ_ONLY="17,330": replay the same draws but run only those cases, print where their labels differ and which of
their knobs the mismatch needs (each left out in turn).
(diagnostic, not part of the test suite; prints each case before it runs so that a fault can be traced
to its parameters; the summary of the round's run is kept under profiles/)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat, scenes
from oracle import oracle_c
from tests.helpers import unpack_sparse_np

KNOBS = {  # (round 6: the sixteen settled tuning keys are retired -- accepted, no effect -- and left the draw)
    "SC_OPT_FLAG_VIEWS": [0, 1, 3, 8, 11], "SC_OPT_DENSE_VIEWS": [1, 2, 3], "SC_OPT_STAGE1_VIEWS": [1, 4, 8, 64],
    "SC_OPT_FULL_BRICKS": [0, 1], "SC_OPT_BRICK": [0, 1, 1, 1], "SC_OPT_COMPACT": [0, 1, 1, 1], "SC_OPT_VIEW_ORDER": [0, 1],
    "SC_OPT_PACK_ROWS": [0, 1, 2, 3, 3, 3, 4, 8], "SC_OPT_VIEWS_PER_LAUNCH": [0, 0, 0, 1, 5], "SC_OPT_PACK_RIDE": [0, 1],
    "SC_OPT_VIEW_BRICK": [0, 1], "SC_OPT_BULK_MIN": [0, 1, 64, 128, 256], "SC_OPT_BULK_FLOOR": [0, 1, 16, 2048, 1 << 30],
    "SC_OPT_UNIT_CULL": [0, 1, 2, 2], "SC_OPT_HOST_PACK": [0, 1, 1], "SC_OPT_BULK_LIVE": [0, 0, 2, 16],
    "SC_OPT_SAFE_KERNELS": [0, 1, 1], "SC_OPT_LATE_ROAD": [0, 1, 1], "SC_OPT_LIST_CAP": [0, 0, 0, 2, 16, 300],
}

ranked = 0


def ranks_assembly(sh, origin, vs, views, dv, opts, device_masks, world, part, cap, want, stack, K, R, t):
    """The labels of `world` rank engines through the brick-sparse wire into one grid; returns what differs (or None)."""
    from plant3dvision_amd.sharded import rank_planes, slab_bounds
    from tests.helpers import sparse_header_np
    planes = [rank_planes(sh[0], world, r, part) for r in range(world)]
    nbmax = nat.sparse_bricks(max(len(p) for p in planes), sh[1], sh[2])
    engines = []
    for r in range(world):
        kw = {"cyclic": (r, world)} if part == "cyclic" else {"slab": slab_bounds(sh[0], world, r)}
        e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE, default_value=dv, **kw)
        for k, v in opts.items():
            e.set_option(getattr(nat, k), v)
        if device_masks:
            ptr = e.dev_alloc(stack.nbytes); e.dev_upload(ptr, stack)
            e.process_views_device(K, R, t, ptr, *stack.shape, nat.SC_MASK_U8)
            e.synchronize(); e.dev_free(ptr)
        else:
            for Kq, Rq, tq, m in views:
                e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8)
        engines.append(e)
    out = {}
    for attempt in range(2):
        stride = nat.sparse_rank_bytes(nbmax, cap)
        wire = np.zeros(stride * world, dtype=np.uint8)
        over = False
        for r, e in enumerate(engines):
            buf = e.get_values_sparse(cap)
            h = sparse_header_np(buf)
            over |= h["nmixed"] > h["cap"]
            if buf.size > stride:
                out["stride"] = (r, int(buf.size), int(stride))
            wire[r * stride:r * stride + min(buf.size, stride)] = buf[:stride]
        if not over:
            break
        cap = nbmax  # a rank ran out of slots: every rank again with room for every brick (what ShardedBackprojection does)
    e0 = engines[0]
    recv = e0.dev_alloc(wire.nbytes); e0.dev_upload(recv, wire)
    n = int(np.prod(sh))
    dst = e0.dev_alloc(n * 4)
    nat.unpack_sparse(0, e0.stream(), recv, stride, world, sh, dst, 4)
    e0.synchronize()
    got = np.empty(n, dtype=np.int32)
    e0.dev_download(got, dst)
    e0.dev_free(dst); e0.dev_free(recv)
    if not np.array_equal(got.reshape(sh), want):
        out["device_unpack"] = int((got.reshape(sh) != want).sum())
    host = nat.widen_sparse_ranks(wire, stride, world, sh)
    if not np.array_equal(host, want):
        out["host_widen"] = int((host != want).sum())
    for e in engines:
        e.close()
    return out or None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    big_share = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    only = {int(x) for x in os.environ.get("FUZZ_ONLY", "").split(",") if x.strip()}
    bad = 0
    certified_views = uncertified_views = averaged = 0
    global ranked
    for c in range(cases):
        shape = (int(rng.integers(2, 24)), int(rng.integers(2, 70)), int(rng.integers(2, 200)))
        kind = str(rng.choice(["plant", "noise", "solid", "empty", "dense"]))
        nviews = int(rng.integers(1, 16))
        if big_share > 0 and rng.random() < big_share:
            bx, by_, bz_ = (int(x) for x in os.environ.get("FUZZ_BIG_MAX", "48,160,320").split(","))  # (e.g. 128,320,640)
            shape = (int(rng.integers(16, bx + 1)), int(rng.integers(48, by_ + 1)), int(rng.integers(100, bz_ + 1)))
            nviews = int(rng.integers(12, 41))
        kw = dict(radius_factor=float(rng.choice([0.3, 0.8, 1.5, 3.0])), tilt_deg=float(rng.choice([0.0, 0.0, 25.0, 50.0])),
                  voxel_size=float(rng.choice([1e-3, 0.5, 0.5, 0.5, 1.7, 1e3])))
        if rng.random() < 0.4:
            w, h = int(rng.integers(2, 26)) * 16 if rng.random() < 0.6 else int(rng.integers(20, 400)), int(rng.integers(20, 300))
            kw.update(width=w, height=h, fx=float(rng.uniform(0.3, 3.0) * w), fy=float(rng.uniform(0.3, 3.0) * w),
                      cx=float(rng.uniform(0.2, 0.8) * w), cy=float(rng.uniform(0.2, 0.8) * h))
            r = rng.random()
            if r < 0.15:    # principal point far outside the picture
                kw.update(cx=float(rng.choice([-1, 1]) * 1e4), cy=float(rng.choice([-1, 1]) * 7e3))
            elif r < 0.3:   # telephoto
                kw.update(fx=1e5, fy=1e5)
        sh, origin, vs, views = scenes.make_scene(shape, nviews, kind, **kw)
        ncert = sum(nat.view_certified(sh, origin, vs, Kq, Rq, tq) for Kq, Rq, tq, _ in views)
        certified_views += ncert
        uncertified_views += len(views) - ncert
        dv = int(rng.choice([0, 0, 0, 1, -1, 5]))
        opts = {k: int(rng.choice(v)) for k, v in KNOBS.items() if rng.random() < 0.5}
        run_it = not only or c in only
        device_masks = rng.random() < 0.5
        # (round 6) the same carve as the ranks of an N > 1 run hold it: W engines on plane sets of the grid, their sparse
        # buffers laid rank-major as an all-gather leaves them, unpacked on the device and on the host into ONE grid
        ranks_w = int(rng.integers(2, 9)) if rng.random() < 0.3 else 0
        ranks_part = str(rng.choice(["cyclic", "slab"]))
        ranks_cap = int(rng.choice([16, 64, 4096, 1 << 20]))
        ok = True
        if run_it:
            print(f"case {c}: shape {sh} {kind} views {nviews} dv {dv} kw {kw} opts {opts}", flush=True)
            want = oracle_c.carve(sh, origin, vs, views, dv, nthreads=4)
            stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
            K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])

            def carve_twice(o):
                e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE, default_value=dv)
                for k, v in o.items():
                    e.set_option(getattr(nat, k), v)
                ptr = e.dev_alloc(stack.nbytes); e.dev_upload(ptr, stack)
                gots = []
                for rnd in range(2):
                    if device_masks:
                        e.process_views_device(K, R, t, ptr, *stack.shape, nat.SC_MASK_U8)
                    else:
                        for Kq, Rq, tq, m in views:
                            e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8)
                    gots.append(e.get_values().copy())
                    # (round 6) the brick-sparse transport form of the same labels, packed from this batch's verdict bytes
                    # and lists (or from the dead bytes / from nothing, by the labels' history): it must decode to them
                    if dv in (-1, 0, 1):
                        buf = e.get_values_sparse(nat.sparse_bricks(*sh))
                        dec = unpack_sparse_np(buf, buf.size, 1, sh)
                        if not np.array_equal(dec, gots[-1]):
                            sparse_bad.append((rnd, int((dec != gots[-1]).sum())))
                counts = e.fused_counts_ex()
                e.dev_free(ptr); e.close()
                return gots, counts

            sparse_bad = []
            gots, counts = carve_twice(opts)
            if sparse_bad:
                ok = False
                print(f"SPARSE MISMATCH case {c}: shape {sh} {kind} views {nviews} dv {dv} opts {opts} device {device_masks}: {sparse_bad}")
            for rnd, got in enumerate(gots):
                if not np.array_equal(got, want):
                    ok = False
                    print(f"MISMATCH case {c} round {rnd}: shape {sh} {kind} views {nviews} dv {dv} kw {kw} opts {opts} "
                          f"device {device_masks}: {int((got != want).sum())} voxels differ")
            if ranks_w and dv in (-1, 0, 1) and sh[0] >= ranks_w:
                nbad = ranks_assembly(sh, origin, vs, views, dv, opts, device_masks, ranks_w, ranks_part, ranks_cap, want, stack, K, R, t)
                ranked += 1
                if nbad:
                    ok = False
                    print(f"RANKS MISMATCH case {c}: shape {sh} {kind} views {nviews} dv {dv} opts {opts} device {device_masks} "
                          f"world {ranks_w} {ranks_part} cap {ranks_cap}: {nbad}")
            if only and not ok:  # where, what, and which knobs it takes
                got = gots[0]
                badv = np.argwhere(got != want)
                print("   counts", counts)
                print("   planes", np.unique(badv[:, 0])[:24], "strips y", np.unique(badv[:, 1] // 16)[:24], "bricks z", np.unique(badv[:, 2] // 64))
                pairs, n = np.unique(np.stack([got[got != want], want[got != want]], 1), axis=0, return_counts=True)
                print("   (got, want):", [(tuple(int(x) for x in pq), int(q)) for pq, q in zip(pairs, n)])
                for b in badv[:12]:
                    print("    ", tuple(int(x) for x in b), "got", int(got[tuple(b)]), "want", int(want[tuple(b)]))
                for k in list(opts):
                    o = dict(opts); del o[k]
                    g2, _ = carve_twice(o)
                    print(f"   without {k}={opts[k]}: {int((g2[0] != want).sum())} / {int((g2[1] != want).sum())} differ", flush=True)
                g3, _ = carve_twice({})
                print(f"   no knobs: {int((g3[0] != want).sum())} / {int((g3[1] != want).sum())} differ", flush=True)
        # the average kernel on the same rig (every third case): uint8 masks through the table or float32
        # masks, binary or grey, fused / per view, brick and tile forms on or off -- bitwise against the oracle
        if c % 3 == 0 and int(np.prod(sh)) * nviews < 4e7:
            from plant3dvision_amd.cl import averaging_table
            log = bool(rng.random() < 0.5)
            table = averaging_table(log)
            form = str(rng.choice(["u8", "f32"]))
            grey = rng.random() < 0.5
            adv = float(rng.choice([0.0, 0.0, -0.0, 1.5]))
            ms = []
            for _, _, _, m in views:
                if form == "u8":
                    ms.append(rng.integers(0, 256, m.shape, dtype=np.uint8) if grey else m)
                else:
                    ms.append(rng.random(m.shape, dtype=np.float32) if grey else table[m])
            aopts = {"SC_OPT_VIEWS_PER_LAUNCH": int(rng.choice([0, 0, 1, 4])), "SC_OPT_AVG_BRICK": int(rng.choice([0, 1, 1])),
                     "SC_OPT_AVG_TILE_F32": int(rng.choice([0, 1, 1]))}
            if not run_it:
                continue
            fviews = [(Kq, Rq, tq, (table[m] if form == "u8" else m)) for (Kq, Rq, tq, _), m in zip(views, ms)]
            wantf = oracle_c.average(sh, origin, vs, fviews, adv, nthreads=4)
            print(f"   average: form {form} grey {grey} log {log} default {adv} opts {aopts}", flush=True)
            ea = nat.Engine(sh, origin, vs, nat.SC_MODE_AVERAGE, default_value=adv)
            ea.set_lut(table)
            for k, v in aopts.items():
                ea.set_option(getattr(nat, k), v)
            for rnd in range(2):
                if rnd:
                    ea.clear()
                for (Kq, Rq, tq, _), m in zip(views, ms):
                    ea.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8_LUT if form == "u8" else nat.SC_MASK_F32)
                gotf = ea.get_values()
                if not np.array_equal(gotf.view(np.uint32), wantf.view(np.uint32)):
                    ok = False
                    print(f"MISMATCH (average) case {c} round {rnd}: {int((gotf.view(np.uint32) != wantf.view(np.uint32)).sum())} voxels differ")
            ea.close()
            averaged += 1
        bad += 0 if ok else 1
    print(f"{ranked} cases also as the ranks of an N > 1 run through the sparse wire")
    print(f"{cases} cases ({averaged} with the average kernel too), {bad} with mismatches; {certified_views} certified views, {uncertified_views} not")
    if len(sys.argv) > 3:
        import json
        json.dump({"tool": "tools/fuzz_carve.py", "cases": cases, "seed": int(sys.argv[2]), "cases_with_mismatches": bad,
                   "cases_with_the_average_kernel_too": averaged,
                   "cases_also_as_ranks_through_the_sparse_wire": ranked,
                   "views_on_the_certified_projection_path": certified_views, "views_on_the_general_path": uncertified_views,
                   "knobs": sorted(KNOBS), "scenes": ["plant", "noise", "solid", "empty", "dense"],
                   "checked": "HIP carve (host masks or device batch, fresh volume and a second batch on the stored one) "
                              "== oracle/spacecarve_oracle.c, every voxel"}, open(sys.argv[3], "w"), indent=1)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
