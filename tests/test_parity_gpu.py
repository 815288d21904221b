"""GPU: the HIP engine, called through the C ABI, against the CPU oracle and the committed
golden fixtures.  Bar: bit-exact for carve (int32) AND for average (float32 -- the sum is
performed in the same order with the same IEEE operations, so tolerance is zero)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_c
from plant3dvision_amd import _native as nat
from plant3dvision_amd.cl import EPS, Backprojection, averaging_table, img_as_float32
from plant3dvision_amd.tasks import cl as tasks_cl
from tests.helpers import files_from_views, histogram3, scene, sha256

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
THREADS = min(32, os.cpu_count() or 8)


def hip_carve(shape, origin, vs, views, default_value=0, views_per_launch=0, device=0,
              view_order=1, compact=1):
    bp = Backprojection(shape, origin, vs, default_value=default_value, device=device,
                        views_per_launch=views_per_launch)
    bp._engine.set_option(nat.SC_OPT_VIEW_ORDER, view_order)
    bp._engine.set_option(nat.SC_OPT_COMPACT, compact)
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    out = bp.get_values().copy()
    bp.close()
    return out


def hip_average(shape, origin, vs, fviews, default_value=0, views_per_launch=0, device=0):
    bp = Backprojection(shape, origin, vs, type="averaging", default_value=default_value,
                        device=device, views_per_launch=views_per_launch)
    for K, R, t, m in fviews:
        bp.process_view(K, R, t, m)
    out = bp.get_values().copy()
    bp.close()
    return out


def test_library_sees_gfx950(gpu_device):
    assert nat.device_count() >= 1


@pytest.mark.parametrize("mode,seed", [(1, 1), (1, 77), (0, 5), (2, 3), (2, 2024)])
def test_shared_reciprocal_division_is_bit_identical(gpu_device, mode, seed):
    """The kernels divide p_x and p_y by p_z through one refined reciprocal; inside its
    operand range that must equal hipcc's IEEE division bit for bit (2^32 triples a run).  Mode 2: numerators
    below that range under a certified view -- the pixel and the picture test must be the exact quotient's."""
    e = nat.Engine([4, 4, 4], [0, 0, 0], 1.0, nat.SC_MODE_CARVE)
    bad, fast = e.selftest_division(1 << 32, seed=seed, mode=mode)
    e.close()
    assert bad == 0
    if mode in (1, 2):
        assert fast == 1 << 32  # projection-like operands all take the fast path
    else:
        assert 0 < fast < 1 << 31  # raw bit patterns mostly fall outside the range


def test_ctor_like_reference_unit_test(gpu_device):
    # reference tests/unit/test_cl.py:5-9, plus what the reference never checks: the values
    bp = Backprojection([10, 10, 10], [0.0, 0.0, 0.0], 1.0)
    v = bp.get_values()
    assert v.dtype == np.int32 and v.shape == (10, 10, 10) and (v == 0).all()
    bp = Backprojection([10, 10, 10], [0.0, 0.0, 0.0], 1.0, 'averaging', default_value=2.5)
    v = bp.get_values()
    assert v.dtype == np.float32 and (v == 2.5).all()


@pytest.mark.parametrize("n,v,kind", [(32, 6, "plant"), (48, 5, "noise"), ((20, 31, 18), 6, "plant"),
                                      ((7, 5, 3), 4, "plant"), ((9, 9, 1), 3, "noise"),
                                      ((61, 45, 113), 8, "plant"), (16, 3, "solid"), (16, 3, "empty")])
@pytest.mark.parametrize("vpl,compact", [(0, 1), (0, 0), (1, 1), (3, 1)])
def test_carve_matches_oracle(gpu_device, n, v, kind, vpl, compact):
    """vpl = views per launch: 0 fused (with / without survivor compaction), 1 the
    reference's one-launch-per-view, 3 chunks."""
    shape, origin, vs, views = scene(n, v, kind)
    want = oracle_c.carve(shape, origin, vs, views, nthreads=4)
    got = hip_carve(shape, origin, vs, views, views_per_launch=vpl, compact=compact)
    assert got.dtype == np.int32
    assert np.array_equal(got, want), histogram3(got)


def test_carve_given_order_equals_perpendicular_first_order(gpu_device):
    shape, origin, vs, views = scene(40, 9, "plant")
    want = oracle_c.carve(shape, origin, vs, views)
    for order in (0, 1):
        for compact in (0, 1):
            assert np.array_equal(hip_carve(shape, origin, vs, views, view_order=order, compact=compact), want)


@pytest.mark.parametrize("kind,n,v", [("solid", 40, 12), ("noise", 40, 12), ("solid", (33, 20, 30), 9)])
def test_masks_that_carve_little_through_both_schedules(gpu_device, kind, n, v):
    shape, origin, vs, views = scene(n, v, kind)
    want = oracle_c.carve(shape, origin, vs, views, nthreads=4)
    assert np.array_equal(hip_carve(shape, origin, vs, views, compact=1), want)
    assert np.array_equal(hip_carve(shape, origin, vs, views, compact=0), want)


def _batch(e, views, host):
    if host:
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        return None
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = e.dev_alloc(stack.nbytes)
    e.dev_upload(ptr, stack)
    K = np.stack([q[0] for q in views]); R = np.stack([q[1] for q in views]); t = np.stack([q[2] for q in views])
    e.process_views_device(K, R, t, ptr, *stack.shape, nat.SC_MASK_U8)
    return ptr


@pytest.mark.parametrize("kind,shape,v", [("noise", (8, 32, 128), 12), ("plant", (12, 48, 128), 10), ("dense", (9, 32, 192), 12)])
@pytest.mark.parametrize("host", [False, True])
@pytest.mark.parametrize("brick", [1, 0])
def test_survivor_list_overflow_takes_the_dense_pass(gpu_device, kind, shape, v, host, brick):
    """A survivor sub-list that runs out of room in the dense stage (SC_OPT_LIST_CAP makes that happen on a small
    grid) raises the overflow flag: the list kernels leave and the special kernel applies the remaining views
    densely -- same labels, on a fresh volume and on a second batch over the stored one."""
    sh, origin, vs, views = scene(shape, v, kind)
    want = oracle_c.carve(sh, origin, vs, views, nthreads=4)
    e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE)
    e.set_option(nat.SC_OPT_LIST_CAP, 4)
    e.set_option(nat.SC_OPT_BRICK, brick)
    e.set_option(nat.SC_OPT_UNIT_CULL, 2)
    ptr = _batch(e, views, host)
    assert np.array_equal(e.get_values(), want), (kind, host, brick, histogram3(want))
    assert e.fused_counts()[3] == 1, "the lists did not overflow: the test tests nothing"
    ptr2 = _batch(e, views, host)
    assert np.array_equal(e.get_values(), want), "second batch over the stored volume"
    for q in (ptr, ptr2):
        if q is not None:
            e.dev_free(q)
    e.close()


@pytest.mark.parametrize("device_masks", [True, False])
def test_failed_reservations_leave_no_holes_in_the_lists(gpu_device, device_masks):
    """Fuzz case 878 of round 4 (a GPU memory fault before the fix): all-foreground masks through a wide-angle
    camera whose pictures the tall grid sticks out of, every unit with a voxel alive on the bulk list, two entries of
    room per survivor sub-list.  The units' voxels find no room on the list; a plain atomic add that fails leaves
    the counter beyond the capacity and the slots below it unwritten, and the final stage then read those slots as
    entries.  Reservations are compare-and-swap now (list_reserve): a counter never passes its capacity."""
    kw = dict(radius_factor=0.8, tilt_deg=0.0, voxel_size=1000.0, width=48, height=71, fx=24.813724957700188,
              fy=131.65757157747763, cx=26.850375195941375, cy=22.261666802611273)
    sh, origin, vs, views = scene((11, 13, 58), 15, "solid", **kw)
    want = oracle_c.carve(sh, origin, vs, views, nthreads=4)
    assert 0 < (want == 0).sum() < want.size  # part of the grid is outside every picture
    opts = {"SC_OPT_DENSE_VIEWS": 3, "SC_OPT_STAGE1_VIEWS": 64, "SC_OPT_DEFER_SHARE": 5, "SC_OPT_COMPACT": 1,
            "SC_OPT_BRICK_WALKERS": 8, "SC_OPT_FILL_BLOCKS": 1, "SC_OPT_STAGE1_VOXELS": 1, "SC_OPT_VIEW_BRICK": 1,
            "SC_OPT_STAGE1_LIST_BLOCKS": 1280, "SC_OPT_BULK_MIN": 1, "SC_OPT_BULK_LIVE": 0, "SC_OPT_ITEM_BIAS": 0, "SC_OPT_UNIT_CULL": 0,
            "SC_OPT_LIST_CAP": 2}
    for stage1 in (64, 4):  # a single (final) list stage, and two
        e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE)
        for k, v in opts.items():
            e.set_option(getattr(nat, k), v)
        e.set_option(nat.SC_OPT_STAGE1_VIEWS, stage1)
        for rnd in range(2):
            ptr = _batch(e, views, not device_masks)
            assert np.array_equal(e.get_values(), want), (device_masks, stage1, rnd, histogram3(want))
            if ptr is not None:
                e.dev_free(ptr)
        e.close()


@pytest.mark.parametrize("nviews", [6, 7, 10, 11, 13])
@pytest.mark.parametrize("kind,shape", [("dense", (9, 32, 192)), ("plant", (12, 48, 128))])
def test_bulk_units_whatever_the_number_of_list_stages(gpu_device, nviews, kind, shape):
    """Bulk units below the floor are taken by the FIRST survivor stage as they are; a batch short enough to have
    only the final stage (<= 10 views) has no such stage and must have its units asked instead."""
    sh, origin, vs, views = scene(shape, nviews, kind)
    want = oracle_c.carve(sh, origin, vs, views, nthreads=4)
    for floor in (None, 0):
        e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE)
        e.set_option(nat.SC_OPT_BULK_MIN, 1)
        e.set_option(nat.SC_OPT_BULK_LIVE, 0)  # the list is kept however few bricks are live
        if floor is not None:
            e.set_option(nat.SC_OPT_BULK_FLOOR, floor)
        ptr = _batch(e, views, False)
        assert np.array_equal(e.get_values(), want), (nviews, kind, floor, histogram3(want))
        assert e.fused_counts_ex()["bulk_units"] > 0
        e.dev_free(ptr)
        e.close()


@pytest.mark.parametrize("kind,shape", [("dense", (9, 32, 192)), ("plant", (12, 48, 128)), ("solid", (6, 32, 128))])
@pytest.mark.parametrize("floor", [0, 1 << 30])
def test_bulk_unit_without_room_in_the_lists_is_carved_on_the_spot(gpu_device, kind, shape, floor):
    """Every unit with a voxel alive goes on the bulk list (SC_OPT_BULK_MIN 1), so the dense stage appends nothing
    and cannot overflow; the special kernel then finds no room for the units' voxels in the sub-lists
    (SC_OPT_LIST_CAP) and takes each such unit through every view itself -- asked first (floor 0) or not."""
    sh, origin, vs, views = scene(shape, 11, kind)
    want = oracle_c.carve(sh, origin, vs, views, nthreads=4)
    e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE)
    for k, val in ((nat.SC_OPT_LIST_CAP, 2), (nat.SC_OPT_BULK_MIN, 1), (nat.SC_OPT_BULK_LIVE, 0), (nat.SC_OPT_BULK_FLOOR, floor),
                   (nat.SC_OPT_ITEM_BIAS, 0), (nat.SC_OPT_UNIT_CULL, 0)):
        e.set_option(k, val)
    ptr = _batch(e, views, False)
    assert np.array_equal(e.get_values(), want), (kind, floor, histogram3(want))
    c = e.fused_counts_ex()
    assert c["list_overflow"] == 0, "only the dense stage may raise the flag"
    if kind != "solid":
        assert c["bulk_units"] > 0
    e.dev_free(ptr)
    e.close()


@pytest.mark.parametrize("shape,kw", [
    ((6, 5, 512), dict(radius_factor=2.0)),                      # tall columns: full-wave z-runs
    ((6, 5, 512), dict(radius_factor=0.2)),                      # camera ring INSIDE the columns' extent
    ((5, 4, 768), dict(radius_factor=0.6, tilt_deg=35.0)),       # oblique: runs cross many tiles
    ((4, 4, 1024), dict(radius_factor=1.0, width=200, height=90, fx=150.0, fy=150.0, cx=100.0, cy=45.0)),
    ((3, 7, 561), dict(radius_factor=1.5)),                      # nz % 4 != 0, wavefronts span columns
])
@pytest.mark.parametrize("kind", ["plant", "noise", "empty"])
def test_tall_columns_close_and_oblique_cameras(gpu_device, shape, kw, kind):
    """Long z-runs (full wavefronts inside one column), cameras inside the columns' extent,
    oblique views, runs that leave the image or pass behind the camera, nz % 4 != 0."""
    sh, origin, vs, views = scene(shape, 7, kind, **kw)
    want = oracle_c.carve(sh, origin, vs, views, nthreads=4)
    for vpl in (0, 1):
        assert np.array_equal(hip_carve(sh, origin, vs, views, views_per_launch=vpl), want), \
            (vpl, histogram3(want))


@pytest.mark.parametrize("shape,kw", [
    ((5, 16, 128), dict(radius_factor=2.0)),
    ((5, 16, 128), dict(radius_factor=0.25)),                    # cameras inside the bricks' extent
    ((3, 32, 192), dict(radius_factor=0.7, tilt_deg=40.0)),      # oblique: bricks cover many tiles
    ((4, 48, 64), dict(radius_factor=1.0, width=200, height=90, fx=150.0, fy=150.0, cx=100.0, cy=45.0)),
    ((2, 16, 64), dict(radius_factor=3.0, width=2000, height=1500, fx=3000.0, fy=3000.0, cx=1000.0, cy=750.0)),
])
@pytest.mark.parametrize("kind", ["plant", "noise", "empty", "solid"])
def test_brick_culling_is_exact(gpu_device, shape, kw, kind):
    """The dense stage works on 16x64-voxel bricks with a conservative emptiness test.  It must never change a label: compare with the oracle and with the brick
    form switched off, on geometries that stress its guards (bricks that leave the image, lie
    behind or around the camera, cover many tiles, or sit over completely empty masks)."""
    sh, origin, vs, views = scene(shape, 8, kind, **kw)
    want = oracle_c.carve(sh, origin, vs, views, nthreads=4)
    for brick in (1, 0):
        bp = Backprojection(sh, origin, vs)
        bp._engine.set_option(nat.SC_OPT_BRICK, brick)
        for K, R, t, m in views:
            bp.process_view(K, R, t, m)
        assert np.array_equal(bp.get_values(), want), (brick, histogram3(want))
        # a second fused launch on the stored (non-fresh) state
        for K, R, t, m in views:
            bp.process_view(K, R, t, m)
        assert np.array_equal(bp.get_values(), want), (brick, "second pass")
        bp.close()


@pytest.mark.parametrize("opts", [
    {},                                                                   # defaults
    {"SC_OPT_FLAG_VIEWS": 0},                                             # every view may veto a brick
    {"SC_OPT_FLAG_VIEWS": 1, "SC_OPT_DENSE_VIEWS": 1},
    {"SC_OPT_FLAG_VIEWS": 11, "SC_OPT_DENSE_VIEWS": 3, "SC_OPT_STAGE1_VIEWS": 3},
    {"SC_OPT_DEFER_STORES": 0},                                           # dense stage fills empty bricks
    {"SC_OPT_DEFER_SHARE": 0},
    {"SC_OPT_DEFER_SHARE": 7, "SC_OPT_DEFER_STORES": 24},                 # fill split between the two
    {"SC_OPT_LIST_BLOCKS": 8, "SC_OPT_DEFER_STORES": 8},                  # tiny persistent grids
    {"SC_OPT_STAGE1_VIEWS": 64},                                          # single (final) list stage
    {"SC_OPT_STAGE1_VIEWS": 2, "SC_OPT_STAGE2_VIEWS": 3, "SC_OPT_VIEW_GROUP": 2},
    {"SC_OPT_VIEW_GROUP": 5, "SC_OPT_PACK_ROWS": 1},
    {"SC_OPT_FULL_BRICKS": 0},
    {"SC_OPT_STAGE1_STORE_SHARE": 0},                                     # the final stage fills everything
    {"SC_OPT_STAGE1_STORE_SHARE": 16, "SC_OPT_STAGE1_LIST_BLOCKS": 8},    # ... the first stage does
    {"SC_OPT_STAGE1_STORE_SHARE": 9, "SC_OPT_DEFER_SHARE": 11, "SC_OPT_DEFER_STORES": 40},  # all three kernels fill
    {"SC_OPT_COMPACT": 0},                                                # bricks without survivor lists
    {"SC_OPT_VIEW_ORDER": 0},
    {"SC_OPT_FILL_BLOCKS": 0},                                            # one short store block per strip
    {"SC_OPT_FILL_BLOCKS": 3, "SC_OPT_STAGE1_STORE_SHARE": 8},            # a few persistent ones
    {"SC_OPT_PACK_RIDE": 0},                                              # every mask packed ahead
    {"SC_OPT_FINAL_VOXELS": 1},                                           # one survivor per lane in the final stage
    {"SC_OPT_FINAL_VOXELS": 4, "SC_OPT_STAGE1_VOXELS": 2},                # four, one view per turn; two in the first stage
    {"SC_OPT_STAGE1_VOXELS": 4, "SC_OPT_VIEW_GROUP": 3},
    {"SC_OPT_BRICK_WALKERS": 8, "SC_OPT_FILL_BLOCKS": 1, "SC_OPT_VIEW_ORDER": 0},
    {"SC_OPT_BULK_MIN": 0},                                               # no unit is finished as a whole
    {"SC_OPT_BULK_MIN": 1, "SC_OPT_BULK_FLOOR": 0, "SC_OPT_BULK_LIVE": 0},  # every unit with a voxel alive is, and asked
    {"SC_OPT_BULK_MIN": 1, "SC_OPT_BULK_FLOOR": 0, "SC_OPT_BULK_LIVE": 16},  # ... unless every brick is live: never
    {"SC_OPT_BULK_MIN": 1, "SC_OPT_BULK_LIVE": 0},                        # ... too few for the default floor: spilled
    {"SC_OPT_BULK_MIN": 1, "SC_OPT_BULK_FLOOR": 0, "SC_OPT_BULK_LIVE": 0, "SC_OPT_ITEM_BIAS": 64},  # items whatever they cost
    {"SC_OPT_BULK_MIN": 1, "SC_OPT_BULK_FLOOR": 0, "SC_OPT_BULK_LIVE": 0, "SC_OPT_ITEM_BIAS": 0},   # never items: asked, then the lists
    {"SC_OPT_BULK_MIN": 256, "SC_OPT_FULL_BRICKS": 0, "SC_OPT_BULK_FLOOR": 0},
    {"SC_OPT_BULK_MIN": 40, "SC_OPT_DENSE_VIEWS": 1, "SC_OPT_LIST_BLOCKS": 8, "SC_OPT_BULK_FLOOR": 0},
    {"SC_OPT_BULK_MIN": 40, "SC_OPT_UNIT_BLOCKS": 1, "SC_OPT_BULK_FLOOR": 3},  # one block does all the special kernel has
    {"SC_OPT_PACK_ROWS": 4},                                              # the panel form of the pack kernel
    {"SC_OPT_PACK_ROWS": 3, "SC_OPT_PACK_RIDE": 0},                       # bands, every mask packed ahead
    {"SC_OPT_PACK_ROWS": 8, "SC_OPT_PACK_RIDE": 0},
    {"SC_OPT_UNIT_CULL": 0},                                              # no unit verdicts in the dense stage
    {"SC_OPT_UNIT_CULL": 2, "SC_OPT_BULK_MIN": 1, "SC_OPT_BULK_FLOOR": 0},  # ... asked whatever the tiles settled
    {"SC_OPT_UNIT_CULL": 2, "SC_OPT_PACK_RIDE": 0, "SC_OPT_BRICK_WALKERS": 8},  # by 16 views, few walkers
    {"SC_OPT_SAFE_KERNELS": 0},                                           # the list kernels with the general path compiled in
    {"SC_OPT_DENSE_EXTRA": 0},                                            # no third pair of dense views for thinned-out units
    {"SC_OPT_SPEC_SHARE": 0},                                             # no fill ahead of the verdicts
    {"SC_OPT_SPEC_SHARE": 16, "SC_OPT_SPEC_BLOCKS": 7},                   # ... all of it, by an odd number of blocks
    {"SC_OPT_SPEC_SHARE": 9, "SC_OPT_FILL_BLOCKS": 0},
    {"SC_OPT_LATE_ROAD": 0},                                              # failed candidates on the late list, always
    {"SC_OPT_LATE_ROAD": 1, "SC_OPT_BULK_MIN": 1, "SC_OPT_BULK_FLOOR": 0, "SC_OPT_BULK_LIVE": 0},  # ... among many bulk units
    {"SC_OPT_LATE_ROAD": 1, "SC_OPT_BULK_FLOOR": 1 << 30},               # ... alone: the bulk units spill, the late ones are asked
    {"SC_OPT_DEFER_SHARE": 5},                                            # the dense kernel fills most strips itself (and nobody rides)
    {"SC_OPT_DEFER_SHARE": 2, "SC_OPT_FILL_BLOCKS": 0},
    {"SC_OPT_FILL_BLOCKS": 64, "SC_OPT_DEFER_SHARE": 11, "SC_OPT_STAGE1_STORE_SHARE": 3},  # few store blocks, odd shares
    {"SC_OPT_DENSE_VIEWS": 3, "SC_OPT_UNIT_CULL": 0},
    {"SC_OPT_DEFER_STORES": 1280, "SC_OPT_STAGE1_VIEWS": 8, "SC_OPT_STAGE1_STORE_SHARE": 4, "SC_OPT_BRICK_WALKERS": 1024},  # rounds 3-5's defaults
])
@pytest.mark.parametrize("kind,shape", [("plant", (24, 32, 128)), ("noise", (6, 16, 64)), ("plant", (9, 48, 192)),
                                        ("dense", (14, 48, 192)),    # a bulky object: whole-brick masks at work
                                        ("plant", (7, 23, 70)),      # bricks stick out in y and z
                                        ("plant", (5, 37, 131))])    # ... and nz % 4 != 0: element accesses
def test_fused_pipeline_knobs_never_change_a_label(gpu_device, opts, kind, shape):
    """The fused carve is a pipeline (bit packing, brick verdicts + live list, dense stage on live
    bricks, survivor stages, -1 fill of empty bricks riding with the final stage).  Every knob
    that moves work between its kernels must leave the volume bit-identical to the oracle, on
    a fresh volume and on a second batch over the stored one, through host and device masks."""
    sh, origin, vs, views = scene(shape, 12, kind)
    want = oracle_c.carve(sh, origin, vs, views, nthreads=4)
    e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE)
    for k, v in opts.items():
        e.set_option(getattr(nat, k), v)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = e.dev_alloc(stack.nbytes)
    e.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
    assert np.array_equal(e.get_values(), want), ("device masks, fresh", opts, histogram3(want))
    live, s0, s1, overflow = e.fused_counts()
    nbricks = sh[0] * ((sh[1] + 15) // 16) * ((sh[2] + 63) // 64)
    assert 0 <= live <= nbricks
    # (bulk units too few to be asked are taken by the first survivor stage as they are: their survivors are on its
    # output list without ever having been on its input list)
    assert overflow or s1 <= s0 + 256 * e.fused_counts_ex()["bulk_units"]
    e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
    assert np.array_equal(e.get_values(), want), ("device masks, stored state", opts)
    e.clear()
    for Kq, Rq, tq, m in views:
        e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8)
    assert np.array_equal(e.get_values(), want), ("host masks", opts)
    e.dev_free(ptr)
    e.close()


@pytest.mark.parametrize("opts", [{"SC_OPT_BRICK_WALKERS": 8}, {"SC_OPT_LIST_BLOCKS": 8, "SC_OPT_PACK_RIDE": 0},
                                  {"SC_OPT_BRICK_WALKERS": 24}, {}])
def test_few_walker_blocks_on_a_long_live_list(gpu_device, opts):
    """The dense stage's walkers draw their bricks from eight ticket counters per XCD, each dealing every eighth run of
    16 live-list entries.  With 8 walker blocks an XCD has four wavefronts: they must share four counters -- with
    eight, the runs of the counters nobody holds went to nobody, and a live list of more than 512 bricks kept labels no
    view had been applied to (round 4, found by the fuzz sweep once it drew grids of this size)."""
    sh, origin, vs, views = scene((36, 142, 208), 12, "noise")
    want = oracle_c.carve(sh, origin, vs, views, nthreads=8)
    e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE)
    for k, v in opts.items():
        e.set_option(getattr(nat, k), v)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = e.dev_alloc(stack.nbytes)
    e.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    for state in ("fresh", "stored state"):
        e.process_views_device(K, R, t, ptr, *stack.shape, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), (opts, state)
    assert e.fused_counts()[0] > 512, "the scene is meant to leave a long live list"
    e.dev_free(ptr)
    e.close()


@pytest.mark.parametrize("kind,shape,nv", [("dense", (40, 128, 256), 30), ("solid", (36, 100, 200), 24),
                                           ("dense", (70, 64, 130), 40)])
def test_full_candidates_on_their_list_over_many_flag_blocks(gpu_device, kind, shape, nv):
    """FULL candidates (bricks every view packed ahead keeps whole) are listed by the flags kernel in eight sub-lists
    by block range and asked about the late views by the confirm kernel, 64 list entries per block (ListCtl::ncand).
    Grids of 16-35 flag blocks: several blocks per sub-list, sub-lists that end inside a group of 64, candidates that
    fail (late bricks) and candidates that hold -- every label equal to the oracle's, on a fresh volume, on a second
    batch over the stored one, and with the riders off (no candidate stays open)."""
    sh, origin, vs, views = scene(shape, nv, kind)
    want = oracle_c.carve(sh, origin, vs, views, nthreads=8)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    late = {}
    for ride in (1, 0):
        e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE)
        e.set_option(nat.SC_OPT_PACK_RIDE, ride)
        ptr = e.dev_alloc(stack.nbytes)
        e.dev_upload(ptr, stack)
        e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), (kind, "fresh", ride, histogram3(want))
        late[ride] = e.fused_counts_ex()["late_bricks"]
        e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), (kind, "stored state", ride)
        e.dev_free(ptr)
        e.close()
    assert late[0] == 0  # every mask packed ahead: the flags kernel settles FULL bricks itself
    if shape == (40, 128, 256):
        assert late[1] > 0, "the scene is meant to leave candidates that a late view rejects"


@pytest.mark.parametrize("shape", [(6, 32, 128), (5, 37, 131), (3, 16, 64)])
@pytest.mark.parametrize("default_value", [0, 1, -1, 7])
@pytest.mark.parametrize("defer", [1536, 0])
def test_bricks_every_view_keeps_whole(gpu_device, shape, default_value, defer):
    """A brick that EVERY view sees whole, in-image, over foreground only is labelled without
    projecting a voxel (0 -> 1, other labels stay).  Masks: all foreground ("solid"), then a
    disc big enough to hold the inner bricks but not the outer ones; a fresh volume, a second
    batch over the stored one, and a batch of inverted masks over that."""
    sh, origin, vs, views = scene(shape, 9, "solid")
    H, W = views[0][3].shape
    yy, xx = np.mgrid[0:H, 0:W]
    disc = (((yy - H / 2) ** 2 + (xx - W / 2) ** 2) < (0.30 * min(H, W)) ** 2).astype(np.uint8) * 255
    for masks in ([m for _, _, _, m in views], [disc] * len(views)):
        vv = [(K, R, t, m) for (K, R, t, _), m in zip(views, masks)]
        want = oracle_c.carve(sh, origin, vs, vv, default_value, nthreads=4)
        e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE, default_value=default_value)
        e.set_option(nat.SC_OPT_DEFER_STORES, defer)
        for K, R, t, m in vv:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), ("fresh", histogram3(want))
        for K, R, t, m in vv:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), "stored state"
        inv = [(K, R, t, np.invert(m)) for K, R, t, m in vv]
        for K, R, t, m in inv:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), oracle_c.carve(sh, origin, vs, vv + inv, default_value, nthreads=4))
        e.close()


@pytest.mark.parametrize("shape", [(6, 32, 128), (5, 37, 131), (9, 48, 192)])
@pytest.mark.parametrize("default_value", [0, 1, -1, 7])
@pytest.mark.parametrize("bulk_min,full", [(128, 1), (1, 0), (200, 0)])
def test_bulk_units_asked_as_a_whole(gpu_device, shape, default_value, bulk_min, full):
    """Units (a wavefront's share of a live brick) with most voxels alive after the dense views are finished
    as a whole: every view is asked about the unit from its summed-area table over 8x8-pixel cells --
    entirely over background: carved; entirely over foreground or out of the picture: nothing to project --
    and only the undecided views project its voxels.  A disc that holds the inner units and cuts through
    the outer ones (all three verdicts at work), the bulky ellipsoid, all-foreground masks; a fresh volume,
    a second batch over the stored one, and a batch of inverted masks over that."""
    sh, origin, vs, views = scene(shape, 11, "solid")
    H, W = views[0][3].shape
    yy, xx = np.mgrid[0:H, 0:W]
    disc = (((yy - H / 2) ** 2 + (xx - W / 2) ** 2) < (0.22 * min(H, W)) ** 2).astype(np.uint8) * 255
    dense = [m for _, _, _, m in scene(shape, 11, "dense")[3]]
    used = 0
    for masks in ([disc] * len(views), dense, [m for _, _, _, m in views]):
        vv = [(K, R, t, m) for (K, R, t, _), m in zip(views, masks)]
        want = oracle_c.carve(sh, origin, vs, vv, default_value, nthreads=4)
        e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE, default_value=default_value)
        e.set_option(nat.SC_OPT_BULK_MIN, bulk_min)
        e.set_option(nat.SC_OPT_BULK_LIVE, 0)
        e.set_option(nat.SC_OPT_BULK_FLOOR, 0)  # asked however few they are (a test grid has fewer than the default floor)
        e.set_option(nat.SC_OPT_FULL_BRICKS, full)
        stack = np.ascontiguousarray(np.stack(masks))
        ptr = e.dev_alloc(stack.nbytes)
        e.dev_upload(ptr, stack)
        K = np.stack([v[0] for v in vv]); R = np.stack([v[1] for v in vv]); t = np.stack([v[2] for v in vv])
        e.process_views_device(K, R, t, ptr, *stack.shape, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), ("device batch, fresh", histogram3(want))
        used += e.fused_counts_ex()["bulk_units"]
        for Kq, Rq, tq, m in vv:
            e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), "host masks, stored state"
        inv = [(Kq, Rq, tq, np.invert(m)) for Kq, Rq, tq, m in vv]
        for Kq, Rq, tq, m in inv:
            e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), oracle_c.carve(sh, origin, vs, vv + inv, default_value, nthreads=4))
        e.dev_free(ptr)
        e.close()
    if default_value != -1:
        assert used > 0, "no unit ever took the bulk path"


def test_fused_compaction_with_slab_and_default_values(gpu_device):
    shape, origin, vs, views = scene((40, 28, 36), 14, "plant")
    for dv in (0, 3):
        want = oracle_c.carve(shape, origin, vs, views, dv, nthreads=4)
        parts = []
        for i0, i1 in ((0, 17), (17, 40)):
            e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=dv, slab=(i0, i1))
            for K, R, t, m in views:
                e.process_view(K, R, t, m, nat.SC_MASK_U8)
            parts.append(e.get_values())
            e.close()
        assert np.array_equal(np.concatenate(parts, axis=0), want)


def test_fused_on_non_fresh_state(gpu_device):
    """Second fused launch starts from stored labels (non-FRESH dense stage + lists)."""
    shape, origin, vs, views = scene(48, 16, "plant")
    bp = Backprojection(shape, origin, vs)
    for K, R, t, m in views[:8]:
        bp.process_view(K, R, t, m)
    bp.synchronize()
    for K, R, t, m in views[8:]:
        bp.process_view(K, R, t, m)
    assert np.array_equal(bp.get_values(), oracle_c.carve(shape, origin, vs, views, nthreads=4))


@pytest.mark.parametrize("default_value", [1, -1, 7])
def test_carve_default_values(gpu_device, default_value):
    shape, origin, vs, views = scene(20, 4, "plant")
    want = oracle_c.carve(shape, origin, vs, views, default_value)
    for vpl in (0, 1):
        got = hip_carve(shape, origin, vs, views, default_value, views_per_launch=vpl)
        assert np.array_equal(got, want)


def test_camera_inside_volume_unseen_and_behind(gpu_device):
    """p_z <= 0 voxels, off-image voxels (label stays 0), tiny images."""
    shape, origin, vs, views = scene(24, 4, "solid", radius_factor=0.3, width=64, height=48,
                                     fx=40.0, fy=40.0, cx=32.0, cy=24.0)
    want = oracle_c.carve(shape, origin, vs, views)
    assert histogram3(want)[1] > 0
    assert np.array_equal(hip_carve(shape, origin, vs, views), want)
    shape, origin, vs, views = scene(30, 6, "noise", radius_factor=0.45, width=100, height=37,
                                     fx=55.0, fy=50.0, cx=50.0, cy=18.0, tilt_deg=12.0)
    want = oracle_c.carve(shape, origin, vs, views)
    assert min(histogram3(want)) > 0  # all three labels occur
    for vpl in (0, 1):
        assert np.array_equal(hip_carve(shape, origin, vs, views, views_per_launch=vpl), want)


def test_degenerate_poses_nan_inf_zero_depth(gpu_device):
    """H4/H5: NaN, inf and p_z == 0 must be rejected exactly as the canonical cast does."""
    shape, origin, vs = [8, 8, 8], [0.0, 0.0, 0.0], 1.0
    m = np.full((16, 16), 255, dtype=np.uint8)
    m[::2] = 0
    eye = np.eye(3, dtype=np.float32).reshape(9)
    views = [
        ([10.0, 10.0, 8.0, 8.0], eye, [0.0, 0.0, 0.0], m),           # p_z == 0 plane at k = 0
        ([10.0, 10.0, 8.0, 8.0], eye, [0.0, 0.0, -3.0], m),          # some voxels behind
        ([10.0, 10.0, 8.0, 8.0], eye, [0.0, 0.0, np.nan], m),        # NaN depth
        ([np.inf, 10.0, 8.0, 8.0], eye, [0.0, 0.0, 1.0], m),         # inf focal
        ([1e30, 1e30, 8.0, 8.0], eye, [1e30, 0.0, 1e-30], m),        # overflow to inf
        ([10.0, 10.0, -0.5, 7.5], eye, [0.0, 0.0, 1.0], m),          # u_f in (-1, 0) for x = 0
    ]
    want = oracle_c.carve(shape, origin, vs, views)
    for vpl in (0, 1):
        assert np.array_equal(hip_carve(shape, origin, vs, views, views_per_launch=vpl), want)


def test_mask_dtypes_bool_int32_grey(gpu_device):
    shape, origin, vs, views = scene(24, 4, "plant")
    want = oracle_c.carve(shape, origin, vs, views)
    as_bool = [(K, R, t, m != 0) for K, R, t, m in views]
    as_i32 = [(K, R, t, m.astype(np.int32) * 1000) for K, R, t, m in views]
    grey = [(K, R, t, np.where(m != 0, 1 + (np.arange(m.size).reshape(m.shape) % 200), 0).astype(np.uint8))
            for K, R, t, m in views]
    as_f64 = [(K, R, t, m.astype(np.float64) / 255.0 * 0.9) for K, R, t, m in views]  # int32 cast -> 0
    for vv in (as_bool, as_i32, grey):
        assert np.array_equal(hip_carve(shape, origin, vs, vv), want)
    assert np.array_equal(hip_carve(shape, origin, vs, as_f64), oracle_c.carve(shape, origin, vs, as_f64))


@pytest.mark.parametrize("host_pack", [1, 0])
@pytest.mark.parametrize("w,h", [(33, 17), (100, 70), (160, 96), (1440, 1080)])
def test_host_masks_cross_pcie_as_bits(gpu_device, host_pack, w, h):
    """a carve mask handed over in host memory is reduced to 1 bit per pixel on host threads
    (SC_OPT_HOST_PACK 1, the default) and the device makes tiles, occupancy bytes and cell maps from the bits; 0 is
    the byte path of rounds 1-3.  Grey levels, bool, int32 (negative values too), inverted uint8 / bool, widths
    that are not multiples of 32 or 16, one launch per view, per 5 views, per batch -- all against the oracle."""
    n = 20 if w < 1000 else 28
    shape, origin, vs, views = scene(n, 7, "plant", width=w, height=h, fx=0.8 * w, fy=0.8 * w, cx=w / 2.0, cy=h / 2.0)
    rng = np.random.default_rng(w * 31 + h)
    grey = [np.where(m != 0, rng.integers(1, 256, m.shape), 0).astype(np.uint8) for _, _, _, m in views]
    forms = [("grey", grey, nat.SC_MASK_U8, lambda m: m),
             ("bool", [m != 0 for m in grey], nat.SC_MASK_U8, lambda m: m),
             ("i32", [(m.astype(np.int64) * int(rng.choice([-70000, 1, 3]))).astype(np.int32) for m in grey], nat.SC_MASK_I32, lambda m: m),
             ("u8 inverted", [np.where(m != 0, 255, rng.integers(0, 255, m.shape)).astype(np.uint8) for m in grey],
              nat.SC_MASK_U8_INV, np.invert),
             ("bool inverted", [m == 0 for m in grey], nat.SC_MASK_BOOL_INV, np.invert)]
    for tag, masks, code, conv in forms:
        want = oracle_c.carve(shape, origin, vs, [(K, R, t, conv(m)) for (K, R, t, _), m in zip(views, masks)], nthreads=4)
        for vpl in (0, 1, 5):
            e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
            e.set_option(nat.SC_OPT_HOST_PACK, host_pack)
            e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
            for (K, R, t, _), m in zip(views, masks):
                e.process_view(K, R, t, m.view(np.uint8) if m.dtype == np.bool_ else m, code)
            assert np.array_equal(e.get_values(), want), (tag, vpl, host_pack, histogram3(want))
            # a second batch over the stored volume, after a clear in between batches of the arena's other half
            for (K, R, t, _), m in zip(views, masks):
                e.process_view(K, R, t, m.view(np.uint8) if m.dtype == np.bool_ else m, code)
            assert np.array_equal(e.get_values(), want), (tag, vpl, host_pack, "second batch")
            e.clear()
            for (K, R, t, _), m in zip(views[:3], masks[:3]):
                e.process_view(K, R, t, m.view(np.uint8) if m.dtype == np.bool_ else m, code)
            e.clear()  # pending host bits dropped
            for (K, R, t, _), m in zip(views, masks):
                e.process_view(K, R, t, m.view(np.uint8) if m.dtype == np.bool_ else m, code)
            assert np.array_equal(e.get_values(), want), (tag, vpl, host_pack, "after clears")
            e.close()


def test_host_mask_arena_grows_with_the_batch(gpu_device):
    """90 masks of 1440 x 1080 are 17.5 MB of bits: more than the arena's first 16 MB, in one batch."""
    shape, origin, vs, views = scene(16, 6, "plant")
    many = [views[q % len(views)] for q in range(90)]
    want = oracle_c.carve(shape, origin, vs, views, nthreads=4)  # a view applied again changes nothing
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    for _ in range(2):
        e.clear()
        for K, R, t, m in many:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want)
    e.close()


class _PngFile:
    """A fileset entry that hands out its bytes like plantdb's ``File`` (``read_raw``) and its pixels (``array``)."""

    def __init__(self, fid, raw, array, cam):
        self.id, self._raw, self.array, self._md = fid, raw, array, {"colmap_camera": cam, "channel": None}

    def read_raw(self):
        return self._raw

    def get_metadata(self, key, default=None):
        return self._md.get(key, default)


def _png(arr, **kw):
    import io
    from PIL import Image
    bio = io.BytesIO()
    Image.fromarray(arr).save(bio, format="PNG", **kw)
    return bio.getvalue()


@pytest.mark.parametrize("w,h", [(160, 96), (101, 67), (1440, 1080)])
@pytest.mark.parametrize("invert", [False, True])
def test_fileset_of_png_masks_decoded_inside_the_library(gpu_device, w, h, invert):
    """``process_fileset`` over files that hand out 8-bit greyscale PNG bytes: one call decodes them on the library's
    threads and reduces each mask to bits as it comes out of the decoder (``sc_process_png_views``) -- the same
    volume as the oracle on the decoded pixels (inverted like cl.py:300-301 where asked), grey levels included; a
    fileset with a file the decoder refuses (RGB) falls back to the decode-ahead loop and gives the same volume."""
    from plant3dvision_amd.scenes import camera_dict
    n = 20 if w < 1000 else 28
    shape, origin, vs, views = scene(n, 7, "plant", width=w, height=h, fx=0.8 * w, fy=0.8 * w, cx=w / 2.0, cy=h / 2.0)
    rng = np.random.default_rng(w + h)
    grey = [np.where(m != 0, rng.integers(1, 256, m.shape), 0).astype(np.uint8) for _, _, _, m in views]
    want = oracle_c.carve(shape, origin, vs, [(K, R, t, np.invert(m) if invert else m) for (K, R, t, _), m in zip(views, grey)], nthreads=4)
    files = [_PngFile(f"im{q}", _png(m, compress_level=int(q % 3) * 4), m, camera_dict(K, R, t))
             for q, ((K, R, t, _), m) in enumerate(zip(views, grey))]
    bp = Backprojection(shape, origin, vs)
    calls = []
    real = bp._engine.process_png_views
    bp._engine.process_png_views = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    got = bp.process_fileset(files, "colmap_camera", invert=invert)
    assert calls and np.array_equal(got, want), (w, h, invert, histogram3(want))
    # one file the decoder does not take: the whole fileset goes through the usual loop, same result
    from PIL import Image
    import io
    bio = io.BytesIO()
    Image.fromarray(grey[2]).convert("RGB").save(bio, format="PNG")
    files[2] = _PngFile("im2", bio.getvalue(), grey[2], files[2]._md["colmap_camera"])
    bp.clear()
    got = bp.process_fileset(files, "colmap_camera", invert=invert)
    assert np.array_equal(got, want), "fallback"
    # a damaged file: refused by the call, nothing enqueued, the loop reads the pixels another way
    bad = bytearray(files[4]._raw); bad[len(bad) // 2] ^= 0xff
    files[4] = _PngFile("im4", bytes(bad), grey[4], files[4]._md["colmap_camera"])
    files[2] = _PngFile("im2", _png(grey[2]), grey[2], files[2]._md["colmap_camera"])
    bp.clear()
    got = bp.process_fileset(files, "colmap_camera", invert=invert)
    assert np.array_equal(got, want), "damaged file"
    bp.close()


@pytest.mark.parametrize("w,h", [(160, 96), (33, 17), (1440, 1080)])
def test_invert_folded_into_device_packing(gpu_device, w, h):
    """``process_fileset(invert=True)``: uint8 / bool masks are inverted by the pack kernels
    (fast 16-px path and general path); other dtypes on the host.  All must equal the
    reference's ``np.invert`` on the raw dtype (cl.py:300-301)."""
    n = 20 if w < 1000 else 32
    shape, origin, vs, views = scene(n, 5, "noise", width=w, height=h, fx=0.8 * w, fy=0.8 * w,
                                     cx=w / 2.0, cy=h / 2.0)
    grey = [(K, R, t, np.where(m != 0, 255 - (np.arange(m.size).reshape(m.shape) % 3), 0).astype(np.uint8))
            for K, R, t, m in views]  # values 0, 253, 254, 255: only 255 becomes background
    for vv, tag in ((grey, "u8"), ([(K, R, t, m != 0) for K, R, t, m in views], "bool"),
                    ([(K, R, t, m.astype(np.int32)) for K, R, t, m in grey], "i32")):
        want = oracle_c.carve(shape, origin, vs, [(K, R, t, np.invert(m)) for K, R, t, m in vv])
        bp = Backprojection(shape, origin, vs)
        got = bp.process_fileset(files_from_views(vv, "colmap_camera"), "colmap_camera", invert=True)
        assert np.array_equal(got, want), tag
        bp.close()


def test_decode_workers_do_not_change_results(gpu_device):
    shape, origin, vs, views = scene(24, 9, "plant")
    rng = np.random.default_rng(2)
    fviews = [(K, R, t, (rng.random(m.shape, dtype=np.float32))) for K, R, t, m in views]
    files = files_from_views(fviews, "colmap_camera")
    outs = []
    for workers in (1, 4):
        bp = Backprojection(shape, origin, vs, type="averaging", decode_workers=workers)
        outs.append(bp.process_fileset(files, "colmap_camera").copy())
        bp.close()
    assert np.array_equal(outs[0], outs[1])  # float sum keeps file order
    assert np.array_equal(outs[0], oracle_c.average(shape, origin, vs, fviews))


def test_odd_image_sizes(gpu_device):
    for (w, h) in ((33, 17), (64, 64), (65, 31), (1, 1), (129, 3)):
        shape, origin, vs, views = scene(20, 4, "noise", width=w, height=h, fx=0.8 * w, fy=0.8 * w,
                                         cx=w / 2.0, cy=h / 2.0)
        want = oracle_c.carve(shape, origin, vs, views)
        assert np.array_equal(hip_carve(shape, origin, vs, views), want), (w, h)


def test_clear_and_reuse(gpu_device):
    shape, origin, vs, views = scene(24, 5, "plant")
    bp = Backprojection(shape, origin, vs)
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    first = bp.get_values().copy()
    # more views after a read-back continue from the stored state (non-fresh path)
    inv = [(K, R, t, np.invert(m)) for K, R, t, m in views[:2]]
    for K, R, t, m in inv:
        bp.process_view(K, R, t, m)
    second = bp.get_values().copy()
    assert np.array_equal(second, oracle_c.carve(shape, origin, vs, list(views) + inv))
    bp.clear()
    assert (bp.get_values() == 0).all()
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    assert np.array_equal(bp.get_values(), first)


def test_arrays_returned_by_get_values_survive_clear(gpu_device):
    """cl.py:229-232,307-311: get_values fills and returns values_h (repeated calls alias it),
    clear() rebinds values_h to a new default-valued array, so an array handed out before a
    clear keeps its contents."""
    shape, origin, vs, views = scene(40, 6, "plant")
    want = oracle_c.carve(shape, origin, vs, views)
    bp = Backprojection(shape, origin, vs, default_value=3)
    assert (bp.values_h == 3).all() and bp.values_h.shape == tuple(shape)  # default * ones, cl.py:173
    bp.close()
    bp = Backprojection(shape, origin, vs)
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    held = bp.get_values()
    assert np.array_equal(held, want)
    assert np.shares_memory(held, bp.get_values())   # same values_h, like the reference
    bp.clear()
    assert (bp.values_h == 0).all()
    inv = [(K, R, t, np.invert(m)) for K, R, t, m in views]
    for K, R, t, m in inv:
        bp.process_view(K, R, t, m)
    other = bp.get_values()
    assert not np.shares_memory(held, other)
    assert np.array_equal(held, want)                # untouched by the second read-back
    assert np.array_equal(other, oracle_c.carve(shape, origin, vs, inv))
    bp.close()


def test_read_back_into_page_locked_memory(gpu_device):
    """sc_host_alloc / pinned_empty: an opt-in destination for Engine.get_values."""
    shape, origin, vs, views = scene(40, 6, "plant")
    want = oracle_c.carve(shape, origin, vs, views)
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    out = nat.pinned_empty(shape, np.int32)
    for K, R, t, m in views:
        e.process_view(K, R, t, m, nat.SC_MASK_U8)
    assert e.get_values(out) is out and np.array_equal(out, want)
    e.close()
    assert np.array_equal(out, want)  # the buffer outlives the engine
    del out


@pytest.mark.parametrize("vpl", [0, 1, 2])
def test_average_matches_oracle_bitwise(gpu_device, vpl):
    shape, origin, vs, views = scene((20, 17, 22), 6, "noise", width=160, height=120, fx=130.0,
                                     fy=130.0, cx=80.0, cy=60.0)
    rng = np.random.default_rng(5)
    fviews = [(K, R, t, (rng.random(m.shape, dtype=np.float32) * 4 - 2)) for K, R, t, m in views]
    want = oracle_c.average(shape, origin, vs, fviews, default_value=0.25)
    got = hip_average(shape, origin, vs, fviews, default_value=0.25, views_per_launch=vpl)
    assert got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))  # tolerance: 0 ulp


def test_average_log_path_uint8_masks(gpu_device):
    shape, origin, vs, views = scene(16, 4, "plant")
    bp = Backprojection(shape, origin, vs, type="averaging", log=True)
    vol = bp.process_fileset(files_from_views(views, "colmap_camera"), "colmap_camera")
    fviews = [(K, R, t, np.log(EPS + img_as_float32(m))) for K, R, t, m in views]
    assert np.array_equal(vol, oracle_c.average(shape, origin, vs, fviews))


@pytest.mark.parametrize("w,h", [(160, 96), (37, 21), (896, 896)])
@pytest.mark.parametrize("log", [False, True])
def test_average_uint8_masks_table_path(gpu_device, w, h, log):
    """uint8 masks in averaging mode travel as bytes + a 256-entry table (SC_MASK_U8_LUT); the
    result must equal the reference's full-array conversion (img_as_float32, log) bit for bit,
    for tile-aligned and ragged image sizes, fused and per-view."""
    n = 20 if w < 800 else 40
    shape, origin, vs, views = scene(n, 5, "noise", width=w, height=h, fx=0.8 * w, fy=0.8 * w,
                                     cx=w / 2.0, cy=h / 2.0)
    rng = np.random.default_rng(w + h)
    grey = [(K, R, t, rng.integers(0, 256, m.shape, dtype=np.uint8)) for K, R, t, m in views]
    conv = (lambda m: np.log(EPS + img_as_float32(m))) if log else img_as_float32
    with np.errstate(divide="ignore"):
        want = oracle_c.average(shape, origin, vs, [(K, R, t, conv(m)) for K, R, t, m in grey])
    for vpl in (0, 1):
        bp = Backprojection(shape, origin, vs, type="averaging", log=log, views_per_launch=vpl)
        for K, R, t, m in grey:
            bp.process_view(K, R, t, m)
        assert np.array_equal(bp.get_values().view(np.uint32), want.view(np.uint32)), (vpl,)
        bp.close()


@pytest.mark.parametrize("shape,kw", [
    ((6, 32, 128), dict()),
    ((5, 37, 131), dict(radius_factor=1.2)),                                     # bricks stick out; nz % 4 != 0
    ((4, 16, 64), dict(radius_factor=0.4)),                                      # cameras among the bricks
    ((3, 48, 64), dict(width=208, height=96, fx=150.0, fy=150.0, cx=100.0, cy=45.0)),
])
@pytest.mark.parametrize("log", [False, True])
@pytest.mark.parametrize("kind", ["plant", "solid", "empty", "grey"])
def test_average_brick_form_on_flat_masks(gpu_device, shape, kw, log, kind):
    """Averaging with uint8 masks + table works on bricks: where a brick's footprint in a view is
    all 0 (or all 255) every voxel adds table[0] (table[255]) without being projected.  Binary
    masks (what Segmentation2D writes), constant masks, and grey ones (nothing is flat) must all
    equal the oracle bit for bit, with the brick form on and off, on a fresh volume and on a
    second batch accumulated over the stored one, through host and device masks."""
    sh, origin, vs, views = scene(shape, 7, "plant" if kind == "grey" else kind, **kw)
    if kind == "grey":
        rng = np.random.default_rng(11)
        views = [(K, R, t, rng.integers(0, 256, m.shape, dtype=np.uint8)) for K, R, t, m in views]
    conv = (lambda m: np.log(EPS + img_as_float32(m))) if log else img_as_float32
    with np.errstate(divide="ignore"):
        fviews = [(K, R, t, conv(m)) for K, R, t, m in views]
        want1 = oracle_c.average(sh, origin, vs, fviews)
        want2 = oracle_c.average(sh, origin, vs, fviews + fviews)
        lut = conv(np.arange(256, dtype=np.uint8))
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    for brick in (1, 0):
        e = nat.Engine(sh, origin, vs, nat.SC_MODE_AVERAGE)
        e.set_option(nat.SC_OPT_AVG_BRICK, brick)
        e.set_lut(lut)
        ptr = e.dev_alloc(stack.nbytes)
        e.dev_upload(ptr, stack)
        e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8_LUT)
        assert np.array_equal(e.get_values().view(np.uint32), want1.view(np.uint32)), (brick, "fresh")
        for Kq, Rq, tq, m in views:  # second batch, host masks, on the stored sums
            e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8_LUT)
        assert np.array_equal(e.get_values().view(np.uint32), want2.view(np.uint32)), (brick, "stored")
        e.clear()
        e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 3)  # batches of 3 views (the last one shorter), in order
        e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8_LUT)
        assert np.array_equal(e.get_values().view(np.uint32), want1.view(np.uint32)), (brick, "batches of 3")
        e.dev_free(ptr)
        e.close()


@pytest.mark.parametrize("L", [2, 3, 4, 5])  # 5: more than one launch takes -- label by label
@pytest.mark.parametrize("shape,kw", [((6, 32, 128), dict()), ((5, 37, 131), dict(width=320, height=208, fx=260.0, fy=260.0, cx=160.0, cy=104.0)),
                                      ((4, 20, 70), dict(width=330, height=207, fx=260.0, fy=260.0, cx=165.0, cy=103.0))])  # odd width: label by label
def test_labels_of_one_scan_in_one_launch(gpu_device, L, shape, kw):
    """``sc_average_labels``: L averaging engines, one set of poses, a voxel projected once per view and every
    label's mask read at that pixel -- each label's volume bit-identical to the oracle's (the same float32
    additions in the same order as its own launch), on fresh volumes and on a second batch over the stored sums;
    binary, grey, all-black and all-white labels together (flat and mixed footprints in one launch)."""
    sh, origin, vs, views = scene(shape, 9, "plant", **kw)
    H, W = views[0][3].shape
    rng = np.random.default_rng(31)
    stacks = []
    for l in range(L):
        kind = l % 5
        if kind == 0:
            st = np.stack([m for _, _, _, m in views])
        elif kind == 1:
            st = rng.integers(0, 256, (len(views), H, W), dtype=np.uint8)
        elif kind == 2:
            st = np.stack([np.invert(m) for _, _, _, m in views])
        elif kind == 3:
            st = np.full((len(views), H, W), 255, dtype=np.uint8)
        else:
            st = np.zeros((len(views), H, W), dtype=np.uint8)
        stacks.append(np.ascontiguousarray(st))
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    for log in (False, True):
        table = averaging_table(log)
        engines = [nat.Engine(sh, origin, vs, nat.SC_MODE_AVERAGE, default_value=float(l)) for l in range(L)]
        ptrs = []
        for e, st in zip(engines, stacks):
            e.set_lut(table)
            ptr = e.dev_alloc(st.nbytes)
            e.dev_upload(ptr, st)
            ptrs.append(ptr)
        for rnd in range(2):
            before = nat.backend().call("sc_average_labels_fused_count")
            nat.average_labels(engines, K, R, t, ptrs, len(views), H, W)
            took_shared_form = nat.backend().call("sc_average_labels_fused_count") - before
            # whole 16-pixel rows and 2 .. 4 labels take the shared launches; an odd width or a fifth label: label by label
            assert took_shared_form == (1 if W % 16 == 0 and L <= 4 else 0), (L, W, took_shared_form)
            for l, (e, st) in enumerate(zip(engines, stacks)):
                fv = [(Kq, Rq, tq, table[st[q]]) for q, (Kq, Rq, tq, _) in enumerate(views)]
                want = oracle_c.average(sh, origin, vs, fv * (rnd + 1), default_value=float(l))
                assert np.array_equal(e.get_values().view(np.uint32), want.view(np.uint32)), (L, l, log, rnd)
        for e, ptr in zip(engines, ptrs):
            e.dev_free(ptr)
            e.close()


def test_average_mixed_uint8_and_float_views(gpu_device):
    shape, origin, vs, views = scene(18, 6, "noise", width=64, height=48, fx=50.0, fy=50.0, cx=32.0, cy=24.0)
    rng = np.random.default_rng(4)
    mixed = [(K, R, t, rng.integers(0, 256, m.shape, dtype=np.uint8) if q % 2 else rng.random(m.shape, dtype=np.float32))
             for q, (K, R, t, m) in enumerate(views)]
    want = oracle_c.average(shape, origin, vs, [(K, R, t, img_as_float32(m)) for K, R, t, m in mixed])
    bp = Backprojection(shape, origin, vs, type="averaging", log=False)
    for K, R, t, m in mixed:
        bp.process_view(K, R, t, m)
    assert np.array_equal(bp.get_values(), want)


def test_process_fileset_with_labels(gpu_device):
    shape, origin, vs, views = scene(16, 3, "plant")
    stem = files_from_views(views, "camera", channel="stem")
    other = files_from_views([(K, R, t, np.invert(m)) for K, R, t, m in views], "camera", channel="bg")
    bp = Backprojection(shape, origin, vs, labels=["stem", "bg"])
    res = bp.process_fileset(stem + other, "camera")
    assert res.dtype == np.float64 and res.shape == (2, *shape)
    assert np.array_equal(res[0], oracle_c.carve(shape, origin, vs, views))
    assert np.array_equal(res[1], oracle_c.carve(shape, origin, vs, [(K, R, t, np.invert(m)) for K, R, t, m in views]))


def test_voxels_run_end_to_end(gpu_device):
    shape, origin, vs, views = scene(24, 6, "plant")
    bbox = {a: [o, o + (n - 1) * vs] for a, o, n in zip("xyz", origin, shape)}
    vol, labels, md = tasks_cl.voxels_run(files_from_views(views), bbox, voxel_size=vs)
    assert labels is None and md["origin"] == origin
    assert np.array_equal(vol, oracle_c.carve(shape, origin, vs, views))


# -- committed golden fixtures ----------------------------------------------------------------
def _vp_views(data, channel, invert=False):
    out = []
    for q in range(data[f"masks_{channel}"].shape[0]):
        m = data[f"masks_{channel}"][q]
        out.append((data[f"K_{channel}"][q].astype(np.float32),
                    data[f"R_{channel}"][q].reshape(9).astype(np.float32),
                    data[f"t_{channel}"][q].astype(np.float32), np.invert(m) if invert else m))
    return out


@pytest.mark.parametrize("tag,vs", [("vs10", 1.0), ("vs05", 0.5)])
def test_virtual_plant_golden(gpu_device, tag, vs):
    data = np.load(os.path.join(GOLDEN, "virtual_plant_inputs.npz"))
    exp = np.load(os.path.join(GOLDEN, "virtual_plant_expected.npz"))
    shape, origin = exp[f"shape_{tag}"].tolist(), exp[f"origin_{tag}"].tolist()
    for vpl in (0, 1):
        got = hip_carve(shape, origin, vs, _vp_views(data, "stem"), views_per_launch=vpl)
        assert np.array_equal(got, exp[f"carve_stem_{tag}"].astype(np.int32))
        got = hip_carve(shape, origin, vs, _vp_views(data, "background", True), views_per_launch=vpl)
        assert np.array_equal(got, exp[f"carve_background_invert_{tag}"].astype(np.int32))


def test_virtual_plant_average_golden(gpu_device):
    data = np.load(os.path.join(GOLDEN, "virtual_plant_inputs.npz"))
    exp = np.load(os.path.join(GOLDEN, "virtual_plant_expected.npz"))
    shape, origin = exp["shape_vs10"].tolist(), exp["origin_vs10"].tolist()
    bp = Backprojection(shape, origin, 1.0, type="averaging", log=False)
    for K, R, t, m in _vp_views(data, "stem"):
        bp.process_view(K, R, t, m)  # uint8 in, img_as_float32 inside (cl.py:205-206)
    assert np.array_equal(bp.get_values(), exp["average_stem_nolog_vs10"])


def test_synthetic_golden(gpu_device):
    exp = np.load(os.path.join(GOLDEN, "synthetic_expected.npz"))
    for key, n, v, kind in (("plant_32_6", 32, 6, "plant"), ("plant_64_12", 64, 12, "plant"),
                            ("noise_48_5", 48, 5, "noise"), ("plant_61x45x113_8", (61, 45, 113), 8, "plant")):
        shape, origin, vs, views = scene(n, v, kind)
        assert np.array_equal(hip_carve(shape, origin, vs, views), exp[key].astype(np.int32)), key


@pytest.mark.parametrize("kind", ["plant", "noise", "solid"])
def test_cfg1_digests(gpu_device, kind):
    """BASELINE cfg 1 (128^3 x 12 views): SHA-256 of the int32 grid vs the committed digest."""
    dig = json.load(open(os.path.join(GOLDEN, "synthetic_digests.json")))[f"{kind}_128_12"]
    shape, origin, vs, views = scene(128, 12, kind)
    for vpl, compact in ((0, 1), (0, 0), (1, 1)):
        got = hip_carve(shape, origin, vs, views, views_per_launch=vpl, compact=compact)
        assert histogram3(got) == dig["hist_m1_0_p1"]
        assert sha256(got) == dig["sha256_int32"]


@pytest.mark.parametrize("vpl", [0, 1, 5])
def test_masks_resident_in_hbm_path(gpu_device, vpl):
    """sc_process_views_device (bench / Masks2D path): masks already in device memory."""
    shape, origin, vs, views = scene(64, 12, "plant")
    want = oracle_c.carve(shape, origin, vs, views, nthreads=4)
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = e.dev_alloc(stack.nbytes)
    e.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    for _ in range(2):  # second round: clear + same device masks again
        e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want)
        e.clear()
    e.dev_free(ptr)
    e.close()


def test_average_masks_resident_in_hbm(gpu_device):
    shape, origin, vs, views = scene(24, 5, "noise", width=96, height=80, fx=80.0, fy=80.0, cx=48.0, cy=40.0)
    rng = np.random.default_rng(9)
    stack = rng.random((len(views), 80, 96), dtype=np.float32)
    fviews = [(K, R, t, stack[q]) for q, (K, R, t, _) in enumerate(views)]
    want = oracle_c.average(shape, origin, vs, fviews)
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE)
    ptr = e.dev_alloc(stack.nbytes)
    e.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    e.process_views_device(K, R, t, ptr, len(views), 80, 96, nat.SC_MASK_F32)
    assert np.array_equal(e.get_values(), want)
    e.dev_free(ptr)
    e.close()


# -- larger sizes: oracle on all host threads, then size-independent properties ----------------
def test_cfg2_256_cubed_36_views_vs_oracle(gpu_device):
    shape, origin, vs, views = scene(256, 36, "plant")
    want = oracle_c.carve(shape, origin, vs, views, nthreads=os.cpu_count() or 8)
    for vpl, compact in ((0, 1), (0, 0), (1, 1)):
        got = hip_carve(shape, origin, vs, views, views_per_launch=vpl, compact=compact)
        assert np.array_equal(got, want)


def test_slabs_concatenate_to_the_whole_grid(gpu_device):
    """Sharding property: X-slabs computed from GLOBAL indices are bit-identical."""
    shape, origin, vs, views = scene((50, 24, 36), 5, "plant")
    want = oracle_c.carve(shape, origin, vs, views)
    parts = []
    for i0, i1 in ((0, 13), (13, 14), (14, 50)):
        e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, slab=(i0, i1))
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        parts.append(e.get_values())
        e.close()
    assert np.array_equal(np.concatenate(parts, axis=0), want)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_cyclic_planes_interleave_to_the_whole_grid(gpu_device, world):
    """Plane-cyclic sharding (the default of ShardedBackprojection): rank r owns planes r, r+W, ...
    computed from GLOBAL indices; interleaved they are the single-engine grid, fused and per view."""
    shape, origin, vs, views = scene((50, 24, 36), 8, "plant")
    want = oracle_c.carve(shape, origin, vs, views)
    for vpl in (0, 1):
        got = np.empty(shape, dtype=np.int32)
        for r in range(world):
            e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, cyclic=(r, world))
            e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
            for K, R, t, m in views:
                e.process_view(K, R, t, m, nat.SC_MASK_U8)
            got[r::world] = e.get_values()
            assert e.slab_shape[0] == len(range(r, shape[0], world))
            e.close()
        assert np.array_equal(got, want), (world, vpl)


@pytest.mark.parametrize("world", [2, 3])
def test_average_on_cyclic_planes_and_slabs(gpu_device, world):
    """The averaging kernels (brick form included) on a rank's planes use GLOBAL x indices too."""
    shape, origin, vs, views = scene((13, 32, 70), 6, "plant")
    lut = img_as_float32(np.arange(256, dtype=np.uint8))
    want = oracle_c.average(shape, origin, vs, [(K, R, t, img_as_float32(m)) for K, R, t, m in views])
    got_c = np.empty(shape, dtype=np.float32)
    parts = []
    for r in range(world):
        e = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE, cyclic=(r, world))
        e.set_lut(lut)
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8_LUT)
        got_c[r::world] = e.get_values()
        e.close()
        i0, i1 = shape[0] * r // world, shape[0] * (r + 1) // world
        e = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE, slab=(i0, i1))
        e.set_lut(lut)
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8_LUT)
        parts.append(e.get_values())
        e.close()
    assert np.array_equal(got_c.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(np.concatenate(parts, axis=0).view(np.uint32), want.view(np.uint32))


def test_full_size_512_cubed_72_views_whole_grid_vs_oracle(gpu_device):
    """BASELINE cfg 3 at full size, the benchmarked scene: EVERY voxel of the fused batch against the oracle
    (backprojection.c:57-84 restated; a fraction of a second on the host's threads) and against the oracle's digest
    committed in tests/golden/synthetic_digests.json; then the properties the domain offers: fused == per-view
    schedule == permuted order (order independence), idempotence (re-applying every view changes nothing)."""
    shape, origin, vs, views = scene(512, 72, "plant")
    fused = hip_carve(shape, origin, vs, views, views_per_launch=0)
    want = oracle_c.carve(shape, origin, vs, views, nthreads=THREADS)
    assert np.array_equal(fused, want), (histogram3(fused), histogram3(want))
    dig = sha256(fused)
    gold = json.load(open(os.path.join(GOLDEN, "synthetic_digests.json")))["plant_512_72"]
    assert dig == gold["sha256_int32"] and histogram3(fused) == gold["hist_m1_0_p1"]
    del want
    assert sha256(hip_carve(shape, origin, vs, views, views_per_launch=0, compact=0)) == dig
    per_view = hip_carve(shape, origin, vs, views, views_per_launch=1, view_order=0)
    assert sha256(per_view) == dig
    del per_view
    rng = np.random.default_rng(7)
    perm = rng.permutation(len(views))
    # idempotence + permutation in one engine
    bp = Backprojection(shape, origin, vs)
    for q in perm:
        bp.process_view(*views[q])
    bp.synchronize()
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    again = bp.get_values()
    assert sha256(again) == dig
    bp.close()
    h = histogram3(fused)
    assert h[2] > 0 and h[0] > 100 * h[2]


# -- seeded random geometry ---------------------------------------------------------------------
def _random_case(rng):
    """Random grid, free (rolled / tilted / off-centre) cameras, random image size and masks."""
    shape = [int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 300))]
    vs = float(rng.choice([0.25, 0.5, 1.0, 1.7]))
    origin = [float(x) for x in rng.uniform(-50, 50, 3)]
    centre = np.array(origin) + (np.array(shape) - 1) * vs / 2
    extent = max(shape) * vs
    W, H = int(rng.integers(8, 400)), int(rng.integers(8, 300))
    views = []
    for _ in range(int(rng.integers(1, 11))):
        # random camera position on a shell around (or inside) the grid, random roll
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        C = centre + d * extent * rng.uniform(0.2, 3.0)
        fwd = centre + rng.normal(size=3) * extent * 0.2 - C
        fwd /= np.linalg.norm(fwd)
        up = rng.normal(size=3)
        right = np.cross(up, fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])
        t = -R @ C
        f = rng.uniform(0.3, 3.0) * W
        K = np.array([f, f * rng.uniform(0.8, 1.25), W * rng.uniform(0.2, 0.8), H * rng.uniform(0.2, 0.8)])
        kind = rng.integers(0, 3)
        if kind == 0:
            m = (rng.random((H, W)) < rng.uniform(0.05, 0.95)).astype(np.uint8) * 255
        elif kind == 1:  # blobs: coherent regions
            yy, xx = np.mgrid[0:H, 0:W]
            m = np.zeros((H, W), dtype=np.uint8)
            for _ in range(4):
                cy, cx, r = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(2, max(W, H) / 2)
                m[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = rng.integers(1, 256)
        else:
            m = np.full((H, W), 255, dtype=np.uint8)
        views.append((K.astype(np.float32), R.reshape(9).astype(np.float32), t.astype(np.float32), m))
    return shape, origin, vs, views


@pytest.mark.parametrize("seed", range(24))
def test_random_geometry_carve_and_average(gpu_device, seed):
    rng = np.random.default_rng(1000 + seed)
    shape, origin, vs, views = _random_case(rng)
    dv = int(rng.choice([0, 0, 0, 1, -1, 5]))
    want = oracle_c.carve(shape, origin, vs, views, dv)
    for vpl, compact in ((0, 1), (0, 0), (1, 1), (2, 1)):
        got = hip_carve(shape, origin, vs, views, default_value=dv, views_per_launch=vpl, compact=compact)
        assert np.array_equal(got, want), (seed, vpl, compact, shape, histogram3(want))
    fviews = [(K, R, t, rng.random(m.shape, dtype=np.float32) - 0.5) for K, R, t, m in views]
    wantf = oracle_c.average(shape, origin, vs, fviews, default_value=0.5)
    for vpl in (0, 1):
        gotf = hip_average(shape, origin, vs, fviews, default_value=0.5, views_per_launch=vpl)
        assert np.array_equal(gotf.view(np.uint32), wantf.view(np.uint32)), (seed, vpl, shape)


# -- device batches: packing at flush, riders beside the dense stage, open FULL candidates -------------
def _device_batch_carve(shape, origin, vs, views, opts=(), default_value=0, preload=None):
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=float(default_value))
    for k, v in opts:
        e.set_option(k, v)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = e.dev_alloc(stack.nbytes)
    e.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    V, H, W = stack.shape
    if preload is not None:  # a stored (non-fresh) volume: some views applied one by one first
        for Kq, Rq, tq, m in preload:
            e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8)
        e.synchronize()
    e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
    out = e.get_values()
    counts = e.fused_counts()
    e.dev_free(ptr)
    e.close()
    return out, counts


@pytest.mark.parametrize("ride", [1, 0])
@pytest.mark.parametrize("kind,shape,v", [("plant", (24, 48, 128), 14), ("solid", (10, 32, 128), 16),
                                          ("dense", (20, 48, 192), 18), ("noise", (6, 16, 64), 12),
                                          ("plant", (9, 37, 131), 13)])
def test_device_batch_packed_at_flush_equals_oracle(gpu_device, ride, kind, shape, v):
    _, origin, vs, views = scene(shape, v, kind)
    want = oracle_c.carve(list(shape), origin, vs, views, nthreads=4)
    for walkers in (1024, 8):
        got, _ = _device_batch_carve(shape, origin, vs, views,
                                     opts=((nat.SC_OPT_PACK_RIDE, ride), (nat.SC_OPT_BRICK_WALKERS, walkers)))
        assert np.array_equal(got, want), (kind, ride, walkers, histogram3(got), histogram3(want))


@pytest.mark.parametrize("odd", [0, 3, 7, 11, 13])
@pytest.mark.parametrize("default_value", [0, 5])
def test_full_candidates_rejected_by_a_late_view(gpu_device, odd, default_value):
    """Every view but one keeps every brick whole (all-foreground pictures); the odd one carves half
    of the volume.  Whether it is among the views packed ahead (and seen by the flags kernel) or among
    the riders (the store blocks' confirmation, then the special kernel's late bricks), the labels are
    the oracle's -- on a fresh volume and on a stored one."""
    shape, origin, vs, views = scene((12, 48, 128), 14, "solid")
    views = [list(v) for v in views]
    half = views[odd][3].copy()
    half[:, : half.shape[1] // 2] = 0
    half[200:260, :] = 0
    views[odd][3] = half
    views = [tuple(v) for v in views]
    want = oracle_c.carve(list(shape), origin, vs, views, default_value, nthreads=4)
    assert 0 < (want == -1).sum() < want.size
    got, _ = _device_batch_carve(shape, origin, vs, views, default_value=default_value)
    assert np.array_equal(got, want), (odd, histogram3(got), histogram3(want))
    # stored volume: two of the views were applied before the batch arrives
    got, _ = _device_batch_carve(shape, origin, vs, views, default_value=default_value, preload=views[4:6])
    assert np.array_equal(got, want), (odd, "stored")
    # a survivor list that overflows while candidates are open
    got, _ = _device_batch_carve(shape, origin, vs, views, default_value=default_value,
                                 opts=((nat.SC_OPT_FULL_BRICKS, 1), (nat.SC_OPT_FLAG_VIEWS, 2)))
    assert np.array_equal(got, want), (odd, "flag_views 2")


def test_device_batch_then_more_views_before_the_flush(gpu_device):
    """A deferred device batch followed by host views (or a second batch) is packed in the order
    given and carved with them."""
    shape, origin, vs, views = scene((16, 32, 64), 12, "plant")
    want = oracle_c.carve(list(shape), origin, vs, views)
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views[:8]]))
    ptr = e.dev_alloc(stack.nbytes)
    e.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    e.process_views_device(K[:8], R[:8], t[:8], ptr, 8, *stack.shape[1:], nat.SC_MASK_U8)
    for Kq, Rq, tq, m in views[8:]:
        e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8)
    assert np.array_equal(e.get_values(), want)
    e.clear()
    e.process_views_device(K[:8], R[:8], t[:8], ptr, 8, *stack.shape[1:], nat.SC_MASK_U8)
    e.clear()  # dropped before it was ever packed
    e.process_views_device(K[:8], R[:8], t[:8], ptr, 8, *stack.shape[1:], nat.SC_MASK_U8)
    e.process_views_device(K[:8], R[:8], t[:8], ptr, 8, *stack.shape[1:], nat.SC_MASK_U8)
    assert np.array_equal(e.get_values(), oracle_c.carve(list(shape), origin, vs, views[:8]))
    e.dev_free(ptr)
    e.close()


def test_device_batches_back_to_back_and_mixed_with_other_launches(gpu_device):
    """Consecutive device batches without a wait in between, batches onto a stored volume, per-view
    launches after a batch, a batch after per-view launches, different masks per batch -- always the
    oracle's labels."""
    shape, origin, vs, views = scene((20, 48, 128), 12, "plant")
    _, _, _, views_b = scene((20, 48, 128), 12, "dense")
    want_a = oracle_c.carve(list(shape), origin, vs, views, nthreads=4)
    want_b = oracle_c.carve(list(shape), origin, vs, views_b, nthreads=4)
    want_ab = oracle_c.carve(list(shape), origin, vs, list(views) + list(views_b), nthreads=4)
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    bufs = []
    for vv in (views, views_b):
        stack = np.ascontiguousarray(np.stack([m for _, _, _, m in vv]))
        ptr = e.dev_alloc(stack.nbytes)
        e.dev_upload(ptr, stack)
        bufs.append((ptr, stack.shape, np.stack([v[0] for v in vv]), np.stack([v[1] for v in vv]), np.stack([v[2] for v in vv])))

    def batch(q):
        ptr, (V, H, W), K, R, t = bufs[q]
        e.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)

    for _ in range(3):  # a, b, a, b ... only the last result is read
        e.clear(); batch(0); e.flush()
        e.clear(); batch(1); e.flush()
    assert np.array_equal(e.get_values(), want_b)
    e.clear(); batch(0); e.flush(); batch(1)  # second batch onto the stored volume of the first
    assert np.array_equal(e.get_values(), want_ab)
    e.clear(); batch(0); e.flush()
    for Kq, Rq, tq, m in views_b:
        e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8)
    assert np.array_equal(e.get_values(), want_ab)
    e.clear()
    for Kq, Rq, tq, m in views_b[:3]:
        e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8)
    e.flush(); batch(0); e.flush(); batch(1)
    assert np.array_equal(e.get_values(), want_ab)
    e.clear(); batch(0)
    assert np.array_equal(e.get_values(), want_a)
    for ptr, *_ in bufs:
        e.dev_free(ptr)
    e.close()


@pytest.mark.parametrize("kw", [
    dict(cx=1.0e4, cy=-7.0e3),                                   # principal point 10^4 px off the picture
    dict(fx=1.0e5, fy=1.0e5),                                    # telephoto: a brick spans thousands of pixels
    dict(fx=1.0e5, fy=1.0e5, radius_factor=40.0),                # ... or a fraction of one
    dict(radius_factor=0.3),                                     # cameras inside the grid: bricks straddle p_z = 0
    dict(radius_factor=0.55, tilt_deg=35.0),
    dict(voxel_size=1.0e-3),                                     # coordinates 375.0xx: float32 steps of 3e-5
    dict(voxel_size=1.0e3),
    dict(voxel_size=1.0e-3, fx=1.0e5, fy=1.0e5),
    dict(width=64, height=48, fx=30.0, fy=30.0, cx=32.0, cy=24.0),  # fisheye-like: most bricks leave the picture
    dict(offaxis=1.0e4),                                         # principal point 10^4 px off AND the grid in the picture:
    dict(offaxis=-3.0e3, radius_factor=0.8),                     # q * f and c nearly cancel in every pixel coordinate
])
@pytest.mark.parametrize("kind", ["plant", "solid", "empty", "dense"])
def test_brick_verdicts_on_adversarial_cameras(gpu_device, kw, kind):
    """The conservative brick verdicts (DESIGN.md 4b gives the bound) must never settle a voxel the
    reference arithmetic would treat differently: device batches and host masks against the oracle
    on rigs that stress every term of the bound."""
    shape = (5, 48, 192)
    kw = dict(kw)
    off = kw.pop("offaxis", None)
    _, origin, vs, views = scene(shape, 10, kind, **kw)
    if off is not None:
        # turn every camera about its own y axis and move the principal point so that the old optical
        # axis still lands on the picture's centre: same pictures in view, a skewed projective camera
        turned = []
        for K, R, t, m in views:
            th = np.arctan2(off, float(K[0]))
            rot = np.array([[np.cos(th), 0, -np.sin(th)], [0, 1, 0], [np.sin(th), 0, np.cos(th)]])
            R2 = (rot @ R.reshape(3, 3).astype(np.float64)).reshape(9).astype(np.float32)
            t2 = (rot @ t.astype(np.float64)).astype(np.float32)
            K2 = K.copy()
            K2[2] = np.float32(float(K[2]) + off)
            turned.append((K2, R2, t2, m))
        views = turned
    want = oracle_c.carve(list(shape), origin, vs, views, nthreads=4)
    if off is not None and kind in ("solid", "dense"):
        assert (want != 0).mean() > 0.03  # part of the grid really is in the picture
    got, _ = _device_batch_carve(shape, origin, vs, views)
    assert np.array_equal(got, want), (kw, kind, "device batch", histogram3(got), histogram3(want))
    # the verdicts below the brick asked of every unit, whatever the heuristics would decide
    got, _ = _device_batch_carve(shape, origin, vs, views, opts=((nat.SC_OPT_UNIT_CULL, 2), (nat.SC_OPT_BULK_MIN, 1), (nat.SC_OPT_BULK_LIVE, 0),
                                                                   (nat.SC_OPT_BULK_FLOOR, 0)))
    assert np.array_equal(got, want), (kw, kind, "device batch, unit verdicts forced", histogram3(got), histogram3(want))
    got = hip_carve(shape, origin, vs, views)
    assert np.array_equal(got, want), (kw, kind, "host masks")
    table = img_as_float32(np.arange(256, dtype=np.uint8))
    wantf = oracle_c.average(list(shape), origin, vs, [(K, R, t, table[m]) for K, R, t, m in views])
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE)
    e.set_lut(table)
    for K, R, t, m in views:
        e.process_view(K, R, t, m, nat.SC_MASK_U8_LUT)
    assert np.array_equal(e.get_values().view(np.uint32), wantf.view(np.uint32)), (kw, kind, "average brick form")
    e.close()


@pytest.mark.parametrize("kind,shape,v", [("plant", (24, 48, 128), 9), ("solid", (6, 32, 128), 7), ("dense", (12, 48, 192), 8),
                                          ("noise", (6, 16, 64), 6), ("empty", (5, 20, 70), 4), ("plant", (7, 37, 131), 7)])
@pytest.mark.parametrize("default_value", [0, 1, -1, 7])
def test_one_view_per_launch_brick_and_streaming_forms(gpu_device, kind, shape, v, default_value):
    """The reference's cadence (one launch per view, cl.py:223-226) in both forms the engine has: brick
    verdicts with dead-brick skipping (default) and the streaming kernel (SC_OPT_VIEW_BRICK 0); mixed
    with fused batches and clears in between, the labels are always the oracle's."""
    _, origin, vs, views = scene(shape, v, kind)
    want = oracle_c.carve(list(shape), origin, vs, views, default_value, nthreads=4)
    for vb in (1, 0):
        e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=float(default_value))
        e.set_option(nat.SC_OPT_VIEW_BRICK, vb)
        e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 1)
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), (kind, vb, "fresh")
        for K, R, t, m in views[::-1]:  # again, on the stored volume, other order: idempotent
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), (kind, vb, "stored")
        e.clear()  # dead bricks are forgotten with the labels
        half = len(views) // 2
        for K, R, t, m in views[:half]:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 0)  # the rest as one fused batch on the stored volume
        for K, R, t, m in views[half:]:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        e.flush()
        e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 1)
        e.process_view(*views[0][:3], views[0][3], nat.SC_MASK_U8)  # and one more single view after it
        assert np.array_equal(e.get_values(), want), (kind, vb, "mixed")
        e.close()


@pytest.mark.parametrize("shape,kw", [((6, 32, 128), dict()), ((5, 37, 131), dict(width=330, height=207, fx=260.0, fy=260.0, cx=165.0, cy=103.0)),
                                      ((4, 20, 70), dict(radius_factor=0.6))])
@pytest.mark.parametrize("kind", ["plant", "solid", "grey"])
@pytest.mark.parametrize("log", [False, True])
def test_average_float32_masks_tiled_and_brick_form(gpu_device, shape, kw, kind, log):
    """float32 masks (what cl.py:205-215 hands the kernel): re-laid in 8x4 tiles, flat footprints add
    their one value without projecting -- bit-identical to the oracle's view-ordered float sum, in every
    form the engine has (tiled + bricks, tiled linear, row-major), fused, in batches of 3 and per view,
    through host masks and a device batch."""
    _, origin, vs, views = scene(shape, 7, "plant" if kind == "grey" else kind, **kw)
    rng = np.random.default_rng(4)
    fviews = []
    for K, R, t, m in views:
        if kind == "grey":
            f = rng.random(m.shape, dtype=np.float32)
        else:
            f = img_as_float32(m)
            if log:
                f = np.log(EPS + f).astype(np.float32)  # two values: flat nearly everywhere
        fviews.append((K, R, t, np.ascontiguousarray(f, dtype=np.float32)))
    want = oracle_c.average(list(shape), origin, vs, fviews, default_value=0.25)
    stack = np.ascontiguousarray(np.stack([f for _, _, _, f in fviews]))
    K = np.stack([v[0] for v in fviews]); R = np.stack([v[1] for v in fviews]); t = np.stack([v[2] for v in fviews])
    for tile, brick in ((1, 1), (1, 0), (0, 1)):
        for vpl in (0, 3, 1):
            e = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE, default_value=0.25)
            e.set_option(nat.SC_OPT_AVG_TILE_F32, tile)
            e.set_option(nat.SC_OPT_AVG_BRICK, brick)
            e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
            for Kq, Rq, tq, f in fviews:
                e.process_view(Kq, Rq, tq, f, nat.SC_MASK_F32)
            got = e.get_values()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (kind, log, tile, brick, vpl, "host masks")
            e.clear()
            ptr = e.dev_alloc(stack.nbytes)
            e.dev_upload(ptr, stack)
            e.process_views_device(K, R, t, ptr, *stack.shape, nat.SC_MASK_F32)
            got = e.get_values()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (kind, log, tile, brick, vpl, "device batch")
            e.dev_free(ptr)
            e.close()


def test_int8_read_back_of_carve_labels(gpu_device):
    """sc_get_values_i8: the labels as bytes (a quarter of the PCIe traffic); the class widens them back
    to the int32 array of cl.py:229-232 for volumes large enough to pay."""
    shape, origin, vs, views = scene((40, 33, 70), 6, "plant")
    for dv in (0, 1, -1, 7, -128, 127):
        want = oracle_c.carve(list(shape), origin, vs, views, dv)
        e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=float(dv))
        assert np.array_equal(e.get_values_i8(), np.full(shape, dv, dtype=np.int8))  # before any view
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        got = e.get_values_i8()
        assert got.dtype == np.int8 and np.array_equal(got.astype(np.int32), want), dv
        assert np.array_equal(e.get_values(), want)
        e.close()
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=300.0)
    with pytest.raises(nat.SpaceCarveError, match="does not fit int8"):
        e.get_values_i8()
    e.close()
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE)
    with pytest.raises(nat.SpaceCarveError):
        e.get_values_i8()
    e.close()
    # the class: a volume of 2^24+ voxels whose default value two bits cannot hold goes through the int8 path and
    # comes back int32 (-1 / 0 / 1 take the 2-bit wire: test_read_back_over_the_two_bit_wire)
    big, origin, vs, views = scene((64, 512, 512), 8, "plant")
    want = oracle_c.carve(list(big), origin, vs, views, 7, nthreads=8)
    bp = Backprojection(big, origin, vs, default_value=7)
    for K, R, t, m in views:
        bp.process_view(K, R, t, m)
    got = bp.get_values()
    assert got.dtype == np.int32 and np.array_equal(got, want) and isinstance(bp._narrow_h, np.ndarray)
    bp.clear()
    assert (bp.get_values() == 7).all() and np.array_equal(got, want)  # the array handed out kept its contents
    bp.close()


@pytest.mark.parametrize("width,height", [(496, 37), (1008, 75), (2032, 64), (2048, 33), (528, 300), (16, 40), (48, 97)])
@pytest.mark.parametrize("invert", [False, True])
def test_masks_packed_in_bands(gpu_device, width, height, invert):
    """The band form of the pack kernel (a block per tile row of a view, 16-pixel tasks in row-major order): widths
    with an odd number of 16-pixel chunks per row (the last tile half empty), heights that are not multiples of 32,
    the narrowest and the widest pictures it takes, plain and inverted masks, packed ahead and by the riders of
    the dense stage; bands forced for the narrow ones (SC_OPT_PACK_ROWS 3).  Labels against the oracle: a wrong
    bit, tile byte or cell map shows as a wrong voxel."""
    shape = (6, 40, 130)
    _, origin, vs, views = scene(shape, 12, "plant", width=width, height=height, fx=0.9 * width, fy=0.9 * width,
                                 cx=0.5 * width, cy=0.5 * height)
    rng = np.random.default_rng(width * 7 + height)
    views = [(K, R, t, np.where(rng.random(m.shape) < 0.02, 255 - m, m).astype(np.uint8)) for K, R, t, m in views]
    want = oracle_c.carve(list(shape), origin, vs, [(K, R, t, (255 - m) if invert else m) for K, R, t, m in views], nthreads=4)
    for ride in (1, 0):
        e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
        for k, v in ((nat.SC_OPT_PACK_ROWS, 3), (nat.SC_OPT_PACK_RIDE, ride), (nat.SC_OPT_UNIT_CULL, 2), (nat.SC_OPT_BULK_MIN, 1), (nat.SC_OPT_BULK_LIVE, 0),
                     (nat.SC_OPT_BULK_FLOOR, 0)):
            e.set_option(k, v)
        stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
        ptr = e.dev_alloc(stack.nbytes)
        e.dev_upload(ptr, stack)
        K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
        e.process_views_device(K, R, t, ptr, len(views), height, width, nat.SC_MASK_U8_INV if invert else nat.SC_MASK_U8)
        got = e.get_values()
        assert np.array_equal(got, want), (width, height, invert, ride, histogram3(got), histogram3(want))
        # host masks: packed one by one as they arrive
        e.clear()
        for Kq, Rq, tq, m in views:
            e.process_view(Kq, Rq, tq, m, nat.SC_MASK_U8_INV if invert else nat.SC_MASK_U8)
        assert np.array_equal(e.get_values(), want), (width, height, invert, "host masks")
        e.dev_free(ptr)
        e.close()


def test_read_back_over_the_two_bit_wire(gpu_device):
    """sc_get_values_wire2: three-state labels cross PCIe at 2 bits each in pieces and are widened to the int32
    array of cl.py:229-232 by host threads inside the library -- voxel counts that are not multiples of 16, more
    pieces than threads and fewer, every default value two bits hold; other default values are refused and the
    class takes the int8 route for them."""
    for shape, nviews in (((40, 33, 70), 6), ((3, 5, 7), 3), ((70, 256, 260), 5)):
        shape, origin, vs, views = scene(shape, nviews, "plant")
        n = int(np.prod(shape))
        for dv in (0, 1, -1):
            want = oracle_c.carve(list(shape), origin, vs, views, dv, nthreads=4)
            e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=float(dv))
            out = np.full(n, 99, dtype=np.int32)
            staging = np.empty((n + 15) // 16 * 4 + 16, dtype=np.uint8)
            e.get_values_wire2(out, staging, threads=3)  # before any view
            assert (out == dv).all()
            for K, R, t, m in views:
                e.process_view(K, R, t, m, nat.SC_MASK_U8)
            for threads in (1, 5, 16):
                out[:] = 99
                e.get_values_wire2(out, staging, threads=threads)
                assert np.array_equal(out.reshape(shape), want), (shape, dv, threads)
            e.get_values_wire2(out)  # the round-3 staging argument is optional now (the engine keeps a page-locked one)
            assert np.array_equal(out.reshape(shape), want), (shape, dv, "no staging argument")
            # a destination that is not 32-byte aligned
            odd = np.full(n + 3, 99, dtype=np.int32)[3:]
            e.get_values_wire2(odd)
            assert np.array_equal(odd.reshape(shape), want), (shape, dv, "unaligned destination")
            e.close()
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=7.0)
    with pytest.raises(nat.SpaceCarveError, match="two bits cannot hold"):
        e.get_values_wire2(np.empty(n, dtype=np.int32), np.empty(n, dtype=np.uint8))
    e.close()
    # the class: 2^24+ voxels and a default value of -1 / 0 / 1 take this route, 7 the int8 one
    big, origin, vs, views = scene((64, 512, 512), 6, "plant")
    for dv in (-1, 7):
        want = oracle_c.carve(list(big), origin, vs, views, dv, nthreads=8)
        bp = Backprojection(big, origin, vs, default_value=dv)
        for K, R, t, m in views:
            bp.process_view(K, R, t, m)
        got = bp.get_values()
        assert got.dtype == np.int32 and np.array_equal(got, want), dv
        bp.close()


def test_read_back_in_pieces_with_a_host_function_beside_the_copies(gpu_device):
    """Engine.get_values_staged / get_values_pipelined (round 5): the volume crosses PCIe in pieces -- through a
    page-locked ring, or straight into the destination -- and a host function takes every piece on host threads while
    the next ones cross.  Same values as get_values + the function afterwards: float32 sums and int32 labels, a padded
    grid (nz % 64 != 0: the device snapshot without the padding), pieces smaller and larger than the volume, a
    destination of another dtype (the reference's float64 label array, cl.py:254), several volumes on one pool."""
    from concurrent.futures import ThreadPoolExecutor
    shape, origin, vs, views = scene((37, 48, 70), 6, "plant")  # nz = 70: rows padded on the device
    fviews = [(K, R, t, img_as_float32(m)) for K, R, t, m in views]
    want_f = oracle_c.average(list(shape), origin, vs, fviews)
    want_i = oracle_c.carve(list(shape), origin, vs, views, nthreads=4)
    ea = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE)
    for K, R, t, m in fviews:
        ea.process_view(K, R, t, m, nat.SC_MASK_F32)
    ec = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    for K, R, t, m in views:
        ec.process_view(K, R, t, m, nat.SC_MASK_U8)

    def exp_from(src, dst):
        np.exp(src, out=dst)

    def exp_in_place(piece):
        np.exp(piece, out=piece)

    for piece_bytes in (4096, 1 << 16, 1 << 26):
        out = np.full(shape, 7, dtype=np.float32)
        assert ea.get_values_staged(out, exp_from, piece_bytes=piece_bytes) is out
        assert np.array_equal(out, np.exp(want_f)), piece_bytes
        out = np.full(shape, 7, dtype=np.float32)
        ea.get_values_pipelined(out, exp_in_place, piece_bytes=piece_bytes)
        assert np.array_equal(out, np.exp(want_f)), piece_bytes
    # another dtype at the destination: float64 <- float32 and float64 <- int32, widened by the function
    out64 = np.zeros((2, *shape))  # float64, like result = np.zeros((len(labels), *shape)) (cl.py:250)
    with ThreadPoolExecutor(max_workers=3) as pool:
        futs = ea.get_values_staged(out64[0], lambda src, dst: np.copyto(dst, src), piece_bytes=1 << 15, pool=pool)
        futs += ec.get_values_staged(out64[1], lambda src, dst: np.copyto(dst, src), piece_bytes=1 << 15, pool=pool)
        for f in futs:
            f.result()
    assert np.array_equal(out64[0], want_f.astype(np.float64)) and np.array_equal(out64[1], want_i.astype(np.float64))
    with pytest.raises(ValueError):
        ea.get_values_staged(np.zeros(5, dtype=np.float32), exp_from)
    ea.close()
    ec.close()


@pytest.mark.parametrize("default_value", [0, 1, -1, 5])
@pytest.mark.parametrize("kind", ["plant", "solid", "dense"])
def test_bricks_no_view_sees_keep_their_labels(gpu_device, kind, default_value):
    """A grid much larger than what the cameras see (narrow pictures, a ring close to the object): most of
    its bricks lie outside every picture.  The verdict for such a brick and view is OUTSIDE (the view does
    nothing to it); kept by every view, seen by none: UNTOUCHED, its labels stay `default_value` -- on a
    fresh volume, on a stored one, one launch per view, and with the views that see part of a brick packed
    late (candidates that fail the confirmation)."""
    shape = (12, 96, 384)
    kw = dict(width=96, height=64, fx=260.0, fy=260.0, cx=48.0, cy=32.0, radius_factor=1.1)
    _, origin, vs, views = scene(shape, 14, kind, **kw)
    want = oracle_c.carve(list(shape), origin, vs, views, default_value, nthreads=4)
    h = histogram3(want) if default_value == 0 else None
    if h is not None:
        assert h[1] > 0.3 * want.size, h  # a good part of the grid is seen by no view (label 0 stays)
        assert h[0] > 0 or kind != "plant", h
    for opts in ((), ((nat.SC_OPT_FLAG_VIEWS, 3),), ((nat.SC_OPT_PACK_RIDE, 0),), ((nat.SC_OPT_FULL_BRICKS, 0),),
                 ((nat.SC_OPT_DEFER_STORES, 0),), ((nat.SC_OPT_VIEWS_PER_LAUNCH, 1),), ((nat.SC_OPT_VIEWS_PER_LAUNCH, 5),),
                 ((nat.SC_OPT_VIEWS_PER_LAUNCH, 1), (nat.SC_OPT_VIEW_BRICK, 0)), ((nat.SC_OPT_COMPACT, 0),)):
        got, counts = _device_batch_carve(shape, origin, vs, views, opts=opts, default_value=default_value)
        assert np.array_equal(got, want), (kind, default_value, opts, histogram3(got), histogram3(want))
    # a stored volume: three views one by one first, then the batch
    got, _ = _device_batch_carve(shape, origin, vs, views[3:], default_value=default_value, preload=views[:3])
    assert np.array_equal(got, want), (kind, default_value, "stored")
    got = hip_carve(shape, origin, vs, views, default_value=default_value)
    assert np.array_equal(got, want), (kind, default_value, "host masks")


@pytest.mark.parametrize("shape,kind", [((12, 32, 128), "plant"), ((7, 21, 70), "plant"), ((5, 17, 1), "solid"),
                                        ((6, 33, 129), "dense"), ((9, 16, 64), "empty")])
@pytest.mark.parametrize("default_value", [0, 1, -1])
def test_labels_packed_at_two_bits_and_one(gpu_device, shape, kind, default_value):
    """``sc_get_values_packed``: label & 3 at two bits per voxel (the three states), label == 1 at one bit,
    rows without their padding; bricks a launch found empty are written without being read.  And back:
    ``sc_unpack_labels`` on one rank's buffer gives the labels again (int8 and int32)."""
    from tests.helpers import pack_labels_np
    sh, origin, vs, views = scene(shape, 9, kind)
    want = oracle_c.carve(sh, origin, vs, views, default_value, nthreads=4)
    e = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE, default_value=default_value)
    for rnd in range(2):  # fused batch on a fresh volume, then view by view on the stored one (dead bricks)
        e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, rnd)
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        for bits in (2, 1):
            got = e.get_values_packed(bits)
            assert np.array_equal(got, pack_labels_np(want, bits)), (bits, rnd)
            ptr, nbytes = e.values_packed(bits)
            assert nbytes == nat.packed_bytes(want.size, bits)
            for out_dtype in (np.int8, np.int32):
                out = e.dev_alloc(want.size * np.dtype(out_dtype).itemsize)
                nat.unpack_labels(0, 0, ptr, nbytes, 1, "cyclic", sh, bits, out, np.dtype(out_dtype).itemsize)
                back = np.empty(want.shape, dtype=out_dtype)
                e.dev_download(back, out)
                e.dev_free(out)
                assert np.array_equal(back, want if bits == 2 else (want == 1)), (bits, out_dtype)
    e.close()
    bad = nat.Engine(sh, origin, vs, nat.SC_MODE_CARVE, default_value=7)
    with pytest.raises(nat.SpaceCarveError):
        bad.get_values_packed(2)
    bad.close()


@pytest.mark.parametrize("shape", [(9, 21, 70), (5, 17, 1), (4, 16, 64), (6, 33, 129), (3, 40, 191)])
def test_row_padding_never_shows(gpu_device, shape):
    """The state's rows are padded to multiples of 64 voxels on the device; every way out of the engine
    (int32 and int8 read-back, the dense device pointer, averaging volumes, a stored volume read twice)
    hands over nx * ny * nz elements in the reference's order."""
    _, origin, vs, views = scene(shape, 7, "plant")
    want = oracle_c.carve(list(shape), origin, vs, views, nthreads=2)
    for vpl in (0, 1, 3):
        e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
        e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8)
        got = e.get_values()
        assert got.shape == tuple(shape) and np.array_equal(got, want), (shape, vpl)
        got8 = e.get_values_i8()
        assert got8.dtype == np.int8 and np.array_equal(got8, want), (shape, vpl)
        dense = np.empty(shape, dtype=np.int32)
        e.dev_download(dense, e.values_device_ptr())
        assert np.array_equal(dense, want), (shape, vpl, "device pointer")
        assert np.array_equal(e.get_values(), want)  # again, unchanged
        e.close()
    table = img_as_float32(np.arange(256, dtype=np.uint8))
    wantf = oracle_c.average(list(shape), origin, vs, [(K, R, t, table[m]) for K, R, t, m in views])
    for vpl in (0, 1):
        e = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE)
        e.set_lut(table)
        e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
        for K, R, t, m in views:
            e.process_view(K, R, t, m, nat.SC_MASK_U8_LUT)
        assert np.array_equal(e.get_values().view(np.uint32), wantf.view(np.uint32)), (shape, vpl, "average")
        densef = np.empty(shape, dtype=np.float32)
        e.dev_download(densef, e.values_device_ptr())
        assert np.array_equal(densef.view(np.uint32), wantf.view(np.uint32))
        e.close()
    # before any view: default_value everywhere
    e = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE, default_value=7.0)
    assert (e.get_values() == 7).all() and (e.get_values_i8() == 7).all()
    e.close()


@pytest.mark.parametrize("form", ["u8", "f32"])
@pytest.mark.parametrize("log", [False, True])
def test_average_of_bricks_no_view_sees(gpu_device, form, log):
    """Averaging on a grid much wider than the pictures: a view that does not see a brick at all adds
    nothing to it -- not even +0.0 (a default of -0.0 keeps its sign where no view reaches)."""
    shape = (6, 96, 384)
    kw = dict(width=96, height=64, fx=260.0, fy=260.0, cx=48.0, cy=32.0, radius_factor=1.1)
    _, origin, vs, views = scene(shape, 9, "dense", **kw)
    table = averaging_table(log)
    rng = np.random.default_rng(8)
    for default_value in (0.0, -0.0, 2.5):
        if form == "u8":
            masks = [m if q % 2 else rng.integers(0, 256, m.shape, dtype=np.uint8) for q, (_, _, _, m) in enumerate(views)]
            fviews = [(K, R, t, table[m]) for (K, R, t, _), m in zip(views, masks)]
        else:
            masks = [table[m] if q % 2 else rng.random(m.shape, dtype=np.float32) for q, (_, _, _, m) in enumerate(views)]
            fviews = [(K, R, t, m) for (K, R, t, _), m in zip(views, masks)]
        want = oracle_c.average(list(shape), origin, vs, fviews, default_value)
        assert (want.view(np.uint32) == np.float32(default_value).view(np.uint32)).mean() > 0.3  # unseen voxels
        for vpl in (0, 1, 4):
            e = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE, default_value=default_value)
            e.set_lut(table)
            e.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
            for (K, R, t, _), m in zip(views, masks):
                e.process_view(K, R, t, m, nat.SC_MASK_U8_LUT if form == "u8" else nat.SC_MASK_F32)
            got = e.get_values()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (form, log, default_value, vpl)
            e.close()
