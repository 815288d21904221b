"""GPU: the timing side of the C ABI (SC_OPT_TIME_KERNELS, sc_kernel_stats, sc_span_begin / sc_span_end,
SC_OPT_RESERVE_EVENTS) -- what bench.py's roofline numbers come from.  Timing never changes a label."""
import numpy as np
import pytest

from oracle import oracle_c
from plant3dvision_amd import _native as nat
from tests.helpers import scene

pytestmark = pytest.mark.gpu

KERNELS = {"carve": nat.SC_KERNEL_CARVE, "list": nat.SC_KERNEL_LIST, "pack": nat.SC_KERNEL_PACK,
           "fill": nat.SC_KERNEL_FILL, "flags": nat.SC_KERNEL_FLAGS, "step": nat.SC_KERNEL_STEP}


def _setup(shape=(40, 48, 128), nviews=12):
    shape, origin, vs, views = scene(shape, nviews, "plant")
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    ptr = eng.dev_alloc(stack.nbytes)
    eng.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    want = oracle_c.carve(shape, origin, vs, views)
    return eng, (K, R, t, ptr, *stack.shape, nat.SC_MASK_U8), want


def _batch(eng, call, n=1):
    for _ in range(n):
        eng.clear()
        eng.process_views_device(*call)
        eng.flush()


def test_kernel_stats_modes_count_what_they_say_and_change_nothing(gpu_device):
    eng, call, want = _setup()
    for mode, batches in ((2, 3), (1, 2), (0, 2)):
        eng.set_option(nat.SC_OPT_TIME_KERNELS, mode)
        eng.reset_kernel_stats()
        _batch(eng, call, batches)
        got = eng.get_values()
        assert np.array_equal(got, want), mode
        stats = {k: eng.kernel_stats(v) for k, v in KERNELS.items()}
        if mode == 0:
            assert all(n == 0 and ms == 0.0 for n, ms in stats.values()), stats
        elif mode == 2:  # one event pair per fused batch, nothing else
            assert stats["step"][0] == batches and 0.0 < stats["step"][1] < 1e3, stats
            assert stats["list"][0] == 0 and stats["flags"][0] == 0, stats
        else:  # a pair around every kernel
            assert stats["step"][0] == batches
            assert stats["carve"][0] >= batches and stats["flags"][0] == batches and stats["list"][0] >= batches, stats
            assert stats["pack"][0] >= batches
            per_kernel = sum(stats[k][1] for k in ("carve", "list", "pack", "flags", "fill"))
            assert 0.0 < per_kernel <= stats["step"][1] * 1.5 + 0.5, stats
    # a one-view-per-launch schedule: mode 2 times the carve kernel of every launch, no batch window
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 1)
    eng.set_option(nat.SC_OPT_TIME_KERNELS, 2)
    eng.reset_kernel_stats()
    _batch(eng, call, 1)
    assert np.array_equal(eng.get_values(), want)
    n, ms = eng.kernel_stats(nat.SC_KERNEL_CARVE)
    assert n == call[4] and ms > 0.0
    assert eng.kernel_stats(nat.SC_KERNEL_STEP)[0] == 0
    # clear() with a window open: the window is dropped, the next batch is timed on its own
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 0)
    eng.reset_kernel_stats()
    eng.process_views_device(*call)  # deferred: opens the window
    eng.clear()
    _batch(eng, call, 1)
    assert eng.kernel_stats(nat.SC_KERNEL_STEP)[0] == 1
    assert np.array_equal(eng.get_values(), want)
    eng.dev_free(call[3])
    eng.close()


def test_span_brackets_batches_with_one_event_pair(gpu_device):
    eng, call, want = _setup()
    eng.set_option(nat.SC_OPT_RESERVE_EVENTS, 16)
    _batch(eng, call, 2)
    eng.synchronize()
    eng.span_begin()
    with pytest.raises(nat.SpaceCarveError):
        eng.span_begin()  # one span at a time
    _batch(eng, call, 5)
    ms5 = eng.span_end()
    assert 0.0 < ms5 < 1e3
    with pytest.raises(nat.SpaceCarveError):
        eng.span_end()
    eng.span_begin()
    ms0 = eng.span_end()  # nothing in between
    assert 0.0 <= ms0 < ms5
    assert np.array_equal(eng.get_values(), want)
    for bad in (-1, 65537):
        with pytest.raises(ValueError):
            eng.set_option(nat.SC_OPT_RESERVE_EVENTS, bad)
    eng.dev_free(call[3])
    eng.close()
