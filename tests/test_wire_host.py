"""Host half of the 2-bit read-back (sc_get_values_wire2 widens the pieces that have crossed PCIe with
sc_widen_labels2's loop): no device needed.  Labels -1 / 0 / 1 packed as the device packs them (label & 3, 16 per
32-bit word, voxel v at bits 2 (v % 16) of word v / 16: include/spacecarve.h) must come back as the int32 array
cl.py:229-232 returns, for voxel counts that are not multiples of 16 and any number of threads."""
import numpy as np
import pytest

from plant3dvision_amd import _native as nat
from tests.helpers import pack_labels_np


@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 65536 + 5, 3 * 262144 * 16 + 7])
def test_two_bit_labels_widen_to_int32_on_host_threads(n):
    rng = np.random.default_rng(n)
    labels = rng.integers(-1, 2, size=n).astype(np.int32)
    packed = pack_labels_np(labels, 2)
    assert packed.size == (n + 15) // 16
    for threads in (1, 3, 16, 64):
        out = np.full(n, 99, dtype=np.int32)
        nat.widen_labels2(packed, n, out, threads=threads)
        assert np.array_equal(out, labels), (n, threads)
    # the words may be longer than the labels need (the device buffer is whole 16-byte groups): the tail is not read
    padded = np.concatenate([packed, np.full(3, 0xFFFFFFFF, dtype=np.uint32)])
    assert np.array_equal(nat.widen_labels2(padded, n), labels)


def test_widening_refuses_short_input():
    with pytest.raises(ValueError):
        nat.widen_labels2(np.zeros(1, dtype=np.uint32), 17)
    with pytest.raises(ValueError):
        nat.widen_labels2(np.zeros(2, dtype=np.uint32), 17, out=np.zeros(16, dtype=np.int32))


def _bits_reference(fg):
    """fg: bool [H][W] -> uint32 [H][ceil(W / 32)], pixel u at bit u & 31 of word u >> 5."""
    H, W = fg.shape
    wpr = (W + 31) // 32
    pad = np.zeros((H, wpr * 32), dtype=np.uint64)
    pad[:, :W] = fg
    return (pad.reshape(H, wpr, 32) << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32)


@pytest.mark.parametrize("shape", [(1, 1), (3, 31), (5, 32), (7, 33), (64, 100), (130, 257), (1080, 1440), (33, 1024 + 17)])
def test_host_bit_packing_is_the_carve_test_on_every_pixel(shape):
    """The host half of mask ingest: what crosses PCIe for a carve mask is `pixel != 0` (backprojection.c:79 on the
    int32 cast of cl.py:215), after np.invert where the fileset loop asks for it (cl.py:300-301) -- for grey levels,
    bool bytes, int32 masks with negative values, padded rows, widths that are not multiples of 32."""
    from plant3dvision_amd import _native as nat
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    grey = rng.integers(0, 256, (H, W), dtype=np.uint8)
    grey[rng.random((H, W)) < 0.5] = 0
    assert np.array_equal(nat.hostpack_bits(grey, nat.SC_MASK_U8), _bits_reference(grey != 0))
    assert np.array_equal(nat.hostpack_bits(grey, nat.SC_MASK_U8_INV), _bits_reference(np.invert(grey) != 0))
    b = rng.random((H, W)) < 0.3
    assert np.array_equal(nat.hostpack_bits(b, nat.SC_MASK_U8), _bits_reference(b))
    assert np.array_equal(nat.hostpack_bits(b, nat.SC_MASK_BOOL_INV), _bits_reference(np.invert(b)))
    i32 = rng.integers(-3, 4, (H, W)).astype(np.int32) * rng.integers(0, 2, (H, W)).astype(np.int32) * 70000
    assert np.array_equal(nat.hostpack_bits(i32, nat.SC_MASK_I32), _bits_reference(i32 != 0))
    wide = np.zeros((H, W + 13), dtype=np.uint8)  # a view with padded rows
    wide[:, :W] = grey
    wide[:, W:] = 255
    assert np.array_equal(nat.hostpack_bits(wide[:, :W], nat.SC_MASK_U8), _bits_reference(grey != 0))


def test_widening_at_every_alignment_and_tail():
    from plant3dvision_amd import _native as nat
    rng = np.random.default_rng(5)
    for n in (1, 15, 16, 17, 1000, (1 << 20) + 5):
        lab = rng.integers(-1, 2, n).astype(np.int32)
        words = np.zeros((n + 15) // 16, dtype=np.uint32)
        for i in range(16):
            part = (lab[i::16] & 3).astype(np.uint32)
            words[:part.size] |= part << np.uint32(2 * i)
        buf = np.empty(n + 8, dtype=np.int32)
        for off in (0, 1, 3):  # 32-byte aligned (streaming stores) or not
            out = buf[off:off + n]
            nat.widen_labels2(words, n, out=out)
            assert np.array_equal(out, lab), (n, off)


@pytest.mark.parametrize("partition", ["cyclic", "slab"])
@pytest.mark.parametrize("world,shape", [(1, (5, 4, 8)), (2, (16, 10, 12)), (3, (17, 9, 8)), (3, (11, 8, 16)), (4, (9, 16, 32)), (8, (64, 16, 16))])
def test_ranks_packed_planes_widen_into_one_grid_in_global_order(world, shape, partition):
    """The host end of gather_to_host: every rank's planes at 2 bits per label, rank-major and padded to the largest
    plane count, into the int32 grid of cl.py:229-232 -- even and uneven plane counts, planes of whole packed words
    and planes that start in the middle of one."""
    from plant3dvision_amd import _native as nat
    from plant3dvision_amd.sharded import rank_planes
    rng = np.random.default_rng(world * 100 + shape[0])
    grid = rng.integers(-1, 2, shape).astype(np.int32)
    plane = shape[1] * shape[2]
    pmax = max(len(rank_planes(shape[0], world, r, partition)) for r in range(world))
    rank_bytes = nat.packed_bytes(pmax * plane, 2)
    buf = np.zeros(world * rank_bytes // 4, dtype=np.uint32)
    for r in range(world):
        mine = grid[list(rank_planes(shape[0], world, r, partition))].reshape(-1)
        w = pack_labels_np(mine, 2)
        buf[r * rank_bytes // 4: r * rank_bytes // 4 + w.size] = w
    got = nat.widen_labels2_ranks(buf, rank_bytes, world, partition, shape)
    assert got.dtype == np.int32 and np.array_equal(got, grid)


def test_host_pool_works_in_a_forked_child():
    """A fork()ed child inherits the pool object without its threads (multiprocessing's fork start
    method, luigi workers): the host-only entry points must still finish there -- a fresh pool after the fork."""
    import os
    import signal
    n = 3 * 262144 * 16 + 7
    rng = np.random.default_rng(11)
    labels = rng.integers(-1, 2, size=n).astype(np.int32)
    packed = pack_labels_np(labels, 2)
    big = rng.integers(0, 256, (1080, 1440), dtype=np.uint8)
    want_bits = nat.hostpack_bits(big, nat.SC_MASK_U8)          # the parent's pool has started its threads
    assert np.array_equal(nat.widen_labels2(packed, n, threads=8), labels)
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:  # the child: no pytest machinery from here on
        code = 1
        try:
            signal.alarm(60)  # a hang is a failure, not a stuck test run
            ok = np.array_equal(nat.widen_labels2(packed, n, threads=8), labels)
            ok = ok and np.array_equal(nat.hostpack_bits(big, nat.SC_MASK_U8), want_bits)
            os.write(w, b"1" if ok else b"0")
            code = 0 if ok else 2
        finally:
            os._exit(code)
    os.close(w)
    _, status = os.waitpid(pid, 0)
    assert os.read(r, 1) == b"1" and os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    os.close(r)
    assert np.array_equal(nat.widen_labels2(packed, n, threads=8), labels)  # the parent's pool is untouched


# -- the brick-sparse label form (round 6): host end -------------------------------------------------------------------
def _random_labels(rng, shape, kind):
    nx, ny, nz = shape
    if kind == "carved":      # mostly -1, a blob of 1s, a little 0
        lab = np.full(shape, -1, dtype=np.int32)
        lab[nx // 3:nx // 3 + max(1, nx // 4), ny // 4:ny // 2 + 1, nz // 5:nz // 2 + 1] = 1
        lab[rng.random(shape) < 0.002] = 0
        return lab
    if kind == "solid":
        return np.ones(shape, dtype=np.int32)
    return rng.integers(-1, 2, size=shape).astype(np.int32)  # noise: every brick mixed


@pytest.mark.parametrize("shape,world,partition", [((8, 32, 128), 1, "cyclic"), ((9, 20, 70), 2, "cyclic"),
                                                   ((9, 20, 70), 3, "slab"), ((5, 16, 64), 5, "cyclic"), ((7, 33, 129), 2, "slab")])
@pytest.mark.parametrize("kind", ["carved", "solid", "noise"])
def test_sparse_labels_widen_on_the_host_to_the_grid(shape, world, partition, kind):
    """``sc_widen_sparse_ranks`` (the host end of gather_to_host over the sparse wire) against the labels themselves and
    the NumPy restatement of the format, ragged shapes and uneven ranks included."""
    from plant3dvision_amd.sharded import rank_planes
    from tests.helpers import pack_sparse_np, sparse_header_np, unpack_sparse_np
    rng = np.random.default_rng(hash((shape, world, kind)) % (1 << 32))
    lab = _random_labels(rng, shape, kind)
    nbmax = max(nat.sparse_bricks(len(rank_planes(shape[0], world, r, partition)), shape[1], shape[2]) for r in range(world))
    stride = nat.sparse_rank_bytes(nbmax, nbmax)
    recv = np.zeros(world * stride, dtype=np.uint8)
    mixed = 0
    for r in range(world):
        pl = rank_planes(shape[0], world, r, partition)
        buf = pack_sparse_np(lab[pl.start:pl.stop:pl.step], pl.start, pl.step if len(pl) > 1 else 1, cap=nbmax)
        assert buf.size <= stride and buf.size == nat.sparse_rank_bytes(nat.sparse_bricks(len(pl), shape[1], shape[2]), nbmax)
        recv[r * stride:r * stride + buf.size] = buf
        mixed += sparse_header_np(buf)["nmixed"]
    if kind == "solid":
        assert mixed == 0
    got = nat.widen_sparse_ranks(recv, stride, world, shape)
    assert got.dtype == np.int32 and np.array_equal(got, lab)
    assert np.array_equal(unpack_sparse_np(recv, stride, world, shape), lab)


def test_sparse_labels_host_end_refuses_what_it_cannot_trust():
    from tests.helpers import pack_sparse_np
    rng = np.random.default_rng(3)
    lab = rng.integers(-1, 2, size=(4, 32, 128)).astype(np.int32)
    nb = nat.sparse_bricks(4, 32, 128)
    full = pack_sparse_np(lab, 0, 1, cap=nb)
    assert np.array_equal(nat.widen_sparse_ranks(full, full.size, 1, lab.shape), lab)
    lab2 = rng.integers(-1, 2, size=(8, 32, 128)).astype(np.int32)
    over = pack_sparse_np(lab2, 0, 1, cap=16)  # 32 mixed bricks, 16 slots: nmixed > cap
    with pytest.raises(nat.SpaceCarveError):
        nat.widen_sparse_ranks(over, over.size, 1, lab2.shape)
    bad = full.copy()
    bad[0] ^= 0xff  # not a sparse buffer
    with pytest.raises(ValueError):
        nat.widen_sparse_ranks(bad, bad.size, 1, lab.shape)
    with pytest.raises(ValueError):
        nat.widen_sparse_ranks(full, full.size, 1, (5, 32, 128))  # the ranks do not hold the grid's planes
