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
