"""CPU: the C-ABI library builds, loads, exports every symbol include/spacecarve.h declares,
and fails loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from plant3dvision_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "spacecarve.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z_0-9]+)\s*\(", text)))


def test_library_is_built_and_loads():
    assert os.path.exists(nat.LIB_PATH), "run __graft_entry__.build() first"
    b = nat.backend()
    assert b.call("sc_abi_version") == 1


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_symbols()
    assert len(declared) >= 20
    lib = ctypes.CDLL(nat.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in spacecarve.h but not exported"
    assert sorted(nat.EXPORTED_SYMBOLS) == declared


def test_header_constants_match_binding():
    text = open(os.path.join(ROOT, "include", "spacecarve.h")).read()
    for name in ("SC_MODE_CARVE", "SC_MODE_AVERAGE", "SC_MASK_U8", "SC_MASK_I32", "SC_MASK_F32",
                 "SC_OPT_VIEWS_PER_LAUNCH", "SC_OPT_VIEW_ORDER", "SC_OPT_TIME_KERNELS",
                 "SC_OPT_MAX_PENDING", "SC_KERNEL_CARVE", "SC_KERNEL_AVERAGE", "SC_KERNEL_PACK",
                 "SC_KERNEL_FILL", "SC_ERR_INVALID", "SC_ERR_DEVICE", "SC_ERR_NOMEM"):
        m = re.search(rf"#define {name} \(?(-?\d+)\)?", text)
        assert m, name
        assert int(m.group(1)) == getattr(nat, name), name


def test_no_cpu_fallback_without_device():
    """Without a GPU the product path must raise, never compute."""
    try:
        n = nat.device_count()
    except nat.SpaceCarveError:
        n = 0
    if n > 0:
        pytest.skip("a gfx950 device is present")
    from plant3dvision_amd.cl import Backprojection
    with pytest.raises(nat.SpaceCarveError):
        Backprojection([10, 10, 10], [0.0, 0.0, 0.0], 1.0)


def test_argument_errors_do_not_need_a_device():
    b = nat.backend()
    out = np.zeros(1, dtype=np.uintp)
    origin = np.zeros(3, dtype=np.float32)
    rc = b.call("sc_create", nat.addr(out), 0, 4, 4, nat.addr(origin), 1.0, 0, 0.0, 0)
    assert rc == nat.SC_ERR_INVALID and "shape" in nat.last_error()
    rc = b.call("sc_create", nat.addr(out), 4, 4, 4, nat.addr(origin), 1.0, 9, 0.0, 0)
    assert rc == nat.SC_ERR_INVALID and "mode" in nat.last_error()
    rc = b.call("sc_create_slab", nat.addr(out), 4, 4, 4, 3, 2, nat.addr(origin), 1.0, 0, 0.0, 0)
    assert rc == nat.SC_ERR_INVALID and "slab" in nat.last_error()
    assert b.call("sc_clear", 0) == nat.SC_ERR_INVALID
    assert int(out[0]) == 0
