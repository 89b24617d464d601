"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares; without a device it fails
loudly (there is no CPU fallback in the product)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("brs.h", "brs_policy.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(brs_[a-z_0-9]+)\s*\(", txt))
    return sorted(names)


def test_header_symbols_exported():
    from balance_robot_mujoco_rl_amd import _lib
    _lib.build()
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/*.h but not exported by libbrs_hip.so"
    assert sorted(_lib.SYMBOLS) == names


def test_sizes_and_bad_args_without_device():
    from balance_robot_mujoco_rl_amd import _lib
    L = _lib.lib()
    nq, nv, no, na = (C.c_int32() for _ in range(4))
    assert L.brs_sizes(1, C.byref(nq), C.byref(nv), C.byref(no), C.byref(na)) == 0
    assert (nq.value, nv.value, no.value, na.value) == (9, 8, 6, 2)
    assert L.brs_sizes(3, C.byref(nq), C.byref(nv), None, None) == 0 and (nq.value, nv.value) == (16, 14)
    assert L.brs_sizes(7, None, None, None, None) == -1
    h = C.c_void_p()
    assert L.brs_create(None, C.byref(h)) == -1
    cfg = _lib.BrsConfig(9, 4, 0, 0, 0, 0, 0, 0, 0.0, 0, 0)
    assert L.brs_create(C.byref(cfg), C.byref(h)) == -1 and b"variant" in L.brs_last_error(None)
    cfg = _lib.BrsConfig(1, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0)
    assert L.brs_create(C.byref(cfg), C.byref(h)) == -1
    cfg = _lib.BrsConfig(1, 8, 0, 6, 0, 0, 0, 0, 0.0, 0, 0)  # NOISE_ON | NOISE_OFF
    assert L.brs_create(C.byref(cfg), C.byref(h)) == -1
    # include/brs_policy.h: argument checks that need no device
    assert L.brs_policy_create(0, None) == -1
    assert L.brs_gae(0, 0, 4, None, None, None, None, None, 0.99, 0.95, None, None, None) == -1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the loud-failure path is for machines without one")
    from balance_robot_mujoco_rl_amd import BatchedSim, BrsError, _lib
    with pytest.raises(BrsError):
        BatchedSim("Env01-v2", 4)
    L = _lib.lib()
    h = C.c_void_p()
    cfg = _lib.BrsConfig(1, 8, 0, 0, 0, 0, 0, 0, 0.0, 0, 0)
    assert L.brs_create(C.byref(cfg), C.byref(h)) == -2  # BRS_ERR_HIP
    assert b"no CPU fallback" in L.brs_last_error(None)


def test_product_never_imports_oracle():
    """the product package must not reference oracle/ or the host test build"""
    pkg = os.path.join(ROOT, "balance_robot_mujoco_rl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                t = open(os.path.join(dp, f)).read()
                assert "import oracle" not in t and "from oracle" not in t and "libbrs_oracle" not in t, f
                assert "libbrs_hostsim" not in t, f


def test_tools_and_entry_points_compile():
    """every script under tools/ and the driver entry points are at least syntactically valid Python (most of them only
    run on a GPU box)"""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "tools", "*.py")) + [os.path.join(root, f) for f in ("bench.py", "__graft_entry__.py")]
    assert len(files) >= 8
    for f in files:
        compile(open(f).read(), f, "exec")
