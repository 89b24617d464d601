import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "envlogic.json")) as f:
        return json.load(f)


def pytest_collection_modifyitems(config, items):
    # GPU tests go through the in-tree HIP library: make sure this checkout has it (no-op when __graft_entry__.build()
    # already ran; the product itself never builds on demand and has no fallback)
    if any(it.get_closest_marker("gpu") for it in items) and "not gpu" not in (config.getoption("-m") or ""):
        from balance_robot_mujoco_rl_amd import _lib
        _lib.build()
