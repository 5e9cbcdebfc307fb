import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_visible() -> bool:
    if not os.path.exists("/dev/kfd"):
        return False
    try:   # the HIP runtime's own answer (no torch import: that costs a minute on a fresh box)
        import ctypes
        n = ctypes.c_int(0)
        return ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return True


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing them with KB_E_DEVICE
    (the engine has no CPU fallback by design).  On a GPU box nothing is skipped: a missing libkbengine.so then fails loudly."""
    if _hip_device_visible():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def kb():
    return importlib.import_module("kube-batch_amd")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
