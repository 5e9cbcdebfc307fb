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


def pytest_generate_tests(metafunc):
    """Every gpu-marked test runs once per commit kernel (kb_device.h: KB_COMMIT_BATCH / KB_COMMIT_RUN): the engine picks one
    per round from the share of dirty-won rows, so both must reproduce the oracle on every input."""
    if metafunc.definition.get_closest_marker("gpu") is not None and "commit_kernel" in metafunc.fixturenames:
        metafunc.parametrize("commit_kernel", ["batch", "run"], indirect=True)


@pytest.fixture(autouse=True)
def commit_kernel(request):
    which = getattr(request, "param", None)
    old = os.environ.get("KB_COMMIT_KERNEL")
    if which:
        os.environ["KB_COMMIT_KERNEL"] = which     # read by kb_engine_create
    yield which
    if which:
        if old is None:
            os.environ.pop("KB_COMMIT_KERNEL", None)
        else:
            os.environ["KB_COMMIT_KERNEL"] = old


@pytest.fixture(scope="session")
def kb():
    return importlib.import_module("kube-batch_amd")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
