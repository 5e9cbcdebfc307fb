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
    """Every gpu-marked test that asks for the `commit_kernel` fixture runs once per commit kernel (kb_device.h: KB_COMMIT_RUN, the plain
    row-by-row restatement, and KB_COMMIT_SELECT, the one the engine runs): each must reproduce the oracle on every input."""
    if metafunc.definition.get_closest_marker("gpu") is not None and "commit_kernel" in metafunc.fixturenames:
        metafunc.parametrize("commit_kernel", ["run", "select"], indirect=True)


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


# ---- the engine's envelope, pinned ---------------------------------------------------------------------------------------------
# The differential suites skip a snapshot when the oracle says the reference would panic on it, or when the engine answers
# KB_E_UNSUPPORTED / KB_E_INVALID (the documented envelope).  A regression that WIDENS the unsupported set would
# show up as more skips, not as a failure — so the set of skipping cases is committed (tests/golden/expected_skips.json, produced by
# KB_RECORD_SKIPS=<file> python -m pytest tests/test_emu_engine_cpu.py; the envelope is host logic, identical on the emulated and the
# real device) and any OTHER skip of a differential case is turned into a failure, on the GPU box and on the emulated device alike.
_ENVELOPE_MODULES = {"test_gpu_parity", "test_gpu_fuzz", "test_gpu_adversarial", "test_gpu_interpod", "test_gpu_preempt",
                     "test_framework_actions", "test_gpu_regressions", "test_gpu_fullsize", "test_gpu_sharded", "test_gpu_wideports"}
_expected_skips = None


def _skip_key(item):
    fn = getattr(item, "function", None)
    mod = getattr(fn, "__module__", "") or ""
    if mod not in _ENVELOPE_MODULES:
        return None
    cs = getattr(item, "callspec", None)
    ident = "-".join(p for p in (cs.id.split("-") if cs is not None else []) if p not in ("run", "select"))   # the commit-kernel axis skips alike
    return f"{mod}::{fn.__name__}[{ident}]"


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    global _expected_skips
    outcome = yield
    rep = outcome.get_result()
    if not rep.skipped or call.when != "call":
        return
    key = _skip_key(item)
    if key is None:
        return
    reason = rep.longrepr[2] if isinstance(rep.longrepr, tuple) else str(rep.longrepr)
    rec = os.environ.get("KB_RECORD_SKIPS")
    if rec:
        with open(rec, "a") as f:
            f.write(key + "\t" + reason.replace("\n", " ") + "\n")
        return
    if _expected_skips is None:
        import json
        _expected_skips = set(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "expected_skips.json")))["skips"])
    if key not in _expected_skips:
        rep.outcome = "failed"
        rep.longrepr = f"UNEXPECTED SKIP (not in tests/golden/expected_skips.json: did the engine's envelope shrink?): {key}: {reason}"


@pytest.fixture(scope="session")
def kb():
    return importlib.import_module("kube-batch_amd")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
