"""The engine's HOST side, whole, on a CPU: kb_engine.cpp + kb_session.cpp + kb_order.cpp + kb_preempt.cpp compiled unchanged with g++ against
tests/host_harness/hip_mock (a synchronous stand-in for the few HIP runtime calls they make) and linked with
tests/host_harness/device_emu.cpp, a sequential restatement of what each kernel launch computes.  The result exports the complete C ABI
of include/kb_engine.h, so the `-m gpu` suites themselves run here — same test functions, same oracle comparison — with the emulated
library loaded through kube-batch_amd/engine.py's own ctypes bindings.

What it covers that nothing else does without a GPU: ActionRun (speculated windows, roll-back and replay, dead shapes, the
feasibility probe), chained rounds and the pinned mailbox protocol, the commit-kernel choice, session load / reset, kb_eval_matrix /
kb_argmax_rows plumbing, kb_round_* (the sharded path's entry points), the evict actions end to end, the aggregate cross-checks.
What it does NOT cover: the kernels.  Their parity with the oracle is the `-m gpu` suite on the MI355X; the product has no CPU path
(tests/test_abi_cpu.py::test_create_without_gpu_fails_loudly) and this library is never loaded outside this file.

Full-size configurations (tests/test_gpu_fullsize.py) are left to the GPU: a sequential matrix over 10k-50k nodes per round is minutes."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE]
CSRC = os.path.join(HERE, "..", "kube-batch_amd", "csrc")
HH = os.path.join(HERE, "host_harness")

engine = importlib.import_module("kube-batch_amd.engine")
kbm = importlib.import_module("kube-batch_amd")
abi = kbm.abi


def build_emulated_library():
    if os.environ.get("KB_EMU_LIB"):                     # an instrumented build (scripts/sanitize_cpu.sh)
        return os.environ["KB_EMU_LIB"]
    out_dir = os.path.join(HH, "build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libkbengine_emu.so")
    srcs = [os.path.join(CSRC, f) for f in ("kb_engine.cpp", "kb_session.cpp", "kb_order.cpp", "kb_preempt.cpp")] + [os.path.join(HH, "device_emu.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("kb_device.h", "kb_eval.hpp", "kb_host.hpp", "kb_preempt.hpp")] + \
        [os.path.join(HH, "hip_mock", "hip", "hip_runtime.h"), os.path.join(HERE, "..", "include", "kb_engine.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        tmp = f"{so}.{os.getpid()}"                      # atomic: parallel pytest workers may build at the same time
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result",
                               "-I" + os.path.join(HH, "hip_mock"), "-o", tmp] + srcs)
        os.replace(tmp, so)
    return so


@pytest.fixture(autouse=True)
def emulated_engine():
    """engine.lib() binds whatever engine.LIB_PATH names: point it at the emulated build for the duration of a test of THIS module."""
    so = build_emulated_library()
    saved = (engine.LIB_PATH, engine._LIB)
    engine.LIB_PATH, engine._LIB = so, None
    yield so
    engine.LIB_PATH, engine._LIB = saved


def test_the_emulated_library_exports_the_whole_c_abi(emulated_engine):
    L = C.CDLL(emulated_engine)
    for name in engine.EXPORTS:
        assert hasattr(L, name), name


def test_product_library_path_is_untouched_outside_this_module():
    assert engine.LIB_PATH.endswith("libkbengine_emu.so")          # inside a test of this module
    assert os.path.basename(os.path.dirname(engine.__file__)) == "kube-batch_amd"
    src = open(engine.__file__).read()
    assert "emu" not in src and "host_harness" not in src          # the product wrapper knows nothing about the emulation


# ---- the `-m gpu` suites, re-collected here without their marker -------------------------------------------------------------
# (module-level `pytestmark = pytest.mark.gpu` belongs to the module a function is collected FROM, so the copies below are plain
# CPU tests; their parametrisation travels with the function objects.)
_SKIP = {
    # sizes that only make sense on the device
    "test_gpu_parity": {"test_wide_cluster_more_than_64k_nodes", "test_full_size_properties_config3"},
}


def _adopt(module_name, only=None):
    mod = importlib.import_module(module_name)
    for name, obj in sorted(vars(mod).items()):
        if not (name.startswith("test_") and callable(obj)) or name in _SKIP.get(module_name, ()):
            continue
        if only is not None and name not in only:
            continue
        globals()[f"{name}__{module_name[5:]}"] = obj


for _m in ("test_gpu_parity", "test_gpu_fuzz", "test_gpu_adversarial", "test_gpu_interpod", "test_gpu_preempt", "test_framework_actions"):
    _adopt(_m)
