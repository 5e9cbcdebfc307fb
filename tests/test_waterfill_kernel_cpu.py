"""kb_waterfill.hip's kernel text, run on the host by 256 threads with a real barrier (tests/host_harness/waterfill_kernel_harness.cpp), against
kb_waterfill.hpp's steps run one after the other — the form the emulated device launches and tests/test_emu_engine_cpu.py holds to the host loop
of kb_session.cpp (and through it to tests/pyref.py and the oracle) — and against the reference's own known answer for the loop
(doc/usage/tutorial.md:297-330, plugins/proportion/proportion.go:101-154).  What the kernel adds to the steps is the assignment of lanes and the
barriers between them; that is what runs here.  No device, no engine: host threads only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HH = os.path.join(HERE, "host_harness")
CSRC = os.path.join(HERE, "..", "kube-batch_amd", "csrc")
MAX_RES = 32                                                  # include/kb_engine.h: KB_MAX_RES


class Res(C.Structure):
    _fields_ = [("v", C.c_double * MAX_RES), ("mask", C.c_uint32), ("_pad", C.c_uint32)]


class WfQueue(C.Structure):
    _fields_ = [("deserved", Res), ("request", Res), ("inc", Res), ("dec", Res), ("weight", C.c_int32), ("has_attr", C.c_uint32), ("meet", C.c_uint32), ("active", C.c_uint32)]


class WfState(C.Structure):
    _fields_ = [("remaining", Res), ("increased", Res), ("decreased", Res), ("total_weight", C.c_int32), ("stop", C.c_uint32), ("share_at_open", C.c_uint32),
                ("underflow", C.c_uint32), ("passes", C.c_uint32), ("_pad", C.c_uint32)]


@pytest.fixture(scope="module")
def harness():
    if os.environ.get("KB_WATERFILL_HARNESS_LIB"):            # an instrumented build (scripts/sanitize_cpu.sh)
        L = C.CDLL(os.environ["KB_WATERFILL_HARNESS_LIB"])
    else:
        out_dir = os.path.join(HH, "build")
        os.makedirs(out_dir, exist_ok=True)
        so = os.path.join(out_dir, "libwaterfillkernel.so")
        src = os.path.join(HH, "waterfill_kernel_harness.cpp")
        deps = [src] + [os.path.join(CSRC, f) for f in ("kb_waterfill.hip", "kb_waterfill.hpp", "kb_res.hpp")] + [os.path.join(HH, "hip_mock", "hip", "hip_runtime.h")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            tmp = f"{so}.{os.getpid()}"
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-pthread",
                                   "-I" + os.path.join(HH, "hip_mock"), "-o", tmp, src])
            os.replace(tmp, so)
        L = C.CDLL(so)
    lay = (C.c_uint32 * 8)()
    L.kbwf_layout(lay)
    assert (lay[0], lay[1], lay[2]) == (C.sizeof(Res), C.sizeof(WfQueue), C.sizeof(WfState))
    assert (lay[4], lay[5], lay[6]) == (WfQueue.weight.offset, WfState.total_weight.offset, Res.mask.offset)
    assert lay[3] == 256
    return L


def _res(dst, vals, mask):
    for d, x in enumerate(vals):
        dst.v[d] = float(x)
    dst.mask = int(mask)


def _case(rng, Q, R):
    """queues as proportion sees them at OnSessionOpen: integral milli / byte quantities, scalar keys present on some operands only"""
    qs = (WfQueue * max(Q, 1))()
    st = WfState()
    full = (1 << (R - 2)) - 1
    tmask = full if rng.rand() < 0.7 else int(rng.randint(0, full + 1))
    total = [float(rng.randint(1, 4000)) * 1000.0, float(rng.randint(1, 4000)) * float(1 << 30)] + [float(rng.randint(0, 64)) * 1000.0 for _ in range(R - 2)]
    total = [x if d < 2 or (tmask >> (d - 2)) & 1 else 0.0 for d, x in enumerate(total)]
    _res(st.remaining, total, tmask)
    style = rng.randint(0, 4)
    for q in range(Q):
        a = qs[q]
        a.has_attr = 0 if rng.rand() < 0.1 else 1
        a.weight = int(rng.choice([0, 1, 1, 2, 3, 7, 100, 2 ** 31 - 1])) if style == 0 else int(rng.randint(1, 10))
        scale = {0: 2.0, 1: 0.3, 2: 1.0, 3: 5.0}[style] / max(Q, 1)   # requests above / below / around an equal split of the total
        rmask = tmask if rng.rand() < 0.6 else int(rng.randint(0, full + 1))
        req = [np.floor(total[d] * scale * rng.rand() * 2.0) for d in range(R)]
        if rng.rand() < 0.1:
            req = [0.0] * R                                   # a queue whose jobs request nothing
        req = [x if d < 2 or (rmask >> (d - 2)) & 1 else 0.0 for d, x in enumerate(req)]
        _res(a.request, req, rmask)
    return qs, st


def _copy(x):
    y = type(x)()
    C.memmove(C.byref(y), C.byref(x), C.sizeof(x))
    return y


@pytest.mark.parametrize("block", range(4))
def test_kernel_text_on_host_threads_equals_the_steps_in_sequence(harness, block):
    rng = np.random.RandomState(1000 + block)
    multi = 0
    for i in range(12):
        Q = int(rng.choice([0, 1, 2, 3, 17, 64, 128, 255, 256, 257, 300, 700]))
        R = int(rng.choice([2, 3, 4, 16, 32]))
        qs, st = _case(rng, Q, R)
        qs2, st2 = _copy(qs), _copy(st)
        harness.kbwf_run_kernel(qs, C.c_uint32(Q), C.byref(st), C.c_int(R))
        harness.kbwf_run_sequential(qs2, C.c_uint32(Q), C.byref(st2), C.c_int(R))
        assert (st.passes, st.underflow, st.share_at_open) == (st2.passes, st2.underflow, st2.share_at_open), (block, i, Q, R)
        assert bytes(st.remaining) == bytes(st2.remaining), (block, i, Q, R)
        for q in range(Q):
            assert bytes(qs[q].deserved) == bytes(qs2[q].deserved) and qs[q].meet == qs2[q].meet, (block, i, q, Q, R)
        multi += st.passes > 1
    assert multi >= 3                                         # the loop really went round: queues met their request one after the other


def test_kernel_text_on_the_tutorial_example(harness):
    """doc/usage/tutorial.md:297-330: 9 cpu / 27 Gi, weights 2 and 4, requests 5 x (1 cpu, 2 Gi) and 10 x (1 cpu, 2 Gi) -> (3, 9 Gi) and (6, 18 Gi)"""
    Gi = float(1 << 30)
    qs = (WfQueue * 2)()
    st = WfState()
    _res(st.remaining, [9000.0, 27.0 * Gi], 0)
    for q, (w, n) in enumerate([(2, 5), (4, 10)]):
        qs[q].has_attr, qs[q].weight = 1, w
        _res(qs[q].request, [1000.0 * n, 2.0 * Gi * n], 0)
    harness.kbwf_run_kernel(qs, C.c_uint32(2), C.byref(st), C.c_int(2))
    assert [qs[0].deserved.v[0], qs[0].deserved.v[1]] == [3000.0, 9.0 * Gi]
    assert [qs[1].deserved.v[0], qs[1].deserved.v[1]] == [6000.0, 18.0 * Gi]
    assert st.underflow == 0 and st.share_at_open == 1


def test_kernel_text_no_weight_at_all(harness):
    """proportion.go:113-116: total weight 0 in the first pass -> the loop never runs, no updateShare: share_at_open = 0, deserved stays empty"""
    qs = (WfQueue * 3)()
    st = WfState()
    _res(st.remaining, [1000.0, 1e9], 0)
    for q in range(3):
        qs[q].has_attr, qs[q].weight = 1, 0
        _res(qs[q].request, [500.0, 1e8], 0)
    harness.kbwf_run_kernel(qs, C.c_uint32(3), C.byref(st), C.c_int(2))
    assert st.share_at_open == 0 and st.passes == 0
    assert all(qs[q].deserved.v[0] == 0.0 and qs[q].meet == 0 for q in range(3))
