"""The per-pair arithmetic of the engine (kube-batch_amd/csrc/kb_eval.hpp: the one header the matrix kernel, both commit kernels and the
host's preempt evaluator share), compiled for the host and compared with the reference's own arithmetic written out literally
(int64 `/`, IEEE double `/`: vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/*.go, api/resource_info.go:268-302):

* div_small_f64 — RN(a / b) from the per-node reciprocal in three operations (the proof is in the header) — on random operands below
  2^48, on denominators with few significant bits / all ones / 2^k +- 1 with numerators next to short binary fractions of them (the
  neighbourhood of the rounding boundaries), on small denominators;
* score_core_f64 — Least / Most / Balanced, branch-free — on cluster-shaped inputs, on small integers that hit every boundary of the
  0..10 scores, and up to the 2^48 envelope;
* le_eps — one subtraction and one compare — around the three epsilons, with negative right-hand sides.
The kernels run the same header on the MI355X; their parity with the oracle is the -m gpu suite."""
import ctypes as C
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HH = os.path.join(HERE, "host_harness")
CSRC = os.path.join(HERE, "..", "kube-batch_amd", "csrc")


@pytest.fixture(scope="module")
def eh():
    out_dir = os.path.join(HH, "build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libevalharness.so")
    src = os.path.join(HH, "eval_harness.cpp")
    deps = [src, os.path.join(CSRC, "kb_eval.hpp"), os.path.join(CSRC, "kb_device.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        tmp = f"{so}.{os.getpid()}"
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall",
                               "-I" + os.path.join(HH, "hip_mock"), "-o", tmp, src])
        os.replace(tmp, so)
    L = C.CDLL(so)
    L.eh_div_small.restype = C.c_double
    L.eh_div_small.argtypes = [C.c_double, C.c_double]
    L.eh_div_mismatches.restype = C.c_uint64
    L.eh_div_mismatches.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.eh_score_mismatches.restype = C.c_uint64
    L.eh_score_mismatches.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_int64)]
    L.eh_le_mismatches.restype = C.c_uint64
    L.eh_le_mismatches.argtypes = [C.c_uint64, C.c_uint64]
    L.eh_score.restype = C.c_uint32
    L.eh_score.argtypes = [C.c_int64] * 6 + [C.c_int] * 3
    L.eh_ref_score.restype = C.c_uint32
    L.eh_ref_score.argtypes = [C.c_int64] * 6 + [C.c_int] * 3
    return L


@pytest.mark.parametrize("mode,bits,n", [(0, 48, 20_000_000), (0, 20, 5_000_000), (1, 48, 20_000_000), (1, 30, 5_000_000), (2, 48, 5_000_000)])
def test_three_operation_division_is_correctly_rounded(eh, mode, bits, n):
    a, b = C.c_double(), C.c_double()
    bad = eh.eh_div_mismatches(0xD1F + 97 * mode + bits, n, mode, bits, C.byref(a), C.byref(b))
    assert bad == 0, f"{bad} mismatches, first: {a.value} / {b.value}"


def test_division_known_values(eh):
    for a, b in [(0, 1), (0, 7), (1, 3), (2, 3), (1, 2), (3, 5), (6, 10), (2 ** 48 - 2, 2 ** 48 - 1), (1, 2 ** 48 - 1), (2 ** 47, 2 ** 48 - 1),
                 (123456789, 987654321), (2400, 24000), (5 << 30, 10 << 30)]:
        assert eh.eh_div_small(float(a), float(b)) == a / b, (a, b)


@pytest.mark.parametrize("mode,n", [(0, 10_000_000), (1, 20_000_000), (2, 10_000_000)])
def test_branch_free_scorers_equal_the_reference_arithmetic(eh, mode, n):
    bad = (C.c_int64 * 9)()
    nb = eh.eh_score_mismatches(0x5C0 + mode, n, mode, bad)
    assert nb == 0, f"{nb} mismatches, first: {list(bad)}"


def test_scorer_guards(eh):
    """capacity 0, request above / equal to capacity, a full node: the reference's early-outs"""
    for args in [(1, 1, 0, 0, 0, 0), (1, 1, 0, 0, 0, 10), (5, 5, 5, 5, 10, 10), (1, 1, 10, 10, 10, 10), (0, 0, 10, 10, 10, 10), (0, 0, 0, 0, 10, 10),
                 (3, 1, 3, 4, 10, 10), (1, 0, 9, 10, 10, 10)]:
        for w in [(1, 0, 1), (0, 5, 1), (2, 3, 1)]:
            assert eh.eh_score(*args, *w) == eh.eh_ref_score(*args, *w), (args, w)


def test_less_equal_as_one_subtraction(eh):
    assert eh.eh_le_mismatches(0x1E, 30_000_000) == 0
