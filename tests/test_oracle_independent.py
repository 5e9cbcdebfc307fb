"""A second, independent restatement (pure Python, written from the vendored source text) of the arithmetic the reference's own
tests do not pin — the nodeorder scorers and Resource.LessEqual — checked against the C oracle on random inputs.  Two
restatements that agree bit for bit make a transcription slip in either unlikely; it is still not a reference run.
"""
import ctypes as C
import importlib
import math

import numpy as np
import pytest
from hypothesis import assume, given, settings, strategies as st

kbm = importlib.import_module("kube-batch_amd")

MAX_PRIORITY = 10    # schedulerapi.MaxPriority


def go_div(a: int, b: int) -> int:
    """Go's int64 division truncates toward zero (operands here are never negative)."""
    return int(a // b) if (a >= 0) == (b > 0) else -int((-a) // b)


def least_requested_score(requested: int, capacity: int) -> int:      # least_requested.go:46-58
    if capacity == 0 or requested > capacity:
        return 0
    return go_div((capacity - requested) * MAX_PRIORITY, capacity)


def most_requested_score(requested: int, capacity: int) -> int:       # most_requested.go:50-61
    if capacity == 0 or requested > capacity:
        return 0
    return go_div(requested * MAX_PRIORITY, capacity)


def fraction_of_capacity(requested: int, capacity: int) -> float:     # balanced_resource_allocation.go:74-79
    return 1.0 if capacity == 0 else float(requested) / float(capacity)


def balanced_score(rc, ac, rm, am) -> int:                            # balanced_resource_allocation.go:42-72 (volume gate off)
    cf, mf = fraction_of_capacity(rc, ac), fraction_of_capacity(rm, am)
    if cf >= 1 or mf >= 1:
        return 0
    return int((1 - math.fabs(cf - mf)) * float(MAX_PRIORITY))       # int64(x) truncates toward zero; x >= 0 here


def node_scores(rc, ac, rm, am):
    # leastResourceScorer / mostResourceScorer: both resources weigh 1 (resource_allocation.go defaults), nodeScore / weightSum
    least = go_div(least_requested_score(rc, ac) + least_requested_score(rm, am), 2)
    most = go_div(most_requested_score(rc, ac) + most_requested_score(rm, am), 2)
    return least, most, balanced_score(rc, ac, rm, am)


quantity = st.one_of(st.integers(0, 200_000), st.integers(0, 1 << 40), st.sampled_from([0, 1, 100, 1000, 4000, 200 << 20, 8 << 30, (1 << 47) - 1]))


@settings(max_examples=600, deadline=None)
@given(rc=quantity, ac=quantity, rm=quantity, am=quantity)
def test_scorers_agree_with_the_oracle(oracle_mod, rc, ac, rm, am):
    L = oracle_mod.lib()
    least, most, bal = C.c_int64(), C.c_int64(), C.c_int64()
    L.kbo_scorers(rc, ac, rm, am, C.byref(least), C.byref(most), C.byref(bal))
    assert (least.value, most.value, bal.value) == node_scores(rc, ac, rm, am)


# ---- Resource.LessEqual (api/resource_info.go:268-302), cpu / memory / one scalar with map-presence semantics
MIN_CPU, MIN_MEM, MIN_SCALAR = 10.0, 10.0 * 1024 * 1024, 10.0


def less_equal(l, r) -> bool:
    """l, r: (cpu, mem, scalars-or-None) with scalars a dict name -> value"""
    def le(a, b, eps):
        return a < b or math.fabs(a - b) < eps
    if not le(l[0], r[0], MIN_CPU):
        return False
    if not le(l[1], r[1], MIN_MEM):
        return False
    if l[2] is None:
        return True
    for name, lq in l[2].items():
        if lq <= MIN_SCALAR:
            continue
        if r[2] is None:
            return False
        if not le(lq, r[2].get(name, 0.0), MIN_SCALAR):
            return False
    return True


val = st.one_of(st.sampled_from([0.0, 5.0, 9.999, 10.0, 10.001, 1000.0, 1009.999, 1010.0, 2000.0, 1e9, 1e9 + 10485759.0, 1e9 + 10485760.0]),
                st.floats(0, 1e12, allow_nan=False, allow_infinity=False))
scalars = st.one_of(st.none(), st.dictionaries(st.sampled_from(["a", "b"]), val, max_size=2))


@settings(max_examples=600, deadline=None)
@given(lc=val, lm=val, ls=scalars, rc=val, rm=val, rs=scalars)
def test_less_equal_agrees_with_the_oracle(oracle_mod, lc, lm, ls, rc, rm, rs):
    L = oracle_mod.lib()
    Res = oracle_mod.OracleRes
    L.kbo_set_dims(4)
    dims = {"a": 2, "b": 3}

    def mk(c, m, s):
        r = Res()
        r.v[0], r.v[1] = c, m
        r.mask = 0
        if s is not None:
            for k, v in s.items():
                r.v[dims[k]] = v
                r.mask |= 1 << (dims[k] - 2)
        return r

    # empty-but-non-nil maps are not representable in the snapshot (the flattener never produces them)
    assume(not (ls is not None and not ls) and not (rs is not None and not rs))
    assert bool(L.kbo_res_less_equal(C.byref(mk(lc, lm, ls)), C.byref(mk(rc, rm, rs)))) == less_equal((lc, lm, ls), (rc, rm, rs))


# ---- the matrix kernel's integer arithmetic (kube-batch_amd/csrc/kb_kernels.hip div10 / score_core), restated in IEEE doubles:
#      floor(10*req/cap) from a reciprocal estimate + one remainder correction, and least = 10 - most - (remainder != 0).
#      Checked against exact integer division over the whole envelope kb_session_load admits (quantities < 2^48).
def _div10(req: int, cap: int):
    a = req * 10
    q = int(float(a) * (1.0 / float(cap)))            # (int)((double)a * inv_cap): truncation, a < 2^52 is exact in a double
    rem = a - q * cap
    if rem < 0:
        q, rem = q - 1, rem + cap
    elif rem >= cap:
        q, rem = q + 1, rem - cap
    return q, rem != 0, rem


envelope = st.one_of(st.integers(1, (1 << 48) - 1), st.integers(1, 200_000), st.sampled_from([1, 2, 3, 7, 10, 1000, 64000, (1 << 48) - 1, (1 << 47) + 1]))


@settings(max_examples=3000, deadline=None)
@given(cap=envelope, frac=st.floats(0.0, 1.0), nudge=st.integers(-2, 2))
def test_reciprocal_div10_is_exact(cap, frac, nudge):
    req = min(cap, max(0, int(cap * frac) + nudge))
    q, rem_nz, rem = _div10(req, cap)
    assert 0 <= rem < cap
    assert q == (req * 10) // cap and rem_nz == ((req * 10) % cap != 0)
    assert most_requested_score(req, cap) == q
    assert least_requested_score(req, cap) == 10 - q - (1 if rem_nz else 0)      # floor(10 - x) = 10 - ceil(x)


def test_reciprocal_div10_on_multiples_and_neighbours():
    """The estimate is worst at exact multiples (q * cap / 10) and one unit either side of them."""
    for cap in (1, 3, 7, 10, 999, 1000, 4000, 64000, (1 << 30), (1 << 40) + 12345, (1 << 48) - 1):
        for k in range(11):
            for d in (-1, 0, 1):
                req = (k * cap) // 10 + d
                if 0 <= req <= cap:
                    q, rem_nz, _ = _div10(req, cap)
                    assert q == (req * 10) // cap and rem_nz == ((req * 10) % cap != 0), (cap, req)
