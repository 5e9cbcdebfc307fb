"""The oracle's fast mode (per-shape cached rows + one-node repairs, oracle/kb_oracle.c: kbo_set_fast) against its faithful mode:
ordered decisions, bind set, node state, shares and counters must be identical.  The fast mode exists so that snapshots the
faithful loop needs minutes for (BASELINE config 5: 1M tasks x 50k nodes) get a golden bind set; it is test infrastructure and
must never be trusted beyond what this file checks."""
import importlib
import json
import os

import numpy as np
import pytest

import rawgen
import test_pyref_vs_oracle as cases

kbm = importlib.import_module("kube-batch_amd")
conf_mod = kbm.conf


def _same(o1, o2):
    assert np.array_equal(o1.decisions(), o2.decisions())
    assert np.array_equal(o1.binds(), o2.binds())
    for a, b in zip(o1.node_state(), o2.node_state()):
        assert np.array_equal(a, b)
    for a, b in zip(o1.shares(), o2.shares()):
        assert np.array_equal(a, b)
    for a, b in zip(o1.task_state(), o2.task_state()):
        assert np.array_equal(a, b)
    assert o1.evals == o2.evals and o1.popped == o2.popped


def _same_evict(o1, o2):
    """the evict actions: journal (every Statement operation with its statement number, commit / discard markers), the
    evictions in cache.Evict order, and everything _same() compares"""
    assert np.array_equal(o1.journal(), o2.journal())
    assert np.array_equal(o1.evictions(), o2.evictions())
    _same(o1, o2)


def _run_order(oracle_mod, cfg, snap, order, fast):
    o = oracle_mod.Oracle(cfg, snap)
    if fast:
        o.set_fast(True)
    o.run(order)
    return o


def _both_modes(oracle_mod, cfg, snap, order):
    try:
        slow = _run_order(oracle_mod, cfg, snap, order, False)
    except RuntimeError:
        with pytest.raises(RuntimeError):
            _run_order(oracle_mod, cfg, snap, order, True)     # the reference would panic: both modes must say so
        return None
    fast = _run_order(oracle_mod, cfg, snap, order, True)
    _same_evict(slow, fast)
    return fast


def _run(oracle_mod, cfg, snap, fast):
    o = oracle_mod.Oracle(cfg, snap)
    if fast:
        o.set_fast(True)
    o.run(["allocate", "backfill"])
    return o


@pytest.mark.parametrize("idx,scale", [(2, 0.3), (3, 0.03), (3, 0.08), (4, 0.03), (5, 0.004)])
def test_fast_equals_faithful_on_synthetic_snapshots(oracle_mod, idx, scale):
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(idx, scale))
    for cfg in (conf_mod.load_scheduler_conf(), conf_mod.load_scheduler_conf(cases.CONF_TMPL.format(wl=0, wm=5, wa=1, wb=1))):
        _same(_run(oracle_mod, cfg, snap, False), _run(oracle_mod, cfg, snap, True))


@pytest.mark.parametrize("seed", range(120))
def test_fast_equals_faithful_on_adversarial_snapshots(oracle_mod, seed):
    snap = rawgen.raw_snapshot(seed)
    rng = np.random.RandomState(seed)
    wl, wm, wa, wb = [int(x) for x in rng.choice([0, 1, 1, 2, 5], size=4)]
    cfg = conf_mod.load_scheduler_conf(cases.CONF_TMPL.format(wl=wl, wm=wm, wa=wa, wb=wb))
    try:
        slow = _run(oracle_mod, cfg, snap, False)
    except RuntimeError:
        with pytest.raises(RuntimeError):
            _run(oracle_mod, cfg, snap, True)     # the reference would panic: both modes must say so
        return
    _same(slow, _run(oracle_mod, cfg, snap, True))


def test_fast_mode_reproduces_the_full_size_digest_of_config_3(oracle_mod):
    """100k x 10k: the committed digest was produced by the faithful mode (tests/golden/make_fullsize_golden.py)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_fullsize_golden as mfg
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_digests.json")))
    for name in ("config3_full", "config4_binpack_full"):
        cfg, snap = mfg.case_inputs(kbm, name)
        o = _run(oracle_mod, cfg, snap, True)
        assert mfg.digest_of(np, o.decisions(), o.binds()) == golden[name]["sha256"], name
        assert o.evals == golden[name]["evals"]


# ---- preempt: the fast mode (per-queue node sets, cached SortNodes lists, one walk per run of identical preemptors) ----

@pytest.mark.parametrize("seed", range(60))
def test_fast_preempt_equals_faithful_on_random_clusters(oracle_mod, seed):
    """the clusters and action orders of tests/test_gpu_preempt.py::test_preempt_on_random_clusters (preempt and reclaim mixed with
    allocate / backfill)"""
    cfg, snap, order = cases._evict_case(seed)
    _both_modes(oracle_mod, cfg, snap, order)


@pytest.mark.parametrize("seed", range(36))
def test_fast_preempt_equals_faithful_on_consecutive_evict_actions(oracle_mod, seed):
    cfg, snap, _ = cases._evict_case(seed)
    _both_modes(oracle_mod, cfg, snap, cases.EVICT_ORDERS[2 + seed % 4])


@pytest.mark.parametrize("seed", range(64))
def test_fast_preempt_equals_faithful_under_other_tier_layouts(oracle_mod, seed):
    cfg, snap, order = cases._evict_variant(seed)
    _both_modes(oracle_mod, cfg, snap, order)


@pytest.mark.parametrize("seed", range(2, 240, 3))
def test_fast_preempt_equals_faithful_on_adversarial_snapshots(oracle_mod, seed):
    snap = rawgen.raw_snapshot(seed)
    order = [["preempt"], ["reclaim", "allocate", "backfill", "preempt"], ["preempt", "allocate"]][(seed // 3) % 3]
    cfg = conf_mod.load_scheduler_conf(cases.CONF_FULL.format(actions=", ".join(order)))
    _both_modes(oracle_mod, cfg, snap, order)


@pytest.mark.parametrize("scale,idx", [(0.02, 3), (0.05, 3), (0.01, 4), (0.002, 5), (0.02, 5)])
def test_fast_preempt_equals_faithful_on_scaled_baseline_configs(oracle_mod, scale, idx):
    """BASELINE configs[4]'s three actions on scaled snapshots; at (5, 0.02) the faithful preempt walks 20k preemptors x 1k nodes"""
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(idx, scale))
    order = ["allocate", "backfill", "preempt"]
    cfg = conf_mod.load_scheduler_conf(cases.CONF_FULL.format(actions=", ".join(order)))
    fast = _both_modes(oracle_mod, cfg, snap, order)
    assert fast is not None


def test_reference_preempt_cases_in_fast_mode(oracle_mod):
    """actions/preempt/preempt_test.go:51-131 through the fast mode: 1 and 2 evictions, and the journal's shape"""
    import test_oracle_kat as kat
    for snap, cfg, want in kat.preempt_reference_cases():
        o = _run_order(oracle_mod, cfg, snap, ["preempt"], True)
        assert [snap.task_name(int(t)) for t in o.evictions()] == want
        j = o.journal()
        assert (j[:, 0] == kbm.abi.OP_EVICT).sum() == len(want) and j[-1, 0] == kbm.abi.OP_COMMIT
