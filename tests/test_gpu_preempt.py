"""The engine's preempt action (kb_run_preempt: statements and victims on the host, PredicateNodes + PrioritizeNodes + SortNodes on
the device) against the oracle's restatement of actions/preempt/preempt.go: committed evictions in cache.Evict order, every
task's status and (sticky) node, the float64 node state, drf / proportion shares and the bind set — on both cases of the
reference's own preempt_test.go:51-131, on random clusters with priorities, gangs, protected pods, scalars and host ports,
under action orders that put preempt before, between and after allocate / backfill, and on adversarial raw snapshots."""
import importlib

import numpy as np
import pytest

import rawgen
import test_pyref_vs_oracle as cases

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
abi, conf, fx = kbm.abi, kbm.conf, kbm.fixtures

pytestmark = [pytest.mark.gpu]


def _has(cfg, plugin):
    return any(po.name == plugin for tier in cfg.tiers for po in tier)


def _compare(e, o, snap, tag, cfg):
    assert [int(t) for t in e.evictions()] == [int(t) for t in o.evictions()], tag
    ej, oj = e.journal(), o.journal()     # every Statement.Evict / Pipeline with its statement number, commit / discard markers; reclaim: stmt 0
    assert ej.shape == oj.shape and np.array_equal(ej, oj), (tag, "journal")
    est, end = e.task_state()
    ost, ond = o.task_state()
    assert np.array_equal(est, ost), (tag, np.nonzero(est != ost)[0][:8])
    assert np.array_equal(end, ond), (tag, np.nonzero(end != ond)[0][:8])
    for name, a, b in zip(("idle", "releasing", "nz_cpu", "nz_mem", "pod_cnt"), e.node_state(), o.node_state()):
        assert np.array_equal(a, b), (tag, name)
    ejs, eqs, _ = e.shares()
    ojs, oqs, _ = o.shares()
    if _has(cfg, "drf"):            # the shares only exist where the plugin that owns them is configured
        assert np.array_equal(ejs, ojs), tag
    if _has(cfg, "proportion"):
        assert np.array_equal(eqs, oqs), tag
    assert np.array_equal(e.binds(), o.binds()), tag


def _run_both(oracle_mod, cfg, snap, order, tag):
    e = engine.Engine(cfg)
    try:
        e.load(snap)
    except engine.EngineError as err:
        e.close()
        if err.code in (abi.KB_E_UNSUPPORTED, abi.KB_E_INVALID):
            pytest.skip(f"outside the engine's envelope: {err}")
        raise
    o = oracle_mod.Oracle(cfg, snap)
    try:
        o.run(order)
    except RuntimeError:
        e.close()
        pytest.skip("the reference would panic on this snapshot")
    try:
        e.run(order)
    except engine.EngineError as err:
        e.close()
        if err.code == abi.KB_E_UNSUPPORTED:
            pytest.skip(f"outside the engine's envelope: {err}")
        raise
    _compare(e, o, snap, tag, cfg)
    e.close()
    o.close()


def _preempt_tiers():
    """preempt_test.go:162-176: one tier, conformance and gang with EnabledPreemptable only."""
    return conf.tiers_literal([conf.PluginOption("conformance", enabled=abi.EN_PREEMPTABLE),
                               conf.PluginOption("gang", enabled=abi.EN_PREEMPTABLE)])


def test_reference_preempt_cases(oracle_mod):
    """actions/preempt/preempt_test.go:51-131, both cases: the FakeEvictor records 1 and 2 evictions."""
    S = kbm.snapshot
    rl = fx.build_resource_list
    cases_ = [
        (S.flatten(nodes=[S.Node("n1", rl("3", "3Gi"))],
                   pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"),
                         fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
                         fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg1"),
                         fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg1")],
                   pod_groups=[S.PodGroup("c1", "pg1", queue="q1")], queues=[S.Queue("q1", 1)]), ["c1/preemptee2"]),
        (S.flatten(nodes=[S.Node("n1", rl("2", "2G"))],
                   pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"),
                         fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
                         fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2"),
                         fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg2")],
                   pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q1")], queues=[S.Queue("q1", 1)]),
         ["c1/preemptee2", "c1/preemptee1"]),
    ]
    for snap, want in cases_:
        cfg = _preempt_tiers()
        e = engine.Engine(cfg)
        e.load(snap)
        e.run(["preempt"])
        o = oracle_mod.Oracle(cfg, snap)
        o.run(["preempt"])
        assert [snap.task_name(int(t)) for t in e.evictions()] == want
        _compare(e, o, snap, want, cfg)
        # the journal: the Evicts, the Pipeline they made room for, and a Commit marker closing the statement
        j = e.last_journal
        assert (j[:, 0] == abi.OP_EVICT).sum() == len(want) and (j[:, 0] == abi.OP_PIPELINE).sum() >= 1 and j[-1, 0] == abi.OP_COMMIT
        e.close()
        o.close()


@pytest.mark.parametrize("seed", range(60))
def test_preempt_on_random_clusters(oracle_mod, seed):
    cfg, snap, order = cases._evict_case(seed)     # orders mix preempt and reclaim with allocate / backfill
    _run_both(oracle_mod, cfg, snap, order, seed)


@pytest.mark.parametrize("seed", range(36))
def test_consecutive_evict_actions(oracle_mod, seed):
    """preempt / reclaim back to back: what a discarded statement leaves behind (a NodeName on a task that is on no node) must survive
    from one action to the next (HostSession::t_off_node; found by tests/test_host_evict_cpu.py)."""
    cfg, snap, _ = cases._evict_case(seed)
    order = cases.EVICT_ORDERS[2 + seed % 4]
    _run_both(oracle_mod, cfg, snap, order, seed)


@pytest.mark.parametrize("seed", range(64))
def test_evict_actions_under_other_tier_layouts(oracle_mod, seed):
    """cases.EVICT_CONFS: the victim rules in other tiers (no priority rule: the whole cached list is walked and the dirty nodes are
    merged into it), bin-packing weights, reclaim decided by proportion."""
    cfg, snap, order = cases._evict_variant(seed)
    _run_both(oracle_mod, cfg, snap, order, seed)


@pytest.mark.parametrize("scale,idx", [(0.02, 3), (0.05, 3), (0.01, 4), (0.002, 5), (0.05, 5)])
def test_preempt_after_allocate_on_scaled_baseline_configs(oracle_mod, scale, idx):
    """BASELINE configs[4] names allocate + backfill + preempt: the three actions in that order on scaled snapshots."""
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(idx, scale))
    order = ["allocate", "backfill", "preempt"]
    cfg = conf.load_scheduler_conf(cases.CONF_FULL.format(actions=", ".join(order)))
    _run_both(oracle_mod, cfg, snap, order, (idx, scale))


def test_reference_reclaim_case(oracle_mod):
    """actions/reclaim/reclaim_test.go:51-99: queue q1 holds the whole node, q2's pending pod reclaims exactly one running pod
    (tiers: conformance + gang with EnabledReclaimable, reclaim_test.go:140-154)."""
    S = kbm.snapshot
    rl = fx.build_resource_list
    snap = S.flatten(nodes=[S.Node("n1", rl("3", "3Gi"))],
                     pods=[fx.build_pod("c1", f"preemptee{i}", "n1", "Running", rl("1", "1G"), "pg1") for i in (1, 2, 3)] +
                          [fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2")],
                     pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q2")], queues=[S.Queue("q1", 1), S.Queue("q2", 1)])
    cfg = conf.tiers_literal([conf.PluginOption("conformance", enabled=abi.EN_RECLAIMABLE), conf.PluginOption("gang", enabled=abi.EN_RECLAIMABLE)])
    e = engine.Engine(cfg)
    e.load(snap)
    e.run(["reclaim"])
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["reclaim"])
    assert [snap.task_name(int(t)) for t in e.evictions()] == ["c1/preemptee1"]
    _compare(e, o, snap, "reclaim_test", cfg)
    e.close()
    o.close()


@pytest.mark.parametrize("seed", range(2, 240, 3))
def test_preempt_on_adversarial_snapshots(oracle_mod, seed):
    snap = rawgen.raw_snapshot(seed)
    order = [["preempt"], ["reclaim", "allocate", "backfill", "preempt"], ["preempt", "allocate"]][(seed // 3) % 3]
    cfg = conf.load_scheduler_conf(cases.CONF_FULL.format(actions=", ".join(order)))
    _run_both(oracle_mod, cfg, snap, order, seed)
