"""The whole allocate + backfill loop restated twice — oracle/kb_oracle.c (C) and tests/pyref.py (pure Python, written from the
Go sources independently) — must agree decision for decision, bind for bind and bit for bit in the final node state and shares on
small random clusters that exercise releasing capacity (Pipeline), init containers, scalar resources with nil-map semantics,
priorities, pod caps, static classes, preferred node affinity and host ports.  See DESIGN.md §5."""
import importlib

import numpy as np
import pytest

import pyref

kbm = importlib.import_module("kube-batch_amd")
abi, conf, snapmod = kbm.abi, kbm.conf, kbm.snapshot

CONF_TMPL = """
actions: "allocate, backfill"
tiers:
- plugins:
  - name: priority
  - name: gang
- plugins:
  - name: drf
  - name: predicates
  - name: proportion
  - name: nodeorder
    arguments:
      leastrequested.weight: {wl}
      mostrequested.weight: {wm}
      nodeaffinity.weight: {wa}
      balancedresource.weight: {wb}
"""

CONF_NO_SHARES = """
actions: "allocate, backfill"
tiers:
- plugins:
  - name: priority
  - name: gang
  - name: predicates
  - name: nodeorder
"""


def _case(seed):
    rng = np.random.RandomState(7000 + seed)
    R = int(rng.choice([2, 3, 5]))
    p = snapmod.SynthParams(
        n_tasks=int(rng.randint(30, 260 if seed < 60 else 1200)), n_nodes=int(rng.randint(3, 40 if seed < 60 else 120)), n_queues=int(rng.randint(1, 5)), n_res=R,
        seed=snapmod.SEED_BASE + 900 + seed, preload_node_frac=float(rng.uniform(0, 0.8)), running_job_frac=float(rng.uniform(0, 0.3)),
        best_effort_frac=float(rng.uniform(0, 0.15)), no_mem_key_frac=float(rng.uniform(0, 0.2)), scalar_job_frac=float(rng.uniform(0, 0.6)),
        zone_selector_frac=float(rng.uniform(0, 0.4)), n_zones=int(rng.randint(1, 5)))
    s = snapmod.synth(p)
    N, T = s.n_nodes, s.n_tasks
    rel = rng.uniform(size=N) < rng.uniform(0, 0.6)
    s.node_releasing[0] = np.where(rel, rng.choice([500, 1000, 4000, 16000], size=N), 0).astype(np.float64)
    s.node_releasing[1] = np.where(rel, rng.choice([1, 4, 16, 64], size=N) * float(1 << 30), 0)
    for d in range(2, R):
        s.node_releasing[d] = np.where(rel & (s.node_allocatable[d] > 0), 1000.0 * rng.randint(0, 3, size=N), 0)
    pending = (s.task_status == abi.TASK_PENDING) & (s.task_resreq[0] > 0)
    bump = pending & (rng.uniform(size=T) < rng.uniform(0, 0.3))
    s.task_init_resreq[0] = np.where(bump, s.task_resreq[0] + rng.choice([100, 500, 2000], size=T), s.task_init_resreq[0])
    s.task_priority[:] = rng.choice([1, 1, 5, 9], size=T).astype(np.int32)
    tight = rng.uniform(size=N) < 0.3
    s.node_max_pods[:] = np.where(tight, s.node_pod_cnt + rng.randint(0, 6, size=N), s.node_max_pods).astype(np.int32)
    if seed % 3 == 0:
        aff = rng.choice([0, 0, 2, 5, 30], size=(s.n_task_classes, s.n_node_classes)).astype(np.int32)
        aff[rng.uniform(size=s.n_task_classes) < 0.4] = 0
        s.class_affinity = aff
    if seed % 4 == 1:
        want = np.zeros(T, np.uint64)
        has = rng.uniform(size=T) < 0.3
        want[has] = (np.uint64(1) << rng.randint(0, 5, size=int(has.sum())).astype(np.uint64))
        cmask = want | np.where(want != 0, np.uint64(1) << np.uint64(5), np.uint64(0)).astype(np.uint64)
        s.task_port_want, s.task_port_conflict = want, cmask
        s.node_ports = np.where(rng.uniform(size=N) < 0.2, rng.randint(1, 64, size=N), 0).astype(np.uint64)
    s._check()
    if seed % 5 == 4:
        cfg = conf.load_scheduler_conf(CONF_NO_SHARES)
    else:
        wl, wm, wa, wb = [int(x) for x in rng.choice([0, 1, 1, 2, 5], size=4)]
        cfg = conf.load_scheduler_conf(CONF_TMPL.format(wl=wl, wm=wm, wa=wa, wb=wb))
    return cfg, s


def _tiers(cfg):
    out = []
    for tier in cfg.tiers:
        opts = []
        for po in tier:
            args = {}
            for k, v in (po.arguments or {}).items():
                try:
                    args[k] = int(str(v), 10)      # framework/arguments.go:29-46: a value Atoi rejects leaves the default
                except ValueError:
                    pass
            opts.append((po.name, po.enabled, args))
        out.append(opts)
    return out


@pytest.mark.parametrize("seed", range(80))
def test_python_restatement_equals_c_oracle(oracle_mod, seed):
    cfg, snap = _case(seed)
    _compare(oracle_mod, cfg, snap, seed)


@pytest.mark.parametrize("seed", range(40))
def test_host_port_masks_of_several_words(oracle_mod, seed):
    """kb_snapshot.port_words > 1: more than 64 interned (ip, protocol, port) triples.  The Python side keeps one unbounded integer per
    mask (the reference keeps sets), the C side words: the two must still agree on every decision."""
    import rawgen
    cfg, snap = _case(seed)
    rawgen.widen_ports(snap, 4400 + seed, words=2 + seed % 3)
    assert snap.port_words == 2 + seed % 3 and snap.task_port_conflict.shape == (snap.n_tasks, snap.port_words)
    _compare(oracle_mod, cfg, snap, seed)


def _compare(oracle_mod, cfg, snap, seed):
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate", "backfill"])
    p = pyref.Session(_tiers(cfg), snap).run(["allocate", "backfill"])

    od = o.decisions()
    pd = np.array(p.decisions, dtype=np.uint32).reshape(-1, 3)
    assert pd.shape == od.shape, (seed, pd.shape, od.shape)
    assert np.array_equal(pd, od), f"seed {seed}: first divergence at decision {int(np.argmax((pd != od).any(axis=1)))}"
    pb = np.full(snap.n_tasks, abi.KB_NONE, np.uint32)
    for t, n in p.binds.items():
        pb[t] = n
    assert np.array_equal(pb, o.binds())
    idle, rel, nzc, nzm, cnt = o.node_state()
    for n in range(snap.n_nodes):
        for d in range(snap.n_res):
            assert p.idle[n].get(d) == idle[d, n], (seed, n, d)
            assert p.rel[n].get(d) == rel[d, n], (seed, n, d)
    assert np.array_equal(np.array(p.nzc), nzc) and np.array_equal(np.array(p.nzm), nzm) and np.array_equal(np.array(p.podcnt), cnt)
    if any(po.name == "drf" for t in cfg.tiers for po in t):
        js, qs, des = o.shares()
        assert np.array_equal(np.array(p.jshare), js)
        for q, a in p.qattr.items():
            assert a["share"] == qs[q], (seed, q)
            for d in range(snap.n_res):
                assert a["deserved"].get(d) == des[d, q], (seed, q, d)
    st, nd = o.task_state()
    assert np.array_equal(np.array(p.status, np.uint8), st)
    assert o.popped == p.popped
    o.close()


# ---- the Python restatement against the reference's own known answers (the same ones tests/test_oracle_kat.py pins the C oracle to)
fixtures = importlib.import_module("kube-batch_amd.fixtures")


@pytest.mark.parametrize("case", range(2))
def test_python_restatement_allocate_reference_cases(case):
    """pkg/scheduler/actions/allocate/allocate_test.go:38-212 TestAllocate"""
    name, snap, expected = fixtures.allocate_cases()[case]
    p = pyref.Session(_tiers(fixtures.allocate_test_tiers()), snap).run(["allocate"])
    pb = np.full(snap.n_tasks, abi.KB_NONE, np.uint32)
    for t, n in p.binds.items():
        pb[t] = n
    assert snap.bind_map(pb) == expected, name


def test_python_restatement_proportion_tutorial_example():
    """doc/usage/tutorial.md:297-330: deserved = (3 cpu, 9 Gi) and (6 cpu, 18 Gi) for weights 2 and 4"""
    Gi = 1 << 30
    pods = [snapmod.Pod("q1", f"p{i}", [{"cpu": "1", "memory": "2Gi"}], group_name="j1") for i in range(5)]
    pods += [snapmod.Pod("q2", f"p{i}", [{"cpu": "1", "memory": "2Gi"}], group_name="j2") for i in range(10)]
    snap = snapmod.flatten(
        nodes=[snapmod.Node("n1", {"cpu": "6", "memory": "15Gi", "pods": "110"}),
               snapmod.Node("n2", {"cpu": "3", "memory": "12Gi", "pods": "110"})],
        pods=pods,
        pod_groups=[snapmod.PodGroup("q1", "j1", queue="queue1"), snapmod.PodGroup("q2", "j2", queue="queue2")],
        queues=[snapmod.Queue("queue1", 2), snapmod.Queue("queue2", 4)])
    p = pyref.Session(_tiers(conf.load_scheduler_conf()), snap)
    assert [p.qattr[0]["deserved"].cpu, p.qattr[0]["deserved"].mem] == [3000.0, 9.0 * Gi]
    assert [p.qattr[1]["deserved"].cpu, p.qattr[1]["deserved"].mem] == [6000.0, 18.0 * Gi]


def test_python_restatement_example_job_spread():
    """BASELINE config 1: example/job.yaml on 3 nodes spreads 2/2/2"""
    cfg, snap = fixtures.example_job()
    p = pyref.Session(_tiers(cfg), snap).run(["allocate"])
    assert len(p.binds) == 6
    assert sorted(np.bincount(list(p.binds.values()), minlength=3).tolist()) == [2, 2, 2]
    assert [d[1] for d in p.decisions[:3]] == [0, 1, 2]


# ---- preempt and reclaim (oracle-only actions so far: groundwork for the engine's next actions)
CONF_FULL = """
actions: "{actions}"
tiers:
- plugins:
  - name: priority
  - name: gang
  - name: conformance
- plugins:
  - name: drf
  - name: predicates
  - name: proportion
  - name: nodeorder
"""


def _evict_case(seed):
    rng = np.random.RandomState(9000 + seed)
    R = int(rng.choice([2, 2, 3]))
    p = snapmod.SynthParams(
        n_tasks=int(rng.randint(40, 400)), n_nodes=int(rng.randint(2, 24)), n_queues=int(rng.randint(1, 5)), n_res=R,
        seed=snapmod.SEED_BASE + 1300 + seed, preload_node_frac=float(rng.uniform(0.3, 1.0)), running_job_frac=float(rng.uniform(0.2, 0.7)),
        best_effort_frac=float(rng.uniform(0, 0.1)), no_mem_key_frac=float(rng.uniform(0, 0.2)), scalar_job_frac=float(rng.uniform(0, 0.5)),
        zone_selector_frac=float(rng.uniform(0, 0.3)), n_zones=int(rng.randint(1, 4)))
    s = snapmod.synth(p)
    s.job_priority[:] = rng.choice([0, 0, 100, 1000], size=s.n_jobs).astype(np.int32)
    s.task_priority[:] = rng.choice([1, 1, 5, 9], size=s.n_tasks).astype(np.int32)
    s.job_min_available[:] = np.minimum(s.job_min_available, rng.choice([1, 1, 2, 64], size=s.n_jobs)).astype(np.int32)
    prot = (rng.uniform(size=s.n_tasks) < 0.05).astype(np.uint8)
    s.task_evict_protected = prot if prot.any() else None
    s._check()
    order = [["preempt"], ["reclaim"], ["reclaim", "allocate", "backfill", "preempt"], ["allocate", "preempt", "reclaim"],
             ["preempt", "allocate", "backfill"], ["preempt", "reclaim", "allocate", "preempt"]][seed % 6]
    return conf.load_scheduler_conf(CONF_FULL.format(actions=", ".join(order))), s, order


# evict-only action orders (consecutive evict actions carry the Statement leftovers — sticky NodeNames, Releasing capacity — across)
EVICT_ORDERS = [["preempt"], ["reclaim"], ["preempt", "reclaim"], ["reclaim", "preempt"], ["preempt", "preempt"], ["reclaim", "reclaim", "preempt"]]

# tier layouts that move the victim rules around (session_plugins.go:80-162 decides per tier): no priority rule; drf + gang deciding
# with bin-packing weights; reclaim by conformance, then proportion + gang; the priority rule and gang's JobPipelined switched off
EVICT_CONFS = ["""
actions: "{actions}"
tiers:
- plugins:
  - name: gang
  - name: conformance
- plugins:
  - name: drf
  - name: predicates
  - name: proportion
  - name: nodeorder
""", """
actions: "{actions}"
tiers:
- plugins:
  - name: drf
  - name: gang
- plugins:
  - name: priority
  - name: predicates
  - name: proportion
  - name: nodeorder
    arguments:
      mostrequested.weight: 5
      leastrequested.weight: 0
""", """
actions: "{actions}"
tiers:
- plugins:
  - name: conformance
- plugins:
  - name: proportion
  - name: gang
""", """
actions: "{actions}"
tiers:
- plugins:
  - name: priority
    enablePreemptable: false
  - name: gang
    enableJobPipelined: false
- plugins:
  - name: drf
  - name: predicates
  - name: nodeorder
"""]


def _evict_variant(seed):
    """-> (conf, snapshot, order): the clusters of _evict_case / the adversarial raw snapshots under EVICT_CONFS and EVICT_ORDERS"""
    import rawgen
    ci = seed % len(EVICT_CONFS)
    order = EVICT_ORDERS[(seed // len(EVICT_CONFS) + ci) % len(EVICT_ORDERS)]
    snap = _evict_case(seed)[1] if (seed // 2) % 2 == 0 else rawgen.raw_snapshot(seed)
    return conf.load_scheduler_conf(EVICT_CONFS[ci].format(actions=", ".join(order))), snap, order


@pytest.mark.parametrize("seed", range(120))
def test_python_restatement_equals_c_oracle_with_preempt_and_reclaim(oracle_mod, seed):
    cfg, snap, order = _evict_case(seed)
    _pyref_vs_oracle_evict(oracle_mod, cfg, snap, order, seed)


@pytest.mark.parametrize("seed", range(96))
def test_python_restatement_equals_c_oracle_under_other_tier_layouts(oracle_mod, seed):
    cfg, snap, order = _evict_variant(seed)
    _pyref_vs_oracle_evict(oracle_mod, cfg, snap, order, seed)


def _pyref_vs_oracle_evict(oracle_mod, cfg, snap, order, seed):
    o = oracle_mod.Oracle(cfg, snap)
    p = pyref.Session(_tiers(cfg), snap)
    o_panic = p_panic = False
    try:
        o.run(order)
    except RuntimeError:
        o_panic = True
    try:
        p.run(order)
    except ArithmeticError:
        p_panic = True
    assert o_panic == p_panic, (seed, o_panic, p_panic)
    if o_panic:
        return
    assert [int(t) for t in o.evictions()] == p.evictions, seed
    st, nd = o.task_state()
    assert np.array_equal(np.array(p.status, np.uint8), st), seed
    assert np.array_equal(np.array(p.tnode, np.uint32), nd), seed      # NodeName is sticky: un-pipelined tasks keep it
    idle, rel, nzc, nzm, cnt = o.node_state()
    for n in range(snap.n_nodes):
        for d in range(snap.n_res):
            assert p.idle[n].get(d) == idle[d, n], (seed, n, d)
            assert p.rel[n].get(d) == rel[d, n], (seed, n, d)
    assert np.array_equal(np.array(p.nzc), nzc) and np.array_equal(np.array(p.nzm), nzm) and np.array_equal(np.array(p.podcnt), cnt)
    js, qs, des = o.shares()
    if p._has("drf"):                                                    # the shares exist where the plugin that owns them is configured
        assert np.array_equal(np.array(p.jshare), js)
    if p._has("proportion"):
        for q, a in p.qattr.items():
            assert a["share"] == qs[q], (seed, q)
    pb = np.full(snap.n_tasks, abi.KB_NONE, np.uint32)
    for t, n in p.binds.items():
        pb[t] = n
    assert np.array_equal(pb, o.binds())
    o.close()


def test_discarded_statement_leaves_a_sticky_node_name(oracle_mod):
    """Hand-derived from the Go sources; pins a quirk both restatements must share.  NodeInfo.RemoveTask never clears
    task.NodeName (api/node_info.go:217-243), so a task that a discarded preempt statement un-pipelined (statement.go:152-187)
    keeps its old host: a later AddTask on another node fails (node_info.go:173-176) AFTER ssn.Allocate flipped its status
    (session.go:243), it still counts as ready (job_info.go:383-394) and dispatch binds it to the stale NodeName (session.go:290-297).

    n1, n2: 1 cpu, each full with one running low-priority pod; n3: 1 cpu, empty.  Gang `high` (minMember 3, three pending pods).
    preempt: no plugin scores -> SortNodes = n3, n2, n1.  high0 evicts on n2, high1 on n1, high2 finds nobody -> 2 < 3, statement
    discarded (no eviction reaches the cache).  allocate: only n3 has room; high0 and high1 fail AddTask there (sticky n2 / n1) but
    are Allocated; high2 lands on n3; 3 >= minMember -> all three are dispatched to their NodeName."""
    S = kbm.snapshot
    rl = fixtures.build_resource_list
    tiers = conf.tiers_literal([conf.PluginOption("priority", enabled=abi.EN_PREEMPTABLE | abi.EN_JOB_ORDER),
                                conf.PluginOption("gang", enabled=abi.EN_PREEMPTABLE | abi.EN_JOB_PIPELINED | abi.EN_JOB_READY)])
    snap = S.flatten(
        nodes=[S.Node(f"n{i}", rl("1", "1G")) for i in (1, 2, 3)],
        pods=[fixtures.build_pod("c1", "low1", "n1", "Running", rl("1", "1G"), "low"),
              fixtures.build_pod("c1", "low2", "n2", "Running", rl("1", "1G"), "low")] +
             [fixtures.build_pod("c1", f"high{i}", "", "Pending", rl("1", "1G"), "high") for i in range(3)],
        pod_groups=[S.PodGroup("c1", "low", queue="q1", priority=1), S.PodGroup("c1", "high", queue="q1", min_member=3, priority=10)],
        queues=[S.Queue("q1", 1)])
    expected = {"c1/high0": "n2", "c1/high1": "n1", "c1/high2": "n3"}
    o = oracle_mod.Oracle(tiers, snap)
    o.run(["preempt", "allocate"])
    assert len(o.evictions()) == 0
    assert snap.bind_map(o.binds()) == expected
    idle, rel, _, _, cnt = o.node_state()
    assert idle[0].tolist() == [0.0, 0.0, 0.0] and rel[0].tolist() == [0.0, 0.0, 0.0] and cnt.tolist() == [1, 1, 1]
    p = pyref.Session(_tiers(tiers), snap).run(["preempt", "allocate"])
    assert p.evictions == [] and {snap.task_name(t): snap.node_name(n) for t, n in p.binds.items()} == expected
    assert [x.cpu for x in p.idle] == [0.0, 0.0, 0.0] and p.podcnt == [1, 1, 1]


def test_python_restatement_reference_preempt_and_reclaim_cases():
    """actions/preempt/preempt_test.go:51-131 (1 and 2 evictions) and actions/reclaim/reclaim_test.go:51-99 (1 eviction): the
    counts the reference's FakeEvictor checks, reproduced by the Python restatement on its own."""
    S = kbm.snapshot
    rl, pod = fixtures.build_resource_list, fixtures.build_pod
    tiers = conf.tiers_literal([conf.PluginOption("conformance", enabled=abi.EN_PREEMPTABLE), conf.PluginOption("gang", enabled=abi.EN_PREEMPTABLE)])
    snap = S.flatten(
        nodes=[S.Node("n1", rl("3", "3Gi"))],
        pods=[pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"), pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
              pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg1"), pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg1")],
        pod_groups=[S.PodGroup("c1", "pg1", queue="q1")], queues=[S.Queue("q1", 1)])
    assert len(pyref.Session(_tiers(tiers), snap).run(["preempt"]).evictions) == 1
    snap = S.flatten(
        nodes=[S.Node("n1", rl("2", "2G"))],
        pods=[pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"), pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
              pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2"), pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg2")],
        pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q1")], queues=[S.Queue("q1", 1)])
    assert len(pyref.Session(_tiers(tiers), snap).run(["preempt"]).evictions) == 2
    tiers = conf.tiers_literal([conf.PluginOption("conformance", enabled=abi.EN_RECLAIMABLE), conf.PluginOption("gang", enabled=abi.EN_RECLAIMABLE)])
    snap = S.flatten(
        nodes=[S.Node("n1", rl("3", "3Gi"))],
        pods=[pod("c1", f"preemptee{i}", "n1", "Running", rl("1", "1G"), "pg1") for i in (1, 2, 3)] + [pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2")],
        pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q2")], queues=[S.Queue("q1", 1), S.Queue("q2", 1)])
    assert len(pyref.Session(_tiers(tiers), snap).run(["reclaim"]).evictions) == 1


# ---- adversarial raw snapshots (tests/rawgen.py): epsilon edges, zero capacities, nil maps, tie-breaks
import rawgen  # noqa: E402  (tests/ is on sys.path under pytest rootdir conftest)


def _compare_final_state(seed, snap, o, p):
    st, nd = o.task_state()
    assert np.array_equal(np.array(p.status, np.uint8), st), seed
    assert np.array_equal(np.array(p.tnode, np.uint32), nd), seed
    idle, rel, nzc, nzm, cnt = o.node_state()
    for n in range(snap.n_nodes):
        for d in range(snap.n_res):
            assert p.idle[n].get(d) == idle[d, n], (seed, n, d)
            assert p.rel[n].get(d) == rel[d, n], (seed, n, d)
    assert np.array_equal(np.array(p.nzc), nzc) and np.array_equal(np.array(p.nzm), nzm) and np.array_equal(np.array(p.podcnt), cnt)
    pb = np.full(snap.n_tasks, abi.KB_NONE, np.uint32)
    for t, n in p.binds.items():
        pb[t] = n
    assert np.array_equal(pb, o.binds()), seed


@pytest.mark.parametrize("seed", range(600))
def test_restatements_agree_on_adversarial_snapshots(oracle_mod, seed):
    snap = rawgen.raw_snapshot(seed)
    rng = np.random.RandomState(seed)
    if seed % 3 == 2:
        order = [["preempt"], ["reclaim", "allocate", "backfill", "preempt"], ["preempt", "allocate"]][(seed // 3) % 3]
        cfg = conf.load_scheduler_conf(CONF_FULL.format(actions=", ".join(order)))
    else:
        order = ["allocate", "backfill"]
        wl, wm, wa, wb = [int(x) for x in rng.choice([0, 1, 1, 2, 5], size=4)]
        cfg = conf.load_scheduler_conf(CONF_TMPL.format(wl=wl, wm=wm, wa=wa, wb=wb))
    o_panic = p_panic = False
    o = p = None
    try:
        o = oracle_mod.Oracle(cfg, snap)
        o.run(order)
    except RuntimeError:
        o_panic = True
    try:
        p = pyref.Session(_tiers(cfg), snap)
        p.run(order)
    except ArithmeticError:
        p_panic = True
    assert o_panic == p_panic, (seed, o_panic, p_panic)
    if o_panic:
        return
    if order == ["allocate", "backfill"]:
        assert np.array_equal(np.array(p.decisions, dtype=np.uint32).reshape(-1, 3), o.decisions()), seed
    assert [int(t) for t in o.evictions()] == p.evictions, seed
    _compare_final_state(seed, snap, o, p)
    js, qs, des = o.shares()
    assert np.array_equal(np.array(p.jshare), js), seed
    for q, a in p.qattr.items():
        assert a["share"] == qs[q], (seed, q)
        for d in range(snap.n_res):
            assert a["deserved"].get(d) == des[d, q], (seed, q, d)
    o.close()
