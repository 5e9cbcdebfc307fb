"""Inter-pod (anti)affinity through the whole allocate + backfill loop: the C oracle (incremental counts) against tests/pyref.py (counts
recomputed from the task statuses on every call) on random clusters where resources bind too — decisions, binds, node state, statuses."""
import importlib

import numpy as np
import pytest

import pyref
from test_interpod_cpu import random_cluster, _tiers

kbm = importlib.import_module("kube-batch_amd")
abi, conf, snapmod = kbm.abi, kbm.conf, kbm.snapshot

CONFS = [None, """
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
      podaffinity.weight: 3
      leastrequested.weight: 0
      mostrequested.weight: 2
"""]


def interpod_case(seed, wide=False):
    """wide: many distinct terms, so that the per-task masks of kb_interpod span several 64-bit words"""
    rng = np.random.RandomState(900 + seed)
    if wide:
        nodes, pods, groups, queues = random_cluster(3000 + seed, n_nodes=int(rng.randint(8, 20)), n_pods=int(rng.randint(150, 260)),
                                                     n_jobs=int(rng.randint(8, 16)), tight=True, pool_size=70)
    else:
        nodes, pods, groups, queues = random_cluster(1000 + seed, n_nodes=int(rng.randint(5, 16)), n_pods=int(rng.randint(30, 90)),
                                                     n_jobs=int(rng.randint(4, 12)), tight=True)
    snap = snapmod.flatten(nodes, pods, groups, queues)
    if seed % 3 == 0:                                                      # capacity that is being released: Pipeline decisions
        N = snap.n_nodes
        rel = rng.uniform(size=N) < 0.5
        snap.node_releasing[0] = np.where(rel, rng.choice([500, 1000, 4000], size=N), 0).astype(np.float64)
        snap.node_releasing[1] = np.where(rel, rng.choice([1, 4, 16], size=N) * float(1 << 30), 0)
    return conf.load_scheduler_conf(CONFS[seed % 2]), snap


def very_wide_interpod_case(n_pods=1100):
    """more than 2 x 1024 distinct predicate counters AND priority classes (round 3's envelope stopped at 1024 of each): every pod repels the
    pods of one label value of its own and prefers those of another, by host name; a third of the pods already run"""
    rng = np.random.RandomState(77)
    nodes = [snapmod.Node(name=f"n{i:02d}", allocatable={"cpu": "64", "memory": "256Gi", "pods": "110"},
                          labels={"kubernetes.io/hostname": f"n{i:02d}", "zone": f"z{i % 3}"}) for i in range(48)]
    groups = [snapmod.PodGroup(namespace="ns1", name=f"pg{j}", min_member=1, queue="default", creation=j) for j in range(40)]
    pods = []
    for i in range(n_pods):
        p = snapmod.Pod(namespace="ns1", name=f"p{i:04d}", containers=[{"cpu": "100m", "memory": "128Mi"}], group_name=f"pg{i % 40}",
                        labels={"app": f"u{(i * 7 + 3) % n_pods}", "tier": f"t{i % 5}"}, creation=i)
        p.pod_anti_affinity_required = [((), ((("app", f"u{i}"),), ()), "kubernetes.io/hostname")]
        p.pod_affinity_preferred = [(int(rng.choice([1, 10, 100])), ((), ((("app", f"u{(i + 1) % n_pods}"),), ()), "zone" if i % 2 else "kubernetes.io/hostname"))]
        if i % 3 == 0:
            p.node_name = nodes[int(rng.randint(len(nodes)))].name
            p.phase = "Running"
        pods.append(p)
    snap = snapmod.flatten(nodes, pods, groups, [snapmod.Queue(name="default")])
    assert snap.interpod["n_counters"] > 2048 and snap.interpod["n_classes"] > 2048
    return conf.load_scheduler_conf(CONFS[1]), snap


def test_oracle_equals_pyref_beyond_1024_counters_and_classes(oracle_mod):
    """the oracle at that width, held to the second restatement (tests/test_gpu_interpod.py holds the engine to the oracle)"""
    cfg, snap = very_wide_interpod_case()
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate", "backfill"])
    p = pyref.Session(_tiers(cfg), snap).run(["allocate", "backfill"])
    pd = np.array(p.decisions, dtype=np.uint32).reshape(-1, 3)
    assert pd.shape == o.decisions().shape and np.array_equal(pd, o.decisions())
    assert len(pd) > 500


@pytest.mark.parametrize("seed", list(range(60)) + [1000 + i for i in range(6)])
def test_oracle_equals_pyref_with_interpod_affinity(oracle_mod, seed):
    try:
        cfg, snap = interpod_case(seed % 1000, wide=seed >= 1000)
        if seed >= 1000:
            assert snap.interpod["n_counters"] > 64 and snap.interpod["n_classes"] > 64   # multi-word masks
    except snapmod.UnsupportedSnapshot as e:
        pytest.skip(str(e))
    if snap.interpod is None:
        pytest.skip("no pod-affinity term drawn")
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
            assert p.idle[n].get(d) == idle[d, n] and p.rel[n].get(d) == rel[d, n], (seed, n, d)
    st, nd = o.task_state()
    assert np.array_equal(np.array(p.status, np.uint8), st)
    assert o.popped == p.popped
    o.close()


def test_interpod_cases_place_pods_and_refuse_nodes(oracle_mod):
    """the generator must reach the interesting regime: decisions exist, and the predicate removes nodes"""
    placed = refused = 0
    for seed in range(20):
        try:
            cfg, snap = interpod_case(seed)
        except snapmod.UnsupportedSnapshot:
            continue
        if snap.interpod is None:
            continue
        o = oracle_mod.Oracle(cfg, snap)
        o.run(["allocate", "backfill"])
        placed += len(o.decisions())
        S = pyref.Session(_tiers(cfg), snap)
        refused += sum(1 for t in range(snap.n_tasks) for n in range(snap.n_nodes) if not S.interpod_predicate(t, n))
        o.close()
    assert placed > 200 and refused > 500, (placed, refused)


# ---- preempt / reclaim in sessions with inter-pod terms (round 3) -------------------------------------------------------------------
# An eviction takes its victim OUT of the predicate's pod list (Releasing is not an allocated status) while it stays in ni.Tasks, a Pipeline
# adds the preemptor to ni.Tasks with an empty Spec.NodeName, a discarded statement undoes both.  The C oracle keeps the kb_interpod counts
# incrementally through all of it; tests/pyref.py recounts them from the task statuses at every predicate / priority call (the PodLister's own
# way): two independent routes to the same journal.
EVICT_ORDERS_IP = [["preempt"], ["reclaim"], ["allocate", "preempt"], ["reclaim", "allocate", "backfill", "preempt"], ["preempt", "reclaim"], ["preempt", "preempt"]]


def interpod_evict_case(seed, objects=False):
    rng = np.random.RandomState(7700 + seed)
    nodes, pods, groups, queues = random_cluster(5000 + seed, n_nodes=int(rng.randint(3, 10)), n_pods=int(rng.randint(30, 110)), n_jobs=int(rng.randint(4, 10)),
                                                 tight=True, n_queues=int(rng.choice([1, 2, 3])))
    room = {n.name: [snapmod.quantity_milli_value(n.allocatable["cpu"]), snapmod.quantity_value(n.allocatable["memory"]) // (1 << 20), int(n.allocatable["pods"])] for n in nodes}
    for p in pods:                                                         # what already runs (50m / 64Mi each) ...
        if p.node_name:
            r = room[p.node_name]; r[0] -= 50; r[1] -= 64; r[2] -= 1
    for p in pods:                                                         # ... made worth taking where the node has the room: victims hold real capacity
        if p.node_name and p.group_name and rng.uniform() < 0.8:
            c, m = int(rng.choice([500, 1000, 2000])), int(rng.choice([512, 1024]))
            r = room[p.node_name]
            if r[0] - c >= 100 and r[1] - m >= 100 and r[2] >= 0:
                r[0] -= c - 50; r[1] -= m - 64
                p.containers = [{"cpu": f"{c}m", "memory": f"{m}Mi"}]
    for p in pods:                                                         # victims that keep others away: while such a pod is in an allocated status no pod
        if p.node_name and p.group_name and p.phase == "Running" and rng.uniform() < 0.35:   # with the label may join its node (or zone); evicted, it no longer counts
            p.pod_anti_affinity_required = [((), ((("app", ["a", "b", "c"][rng.randint(3)]),), ()), ["kubernetes.io/hostname", "zone"][rng.randint(2)])]
    snap = snapmod.flatten(nodes, pods, groups, queues)
    order = EVICT_ORDERS_IP[seed % len(EVICT_ORDERS_IP)]
    text = (CONFS[seed % 2] or conf.DEFAULT_SCHEDULER_CONF).replace('actions: "allocate, backfill"', 'actions: "%s"' % ", ".join(order))
    if objects:
        return conf.load_scheduler_conf(text), snap, order, nodes, pods
    return conf.load_scheduler_conf(text), snap, order


@pytest.mark.parametrize("seed", range(150))
def test_oracle_equals_pyref_with_interpod_affinity_under_preempt_and_reclaim(oracle_mod, seed):
    from test_pyref_vs_oracle import _pyref_vs_oracle_evict
    try:
        cfg, snap, order = interpod_evict_case(seed)
    except (snapmod.UnsupportedSnapshot, ValueError) as e:
        pytest.skip(str(e))
    if snap.interpod is None:
        pytest.skip("no pod-affinity term drawn")
    _pyref_vs_oracle_evict(oracle_mod, cfg, snap, order, seed)   # evictions in order, task statuses and (sticky) node names, node state, shares, binds


@pytest.mark.parametrize("seed", range(150))
def test_predicate_after_evict_actions_equals_the_object_level_answer(oracle_mod, seed):
    """Oracle, pyref and engine agree with each other above — but all three read the kb_interpod TABLES.  Here the session the oracle leaves behind
    after its evict actions (who is Releasing, who was pipelined where, who went back to Pending) is rebuilt as Kubernetes-shaped objects, and the
    inter-pod predicate of every still-Pending task on every node is asked of tests/interpod_objref.py (labels, selectors, namespaces, topology
    keys: no table) and compared with what the oracle's live counts answer: an eviction that left the counts alone, a Pipeline booked as allocated,
    a discard that did not put a victim back would show here."""
    import interpod_objref as objref
    cfg, snap, order, nodes, pods = interpod_evict_case(seed, objects=True)
    if snap.interpod is None:
        pytest.skip("no pod-affinity term drawn")
    nbits = snap.n_task_classes * snap.n_node_classes
    all_compatible = all((snap.class_compat[b >> 3] >> (b & 7)) & 1 for b in range(nbits))
    if snap.node_ports is not None or not all_compatible:
        pytest.skip("other plugin predicates in play")           # random_cluster draws none: the mask below is pod cap + inter-pod only
    o = oracle_mod.Oracle(cfg, snap)
    try:
        o.run(order)
    except RuntimeError:
        pytest.skip("the reference panics on this snapshot")
    st, nd = o.task_state()
    _, _, _, _, cnt = o.node_state()
    nodes_sorted = sorted(nodes, key=lambda n: n.name)
    by_name = {f"{p.namespace}/{p.name}": p for p in pods}
    status_name = {abi.TASK_PENDING: "Pending", abi.TASK_ALLOCATED: "Allocated", abi.TASK_PIPELINED: "Pipelined", abi.TASK_BINDING: "Binding",
                   abi.TASK_BOUND: "Bound", abi.TASK_RUNNING: "Running", abi.TASK_RELEASING: "Releasing"}
    states = []
    for t, name in enumerate(snap.names["tasks"]):
        p = by_name[name]
        on = int(st[t]) != abi.TASK_PENDING and int(nd[t]) != abi.KB_NONE        # an un-pipelined task keeps its NodeName but is in no ni.Tasks
        opened_on_node = int(snap.task_node[t]) != abi.KB_NONE
        spec = (p.node_name if not p.spec_node_name_empty else "") if opened_on_node else ""   # placed by this session: Spec.NodeName still empty
        states.append(objref.PodState(p, nodes_sorted[int(nd[t])].name if on else None, status_name[int(st[t])], spec))
    session = set(snap.names["tasks"])
    for p in pods:
        if f"{p.namespace}/{p.name}" not in session and p.node_name:
            s_ = objref.PodState(p, p.node_name, "Running", "" if p.spec_node_name_empty else p.node_name)
            s_.in_session = False
            states.append(s_)
    W = objref.World(nodes, states)
    pending = [t for t in range(snap.n_tasks) if int(st[t]) == abi.TASK_PENDING]
    if not pending:
        pytest.skip("nothing left pending")
    mask, _ = o.eval_matrix(0, snap.n_tasks, 0)                                  # plugin predicates only, against the session as it stands
    for t in pending:
        pod = by_name[snap.names["tasks"][t]]
        for n, node in enumerate(nodes_sorted):
            want = bool(snap.node_max_pods[n] > cnt[n]) and objref.predicate(W, pod, node)
            got = bool((mask[t, n >> 3] >> (n & 7)) & 1)
            assert got == want, (seed, order, snap.names["tasks"][t], node.name, got, want)
    o.close()
