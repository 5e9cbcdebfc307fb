"""Inter-pod (anti)affinity, CPU side: what kube-batch_amd/snapshot.py:build_interpod folds into the kb_interpod tables must give, for
every (pod, node) pair of random clusters in random states, the answers of the object-level restatement of the Go code
(tests/interpod_objref.py): predicate p8 and InterPodAffinityPriority.  The table arithmetic is tests/pyref.py's, the same the
oracle and the engine implement."""
import importlib

import numpy as np
import pytest

import interpod_objref as objref
import pyref

kbm = importlib.import_module("kube-batch_amd")
abi, conf, snapmod = kbm.abi, kbm.conf, kbm.snapshot


def _tiers(cfg):
    out = []
    for tier in cfg.tiers:
        out.append([(po.name, po.enabled, {k: int(v) for k, v in (po.arguments or {}).items() if str(v).lstrip("-").isdigit()}) for po in tier])
    return out


def random_cluster(seed, n_nodes=7, n_pods=22, n_jobs=5, tight=False, pool_size=8, weights=(1, 10, 100), templates=False, narrow=False, n_queues=1):
    """tight: node capacities and pod requests sized so that resources bind too (Pipelines, gangs that do not fit)"""
    rng = np.random.RandomState(4200 + seed)
    zones = ["z0", "z1", "z2"]
    nodes = []
    for i in range(n_nodes):
        labels = {"kubernetes.io/hostname": f"n{i:02d}"}
        if rng.uniform() < 0.85:
            labels["zone"] = zones[rng.randint(len(zones))]
        if rng.uniform() < 0.5:
            labels["rack"] = f"r{rng.randint(2)}"
        alloc = {"cpu": "64", "memory": "256Gi", "pods": "110"}
        if tight:
            alloc = {"cpu": str(int(rng.choice([2, 4, 8]))), "memory": f"{int(rng.choice([8, 16, 32]))}Gi", "pods": str(int(rng.choice([6, 20, 110])))}
        nodes.append(snapmod.Node(name=f"n{i:02d}", allocatable=alloc, labels=labels))
    apps, tiers_, nss = ["a", "b", "c"], ["fe", "be"], ["ns1", "ns2"]
    keys = ["zone", "kubernetes.io/hostname", "rack"]

    def selector():
        if narrow:                                                         # terms that name one workload, like real templates do
            return ((("app", f"a{rng.randint(n_jobs)}"),), ())
        r = rng.uniform()
        if r < 0.08:
            return None                                                  # nil selector: matches nothing
        if r < 0.16:
            return ((), ())                                              # empty selector: matches everything
        ml, ex = [], []
        if rng.uniform() < 0.6:
            ml.append(("app", apps[rng.randint(3)]))
        if rng.uniform() < 0.4 or not ml:
            op = ["In", "NotIn", "Exists", "DoesNotExist"][rng.randint(4)]
            vals = tuple(sorted(set(rng.choice(apps if rng.uniform() < 0.5 else tiers_, size=rng.randint(1, 3))))) if op in ("In", "NotIn") else ()
            ex.append(("app" if vals and vals[0] in apps else "tier", op, vals))
        return (tuple(ml), tuple(ex))

    def fresh_term(allow_empty_key=False):
        ns = () if rng.uniform() < 0.6 else tuple(sorted(set(rng.choice(nss, size=rng.randint(1, 3)))))
        key = keys[rng.randint(3)] if not (allow_empty_key and rng.uniform() < 0.1) else ""
        return (ns, selector(), key)

    pool = [fresh_term() for _ in range(pool_size)] if tight else None            # workloads share a handful of terms (deployment templates)

    def term(allow_empty_key=False):
        return pool[rng.randint(len(pool))] if pool else fresh_term(allow_empty_key)

    pods, groups = [], []
    job_spec = {}
    for j in range(n_jobs):
        groups.append(snapmod.PodGroup(namespace=nss[j % 2], name=f"pg{j}", min_member=int(rng.randint(1, 4)) if tight else 1,
                                       queue="default" if n_queues == 1 else f"q{j % n_queues}",
                                       creation=j, priority=int(rng.randint(0, 3)) if tight else 0))
    for i in range(n_pods):
        j = rng.randint(n_jobs + 1)                                       # n_jobs: a pod outside the session
        ns = nss[j % 2] if j < n_jobs else nss[rng.randint(2)]
        labels = {}
        if narrow:
            labels["app"] = f"a{j}"
        elif rng.uniform() < 0.9:
            labels["app"] = apps[rng.randint(3)]
        if rng.uniform() < 0.5:
            labels["tier"] = tiers_[rng.randint(2)]
        req = {"cpu": "100m", "memory": "128Mi"}
        if tight:
            req = {"cpu": f"{int(rng.choice([250, 500, 1000, 2000]))}m", "memory": f"{int(rng.choice([512, 1024, 4096]))}Mi"}
            if rng.uniform() < 0.08:
                req = {}                                                   # BestEffort: backfill places it
        p = snapmod.Pod(namespace=ns, name=f"p{i:03d}", containers=[req],
                        group_name=f"pg{j}" if j < n_jobs else "", labels=labels, creation=i)
        def draw_spec():
            r = rng.uniform()
            spec = [[], [], [], []]
            if r < 0.30:
                spec[0] = [term() for _ in range(rng.randint(1, 3))]
            if 0.2 < r < 0.45:
                spec[1] = [term() for _ in range(rng.randint(1, 3))]
            if rng.uniform() < 0.3:
                spec[2] = [(int(rng.choice(list(weights))), term()) for _ in range(rng.randint(1, 3))]
            if rng.uniform() < 0.3:
                spec[3] = [(int(rng.choice(list(weights))), term(True)) for _ in range(rng.randint(1, 3))]
            return spec
        if templates:                                                     # the pods of a job share one template, like a Deployment's
            if j not in job_spec:
                job_spec[j] = draw_spec() if rng.uniform() < 0.4 else [[], [], [], []]
            spec = job_spec[j]
        else:
            spec = draw_spec()
        p.pod_anti_affinity_required, p.pod_affinity_required, p.pod_affinity_preferred, p.pod_anti_affinity_preferred = (list(x) for x in spec)
        placed = rng.uniform() < (0.35 if j < n_jobs else 1.0)
        if placed:
            if tight:
                p.containers = [{"cpu": "50m", "memory": "64Mi"}]           # what already runs must fit (an overfull node is outside the envelope)
            p.node_name = nodes[rng.randint(n_nodes)].name
            p.phase = "Running" if rng.uniform() < 0.7 else "Pending"     # Pending + nodeName = Bound
            if p.phase == "Pending" and rng.uniform() < 0.5:
                p.spec_node_name_empty = True                              # the cache's Binding: Spec.NodeName not written yet
        pods.append(p)
    if n_queues > 1:                                                       # several queues with different weights: reclaim has something to take
        return nodes, pods, groups, [snapmod.Queue(name=f"q{i}", weight=1 + 2 * i) for i in range(n_queues)]
    return nodes, pods, groups, [snapmod.Queue(name="default")]


@pytest.mark.parametrize("seed", range(40))
def test_tables_agree_with_the_object_level_restatement(seed):
    nodes, pods, groups, queues = random_cluster(seed)
    try:
        snap = snapmod.flatten(nodes, pods, groups, queues)
    except snapmod.UnsupportedSnapshot as e:
        pytest.skip(str(e))
    if snap.interpod is None:
        pytest.skip("no pod-affinity term drawn")
    S = pyref.Session(_tiers(conf.load_scheduler_conf()), snap)
    by_name = {f"{p.namespace}/{p.name}": p for p in pods}
    task_pod = [by_name[nm] for nm in snap.names["tasks"]]
    nodes_sorted = sorted(nodes, key=lambda n: n.name)
    status_name = {abi.TASK_PENDING: "Pending", abi.TASK_BOUND: "Bound", abi.TASK_RUNNING: "Running", abi.TASK_RELEASING: "Releasing",
                   abi.TASK_BINDING: "Binding"}
    states, task_state = [], {}
    for t, p in enumerate(task_pod):
        n = int(snap.task_node[t])
        st = objref.PodState(p, nodes_sorted[n].name if n != abi.KB_NONE else None, status_name[int(snap.task_status[t])],
                             "" if (p.spec_node_name_empty or n == abi.KB_NONE) else p.node_name)
        states.append(st); task_state[t] = st
    session_names = set(snap.names["tasks"])
    for p in pods:
        if f"{p.namespace}/{p.name}" not in session_names and p.node_name:
            st = objref.PodState(p, p.node_name, "Running", "" if p.spec_node_name_empty else p.node_name)
            st.in_session = False
            states.append(st)
    W = objref.World(nodes, states)
    rng = np.random.RandomState(seed)
    pending = [t for t in range(snap.n_tasks) if int(snap.task_status[t]) == abi.TASK_PENDING]
    rng.shuffle(pending)
    checked = 0
    for t in pending:
        pod = task_pod[t]
        ok_tab = [S.interpod_predicate(t, n) for n in range(snap.n_nodes)]
        ok_obj = [objref.predicate(W, pod, nodes_sorted[n]) for n in range(snap.n_nodes)]
        assert ok_tab == ok_obj, (seed, snap.names["tasks"][t], ok_tab, ok_obj)
        feasible = [n for n in range(snap.n_nodes) if ok_tab[n] and rng.uniform() < 0.8]
        if feasible:
            sc_tab = S.interpod_scores(t, feasible)
            sc_obj = objref.priority(W, pod, [nodes_sorted[n] for n in feasible])
            assert {nodes_sorted[n].name: v for n, v in sc_tab.items()} == sc_obj, (seed, snap.names["tasks"][t])
            checked += 1
            n = feasible[rng.randint(len(feasible))]
            kind = "Allocated" if rng.uniform() < 0.75 else "Pipelined"
            S.status[t] = pyref.ALLOCATED if kind == "Allocated" else pyref.PIPELINED
            S.tnode[t], S.onnode[t] = n, True
            S.ip_added[n].add(t)
            task_state[t].node, task_state[t].status, task_state[t].spec_node_name = nodes_sorted[n].name, kind, ""
    assert checked > 0


def test_open_counts_equal_the_task_statuses():
    """ctr_count / ctr_total of the tables (what the oracle and the engine start from) == the from-scratch count over the statuses"""
    for seed in range(40):
        nodes, pods, groups, queues = random_cluster(seed)
        try:
            snap = snapmod.flatten(nodes, pods, groups, queues)
        except snapmod.UnsupportedSnapshot:
            continue
        ip = snap.interpod
        if ip is None:
            continue
        S = pyref.Session(_tiers(conf.load_scheduler_conf()), snap)
        for c in range(ip["n_counters"]):
            assert int(ip["ctr_total"][c]) == S._ip_count(c, None)
            for d in range(ip["n_domains"]):
                assert int(ip["ctr_count"][c][d]) == S._ip_count(c, d), (seed, c, d)
