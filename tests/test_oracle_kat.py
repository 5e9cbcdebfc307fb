"""Pins the CPU oracle against the reference's own known-answer tests.

Each test names the reference test whose table it restates (paths relative to /root/reference).
Nothing here reads /root/reference at run time.
"""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

kbm = importlib.import_module("kube-batch_amd")
abi = kbm.abi
fixtures = importlib.import_module("kube-batch_amd.fixtures")
snapmod = kbm.snapshot
conf = kbm.conf

S1, HP = 2, 3   # dims of "scalar.test/scalar1" and "hugepages-test"


@pytest.fixture(scope="module")
def L(oracle_mod):
    lib = oracle_mod.lib()
    lib.kbo_set_dims(4)
    return lib


def R(oracle_mod, cpu=0.0, mem=0.0, **sc):
    m = {}
    if "s1" in sc:
        m[S1] = sc["s1"]
    if "hp" in sc:
        m[HP] = sc["hp"]
    return oracle_mod.OracleRes.make(cpu, mem, m)


def tup(r):
    return r.as_tuple(4)


# ---- pkg/scheduler/api/resource_info_test.go:27-58 TestNewResource (through the flattener's NewResource) ----
def test_new_resource_milli_scaling():
    v, mask, _ = snapmod._resource({"cpu": "4m", "memory": "2000", "scalar.test/scalar1": "1", "hugepages-test": "2"},
                                   {"hugepages-test": 2, "scalar.test/scalar1": 3}, 4)
    assert v.tolist() == [4.0, 2000.0, 2000.0, 1000.0] and mask == 0b11
    v, mask, mt = snapmod._resource({}, {}, 2)
    assert v.tolist() == [0.0, 0.0] and mask == 0 and mt == 0


# ---- resource_info_test.go:60-100 TestResourceAddScalar is SetScalar on a map: covered by OracleRes.make ----

# ---- resource_info_test.go:102-145 TestSetMaxResource ----
def test_set_max_resource(L, oracle_mod):
    r1, r2 = R(oracle_mod), R(oracle_mod, 4000, 2000, s1=1, hp=2)
    L.kbo_res_set_max(C.byref(r1), C.byref(r2))
    assert tup(r1) == (4000, 2000, {S1: 1, HP: 2})
    r1, r2 = R(oracle_mod, 4000, 4000, s1=1, hp=2), R(oracle_mod, 4000, 2000, s1=4, hp=5)
    L.kbo_res_set_max(C.byref(r1), C.byref(r2))
    assert tup(r1) == (4000, 4000, {S1: 4, HP: 5})


# ---- resource_info_test.go:147-186 TestIsZero ----
def test_is_zero(L, oracle_mod):
    assert L.kbo_res_is_zero(C.byref(R(oracle_mod)), 0) == 1
    full = R(oracle_mod, 4000, 4000, s1=4, hp=5)
    assert L.kbo_res_is_zero(C.byref(full), 0) == 0
    assert L.kbo_res_is_zero(C.byref(full), S1) == 1


# ---- resource_info_test.go:188-244 TestAddResource ----
def test_add_resource(L, oracle_mod):
    r1, r2 = R(oracle_mod), R(oracle_mod, 4000, 2000, s1=1, hp=2)
    L.kbo_res_add(C.byref(r1), C.byref(r2))
    assert tup(r1) == (4000, 2000, {S1: 1, HP: 2})
    r1, r2 = R(oracle_mod, 4000, 4000, s1=1, hp=2), R(oracle_mod, 4000, 2000, s1=4, hp=5)
    L.kbo_res_add(C.byref(r1), C.byref(r2))
    assert tup(r1) == (8000, 6000, {S1: 5, HP: 7})
    r1, r2 = R(oracle_mod, 4000, 4000, s1=1), R(oracle_mod, 4000, 2000, s1=4, hp=5)
    L.kbo_res_add(C.byref(r1), C.byref(r2))
    assert tup(r1) == (8000, 6000, {S1: 5, HP: 5})


# ---- resource_info_test.go:246-304 TestLessEqual ----
def test_less_equal(L, oracle_mod):
    cases = [
        (R(oracle_mod), R(oracle_mod, 4000, 2000, s1=1000, hp=2000), True),
        (R(oracle_mod, 4000, 4000, s1=1000, hp=2000), R(oracle_mod, 2000, 2000, s1=4000, hp=5000), False),
        (R(oracle_mod, 4, 4000, s1=1), R(oracle_mod), True),
        (R(oracle_mod, 4000, 4000, s1=1000, hp=2000), R(oracle_mod, 8000, 8000, s1=4000, hp=5000), True),
    ]
    for a, b, exp in cases:
        assert bool(L.kbo_res_less_equal(C.byref(a), C.byref(b))) is exp


# ---- resource_info_test.go:306-350 TestSubResource ----
def test_sub_resource(L, oracle_mod):
    r1, r2 = R(oracle_mod, 4000, 2000, s1=1, hp=2), R(oracle_mod)
    assert L.kbo_res_sub(C.byref(r1), C.byref(r2)) == 0
    assert tup(r1) == (4000, 2000, {S1: 1, HP: 2})
    r1, r2 = R(oracle_mod, 4000, 4000, s1=1000, hp=2000), R(oracle_mod, 3000, 2000, s1=500, hp=1000)
    assert L.kbo_res_sub(C.byref(r1), C.byref(r2)) == 0
    assert tup(r1) == (1000, 2000, {S1: 500, HP: 1000})
    # resource_info.go:158: Sub panics unless rr.LessEqual(r)
    r1, r2 = R(oracle_mod, 1000, 1000), R(oracle_mod, 3000, 1000)
    assert L.kbo_res_sub(C.byref(r1), C.byref(r2)) == -100


# ---- resource_info_test.go:352-419 TestLess ----
def test_less(L, oracle_mod):
    cases = [
        (R(oracle_mod), R(oracle_mod), False),
        (R(oracle_mod), R(oracle_mod, 4000, 2000, s1=1000, hp=2000), True),
        (R(oracle_mod, 4000, 4000, s1=1000, hp=2000), R(oracle_mod, 8000, 8000, s1=4000, hp=5000), True),
        (R(oracle_mod, 4000, 4000, s1=5000, hp=2000), R(oracle_mod, 8000, 8000, s1=4000, hp=5000), False),
        (R(oracle_mod, 9000, 4000, s1=1000, hp=2000), R(oracle_mod, 8000, 8000, s1=4000, hp=5000), False),
    ]
    for a, b, exp in cases:
        assert bool(L.kbo_res_less(C.byref(a), C.byref(b))) is exp


# ---- pkg/scheduler/api/node_info_test.go:35-105 TestNodeInfo_AddPod (through the flattener's AddTask) ----
def test_node_info_add_pod():
    G = 10**9
    snap = snapmod.flatten(
        nodes=[snapmod.Node("n1", {"cpu": "8000m", "memory": "10G"})],
        pods=[snapmod.Pod("c1", "p1", [{"cpu": "1000m", "memory": "1G"}], node_name="n1", phase="Running"),
              snapmod.Pod("c1", "p2", [{"cpu": "2000m", "memory": "2G"}], node_name="n1", phase="Running")],
        pod_groups=[], queues=[])
    assert snap.node_idle[:, 0].tolist() == [5000.0, 7.0 * G]
    assert snap.node_allocatable[:, 0].tolist() == [8000.0, 10.0 * G]
    assert snap.node_pod_cnt[0] == 2
    # "add 1 unknown pod": 1000m/2G on a 2000m/1G node -> AddTask errors, node unchanged
    snap = snapmod.flatten(
        nodes=[snapmod.Node("n2", {"cpu": "2000m", "memory": "1G"})],
        pods=[snapmod.Pod("c2", "p1", [{"cpu": "1000m", "memory": "2G"}], node_name="n2", phase="Unknown")],
        pod_groups=[], queues=[])
    assert snap.node_idle[:, 0].tolist() == [2000.0, 1.0 * G] and snap.node_pod_cnt[0] == 0


# ---- pkg/scheduler/api/pod_info_test.go:26-100 TestGetPodResourceRequest (init-container max rule) ----
def test_pod_resource_request_init_containers():
    G = 10**9
    snap = snapmod.flatten(
        nodes=[snapmod.Node("n1", {"cpu": "64", "memory": "64G", "pods": "10"})],
        pods=[snapmod.Pod("ns", "a", [{"cpu": "1000m", "memory": "1G"}, {"cpu": "2000m", "memory": "1G"}], group_name="g"),
              snapmod.Pod("ns", "b", [{"cpu": "1000m", "memory": "1G"}, {"cpu": "2000m", "memory": "1G"}], group_name="g",
                          init_containers=[{"cpu": "2000m", "memory": "5G"}, {"cpu": "2000m", "memory": "1G"}])],
        pod_groups=[snapmod.PodGroup("ns", "g")], queues=[snapmod.Queue("default")])
    assert snap.task_init_resreq[:, 0].tolist() == [3000.0, 2.0 * G]
    assert snap.task_init_resreq[:, 1].tolist() == [3000.0, 5.0 * G]
    assert snap.task_resreq[:, 1].tolist() == [3000.0, 2.0 * G]        # pod_info_test.go:102-162 (without init containers)


# ---- pkg/scheduler/util/scheduler_helper_test.go:24-92 TestSelectBestNode ----
def test_select_best_node(L):
    for scores, expected in (([1.0, 1.0, 2.0, 2.0], {2, 3}), ([1.0, 1.0, 3.0, 2.0, 2.0], {2})):
        arr = (C.c_double * len(scores))(*scores)
        idx = (C.c_int * len(scores))()
        n = C.c_int()
        best = L.kbo_select_best(arr, len(scores), idx, C.byref(n))
        assert set(idx[: n.value]) == expected and best in expected
        assert best == min(expected)        # canonical tie-break


# ---- pkg/scheduler/actions/allocate/allocate_test.go:38-212 TestAllocate ----
@pytest.mark.parametrize("case", range(2))
def test_allocate_reference_cases(oracle_mod, case):
    name, snap, expected = fixtures.allocate_cases()[case]
    o = oracle_mod.Oracle(fixtures.allocate_test_tiers(), snap)
    o.allocate()
    assert snap.bind_map(o.binds()) == expected, name


# ---- doc/usage/tutorial.md:297-330 proportion worked example ----
def test_proportion_tutorial_example(oracle_mod):
    Gi = 1 << 30
    pods = []
    for i in range(5):
        pods.append(snapmod.Pod("q1", f"p{i}", [{"cpu": "1", "memory": "2Gi"}], group_name="j1"))
    for i in range(10):
        pods.append(snapmod.Pod("q2", f"p{i}", [{"cpu": "1", "memory": "2Gi"}], group_name="j2"))
    snap = snapmod.flatten(
        nodes=[snapmod.Node("n1", {"cpu": "6", "memory": "15Gi", "pods": "110"}),
               snapmod.Node("n2", {"cpu": "3", "memory": "12Gi", "pods": "110"})],
        pods=pods,
        pod_groups=[snapmod.PodGroup("q1", "j1", queue="queue1"), snapmod.PodGroup("q2", "j2", queue="queue2")],
        queues=[snapmod.Queue("queue1", 2), snapmod.Queue("queue2", 4)])
    o = oracle_mod.Oracle(conf.load_scheduler_conf(), snap)
    _, _, des = o.shares()
    assert des[:, 0].tolist() == [3000.0, 9.0 * Gi]
    assert des[:, 1].tolist() == [6000.0, 18.0 * Gi]


# ---- SURVEY.md §8a hand-derived KAT from the vendored formulas (least_requested.go:31-58, most_requested.go:30-61,
#      balanced_resource_allocation.go:35-79, non_zero.go:32-37) ----
def test_scorer_hand_kat(L):
    least, most, bal = C.c_int64(), C.c_int64(), C.c_int64()
    ac, am = 4000, 8 << 30
    L.kbo_scorers(1000, ac, 200 << 20, am, C.byref(least), C.byref(most), C.byref(bal))
    assert (least.value, most.value, bal.value) == (8, 1, 7)
    L.kbo_scorers(2000, ac, 400 << 20, am, C.byref(least), C.byref(most), C.byref(bal))
    assert (least.value, bal.value) == (7, 5)
    # capacity 0 and requested > capacity edges
    L.kbo_scorers(10, 0, 10, 0, C.byref(least), C.byref(most), C.byref(bal))
    assert (least.value, most.value, bal.value) == (0, 0, 0)
    L.kbo_scorers(5000, 4000, 1, am, C.byref(least), C.byref(most), C.byref(bal))
    assert (least.value, most.value, bal.value) == (4, 0, 0)     # cpu side 0, mem side 9/0; balanced 0 (fraction >= 1)


# ---- BASELINE config 1: example/job.yaml on 3 nodes spreads 2/2/2 (SURVEY.md §8a KAT) ----
def test_example_job_spread(oracle_mod):
    cfg, snap = fixtures.example_job()
    o = oracle_mod.Oracle(cfg, snap)
    o.allocate()
    binds = o.binds()
    assert (binds != abi.KB_NONE).sum() == 6
    assert sorted(np.bincount(binds, minlength=3).tolist()) == [2, 2, 2]
    dec = o.decisions()
    # first pod -> first node at score 15, second pod must leave it (12 < 15)
    assert dec[0, 1] == 0 and dec[1, 1] == 1 and dec[2, 1] == 2
    assert o.evals == 6 * 3


def _go_heap_reference(keys, pushes):
    """container/heap restated independently in Python (go1.13 src/container/heap/heap.go) for the sift KAT."""
    items = []

    def less(i, j):
        return keys[items[i]] < keys[items[j]]

    def up(j):
        while True:
            i = (j - 1) // 2 if j > 0 else 0
            if i == j or not less(j, i):
                break
            items[i], items[j] = items[j], items[i]
            j = i

    def down(i0, n):
        i = i0
        while True:
            j1 = 2 * i + 1
            if j1 >= n or j1 < 0:
                break
            j = j1
            if j1 + 1 < n and less(j1 + 1, j1):
                j = j1 + 1
            if not less(j, i):
                break
            items[i], items[j] = items[j], items[i]
            i = j

    for x in pushes:
        items.append(x)
        up(len(items) - 1)
    out = []
    while items:
        n = len(items) - 1
        items[0], items[n] = items[n], items[0]
        down(0, n)
        out.append(items.pop())
    return out


def test_heap_mechanics_with_duplicate_keys(L):
    rng = np.random.RandomState(7)
    for n in (1, 2, 3, 7, 64, 257):
        keys = rng.randint(0, 5, size=n).astype(np.float64)     # many ties: pop order is decided by sift mechanics
        pushes = rng.permutation(n).astype(np.uint32)
        out = np.empty(n, np.uint32)
        L.kbo_heap_order(keys.ctypes.data_as(C.POINTER(C.c_double)), pushes.ctypes.data_as(C.POINTER(C.c_uint32)), n,
                         out.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert out.tolist() == _go_heap_reference(keys, pushes.tolist())
        assert np.all(np.diff(keys[out]) >= 0)


def test_node_affinity_priority_map_and_reduce(oracle_mod):
    """nodeorder's NodeAffinity config (vendor/.../priorities/node_affinity.go:34-77 + reduce.go:28-63), hand-derived from the
    vendored formulas (no reference test pins it): counts a=8 (5 for zone In [x] + 3 for gen Gt 2), b=2, c=2 (zone NotIn [x];
    the term without expressions selects nothing, the weight-0 term is skipped); NormalizeReduce(10) over the feasible nodes
    gives 10, 2, 2; on top of the 15 of the empty 4-cpu / 8-GiB node KAT -> 25, 17, 17.  With node a unschedulable the maximum
    over the FEASIBLE set drops to 2 -> b and c get the full 10."""
    S = kbm.snapshot

    def mk(unsched):
        nodes = [S.Node("a", {"cpu": "4", "memory": "8Gi", "pods": "10"}, labels={"zone": "x", "gen": "3"}, unschedulable=unsched),
                 S.Node("b", {"cpu": "4", "memory": "8Gi", "pods": "10"}, labels={"zone": "y"}),
                 S.Node("c", {"cpu": "4", "memory": "8Gi", "pods": "10"})]
        pods = [S.Pod("ns", "p0", [{"cpu": "1"}], group_name="g",
                      preferred_affinity=[(5, [("zone", "In", ("x",))]), (3, [("gen", "Gt", ("2",))]), (7, []),
                                          (0, [("zone", "Exists", ())]), (2, [("zone", "NotIn", ("x",))])])]
        return S.flatten(nodes, pods, [S.PodGroup("ns", "g")], [S.Queue("default")])

    snap = mk(False)
    assert snap.class_affinity.tolist() == [[2, 8, 2]]          # node classes sort as (no labels), (zone=x, gen=3), (zone=y)
    o = oracle_mod.Oracle(kbm.conf.load_scheduler_conf(), snap)
    mask, score = o.eval_matrix(0, 1, 1)
    assert score.tolist() == [[25, 17, 17]] and mask.tolist() == [[7]]
    o.run(["allocate"])
    assert snap.bind_map(o.binds()) == {"ns/p0": "a"}
    snap = mk(True)
    o = oracle_mod.Oracle(kbm.conf.load_scheduler_conf(), snap)
    mask, score = o.eval_matrix(0, 1, 1)
    assert score.tolist() == [[0, 25, 25]] and mask.tolist() == [[6]]


def test_host_ports_predicate_and_accounting(oracle_mod):
    """PodFitsHostPorts over nodeinfo.HostPortInfo (vendor/.../predicates/predicates.go:1153-1175, nodeinfo/host_ports.go:107-135),
    hand-derived: conflicts need the same protocol and port and equal IPs or a 0.0.0.0 on either side; a placed pod's ports join
    the node's used set.  Bits (sorted universe): 0 = 0.0.0.0/TCP/80, 1 = 0.0.0.0/UDP/80, 2 = 10.0.0.1/TCP/80, 3 = 10.0.0.2/TCP/80."""
    S = kbm.snapshot
    nodes = [S.Node(f"n{i}", {"cpu": "4", "memory": "8Gi", "pods": "10"}) for i in range(3)]
    pods = [S.Pod("ns", "web0", [{"cpu": "1"}], group_name="g", host_ports=[("", "", 80)]),
            S.Pod("ns", "web1", [{"cpu": "1"}], group_name="g", host_ports=[("10.0.0.1", "TCP", 80)]),
            S.Pod("ns", "udp", [{"cpu": "1"}], group_name="g", host_ports=[("", "UDP", 80), ("", "TCP", 0)]),
            S.Pod("ns", "plain", [{"cpu": "1"}], group_name="g"),
            S.Pod("kube-system", "run", [{"cpu": "1"}], node_name="n1", phase="Running", host_ports=[("10.0.0.2", "TCP", 80)])]
    snap = S.flatten(nodes, pods, [S.PodGroup("ns", "g")], [S.Queue("default")])
    assert snap.names["tasks"] == ["ns/plain", "ns/udp", "ns/web0", "ns/web1"]
    assert snap.node_ports.tolist() == [0, 8, 0]
    assert snap.task_port_want.tolist() == [0, 2, 1, 4]
    assert snap.task_port_conflict.tolist() == [0, 2, 13, 5]          # web0 (0.0.0.0) collides with every TCP/80, web1 not with 10.0.0.2
    o = oracle_mod.Oracle(kbm.conf.load_scheduler_conf(), snap)
    mask, _ = o.eval_matrix(0, 4, 1)
    assert mask[:, 0].tolist() == [7, 7, 5, 7]                        # only web0 is kept off n1 (10.0.0.2/TCP/80 in use there)
    # three pods that all want 0.0.0.0:80 on two nodes: the third finds no node and its job is abandoned (allocate.go:144-148)
    nodes = [S.Node(f"n{i}", {"cpu": "4", "memory": "8Gi", "pods": "10"}) for i in range(2)]
    pods = [S.Pod("ns", f"web{i}", [{"cpu": "1"}], group_name="g", host_ports=[("", "", 80)]) for i in range(3)]
    pods.append(S.Pod("ns", "plain", [{"cpu": "1"}], group_name="g"))
    snap = S.flatten(nodes, pods, [S.PodGroup("ns", "g")], [S.Queue("default")])
    o = oracle_mod.Oracle(kbm.conf.load_scheduler_conf(), snap)
    o.run(["allocate", "backfill"])
    assert snap.bind_map(o.binds()) == {"ns/plain": "n0", "ns/web0": "n1", "ns/web1": "n0"}


def test_host_ports_that_no_pending_pod_can_conflict_with_get_no_bit(oracle_mod):
    """The host-port table only interns triples a Pending pod's port can conflict with: what running daemons listen on does not widen the
    masks, and the pruned table decides like the complete one."""
    S = kbm.snapshot
    nodes = [S.Node(f"n{i}", {"cpu": "8", "memory": "16Gi", "pods": "110"}) for i in range(4)]
    daemons = [S.Pod("kube-system", f"d{i:03d}", [{"cpu": "10m"}], node_name=f"n{i % 4}", phase="Running", host_ports=[("", "TCP", 9000 + i)])
               for i in range(100)]                                    # 100 distinct ports: beyond the table without pruning
    daemons.append(S.Pod("kube-system", "web-n2", [{"cpu": "10m"}], node_name="n2", phase="Running", host_ports=[("10.0.0.7", "TCP", 80)]))
    pending = [S.Pod("ns", f"web{i}", [{"cpu": "1"}], group_name="g", host_ports=[("", "", 80)]) for i in range(4)] + \
              [S.Pod("ns", "plain", [{"cpu": "1"}], group_name="g")]
    snap = S.flatten(nodes, daemons + pending, [S.PodGroup("ns", "g")], [S.Queue("default")])
    assert snap.port_words == 1
    assert snap.node_ports.tolist() == [0, 0, 2, 0]                    # bits: 0 = 0.0.0.0/TCP/80 (asked), 1 = 10.0.0.7/TCP/80 (conflicts with it)
    assert sorted(set(snap.task_port_want.tolist())) == [0, 1] and sorted(set(snap.task_port_conflict.tolist())) == [0, 3]
    o = oracle_mod.Oracle(kbm.conf.load_scheduler_conf(), snap)
    o.run(["allocate", "backfill"])
    binds = snap.bind_map(o.binds())
    assert sorted(binds[f"ns/web{i}"] for i in range(4) if f"ns/web{i}" in binds) == ["n0", "n1", "n3"]   # n2 is taken, the fourth finds none
    # the complete tables — one word with few daemons, two words with all of them (kb_snapshot.port_words; the triples the Pending pods'
    # ports conflict with take the low bits) — decide like the pruned one
    few = daemons[:40] + daemons[-1:] + pending
    a = S.flatten(nodes, few, [S.PodGroup("ns", "g")], [S.Queue("default")])
    b = S.flatten(nodes, few, [S.PodGroup("ns", "g")], [S.Queue("default")], prune_ports=False)
    c = S.flatten(nodes, daemons + pending, [S.PodGroup("ns", "g")], [S.Queue("default")], prune_ports=False)
    assert (a.port_words, b.port_words, c.port_words) == (1, 1, 2) and c.node_ports.shape == (4, 2)
    assert c.node_ports[:, 0].tolist()[2] & 3 == 2 and sorted(set(c.task_port_conflict[:, 0].tolist())) == [0, 3]   # the same two low bits
    res = []
    for sn in (a, b, c):
        o = oracle_mod.Oracle(kbm.conf.load_scheduler_conf(), sn)
        o.run(["allocate", "backfill"])
        res.append(sn.bind_map(o.binds()))
    assert res[0] == res[1] and len(res[0]) == 4
    assert res[2] == binds                                             # all daemons: two words against the pruned single word


def _preempt_tiers():
    """preempt_test.go:162-176: one tier, conformance and gang with EnabledPreemptable only."""
    return kbm.conf.tiers_literal([kbm.conf.PluginOption("conformance", enabled=abi.EN_PREEMPTABLE),
                                   kbm.conf.PluginOption("gang", enabled=abi.EN_PREEMPTABLE)])


def preempt_reference_cases():
    """-> [(snapshot, conf, evictions in cache.Evict order)]: both cases of actions/preempt/preempt_test.go:51-131"""
    fx = kbm.fixtures
    S = kbm.snapshot
    rl = fx.build_resource_list
    one = S.flatten(
        nodes=[S.Node("n1", rl("3", "3Gi"))],
        pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg1")],
        pod_groups=[S.PodGroup("c1", "pg1", queue="q1")], queues=[S.Queue("q1", 1)])
    two = S.flatten(
        nodes=[S.Node("n1", rl("2", "2G"))],
        pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2"),
              fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg2")],
        pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q1")], queues=[S.Queue("q1", 1)])
    return [(one, _preempt_tiers(), ["c1/preemptee2"]), (two, _preempt_tiers(), ["c1/preemptee2", "c1/preemptee1"])]


def test_reference_preempt_cases(oracle_mod):
    """actions/preempt/preempt_test.go:51-131, both cases: the number of evictions the FakeEvictor records (1 and 2).
    The restatement also says WHO is evicted: victims leave in reverse task order (preempt.go:223-225 negates TaskOrderFn)."""
    fx = kbm.fixtures
    S = kbm.snapshot
    rl = fx.build_resource_list
    # case 1: one job, two running + two pending pods, node 3 cpu / 3Gi
    snap = S.flatten(
        nodes=[S.Node("n1", rl("3", "3Gi"))],
        pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg1")],
        pod_groups=[S.PodGroup("c1", "pg1", queue="q1")], queues=[S.Queue("q1", 1)])
    o = oracle_mod.Oracle(_preempt_tiers(), snap)
    o.run(["preempt"])
    ev = [snap.task_name(int(t)) for t in o.evictions()]
    assert len(ev) == 1 and ev == ["c1/preemptee2"]
    st, nd = o.task_state()
    names = snap.names["tasks"]
    # the first preemptor found no victim outside its own job (phase 1) and was consumed; the second one got the slot
    assert st[names.index("c1/preemptor2")] == abi.TASK_PIPELINED and snap.node_name(int(nd[names.index("c1/preemptor2")])) == "n1"
    assert st[names.index("c1/preemptor1")] == abi.TASK_PENDING and st[names.index("c1/preemptee2")] == abi.TASK_RELEASING
    # case 2: the pending pods belong to a second job of the same queue, node 2 cpu / 2G completely used
    snap = S.flatten(
        nodes=[S.Node("n1", rl("2", "2G"))],
        pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
              fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2"),
              fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg2")],
        pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q1")], queues=[S.Queue("q1", 1)])
    o = oracle_mod.Oracle(_preempt_tiers(), snap)
    o.run(["preempt"])
    ev = [snap.task_name(int(t)) for t in o.evictions()]
    assert ev == ["c1/preemptee2", "c1/preemptee1"]
    st, _ = o.task_state()
    names = snap.names["tasks"]
    assert all(st[names.index(n)] == abi.TASK_PIPELINED for n in ("c1/preemptor1", "c1/preemptor2"))
    idle, rel, _, _, cnt = o.node_state()
    assert idle[0, 0] == 0.0 and rel[0, 0] == 0.0 and cnt[0] == 4      # both releasing slots are spoken for


def test_preempt_gang_discard_and_priority(oracle_mod):
    """Hand-derived: a gang of 3 (minAvailable 3) can only displace two running pods -> the job never becomes pipelined and the
    whole statement is discarded (preempt.go:129-133, statement.go:193-205): no eviction, every status and the node restored.
    With room for all three the statement commits; the priority plugin only lets lower-priority jobs be victims."""
    fx = kbm.fixtures
    S = kbm.snapshot
    rl = fx.build_resource_list
    tiers = kbm.conf.tiers_literal([kbm.conf.PluginOption("priority", enabled=abi.EN_PREEMPTABLE | abi.EN_JOB_ORDER),
                                    kbm.conf.PluginOption("gang", enabled=abi.EN_PREEMPTABLE | abi.EN_JOB_PIPELINED)])

    def cluster(n_victims, victim_prio):
        pods = [fx.build_pod("c1", f"low{i}", "n1", "Running", rl("1", "1G"), "low") for i in range(n_victims)]
        pods += [fx.build_pod("c1", f"high{i}", "", "Pending", rl("1", "1G"), "high") for i in range(3)]
        return S.flatten(nodes=[S.Node("n1", rl(str(n_victims), f"{n_victims}G"))], pods=pods,
                         pod_groups=[S.PodGroup("c1", "low", queue="q1", priority=victim_prio),
                                     S.PodGroup("c1", "high", queue="q1", min_member=3, priority=10)], queues=[S.Queue("q1", 1)])

    snap = cluster(2, 1)
    o = oracle_mod.Oracle(tiers, snap)
    before = [a.copy() for a in o.node_state()]
    o.run(["preempt"])
    assert len(o.evictions()) == 0
    st, _ = o.task_state()
    assert sorted(st.tolist()) == [abi.TASK_PENDING] * 3 + [abi.TASK_RUNNING] * 2
    for a, b in zip(o.node_state(), before):
        assert np.array_equal(a, b)
    snap = cluster(3, 1)
    o = oracle_mod.Oracle(tiers, snap)
    o.run(["preempt"])
    assert sorted(snap.task_name(int(t)) for t in o.evictions()) == ["c1/low0", "c1/low1", "c1/low2"]
    snap = cluster(3, 10)                                              # equal priority: nobody may be preempted (priority.go:86-92)
    o = oracle_mod.Oracle(tiers, snap)
    o.run(["preempt"])
    assert len(o.evictions()) == 0


def test_reference_reclaim_case(oracle_mod):
    """actions/reclaim/reclaim_test.go:51-99: queue q1 holds the whole node, q2's pending pod reclaims exactly one running pod
    (tiers: conformance + gang with EnabledReclaimable, reclaim_test.go:140-154).  Victims go in list order (reclaim.go:156-169)."""
    fx = kbm.fixtures
    S = kbm.snapshot
    rl = fx.build_resource_list
    snap = S.flatten(
        nodes=[S.Node("n1", rl("3", "3Gi"))],
        pods=[fx.build_pod("c1", f"preemptee{i}", "n1", "Running", rl("1", "1G"), "pg1") for i in (1, 2, 3)] +
             [fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2")],
        pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q2")],
        queues=[S.Queue("q1", 1), S.Queue("q2", 1)])
    tiers = kbm.conf.tiers_literal([kbm.conf.PluginOption("conformance", enabled=abi.EN_RECLAIMABLE),
                                    kbm.conf.PluginOption("gang", enabled=abi.EN_RECLAIMABLE)])
    o = oracle_mod.Oracle(tiers, snap)
    o.run(["reclaim"])
    assert [snap.task_name(int(t)) for t in o.evictions()] == ["c1/preemptee1"]
    st, nd = o.task_state()
    names = snap.names["tasks"]
    assert st[names.index("c1/preemptor1")] == abi.TASK_PIPELINED and nd[names.index("c1/preemptor1")] == 0
    # proportion's reclaimable fn (proportion.go:171-196): a victim must leave its queue at or above its deserved share.
    # Two equal-weight queues on 4 cpu: q1 runs 3 pods (deserved 2 cpu... but capped by its request) -> exactly one may go.
    tiers = kbm.conf.tiers_literal([kbm.conf.PluginOption("proportion", enabled=abi.EN_RECLAIMABLE | abi.EN_QUEUE_ORDER),
                                    kbm.conf.PluginOption("gang", enabled=abi.EN_RECLAIMABLE)])
    snap = S.flatten(
        nodes=[S.Node("n1", rl("4", "4G"))],
        pods=[fx.build_pod("c1", f"run{i}", "n1", "Running", rl("1", "1G"), "pg1") for i in (1, 2, 3, 4)] +
             [fx.build_pod("c1", f"want{i}", "", "Pending", rl("1", "1G"), "pg2") for i in (1, 2, 3)],
        pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q2")],
        queues=[S.Queue("q1", 1), S.Queue("q2", 1)])
    o = oracle_mod.Oracle(tiers, snap)
    o.run(["reclaim"])
    # deserved = 2 cpu each (water-fill over equal weights, both queues request >= 2): q1 may shrink from 4 to 2, no further;
    # but a job gets ONE reclaim attempt per pop of its queue and is not re-pushed (reclaim.go:100-114): one eviction
    assert len(o.evictions()) == 1


def test_conformance_protects_system_pods_from_preemption(oracle_mod):
    """plugins/conformance/conformance.go:44-58: pods in kube-system or with a system-critical priority class are never victims."""
    fx = kbm.fixtures
    S = kbm.snapshot
    rl = fx.build_resource_list
    pods = [S.Pod("kube-system", "dns", [rl("1", "1G")], group_name="sys", node_name="n1", phase="Running"),
            S.Pod("c1", "critical", [rl("1", "1G")], group_name="low", node_name="n1", phase="Running", priority_class_name="system-node-critical"),
            S.Pod("c1", "plain", [rl("1", "1G")], group_name="low", node_name="n1", phase="Running"),
            S.Pod("c1", "want1", [rl("1", "1G")], group_name="high"), S.Pod("c1", "want2", [rl("1", "1G")], group_name="high")]
    snap = S.flatten(nodes=[S.Node("n1", rl("3", "3G"))], pods=pods,
                     pod_groups=[S.PodGroup("kube-system", "sys", queue="q1"), S.PodGroup("c1", "low", queue="q1"), S.PodGroup("c1", "high", queue="q1")],
                     queues=[S.Queue("q1", 1)])
    assert snap.task_evict_protected.sum() == 2
    o = oracle_mod.Oracle(_preempt_tiers(), snap)
    o.run(["preempt"])
    assert [snap.task_name(int(t)) for t in o.evictions()] == ["c1/plain"]


def test_oracle_matches_its_committed_digests(oracle_mod):
    """tests/golden/oracle_digests.json (made by tests/golden/make_golden.py): the parity checker must not drift silently."""
    import json
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(here, "golden", "oracle_digests.json")))
    for name, idx, scale, actions in mg.CASES:
        assert mg.digest(kbm, oracle_mod, idx, scale, actions) == want[name], name


def test_threaded_oracle_equals_single_threaded(oracle_mod):
    """The cpu_baseline leg times the oracle with the reference's 16-worker fan-out per task (util/scheduler_helper.go:63-86; used for
    N >= 256).  The fan-out must not change a single decision."""
    cfg = conf.load_scheduler_conf()
    snap = snapmod.synth(snapmod.synth_config(2, 0.4))
    assert snap.n_nodes >= 256
    runs = []
    for threads in (1, 8, 16):
        o = oracle_mod.Oracle(cfg, snap, threads=threads)
        o.run(["allocate", "backfill"])
        runs.append((o.decisions(), o.binds(), o.evals, [a.copy() for a in o.node_state()]))
        o.close()
    for r in runs[1:]:
        assert np.array_equal(r[0], runs[0][0]) and np.array_equal(r[1], runs[0][1]) and r[2] == runs[0][2]
        for a, b in zip(r[3], runs[0][3]):
            assert np.array_equal(a, b)
