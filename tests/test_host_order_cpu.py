"""The engine's host order machine (kube-batch_amd/csrc/kb_order.cpp, compiled unchanged with g++ into a test harness) driven
on CPU through the same protocol the engine's ActionRun uses around the device rounds — plan a speculated window behind a
checkpoint, plan the next one behind a second checkpoint, absorb the device's answer (confirm / roll back + replay / mark dead
shapes / undo the last pop) — with tests/pyref.py playing the device.  The task sequence, the decisions, the popped count and
the running drf / proportion / gang aggregates must equal the sequential reference loop's, for every window size and for
random break patterns.  This is host logic only: no kernel runs here (the kernels' parity tests are the `-m gpu` ones)."""
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
abi, conf = kbm.abi, kbm.conf

HERE = os.path.dirname(os.path.abspath(__file__))
DONE, NO_FEASIBLE, PIPELINED, RENORM = 0, 1, 2, 4           # KB_REASON_* (kube-batch_amd/csrc/kb_device.h)


@pytest.fixture(scope="module")
def harness():
    return _build()


def _build():
    if os.environ.get("KB_ORDER_HARNESS_LIB"):               # an instrumented build (scripts/sanitize_cpu.sh)
        return _bind(C.CDLL(os.environ["KB_ORDER_HARNESS_LIB"]))
    out_dir = os.path.join(HERE, "host_harness", "build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liborderharness.so")
    srcs = [os.path.join(HERE, "host_harness", "order_harness.cpp"), os.path.join(HERE, "..", "kube-batch_amd", "csrc", "kb_order.cpp")]
    deps = srcs + [os.path.join(HERE, "..", "kube-batch_amd", "csrc", f) for f in ("kb_host.hpp", "kb_res.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        tmp = f"{so}.{os.getpid()}"                          # atomic: several pytest workers may build at once
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o", tmp] + srcs)
        os.replace(tmp, so)
    return _bind(C.CDLL(so))


def _bind(L):
    L.hh_create.restype = C.c_void_p
    L.hh_steps.restype = C.c_uint64
    return L


def _policy(cfg):
    """compile_policy (kube-batch_amd/csrc/kb_session.cpp) restated for this harness (tests/test_host_evict_cpu.py runs the real one)."""
    chain, pol = [], dict(qprop=0, tprio=0, gready=0, gang=0, drf=0, prop=0)
    for tier in cfg.tiers:
        for po in tier:
            en = po.enabled
            if po.name == "priority":
                if en & abi.EN_JOB_ORDER:
                    chain.append(abi.PLUGIN_IDS["priority"])
                if en & abi.EN_TASK_ORDER:
                    pol["tprio"] = 1
            elif po.name == "gang":
                pol["gang"] = 1
                if en & abi.EN_JOB_ORDER:
                    chain.append(abi.PLUGIN_IDS["gang"])
                if en & abi.EN_JOB_READY:
                    pol["gready"] = 1
            elif po.name == "drf":
                pol["drf"] = 1
                if en & abi.EN_JOB_ORDER:
                    chain.append(abi.PLUGIN_IDS["drf"])
            elif po.name == "proportion":
                pol["prop"] = 1
                if en & abi.EN_QUEUE_ORDER:
                    pol["qprop"] = 1
    return chain, pol


class Machine:
    def __init__(self, L, cfg, snap, p):
        self.L = L
        R, T, J, Q = snap.n_res, snap.n_tasks, snap.n_jobs, snap.n_queues
        self.R, self.T, self.J, self.Q = R, T, J, Q
        rows = np.zeros((T, R))
        for t in range(T):
            for d in range(R):
                rows[t, d] = p.resreq[t].get(d)
        empty = np.array([p.resreq[t].is_empty() for t in range(T)], np.uint8)

        def mask(r):
            return sum(1 << (d - 2) for d in (r.scalars or {}))
        total = np.array([p.total.get(d) for d in range(R)])
        des = np.zeros((Q, R))
        desm = np.zeros(Q, np.uint32)
        qalloc, qshare = np.zeros((Q, R)), np.zeros(Q)
        for q, a in p.qattr.items():
            des[q] = [a["deserved"].get(d) for d in range(R)]
            desm[q] = mask(a["deserved"])
            qalloc[q] = [a["allocated"].get(d) for d in range(R)]
            qshare[q] = a["share"]
        jalloc, jshare = np.zeros((J, R)), np.zeros(J)
        if p.jalloc:
            for j in range(J):
                jalloc[j] = [p.jalloc[j].get(d) for d in range(R)]
                jshare[j] = p.jshare[j]
        ready = np.array([p.ready_num(j) for j in range(J)], np.int32)
        chain, pol = _policy(cfg)
        chain_a = np.array(chain + [0], np.uint8)
        keep = [rows, np.ascontiguousarray(snap.task_scalar_mask, np.uint32), np.ascontiguousarray(snap.task_priority, np.int32),
                np.ascontiguousarray(snap.task_creation, np.int64), np.ascontiguousarray(snap.task_status, np.uint8), empty,
                np.ascontiguousarray(snap.job_task_begin, np.uint32), np.ascontiguousarray(snap.job_queue, np.uint32),
                np.ascontiguousarray(snap.job_min_available, np.int32), np.ascontiguousarray(snap.job_priority, np.int32),
                np.ascontiguousarray(snap.job_creation, np.int64), np.ascontiguousarray(snap.queue_creation, np.int64),
                total, des, desm, jalloc, jshare, qalloc, qshare, ready, chain_a]
        ptr = [a.ctypes.data_as(C.c_void_p) for a in keep]
        self.h = C.c_void_p(L.hh_create(
            C.c_int(R), C.c_uint32(T), C.c_uint32(J), C.c_uint32(Q), *ptr[:12], ptr[12], C.c_uint32(mask(p.total)), ptr[13], ptr[14],
            ptr[15], ptr[16], ptr[17], ptr[18], ptr[19], ptr[20], C.c_int(len(chain)), C.c_int(pol["qprop"]), C.c_int(pol["tprio"]),
            C.c_int(pol["gready"]), C.c_int(pol["gang"]), C.c_int(pol["drf"]), C.c_int(pol["prop"])))

    def next(self):
        t = C.c_uint32()
        return int(t.value) if self.L.hh_next(self.h, C.byref(t)) else None

    def report(self, outcome):
        self.L.hh_report(self.h, C.c_int({"alloc": 0, "pipe": 1, "none": 2}[outcome]))

    def state(self):
        js, qs, rd = np.zeros(self.J), np.zeros(self.Q), np.zeros(self.J, np.int32)
        ja, qa = np.zeros((self.J, self.R)), np.zeros((self.Q, self.R))
        self.L.hh_state(self.h, *[a.ctypes.data_as(C.c_void_p) for a in (js, qs, rd, ja, qa)])
        return js, qs, rd, ja, qa

    def close(self):
        self.L.hh_destroy(self.h)


def _feas_shapes(snap, p):
    """t_feas_shape + the vectors mark_dead compares (kb_session_load): InitResreq with sub-epsilon scalars read as 0, class, conflicts."""
    ids, shape, eff = {}, [], []
    for t in range(snap.n_tasks):
        v = tuple(p.init[t].get(d) if (d < 2 or p.init[t].get(d) > 10.0) else 0.0 for d in range(snap.n_res))
        key = (tuple(p.init[t].get(d) for d in range(snap.n_res)), tuple(sorted((p.init[t].scalars or {}).keys())), p.tcls[t], p.tconf[t])
        if key not in ids:
            ids[key] = len(eff)
            eff.append((v, p.tcls[t], p.tconf[t]))
        shape.append(ids[key])
    return shape, eff


def _drive(L, cfg, snap, window, ahead, renorm_prob, rng):
    """ActionRun::plan / plan_ahead / promote / absorb and run_action's loop (kb_engine.cpp), with pyref as the device."""
    p = pyref.Session(cases._tiers(cfg), snap)
    m = Machine(L, cfg, snap, p)
    shape, eff = _feas_shapes(snap, p)
    dead = [False] * len(eff)
    decs, popped = [], 0

    def mark_dead(x):
        for y in range(len(eff)):
            if not dead[y] and eff[y][1] == eff[x][1] and eff[y][2] == eff[x][2] and all(a >= b for a, b in zip(eff[y][0], eff[x][0])):
                dead[y] = True
        dead[x] = True

    def speculate():
        rows, pops = [], 0
        while len(rows) < window:
            t = m.next()
            if t is None:
                break
            pops += 1
            if dead[shape[t]]:
                m.report("none")
                continue
            rows.append(t)
            m.report("alloc")
        return rows, pops

    def device(rows):
        out = []
        for i, t in enumerate(rows):
            if i > 0 and rng.uniform() < renorm_prob:
                return i, RENORM, out
            r = p.place(t)
            if r == "none":
                return i, NO_FEASIBLE, out
            assert r in ("alloc", "pipe")
            out.append((t, p.tnode[t], 1 if r == "pipe" else 0))
            if r == "pipe":
                return i + 1, PIPELINED, out
        return len(rows), DONE, out

    L.hh_checkpoint(m.h)
    rows, spec_pops = speculate()
    if not rows:
        popped += spec_pops
    ahead_windows = []                                       # the windows speculated behind the current one, each behind its own roll-back point
    depth = 2 if ahead == 2 else (1 if ahead else 0)         # run_action: one queued behind the round in flight, one more only planned (ahead == 2)
    while rows:
        while len(ahead_windows) < depth and (not ahead_windows or ahead_windows[-1][0]):
            L.hh_push_checkpoint(m.h)
            ahead_windows.append(speculate())
        n_done, reason, out = device(rows)
        if reason == DONE:
            popped += spec_pops
            decs += out
        else:
            L.hh_rollback(m.h)
            ahead_windows = []
            i = 0
            while True:
                t = m.next()
                assert t is not None, "order replay ran out of tasks"
                popped += 1
                if dead[shape[t]]:
                    m.report("none")
                    continue
                assert t == rows[i], "order replay diverged from the speculated sequence"
                if reason == NO_FEASIBLE and i == n_done:
                    mark_dead(shape[t])
                    m.report("none")
                    break
                if reason == RENORM and i == n_done:
                    L.hh_rollback_last_pop(m.h)
                    popped -= 1
                    break
                decs.append(out[i])
                m.report("pipe" if out[i][2] else "alloc")
                i += 1
                if reason == PIPELINED and i == n_done:
                    break
        if ahead and reason == DONE:
            L.hh_pop_commit(m.h)
            rows, spec_pops = ahead_windows.pop(0)
            if not rows:
                popped += spec_pops
        else:
            L.hh_checkpoint(m.h)
            rows, spec_pops = speculate()
            if not rows:
                popped += spec_pops
    state = m.state()
    m.close()
    return p, decs, popped, state


def _check(L, cfg, snap, seed):
    ref = pyref.Session(cases._tiers(cfg), snap).run(["allocate"])
    rng = np.random.RandomState(seed)
    # ahead: 0 = one window at a time, 1 = the next window speculated behind the one in flight, 2 = two windows (run_action since round 6)
    for window, ahead, renorm in ((1, 0, 0.0), (int(rng.choice([2, 3, 5, 8])), 1, 0.0), (int(rng.choice([16, 64, 256])), 2, 0.0),
                                  (int(rng.choice([2, 3, 5, 8])), 2, 0.0), (int(rng.choice([4, 7, 32])), int(rng.randint(3)), 0.15)):
        p, decs, popped, (js, qs, rd, ja, qa) = _drive(L, cfg, snap, window, ahead, renorm, rng)
        tag = (seed, window, ahead, renorm)
        assert decs == ref.decisions, tag
        assert popped == ref.popped, tag
        assert p.binds == ref.binds, tag
        assert rd.tolist() == [ref.ready_num(j) for j in range(snap.n_jobs)], tag
        if ref.jalloc:
            assert js.tolist() == ref.jshare, tag
            assert ja.tolist() == [[ref.jalloc[j].get(d) for d in range(snap.n_res)] for j in range(snap.n_jobs)], tag
        for q, a in ref.qattr.items():
            assert qs[q] == a["share"], tag
            assert qa[q].tolist() == [a["allocated"].get(d) for d in range(snap.n_res)], tag


@pytest.fixture(params=[0, 1], ids=["heap-copies", "heap-journals"])
def heap_mode(harness, request):
    """The roll-back points keep the two heap arrays as copies (small sessions) or as first-write journals (from OrderMachine::kJournalJobs jobs on):
    every case both ways — the sessions of this file are all small."""
    harness.hh_set_journal(C.c_int(request.param))
    yield request.param
    harness.hh_set_journal(C.c_int(-1))


@pytest.mark.parametrize("seed", range(40))
def test_order_machine_on_synthetic_clusters(harness, heap_mode, seed):
    cfg, snap = cases._case(seed)
    _check(harness, cfg, snap, seed)


@pytest.mark.parametrize("seed", range(200))
def test_order_machine_on_adversarial_snapshots(harness, heap_mode, seed):
    snap = rawgen.raw_snapshot(seed)
    rng = np.random.RandomState(seed)
    if seed % 4 == 3:
        cfg = conf.load_scheduler_conf(cases.CONF_NO_SHARES)
    else:
        wl, wm, wa, wb = [int(x) for x in rng.choice([0, 1, 1, 2, 5], size=4)]
        cfg = conf.load_scheduler_conf(cases.CONF_TMPL.format(wl=wl, wm=wm, wa=wa, wb=wb))
    try:
        pyref.Session(cases._tiers(cfg), snap).run(["allocate"])
    except ArithmeticError:
        pytest.skip("the reference would panic on this snapshot")
    _check(harness, cfg, snap, seed)
