"""Python mirror of the reference's framework layer for the hot path — same names, argument meaning and error
behaviour as pkg/scheduler/framework, so the harness and the parity tests read like the reference's own action tests.

  OpenSession / CloseSession     framework/framework.go:30-63
  Session.Allocate / Pipeline    framework/session.go:194-288 (status index, node accounting, gang-gated dispatch)
  Action {Name, Initialize, Execute, UnInitialize}; RegisterAction / GetAction   framework/interface.go:20-32, plugins.go:58-72
  FakeBinder                     pkg/scheduler/util/test_utils.go:94-118

The Actions registered here are the engine-backed replacements: Execute(ssn) flattens nothing (the Session already wraps
the SoA snapshot), calls the C ABI (kb_run_allocate / kb_run_backfill) and REPLAYS the returned decisions through
ssn.Allocate / ssn.Pipeline — exactly what the Go shim of INTEGRATION.md does — so the bind set is produced by the
Session's own gang-gated dispatch, independently of the device's K2 reduction (the tests compare the two).
"""
from typing import Dict, List, Optional

import numpy as np

from . import abi
from .conf import SchedulerConf
from .engine import Engine
from .snapshot import SessionSnapshot

_ALLOCATED_STATUS = (abi.TASK_BOUND, abi.TASK_BINDING, abi.TASK_RUNNING, abi.TASK_ALLOCATED)   # api/helpers.go:64-71


class FakeBinder:
    """util.FakeBinder: records {ns/name: hostname}."""

    def __init__(self):
        self.Binds: Dict[str, str] = {}
        self.order: List[str] = []

    def Bind(self, task_key: str, hostname: str):
        self.Binds[task_key] = hostname
        self.order.append(task_key)


class Session:
    """framework.Session restricted to what an Action on this path touches."""

    def __init__(self, snap: SessionSnapshot, conf: SchedulerConf, binder: Optional[FakeBinder] = None, **engine_kw):
        self.snap = snap
        self.Tiers = conf.tiers
        self.conf = conf
        self.binder = binder or FakeBinder()
        self.task_status = snap.task_status.copy()
        self.task_node = snap.task_node.copy()
        self.node_idle = snap.node_idle.copy()
        self.node_releasing = snap.node_releasing.copy()
        self.node_pod_cnt = snap.node_pod_cnt.copy()
        self._gang_ready = any(p.name == "gang" and (p.enabled & abi.EN_JOB_READY) for t in conf.tiers for p in t)
        # engine=<object>: a caller-provided engine (tests replay recorded decisions through the Session without a device)
        self.engine = engine_kw.pop("engine", None)
        if self.engine is None:
            self.engine = Engine(conf, **engine_kw)
            self.engine.load(snap)

    # ---- JobInfo counters (api/job_info.go:383-434)
    def _job_tasks(self, j):
        return range(int(self.snap.job_task_begin[j]), int(self.snap.job_task_begin[j + 1]))

    def ReadyTaskNum(self, j) -> int:
        st = self.task_status[self.snap.job_task_begin[j]:self.snap.job_task_begin[j + 1]]
        return int(np.isin(st, _ALLOCATED_STATUS + (abi.TASK_SUCCEEDED,)).sum())

    def JobReady(self, j) -> bool:   # session_plugins.go:182-200 + gang.go:122-125
        if not self._gang_ready:
            return True
        return self.ReadyTaskNum(j) >= int(self.snap.job_min_available[j])

    def _dispatch(self, t):          # session.go:290-314
        self.binder.Bind(self.snap.task_name(t), self.snap.node_name(int(self.task_node[t])))
        self.task_status[t] = abi.TASK_BINDING

    def Allocate(self, t: int, n: int):
        """ssn.Allocate (session.go:235-288)."""
        s = self.snap
        self.task_status[t] = abi.TASK_ALLOCATED
        self.node_idle[:, n] -= s.task_resreq[:, t]          # node.AddTask -> allocateIdleResource (node_info.go:161-203)
        self.node_pod_cnt[n] += 1
        self.task_node[t] = n
        j = int(s.task_job[t])
        if self.JobReady(j):
            for tt in self._job_tasks(j):
                if self.task_status[tt] == abi.TASK_ALLOCATED:
                    self._dispatch(tt)

    def Pipeline(self, t: int, n: int):
        """ssn.Pipeline (session.go:194-232)."""
        s = self.snap
        self.task_status[t] = abi.TASK_PIPELINED
        self.node_releasing[:, n] -= s.task_resreq[:, t]
        self.node_pod_cnt[n] += 1
        self.task_node[t] = n

    def binds_array(self) -> np.ndarray:
        out = np.full(self.snap.n_tasks, abi.KB_NONE, np.uint32)
        names = {self.snap.task_name(t): t for t in range(self.snap.n_tasks)}
        nodes = {self.snap.node_name(n): n for n in range(self.snap.n_nodes)}
        for k, v in self.binder.Binds.items():
            out[names[k]] = nodes[v]
        return out


class Action:
    def Name(self) -> str:
        raise NotImplementedError

    def Initialize(self):
        pass

    def Execute(self, ssn: Session):
        raise NotImplementedError

    def UnInitialize(self):
        pass


def _replay(ssn: Session, decisions: np.ndarray):
    for task, node, kind in decisions:
        if kind == 0:
            ssn.Allocate(int(task), int(node))
        else:
            ssn.Pipeline(int(task), int(node))


class GpuAllocateAction(Action):
    """Replacement for actions/allocate (allocate.go:43-194): device rounds + replay."""

    def Name(self):
        return "allocate"

    def Execute(self, ssn: Session):
        _replay(ssn, ssn.engine.run_allocate())


class GpuBackfillAction(Action):
    """Replacement for actions/backfill (backfill.go:40-71)."""

    def Name(self):
        return "backfill"

    def Execute(self, ssn: Session):
        _replay(ssn, ssn.engine.run_backfill())


_ACTIONS: Dict[str, Action] = {}


def RegisterAction(act: Action):            # framework/plugins.go:58-63
    _ACTIONS[act.Name()] = act


def GetAction(name: str) -> Action:         # framework/plugins.go:66-72
    if name not in _ACTIONS:
        raise KeyError(f"failed to found Action {name}")
    return _ACTIONS[name]


RegisterAction(GpuAllocateAction())
RegisterAction(GpuBackfillAction())


def OpenSession(snap: SessionSnapshot, conf: SchedulerConf, **engine_kw) -> Session:
    return Session(snap, conf, **engine_kw)


def CloseSession(ssn: Session):
    ssn.engine.close()
