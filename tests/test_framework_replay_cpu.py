"""The replay half of the drop-in on CPU: decisions recorded from the oracle are fed through the Python mirror of framework.Session
(Allocate / Pipeline / gang-gated dispatch) by the same Action objects the engine-backed path uses; the Binder must see the
oracle's bind set and the Session's node accounting must match.  (The engine-backed half is tests/test_framework_actions.py, -m gpu.)"""
import importlib

import numpy as np
import pytest

import test_pyref_vs_oracle as cases

kbm = importlib.import_module("kube-batch_amd")
fw = importlib.import_module("kube-batch_amd.framework")
fixtures = importlib.import_module("kube-batch_amd.fixtures")
abi = kbm.abi


class RecordedEngine:
    """stands in for kube-batch_amd.engine.Engine: hands out what the oracle decided, action by action"""

    def __init__(self, oracle_mod, conf, snap):
        self.o = oracle_mod.Oracle(conf, snap)
        self.n = 0

    def _take(self, action):
        self.o.run([action])
        d = self.o.decisions()
        out, self.n = d[self.n:], len(d)
        return out

    def run_allocate(self):
        return self._take("allocate")

    def run_backfill(self):
        return self._take("backfill")

    def close(self):
        self.o.close()


@pytest.mark.parametrize("case", range(2))
def test_reference_allocate_cases_through_the_session_mirror(oracle_mod, case):
    """allocate_test.go:38-212: the object the reference's test compares is the FakeBinder's map"""
    name, snap, expected = fixtures.allocate_cases()[case]
    tiers = fixtures.allocate_test_tiers()
    ssn = fw.OpenSession(snap, tiers, engine=RecordedEngine(oracle_mod, tiers, snap))
    fw.GetAction("allocate").Execute(ssn)
    assert ssn.binder.Binds == expected, name


@pytest.mark.parametrize("seed", range(12))
def test_replay_reproduces_binds_and_node_accounting(oracle_mod, seed):
    cfg, snap = cases._case(seed)
    eng = RecordedEngine(oracle_mod, cfg, snap)
    ssn = fw.OpenSession(snap, cfg, engine=eng)
    for a in ("allocate", "backfill"):
        fw.GetAction(a).Execute(ssn)
    o = eng.o
    assert np.array_equal(ssn.binds_array(), o.binds())
    idle, rel, _, _, cnt = o.node_state()
    assert np.array_equal(ssn.node_idle, idle) and np.array_equal(ssn.node_releasing, rel) and np.array_equal(ssn.node_pod_cnt, cnt)
    st, nd = o.task_state()
    assert np.array_equal(ssn.task_status, st) and np.array_equal(ssn.task_node, nd)
