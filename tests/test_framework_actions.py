"""The reference's action tests, driven through the Python mirror of framework.Session / Action over the C ABI
(pkg/scheduler/actions/allocate/allocate_test.go:38-212).  Needs the GPU: the Action is the engine-backed one."""
import importlib

import numpy as np
import pytest

kbm = importlib.import_module("kube-batch_amd")
fixtures = importlib.import_module("kube-batch_amd.fixtures")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(2))
def test_allocate(case):
    fw = importlib.import_module("kube-batch_amd.framework")
    name, snap, expected = fixtures.allocate_cases()[case]
    ssn = fw.OpenSession(snap, fixtures.allocate_test_tiers())
    try:
        allocate = fw.GetAction("allocate")
        allocate.Execute(ssn)
        assert ssn.binder.Binds == expected, f"case {case} ({name}): expected {expected}, got {ssn.binder.Binds}"
        # the Session's own gang-gated dispatch (replay) and the device's gang ballot (K2) agree
        assert np.array_equal(ssn.binds_array(), ssn.engine.binds())
    finally:
        fw.CloseSession(ssn)


def test_default_conf_actions_on_gang_cluster(oracle_mod):
    """allocate then backfill from the default scheduler conf; replayed Session state == device state == oracle."""
    fw = importlib.import_module("kube-batch_amd.framework")
    conf = kbm.conf.load_scheduler_conf()
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.03))
    ssn = fw.OpenSession(snap, conf)
    try:
        for name in conf.actions:
            fw.GetAction(name).Execute(ssn)
        o = oracle_mod.Oracle(conf, snap)
        o.run(conf.actions)
        assert np.array_equal(ssn.binds_array(), o.binds())
        assert np.array_equal(ssn.engine.binds(), o.binds())
        idle, rel, _, _, cnt = o.node_state()
        assert np.array_equal(ssn.node_idle, idle) and np.array_equal(ssn.node_pod_cnt, cnt)
        assert [snap.task_name(int(t)) for t in o.bind_order()] == ssn.binder.order
    finally:
        fw.CloseSession(ssn)
