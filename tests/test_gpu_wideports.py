"""Host-port masks of several 64-bit words (kb_snapshot.port_words > 1: more than 64 interned (ip, protocol, port) triples in one session).
K1 evaluates the whole mask; the commit kernels keep to word 0, so a Pending pod that reaches beyond it is decided in a round of its own
and the host ORs its high words into the node's afterwards (kb_host.hpp: t_wide).  The oracle keeps masks of any width (the reference
keeps sets): decisions, binds, node state, shares, evictions and journals must be identical — allocate / backfill on the fuzz clusters,
the evict actions, adversarial raw snapshots, and pods / nodes flattened from objects with more than 64 distinct host ports."""
import importlib

import numpy as np
import pytest

import rawgen
import test_gpu_fuzz as fuzz
import test_gpu_preempt as pre
import test_pyref_vs_oracle as cases

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
abi, conf, snapmod = kbm.abi, kbm.conf, kbm.snapshot

pytestmark = pytest.mark.gpu


def _equal_cycle(oracle_mod, cfg, snap, tag, window=0, batch=0, again=False):
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate", "backfill"])
    e = engine.Engine(cfg, window=window, commit_batch=batch)
    e.load(snap)
    dec = e.run(["allocate", "backfill"])
    od = o.decisions()
    assert dec.shape == od.shape, (tag, dec.shape, od.shape)
    assert np.array_equal(dec, od), f"{tag}: first divergence at decision {int(np.argmax((dec != od).any(axis=1)))}"
    assert np.array_equal(e.binds(), o.binds())
    for a, b in zip(e.node_state(), o.node_state()):
        assert np.array_equal(a, b)
    for a, b in zip(e.shares(), o.shares()):
        assert np.array_equal(a, b)
    if again:                              # the pristine copy in HBM carries the high words too
        e.reset()
        assert np.array_equal(e.run(["allocate", "backfill"]), od)
    st = e.stats()
    e.close()
    o.close()
    return dec, st


@pytest.mark.parametrize("seed", range(40))
def test_allocate_and_backfill(oracle_mod, seed):
    cfg, snap, window, batch = fuzz._case(seed)
    rawgen.widen_ports(snap, 9100 + seed, words=2 + seed % 3, low_share=[0.5, 0.9, 0.0][seed % 3])
    _equal_cycle(oracle_mod, cfg, snap, seed, window, batch, again=seed % 4 == 0)


def test_the_high_words_decide(oracle_mod):
    """the same cluster with the words behind the first cleared places pods differently: the test above is not vacuous"""
    differ = 0
    for seed in range(8):
        cfg, snap, window, batch = fuzz._case(seed)
        rawgen.widen_ports(snap, 9100 + seed, words=3, low_share=0.3)
        dec, _ = _equal_cycle(oracle_mod, cfg, snap, seed, window, batch)
        snap.node_ports[:, 1:] = 0; snap.task_port_want[:, 1:] = 0; snap.task_port_conflict[:, 1:] = 0
        dec0, _ = _equal_cycle(oracle_mod, cfg, snap, seed, window, batch)
        differ += dec.shape != dec0.shape or not np.array_equal(dec, dec0)
    assert differ >= 4


@pytest.mark.parametrize("seed", range(40))
def test_evict_actions(oracle_mod, seed):
    cfg, snap, order = cases._evict_case(seed) if seed % 2 == 0 else cases._evict_variant(seed)
    rawgen.widen_ports(snap, 9300 + seed, words=2 + seed % 2, p_task=0.6)
    pre._run_both(oracle_mod, cfg, snap, order, seed)


@pytest.mark.parametrize("seed", range(1, 120, 3))
def test_adversarial_snapshots(oracle_mod, seed):
    snap = rawgen.raw_snapshot(seed)
    rawgen.widen_ports(snap, 9500 + seed, words=2)
    order = [["allocate", "backfill"], ["preempt"], ["reclaim", "allocate", "backfill", "preempt"]][(seed // 3) % 3]
    cfg = conf.load_scheduler_conf(cases.CONF_FULL.format(actions=", ".join(order)))
    pre._run_both(oracle_mod, cfg, snap, order, seed)


def _objects(n_nodes=12, n_ports=150, seed=0):
    """nodes with daemon pods on many distinct host ports, pending pods that want some of them (a few on the wildcard address)"""
    rng = np.random.RandomState(seed)
    nodes = [snapmod.Node(name=f"n{i:02d}", allocatable={"cpu": "16", "memory": "64Gi", "pods": "110"}) for i in range(n_nodes)]
    queues = [snapmod.Queue(name="q", weight=1)]
    groups, pods = [], []
    ports = [int(p) for p in rng.choice(np.arange(20000, 20000 + 4 * n_ports), size=n_ports, replace=False)]
    for i, n in enumerate(nodes):          # what already runs: each node holds a different slice of the ports
        for k in range(6):
            port = ports[(i * 11 + k * 7) % n_ports]
            pods.append(snapmod.Pod(name=f"daemon-{i}-{k}", namespace="sys", node_name=n.name, phase="Running",      # no PodGroup: outside the session
                                    containers=[{"cpu": "100m", "memory": "128Mi"}], host_ports=[("" if k % 3 else "10.0.0.1", "TCP", port)]))
    for j in range(40):
        groups.append(snapmod.PodGroup(name=f"pg{j}", namespace="ns", queue="q", min_member=1))
        for k in range(3):
            port = ports[(j * 5 + k * 13) % n_ports]
            pods.append(snapmod.Pod(name=f"p{j}-{k}", namespace="ns", group_name=f"pg{j}", phase="Pending", creation=1_600_000_000 + j,
                                    containers=[{"cpu": "500m", "memory": "1Gi"}],
                                    host_ports=[("" if (j + k) % 4 == 0 else "10.0.0.1", "TCP", port)] + ([("", "UDP", port)] if k == 2 else [])))
    return nodes, queues, groups, pods


def test_flattened_objects_with_more_than_64_host_ports(oracle_mod):
    nodes, queues, groups, pods = _objects()
    snap = snapmod.flatten(nodes, pods, groups, queues)
    assert snap.port_words >= 2 and snap.task_port_conflict.shape == (snap.n_tasks, snap.port_words)
    # the low word goes to the triples the pending pods' ports conflict with most often
    cfg = conf.load_scheduler_conf()
    dec, st = _equal_cycle(oracle_mod, cfg, snap, "objects", again=True)
    assert len(dec) > 0
    # the unpruned table (every host port of the cluster gets a bit, wider masks) must decide alike
    wide = snapmod.flatten(nodes, pods, groups, queues, prune_ports=False)
    assert wide.port_words >= snap.port_words
    dec2, _ = _equal_cycle(oracle_mod, cfg, wide, "objects, unpruned")
    assert np.array_equal(dec, dec2)
