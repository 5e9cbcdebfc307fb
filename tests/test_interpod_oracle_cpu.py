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
