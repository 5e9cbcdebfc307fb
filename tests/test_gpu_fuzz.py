"""Differential fuzzing of the engine against the oracle (through the C ABI) on small random clusters that exercise the
commit kernel's rarer paths together: Releasing capacity (Pipeline decisions end a speculated round), init containers
(InitResreq > Resreq), scalar resources, task priorities, tight pod-count caps, bin-packing weights, small and odd windows and
commit-batch sizes.  Decisions (ordered), bind set, final node state and shares must be bit-identical."""
import importlib

import numpy as np
import pytest

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
abi, conf, snapmod = kbm.abi, kbm.conf, kbm.snapshot

pytestmark = pytest.mark.gpu

CONF_TMPL = """
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
      leastrequested.weight: {wl}
      mostrequested.weight: {wm}
      balancedresource.weight: {wb}
"""


def _case(seed):
    rng = np.random.RandomState(1000 + seed)
    R = int(rng.choice([2, 2, 3, 6]))
    p = snapmod.SynthParams(
        n_tasks=int(rng.randint(200, 2500 if seed < 24 else 9000)), n_nodes=int(rng.randint(12, 300 if seed < 24 else 900)), n_queues=int(rng.randint(1, 7)), n_res=R,
        seed=snapmod.SEED_BASE + 500 + seed, preload_node_frac=float(rng.uniform(0, 0.8)), running_job_frac=float(rng.uniform(0, 0.3)),
        best_effort_frac=float(rng.uniform(0, 0.1)), no_mem_key_frac=float(rng.uniform(0, 0.2)), scalar_job_frac=float(rng.uniform(0, 0.6)),
        zone_selector_frac=float(rng.uniform(0, 0.4)), n_zones=int(rng.randint(1, 9)))
    s = snapmod.synth(p)
    N, T = s.n_nodes, s.n_tasks
    # Releasing capacity on some nodes (tasks being deleted): feeds the Pipeline branch of allocate.go:160-183
    rel = rng.uniform(size=N) < rng.uniform(0, 0.5)
    s.node_releasing[0] = np.where(rel, rng.choice([500, 1000, 4000, 16000], size=N), 0).astype(np.float64)
    s.node_releasing[1] = np.where(rel, rng.choice([1, 4, 16, 64], size=N) * float(1 << 30), 0)
    for d in range(2, R):
        s.node_releasing[d] = np.where(rel & (s.node_allocatable[d] > 0), 1000.0 * rng.randint(0, 3, size=N), 0)
    # init containers: InitResreq = max(Resreq, init) per dimension, on non-BestEffort pending tasks
    pending = (s.task_status == abi.TASK_PENDING) & (s.task_resreq[0] > 0)
    bump = pending & (rng.uniform(size=T) < rng.uniform(0, 0.3))
    s.task_init_resreq[0] = np.where(bump, s.task_resreq[0] + rng.choice([100, 500, 2000], size=T), s.task_init_resreq[0])
    bump2 = pending & (rng.uniform(size=T) < 0.1)
    s.task_init_resreq[1] = np.where(bump2, s.task_resreq[1] + float(1 << 28), s.task_init_resreq[1])
    # task priorities (priority plugin's TaskOrderFn), tight pod caps on some nodes
    s.task_priority[:] = rng.choice([1, 1, 5, 9], size=T).astype(np.int32)
    tight = rng.uniform(size=N) < 0.3
    s.node_max_pods[:] = np.where(tight, s.node_pod_cnt + rng.randint(0, 6, size=N), s.node_max_pods).astype(np.int32)
    if seed % 3 == 0:     # preferred node affinity on about half of the task classes
        aff = rng.choice([0, 0, 2, 5, 30], size=(s.n_task_classes, s.n_node_classes)).astype(np.int32)
        aff[rng.uniform(size=s.n_task_classes) < 0.5] = 0
        s.class_affinity = aff
    if seed % 4 == 1:     # host ports
        want = np.zeros(T, np.uint64)
        has = rng.uniform(size=T) < 0.3
        want[has] = (np.uint64(1) << rng.randint(0, 5, size=int(has.sum())).astype(np.uint64))
        cmask = want | np.where(want != 0, np.uint64(1) << np.uint64(5), np.uint64(0)).astype(np.uint64)   # bit 5: a wildcard sibling
        want = np.where(rng.uniform(size=T) < 0.05, want | (np.uint64(1) << np.uint64(5)), want).astype(np.uint64)
        cmask = cmask | want
        s.task_port_want, s.task_port_conflict = want, cmask
        s.node_ports = np.where(rng.uniform(size=N) < 0.2, rng.randint(1, 64, size=N), 0).astype(np.uint64)
    s._check()
    wl, wm, wb = [int(x) for x in rng.choice([0, 1, 1, 2, 5], size=3)]
    cfg = conf.load_scheduler_conf(CONF_TMPL.format(wl=wl, wm=wm, wb=wb))
    window = int(rng.choice([0, 64, 100, 256, 333, 512, 1024]))
    batch = int(rng.choice([0, 1, 3, 8, 16]))
    return cfg, s, window, batch


@pytest.mark.parametrize("seed", range(40))
def test_engine_equals_oracle_on_random_clusters(oracle_mod, seed):
    cfg, snap, window, batch = _case(seed)
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate", "backfill"])
    e = engine.Engine(cfg, window=window, commit_batch=batch)
    e.load(snap)
    dec = e.run(["allocate", "backfill"])
    od = o.decisions()
    assert dec.shape == od.shape, (seed, dec.shape, od.shape)
    assert np.array_equal(dec, od), f"seed {seed}: first divergence at decision {int(np.argmax((dec != od).any(axis=1)))}"
    assert np.array_equal(e.binds(), o.binds())
    for a, b in zip(e.node_state(), o.node_state()):
        assert np.array_equal(a, b)
    for a, b in zip(e.shares(), o.shares()):
        assert np.array_equal(a, b)
    if seed % 4 == 0:                      # a second cycle from the pristine copy kept in HBM is identical
        e.reset()
        assert np.array_equal(e.run(["allocate", "backfill"]), od)
    e.close()
