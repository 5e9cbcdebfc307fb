"""Inter-pod (anti)affinity on the device — predicate p8 in the matrix kernel, InterPodAffinityPriority in k_interpod, the counters
advanced by the commit kernels' epilogue — against the C oracle, through the C ABI, on the random clusters of
tests/test_interpod_oracle_cpu.py (which pin the oracle to a second restatement, itself pinned to the object-level Go semantics by
tests/test_interpod_cpu.py) and on larger ones.  Needs a real MI355X: -m gpu."""
import importlib

import numpy as np
import pytest

from test_gpu_parity import assert_same_outcome
from test_interpod_cpu import random_cluster
from test_interpod_oracle_cpu import CONFS, interpod_case

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
abi, conf, snapmod = kbm.abi, kbm.conf, kbm.snapshot

pytestmark = pytest.mark.gpu


def _run(oracle_mod, cfg, snap, **ekw):
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate", "backfill"])
    e = engine.Engine(cfg, **ekw)
    e.load(snap)
    dec = e.run(["allocate", "backfill"])
    assert_same_outcome(o, e, dec)
    # a second cycle after kb_session_reset restores the live counters too
    e.reset()
    dec2 = np.concatenate([e.run_allocate(), e.run_backfill()]) if hasattr(e, "run_allocate") else e.run(["allocate", "backfill"])
    assert np.array_equal(dec2[:, :3], dec[:, :3])
    o.close(); e.close()
    return len(dec)


@pytest.mark.parametrize("seed", list(range(60)) + [1000 + i for i in range(6)])
def test_engine_equals_oracle_with_interpod_affinity(oracle_mod, commit_kernel, seed):
    try:
        cfg, snap = interpod_case(seed % 1000, wide=seed >= 1000)   # wide: masks of several 64-bit words
    except snapmod.UnsupportedSnapshot as e:
        pytest.skip(str(e))
    if snap.interpod is None:
        pytest.skip("no pod-affinity term drawn")
    _run(oracle_mod, cfg, snap, window=[0, 64, 16][seed % 3])


@pytest.mark.parametrize("seed", range(6))
def test_engine_equals_oracle_on_larger_interpod_clusters(oracle_mod, commit_kernel, seed):
    rng = np.random.RandomState(70 + seed)
    for attempt in range(8):
        nodes, pods, groups, queues = random_cluster(5000 + 10 * seed + attempt, n_nodes=int(rng.randint(60, 200)), n_pods=int(rng.randint(400, 1200)),
                                                     n_jobs=int(rng.randint(20, 80)), tight=True, pool_size=4, weights=(10, 40), templates=True)
        try:
            snap = snapmod.flatten(nodes, pods, groups, queues)
        except snapmod.UnsupportedSnapshot:
            continue
        if snap.interpod is not None:
            break
    else:
        pytest.skip("no supported cluster drawn")
    n = _run(oracle_mod, conf.load_scheduler_conf(CONFS[seed % 2]), snap)
    assert n > 50


@pytest.mark.parametrize("seed", range(6))
def test_engine_equals_oracle_when_few_pods_are_subjects(oracle_mod, commit_kernel, seed):
    """templates that name one workload each: most pods are not inter-pod subjects and share windows with the few that are"""
    rng = np.random.RandomState(170 + seed)
    nodes, pods, groups, queues = random_cluster(7000 + seed, n_nodes=int(rng.randint(60, 200)), n_pods=int(rng.randint(600, 1500)),
                                                 n_jobs=int(rng.randint(30, 60)), tight=True, pool_size=6, weights=(10, 40), templates=True, narrow=True)
    snap = snapmod.flatten(nodes, pods, groups, queues)
    assert snap.interpod is not None
    ip = snap.interpod
    subjects = int(((ip["task_forbid"] != 0).any(axis=1) | (ip["task_require"] != 0xFFFF) | (ip["task_sig"] != abi.KB_NONE)).sum())
    assert 0 < subjects < snap.n_tasks
    n = _run(oracle_mod, conf.load_scheduler_conf(CONFS[seed % 2]), snap)
    assert n > 50


@pytest.mark.parametrize("seed", range(150))
def test_evict_actions_with_interpod_terms_equal_the_oracle(oracle_mod, commit_kernel, seed):
    """preempt / reclaim in sessions with inter-pod (anti)affinity terms (actions/preempt/preempt.go:182 calls the same ssn.PredicateFn, i.e.
    plugins/predicates/predicates.go:249-262): the evict machine keeps the kb_interpod counts on the host (an eviction takes its victim out
    of the predicate's pod list, a Pipeline adds the preemptor with an empty Spec.NodeName, a discarded statement undoes both), puts them
    on the device in front of every list it asks for and rebuilds every list after a change: journal, evictions, statuses, node state and
    shares equal the oracle's.  Round 3 wrote it behind a switch; since its first device run (round 4) there is no switch and no refusal."""
    import test_gpu_preempt as gp
    from test_interpod_oracle_cpu import interpod_evict_case
    cfg, snap, order = interpod_evict_case(seed)     # every one of the 150 seeds draws a cluster the flattener takes, with pod-affinity terms:
    assert snap.interpod is not None                 # a seed that stops doing so fails here instead of passing with nothing compared
    gp._run_both(oracle_mod, cfg, snap, order, seed)


def test_more_than_1024_counters_and_classes(oracle_mod):
    """round 3's envelope stopped at 1024 inter-pod predicate counters / priority classes per session; the tables were multi-word already
    (vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/predicates/predicates.go:1153-1175,1261-1290 know no such limit).  More than 2 x 1024
    of each: the engine equals the oracle (which tests/test_interpod_oracle_cpu.py holds to the second restatement at this width)."""
    from test_interpod_oracle_cpu import very_wide_interpod_case
    cfg, snap = very_wide_interpod_case()
    assert _run(oracle_mod, cfg, snap) > 500


def test_hand_derived_known_answer_on_the_device(oracle_mod, commit_kernel):
    """tests/test_manifests_cpu.py derives the outcome of this cluster by hand from the Go code (first-node ties, the empty
    Spec.NodeName quirk of nodeorder's cachedNodeInfo, only the feasible nodes' pods being seen): the HIP path must produce it too."""
    from test_manifests_cpu import interpod_kat_text
    manifests = importlib.import_module("kube-batch_amd.manifests")
    snap = manifests.load_snapshot(interpod_kat_text())
    e = engine.Engine(conf.load_scheduler_conf())
    e.load(snap)
    e.run(["allocate", "backfill"])
    assert snap.bind_map(e.binds()) == {"ns/web-0": "n1", "ns/cache-0": "n1", "ns/web-1": "n2", "ns/web-2": "n3"}
    e.close()


@pytest.mark.parametrize("seed", [1, 4, 6, 9])
def test_matrix_with_interpod_predicate_and_priority(oracle_mod, seed):
    """kb_eval_matrix: mask bits (predicate p8 included) and u16 scores (InterPodAffinityPriority included) of every (task, node) pair
    against the session-open state, both fit modes; then again after the allocate action has advanced the counters"""
    cfg, snap = interpod_case(seed)
    assert snap.interpod is not None
    o = oracle_mod.Oracle(cfg, snap)
    e = engine.Engine(cfg)
    e.load(snap)
    for stage in range(2):
        for fit in (1, 0):
            em, es = e.eval_matrix(0, snap.n_tasks, fit)
            om, os_ = o.eval_matrix(0, snap.n_tasks, fit)
            assert np.array_equal(em, om), (seed, stage, fit)
            assert np.array_equal(es, os_), (seed, stage, fit)
        if stage == 0:
            o.run(["allocate"])
            e.run(["allocate"])
    o.close(); e.close()
