"""The claim behind the selection kernel's bulk commit of backfill rows (kb_commit_sel.hip: bf_bulk), held to the oracle on the CPU — the
emulated device commits row by row, so this is where the rule itself is checked without a GPU (on the device: every backfill case of the
-m gpu suite under KB_COMMIT_KERNEL=select).

backfill.go:50-66 gives a BestEffort task the FIRST node (ascending name) that passes the predicates.  For a plain BestEffort row — empty
InitResreq and Resreq, no host port — a placement changes nothing the predicates read of the node except its pod count.  So within a run of
consecutive backfill rows of one shape: the node that took a row takes the next one too, exactly as long as it has a pod slot left
(predicates.go:127), and otherwise the next row goes to a node with a HIGHER index (everything in front was infeasible and stays so)."""
import importlib

import numpy as np
import pytest

kbm = importlib.import_module("kube-batch_amd")
abi, conf, snapmod = kbm.abi, kbm.conf, kbm.snapshot


def _case(seed):
    rng = np.random.RandomState(31000 + seed)
    p = snapmod.SynthParams(n_tasks=int(rng.randint(300, 3000)), n_nodes=int(rng.randint(4, 120)), n_queues=int(rng.randint(1, 5)), n_res=int(rng.choice([2, 3])),
                            seed=snapmod.SEED_BASE + 3100 + seed, preload_node_frac=float(rng.uniform(0, 0.8)), running_job_frac=float(rng.uniform(0, 0.3)),
                            best_effort_frac=float(rng.uniform(0.2, 0.7)), zone_selector_frac=float(rng.uniform(0, 0.4)), n_zones=int(rng.randint(1, 5)))
    s = snapmod.synth(p)
    tight = rng.uniform(size=s.n_nodes) < 0.7                       # pod caps a few slots above what is there: nodes fill up inside a run
    s.node_max_pods[:] = np.where(tight, s.node_pod_cnt + rng.randint(0, 9, size=s.n_nodes), s.node_max_pods).astype(np.int32)
    s._check()
    return conf.load_scheduler_conf(), s


@pytest.mark.parametrize("seed", range(30))
def test_a_backfill_winner_keeps_winning_until_it_is_full(oracle_mod, seed):
    cfg, snap = _case(seed)
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate"])
    n_alloc = len(o.decisions())
    _, _, _, _, cnt = o.node_state()
    cnt = cnt.astype(np.int64).copy()                                # pod counts backfill starts from
    o.run(["backfill"])
    dec = o.decisions()[n_alloc:]
    o.close()
    assert len(dec) > 20
    R, T = snap.n_res, snap.n_tasks
    plain = (snap.task_init_resreq.reshape(R, T) == 0).all(axis=0) & (snap.task_resreq.reshape(R, T) == 0).all(axis=0)
    shape = lambda t: (int(snap.task_class[t]), int(snap.task_nz_cpu[t]), int(snap.task_nz_mem[t]))
    runs = checked = 0
    prev = None                                                      # (task, node) of the row in front, when it was a plain row of the same job
    for t, n, kind in dec:
        t, n = int(t), int(n)
        assert kind == 0 and n != abi.KB_NONE
        if prev is not None and plain[t] and t == prev[0] + 1 and snap.task_job[t] == snap.task_job[prev[0]] and shape(t) == shape(prev[0]):
            pn = prev[1]
            if cnt[pn] < snap.node_max_pods[pn]:
                assert n == pn, (seed, t, "the node in front still had a pod slot")
            else:
                assert n > pn, (seed, t, "a full node is followed by a later one")
            checked += 1
        else:
            runs += 1
        cnt[n] += 1
        prev = (t, n) if plain[t] else None
    assert checked > 10 and runs > 1
