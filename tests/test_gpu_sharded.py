"""The task-row-sharded cycle on real kernels: world_size 1 (no process group) and world_size 2 with both ranks on the
one GPU of the test box (gloo transport, buffers staged through host memory).  Decisions and binds must equal the oracle
and the unsharded engine."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

kbm = importlib.import_module("kube-batch_amd")
distmod = importlib.import_module("kube-batch_amd.dist")

pytestmark = pytest.mark.gpu


def _snap():
    return kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.05))


def test_sharded_world1_equals_oracle(oracle_mod):
    conf = kbm.conf.load_scheduler_conf()
    snap = _snap()
    cyc = distmod.ShardedCycle(conf, snap, device=0, window=256)
    dec = cyc.step()
    o = oracle_mod.Oracle(conf, snap)
    o.run(["allocate", "backfill"])
    assert np.array_equal(dec, o.decisions())
    assert np.array_equal(cyc.engine.binds(), o.binds())
    for a, b in zip(cyc.engine.node_state(), o.node_state()):
        assert np.array_equal(a, b)
    dec2 = cyc.step()                      # reset + second cycle: identical
    assert np.array_equal(dec2, dec)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        conf = kbm.conf.load_scheduler_conf()
        cyc = distmod.ShardedCycle(conf, _snap(), device=0, window=256, min_rows_per_rank=0)   # always exchange
        dec = cyc.step()
        np.save(os.path.join(out_dir, f"dec{rank}.npy"), dec)
        np.save(os.path.join(out_dir, f"binds{rank}.npy"), cyc.engine.binds())
        st = cyc.engine.stats()
        np.save(os.path.join(out_dir, f"mevals{rank}.npy"), np.array([st["matrix_evals"], st["rounds"]]))
    finally:
        dist.destroy_process_group()


def test_sharded_two_ranks_one_gpu(oracle_mod, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    conf = kbm.conf.load_scheduler_conf()
    o = oracle_mod.Oracle(conf, _snap())
    o.run(["allocate", "backfill"])
    for r in (0, 1):
        assert np.array_equal(np.load(tmp_path / f"dec{r}.npy"), o.decisions()), f"rank {r}"
        assert np.array_equal(np.load(tmp_path / f"binds{r}.npy"), o.binds()), f"rank {r}"
    m0, m1 = np.load(tmp_path / "mevals0.npy"), np.load(tmp_path / "mevals1.npy")
    assert m0[1] == m1[1]                                   # same number of rounds
    assert abs(int(m0[0]) - int(m1[0])) <= m0[1] * 10_000   # each rank evaluated about half of the matrix rows


def _rccl_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["KB_DIST_ALWAYS_COLLECT"] = "1"           # a one-rank group still goes through the collectives
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        class Delayed(distmod.ShardedCycle):
            """every collective is preceded by ~10 ms of busy-wait ON THE COLLECTIVE STREAM: an engine that is not ordered on that
            stream reads the gathered keys / the reduced delta before they exist"""
            def _all_gather_keys(self, local, chunk, L):
                torch.cuda._sleep(20_000_000)
                return super()._all_gather_keys(local, chunk, L)

            def _all_reduce_delta(self):
                torch.cuda._sleep(20_000_000)
                return super()._all_reduce_delta()
        conf = kbm.conf.load_scheduler_conf()
        cyc = Delayed(conf, kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.01)), device=0, window=256, min_rows_per_rank=0)
        assert cyc.stream_ordered and cyc._stream is not None and cyc._stream.cuda_stream != 0
        dec = cyc.step()
        np.save(os.path.join(out_dir, "dec.npy"), dec)
        np.save(os.path.join(out_dir, "binds.npy"), cyc.engine.binds())
    finally:
        dist.destroy_process_group()


def test_rccl_stream_ordering_with_delayed_collectives(oracle_mod, tmp_path):
    """The RCCL path (backend "nccl", world size 1: the only group a one-GPU box allows) skips every host synchronisation inside a
    round because the engine's kernels and the collectives share ONE stream.  Round 2 passed torch's default stream (handle 0), which
    kb_engine_use_stream reads as "the engine's own stream": nothing was ordered, and only launch latency hid it.  Here every
    collective is delayed by a busy-wait kernel on its stream; the decisions must still be the oracle's."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    conf = kbm.conf.load_scheduler_conf()
    o = oracle_mod.Oracle(conf, kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.01)))
    o.run(["allocate", "backfill"])
    assert np.array_equal(np.load(tmp_path / "dec.npy"), o.decisions())
    assert np.array_equal(np.load(tmp_path / "binds.npy"), o.binds())


# ---- BASELINE configs[4] names a third action: preempt (and reclaim) in the sharded mode — every replica runs the evict action, one all-reduce
#      compares the journals (kube-batch_amd/dist.py)
def sharded_evict_inputs(case):
    """-> (conf, snapshot, action order): preempt_test.go's two cases (0, 1), a scaled 1M x 50k cycle with its third action (2), all four actions (3)"""
    import test_gpu_preempt as gp
    import test_pyref_vs_oracle as cases
    if case in (0, 1):      # actions/preempt/preempt_test.go:51-131
        S, fx = kbm.snapshot, kbm.fixtures
        rl = fx.build_resource_list
        if case == 0:
            snap = S.flatten(nodes=[S.Node("n1", rl("3", "3Gi"))],
                             pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"), fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
                                   fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg1"), fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg1")],
                             pod_groups=[S.PodGroup("c1", "pg1", queue="q1")], queues=[S.Queue("q1", 1)])
        else:
            snap = S.flatten(nodes=[S.Node("n1", rl("2", "2G"))],
                             pods=[fx.build_pod("c1", "preemptee1", "n1", "Running", rl("1", "1G"), "pg1"), fx.build_pod("c1", "preemptee2", "n1", "Running", rl("1", "1G"), "pg1"),
                                   fx.build_pod("c1", "preemptor1", "", "Pending", rl("1", "1G"), "pg2"), fx.build_pod("c1", "preemptor2", "", "Pending", rl("1", "1G"), "pg2")],
                             pod_groups=[S.PodGroup("c1", "pg1", queue="q1"), S.PodGroup("c1", "pg2", queue="q1")], queues=[S.Queue("q1", 1)])
        return gp._preempt_tiers(), snap, ["preempt"]
    order = ["allocate", "backfill", "preempt"] if case == 2 else ["reclaim", "allocate", "backfill", "preempt"]
    cfg = kbm.conf.load_scheduler_conf(cases.CONF_FULL.format(actions=", ".join(order)))
    return cfg, kbm.snapshot.synth(kbm.snapshot.synth_config(5, 0.004 if case == 2 else 0.002)), order



def check_sharded_evict_outputs(oracle_mod, out_dir, case):
    """what both ranks saved == the oracle: decisions of allocate / backfill, Statement journal, evictions, binds, node state"""
    cfg, snap, order = sharded_evict_inputs(case)
    o = oracle_mod.Oracle(cfg, snap)
    n0, odec = 0, []
    for a in order:                                   # the engine's decision list: allocate / backfill only (reclaim's ssn.Pipeline calls are journal entries)
        o.run([a])
        d = o.decisions()[n0:]
        n0 += len(d)
        if a in ("allocate", "backfill"):
            odec.append(d)
    odec = np.concatenate(odec) if odec else np.zeros((0, 3), np.uint32)
    for r in (0, 1):
        assert np.array_equal(np.load(out_dir / f"dec{r}.npy").reshape(-1, 3), odec), f"rank {r}"
        assert np.array_equal(np.load(out_dir / f"binds{r}.npy"), o.binds()), f"rank {r}"
        j = np.load(out_dir / f"journal{r}.npy")
        assert j.shape == o.journal().shape and np.array_equal(j, o.journal()), f"rank {r}: journal"
        assert [int(t) for t in np.load(out_dir / f"evict{r}.npy")] == [int(t) for t in o.evictions()], f"rank {r}"
        for i, a in enumerate(o.node_state()):
            assert np.array_equal(np.load(out_dir / f"node{i}_{r}.npy"), a), f"rank {r}: node state {i}"
    if case < 2:
        assert len(o.evictions()) == (1, 2)[case]      # what preempt_test.go's FakeEvictor records


def _evict_worker(rank, world, port, out_dir, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, snap, order = sharded_evict_inputs(case)
        cyc = distmod.ShardedCycle(cfg, snap, device=0, window=256, min_rows_per_rank=0, actions=order)
        dec = cyc.step()
        eng = cyc.engine
        assert cyc.evict_actions == sum(a in ("preempt", "reclaim") for a in order)
        np.save(os.path.join(out_dir, f"dec{rank}.npy"), dec)
        np.save(os.path.join(out_dir, f"binds{rank}.npy"), eng.binds())
        np.save(os.path.join(out_dir, f"journal{rank}.npy"), eng.journal())
        np.save(os.path.join(out_dir, f"evict{rank}.npy"), np.array(eng.evictions(), np.uint32))
        for i, a in enumerate(eng.node_state()):
            np.save(os.path.join(out_dir, f"node{i}_{rank}.npy"), a)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", range(4))
def test_sharded_cycle_with_evict_actions_two_ranks_one_gpu(oracle_mod, tmp_path, case):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_evict_worker, args=(2, port, str(tmp_path), case), nprocs=2, join=True)
    check_sharded_evict_outputs(oracle_mod, tmp_path, case)
