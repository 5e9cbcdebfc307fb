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
