"""The engine's round protocol as an executable model (tests/round_model.py): candidate lists from the round-start state, dirty-node
repair, speculation breaks, L = W + 1 — checked against the sequential reference loop on synthetic and adversarial snapshots,
single process and sharded over two gloo ranks through kube-batch_amd/dist.py."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pyref
import rawgen
import round_model
import test_host_order_cpu as hoc
import test_pyref_vs_oracle as cases

kbm = importlib.import_module("kube-batch_amd")
conf = kbm.conf
distmod = importlib.import_module("kube-batch_amd.dist")
harness = hoc.harness


def _reference(cfg, snap):
    return pyref.Session(cases._tiers(cfg), snap).run(["allocate"])


def _check(L, cfg, snap, seed):
    ref = _reference(cfg, snap)
    rng = np.random.RandomState(seed)
    for window in (1, int(rng.choice([2, 3, 5])), int(rng.choice([8, 16, 64])), 256):
        m = round_model.RoundModel(L, cfg, snap, cases._tiers(cfg), window).run_single()
        assert m.decs == ref.decisions, (seed, window)
        assert m.popped == ref.popped, (seed, window)
        assert m.p.binds == ref.binds, (seed, window)
        for n in range(snap.n_nodes):
            for d in range(snap.n_res):
                assert m.p.idle[n].get(d) == ref.idle[n].get(d) and m.p.rel[n].get(d) == ref.rel[n].get(d)
        m.close()
    # the feasibility probe at speculation breaks marks dead shapes early; the decisions must not move, and it may only
    # ever remove breaks
    for window in (int(rng.choice([3, 8])), 64):
        plain = round_model.RoundModel(L, cfg, snap, cases._tiers(cfg), window).run_single()
        m = round_model.RoundModel(L, cfg, snap, cases._tiers(cfg), window).run_single(probe=True)
        assert m.decs == ref.decisions and m.popped == ref.popped and m.p.binds == ref.binds, (seed, window, "probe")
        assert m.breaks <= plain.breaks, (seed, window, m.breaks, plain.breaks)
        plain.close(); m.close()
    # lists one round stale (built while the previous window commits), 2W + 1 entries, previous round's nodes dirty
    for window in (int(rng.choice([2, 4, 7])), int(rng.choice([16, 48]))):
        m = round_model.RoundModel(L, cfg, snap, cases._tiers(cfg), window).run_single_stale()
        assert m.decs == ref.decisions, (seed, window, "stale")
        assert m.popped == ref.popped, (seed, window, "stale")
        m.close()


@pytest.mark.parametrize("seed", range(30))
def test_round_model_on_synthetic_clusters(harness, seed):
    cfg, snap = cases._case(seed)
    _check(harness, cfg, snap, seed)


@pytest.mark.parametrize("seed", range(150))
def test_round_model_on_adversarial_snapshots(harness, seed):
    snap = rawgen.raw_snapshot(seed)
    rng = np.random.RandomState(seed)
    wl, wm, wa, wb = [int(x) for x in rng.choice([0, 1, 1, 2, 5], size=4)]
    cfg = conf.load_scheduler_conf(cases.CONF_TMPL.format(wl=wl, wm=wm, wa=wa, wb=wb))
    try:
        _reference(cfg, snap)
    except ArithmeticError:
        pytest.skip("the reference would panic on this snapshot")
    _check(harness, cfg, snap, seed)


def _worker(rank, world, port, out_dir, seed, min_rows):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = hoc._bind(__import__("ctypes").CDLL(os.path.join(hoc.HERE, "host_harness", "build", "liborderharness.so")))
        cfg, snap = cases._case(seed)
        be = round_model.RoundModel(L, cfg, snap, cases._tiers(cfg), window=24)
        cyc = distmod.ShardedCycle(None, None, backend=be, buffer_device=torch.device("cpu"), actions=("allocate",), min_rows_per_rank=min_rows)
        dec = cyc.run_action(0)
        np.save(os.path.join(out_dir, f"dec{rank}.npy"), dec)
        np.save(os.path.join(out_dir, f"rounds{rank}.npy"), np.array([cyc.rounds, cyc.replicated_rounds]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed,min_rows", [(1, 0), (11, 0), (14, 0), (11, 4)])
def test_sharded_round_model_two_ranks_equals_the_reference(harness, tmp_path, seed, min_rows):
    """dist.py end to end on CPU: two gloo ranks, each evaluating its shard of the window's distinct shapes, all-gather of the
    candidate lists, replicated commit, all-reduce of the per-node deltas — the decisions equal the sequential reference loop's."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), seed, min_rows), nprocs=2, join=True)
    cfg, snap = cases._case(seed)
    ref = _reference(cfg, snap)
    d0, d1 = np.load(tmp_path / "dec0.npy"), np.load(tmp_path / "dec1.npy")
    assert np.array_equal(d0, d1)
    assert d0.tolist() == [list(x) for x in ref.decisions]
    r = np.load(tmp_path / "rounds0.npy")
    assert r[0] > 0 and (r[1] < r[0] if min_rows == 0 else True)       # with min_rows 0 every round really exchanged candidate lists
