"""world_size-2 gloo tests of the task-row sharding layer (kube-batch_amd/dist.py) on CPU.

The engine needs a GPU, so these tests substitute a small deterministic stand-in backend for kb_round_* and check what
the N>1 path adds: the shard partition, the all-gather of candidate lists into one table that is identical on every
rank, the all-reduce of per-node deltas, and that every rank takes the same decisions."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

distmod = importlib.import_module("kube-batch_amd.dist")


def test_shard_bounds_partition():
    for n in (0, 1, 5, 64, 1001):
        for world in (1, 2, 3, 8):
            spans = [distmod.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class FakeBackend:
    """Stands in for kb_round_*: rounds of a toy scheduler whose 'candidates' are a deterministic function of
    (round, matrix row) and whose commit places window row i on candidate node keys[row % n_mrows][0]."""

    def __init__(self, n_nodes=64, R=2, rounds=3, rows=10, mrows=5, L=11):
        self.N, self.R = n_nodes, R
        self.delta_len = n_nodes * (2 * R + 3)
        self.plan = [(rows + k, mrows + k, L + k) for k in range(rounds)]
        self.k = 0
        self.state = np.zeros(self.delta_len)
        self.decs = []
        self.seen_tables = []

    def begin(self, action):
        if self.k >= len(self.plan):
            return 0, 0, 0
        return self.plan[self.k]

    def _key(self, m, j):
        return (self.k + 1) * 1_000_000 + m * 1_000 + j + 1

    def candidates(self, m0, m1, keys):
        _, _, L = self.plan[self.k]
        for m in range(m0, m1):
            keys[m - m0] = torch.tensor([self._key(m, j) for j in range(L)], dtype=torch.int64)

    def commit(self, table, r0, r1, delta):
        n_rows, n_mrows, L = self.plan[self.k]
        self.seen_tables.append(table.clone())
        delta.zero_()
        for i in range(n_rows):
            node = int(table[i % n_mrows, 0].item()) % self.N
            self.decs.append((self.k, i, node))
            self.state[node] -= 1.0
            if r0 <= i < r1:
                delta[node] -= 1.0

    def apply(self, delta):
        # the reduced deltas of all ranks must reproduce the replica's own commit
        n_rows, n_mrows, _ = self.plan[self.k]
        ref = np.zeros(self.delta_len)
        for i in range(n_rows):
            ref[int(self.seen_tables[-1][i % n_mrows, 0].item()) % self.N] -= 1.0
        assert np.array_equal(delta.numpy(), ref), "all-reduced deltas differ from the replicated commit"
        self.k += 1

    def decisions(self):
        return np.array(self.decs, dtype=np.int64)


def _worker(rank, world, port, out_dir, min_rows):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        be = FakeBackend()
        cyc = distmod.ShardedCycle(None, None, backend=be, buffer_device=torch.device("cpu"), actions=("allocate",),
                                   min_rows_per_rank=min_rows)
        dec = cyc.step()
        # every table a rank saw must be the complete, rank-independent candidate table
        for k, tab in enumerate(be.seen_tables):
            n_rows, n_mrows, L = be.plan[k]
            for m in range(n_mrows):
                expect = [(k + 1) * 1_000_000 + m * 1_000 + j + 1 for j in range(L)]
                assert tab[m].tolist() == expect, (rank, k, m)
            assert int(tab[n_mrows:].abs().sum()) == 0          # padding rows stay empty
        np.save(os.path.join(out_dir, f"dec{rank}.npy"), dec)
        np.save(os.path.join(out_dir, f"state{rank}.npy"), be.state)
        assert cyc.rounds == 3
        # min_rows == 0: every round exchanges its rows; a large threshold: every rank evaluates all rows itself
        assert cyc.replicated_rounds == (0 if min_rows == 0 else 3)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("min_rows", [0, 1000])
def test_sharded_cycle_two_ranks_gloo(tmp_path, min_rows):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), min_rows), nprocs=2, join=True)
    d0, d1 = np.load(tmp_path / "dec0.npy"), np.load(tmp_path / "dec1.npy")
    assert d0.shape[0] == 10 + 11 + 12 and np.array_equal(d0, d1)          # replicas take identical decisions
    assert np.array_equal(np.load(tmp_path / "state0.npy"), np.load(tmp_path / "state1.npy"))
