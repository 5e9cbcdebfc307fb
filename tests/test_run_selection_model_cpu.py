"""DESIGN section 9.2's parallel selection for a run of identical rows (tests/run_selection_model.py) against the serial loop it replaces, on random
clusters: the same (node, Allocate / Pipeline) per row, in the same order — with Balanced's non-monotone scores, ties between nodes, nodes that
run out of Idle and continue on Releasing, pod-count caps, runs longer than the cluster can take."""
import numpy as np
import pytest

import run_selection_model as m

WEIGHTS = [(1, 0, 1), (0, 5, 1), (1, 0, 0), (1, 1, 1), (0, 0, 1), (2, 3, 5)]


def _cluster(rng, N):
    cores = rng.choice([4, 8, 16, 32, 64, 96, 128], size=N)
    gib = rng.choice([16, 32, 64, 128, 256, 512], size=N)
    ac, am = cores * 1000, gib * (1 << 30)
    used_c = (ac * rng.choice([0.0, 0.1, 0.3, 0.5, 0.8, 0.95], size=N)).astype(np.int64) // 100 * 100
    used_m = (am * rng.choice([0.0, 0.1, 0.3, 0.5, 0.8, 0.95], size=N)).astype(np.int64) // (1 << 20) * (1 << 20)
    idle = np.stack([ac - used_c, am - used_m], 1).astype(np.float64)
    rel = np.stack([used_c * rng.choice([0.0, 0.0, 0.5, 1.0], size=N), used_m * rng.choice([0.0, 0.0, 0.5, 1.0], size=N)], 1).astype(np.float64)
    podcnt = rng.randint(0, 20, size=N)
    maxpods = rng.choice([110, 110, 30, 22], size=N)
    return m.Cluster(ac, am, used_c, used_m, idle, rel, podcnt, maxpods)


def _shape(rng):
    cpu = int(rng.choice([100, 250, 500, 1000, 2000, 4000, 8000]))
    mem = int(rng.choice([128, 256, 512, 1024, 4096, 16384])) << 20
    return (cpu, mem, cpu, mem)


@pytest.mark.parametrize("seed", range(60))
def test_selection_equals_the_serial_loop(seed):
    rng = np.random.RandomState(seed)
    N = int(rng.choice([1, 2, 5, 20, 60]))
    c = _cluster(rng, N)
    for _ in range(4):
        shape, r, w = _shape(rng), int(rng.choice([1, 2, 7, 32, 64])), WEIGHTS[rng.randint(len(WEIGHTS))]
        want = m.serial_run(c, shape, r, w)
        assert m.selected_run(c, shape, r, w) == want, (seed, N, shape, r, w)
        assert m.selected_run(c, shape, r, w, lanes=r) == want          # the r best initial keys hold every winner


def test_identical_nodes_and_a_key_that_rises():
    """every node the same (all keys tie: lowest index first, and it keeps winning while its key does not fall below the others'), and a shape whose
    Balanced score RISES with the first placements (cpu-heavy node state, memory-heavy request): the prefix minimum keeps the node in front"""
    N = 6
    c = m.Cluster([16000] * N, [64 << 30] * N, [8000] * N, [4 << 30] * N, [[8000.0, float(60 << 30)]] * N, [[0.0, 0.0]] * N, [0] * N, [110] * N)
    shape = (100, 4 << 30, 100, 4 << 30)
    seq = m.node_sequence(c, 0, shape, 6, (0, 0, 1))
    assert any(b[0] > a[0] for a, b in zip(seq, seq[1:]))            # the premise "keys only fall" does not hold here ...
    for w in WEIGHTS:
        for r in (1, 5, 40, 100):
            assert m.selected_run(c, shape, r, w) == m.serial_run(c, shape, r, w)      # ... and the selection is still the loop



@pytest.mark.parametrize("seed", range(80))
def test_shots_equal_the_serial_loop(seed):
    """round 6: bounded tables, one rank per shot, only the FINAL picks committed (table_run) against the serial loop — every lane budget from one
    contender with a deep table to 32 contenders with two entries each.  A Pipeline ends the serial loop's round too: the comparison stops behind
    the first one."""
    rng = np.random.RandomState(2000 + seed)
    N = int(rng.choice([1, 2, 5, 20, 60]))
    c = _cluster(rng, N)
    for _ in range(4):
        shape, r, w = _shape(rng), int(rng.choice([1, 2, 7, 32])), WEIGHTS[rng.randint(len(WEIGHTS))]
        want = m.serial_run(c, shape, r, w)
        for i, (_, kind) in enumerate(want):
            if kind == m.PIPELINE:
                want = want[:i + 1]
                break
        for cap, lanes in ((32, 64), (8, 64), (2, 64), (1, 64), (32, 32), (3, 8), (1, 2), (5, 5)):
            assert m.table_run(c, shape, r, w, cap, lanes) == want, (seed, N, shape, r, w, cap, lanes)
