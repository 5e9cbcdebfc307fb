"""The exactness claim behind the overlapped candidate lists (kb_kernels.hip: k_repair), stated on plain integers:

  a round's candidate list = the first L entries of its nodes in descending key order (key 0 = infeasible; keys are distinct, the node index is
  part of them).  The overlapped launches see the TRUE key of every node the predecessor round leaves alone and an ARBITRARY key (old, new,
  torn: anything, feasible or not) for the at most n_prev nodes it changes, and keep the first n_prev + L entries of that stale order.
  Dropping the predecessor's nodes from the stale list and merging their true keys in gives the true list.

The bound is tight: one entry less and a counterexample exists (the second test builds it).  This is the model; the kernel's own merge (block
scan + binary search + counts) is checked against the oracle on the device (`-m gpu`) and, as a contract, on the emulated device."""
import random

import pytest


def top(keys, n):
    """descending, feasible only, at most n"""
    return sorted((k for k in keys if k), reverse=True)[:n]


def repaired(stale_keys, true_keys, changed, L, stale_len):
    node_of = {k: i for i, k in enumerate(stale_keys) if k}
    stale_list = top(stale_keys, stale_len)
    survivors = [k for k in stale_list if node_of[k] not in changed]
    fresh = [true_keys[i] for i in changed if true_keys[i]]
    return sorted(survivors + fresh, reverse=True)[:L]


def make_case(rng, n_nodes, n_prev, p_infeasible):
    # distinct keys: (score << 20) | (2^20 - 1 - node), like KB_KEY's order (score descending, node ascending)
    def key(score, node):
        return (score + 1) << 20 | (0xFFFFF - node)
    true_keys = [0 if rng.random() < p_infeasible else key(rng.randrange(0, 40), i) for i in range(n_nodes)]
    changed = set(rng.sample(range(n_nodes), min(n_prev, n_nodes)))
    stale_keys = list(true_keys)
    for i in changed:   # whatever the overlapped launch happened to see: infeasible, the old key, a wildly better or worse one
        stale_keys[i] = rng.choice([0, true_keys[i], key(rng.randrange(0, 40), i), key(10_000, i), key(0, i)])
    return true_keys, stale_keys, changed


@pytest.mark.parametrize("seed", range(200))
def test_repaired_list_is_the_true_list(seed):
    rng = random.Random(seed)
    n_nodes = rng.choice([1, 2, 5, 40, 300, 2000])
    n_prev = rng.choice([0, 1, 3, 17, 64, 256])
    L = rng.choice([1, 2, 9, 65, 257])
    true_keys, stale_keys, changed = make_case(rng, n_nodes, n_prev, rng.choice([0.0, 0.3, 0.95]))
    assert repaired(stale_keys, true_keys, changed, L, len(changed) + L) == top(true_keys, L)
    assert repaired(stale_keys, true_keys, changed, L, n_prev + L) == top(true_keys, L)      # what the engine asks for (n_prev rows >= nodes changed)


def test_the_bound_is_tight():
    """n_prev changed nodes whose stale keys all sit on top push a needed clean node out of a stale list that is one entry short."""
    n_prev, L, n_nodes = 4, 3, 20
    key = lambda score, node: (score + 1) << 20 | (0xFFFFF - node)
    true_keys = [key(5, i) for i in range(n_nodes)]
    changed = set(range(10, 10 + n_prev))
    stale_keys = list(true_keys)
    for i in changed:
        stale_keys[i] = key(1000, i)            # looked great to the overlapped launch ...
        true_keys[i] = 0                        # ... and is full now
    want = top(true_keys, L)
    assert repaired(stale_keys, true_keys, changed, L, n_prev + L) == want
    assert repaired(stale_keys, true_keys, changed, L, n_prev + L - 1) != want


def ranks_by_place_marks(stale_list, dead, fresh, threads=1024):
    """k_repair's merge as the kernel computes it (kb_kernels.hip): `stale_list` descending and padded with zeros to `threads` entries, `dead[i]` = entry i
    is one of the predecessor's nodes, `fresh` = its new keys (0: infeasible).  Returns {rank: key}."""
    assert len(stale_list) == threads and len(dead) == threads
    alive = [k != 0 and not dd for k, dd in zip(stale_list, dead)]
    alive_before = [0] * (threads + 1)
    for i in range(threads):
        alive_before[i + 1] = alive_before[i] + (1 if alive[i] else 0)
    marks = [0] * (threads + 1)
    place = {}
    for j, f in enumerate(fresh):
        if not f:
            continue
        lo, hi = 0, threads                       # stale entries above f: binary search on the descending list (zeros are never above)
        while lo < hi:
            mid = (lo + hi) >> 1
            if stale_list[mid] > f:
                lo = mid + 1
            else:
                hi = mid
        place[j] = lo
        marks[lo] += 1
    out = {}
    run = 0
    for i in range(threads):
        run += marks[i]                           # inclusive prefix sum: new keys whose place is at or in front of entry i
        if alive[i]:
            rank = alive_before[i] + run
            assert rank not in out
            out[rank] = stale_list[i]
    n_parts = 4
    np8 = (len(fresh) + 7) & ~7
    padded = list(fresh) + [0] * (np8 - len(fresh))
    per = (((np8 + 3) >> 2) + 7) & ~7
    for j, f in enumerate(fresh):
        if not f:
            continue
        above = 0
        for part in range(n_parts):               # thread (part, j): one quarter of the new keys, eight at a time
            i0, i1 = part * per, min(np8, part * per + per)
            for i in range(i0, i1, 8):
                above += sum(1 for g in padded[i:i + 8] if g > f)
        rank = alive_before[place[j]] + above
        assert rank not in out
        out[rank] = f
    return out


@pytest.mark.parametrize("seed", range(300))
def test_ranks_from_place_marks_are_the_merged_order(seed):
    """The rank arithmetic of k_repair (place of a new key in the stale list by binary search, a mark there, prefix sums of marks and of survivor
    flags) gives every survivor and every new key its position in the merged descending order — including new keys equal to their node's own
    stale entry, lists that end early, infeasible new keys, and n_prev from 0 to 256."""
    rng = random.Random(1000 + seed)
    n_nodes = rng.choice([1, 2, 5, 40, 300, 2000])
    n_prev = rng.choice([0, 1, 3, 7, 8, 9, 17, 64, 255, 256])
    L = rng.choice([1, 2, 9, 65, 257])
    true_keys, stale_keys, changed = make_case(rng, n_nodes, n_prev, rng.choice([0.0, 0.3, 0.95]))
    node_of = {}
    for i, k in enumerate(stale_keys):
        if k:
            node_of[k] = i
    stale_len = n_prev + L
    stale_list = top(stale_keys, stale_len)
    dead = [node_of[k] in changed for k in stale_list]
    threads = 1024
    stale_list = stale_list + [0] * (threads - len(stale_list))
    dead = dead + [False] * (threads - len(dead))
    # decision records: a node may have taken several rows (one owner each: the other records carry key 0), and the records are in row order
    records = list(changed)
    if records:
        records += [rng.choice(records) for _ in range(rng.randrange(0, 256 - len(records) + 1))]
    rng.shuffle(records)
    seen, fresh = set(), []
    for n in records:
        fresh.append(true_keys[n] if n not in seen else 0)
        seen.add(n)
    got = ranks_by_place_marks(stale_list, dead, fresh, threads)
    merged = sorted([k for k, dd in zip(stale_list, dead) if k and not dd] + [f for f in fresh if f], reverse=True)
    assert sorted(got) == list(range(len(merged)))
    assert [got[r] for r in range(len(merged))] == merged
    assert merged[:L] == top(true_keys, L)
