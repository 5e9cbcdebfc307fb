"""Adversarial raw session snapshots for differential tests (test infrastructure): values sit on and around the reference's
epsilons (10 milli, 10 MiB), capacities of 0, scalar keys present with tiny or zero values, nil scalar maps on either side, equal
creation stamps (UID tie-breaks), minAvailable of 0 / size / size + 1, zero queue weights, full and over-full nodes.  The arrays are
NOT a consistent cluster (idle is not allocatable minus usage): the restatements take them as given, which is the point."""
import importlib

import numpy as np

kbm = importlib.import_module("kube-batch_amd")
abi, snapmod = kbm.abi, kbm.snapshot

MiB, GiB = float(1 << 20), float(1 << 30)


def raw_snapshot(seed: int):
    rng = np.random.RandomState(424242 + seed)
    R = int(rng.choice([2, 3, 4, 5]))
    N = int(rng.randint(1, 14))
    Q = int(rng.randint(1, 5))
    J = int(rng.randint(1, 30))
    sizes = rng.randint(1, 7, size=J)
    T = int(sizes.sum())
    begin = np.zeros(J + 1, np.uint32)
    begin[1:] = np.cumsum(sizes)
    task_job = np.repeat(np.arange(J, dtype=np.uint32), sizes)

    # nodes
    alloc = np.zeros((R, N))
    alloc[0] = rng.choice([0, 5, 1000, 2000, 4000, 64000], size=N, p=[.05, .05, .2, .3, .3, .1])
    alloc[1] = rng.choice([0, 5 * MiB, 1 * GiB, 4 * GiB, 8 * GiB], size=N, p=[.05, .05, .3, .3, .3])
    nmask = np.zeros(N, np.uint32)
    for d in range(2, R):
        has = rng.uniform(size=N) < 0.6
        alloc[d] = np.where(has, rng.choice([0, 5, 10, 1000, 4000], size=N), 0)
        nmask |= has.astype(np.uint32) << np.uint32(d - 2)
    used = np.zeros((R, N))
    used[0] = rng.choice([0, 0, 100, 500, 1000, 1005], size=N)
    used[1] = rng.choice([0, 0, 128 * MiB, 1 * GiB, 1 * GiB + 5 * MiB], size=N)
    idle = alloc - used                                   # may be slightly negative, like after an epsilon-admitted Sub
    for d in range(2, R):
        idle[d] = np.where((nmask >> np.uint32(d - 2)) & 1, alloc[d] - rng.choice([0, 0, 5, 1000], size=N), 0)
    # a real Idle never drops below -epsilon (Sub only admits requests LessEqual accepts); below that UpdateTask is glog.Fatalf
    idle[0] = np.maximum(idle[0], -5)
    idle[1] = np.maximum(idle[1], -5 * MiB)
    for d in range(2, R):
        idle[d] = np.maximum(idle[d], -5)
    rel = np.zeros((R, N))
    relon = rng.uniform(size=N) < 0.4
    rel[0] = np.where(relon, rng.choice([5, 100, 1000, 4000], size=N), 0)
    rel[1] = np.where(relon, rng.choice([5 * MiB, 1 * GiB, 4 * GiB], size=N), 0)
    for d in range(2, R):
        rel[d] = np.where(relon & (rng.uniform(size=N) < 0.5), rng.choice([5, 11, 1000], size=N), 0)
    acpu = np.where(rng.uniform(size=N) < 0.1, 0, alloc[0]).astype(np.int64)
    amem = np.where(rng.uniform(size=N) < 0.1, 0, alloc[1]).astype(np.int64)
    nzc = rng.choice([0, 100, 900, 1000, 3999, 5000], size=N).astype(np.int64)
    nzm = (rng.choice([0, 200, 512, 1024, 8192], size=N) * MiB).astype(np.int64)
    maxpods = rng.choice([0, 1, 3, 110, 110], size=N).astype(np.int32)
    podcnt = rng.randint(0, 4, size=N).astype(np.int32)

    # jobs / tasks: one request shape per job, some tasks bumped individually
    jcpu = rng.choice([0, 5, 10, 11, 100, 1000, 1005, 2000], size=J)
    jmem = rng.choice([0, 5 * MiB, 10 * MiB, 10 * MiB + 1, 256 * MiB, 1 * GiB], size=J)
    res = np.zeros((R, T))
    res[0], res[1] = jcpu[task_job], jmem[task_job]
    tmask = np.zeros(T, np.uint32)
    for d in range(2, R):
        jhas = rng.uniform(size=J) < 0.4
        jval = rng.choice([0, 5, 10, 11, 1000, 2000], size=J)
        res[d] = np.where(jhas, jval, 0)[task_job]
        tmask |= jhas[task_job].astype(np.uint32) << np.uint32(d - 2)
    init = res.copy()
    bump = rng.uniform(size=T) < 0.2
    init[0] = np.where(bump, res[0] + rng.choice([5, 100, 1000], size=T), res[0])
    for d in range(2, R):                                    # an init container may raise a scalar the containers do not name
        b = rng.uniform(size=T) < 0.05
        init[d] = np.where(b, np.maximum(res[d], 1000), init[d])
    tnzc = np.where(res[0] == 0, 100, res[0]).astype(np.int64)
    tnzm = np.where(res[1] == 0, 200 * MiB, res[1]).astype(np.int64)
    status = rng.choice([abi.TASK_PENDING] * 12 + [abi.TASK_RUNNING, abi.TASK_RUNNING, abi.TASK_BOUND, abi.TASK_ALLOCATED, abi.TASK_RELEASING,
                                                      abi.TASK_SUCCEEDED, abi.TASK_FAILED, abi.TASK_UNKNOWN, abi.TASK_PIPELINED], size=T).astype(np.uint8)
    tnode = np.where(np.isin(status, [abi.TASK_PENDING, abi.TASK_SUCCEEDED, abi.TASK_FAILED, abi.TASK_UNKNOWN]), abi.KB_NONE,
                     rng.randint(0, N, size=T)).astype(np.uint32)
    n_tc, n_nc = int(rng.randint(1, 4)), int(rng.randint(1, 4))
    compat = np.zeros((n_tc * n_nc + 7) // 8, np.uint8)
    for b in range(n_tc * n_nc):
        if rng.uniform() < 0.8:
            compat[b >> 3] |= 1 << (b & 7)
    minav = np.where(rng.uniform(size=J) < 0.6, sizes, rng.choice([0, 1, 2], size=J))
    minav = np.where(rng.uniform(size=J) < 0.08, sizes + 1, minav).astype(np.int32)
    s = snapmod.SessionSnapshot(
        n_res=R, n_nodes=N, n_tasks=T, n_jobs=J, n_queues=Q, n_task_classes=n_tc, n_node_classes=n_nc,
        node_idle=idle, node_releasing=rel, node_allocatable=alloc, node_scalar_mask=nmask, node_alloc_cpu=acpu, node_alloc_mem=amem,
        node_nz_cpu=nzc, node_nz_mem=nzm, node_max_pods=maxpods, node_pod_cnt=podcnt, node_class=rng.randint(0, n_nc, size=N).astype(np.uint32),
        task_resreq=res, task_init_resreq=init, task_scalar_mask=tmask, task_nz_cpu=tnzc, task_nz_mem=tnzm, task_job=task_job,
        task_class=rng.randint(0, n_tc, size=T).astype(np.uint32), task_priority=rng.choice([1, 1, 1, 5], size=T).astype(np.int32),
        task_creation=(1_600_000_000 + rng.randint(0, 3, size=T)).astype(np.int64), task_status=status, task_node=tnode,
        job_task_begin=begin, job_queue=rng.randint(0, Q, size=J).astype(np.uint32), job_min_available=minav,
        job_priority=rng.choice([0, 0, 100], size=J).astype(np.int32), job_creation=(1_600_000_000 + rng.randint(0, 3, size=J)).astype(np.int64),
        queue_weight=rng.choice([0, 1, 1, 2, 8], size=Q).astype(np.int32), queue_creation=rng.randint(0, 2, size=Q).astype(np.int64),
        class_compat=compat if rng.uniform() < 0.8 else None)
    if rng.uniform() < 0.3:
        s.class_affinity = rng.choice([0, 0, 1, 7, 100], size=(n_tc, n_nc)).astype(np.int32)
    if rng.uniform() < 0.3:
        s.task_port_want = (np.uint64(1) << rng.randint(0, 4, size=T).astype(np.uint64)) * (rng.uniform(size=T) < 0.4).astype(np.uint64)
        s.task_port_conflict = s.task_port_want | ((rng.uniform(size=T) < 0.2).astype(np.uint64) * np.uint64(0xF))
        s.node_ports = rng.randint(0, 16, size=N).astype(np.uint64) * (rng.uniform(size=N) < 0.3).astype(np.uint64)
        for t in range(T):                                   # the one consistency rule both restatements rely on: a node's used
            if tnode[t] != abi.KB_NONE:                      # ports include those of the session tasks already on it
                s.node_ports[tnode[t]] |= s.task_port_want[t]
    s._check()
    return s


def widen_ports(s, seed, words=3, p_task=0.5, p_node=0.4, low_share=0.5):
    """Host-port masks of `words` 64-bit words on a raw snapshot (its own random stream: the snapshot's other fields and every fixture
    drawn from raw_snapshot stay as they are).  A share of the pods keeps to word 0 (the engine's fast path), the others name triples
    anywhere; a pod conflicts with what it wants plus, sometimes, a few neighbours (a wildcard IP)."""
    rng = np.random.RandomState(seed)
    N, T = int(s.n_nodes), int(s.n_tasks)
    bits = 64 * words
    want = np.zeros((T, words), np.uint64); conf = np.zeros((T, words), np.uint64); nodes = np.zeros((N, words), np.uint64)

    def setbit(a, i, b):
        a[i, b // 64] |= np.uint64(1) << np.uint64(b % 64)
    pool = sorted(set(int(x) for x in rng.randint(0, bits, size=max(4, T // 6))) | {0, 1, 63, 64, bits - 1})
    for t in range(T):
        if rng.uniform() >= p_task:
            continue
        for _ in range(int(rng.randint(1, 3))):
            b = int(rng.randint(0, 64)) if rng.uniform() < low_share else int(pool[rng.randint(0, len(pool))])
            setbit(want, t, b); setbit(conf, t, b)
            if rng.uniform() < 0.3:
                for nb in (b ^ 1, (b + 64) % bits):
                    setbit(conf, t, nb)
    for n in range(N):
        if rng.uniform() < p_node:
            for _ in range(int(rng.randint(1, 4))):
                setbit(nodes, n, int(pool[rng.randint(0, len(pool))]))
    for t in range(T):
        if s.task_node[t] != abi.KB_NONE:
            nodes[s.task_node[t]] |= want[t]
    s.node_ports, s.task_port_want, s.task_port_conflict = nodes, want, conf
    s.port_words = words
    s._check()
    return s
