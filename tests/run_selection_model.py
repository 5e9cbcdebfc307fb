"""Executable model (CPU, test infrastructure) for DESIGN section 9.2: a RUN of identical window rows committed by a PARALLEL SELECTION instead of the
serial loop `arg-max -> place -> re-score that node -> arg-max again`.

The serial loop (what allocate.go:143-183 does for r consecutive tasks of one shape, and what the run commit kernel's rows phase does today):
    for each of the r rows: feasible nodes = InitResreq fits Idle or fits Releasing (allocate.go:113-131); best = highest nodeorder score, lowest node
    index among equals (SelectBestNode over the sorted list); fits Idle -> Allocate (Idle -= Resreq) else Pipeline (Releasing -= Resreq); the node's
    non-zero request sums grow either way (its next score changes); no feasible node ends the run.

The selection: per node n its own sequence is a function of its own state only — key(n, j) = its score for the shape after j placements of the shape on
it, defined while placement j + 1 still fits.  Let eff(n, j) = min(key(n, 0..j)) (prefix minimum: non-increasing by construction).  Claim: the loop's
picks are the first r elements of all (n, j) ordered by (eff desc, n asc, j asc).  Why: when n is picked at key s it was the maximum, lowest index among
equals; while its following keys stay >= s it is picked again at once (strictly greater than everything else, or equal and still the lowest index among
the equals — nobody else changed), and for exactly those steps eff stays s, which the order places right behind (n, j); once its key falls below s
the prefix minimum IS the real key again.  Feasibility only shrinks inside a run (Idle and Releasing only shrink), so a sequence ends where it first
fails to fit.  The distinct nodes among the picks are a subset of the r best initial keys: r lanes, each walking one node's sequence, feed a top-r
selection — no serial dependency between rows.

`serial_run` and `selected_run` below return the same list of (node, kind) for any inputs (tests/test_run_selection_model_cpu.py: random clusters, the
three nodeorder weight sets of the bench configurations, Balanced's non-monotone scores included)."""
import numpy as np

ALLOCATE, PIPELINE = 0, 1


def score(rc, rm, ac, am, wl, wm, wb):
    """LeastRequested / MostRequested / BalancedResourceAllocation for one (request sums, allocatable) pair, weights applied (int64 / float64 as in
    vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities)"""
    ok_c, ok_m = ac > 0 and rc <= ac, am > 0 and rm <= am
    lc = (ac - rc) * 10 // ac if ok_c else 0
    lm = (am - rm) * 10 // am if ok_m else 0
    mc = rc * 10 // ac if ok_c else 0
    mm = rm * 10 // am if ok_m else 0
    cf = 1.0 if ac == 0 else float(rc) / float(ac)
    mf = 1.0 if am == 0 else float(rm) / float(am)
    bal = 0 if (cf >= 1 or mf >= 1) else int((1 - abs(cf - mf)) * 10.0)
    return wl * ((lc + lm) // 2) + wm * ((mc + mm) // 2) + wb * bal


class Cluster:
    def __init__(self, ac, am, nzc, nzm, idle, rel, podcnt, maxpods):
        self.ac, self.am, self.nzc, self.nzm = [np.array(x, np.int64) for x in (ac, am, nzc, nzm)]
        self.idle, self.rel = np.array(idle, np.float64), np.array(rel, np.float64)        # [N][2] cpu, memory
        self.podcnt, self.maxpods = np.array(podcnt, np.int64), np.array(maxpods, np.int64)

    def copy(self):
        return Cluster(self.ac, self.am, self.nzc, self.nzm, self.idle, self.rel, self.podcnt, self.maxpods)


def _fits(v, req):
    return v[0] >= req[0] and v[1] >= req[1]


def serial_run(c, shape, r, w):
    """shape = (nz cpu, nz mem, request cpu, request mem); r rows; w = (least, most, balanced) weights"""
    c = c.copy()
    tc, tm, req = shape[0], shape[1], (float(shape[2]), float(shape[3]))
    out = []
    for _ in range(r):
        best, best_key = -1, None
        for n in range(len(c.ac)):
            if c.podcnt[n] >= c.maxpods[n] or not (_fits(c.idle[n], req) or _fits(c.rel[n], req)):
                continue
            k = score(int(c.nzc[n]) + tc, int(c.nzm[n]) + tm, int(c.ac[n]), int(c.am[n]), *w)
            if best_key is None or k > best_key:                     # first maximum: lowest index among equals
                best, best_key = n, k
        if best < 0:
            break
        if _fits(c.idle[best], req):
            c.idle[best] -= req
            out.append((best, ALLOCATE))
        else:
            c.rel[best] -= req
            out.append((best, PIPELINE))
        c.nzc[best] += tc
        c.nzm[best] += tm
        c.podcnt[best] += 1
    return out


def node_sequence(c, n, shape, depth, w):
    """[(real key, kind)] of node n for placements 1..depth of the shape, while they fit — a function of the node's own state"""
    tc, tm, req = shape[0], shape[1], np.array([float(shape[2]), float(shape[3])])
    idle, rel, nzc, nzm, pods = c.idle[n].copy(), c.rel[n].copy(), int(c.nzc[n]), int(c.nzm[n]), int(c.podcnt[n])
    seq = []
    for _ in range(depth):
        if pods >= c.maxpods[n] or not (_fits(idle, req) or _fits(rel, req)):
            break
        k = score(nzc + tc, nzm + tm, int(c.ac[n]), int(c.am[n]), *w)
        if _fits(idle, req):
            idle -= req
            seq.append((k, ALLOCATE))
        else:
            rel -= req
            seq.append((k, PIPELINE))
        nzc += tc
        nzm += tm
        pods += 1
    return seq


def selected_run(c, shape, r, w, lanes=None):
    """the same picks without a serial dependency between rows: every candidate node's sequence (independent: one lane each), prefix minima, ONE
    selection of the r first elements by (eff desc, node asc, j asc).  lanes: only the `lanes` best initial keys are walked (r suffices)."""
    N = len(c.ac)
    first = []
    for n in range(N):
        s = node_sequence(c, n, shape, 1, w)
        if s:
            first.append((-s[0][0], n))
    first.sort()
    cand = [n for _, n in first[: (r if lanes is None else lanes)]]
    entries = []
    for n in cand:
        eff = None
        for j, (k, kind) in enumerate(node_sequence(c, n, shape, r, w)):
            eff = k if eff is None else min(eff, k)
            entries.append((-eff, n, j, kind))
    entries.sort()
    return [(n, kind) for _, n, _, kind in entries[:r]]



def table_run(c, shape, r, w, cap=32, lanes=64):
    """Round 6: what the selection kernel's SHOTS do (kb_commit_sel.hip) — the selection above with bounded tables, committing per shot exactly the
    picks that are final.  One shot: the nC best contenders by current key (nC <= cap), F = the best key of everybody else, and per contender its next
    D = 2^floor(log2(lanes / nC)) entries (key after 0 .. D - 1 further placements: one parallel evaluation), prefix minima per contender, ONE rank
    of all entries by (eff desc, step asc).  In rank order an entry is FINAL — the serial loop's next pick — as long as (a) its eff is above F (no
    outsider comes first) and (b) no entry in front of it is the LAST entry of a table whose sequence may go on (what lies behind a table's end is
    unknown, but its eff is at most the last entry's: it ranks behind that entry, nowhere else).  The final picks are committed (at least one per
    shot: the best contender's first entry), the node states move on, the next shot sees the rest of the run.  A Pipeline ends the round behind its
    row."""
    c = c.copy()
    tc, tm, req = shape[0], shape[1], (float(shape[2]), float(shape[3]))
    N = len(c.ac)
    out = []
    while len(out) < r:
        rem = r - len(out)
        pool = []
        for n in range(N):
            s0 = node_sequence(c, n, shape, 1, w)
            if s0:
                pool.append(((s0[0][0], -n), n))
        pool.sort(reverse=True)
        if not pool:
            break
        nC = min(len(pool), cap, max(rem, 1) + 3)      # (the kernel: everybody at or above the rem-th clean candidate's key; any superset of the top rem works)
        chosen = [n for _, n in pool[:nC]]
        F = pool[nC][0] if len(pool) > nC else None
        D = 1
        while D * 2 * nC <= lanes:
            D *= 2
        entries = []
        for n in chosen:
            seq = node_sequence(c, n, shape, D, w)
            eff = None
            for u, (k, kind) in enumerate(seq):
                kk = (k, -n)
                eff = kk if eff is None else min(eff, kk)
                open_end = (u == D - 1 and kind == ALLOCATE)       # the table ends here, the sequence may not
                entries.append((eff, -u, n, kind, open_end))
                if kind == PIPELINE:
                    break
        entries.sort(key=lambda e: (e[0], e[1]), reverse=True)
        take = []
        for eff, mu, n, kind, open_end in entries:
            if len(take) == rem or (F is not None and eff < F):
                break
            take.append((n, kind))
            if open_end or kind == PIPELINE:
                break
        assert take, "a shot always commits the best contender's first entry"
        stop = False
        for n, kind in take:
            if kind == ALLOCATE:
                c.idle[n] -= req
            else:
                c.rel[n] -= req
                stop = True
            c.nzc[n] += tc
            c.nzm[n] += tm
            c.podcnt[n] += 1
            out.append((n, kind))
        if stop:
            break
    return out
