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
