"""round_model — an executable model of the engine's round protocol on top of tests/pyref.py and the
engine's own host order machine (tests/host_harness): speculate a window, build one sorted candidate list per distinct task
shape from the ROUND-START node state, then commit the window row by row taking the best of (first list entry whose node is
still clean, exact re-evaluation of the nodes this round already changed), and stop where the speculation breaks.  It models
the ALGORITHM (what the kernels are supposed to compute), not the kernels: the `-m gpu` suite checks those.  Test infrastructure.

The model also implements dist.py's backend interface (begin / candidates / commit / apply / decisions), so the world-size-2
gloo test can run the sharded protocol end to end on CPU and compare its decisions with the sequential reference loop."""
import numpy as np
import torch

import pyref
import test_host_order_cpu as hoc

DONE, NO_FEASIBLE, PIPELINED, RENORM = hoc.DONE, hoc.NO_FEASIBLE, hoc.PIPELINED, hoc.RENORM
NODE_BITS = 32


def key_of(score, node):
    """descending score, then ascending node index; 0 terminates a list"""
    return ((int(score) + 1) << NODE_BITS) | (0xFFFFFFFF - node)


def node_of(key):
    return 0xFFFFFFFF - (key & 0xFFFFFFFF)


class RoundModel:
    def __init__(self, L, cfg, snap, tiers, window):
        self.p = pyref.Session(tiers, snap)
        self.m = hoc.Machine(L, cfg, snap, self.p)
        self.Lh = L
        self.snap, self.W = snap, window
        self.shape, self.eff = hoc._feas_shapes(snap, self.p)
        self.dead = [False] * len(self.eff)
        p = self.p
        # identical matrix rows: InitResreq (+ key set), non-zero requests, class, port masks (kb_session_load's row shapes)
        self.row_shape = [(tuple(p.init[t].get(d) for d in range(snap.n_res)), tuple(sorted((p.init[t].scalars or {}).keys())),
                           p.tnzc[t], p.tnzm[t], p.tcls[t], p.tconf[t], p.twant[t]) for t in range(snap.n_tasks)]
        self.aff_row = [p.affinity is not None and any(int(x) != 0 for x in p.affinity[p.tcls[t]]) for t in range(snap.n_tasks)]
        self.decs, self.popped = [], 0
        self.rows, self.spec_pops = [], 0
        self.started = False
        self.delta_len = snap.n_nodes * (2 * snap.n_res + 3)

    # ---- host side: ActionRun::plan / absorb
    def _mark_dead(self, x):
        for y in range(len(self.eff)):
            if not self.dead[y] and self.eff[y][1] == self.eff[x][1] and self.eff[y][2] == self.eff[x][2] and \
                    all(a >= b for a, b in zip(self.eff[y][0], self.eff[x][0])):
                self.dead[y] = True
        self.dead[x] = True

    def plan(self, ahead=False):
        (self.Lh.hh_push_checkpoint if ahead else self.Lh.hh_checkpoint)(self.m.h)
        rows, pops = [], 0
        while len(rows) < self.W:
            t = self.m.next()
            if t is None:
                break
            pops += 1
            if self.dead[self.shape[t]]:
                self.m.report("none")
                continue
            rows.append(t)
            self.m.report("alloc")
        if ahead:
            return rows, pops
        self._set_window(rows, pops)
        return len(rows)

    def _set_window(self, rows, pops, list_len=None):
        self.rows, self.spec_pops = rows, pops
        if not rows:
            self.popped += pops
        # distinct shapes of the window in order of first appearance = the round's matrix rows
        self.mrow_of, self.mrows = [], []
        seen = {}
        for t in rows:
            k = self.row_shape[t]
            if k not in seen:
                seen[k] = len(self.mrows)
                self.mrows.append(t)
            self.mrow_of.append(seen[k])
        self.L = list_len or len(rows) + 1

    def absorb(self, n_done, reason, out):
        rows = self.rows
        if reason == DONE:
            self.popped += self.spec_pops
            self.decs += out
            return
        self.Lh.hh_rollback(self.m.h)
        i = 0
        while True:
            t = self.m.next()
            assert t is not None, "order replay ran out of tasks"
            self.popped += 1
            if self.dead[self.shape[t]]:
                self.m.report("none")
                continue
            assert t == rows[i], "order replay diverged from the speculated sequence"
            if reason == NO_FEASIBLE and i == n_done:
                self._mark_dead(self.shape[t])
                self.m.report("none")
                break
            if reason == RENORM and i == n_done:
                self.Lh.hh_rollback_last_pop(self.m.h)
                self.popped -= 1
                break
            self.decs.append(out[i])
            self.m.report("pipe" if out[i][2] else "alloc")
            i += 1
            if reason == PIPELINED and i == n_done:
                break

    # ---- device side
    def candidate_list(self, m):
        """K1 + K3 for matrix row m: every feasible node of the round-start state, best first, cut to L entries"""
        p, t = self.p, self.mrows[m]
        feasible = [n for n in range(p.N) if (p.init[t].less_equal(p.idle[n]) or p.init[t].less_equal(p.rel[n])) and p.plugin_predicate(t, n)]
        scores = p.prioritize(t, feasible)
        keys = sorted((key_of(scores[n], n) for n in feasible), reverse=True)[: self.L]
        return keys + [0] * (self.L - len(keys))

    def commit_window(self, table, own=None, pre_dirty=()):
        """K5: the sequential commit over the window against the gathered candidate table.  Returns (n_done, reason, decisions,
        own-row deltas as {node: [dIdle R, dRel R, dnzc, dnzm, dcnt]})."""
        p, R = self.p, self.snap.n_res
        dirty, cursor, out, delta = list(pre_dirty), {}, [], {}
        self._total = {}                      # what the whole window did to the nodes (every row, not only the own ones)
        for i, t in enumerate(self.rows):
            if self.aff_row[t] and i > 0:
                return i, RENORM, out, delta
            m = self.mrow_of[i]
            lst = table[m]
            c = cursor.get(m, 0)
            while c < len(lst) and lst[c] != 0 and node_of(lst[c]) in dirty:
                c += 1
            cursor[m] = c
            best = lst[c] if c < len(lst) else 0
            assert not (c >= len(lst) and len(lst) == self.L and lst[-1] != 0), "candidate list exhausted: L = W + 1 must prevent this"
            live = [n for n in dirty if (p.init[t].less_equal(p.idle[n]) or p.init[t].less_equal(p.rel[n])) and p.plugin_predicate(t, n)]
            if live:
                sc = p.prioritize(t, live)
                best = max(best, max(key_of(sc[n], n) for n in live))
            if best == 0:
                return i, NO_FEASIBLE, out, delta
            n = node_of(best)
            before = ([p.idle[n].get(d) for d in range(R)], [p.rel[n].get(d) for d in range(R)], p.nzc[n], p.nzm[n], p.podcnt[n])
            p.popped += 1
            if p.init[t].less_equal(p.idle[n]):
                assert p.ssn_allocate(t, n)
                kind = 0
            else:
                assert p.init[t].less_equal(p.rel[n]) and p.ssn_pipeline(t, n)
                kind = 1
            out.append((t, n, kind))
            if n not in dirty:
                dirty.append(n)
            for tgt in ([delta] if own is None or own[0] <= i < own[1] else []) + [self._total]:
                d = tgt.setdefault(n, [0.0] * (2 * R + 3))
                for k in range(R):
                    d[k] += p.idle[n].get(k) - before[0][k]
                    d[R + k] += p.rel[n].get(k) - before[1][k]
                d[2 * R] += p.nzc[n] - before[2]
                d[2 * R + 1] += p.nzm[n] - before[3]
                d[2 * R + 2] += p.podcnt[n] - before[4]
            if kind == 1:
                return i + 1, PIPELINED, out, delta
        return len(self.rows), DONE, out, delta

    def probe_dead_shapes(self):
        """ActionRun::probe_dead_shapes: between absorb() and plan(), every feasibility shape that is still alive
        is checked against the CURRENT node state; a shape without a feasible node is marked dead now.  Returns how many died."""
        p = self.p
        rep = {}
        for t in range(self.snap.n_tasks):
            rep.setdefault(self.shape[t], t)
        died = 0
        for f, t in rep.items():
            if self.dead[f]:
                continue
            if not any((p.init[t].less_equal(p.idle[n]) or p.init[t].less_equal(p.rel[n])) and p.plugin_predicate(t, n) for n in range(p.N)):
                self.dead[f] = True
                died += 1
        return died

    def run_single(self, probe=False):
        self.probe_deaths = self.breaks = 0
        if probe:
            self.probe_deaths += self.probe_dead_shapes()
        while self.plan():
            table = [self.candidate_list(m) for m in range(len(self.mrows))]
            n_done, reason, out, _ = self.commit_window(table)
            self.absorb(n_done, reason, out)
            if reason != DONE:
                self.breaks += 1
                if probe and reason != RENORM:
                    self.probe_deaths += self.probe_dead_shapes()
        return self

    def run_single_stale(self):
        """the lists of window k+1 are built while window k commits, i.e. from the state at the START of round k,
        for the window speculated behind the second checkpoint; its commit starts with the nodes round k changed already dirty
        (lists of length 2W + 1).  A break re-plans with fresh lists; a next window holding an affinity row (its score is
        normalised over the feasible set, which the stale state may overstate) gets fresh lists too."""
        n = self.plan()
        table = [self.candidate_list(m) for m in range(len(self.mrows))]
        pre_dirty = []
        self.stale_rounds = 0
        while n:
            rows_next, pops_next = self.plan(ahead=True)
            cur = (self.rows, self.spec_pops, self.mrow_of, self.mrows, self.L)
            stale_ok = bool(rows_next) and not any(self.aff_row[t] for t in rows_next)
            if stale_ok:                                        # K1 + K3 of the NEXT window, before this window's commit
                popped = self.popped
                self._set_window(rows_next, pops_next, list_len=2 * self.W + 1)
                self.popped = popped
                next_shapes = (self.mrow_of, self.mrows, self.L)
                table_next = [self.candidate_list(m) for m in range(len(self.mrows))]
            self.rows, self.spec_pops, self.mrow_of, self.mrows, self.L = cur
            n_done, reason, out, _ = self.commit_window(table, pre_dirty=pre_dirty)
            self.absorb(n_done, reason, out)
            if reason != DONE:                                  # absorb() rolled both speculations back
                n = self.plan()
                table, pre_dirty = [self.candidate_list(m) for m in range(len(self.mrows))], []
                continue
            self.Lh.hh_pop_commit(self.m.h)                     # ActionRun::promote
            if stale_ok:
                self.rows, self.spec_pops = rows_next, pops_next
                self.mrow_of, self.mrows, self.L = next_shapes
                table, pre_dirty = table_next, sorted({nd for _, nd, _ in out})   # what round k changed after the lists were built
                self.stale_rounds += 1
            else:
                self._set_window(rows_next, pops_next)
                table, pre_dirty = [self.candidate_list(m) for m in range(len(self.mrows))], []
            n = len(rows_next)
        return self

    # ---- dist.py backend interface (kb_round_* of the C ABI)
    def begin(self, action):
        n = self.plan()
        return (n, len(self.mrows), self.L) if n else (0, 0, 0)

    def candidates(self, m0, m1, keys):
        for m in range(m0, min(m1, len(self.mrows))):
            keys[m - m0] = torch.tensor(self.candidate_list(m), dtype=torch.int64)

    def commit(self, table, r0, r1, delta):
        lists = [[int(x) for x in table[m].tolist()] for m in range(len(self.mrows))]
        self._last = self.commit_window(lists, own=(r0, r1))
        N, R = self.snap.n_nodes, self.snap.n_res
        delta.zero_()
        for n, d in self._last[3].items():
            for k in range(2 * R + 3):
                delta[k * N + n] += d[k]

    def apply(self, delta):
        """the all-reduced deltas must describe exactly what this replica's own commit did to its nodes this round"""
        N, R = self.snap.n_nodes, self.snap.n_res
        n_done, reason, out, _ = self._last
        want = np.zeros((2 * R + 3, N))
        for n, d in self._total.items():
            want[:, n] = d
        assert np.array_equal(delta.numpy().reshape(2 * R + 3, N), want), "replicas diverged: reduced deltas differ from the local commit"
        self.absorb(n_done, reason, out)

    def decisions(self):
        return np.array(self.decs, dtype=np.int64).reshape(-1, 3)

    def close(self):
        self.m.close()
