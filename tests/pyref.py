"""pyref — a SECOND restatement of the reference's allocate + backfill path, pure Python, written from the Go sources (not from
oracle/kb_oracle.c).  Test infrastructure only (small clusters; it is slow on purpose: plain dicts and loops that read like the
Go code).  tests/test_oracle_independent.py runs it against the C oracle on random snapshots: two independent restatements
that agree decision for decision make a transcription slip in either unlikely.  It is still not a reference run.

Canonical orders wherever the Go code ranges over a map: ascending key; SelectBestNode takes the first maximum.
"""
import math
from typing import Dict, List, Optional

import numpy as np

MIN_CPU, MIN_SCALAR, MIN_MEM = 10.0, 10.0, 10.0 * 1024 * 1024          # api/resource_info.go:68-70
PENDING, ALLOCATED, PIPELINED, BINDING, BOUND, RUNNING, RELEASING, SUCCEEDED, FAILED, UNKNOWN = range(10)
NONE = 0xFFFFFFFF
EN_JOB_ORDER, EN_JOB_READY, EN_JOB_PIPELINED, EN_TASK_ORDER, EN_PREEMPTABLE, EN_RECLAIMABLE, EN_QUEUE_ORDER, EN_PREDICATE, EN_NODE_ORDER = (1 << i for i in range(9))


class Fatal(ArithmeticError):
    """glog.Fatalf in the reference: the process dies; reported like the Resource.Sub panic"""


class Resource:
    """api.Resource (api/resource_info.go:28-38): scalars is None for a nil map."""

    def __init__(self, cpu=0.0, mem=0.0, scalars: Optional[Dict[int, float]] = None):
        self.cpu, self.mem, self.scalars = float(cpu), float(mem), scalars

    def clone(self):
        return Resource(self.cpu, self.mem, None if self.scalars is None else dict(self.scalars))

    def is_empty(self):                                               # :93-105
        if not (self.cpu < MIN_CPU and self.mem < MIN_MEM):
            return False
        return all(q < MIN_SCALAR for q in (self.scalars or {}).values())

    def add(self, rr):                                                # :128-140
        self.cpu += rr.cpu
        self.mem += rr.mem
        for name, q in (rr.scalars or {}).items():
            if self.scalars is None:
                self.scalars = {}
            self.scalars[name] = self.scalars.get(name, 0.0) + q
        return self

    def sub(self, rr):                                                # :143-160
        if not rr.less_equal(self):
            raise ArithmeticError("Resource is not sufficient to do operation")
        self.cpu -= rr.cpu
        self.mem -= rr.mem
        for name, q in (rr.scalars or {}).items():
            if self.scalars is None:
                return self
            self.scalars[name] = self.scalars.get(name, 0.0) - q
        return self

    def multi(self, ratio):                                           # :217-224
        self.cpu *= ratio
        self.mem *= ratio
        for name in list((self.scalars or {}).keys()):
            self.scalars[name] *= ratio
        return self

    def less(self, rr):                                               # :227-265
        if not self.cpu < rr.cpu or not self.mem < rr.mem:
            return False
        if self.scalars is None:
            if rr.scalars is not None:
                for q in rr.scalars.values():
                    if q <= MIN_SCALAR:
                        return False
            return True
        if rr.scalars is None:
            return False
        return all(q < rr.scalars.get(name, 0.0) for name, q in self.scalars.items())

    def less_equal(self, rr):                                         # :268-302
        def le(l, r, diff):
            return l < r or math.fabs(l - r) < diff
        if not le(self.cpu, rr.cpu, MIN_CPU) or not le(self.mem, rr.mem, MIN_MEM):
            return False
        if self.scalars is None:
            return True
        for name, q in self.scalars.items():
            if q <= MIN_SCALAR:
                continue
            if rr.scalars is None:
                return False
            if not le(q, rr.scalars.get(name, 0.0), MIN_SCALAR):
                return False
        return True

    def diff(self, rr):                                               # :305-337
        inc, dec = Resource(), Resource()
        if self.cpu > rr.cpu:
            inc.cpu += self.cpu - rr.cpu
        else:
            dec.cpu += rr.cpu - self.cpu
        if self.mem > rr.mem:
            inc.mem += self.mem - rr.mem
        else:
            dec.mem += rr.mem - self.mem
        for name, q in (self.scalars or {}).items():
            rq = (rr.scalars or {}).get(name, 0.0)
            tgt = inc if q > rq else dec
            if tgt.scalars is None:
                tgt.scalars = {}
            tgt.scalars[name] = tgt.scalars.get(name, 0.0) + (q - rq if q > rq else rq - q)
        return inc, dec

    def get(self, name):                                              # :349-361 (0 = cpu, 1 = memory)
        if name == 0:
            return self.cpu
        if name == 1:
            return self.mem
        return 0.0 if self.scalars is None else self.scalars.get(name, 0.0)

    def names(self):                                                  # :364-372
        return [0, 1] + sorted((self.scalars or {}).keys())


def helpers_min(l, r):                                                # api/helpers/helpers.go:28-44
    res = Resource(min(l.cpu, r.cpu), min(l.mem, r.mem))
    if l.scalars is None or r.scalars is None:
        return res
    res.scalars = {name: min(q, r.scalars.get(name, 0.0)) for name, q in l.scalars.items()}
    return res


def helpers_share(l, r):                                              # api/helpers/helpers.go:47-60
    if r == 0:
        return 0.0 if l == 0 else 1.0
    return l / r


class GoHeap:
    """container/heap (go1.13) behind util.PriorityQueue (util/priority_queue.go:26-94)."""

    def __init__(self, less):
        self.items, self.less = [], less

    def __len__(self):
        return len(self.items)

    def push(self, x):
        h = self.items
        h.append(x)
        j = len(h) - 1
        while True:
            i = (j - 1) // 2 if j > 0 else 0
            if i == j or not self.less(h[j], h[i]):
                break
            h[i], h[j] = h[j], h[i]
            j = i

    def pop(self):
        h = self.items
        n = len(h) - 1
        h[0], h[n] = h[n], h[0]
        i = 0
        while True:
            j1 = 2 * i + 1
            if j1 >= n or j1 < 0:
                break
            j = j1
            if j1 + 1 < n and self.less(h[j1 + 1], h[j1]):
                j = j1 + 1
            if not self.less(h[j], h[i]):
                break
            h[i], h[j] = h[j], h[i]
            i = j
        return h.pop()


def go_div(a, b):
    return a // b        # operands are non-negative here: Go's truncation equals floor


class Session:
    def __init__(self, tiers, snap):
        """tiers: [[(plugin name, enabled bits, {arg: int})]]; snap: kube-batch_amd SessionSnapshot."""
        s = snap
        self.tiers = tiers
        self.R, self.N, self.T, self.J, self.Q = int(s.n_res), int(s.n_nodes), int(s.n_tasks), int(s.n_jobs), int(s.n_queues)

        def res(mat, mask, i):
            sc = None
            m = int(mask[i]) if mask is not None else 0
            if m:
                sc = {d: float(mat[d, i]) for d in range(2, self.R) if (m >> (d - 2)) & 1}
            return Resource(mat[0, i], mat[1, i], sc)

        # api.NodeInfo: Idle has Allocatable's scalar key set (NewResource of the same list, node_info.go:49-84); Releasing
        # starts as EmptyResource() and gains keys only through Add (node_info.go:65,154): the keys with a non-zero value
        self.idle = [res(s.node_idle, s.node_scalar_mask, n) for n in range(self.N)]
        self.rel = []
        for n in range(self.N):
            sc = {d: float(s.node_releasing[d, n]) for d in range(2, self.R) if s.node_releasing[d, n] != 0.0}
            self.rel.append(Resource(s.node_releasing[0, n], s.node_releasing[1, n], sc or None))
        self.alloc = [res(s.node_allocatable, s.node_scalar_mask, n) for n in range(self.N)]
        self.acpu, self.amem = [int(x) for x in s.node_alloc_cpu], [int(x) for x in s.node_alloc_mem]
        self.nzc, self.nzm = [int(x) for x in s.node_nz_cpu], [int(x) for x in s.node_nz_mem]
        self.maxpods, self.podcnt = [int(x) for x in s.node_max_pods], [int(x) for x in s.node_pod_cnt]
        self.ncls = [int(x) for x in s.node_class]
        def masks(a, n):   # [n] or [n][Wh] 64-bit words -> one Python int per row (the reference keeps sets: no width)
            if a is None:
                return [0] * n
            a = np.asarray(a, dtype=np.uint64).reshape(n, -1)
            return [sum(int(a[i, w]) << (64 * w) for w in range(a.shape[1])) for i in range(n)]
        self.nports = masks(getattr(s, "node_ports", None), self.N)
        # TaskInfo: InitResreq carries the keys of Resreq plus those an init container raised (pod_info.go:53-62)
        self.resreq = [res(s.task_resreq, s.task_scalar_mask, t) for t in range(self.T)]
        self.init = []
        for t in range(self.T):
            m = int(s.task_scalar_mask[t])
            for d in range(2, self.R):
                if s.task_init_resreq[d, t] != 0.0:
                    m |= 1 << (d - 2)
            self.init.append(Resource(s.task_init_resreq[0, t], s.task_init_resreq[1, t],
                                      {d: float(s.task_init_resreq[d, t]) for d in range(2, self.R) if (m >> (d - 2)) & 1} if m else None))
        self.tnzc, self.tnzm = [int(x) for x in s.task_nz_cpu], [int(x) for x in s.task_nz_mem]
        self.tjob, self.tcls = [int(x) for x in s.task_job], [int(x) for x in s.task_class]
        self.tprio, self.tcre = [int(x) for x in s.task_priority], [int(x) for x in s.task_creation]
        self.status = [int(x) for x in s.task_status]
        self.tnode = [int(x) for x in s.task_node]
        want, conf = getattr(s, "task_port_want", None), getattr(s, "task_port_conflict", None)
        self.twant, self.tconf = masks(want, self.T), masks(conf, self.T)
        self.jbegin = [int(x) for x in s.job_task_begin]
        self.jqueue, self.jmin = [int(x) for x in s.job_queue], [int(x) for x in s.job_min_available]
        self.jprio, self.jcre = [int(x) for x in s.job_priority], [int(x) for x in s.job_creation]
        self.qweight, self.qcre = [int(x) for x in s.queue_weight], [int(x) for x in s.queue_creation]
        self.ntc, self.nnc = int(s.n_task_classes), int(s.n_node_classes)
        self.compat = s.class_compat
        self.affinity = getattr(s, "class_affinity", None)
        prot = getattr(s, "task_evict_protected", None)
        self.protected = [bool(x) for x in prot] if prot is not None else [False] * self.T
        self.onnode = [n != NONE for n in self.tnode]
        self.nstatus = [st if on else None for st, on in zip(self.status, self.onnode)]
        self.base_ports = list(self.nports)
        self.session_on = [set() for _ in range(self.N)]
        self.decisions, self.binds, self.popped, self.evictions, self.pop_order = [], {}, 0, [], []
        # inter-pod (anti)affinity tables (include/kb_engine.h: kb_interpod); the per-domain counts are recomputed from the task
        # statuses on every call here (the PodLister of plugins/util/util.go:37-90), not kept incrementally
        ip = getattr(s, "interpod", None)
        self.ip = None
        if ip is not None:
            self.ip = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in ip.items()}
            for k in ("task_inc", "task_forbid", "task_cls_inc"):      # [T][W] 64-bit words -> one Python integer per task
                self.ip[k] = [sum(int(w) << (64 * i) for i, w in enumerate(row)) for row in self.ip[k]]
            self.ip_at_open = [on for on in self.onnode]
            self.ip_added = [set() for _ in range(self.N)]      # tasks whose first AddTask happened in this session: Spec.NodeName == ""
        self._open_plugins()

    # ---- inter-pod (anti)affinity on the kb_interpod tables
    def _ip_count(self, c, dom):
        """allocated-status session pods counted by predicate counter c in domain dom (None: in any domain or none)"""
        ip, k = self.ip, 0
        for t in range(self.T):
            if self.status[t] in (ALLOCATED, BINDING, BOUND, RUNNING) and (ip["task_inc"][t] >> c) & 1 and self.tnode[t] != NONE:
                if dom is None or ip["ctr_dom"][c][self.tnode[t]] == dom:
                    k += 1
        return k

    def interpod_predicate(self, t, n):
        """InterPodAffinityMatches (vendor/.../algorithm/predicates/predicates.go:1261-1290) as kb_interpod states it"""
        ip = self.ip
        if ip is None:
            return True
        for c in range(ip["n_counters"]):
            if (ip["task_forbid"][t] >> c) & 1:
                d = ip["ctr_dom"][c][n]
                if d != NONE and self._ip_count(c, d) > 0:
                    return False
        r = ip["task_require"][t]
        if r != 0xFFFF:
            d = ip["ctr_dom"][r][n]
            if not (d != NONE and self._ip_count(r, d) > 0):
                if self._ip_count(r, None) > 0 or not ip["task_self"][t]:
                    return False
        return True

    def interpod_scores(self, t, feasible):
        """CalculateInterPodAffinityPriority (vendor/.../priorities/interpod_affinity.go:99-235) as kb_interpod states it"""
        ip = self.ip
        if ip is None or ip["task_sig"][t] == NONE:
            return {n: 0 for n in feasible}
        w = ip["sig_weight"][ip["task_sig"][t]]
        z = ip["first_unbound_node"]
        for n in range(self.N):
            if self.ip_added[n]:
                z = min(z, n)
                break
        counts = {n: 0 for n in feasible}
        for p in range(ip["n_classes"]):
            if w[p] == 0:
                continue
            dom = ip["cls_dom"][p]
            bound, zs = {}, 0
            for n in feasible:
                unb = ip["cls_unbound"][p][n] + sum(1 for i in self.ip_added[n] if (ip["task_cls_inc"][i] >> p) & 1)
                zs += unb
                if dom[n] != NONE:
                    bound[dom[n]] = bound.get(dom[n], 0) + ip["cls_bound"][p][n]
            zdom = dom[z] if z != NONE else NONE
            for i in feasible:
                if dom[i] == NONE:
                    continue
                counts[i] += w[p] * (bound.get(dom[i], 0) + (zs if dom[i] == zdom else 0))
        mx, mn = max([0] + list(counts.values())), min([0] + list(counts.values()))
        out = {}
        for n in feasible:
            f = 0.0
            if mx - mn > 0:
                f = 10.0 * (float(counts[n] - mn) / float(mx - mn))
            out[n] = int(f)
        return out

    # ---- conf helpers (framework/session_plugins.go isEnabled)
    def _opts(self):
        for tier in self.tiers:
            for opt in tier:
                yield opt

    def _has(self, name):
        return any(o[0] == name for o in self._opts())

    def _enabled(self, name, bit):
        return any(o[0] == name and (o[1] & bit) for o in self._opts())

    # ---- JobInfo counters (api/job_info.go:383-434)
    def _tasks(self, j):
        return range(self.jbegin[j], self.jbegin[j + 1])

    def ready_num(self, j):
        return sum(1 for t in self._tasks(j) if self.status[t] in (BOUND, BINDING, RUNNING, ALLOCATED, SUCCEEDED))

    def job_ready(self, j):                                           # gang.go:122-125 behind session_plugins.go:182-200
        if self._enabled("gang", EN_JOB_READY):
            return self.ready_num(j) >= self.jmin[j]
        return True

    # ---- OnSessionOpen of drf and proportion
    def _open_plugins(self):
        total = Resource()
        for n in range(self.N):
            total.add(self.alloc[n])
        self.total = total
        self.jalloc = []
        for j in range(self.J if self._has("drf") else 0):            # drf.go:60-83
            a = Resource()
            for t in self._tasks(j):
                if self.status[t] in (BOUND, BINDING, RUNNING, ALLOCATED):
                    a.add(self.resreq[t])
            self.jalloc.append(a)
        self.jshare = [self._drf_share(a) for a in self.jalloc]
        self.qattr = {}                                               # proportion.go:58-154
        if not self._has("proportion"):
            return
        for j in range(self.J):
            q = self.jqueue[j]
            if q not in self.qattr:
                self.qattr[q] = {"deserved": Resource(), "allocated": Resource(), "request": Resource(), "share": 0.0, "weight": self.qweight[q]}
            a = self.qattr[q]
            for t in self._tasks(j):
                if self.status[t] in (BOUND, BINDING, RUNNING, ALLOCATED):
                    a["allocated"].add(self.resreq[t])
                    a["request"].add(self.resreq[t])
                elif self.status[t] == PENDING:
                    a["request"].add(self.resreq[t])
        remaining = total.clone()
        meet = set()
        while True:
            tw = sum(a["weight"] for q, a in self.qattr.items() if q not in meet)
            if tw == 0:
                break
            inc_all, dec_all = Resource(), Resource()
            for q in sorted(self.qattr):
                a = self.qattr[q]
                if q in meet:
                    continue
                old = a["deserved"].clone()
                a["deserved"].add(remaining.clone().multi(float(a["weight"]) / float(tw)))
                if a["request"].less(a["deserved"]):
                    a["deserved"] = helpers_min(a["deserved"], a["request"])
                    meet.add(q)
                self._prop_share(a)
                inc, dec = a["deserved"].diff(old)
                inc_all.add(inc)
                dec_all.add(dec)
            remaining.sub(inc_all).add(dec_all)
            if remaining.is_empty():
                break

    def _drf_share(self, allocated):                                  # drf.go:157-171
        res = 0.0
        for rn in self.total.names():
            res = max(res, helpers_share(allocated.get(rn), self.total.get(rn)))
        return res

    def _prop_share(self, a):                                         # proportion.go:241-253
        res = 0.0
        for rn in a["deserved"].names():
            res = max(res, helpers_share(a["allocated"].get(rn), a["deserved"].get(rn)))
        a["share"] = res

    # ---- tiered order functions (framework/session_plugins.go:243-331)
    def job_less(self, l, r):
        for name, en, _ in self._opts():
            if not en & EN_JOB_ORDER:
                continue
            j = 0
            if name == "priority":                                    # priority.go:61-77
                j = -1 if self.jprio[l] > self.jprio[r] else (1 if self.jprio[l] < self.jprio[r] else 0)
            elif name == "gang":                                      # gang.go:96-119
                lr, rr = self.ready_num(l) >= self.jmin[l], self.ready_num(r) >= self.jmin[r]
                j = 0 if (lr and rr) else (1 if lr else (-1 if rr else 0))
            elif name == "drf":                                       # drf.go:114-130
                j = 0 if self.jshare[l] == self.jshare[r] else (-1 if self.jshare[l] < self.jshare[r] else 1)
            else:
                continue
            if j != 0:
                return j < 0
        if self.jcre[l] == self.jcre[r]:
            return l < r
        return self.jcre[l] < self.jcre[r]

    def queue_less(self, l, r):
        for name, en, _ in self._opts():
            if name == "proportion" and en & EN_QUEUE_ORDER:          # proportion.go:156-169
                ls, rs = self.qattr[l]["share"], self.qattr[r]["share"]
                if ls != rs:
                    return ls < rs
        if self.qcre[l] == self.qcre[r]:
            return l < r
        return self.qcre[l] < self.qcre[r]

    def task_less(self, l, r):
        for name, en, _ in self._opts():
            if name == "priority" and en & EN_TASK_ORDER:             # priority.go:40-56
                if self.tprio[l] != self.tprio[r]:
                    return self.tprio[l] > self.tprio[r]
        if self.tcre[l] == self.tcre[r]:
            return l < r
        return self.tcre[l] < self.tcre[r]

    def overused(self, q):                                            # proportion.go:198-209 (no Enabled* gate, session_plugins.go:165-179)
        if not self._has("proportion"):
            return False
        a = self.qattr[q]
        return a["deserved"].less_equal(a["allocated"])

    # ---- predicates plugin (plugins/predicates/predicates.go:123-265) with the static checks folded into classes
    def plugin_predicate(self, t, n):
        if not self._enabled("predicates", EN_PREDICATE):
            return True
        if self.maxpods[n] <= self.podcnt[n]:
            return False
        if self.compat is not None:
            bit = self.tcls[t] * self.nnc + self.ncls[n]
            if not (int(self.compat[bit >> 3]) >> (bit & 7)) & 1:
                return False
        if (self.nports[n] & self.tconf[t]) != 0:
            return False
        return self.interpod_predicate(t, n)

    # ---- nodeorder (plugins/nodeorder/nodeorder.go:107-168 over vendor/.../priorities)
    def _weights(self):
        w = {"leastrequested.weight": 1, "mostrequested.weight": 0, "nodeaffinity.weight": 1, "podaffinity.weight": 1, "balancedresource.weight": 1}
        for name, en, args in self._opts():
            if name == "nodeorder":
                w.update(args or {})
        return w

    def prioritize(self, t, feasible):
        """util.PrioritizeNodes (util/scheduler_helper.go:89-171): map, reduce, weighted sum."""
        if not self._enabled("nodeorder", EN_NODE_ORDER):
            return {n: 0.0 for n in feasible}
        w = self._weights()
        counts = {}
        scores = {}
        for n in feasible:
            rc, rm = self.nzc[n] + self.tnzc[t], self.nzm[n] + self.tnzm[t]
            ac, am = self.acpu[n], self.amem[n]
            lc = 0 if (ac == 0 or rc > ac) else go_div((ac - rc) * 10, ac)
            lm = 0 if (am == 0 or rm > am) else go_div((am - rm) * 10, am)
            mc = 0 if (ac == 0 or rc > ac) else go_div(rc * 10, ac)
            mm = 0 if (am == 0 or rm > am) else go_div(rm * 10, am)
            cf = 1.0 if ac == 0 else float(rc) / float(ac)
            mf = 1.0 if am == 0 else float(rm) / float(am)
            bal = 0 if (cf >= 1 or mf >= 1) else int((1 - math.fabs(cf - mf)) * 10.0)
            counts[n] = int(self.affinity[self.tcls[t]][self.ncls[n]]) if self.affinity is not None else 0
            scores[n] = [go_div(lc + lm, 2), go_div(mc + mm, 2), 0, 0, bal]
        mx = max(counts.values()) if counts else 0                    # NormalizeReduce(10, false) (reduce.go:28-63)
        ips = self.interpod_scores(t, feasible)
        for n in feasible:
            scores[n][2] = go_div(10 * counts[n], mx) if mx > 0 else counts[n]
            scores[n][3] = ips[n]
        ws = [w["leastrequested.weight"], w["mostrequested.weight"], w["nodeaffinity.weight"], w["podaffinity.weight"], w["balancedresource.weight"]]
        return {n: float(sum(float(sc * wt) for sc, wt in zip(scores[n], ws))) for n in feasible}

    # ---- Session.Allocate / Pipeline (framework/session.go:194-288)
    def _fire_allocate(self, t):
        j = self.tjob[t]
        if self._has("drf"):
            self.jalloc[j].add(self.resreq[t])
            self.jshare[j] = self._drf_share(self.jalloc[j])
        if self._has("proportion"):
            a = self.qattr[self.jqueue[j]]
            a["allocated"].add(self.resreq[t])
            self._prop_share(a)

    def _fire_deallocate(self, t):                                    # drf.go:143-151, proportion.go:223-233
        j = self.tjob[t]
        if self._has("drf"):
            self.jalloc[j].sub(self.resreq[t])
            self.jshare[j] = self._drf_share(self.jalloc[j])
        if self._has("proportion"):
            a = self.qattr[self.jqueue[j]]
            a["allocated"].sub(self.resreq[t])
            self._prop_share(a)

    # ---- api.NodeInfo.AddTask / RemoveTask / UpdateTask (api/node_info.go:172-256); the node keeps a CLONE of the task, so
    # the status the node sees (nstatus) is the task's status at the time of the last AddTask
    def node_add_task(self, t, n):
        if self.tnode[t] != NONE and self.tnode[t] != n:              # :173-176 "already on different node" (NodeName is sticky)
            return False
        if self.onnode[t]:                                            # :178-182 "already on node"
            return False
        st = self.status[t]
        if st == RELEASING:
            if not self.resreq[t].less_equal(self.idle[n]):
                return False
            self.idle[n].sub(self.resreq[t])
            self.rel[n].add(self.resreq[t])
        elif st == PIPELINED:
            self.rel[n].sub(self.resreq[t])
        else:
            if not self.resreq[t].less_equal(self.idle[n]):           # allocateIdleResource :161-167
                return False
            self.idle[n].sub(self.resreq[t])
        self.tnode[t], self.onnode[t], self.nstatus[t] = n, True, st
        self.podcnt[n] += 1                                           # k8s-side sums range over ni.Tasks (nodeinfo rebuilt per call)
        self.nzc[n] += self.tnzc[t]
        self.nzm[n] += self.tnzm[t]
        self.session_on[n].add(t)
        self.nports[n] |= self.twant[t]
        if self.ip is not None and not self.ip_at_open[t]:
            self.ip_added[n].add(t)
        return True

    def node_remove_task(self, t):
        if not self.onnode[t]:
            return False
        n, st = self.tnode[t], self.nstatus[t]
        if st == RELEASING:
            self.rel[n].sub(self.resreq[t])
            self.idle[n].add(self.resreq[t])
        elif st == PIPELINED:
            self.rel[n].add(self.resreq[t])
        else:
            self.idle[n].add(self.resreq[t])
        self.onnode[t] = False
        self.podcnt[n] -= 1
        self.nzc[n] -= self.tnzc[t]
        self.nzm[n] -= self.tnzm[t]
        self.session_on[n].discard(t)
        if self.ip is not None:
            self.ip_added[n].discard(t)
        ports = self.base_ports[n]
        for i in self.session_on[n]:
            ports |= self.twant[i]
        self.nports[n] = ports
        return True

    def node_update_task(self, t):
        if self.node_remove_task(t):
            if not self.node_add_task(t, self.tnode[t]):
                raise Fatal("glog.Fatalf: Failed to add Task during task update (node_info.go:250-254)")

    def _dispatch_ready(self, j):                                     # session.go:277-285
        if self.job_ready(j):
            for i in self._tasks(j):
                if self.status[i] == ALLOCATED:
                    self.binds[i] = self.tnode[i]
                    self.status[i] = BINDING

    def ssn_allocate(self, t, n):
        self.status[t] = ALLOCATED                                    # session.go:243, before node.AddTask
        if not self.node_add_task(t, n):
            return False
        self.decisions.append((t, n, 0))
        self._fire_allocate(t)
        self._dispatch_ready(self.tjob[t])
        return True

    def ssn_pipeline(self, t, n):                                     # session.go:194-232
        self.status[t] = PIPELINED
        if not self.node_add_task(t, n):
            return False
        self.decisions.append((t, n, 1))
        self._fire_allocate(t)
        return True

    def ssn_evict(self, t):                                           # session.go:317-354 (reclaim: evicts at once)
        self.evictions.append(t)
        self.status[t] = RELEASING
        self.node_update_task(t)
        self._fire_deallocate(t)

    def waiting_num(self, j):
        return sum(1 for t in self._tasks(j) if self.status[t] == PIPELINED)

    def job_pipelined(self, j):                                       # gang.go:126-129 behind session_plugins.go:203-221
        if self._enabled("gang", EN_JOB_PIPELINED):
            return self.waiting_num(j) + self.ready_num(j) >= self.jmin[j]
        return True

    # ---- victims: tier-wise intersection (session_plugins.go:80-162).  A Go nil slice (no append happened) is "no decision".
    def _victims(self, bit, fns, actor, candidates):
        victims, init = None, False
        for tier in self.tiers:
            for name, en, _ in tier:
                if not en & bit or name not in fns:
                    continue
                cand = fns[name](actor, candidates)
                if not init:
                    victims, init = cand, True
                else:
                    inter = [v for v in (victims or []) for c in (cand or []) if v == c]
                    victims = inter or None
            if victims is not None:
                return victims
        return victims

    def _gang_evictable(self, actor, tasks):                          # gang.go:71-90
        out = []
        for t in tasks:
            j = self.tjob[t]
            if self.jmin[j] <= self.ready_num(j) - 1 or self.jmin[j] == 1:
                out.append(t)
        return out or None

    def _conformance_evictable(self, actor, tasks):                   # conformance.go:41-58
        return [t for t in tasks if not self.protected[t]] or None

    def _priority_preemptable(self, actor, tasks):                    # priority.go:79-97
        return [t for t in tasks if self.jprio[self.tjob[t]] < self.jprio[self.tjob[actor]]] or None

    def _drf_preemptable(self, actor, tasks):                         # drf.go:85-110
        ls = self._drf_share(self.jalloc[self.tjob[actor]].clone().add(self.resreq[actor]))
        allocations, out = {}, []
        for t in tasks:
            j = self.tjob[t]
            if j not in allocations:
                allocations[j] = self.jalloc[j].clone()
            rs = self._drf_share(allocations[j].sub(self.resreq[t]))
            if ls < rs or math.fabs(ls - rs) <= 0.000001:
                out.append(t)
        return out or None

    def _proportion_reclaimable(self, actor, tasks):                  # proportion.go:171-196
        allocations, out = {}, []
        for t in tasks:
            q = self.jqueue[self.tjob[t]]
            a = self.qattr[q]
            if q not in allocations:
                allocations[q] = a["allocated"].clone()
            if allocations[q].less(self.resreq[t]):
                continue
            allocations[q].sub(self.resreq[t])
            if a["deserved"].less_equal(allocations[q]):
                out.append(t)
        return out or None

    def preemptable(self, actor, tasks):
        fns = {"gang": self._gang_evictable, "conformance": self._conformance_evictable, "priority": self._priority_preemptable}
        if self._has("drf"):
            fns["drf"] = self._drf_preemptable
        return self._victims(EN_PREEMPTABLE, {k: v for k, v in fns.items() if self._has(k)}, actor, tasks)

    def reclaimable(self, actor, tasks):
        fns = {"gang": self._gang_evictable, "conformance": self._conformance_evictable, "proportion": self._proportion_reclaimable}
        return self._victims(EN_RECLAIMABLE, {k: v for k, v in fns.items() if self._has(k)}, actor, tasks)

    # ---- actions
    def place(self, t):
        """allocate.go:129-183 for one popped task: PredicateNodes, PrioritizeNodes, SelectBestNode, Allocate or Pipeline.
        Returns "none" (no feasible node), "alloc" or "pipe" ("stay" when neither branch fired)."""
        self.popped += 1
        self.pop_order.append(t)
        feasible = [n for n in range(self.N)
                    if (self.init[t].less_equal(self.idle[n]) or self.init[t].less_equal(self.rel[n])) and self.plugin_predicate(t, n)]
        if not feasible:
            return "none"
        scores = self.prioritize(t, feasible)
        best = max(scores.values())
        n = min(k for k, v in scores.items() if v == best)            # canonical first maximum
        if self.init[t].less_equal(self.idle[n]):
            self.ssn_allocate(t, n)
            return "alloc"
        if self.init[t].less_equal(self.rel[n]):
            self.ssn_pipeline(t, n)
            return "pipe"
        return "stay"

    def allocate(self):                                               # actions/allocate/allocate.go:43-194
        queues = GoHeap(self.queue_less)
        jobs_map = {}
        for j in range(self.J):
            q = self.jqueue[j]
            if q >= self.Q:
                continue
            queues.push(q)
            jobs_map.setdefault(q, GoHeap(self.job_less)).push(j)
        pending = {}
        while len(queues):
            q = queues.pop()
            if self.overused(q):
                continue
            jobs = jobs_map.get(q)
            if jobs is None or not len(jobs):
                continue
            j = jobs.pop()
            if j not in pending:
                tasks = GoHeap(self.task_less)
                for t in self._tasks(j):
                    if self.status[t] == PENDING and not self.resreq[t].is_empty():
                        tasks.push(t)
                pending[j] = tasks
            tasks = pending[j]
            while len(tasks):
                t = tasks.pop()
                if self.place(t) == "none":
                    break
                if self.job_ready(j) and len(tasks):
                    jobs.push(j)
                    break
            queues.push(q)

    def backfill(self):                                               # actions/backfill/backfill.go:40-71
        for j in range(self.J):
            for t in self._tasks(j):
                if self.status[t] != PENDING or not self.init[t].is_empty():
                    continue
                self.popped += 1
                for n in range(self.N):
                    if not self.plugin_predicate(t, n):
                        continue
                    if self.ssn_allocate(t, n):
                        break

    # ---- framework.Statement (framework/statement.go:30-240)
    def stmt_evict(self, ops, t):
        self.status[t] = RELEASING
        self.node_update_task(t)
        self._fire_deallocate(t)
        ops.append(("evict", t))

    def stmt_pipeline(self, ops, t, n):
        self.status[t] = PIPELINED
        if self.node_add_task(t, n):                                  # an AddTask error is logged; handlers still fire (:128-147)
            self.decisions.append((t, n, 1))
        self._fire_allocate(t)
        ops.append(("pipeline", t))

    def stmt_discard(self, ops):
        for name, t in reversed(ops):
            if name == "evict":                                       # unevict :80-107
                self.status[t] = RUNNING
                self.node_update_task(t)
                self._fire_allocate(t)
            else:                                                     # unpipeline :152-187
                self.status[t] = PENDING
                if self.node_remove_task(t):
                    self.decisions = [d for d in self.decisions if d[0] != t]
                self._fire_deallocate(t)

    def stmt_commit(self, ops):
        for name, t in ops:
            if name == "evict":
                self.evictions.append(t)                              # cache.Evict

    def _node_tasks(self, n):
        return [t for t in range(self.T) if self.onnode[t] and self.tnode[t] == n]      # canonical: ascending task index

    def _preempt_one(self, ops, preemptor, flt):                      # actions/preempt/preempt.go:171-254
        self.popped += 1
        feasible = [n for n in range(self.N) if self.plugin_predicate(preemptor, n)]
        scores = self.prioritize(preemptor, feasible)
        order = sorted(feasible, key=lambda n: (scores[n], n), reverse=True)        # SortNodes: sort.Reverse(score, then host name)
        for n in order:
            preemptees = [t for t in self._node_tasks(n) if flt(t)]
            victims = self.preemptable(preemptor, preemptees)
            if not victims:                                           # validateVictims :256-270
                continue
            all_res = Resource()
            for v in victims:
                all_res.add(self.resreq[v])
            if not self.init[preemptor].less_equal(all_res):
                continue
            vq = GoHeap(lambda l, r: not self.task_less(l, r))
            for v in victims:
                vq.push(v)
            preempted = Resource()
            while len(vq):
                v = vq.pop()
                self.stmt_evict(ops, v)
                preempted.add(self.resreq[v])
                if self.init[preemptor].less_equal(preempted):
                    break
            if self.init[preemptor].less_equal(preempted):
                self.stmt_pipeline(ops, preemptor, n)
                return True
        return False

    def preempt(self):                                                # actions/preempt/preempt.go:44-166
        preemptors_map, preemptor_tasks, under_request, queues = {}, {}, [], []
        for j in range(self.J):
            q = self.jqueue[j]
            if q >= self.Q:
                continue
            if q not in queues:
                queues.append(q)
            pend = [t for t in self._tasks(j) if self.status[t] == PENDING]
            if pend:
                preemptors_map.setdefault(q, GoHeap(self.job_less)).push(j)
                under_request.append(j)
                preemptor_tasks[j] = GoHeap(self.task_less)
                for t in pend:
                    preemptor_tasks[j].push(t)
        for q in sorted(queues):
            while True:
                preemptors = preemptors_map.get(q)
                if preemptors is None or not len(preemptors):
                    break
                pj = preemptors.pop()
                ops, assigned = [], False
                while True:
                    if not len(preemptor_tasks[pj]):
                        break
                    p = preemptor_tasks[pj].pop()
                    if self._preempt_one(ops, p, lambda t: self.nstatus[t] == RUNNING and self.jqueue[self.tjob[t]] == self.jqueue[pj]
                                         and self.tjob[t] != self.tjob[p]):
                        assigned = True
                    if self.job_pipelined(pj):
                        self.stmt_commit(ops)
                        break
                if not self.job_pipelined(pj):
                    self.stmt_discard(ops)
                    continue
                if assigned:
                    preemptors.push(pj)
            for j in under_request:                                   # preemption between tasks within a job
                while True:
                    if j not in preemptor_tasks or not len(preemptor_tasks[j]):
                        break
                    p = preemptor_tasks[j].pop()
                    ops = []
                    assigned = self._preempt_one(ops, p, lambda t: self.nstatus[t] == RUNNING and self.tjob[t] == self.tjob[p])
                    self.stmt_commit(ops)
                    if not assigned:
                        break

    def reclaim(self):                                                # actions/reclaim/reclaim.go:41-193
        queues, seen = GoHeap(self.queue_less), set()
        preemptors_map, preemptor_tasks = {}, {}
        for j in range(self.J):
            q = self.jqueue[j]
            if q >= self.Q:
                continue
            if q not in seen:
                seen.add(q)
                queues.push(q)
            pend = [t for t in self._tasks(j) if self.status[t] == PENDING]
            if pend:
                preemptors_map.setdefault(q, GoHeap(self.job_less)).push(j)
                preemptor_tasks[j] = GoHeap(self.task_less)
                for t in pend:
                    preemptor_tasks[j].push(t)
        while len(queues):
            q = queues.pop()
            if self.overused(q):
                continue
            jobs = preemptors_map.get(q)
            if jobs is None or not len(jobs):
                continue
            j = jobs.pop()
            tasks = preemptor_tasks.get(j)
            if tasks is None or not len(tasks):
                continue
            task = tasks.pop()
            self.popped += 1
            assigned = False
            for n in range(self.N):
                if not self.plugin_predicate(task, n):
                    continue
                reclaimees = [t for t in self._node_tasks(n) if self.nstatus[t] == RUNNING and self.jqueue[self.tjob[t]] != self.jqueue[j]]
                victims = self.reclaimable(task, reclaimees)
                if not victims:
                    continue
                all_res = Resource()
                for v in victims:
                    all_res.add(self.resreq[v])
                if not self.init[task].less_equal(all_res):
                    continue
                reclaimed = Resource()
                for v in victims:
                    self.ssn_evict(v)
                    reclaimed.add(self.resreq[v])
                    if self.init[task].less_equal(reclaimed):
                        break
                if self.init[task].less_equal(reclaimed):
                    self.ssn_pipeline(task, n)
                    assigned = True
                    break
            if assigned:
                queues.push(q)

    def run(self, actions):
        for a in actions:
            getattr(self, a)()
        return self
