// evict_harness.cpp — test infrastructure: the ENGINE'S OWN host code for kb_engine_create / kb_session_load / kb_run_preempt /
// kb_run_reclaim (kube-batch_amd/csrc/kb_session.cpp and kb_preempt.cpp, compiled unchanged with g++) behind a tiny C interface,
// so that the CPU suite can run the policy compiler, the session build (validation, task shapes, proportion's water-filling) and the
// evict actions' statement / victim bookkeeping against the oracle without a GPU (tests/test_host_evict_cpu.py).  The device's part
// — one sorted node list per preemptor shape (PredicateNodes + PrioritizeNodes + SortNodes) against the node state it was last
// handed — is played by tests/pyref.py through the two callbacks.  What kb_engine.cpp does around the machine (device <-> host copies
// of the node state, the share reduction kernel) is replaced by plain copies here.  Nothing here is linked into libkbengine.so.
#include <cstdio>

#include "../../kube-batch_amd/csrc/kb_preempt.hpp"

using namespace kb;

extern "C" {
// out: up to N keys, (score << 32 | node), best first in SortNodes' order; returns the count
typedef uint32_t (*eh_list_fn)(uint32_t task, uint64_t *out);
// the "device" copy of these nodes is brought up to date: what the plugin predicates and scorers read
typedef void (*eh_refresh_fn)(const uint32_t *nodes, uint32_t n, const int64_t *nzc, const int64_t *nzm, const int32_t *podcnt, const uint64_t *ports);
}

struct EH {
  Policy pol;
  HostSession hs;
  LiveNodes ln;
  std::vector<uint8_t> status, counted;
  std::vector<uint32_t> tnode;
  std::vector<uint32_t> t_active, nmask;
  std::vector<StmtOp> ops;
  std::vector<uint32_t> evictions;
  uint64_t popped = 0, evals = 0;
  eh_list_fn list_fn = nullptr;
  eh_refresh_fn refresh_fn = nullptr;
  std::string err;
  int code = 0;
};

namespace {

template <typename F> int guarded(EH *h, F f) {
  try {
    f();
    h->code = KB_OK;
  } catch (const EngineError &e) {
    h->code = e.code;
    h->err = e.what();
  } catch (const std::exception &e) {
    h->code = KB_E_INTERNAL;
    h->err = e.what();
  }
  return h->code;
}

}  // namespace

extern "C" {

EH *eh_create() { return new EH(); }
void eh_destroy(EH *h) { delete h; }
const char *eh_error(const EH *h) { return h->err.c_str(); }

// kb_engine_create's host half, then kb_session_load's.  The running aggregates come from the caller (the engine takes them from its
// share-reduction kernel): job_alloc [J][R], job_share [J], queue_alloc [Q][R], queue_share [Q].
int eh_load(EH *h, const kb_config *cfg, const kb_snapshot *sn, const double *job_alloc, const double *job_share, const double *queue_alloc,
            const double *queue_share) {
  return guarded(h, [&]() {
    h->pol = compile_policy(cfg);
    if (sn->version != KB_ABI_VERSION) throw EngineError(KB_E_INVALID, "snapshot ABI version mismatch");
    if (sn->n_res < 2 || sn->n_res > KB_MAX_RES) throw EngineError(KB_E_INVALID, "n_res out of range");
    const uint32_t NP = sn->n_nodes ? sn->n_nodes : 1;   // no device padding here
    build_host_session(sn, h->pol, NP, h->hs, h->t_active, h->nmask);
    HostSession &hs = h->hs;
    const int R = hs.R;
    const uint32_t N = hs.N, T = hs.T, J = hs.J, Q = hs.Q;
    // kb_session_load sets this while it uploads the preferred node-affinity table (kb_engine.cpp): any non-zero count under a non-zero weight
    hs.cls_has_aff.clear();
    if (sn->class_affinity && h->pol.wNA != 0) {
      std::vector<uint8_t> has(sn->n_task_classes ? sn->n_task_classes : 1, 0);
      for (size_t i = 0; i < (size_t)sn->n_task_classes * sn->n_node_classes; i++)
        if (sn->class_affinity[i]) { hs.has_affinity = true; has[i / sn->n_node_classes] = 1; }
      if (hs.has_affinity) hs.cls_has_aff = has;
    }
    hs.job_alloc.assign(job_alloc, job_alloc + (size_t)J * R);
    hs.job_share.assign(job_share, job_share + J);
    hs.queue_alloc.assign(queue_alloc, queue_alloc + (size_t)Q * R);
    hs.queue_share.assign(queue_share, queue_share + Q);
    // the live node state as kb_session_load uploads it (kb_engine.cpp: run_evict_action reads it back in this form)
    LiveNodes &ln = h->ln;
    ln = LiveNodes();
    ln.idle.assign(N, Res()); ln.rel.assign(N, Res());
    ln.nzc.assign(sn->node_nz_cpu, sn->node_nz_cpu + N); ln.nzm.assign(sn->node_nz_mem, sn->node_nz_mem + N);
    ln.podcnt.assign(sn->node_pod_cnt, sn->node_pod_cnt + N);
    ln.ports.assign(N, 0);
    const size_t Wh = sn->port_words ? sn->port_words : 1;   // host-port masks: word 0 and, behind it, what HostSession::port_xw kept
    for (uint32_t n = 0; n < N && sn->node_ports; n++) ln.ports[n] = sn->node_ports[(size_t)n * Wh];
    ln.ports_x.assign((size_t)N * h->hs.port_xw, 0);
    for (uint32_t n = 0; n < N && sn->node_ports; n++)
      for (uint32_t w = 0; w < h->hs.port_xw; w++) ln.ports_x[(size_t)n * h->hs.port_xw + w] = sn->node_ports[(size_t)n * Wh + 1 + w];
    for (uint32_t n = 0; n < N; n++) {
      ln.idle[n].mask = h->nmask[n] & 0x3FFFFFFFu;
      for (int d = 0; d < R; d++) {
        ln.idle[n].v[d] = sn->node_idle[(size_t)d * N + n];
        ln.rel[n].v[d] = sn->node_releasing[(size_t)d * N + n];
        if (d >= 2 && ln.rel[n].v[d] != 0.0) ln.rel[n].setk(d);
      }
    }
    ln.ac.assign(hs.n_ac.begin(), hs.n_ac.end()); ln.am.assign(hs.n_am.begin(), hs.n_am.end());
    ln.maxpods = hs.n_maxpods; ln.cls = hs.n_cls;
    h->status = hs.t_status;
    h->tnode = hs.t_node;
    h->counted.assign(T ? T : 1, 0);
    for (uint32_t t = 0; t < T; t++) {
      const int st = hs.t_status[t];
      h->counted[t] = (st == KB_TASK_BOUND || st == KB_TASK_BINDING || st == KB_TASK_RUNNING || st == KB_TASK_ALLOCATED) ? 1 : 0;   // drf.go:71-77
    }
    h->ops.clear(); h->evictions.clear();
    h->popped = h->evals = 0;
  });
}

void eh_set_callbacks(EH *h, eh_list_fn l, eh_refresh_fn r) { h->list_fn = l; h->refresh_fn = r; }

// kb_run_preempt / kb_run_reclaim (kb_engine.cpp: run_evict_action) around the same PreemptMachine
int eh_run(EH *h, int reclaim) {
  return guarded(h, [&]() {
    HostSession &hs = h->hs;
    if (hs.has_interpod) throw EngineError(KB_E_UNSUPPORTED, "preempt / reclaim in a session with inter-pod (anti)affinity terms is not modelled");
    const uint32_t N = hs.N, T = hs.T, J = hs.J, Q = hs.Q;
    PreemptMachine pm;
    pm.counted = h->counted;
    pm.jalloc = hs.job_alloc; pm.jshare = hs.job_share; pm.qalloc = hs.queue_alloc; pm.qshare = hs.queue_share;
    pm.jmask.assign(J ? J : 1, 0); pm.qmask.assign(Q ? Q : 1, 0);
    for (uint32_t t = 0; t < T; t++)
      if (pm.counted[t] && hs.t_job[t] < J) {
        pm.jmask[hs.t_job[t]] |= hs.t_resmask[t];
        if (hs.job_queue[hs.t_job[t]] < Q) pm.qmask[hs.job_queue[hs.t_job[t]]] |= hs.t_resmask[t];
      }
    auto lists = [&](uint32_t task, std::vector<uint64_t> &keys) {
      keys.assign(N ? N : 1, 0);
      const uint32_t k = h->list_fn(task, keys.data());
      keys.resize(k);
    };
    auto refresh = [&](const std::vector<uint32_t> &nodes) {
      std::vector<int64_t> c, m;
      std::vector<int32_t> p;
      std::vector<uint64_t> po;
      for (uint32_t n : nodes) { c.push_back(h->ln.nzc[n]); m.push_back(h->ln.nzm[n]); p.push_back(h->ln.podcnt[n]); po.push_back(h->ln.ports[n]); }
      h->refresh_fn(nodes.data(), (uint32_t)nodes.size(), c.data(), m.data(), p.data(), po.data());
    };
    pm.init(&hs, &h->pol, &h->ln, &h->status, &h->tnode, lists, refresh);
    if (reclaim) pm.run_reclaim(); else pm.run();
    // the action leaves the "device" current for the next one
    refresh(pm.touched_nodes);
    hs.t_status = h->status;
    hs.t_node = h->tnode;
    pm.off_node_tasks(hs.t_off_node);
    h->counted = pm.counted;
    hs.job_alloc = pm.jalloc; hs.job_share = pm.jshare; hs.queue_alloc = pm.qalloc; hs.queue_share = pm.qshare;
    h->ops.insert(h->ops.end(), pm.ops.begin(), pm.ops.end());
    h->evictions.insert(h->evictions.end(), pm.evictions.begin(), pm.evictions.end());
    h->popped += pm.popped;
    h->evals += pm.evals;
  });
}

uint64_t eh_n_ops(const EH *h) { return h->ops.size(); }
void eh_ops(const EH *h, uint32_t *out) {   // [n][4]: op, task, node, stmt
  for (size_t i = 0; i < h->ops.size(); i++) { out[4 * i] = h->ops[i].op; out[4 * i + 1] = h->ops[i].task; out[4 * i + 2] = h->ops[i].node; out[4 * i + 3] = h->ops[i].stmt; }
}
uint64_t eh_n_evictions(const EH *h) { return h->evictions.size(); }
void eh_evictions(const EH *h, uint32_t *out) { std::memcpy(out, h->evictions.data(), sizeof(uint32_t) * h->evictions.size()); }
uint64_t eh_popped(const EH *h) { return h->popped; }
void eh_task_state(const EH *h, uint8_t *status, uint32_t *node) {
  std::memcpy(status, h->status.data(), h->hs.T);
  std::memcpy(node, h->tnode.data(), sizeof(uint32_t) * h->hs.T);
}
void eh_node_state(const EH *h, double *idle, double *rel, int64_t *nzc, int64_t *nzm, int32_t *podcnt) {   // idle / rel: [R][N]
  const int R = h->hs.R;
  const uint32_t N = h->hs.N;
  for (uint32_t n = 0; n < N; n++) {
    for (int d = 0; d < R; d++) { idle[(size_t)d * N + n] = h->ln.idle[n].get(d); rel[(size_t)d * N + n] = h->ln.rel[n].get(d); }
    nzc[n] = h->ln.nzc[n]; nzm[n] = h->ln.nzm[n]; podcnt[n] = h->ln.podcnt[n];
  }
}
void eh_shares(const EH *h, double *jshare, double *qshare) {
  std::memcpy(jshare, h->hs.job_share.data(), sizeof(double) * h->hs.J);
  std::memcpy(qshare, h->hs.queue_share.data(), sizeof(double) * h->hs.Q);
}
// what the session build derived (checked against the second restatement): proportion's deserved [Q][R], the totals [R],
// the shape ids [T] and their counts
void eh_session(const EH *h, double *deserved, double *total, uint32_t *feas_shape, uint32_t *row_shape, uint32_t *n_shapes) {
  const int R = h->hs.R;
  for (uint32_t q = 0; q < h->hs.Q; q++)
    for (int d = 0; d < R; d++) deserved[(size_t)q * R + d] = h->hs.deserved[q].get(d);
  for (int d = 0; d < R; d++) total[d] = h->hs.total.get(d);
  std::memcpy(feas_shape, h->hs.t_feas_shape.data(), sizeof(uint32_t) * h->hs.T);
  std::memcpy(row_shape, h->hs.t_row_shape.data(), sizeof(uint32_t) * h->hs.T);
  n_shapes[0] = h->hs.n_feas_shapes;
  n_shapes[1] = h->hs.n_row_shapes;
}

// FNV-1a over everything the session build leaves in the HostSession (and its two side outputs): two builds of the same snapshot
// agree on this iff they agree on every derived array (used to compare an optimised build with its predecessor)
uint64_t eh_digest(const EH *h) {
  uint64_t x = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; i++) { x ^= b[i]; x *= 1099511628211ull; }
    x ^= n; x *= 1099511628211ull;
  };
  auto vec = [&](const auto &v) { mix(v.data(), v.size() * sizeof(v[0])); };
  const HostSession &s = h->hs;
  vec(s.t_res); vec(s.t_init); vec(s.t_res_rows); vec(s.t_resmask); vec(s.t_job); vec(s.t_cls); vec(s.t_node); vec(s.t_prio); vec(s.t_creation);
  vec(s.t_status); vec(s.t_res_empty); vec(s.t_init_empty); vec(s.t_feas_shape); vec(s.t_row_shape); vec(s.feas_eff); vec(s.feas_rep);
  vec(s.feas_cls); vec(s.feas_conf); vec(s.job_begin); vec(s.job_queue); vec(s.job_min); vec(s.job_prio); vec(s.job_creation);
  vec(s.queue_weight); vec(s.queue_creation); vec(s.t_nzc); vec(s.t_nzm); vec(s.t_want); vec(s.t_conf); vec(s.t_protected); vec(s.n_ac);
  vec(s.n_am); vec(s.n_maxpods); vec(s.n_cls); vec(s.n_idle_mask); vec(s.compat); vec(s.t_ip_subject); vec(s.t_ip_checks);
  vec(s.feas_ip_require); vec(s.feas_ip); vec(s.queue_has_attr); vec(h->t_active); vec(h->nmask);
  mix(s.total.v, sizeof(s.total.v)); mix(&s.total.mask, sizeof(uint32_t)); mix(&s.queue_share_at_open, 1);
  for (const Res &r : s.deserved) { mix(r.v, sizeof(r.v)); mix(&r.mask, sizeof(uint32_t)); }
  const uint32_t n[4] = {s.n_feas_shapes, s.n_row_shapes, s.n_tc, s.n_nc};
  mix(n, sizeof(n));
  if (s.port_xw) { vec(s.t_want_x); vec(s.t_conf_x); vec(s.t_wide); }   // host-port masks of several words (the pinned digests predate them: one-word sessions hash as before)
  return x;
}

}  // extern "C"
