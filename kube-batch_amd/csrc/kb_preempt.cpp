// kb_preempt.cpp — host side of the engine's preempt action; see kb_preempt.hpp.
#include "kb_preempt.hpp"

#include <chrono>
#include <cstdlib>

#include <climits>

namespace kb {

namespace {

// container/heap with Go's sift mechanics (util/priority_queue.go:26-94 wraps it): the job heap's keys (gang readiness, drf share)
// move while jobs sit in it, so the pop order depends on them
template <typename Less> struct GoHeap {
  std::vector<uint32_t> a;
  Less less;
  explicit GoHeap(Less l) : less(l) {}
  bool empty() const { return a.empty(); }
  void push(uint32_t x) {
    a.push_back(x);
    size_t j = a.size() - 1;
    for (;;) {
      size_t i = j == 0 ? 0 : (j - 1) / 2;
      if (i == j || !less(a[j], a[i])) break;
      std::swap(a[i], a[j]);
      j = i;
    }
  }
  uint32_t pop() {
    size_t n = a.size() - 1;
    std::swap(a[0], a[n]);
    size_t i = 0;
    for (;;) {
      size_t j1 = 2 * i + 1;
      if (j1 >= n) break;
      size_t j = j1;
      if (j1 + 1 < n && less(a[j1 + 1], a[j1])) j = j1 + 1;
      if (!less(a[j], a[i])) break;
      std::swap(a[i], a[j]);
      i = j;
    }
    uint32_t x = a.back();
    a.pop_back();
    return x;
  }
};

// The per-job queues of pending tasks (preemptorTasks / reclaim's task queues).  Their keys (priority, creation stamp, index: a strict
// total order) never change while a task waits, so a priority queue hands them out in sorted order whatever its sift mechanics are:
// one flat array, each job's segment sorted once, a cursor per job.
struct TaskQueues {
  std::vector<uint32_t> items, off, cur;
  template <typename Less> void build(uint32_t J, const std::vector<uint32_t> &job_begin, const std::vector<uint8_t> &want_job, const std::vector<uint8_t> &status, Less less) {
    off.assign((size_t)J + 1, 0);
    items.clear();
    for (uint32_t j = 0; j < J; j++) {
      off[j] = (uint32_t)items.size();
      if (!want_job[j]) continue;
      for (uint32_t t = job_begin[j]; t < job_begin[j + 1]; t++)
        if (status[t] == KB_TASK_PENDING) items.push_back(t);
      std::sort(items.begin() + off[j], items.end(), less);
    }
    off[J] = (uint32_t)items.size();
    cur.assign(off.begin(), off.end() - 1);
  }
  bool empty(uint32_t j) const { return cur[j] == off[j + 1]; }
  uint32_t pop(uint32_t j) { return items[cur[j]++]; }
  const uint32_t *rest(uint32_t j) const { return items.data() + cur[j]; }   // what is left of job j's queue, in pop order
  uint32_t left(uint32_t j) const { return off[j + 1] - cur[j]; }
  void drain(uint32_t j) { cur[j] = off[j + 1]; }
};

}  // namespace

Res PreemptMachine::task_res(uint32_t t) const {
  Res r;
  r.mask = hs_->t_resmask[t];
  for (int d = 0; d < hs_->R; d++) r.v[d] = hs_->t_res[(size_t)d * hs_->T + t];
  return r;
}
Res PreemptMachine::task_init(uint32_t t) const {   // keys: Resreq's plus every dimension an init container raised (SetMaxResource)
  Res r;
  r.mask = hs_->t_resmask[t];
  for (int d = 0; d < hs_->R; d++) {
    r.v[d] = hs_->t_init[(size_t)d * hs_->T + t];
    if (d >= 2 && r.v[d] != 0.0) r.setk(d);
  }
  return r;
}
int PreemptMachine::ready_num(uint32_t j) const {   // api/job_info.go:383-394
  const int32_t *c = &cnt[(size_t)j * 10];
  return c[KB_TASK_BOUND] + c[KB_TASK_BINDING] + c[KB_TASK_RUNNING] + c[KB_TASK_ALLOCATED] + c[KB_TASK_SUCCEEDED];
}
bool PreemptMachine::job_pipelined(uint32_t j) const {   // session_plugins.go:202-222 + gang.go:126-129 -> job_info.go Pipelined()
  if (!pol_->gang_job_pipelined) return true;
  return cnt[(size_t)j * 10 + KB_TASK_PIPELINED] + ready_num(j) >= hs_->job_min[j];
}
bool PreemptMachine::job_less(uint32_t l, uint32_t r) const {   // session_plugins.go:243-267 + priority.go:61-77, gang.go:96-119, drf.go:114-130
  for (uint8_t p : pol_->job_chain) {
    int j = 0;
    if (p == KB_PLUGIN_PRIORITY) {
      if (hs_->job_prio[l] > hs_->job_prio[r]) j = -1;
      else if (hs_->job_prio[l] < hs_->job_prio[r]) j = 1;
    } else if (p == KB_PLUGIN_GANG) {
      bool lr = ready_num(l) >= hs_->job_min[l], rr = ready_num(r) >= hs_->job_min[r];
      if (lr && rr) j = 0; else if (lr) j = 1; else if (rr) j = -1;
    } else if (p == KB_PLUGIN_DRF) {
      if (jshare[l] == jshare[r]) j = 0; else if (jshare[l] < jshare[r]) j = -1; else j = 1;
    }
    if (j != 0) return j < 0;
  }
  if (hs_->job_creation[l] == hs_->job_creation[r]) return l < r;
  return hs_->job_creation[l] < hs_->job_creation[r];
}
bool PreemptMachine::task_less(uint32_t l, uint32_t r) const {   // session_plugins.go:298-331 + priority.go:40-56
  if (pol_->task_order_priority && hs_->t_prio[l] != hs_->t_prio[r]) return hs_->t_prio[l] > hs_->t_prio[r];
  if (hs_->t_creation[l] != hs_->t_creation[r]) return hs_->t_creation[l] < hs_->t_creation[r];
  return l < r;
}
double PreemptMachine::drf_share(const double *alloc, uint32_t mask) const {   // drf.go:157-171
  double share = 0;
  for (int d = 0; d < hs_->R; d++) {
    if (d >= 2 && !hs_->total.has(d)) continue;
    const double a = (d < 2 || ((mask >> (d - 2)) & 1u)) ? alloc[d] : 0.0;
    const double s = helpers_share(a, hs_->total.get(d));
    if (s > share) share = s;
  }
  return share;
}

// drf.go:135-156, proportion.go:212-235: AllocateFunc / DeallocateFunc (fired by Evict, Pipeline and their undo)
void PreemptMachine::fire_allocate(uint32_t t) {
  const int R = hs_->R;
  const uint32_t j = hs_->t_job[t], tm = hs_->t_resmask[t];
  if (pol_->has_drf) {
    double *a = &jalloc[(size_t)j * R];
    for (int d = 0; d < R; d++)
      if (d < 2 || ((tm >> (d - 2)) & 1u)) a[d] += hs_->t_res[(size_t)d * hs_->T + t];
    jmask[j] |= tm;
    jshare[j] = drf_share(a, jmask[j]);
  }
  if (pol_->has_proportion && hs_->job_queue[j] < hs_->Q) {
    const uint32_t q = hs_->job_queue[j];
    double *a = &qalloc[(size_t)q * R];
    for (int d = 0; d < R; d++)
      if (d < 2 || ((tm >> (d - 2)) & 1u)) a[d] += hs_->t_res[(size_t)d * hs_->T + t];
    qmask[q] |= tm;
    const Res &des = hs_->deserved[q];
    double share = 0;
    for (int d = 0; d < R; d++) {
      if (d >= 2 && !des.has(d)) continue;
      const double s = helpers_share((d < 2 || ((qmask[q] >> (d - 2)) & 1u)) ? a[d] : 0.0, des.get(d));
      if (s > share) share = s;
    }
    qshare[q] = share;
  }
  counted[t] = 1;
}
void PreemptMachine::fire_deallocate(uint32_t t) {
  const int R = hs_->R;
  const uint32_t j = hs_->t_job[t], tm = hs_->t_resmask[t];
  auto sub = [&](double *a, uint32_t &mask) {   // Resource.Sub (resource_info.go:143-160) on the dense vector + key mask
    Res r, rr = task_res(t);
    r.mask = mask;
    for (int d = 0; d < R; d++) r.v[d] = a[d];
    if (!res_sub(r, rr, R)) throw EngineError(KB_E_UNSUPPORTED, "preempt: a plugin's allocated vector would underflow (the reference panics in Resource.Sub)");
    for (int d = 0; d < R; d++) a[d] = r.v[d];
    mask = r.mask;
  };
  (void)tm;
  if (pol_->has_drf) {
    double *a = &jalloc[(size_t)j * R];
    sub(a, jmask[j]);
    jshare[j] = drf_share(a, jmask[j]);
  }
  if (pol_->has_proportion && hs_->job_queue[j] < hs_->Q) {
    const uint32_t q = hs_->job_queue[j];
    double *a = &qalloc[(size_t)q * R];
    sub(a, qmask[q]);
    const Res &des = hs_->deserved[q];
    double share = 0;
    for (int d = 0; d < R; d++) {
      if (d >= 2 && !des.has(d)) continue;
      const double s = helpers_share((d < 2 || ((qmask[q] >> (d - 2)) & 1u)) ? a[d] : 0.0, des.get(d));
      if (s > share) share = s;
    }
    qshare[q] = share;
  }
  counted[t] = 0;
}

void PreemptMachine::set_status(uint32_t t, int st) {   // job_info.go:247-264 UpdateTaskStatus = delete + add
  const uint32_t j = hs_->t_job[t];
  cnt[(size_t)j * 10 + (*status_)[t]]--;
  (*status_)[t] = (uint8_t)st;
  cnt[(size_t)j * 10 + st]++;
}

void PreemptMachine::touch_node(uint32_t n) {
  if (!touched_[n]) { touched_[n] = 1; touched_nodes.push_back(n); }
}
void PreemptMachine::mark_dirty(uint32_t n) {   // what the plugin predicates / scorers read changed: cached lists are stale for n
  if (!dirty_[n]) { dirty_[n] = 1; dirty_nodes_.push_back(n); }
}
void PreemptMachine::recompute_minprio(uint32_t n) {
  if (!prio_prunes_) return;
  const uint32_t Q = hs_->Q;
  std::vector<std::pair<uint32_t, int32_t>> &v = nq_minprio_[n];
  v.clear();
  for (uint32_t t : ntasks_[n]) {
    if (node_status[t] != KB_TASK_RUNNING) continue;
    const uint32_t j = hs_->t_job[t], q = hs_->job_queue[j];
    if (q >= Q) continue;
    size_t k = 0;
    while (k < v.size() && v[k].first != q) k++;
    if (k == v.size()) v.emplace_back(q, hs_->job_prio[j]);
    else if (hs_->job_prio[j] < v[k].second) v[k].second = hs_->job_prio[j];
  }
}
// lowest job priority among node n's Running session tasks of queue q (INT_MAX: none): a node holds tasks of a handful of queues
int32_t PreemptMachine::minprio(uint32_t q, uint32_t n) const {
  for (const std::pair<uint32_t, int32_t> &e : nq_minprio_[n])
    if (e.first == q) return e.second;
  return INT_MAX;
}

// NodeInfo.RemoveTask (api/node_info.go:217-243): accounting by the status of the node's own clone
void PreemptMachine::node_remove(uint32_t t) {
  if (!on_node[t]) return;   // "failed to find task on host": logged, nothing changes
  const int R = hs_->R;
  const uint32_t n = (*tnode_)[t];
  const Res rq = task_res(t);
  switch (node_status[t]) {
    case KB_TASK_RELEASING:
      if (!res_sub(nd_->rel[n], rq, R)) throw EngineError(KB_E_UNSUPPORTED, "preempt: Releasing would underflow (the reference panics in Resource.Sub)");
      res_add(nd_->idle[n], rq, R);
      break;
    case KB_TASK_PIPELINED: res_add(nd_->rel[n], rq, R); break;
    default: res_add(nd_->idle[n], rq, R); break;
  }
  nd_->podcnt[n] -= 1;
  nd_->nzc[n] -= hs_->t_nzc[t];
  nd_->nzm[n] -= hs_->t_nzm[t];
  std::vector<uint32_t> &v = ntasks_[n];
  v.erase(std::lower_bound(v.begin(), v.end(), t));
  if (!hs_->t_want.empty()) {   // UsedPorts is rebuilt from the remaining pods
    uint64_t ports = nd_->base_ports[n];
    for (uint32_t o : v) ports |= hs_->t_want[o];
    nd_->ports[n] = ports;
    const uint32_t X = hs_->port_xw;
    for (uint32_t w = 0; w < X; w++) {
      uint64_t px = nd_->base_ports_x[(size_t)n * X + w];
      for (uint32_t o : v) px |= hs_->t_want_x[(size_t)o * X + w];
      nd_->ports_x[(size_t)n * X + w] = px;
    }
  }
  on_node[t] = 0;
  touch_node(n);
}
// NodeInfo.AddTask (api/node_info.go:172-212) for a task whose session status is already `st`
bool PreemptMachine::node_add(uint32_t t, uint32_t n, int st) {
  const int R = hs_->R;
  if (((*tnode_)[t] != KB_NONE && (*tnode_)[t] != n) || on_node[t]) return false;   // NodeName is sticky (node_info.go:173-182)
  const Res rq = task_res(t);
  switch (st) {
    case KB_TASK_RELEASING:
      if (!res_less_equal(rq, nd_->idle[n], R)) return false;
      if (!res_sub(nd_->idle[n], rq, R)) throw EngineError(KB_E_UNSUPPORTED, "preempt: Idle would underflow");
      res_add(nd_->rel[n], rq, R);
      break;
    case KB_TASK_PIPELINED:
      if (!res_sub(nd_->rel[n], rq, R)) throw EngineError(KB_E_UNSUPPORTED, "preempt: Releasing would underflow (the reference panics in Resource.Sub)");
      break;
    default:
      if (!res_less_equal(rq, nd_->idle[n], R)) return false;
      if (!res_sub(nd_->idle[n], rq, R)) throw EngineError(KB_E_UNSUPPORTED, "preempt: Idle would underflow");
      break;
  }
  (*tnode_)[t] = n;
  node_status[t] = (uint8_t)st;
  on_node[t] = 1;
  nd_->podcnt[n] += 1;
  nd_->nzc[n] += hs_->t_nzc[t];
  nd_->nzm[n] += hs_->t_nzm[t];
  if (!hs_->t_want.empty()) nd_->ports[n] |= hs_->t_want[t];
  for (uint32_t w = 0, X = hs_->port_xw; w < X; w++) nd_->ports_x[(size_t)n * X + w] |= hs_->t_want_x[(size_t)t * X + w];
  std::vector<uint32_t> &v = ntasks_[n];
  v.insert(std::lower_bound(v.begin(), v.end(), t), t);
  touch_node(n);
  return true;
}
// NodeInfo.UpdateTask (node_info.go:245-256) = RemoveTask + AddTask; an AddTask error there is glog.Fatalf
void PreemptMachine::node_update(uint32_t t, int st) {
  if (!on_node[t]) return;
  const uint32_t n = (*tnode_)[t];
  node_remove(t);
  if (!node_add(t, n, st)) throw EngineError(KB_E_UNSUPPORTED, "preempt: NodeInfo.UpdateTask could not re-add the task (the reference aborts with glog.Fatalf)");
  recompute_minprio(n);
}

// ---- inter-pod (anti)affinity under the evict actions ----
void PreemptMachine::set_interpod(IpLive *ip, std::function<void()> upload) {
  ip_ = ip; ip_upload_ = std::move(upload);
  ip_changed_ = false;
  ip_z0_ = ip ? ip->z : KB_NONE;   // the pods Z stands for when the action starts stay where they are: only this action's Pipelines come and go
  ip_unb_n_.assign(hs_->N ? hs_->N : 1, 0);
}
void PreemptMachine::ip_allocated_status(uint32_t t, int joins) {   // task t, in its node's Tasks, enters (+1) / leaves (-1) the allocated statuses
#ifdef KB_NEGATIVE_CONTROL_NO_IP_EVICT   // a test build (tests/README.md): the differential cases must notice an eviction that leaves the counts alone
  return;
#endif
  if (!ip_ || !on_node[t]) return;
  const uint32_t n = (*tnode_)[t], Wc = hs_->ip_Wc;
  for (uint32_t w = 0; w < Wc; w++) {
    uint64_t m = hs_->ip_task_inc[(size_t)t * Wc + w];
    for (uint32_t c = 64 * w; m; c++, m >>= 1) {
      if (!(m & 1)) continue;
      ip_->ctot[c] += joins;
      const uint32_t d = hs_->ip_ctr_dom[(size_t)c * hs_->N + n];
      if (d != KB_NONE) ip_->ccnt[(size_t)c * hs_->ip_D + d] += joins;
    }
  }
  ip_changed_ = true;
}
void PreemptMachine::ip_placed(uint32_t t, uint32_t n) {   // a Pipeline put task t into ni.Tasks of node n (Spec.NodeName still empty)
  if (!ip_) return;
  const uint32_t Wp = hs_->ip_Wp;
  for (uint32_t w = 0; w < Wp; w++) {
    uint64_t m = hs_->ip_task_cls_inc[(size_t)t * Wp + w];
    for (uint32_t p = 64 * w; m; p++, m >>= 1)
      if (m & 1) ip_->punb[(size_t)p * ip_->NP + n] += 1;
  }
  ip_unb_n_[n] += 1;
  if (n < ip_->z) ip_->z = n;
  ip_changed_ = true;
}
void PreemptMachine::ip_unplaced(uint32_t t, uint32_t n) {   // ... and its undo takes it out again
  if (!ip_) return;
  const uint32_t Wp = hs_->ip_Wp;
  for (uint32_t w = 0; w < Wp; w++) {
    uint64_t m = hs_->ip_task_cls_inc[(size_t)t * Wp + w];
    for (uint32_t p = 64 * w; m; p++, m >>= 1)
      if (m & 1) ip_->punb[(size_t)p * ip_->NP + n] -= 1;
  }
  ip_unb_n_[n] -= 1;
  if (n == ip_->z && ip_unb_n_[n] == 0 && n != ip_z0_) {   // Z = the first node (ascending) that holds any pod with an empty Spec.NodeName
    uint32_t z = ip_z0_;
    for (uint32_t i = n + 1; i < hs_->N && i < ip_z0_; i++)
      if (ip_unb_n_[i] > 0) { z = i; break; }
    ip_->z = z;
  }
  ip_changed_ = true;
}
// PodAffinityChecker.InterPodAffinityMatches on the kb_interpod counters (the arithmetic of kb_kernels.hip: interpod_ok, on the host's live counts)
bool PreemptMachine::ip_predicate(uint32_t t, uint32_t n) const {
  const uint32_t Wc = hs_->ip_Wc;
  for (uint32_t w = 0; w < Wc; w++) {
    uint64_t fb = hs_->ip_task_forbid[(size_t)t * Wc + w];
    for (uint32_t c = 64 * w; fb; c++, fb >>= 1) {
      if (!(fb & 1)) continue;
      const uint32_t d = hs_->ip_ctr_dom[(size_t)c * hs_->N + n];
      if (d != KB_NONE && ip_->ccnt[(size_t)c * hs_->ip_D + d] > 0) return false;
    }
  }
  const uint32_t r = hs_->ip_task_require[t];
  if (r != 0xFFFFu) {
    const uint32_t d = hs_->ip_ctr_dom[(size_t)r * hs_->N + n];
    if (!(d != KB_NONE && ip_->ccnt[(size_t)r * hs_->ip_D + d] > 0))
      if (ip_->ctot[r] > 0 || !hs_->ip_task_self[t]) return false;
  }
  return true;
}

void PreemptMachine::evict(uint32_t t) {   // statement.go:36-69
  version_++;
  ip_allocated_status(t, -1);
  set_status(t, KB_TASK_RELEASING);
  node_update(t, KB_TASK_RELEASING);
  fire_deallocate(t);
  ops.push_back(StmtOp{KB_OP_EVICT, t, (*tnode_)[t], stmt_no_});
}
void PreemptMachine::unevict(uint32_t t) {   // statement.go:83-110
  version_++;
  set_status(t, KB_TASK_RUNNING);
  node_update(t, KB_TASK_RUNNING);
  ip_allocated_status(t, +1);
  fire_allocate(t);
}
void PreemptMachine::pipeline(uint32_t t, uint32_t n) {   // statement.go:113-150 (an AddTask error is logged, the handlers still run)
  version_++;
  set_status(t, KB_TASK_PIPELINED);
  if (node_add(t, n, KB_TASK_PIPELINED)) { mark_dirty(n); ip_placed(t, n); }
  fire_allocate(t);
  ops.push_back(StmtOp{KB_OP_PIPELINE, t, n, stmt_no_});
}
// ssn.Pipeline (framework/session.go:194-232), what reclaim calls: unlike Statement.Pipeline an AddTask error (a sticky NodeName
// left by a discarded preempt statement) returns BEFORE the plugin event handlers — the status is Pipelined, nothing else moved
void PreemptMachine::pipeline_session(uint32_t t, uint32_t n) {
  version_++;
  set_status(t, KB_TASK_PIPELINED);
  ops.push_back(StmtOp{KB_OP_PIPELINE, t, n, stmt_no_});
  if (!node_add(t, n, KB_TASK_PIPELINED)) return;
  mark_dirty(n);
  ip_placed(t, n);
  fire_allocate(t);
}
void PreemptMachine::unpipeline(uint32_t t) {   // statement.go:155-190; task.NodeName keeps the old host (RemoveTask never clears it)
  version_++;
  set_status(t, KB_TASK_PENDING);
  if (on_node[t]) { const uint32_t n = (*tnode_)[t]; ip_unplaced(t, n); node_remove(t); mark_dirty(n); }
  fire_deallocate(t);
}
void PreemptMachine::begin_stmt() {
  stmt_no_++;
  stmt_begin_ = ops.size();
}
void PreemptMachine::commit() {   // statement.go:208-220: evict -> cache.Evict, pipeline -> no-op
  const size_t end = ops.size();
  for (size_t i = stmt_begin_; i < end; i++)
    if (ops[i].op == KB_OP_EVICT) evictions.push_back(ops[i].task);
  if (end > stmt_begin_) ops.push_back(StmtOp{KB_OP_COMMIT, KB_NONE, KB_NONE, stmt_no_});   // an empty statement leaves no trace
  stmt_begin_ = ops.size();
}
void PreemptMachine::discard() {   // statement.go:193-205: newest first
  const size_t end = ops.size();
  for (size_t i = end; i-- > stmt_begin_;) {
    if (ops[i].op == KB_OP_EVICT) unevict(ops[i].task);
    else if (ops[i].op == KB_OP_PIPELINE) unpipeline(ops[i].task);
  }
  if (end > stmt_begin_) ops.push_back(StmtOp{KB_OP_DISCARD, KB_NONE, KB_NONE, stmt_no_});
  stmt_begin_ = ops.size();
}

// session_plugins.go:122-162: per tier the intersection of the enabled plugins' candidates; once a plugin has spoken an empty
// set stays empty (Go: a nil slice intersected with anything is nil), and the first tier that leaves a non-empty set decides
size_t PreemptMachine::evictable(uint32_t preemptor, const std::vector<uint32_t> &pre, std::vector<uint32_t> &victims, bool reclaim) {
  const int R = hs_->R;
  const size_t n = pre.size();
  bool init = false;
  victims.clear();
  std::vector<uint8_t> &keep = scratch_keep_;   // (members: a node tried costs no allocation)
  keep.assign(n ? n : 1, 0);
  for (const std::vector<uint8_t> &tier : (reclaim ? pol_->reclaim_tiers : pol_->preempt_tiers)) {
    for (uint8_t plugin : tier) {
      std::fill(keep.begin(), keep.end(), 0);
      if (plugin == KB_PLUGIN_PROPORTION) {   // proportion.go:171-196: running per-queue allocation, in reclaimee order
        std::vector<uint32_t> &aq = scratch_ids_;
        std::vector<Res> &alloc = scratch_alloc_;
        aq.clear(); alloc.clear();
        for (size_t i = 0; i < n; i++) {
          const uint32_t q = hs_->job_queue[hs_->t_job[pre[i]]];
          if (q >= hs_->Q) continue;
          size_t a = 0;
          while (a < aq.size() && aq[a] != q) a++;
          if (a == aq.size()) {
            aq.push_back(q);
            Res r;
            r.mask = qmask[q];
            for (int d = 0; d < R; d++) r.v[d] = qalloc[(size_t)q * R + d];
            alloc.push_back(r);
          }
          const Res rq = task_res(pre[i]);
          if (res_less(alloc[a], rq, R)) continue;
          if (!res_sub(alloc[a], rq, R)) throw EngineError(KB_E_UNSUPPORTED, "reclaim: queue allocation would underflow (the reference panics in Resource.Sub)");
          keep[i] = res_less_equal(hs_->deserved[q], alloc[a], R);
        }
      } else
      if (plugin == KB_PLUGIN_CONFORMANCE) {   // conformance.go:44-58
        for (size_t i = 0; i < n; i++) keep[i] = hs_->t_protected.empty() || !hs_->t_protected[pre[i]];
      } else if (plugin == KB_PLUGIN_GANG) {   // gang.go:71-90
        for (size_t i = 0; i < n; i++) {
          const uint32_t j = hs_->t_job[pre[i]];
          keep[i] = (hs_->job_min[j] <= ready_num(j) - 1) || hs_->job_min[j] == 1;
        }
      } else if (plugin == KB_PLUGIN_PRIORITY) {   // priority.go:81-98
        const int32_t pp = hs_->job_prio[hs_->t_job[preemptor]];
        for (size_t i = 0; i < n; i++) keep[i] = hs_->job_prio[hs_->t_job[pre[i]]] < pp;
      } else if (plugin == KB_PLUGIN_DRF) {   // drf.go:84-109: running per-job allocation, in preemptee order
        const uint32_t pj = hs_->t_job[preemptor];
        Res lalloc;
        lalloc.mask = jmask[pj];
        for (int d = 0; d < R; d++) lalloc.v[d] = jalloc[(size_t)pj * R + d];
        res_add(lalloc, task_res(preemptor), R);
        const double ls = drf_share(lalloc.v, lalloc.mask);
        std::vector<uint32_t> &ajob = scratch_ids_;
        std::vector<Res> &alloc = scratch_alloc_;
        ajob.clear(); alloc.clear();
        for (size_t i = 0; i < n; i++) {
          const uint32_t jb = hs_->t_job[pre[i]];
          size_t a = 0;
          while (a < ajob.size() && ajob[a] != jb) a++;
          if (a == ajob.size()) {
            ajob.push_back(jb);
            Res r;
            r.mask = jmask[jb];
            for (int d = 0; d < R; d++) r.v[d] = jalloc[(size_t)jb * R + d];
            alloc.push_back(r);
          }
          if (!res_sub(alloc[a], task_res(pre[i]), R)) throw EngineError(KB_E_UNSUPPORTED, "preempt: drf allocation would underflow (the reference panics in Resource.Sub)");
          const double rs = drf_share(alloc[a].v, alloc[a].mask);
          keep[i] = (ls < rs) || (std::fabs(ls - rs) <= 0.000001);   // shareDelta (drf.go:33)
        }
      } else {
        continue;
      }
      if (!init) {
        for (size_t i = 0; i < n; i++) if (keep[i]) victims.push_back(pre[i]);
        init = true;
      } else {
        size_t w = 0;
        for (size_t v = 0; v < victims.size(); v++) {
          bool in = false;
          for (size_t i = 0; i < n && !in; i++) in = keep[i] && pre[i] == victims[v];
          if (in) victims[w++] = victims[v];
        }
        victims.resize(w);
      }
    }
    if (!victims.empty()) break;
  }
  return victims.size();
}

// plugin predicates (predicates.go:123-265 with the static checks folded into classes) and nodeorder's resource scorers against
// the LIVE host state of one node — the repair path for nodes a Pipeline changed since the device built the lists
bool PreemptMachine::host_eval(uint32_t t, uint32_t n, long long &score) const {
  score = 0;
  if (pol_->pred_enabled) {
    if (nd_->maxpods[n] <= nd_->podcnt[n]) return false;   // predicates.go:127
    if (!hs_->compat.empty()) {
      const uint32_t bit = hs_->t_cls[t] * hs_->n_nc + nd_->cls[n];
      if (!((hs_->compat[bit >> 3] >> (bit & 7)) & 1)) return false;
    }
    if (!hs_->t_conf.empty() && (nd_->ports[n] & hs_->t_conf[t])) return false;
    for (uint32_t w = 0, X = hs_->port_xw; w < X; w++)
      if (nd_->ports_x[(size_t)n * X + w] & hs_->t_conf_x[(size_t)t * X + w]) return false;
    if (ip_ && hs_->t_ip_checks[t] && !ip_predicate(t, n)) return false;   // predicates.go:249-262 (the score below is only read where lists are repaired: never with inter-pod terms)
  }
  if (!pol_->nodeorder_enabled) return true;
  const long long rc = nd_->nzc[n] + hs_->t_nzc[t], rm = nd_->nzm[n] + hs_->t_nzm[t], ac = nd_->ac[n], am = nd_->am[n];
  auto least = [](long long req, long long cap) -> long long { return (cap == 0 || req > cap) ? 0 : ((cap - req) * 10) / cap; };   // least_requested.go:36-58
  auto most = [](long long req, long long cap) -> long long { return (cap == 0 || req > cap) ? 0 : (req * 10) / cap; };             // most_requested.go:34-61
  const long long l = (least(rc, ac) + least(rm, am)) / 2, m = (most(rc, ac) + most(rm, am)) / 2;
  const double cf = ac == 0 ? 1.0 : (double)rc / (double)ac, mf = am == 0 ? 1.0 : (double)rm / (double)am;   // balanced_resource_allocation.go:42-79
  long long b = 0;
  if (!(cf >= 1.0 || mf >= 1.0)) b = (long long)((1.0 - std::fabs(cf - mf)) * 10.0);
  score = l * pol_->wL + m * pol_->wM + b * pol_->wB;
  return true;
}

// one node of preempt()'s walk (preempt.go:195-254); mode 0: victims are Running tasks of OTHER jobs in the preemptor job's queue,
// mode 1: of the preemptor's own job
bool PreemptMachine::try_node(uint32_t preemptor, int mode, uint32_t n) {
  tr_tries++;
  const int R = hs_->R;
  const uint32_t pj = hs_->t_job[preemptor], pq = hs_->job_queue[pj];
  if (prio_prunes_ && mode == 0 && pq < hs_->Q && minprio(pq, n) >= hs_->job_prio[pj]) return false;   // no task the priority rule would let go
  std::vector<uint32_t> &pre = scratch_pre_, &victims = scratch_victims_;   // (members: no allocation per node tried)
  pre.clear(); victims.clear();
  for (uint32_t t : ntasks_[n]) {   // node.Tasks in canonical order, filtered (preempt.go:112-124 / :150-157)
    if (node_status[t] != KB_TASK_RUNNING) continue;
    const uint32_t j = hs_->t_job[t];
    if (mode == 0) { if (!(hs_->job_queue[j] == pq && j != pj)) continue; }
    else if (j != pj) continue;
    pre.push_back(t);
  }
  if (pre.empty()) return false;
  if (evictable(preemptor, pre, victims) == 0) return false;   // validateVictims: "no victims"
  const Res init = task_init(preemptor);
  Res all;
  for (uint32_t v : victims) res_add(all, task_res(v), R);
  if (!res_less_equal(init, all, R)) return false;   // "not enough resources"
  auto vless = [this](uint32_t l, uint32_t r) { return !task_less(l, r); };   // preempt.go:223-225
  GoHeap<decltype(vless)> vq(vless);
  for (uint32_t v : victims) vq.push(v);
  Res preempted;
  while (!vq.empty()) {   // lowest priority first (preempt.go:229-241)
    const uint32_t v = vq.pop();
    evict(v);
    res_add(preempted, task_res(v), R);
    if (res_less_equal(init, preempted, R)) break;
  }
  if (res_less_equal(init, preempted, R)) {   // preempt.go:247-256
    pipeline(preemptor, n);
    return true;
  }
  return false;
}

// preempt(): preempt.go:171-254
bool PreemptMachine::preempt_one(uint32_t preemptor, int mode) {
  popped++;
  evals += hs_->N;
  // with the priority rule in the deciding tier a task of the preemptor's own job can never be a victim
  if (prio_prunes_ && mode == 1) { tr_pruned++; return false; }
  // What a preemptor tries depends on its job, its matrix row and its Resreq alone, so when one found nothing AND left everything as it
  // was (try_node evicts only once the victims are known to cover InitResreq; the version check covers the rounding corner where the
  // evictions then fall short), the next task of the same job with the same three, met before anything else moved, finds nothing
  // either.  The tasks of a gang are popped back to back: one walk per job instead of one per task.
  if (fail_version_ == version_ && fail_mode_ == mode && same_preemptor_class(fail_task_, preemptor)) { tr_shortcut++; return false; }
  const uint64_t before = version_;
  const auto tw0 = std::chrono::steady_clock::now();
  const bool found = preempt_walk(preemptor, mode);
  tr_walks++; tr_walk_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
  if (!found && version_ == before) { fail_version_ = version_; fail_mode_ = mode; fail_task_ = preemptor; }
  return found;
}

bool PreemptMachine::same_preemptor_class(uint32_t a, uint32_t b) const {
  if (a == KB_NONE || hs_->t_job[a] != hs_->t_job[b] || hs_->t_row_shape[a] != hs_->t_row_shape[b] || hs_->t_resmask[a] != hs_->t_resmask[b]) return false;
  for (int d = 0; d < hs_->R; d++)
    if (hs_->t_res[(size_t)d * hs_->T + a] != hs_->t_res[(size_t)d * hs_->T + b]) return false;
  return true;
}

// A preemptor whose class has preferred node-affinity terms is scored with NormalizeReduce over ITS feasible set
// (vendor/.../priorities/reduce.go:28-63 behind util.PrioritizeNodes): one node leaving that set (pod cap reached, a host port
// taken) can change every other node's score, so a cached list cannot be repaired node by node — it is rebuilt on the device.
bool PreemptMachine::needs_exact_list(uint32_t preemptor) const {
  return pol_->nodeorder_enabled && !hs_->cls_has_aff.empty() && hs_->cls_has_aff[hs_->t_cls[preemptor]];
}

bool PreemptMachine::preempt_walk(uint32_t preemptor, int mode) {
  // too many repaired nodes: bring the device up to date and rebuild the lists on demand.  Only the full walk below pays per dirty
  // node; with the priority rule deciding, a dirty node costs one evaluation when it is a candidate and nothing otherwise.
  if ((dirty_nodes_.size() > 256 && !prio_prunes_) || (!dirty_nodes_.empty() && needs_exact_list(preemptor)) || (ip_ && (ip_changed_ || !dirty_nodes_.empty()))) {
    refresh_(dirty_nodes_);
    for (uint32_t n : dirty_nodes_) dirty_[n] = 0;
    dirty_nodes_.clear();
    std::fill(shape_have_.begin(), shape_have_.end(), 0);
    if (ip_) {   // inter-pod terms: a placement or an eviction moves a whole topology domain, and the priority is normalised over the feasible
      ip_upload_();   // set: no list survives a change; the device evaluates against the counts as they stand now
      ip_changed_ = false;
    }
  }
  const uint32_t sh = hs_->t_row_shape[preemptor];
  const uint32_t pj = hs_->t_job[preemptor], pq = hs_->job_queue[pj];
  // With the priority rule deciding, only the few nodes that hold a lower-priority Running task of the preemptor's queue are ever tried
  // (below): their keys are a handful of evaluations, which the host makes itself (host_eval: the scorer the lists are repaired with,
  // bit for bit the device's) — no list of all N nodes is fetched, sorted and ranked for the shape.  Round 4: 319 lists = 26 ms on the
  // device + 4 ms of host reordering + a 50k-entry rank table each, at 1M x 50k.  Scores that are normalised over the feasible set
  // (preferred node affinity, inter-pod terms) still need the device's list.
  const bool queue_candidates = prio_prunes_ && mode == 0 && pq < hs_->Q;
  const bool host_keys = queue_candidates && !needs_exact_list(preemptor) && !ip_;
  if (!host_keys && !shape_have_[sh]) {
    std::vector<uint64_t> keys;
    lists_(preemptor, keys);
    shape_list_[sh].assign(keys.begin(), keys.end());
    shape_have_[sh] = 1;
    std::vector<int32_t> &rk = shape_rank_[sh];
    rk.assign(hs_->N ? hs_->N : 1, -1);
    for (size_t i = 0; i < keys.size(); i++) rk[(uint32_t)keys[i]] = (int32_t)i;
  }
  const std::vector<uint64_t> &L = shape_list_[sh];
  if (queue_candidates) {
    // Only nodes that hold a Running task of the preemptor's queue with a lower job priority can yield victims (try_node would
    // reject every other node at once), and there are few of them: take THEIR keys — the list entry for a clean node, a live
    // re-evaluation for a node a Pipeline changed — and walk them in SortNodes' order.  Same nodes, same order, same outcome
    // as walking the whole list.
    std::vector<uint64_t> C;
    const int32_t pp = hs_->job_prio[pj];
    const std::vector<int32_t> &rank = shape_rank_[sh];
    const auto tsc0 = std::chrono::steady_clock::now();
    tr_scan_nodes += qnodes_[pq].size();
    for (uint32_t n : qnodes_[pq]) {
      if (minprio(pq, n) >= pp) continue;
      if (host_keys || dirty_[n]) {
        long long sc;
        if (host_eval(preemptor, n, sc)) C.push_back(((uint64_t)sc << 32) | n);
      } else if (rank[n] >= 0) {
        C.push_back(L[(size_t)rank[n]]);
      }
    }
    tr_scan_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tsc0).count();
    // descending key order, lazily: the walk usually ends at one of the first candidates (keys are distinct: the node is part of them)
    std::make_heap(C.begin(), C.end());
    for (auto end = C.end(); end != C.begin(); --end) {
      std::pop_heap(C.begin(), end);
      if (try_node(preemptor, mode, (uint32_t)*(end - 1))) return true;
    }
    return false;
  }
  // the nodes changed since the lists were built, re-evaluated against their live state
  std::vector<uint64_t> D;
  for (uint32_t n : dirty_nodes_) {
    long long sc;
    if (host_eval(preemptor, n, sc)) D.push_back(((uint64_t)sc << 32) | n);
  }
  std::sort(D.begin(), D.end(), [](uint64_t a, uint64_t b) { return a > b; });
  size_t i = 0, k = 0;
  for (;;) {
    while (i < L.size() && dirty_[(uint32_t)L[i]]) i++;
    uint32_t n;
    if (i < L.size() && (k >= D.size() || L[i] > D[k])) n = (uint32_t)L[i++];
    else if (k < D.size()) n = (uint32_t)D[k++];
    else break;
    if (try_node(preemptor, mode, n)) return true;
  }
  return false;
}

void PreemptMachine::init(const HostSession *hs, const Policy *pol, LiveNodes *live, std::vector<uint8_t> *status, std::vector<uint32_t> *tnode,
                          ListFn lists, RefreshFn refresh) {
  hs_ = hs; pol_ = pol; nd_ = live; status_ = status; tnode_ = tnode;
  lists_ = std::move(lists); refresh_ = std::move(refresh);
  const uint32_t N = hs->N, T = hs->T, J = hs->J, Q = hs->Q;
  ops.clear(); evictions.clear(); touched_nodes.clear();
  popped = evals = 0;
  stmt_no_ = 0; stmt_begin_ = 0;
  version_ = 0; fail_version_ = ~0ull; fail_task_ = KB_NONE; fail_mode_ = -1;
  // (the engine keeps ONE machine for its life — its per-node lists, pruning index and scratch vectors keep their storage from one action and one
  //  cycle to the next —, so everything an earlier action may have left is put back here)
  tr_walks = tr_tries = tr_skipped = tr_pruned = tr_shortcut = 0; tr_walk_ms = tr_setup_ms = tr_scan_ms = 0.0; tr_scan_nodes = 0;
  ip_ = nullptr; ip_upload_ = nullptr; ip_changed_ = false; ip_z0_ = KB_NONE; ip_unb_n_.clear();
  cnt.assign((size_t)(J ? J : 1) * 10, 0);
  for (uint32_t t = 0; t < T; t++) if (hs->t_job[t] < J) cnt[(size_t)hs->t_job[t] * 10 + (*status)[t]]++;
  node_status.assign(T, 0);
  on_node.assign(T, 0);
  ntasks_.resize(N);
  {   // a node's tasks, ascending: counted first, so that every list is allocated once (1M tasks on 50k nodes: no regrowth), and the lists
      // keep their storage from one action to the next
    std::vector<uint32_t> per(N ? N : 1, 0);
    for (uint32_t t = 0; t < T; t++) {
      node_status[t] = (*status)[t];
      if ((*tnode)[t] != KB_NONE && (*tnode)[t] < N && !(!hs->t_off_node.empty() && hs->t_off_node[t])) { on_node[t] = 1; per[(*tnode)[t]]++; }
    }
    for (uint32_t n = 0; n < N; n++) { ntasks_[n].clear(); ntasks_[n].reserve(per[n] + 4u); }
    for (uint32_t t = 0; t < T; t++)
      if (on_node[t]) ntasks_[(*tnode)[t]].push_back(t);
  }
  nd_->base_ports.assign(N, 0);
  if (!hs->t_want.empty())
    for (uint32_t n = 0; n < N; n++) {   // ports of pods outside the session stay on the node whatever moves
      uint64_t mine = 0;
      for (uint32_t t : ntasks_[n]) mine |= hs->t_want[t];
      nd_->base_ports[n] = nd_->ports[n] & ~mine;
    }
  nd_->base_ports_x.assign((size_t)N * hs->port_xw, 0);
  for (uint32_t n = 0, X = hs->port_xw; n < N && X; n++)
    for (uint32_t w = 0; w < X; w++) {
      uint64_t mine = 0;
      for (uint32_t t : ntasks_[n]) mine |= hs->t_want_x[(size_t)t * X + w];
      nd_->base_ports_x[(size_t)n * X + w] = nd_->ports_x[(size_t)n * X + w] & ~mine;
    }
  shape_list_.assign(hs->n_row_shapes ? hs->n_row_shapes : 1, {});
  shape_have_.assign(hs->n_row_shapes ? hs->n_row_shapes : 1, 0);
  shape_rank_.assign(hs->n_row_shapes ? hs->n_row_shapes : 1, {});
  dirty_.assign(N ? N : 1, 0);
  dirty_nodes_.clear();
  touched_.assign(N ? N : 1, 0);
  // the priority rule prunes when it sits in the first tier that owns any victim rule (every later verdict is a subset of its own)
  prio_prunes_ = false;
  for (const std::vector<uint8_t> &tier : pol->preempt_tiers) {
    if (tier.empty()) continue;
    prio_prunes_ = std::find(tier.begin(), tier.end(), (uint8_t)KB_PLUGIN_PRIORITY) != tier.end();
    break;
  }
  if (prio_prunes_) {
    // sparse: per node the (queue, lowest priority) pairs of its Running session tasks — a dense Q x N table cost tens of MB and an
    // O(Q N) pass per action at 1M x 50k with hundreds of queues (round-2 advisory)
    nq_minprio_.assign(N ? N : 1, {});
    qnodes_.assign(Q ? Q : 1, {});
    for (uint32_t n = 0; n < N; n++) {
      recompute_minprio(n);
      for (const std::pair<uint32_t, int32_t> &e : nq_minprio_[n]) qnodes_[e.first].push_back(n);   // ascending n per queue
    }
  }
}

// tasks that end the action with a NodeName but outside that node's Tasks (HostSession::t_off_node); empty when there are none
void PreemptMachine::off_node_tasks(std::vector<uint8_t> &off) const {
  off.clear();
  for (uint32_t t = 0; t < hs_->T; t++)
    if ((*tnode_)[t] != KB_NONE && !on_node[t]) {
      if (off.empty()) off.assign(hs_->T, 0);
      off[t] = 1;
    }
}

// preemptAction.Execute (preempt.go:45-168).  Canonical orders where the reference ranges over Go maps: queues ascending
// QueueID, jobs ascending JobID (underRequest), a node's tasks ascending task index.
void PreemptMachine::run() {
  const uint32_t J = hs_->J, Q = hs_->Q;
  const auto t_run0 = std::chrono::steady_clock::now();
  auto jl = [this](uint32_t l, uint32_t r) { return job_less(l, r); };
  auto tl = [this](uint32_t l, uint32_t r) { return task_less(l, r); };
  std::vector<GoHeap<decltype(jl)>> qjobs(Q ? Q : 1, GoHeap<decltype(jl)>(jl));   // preemptorsMap
  TaskQueues jtasks;                                                               // preemptorTasks
  std::vector<uint8_t> qseen(Q ? Q : 1, 0), under(J ? J : 1, 0);
  for (uint32_t j = 0; j < J; j++) {   // preempt.go:55-76
    const uint32_t q = hs_->job_queue[j];
    if (q >= Q) continue;
    qseen[q] = 1;
    if (cnt[(size_t)j * 10 + KB_TASK_PENDING] != 0) {
      qjobs[q].push(j);
      under[j] = 1;
    }
  }
  jtasks.build(J, hs_->job_begin, under, *status_, tl);
  std::vector<uint32_t> active;   // under-request jobs whose task queue is not empty, ascending
  for (uint32_t j = 0; j < J; j++)
    if (under[j] && !jtasks.empty(j)) active.push_back(j);
  tr_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run0).count();
  for (uint32_t q = 0; q < Q; q++) {
    if (!qseen[q]) continue;
    for (;;) {   // between jobs within the queue (preempt.go:80-139)
      if (qjobs[q].empty()) break;
      const uint32_t pj = qjobs[q].pop();
      bool assigned = false;
      begin_stmt();
      for (;;) {
        if (jtasks.empty(pj)) break;
        const uint32_t preemptor = jtasks.pop(pj);
        const bool got = preempt_one(preemptor, 0);
        if (got) assigned = true;
        if (job_pipelined(pj)) { commit(); break; }
        // The tasks of a gang are popped back to back and are the same preemptor (job, matrix row, Resreq): when this one found nothing and
        // left everything as it was, each of the others would be popped, counted and turned away by preempt_one's first lines, one by one,
        // with nothing moving in between (the job is not pipelined and stays so).  Count them all at once.
        if (!got && fail_version_ == version_ && fail_mode_ == 0 && fail_task_ != KB_NONE && same_preemptor_class(fail_task_, preemptor)) {
          const uint32_t nrest = jtasks.left(pj);
          const uint32_t *rest = jtasks.rest(pj);
          bool all_same = true;
          for (uint32_t i = 0; i < nrest && all_same; i++) all_same = same_preemptor_class(preemptor, rest[i]);
          if (all_same && nrest) {
            popped += nrest; evals += (uint64_t)nrest * hs_->N; tr_skipped += nrest;
            jtasks.drain(pj);
          }
        }
      }
      if (!job_pipelined(pj)) { discard(); continue; }
      if (assigned) qjobs[q].push(pj);
    }
    // between tasks within a job (preempt.go:142-166): EVERY under-request job, once per queue of the outer loop, in ascending job order.
    // Only the jobs that still have a pending task in their queue do anything: they are kept as a list (1M x 50k: 100k jobs x 128
    // queues of empty visits otherwise)
    size_t keep = 0;
    for (size_t ai = 0; ai < active.size(); ai++) {
      const uint32_t j = active[ai];
      for (;;) {
        if (jtasks.empty(j)) break;
        const uint32_t preemptor = jtasks.pop(j);
        begin_stmt();
        const bool assigned = preempt_one(preemptor, 1);
        commit();
        if (!assigned) break;
      }
      if (!jtasks.empty(j)) active[keep++] = j;
    }
    active.resize(keep);
  }
}

// session_plugins.go:270-295 + proportion.go:156-169
bool PreemptMachine::queue_less(uint32_t l, uint32_t r) const {
  if (pol_->queue_order_proportion) {
    const double ls = qshare[l], rs = qshare[r];
    if (!(ls == rs)) return ls < rs;
  }
  if (hs_->queue_creation[l] == hs_->queue_creation[r]) return l < r;
  return hs_->queue_creation[l] < hs_->queue_creation[r];
}
// session_plugins.go:165-179 + proportion.go:198-209: deserved.LessEqual(allocated)
bool PreemptMachine::overused(uint32_t q) const {
  if (!pol_->has_proportion) return false;
  Res a;
  a.mask = qmask[q];
  for (int d = 0; d < hs_->R; d++) a.v[d] = qalloc[(size_t)q * hs_->R + d];
  return res_less_equal(hs_->deserved[q], a, hs_->R);
}

// reclaimAction.Execute (reclaim.go:41-193).  Canonical orders where the reference ranges over Go maps: jobs ascending JobID,
// nodes ascending name, a node's tasks ascending task index.
void PreemptMachine::run_reclaim() {
  const int R = hs_->R;
  const uint32_t J = hs_->J, Q = hs_->Q, N = hs_->N;
  auto ql = [this](uint32_t l, uint32_t r) { return queue_less(l, r); };
  auto jl = [this](uint32_t l, uint32_t r) { return job_less(l, r); };
  auto tl = [this](uint32_t l, uint32_t r) { return task_less(l, r); };
  GoHeap<decltype(ql)> queues(ql);
  std::vector<GoHeap<decltype(jl)>> qjobs(Q ? Q : 1, GoHeap<decltype(jl)>(jl));
  TaskQueues jtasks;
  std::vector<uint8_t> has_pending(J ? J : 1, 0);
  std::vector<uint8_t> qseen(Q ? Q : 1, 0);
  for (uint32_t j = 0; j < J; j++) {   // reclaim.go:55-83
    const uint32_t q = hs_->job_queue[j];
    if (q >= Q) continue;
    if (!qseen[q]) { qseen[q] = 1; queues.push(q); }
    if (cnt[(size_t)j * 10 + KB_TASK_PENDING] != 0) {
      qjobs[q].push(j);
      has_pending[j] = 1;
    }
  }
  jtasks.build(J, hs_->job_begin, has_pending, *status_, tl);
  stmt_no_ = 0;
  for (;;) {
    if (queues.empty()) break;
    const uint32_t q = queues.pop();
    if (overused(q)) continue;             // reclaim.go:96-99
    if (qjobs[q].empty()) continue;        // :102-106
    const uint32_t j = qjobs[q].pop();
    if (jtasks.empty(j)) continue;         // :109-113
    const uint32_t task = jtasks.pop(j);
    popped++;
    const Res init = task_init(task);
    bool assigned = false;
    for (uint32_t n = 0; n < N && !assigned; n++) {
      long long sc;
      evals++;
      if (!host_eval(task, n, sc)) continue;   // ssn.PredicateFn (:118-120): the plugin predicates against the live node
      std::vector<uint32_t> pre, victims;
      for (uint32_t t : ntasks_[n]) {          // :128-141: Running tasks of OTHER queues
        if (node_status[t] != KB_TASK_RUNNING) continue;
        if (hs_->job_queue[hs_->t_job[t]] != q) pre.push_back(t);
      }
      if (pre.empty()) continue;
      if (evictable(task, pre, victims, true) == 0) continue;   // :142-147
      Res all;
      for (uint32_t v : victims) res_add(all, task_res(v), R);
      if (!res_less_equal(init, all, R)) continue;              // :149-157
      Res reclaimed;
      for (uint32_t v : victims) {             // :160-172: ssn.Evict acts at once (framework/session.go:317-354)
        evict(v);
        evictions.push_back(v);
        res_add(reclaimed, task_res(v), R);
        if (res_less_equal(init, reclaimed, R)) break;
      }
      if (res_less_equal(init, reclaimed, R)) {   // :177-186: ssn.Pipeline (framework/session.go:194-232)
        pipeline_session(task, n);
        assigned = true;
      }
    }
    if (assigned) queues.push(q);            // :189-191
  }
}

}  // namespace kb
