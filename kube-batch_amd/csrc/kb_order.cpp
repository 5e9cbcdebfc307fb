// kb_order.cpp — the order machine: queue / job / task ordering of allocate.go around the device rounds.
//
// Heaps follow Go's container/heap exactly (util/priority_queue.go:26-94 wraps it): the queue heap holds one entry
// PER JOB (allocate.go:50-52), so the same queue sits in it many times while its share mutates, and the pop order
// then depends on the sift mechanics:
//   Push: append, up(n-1);  Pop: swap(0,n-1), down(0,n-1), remove last
//   up(j):   i=(j-1)/2; stop if i==j or !less(j,i); swap; j=i
//   down(i): j1=2i+1; stop if j1>=n; j=j1; if j1+1<n && less(j1+1,j1) j=j1+1; stop if !less(j,i); swap; i=j
#include "kb_host.hpp"

namespace kb {

// session_plugins.go:270-295 + proportion.go:156-169
bool OrderMachine::queue_less(uint32_t l, uint32_t r) const {
  if (pol_->queue_order_proportion) {
    double ls = qshare[l], rs = qshare[r];
    if (!(ls == rs)) return ls < rs;
  }
  if (hs_->queue_creation[l] == hs_->queue_creation[r]) return l < r;   // UID order == canonical index order
  return hs_->queue_creation[l] < hs_->queue_creation[r];
}

// session_plugins.go:243-267 + priority.go:61-77, gang.go:96-119, drf.go:114-130
bool OrderMachine::job_less(uint32_t l, uint32_t r) const {
  for (uint8_t p : pol_->job_chain) {
    int j = 0;
    if (p == KB_PLUGIN_PRIORITY) {
      if (hs_->job_prio[l] > hs_->job_prio[r]) j = -1;
      else if (hs_->job_prio[l] < hs_->job_prio[r]) j = 1;
    } else if (p == KB_PLUGIN_GANG) {
      bool lr = ready[l] >= hs_->job_min[l], rr = ready[r] >= hs_->job_min[r];
      if (lr && rr) j = 0; else if (lr) j = 1; else if (rr) j = -1;
    } else if (p == KB_PLUGIN_DRF) {
      if (jshare[l] == jshare[r]) j = 0; else if (jshare[l] < jshare[r]) j = -1; else j = 1;
    }
    if (j != 0) return j < 0;
  }
  if (hs_->job_creation[l] == hs_->job_creation[r]) return l < r;
  return hs_->job_creation[l] < hs_->job_creation[r];
}

// session_plugins.go:165-179 + proportion.go:198-209: deserved.LessEqual(allocated)
bool OrderMachine::overused(uint32_t q) const {
  if (!pol_->has_proportion) return false;
  const int R = hs_->R;
  const Res &des = hs_->deserved[q];
  const double *al = &qalloc[(size_t)q * R];
  if (!le_func(des.v[0], al[0], kMinMilliCPU)) return false;
  if (!le_func(des.v[1], al[1], kMinMemory)) return false;
  for (int d = 2; d < R; d++) {
    if (!des.has(d) || des.v[d] <= kMinMilliScalar) continue;
    // an absent key on the right reads 0 and a nil map fails exactly when the compare against 0 fails
    if (!le_func(des.v[d], al[d], kMinMilliScalar)) return false;
  }
  return true;
}

// The sifts move a hole instead of swapping (the element on its way is compared by value, the others move one level): the same comparisons in
// the same order as container/heap's up / down, the same final array, half the writes.
void OrderMachine::qpush(uint32_t q) {
  uint32_t j = qn_++;
  while (j > 0) {
    const uint32_t i = (j - 1) / 2;
    if (!queue_less(q, qheap_[i])) break;
    qset(j, qheap_[i]);
    j = i;
  }
  qset(j, q);
}
uint32_t OrderMachine::qpop() {
  const uint32_t n = --qn_;          // Pop: swap(0, n), down(0, n), remove last — the old root leaves, the last element sinks from the root
  const uint32_t top = qheap_[0];
  if (n == 0) return top;
  const uint32_t x = qheap_[n];
  uint32_t i = 0;
  for (;;) {
    const uint32_t j1 = 2 * i + 1;
    if (j1 >= n) break;
    uint32_t j = j1;
    if (j1 + 1 < n && queue_less(qheap_[j1 + 1], qheap_[j1])) j = j1 + 1;
    if (!queue_less(qheap_[j], x)) break;
    qset(i, qheap_[j]);
    i = j;
  }
  qset(i, x);
  return top;
}
void OrderMachine::jpush(uint32_t q, uint32_t job) {
  const uint32_t b = jheap_off_[q];
  const uint32_t *h = &jheap_items_[b];
  uint32_t j = jheap_n_[q]++;
  while (j > 0) {
    const uint32_t i = (j - 1) / 2;
    if (!job_less(job, h[i])) break;
    jset(b + j, h[i]);
    j = i;
  }
  jset(b + j, job);
}
uint32_t OrderMachine::jpop(uint32_t q) {
  const uint32_t b = jheap_off_[q];
  const uint32_t *h = &jheap_items_[b];
  const uint32_t n = --jheap_n_[q];
  const uint32_t top = h[0];
  if (n == 0) return top;
  const uint32_t x = h[n];
  uint32_t i = 0;
  for (;;) {
    const uint32_t j1 = 2 * i + 1;
    if (j1 >= n) break;
    uint32_t j = j1;
    if (j1 + 1 < n && job_less(h[j1 + 1], h[j1])) j = j1 + 1;
    if (!job_less(h[j], x)) break;
    jset(b + i, h[j]);
    i = j;
  }
  jset(b + i, x);
  return top;
}

void OrderMachine::init_allocate(const HostSession *hs, const Policy *pol) {
  hs_ = hs;
  pol_ = pol;
  const uint32_t J = hs->J, Q = hs->Q;
  jalloc = hs->job_alloc;
  jshare = hs->job_share;
  qalloc = hs->queue_alloc;
  qshare = hs->queue_share;
  ready = hs->job_ready;
  steps = 0;
  // per-queue job heaps share one flat array; a queue's heap never holds more than its job count
  jheap_off_.assign(Q + 1, 0);
  for (uint32_t j = 0; j < J; j++)
    if (hs->job_queue[j] < Q) jheap_off_[hs->job_queue[j] + 1]++;
  for (uint32_t q = 0; q < Q; q++) jheap_off_[q + 1] += jheap_off_[q];
  jheap_items_.assign(J ? J : 1, 0);
  // A queue's job heap is built the first time the queue is popped (build_jobs; kNotBuilt stands in its size until then, and travels with
  // the sizes through the roll-back points).  What job_less reads of a job (ReadyTaskNum, drf share) moves only when one of the job's
  // tasks is handled, i.e. after the job was popped from that very heap: the pushes find the values they would have found here.
  jheap_n_.assign(Q, kNotBuilt);
  qjobs_.assign(J ? J : 1, 0);   // the jobs of every queue, ascending (counting sort over the offsets above)
  {
    std::vector<uint32_t> fill(jheap_off_.begin(), jheap_off_.end() - 1);
    for (uint32_t j = 0; j < J; j++)
      if (hs->job_queue[j] < Q) qjobs_[fill[hs->job_queue[j]]++] = j;
  }
  qheap_.assign(J ? J : 1, 0);
  qn_ = 0;
  // the heaps' journals (large sessions): every stamp equals frame 0's epoch (0) while the heap is filled below, so nothing is logged;
  // checkpoint() starts epoch 1
  journal_ = force_journal < 0 ? J >= kJournalJobs : force_journal != 0;
  if (journal_) { qstamp_.assign(J ? J : 1, 0); jstamp_.assign(J ? J : 1, 0); }
  stamp_.assign(J ? J : 1, 0);
  epoch_ = 0;
  depth_ = 1;
  fr_[0].epoch = 0;
  // pendingTasks[job] (allocate.go:110-123): Pending tasks whose Resreq is not empty, in TaskOrderFn order.  A job's list is built the
  // first time the job is popped (build_pending): a job's tasks are adjacent in the snapshot, so its list lives in pend_ at the job's own
  // task range, and nothing about it depends on what the action has done so far (statuses only change when the action closes).  The host
  // builds them while the device works on the window in front; a cycle that ends on exhausted capacity never builds most of them.
  if (pend_.size() < hs->T) pend_.resize(hs->T);
  pend_end_.assign(J ? J : 1, 0);
  pend_built_.assign(J ? J : 1, 0);
  cursor_.assign(J, 0);
  for (uint32_t j = 0; j < J; j++) cursor_[j] = hs->job_begin[j];
  // allocate.go:50-65: ssn.Jobs in ascending JobID; one queue-heap entry per job
  for (uint32_t j = 0; j < J; j++) {
    uint32_t q = hs->job_queue[j];
    if (q >= Q) continue;   // "queue not found": job skipped
    qpush(q);
  }
  inner_ = false;
  cur_q_ = cur_j_ = -1;
  cur_t_ = KB_NONE;
  checkpoint();
}

// allocate.go:50-65 walks ssn.Jobs in ascending JobID and pushes each job into its queue's heap: per queue, its jobs in ascending order
void OrderMachine::build_jobs(uint32_t q) {
  jheap_n_[q] = 0;
  for (uint32_t k = jheap_off_[q]; k < jheap_off_[q + 1]; k++) jpush(q, qjobs_[k]);
}

// session_plugins.go:298-331 TaskOrderFn: priority plugin, then pod creation time, then UID; the comparator is a strict total order over
// immutable keys, so the heap's pop order (allocate.go:110-123 pushes into a PriorityQueue) is the sorted order
void OrderMachine::build_pending(uint32_t j) {
  const HostSession *hs = hs_;
  const uint32_t b = hs->job_begin[j];
  uint32_t n = b;
  for (uint32_t t = b; t < hs->job_begin[j + 1]; t++)
    if (hs->t_status[t] == KB_TASK_PENDING && !hs->t_res_empty[t]) pend_[n++] = t;
  const bool by_prio = pol_->task_order_priority;
  std::sort(pend_.begin() + b, pend_.begin() + n, [hs, by_prio](uint32_t l, uint32_t r) {
    if (by_prio && hs->t_prio[l] != hs->t_prio[r]) return hs->t_prio[l] > hs->t_prio[r];
    if (hs->t_creation[l] != hs->t_creation[r]) return hs->t_creation[l] < hs->t_creation[r];
    return l < r;
  });
  pend_end_[j] = n;
  pend_built_[j] = 1;
}

void OrderMachine::arm(Frame &f) {
  f.epoch = ++epoch_;
  f.jobs.clear(); f.cursor.clear(); f.ready.clear(); f.vals.clear();
  f.qlog.clear(); f.jlog.clear();
  if (!journal_) { f.qheap.assign(qheap_.begin(), qheap_.begin() + qn_); f.jheap_items = jheap_items_; }
  f.qn = qn_;
  f.jheap_n = jheap_n_;
  f.qalloc = qalloc;
  f.qshare = qshare;
  f.cur_q = cur_q_; f.cur_j = cur_j_; f.cur_t = cur_t_; f.inner = inner_; f.steps = steps;
}

void OrderMachine::undo(Frame &f) {
  const int R = hs_->R;
  for (size_t k = f.jobs.size(); k-- > 0;) {   // newest first: a job logged twice ends with its oldest value
    const uint32_t j = f.jobs[k];
    cursor_[j] = f.cursor[k];
    ready[j] = f.ready[k];
    const double *v = &f.vals[k * (size_t)(R + 1)];
    for (int d = 0; d < R; d++) jalloc[(size_t)j * R + d] = v[d];
    jshare[j] = v[R];
  }
  if (journal_) {
    for (const uint64_t w : f.qlog) qheap_[(uint32_t)(w >> 32)] = (uint32_t)w;        // (a frame holds a slot once: any order)
    for (const uint64_t w : f.jlog) jheap_items_[(uint32_t)(w >> 32)] = (uint32_t)w;
  } else {
    std::copy(f.qheap.begin(), f.qheap.end(), qheap_.begin());   // (slots behind the size then: never read before they are written)
    jheap_items_ = f.jheap_items;
  }
  qn_ = f.qn;
  jheap_n_ = f.jheap_n;
  qalloc = f.qalloc;
  qshare = f.qshare;
  cur_q_ = f.cur_q; cur_j_ = f.cur_j; cur_t_ = f.cur_t; inner_ = f.inner; steps = f.steps;
}

void OrderMachine::checkpoint() {
  depth_ = 1;
  arm(fr_[0]);
}

void OrderMachine::push_checkpoint() {
  if (depth_ >= kFrames) throw std::logic_error("order machine: too many roll-back points");
  depth_++;
  arm(fr_[depth_ - 1]);
}

void OrderMachine::pop_commit() {
  if (depth_ >= 2) {   // the oldest frame goes (its storage moves to the end for the next push), the others move down
    std::rotate(fr_, fr_ + 1, fr_ + depth_);
    depth_--;
  }
}

void OrderMachine::rollback() {
  for (int k = depth_; k-- > 0;) undo(fr_[k]);   // newest first
  checkpoint();   // the restored state is the new roll-back point (fresh journal)
}

bool OrderMachine::next(uint32_t &task) {
  for (;;) {
    if (inner_) {
      uint32_t j = (uint32_t)cur_j_;
      if (!pend_built_[j]) build_pending(j);
      if (cursor_[j] < pend_end_[j]) {       // allocate.go:129-130
        touch(j);
        task = cur_t_ = pend_[cursor_[j]++];
        steps++;
        return true;
      }
      inner_ = false;
      qpush((uint32_t)cur_q_);               // allocate.go:192
    }
    if (qn_ == 0) return false;              // allocate.go:90-92
    uint32_t q = qpop();
    if (overused(q)) continue;               // allocate.go:95-98
    if (jheap_n_[q] == kNotBuilt) build_jobs(q);
    if (jheap_n_[q] == 0) continue;          // allocate.go:104-107
    uint32_t j = jpop(q);
    cur_q_ = (int)q;
    cur_j_ = (int)j;
    inner_ = true;
  }
}

// drf.go:135-145 and proportion.go:212-223 AllocateFunc (fired by ssn.Allocate and ssn.Pipeline alike)
void OrderMachine::update_shares(uint32_t j, uint32_t t) {
  const int R = hs_->R;
  const double *tr = &hs_->t_res_rows[(size_t)t * R];   // absent scalar keys hold 0.0
  const uint32_t tmask = hs_->t_resmask[t];
  if (pol_->has_drf) {
    double *a = &jalloc[(size_t)j * R];
    double share = 0;
    for (int d = 0; d < R; d++) {
      if (d < 2 || ((tmask >> (d - 2)) & 1u)) a[d] += tr[d];
      if (d >= 2 && !hs_->total.has(d)) continue;
      double s = helpers_share(a[d], hs_->total.get(d));
      if (s > share) share = s;
    }
    jshare[j] = share;
  }
  if (pol_->has_proportion) {
    uint32_t q = hs_->job_queue[j];
    double *a = &qalloc[(size_t)q * R];
    const Res &des = hs_->deserved[q];
    double share = 0;
    for (int d = 0; d < R; d++) {
      if (d < 2 || ((tmask >> (d - 2)) & 1u)) a[d] += tr[d];
      if (d >= 2 && !des.has(d)) continue;
      double s = helpers_share(a[d], des.get(d));
      if (s > share) share = s;
    }
    qshare[q] = share;
  }
}

void OrderMachine::report(Outcome o) {
  const uint32_t j = (uint32_t)cur_j_, q = (uint32_t)cur_q_;
  if (o == Outcome::NoFeasibleNode) {        // allocate.go:144-148: break; then queues.Push(queue)
    inner_ = false;
    qpush(q);
    return;
  }
  touch(j);
  if (o == Outcome::Allocated) ready[j] += 1;   // status Allocated counts towards ReadyTaskNum; Pipelined does not
  update_shares(j, cur_t_);
  if (job_ready(j) && cursor_[j] < pend_end_[j]) {   // allocate.go:185-188 (the current job: its list is built)
    jpush(q, j);
    inner_ = false;
    qpush(q);
  }
}

}  // namespace kb
