// kb_host.hpp — host-side model of the session for the engine: api.Resource algebra with map-presence
// semantics, plugin OnSessionOpen state (drf totals, proportion deserved), and the order machine that
// reproduces the control flow of allocate.go:43-194 around the device rounds.
//
// This is the engine's own implementation (C++), independent of oracle/kb_oracle.c (test infrastructure).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/kb_engine.h"
#include "kb_res.hpp"

namespace kb {

struct EngineError : std::runtime_error {
  int code;
  EngineError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// ---- policy compiled from conf.Tier lists (framework/session_plugins.go dispatchers) ----
struct Policy {
  std::vector<uint8_t> job_chain;   // plugins contributing to JobOrderFn, tier-major order (session_plugins.go:243-267)
  bool queue_order_proportion = false;   // QueueOrderFn (session_plugins.go:270-295)
  bool task_order_priority = false;      // TaskOrderFn (session_plugins.go:298-331)
  bool gang_job_ready = false;           // JobReady (session_plugins.go:182-200 + gang.go:122-125)
  bool has_gang = false, has_drf = false, has_proportion = false;
  bool pred_enabled = false, nodeorder_enabled = false;
  int wL = 1, wM = 0, wNA = 1, wPA = 1, wB = 1;   // nodeorder.go:111-117
  // Preemptable (session_plugins.go:122-162): per tier the plugins registered with EnabledPreemptable that own a victim rule
  std::vector<std::vector<uint8_t>> preempt_tiers;
  // Reclaimable (session_plugins.go:80-119): the same with EnabledReclaimable (conformance, gang, proportion own a rule)
  std::vector<std::vector<uint8_t>> reclaim_tiers;
  bool gang_job_pipelined = false;       // JobPipelined (session_plugins.go:202-222 + gang.go:126-129)
};

// ---- host mirror of the session ----
struct HostSession {
  int R = 2;
  uint32_t N = 0, T = 0, J = 0, Q = 0;
  std::vector<double> t_res, t_init;      // [R][T]
  std::vector<double> t_res_rows;         // [T][R] Resreq again, task-major (host order machine)
  std::vector<uint32_t> t_resmask, t_job, t_cls, t_node;
  std::vector<int32_t> t_prio;
  std::vector<int64_t> t_creation;
  std::vector<uint8_t> t_status;
  // [T] (empty: none) the task carries a NodeName but is NOT in that node's Tasks: un-pipelined by a discarded preempt statement, or
  // pipelined while the stale name made AddTask fail (api/node_info.go:217-243 never clears NodeName; statement.go:113-150 only logs
  // the AddTask error).  Status and NodeName alone cannot tell these from tasks that sit on their node, so it is kept between actions.
  std::vector<uint8_t> t_off_node;
  std::vector<uint8_t> t_res_empty;       // Resreq.IsEmpty()   (allocate.go:114)
  std::vector<uint8_t> t_init_empty;      // InitResreq.IsEmpty() (backfill.go:47)
  std::vector<uint32_t> init_empty_tasks; // the tasks with t_init_empty set, ascending (a request never changes inside a session)
  std::vector<uint32_t> t_feas_shape;     // id of (InitResreq, class): tasks sharing it share a feasibility row
  std::vector<uint32_t> t_row_shape;      // id of (InitResreq, non-zero request, class): identical matrix rows
  uint32_t n_feas_shapes = 0, n_row_shapes = 0;
  std::vector<double> feas_eff;           // [n_feas_shapes][R] the InitResreq values LessEqual compares (0 where the dimension is skipped)
  std::vector<uint32_t> feas_rep;         // [n_feas_shapes] one task of the shape (the feasibility probe evaluates it)
  std::vector<uint32_t> feas_cls;         // [n_feas_shapes] static-predicate class
  std::vector<uint64_t> feas_conf;        // [n_feas_shapes] host-port conflict mask
  std::vector<uint32_t> job_begin, job_queue;
  std::vector<int32_t> job_min, job_prio;
  std::vector<int64_t> job_creation;
  std::vector<int32_t> queue_weight;
  std::vector<int64_t> queue_creation;
  // static data the preempt action reads on the host (the allocate / backfill path has it on the device only)
  std::vector<int64_t> t_nzc, t_nzm;       // pod non-zero request
  std::vector<uint64_t> t_want, t_conf;    // host ports the pod occupies / that conflict with it — word 0 of the masks (empty: no host ports)
  // kb_snapshot.port_words > 1: the words behind the first, [T][port_xw].  A task with a bit in any of them (t_wide) is decided in a
  // round of its own on the plain path: K1 evaluates the full mask (kb_k1.hpp), the commit kernels keep to word 0, and the host ORs the
  // pod's high words into the node's after the round (ActionRun::absorb).  Pods that stay inside word 0 never read or change the others.
  uint32_t port_xw = 0;
  std::vector<uint64_t> t_want_x, t_conf_x;
  std::vector<uint8_t> t_wide;             // [T] (empty: port_xw == 0)
  bool wide(uint32_t t) const { return !t_wide.empty() && t_wide[t]; }
  std::vector<uint8_t> t_protected;        // conformance: never a victim (empty: none)
  std::vector<int64_t> n_ac, n_am;         // nodeinfo.allocatableResource
  std::vector<int32_t> n_maxpods;
  std::vector<uint32_t> n_cls, n_idle_mask;   // static class; scalar keys of Idle / Allocatable
  std::vector<uint8_t> compat;             // class x class bit table (empty: every pair compatible)
  uint32_t n_tc = 0, n_nc = 1;
  bool whole = true;                       // every request / Idle / Releasing value is a whole number below 2^47 (KbDev::whole)
  bool has_affinity = false;               // some class carries preferred node-affinity terms (NormalizeReduce)
  std::vector<uint8_t> cls_has_aff;        // [n_tc] the task class has a non-zero preferred node-affinity count (empty: has_affinity is false)
  // inter-pod (anti)affinity (kb_interpod): a SUBJECT task (predicate checks or priority weights) is planned as the first row of its
  // window; a shape whose affinity REQUIRES a positive count can become feasible again (counts only grow): never marked dead
  bool has_interpod = false;
  std::vector<uint8_t> t_ip_subject;       // [T] (empty: no inter-pod affinity in the session)
  std::vector<uint8_t> t_ip_checks;        // [T] the task has predicate checks (forbid bits or a required counter)
  // the kb_interpod tables the evict actions need on the host (what a task joins and checks, the counters' node -> domain maps): the
  // live counts themselves come back from the device when such an action starts (kb_engine.cpp: run_evict_action)
  uint32_t ip_C = 0, ip_D = 1, ip_P = 0, ip_Wc = 1, ip_Wp = 1;
  std::vector<uint64_t> ip_task_inc, ip_task_forbid;   // [T][ip_Wc]
  std::vector<uint64_t> ip_task_cls_inc;               // [T][ip_Wp]
  std::vector<uint16_t> ip_task_require;               // [T]
  std::vector<uint8_t> ip_task_self;                   // [T]
  std::vector<uint32_t> ip_ctr_dom;                    // [ip_C][N]
  std::vector<uint8_t> feas_ip_require;    // [n_feas_shapes]
  std::vector<uint32_t> feas_ip;           // [n_feas_shapes] id of the shape's (forbid, require, self) triple: dominance needs equality
  // plugin state
  Res total;                               // drf.totalResource == proportion.totalResource
  std::vector<Res> deserved;               // [Q] proportion queueOpts[q].deserved
  std::vector<uint8_t> queue_has_attr;     // queue has a job in the session (proportion.go:69-83)
  // proportion only calls updateShare inside its water-fill loop (proportion.go:101-154) and in its event handlers
  // (:212-235).  When the loop's first pass finds total weight 0 it breaks before any updateShare, so every queue's share
  // stays at its zero value — whatever it has allocated — until an Allocate / Pipeline event touches that queue.
  uint8_t queue_share_at_open = 1;         // the water-fill loop ran at least one pass
  std::vector<Res> queue_request;          // [Q] attr.request (proportion.go:75-83): what the water-fill fills towards
  bool waterfill_on_device = false;        // set by the engine before build_host_session: the loop is left to kb_launch_waterfill
  std::vector<uint8_t> queue_share_live;   // [Q] updateShare has run for the queue (at open or through an event)
  // live aggregates (refreshed from the device share reduction after every action)
  std::vector<double> job_alloc;           // [J][R]
  std::vector<double> job_share;           // [J]
  std::vector<double> queue_alloc;         // [Q][R]
  std::vector<double> queue_share;         // [Q]
  std::vector<int32_t> job_ready;          // [J] ReadyTaskNum
};

// kb_session.cpp: the host half of kb_engine_create / kb_session_load (no device code; also built into the CPU test harnesses)
Policy compile_policy(const kb_config *cfg);
void build_host_session(const kb_snapshot *sn, const Policy &pol, uint32_t NP, HostSession &hs, std::vector<uint32_t> &t_active, std::vector<uint32_t> &nmask);

enum class Outcome { Allocated, Pipelined, NoFeasibleNode };

// The allocate action's control flow (allocate.go:43-194) as a resumable machine.  next() runs the reference loop up
// to the point where PredicateNodes would be called for a task and returns that task; report() feeds back what the
// device decided and runs the rest of the iteration.  checkpoint() / rollback() bracket a speculated round: the small
// state (heaps, queue aggregates) is copied, the per-job state (allocated vector, share, ready count, task cursor) is
// journalled on first touch, so a round costs O(window), not O(jobs x resources).
class OrderMachine {
 public:
  void init_allocate(const HostSession *hs, const Policy *pol);
  bool next(uint32_t &task);
  void report(Outcome o);
  void checkpoint();        // roll-back point = now (drops every earlier one)
  void push_checkpoint();   // another roll-back point on top (the windows behind the one in flight are speculated behind their own: at most kFrames in all)
  void pop_commit();        // the oldest window is confirmed: its roll-back point goes, the newer ones stay
  void rollback();          // back to the OLDEST roll-back point, which stays armed
  static constexpr int kFrames = 3;   // the window in flight, the one queued behind it, the one planned behind that (run_action)
  // undo the effect of the last next(): the task it returned is handed out again by the following next()
  void rollback_last_pop() { cursor_[(uint32_t)cur_j_]--; steps--; }
  // running aggregates, compared with the device reduction after the action
  std::vector<double> jalloc, jshare, qalloc, qshare;
  std::vector<int32_t> ready;
  uint64_t steps = 0;
  // how the roll-back points keep the two heap arrays (one slot per job each): -1 = by the session's size (journals from kJournalJobs jobs
  // on, copies below), 0 / 1 = copies / journals (tests/host_harness/order_harness.cpp runs every case both ways)
  int force_journal = -1;
#ifndef KB_ORDER_JOURNAL_JOBS
#define KB_ORDER_JOURNAL_JOBS 32768   // (make EXTRA=-DKB_ORDER_JOURNAL_JOBS=... for an A/B build)
#endif
  static constexpr uint32_t kJournalJobs = KB_ORDER_JOURNAL_JOBS;

 private:
  const HostSession *hs_ = nullptr;
  const Policy *pol_ = nullptr;
  // the queue heap: a fixed array of one slot per job (allocate.go:50-52 pushes one entry per job, and nothing is pushed that was not popped) and
  // its size.  Writes go through qset / jset: the FIRST write to a slot behind a roll-back point logs the slot's old value in that point's frame
  std::vector<uint32_t> qheap_;
  uint32_t qn_ = 0;
  static constexpr uint32_t kNotBuilt = 0xFFFFFFFFu;   // jheap_n_: the queue has not been popped yet, its job heap is still to be built
  std::vector<uint32_t> jheap_items_, jheap_off_, jheap_n_, qjobs_;
  void build_jobs(uint32_t q);
  // per job: Pending non-BestEffort tasks in TaskOrderFn order, built on the job's first pop into the job's own task range of pend_
  std::vector<uint32_t> pend_, pend_end_, cursor_;
  std::vector<uint8_t> pend_built_;
  void build_pending(uint32_t j);
  int cur_q_ = -1, cur_j_ = -1;
  uint32_t cur_t_ = KB_NONE;
  bool inner_ = false;
  // roll-back points: copies of the small state (per queue) + first-touch journal of the per-job state, at most kFrames deep.  The two heap arrays
  // (one slot per job each) are COPIED into the frame in small sessions and JOURNALLED in large ones (journal_): at 100k jobs the copies were
  // 0.8 MB per roll-back point, one point per round, in front of every round's launches — the next round's matrix went out ~25 us later and its
  // lists were late for the commit launch (1M x 50k: 252 -> 242 ms); at 10k jobs the copy is 80 KB and cheaper than a stamp test on each of the
  // ~8k heap writes of a round (100k x 10k, R = 16: 57.8 against 58.6 ms with journals).
  struct Frame {
    std::vector<uint32_t> qheap, jheap_items, jheap_n;   // (qheap / jheap_items: copy mode)
    std::vector<uint64_t> qlog, jlog;   // journal mode: (slot << 32) | the slot's value when the frame was armed
    uint32_t qn = 0;
    std::vector<double> qalloc, qshare;
    int cur_q = -1, cur_j = -1;
    uint32_t cur_t = KB_NONE;
    bool inner = false;
    uint64_t steps = 0;
    std::vector<uint32_t> jobs, cursor;
    std::vector<int32_t> ready;
    std::vector<double> vals;   // per journalled job: allocated[R], share
    uint32_t epoch = 0;
  };
  Frame fr_[kFrames];
  int depth_ = 0;
  std::vector<uint32_t> stamp_, qstamp_, jstamp_;   // per job / queue-heap slot / job-heap slot: the epoch of the frame that holds its old value
  uint32_t epoch_ = 0;
  bool journal_ = false;
  void qset(uint32_t i, uint32_t v) {
    if (journal_) {
      Frame &f = fr_[depth_ - 1];
      if (qstamp_[i] != f.epoch) { qstamp_[i] = f.epoch; f.qlog.push_back(((uint64_t)i << 32) | qheap_[i]); }
    }
    qheap_[i] = v;
  }
  void jset(uint32_t i, uint32_t v) {   // i: index into jheap_items_ (the queue's offset included)
    if (journal_) {
      Frame &f = fr_[depth_ - 1];
      if (jstamp_[i] != f.epoch) { jstamp_[i] = f.epoch; f.jlog.push_back(((uint64_t)i << 32) | jheap_items_[i]); }
    }
    jheap_items_[i] = v;
  }
  void arm(Frame &f);
  void undo(Frame &f);
  void touch(uint32_t j) {
    Frame &f = fr_[depth_ - 1];
    if (stamp_[j] == f.epoch) return;
    stamp_[j] = f.epoch;
    f.jobs.push_back(j);
    f.cursor.push_back(cursor_[j]);
    f.ready.push_back(ready[j]);
    const int R = hs_->R;
    f.vals.insert(f.vals.end(), jalloc.begin() + (size_t)j * R, jalloc.begin() + (size_t)(j + 1) * R);
    f.vals.push_back(jshare[j]);
  }

  bool job_ready(uint32_t j) const { return pol_->gang_job_ready ? ready[j] >= hs_->job_min[j] : true; }
  bool queue_less(uint32_t l, uint32_t r) const;
  bool job_less(uint32_t l, uint32_t r) const;
  bool overused(uint32_t q) const;
  void qpush(uint32_t q);
  uint32_t qpop();
  void jpush(uint32_t q, uint32_t j);
  uint32_t jpop(uint32_t q);
  void update_shares(uint32_t j, uint32_t t);
};

}  // namespace kb
