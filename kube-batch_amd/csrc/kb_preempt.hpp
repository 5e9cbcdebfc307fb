// kb_preempt.hpp — host side of the preempt action (actions/preempt/preempt.go:45-271) for the engine.
//
// The data-parallel part of preempt() — PredicateNodes with the plugin predicates only, PrioritizeNodes, SortNodes
// (preempt.go:183-193) — runs on the device: one K1 row (fit_mode 0) and one complete K3 list per distinct preemptor shape,
// valid until a Pipeline changes a node (evictions touch Idle / Releasing only, which neither the plugin predicates nor the
// scorers read).  Everything else is inherently sequential bookkeeping over small sets — the Statement journal
// (framework/statement.go:36-220), the tiered Preemptable intersection (framework/session_plugins.go:122-162 with the gang, drf,
// priority and conformance victim rules), the victim heap of one node — and stays on the host, in this file.
//
// This is the engine's own implementation, independent of oracle/kb_oracle.c (test infrastructure).
#pragma once
#include <functional>
#include <vector>

#include "kb_host.hpp"

namespace kb {

// one journal entry handed back through the C ABI (include/kb_engine.h: kb_stmt_op)
struct StmtOp {
  uint32_t op;     // KB_OP_*
  uint32_t task;
  uint32_t node;
  uint32_t stmt;   // running statement number inside the action
};

// live node state on the host while the action runs (Res: dense vector + scalar-key mask, mask == 0 <=> nil map)
struct LiveNodes {
  std::vector<Res> idle, rel;
  std::vector<long long> nzc, nzm, ac, am;
  std::vector<int32_t> podcnt, maxpods;
  std::vector<uint32_t> cls;
  std::vector<uint64_t> ports, base_ports;   // base: the share that belongs to pods outside the session
  std::vector<uint64_t> ports_x, base_ports_x;   // [N][HostSession::port_xw] the masks' words behind the first (empty: one word)
};

// the live kb_interpod counts while an evict action runs (device -> host when it starts, host -> device before every list and when it ends)
struct IpLive {
  std::vector<int32_t> ccnt;   // [C][D] allocated-status pods per counter and domain
  std::vector<int32_t> ctot;   // [C]
  std::vector<int32_t> punb;   // [P][NP] pods with an empty Spec.NodeName per priority class and node
  uint32_t z = KB_NONE;        // first node (ascending) holding a pod with an empty Spec.NodeName
  uint32_t NP = 0;
};

class PreemptMachine {
 public:
  // `lists(shape_task, out)`: nodes passing the plugin predicates for the shape of task `shape_task`, best first in SortNodes'
  // order (descending score, ties by DESCENDING node name: util/scheduler_helper.go:51-56,174-185), against the node state
  // the device currently holds.  `refresh(nodes)`: the device's copy of those nodes is brought up to date (cached lists die).
  using ListFn = std::function<void(uint32_t shape_task, std::vector<uint64_t> &out)>;   // entries: score << 32 | node, descending
  using RefreshFn = std::function<void(const std::vector<uint32_t> &nodes)>;

  void init(const HostSession *hs, const Policy *pol, LiveNodes *live, std::vector<uint8_t> *status, std::vector<uint32_t> *tnode,
            ListFn lists, RefreshFn refresh);
  // sessions with inter-pod terms: the live counts and how to put them on the device (call after init)
  void set_interpod(IpLive *ip, std::function<void()> upload);
  void run();           // the preempt action
  void off_node_tasks(std::vector<uint8_t> &off) const;   // after run(): what the next action's init() must know (HostSession::t_off_node)
  void run_reclaim();   // the reclaim action (actions/reclaim/reclaim.go:41-193): no Statement, ssn.Evict / ssn.Pipeline act at once

  std::vector<StmtOp> ops;             // every Evict / Pipeline / Commit / Discard, in order
  std::vector<uint32_t> evictions;     // committed evictions in the order stmt.Commit hands them to cache.Evict
  std::vector<uint8_t> counted;        // [T] task's Resreq is part of drf / proportion "allocated" (in: as of action start)
  std::vector<double> jalloc, jshare, qalloc, qshare;   // running drf / proportion aggregates (in: as of action start)
  std::vector<uint32_t> jmask, qmask;  // scalar-key masks of the allocated vectors
  std::vector<int32_t> cnt;            // [J][10] len(TaskStatusIndex[status])
  std::vector<uint8_t> node_status, on_node;   // node-side view of every task (api/node_info.go:186: the node keeps its own clone)
  std::vector<uint32_t> touched_nodes; // nodes whose state changed during the action (for the upload)
  uint64_t popped = 0, evals = 0;
  // where an action's time goes (KB_EVICT_TRACE): preemptors that walked nodes, nodes tried, the walks' time, preemptors skipped with their job
  uint64_t tr_walks = 0, tr_tries = 0, tr_skipped = 0, tr_pruned = 0, tr_shortcut = 0;
  double tr_walk_ms = 0.0, tr_setup_ms = 0.0, tr_scan_ms = 0.0;
  uint64_t tr_scan_nodes = 0;
  std::vector<uint32_t> scratch_pre_, scratch_victims_, scratch_ids_;
  std::vector<uint8_t> scratch_keep_;
  std::vector<Res> scratch_alloc_;

 private:
  const HostSession *hs_ = nullptr;
  const Policy *pol_ = nullptr;
  LiveNodes *nd_ = nullptr;
  std::vector<uint8_t> *status_ = nullptr;
  std::vector<uint32_t> *tnode_ = nullptr;
  ListFn lists_;
  RefreshFn refresh_;
  std::vector<std::vector<uint32_t>> ntasks_;   // node -> tasks in ni.Tasks, ascending task index
  size_t stmt_begin_ = 0;
  uint32_t stmt_no_ = 0;
  // per-shape cached lists + the nodes Pipelines changed since they were built
  std::vector<std::vector<uint64_t>> shape_list_;
  std::vector<uint8_t> shape_have_;
  std::vector<uint8_t> dirty_;
  std::vector<uint32_t> dirty_nodes_;
  std::vector<uint8_t> touched_;
  // pruning index: per (queue, node) the lowest job priority among the node's Running session tasks of that queue
  std::vector<std::vector<std::pair<uint32_t, int32_t>>> nq_minprio_;   // [node] -> (queue, lowest priority), sparse
  bool prio_prunes_ = false;
  std::vector<std::vector<uint32_t>> qnodes_;        // per queue: nodes that hold a Running session task of the queue (superset, fixed)
  // every Evict / Pipeline and their undo bumps version_; the last preemptor that found nothing, as of which version
  uint64_t version_ = 0, fail_version_ = ~0ull;
  uint32_t fail_task_ = KB_NONE;
  int fail_mode_ = -1;
  std::vector<std::vector<int32_t>> shape_rank_;     // per shape: position of every node in the shape's list (-1: not in it)

  // inter-pod (anti)affinity: an eviction takes its victim out of the predicate's pod list (Releasing is no allocated status: plugins/util/
  // util.go:37-60) while it stays in ni.Tasks; a Pipeline adds the preemptor to ni.Tasks with an empty Spec.NodeName (the priority's
  // "unbound" count, and Z); a discarded statement undoes both.  Lists are rebuilt from the device after every such change.
  IpLive *ip_ = nullptr;
  std::function<void()> ip_upload_;
  bool ip_changed_ = false;
  uint32_t ip_z0_ = KB_NONE;
  std::vector<int32_t> ip_unb_n_;   // [N] pods this action pipelined onto the node and that are still there
  void ip_allocated_status(uint32_t t, int joins);
  void ip_placed(uint32_t t, uint32_t n);
  void ip_unplaced(uint32_t t, uint32_t n);
  bool ip_predicate(uint32_t t, uint32_t n) const;

  Res task_res(uint32_t t) const;
  Res task_init(uint32_t t) const;
  int ready_num(uint32_t j) const;
  bool job_pipelined(uint32_t j) const;
  bool job_less(uint32_t l, uint32_t r) const;
  bool task_less(uint32_t l, uint32_t r) const;
  double drf_share(const double *alloc, uint32_t mask) const;
  void fire_allocate(uint32_t t);
  void fire_deallocate(uint32_t t);
  void set_status(uint32_t t, int st);
  void node_remove(uint32_t t);
  bool node_add(uint32_t t, uint32_t n, int st);
  void node_update(uint32_t t, int st);
  void touch_node(uint32_t n);
  void mark_dirty(uint32_t n);
  void recompute_minprio(uint32_t n);
  int32_t minprio(uint32_t q, uint32_t n) const;
  void evict(uint32_t t);
  void unevict(uint32_t t);
  void pipeline(uint32_t t, uint32_t n);
  void pipeline_session(uint32_t t, uint32_t n);
  void unpipeline(uint32_t t);
  void begin_stmt();
  void commit();
  void discard();
  size_t evictable(uint32_t preemptor, const std::vector<uint32_t> &pre, std::vector<uint32_t> &victims, bool reclaim = false);
  bool queue_less(uint32_t l, uint32_t r) const;
  bool overused(uint32_t q) const;
  bool host_eval(uint32_t t, uint32_t n, long long &score) const;
  bool preempt_one(uint32_t preemptor, int mode);
  bool preempt_walk(uint32_t preemptor, int mode);
  bool same_preemptor_class(uint32_t a, uint32_t b) const;
  bool needs_exact_list(uint32_t preemptor) const;
  bool try_node(uint32_t preemptor, int mode, uint32_t n);
};

}  // namespace kb
