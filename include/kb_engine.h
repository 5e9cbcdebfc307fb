/*
 * kb_engine.h — C ABI of the MI355X-native allocate/backfill engine for kube-batch.
 *
 * This is the drop-in boundary.  The Go side of kube-batch reaches the
 * device through these entry points only (cgo -> C ABI -> HIP); nothing here exposes a
 * torch / HIP / C++ type.  Every function returns 0 (KB_OK) or a negative KB_E_* code;
 * no exception or abort crosses the boundary; all buffers are caller-allocated and no
 * caller pointer is retained after a call returns (cgo pointer rules).
 *
 * What each entry point stands in for in the reference (paths relative to /root/reference):
 *
 *   kb_engine_create / kb_engine_destroy
 *       framework.OpenSession's plugin construction from conf.Tier / conf.PluginOption
 *       (pkg/scheduler/framework/framework.go:30-52, pkg/scheduler/conf/scheduler_conf.go:20-56)
 *       and the action registry lookup (pkg/scheduler/framework/plugins.go:58-72).
 *   kb_session_load
 *       cache.Snapshot() -> Session{Jobs,Nodes,Queues} (pkg/scheduler/cache/cache.go:627-683,
 *       pkg/scheduler/framework/session.go:63-115) plus every plugin's OnSessionOpen state:
 *       drf totals/shares (plugins/drf/drf.go:60-83), proportion deserved water-fill
 *       (plugins/proportion/proportion.go:58-154).  (gang's JobValid filter never fires at this commit: DESIGN.md §1.)
 *   kb_session_reset
 *       a second framework.OpenSession on an unchanged cache (same Snapshot(), same OnSessionOpen results).
 *   kb_run_allocate
 *       allocateAction.Execute (pkg/scheduler/actions/allocate/allocate.go:43-194) with
 *       util.PredicateNodes / PrioritizeNodes / SelectBestNode
 *       (pkg/scheduler/util/scheduler_helper.go:63-208) and Session.Allocate / Pipeline
 *       (pkg/scheduler/framework/session.go:194-288).
 *   kb_run_backfill
 *       backfillAction.Execute (pkg/scheduler/actions/backfill/backfill.go:40-71).
 *   kb_run_preempt
 *       preemptAction.Execute (pkg/scheduler/actions/preempt/preempt.go:45-168) with preempt() (:171-254: PredicateNodes with
 *       the plugin predicates only, PrioritizeNodes, util.SortNodes scheduler_helper.go:174-185, ssn.Preemptable
 *       framework/session_plugins.go:122-162, victims lowest TaskOrderFn first) and framework.Statement
 *       (framework/statement.go:36-220: Evict / Pipeline / Commit / Discard).
 *   kb_run_reclaim
 *       reclaimAction.Execute (pkg/scheduler/actions/reclaim/reclaim.go:41-193).
 *   kb_eval_matrix
 *       the per-(task,node) predicate closure (allocate.go:73-87), the predicates plugin
 *       (plugins/predicates/predicates.go:123-265) and the nodeorder scorers
 *       (plugins/nodeorder/nodeorder.go:140-168 -> vendor/k8s.io/kubernetes/pkg/scheduler/
 *       algorithm/priorities/{least_requested,most_requested,balanced_resource_allocation}.go)
 *       evaluated for a contiguous range of task rows against the session's live node state.
 *   kb_argmax_rows
 *       util.SelectBestNode / findMaxScores (scheduler_helper.go:188-208) with the canonical
 *       tie-break (first max in ascending node order), for a range of task rows.
 *   kb_get_binds
 *       the gang-gated dispatch of Session.Allocate (session.go:277-285 -> cache.Bind):
 *       which Allocated tasks were handed to the Binder.
 *   kb_get_shares
 *       drf jobOpts[*].share / proportion queueOpts[*].share after the run
 *       (drf.go:157-171, proportion.go:241-253).
 *
 * Canonical order: nodes ascending by name, queues ascending by QueueID,
 * jobs ascending by JobID ("ns/name"), tasks grouped by job and ascending by pod UID inside
 * a job.  All indices in this ABI are ranks in that order, so "UID string <" in the
 * reference's fall-back comparators is "index <" here.
 */
#ifndef KB_ENGINE_H
#define KB_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KB_ABI_VERSION 8u
#define KB_MAX_RES 32u          /* resource dimensions: 0 = cpu (milli), 1 = memory (bytes), 2.. = scalar resources (milli) */
#define KB_NONE 0xFFFFFFFFu

/* ---- error codes ---------------------------------------------------------------- */
#define KB_OK              0
#define KB_E_INVALID      -1    /* bad argument / malformed snapshot */
#define KB_E_UNSUPPORTED  -2    /* configuration outside the engine's exact envelope: caller falls back to the stock action */
#define KB_E_DEVICE       -3    /* HIP runtime error (message via kb_last_error) */
#define KB_E_NOMEM        -4
#define KB_E_STATE        -5    /* call order violated (e.g. run before load) */
#define KB_E_CAPACITY     -6    /* caller's output buffer too small; *n_out holds the required count */
#define KB_E_INTERNAL     -7    /* internal consistency check failed; no decisions emitted */

/* ---- task status (pkg/scheduler/api/types.go:27-61) ------------------------------ */
enum {
  KB_TASK_PENDING = 0, KB_TASK_ALLOCATED = 1, KB_TASK_PIPELINED = 2, KB_TASK_BINDING = 3,
  KB_TASK_BOUND = 4, KB_TASK_RUNNING = 5, KB_TASK_RELEASING = 6, KB_TASK_SUCCEEDED = 7,
  KB_TASK_FAILED = 8, KB_TASK_UNKNOWN = 9
};

/* ---- plugins (pkg/scheduler/plugins/factory.go:31-42) ---------------------------- */
enum {
  KB_PLUGIN_PRIORITY = 0, KB_PLUGIN_GANG = 1, KB_PLUGIN_CONFORMANCE = 2, KB_PLUGIN_DRF = 3,
  KB_PLUGIN_PREDICATES = 4, KB_PLUGIN_PROPORTION = 5, KB_PLUGIN_NODEORDER = 6
};

/* conf.PluginOption.Enabled* (pkg/scheduler/conf/scheduler_conf.go:33-56) */
#define KB_EN_JOB_ORDER     (1u << 0)
#define KB_EN_JOB_READY     (1u << 1)
#define KB_EN_JOB_PIPELINED (1u << 2)
#define KB_EN_TASK_ORDER    (1u << 3)
#define KB_EN_PREEMPTABLE   (1u << 4)
#define KB_EN_RECLAIMABLE   (1u << 5)
#define KB_EN_QUEUE_ORDER   (1u << 6)
#define KB_EN_PREDICATE     (1u << 7)
#define KB_EN_NODE_ORDER    (1u << 8)
#define KB_EN_ALL           0x1FFu   /* plugins.ApplyPluginConfDefaults (plugins/defaults.go:22-52) */

/* kb_plugin_option.args slots */
#define KB_ARG_NODEORDER_LEAST    0  /* leastrequested.weight   default 1 (nodeorder.go:111-117) */
#define KB_ARG_NODEORDER_MOST     1  /* mostrequested.weight    default 0 */
#define KB_ARG_NODEORDER_NODEAFF  2  /* nodeaffinity.weight     default 1 */
#define KB_ARG_NODEORDER_PODAFF   3  /* podaffinity.weight      default 1 */
#define KB_ARG_NODEORDER_BALANCED 4  /* balancedresource.weight default 1 */
/* The predicates plugin's optional pressure checks are static per (pod class, node class): the CALLER folds them into
   class_compat (flatten.go pressureArgs / snapshot.flatten(pressure=...)) and does not pass them; a config that sets one of
   these slots gets KB_E_UNSUPPORTED (the engine holds no node-condition state). */
#define KB_ARG_PRED_MEM_PRESSURE  0  /* predicate.MemoryPressureEnable (predicates.go:94-107) */
#define KB_ARG_PRED_DISK_PRESSURE 1
#define KB_ARG_PRED_PID_PRESSURE  2

typedef struct kb_plugin_option {
  uint32_t plugin;     /* KB_PLUGIN_* */
  uint32_t enabled;    /* KB_EN_* bitmask */
  int32_t  args[8];    /* plugin-specific, see KB_ARG_* ; unused slots 0 */
  uint32_t args_set;   /* bit i set <=> args[i] was given in the YAML (otherwise the default applies) */
} kb_plugin_option;

#define KB_FLAG_SYNC_ROUNDS 1u  /* disable host/device overlap (debug) */
/* While a round runs the calling thread POLLS the round's sequence word in pinned memory (no stream synchronisation per round: ~20 us of a
 * ~110 us round).  Default since round 6: a short spin (`pause`, KB_WAIT_SPIN_US microseconds — most answers arrive inside it, the host reaches
 * the wait with the round half over), then the thread gives the core up between two polls (sched_yield): inside cmd/kube-batch the informer
 * goroutines' threads (pkg/scheduler/cache) get the core while a round runs, the answer is noticed a few microseconds late at worst.
 * KB_FLAG_SPIN_WAIT: spin for the whole wait (round 5's default; one core busy for the length of the action).  KB_FLAG_YIELD_WAIT (round 5's
 * opt-in) is accepted and means the default. */
#define KB_FLAG_YIELD_WAIT 2u
#define KB_FLAG_SPIN_WAIT 4u
#define KB_WAIT_SPIN_US 20.0

typedef struct kb_config {
  uint32_t version;                /* KB_ABI_VERSION */
  uint32_t n_tiers;
  const uint32_t *tier_begin;      /* [n_tiers+1] offsets into plugins[] */
  const kb_plugin_option *plugins;
  int32_t  device;                 /* HIP device ordinal */
  uint32_t window;                 /* task rows per device round (<= 1024); 0 = engine default (256) */
  uint32_t commit_batch;           /* rows the commit kernel speculates per batch, 1..16 (doubled after a fully valid batch); 0 = default (16) */
  uint32_t flags;                  /* KB_FLAG_* */
} kb_config;

/*
 * Session snapshot, structure-of-arrays, caller-owned and read-only.
 * Matrix-shaped fields are dimension-major: x[d * n + i].
 */
/* Inter-pod (anti)affinity: the predicate of plugins/predicates/predicates.go:249-262 (predicates.NewPodAffinityPredicate,
   vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/predicates/predicates.go:1261-1575 on its metadata-free path, over the PodLister
   of plugins/util/util.go:37-90) and nodeorder's InterPodAffinityPriority (vendor/.../priorities/interpod_affinity.go:99-235 behind
   plugins/nodeorder/nodeorder.go:48-62,156-160).  The flattener resolves namespaces, label selectors and topology keys into:

   predicate COUNTERS (at most KB_INTERPOD_MAX).  Counter c maps every node to a domain id (ctr_dom; KB_NONE: the node lacks a topology label of
   the counter) and counts, per domain, the session's pods in an allocated status (api.AllocatedStatus) that
     - own a given required anti-affinity term                     ("existing pods' anti-affinity", predicates.go:1400-1441), or
     - match ALL terms of a given set of required terms            (a pod's own affinity / anti-affinity, :1474-1575: the slow path
                                                                    ANDs the terms of one pod).
   ctr_total counts them regardless of the domain.  Per task: task_inc = counters the pod joins when ssn.Allocate places it;
   task_forbid = counters that must be 0 in the candidate node's domain — bit masks of Wc = max(1, ceil(C / 64)) 64-bit words per
   task (counter c: bit c % 64 of word c / 64); task_require = the counter that must be positive there (0xFFFF: none) unless
   ctr_total of it is 0 and task_self is set (the first-pod rule, :1550-1565).

   priority CLASSES (at most KB_INTERPOD_MAX; task_cls_inc has Wp = max(1, ceil(P / 64)) words per task).  Class p maps every node to the id of its value of ONE topology key (cls_dom) and counts, per node,
   the pods in ni.Tasks that the class covers (a preferred term of the scored pod: the pods matching it; a term some pod owns -
   required affinity with hardPodAffinityWeight 1, preferred (anti)affinity with its signed weight: its owners): cls_bound for pods
   whose Spec.NodeName is set, cls_unbound for pods whose Spec.NodeName is still empty - those are looked up through nodeorder's
   cachedNodeInfo and all resolve to ONE node, the first in ascending name order that holds any such pod (first_unbound_node as of
   session open; every in-session placement creates such a pod).  task_cls_inc = classes the pod joins when placed (Allocate or
   Pipeline: both AddTask); task_sig = row of sig_weight with the pod's signed weight per class (KB_NONE: all zero).
     count(i) = sum_p weight_p * ( sum over FEASIBLE nodes n with cls_dom[p][n] == cls_dom[p][i] of cls_bound[p][n]
                                   + [cls_dom[p][i] == cls_dom[p][Z]] * sum over feasible n of cls_unbound[p][n] )
     score(i) = int(10 * (count(i) - min) / (max - min)) * podaffinity.weight, min / max over the feasible nodes and 0
   (interpod_affinity.go:213-233; only the pods of the feasible nodes are seen: util/scheduler_helper.go:226-238). */
#define KB_INTERPOD_MAX 65534u   /* the width of task_require (uint16, 0xFFFF = none); the tables themselves are multi-word / dense: no other limit */
typedef struct kb_interpod {
  uint32_t n_counters;           /* C <= KB_INTERPOD_MAX */
  uint32_t n_domains;            /* D: every domain id of every counter is < D */
  uint32_t n_classes;            /* P <= KB_INTERPOD_MAX */
  uint32_t n_sigs;               /* S rows of sig_weight */
  uint32_t first_unbound_node;   /* Z at session open, KB_NONE: no pod with an empty Spec.NodeName sits on a node */
  uint32_t pad;
  const uint32_t *ctr_dom;       /* [C][N] */
  const int32_t  *ctr_count;     /* [C][D] */
  const int32_t  *ctr_total;     /* [C] */
  const uint64_t *task_inc;      /* [T][Wc] */
  const uint64_t *task_forbid;   /* [T][Wc] */
  const uint16_t *task_require;  /* [T] */
  const uint8_t  *task_self;     /* [T] */
  const uint32_t *cls_dom;       /* [P][N] */
  const int32_t  *cls_bound;     /* [P][N] */
  const int32_t  *cls_unbound;   /* [P][N] */
  const uint64_t *task_cls_inc;  /* [T][Wp] */
  const uint32_t *task_sig;      /* [T] */
  const int32_t  *sig_weight;    /* [S][P] */
} kb_interpod;

typedef struct kb_snapshot {
  uint32_t version;                /* KB_ABI_VERSION */
  uint32_t n_res;                  /* R, 2..KB_MAX_RES */
  uint32_t n_nodes, n_tasks, n_jobs, n_queues;
  uint32_t n_task_classes, n_node_classes;

  /* nodes: api.NodeInfo (pkg/scheduler/api/node_info.go:28-47) */
  const double   *node_idle;          /* [R][N]  NodeInfo.Idle      */
  const double   *node_releasing;     /* [R][N]  NodeInfo.Releasing */
  const double   *node_allocatable;   /* [R][N]  NodeInfo.Allocatable (kube-batch float64 view) */
  const uint32_t *node_scalar_mask;   /* [N] bit (d-2) set <=> scalar key d exists in Allocatable.ScalarResources */
  const int64_t  *node_alloc_cpu;     /* [N] k8s nodeinfo.allocatableResource.MilliCPU (vendor/.../nodeinfo/node_info.go:625-628) */
  const int64_t  *node_alloc_mem;     /* [N] ... .Memory */
  const int64_t  *node_nz_cpu;        /* [N] nodeinfo.nonzeroRequest.MilliCPU over every pod in ni.Tasks (node_info.go:502-517) */
  const int64_t  *node_nz_mem;        /* [N] ... .Memory */
  const int32_t  *node_max_pods;      /* [N] Allocatable.MaxTaskNum (api/resource_info.go:81-82) */
  const int32_t  *node_pod_cnt;       /* [N] len(ni.Tasks) */
  const uint32_t *node_class;         /* [N] static-predicate class (labels/taints/conditions flattened by the caller) */

  /* tasks: api.TaskInfo (pkg/scheduler/api/job_info.go:36-54) */
  const double   *task_resreq;        /* [R][T] TaskInfo.Resreq     */
  const double   *task_init_resreq;   /* [R][T] TaskInfo.InitResreq */
  const uint32_t *task_scalar_mask;   /* [T] scalar keys present in Resreq.ScalarResources */
  const int64_t  *task_nz_cpu;        /* [T] sum over containers of GetNonzeroRequests cpu (vendor/.../priorities/util/non_zero.go:48-61) */
  const int64_t  *task_nz_mem;        /* [T] */
  const uint32_t *task_job;           /* [T] job index */
  const uint32_t *task_class;         /* [T] static-predicate class */
  const int32_t  *task_priority;      /* [T] TaskInfo.Priority */
  const int64_t  *task_creation;      /* [T] pod CreationTimestamp (seconds) */
  const uint8_t  *task_status;        /* [T] KB_TASK_* */
  const uint32_t *task_node;          /* [T] node index for placed tasks, KB_NONE otherwise.  Set <=> the task is in that node's
                                         ni.Tasks: a task left with a NodeName but on no node (by a discarded or failed Statement
                                         step of an earlier action) cannot be expressed; the flatteners refuse such a session */

  /* jobs: api.JobInfo (job_info.go:127-154); tasks of job j are [job_task_begin[j], job_task_begin[j+1]) */
  const uint32_t *job_task_begin;     /* [J+1] */
  const uint32_t *job_queue;          /* [J] queue index */
  const int32_t  *job_min_available;  /* [J] */
  const int32_t  *job_priority;       /* [J] */
  const int64_t  *job_creation;       /* [J] PodGroup CreationTimestamp (seconds) */

  /* queues: api.QueueInfo (api/queue_info.go:74-93) */
  const int32_t  *queue_weight;       /* [Q] */
  const int64_t  *queue_creation;     /* [Q] */

  /* static predicates p2..p7 folded to a class x class bit table:
     bit (tc * n_node_classes + nc) of class_compat; NULL => every pair compatible */
  const uint8_t  *class_compat;

  /* preferred node affinity (nodeorder's NodeAffinity priority, vendor/.../priorities/node_affinity.go:34-77): the Map
     step's count for (task class tc, node class nc) = sum of the weights of the pod's preferred scheduling terms whose
     match expressions select the node's labels; [n_task_classes][n_node_classes], NULL => no pod has preferred terms.
     The engine applies NormalizeReduce(10) over the task's feasible nodes (reduce.go:28-63) and the plugin weight. */
  const int32_t  *class_affinity;

  /* host ports (predicates.PodFitsHostPorts, vendor/.../algorithm/predicates/predicates.go:1153-1175 over
     nodeinfo.HostPortInfo, vendor/.../nodeinfo/host_ports.go:107-135): the caller interns every distinct
     (hostIP, protocol, hostPort) of the session's pods that can conflict with a port of a Pending pod (the only pods the
     predicate is asked for; the others' ports can never decide anything) into a bit: triple i is bit i % 64 of word i / 64 of a
     mask of Wh = max(1, port_words) 64-bit words (port_words, at the end of this struct; any number of triples).
     node_ports[n] = bits used by the pods in ni.Tasks; task_port_want[t] = bits the pod occupies once placed;
     task_port_conflict[t] = every bit that conflicts with one of the pod's ports (same protocol and port, and equal IPs or
     either side 0.0.0.0).  A node fails the predicate iff node_ports & task_port_conflict != 0 in any word; placing the pod ORs
     task_port_want in.  All three NULL => no host ports.  Word 0 is the fast one: a Pending pod whose masks reach into the
     words behind it is decided in a device round of its own, so the flattener hands the low bits to the
     triples the Pending pods name most often (kube-batch_amd/snapshot.py, integration/go/gpuallocate/flatten.go). */
  const uint64_t *node_ports;          /* [N][Wh] */
  const uint64_t *task_port_want;      /* [T][Wh] */
  const uint64_t *task_port_conflict;  /* [T][Wh] */

  /* conformance plugin (plugins/conformance/conformance.go:44-58): 1 = the pod may not be evicted (kube-system namespace or a
     system-cluster-critical / system-node-critical priority class).  Read by kb_run_preempt; NULL => no pod is protected. */
  const uint8_t  *task_evict_protected; /* [T] */

  /* inter-pod (anti)affinity tables; NULL => no pod of the cluster carries a podAffinity / podAntiAffinity term */
  const kb_interpod *interpod;

  uint32_t port_words;             /* Wh: 64-bit words per host-port mask; 0 reads as 1 */
  uint32_t pad;
} kb_snapshot;

/* one placement decision, in the order the reference loop would have made it */
typedef struct kb_decision {
  uint32_t task;
  uint32_t node;
  uint32_t kind;    /* 0 = ssn.Allocate, 1 = ssn.Pipeline */
  uint32_t round;   /* device round that produced it (diagnostic) */
} kb_decision;

/* one entry of the preempt action's journal: what the reference does through framework.Statement, in order.  The Go action
   replays it: a new ssn.Statement() whenever `stmt` changes, stmt.Evict(node.Tasks[...].Clone(), "preempt") / stmt.Pipeline(task,
   node) per entry, stmt.Commit() / stmt.Discard() at the markers (a discarded statement is replayed too: its Pipeline leaves the
   sticky NodeName behind, framework/statement.go:155-190 + api/node_info.go:217-243). */
enum { KB_OP_EVICT = 0, KB_OP_PIPELINE = 1, KB_OP_COMMIT = 2, KB_OP_DISCARD = 3 };
typedef struct kb_stmt_op {
  uint32_t op;      /* KB_OP_* */
  uint32_t task;    /* KB_NONE for the markers */
  uint32_t node;    /* EVICT: the node the victim runs on; PIPELINE: the node the preemptor waits for */
  uint32_t stmt;    /* statement number inside the action (1-based) */
} kb_stmt_op;

typedef struct kb_stats {
  uint64_t evals;             /* (task,node) evaluations the reference algorithm performs for the work done so far: sum over popped tasks of N */
  uint64_t tasks_popped;      /* tasks that went through PredicateNodes */
  uint64_t decisions;         /* Allocate + Pipeline calls */
  uint64_t binds;             /* tasks dispatched to the Binder */
  uint64_t rounds;            /* device rounds (matrix -> arg-max -> commit) */
  uint64_t spec_breaks;       /* rounds cut short because the speculated order diverged */
  uint64_t row_fallbacks;     /* rows won by a node the round had already changed (a dirty slot): committed by shots, single rows one by one (the name is round 3's) */
  uint64_t matrix_launches;   /* launches of the mask+score matrix kernel */
  uint64_t matrix_evals;      /* (task,node) pairs those launches evaluated */
  double   matrix_ms;         /* HIP-event time of those launches on the engine stream */
  double   argmax_ms;         /* segmented arg-max kernel */
  double   commit_ms;         /* sequential commit kernel */
  double   reduce_ms;         /* gang ballot + share reduction kernel */
  double   host_order_ms;     /* host time in the order machine (queue/job/task ordering) */
  double   total_ms;          /* wall time of the run_* calls */
  uint64_t rounds_select;     /* of `rounds`: committed by the selection kernel (k_commit_select); the others by the batch or the run kernel */
  uint64_t select_runs_clean;   /* selection kernel, runs of >= 2 plain rows: every pick a clean candidate's first placement */
  uint64_t select_runs_shots;   /* ... committed by shots: bounded tables of the contenders' next keys, one rank, the final picks (kb_commit_sel.hip) */
  uint64_t select_shots;        /* ... the shots those runs took (a table that ends while its sequence may go on costs another one) */
} kb_stats;

typedef struct kb_engine kb_engine;

int  kb_engine_create(const kb_config *cfg, kb_engine **out);
void kb_engine_destroy(kb_engine *e);
const char *kb_last_error(const kb_engine *e);   /* valid until the next call on e; e == NULL -> creation error */

int  kb_session_load(kb_engine *e, const kb_snapshot *snap);
/* restore the loaded session to its just-loaded state from the pristine copy kept in HBM (device-to-device);
   the snapshot is not read again.  Used to run the same cycle repeatedly (bench steps) without a host upload.
   The copies are queued on the engine's stream and the call returns without waiting for them: the actions that follow are
   ordered behind them on that stream, and every other entry point that looks at the state (the getters, the evict actions,
   kb_engine_use_stream, kb_session_load) waits first.  The drf / proportion / gang aggregates come back from the host copies
   made when the session was loaded (the restored state is bit for bit the one that reduction saw). */
int  kb_session_reset(kb_engine *e);

int  kb_run_allocate(kb_engine *e, kb_decision *out, uint64_t cap, uint64_t *n_out);
int  kb_run_backfill(kb_engine *e, kb_decision *out, uint64_t cap, uint64_t *n_out);
/* the preempt action on the session's current state; journal entries in order.  ANY non-OK answer of kb_run_preempt / kb_run_reclaim after
   the action has started (KB_E_CAPACITY: *n_out = required count, no result was applied; KB_E_UNSUPPORTED met mid-action; KB_E_INTERNAL from
   the closing cross-check) may leave refreshed nodes or committed state behind: the session is then marked and every kb_run_* answers
   KB_E_STATE until kb_session_load or kb_session_reset; the getters keep working.
   Sessions with preferred node-affinity terms are scored with NormalizeReduce over the preemptor's feasible set, which one Pipeline can
   change for every node: their lists are rebuilt after every Pipeline instead of repaired (KB_PREEMPT_NODE_AFFINITY=0 in the environment
   restores the round-2 refusal, KB_E_UNSUPPORTED).  Sessions with inter-pod (anti)affinity terms: the evict machine keeps the kb_interpod
   counts current on the host and rebuilds every list after a change (tests/test_gpu_interpod.py).  KB_E_UNSUPPORTED: states in which the reference itself would panic / abort
   (Resource.Sub underflow, NodeInfo.UpdateTask). */
int  kb_run_preempt(kb_engine *e, kb_stmt_op *out, uint64_t cap, uint64_t *n_out);
/* the reclaim action (pkg/scheduler/actions/reclaim/reclaim.go:41-193; victims through ssn.Reclaimable, framework/session_plugins.go:
   80-119, with proportion's rule plugins/proportion/proportion.go:171-196).  There is no Statement: every EVICT entry is an
   ssn.Evict (framework/session.go:317-354), every PIPELINE an ssn.Pipeline, in order; `stmt` is 0.  The walk is first-fit over
   nodes in name order with the plugin predicates only: sequential bookkeeping, it runs on the host side of the engine. */
int  kb_run_reclaim(kb_engine *e, kb_stmt_op *out, uint64_t cap, uint64_t *n_out);
/* tasks the committed statements handed to cache.Evict so far, in that order */
int  kb_get_evictions(kb_engine *e, uint32_t *out, uint64_t cap, uint64_t *n_out);

/* rows [t0,t1) of the task x node matrix against the session's current node state.
   mask_bits: (t1-t0) rows of ceil(N/8) bytes, bit (n & 7) of byte n >> 3; score: (t1-t0) x N uint16.
   fit_mode: 1 = allocate's predicate (resource fit + plugin predicates), 0 = plugin predicates only (backfill/preempt),
   optionally OR'ed with KB_MATRIX_DIRECT / KB_MATRIX_NO_DEDUP (how the launch is organised, never what it computes). */
#define KB_MATRIX_DIRECT   0x100u   /* every task row is evaluated by the matrix kernel itself (no per-shape rows + row expansion); a row equal
                                       to its predecessor (the tasks of a job) re-stores the predecessor's result */
#define KB_MATRIX_NO_DEDUP 0x200u   /* with KB_MATRIX_DIRECT: not even adjacent equal rows share an evaluation: rows x N evaluations,
                                       the evaluator's own rate (bench.py: roofline_eval_all_rows) */
int  kb_eval_matrix(kb_engine *e, uint32_t t0, uint32_t t1, uint32_t fit_mode, uint8_t *mask_bits, uint16_t *score);

/* per row of [t0,t1): the k best feasible nodes, descending score then ascending node index;
   out_node[(t-t0)*k + i] = node index or KB_NONE, out_score likewise (0 where none). */
int  kb_argmax_rows(kb_engine *e, uint32_t t0, uint32_t t1, uint32_t fit_mode, uint32_t k,
                    uint32_t *out_node, uint16_t *out_score);

/* device-resident timing of the matrix kernel for the roofline figure: reps launches over rows [t0,t1),
   nothing copied back; *ms_avg = average launch duration measured with HIP events on the engine stream. */
int  kb_bench_matrix(kb_engine *e, uint32_t t0, uint32_t t1, uint32_t fit_mode, uint32_t reps, double *ms_avg);

/* task -> node for every task handed to the Binder so far (KB_NONE otherwise); out has n_tasks entries */
int  kb_get_binds(kb_engine *e, uint32_t *task_node_out);
/* current task status (KB_TASK_*) and node (or KB_NONE), n_tasks entries each; either pointer may be NULL */
int  kb_get_task_state(kb_engine *e, uint8_t *status_out, uint32_t *node_out);
/* live node state: idle/releasing [R][N], nz_cpu/nz_mem [N], pod_cnt [N]; any pointer may be NULL */
int  kb_get_node_state(kb_engine *e, double *idle, double *releasing, int64_t *nz_cpu, int64_t *nz_mem, int32_t *pod_cnt);
/* drf job shares [J], proportion queue shares [Q], proportion deserved [R][Q]; any pointer may be NULL */
int  kb_get_shares(kb_engine *e, double *job_share, double *queue_share, double *queue_deserved);
int  kb_get_stats(kb_engine *e, kb_stats *out);

/* ---- round-granular entry points for task-row sharding across GPUs ----
 * One process per GPU holds a full replica of the session.  Per round every rank calls, in this order:
 *   kb_round_begin      -> the next speculated window (identical on every rank): n_rows task rows, n_mrows distinct
 *                          task shapes (= matrix rows), list_len candidates per matrix row.  n_rows == 0: the action is
 *                          complete (gang ballot + share reduction done, decisions available).
 *   kb_round_candidates -> mask+score matrix and sorted candidate lists for matrix rows [mrow0, mrow1) — this rank's
 *                          shard — written to a caller-provided DEVICE buffer [mrow1-mrow0][list_len] of uint64 keys
 *   (all-gather of the key buffers across ranks: RCCL, done by the caller)
 *   kb_round_commit     -> the sequential commit over the whole window using the gathered table [n_mrows][list_len];
 *                          fills the caller-provided DEVICE delta buffer (kb_round_delta_doubles float64: per node
 *                          dIdle[R], dReleasing[R], d(non-zero cpu), d(non-zero mem), d(pod count)) with the committed
 *                          deltas of window rows [own_row0, own_row1) only
 *   (all-reduce(sum) of the delta buffers: RCCL, done by the caller; integer-valued float64 -> exact, order-independent)
 *   kb_round_apply      -> node state for the next round := round-start state + reduced deltas; KB_E_INTERNAL if that
 *                          differs from the replica's own commit (replicas diverged)
 * The reduced deltas are a CROSS-CHECK (every replica commits the whole window itself): they need not sit between two rounds.  The
 * deferred form takes the all-reduce off the critical path: kb_round_apply with a NULL buffer only absorbs the round's result; the
 * caller reduces round k's buffer on a side stream while round k + 1 is planned, evaluated and committed, and hands it to
 *   kb_round_check      -> queued, nothing waited for: (state at the start of round k) + reduced deltas == (state at the start of
 *                          round k + 1), called after kb_round_begin of round k + 1 has returned rows and before the one of round
 *                          k + 2 (kube-batch_amd/dist.py: behind kb_round_commit of round k + 1, in front of its kb_round_apply); with against_live != 0
 *                          after the kb_round_begin that returned n_rows == 0: == the live state (the action's last round)
 *   kb_round_check_result -> the number of values that differed in any check since the action began (one synchronisation);
 *                          non-zero: the replicas diverged
 */
/* run the engine's kernels on the caller's HIP stream (e.g. the stream its RCCL collectives are ordered on) instead of the
   engine's own: launches, collectives and the delta kernel are then ordered by the stream alone, with no host synchronisation
   between them.  0 restores the engine's own stream.  The caller keeps the stream alive. */
int  kb_engine_use_stream(kb_engine *e, uint64_t hip_stream);
int  kb_round_begin(kb_engine *e, uint32_t action /*0 allocate, 1 backfill*/, uint32_t *n_rows, uint32_t *n_mrows, uint32_t *list_len);
int  kb_round_candidates(kb_engine *e, uint32_t mrow0, uint32_t mrow1, uint64_t dev_keys_ptr);
int  kb_round_commit(kb_engine *e, uint64_t dev_all_keys_ptr, uint32_t own_row0, uint32_t own_row1, uint64_t dev_delta_ptr);
int  kb_round_apply(kb_engine *e, uint64_t dev_delta_ptr, uint32_t *done);
int  kb_round_check(kb_engine *e, uint64_t dev_delta_ptr, uint32_t against_live);
int  kb_round_check_result(kb_engine *e, uint32_t *mismatches);
int  kb_round_delta_doubles(const kb_engine *e, uint64_t *n_doubles);   /* NP * (2R + 3) float64 per buffer */
int  kb_round_decisions(kb_engine *e, kb_decision *out, uint64_t cap, uint64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* KB_ENGINE_H */
