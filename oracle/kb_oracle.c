/*
 * kb_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of kube-batch's allocate/backfill hot path (reference @ /root/reference, Go,
 * RELEASE v0.5).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (kube-batch_amd/csrc) never links, imports or calls it.
 *
 * The reference is Go and there is no Go toolchain in the build image, so the real binary cannot
 * be executed here.  Parity of this restatement is pinned by the reference's OWN known-answer
 * tests, re-stated in tests/test_oracle_kat.py (api/resource_info_test.go, api/node_info_test.go,
 * actions/allocate/allocate_test.go, util/scheduler_helper_test.go) and by the doc-level worked
 * example doc/usage/tutorial.md:297-330.  What no in-tree test pins (the vendored k8s scorers'
 * numeric outputs, container/heap pop order with duplicate queue entries, math/rand tie-break) is
 * "parity unpinned by tests": it follows the vendored source text line by line and the
 * canonicalisation of SURVEY.md §8c (first max-score node in ascending node order; maps iterated
 * in ascending key order).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/kb_engine.h"

#define KBO_PANIC (-100) /* the Go code would have panicked (Resource.Sub underflow) */

/* ================================================================================================
 * api.Resource — pkg/scheduler/api/resource_info.go:28-38
 * Dense vector + presence mask for ScalarResources (bit d-2 <=> key of dimension d exists in the map;
 * mask == 0 <=> the map is nil: the Go code never creates an empty non-nil map on this path).
 * ============================================================================================== */
typedef struct kbo_res {
  double v[KB_MAX_RES];
  uint32_t mask;
  int32_t max_task_num;
} kbo_res;

static const double minMilliCPU = 10.0;               /* resource_info.go:68 */
static const double minMilliScalar = 10.0;            /* resource_info.go:69 */
static const double minMemory = 10.0 * 1024 * 1024;   /* resource_info.go:70 */

static int g_R = 2; /* dimensions in use for the KAT helpers (session code passes R explicitly) */

static void res_zero(kbo_res *r) { memset(r, 0, sizeof(*r)); }
#define HAS(r, d) (((r)->mask >> ((d)-2)) & 1u)
#define SETK(r, d) ((r)->mask |= (1u << ((d)-2)))

/* resource_info.go:93-105 */
static int res_is_empty(const kbo_res *r, int R) {
  if (!(r->v[0] < minMilliCPU && r->v[1] < minMemory)) return 0;
  for (int d = 2; d < R; d++)
    if (HAS(r, d) && r->v[d] >= minMilliScalar) return 0;
  return 1;
}

/* resource_info.go:128-140 */
static void res_add(kbo_res *r, const kbo_res *rr, int R) {
  r->v[0] += rr->v[0];
  r->v[1] += rr->v[1];
  for (int d = 2; d < R; d++)
    if (HAS(rr, d)) { SETK(r, d); r->v[d] += rr->v[d]; }
}

/* resource_info.go:268-302 */
static int le_func(double l, double r, double diff) { return (l < r || fabs(l - r) < diff) ? 1 : 0; }
static int res_less_equal(const kbo_res *r, const kbo_res *rr, int R) {
  if (!le_func(r->v[0], rr->v[0], minMilliCPU)) return 0;
  if (!le_func(r->v[1], rr->v[1], minMemory)) return 0;
  if (r->mask == 0) return 1;
  for (int d = 2; d < R; d++) {
    if (!HAS(r, d)) continue;
    double q = r->v[d];
    if (q <= minMilliScalar) continue;
    if (rr->mask == 0) return 0;
    double rq = HAS(rr, d) ? rr->v[d] : 0.0;
    if (!le_func(q, rq, minMilliScalar)) return 0;
  }
  return 1;
}

/* resource_info.go:143-160 — returns KBO_PANIC where Go panics */
static int res_sub(kbo_res *r, const kbo_res *rr, int R) {
  if (!res_less_equal(rr, r, R)) return KBO_PANIC;
  r->v[0] -= rr->v[0];
  r->v[1] -= rr->v[1];
  for (int d = 2; d < R; d++) {
    if (!HAS(rr, d)) continue;
    if (r->mask == 0) return 0; /* "if r.ScalarResources == nil { return r }" */
    SETK(r, d);
    r->v[d] -= rr->v[d];
  }
  return 0;
}

/* resource_info.go:163-188 */
static void res_set_max(kbo_res *r, const kbo_res *rr, int R) {
  if (rr->v[0] > r->v[0]) r->v[0] = rr->v[0];
  if (rr->v[1] > r->v[1]) r->v[1] = rr->v[1];
  if (rr->mask == 0) return;
  if (r->mask == 0) { /* copies the whole map on first key and returns */
    for (int d = 2; d < R; d++)
      if (HAS(rr, d)) { SETK(r, d); r->v[d] = rr->v[d]; }
    return;
  }
  for (int d = 2; d < R; d++)
    if (HAS(rr, d)) {
      double cur = HAS(r, d) ? r->v[d] : 0.0;
      if (rr->v[d] > cur) { SETK(r, d); r->v[d] = rr->v[d]; }
    }
}

/* resource_info.go:194-214 */
static void res_fit_delta(kbo_res *r, const kbo_res *rr, int R) {
  if (rr->v[0] > 0) r->v[0] -= rr->v[0] + minMilliCPU;
  if (rr->v[1] > 0) r->v[1] -= rr->v[1] + minMemory;
  for (int d = 2; d < R; d++)
    if (HAS(rr, d) && rr->v[d] > 0) { SETK(r, d); r->v[d] -= rr->v[d] + minMilliScalar; }
}

/* resource_info.go:217-224 */
static void res_multi(kbo_res *r, double ratio, int R) {
  r->v[0] = r->v[0] * ratio;
  r->v[1] = r->v[1] * ratio;
  for (int d = 2; d < R; d++)
    if (HAS(r, d)) r->v[d] = r->v[d] * ratio;
}

/* resource_info.go:227-265 */
static int res_less(const kbo_res *r, const kbo_res *rr, int R) {
  if (!(r->v[0] < rr->v[0])) return 0;
  if (!(r->v[1] < rr->v[1])) return 0;
  if (r->mask == 0) {
    if (rr->mask != 0)
      for (int d = 2; d < R; d++)
        if (HAS(rr, d) && rr->v[d] <= minMilliScalar) return 0;
    return 1;
  }
  if (rr->mask == 0) return 0;
  for (int d = 2; d < R; d++) {
    if (!HAS(r, d)) continue;
    double rq = HAS(rr, d) ? rr->v[d] : 0.0;
    if (!(r->v[d] < rq)) return 0;
  }
  return 1;
}

/* resource_info.go:305-337 */
static void res_diff(const kbo_res *r, const kbo_res *rr, kbo_res *inc, kbo_res *dec, int R) {
  res_zero(inc);
  res_zero(dec);
  if (r->v[0] > rr->v[0]) inc->v[0] += r->v[0] - rr->v[0]; else dec->v[0] += rr->v[0] - r->v[0];
  if (r->v[1] > rr->v[1]) inc->v[1] += r->v[1] - rr->v[1]; else dec->v[1] += rr->v[1] - r->v[1];
  for (int d = 2; d < R; d++) {
    if (!HAS(r, d)) continue;
    double rq = HAS(rr, d) ? rr->v[d] : 0.0;
    if (r->v[d] > rq) { SETK(inc, d); inc->v[d] += r->v[d] - rq; }
    else { SETK(dec, d); dec->v[d] += rq - r->v[d]; }
  }
}

/* resource_info.go:349-361 */
static double res_get(const kbo_res *r, int d) {
  if (d < 2) return r->v[d];
  return HAS(r, d) ? r->v[d] : 0.0;
}

/* api/helpers/helpers.go:28-44 */
static void helpers_min(const kbo_res *l, const kbo_res *r, kbo_res *res, int R) {
  res_zero(res);
  res->v[0] = fmin(l->v[0], r->v[0]);
  res->v[1] = fmin(l->v[1], r->v[1]);
  if (l->mask == 0 || r->mask == 0) return;
  for (int d = 2; d < R; d++)
    if (HAS(l, d)) { SETK(res, d); res->v[d] = fmin(l->v[d], HAS(r, d) ? r->v[d] : 0.0); }
}

/* api/helpers/helpers.go:47-60 */
static double helpers_share(double l, double r) {
  if (r == 0) return (l == 0) ? 0.0 : 1.0;
  return l / r;
}

/* ---- KAT entry points (tests/test_oracle_kat.py drives resource_info_test.go's tables through these) ---- */
void kbo_set_dims(int R) { g_R = R; }
int kbo_res_is_empty(const kbo_res *r) { return res_is_empty(r, g_R); }
void kbo_res_add(kbo_res *r, const kbo_res *rr) { res_add(r, rr, g_R); }
int kbo_res_sub(kbo_res *r, const kbo_res *rr) { return res_sub(r, rr, g_R); }
int kbo_res_less(const kbo_res *r, const kbo_res *rr) { return res_less(r, rr, g_R); }
int kbo_res_less_equal(const kbo_res *r, const kbo_res *rr) { return res_less_equal(r, rr, g_R); }
void kbo_res_set_max(kbo_res *r, const kbo_res *rr) { res_set_max(r, rr, g_R); }
void kbo_res_fit_delta(kbo_res *r, const kbo_res *rr) { res_fit_delta(r, rr, g_R); }
void kbo_res_multi(kbo_res *r, double ratio) { res_multi(r, ratio, g_R); }
void kbo_res_diff(const kbo_res *r, const kbo_res *rr, kbo_res *inc, kbo_res *dec) { res_diff(r, rr, inc, dec, g_R); }
void kbo_res_min(const kbo_res *l, const kbo_res *r, kbo_res *out) { helpers_min(l, r, out, g_R); }
double kbo_share(double l, double r) { return helpers_share(l, r); }
/* resource_info.go:108-126 IsZero; returns -1 for the "unknown resource" panic */
int kbo_res_is_zero(const kbo_res *r, int d) {
  if (d == 0) return r->v[0] < minMilliCPU;
  if (d == 1) return r->v[1] < minMemory;
  if (r->mask == 0) return 1;
  if (!HAS(r, d)) return -1;
  return r->v[d] < minMilliScalar;
}

/* ================================================================================================
 * container/heap (Go stdlib; not in the tree) as used by util.PriorityQueue —
 * pkg/scheduler/util/priority_queue.go:26-94.  Published algorithm (go1.13 src/container/heap/heap.go):
 *   Push: append; up(n-1)         Pop: n=Len-1; Swap(0,n); down(0,n); remove last
 *   up(j):   for { i=(j-1)/2; if i==j || !less(j,i) break; swap(i,j); j=i }
 *   down(i0,n): i=i0; for { j1=2i+1; if j1>=n||j1<0 break; j=j1; if j2=j1+1<n && less(j2,j1) j=j2;
 *                           if !less(j,i) break; swap(i,j); i=j }
 * less(i,j) = lessFn(items[i], items[j])  (priority_queue.go:71-78).
 * ============================================================================================== */
typedef int (*less_fn)(void *ctx, uint32_t a, uint32_t b);
typedef struct heap_t {
  uint32_t *items;
  int n, cap;
  less_fn less;
  void *ctx;
} heap_t;

static void heap_init(heap_t *h, less_fn less, void *ctx) { h->items = NULL; h->n = 0; h->cap = 0; h->less = less; h->ctx = ctx; }
static void heap_free(heap_t *h) { free(h->items); h->items = NULL; h->n = h->cap = 0; }
static int heap_less(heap_t *h, int i, int j) { return h->less(h->ctx, h->items[i], h->items[j]); }
static void heap_swap(heap_t *h, int i, int j) { uint32_t t = h->items[i]; h->items[i] = h->items[j]; h->items[j] = t; }
static void heap_up(heap_t *h, int j) {
  for (;;) {
    int i = (j - 1) / 2; /* parent; Go's (j-1)/2 with j==0 gives 0 (trunc toward zero) */
    if (i == j || !heap_less(h, j, i)) break;
    heap_swap(h, i, j);
    j = i;
  }
}
static void heap_down(heap_t *h, int i0, int n) {
  int i = i0;
  for (;;) {
    int j1 = 2 * i + 1;
    if (j1 >= n || j1 < 0) break;
    int j = j1;
    int j2 = j1 + 1;
    if (j2 < n && heap_less(h, j2, j1)) j = j2;
    if (!heap_less(h, j, i)) break;
    heap_swap(h, i, j);
    i = j;
  }
}
static void heap_push(heap_t *h, uint32_t x) {
  if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 16; h->items = (uint32_t *)realloc(h->items, sizeof(uint32_t) * (size_t)h->cap); }
  h->items[h->n++] = x;
  heap_up(h, h->n - 1);
}
static uint32_t heap_pop(heap_t *h) {
  int n = h->n - 1;
  heap_swap(h, 0, n);
  heap_down(h, 0, n);
  uint32_t x = h->items[h->n - 1];
  h->n--;
  return x;
}

/* KAT helper: heap over integer keys with "less = key[a] < key[b]" to test the sift mechanics */
static int key_less(void *ctx, uint32_t a, uint32_t b) { const double *k = (const double *)ctx; return k[a] < k[b]; }
int kbo_heap_order(const double *keys, const uint32_t *push_ids, int n, uint32_t *pop_out) {
  heap_t h;
  heap_init(&h, key_less, (void *)keys);
  for (int i = 0; i < n; i++) heap_push(&h, push_ids[i]);
  for (int i = 0; i < n; i++) pop_out[i] = heap_pop(&h);
  heap_free(&h);
  return 0;
}

/* ================================================================================================
 * util.SelectBestNode / findMaxScores — pkg/scheduler/util/scheduler_helper.go:188-208
 * Canonical tie-break: rand.Intn(len(maxScores)) replaced by index 0.
 * Returns the position in the list of the selected entry; fills all max positions for the KAT.
 * ============================================================================================== */
int kbo_select_best(const double *scores, int n, int *max_idx_out, int *n_max_out) {
  int cnt = 0;
  double maxScore = scores[0];
  for (int i = 0; i < n; i++) {
    if (scores[i] > maxScore) { maxScore = scores[i]; cnt = 0; max_idx_out[cnt++] = i; }
    else if (scores[i] == maxScore) { max_idx_out[cnt++] = i; }
  }
  *n_max_out = cnt;
  return max_idx_out[0];
}

/* ================================================================================================
 * k8s scorers (vendored k8s.io/kubernetes v1.16.8)
 * ============================================================================================== */
#define MAX_PRIORITY 10 /* vendor/k8s.io/kubernetes/pkg/scheduler/api/types.go:35 */

/* vendor/.../priorities/least_requested.go:50-58 */
static int64_t least_requested_score(int64_t requested, int64_t capacity) {
  if (capacity == 0) return 0;
  if (requested > capacity) return 0;
  return ((capacity - requested) * (int64_t)MAX_PRIORITY) / capacity;
}
/* vendor/.../priorities/most_requested.go:52-61 */
static int64_t most_requested_score(int64_t requested, int64_t capacity) {
  if (capacity == 0) return 0;
  if (requested > capacity) return 0;
  return (requested * MAX_PRIORITY) / capacity;
}
/* vendor/.../priorities/balanced_resource_allocation.go:74-79 */
static double fraction_of_capacity(int64_t requested, int64_t capacity) {
  if (capacity == 0) return 1;
  return (double)requested / (double)capacity;
}
/* least_requested.go:36-45 (weights cpu 1, memory 1: resource_allocation.go:46) */
static int64_t least_scorer(int64_t rc, int64_t ac, int64_t rm, int64_t am) {
  int64_t nodeScore = 0, weightSum = 0;
  nodeScore += least_requested_score(rm, am) * 1; weightSum += 1;
  nodeScore += least_requested_score(rc, ac) * 1; weightSum += 1;
  return nodeScore / weightSum;
}
/* most_requested.go:34-43 */
static int64_t most_scorer(int64_t rc, int64_t ac, int64_t rm, int64_t am) {
  int64_t nodeScore = 0, weightSum = 0;
  nodeScore += most_requested_score(rm, am) * 1; weightSum += 1;
  nodeScore += most_requested_score(rc, ac) * 1; weightSum += 1;
  return nodeScore / weightSum;
}
/* balanced_resource_allocation.go:42-71 (BalanceAttachedNodeVolumes gate off) */
static int64_t balanced_scorer(int64_t rc, int64_t ac, int64_t rm, int64_t am) {
  double cpuFraction = fraction_of_capacity(rc, ac);
  double memoryFraction = fraction_of_capacity(rm, am);
  if (cpuFraction >= 1 || memoryFraction >= 1) return 0;
  double diff = fabs(cpuFraction - memoryFraction);
  return (int64_t)((1 - diff) * (double)MAX_PRIORITY);
}
/* KAT entry: the three scorers for one (pod, node) */
void kbo_scorers(int64_t rc, int64_t ac, int64_t rm, int64_t am, int64_t *least, int64_t *most, int64_t *bal) {
  *least = least_scorer(rc, ac, rm, am);
  *most = most_scorer(rc, ac, rm, am);
  *bal = balanced_scorer(rc, ac, rm, am);
}

/* ================================================================================================
 * Session
 * ============================================================================================== */
typedef struct o_node {
  kbo_res idle, releasing, used, allocatable;
  int64_t alloc_cpu, alloc_mem, nz_cpu, nz_mem; /* k8s nodeinfo view */
  int32_t pod_cnt;
  uint32_t cls;
  uint64_t *ports;   /* [Wh] nodeinfo.UsedPorts() as interned bits (vendor/.../nodeinfo/host_ports.go); Wh = kbo_session.Wh words per mask */
  uint64_t *base_ports;   /* [Wh] the share of `ports` that belongs to pods outside the session */
} o_node;

/* host-port masks of Wh words (kb_snapshot.port_words): the reference keeps sets of (ip, protocol, port) without a width */
static inline int pm_meet(const uint64_t *a, const uint64_t *b, uint32_t W) { for (uint32_t w = 0; w < W; w++) if (a[w] & b[w]) return 1; return 0; }
static inline void pm_or(uint64_t *a, const uint64_t *b, uint32_t W) { for (uint32_t w = 0; w < W; w++) a[w] |= b[w]; }
static inline int pm_eq(const uint64_t *a, const uint64_t *b, uint32_t W) { for (uint32_t w = 0; w < W; w++) if (a[w] != b[w]) return 0; return 1; }

typedef struct o_task {
  kbo_res resreq, init_resreq;
  int64_t nz_cpu, nz_mem;
  uint32_t job, cls, node;
  const uint64_t *port_want, *port_conflict;   /* [Wh] */
  uint8_t on_node;         /* the task is in some ni.Tasks (preempt bookkeeping) */
  uint8_t node_status;     /* status of the clone ni.Tasks holds (api/node_info.go:186: the node keeps a copy taken at AddTask) */
  uint8_t evict_protected; /* conformance: kube-system namespace or a system-critical priority class (conformance.go:44-58) */
  int32_t priority;
  int64_t creation;
  uint8_t status;
} o_task;

typedef struct o_job {
  uint32_t queue, t0, t1;
  int32_t min_available, priority;
  int64_t creation;
  int valid;            /* survived the JobValid filter (framework/session.go:89-107) */
  int32_t cnt[10];      /* len(TaskStatusIndex[status]) */
  kbo_res drf_allocated;
  double drf_share;
  int tasks_init;       /* pendingTasks[job.UID] created (allocate.go:110) */
  heap_t tasks;
} o_job;

typedef struct o_queue {
  int32_t weight;
  int64_t creation;
  int has_attr;         /* proportion: queueOpts entry exists (queue has a job in the session) */
  kbo_res deserved, allocated, request;
  double share;
  int has_jobs_heap;
  heap_t jobs;
} o_queue;

typedef struct plug_opt { uint32_t plugin, enabled; int32_t args[8]; uint32_t args_set; } plug_opt;

typedef struct kbo_session {
  int R;
  uint32_t N, T, J, Q, n_tc, n_nc;
  o_node *nodes;
  uint32_t Wh;            /* 64-bit words per host-port mask (kb_snapshot.port_words, 0 reads as 1) */
  uint64_t *pm_nodes, *pm_base, *pm_want, *pm_conf, *pm_scratch;   /* [N][Wh], [N][Wh], [T][Wh], [T][Wh], [Wh] */
  o_task *tasks;
  o_job *jobs;
  o_queue *queues;
  uint8_t *compat;
  int32_t *affinity;   /* [n_tc][n_nc] NodeAffinity Map counts (node_affinity.go:34-77), NULL: no preferred terms anywhere */
  /* conf */
  int n_tiers;
  uint32_t *tier_begin;
  plug_opt *plugins;
  int has_plugin[8];
  int w_least, w_most, w_nodeaff, w_podaff, w_bal;
  int pred_enabled, nodeorder_enabled;
  /* drf / proportion */
  kbo_res drf_total, prop_total;
  /* outputs */
  kb_decision *decisions;
  uint64_t n_dec, cap_dec;
  uint32_t *bind_node;  /* [T] */
  uint32_t *bind_order; /* task ids in dispatch order */
  uint64_t n_binds;
  uint64_t evals, popped;
  uint32_t *evictions;  /* task ids in the order stmt.Commit hands them to cache.Evict */
  uint64_t n_evict, cap_evict;
  int panic;
  int threads;
  uint64_t task_limit;  /* cpu_baseline sample: stop the allocate loop after this many popped tasks (0 = none) */
  /* fast mode (kbo_set_fast): per task shape a cached row of keys over all nodes + a max-tree, repaired one node at a time */
  int fast;
  struct fast_t *fx;
  /* node -> tasks in ni.Tasks (bitmap over T per node would be too big: a linked list through next_on_node, head per node;
     the scans below sort what they collect, so list order does not matter).  Built by node_index_build(). */
  uint32_t *node_head, *next_on_node, *prev_on_node;
  /* what preempt / reclaim did, in order, in the shape the engine's C ABI reports it (include/kb_engine.h: kb_stmt_op): every
     Statement.Evict / Statement.Pipeline with the number of its ssn.Statement() (1-based, in creation order), a COMMIT / DISCARD
     marker closing every statement that holds at least one operation; reclaim has no Statement: ssn.Evict / ssn.Pipeline with stmt 0 */
  kb_stmt_op *journal;
  uint64_t n_journal, cap_journal;
  uint32_t stmt_no;
  uint64_t mutations;   /* Statement operations and their undos so far (fast preempt: "nothing has changed since") */
  struct pfast_t *px;   /* fast mode of preempt (below) */
  /* inter-pod (anti)affinity (include/kb_engine.h: kb_interpod), NULL tables: no pod carries a term.  Counts are kept
     incrementally by ssn_allocate / ssn_pipeline (allocate and backfill only add; preempt / reclaim refuse such sessions). */
  int ip_on;
  uint32_t ip_C, ip_D, ip_P, ip_S, ip_Z, ip_Wc, ip_Wp;   /* Wc / Wp: 64-bit words per task mask */
  uint32_t ip_Z0;        /* Z as the snapshot gave it: the pods it stands for never leave their ni.Tasks inside a session */
  int32_t *ip_unb_n;     /* [N] pods with an empty Spec.NodeName that THIS session added to ni.Tasks of node n and that are still there */
  uint32_t *ip_ctr_dom, *ip_cls_dom, *ip_task_sig;
  int32_t *ip_ctr_count, *ip_ctr_total, *ip_cls_bound, *ip_cls_unbound, *ip_sig_weight;
  uint64_t *ip_task_inc, *ip_task_forbid, *ip_task_cls_inc;
  uint16_t *ip_task_require;
  uint8_t *ip_task_self;
} kbo_session;

static int find_plugin_enabled(const kbo_session *s, uint32_t plugin, uint32_t en_bit) {
  /* is there an option for `plugin` whose Enabled bit is set (session_plugins.go isEnabled) */
  for (int t = 0; t < s->n_tiers; t++)
    for (uint32_t p = s->tier_begin[t]; p < s->tier_begin[t + 1]; p++)
      if (s->plugins[p].plugin == plugin && (s->plugins[p].enabled & en_bit)) return 1;
  return 0;
}

/* ---- JobInfo counters: pkg/scheduler/api/job_info.go:383-434, api/helpers.go:64-71 ---- */
static int allocated_status(int st) { return st == KB_TASK_BOUND || st == KB_TASK_BINDING || st == KB_TASK_RUNNING || st == KB_TASK_ALLOCATED; }
static int32_t job_ready_num(const o_job *j) {
  int32_t n = 0;
  for (int st = 0; st < 10; st++)
    if (allocated_status(st) || st == KB_TASK_SUCCEEDED) n += j->cnt[st];
  return n;
}
static int32_t job_valid_num(const o_job *j) {
  int32_t n = 0;
  for (int st = 0; st < 10; st++)
    if (allocated_status(st) || st == KB_TASK_SUCCEEDED || st == KB_TASK_PIPELINED || st == KB_TASK_PENDING) n += j->cnt[st];
  return n;
}
static int job_ready(const o_job *j) { return job_ready_num(j) >= j->min_available; }

/* ---- drf: plugins/drf/drf.go:157-171 ---- */
static double drf_calc_share(const kbo_session *s, const kbo_res *allocated) {
  double res = 0;
  for (int d = 0; d < s->R; d++) {
    if (d >= 2 && !HAS(&s->drf_total, d)) continue; /* totalResource.ResourceNames() */
    double share = helpers_share(res_get(allocated, d), res_get(&s->drf_total, d));
    if (share > res) res = share;
  }
  return res;
}
/* ---- proportion: plugins/proportion/proportion.go:241-253 ---- */
static void prop_update_share(const kbo_session *s, o_queue *q) {
  double res = 0;
  for (int d = 0; d < s->R; d++) {
    if (d >= 2 && !HAS(&q->deserved, d)) continue; /* attr.deserved.ResourceNames() */
    double share = helpers_share(res_get(&q->allocated, d), res_get(&q->deserved, d));
    if (share > res) res = share;
  }
  q->share = res;
}

/* ---- tiered order functions: framework/session_plugins.go:243-331 ---- */
static int job_order_less(void *ctx, uint32_t l, uint32_t r) {
  kbo_session *s = (kbo_session *)ctx;
  const o_job *lv = &s->jobs[l], *rv = &s->jobs[r];
  for (int t = 0; t < s->n_tiers; t++)
    for (uint32_t p = s->tier_begin[t]; p < s->tier_begin[t + 1]; p++) {
      const plug_opt *po = &s->plugins[p];
      if (!(po->enabled & KB_EN_JOB_ORDER)) continue;
      int j = 0;
      if (po->plugin == KB_PLUGIN_PRIORITY) { /* priority.go:61-77 */
        if (lv->priority > rv->priority) j = -1; else if (lv->priority < rv->priority) j = 1;
      } else if (po->plugin == KB_PLUGIN_GANG) { /* gang.go:96-119 */
        int lReady = job_ready(lv), rReady = job_ready(rv);
        if (lReady && rReady) j = 0; else if (lReady) j = 1; else if (rReady) j = -1; else j = 0;
      } else if (po->plugin == KB_PLUGIN_DRF) { /* drf.go:114-130 */
        if (lv->drf_share == rv->drf_share) j = 0; else if (lv->drf_share < rv->drf_share) j = -1; else j = 1;
      } else continue;
      if (j != 0) return j < 0;
    }
  if (lv->creation == rv->creation) return l < r; /* UID order == canonical index order */
  return lv->creation < rv->creation;
}
static int queue_order_less(void *ctx, uint32_t l, uint32_t r) {
  kbo_session *s = (kbo_session *)ctx;
  const o_queue *lv = &s->queues[l], *rv = &s->queues[r];
  for (int t = 0; t < s->n_tiers; t++)
    for (uint32_t p = s->tier_begin[t]; p < s->tier_begin[t + 1]; p++) {
      const plug_opt *po = &s->plugins[p];
      if (!(po->enabled & KB_EN_QUEUE_ORDER)) continue;
      if (po->plugin != KB_PLUGIN_PROPORTION) continue;
      int j; /* proportion.go:156-169 */
      if (lv->share == rv->share) j = 0; else if (lv->share < rv->share) j = -1; else j = 1;
      if (j != 0) return j < 0;
    }
  if (lv->creation == rv->creation) return l < r;
  return lv->creation < rv->creation;
}
static int task_order_less(void *ctx, uint32_t l, uint32_t r) {
  kbo_session *s = (kbo_session *)ctx;
  const o_task *lv = &s->tasks[l], *rv = &s->tasks[r];
  for (int t = 0; t < s->n_tiers; t++)
    for (uint32_t p = s->tier_begin[t]; p < s->tier_begin[t + 1]; p++) {
      const plug_opt *po = &s->plugins[p];
      if (!(po->enabled & KB_EN_TASK_ORDER)) continue;
      if (po->plugin != KB_PLUGIN_PRIORITY) continue;
      int j; /* priority.go:40-56 */
      if (lv->priority == rv->priority) j = 0; else if (lv->priority > rv->priority) j = -1; else j = 1;
      if (j != 0) return j < 0;
    }
  if (lv->creation == rv->creation) return l < r;
  return lv->creation < rv->creation;
}

/* session_plugins.go:165-179 + proportion.go:198-209 (no Enabled* check on Overused) */
static int ssn_overused(const kbo_session *s, uint32_t q) {
  if (!s->has_plugin[KB_PLUGIN_PROPORTION]) return 0;
  const o_queue *attr = &s->queues[q];
  return res_less_equal(&attr->deserved, &attr->allocated, s->R);
}
/* session_plugins.go:182-200 + gang.go:122-125 */
static int ssn_job_ready(const kbo_session *s, const o_job *j) {
  if (find_plugin_enabled(s, KB_PLUGIN_GANG, KB_EN_JOB_READY)) return job_ready(j);
  return 1;
}

/* ---- per-(task,node) predicate and score ---- */
static int class_ok(const kbo_session *s, uint32_t tc, uint32_t nc) {
  if (!s->compat) return 1;
  uint32_t bit = tc * s->n_nc + nc;
  return (s->compat[bit >> 3] >> (bit & 7)) & 1;
}
/* plugins/predicates/predicates.go:123-265 with the static checks p2..p7 folded into class_ok */
/* plugins/predicates/predicates.go:249-262 -> PodAffinityChecker.InterPodAffinityMatches (vendor/.../algorithm/predicates/
   predicates.go:1261-1290, meta == nil) on the kb_interpod tables: existing pods' anti-affinity and the pod's own anti-affinity
   forbid a positive count in the node's domain (:1400-1441, :1535-1543); the pod's own affinity needs one, unless no pod at all
   matches its terms and it matches them itself (:1519-1566). */
static int interpod_predicate(const kbo_session *s, uint32_t t, uint32_t n) {
  for (uint32_t w = 0; w < s->ip_Wc; w++) {
    uint64_t fb = s->ip_task_forbid[(size_t)t * s->ip_Wc + w];
    for (uint32_t c = 64 * w; fb; c++, fb >>= 1) {
      if (!(fb & 1)) continue;
      uint32_t d = s->ip_ctr_dom[(size_t)c * s->N + n];
      if (d != KB_NONE && s->ip_ctr_count[(size_t)c * s->ip_D + d] > 0) return 0;
    }
  }
  uint32_t r = s->ip_task_require[t];
  if (r != 0xFFFF) {
    uint32_t d = s->ip_ctr_dom[(size_t)r * s->N + n];
    if (!(d != KB_NONE && s->ip_ctr_count[(size_t)r * s->ip_D + d] > 0)) {
      if (s->ip_ctr_total[r] > 0 || !s->ip_task_self[t]) return 0;
    }
  }
  return 1;
}
static int plugin_predicate(const kbo_session *s, const o_task *t, const o_node *n) {
  if (!s->pred_enabled) return 1; /* session_plugins.go:334-351: no enabled predicate fn => nil */
  if (n->allocatable.max_task_num <= n->pod_cnt) return 0; /* predicates.go:127 */
  if (!class_ok(s, t->cls, n->cls)) return 0;
  /* PodFitsHostPorts (predicates.go:181-190 -> vendor/.../predicates/predicates.go:1153-1175): any wanted port in conflict with a used one */
  if (pm_meet(n->ports, t->port_conflict, s->Wh)) return 0;
  if (s->ip_on && !interpod_predicate(s, (uint32_t)(t - s->tasks), (uint32_t)(n - s->nodes))) return 0;
  return 1;
}
/* actions/allocate/allocate.go:73-87 */
static int allocate_predicate(const kbo_session *s, const o_task *t, const o_node *n) {
  if (!res_less_equal(&t->init_resreq, &n->idle, s->R) && !res_less_equal(&t->init_resreq, &n->releasing, s->R)) return 0;
  return plugin_predicate(s, t, n);
}
/* util/scheduler_helper.go:89-171 with nodeorder's five configs (plugins/nodeorder/nodeorder.go:140-168);
   requested = nodeInfo.NonZeroRequest + pod non-zero request (resource_allocation.go:100-112). */
static double node_score(const kbo_session *s, const o_task *t, const o_node *n) {
  if (!s->nodeorder_enabled) return 0.0;
  int64_t rc = n->nz_cpu + t->nz_cpu, rm = n->nz_mem + t->nz_mem;
  int least = (int)least_scorer(rc, n->alloc_cpu, rm, n->alloc_mem);
  int most = (int)most_scorer(rc, n->alloc_cpu, rm, n->alloc_mem);
  int nodeaff = 0; /* the NodeAffinity config needs its Reduce over the whole feasible list: added by node_affinity_reduce() below */
  int podaff = 0;  /* no pod (anti)affinity: interpod_affinity.go yields 0 for every node */
  int bal = (int)balanced_scorer(rc, n->alloc_cpu, rm, n->alloc_mem);
  double score = 0;
  score += (double)(least * s->w_least);   /* scheduler_helper.go:162-168: result[i].Score += float64(results[j][i].Score * Weight) */
  score += (double)(most * s->w_most);
  score += (double)(nodeaff * s->w_nodeaff);
  score += (double)(podaff * s->w_podaff);
  score += (double)(bal * s->w_bal);
  return score;
}

/* ---- worker pool mirroring workqueue.ParallelizeUntil(ctx, 16, len(nodes), fn)
        (vendor/k8s.io/client-go/util/workqueue/parallelizer.go:29-63): a shared piece counter drained by workers ---- */
typedef struct pool_t {
  int nthreads;
  pthread_t *th;
  pthread_mutex_t mu;
  pthread_cond_t cv_start, cv_done;
  int generation, running, stop;
  kbo_session *s;
  const o_task *task;
  int fit_mode;
  uint8_t *feas;     /* [N] */
  double *score;     /* [N] */
  volatile int next;
} pool_t;
static pool_t g_pool;
static int g_pool_threads = 0;
#define PIECE_CHUNK 64

static void eval_range(kbo_session *s, const o_task *t, int fit_mode, uint8_t *feas, double *score, uint32_t a, uint32_t b) {
  for (uint32_t i = a; i < b; i++) {
    const o_node *n = &s->nodes[i];
    int ok = fit_mode ? allocate_predicate(s, t, n) : plugin_predicate(s, t, n);
    feas[i] = (uint8_t)ok;
    score[i] = ok ? node_score(s, t, n) : 0.0;
  }
}
static void *pool_worker(void *arg) {
  pool_t *p = (pool_t *)arg;
  int seen = 0;
  for (;;) {
    pthread_mutex_lock(&p->mu);
    while (p->generation == seen && !p->stop) pthread_cond_wait(&p->cv_start, &p->mu);
    if (p->stop) { pthread_mutex_unlock(&p->mu); return NULL; }
    seen = p->generation;
    pthread_mutex_unlock(&p->mu);
    for (;;) {
      int piece = __sync_fetch_and_add(&p->next, PIECE_CHUNK);
      if ((uint32_t)piece >= p->s->N) break;
      uint32_t b = (uint32_t)piece + PIECE_CHUNK;
      if (b > p->s->N) b = p->s->N;
      eval_range(p->s, p->task, p->fit_mode, p->feas, p->score, (uint32_t)piece, b);
    }
    pthread_mutex_lock(&p->mu);
    if (--p->running == 0) pthread_cond_signal(&p->cv_done);
    pthread_mutex_unlock(&p->mu);
  }
}
static void pool_start(int nthreads) {
  if (g_pool_threads == nthreads) return;
  if (g_pool_threads) {
    pthread_mutex_lock(&g_pool.mu); g_pool.stop = 1; pthread_cond_broadcast(&g_pool.cv_start); pthread_mutex_unlock(&g_pool.mu);
    for (int i = 0; i < g_pool.nthreads; i++) pthread_join(g_pool.th[i], NULL);
    free(g_pool.th); g_pool_threads = 0;
  }
  if (nthreads <= 1) return;
  memset(&g_pool, 0, sizeof(g_pool));
  g_pool.nthreads = nthreads;
  g_pool.th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  pthread_mutex_init(&g_pool.mu, NULL);
  pthread_cond_init(&g_pool.cv_start, NULL);
  pthread_cond_init(&g_pool.cv_done, NULL);
  for (int i = 0; i < nthreads; i++) pthread_create(&g_pool.th[i], NULL, pool_worker, &g_pool);
  g_pool_threads = nthreads;
}
/* nodeorder's NodeAffinity config: Map = CalculateNodeAffinityPriorityMap (node_affinity.go:34-77, the per-class-pair count
   of the snapshot), Reduce = NormalizeReduce(MaxPriority=10, reverse=false) over the FEASIBLE nodes (reduce.go:28-63: maxCount == 0
   leaves the zeros; else score = 10 * count / maxCount, integer division), then Score += float64(score * weight)
   (scheduler_helper.go:162-168). */
static void node_affinity_reduce(const kbo_session *s, const o_task *t, const uint8_t *feas, double *score) {
  if (!s->affinity || !s->nodeorder_enabled) return;
  const int32_t *row = &s->affinity[(size_t)t->cls * s->n_nc];
  int max_count = 0;
  for (uint32_t n = 0; n < s->N; n++)
    if (feas[n] && row[s->nodes[n].cls] > max_count) max_count = row[s->nodes[n].cls];
  if (max_count == 0) return;
  for (uint32_t n = 0; n < s->N; n++)
    if (feas[n]) score[n] += (double)((10 * row[s->nodes[n].cls] / max_count) * s->w_nodeaff);
}
/* nodeorder's InterPodAffinityPriority (plugins/nodeorder/nodeorder.go:156-160 -> vendor/.../priorities/interpod_affinity.go:99-235)
   on the kb_interpod tables.  Only the pods of the FEASIBLE nodes are seen (util/scheduler_helper.go:226-238); a pod whose
   Spec.NodeName is still empty "lives" on the first node that holds any such pod (nodeorder.go:48-62, canonical: ascending name). */
static void interpod_priority(const kbo_session *s, const o_task *tk, const uint8_t *feas, double *score) {
  if (!s->ip_on || !s->nodeorder_enabled) return;
  const uint32_t t = (uint32_t)(tk - s->tasks);
  if (s->ip_task_sig[t] == KB_NONE) return;
  const int32_t *w = &s->ip_sig_weight[(size_t)s->ip_task_sig[t] * s->ip_P];
  const uint32_t N = s->N;
  int64_t *counts = (int64_t *)calloc(N ? N : 1, sizeof(int64_t));
  int64_t *bound = (int64_t *)malloc(sizeof(int64_t) * (N ? N : 1));   /* per domain id (< N) */
  for (uint32_t p = 0; p < s->ip_P; p++) {
    if (w[p] == 0) continue;
    const uint32_t *dom = &s->ip_cls_dom[(size_t)p * N];
    const int32_t *cb = &s->ip_cls_bound[(size_t)p * N], *cu = &s->ip_cls_unbound[(size_t)p * N];
    memset(bound, 0, sizeof(int64_t) * (N ? N : 1));
    int64_t zs = 0;
    for (uint32_t n = 0; n < N; n++) {
      if (!feas[n]) continue;
      zs += cu[n];
      if (dom[n] != KB_NONE) bound[dom[n]] += cb[n];
    }
    const uint32_t zdom = s->ip_Z != KB_NONE ? dom[s->ip_Z] : KB_NONE;
    for (uint32_t i = 0; i < N; i++) {
      if (!feas[i] || dom[i] == KB_NONE) continue;
      counts[i] += (int64_t)w[p] * (bound[dom[i]] + (dom[i] == zdom ? zs : 0));
    }
  }
  int64_t mx = 0, mn = 0;                                      /* interpod_affinity.go:213-220: both start at 0 */
  for (uint32_t i = 0; i < N; i++) {
    if (!feas[i]) continue;
    if (counts[i] > mx) mx = counts[i];
    if (counts[i] < mn) mn = counts[i];
  }
  if (mx - mn > 0)
    for (uint32_t i = 0; i < N; i++) {
      if (!feas[i]) continue;
      double f = 10.0 * ((double)(counts[i] - mn) / (double)(mx - mn));   /* :226-228 */
      score[i] += (double)((int)f * s->w_podaff);
    }
  free(counts); free(bound);
}
/* the pod joins ni.Tasks of node n (Allocate or Pipeline: both AddTask) and, for ssn.Allocate, the PodLister's allocated set */
static void interpod_placed(kbo_session *s, uint32_t t, uint32_t n, int allocated) {
  if (!s->ip_on) return;
  for (uint32_t w = 0; w < s->ip_Wp; w++) {
    uint64_t m = s->ip_task_cls_inc[(size_t)t * s->ip_Wp + w];
    for (uint32_t p = 64 * w; m; p++, m >>= 1)
      if (m & 1) s->ip_cls_unbound[(size_t)p * s->N + n] += 1;
  }
  s->ip_unb_n[n] += 1;
  if (n < s->ip_Z) s->ip_Z = n;
  if (!allocated) return;
  for (uint32_t w = 0; w < s->ip_Wc; w++) {
    uint64_t m = s->ip_task_inc[(size_t)t * s->ip_Wc + w];
    for (uint32_t c = 64 * w; m; c++, m >>= 1) {
      if (!(m & 1)) continue;
      s->ip_ctr_total[c] += 1;
      uint32_t d = s->ip_ctr_dom[(size_t)c * s->N + n];
      if (d != KB_NONE) s->ip_ctr_count[(size_t)c * s->ip_D + d] += 1;
    }
  }
}
/* The evict actions move pods the other way.  What each table counts decides what moves:
     predicate counters (ip_ctr_*): the PodLister's pods = session tasks in an ALLOCATED status (plugins/util/util.go:37-60).  Evict makes its
       victim Releasing (statement.go:36-69, session.go:317-354 -> job.UpdateTaskStatus): it leaves the list although it stays in ni.Tasks;
       the undo of a discarded statement brings it back (statement.go:83-110);
     priority classes (ip_cls_*): the pods in ni.Tasks (nodeorder builds its nodeInfo from node.Pods()): an eviction changes nothing there, a
       Pipeline adds the preemptor (Spec.NodeName still empty: the "unbound" count, and Z), its undo takes it out again. */
static void interpod_allocated_status(kbo_session *s, uint32_t t, int joins) {   /* task t (on its node) enters (+1) / leaves (-1) the allocated statuses */
#ifdef KBO_NEGATIVE_CONTROL_NO_IP_EVICT   /* tests/test_interpod_oracle_cpu.py builds this variant to show that its cases notice */
  return;
#endif
  if (!s->ip_on || !s->tasks[t].on_node) return;
  const uint32_t n = s->tasks[t].node;
  for (uint32_t w = 0; w < s->ip_Wc; w++) {
    uint64_t m = s->ip_task_inc[(size_t)t * s->ip_Wc + w];
    for (uint32_t c = 64 * w; m; c++, m >>= 1) {
      if (!(m & 1)) continue;
      s->ip_ctr_total[c] += joins;
      uint32_t d = s->ip_ctr_dom[(size_t)c * s->N + n];
      if (d != KB_NONE) s->ip_ctr_count[(size_t)c * s->ip_D + d] += joins;
    }
  }
}
static void interpod_unpipelined(kbo_session *s, uint32_t t, uint32_t n) {   /* a task this session pipelined leaves ni.Tasks of node n again */
  if (!s->ip_on) return;
  for (uint32_t w = 0; w < s->ip_Wp; w++) {
    uint64_t m = s->ip_task_cls_inc[(size_t)t * s->ip_Wp + w];
    for (uint32_t p = 64 * w; m; p++, m >>= 1)
      if (m & 1) s->ip_cls_unbound[(size_t)p * s->N + n] -= 1;
  }
  s->ip_unb_n[n] -= 1;
  if (n == s->ip_Z && s->ip_unb_n[n] == 0 && n != s->ip_Z0) {   /* Z = the first node (ascending) that holds any pod with an empty Spec.NodeName */
    uint32_t z = s->ip_Z0;
    for (uint32_t i = n + 1; i < s->N && i < s->ip_Z0; i++)
      if (s->ip_unb_n[i] > 0) { z = i; break; }
    s->ip_Z = z;
  }
}
static void eval_all_nodes_raw(kbo_session *s, const o_task *t, int fit_mode, uint8_t *feas, double *score);
static void eval_all_nodes(kbo_session *s, const o_task *t, int fit_mode, uint8_t *feas, double *score) {
  eval_all_nodes_raw(s, t, fit_mode, feas, score);
  node_affinity_reduce(s, t, feas, score);
  interpod_priority(s, t, feas, score);
}
static void eval_all_nodes_raw(kbo_session *s, const o_task *t, int fit_mode, uint8_t *feas, double *score) {
  if (s->threads <= 1 || s->N < 256) { eval_range(s, t, fit_mode, feas, score, 0, s->N); return; }
  pool_start(s->threads);
  pool_t *p = &g_pool;
  pthread_mutex_lock(&p->mu);
  p->s = s; p->task = t; p->fit_mode = fit_mode; p->feas = feas; p->score = score; p->next = 0;
  p->running = p->nthreads; p->generation++;
  pthread_cond_broadcast(&p->cv_start);
  while (p->running) pthread_cond_wait(&p->cv_done, &p->mu);
  pthread_mutex_unlock(&p->mu);
}

/* ---- job status bookkeeping: job_info.go:247-264 (UpdateTaskStatus = delete + add) ---- */
static void job_set_status(kbo_session *s, uint32_t t, int st) {
  o_task *tk = &s->tasks[t];
  o_job *j = &s->jobs[tk->job];
  j->cnt[tk->status]--;
  tk->status = (uint8_t)st;
  j->cnt[st]++;
}

static void push_decision(kbo_session *s, uint32_t task, uint32_t node, uint32_t kind) {
  if (s->n_dec == s->cap_dec) { s->cap_dec = s->cap_dec ? s->cap_dec * 2 : 1024; s->decisions = (kb_decision *)realloc(s->decisions, sizeof(kb_decision) * s->cap_dec); }
  kb_decision d = {task, node, kind, 0};
  s->decisions[s->n_dec++] = d;
}

/* plugin event handlers: drf.go:135-145, proportion.go:212-223 (fired by both Allocate and Pipeline) */
static void fire_allocate_event(kbo_session *s, uint32_t t) {
  o_task *tk = &s->tasks[t];
  o_job *j = &s->jobs[tk->job];
  if (s->has_plugin[KB_PLUGIN_DRF]) { res_add(&j->drf_allocated, &tk->resreq, s->R); j->drf_share = drf_calc_share(s, &j->drf_allocated); }
  if (s->has_plugin[KB_PLUGIN_PROPORTION]) { o_queue *q = &s->queues[j->queue]; res_add(&q->allocated, &tk->resreq, s->R); prop_update_share(s, q); }
}

/* framework/session.go:290-314 dispatch: cache.Bind + status Binding */
static void ssn_dispatch(kbo_session *s, uint32_t t) {
  s->bind_node[t] = s->tasks[t].node;
  s->bind_order[s->n_binds++] = t;
  job_set_status(s, t, KB_TASK_BINDING);
}

/* framework/session.go:235-288 */
static int ssn_allocate(kbo_session *s, uint32_t t, uint32_t n) {
  o_task *tk = &s->tasks[t];
  o_node *nd = &s->nodes[n];
  o_job *j = &s->jobs[tk->job];
  job_set_status(s, t, KB_TASK_ALLOCATED);                          /* session.go:243 (before node.AddTask) */
  /* node.AddTask: api/node_info.go:172-212.  :173-176 task.NodeName is sticky (RemoveTask never clears it: a task a discarded
     statement un-pipelined keeps its old NodeName and cannot join another node); :178-182 already in ni.Tasks */
  if ((tk->node != KB_NONE && tk->node != n) || tk->on_node) return -1;
  /* status Allocated -> allocateIdleResource (node_info.go:161-167) */
  if (!res_less_equal(&tk->resreq, &nd->idle, s->R)) return -1;     /* "Selected node NotReady": returns before callbacks */
  if (res_sub(&nd->idle, &tk->resreq, s->R) == KBO_PANIC) { s->panic = 1; return KBO_PANIC; }
  res_add(&nd->used, &tk->resreq, s->R);
  tk->node = n;
  nd->pod_cnt += 1;                 /* ni.Tasks[key] = ti ; k8s NodeInfo is rebuilt from ni.Pods() per evaluation */
  pm_or(nd->ports, tk->port_want, s->Wh);       /* ... including its UsedPorts (node_info.go:582-607 updateUsedPorts) */
  nd->nz_cpu += tk->nz_cpu;         /* vendor/.../nodeinfo/node_info.go:502-517 AddPod */
  nd->nz_mem += tk->nz_mem;
  tk->node_status = KB_TASK_ALLOCATED;
  tk->on_node = 1;
  interpod_placed(s, t, n, 1);
  push_decision(s, t, n, 0);
  fire_allocate_event(s, t);
  if (ssn_job_ready(s, j)) {        /* session.go:277-285: dispatch every Allocated task of the job (canonical: ascending UID) */
    for (uint32_t i = j->t0; i < j->t1; i++)
      if (s->tasks[i].status == KB_TASK_ALLOCATED) ssn_dispatch(s, i);
  }
  return 0;
}
/* framework/session.go:194-232 */
static int ssn_pipeline(kbo_session *s, uint32_t t, uint32_t n) {
  o_task *tk = &s->tasks[t];
  o_node *nd = &s->nodes[n];
  job_set_status(s, t, KB_TASK_PIPELINED);
  if ((tk->node != KB_NONE && tk->node != n) || tk->on_node) return -1;   /* node_info.go:173-182, error returned before the handlers */
  /* node.AddTask with status Pipelined: node_info.go:196-197 Releasing.Sub(Resreq) */
  if (res_sub(&nd->releasing, &tk->resreq, s->R) == KBO_PANIC) { s->panic = 1; return KBO_PANIC; }
  res_add(&nd->used, &tk->resreq, s->R);
  tk->node = n;
  nd->pod_cnt += 1;
  pm_or(nd->ports, tk->port_want, s->Wh);
  nd->nz_cpu += tk->nz_cpu;
  nd->nz_mem += tk->nz_mem;
  tk->node_status = KB_TASK_PIPELINED;
  tk->on_node = 1;
  interpod_placed(s, t, n, 0);
  push_decision(s, t, n, 1);
  fire_allocate_event(s, t);
  return 0;
}

/* ---- OnSessionOpen of drf / proportion, gang's JobValid filter ---- */
static int open_plugins(kbo_session *s) {
  int R = s->R;
  /* framework/session.go:89-107: openSession() calls ssn.JobValid(job) BEFORE OpenSession assigns ssn.Tiers
     (framework/framework.go:31-32) and before any plugin has run OnSessionOpen, so JobValid iterates a nil tier list
     (session_plugins.go:225-240) and returns nil for every job: at this commit the gang JobValid filter
     (gang.go:48-69) never removes a job.  Faithful restatement: every snapshot job stays in ssn.Jobs. */
  for (uint32_t j = 0; j < s->J; j++) s->jobs[j].valid = 1;
  /* drf.go:60-83 */
  res_zero(&s->drf_total);
  for (uint32_t n = 0; n < s->N; n++) res_add(&s->drf_total, &s->nodes[n].allocatable, R);
  for (uint32_t j = 0; j < s->J; j++) {
    o_job *job = &s->jobs[j];
    res_zero(&job->drf_allocated);
    for (uint32_t t = job->t0; t < job->t1; t++)
      if (allocated_status(s->tasks[t].status)) res_add(&job->drf_allocated, &s->tasks[t].resreq, R);
    job->drf_share = drf_calc_share(s, &job->drf_allocated);
  }
  /* proportion.go:58-154 */
  res_zero(&s->prop_total);
  for (uint32_t n = 0; n < s->N; n++) res_add(&s->prop_total, &s->nodes[n].allocatable, R);
  for (uint32_t q = 0; q < s->Q; q++) { o_queue *a = &s->queues[q]; a->has_attr = 0; res_zero(&a->deserved); res_zero(&a->allocated); res_zero(&a->request); a->share = 0; }
  for (uint32_t j = 0; j < s->J; j++) {
    o_job *job = &s->jobs[j];
    if (!job->valid) continue;
    if (job->queue >= s->Q) { /* proportion.go:70-73 reads ssn.Queues[job.Queue].UID: nil pointer when the queue is missing */
      if (s->has_plugin[KB_PLUGIN_PROPORTION]) { s->panic = 1; return KBO_PANIC; }
      continue;
    }
    o_queue *a = &s->queues[job->queue];
    a->has_attr = 1;
    for (uint32_t t = job->t0; t < job->t1; t++) {
      int st = s->tasks[t].status;
      if (allocated_status(st)) { res_add(&a->allocated, &s->tasks[t].resreq, R); res_add(&a->request, &s->tasks[t].resreq, R); }
      else if (st == KB_TASK_PENDING) res_add(&a->request, &s->tasks[t].resreq, R);
    }
  }
  kbo_res remaining = s->prop_total;
  uint8_t *meet = (uint8_t *)calloc(s->Q ? s->Q : 1, 1);
  for (;;) {
    int32_t totalWeight = 0;
    for (uint32_t q = 0; q < s->Q; q++) {
      if (!s->queues[q].has_attr || meet[q]) continue;
      totalWeight += s->queues[q].weight;
    }
    if (totalWeight == 0) break;
    kbo_res increasedDeserved, decreasedDeserved;
    res_zero(&increasedDeserved);
    res_zero(&decreasedDeserved);
    for (uint32_t q = 0; q < s->Q; q++) {
      o_queue *attr = &s->queues[q];
      if (!attr->has_attr || meet[q]) continue;
      kbo_res oldDeserved = attr->deserved;
      kbo_res inc = remaining;
      res_multi(&inc, (double)attr->weight / (double)totalWeight, R);
      res_add(&attr->deserved, &inc, R);
      if (res_less(&attr->request, &attr->deserved, R)) {
        kbo_res m;
        helpers_min(&attr->deserved, &attr->request, &m, R);
        attr->deserved = m;
        meet[q] = 1;
      }
      prop_update_share(s, attr);
      kbo_res increased, decreased;
      res_diff(&attr->deserved, &oldDeserved, &increased, &decreased, R);
      res_add(&increasedDeserved, &increased, R);
      res_add(&decreasedDeserved, &decreased, R);
    }
    if (res_sub(&remaining, &increasedDeserved, R) == KBO_PANIC) { free(meet); s->panic = 1; return KBO_PANIC; }
    res_add(&remaining, &decreasedDeserved, R);
    if (res_is_empty(&remaining, R)) break;
  }
  free(meet);
  return 0;
}

/* ================================================================================================
 * public oracle API
 * ============================================================================================== */
static void fill_res(kbo_res *r, const double *col, uint32_t stride, uint32_t i, int R, uint32_t mask) {
  res_zero(r);
  for (int d = 0; d < R; d++) r->v[d] = col[(size_t)d * stride + i];
  r->mask = mask;
  /* a dense value for a key that is absent from the map must read as 0 */
  for (int d = 2; d < R; d++)
    if (!HAS(r, d)) r->v[d] = 0.0;
}

kbo_session *kbo_open(const kb_config *cfg, const kb_snapshot *sn, int threads) {
  if (!cfg || !sn || sn->n_res < 2 || sn->n_res > KB_MAX_RES) return NULL;
  kbo_session *s = (kbo_session *)calloc(1, sizeof(*s));
  int R = s->R = (int)sn->n_res;
  s->N = sn->n_nodes; s->T = sn->n_tasks; s->J = sn->n_jobs; s->Q = sn->n_queues;
  s->n_tc = sn->n_task_classes; s->n_nc = sn->n_node_classes;
  s->threads = threads;
  s->n_tiers = (int)cfg->n_tiers;
  s->tier_begin = (uint32_t *)malloc(sizeof(uint32_t) * (cfg->n_tiers + 1));
  memcpy(s->tier_begin, cfg->tier_begin, sizeof(uint32_t) * (cfg->n_tiers + 1));
  uint32_t np = cfg->tier_begin[cfg->n_tiers];
  s->plugins = (plug_opt *)malloc(sizeof(plug_opt) * (np ? np : 1));
  s->w_least = 1; s->w_most = 0; s->w_nodeaff = 1; s->w_podaff = 1; s->w_bal = 1; /* nodeorder.go:111-117 */
  for (uint32_t p = 0; p < np; p++) {
    s->plugins[p].plugin = cfg->plugins[p].plugin;
    s->plugins[p].enabled = cfg->plugins[p].enabled;
    memcpy(s->plugins[p].args, cfg->plugins[p].args, sizeof(int32_t) * 8);
    s->plugins[p].args_set = cfg->plugins[p].args_set;
    if (cfg->plugins[p].plugin < 8) s->has_plugin[cfg->plugins[p].plugin] = 1;
    if (cfg->plugins[p].plugin == KB_PLUGIN_NODEORDER) { /* nodeorder.go:119-129 */
      const kb_plugin_option *o = &cfg->plugins[p];
      if (o->args_set & 1u) s->w_least = o->args[0];
      if (o->args_set & 2u) s->w_most = o->args[1];
      if (o->args_set & 4u) s->w_nodeaff = o->args[2];
      if (o->args_set & 8u) s->w_podaff = o->args[3];
      if (o->args_set & 16u) s->w_bal = o->args[4];
    }
  }
  s->pred_enabled = find_plugin_enabled(s, KB_PLUGIN_PREDICATES, KB_EN_PREDICATE);
  s->nodeorder_enabled = find_plugin_enabled(s, KB_PLUGIN_NODEORDER, KB_EN_NODE_ORDER);

  s->nodes = (o_node *)calloc(s->N ? s->N : 1, sizeof(o_node));
  s->Wh = sn->port_words ? sn->port_words : 1;
  s->pm_nodes = (uint64_t *)calloc((size_t)(s->N ? s->N : 1) * s->Wh, sizeof(uint64_t));
  s->pm_base = (uint64_t *)calloc((size_t)(s->N ? s->N : 1) * s->Wh, sizeof(uint64_t));
  s->pm_want = (uint64_t *)calloc((size_t)(s->T ? s->T : 1) * s->Wh, sizeof(uint64_t));
  s->pm_conf = (uint64_t *)calloc((size_t)(s->T ? s->T : 1) * s->Wh, sizeof(uint64_t));
  s->pm_scratch = (uint64_t *)calloc(s->Wh, sizeof(uint64_t));
  for (uint32_t n = 0; n < s->N; n++) {
    o_node *nd = &s->nodes[n];
    uint32_t m = sn->node_scalar_mask ? sn->node_scalar_mask[n] : 0;
    fill_res(&nd->idle, sn->node_idle, s->N, n, R, m);
    fill_res(&nd->allocatable, sn->node_allocatable, s->N, n, R, m);
    /* Releasing starts EmptyResource() and only gains keys through Add (node_info.go:65,154): keys with a non-zero value */
    uint32_t rm = 0;
    for (int d = 2; d < R; d++) if (sn->node_releasing[(size_t)d * s->N + n] != 0.0) rm |= 1u << (d - 2);
    fill_res(&nd->releasing, sn->node_releasing, s->N, n, R, rm);
    nd->allocatable.max_task_num = sn->node_max_pods[n];
    nd->alloc_cpu = sn->node_alloc_cpu[n]; nd->alloc_mem = sn->node_alloc_mem[n];
    nd->nz_cpu = sn->node_nz_cpu[n]; nd->nz_mem = sn->node_nz_mem[n];
    nd->pod_cnt = sn->node_pod_cnt[n];
    nd->cls = sn->node_class ? sn->node_class[n] : 0;
    nd->ports = s->pm_nodes + (size_t)n * s->Wh; nd->base_ports = s->pm_base + (size_t)n * s->Wh;
    if (sn->node_ports) memcpy(nd->ports, sn->node_ports + (size_t)n * s->Wh, sizeof(uint64_t) * s->Wh);
  }
  s->tasks = (o_task *)calloc(s->T ? s->T : 1, sizeof(o_task));
  for (uint32_t t = 0; t < s->T; t++) {
    o_task *tk = &s->tasks[t];
    uint32_t m = sn->task_scalar_mask ? sn->task_scalar_mask[t] : 0;
    fill_res(&tk->resreq, sn->task_resreq, s->T, t, R, m);
    /* InitResreq keys: Resreq's keys plus any key an init container raised (SetMaxResource); dense non-zero => present */
    uint32_t im = m;
    for (int d = 2; d < R; d++) if (sn->task_init_resreq[(size_t)d * s->T + t] != 0.0) im |= 1u << (d - 2);
    fill_res(&tk->init_resreq, sn->task_init_resreq, s->T, t, R, im);
    tk->nz_cpu = sn->task_nz_cpu[t]; tk->nz_mem = sn->task_nz_mem[t];
    tk->job = sn->task_job[t];
    tk->cls = sn->task_class ? sn->task_class[t] : 0;
    tk->evict_protected = sn->task_evict_protected ? sn->task_evict_protected[t] : 0;
    tk->port_want = s->pm_want + (size_t)t * s->Wh; tk->port_conflict = s->pm_conf + (size_t)t * s->Wh;
    if (sn->task_port_want) memcpy(s->pm_want + (size_t)t * s->Wh, sn->task_port_want + (size_t)t * s->Wh, sizeof(uint64_t) * s->Wh);
    if (sn->task_port_conflict) memcpy(s->pm_conf + (size_t)t * s->Wh, sn->task_port_conflict + (size_t)t * s->Wh, sizeof(uint64_t) * s->Wh);
    tk->priority = sn->task_priority[t];
    tk->creation = sn->task_creation[t];
    tk->status = sn->task_status[t];
    tk->node_status = tk->status;
    tk->node = sn->task_node ? sn->task_node[t] : KB_NONE;
    tk->on_node = tk->node != KB_NONE;
  }
  s->jobs = (o_job *)calloc(s->J ? s->J : 1, sizeof(o_job));
  for (uint32_t j = 0; j < s->J; j++) {
    o_job *job = &s->jobs[j];
    job->t0 = sn->job_task_begin[j]; job->t1 = sn->job_task_begin[j + 1];
    job->queue = sn->job_queue[j];
    job->min_available = sn->job_min_available[j];
    job->priority = sn->job_priority[j];
    job->creation = sn->job_creation[j];
    for (uint32_t t = job->t0; t < job->t1; t++) job->cnt[s->tasks[t].status]++;
    heap_init(&job->tasks, task_order_less, s);
  }
  s->queues = (o_queue *)calloc(s->Q ? s->Q : 1, sizeof(o_queue));
  for (uint32_t q = 0; q < s->Q; q++) {
    s->queues[q].weight = sn->queue_weight[q];
    s->queues[q].creation = sn->queue_creation ? sn->queue_creation[q] : 0;
    heap_init(&s->queues[q].jobs, job_order_less, s);
  }
  if (sn->class_compat) {
    size_t nb = ((size_t)s->n_tc * s->n_nc + 7) / 8;
    s->compat = (uint8_t *)malloc(nb);
    memcpy(s->compat, sn->class_compat, nb);
  }
  if (sn->class_affinity) {
    size_t na = (size_t)s->n_tc * s->n_nc;
    s->affinity = (int32_t *)malloc(sizeof(int32_t) * (na ? na : 1));
    memcpy(s->affinity, sn->class_affinity, sizeof(int32_t) * na);
  }
  if (sn->interpod) {
    const kb_interpod *ip = sn->interpod;
    s->ip_on = 1;
    s->ip_C = ip->n_counters; s->ip_D = ip->n_domains ? ip->n_domains : 1; s->ip_P = ip->n_classes; s->ip_S = ip->n_sigs; s->ip_Z = ip->first_unbound_node;
    s->ip_Z0 = s->ip_Z;
    s->ip_unb_n = (int32_t *)calloc(s->N ? s->N : 1, sizeof(int32_t));
    s->ip_Wc = s->ip_C ? (s->ip_C + 63) / 64 : 1; s->ip_Wp = s->ip_P ? (s->ip_P + 63) / 64 : 1;
#define IP_COPY(dst, src, type, count) do { size_t n_ = (size_t)(count); dst = (type *)malloc(sizeof(type) * (n_ ? n_ : 1)); if (n_) memcpy(dst, src, sizeof(type) * n_); } while (0)
    IP_COPY(s->ip_ctr_dom, ip->ctr_dom, uint32_t, (size_t)s->ip_C * s->N);
    IP_COPY(s->ip_ctr_count, ip->ctr_count, int32_t, (size_t)s->ip_C * s->ip_D);
    IP_COPY(s->ip_ctr_total, ip->ctr_total, int32_t, s->ip_C);
    IP_COPY(s->ip_task_inc, ip->task_inc, uint64_t, (size_t)s->T * s->ip_Wc);
    IP_COPY(s->ip_task_forbid, ip->task_forbid, uint64_t, (size_t)s->T * s->ip_Wc);
    IP_COPY(s->ip_task_require, ip->task_require, uint16_t, s->T);
    IP_COPY(s->ip_task_self, ip->task_self, uint8_t, s->T);
    IP_COPY(s->ip_cls_dom, ip->cls_dom, uint32_t, (size_t)s->ip_P * s->N);
    IP_COPY(s->ip_cls_bound, ip->cls_bound, int32_t, (size_t)s->ip_P * s->N);
    IP_COPY(s->ip_cls_unbound, ip->cls_unbound, int32_t, (size_t)s->ip_P * s->N);
    IP_COPY(s->ip_task_cls_inc, ip->task_cls_inc, uint64_t, (size_t)s->T * s->ip_Wp);
    IP_COPY(s->ip_task_sig, ip->task_sig, uint32_t, s->T);
    IP_COPY(s->ip_sig_weight, ip->sig_weight, int32_t, (size_t)s->ip_S * s->ip_P);
#undef IP_COPY
  }
  s->bind_node = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  s->bind_order = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  for (uint32_t t = 0; t < s->T; t++) s->bind_node[t] = KB_NONE;
  if (open_plugins(s) != 0) { /* keep the session so the caller can read ->panic */ }
  return s;
}

static void fast_free(kbo_session *s);
void kbo_close(kbo_session *s) {
  if (!s) return;
  for (uint32_t j = 0; j < s->J; j++) heap_free(&s->jobs[j].tasks);
  for (uint32_t q = 0; q < s->Q; q++) heap_free(&s->queues[q].jobs);
  free(s->pm_nodes); free(s->pm_base); free(s->pm_want); free(s->pm_conf); free(s->pm_scratch);
  free(s->nodes); free(s->tasks); free(s->jobs); free(s->queues); free(s->compat); free(s->affinity); free(s->evictions);
  fast_free(s);
  free(s->journal);
  free(s->ip_ctr_dom); free(s->ip_ctr_count); free(s->ip_ctr_total); free(s->ip_task_inc); free(s->ip_task_forbid); free(s->ip_task_require);
  free(s->ip_task_self); free(s->ip_cls_dom); free(s->ip_cls_bound); free(s->ip_cls_unbound); free(s->ip_task_cls_inc); free(s->ip_task_sig);
  free(s->ip_sig_weight); free(s->ip_unb_n);
  free(s->tier_begin); free(s->plugins); free(s->decisions); free(s->bind_node); free(s->bind_order);
  free(s);
}

/* ================================================================================================
 * FAST MODE of the allocate loop: the same decisions as the faithful loop above, without
 * re-evaluating N nodes per popped task.  Test infrastructure like the rest of this file: it exists to produce golden bind
 * sets for snapshots the faithful mode needs minutes for (config 5: 1M x 50k), and is itself checked against the faithful mode
 * (tests/test_oracle_fast_cpu.py: identical decisions on synthetic and adversarial snapshots, and on the committed full-size
 * digests of configs 3 and 4).
 *   - tasks with equal (InitResreq, non-zero request, static class, host ports) have identical rows: one cached row per SHAPE,
 *     key[n] = feasible ? score + 1 : 0, built with the faithful per-pair functions the first time the shape is popped;
 *   - a placement changes ONE node: its entry is re-evaluated in every cached row (again with the faithful functions);
 *   - SelectBestNode's canonical choice (highest score, lowest index) is the root of a max-tree over the row whose combine
 *     step prefers the left child on ties.
 * Shapes whose class carries preferred node-affinity terms are normalised over the current feasible set: they take the
 * faithful path.  `evals` still counts N per popped task (what the reference does).
 * ============================================================================================== */
typedef struct fast_shape {
  uint32_t rep;       /* a task with this shape */
  double *key;        /* [N] */
  uint32_t *tree;     /* [2 * P] best node of each subtree (KB_NONE: none feasible); leaves at P + n */
} fast_shape;
typedef struct fast_t {
  uint32_t P;               /* leaves of the max-tree: N rounded up to a power of two */
  uint32_t n_shapes, cap;
  fast_shape *shapes;
  uint32_t *task_shape;     /* [T] shape id, KB_NONE until the task is first popped */
  uint32_t *bucket_head;    /* hash -> first shape, chained through next */
  uint32_t *next;
  uint32_t n_buckets;
} fast_t;

static int fast_same_shape(const kbo_session *s, const o_task *a, const o_task *b) {
  if (a->cls != b->cls || a->nz_cpu != b->nz_cpu || a->nz_mem != b->nz_mem || !pm_eq(a->port_want, b->port_want, s->Wh) || !pm_eq(a->port_conflict, b->port_conflict, s->Wh)) return 0;
  if (a->init_resreq.mask != b->init_resreq.mask) return 0;
  for (int d = 0; d < s->R; d++) if (a->init_resreq.v[d] != b->init_resreq.v[d]) return 0;
  return 1;
}
static uint64_t fast_hash(const kbo_session *s, const o_task *t) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ t->cls;
  for (int d = 0; d < s->R; d++) { uint64_t w; memcpy(&w, &t->init_resreq.v[d], 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
  h = (h ^ (uint64_t)t->nz_cpu) * 0xFF51AFD7ED558CCDull; h ^= h >> 29;
  uint64_t pc = 0, pw = 0;
  for (uint32_t w = 0; w < s->Wh; w++) { pc = (pc * 0x9E3779B97F4A7C15ull) ^ t->port_conflict[w]; pw = (pw * 0xC2B2AE3D27D4EB4Full) ^ t->port_want[w]; }
  h = (h ^ (uint64_t)t->nz_mem ^ pc ^ (pw << 1) ^ t->init_resreq.mask) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 32;
  return h;
}
static inline uint32_t fast_better(const double *key, uint32_t l, uint32_t r) {   /* left wins ties: lowest index among equal scores */
  if (l == KB_NONE) return r;
  if (r == KB_NONE) return l;
  return key[r] > key[l] ? r : l;
}
static double fast_key(const kbo_session *s, const o_task *t, const o_node *n) {
  return allocate_predicate(s, t, n) ? node_score(s, t, n) + 1.0 : 0.0;
}
static void fast_free(kbo_session *s) {
  fast_t *f = s->fx;
  if (!f) return;
  for (uint32_t i = 0; i < f->n_shapes; i++) { free(f->shapes[i].key); free(f->shapes[i].tree); }
  free(f->shapes); free(f->task_shape); free(f->bucket_head); free(f->next); free(f);
  s->fx = NULL;
}
static uint32_t fast_shape_of(kbo_session *s, uint32_t t) {
  fast_t *f = s->fx;
  if (!f) {
    f = (fast_t *)calloc(1, sizeof(fast_t));
    f->P = 1; while (f->P < (s->N ? s->N : 1)) f->P <<= 1;
    f->task_shape = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
    for (uint32_t i = 0; i < s->T; i++) f->task_shape[i] = KB_NONE;
    f->n_buckets = 1u << 16;
    f->bucket_head = (uint32_t *)malloc(sizeof(uint32_t) * f->n_buckets);
    for (uint32_t i = 0; i < f->n_buckets; i++) f->bucket_head[i] = KB_NONE;
    s->fx = f;
  }
  if (f->task_shape[t] != KB_NONE) return f->task_shape[t];
  const o_task *tk = &s->tasks[t];
  const uint32_t b = (uint32_t)(fast_hash(s, tk) & (f->n_buckets - 1));
  for (uint32_t i = f->bucket_head[b]; i != KB_NONE; i = f->next[i])
    if (fast_same_shape(s, tk, &s->tasks[f->shapes[i].rep])) return f->task_shape[t] = i;
  if (f->n_shapes == f->cap) {
    f->cap = f->cap ? f->cap * 2 : 64;
    f->shapes = (fast_shape *)realloc(f->shapes, sizeof(fast_shape) * f->cap);
    f->next = (uint32_t *)realloc(f->next, sizeof(uint32_t) * f->cap);
  }
  const uint32_t id = f->n_shapes++;
  fast_shape *sh = &f->shapes[id];
  sh->rep = t;
  sh->key = (double *)malloc(sizeof(double) * (s->N ? s->N : 1));
  sh->tree = (uint32_t *)malloc(sizeof(uint32_t) * 2 * f->P);
  for (uint32_t n = 0; n < s->N; n++) sh->key[n] = fast_key(s, tk, &s->nodes[n]);
  for (uint32_t n = 0; n < f->P; n++) sh->tree[f->P + n] = (n < s->N && sh->key[n] > 0.0) ? n : KB_NONE;
  for (uint32_t i = f->P - 1; i >= 1; i--) sh->tree[i] = fast_better(sh->key, sh->tree[2 * i], sh->tree[2 * i + 1]);
  f->next[id] = f->bucket_head[b];
  f->bucket_head[b] = id;
  return f->task_shape[t] = id;
}
/* node n changed (AddTask): repair its entry in every cached row */
static void fast_node_changed(kbo_session *s, uint32_t n) {
  fast_t *f = s->fx;
  if (!f) return;
  for (uint32_t i = 0; i < f->n_shapes; i++) {
    fast_shape *sh = &f->shapes[i];
    sh->key[n] = fast_key(s, &s->tasks[sh->rep], &s->nodes[n]);
    uint32_t p = f->P + n;
    sh->tree[p] = sh->key[n] > 0.0 ? n : KB_NONE;
    for (p >>= 1; p >= 1; p >>= 1) sh->tree[p] = fast_better(sh->key, sh->tree[2 * p], sh->tree[2 * p + 1]);
  }
}
void kbo_set_fast(kbo_session *s, int on) { s->fast = (on && !s->ip_on) ? 1 : 0; }   /* a placement changes a whole topology domain: the one-node repair does not apply */
uint32_t kbo_fast_shapes(const kbo_session *s) { return s->fx ? s->fx->n_shapes : 0; }

/* actions/allocate/allocate.go:43-194 */
int kbo_allocate(kbo_session *s) {
  if (s->panic) return KBO_PANIC;
  heap_t queues;
  heap_init(&queues, queue_order_less, s);
  for (uint32_t q = 0; q < s->Q; q++) { s->queues[q].has_jobs_heap = 0; s->queues[q].jobs.n = 0; }
  for (uint32_t j = 0; j < s->J; j++) { s->jobs[j].tasks_init = 0; s->jobs[j].tasks.n = 0; }
  /* allocate.go:50-65, ssn.Jobs iterated in ascending JobID (canonical) */
  for (uint32_t j = 0; j < s->J; j++) {
    o_job *job = &s->jobs[j];
    if (!job->valid) continue;
    if (job->queue >= s->Q) continue; /* queue not found */
    heap_push(&queues, job->queue);
    s->queues[job->queue].has_jobs_heap = 1;
    heap_push(&s->queues[job->queue].jobs, j);
  }
  uint8_t *feas = (uint8_t *)malloc(s->N ? s->N : 1);
  double *score = (double *)malloc(sizeof(double) * (s->N ? s->N : 1));
  int rc = 0;
  while (queues.n > 0) {
    uint32_t q = heap_pop(&queues);
    if (ssn_overused(s, q)) continue;                       /* allocate.go:95-98 */
    o_queue *queue = &s->queues[q];
    if (!queue->has_jobs_heap || queue->jobs.n == 0) continue; /* allocate.go:104-107 */
    uint32_t j = heap_pop(&queue->jobs);
    o_job *job = &s->jobs[j];
    if (!job->tasks_init) {                                 /* allocate.go:110-123 */
      job->tasks_init = 1;
      for (uint32_t t = job->t0; t < job->t1; t++) {
        if (s->tasks[t].status != KB_TASK_PENDING) continue;
        if (res_is_empty(&s->tasks[t].resreq, s->R)) continue; /* BestEffort skipped on Resreq */
        heap_push(&job->tasks, t);
      }
    }
    while (job->tasks.n > 0) {                              /* allocate.go:129 */
      if (s->task_limit && s->popped >= s->task_limit) goto done;   /* bounded timing sample, not part of the algorithm */
      uint32_t t = heap_pop(&job->tasks);
      o_task *tk = &s->tasks[t];
      int best = -1;                                        /* SelectBestNode, canonical first max */
      const int fast_row = s->fast && !(s->affinity && s->nodeorder_enabled);   /* NormalizeReduce rows stay faithful */
      if (fast_row) {
        const uint32_t sh = fast_shape_of(s, t);
        const uint32_t root = s->N ? s->fx->shapes[sh].tree[1] : KB_NONE;
        best = root == KB_NONE ? -1 : (int)root;
      } else {
        eval_all_nodes(s, tk, 1, feas, score);              /* PredicateNodes + PrioritizeNodes */
        double maxScore = 0;
        for (uint32_t n = 0; n < s->N; n++) {
          if (!feas[n]) continue;
          if (best < 0 || score[n] > maxScore) { best = (int)n; maxScore = score[n]; }
        }
      }
      s->evals += s->N;
      s->popped++;
      if (best < 0) break;                                  /* allocate.go:144-148 */
      o_node *node = &s->nodes[best];
      if (res_less_equal(&tk->init_resreq, &node->idle, s->R)) {          /* allocate.go:160-166 */
        int e = ssn_allocate(s, t, (uint32_t)best);
        if (e == KBO_PANIC) { rc = KBO_PANIC; goto done; }
      } else {                                              /* allocate.go:167-183 (NodesFitDelta is diagnostic only) */
        if (res_less_equal(&tk->init_resreq, &node->releasing, s->R)) {
          int e = ssn_pipeline(s, t, (uint32_t)best);
          if (e == KBO_PANIC) { rc = KBO_PANIC; goto done; }
        }
      }
      if (s->fast) fast_node_changed(s, (uint32_t)best);
      if (ssn_job_ready(s, job) && job->tasks.n > 0) {      /* allocate.go:185-188 */
        heap_push(&queue->jobs, j);
        break;
      }
    }
    heap_push(&queues, q);                                  /* allocate.go:192 */
  }
done:
  free(feas); free(score);
  heap_free(&queues);
  fast_free(s);   /* the cached rows are valid inside one action only (other actions change nodes behind their back) */
  return rc;
}

/* ================================================================================================
 * preempt (actions/preempt/preempt.go:45-271) with framework.Statement (framework/statement.go:36-220).
 * Canonical orders where the reference ranges over Go maps: queues ascending QueueID, jobs ascending JobID (underRequest),
 * a node's tasks ascending task index.  SortNodes needs no canonicalisation: sort.Reverse over Less(score, then host name)
 * is a strict total order -> descending score, ties by DESCENDING node name (scheduler_helper.go:51-56,174-185).
 * ============================================================================================== */
static void fire_deallocate_event(kbo_session *s, uint32_t t) {   /* drf.go:146-156, proportion.go:224-235 */
  o_task *tk = &s->tasks[t];
  o_job *j = &s->jobs[tk->job];
  if (s->has_plugin[KB_PLUGIN_DRF]) { if (res_sub(&j->drf_allocated, &tk->resreq, s->R) == KBO_PANIC) s->panic = 1; j->drf_share = drf_calc_share(s, &j->drf_allocated); }
  if (s->has_plugin[KB_PLUGIN_PROPORTION]) { o_queue *q = &s->queues[j->queue]; if (res_sub(&q->allocated, &tk->resreq, s->R) == KBO_PANIC) s->panic = 1; prop_update_share(s, q); }
}
/* NodeInfo.RemoveTask (api/node_info.go:217-243): accounting by the status of the node's own clone */
static void node_index_unlink(kbo_session *s, uint32_t t) {
  if (!s->node_head) return;
  const uint32_t n = s->tasks[t].node, nx = s->next_on_node[t], pv = s->prev_on_node[t];
  if (pv == KB_NONE) s->node_head[n] = nx; else s->next_on_node[pv] = nx;
  if (nx != KB_NONE) s->prev_on_node[nx] = pv;
}
static void node_index_link(kbo_session *s, uint32_t t, uint32_t n) {
  if (!s->node_head) return;
  s->prev_on_node[t] = KB_NONE;
  s->next_on_node[t] = s->node_head[n];
  if (s->node_head[n] != KB_NONE) s->prev_on_node[s->node_head[n]] = t;
  s->node_head[n] = t;
}
static void node_index_build(kbo_session *s) {
  free(s->node_head); free(s->next_on_node); free(s->prev_on_node);
  s->node_head = (uint32_t *)malloc(sizeof(uint32_t) * (s->N ? s->N : 1));
  s->next_on_node = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  s->prev_on_node = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  for (uint32_t n = 0; n < s->N; n++) s->node_head[n] = KB_NONE;
  for (uint32_t t = s->T; t-- > 0;)
    if (s->tasks[t].on_node && s->tasks[t].node < s->N) node_index_link(s, t, s->tasks[t].node);
}
static int cmp_u32(const void *a, const void *b) { const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : (x > y ? 1 : 0); }
static void node_remove_task(kbo_session *s, uint32_t t) {
  o_task *tk = &s->tasks[t];
  if (!tk->on_node) return;                     /* node_info.go:220-224 "failed to find task on host": logged, nothing changes */
  o_node *nd = &s->nodes[tk->node];
  switch (tk->node_status) {
    case KB_TASK_RELEASING: if (res_sub(&nd->releasing, &tk->resreq, s->R) == KBO_PANIC) s->panic = 1; res_add(&nd->idle, &tk->resreq, s->R); break;
    case KB_TASK_PIPELINED: res_add(&nd->releasing, &tk->resreq, s->R); break;
    default: res_add(&nd->idle, &tk->resreq, s->R); break;
  }
  /* ni.Used.Sub(task.Resreq): Used is write-only on this path and the snapshot does not carry it (it always covers its own tasks) */
  nd->pod_cnt -= 1;
  nd->nz_cpu -= tk->nz_cpu;
  nd->nz_mem -= tk->nz_mem;
  /* host ports: UsedPorts is rebuilt from the remaining pods; with interned bits that needs the other pods' masks */
  memcpy(nd->ports, nd->base_ports, sizeof(uint64_t) * s->Wh);
  if (s->node_head) {
    for (uint32_t i = s->node_head[tk->node]; i != KB_NONE; i = s->next_on_node[i]) if (i != t) pm_or(nd->ports, s->tasks[i].port_want, s->Wh);
  } else {
    for (uint32_t i = 0; i < s->T; i++)
      if (i != t && s->tasks[i].node == tk->node && s->tasks[i].on_node) pm_or(nd->ports, s->tasks[i].port_want, s->Wh);
  }
  node_index_unlink(s, t);
  tk->on_node = 0;
}
/* NodeInfo.AddTask (api/node_info.go:172-212) for a task whose session status is already `status` */
static int node_add_task(kbo_session *s, uint32_t t, uint32_t n, int status) {
  o_task *tk = &s->tasks[t];
  o_node *nd = &s->nodes[n];
  if ((tk->node != KB_NONE && tk->node != n) || tk->on_node) return -1;   /* node_info.go:173-182: NodeName is sticky */
  switch (status) {
    case KB_TASK_RELEASING:
      if (!res_less_equal(&tk->resreq, &nd->idle, s->R)) return -1;
      if (res_sub(&nd->idle, &tk->resreq, s->R) == KBO_PANIC) { s->panic = 1; return KBO_PANIC; }
      res_add(&nd->releasing, &tk->resreq, s->R);
      break;
    case KB_TASK_PIPELINED:
      if (res_sub(&nd->releasing, &tk->resreq, s->R) == KBO_PANIC) { s->panic = 1; return KBO_PANIC; }
      break;
    default:
      if (!res_less_equal(&tk->resreq, &nd->idle, s->R)) return -1;
      if (res_sub(&nd->idle, &tk->resreq, s->R) == KBO_PANIC) { s->panic = 1; return KBO_PANIC; }
      break;
  }
  res_add(&nd->used, &tk->resreq, s->R);
  tk->node = n;
  tk->node_status = (uint8_t)status;
  tk->on_node = 1;
  node_index_link(s, t, n);
  nd->pod_cnt += 1;
  nd->nz_cpu += tk->nz_cpu;
  nd->nz_mem += tk->nz_mem;
  pm_or(nd->ports, tk->port_want, s->Wh);
  return 0;
}
/* NodeInfo.UpdateTask (node_info.go:245-256) = RemoveTask + AddTask; an AddTask error there is glog.Fatalf — the process dies,
   which this restatement reports like a panic (reachable only when Idle has drifted below -epsilon, e.g. through sub-epsilon
   scalar requests that LessEqual skips, resource_info.go:283-287) */
static void node_update_task(kbo_session *s, uint32_t t, int status) {
  if (!s->tasks[t].on_node) return;            /* RemoveTask error: UpdateTask returns it, nothing changes */
  node_remove_task(s, t);
  if (node_add_task(s, t, s->tasks[t].node, status) != 0) s->panic = 1;
}
static void journal_push(kbo_session *s, uint32_t op, uint32_t task, uint32_t node, uint32_t stmt) {
  if (s->n_journal == s->cap_journal) { s->cap_journal = s->cap_journal ? s->cap_journal * 2 : 64; s->journal = (kb_stmt_op *)realloc(s->journal, sizeof(kb_stmt_op) * s->cap_journal); }
  kb_stmt_op e = {op, task, node, stmt};
  s->journal[s->n_journal++] = e;
}
static void pfast_node_changed(kbo_session *s, uint32_t n);
typedef struct stmt_op { uint8_t kind; uint32_t task; } stmt_op;   /* 0 evict, 1 pipeline */
typedef struct stmt_t { stmt_op *ops; size_t n, cap; } stmt_t;
static void stmt_push(stmt_t *st, uint8_t kind, uint32_t task) {
  if (st->n == st->cap) { st->cap = st->cap ? st->cap * 2 : 16; st->ops = (stmt_op *)realloc(st->ops, sizeof(stmt_op) * st->cap); }
  st->ops[st->n].kind = kind; st->ops[st->n].task = task; st->n++;
}
static void stmt_evict(kbo_session *s, stmt_t *st, uint32_t t) {          /* statement.go:36-69 */
  s->mutations++;
  journal_push(s, KB_OP_EVICT, t, s->tasks[t].node, s->stmt_no);
  interpod_allocated_status(s, t, -1);
  job_set_status(s, t, KB_TASK_RELEASING);
  node_update_task(s, t, KB_TASK_RELEASING);
  fire_deallocate_event(s, t);
  stmt_push(st, 0, t);
}
static void stmt_unevict(kbo_session *s, uint32_t t) {                     /* statement.go:83-110 */
  s->mutations++;
  job_set_status(s, t, KB_TASK_RUNNING);
  node_update_task(s, t, KB_TASK_RUNNING);
  interpod_allocated_status(s, t, +1);
  fire_allocate_event(s, t);
}
static void stmt_pipeline(kbo_session *s, stmt_t *st, uint32_t t, uint32_t n) {   /* statement.go:113-150 */
  s->mutations++;
  journal_push(s, KB_OP_PIPELINE, t, n, s->stmt_no);
  job_set_status(s, t, KB_TASK_PIPELINED);
  if (node_add_task(s, t, n, KB_TASK_PIPELINED) == 0) { pfast_node_changed(s, n); interpod_placed(s, t, n, 0); }
  fire_allocate_event(s, t);
  stmt_push(st, 1, t);
}
static void stmt_unpipeline(kbo_session *s, uint32_t t) {                  /* statement.go:155-190 */
  s->mutations++;
  job_set_status(s, t, KB_TASK_PENDING);
  if (s->tasks[t].on_node) { pfast_node_changed(s, s->tasks[t].node); interpod_unpipelined(s, t, s->tasks[t].node); }
  node_remove_task(s, t);                        /* task.NodeName keeps the old host (node_info.go:217-243 never clears it) */
  fire_deallocate_event(s, t);
}
static void stmt_discard(kbo_session *s, stmt_t *st) {                     /* statement.go:193-205: newest first */
  if (st->n) journal_push(s, KB_OP_DISCARD, KB_NONE, KB_NONE, s->stmt_no);
  for (size_t i = st->n; i-- > 0;) {
    if (st->ops[i].kind == 0) stmt_unevict(s, st->ops[i].task);
    else stmt_unpipeline(s, st->ops[i].task);
  }
  st->n = 0;
}
static void stmt_commit(kbo_session *s, stmt_t *st) {                      /* statement.go:208-220: evict -> cache.Evict */
  if (st->n) journal_push(s, KB_OP_COMMIT, KB_NONE, KB_NONE, s->stmt_no);
  for (size_t i = 0; i < st->n; i++) {
    if (st->ops[i].kind != 0) continue;
    if (s->n_evict == s->cap_evict) { s->cap_evict = s->cap_evict ? s->cap_evict * 2 : 64; s->evictions = (uint32_t *)realloc(s->evictions, sizeof(uint32_t) * s->cap_evict); }
    s->evictions[s->n_evict++] = st->ops[i].task;
  }
  st->n = 0;
}
/* session_plugins.go:202-222 + gang.go:126-129 */
static int ssn_job_pipelined(const kbo_session *s, const o_job *j) {
  if (find_plugin_enabled(s, KB_PLUGIN_GANG, KB_EN_JOB_PIPELINED)) return j->cnt[KB_TASK_PIPELINED] + job_ready_num(j) >= j->min_available;
  return 1;
}
/* session_plugins.go:121-162: per tier the intersection of the enabled plugins' candidates; the first tier that leaves a
   non-empty set decides (a nil slice is an empty one).  Returns the number of victims written (input order kept). */
static size_t ssn_evictable(kbo_session *s, uint32_t preemptor, const uint32_t *preemptees, size_t n, uint32_t *victims, int reclaim) {
  const uint32_t en_bit = reclaim ? KB_EN_RECLAIMABLE : KB_EN_PREEMPTABLE;   /* session_plugins.go:80-118 is the same code with EnabledReclaimable */
  int init = 0;
  size_t nv = 0;
  uint8_t *keep = (uint8_t *)malloc(n ? n : 1);
  for (int t = 0; t < s->n_tiers; t++) {
    for (uint32_t p = s->tier_begin[t]; p < s->tier_begin[t + 1]; p++) {
      const plug_opt *po = &s->plugins[p];
      if (!(po->enabled & en_bit)) continue;
      memset(keep, 0, n ? n : 1);
      if (po->plugin == KB_PLUGIN_CONFORMANCE) {            /* conformance.go:44-58 */
        for (size_t i = 0; i < n; i++) keep[i] = !s->tasks[preemptees[i]].evict_protected;
      } else if (po->plugin == KB_PLUGIN_GANG) {            /* gang.go:71-90 */
        for (size_t i = 0; i < n; i++) {
          const o_job *job = &s->jobs[s->tasks[preemptees[i]].job];
          keep[i] = (job->min_available <= job_ready_num(job) - 1) || job->min_available == 1;
        }
      } else if (po->plugin == KB_PLUGIN_PRIORITY && !reclaim) {        /* priority.go:81-98 (preemptable only) */
        const o_job *pj = &s->jobs[s->tasks[preemptor].job];
        for (size_t i = 0; i < n; i++) keep[i] = s->jobs[s->tasks[preemptees[i]].job].priority < pj->priority;
      } else if (po->plugin == KB_PLUGIN_PROPORTION && reclaim) {   /* proportion.go:171-196: running per-queue allocation, in reclaimee order */
        kbo_res *alloc = (kbo_res *)malloc(sizeof(kbo_res) * (n ? n : 1));
        uint32_t *aq = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
        size_t na = 0;
        for (size_t i = 0; i < n; i++) {
          const uint32_t q = s->jobs[s->tasks[preemptees[i]].job].queue;
          size_t a = 0;
          while (a < na && aq[a] != q) a++;
          if (a == na) { aq[na] = q; alloc[na] = s->queues[q].allocated; na++; }
          if (res_less(&alloc[a], &s->tasks[preemptees[i]].resreq, s->R)) continue;
          if (res_sub(&alloc[a], &s->tasks[preemptees[i]].resreq, s->R) == KBO_PANIC) s->panic = 1;
          keep[i] = res_less_equal(&s->queues[q].deserved, &alloc[a], s->R);
        }
        free(alloc); free(aq);
      } else if (po->plugin == KB_PLUGIN_DRF && !reclaim) {             /* drf.go:84-109: running per-job allocation, in preemptee order */
        const o_job *pj = &s->jobs[s->tasks[preemptor].job];
        kbo_res lalloc = pj->drf_allocated;
        res_add(&lalloc, &s->tasks[preemptor].resreq, s->R);
        const double ls = drf_calc_share(s, &lalloc);
        kbo_res *alloc = (kbo_res *)malloc(sizeof(kbo_res) * (n ? n : 1));   /* allocations[job], keyed by first occurrence */
        uint32_t *ajob = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
        size_t na = 0;
        for (size_t i = 0; i < n; i++) {
          const uint32_t jb = s->tasks[preemptees[i]].job;
          size_t a = 0;
          while (a < na && ajob[a] != jb) a++;
          if (a == na) { ajob[na] = jb; alloc[na] = s->jobs[jb].drf_allocated; na++; }
          if (res_sub(&alloc[a], &s->tasks[preemptees[i]].resreq, s->R) == KBO_PANIC) s->panic = 1;
          const double rs = drf_calc_share(s, &alloc[a]);
          keep[i] = (ls < rs) || (fabs(ls - rs) <= 0.000001);   /* shareDelta (drf.go:33) */
        }
        free(alloc); free(ajob);
      } else {
        continue;   /* predicates / proportion / nodeorder register no preemptable fn */
      }
      if (!init) {
        nv = 0;
        for (size_t i = 0; i < n; i++) if (keep[i]) victims[nv++] = preemptees[i];
        init = 1;
      } else {
        size_t w = 0;
        for (size_t v = 0; v < nv; v++) {
          int in = 0;
          for (size_t i = 0; i < n && !in; i++) in = keep[i] && preemptees[i] == victims[v];
          if (in) victims[w++] = victims[v];
        }
        nv = w;
      }
    }
    if (nv > 0) break;
  }
  free(keep);
  return nv;
}
static int victim_less(void *ctx, uint32_t l, uint32_t r) { return !task_order_less(ctx, l, r); }   /* preempt.go:223-225 */
static const double *g_sort_score;
static int sort_nodes_cmp(const void *a, const void *b) {
  const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  if (g_sort_score[x] != g_sort_score[y]) return g_sort_score[x] > g_sort_score[y] ? -1 : 1;
  return x > y ? -1 : (x < y ? 1 : 0);
}
/* the body of preempt()'s loop over selectedNodes for ONE node (preempt.go:192-251): 1 = the preemptor was pipelined here.
   mode 0: victims are Running tasks of OTHER jobs in the preemptor job's queue (:112-124); mode 1: of the same job (:150-157) */
static int preempt_try_node(kbo_session *s, stmt_t *st, uint32_t preemptor, int mode, uint32_t n, uint32_t *pre, uint32_t *vic) {
  o_task *pt = &s->tasks[preemptor];
  const o_job *pj = &s->jobs[pt->job];
  size_t np_ = 0;
  for (uint32_t t = s->node_head[n]; t != KB_NONE; t = s->next_on_node[t]) {   /* node.Tasks, filtered */
    const o_task *tk = &s->tasks[t];
    if (tk->node_status != KB_TASK_RUNNING) continue;
    if (mode == 0) { if (!(s->jobs[tk->job].queue == pj->queue && tk->job != pt->job)) continue; }
    else if (tk->job != pt->job) continue;
    pre[np_++] = t;
  }
  if (np_ == 0) return 0;                      /* ssn.Preemptable of nothing is nothing: validateVictims "no victims" */
  qsort(pre, np_, sizeof(uint32_t), cmp_u32);   /* canonical order: ascending task index */
  size_t nv = ssn_evictable(s, preemptor, pre, np_, vic, 0);
  if (nv == 0) return 0;                       /* validateVictims: "no victims" */
  kbo_res all; res_zero(&all);
  for (size_t i = 0; i < nv; i++) res_add(&all, &s->tasks[vic[i]].resreq, s->R);
  if (!res_less_equal(&pt->init_resreq, &all, s->R)) return 0;   /* "not enough resources" */
  heap_t vq; heap_init(&vq, victim_less, s);
  for (size_t i = 0; i < nv; i++) heap_push(&vq, vic[i]);
  kbo_res preempted; res_zero(&preempted);
  while (vq.n > 0) {                          /* lowest priority first (preempt.go:229-241) */
    uint32_t v = heap_pop(&vq);
    stmt_evict(s, st, v);
    res_add(&preempted, &s->tasks[v].resreq, s->R);
    if (res_less_equal(&pt->init_resreq, &preempted, s->R)) break;
  }
  heap_free(&vq);
  if (res_less_equal(&pt->init_resreq, &preempted, s->R)) {      /* preempt.go:247-256 */
    stmt_pipeline(s, st, preemptor, n);
    return 1;
  }
  return 0;
}
/* preempt(): preempt.go:171-254, the faithful form: every node evaluated, sorted and visited for every preemptor */
static int preempt_one(kbo_session *s, stmt_t *st, uint32_t preemptor, int mode, uint8_t *feas, double *score, uint32_t *order) {
  o_task *pt = &s->tasks[preemptor];
  eval_all_nodes(s, pt, 0, feas, score);       /* PredicateNodes with ssn.PredicateFn only, PrioritizeNodes */
  s->evals += s->N;
  uint32_t nf = 0;
  for (uint32_t n = 0; n < s->N; n++) if (feas[n]) order[nf++] = n;
  /* SortNodes: descending score, ties by descending host name (= descending canonical index); a strict total order, so any
     comparison sort yields the one sequence */
  g_sort_score = score;
  qsort(order, nf, sizeof(uint32_t), sort_nodes_cmp);
  uint32_t *pre = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  uint32_t *vic = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  int assigned = 0;
  for (uint32_t oi = 0; oi < nf && !assigned; oi++) assigned = preempt_try_node(s, st, preemptor, mode, order[oi], pre, vic);
  free(pre); free(vic);
  return assigned;
}

/* ================================================================================================
 * FAST MODE of preempt (kbo_set_fast; the golden journal of BASELINE configs[4], 1M tasks x 50k nodes, where the faithful
 * preempt_one above — N evaluations, a sort and a walk over every node for each of a million preemptors — needs hours).
 * Derived from preempt.go:171-254 and scheduler_helper.go:51-56,174-185 alone; three facts about that code, each exact:
 *  (1) A node contributes nothing to preempt()'s loop unless node.Tasks holds a Running task the filter accepts (an empty
 *      preemptee list makes ssn.Preemptable return no victims -> validateVictims "no victims" -> continue).  Both filters accept
 *      only Running tasks of the preemptor's queue.  Inside the action a task is Running on a node only if it was there when
 *      the action started (Evict makes it Releasing, the undo of a discarded statement puts it back on the same node), so the
 *      nodes that held a Running task of queue q at the start are a superset of the nodes that can matter for q's preemptors:
 *      only those are visited, in SortNodes order.
 *  (2) The order — descending score, ties by descending node name — is a function of the plugin predicates and the scorers,
 *      which read the node's pod count, non-zero request sums and host ports (predicates.go:127,181-190; nodeorder), not
 *      Idle / Releasing.  Evict + its undo leave those as they were (UpdateTask = RemoveTask + AddTask of the same pod); only
 *      Statement.Pipeline and its undo change them, for one node.  So the sorted candidate list of (preemptor shape, queue) is
 *      cached and rebuilt when a node of that queue's set was pipelined onto or un-pipelined since (a change log).  Shape = what
 *      plugin_predicate and node_score read of the task: static class, non-zero request, conflicting ports.
 *  (3) preempt() is a function of (session state, preemptor's job, shape, InitResreq, Resreq, filter).  If it ended without a
 *      single Statement operation for one preemptor, and no operation or undo has happened since, it ends the same way for the
 *      next preemptor of the same job with the same shape and requests (the tasks of a gang): answered without a walk.
 * Sessions whose scores are normalised over the feasible set (preferred node affinity) keep the faithful path, like allocate.
 * `evals` counts N per preemptor in both modes.  tests/test_oracle_fast_cpu.py holds this mode to the faithful one.
 * ============================================================================================== */
typedef struct pfast_list { uint32_t *nodes; uint32_t n; uint64_t built_at; int valid; } pfast_list;
typedef struct pfast_shape { uint32_t cls; int64_t nz_cpu, nz_mem; const uint64_t *port_conflict; /* [Wh], the representative's */ uint32_t rep; pfast_list *lists; /* [Q] */ } pfast_shape;
typedef struct pfast_t {
  uint32_t *qn_begin, *qn_nodes;   /* per queue: nodes that held a Running task of it when the action started, ascending */
  uint8_t *qn_bits;                /* [Q][ceil(N/8)] the same as a bitmap */
  size_t row_bytes;
  uint32_t *changed; uint64_t n_changed, cap_changed;   /* nodes a Pipeline or its undo touched, in order */
  pfast_shape *shapes; uint32_t n_shapes, cap_shapes;
  uint32_t *task_shape;            /* [T] */
  double *key;                     /* [N] scratch: score of the candidates being sorted */
  /* (3): the last preemptor that left no trace */
  uint32_t memo_task; int memo_mode; uint64_t memo_mut; int memo_valid;
} pfast_t;
static void pfast_free(kbo_session *s) {
  pfast_t *f = s->px;
  if (!f) return;
  for (uint32_t i = 0; i < f->n_shapes; i++) {
    for (uint32_t q = 0; q < s->Q; q++) free(f->shapes[i].lists[q].nodes);
    free(f->shapes[i].lists);
  }
  free(f->qn_begin); free(f->qn_nodes); free(f->qn_bits); free(f->changed); free(f->shapes); free(f->task_shape); free(f->key); free(f);
  s->px = NULL;
}
static void pfast_init(kbo_session *s) {
  pfast_t *f = (pfast_t *)calloc(1, sizeof(pfast_t));
  const uint32_t N = s->N, Q = s->Q;
  f->row_bytes = ((size_t)N + 7) / 8;
  f->qn_bits = (uint8_t *)calloc((size_t)(Q ? Q : 1) * (f->row_bytes ? f->row_bytes : 1), 1);
  f->qn_begin = (uint32_t *)calloc((size_t)Q + 2, sizeof(uint32_t));
  for (uint32_t t = 0; t < s->T; t++) {
    const o_task *tk = &s->tasks[t];
    if (!tk->on_node || tk->node >= N || tk->node_status != KB_TASK_RUNNING) continue;
    const uint32_t q = s->jobs[tk->job].queue;
    if (q >= Q) continue;
    uint8_t *b = &f->qn_bits[(size_t)q * f->row_bytes + (tk->node >> 3)];
    if (!(*b & (1u << (tk->node & 7)))) { *b |= (uint8_t)(1u << (tk->node & 7)); f->qn_begin[q + 1]++; }
  }
  for (uint32_t q = 0; q < Q; q++) f->qn_begin[q + 1] += f->qn_begin[q];
  f->qn_nodes = (uint32_t *)malloc(sizeof(uint32_t) * (f->qn_begin[Q] ? f->qn_begin[Q] : 1));
  for (uint32_t q = 0; q < Q; q++) {
    uint32_t w = f->qn_begin[q];
    const uint8_t *row = &f->qn_bits[(size_t)q * f->row_bytes];
    for (uint32_t n = 0; n < N; n++) if (row[n >> 3] & (1u << (n & 7))) f->qn_nodes[w++] = n;
  }
  f->task_shape = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  for (uint32_t t = 0; t < s->T; t++) f->task_shape[t] = KB_NONE;
  f->key = (double *)malloc(sizeof(double) * (N ? N : 1));
  s->px = f;
}
static void pfast_node_changed(kbo_session *s, uint32_t n) {
  pfast_t *f = s->px;
  if (!f) return;
  if (f->n_changed == f->cap_changed) { f->cap_changed = f->cap_changed ? f->cap_changed * 2 : 256; f->changed = (uint32_t *)realloc(f->changed, sizeof(uint32_t) * f->cap_changed); }
  f->changed[f->n_changed++] = n;
}
static uint32_t pfast_shape_of(kbo_session *s, uint32_t t) {
  pfast_t *f = s->px;
  if (f->task_shape[t] != KB_NONE) return f->task_shape[t];
  const o_task *tk = &s->tasks[t];
  if (t > 0 && f->task_shape[t - 1] != KB_NONE) {          /* the tasks of a job mostly repeat their predecessor */
    const pfast_shape *p = &f->shapes[f->task_shape[t - 1]];
    if (p->cls == tk->cls && p->nz_cpu == tk->nz_cpu && p->nz_mem == tk->nz_mem && pm_eq(p->port_conflict, tk->port_conflict, s->Wh)) return f->task_shape[t] = f->task_shape[t - 1];
  }
  for (uint32_t i = 0; i < f->n_shapes; i++) {
    const pfast_shape *p = &f->shapes[i];
    if (p->cls == tk->cls && p->nz_cpu == tk->nz_cpu && p->nz_mem == tk->nz_mem && pm_eq(p->port_conflict, tk->port_conflict, s->Wh)) return f->task_shape[t] = i;
  }
  if (f->n_shapes == f->cap_shapes) { f->cap_shapes = f->cap_shapes ? f->cap_shapes * 2 : 64; f->shapes = (pfast_shape *)realloc(f->shapes, sizeof(pfast_shape) * f->cap_shapes); }
  pfast_shape *p = &f->shapes[f->n_shapes];
  p->cls = tk->cls; p->nz_cpu = tk->nz_cpu; p->nz_mem = tk->nz_mem; p->port_conflict = tk->port_conflict; p->rep = t;
  p->lists = (pfast_list *)calloc(s->Q ? s->Q : 1, sizeof(pfast_list));
  return f->task_shape[t] = f->n_shapes++;
}
/* SortNodes over the queue's node set for this shape against the CURRENT node state */
static const pfast_list *pfast_list_of(kbo_session *s, uint32_t shape, uint32_t q) {
  pfast_t *f = s->px;
  pfast_list *l = &f->shapes[shape].lists[q];
  if (l->valid) {
    const uint8_t *row = &f->qn_bits[(size_t)q * f->row_bytes];
    for (uint64_t i = l->built_at; i < f->n_changed && l->valid; i++) {
      const uint32_t n = f->changed[i];
      if (row[n >> 3] & (1u << (n & 7))) l->valid = 0;
    }
    l->built_at = f->n_changed;
    if (l->valid) return l;
  }
  const uint32_t a = f->qn_begin[q], b = f->qn_begin[q + 1];
  if (!l->nodes) l->nodes = (uint32_t *)malloc(sizeof(uint32_t) * (b - a ? b - a : 1));
  const o_task *rep = &s->tasks[f->shapes[shape].rep];
  uint32_t nf = 0;
  for (uint32_t i = a; i < b; i++) {
    const uint32_t n = f->qn_nodes[i];
    if (!plugin_predicate(s, rep, &s->nodes[n])) continue;
    f->key[n] = node_score(s, rep, &s->nodes[n]);
    l->nodes[nf++] = n;
  }
  g_sort_score = f->key;
  qsort(l->nodes, nf, sizeof(uint32_t), sort_nodes_cmp);
  l->n = nf; l->built_at = f->n_changed; l->valid = 1;
  return l;
}
static int pfast_same_request(const kbo_session *s, const o_task *a, const o_task *b) {
  if (a->resreq.mask != b->resreq.mask || a->init_resreq.mask != b->init_resreq.mask) return 0;
  for (int d = 0; d < s->R; d++) if (a->resreq.v[d] != b->resreq.v[d] || a->init_resreq.v[d] != b->init_resreq.v[d]) return 0;
  return 1;
}
static int preempt_one_fast(kbo_session *s, stmt_t *st, uint32_t preemptor, int mode, uint32_t *pre, uint32_t *vic) {
  pfast_t *f = s->px;
  const o_task *pt = &s->tasks[preemptor];
  s->evals += s->N;
  const uint32_t q = s->jobs[pt->job].queue;
  const uint32_t shape = pfast_shape_of(s, preemptor);
  if (f->memo_valid && f->memo_mut == s->mutations && f->memo_mode == mode) {          /* (3) */
    const o_task *m = &s->tasks[f->memo_task];
    if (m->job == pt->job && f->task_shape[f->memo_task] == shape && pfast_same_request(s, m, pt)) return 0;
  }
  const uint64_t before = s->mutations;
  const pfast_list *l = pfast_list_of(s, shape, q);
  int assigned = 0;
  for (uint32_t i = 0; i < l->n && !assigned; i++) assigned = preempt_try_node(s, st, preemptor, mode, l->nodes[i], pre, vic);
  if (!assigned && s->mutations == before) { f->memo_valid = 1; f->memo_task = preemptor; f->memo_mode = mode; f->memo_mut = before; }
  return assigned;
}

int kbo_preempt(kbo_session *s) {
  if (s->panic) return KBO_PANIC;
  node_index_build(s);
  for (uint32_t n = 0; n < s->N; n++) {        /* ports of pods outside the session stay on the node whatever moves */
    uint64_t *mine = s->pm_scratch;
    memset(mine, 0, sizeof(uint64_t) * s->Wh);
    for (uint32_t t = s->node_head[n]; t != KB_NONE; t = s->next_on_node[t]) pm_or(mine, s->tasks[t].port_want, s->Wh);
    for (uint32_t w = 0; w < s->Wh; w++) s->nodes[n].base_ports[w] = s->nodes[n].ports[w] & ~mine[w];
  }
  const int fast = s->fast && !(s->affinity && s->nodeorder_enabled) && !s->ip_on;   /* NormalizeReduce over the feasible set / inter-pod counters that evictions change: faithful */
  uint8_t *feas = (uint8_t *)malloc(s->N ? s->N : 1);
  double *score = (double *)malloc(sizeof(double) * (s->N ? s->N : 1));
  uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * (s->N ? s->N : 1));
  uint32_t *pre = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  uint32_t *vic = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  if (fast) pfast_init(s);
  heap_t *qjobs = (heap_t *)calloc(s->Q ? s->Q : 1, sizeof(heap_t));     /* preemptorsMap */
  heap_t *jtasks = (heap_t *)calloc(s->J ? s->J : 1, sizeof(heap_t));    /* preemptorTasks */
  uint8_t *qseen = (uint8_t *)calloc(s->Q ? s->Q : 1, 1), *under = (uint8_t *)calloc(s->J ? s->J : 1, 1);
  for (uint32_t q = 0; q < s->Q; q++) heap_init(&qjobs[q], job_order_less, s);
  for (uint32_t j = 0; j < s->J; j++) heap_init(&jtasks[j], task_order_less, s);
  for (uint32_t j = 0; j < s->J; j++) {                                   /* preempt.go:55-76 */
    o_job *job = &s->jobs[j];
    if (!job->valid || job->queue >= s->Q) continue;
    qseen[job->queue] = 1;
    if (job->cnt[KB_TASK_PENDING] != 0) {
      heap_push(&qjobs[job->queue], j);
      under[j] = 1;
      for (uint32_t t = job->t0; t < job->t1; t++) if (s->tasks[t].status == KB_TASK_PENDING) heap_push(&jtasks[j], t);
    }
  }
  stmt_t st = {0};
  s->stmt_no = 0;
  for (uint32_t q = 0; q < s->Q; q++) {
    if (!qseen[q]) continue;
    for (;;) {                                                             /* between jobs within the queue (preempt.go:80-139) */
      if (qjobs[q].n == 0) break;
      uint32_t pj = heap_pop(&qjobs[q]);
      int assigned = 0;
      st.n = 0;
      s->stmt_no++;                                                        /* stmt := ssn.Statement() (preempt.go:91) */
      for (;;) {
        if (jtasks[pj].n == 0) break;
        uint32_t preemptor = heap_pop(&jtasks[pj]);
        s->popped++;
        if (fast ? preempt_one_fast(s, &st, preemptor, 0, pre, vic) : preempt_one(s, &st, preemptor, 0, feas, score, order)) assigned = 1;
        if (ssn_job_pipelined(s, &s->jobs[pj])) { stmt_commit(s, &st); break; }
      }
      if (!ssn_job_pipelined(s, &s->jobs[pj])) { stmt_discard(s, &st); continue; }
      if (assigned) heap_push(&qjobs[q], pj);
    }
    for (uint32_t j = 0; j < s->J; j++) {                                  /* between tasks within a job (preempt.go:142-166) */
      if (!under[j]) continue;
      for (;;) {
        if (jtasks[j].n == 0) break;
        uint32_t preemptor = heap_pop(&jtasks[j]);
        s->popped++;
        st.n = 0;
        s->stmt_no++;                                                      /* preempt.go:153 */
        int assigned = fast ? preempt_one_fast(s, &st, preemptor, 1, pre, vic) : preempt_one(s, &st, preemptor, 1, feas, score, order);
        stmt_commit(s, &st);
        if (!assigned) break;
      }
    }
  }
  for (uint32_t q = 0; q < s->Q; q++) heap_free(&qjobs[q]);
  for (uint32_t j = 0; j < s->J; j++) heap_free(&jtasks[j]);
  free(qjobs); free(jtasks); free(qseen); free(under); free(st.ops); free(feas); free(score); free(order); free(pre); free(vic);
  pfast_free(s);
  free(s->node_head); free(s->next_on_node); free(s->prev_on_node);
  s->node_head = s->next_on_node = s->prev_on_node = NULL;   /* the other actions add tasks without the index */
  return s->panic ? KBO_PANIC : 0;
}
/* ================================================================================================
 * reclaim (actions/reclaim/reclaim.go:40-193): across queues, no Statement — ssn.Evict (framework/session.go:317-354) and
 * ssn.Pipeline act immediately.  Canonical orders: jobs ascending JobID, nodes ascending name, a node's tasks ascending index.
 * ============================================================================================== */
static void record_eviction(kbo_session *s, uint32_t t) {
  if (s->n_evict == s->cap_evict) { s->cap_evict = s->cap_evict ? s->cap_evict * 2 : 64; s->evictions = (uint32_t *)realloc(s->evictions, sizeof(uint32_t) * s->cap_evict); }
  s->evictions[s->n_evict++] = t;
}
int kbo_reclaim(kbo_session *s) {
  if (s->panic) return KBO_PANIC;
  for (uint32_t n = 0; n < s->N; n++) {
    uint64_t *mine = s->pm_scratch;
    memset(mine, 0, sizeof(uint64_t) * s->Wh);
    for (uint32_t t = 0; t < s->T; t++) if (s->tasks[t].on_node && s->tasks[t].node == n) pm_or(mine, s->tasks[t].port_want, s->Wh);
    for (uint32_t w = 0; w < s->Wh; w++) s->nodes[n].base_ports[w] = s->nodes[n].ports[w] & ~mine[w];
  }
  heap_t queues; heap_init(&queues, queue_order_less, s);
  heap_t *qjobs = (heap_t *)calloc(s->Q ? s->Q : 1, sizeof(heap_t));
  heap_t *jtasks = (heap_t *)calloc(s->J ? s->J : 1, sizeof(heap_t));
  uint8_t *qseen = (uint8_t *)calloc(s->Q ? s->Q : 1, 1);
  for (uint32_t q = 0; q < s->Q; q++) heap_init(&qjobs[q], job_order_less, s);
  for (uint32_t j = 0; j < s->J; j++) heap_init(&jtasks[j], task_order_less, s);
  for (uint32_t j = 0; j < s->J; j++) {                                   /* reclaim.go:54-81 */
    o_job *job = &s->jobs[j];
    if (!job->valid || job->queue >= s->Q) continue;
    if (!qseen[job->queue]) { qseen[job->queue] = 1; heap_push(&queues, job->queue); }
    if (job->cnt[KB_TASK_PENDING] != 0) {
      heap_push(&qjobs[job->queue], j);
      for (uint32_t t = job->t0; t < job->t1; t++) if (s->tasks[t].status == KB_TASK_PENDING) heap_push(&jtasks[j], t);
    }
  }
  uint32_t *pre = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  uint32_t *vic = (uint32_t *)malloc(sizeof(uint32_t) * (s->T ? s->T : 1));
  while (queues.n > 0) {                                                   /* reclaim.go:83-190 */
    uint32_t q = heap_pop(&queues);
    if (ssn_overused(s, q)) continue;
    if (qjobs[q].n == 0) continue;
    uint32_t j = heap_pop(&qjobs[q]);
    if (jtasks[j].n == 0) continue;
    uint32_t task = heap_pop(&jtasks[j]);
    o_task *pt = &s->tasks[task];
    s->popped++;
    int assigned = 0;
    for (uint32_t n = 0; n < s->N && !assigned; n++) {
      s->evals++;
      if (!plugin_predicate(s, pt, &s->nodes[n])) continue;                /* ssn.PredicateFn only */
      size_t np_ = 0;
      for (uint32_t t = 0; t < s->T; t++) {
        const o_task *tk = &s->tasks[t];
        if (!tk->on_node || tk->node != n || tk->node_status != KB_TASK_RUNNING) continue;
        if (s->jobs[tk->job].queue != s->jobs[j].queue) pre[np_++] = t;
      }
      size_t nv = ssn_evictable(s, task, pre, np_, vic, 1);
      if (nv == 0) continue;
      kbo_res all; res_zero(&all);
      for (size_t i = 0; i < nv; i++) res_add(&all, &s->tasks[vic[i]].resreq, s->R);
      if (!res_less_equal(&pt->init_resreq, &all, s->R)) continue;
      kbo_res reclaimed; res_zero(&reclaimed);
      for (size_t i = 0; i < nv; i++) {                                    /* in victim-list order (reclaim.go:156-169) */
        const uint32_t v = vic[i];
        record_eviction(s, v);                                             /* ssn.Evict: cache.Evict first */
        journal_push(s, KB_OP_EVICT, v, s->tasks[v].node, 0);
        interpod_allocated_status(s, v, -1);
        job_set_status(s, v, KB_TASK_RELEASING);
        node_update_task(s, v, KB_TASK_RELEASING);
        fire_deallocate_event(s, v);
        res_add(&reclaimed, &s->tasks[v].resreq, s->R);
        if (res_less_equal(&pt->init_resreq, &reclaimed, s->R)) break;
      }
      if (res_less_equal(&pt->init_resreq, &reclaimed, s->R)) {            /* reclaim.go:174-183 */
        journal_push(s, KB_OP_PIPELINE, task, n, 0);
        if (ssn_pipeline(s, task, n) == KBO_PANIC) break;
        assigned = 1;
      }
    }
    if (assigned) heap_push(&queues, q);
  }
  heap_free(&queues);
  for (uint32_t q = 0; q < s->Q; q++) heap_free(&qjobs[q]);
  for (uint32_t j = 0; j < s->J; j++) heap_free(&jtasks[j]);
  free(qjobs); free(jtasks); free(qseen); free(pre); free(vic);
  return s->panic ? KBO_PANIC : 0;
}
uint64_t kbo_n_evictions(const kbo_session *s) { return s->n_evict; }
uint64_t kbo_n_journal(const kbo_session *s) { return s->n_journal; }
void kbo_get_journal(const kbo_session *s, kb_stmt_op *out) { memcpy(out, s->journal, sizeof(kb_stmt_op) * s->n_journal); }
void kbo_get_evictions(const kbo_session *s, uint32_t *out) { memcpy(out, s->evictions, sizeof(uint32_t) * s->n_evict); }

/* actions/backfill/backfill.go:40-71 (jobs ascending JobID, Pending tasks ascending UID, nodes ascending name) */
int kbo_backfill(kbo_session *s) {
  if (s->panic) return KBO_PANIC;
  for (uint32_t j = 0; j < s->J; j++) {
    o_job *job = &s->jobs[j];
    if (!job->valid) continue;
    for (uint32_t t = job->t0; t < job->t1; t++) {
      o_task *tk = &s->tasks[t];
      if (tk->status != KB_TASK_PENDING) continue;
      if (!res_is_empty(&tk->init_resreq, s->R)) continue;
      s->popped++;
      for (uint32_t n = 0; n < s->N; n++) {
        s->evals++;
        if (!plugin_predicate(s, tk, &s->nodes[n])) continue;
        int e = ssn_allocate(s, t, n);
        if (e == KBO_PANIC) return KBO_PANIC;
        if (e != 0) { /* backfill.go:61-64: Allocate failed -> status was already flipped; try the next node */ continue; }
        break;
      }
    }
  }
  return 0;
}

/* matrix rows [t0,t1) against current node state (parity target of kb_eval_matrix) */
int kbo_eval_matrix(kbo_session *s, uint32_t t0, uint32_t t1, uint32_t fit_mode, uint8_t *mask_bits, uint16_t *score_out) {
  size_t rowb = ((size_t)s->N + 7) / 8;
  uint8_t *feas = (uint8_t *)malloc(s->N ? s->N : 1);
  double *score = (double *)malloc(sizeof(double) * (s->N ? s->N : 1));
  for (uint32_t t = t0; t < t1; t++) {
    eval_all_nodes(s, &s->tasks[t], (int)fit_mode, feas, score);
    uint8_t *mrow = mask_bits + (size_t)(t - t0) * rowb;
    memset(mrow, 0, rowb);
    for (uint32_t n = 0; n < s->N; n++) {
      if (feas[n]) mrow[n >> 3] |= (uint8_t)(1u << (n & 7));
      double sc = feas[n] ? score[n] : 0.0;
      score_out[(size_t)(t - t0) * s->N + n] = (uint16_t)sc;
    }
  }
  free(feas); free(score);
  return 0;
}

/* top-k per row: descending score, ascending node index (SelectBestNode generalised; SortNodes' order differs, see preempt) */
int kbo_argmax_rows(kbo_session *s, uint32_t t0, uint32_t t1, uint32_t fit_mode, uint32_t k, uint32_t *out_node, uint16_t *out_score) {
  uint8_t *feas = (uint8_t *)malloc(s->N ? s->N : 1);
  double *score = (double *)malloc(sizeof(double) * (s->N ? s->N : 1));
  uint8_t *used = (uint8_t *)malloc(s->N ? s->N : 1);
  for (uint32_t t = t0; t < t1; t++) {
    eval_all_nodes(s, &s->tasks[t], (int)fit_mode, feas, score);
    memset(used, 0, s->N);
    for (uint32_t i = 0; i < k; i++) {
      int best = -1; double mx = 0;
      for (uint32_t n = 0; n < s->N; n++) {
        if (!feas[n] || used[n]) continue;
        if (best < 0 || score[n] > mx) { best = (int)n; mx = score[n]; }
      }
      size_t o = (size_t)(t - t0) * k + i;
      if (best < 0) { out_node[o] = KB_NONE; out_score[o] = 0; }
      else { used[best] = 1; out_node[o] = (uint32_t)best; out_score[o] = (uint16_t)mx; }
    }
  }
  free(feas); free(score); free(used);
  return 0;
}

/* ---- getters ---- */
uint64_t kbo_n_decisions(const kbo_session *s) { return s->n_dec; }
void kbo_get_decisions(const kbo_session *s, kb_decision *out) { memcpy(out, s->decisions, sizeof(kb_decision) * s->n_dec); }
uint64_t kbo_n_binds(const kbo_session *s) { return s->n_binds; }
void kbo_get_binds(const kbo_session *s, uint32_t *task_node_out) { memcpy(task_node_out, s->bind_node, sizeof(uint32_t) * s->T); }
void kbo_get_bind_order(const kbo_session *s, uint32_t *out) { memcpy(out, s->bind_order, sizeof(uint32_t) * s->n_binds); }
uint64_t kbo_evals(const kbo_session *s) { return s->evals; }
uint64_t kbo_popped(const kbo_session *s) { return s->popped; }
int kbo_panicked(const kbo_session *s) { return s->panic; }
void kbo_get_task_state(const kbo_session *s, uint8_t *status, uint32_t *node) {
  for (uint32_t t = 0; t < s->T; t++) { if (status) status[t] = s->tasks[t].status; if (node) node[t] = s->tasks[t].node; }
}
void kbo_get_node_state(const kbo_session *s, double *idle, double *releasing, int64_t *nz_cpu, int64_t *nz_mem, int32_t *pod_cnt) {
  for (uint32_t n = 0; n < s->N; n++) {
    for (int d = 0; d < s->R; d++) {
      if (idle) idle[(size_t)d * s->N + n] = s->nodes[n].idle.v[d];
      if (releasing) releasing[(size_t)d * s->N + n] = s->nodes[n].releasing.v[d];
    }
    if (nz_cpu) nz_cpu[n] = s->nodes[n].nz_cpu;
    if (nz_mem) nz_mem[n] = s->nodes[n].nz_mem;
    if (pod_cnt) pod_cnt[n] = s->nodes[n].pod_cnt;
  }
}
void kbo_get_shares(const kbo_session *s, double *job_share, double *queue_share, double *queue_deserved) {
  if (job_share) for (uint32_t j = 0; j < s->J; j++) job_share[j] = s->jobs[j].drf_share;
  if (queue_share) for (uint32_t q = 0; q < s->Q; q++) queue_share[q] = s->queues[q].share;
  if (queue_deserved)
    for (uint32_t q = 0; q < s->Q; q++)
      for (int d = 0; d < s->R; d++) queue_deserved[(size_t)d * s->Q + q] = res_get(&s->queues[q].deserved, d);
}
void kbo_get_job_valid(const kbo_session *s, uint8_t *valid) { for (uint32_t j = 0; j < s->J; j++) valid[j] = (uint8_t)s->jobs[j].valid; }
int32_t kbo_job_valid_num(const kbo_session *s, uint32_t j) { return job_valid_num(&s->jobs[j]); }
int32_t kbo_job_ready_num(const kbo_session *s, uint32_t j) { return job_ready_num(&s->jobs[j]); }
void kbo_set_task_limit(kbo_session *s, uint64_t limit) { s->task_limit = limit; }
