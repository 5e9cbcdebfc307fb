// kb_device.h — plain-data views shared by the host engine (kb_engine.cpp) and the HIP kernels (kb_kernels.hip).
//
// HBM layout of a session: every per-node field is a dense array over NP = N rounded up to 2048
// (so a wave's 16-byte loads never straddle a row end), resource vectors are dimension-major f64[R][NP] /
// f64[R][T]; the k8s-side scorer inputs are int64[NP]; the per-round mask+score matrix is u16[W][NP] plus
// bit-packed u32[W][NP/32].
#pragma once
#include <stdint.h>

#define KB_NODE_PAD 2048u   // NP/8 node chunks split evenly over 256 threads (K3), 16-byte loads never straddle a row
#define KB_MAX_TOPK 4096   // longest candidate list kb_argmax_rows hands out

// arg-max key: (0x40000000 + score + 1) << 32 | (0xFFFFFFFF - node).  0 = no feasible node.  max() over keys = highest
// score, then lowest node index = util.SelectBestNode with the canonical tie-break (scheduler_helper.go:188-208).
// The 0x40000000 bias makes every key the bit pattern of a positive normal float64, and for such patterns integer
// order equals floating-point order: wave reductions use v_max_f64 on DPP-moved halves (3 instructions per step).
#define KB_KEY_BIAS 0x40000000ull
#define KB_KEY(score, node) (((KB_KEY_BIAS + (unsigned long long)(score) + 1ull) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)(node)))
#define KB_KEY_NODE(key) (0xFFFFFFFFu - (uint32_t)((key)&0xFFFFFFFFull))
#define KB_KEY_SCORE(key) ((uint32_t)(((key) >> 32) - KB_KEY_BIAS) - 1u)

struct KbDev {
  int R;
  uint32_t N, NP, T, J, Q;
  // live node state (mutated by the commit kernel)
  double *idle;        // [R][NP] NodeInfo.Idle
  double *rel;         // [R][NP] NodeInfo.Releasing
  long long *nzc;      // [NP] nodeinfo.nonzeroRequest.MilliCPU
  long long *nzm;      // [NP] nodeinfo.nonzeroRequest.Memory
  int *podcnt;         // [NP] len(ni.Tasks)
  unsigned long long *ports;   // [NP] host-port bits (word 0 of the masks) used on the node (nodeinfo.UsedPorts), nullptr: no host ports in the session
  // host-port masks of several words (kb_snapshot.port_words > 1): the port_xw words behind the first.  Only K1's evaluation reads them
  // (kb_k1.hpp: eval_row) and only the host changes them, between rounds (kb_launch_or_ports_x): a pod with a bit in them is decided in
  // a round of its own, and every other pod carries zeros there — the commit kernels never need them
  unsigned long long *ports_x;               // [port_xw][NP]
  const unsigned long long *t_want_x, *t_conf_x;   // [T][port_xw]
  uint32_t port_xw;                          // 0: one word
  // static node data
  const long long *acpu, *amem;   // [NP] nodeinfo.allocatableResource
  const int *maxpods;             // [NP]
  const uint32_t *ncls;           // [NP]
  const uint32_t *nmask;          // [NP] bits 0..29: scalar keys of Idle (!= 0 <=> Idle.ScalarResources != nil: Allocatable had scalar keys);
                                  //      bit 31: Releasing.ScalarResources != nil (it starts EmptyResource() and only gains keys through
                                  //      Add, node_info.go:65,154; inside allocate / backfill it is only ever Sub'ed, so this is static)
  const double *inv_acpu, *inv_amem;  // [NP] 1.0/(double)allocatable, for the exact integer-division estimate
  // tasks
  const double *t_init;           // [R][T]
  const double *t_res;            // [R][T]
  const long long *t_nzc, *t_nzm; // [T]
  const uint32_t *t_cls;          // [T]
  const uint32_t *t_active;       // [T] dims that LessEqual must compare: bits 0,1 always; bit d iff InitResreq[d] > 10
  const uint32_t *t_resmask;      // [T] scalar keys present in Resreq (bit d-2)
  const uint32_t *t_job;          // [T]
  const unsigned long long *t_want, *t_conf;   // [T] host-port bits the pod occupies / that conflict with it (nullptr: none)
  uint8_t *t_status;              // [T] KB_TASK_*
  uint32_t *t_node;               // [T]
  uint32_t *t_bind;               // [T] node handed to the Binder, KB_NONE otherwise
  uint8_t *t_counted;             // [T] task's Resreq is part of drf/proportion "allocated" (AllocatedStatus at open, or placed in-session)
  uint8_t *j_allocated;           // [J] an ssn.Allocate ran for the job in the current action: only then does session.go:277-285
                                  //     dispatch the job's Allocated tasks (cleared by the gang ballot kernel)
  // static predicates
  const uint8_t *compat;          // bit (tc*n_nc + nc); nullptr => all compatible
  const uint32_t *crows;          // the same table as word-aligned rows [n_tc][8] (bit nc of row tc), nullptr if n_nc > 256 or no table
  uint32_t n_nc;
  // policy
  int wL, wM, wB;                 // nodeorder weights (least, most, balanced); inter-pod affinity contributes 0
  // preferred node affinity (nodeorder's NodeAffinity config): Map counts per (task class, node class), nullptr if no pod has any
  const int32_t *aff;             // [n_tc][n_nc]
  const uint8_t *aff_cls;         // [n_tc] the class's row has a non-zero count
  int wNA;
  int pred_enabled;               // predicates plugin registered with EnabledPredicate
  int score_enabled;              // nodeorder plugin registered with EnabledNodeOrder
  uint32_t whole;                 // every request, Idle and Releasing value of the session is a whole number below 2^47 (k8s quantities in milli-units /
                                  // bytes always are): u placements are then Idle - u * Resreq exactly, which the selection kernel's shots rely on
  // inter-pod (anti)affinity (include/kb_engine.h: kb_interpod), all nullptr when no pod carries a term.  A task that is a SUBJECT
  // (it has predicate checks or non-zero priority weights) is only ever evaluated by the matrix kernel against the live counters:
  // it is the first row of its round (the host plans it so; t_ip_subject also sets the row's "fresh matrix" flag).  A placement
  // changes a whole topology domain, which the dirty-node repair of the commit kernels cannot express — and need not: rows that
  // are not subjects read none of this, and the counters are advanced by the commit kernels' epilogue, once per placed task.
  const uint32_t *ip_ctr_dom;     // [C][NP] domain of the node for predicate counter c (KB_NONE_U32: label missing)
  int32_t *ip_ctr_count;          // [C][D] allocated-status pods per domain (live)
  int32_t *ip_ctr_total;          // [C] (live)
  const unsigned long long *t_ip_inc, *t_ip_forbid;   // [T][ip_Wc] 64-bit words: counter c is bit c % 64 of word c / 64
  const uint16_t *t_ip_req;                           // [T] counter that must be positive (0xFFFF: none)
  const uint8_t *t_ip_self, *t_ip_subject, *t_ip_checks;   // [T]; checks: the task has forbid bits or a required counter
  const uint32_t *ip_cls_dom;     // [P][NP]
  const int32_t *ip_cls_bound;    // [P][NP]
  int32_t *ip_cls_unbound;        // [P][NP] (live)
  const unsigned long long *t_ip_cls_inc;   // [T][ip_Wp]
  const uint32_t *t_ip_sig;       // [T]
  const int32_t *ip_sig_w;        // [S][P]
  uint32_t *ip_z;                 // [1] first node (ascending) holding a pod with an empty Spec.NodeName (live)
  long long *ip_scratch_cnt;      // [matrix rows][NP] per-node counts of the priority kernel
  int32_t *ip_scratch_hist;       // [matrix rows][NP] per-domain sums of the priority kernel
  uint32_t ip_C, ip_D, ip_P, ip_Wc, ip_Wp;
  int wPA;
};

// one device round (matrix -> sorted candidates -> commit)
//
// Rows of the window that share a shape (InitResreq, non-zero request, static class) have identical matrix rows, so
// the matrix and the candidate lists are built once per distinct shape of the window ("mrows"); row i of the
// window refers to its shape through shape_slot[i].
// everything the commit kernel needs to know about one window row, gathered contiguously so that the row i+2 can be
// fetched with one wave-uniform (scalar) load while row i is being committed
struct KbRowDesc {          // 56 bytes
  double init0, init1;     // InitResreq cpu, memory
  long long nzc, nzm;      // pod non-zero request
  uint32_t task, active, resmask, cls;
  uint16_t slot;           // index of the row's shape among the mrows
  uint16_t flags;          // bit 0: Resreq cpu/memory == InitResreq cpu/memory (no init container raised them)
                           // bit 1: the task's class has preferred node-affinity terms (score normalised over the feasible set)
                           // bit 2: Resreq == InitResreq in every scalar dimension Resreq names
  uint32_t crow;           // the task class's row of the static-predicate table (bit nc), valid when n_node_classes <= 32
};

struct KbRound {
  // the window, in the reference's task order
  const uint32_t *rows;        // [n_rows] task ids
  const uint32_t *shape_slot;  // [n_rows] index of the row's shape among the mrows
  uint32_t n_rows;
  KbRowDesc *desc;             // [n_rows] built on the device by kb_launch_gather
  unsigned long long *trace;   // optional cycle stamps of the commit kernel (debug), nullptr otherwise
  uint32_t cap;                // slot capacity of the commit kernel's LDS tables (>= n_rows)
  // matrix rows
  const uint32_t *mrows;       // [n_mrows] representative task of each shape (nullptr: mrow i is task mrow_task0 + i)
  uint32_t mrow_task0;
  const uint8_t *same_prev;    // [n_mrows] mrow has the same shape as mrow-1 (only set by kb_eval_matrix's canonical rows)
  uint32_t n_mrows;
  int fit_mode;                // 1 allocate (resource fit + plugin predicates), 0 plugin predicates only
  uint16_t *score;             // [n_mrows][NP]
  uint32_t *maskw;             // [n_mrows][NP/32]
  unsigned long long *keys;    // [n_mrows][L] candidates, best first (descending score, ascending node index), 0-terminated
  uint32_t L;                  // list length; L >= n_rows + 1 guarantees a clean candidate survives any dirty set of the round
  // commit outputs
  unsigned long long *dec;     // [n_rows] decision records: low word node (KB_NONE = stayed Pending), high word kind
  uint32_t *result;            // [8]: n_done, reason, n_dirty, list_exhausted (live rescans), window_refills
  int backfill;                // commit semantics of backfill.go (first node passing the predicates, no score)
  uint32_t batch;              // unused
  uint32_t gather;             // the matrix launch also builds the row descriptors (one extra block row)
  unsigned long long *host_out;   // see KbCommitArgs
  unsigned long long seq;
  double *delta;               // optional per-node committed deltas of owned rows [NP*(2R+3)] (multi-GPU), may be nullptr
  uint32_t own_row0, own_row1;
  // Chained rounds: the host queues round k+1 behind round k before it has seen round k's result.  A commit kernel leaves
  // chain_tag in *chain when its whole window committed (KB_REASON_DONE), 0 otherwise; a round launched with a non-zero
  // chain_expect runs only if it finds that value there and is skipped otherwise (its kernels return at once, its commit
  // kernel reports KB_REASON_SKIPPED).  chain == nullptr: not part of a chain.
  uint32_t *chain;
  uint32_t chain_expect, chain_tag;
  // Overlapped candidate lists (round 3): for a chained round the matrix and arg-max launches run on a SECOND stream beside the
  // predecessor's commit kernel, against whatever node state they find: exact for every node the predecessor leaves alone, arbitrary for
  // the (at most n_prev) nodes it changes.  The arg-max kernel leaves ready_tag in ready[row] behind each finished list; kb_launch_repair
  // (first stream, behind the predecessor's commit) waits for it, re-evaluates the predecessor's nodes against the state it left and
  // merges: `stale` [n_mrows][stale_L] -> keys [n_mrows][L].  ready == nullptr: not an overlapped round.
  uint32_t *ready;
  uint32_t ready_tag;
  void *task_rows;                   // [n_mrows] 64-byte records: each row's task as the matrix kernel evaluates it, written by the arg-max launch in
                                     // front of the tag (the repair launch then needs one load for it instead of a walk through the task arrays)
  const unsigned long long *stale;   // kb_launch_repair: the lists of the overlapped arg-max launch
  uint32_t stale_L;                  // >= n_prev + L: what is left of a stale list without the predecessor's nodes still holds the true top L
  const unsigned long long *prev_dec;   // decision records of the predecessor round (low word: node), n_prev of them
  uint32_t n_prev;
  // Round 5: the selection kernel's launch carries the repair workgroups itself (one per matrix row behind its KB_WARM_GRID workgroups; kb_repair.hpp)
  // instead of a launch of its own in front of it: they leave lists_tag in lists_ready[row] behind each repaired list, and the commit
  // workgroup — which has staged everything else meanwhile — waits for the tags before it stages the lists (bounded; a chain word cleared
  // meanwhile, i.e. a stale list that never arrived, makes it report KB_REASON_SKIPPED like a round queued behind a stopped one).
  // nullptr: the lists are final when the commit launch starts.
  uint32_t *lists_ready;
  uint32_t lists_tag;
};
// true when the round was queued behind a predecessor that did not complete
#define KB_CHAIN_BROKEN(r) ((r).chain_expect != 0u && *(r).chain != (r).chain_expect)

// Hot arguments of the commit kernel: the ~25 scalars its loops touch (they live in SGPRs).  The full session / round
// views are read through `dev` / `round` on rare paths only (scalar resource dimensions, multi-GPU deltas, prologue /
// epilogue); keeping ~110 SGPRs of pointers live made the compiler spill SGPRs into VGPR lanes all over the loop.
struct KbCommitArgs {
  const KbDev *dev;
  const KbRound *round;
  const unsigned long long *keys;
  unsigned long long *dec;
  const KbRowDesc *desc;
  uint32_t *result;
  unsigned long long *trace;
  uint32_t n_rows, n_mrows, L, cap, N, NP, T;
  int fit_mode, backfill, pred_enabled, score_enabled, wL, wM, wB;
  uint32_t use_crow, has_delta, has_aff, has_ports;
  int R;
  uint32_t batch;   // unused (the batch kernel's rows per batch)
  unsigned long long *host_out;   // pinned host mirror of the output block (fast rounds), or nullptr
  unsigned long long seq;         // sequence number published last into host_out[KB_OUT_SEQ]
  uint32_t node_bits;             // width of the node field of the commit kernel's 32-bit keys (kb_node_bits)
  uint32_t prewalk;               // unused
  uint32_t whole;                 // KbDev::whole
};

// 32-bit keys of the commit kernel: (score + 1) << node_bits | (2^node_bits - 1 - node); needs (max score + 2) << node_bits <= 2^32
static inline uint32_t kb_node_bits(uint32_t NP) {
  uint32_t b = 1;
  while ((1ull << b) < (unsigned long long)NP) b++;
  return b;
}

// Output block of a round, 8-byte words: [0..3] eight 32-bit result words, [4] divergence counter of kb_apply_deltas,
// [8..11] wall-clock stamps (round start, candidate lists start, commit start, commit end), [12] sequence number,
// [16..] decision records.
#define KB_OUT_STAMP0 8u
#define KB_OUT_SEQ 12u
#define KB_OUT_HDR 16u
#define KB_OUT_STRIDE (KB_OUT_HDR + KB_K5_MAX_WINDOW)   // words per half of the pinned host mirror (two halves: chained rounds alternate)

#define KB_K5_MAX_WINDOW 1024u   // staging buffers are sized for it; the commit kernel's LDS budget decides the window actually used
#define KB_K5_MAX_ROWS 256u      // rows per window: one thread of the commit kernel's workgroup per dirty slot
#define KB_K5_MAX_SHAPES 256u    // distinct task shapes per window (each keeps its candidate list in LDS: the budget decides)

// Two commit kernels, same decisions bit for bit: KB_COMMIT_SELECT (kb_commit_sel.hip, k_commit_select: a run of same-shape rows committed by
// ONE selection, the workgroup a pipeline of waves) runs every round; KB_COMMIT_RUN (kb_commit.hip, k_commit_run: the same runs, row by row,
// no speculation) is the plain restatement it is held to (KB_COMMIT_KERNEL=run, one axis of the -m gpu suite).  Round 3's batch kernel
// (value 0: speculation across shapes) was retired in round 5.
enum { KB_COMMIT_RUN = 1, KB_COMMIT_SELECT = 2 };
// dynamic LDS a round of n_rows rows with n_shapes distinct shapes over NP padded nodes needs (either kernel)
size_t kb_commit_smem_bytes(uint32_t n_rows, uint32_t n_shapes, uint32_t NP, int R);

// KB_REASON_RENORM: the next row's score needs NormalizeReduce over its CURRENT feasible set (preferred node affinity): it is
// committed as the first row of a fresh round, whose matrix is exact for it
// KB_REASON_SKIPPED: the round was chained to a predecessor that stopped early; nothing was evaluated or committed
enum { KB_REASON_DONE = 0, KB_REASON_NO_FEASIBLE = 1, KB_REASON_PIPELINED = 2, KB_REASON_INTERNAL = 3, KB_REASON_RENORM = 4, KB_REASON_SKIPPED = 5 };

// launch wrappers (kb_kernels.hip); all asynchronous on `stream`
void kb_launch_gather(const KbDev &d, const KbRound &r, void *stream);
void kb_launch_matrix(const KbDev &d, const KbRound &r, void *stream);
// NodeAffinity Map + NormalizeReduce + weight added to the score rows of the matrix (no-op without affinity terms)
void kb_launch_affinity(const KbDev &d, const KbRound &r, void *stream);
// feasibility probe: alive[i] |= 1 iff task rows[i] (allocate's predicate: resource fit + plugin predicates) has a feasible node
// against the current node state; alive must be zero on entry
void kb_launch_probe(const KbDev &d, const uint32_t *rows, uint32_t n_rows, uint32_t *alive, void *stream);
// the pod's host-port words behind the first join the node's (sessions with kb_snapshot.port_words > 1, after the round that placed the pod)
void kb_launch_or_ports_x(const KbDev &d, uint32_t task, uint32_t node, void *stream);
// nodeorder's InterPodAffinityPriority added to the score rows of the matrix rows whose task carries weights (no-op otherwise)
void kb_launch_interpod(const KbDev &d, const KbRound &r, void *stream);
void kb_launch_argmax(const KbDev &d, const KbRound &r, void *stream);
// overlapped rounds (KbRound::ready): stale lists + the predecessor's nodes re-evaluated -> the round's candidate lists
void kb_launch_repair(const KbDev &d, const KbRound &r, void *stream);
size_t kb_repair_smem_bytes(uint32_t NP);   // its dynamic LDS (<= 150 KiB or the engine does not overlap rounds)
// per-task rows out of the per-shape rows (kb_eval_matrix / kb_bench_matrix: the materialised T x N matrix)
// order: the rows sorted by shape slot (nullptr: row order).  chunks: stretches of `order` (of the rows themselves without it) that share a shape, at most KB_XCHUNK_ROWS rows
// each — a workgroup then loads its tile of the shape row (up to 16 384 nodes) ONCE and stores it to every row of its chunk (round 6: the source is read
// ~n_rows / KB_XCHUNK_ROWS times instead of n_rows times; round 5's copy, a workgroup per task row, re-fetched a few MB of shape rows through eight L2s 226 MB worth per launch)
struct KbXChunk { uint32_t slot, first, count, pad; };
#ifndef KB_XCHUNK_ROWS
#define KB_XCHUNK_ROWS 64u   // at most 64: one lane of a wave per row of a chunk
#endif
void kb_launch_expand(const KbDev &d, const uint16_t *s_score, const uint32_t *s_mask, const uint32_t *row_slot, const uint32_t *order, uint32_t n_rows,
                      uint16_t *score, uint32_t *maskw, void *stream, const KbXChunk *chunks = nullptr, uint32_t n_chunks = 0);
void kb_launch_commit(const KbDev &d, const KbRound &r, void *stream);         // KB_COMMIT_RUN
void kb_launch_commit_sel(const KbDev &d, const KbRound &r, void *stream);     // KB_COMMIT_SELECT
// node state := round-start state + reduced deltas; returns how many values differ from the live (locally committed) state
uint32_t kb_apply_deltas(const KbDev &d, const double *s_idle, const double *s_rel, const long long *s_nzc, const long long *s_nzm,
                         const int *s_podcnt, const double *delta, uint32_t *dev_counter, void *stream);
// (s0 + delta) == s1 for every node value, s0 / s1: two copies of the node state {Idle [R][NP], Releasing [R][NP], nz cpu, nz mem, pod count};
// differences are ADDED to *dev_counter; queued on `stream`, nothing waited for (the deferred cross-check of the sharded path)
struct KbNodeCopy { const double *idle, *rel; const long long *nzc, *nzm; const int *podcnt; };
void kb_check_deltas(const KbDev &d, const KbNodeCopy &s0, const KbNodeCopy &s1, const double *delta, uint32_t *dev_counter, void *stream);
// live state of n nodes from packed records of 5 + 2R 8-byte words {node | nmask << 32, podcnt, nzc, nzm, ports, Idle[R], Releasing[R]}
// (the evict actions' host mirror -> device); nmask: the writable view of KbDev::nmask
void kb_launch_scatter_nodes(const KbDev &d, const unsigned long long *rec, uint32_t n, uint32_t *nmask, void *stream);
// gang ballot + drf/proportion share reduction over the task table
void kb_launch_finalize(const KbDev &d, const uint32_t *job_task_begin, const int *job_min_avail, const uint32_t *job_queue,
                        int gang_ready_enabled, const double *total /*[R]*/, uint32_t total_mask, const double *deserved /*[R][Q]*/,
                        const uint32_t *deserved_mask /*[Q]*/, double *job_alloc /*[J][R]*/, double *job_share /*[J]*/,
                        double *queue_alloc /*[Q][R]*/, double *queue_share /*[Q]*/, int *job_ready_cnt /*[J]*/, void *stream);
