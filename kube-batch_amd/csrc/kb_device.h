// kb_device.h — plain-data views shared by the host engine (kb_engine.cpp) and the HIP kernels (kb_kernels.hip).
//
// HBM layout of a session: every per-node field is a dense array over NP = N rounded up to 1024
// (so a wave's 16-byte loads never straddle a row end), resource vectors are dimension-major f64[R][NP] /
// f64[R][T]; the k8s-side scorer inputs are int64[NP]; the per-round mask+score matrix is u16[W][NP] plus
// bit-packed u32[W][NP/32].
#pragma once
#include <stdint.h>

#define KB_NODE_PAD 1024u
#define KB_K5_THREADS 1024
#define KB_MAX_TOPK 32

// arg-max key: (score+1) << 32 | (0xFFFFFFFF - node). 0 = no feasible node.  max() over keys = highest score,
// then lowest node index = util.SelectBestNode with the canonical tie-break (scheduler_helper.go:188-208).
#define KB_KEY(score, node) ((((unsigned long long)(score) + 1ull) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)(node)))
#define KB_KEY_NODE(key) (0xFFFFFFFFu - (uint32_t)((key)&0xFFFFFFFFull))
#define KB_KEY_SCORE(key) ((uint32_t)((key) >> 32) - 1u)

struct KbDev {
  int R;
  uint32_t N, NP, T, J, Q;
  // live node state (mutated by the commit kernel)
  double *idle;        // [R][NP] NodeInfo.Idle
  double *rel;         // [R][NP] NodeInfo.Releasing
  long long *nzc;      // [NP] nodeinfo.nonzeroRequest.MilliCPU
  long long *nzm;      // [NP] nodeinfo.nonzeroRequest.Memory
  int *podcnt;         // [NP] len(ni.Tasks)
  // static node data
  const long long *acpu, *amem;   // [NP] nodeinfo.allocatableResource
  const int *maxpods;             // [NP]
  const uint32_t *ncls;           // [NP]
  const uint32_t *nmask;          // [NP] Idle.ScalarResources != nil  (Allocatable had scalar keys)
  // tasks
  const double *t_init;           // [R][T]
  const double *t_res;            // [R][T]
  const long long *t_nzc, *t_nzm; // [T]
  const uint32_t *t_cls;          // [T]
  const uint32_t *t_active;       // [T] dims that LessEqual must compare: bits 0,1 always; bit d iff InitResreq[d] > 10
  const uint32_t *t_resmask;      // [T] scalar keys present in Resreq (bit d-2)
  const uint32_t *t_job;          // [T]
  uint8_t *t_status;              // [T] KB_TASK_*
  uint32_t *t_node;               // [T]
  uint32_t *t_bind;               // [T] node handed to the Binder, KB_NONE otherwise
  uint8_t *t_counted;             // [T] task's Resreq is part of drf/proportion "allocated" (AllocatedStatus at open, or placed in-session)
  // static predicates
  const uint8_t *compat;          // bit (tc*n_nc + nc); nullptr => all compatible
  uint32_t n_nc;
  // policy
  int wL, wM, wB;                 // nodeorder weights (least, most, balanced); node/pod affinity contribute 0
  int pred_enabled;               // predicates plugin registered with EnabledPredicate
  int score_enabled;              // nodeorder plugin registered with EnabledNodeOrder
};

// one device round (matrix -> arg-max -> commit)
struct KbRound {
  const uint32_t *rows;       // [n_rows] task ids in speculated order (nullptr: row i is task row_task0 + i)
  uint32_t row_task0;
  const uint8_t *same_prev;   // [n_rows] row has the same shape (InitResreq, non-zero request, class) as the previous row
  uint32_t n_rows;
  int fit_mode;               // 1 allocate (resource fit + plugin predicates), 0 plugin predicates only
  uint16_t *score;            // [n_rows][NP]
  uint32_t *maskw;            // [n_rows][NP/32]
  unsigned long long *keys;   // [n_rows][topk]
  uint32_t topk;
  // commit outputs
  uint32_t *dec_node;         // [n_rows]
  uint32_t *dec_kind;         // [n_rows]
  uint32_t *result;           // [8]: n_done, reason, n_dirty, fallbacks, live_rescans
  uint32_t *dirty_list;       // [n_rows]
  int use_rows;               // rows of the matrix are resident locally (single GPU); 0 = candidates only
  int backfill;               // commit semantics of backfill.go (status Allocated, Resreq accounting, no share feedback)
  double *delta;              // optional per-node committed deltas of owned rows [NP*(2R+3)] (multi-GPU), may be nullptr
  uint32_t own_row0, own_row1;
};

enum { KB_REASON_DONE = 0, KB_REASON_NO_FEASIBLE = 1, KB_REASON_PIPELINED = 2 };

// launch wrappers (kb_kernels.hip); all asynchronous on `stream`
void kb_launch_matrix(const KbDev &d, const KbRound &r, void *stream);
void kb_launch_argmax(const KbDev &d, const KbRound &r, void *stream);
void kb_launch_commit(const KbDev &d, const KbRound &r, void *stream);
// gang ballot + drf/proportion share reduction over the task table
void kb_launch_finalize(const KbDev &d, const uint32_t *job_task_begin, const int *job_min_avail, const uint32_t *job_queue,
                        int gang_ready_enabled, const double *total /*[R]*/, uint32_t total_mask, const double *deserved /*[R][Q]*/,
                        const uint32_t *deserved_mask /*[Q]*/, double *job_alloc /*[J][R]*/, double *job_share /*[J]*/,
                        double *queue_alloc /*[Q][R]*/, double *queue_share /*[Q]*/, int *job_ready_cnt /*[J]*/, void *stream);
