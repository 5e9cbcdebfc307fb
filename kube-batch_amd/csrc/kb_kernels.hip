// kb_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for kube-batch's allocate/backfill hot path.
//
//   K1 k_matrix    mask + score matrix for a window of task rows x all nodes
//                  = allocate.go:73-87 predicate closure + plugins/predicates/predicates.go:123-265 (pod-count cap,
//                    static checks as a class bit table) + nodeorder's LeastRequested / MostRequested /
//                    BalancedResourceAllocation (vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/*.go)
//                    summed as util.PrioritizeNodes does (scheduler_helper.go:162-168).
//   K3 k_argmax    per matrix row, the first K entries of (score descending, node ascending) = util.SelectBestNode
//                  (scheduler_helper.go:188-208, first max in ascending node order) generalised to a sorted candidate list
//                  (also the order util.SortNodes gives preempt, scheduler_helper.go:174-185).
//   K5 k_commit    the sequential part of allocate.go:129-193 / backfill.go:44-67: for each row in reference order the best
//                  node among the clean columns (next untouched entry of the shape's list) and the re-evaluated dirty
//                  columns, then NodeInfo.AddTask accounting (api/node_info.go:161-212); 16-32 rows speculated per batch.
//   K2+K4 k_finalize   gang ready count by wavefront ballot (api/job_info.go:383-394, gang.go:122-125), gang-gated bind
//                  set (framework/session.go:277-285), drf / proportion share reduction (drf.go:157-171,
//                  proportion.go:241-253, api/helpers/helpers.go:47-60).
//
// Exactness: float64 compares with the reference's epsilons, IEEE double division (no fast-math, -ffp-contract=off),
// int64 truncating division reproduced exactly through a reciprocal estimate + integer remainder fix-up.
// The path is elementwise compare + integer scoring, HBM/latency bound: no MFMA.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "kb_device.h"

#define EPS_CPU 10.0                    // resource_info.go:68
#define EPS_SCALAR 10.0                 // resource_info.go:69
#define EPS_MEM (10.0 * 1024 * 1024)    // resource_info.go:70
#define KB_TASK_ALLOCATED 1
#define KB_TASK_PIPELINED 2
#define KB_TASK_BINDING 3
#define KB_TASK_BOUND 4
#define KB_TASK_RUNNING 5
#define KB_TASK_SUCCEEDED 7
#define KB_NONE_U32 0xFFFFFFFFu

struct TaskVals {
  double init0, init1;
  long long nzc, nzm;
  uint32_t cls, active, task, pad;
  unsigned long long conf;   // host-port bits that conflict with this pod's ports (0: none)
};
struct NodeVals {
  double idle0, idle1, rel0, rel1;
  long long ac, am, nzc, nzm;
  double inv_ac, inv_am;
  uint32_t cls;
  int slots;   // Allocatable.MaxTaskNum > len(pods)  (predicates.go:127 fails on <=)
  int valid;   // node index < N
  unsigned long long ports;   // host-port bits used by the pods on the node
};

__device__ __forceinline__ bool le_eps(double l, double r, double eps) { return (l < r) || (fabs(l - r) < eps); }

__device__ __forceinline__ TaskVals load_task(const KbDev &d, uint32_t t) {
  TaskVals tv;
  tv.init0 = d.t_init[t];
  tv.init1 = d.t_init[(size_t)d.T + t];
  tv.nzc = d.t_nzc[t];
  tv.nzm = d.t_nzm[t];
  tv.cls = d.t_cls[t];
  tv.active = d.t_active[t];
  tv.task = t;
  tv.pad = 0;
  tv.conf = d.t_conf ? d.t_conf[t] : 0ull;
  return tv;
}

__device__ __forceinline__ NodeVals load_node(const KbDev &d, uint32_t n) {
  NodeVals nv;
  nv.valid = n < d.N;
  uint32_t m = nv.valid ? n : 0;
  nv.idle0 = d.idle[m];
  nv.idle1 = d.idle[(size_t)d.NP + m];
  nv.rel0 = d.rel[m];
  nv.rel1 = d.rel[(size_t)d.NP + m];
  nv.ac = d.acpu[m];
  nv.am = d.amem[m];
  nv.nzc = d.nzc[m];
  nv.nzm = d.nzm[m];
  nv.inv_ac = d.inv_acpu[m];
  nv.inv_am = d.inv_amem[m];
  nv.cls = d.ncls[m];
  nv.slots = d.maxpods[m] > d.podcnt[m];
  nv.ports = d.ports ? d.ports[m] : 0ull;
  return nv;
}

// floor(10*req/cap) for 0 <= req <= cap, cap > 0, exact: reciprocal estimate, then one integer remainder correction.
__device__ __forceinline__ int div10(long long req, long long cap, double inv_cap, int &rem_nonzero) {
  long long a = req * 10;
  int q = (int)((double)a * inv_cap);
  long long rem = a - (long long)q * cap;
  if (rem < 0) { q -= 1; rem += cap; }
  else if (rem >= cap) { q += 1; rem -= cap; }
  rem_nonzero = rem != 0;
  return q;
}

// One (task,node) evaluation.  Returns 0 if infeasible, else 0x10000 | score.
// nodeorder's three resource scorers summed with their weights (scheduler_helper.go:162-168); shared by every evaluation path
__device__ __forceinline__ uint32_t score_core(const TaskVals &t, const NodeVals &n, int wL, int wM, int wB) {
  long long rc = n.nzc + t.nzc, rm = n.nzm + t.nzm;   // resource_allocation.go:100-112
  int lc = 0, mc = 0, lm = 0, mm = 0, rem;
  if (!(n.ac == 0 || rc > n.ac)) { mc = div10(rc, n.ac, n.inv_ac, rem); lc = 10 - mc - rem; }   // most/least_requested.go
  if (!(n.am == 0 || rm > n.am)) { mm = div10(rm, n.am, n.inv_am, rem); lm = 10 - mm - rem; }
  int least = (lc + lm) / 2, most = (mc + mm) / 2;
  double cf = (n.ac == 0) ? 1.0 : (double)rc / (double)n.ac;      // balanced_resource_allocation.go:74-79
  double mf = (n.am == 0) ? 1.0 : (double)rm / (double)n.am;
  int bal = 0;
  if (!(cf >= 1.0 || mf >= 1.0)) bal = (int)(long long)((1.0 - fabs(cf - mf)) * 10.0);
  return (uint32_t)(least * wL + most * wM + bal * wB);
}

// class_row: nullptr -> look the class pair up in the global bit table; otherwise the task class's row of the table
// (bit nc), e.g. staged in LDS by the commit kernel so that no global load sits on its critical path.
__device__ __forceinline__ uint32_t eval_pair(const KbDev &d, const TaskVals &t, const NodeVals &n, uint32_t node, int fit_mode,
                                              const uint32_t *class_row = nullptr) {
  if (!n.valid) return 0;
  bool ok = true;
  if (fit_mode) {   // allocate.go:81: !InitResreq.LessEqual(Idle) && !InitResreq.LessEqual(Releasing) -> fail
    bool fi = le_eps(t.init0, n.idle0, EPS_CPU) && le_eps(t.init1, n.idle1, EPS_MEM);
    bool fr = le_eps(t.init0, n.rel0, EPS_CPU) && le_eps(t.init1, n.rel1, EPS_MEM);
    uint32_t a = t.active >> 2;     // scalar dims with InitResreq > 10 (resource_info.go:286-299)
    uint32_t dd = 2;
    while (a) {
      if (a & 1u) {
        double l = d.t_init[(size_t)dd * d.T + t.task];
        fi = fi && le_eps(l, d.idle[(size_t)dd * d.NP + node], EPS_SCALAR);
        fr = fr && le_eps(l, d.rel[(size_t)dd * d.NP + node], EPS_SCALAR);
      }
      a >>= 1;
      dd++;
    }
    ok = fi || fr;
  }
  if (d.pred_enabled) {
    ok = ok && n.slots && ((n.ports & t.conf) == 0ull);   // pod count (predicates.go:127), PodFitsHostPorts (predicates.go:181-190)
    if (class_row) {
      ok = ok && ((class_row[n.cls >> 5] >> (n.cls & 31)) & 1u);
    } else if (d.compat) {
      uint32_t bit = t.cls * d.n_nc + n.cls;
      ok = ok && ((d.compat[bit >> 3] >> (bit & 7)) & 1);
    }
  }
  if (!ok) return 0;
  uint32_t score = 0;
  if (d.score_enabled) score = score_core(t, n, d.wL, d.wM, d.wB);
  return 0x10000u | (score & 0xFFFFu);
}

// ------------------------------------------------------------------------------------------------------------
// K1: mask + score matrix.  grid (NP / (256*NPT), ceil(n_rows/TR)); thread <-> NPT consecutive nodes kept in
// registers for all TR rows of the tile; the tile's task vectors are staged once in LDS.
// Stores: one 8-byte score vector per thread per row (512 B contiguous per wave), one 4-byte mask word per 8 lanes.
// ------------------------------------------------------------------------------------------------------------
// Two tile shapes: <4 nodes/thread, 32 rows/block> for matrix-sized launches (8-byte score stores, node state amortised
// over 32 rows) and <1 node/thread, 4 rows/block> for the small per-round launches (a few dozen distinct shapes), where
// the work has to be spread over all 256 CUs instead of being serialised inside few threads.
__device__ __forceinline__ void gather_row(const KbDev &d, const KbRound &r, uint32_t i) {
  if (i >= r.n_rows) return;
  const uint32_t t = r.rows[i];
  KbRowDesc k;
  k.init0 = d.t_init[t]; k.init1 = d.t_init[(size_t)d.T + t];
  k.nzc = d.t_nzc[t]; k.nzm = d.t_nzm[t];
  k.task = t; k.active = d.t_active[t]; k.resmask = d.t_resmask[t]; k.cls = d.t_cls[t];
  k.slot = (uint16_t)r.shape_slot[i];
  k.flags = (d.t_res[t] == k.init0 && d.t_res[(size_t)d.T + t] == k.init1) ? 1 : 0;
  if (d.aff_cls && d.aff_cls[k.cls]) k.flags |= 2;
  k.crow = d.crows ? d.crows[(size_t)k.cls * 8] : 0xFFFFFFFFu;
  r.desc[i] = k;
}

template <int NPT, int TR>
__global__ void __launch_bounds__(256) k_matrix(KbDev d, KbRound r) {
  __shared__ TaskVals srow[TR];
  __shared__ uint8_t ssame[TR];
  if (r.gather && blockIdx.y == gridDim.y - 1) {   // the extra block row of a single-GPU round: the window's row descriptors
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long *>(r.result)[KB_OUT_STAMP0] = wall_clock64();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < r.n_rows; i += gridDim.x * 256) gather_row(d, r, i);
    return;
  }
  const uint32_t row0 = blockIdx.y * TR;
  const uint32_t nr = min((uint32_t)TR, r.n_mrows - row0);
  if (threadIdx.x < nr) {
    uint32_t i = row0 + threadIdx.x;
    uint32_t t = r.mrows ? r.mrows[i] : r.mrow_task0 + i;
    srow[threadIdx.x] = load_task(d, t);
    ssame[threadIdx.x] = r.same_prev ? r.same_prev[i] : 0;
  }
  __syncthreads();
  const uint32_t n0 = (blockIdx.x * 256 + threadIdx.x) * NPT;
  const uint32_t lane = threadIdx.x & 63;
  NodeVals nv[NPT];
#pragma unroll
  for (int j = 0; j < NPT; j++) nv[j] = load_node(d, n0 + j);
  uint32_t res[NPT];
#pragma unroll
  for (int j = 0; j < NPT; j++) res[j] = 0;
  const size_t mstride = d.NP / 32;
  uint2 pk = make_uint2(0u, 0u);   // packed scores / mask word of the last evaluated row (re-stored for identical rows)
  uint32_t mw = 0;
  for (uint32_t rr = 0; rr < nr; rr++) {
#ifdef KB_K1_OLDLOOP
    const bool fresh = true;
    const bool evalrow = !(ssame[rr] && rr > 0);
    const TaskVals tv = srow[rr];
    if (evalrow) {
#else
    const bool fresh = !(ssame[rr] && rr > 0);
    if (fresh) {
      const TaskVals tv = srow[rr];
#endif
#ifdef KB_K1_NOEVAL   // timing experiment: store path only
#pragma unroll
      for (int j = 0; j < NPT; j++) res[j] = 0x10000u | (tv.cls + (uint32_t)nv[j].nzc);
#else
#pragma unroll
      for (int j = 0; j < NPT; j++) res[j] = eval_pair(d, tv, nv[j], n0 + j, r.fit_mode);
#endif
    }
    const size_t row = row0 + rr;
    if (NPT == 4) {
      if (fresh) {
        pk.x = (res[0] & 0xFFFFu) | (res[1 % NPT] << 16);
        pk.y = (res[2 % NPT] & 0xFFFFu) | (res[3 % NPT] << 16);
        uint32_t nib = ((res[0] >> 16) & 1u) | (((res[1 % NPT] >> 16) & 1u) << 1) | (((res[2 % NPT] >> 16) & 1u) << 2) | (((res[3 % NPT] >> 16) & 1u) << 3);
        // OR the 8 lanes' nibbles into one mask word with DPP moves (no LDS crossbar): xor 1, xor 2, mirror within 8
        uint32_t w = nib << (4 * (lane & 7));
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x141, 0xf, 0xf, false);   // row_half_mirror
        mw = w;
      }
      // streaming stores: the matrix is written once and read by another kernel
#ifdef KB_K1_NT
      __builtin_nontemporal_store(((unsigned long long)pk.y << 32) | pk.x, reinterpret_cast<unsigned long long *>(r.score + row * d.NP + n0));
      if ((lane & 7) == 0) __builtin_nontemporal_store(mw, &r.maskw[row * mstride + (n0 >> 5)]);
#elif defined(KB_K1_NOSTORE)   // timing experiment: evaluation only
      if (pk.x == 0x12345678u && mw == 0x9abcdefu) *reinterpret_cast<uint2 *>(r.score + row * d.NP + n0) = pk;
#else
      *reinterpret_cast<uint2 *>(r.score + row * d.NP + n0) = pk;
      if ((lane & 7) == 0) r.maskw[row * mstride + (n0 >> 5)] = mw;
#endif
    } else {
      r.score[row * d.NP + n0] = (uint16_t)(res[0] & 0xFFFFu);
      unsigned long long b = __ballot((res[0] >> 16) & 1u);   // one wave = 64 consecutive nodes = two mask words
      if (lane == 0) *reinterpret_cast<unsigned long long *>(r.maskw + row * mstride + (n0 >> 5)) = b;
    }
  }
}

// K1b: row expansion.  Tasks with the same shape (InitResreq, non-zero request, class) have identical matrix rows, so the
// materialised T x N matrix is produced by evaluating each distinct shape once (k_matrix over the representative rows)
// and streaming every task row out of its shape's row: 16-byte loads that hit L2 / Infinity Cache (S x N is a few MB),
// 16-byte stores that are the launch's HBM traffic (2 B score + 1 mask bit per evaluation).  One workgroup per task row.
__global__ void __launch_bounds__(256) k_expand(const uint16_t *__restrict__ s_score, const uint32_t *__restrict__ s_mask,
                                                const uint32_t *__restrict__ row_slot, uint32_t n_rows, uint32_t NP,
                                                uint16_t *__restrict__ score, uint32_t *__restrict__ maskw) {
  const uint32_t row = blockIdx.x;
  if (row >= n_rows) return;
  const uint32_t slot = row_slot[row];
  const uint4 *src = reinterpret_cast<const uint4 *>(s_score + (size_t)slot * NP);
  uint4 *dst = reinterpret_cast<uint4 *>(score + (size_t)row * NP);
  const uint32_t n16 = NP / 8;                 // 16-byte chunks per row (NP is a multiple of 2048)
  for (uint32_t c = threadIdx.x; c < n16; c += 256) dst[c] = src[c];
  const uint4 *msrc = reinterpret_cast<const uint4 *>(s_mask + (size_t)slot * (NP / 32));
  uint4 *mdst = reinterpret_cast<uint4 *>(maskw + (size_t)row * (NP / 32));
  const uint32_t m16 = NP / 128;               // NP/32 words = NP/128 16-byte chunks
  for (uint32_t c = threadIdx.x; c < m16; c += 256) mdst[c] = msrc[c];
}

// ------------------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------------------
// max over keys (bit patterns of positive normal doubles, or 0) with DPP moves + v_max_f64
#define KB_DPP_STEP(v, ctrl, row_mask)                                                                    \
  do {                                                                                                    \
    int _lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), (ctrl), (row_mask), 0xf, false);          \
    int _hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), (ctrl), (row_mask), 0xf, false);          \
    v = fmax(v, __hiloint2double(_hi, _lo));                                                              \
  } while (0)
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k) {
  double v = __longlong_as_double((long long)k);
  KB_DPP_STEP(v, 0xB1, 0xf);    // quad_perm [1,0,3,2]
  KB_DPP_STEP(v, 0x4E, 0xf);    // quad_perm [2,3,0,1]
  KB_DPP_STEP(v, 0x141, 0xf);   // row_half_mirror
  KB_DPP_STEP(v, 0x140, 0xf);   // row_mirror: every lane of a 16-lane row holds the row maximum
  KB_DPP_STEP(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
  KB_DPP_STEP(v, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave maximum
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
// max over each aligned group of 8 lanes (every lane of the group gets it)
__device__ __forceinline__ unsigned long long oct_max_key(unsigned long long k) {
  double v = __longlong_as_double((long long)k);
  KB_DPP_STEP(v, 0xB1, 0xf);
  KB_DPP_STEP(v, 0x4E, 0xf);
  KB_DPP_STEP(v, 0x141, 0xf);
  return (unsigned long long)__double_as_longlong(v);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) { return wave_max_key(v); }
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ uint32_t lane_prefix_popc(unsigned long long ballot_mask, uint32_t lane) {
  return __popcll(ballot_mask & ((1ull << lane) - 1ull));
}

// ------------------------------------------------------------------------------------------------------------
// K3: segmented arg-max generalised to a sorted candidate list (best first: descending score, ascending node index).
// One 256-thread workgroup per matrix row.  The row (u16 scores + mask bits) is staged once in LDS with 16-byte loads;
// each thread owns a contiguous run of NP/256 nodes, so "ascending node index" is thread order.  Per score level:
// pass A = block max of the scores below the previous level, pass B = count + block exclusive scan + ordered write.
// The first entry is util.SelectBestNode's choice (scheduler_helper.go:188-208, canonical first-max tie-break).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_max_i32_dpp(int v) {
#define KB_DPP_IMAX(ctrl, row_mask) v = max(v, __builtin_amdgcn_update_dpp(v, v, (ctrl), (row_mask), 0xf, false))
  KB_DPP_IMAX(0xB1, 0xf);
  KB_DPP_IMAX(0x4E, 0xf);
  KB_DPP_IMAX(0x141, 0xf);
  KB_DPP_IMAX(0x140, 0xf);
  KB_DPP_IMAX(0x142, 0xa);
  KB_DPP_IMAX(0x143, 0xc);
#undef KB_DPP_IMAX
  return __builtin_amdgcn_readlane(v, 63);
}
// inclusive scan inside the wave: DPP row shifts, then row broadcasts (the sequence LLVM's buildScan emits)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
  return v;
}

// Sorted candidate list of one matrix row: the first K entries of (score descending, node ascending).  Scores are
// integers, so the list is built FOUR consecutive score values at a time: one pass over the row (staged in LDS) counts
// each thread's nodes at top, top-1, top-2, top-3 (four 16-bit counters packed in two words) and finds the best score
// below that band; one packed scan turns the counts into output positions; one more pass scatters.  With the default
// weights the first band already holds more than a window's worth of nodes.  WIDE (rows of 65536 nodes or more, where a
// 16-bit counter could overflow): two score values per band, one full word each.
template <bool WIDE, int THREADS>
__global__ void __launch_bounds__(THREADS) k_argmax(KbDev d, KbRound r) {
  constexpr int WAVES = THREADS / 64;
  extern __shared__ __align__(16) unsigned char k3_smem[];
  uint16_t *ls = reinterpret_cast<uint16_t *>(k3_smem);                        // [NP] scores
  uint8_t *lm = reinterpret_cast<uint8_t *>(k3_smem) + (size_t)d.NP * 2;      // [NP/8] mask bytes
  __shared__ int s_wmax[WAVES];
  __shared__ uint32_t s_wcnt[WAVES][2];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t row = blockIdx.x;
  if (row == 0 && tid == 0 && r.mrow_task0 == 0 && r.mrows != nullptr)   // round launches only (not kb_eval_matrix's expanded rows)
    reinterpret_cast<unsigned long long *>(r.result)[KB_OUT_STAMP0 + 1] = wall_clock64();
  const uint32_t K = r.L;
  unsigned long long *out = r.keys + (size_t)row * K;
  // contiguous, balanced ranges of 8-node chunks: thread order == node order (NP/8 need not be a multiple of THREADS)
  const uint32_t nchunks = d.NP / 8;
  const uint32_t cbase = (uint32_t)(((unsigned long long)tid * nchunks) / THREADS);
  const uint32_t per8 = (uint32_t)(((unsigned long long)(tid + 1) * nchunks) / THREADS) - cbase;
  const uint4 *ls4 = reinterpret_cast<const uint4 *>(ls);
  int top = -1;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(r.score + (size_t)row * d.NP);
    uint4 *dst = reinterpret_cast<uint4 *>(ls);
    for (uint32_t c = tid; c < d.NP / 8; c += THREADS) dst[c] = src[c];
    const uint32_t *msrc = r.maskw + (size_t)row * (d.NP / 32);
    uint32_t *mdst = reinterpret_cast<uint32_t *>(lm);
    for (uint32_t c = tid; c < d.NP / 32; c += THREADS) mdst[c] = msrc[c];
  }
  __syncthreads();
  for (uint32_t c = 0; c < per8; c++) {   // best score of the row
    const uint32_t mb = lm[cbase + c];
    const uint4 sv = ls4[cbase + c];
    const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
      if (((mb >> e) & 1u) && sc > top) top = sc;
    }
  }
  top = wave_max_i32_dpp(top);
  if (lane == 0) s_wmax[wave] = top;
  __syncthreads();
  top = s_wmax[0];
#pragma unroll
  for (int w2 = 1; w2 < WAVES; w2++) top = max(top, s_wmax[w2]);
  uint32_t found = 0;
  while (top >= 0 && found < K) {
    __syncthreads();   // s_wmax / s_wcnt are reused
    // counts of this thread's nodes at the four scores of the band; best score below the band
    uint32_t c01 = 0, c23 = 0;   // 16-bit fields: (top, top-1), (top-2, top-3)
    int below = -1;
    for (uint32_t c = 0; c < per8; c++) {
      const uint32_t mb = lm[cbase + c];
      const uint4 sv = ls4[cbase + c];
      const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
        if ((mb >> e) & 1u) {
          const int l = top - sc;   // 0..3 inside the band
          if (WIDE) {
            if (l == 0) c01++;
            else if (l == 1) c23++;
            else if (l >= 2 && sc > below) below = sc;
          } else {
            if (l >= 0 && l < 2) c01 += 1u << (16 * l);
            else if (l >= 2 && l < 4) c23 += 1u << (16 * (l - 2));
            else if (l >= 4 && sc > below) below = sc;
          }
        }
      }
    }
    const uint32_t p01 = wave_incl_scan_u32(c01), p23 = wave_incl_scan_u32(c23);
    below = wave_max_i32_dpp(below);
    if (lane == 63) { s_wcnt[wave][0] = p01; s_wcnt[wave][1] = p23; }
    if (lane == 0) s_wmax[wave] = below;
    __syncthreads();
    uint32_t off01 = 0, off23 = 0, tot01 = 0, tot23 = 0;
#pragma unroll
    for (int w2 = 0; w2 < WAVES; w2++) {
      if (w2 < (int)wave) { off01 += s_wcnt[w2][0]; off23 += s_wcnt[w2][1]; }
      tot01 += s_wcnt[w2][0];
      tot23 += s_wcnt[w2][1];
    }
    below = s_wmax[0];
#pragma unroll
    for (int w2 = 1; w2 < WAVES; w2++) below = max(below, s_wmax[w2]);
    // first output position of this thread's nodes at each score of the band
    const uint32_t e01 = off01 + p01 - c01, e23 = off23 + p23 - c23;
    uint32_t t0, t1, t2, t3, pos[4];
    if (WIDE) {
      t0 = tot01; t1 = tot23; t2 = 0; t3 = 0;
      pos[0] = found + e01; pos[1] = found + t0 + e23; pos[2] = 0xFFFFFFFFu; pos[3] = 0xFFFFFFFFu;
    } else {
      t0 = tot01 & 0xFFFFu; t1 = tot01 >> 16; t2 = tot23 & 0xFFFFu; t3 = tot23 >> 16;
      pos[0] = found + (e01 & 0xFFFFu); pos[1] = found + t0 + (e01 >> 16);
      pos[2] = found + t0 + t1 + (e23 & 0xFFFFu); pos[3] = found + t0 + t1 + t2 + (e23 >> 16);
    }
    if ((c01 | c23) && min(min(pos[0], pos[1]), min(pos[2], pos[3])) < K) {
      for (uint32_t c = 0; c < per8; c++) {
        const uint32_t mb = lm[cbase + c];
        const uint4 sv = ls4[cbase + c];
        const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
          const int l = top - sc;
          if (((mb >> e) & 1u) && l >= 0 && l < (WIDE ? 2 : 4)) {
            const uint32_t at = (l == 0) ? pos[0]++ : (l == 1) ? pos[1]++ : (l == 2) ? pos[2]++ : pos[3]++;
            if (at < K) out[at] = KB_KEY(sc, (cbase + c) * 8 + e);
          }
        }
      }
    }
    found += t0 + t1 + t2 + t3;
    top = below;
  }
  if (found > K) found = K;
  for (uint32_t i = found + tid; i < K; i += THREADS) out[i] = 0ull;
}

__device__ __forceinline__ bool bit_test(const uint32_t *bm, uint32_t n) { return (bm[n >> 5] >> (n & 31)) & 1u; }


// window rows -> contiguous descriptors (multi-GPU rounds launch it on its own; single-GPU rounds run it as one extra block
// row of the matrix launch)
__global__ void __launch_bounds__(256) k_gather(KbDev d, KbRound r) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) reinterpret_cast<unsigned long long *>(r.result)[KB_OUT_STAMP0] = wall_clock64();   // round start (constant-rate clock)
  gather_row(d, r, i);
}

// nodeorder's NodeAffinity priority for the matrix rows whose task class has preferred terms (rare): Map = the class-pair count,
// Reduce = NormalizeReduce(10) over the row's FEASIBLE nodes (vendor/.../priorities/reduce.go:28-63: max == 0 leaves zeros,
// else 10 * count / max, integer division), then Score += score * weight (scheduler_helper.go:162-168).  One workgroup per row.
__global__ void __launch_bounds__(256) k_affinity(KbDev d, KbRound r) {
  __shared__ int s_max[4];
  const uint32_t row = blockIdx.x, tid = threadIdx.x;
  const uint32_t t = r.mrows ? r.mrows[row] : r.mrow_task0 + row;
  const uint32_t tc = d.t_cls[t];
  if (!d.aff_cls[tc]) return;   // uniform per block
  const int32_t *arow = d.aff + (size_t)tc * d.n_nc;
  const uint32_t *mw = r.maskw + (size_t)row * (d.NP / 32);
  uint16_t *sc = r.score + (size_t)row * d.NP;
  int mx = 0;
  for (uint32_t n = tid; n < d.N; n += 256)
    if ((mw[n >> 5] >> (n & 31)) & 1u) mx = max(mx, (int)arow[d.ncls[n]]);
  mx = wave_max_i32(mx);
  if ((tid & 63) == 0) s_max[tid >> 6] = mx;
  __syncthreads();
  mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
  if (mx == 0) return;
  for (uint32_t n = tid; n < d.N; n += 256)
    if ((mw[n >> 5] >> (n & 31)) & 1u) sc[n] = (uint16_t)(sc[n] + (10 * arow[d.ncls[n]] / mx) * d.wNA);
}

// LDS layout of the commit kernel for slot capacity `cap`:
//   8-byte tables [10][cap]: idle0 idle1 rel0 rel1 inv_ac inv_am ac am nzc nzm   (field k of slot s at (k*cap + s)*8)
//   4-byte tables [4][cap]:  cls node left cursor
//   KbRowDesc [cap], dirty bitmap [NP/32], header
#define K5F_IDLE0 0
#define K5F_IDLE1 1
#define K5F_REL0 2
#define K5F_REL1 3
#define K5F_INVAC 4
#define K5F_INVAM 5
#define K5F_AC 6
#define K5F_AM 7
#define K5F_NZC 8
#define K5F_NZM 9
#define K5_NF8 10
#define K5_WAVES (KB_K5_THREADS / 64)

__device__ __forceinline__ NodeVals k5_slot_vals(const unsigned long long *tab, const uint32_t *t_cls, const int *t_left, uint32_t cap, uint32_t slot,
                                              const unsigned long long *ptab = nullptr) {
  NodeVals nv;
  nv.idle0 = __longlong_as_double((long long)tab[K5F_IDLE0 * cap + slot]);
  nv.idle1 = __longlong_as_double((long long)tab[K5F_IDLE1 * cap + slot]);
  nv.rel0 = __longlong_as_double((long long)tab[K5F_REL0 * cap + slot]);
  nv.rel1 = __longlong_as_double((long long)tab[K5F_REL1 * cap + slot]);
  nv.inv_ac = __longlong_as_double((long long)tab[K5F_INVAC * cap + slot]);
  nv.inv_am = __longlong_as_double((long long)tab[K5F_INVAM * cap + slot]);
  nv.ac = (long long)tab[K5F_AC * cap + slot];
  nv.am = (long long)tab[K5F_AM * cap + slot];
  nv.nzc = (long long)tab[K5F_NZC * cap + slot];
  nv.nzm = (long long)tab[K5F_NZM * cap + slot];
  nv.ports = ptab ? ptab[slot] : 0ull;
  nv.cls = t_cls[slot];
  nv.slots = t_left[slot] > 0;
  nv.valid = 1;
  return nv;
}

// eval_pair for the commit kernel: policy scalars from the by-value argument struct, session arrays (scalar resource
// dimensions, wide class tables) through the device-memory copy of KbDev on the rare paths only
__device__ __forceinline__ uint32_t eval_pair_k5(const KbCommitArgs &a, const TaskVals &t, const NodeVals &n, uint32_t node, const uint32_t *class_row) {
  bool ok = true;
  if (a.fit_mode) {   // allocate.go:81
    bool fi = le_eps(t.init0, n.idle0, EPS_CPU) && le_eps(t.init1, n.idle1, EPS_MEM);
    bool fr = le_eps(t.init0, n.rel0, EPS_CPU) && le_eps(t.init1, n.rel1, EPS_MEM);
    uint32_t act = t.active >> 2;
    if (act) {
      const KbDev &d = *a.dev;
      uint32_t dd = 2;
      while (act) {
        if (act & 1u) {
          double l = d.t_init[(size_t)dd * d.T + t.task];
          fi = fi && le_eps(l, d.idle[(size_t)dd * d.NP + node], EPS_SCALAR);
          fr = fr && le_eps(l, d.rel[(size_t)dd * d.NP + node], EPS_SCALAR);
        }
        act >>= 1;
        dd++;
      }
    }
    ok = fi || fr;
  }
  if (a.pred_enabled) {
    ok = ok && n.slots && ((n.ports & t.conf) == 0ull);   // pod count (predicates.go:127), PodFitsHostPorts (predicates.go:181-190)
    if (class_row) {
      ok = ok && ((class_row[n.cls >> 5] >> (n.cls & 31)) & 1u);
    } else {
      const KbDev &d = *a.dev;
      if (d.compat) {
        uint32_t bit = t.cls * d.n_nc + n.cls;
        ok = ok && ((d.compat[bit >> 3] >> (bit & 7)) & 1);
      }
    }
  }
  if (!ok) return 0;
  uint32_t score = 0;
  if (a.score_enabled) score = score_core(t, n, a.wL, a.wM, a.wB);
  return 0x10000u | (score & 0xFFFFu);
}

// per-lane source arrays of the one-instruction node-state fetch: fld < 10 -> 8-byte field fld; 10..12 -> cls, maxpods, podcnt
__device__ __forceinline__ void k5_field_ptrs(const KbDev &d, uint32_t fld, const unsigned long long *&g8, const uint32_t *&g4) {
  g8 = nullptr; g4 = nullptr;
  switch (fld) {
    case K5F_IDLE0: g8 = reinterpret_cast<const unsigned long long *>(d.idle); break;
    case K5F_IDLE1: g8 = reinterpret_cast<const unsigned long long *>(d.idle + d.NP); break;
    case K5F_REL0: g8 = reinterpret_cast<const unsigned long long *>(d.rel); break;
    case K5F_REL1: g8 = reinterpret_cast<const unsigned long long *>(d.rel + d.NP); break;
    case K5F_INVAC: g8 = reinterpret_cast<const unsigned long long *>(d.inv_acpu); break;
    case K5F_INVAM: g8 = reinterpret_cast<const unsigned long long *>(d.inv_amem); break;
    case K5F_AC: g8 = reinterpret_cast<const unsigned long long *>(d.acpu); break;
    case K5F_AM: g8 = reinterpret_cast<const unsigned long long *>(d.amem); break;
    case K5F_NZC: g8 = reinterpret_cast<const unsigned long long *>(d.nzc); break;
    case K5F_NZM: g8 = reinterpret_cast<const unsigned long long *>(d.nzm); break;
    case 10: g4 = d.ncls; break;
    case 11: g4 = reinterpret_cast<const uint32_t *>(d.maxpods); break;
    case 12: g4 = reinterpret_cast<const uint32_t *>(d.podcnt); break;
    default: break;
  }
}

// ------------------------------------------------------------------------------------------------------------
// K5: commit.  One workgroup walks the window in the reference's task order (allocate.go:129-193 / backfill.go:44-67).
// The reference commits one task at a time, but ~90 % of the tasks take the best CLEAN node of their
// shape (the next untouched entry of the shape's sorted candidate list).  The kernel therefore speculates K7_B rows
// at once:
//   walk      one wave hands every row of the batch the next clean entry of its shape's list, in row order, marking
//             the nodes in the dirty bitmap as it goes (so a later row of another shape skips them);
//   fetch     16 threads per row pull the 13 state fields of the row's node into a NEW dirty slot (one load each);
//   apply     one thread per row decides Allocate / Pipeline (allocate.go:160) and applies NodeInfo.AddTask
//             (api/node_info.go:172-212) to its slot, i.e. the slot holds the state AFTER the row committed;
//   evaluate  all threads: key(shape q, slot x) for every distinct shape q of the batch and every dirty slot x, old
//             (-> dmax[q]) and new (-> kb[row][q]);
//   validate  row j really takes its clean candidate iff  c_j > dmax[q_j]  and  c_j > kb[l][q_j] for every earlier
//             batch row l: then no dirty node beats it, exactly the reference's arg-max.  The first row that fails is
//             the batch's "dirty row": its winner is the best dirty key, computed by the same evaluation;
//   commit    rows before the first failure are final (decision records, cursors); later rows are rolled back
//             (bitmap bits, speculative scalar-dimension writes) and re-speculated by the next batch; the dirty row is
//             applied to the slot that owns the winning node.
// A batch costs about as much as two rows of a row-at-a-time protocol, and commits ~10 rows on the benchmark snapshot.
// ------------------------------------------------------------------------------------------------------------
#define K7_B 32u          // most rows one batch can speculate
#define K7_B_DEFAULT 16u  // batch size after a batch that was cut short; doubled after a fully valid one
#define K7_D 96u   // row descriptors staged per refill

struct K7Hdr {
  unsigned long long c[K7_B];           // clean candidate key of batch row j (0: the list has no clean feasible node left)
  unsigned long long dmax[K7_B];        // per distinct shape q of the batch: best key over the pre-batch dirty slots
  unsigned long long kb[K7_B][K7_B];    // [row l][shape q]: key of row l's node in its post-commit state
  unsigned long long win[K7_B][64];     // candidate window of shape q, starting at win_base[q]
  uint32_t win_base[K7_B];
  uint32_t rep[K7_B];                   // batch row whose descriptor represents shape q
  uint32_t q_of[K7_B];                  // shape index of batch row j
  uint32_t idx[K7_B];                   // list position of c[j]
  uint32_t kind[K7_B];                  // 0 Allocate, 1 Pipeline
  uint32_t has_map[K7_B];               // the row's speculative commit wrote scalar dimensions (saved values are valid)
  // evaluation work list of shape q: slots [e_start, nd), then dlog[e_log0 .. e_log0 + e_nlog), then the batch's new slots
  uint32_t e_off[K7_B], e_start[K7_B], e_nlog[K7_B], e_log0[K7_B];
  KbRowDesc dbuf[K7_D];                 // row descriptors [dbase, dbase + dcnt) of the window, staged ahead of the batches
  unsigned long long kstar;             // winner of the dirty row
  uint32_t nshapes, p, dirty_row, reason, exhausted, pad;
  uint32_t n_pairs, nlog, n_full, pad3;
  uint32_t seq_rows, seq_pc, n_seq_rows, pad4;
  uint32_t n_batches, n_dirty_rows, n_refills, pad2;
};

struct K7Mem {
  unsigned long long *tab;              // [K5_NF8][cap2]
  uint32_t *t_cls, *t_node;             // [cap2]
  int *t_left;                          // [cap2]
  uint32_t *cursor;                     // [cap] per shape
  uint32_t *qstamp;                     // [cap] per shape: first batch row with that shape (0xFFFFFFFF between batches)
  // per shape: best key over the dirty slots [0, dc_nd) as of dirty-log position dc_log.  Still valid later for the slots
  // it covered unless the node of dc_key itself was changed by a dirty row since (the commit step then sets dc_nd to
  // 0xFFFFFFFF); newer slots and slots changed since are simply evaluated again and max'ed in.
  unsigned long long *dc_key;           // [cap]
  uint32_t *dc_nd, *dc_log;             // [cap]
  uint32_t *dlog;                       // [cap] slots changed by dirty rows, in order
  unsigned long long *keyq;             // [cap2] keys of one shape against every dirty slot (row-at-a-time mode)
  unsigned long long *ptab;             // [cap2] host-port bits of the slot's node (sessions with host ports only)
  uint32_t *bitmap;                     // [NP/32]
  double *save;                         // [K7_B][R-2] scalar-dimension values overwritten by speculative commits
  K7Hdr *H;
  uint32_t cap2;
};

__host__ __device__ inline size_t k7_smem_bytes(uint32_t cap, uint32_t NP, int R) {
  size_t cap2 = (size_t)cap + K7_B;
  return cap2 * (K5_NF8 * 8 + 8 + 8 + 3 * 4) + (size_t)cap * (8 + 8 + 12) + (size_t)(NP / 32) * 4 + (size_t)K7_B * (R > 2 ? R - 2 : 0) * 8 + sizeof(K7Hdr) + 64;
}

__device__ __forceinline__ TaskVals k7_task_vals(const KbCommitArgs &a, const KbRowDesc &k) {
  TaskVals tv;
  tv.init0 = k.init0; tv.init1 = k.init1; tv.nzc = k.nzc; tv.nzm = k.nzm;
  tv.cls = k.cls; tv.active = k.active; tv.task = k.task; tv.pad = 0;
  tv.conf = a.has_ports ? a.dev->t_conf[k.task] : 0ull;   // host-port sessions only: straight from the task table
  return tv;
}

// TaskInfo.Resreq cpu / memory of a row: equal to InitResreq unless an init container raised the latter (flags bit 0 clear,
// rare).  The LDS values are materialised before the branch so that the compiler does not turn "LDS address or global
// address" into flat loads (which wait on both memory counters).
__device__ __forceinline__ void k7_resreq(const KbCommitArgs &a, const KbRowDesc &k, double &res0, double &res1) {
  res0 = k.init0;
  res1 = k.init1;
  asm volatile("" : "+v"(res0), "+v"(res1));
  if (!(k.flags & 1)) { const KbDev &d = *a.dev; res0 = d.t_res[k.task]; res1 = d.t_res[(size_t)d.T + k.task]; }
}

// decision record + multi-GPU deltas of one committed row; SUB: also apply the scalar dimensions of NodeInfo.AddTask
// (the batched clean rows did that speculatively in the apply step)
template <bool SUB>
__device__ __forceinline__ void k7_commit_globals(const KbCommitArgs &a, const KbRowDesc &k, uint32_t i, uint32_t n, uint32_t kind) {
  const uint32_t km = k.resmask;
  uint32_t has_map = 0;
  if (km) {
    const KbDev &d = *a.dev;
    has_map = kind ? (d.nmask[n] >> 31) : (d.nmask[n] & 0x7FFFFFFFu);   // Sub returns early when the receiver's scalar map is nil (resource_info.go:148-153)
    if (SUB && has_map) {
      double *vec = kind ? d.rel : d.idle;
      uint32_t dd = 2, m2 = km;
      while (m2) {
        if (m2 & 1u) vec[(size_t)dd * d.NP + n] -= d.t_res[(size_t)dd * d.T + k.task];
        m2 >>= 1; dd++;
      }
    }
  }
  *reinterpret_cast<uint2 *>(&a.dec[i]) = make_uint2(n, kind);
  if (a.has_delta) {
    const KbDev &d = *a.dev;
    const KbRound &r = *a.round;
    if (i >= r.own_row0 && i < r.own_row1) {
      double res0, res1;
      k7_resreq(a, k, res0, res1);
      // per-node committed deltas of the rows this rank owns: [dIdle R][dRel R][dnzc][dnzm][dpodcnt] x NP
      double *dv = r.delta + (size_t)(kind ? d.R : 0) * d.NP;
      dv[n] -= res0;
      dv[(size_t)d.NP + n] -= res1;
      if (km && has_map) {
        uint32_t dd = 2, m2 = km;
        while (m2) {
          if (m2 & 1u) dv[(size_t)dd * d.NP + n] -= d.t_res[(size_t)dd * d.T + k.task];
          m2 >>= 1; dd++;
        }
      }
      double *tail = r.delta + (size_t)2 * d.R * d.NP;
      tail[n] += (double)k.nzc;
      tail[(size_t)d.NP + n] += (double)k.nzm;
      tail[(size_t)2 * d.NP + n] += 1.0;
    }
  }
}

// Allocate or Pipeline for row k on the node held in `slot` (allocate.go:160), then NodeInfo.AddTask on the LDS copy
__device__ __forceinline__ uint32_t k7_apply_slot(const KbCommitArgs &a, const K7Mem &M, const KbRowDesc &k, uint32_t slot, uint32_t n) {
  const uint32_t cap2 = M.cap2;
  double res0, res1;
  k7_resreq(a, k, res0, res1);
  uint32_t kind = 0;
  if (!a.backfill) {
    bool fi = le_eps(k.init0, __longlong_as_double((long long)M.tab[K5F_IDLE0 * cap2 + slot]), EPS_CPU) &&
              le_eps(k.init1, __longlong_as_double((long long)M.tab[K5F_IDLE1 * cap2 + slot]), EPS_MEM);
    uint32_t act = k.active >> 2;
    if (act) {
      const KbDev &d = *a.dev;
      uint32_t dd = 2;
      while (act) {
        if (act & 1u) fi = fi && le_eps(d.t_init[(size_t)dd * d.T + k.task], d.idle[(size_t)dd * d.NP + n], EPS_SCALAR);
        act >>= 1; dd++;
      }
    }
    kind = fi ? 0u : 1u;
  }
  const uint32_t f0 = kind ? K5F_REL0 : K5F_IDLE0;
  M.tab[(size_t)f0 * cap2 + slot] = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)M.tab[(size_t)f0 * cap2 + slot]) - res0);
  M.tab[(size_t)(f0 + 1) * cap2 + slot] = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)M.tab[(size_t)(f0 + 1) * cap2 + slot]) - res1);
  M.tab[(size_t)K5F_NZC * cap2 + slot] += (unsigned long long)k.nzc;
  M.tab[(size_t)K5F_NZM * cap2 + slot] += (unsigned long long)k.nzm;
  if (a.has_ports) M.ptab[slot] |= a.dev->t_want[k.task];   // the pod's host ports join nodeinfo.UsedPorts()
  return kind;
}

#ifdef KB_K5_TRACE
#define K7_STAMP(k) do { if (threadIdx.x == 0) { unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } } while (0)
#else
#define K7_STAMP(k) do { } while (0)
#endif

// The launch passes {hot scalars, KbDev, KbRound} as ONE by-value block.  Only `hot` is named in the code (-> SGPRs); the two
// views are reached through the kernel-argument segment pointer, i.e. read from constant memory where a rare path needs them.
struct K7KernArgs {
  KbCommitArgs hot;
  KbDev dev;
  KbRound round;
};

__global__ void __launch_bounds__(KB_K5_THREADS) k_commit(const K7KernArgs ka) {
  KbCommitArgs a = ka.hot;
  {
    const unsigned char __attribute__((address_space(4))) *kp = (const unsigned char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    a.dev = (const KbDev *)(kp + offsetof(K7KernArgs, dev));
    a.round = (const KbRound *)(kp + offsetof(K7KernArgs, round));
  }
  extern __shared__ __align__(16) unsigned char k5_smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cap = a.cap, cap2 = a.cap + K7_B;
  K7Mem M;
  M.cap2 = cap2;
  M.tab = reinterpret_cast<unsigned long long *>(k5_smem);
  M.keyq = M.tab + (size_t)K5_NF8 * cap2;   // 8-byte tables first
  M.ptab = M.keyq + cap2;
  M.dc_key = M.ptab + cap2;
  M.t_cls = reinterpret_cast<uint32_t *>(M.dc_key + cap);
  M.t_node = M.t_cls + cap2;
  M.t_left = reinterpret_cast<int *>(M.t_node + cap2);
  M.cursor = reinterpret_cast<uint32_t *>(M.t_left + cap2);
  M.qstamp = M.cursor + cap;
  M.dc_nd = M.qstamp + cap;
  M.dc_log = M.dc_nd + cap;
  M.dlog = M.dc_log + cap;
  M.bitmap = M.dlog + cap;
  {
    // byte offsets from the LDS base (pointer -> integer -> pointer round trips would lose the address space)
    size_t off = (size_t)cap2 * (K5_NF8 * 8 + 8 + 8 + 3 * 4) + (size_t)cap * (8 + 8 + 12) + (size_t)(a.NP / 32) * 4;
    off = (off + 15) & ~(size_t)15;
    M.H = reinterpret_cast<K7Hdr *>(k5_smem + off);
    M.save = reinterpret_cast<double *>(k5_smem + off + sizeof(K7Hdr));
  }
  K7Hdr &H = *M.H;
  const int RS = a.R > 2 ? a.R - 2 : 0;
  const unsigned long long t_start = wall_clock64();
#ifdef KB_K5_TRACE
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif

  for (uint32_t w = tid; w < a.NP / 32; w += KB_K5_THREADS) M.bitmap[w] = 0;
  for (uint32_t w = tid; w < cap; w += KB_K5_THREADS) { M.cursor[w] = 0; M.qstamp[w] = 0xFFFFFFFFu; M.dc_key[w] = 0ull; M.dc_nd[w] = 0; M.dc_log[w] = 0; }
  if (tid == 0) { H.reason = KB_REASON_DONE; H.exhausted = 0; H.n_batches = 0; H.n_dirty_rows = 0; H.n_refills = 0; H.p = 0; H.dirty_row = 0; H.pad = 0; H.nlog = 0; H.n_full = 0; H.pad2 = 0; H.pad3 = 0; H.kstar = 0ull; H.n_seq_rows = 0; }
  // per-thread source array of the fetch step: thread (row*16 + f) reads field f of the row's node
  // (pointers read from the KbDev copy are generic; the fetch step wants global_load, not flat_load)
  typedef const unsigned long long __attribute__((address_space(1))) *gptr8;
  typedef const uint32_t __attribute__((address_space(1))) *gptr4;
  gptr8 g8;
  gptr4 g4;
  {
    const unsigned long long *f8 = nullptr;
    const uint32_t *f4 = nullptr;
    k5_field_ptrs(*a.dev, tid & 15, f8, f4);
    g8 = (gptr8)f8;
    g4 = (gptr4)f4;
  }
  {
    // The loop is latency-bound and this workgroup starts on a cold L2 (kernel boundary): touch every 128-byte line of the
    // node state arrays and of the candidate lists once, with all threads.
    const KbDev &d = *a.dev;
    unsigned long long acc = 0;
    const uint32_t lines = a.NP / 16;
    const unsigned long long *arrs[10] = {
        reinterpret_cast<const unsigned long long *>(d.idle), reinterpret_cast<const unsigned long long *>(d.idle + d.NP),
        reinterpret_cast<const unsigned long long *>(d.rel), reinterpret_cast<const unsigned long long *>(d.rel + d.NP),
        reinterpret_cast<const unsigned long long *>(d.inv_acpu), reinterpret_cast<const unsigned long long *>(d.inv_amem),
        reinterpret_cast<const unsigned long long *>(d.acpu), reinterpret_cast<const unsigned long long *>(d.amem),
        reinterpret_cast<const unsigned long long *>(d.nzc), reinterpret_cast<const unsigned long long *>(d.nzm)};
#pragma unroll
    for (int f = 0; f < 10; f++)
      for (uint32_t l = tid; l < lines; l += KB_K5_THREADS) acc += arrs[f][(size_t)l * 16];
    const uint32_t *arr4[3] = {d.ncls, reinterpret_cast<const uint32_t *>(d.maxpods), reinterpret_cast<const uint32_t *>(d.podcnt)};
#pragma unroll
    for (int f = 0; f < 3; f++)
      for (uint32_t l = tid; l < a.NP / 32; l += KB_K5_THREADS) acc += arr4[f][(size_t)l * 32];
    const size_t klines = ((size_t)a.n_mrows * a.L + 15) / 16;
    for (size_t l = tid; l < klines; l += KB_K5_THREADS) acc += a.keys[l * 16];
    if (acc == 0x123456789abcdefull) H.pad = 1;   // keep the loads alive
  }
  __syncthreads();
  K7_STAMP(0);

  uint32_t i0 = 0, nd = 0, n_done = 0, reason = KB_REASON_DONE, dbase = 0, dcnt = 0, nb_cur = a.batch;
  while (i0 < a.n_rows) {
    uint32_t nb = min(nb_cur, a.n_rows - i0);
    // ---- stage row descriptors: three batches' worth per refill, so most batches find theirs already in LDS
    if (i0 < dbase || i0 + nb > dbase + dcnt) {
      dbase = i0;
      dcnt = min(K7_D, a.n_rows - i0);
      const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.desc + i0);
      unsigned long long *dst = reinterpret_cast<unsigned long long *>(H.dbuf);
      for (uint32_t w = tid; w < dcnt * (uint32_t)(sizeof(KbRowDesc) / 8); w += KB_K5_THREADS) dst[w] = src[w];
      __syncthreads();
    }
    const KbRowDesc *bd = H.dbuf + (i0 - dbase);
    if (a.has_aff && !a.backfill) {
      // a row whose score is normalised over its feasible set (preferred node affinity) is exact only against a fresh matrix:
      // it may be the first row of a round, nothing else; the batch stops in front of it and the round ends there
      uint32_t ja = nb;
      for (uint32_t j = (i0 == 0) ? 1u : 0u; j < nb; j++)
        if (bd[j].flags & 2) { ja = j; break; }
      if (ja == 0) { reason = KB_REASON_RENORM; n_done = i0; break; }
      nb = ja;
    }
    K7_STAMP(1);
    // ---- distinct shapes of the batch (wave 0): q_of[j] = rank of the first row with row j's shape
    if (wave == 0) {
      const bool in = lane < nb;
      const uint32_t s = in ? (uint32_t)bd[in ? lane : 0].slot : 0u;
      if (in) atomicMin(&M.qstamp[s], lane);
      const uint32_t first = in ? M.qstamp[s] : 0xFFFFFFFFu;
      const bool isrep = in && first == lane;
      const unsigned long long repmask = __ballot(isrep);
      // evaluation work list of every distinct shape: reuse the shape's cached dirty max when its node is untouched
      uint32_t cnt = 0, q = 0;
      if (in) {
        q = (uint32_t)__popcll(repmask & ((1ull << first) - 1ull));
        H.q_of[lane] = q;
        M.qstamp[s] = 0xFFFFFFFFu;
      }
      if (isrep) {
        const uint32_t nlog = H.nlog;
        unsigned long long ck = M.dc_key[s];
        uint32_t start = M.dc_nd[s];
        const uint32_t log0 = M.dc_log[s];
        uint32_t nl = nlog - log0;
        const bool full = start == 0xFFFFFFFFu || nl > 16;   // invalidated by a dirty row on its arg-max node, or too stale
        if (full) { start = 0; nl = 0; ck = 0ull; }
        cnt = (nd - start) + nl + nb;
        H.rep[q] = lane; H.win_base[q] = M.cursor[s]; H.dmax[q] = ck;
        H.e_start[q] = start; H.e_nlog[q] = nl; H.e_log0[q] = log0;
        if (full) atomicAdd(&H.n_full, 1u);
      }
      // exclusive prefix of the counts over the representative lanes
      const uint32_t incl = wave_incl_scan_u32(cnt);
      if (isrep) H.e_off[q] = incl - cnt;
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      const uint32_t nsh = (uint32_t)__popcll(repmask);
      if (lane >= nsh && lane < K7_B) H.e_off[lane] = 0xFFFFFFFFu;
      if (lane == 0) { H.nshapes = nsh; H.n_pairs = total; H.n_batches++; }
    }
    __syncthreads();
    // ---- candidate windows of the distinct shapes: one wave per shape
    const uint32_t nshapes = H.nshapes;
    for (uint32_t q = wave; q < nshapes; q += K5_WAVES) {
      const uint32_t s = bd[H.rep[q]].slot;
      const uint32_t e = H.win_base[q] + lane;
      H.win[q][lane] = (e < a.L) ? a.keys[(size_t)s * a.L + e] : 0ull;
    }
    __syncthreads();
    K7_STAMP(2);
    // ---- walk (wave 0): runs of consecutive rows with the same shape take successive clean entries of its window
    if (wave == 0) {
      const uint32_t myq = (lane < nb) ? H.q_of[lane] : 0xFFFFFFFFu;
      uint32_t j = 0;
      while (j < nb) {
        const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)myq, (int)j);
        const unsigned long long diff = __ballot(lane >= j && lane < nb && myq != q);
        const uint32_t j1 = diff ? (uint32_t)(__ffsll((unsigned long long)diff) - 1) : nb;
        uint32_t m = j1 - j;
        for (;;) {
          const unsigned long long wkey = H.win[q][lane];
          const uint32_t base = H.win_base[q];
          const bool nz = wkey != 0ull;
          const uint32_t node = KB_KEY_NODE(wkey);
          const bool cl = nz && !bit_test(M.bitmap, nz ? node : 0u);
          const unsigned long long clean = __ballot(cl);
          const unsigned long long zeros = __ballot(!nz);
          const uint32_t cnt = (uint32_t)__popcll(clean);
          const uint32_t take = cnt < m ? cnt : m;
          const uint32_t rank = (uint32_t)__popcll(clean & ((1ull << lane) - 1ull));
          if (cl && rank < take) {
            H.c[j + rank] = wkey;
            H.idx[j + rank] = base + lane;
            atomicOr(&M.bitmap[node >> 5], 1u << (node & 31));
          }
          j += take;
          m -= take;
          if (m == 0) break;
          if (zeros) {   // the list ended: no clean feasible node is left for the remaining rows of the run
            if (lane < m) { H.c[j + lane] = 0ull; H.idx[j + lane] = 0; }
            j += m;
            break;
          }
          // every entry of the window is dirty: slide it (entries before it stay dirty for the rest of the round or are
          // rolled back together with this batch)
          const uint32_t nbase = base + 64;
          if (nbase >= a.L) {   // cannot happen while L > window: reported, never silently mis-scheduled
            if (lane == 0) H.exhausted = 1;
            if (lane < m) { H.c[j + lane] = 0ull; H.idx[j + lane] = 0; }
            j += m;
            break;
          }
          const uint32_t e = nbase + lane;
          const uint32_t s = bd[H.rep[q]].slot;
          H.win[q][lane] = (e < a.L) ? a.keys[(size_t)s * a.L + e] : 0ull;
          if (lane == 0) { H.win_base[q] = nbase; H.n_refills++; }
        }
      }
    }
    __syncthreads();
    K7_STAMP(3);
    // ---- fetch + apply: the 16 lanes of one DPP row handle one batch row.  Lane f reads field f of the row's candidate
    //      node (one load), the group votes Allocate / Pipeline (allocate.go:160) with a ballot, every lane applies
    //      NodeInfo.AddTask (api/node_info.go:172-212) to its own field and stores it into the row's NEW dirty slot: the slot
    //      holds the node's state AFTER the row committed.  Scalar dimensions (global memory, rare) are written
    //      speculatively by lane 15 and their old values saved for the rollback.
    for (uint32_t w = tid; w < nb * 16; w += KB_K5_THREADS) {
      const uint32_t j = w >> 4, f = w & 15;
      const unsigned long long cj = H.c[j];
      uint32_t kind = 0, has_map = 0;
      if (cj) {
        const KbRowDesc &k = bd[j];
        const uint32_t n = KB_KEY_NODE(cj), slot = nd + j;
        unsigned long long v8 = 0ull;
        uint32_t v4 = 0;
        if (f < K5_NF8) v8 = g8[n];
        else if (f <= 12) v4 = g4[n];
        double res0, res1;
        k7_resreq(a, k, res0, res1);
        bool ok = true;
        if (!a.backfill) {
          const double dv = __longlong_as_double((long long)v8);
          if (f == K5F_IDLE0) ok = le_eps(k.init0, dv, EPS_CPU);
          else if (f == K5F_IDLE1) ok = le_eps(k.init1, dv, EPS_MEM);
          else if (f == 15 && (k.active >> 2)) {
            const KbDev &d = *a.dev;
            uint32_t act = k.active >> 2, dd = 2;
            while (act) {
              if (act & 1u) ok = ok && le_eps(d.t_init[(size_t)dd * d.T + k.task], d.idle[(size_t)dd * d.NP + n], EPS_SCALAR);
              act >>= 1; dd++;
            }
          }
        }
        const unsigned long long bad = __ballot(!ok);
        kind = ((bad >> (lane & 48u)) & 0xFFFFull) ? 1u : 0u;
        const uint32_t f0 = kind ? K5F_REL0 : K5F_IDLE0;
        const double dv = __longlong_as_double((long long)v8);
        if (f == f0) v8 = (unsigned long long)__double_as_longlong(dv - res0);
        else if (f == f0 + 1) v8 = (unsigned long long)__double_as_longlong(dv - res1);
        else if (f == K5F_NZC) v8 += (unsigned long long)k.nzc;
        else if (f == K5F_NZM) v8 += (unsigned long long)k.nzm;
        if (f < K5_NF8) M.tab[(size_t)f * cap2 + slot] = v8;
        const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v4, 0x101, 0xf, 0xf, true);   // row_shl:1: lane 11 <- pods
        if (f == 10) M.t_cls[slot] = v4;
        else if (f == 11) M.t_left[slot] = (int)v4 - (int)nxt - 1;
        else if (f == 13) {
          M.t_node[slot] = n;
          if (a.has_ports) M.ptab[slot] = a.dev->ports[n] | a.dev->t_want[k.task];
        }
        else if (f == 15 && k.resmask) {
          const KbDev &d = *a.dev;
          has_map = kind ? (d.nmask[n] >> 31) : (d.nmask[n] & 0x7FFFFFFFu);   // Sub returns early when the receiver's scalar map is nil (resource_info.go:148-153)
          if (has_map) {
            double *vec = kind ? d.rel : d.idle;
            uint32_t dd = 2, m2 = k.resmask;
            while (m2) {
              if (m2 & 1u) {
                const double old = vec[(size_t)dd * d.NP + n];
                M.save[(size_t)j * RS + (dd - 2)] = old;
                vec[(size_t)dd * d.NP + n] = old - d.t_res[(size_t)dd * d.T + k.task];
              }
              m2 >>= 1; dd++;
            }
          }
        }
      }
      if (f == 14) H.kind[j] = kind;
      if (f == 15) H.has_map[j] = has_map;
    }
    __syncthreads();
    K7_STAMP(5);
    // ---- evaluate the work lists: (shape q, slot x) -> dmax[q] for slots older than the batch, kb[row][q] for its own
    {
      const uint32_t P = H.n_pairs;
      uint32_t off[K7_B];
#pragma unroll
      for (int k = 0; k < (int)K7_B; k += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&H.e_off[k]);
        off[k] = v.x; off[k + 1] = v.y; off[k + 2] = v.z; off[k + 3] = v.w;
      }
      for (uint32_t e = tid; e < P; e += KB_K5_THREADS) {
        uint32_t q = 0;
#pragma unroll
        for (int k = 1; k < (int)K7_B; k++) q += (e >= off[k]) ? 1u : 0u;
        uint32_t r = e - H.e_off[q];
        const uint32_t start = H.e_start[q], nn = nd - start, nl = H.e_nlog[q];
        uint32_t x;
        if (r < nn) x = start + r;
        else if (r < nn + nl) x = M.dlog[H.e_log0[q] + (r - nn)];
        else x = nd + (r - nn - nl);
        unsigned long long key = 0ull;
        if (x < nd || H.c[x - nd] != 0ull) {
          const KbRowDesc &k = bd[H.rep[q]];
          const TaskVals tv = k7_task_vals(a, k);
          const NodeVals nv = k5_slot_vals(M.tab, M.t_cls, M.t_left, cap2, x, a.has_ports ? M.ptab : nullptr);
          const uint32_t node = M.t_node[x];
          const uint32_t res = eval_pair_k5(a, tv, nv, node, a.use_crow ? &k.crow : nullptr);
          key = res ? KB_KEY(res & 0xFFFFu, node) : 0ull;
        }
        if (x < nd) { if (key) atomicMax(&H.dmax[q], key); }
        else H.kb[x - nd][q] = key;
      }
    }
    __syncthreads();
    K7_STAMP(6);
    // ---- validate (wave 0)
    if (wave == 0) {
      const bool in = lane < nb;
      const unsigned long long cj = in ? H.c[lane] : 0ull;
      const uint32_t q = in ? H.q_of[lane] : 0u;
      unsigned long long m = in ? H.dmax[q] : 0ull;
#pragma unroll 8
      for (uint32_t l = 0; l + 1 < nb; l++) {
        const unsigned long long kk = H.kb[l][q];
        if (l < lane && kk > m) m = kk;
      }
      if (in && H.rep[q] == lane) {   // the shape's dirty max as of this batch's start becomes its cache
        const uint32_t s = bd[lane].slot;
        M.dc_key[s] = H.dmax[q]; M.dc_nd[s] = nd; M.dc_log[s] = H.nlog;
      }
      const bool valid = in && cj != 0ull && cj > m;
      const unsigned long long inval = __ballot(in && !valid);
      const unsigned long long pipe = __ballot(valid && H.kind[in ? lane : 0] != 0u);
      uint32_t p = inval ? (uint32_t)(__ffsll((unsigned long long)inval) - 1) : nb;
      const unsigned long long pipe_before = pipe & ((1ull << p) - 1ull);
      uint32_t dirty_row = 0, rsn = KB_REASON_DONE;
      unsigned long long kstar = 0ull;
      if (pipe_before) {   // a Pipeline ends the speculation (the host re-plans): commit up to and including that row
        p = (uint32_t)__ffsll((unsigned long long)pipe_before);
        rsn = KB_REASON_PIPELINED;
      } else if (p < nb) {
        kstar = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(m >> 32), (int)p) << 32) |
                (uint32_t)__builtin_amdgcn_readlane((int)(m & 0xFFFFFFFFull), (int)p);
        if (kstar) dirty_row = 1;
        else if (a.backfill) dirty_row = 2;   // backfill.go:50-66: no node passes the predicates -> the task stays Pending
        else rsn = KB_REASON_NO_FEASIBLE;     // allocate.go:144-148: the job is abandoned; the host re-plans from here
      }
      if (lane == 0) {
#ifdef KB_K5_TRACE
        if (dirty_row == 1 && p == 0) { H.pad2++; if (KB_KEY_NODE(kstar) == KB_KEY_NODE(H.kstar)) H.pad3++; }
        if (a.trace) { a.trace[16 + (p < 16 ? p : 16)] += 1; }
#endif
        H.p = p; H.dirty_row = dirty_row; H.reason = rsn; H.kstar = kstar;
      }
    }
    __syncthreads();
    K7_STAMP(7);
    // ---- commit the valid prefix
    const uint32_t p = H.p, dirty_row = H.dirty_row;
    uint32_t pc = p, rows = p + (dirty_row == 2 ? 1u : 0u);   // candidates consumed, rows consumed
    if (tid < p) {
      const uint32_t j = tid;
      const KbRowDesc &k = bd[j];
      atomicMax(&M.cursor[k.slot], H.idx[j] + 1);
      k7_commit_globals<false>(a, k, i0 + j, KB_KEY_NODE(H.c[j]), H.kind[j]);
    }
    if (dirty_row == 2 && tid == 0) *reinterpret_cast<uint2 *>(&a.dec[i0 + p]) = make_uint2(KB_NONE_U32, 0u);
    if (dirty_row == 1) {
      // ---- a dirty node beats row p's clean candidate.  Dirty winners come in chains (a big node keeps the best score for
      // several tasks), so the rest of row p's run of same-shape rows is committed one row at a time by wave 0 alone, with no
      // workgroup barrier per row: every thread first evaluates the shape against all dirty slots (keyq), then per row
      //     winner = max( max(keyq) , next unconsumed clean candidate of the run )
      // a dirty winner's slot is updated in LDS and its key re-evaluated by one lane; a clean winner takes the slot the fetch /
      // apply steps already prepared for that candidate (same shape => same post-commit state) with the key the evaluate step
      // already computed (kb).
      const uint32_t q = H.q_of[p];
      {
        const KbRowDesc &k = bd[p];
        const TaskVals tv = k7_task_vals(a, k);
        for (uint32_t x = tid; x < nd + p; x += KB_K5_THREADS) {
          const NodeVals nv = k5_slot_vals(M.tab, M.t_cls, M.t_left, cap2, x, a.has_ports ? M.ptab : nullptr);
          const uint32_t node = M.t_node[x];
          const uint32_t res = eval_pair_k5(a, tv, nv, node, a.use_crow ? &k.crow : nullptr);
          M.keyq[x] = res ? KB_KEY(res & 0xFFFFu, node) : 0ull;
        }
      }
      __syncthreads();
      K7_STAMP(9);
      if (wave == 0) {
        const uint32_t myq = (lane < nb) ? H.q_of[lane] : 0xFFFFFFFFu;
        const unsigned long long diff = __ballot(lane > p && lane < nb && myq != q);
        const uint32_t run_end = diff ? (uint32_t)(__ffsll((unsigned long long)diff) - 1) : nb;
        const uint32_t shape = bd[p].slot;
        uint32_t r = p, ndc = nd + p, rsn = KB_REASON_DONE, last_n = 0xFFFFFFFFu, nlog = H.nlog;
        while (r < run_end) {
          unsigned long long kmax = 0ull;
          uint32_t xmax = 0;
          for (uint32_t x = lane; x < ndc; x += 64) {
            const unsigned long long kk = M.keyq[x];
            if (kk > kmax) { kmax = kk; xmax = x; }
          }
          const unsigned long long best = wave_max_key(kmax);
          const unsigned long long cc = H.c[pc];
          const KbRowDesc &k = bd[r];
          if (best == 0ull && cc == 0ull) {
            if (a.backfill) {   // backfill.go:50-66: the task stays Pending
              if (lane == 0) *reinterpret_cast<uint2 *>(&a.dec[i0 + r]) = make_uint2(KB_NONE_U32, 0u);
              r++;
              continue;
            }
            rsn = KB_REASON_NO_FEASIBLE;   // allocate.go:144-148
            break;
          }
          if (best > cc) {
            const unsigned long long own = __ballot(kmax == best);
            const uint32_t xs = (uint32_t)__builtin_amdgcn_readlane((int)xmax, __ffsll((unsigned long long)own) - 1);
            const uint32_t n = KB_KEY_NODE(best);
            // shapes whose cached dirty max sits on the node that changes lose their cache (once per node of a chain)
            if (n != last_n) {
              for (uint32_t sh = lane; sh < a.n_mrows; sh += 64) {
                const unsigned long long ck = M.dc_key[sh];
                if (ck != 0ull && KB_KEY_NODE(ck) == n) M.dc_nd[sh] = 0xFFFFFFFFu;
              }
              last_n = n;
            }
            uint32_t kind = 0;
            if (lane == 0) {
              kind = k7_apply_slot(a, M, k, xs, n);
              M.t_left[xs] -= 1;
              k7_commit_globals<true>(a, k, i0 + r, n, kind);
              M.dlog[nlog] = xs;
              const TaskVals tv = k7_task_vals(a, k);
              const NodeVals nv = k5_slot_vals(M.tab, M.t_cls, M.t_left, cap2, xs, a.has_ports ? M.ptab : nullptr);
              const uint32_t res = eval_pair_k5(a, tv, nv, n, a.use_crow ? &k.crow : nullptr);
              M.keyq[xs] = res ? KB_KEY(res & 0xFFFFu, n) : 0ull;
            }
            kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)kind);
            nlog++;
            r++;
            if (kind) { rsn = KB_REASON_PIPELINED; break; }
          } else {
            // the prepared slot nd+pc holds candidate pc's node after ROW pc's task; identical for row r's task when both
            // rows carry plain requests (same shape; no init-container maximum, no scalar resources)
            const KbRowDesc &kc = bd[pc];
            const uint32_t plain = (uint32_t)(k.flags & kc.flags & 1) && k.resmask == 0 && kc.resmask == 0;
            if (!plain) break;
            const uint32_t kind = H.kind[pc];
            if (lane == 0) {
              atomicMax(&M.cursor[shape], H.idx[pc] + 1);
              k7_commit_globals<false>(a, k, i0 + r, KB_KEY_NODE(cc), kind);
              M.keyq[ndc] = H.kb[pc][q];
            }
            ndc++; pc++; r++;
            if (kind) { rsn = KB_REASON_PIPELINED; break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        // the shape's dirty max as of now becomes its cache
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned long long kmax = 0ull;
        for (uint32_t x = lane; x < ndc; x += 64) { const unsigned long long kk = M.keyq[x]; if (kk > kmax) kmax = kk; }
        kmax = wave_max_key(kmax);
        if (lane == 0) {
          M.dc_key[shape] = kmax; M.dc_nd[shape] = ndc; M.dc_log[shape] = nlog;
          H.n_dirty_rows += nlog - H.nlog;
          H.nlog = nlog;
          H.seq_rows = r; H.seq_pc = pc; H.reason = rsn; H.n_seq_rows += r - p;
        }
      }
      __syncthreads();
      rows = H.seq_rows;
      pc = H.seq_pc;
    }
    // ---- roll back the candidates nobody consumed: they are re-speculated by the next batch
    if (tid < nb && tid >= pc) {
      const uint32_t j = tid;
      const unsigned long long cj = H.c[j];
      if (cj) {
        const KbRowDesc &k = bd[j];
        const uint32_t n = KB_KEY_NODE(cj);
        atomicAnd(&M.bitmap[n >> 5], ~(1u << (n & 31)));
        if (H.has_map[j]) {
          const KbDev &d = *a.dev;
          double *vec = H.kind[j] ? d.rel : d.idle;
          uint32_t dd = 2, m2 = k.resmask;
          while (m2) {
            if (m2 & 1u) vec[(size_t)dd * d.NP + n] = M.save[(size_t)j * RS + (dd - 2)];
            m2 >>= 1; dd++;
          }
        }
      }
    }
    __syncthreads();
    K7_STAMP(8);
    nd += pc;
    i0 += rows;
    // clean streaks are long (56 % of the batches commit every row): speculate twice as many rows after a fully valid batch
    nb_cur = (dirty_row == 0 && p == nb) ? min(2u * a.batch, K7_B) : a.batch;
    n_done = i0;
    reason = H.reason;
    if (H.exhausted) reason = KB_REASON_INTERNAL;
    if (reason != KB_REASON_DONE) break;
  }

  // ---- write the dirty nodes' live state back to HBM
  __syncthreads();
  {
    const KbDev &d = *a.dev;
    for (uint32_t slot = tid; slot < nd; slot += KB_K5_THREADS) {
      const uint32_t n = M.t_node[slot];
      d.idle[n] = __longlong_as_double((long long)M.tab[K5F_IDLE0 * cap2 + slot]);
      d.idle[(size_t)d.NP + n] = __longlong_as_double((long long)M.tab[K5F_IDLE1 * cap2 + slot]);
      d.rel[n] = __longlong_as_double((long long)M.tab[K5F_REL0 * cap2 + slot]);
      d.rel[(size_t)d.NP + n] = __longlong_as_double((long long)M.tab[K5F_REL1 * cap2 + slot]);
      d.nzc[n] = (long long)M.tab[K5F_NZC * cap2 + slot];
      d.nzm[n] = (long long)M.tab[K5F_NZM * cap2 + slot];
      d.podcnt[n] = d.maxpods[n] - M.t_left[slot];
      if (a.has_ports) d.ports[n] = M.ptab[slot];
    }
  }
  // task-table side of ssn.Allocate / ssn.Pipeline for the committed rows (job.UpdateTaskStatus, task.NodeName:
  // framework/session.go:243,205; api/node_info.go:206-209)
  {
    const KbDev &d = *a.dev;
    for (uint32_t i = tid; i < n_done; i += KB_K5_THREADS) {
      const uint2 dc = *reinterpret_cast<const uint2 *>(&a.dec[i]);
      if (dc.x == KB_NONE_U32) continue;
      const uint32_t t = a.desc[i].task;
      d.t_status[t] = dc.y ? KB_TASK_PIPELINED : KB_TASK_ALLOCATED;
      d.t_node[t] = dc.x;
      d.t_counted[t] = 1;
      if (!dc.y) d.j_allocated[d.t_job[t]] = 1;   // ssn.Allocate ran for the job: its Allocated tasks are dispatched if it is ready
    }
  }
  if (tid == 0) {
    a.result[0] = n_done; a.result[1] = reason; a.result[2] = nd; a.result[4] = H.n_refills; a.result[7] = H.n_full; a.result[3] = H.n_seq_rows;
    a.result[5] = H.n_batches; a.result[6] = H.n_dirty_rows;
    unsigned long long *st = reinterpret_cast<unsigned long long *>(a.result) + KB_OUT_STAMP0;
    st[2] = t_start;
    st[3] = wall_clock64();
#ifdef KB_K5_TRACE
    if (a.trace) { for (int k = 0; k < 12; k++) a.trace[k] = tacc[k]; a.trace[12] = H.pad2; a.trace[13] = H.pad3; }
#endif
  }
  // ---- fast rounds: mirror the header and the decision records into pinned host memory and publish the round's sequence
  //      number last; the host spins on that word instead of paying a stream synchronisation + D2H copy per round
  if (a.host_out) {
    __syncthreads();
    const unsigned long long *hdr = reinterpret_cast<const unsigned long long *>(a.result);
    for (uint32_t i = tid; i < KB_OUT_SEQ; i += KB_K5_THREADS) a.host_out[i] = hdr[i];
    for (uint32_t i = tid; i < n_done; i += KB_K5_THREADS) a.host_out[KB_OUT_HDR + i] = a.dec[i];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&a.host_out[KB_OUT_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------------------------------------------------
// K2 + K4: one wave per job.  Ready count by ballot/popcount, gang-gated Allocated -> Binding flip, segmented sum
// of Resreq per job (drf) and per queue (proportion, f64 atomics: the addends are integer-valued milli-units /
// bytes below 2^53, so the sum is exact and order-independent), then the share maxima.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double share_of(double l, double r) { return (r == 0.0) ? ((l == 0.0) ? 0.0 : 1.0) : l / r; }

__global__ void __launch_bounds__(256) k_finalize_jobs(KbDev d, const uint32_t *job_task_begin, const int *job_min_avail,
                                                       const uint32_t *job_queue, int gang_ready_enabled, const double *total,
                                                       uint32_t total_mask, double *job_alloc, double *job_share, double *queue_alloc,
                                                       int *job_ready_cnt) {
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t j = blockIdx.x * 4 + wave;
  if (j >= d.J) return;
  const uint32_t t0 = job_task_begin[j], t1 = job_task_begin[j + 1];
  int ready = 0;
  for (uint32_t base = t0; base < t1; base += 64) {
    uint32_t t = base + lane;
    int st = (t < t1) ? (int)d.t_status[t] : -1;
    bool isr = st == KB_TASK_BOUND || st == KB_TASK_BINDING || st == KB_TASK_RUNNING || st == KB_TASK_ALLOCATED || st == KB_TASK_SUCCEEDED;
    ready += __popcll(__ballot(isr));     // JobInfo.ReadyTaskNum (job_info.go:383-394)
  }
  const bool job_ready = gang_ready_enabled ? (ready >= job_min_avail[j]) : true;   // gang.go:122-125 / session_plugins.go:182-200
  // session.go:277-285: the dispatch sits inside ssn.Allocate, so it needs an Allocate on this job in this action; the ready
  // count only changes through ssn.Allocate here, hence "ready after the job's last Allocate" == "ready now".  Every task in
  // TaskStatusIndex[Allocated] goes, including ones the snapshot already carried as Allocated.
  if (job_ready && d.j_allocated[j]) {
    for (uint32_t t = t0 + lane; t < t1; t += 64)
      if (d.t_status[t] == KB_TASK_ALLOCATED) { d.t_status[t] = KB_TASK_BINDING; d.t_bind[t] = d.t_node[t]; }
  }
  if (lane == 0) { job_ready_cnt[j] = ready; d.j_allocated[j] = 0; }
  const uint32_t q = job_queue[j];
  double share = 0.0;
  for (int dim = 0; dim < d.R; dim++) {
    double s = 0.0;
    for (uint32_t t = t0 + lane; t < t1; t += 64)
      if (d.t_counted[t]) s += d.t_res[(size_t)dim * d.T + t];
    s = wave_sum_f64(s);
    if (lane == 0) {
      job_alloc[(size_t)j * d.R + dim] = s;
      if (s != 0.0) atomicAdd(&queue_alloc[(size_t)q * d.R + dim], s);
      if (dim < 2 || ((total_mask >> (dim - 2)) & 1u)) {   // totalResource.ResourceNames() (drf.go:161)
        double sh = share_of(s, total[dim]);
        if (sh > share) share = sh;
      }
    }
  }
  if (lane == 0) job_share[j] = share;
}

__global__ void k_finalize_queues(KbDev d, const double *deserved, const uint32_t *deserved_mask, const double *queue_alloc, double *queue_share) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= d.Q) return;
  double share = 0.0;
  for (int dim = 0; dim < d.R; dim++) {
    if (dim >= 2 && !((deserved_mask[q] >> (dim - 2)) & 1u)) continue;   // attr.deserved.ResourceNames() (proportion.go:245)
    double sh = share_of(queue_alloc[(size_t)q * d.R + dim], deserved[(size_t)dim * d.Q + q]);
    if (sh > share) share = sh;
  }
  queue_share[q] = share;
}

// ------------------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------------------
// multi-GPU: install round-start state + all-reduced deltas, counting values that differ from the replica's own commit
__global__ void __launch_bounds__(256) k_apply_deltas(KbDev d, const double *s_idle, const double *s_rel, const long long *s_nzc,
                                                      const long long *s_nzm, const int *s_podcnt, const double *delta, uint32_t *counter) {
  uint32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= d.NP) return;
  uint32_t bad = 0;
  for (int dim = 0; dim < d.R; dim++) {
    size_t o = (size_t)dim * d.NP + n;
    double vi = s_idle[o] + delta[o];
    double vr = s_rel[o] + delta[(size_t)d.R * d.NP + o];
    bad += (vi != d.idle[o]) + (vr != d.rel[o]);
    d.idle[o] = vi;
    d.rel[o] = vr;
  }
  const double *tail = delta + (size_t)2 * d.R * d.NP;
  long long c = s_nzc[n] + (long long)tail[n], m = s_nzm[n] + (long long)tail[(size_t)d.NP + n];
  int p = s_podcnt[n] + (int)tail[(size_t)2 * d.NP + n];
  bad += (c != d.nzc[n]) + (m != d.nzm[n]) + (p != d.podcnt[n]);
  d.nzc[n] = c; d.nzm[n] = m; d.podcnt[n] = p;
  if (bad) atomicAdd(counter, bad);
}
uint32_t kb_apply_deltas(const KbDev &d, const double *s_idle, const double *s_rel, const long long *s_nzc, const long long *s_nzm,
                         const int *s_podcnt, const double *delta, uint32_t *dev_counter, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(dev_counter, 0, sizeof(uint32_t), s);
  hipLaunchKernelGGL(k_apply_deltas, dim3((d.NP + 255) / 256), dim3(256), 0, s, d, s_idle, s_rel, s_nzc, s_nzm, s_podcnt, delta, dev_counter);
  uint32_t h = 0;
  (void)hipMemcpyAsync(&h, dev_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  return h;
}
size_t kb_commit_smem_bytes(uint32_t cap, uint32_t NP, int R) { return k7_smem_bytes(cap, NP, R); }
void kb_launch_gather(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  hipLaunchKernelGGL(k_gather, dim3((r.n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, r);
}
void kb_launch_matrix(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_mrows == 0) return;
  if ((size_t)r.n_mrows * d.NP >= (4u << 20)) {
    dim3 grid(d.NP / (256 * 4), (r.n_mrows + 31) / 32 + (r.gather ? 1 : 0));
    hipLaunchKernelGGL((k_matrix<4, 32>), grid, dim3(256), 0, (hipStream_t)stream, d, r);
  } else {
    dim3 grid(d.NP / 256, (r.n_mrows + 3) / 4 + (r.gather ? 1 : 0));
    hipLaunchKernelGGL((k_matrix<1, 4>), grid, dim3(256), 0, (hipStream_t)stream, d, r);
  }
}
void kb_launch_affinity(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_mrows == 0 || !d.aff || !d.score_enabled) return;
  hipLaunchKernelGGL(k_affinity, dim3(r.n_mrows), dim3(256), 0, (hipStream_t)stream, d, r);
}
void kb_launch_expand(const KbDev &d, const uint16_t *s_score, const uint32_t *s_mask, const uint32_t *row_slot, uint32_t n_rows,
                      uint16_t *score, uint32_t *maskw, void *stream) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(k_expand, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, s_score, s_mask, row_slot, n_rows, d.NP, score, maskw);
}
void kb_launch_argmax(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_mrows == 0) return;
  size_t sh = (size_t)d.NP * 2 + d.NP / 8;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_argmax<false, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_argmax<false, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_argmax<true, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  // few rows (a round's distinct shapes): the launch is latency-sized, 1024 threads shorten every pass over the row; many rows
  // (kb_argmax_rows over whole task ranges): 256 threads keep more rows resident per CU
  if (d.NP >= 65536u) hipLaunchKernelGGL((k_argmax<true, 1024>), dim3(r.n_mrows), dim3(1024), sh, st, d, r);
  else if (r.n_mrows <= 512) hipLaunchKernelGGL((k_argmax<false, 1024>), dim3(r.n_mrows), dim3(1024), sh, st, d, r);
  else hipLaunchKernelGGL((k_argmax<false, 256>), dim3(r.n_mrows), dim3(256), sh, st, d, r);
}
void kb_launch_commit(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  static bool attr_set = false;
  static uint32_t env_batch = 0;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_commit), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const char *b = getenv("KB_K5_BATCH");   // tuning override of kb_config.commit_batch
    env_batch = b ? (uint32_t)atoi(b) : 0;
    attr_set = true;
  }
  uint32_t batch = env_batch ? env_batch : (r.batch ? r.batch : K7_B_DEFAULT);
  if (batch > K7_B) batch = K7_B;
  const size_t sh = k7_smem_bytes(r.cap, d.NP, d.R);
  K7KernArgs ka;
  ka.dev = d;
  ka.round = r;
  KbCommitArgs &a = ka.hot;
  a.dev = nullptr; a.round = nullptr;   // set from the kernel-argument segment inside the kernel
  a.keys = r.keys; a.dec = r.dec; a.desc = r.desc; a.result = r.result; a.trace = r.trace;
  a.n_rows = r.n_rows; a.n_mrows = r.n_mrows; a.L = r.L; a.cap = r.cap; a.N = d.N; a.NP = d.NP;
  a.fit_mode = r.fit_mode; a.backfill = r.backfill; a.pred_enabled = d.pred_enabled; a.score_enabled = d.score_enabled;
  a.wL = d.wL; a.wM = d.wM; a.wB = d.wB;
  a.use_crow = (d.pred_enabled && d.crows != nullptr && d.n_nc <= 32) ? 1u : 0u;
  a.has_delta = r.delta != nullptr ? 1u : 0u;
  a.has_aff = (d.aff != nullptr && d.score_enabled) ? 1u : 0u;
  a.has_ports = d.ports != nullptr ? 1u : 0u;
  a.R = d.R;
  a.batch = batch;
  a.host_out = r.host_out;
  a.seq = r.seq;
  hipLaunchKernelGGL(k_commit, dim3(1), dim3(KB_K5_THREADS), sh, (hipStream_t)stream, ka);
}
void kb_launch_finalize(const KbDev &d, const uint32_t *job_task_begin, const int *job_min_avail, const uint32_t *job_queue,
                        int gang_ready_enabled, const double *total, uint32_t total_mask, const double *deserved,
                        const uint32_t *deserved_mask, double *job_alloc, double *job_share, double *queue_alloc,
                        double *queue_share, int *job_ready_cnt, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(queue_alloc, 0, sizeof(double) * (size_t)d.Q * d.R, s);
  if (d.J) hipLaunchKernelGGL(k_finalize_jobs, dim3((d.J + 3) / 4), dim3(256), 0, s, d, job_task_begin, job_min_avail, job_queue,
                              gang_ready_enabled, total, total_mask, job_alloc, job_share, queue_alloc, job_ready_cnt);
  if (d.Q) hipLaunchKernelGGL(k_finalize_queues, dim3((d.Q + 127) / 128), dim3(128), 0, s, d, deserved, deserved_mask, queue_alloc, queue_share);
}
