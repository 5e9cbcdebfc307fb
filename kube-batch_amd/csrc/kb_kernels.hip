// kb_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for kube-batch's allocate/backfill hot path.
//
//   K1 k_matrix    mask + score matrix for a window of task rows x all nodes
//                  = allocate.go:73-87 predicate closure + plugins/predicates/predicates.go:123-265 (pod-count cap,
//                    static checks as a class bit table) + nodeorder's LeastRequested / MostRequested /
//                    BalancedResourceAllocation (vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/*.go)
//                    summed as util.PrioritizeNodes does (scheduler_helper.go:162-168).
//   K3 k_argmax    segmented per-row top-K arg-max = util.SelectBestNode (scheduler_helper.go:188-208), first max in
//                  ascending node order, generalised to K candidates.
//   K5 k_commit    the sequential part of allocate.go:129-193 / backfill.go:44-67: for each row in reference order pick the
//                  best node among the clean columns (from K3 / the stored row) and the re-evaluated dirty columns, then
//                  apply NodeInfo.AddTask accounting (api/node_info.go:161-212).
//   K2+K4 k_finalize   gang ready count by wavefront ballot (api/job_info.go:383-394, gang.go:122-125), gang-gated bind
//                  set (framework/session.go:277-285), drf / proportion share reduction (drf.go:157-171,
//                  proportion.go:241-253, api/helpers/helpers.go:47-60).
//
// Exactness: float64 compares with the reference's epsilons, IEEE double division (no fast-math, -ffp-contract=off),
// int64 truncating division reproduced exactly through a reciprocal estimate + integer remainder fix-up.
// The path is elementwise compare + integer scoring, HBM/latency bound: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

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
};
struct NodeVals {
  double idle0, idle1, rel0, rel1;
  long long ac, am, nzc, nzm;
  double inv_ac, inv_am;
  uint32_t cls;
  int slots;   // Allocatable.MaxTaskNum > len(pods)  (predicates.go:127 fails on <=)
  int valid;   // node index < N
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
    ok = ok && n.slots;
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
template <int NPT, int TR>
__global__ void __launch_bounds__(256) k_matrix(KbDev d, KbRound r) {
  __shared__ TaskVals srow[TR];
  __shared__ uint8_t ssame[TR];
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
__global__ void __launch_bounds__(256) k_argmax(KbDev d, KbRound r) {
  extern __shared__ __align__(16) unsigned char k3_smem[];
  uint16_t *ls = reinterpret_cast<uint16_t *>(k3_smem);                        // [NP] scores
  uint8_t *lm = reinterpret_cast<uint8_t *>(k3_smem) + (size_t)d.NP * 2;      // [NP/8] mask bytes
  __shared__ int s_wmax[4];
  __shared__ uint32_t s_wcnt[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t row = blockIdx.x;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(r.score + (size_t)row * d.NP);
    uint4 *dst = reinterpret_cast<uint4 *>(ls);
    for (uint32_t c = tid; c < d.NP / 8; c += 256) dst[c] = src[c];
    const uint32_t *msrc = r.maskw + (size_t)row * (d.NP / 32);
    uint32_t *mdst = reinterpret_cast<uint32_t *>(lm);
    for (uint32_t c = tid; c < d.NP / 32; c += 256) mdst[c] = msrc[c];
  }
  __syncthreads();
  const uint32_t K = r.L;
  unsigned long long *out = r.keys + (size_t)row * K;
  const uint32_t per8 = d.NP / (256 * 8);    // 8-node chunks per thread (NP is a multiple of 2048? no: of 1024 -> see below)
  const uint32_t cbase = tid * per8;         // first chunk of this thread; chunks beyond NP/8 do not exist when NP % 2048 == 0
  const uint4 *ls4 = reinterpret_cast<const uint4 *>(ls);
  uint32_t found = 0;
  int cur = 0x10000;
  while (found < K) {
    int m = -1;
    for (uint32_t c = 0; c < per8; c++) {
      uint32_t mb = lm[cbase + c];
      uint4 sv = ls4[cbase + c];
      uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
      for (int e = 0; e < 8; e++) {
        int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
        if (((mb >> e) & 1u) && sc < cur && sc > m) m = sc;
      }
    }
    m = wave_max_i32(m);
    if (lane == 0) s_wmax[wave] = m;
    __syncthreads();
    m = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    if (m < 0) break;                          // uniform
    uint32_t cnt = 0;
    for (uint32_t c = 0; c < per8; c++) {
      uint32_t mb = lm[cbase + c];
      uint4 sv = ls4[cbase + c];
      uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
      for (int e = 0; e < 8; e++) {
        int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
        cnt += (((mb >> e) & 1u) && sc == m) ? 1u : 0u;
      }
    }
    // inclusive scan inside the wave: DPP row shifts, then row broadcasts (the sequence LLVM's buildScan emits)
    uint32_t pre = cnt;
    pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x111, 0xf, 0xf, false);   // row_shr:1
    pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x112, 0xf, 0xf, false);   // row_shr:2
    pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x114, 0xf, 0xf, false);   // row_shr:4
    pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x118, 0xf, 0xf, false);   // row_shr:8
    pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x142, 0xa, 0xf, false);   // row_bcast:15
    pre += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pre, 0x143, 0xc, 0xf, false);   // row_bcast:31
    if (lane == 63) s_wcnt[wave] = pre;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; w2++) {
      if (w2 < (int)wave) woff += s_wcnt[w2];
      total += s_wcnt[w2];
    }
    uint32_t pos = found + woff + pre - cnt;
    if (cnt && pos < K) {
      for (uint32_t c = 0; c < per8 && pos < K; c++) {
        uint32_t mb = lm[cbase + c];
        uint4 sv = ls4[cbase + c];
        uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
          int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
          if (((mb >> e) & 1u) && sc == m && pos < K) out[pos++] = KB_KEY(m, (cbase + c) * 8 + e);
        }
      }
    }
    found += total;
    cur = m;
    __syncthreads();                           // s_wmax / s_wcnt are reused by the next level
  }
  if (found > K) found = K;
  for (uint32_t i = found + tid; i < K; i += 256) out[i] = 0ull;
}

__device__ __forceinline__ bool bit_test(const uint32_t *bm, uint32_t n) { return (bm[n >> 5] >> (n & 31)) & 1u; }


// window rows -> contiguous descriptors
__global__ void __launch_bounds__(256) k_gather(KbDev d, KbRound r) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= r.n_rows) return;
  uint32_t t = r.rows[i];
  KbRowDesc k;
  k.init0 = d.t_init[t]; k.init1 = d.t_init[(size_t)d.T + t];
  k.nzc = d.t_nzc[t]; k.nzm = d.t_nzm[t];
  k.task = t; k.active = d.t_active[t]; k.resmask = d.t_resmask[t]; k.cls = d.t_cls[t];
  k.slot = (uint16_t)r.shape_slot[i];
  k.flags = (d.t_res[t] == k.init0 && d.t_res[(size_t)d.T + t] == k.init1) ? 1 : 0;
  k.crow = d.crows ? d.crows[(size_t)k.cls * 8] : 0xFFFFFFFFu;
  r.desc[i] = k;
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
#define K5_EVAL (KB_K5_THREADS - 128)                                  // threads that own dirty slots (all but the loader and candidate waves)
#define K5_SPT ((KB_K5_MAX_WINDOW + K5_EVAL - 1) / K5_EVAL)

struct K5Hdr {
  unsigned long long best[2];       // cross-wave max of the row's keys (ds_max_u64), double-buffered by row parity
  unsigned long long cand;          // \  one 16-byte read after barrier 1
  uint32_t exhausted, pad0;         // /
  uint32_t last_slot, stop;         //    one 8-byte read after barrier 2
  uint32_t refills, rescans;
  uint32_t win_base[2];             // candidate window of the next row (written by the loader wave)
  uint32_t cs_tag[2];               // staged clean-node states: node ids (0xFFFFFFFF = empty)
  unsigned long long win[2][64];
  unsigned long long cs8[2][16];    // staged 8-byte fields (index = field id), written by the loader wave one row ahead
  uint32_t cs4[2][4];               // cls, maxpods, podcnt
  unsigned long long red[K5_WAVES]; // live-rescan path only
};

__host__ __device__ inline size_t k5_smem_bytes(uint32_t cap, uint32_t NP) {
  return (size_t)cap * (K5_NF8 * 8 + 4 * 4 + sizeof(KbRowDesc)) + (size_t)(NP / 32) * 4 + sizeof(K5Hdr);
}

__device__ __forceinline__ NodeVals k5_slot_vals(const unsigned long long *tab, const uint32_t *t_cls, const int *t_left, uint32_t cap, uint32_t slot) {
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
  nv.cls = t_cls[slot];
  nv.slots = t_left[slot] > 0;
  nv.valid = 1;
  return nv;
}

// scalar dimensions stay in global memory (rare path): NodeInfo.AddTask's Sub on dims >= 2, task bookkeeping, and the
// multi-GPU per-node deltas
__device__ __forceinline__ void k5_commit_globals(const KbCommitArgs &a, const KbRowDesc &k, double res0, double res1,
                                                  uint32_t i, uint32_t n, uint32_t kind) {
  uint32_t km = k.resmask;
  uint32_t has_map = 0;
  if (km) {
    const KbDev &d = *a.dev;
    has_map = kind ? 1u : d.nmask[n];   // Sub returns early when the receiver's scalar map is nil (resource_info.go:148-153)
    if (has_map) {
      double *vec = kind ? d.rel : d.idle;
      uint32_t dd = 2, m2 = km;
      while (m2) {
        if (m2 & 1u) vec[(size_t)dd * d.NP + n] -= d.t_res[(size_t)dd * d.T + k.task];
        m2 >>= 1; dd++;
      }
      __threadfence_block();
    }
  }
  // one 8-byte decision record; the task table (status, node, counted) is updated from the records by k_apply
  *reinterpret_cast<uint2 *>(&a.dec[i]) = make_uint2(n, kind);
  if (a.has_delta) {
    const KbDev &d = *a.dev;
    const KbRound &r = *a.round;
    if (i >= r.own_row0 && i < r.own_row1) {
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

// ------------------------------------------------------------------------------------------------------------
// K5: sequential commit.  One workgroup walks the window in the reference's task order (allocate.go:129-193 /
// backfill.go:44-67).  Node state changes one node at a time, so for every task
//     best node = max( best CLEAN node of the task's shape , best DIRTY node re-evaluated against live state )
//   * clean side: the shape's candidate list from K3, sorted best-first, with a monotone cursor in LDS that skips
//     entries whose node has become dirty (the dirty set only grows inside a round and the list is longer than the
//     window, so a clean entry always survives unless the list ran out of feasible nodes);
//   * dirty side: the live state of every node touched in this round sits in LDS tables (one slot per node); each
//     thread owns slots {tid, tid+THREADS} and caches their keys, so a task with the same shape as its predecessor
//     re-evaluates only the slot that just changed (gang members are consecutive and identical).
// The loop is latency-bound, so it is written for few dependent instructions: row descriptors are staged in LDS once,
// reductions are DPP + v_max_f64 on biased keys, the candidate's node state is fetched by 13 lanes with ONE load
// instruction (lane k reads field k from its own array) and a clean winner's slot is initialised by the same lanes with
// ONE LDS store; two workgroup barriers per task.
// ------------------------------------------------------------------------------------------------------------
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
    ok = ok && n.slots;
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

// Shared view of the commit kernel's LDS (see k5_smem_bytes)
struct K5Mem {
  unsigned long long *tab;
  uint32_t *t_cls, *t_node, *cursor, *bitmap;
  int *t_left;
  KbRowDesc *desc;
  K5Hdr *H;
  uint32_t cap;
};

// What every role reads after barrier 1 (identical values in every wave): the row's best key, the clean candidate, and
// whether the candidate list ran dry while still full (live-rescan protocol, two extra barriers for everybody).
struct K5Pick {
  unsigned long long best, cand;
  uint32_t exhausted;
};
__device__ __forceinline__ K5Pick k5_pick(const K5Mem M, uint32_t par) {
  const uint4 h = *reinterpret_cast<const uint4 *>(&M.H->cand);
  K5Pick p;
  p.best = M.H->best[par];
  p.cand = ((unsigned long long)h.y << 32) | h.x;
  p.exhausted = h.z;
  return p;
}

// Live rescan of every CLEAN node (the candidate list ran out while still full; cannot happen with L > window).  All
// waves take part: two barriers.  Returns the best clean key; H.cand is replaced by it.
__device__ __forceinline__ unsigned long long k5_rescan(const KbCommitArgs &a, const K5Mem M, const KbRowDesc &cur) {
  const KbDev &d = *a.dev;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TaskVals tv;
  tv.init0 = cur.init0; tv.init1 = cur.init1; tv.nzc = cur.nzc; tv.nzm = cur.nzm;
  tv.cls = cur.cls; tv.active = cur.active; tv.task = cur.task; tv.pad = 0;
  unsigned long long k2 = 0ull;
  for (uint32_t n = tid; n < d.N; n += KB_K5_THREADS) {
    if (bit_test(M.bitmap, n)) continue;
    NodeVals nv = load_node(d, n);
    uint32_t res = eval_pair(d, tv, nv, n, a.fit_mode);
    if (res) { unsigned long long k3 = KB_KEY(res & 0xFFFFu, n); k2 = k3 > k2 ? k3 : k2; }
  }
  k2 = wave_max_key(k2);
  if (lane == 0) M.H->red[wave] = k2;
  __syncthreads();
  unsigned long long bclean = oct_max_key(M.H->red[lane & (K5_WAVES - 1)]);
  if (tid == 0) { M.H->rescans++; M.H->cand = bclean; }
  __syncthreads();
  return bclean;
}

#ifdef KB_K5_TRACE
#define K5R_DECL(role) const bool trc = a.trace != nullptr && (threadIdx.x & 63) == 0 && ((role) != 0 || threadIdx.x == 0); unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0;
#define K5R_T0() do { if (trc) tr0 = __builtin_readcyclecounter(); } while (0)
#define K5R_T1() do { if (trc) tr1 = __builtin_readcyclecounter(); } while (0)
#define K5R_T2() do { if (trc) tr2 = __builtin_readcyclecounter(); } while (0)
#define K5R_T3() do { if (trc) tr3 = __builtin_readcyclecounter(); } while (0)
#define K5R_END(role, i, cw, same) do { if (trc && (i) < 512) { unsigned long long *dst = a.trace + ((size_t)(role) * 512 + (i)) * 8; dst[0] = tr0; dst[1] = tr1; dst[2] = tr2; dst[3] = tr3; dst[4] = __builtin_readcyclecounter(); dst[5] = ((cw) ? 1ull : 0ull) | ((same) ? 2ull : 0ull); } } while (0)
#else
#define K5R_DECL(role)
#define K5R_T0() do { } while (0)
#define K5R_T1() do { } while (0)
#define K5R_T2() do { } while (0)
#define K5R_T3() do { } while (0)
#define K5R_END(role, i, cw, same) do { } while (0)
#endif

// ---- role: evaluation waves (own the dirty slots) --------------------------------------------------------------
__device__ __forceinline__ void k5_eval_role(const KbCommitArgs a, const K5Mem M, uint32_t &nd_out, uint32_t &n_done, uint32_t &reason) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, cap = M.cap;
  K5Hdr &H = *M.H;
  unsigned long long ck[K5_SPT];     // cached keys of the dirty slots this thread owns, valid for shape `prev_shape`
#pragma unroll
  for (int j = 0; j < K5_SPT; j++) ck[j] = 0ull;
  unsigned long long wmax = 0ull;    // this wave's max over ck[], valid while no lane of the wave re-evaluates
  uint32_t prev_shape = 0xFFFFFFFFu, nd = 0, last_slot = 0xFFFFFFFFu;
  K5R_DECL(0)
  const uint32_t tid_id = tid, lane_id = lane;
  for (uint32_t i = 0; i < a.n_rows; i++) {
    K5R_T0();
    uint32_t tid = tid_id, lane = lane_id;   // opaque per row: keeps thread/lane masks out of (spilled) SGPR pairs
    asm volatile("" : "+v"(tid), "+v"(lane));
    const uint32_t par = i & 1;
    const KbRowDesc &cur = M.desc[i];
    const uint32_t shape = cur.slot;
    const bool same = shape == prev_shape;
    prev_shape = shape;
    // ---- phase 1: keys of my dirty slots (re-evaluated only when the shape changed or the slot was just committed)
    bool mine = false;
#pragma unroll
    for (int j = 0; j < K5_SPT; j++) {
      uint32_t slot = tid + j * K5_EVAL;
      mine = mine || (slot < nd && (!same || slot == last_slot));
    }
    if (__ballot(mine)) {
      if (mine) {
        TaskVals tv;
        tv.init0 = cur.init0; tv.init1 = cur.init1; tv.nzc = cur.nzc; tv.nzm = cur.nzm;
        tv.cls = cur.cls; tv.active = cur.active; tv.task = cur.task; tv.pad = 0;
        const uint32_t *crow = a.use_crow ? &cur.crow : nullptr;
#pragma unroll
        for (int j = 0; j < K5_SPT; j++) {
          uint32_t slot = tid + j * K5_EVAL;
          if (slot < nd && (!same || slot == last_slot)) {
            NodeVals nv = k5_slot_vals(M.tab, M.t_cls, M.t_left, cap, slot);
            uint32_t node = M.t_node[slot];
            uint32_t res = eval_pair_k5(a, tv, nv, node, crow);
            ck[j] = res ? KB_KEY(res & 0xFFFFu, node) : 0ull;
          }
        }
      }
      unsigned long long key = 0ull;
#pragma unroll
      for (int j = 0; j < K5_SPT; j++) key = ck[j] > key ? ck[j] : key;
      wmax = wave_max_key(key);
    }
    if (lane == 0 && wmax) atomicMax(&H.best[par], wmax);
    K5R_T1();
    __syncthreads();
    K5R_T2();
    // ---- phase 2
    K5Pick p = k5_pick(M, par);
    if (tid == 0) H.best[par ^ 1] = 0ull;
    if (p.exhausted) { unsigned long long bc = k5_rescan(a, M, cur); p.cand = bc; p.best = bc > p.best ? bc : p.best; }
    if (p.best == 0ull) {
      if (a.backfill) {   // backfill.go:50-66: no node passes the predicates -> the task simply stays Pending
        if (tid == 0) *reinterpret_cast<uint2 *>(&a.dec[i]) = make_uint2(KB_NONE_U32, 0u);
        __syncthreads();
        n_done = i + 1;
        continue;
      }
      n_done = i; reason = KB_REASON_NO_FEASIBLE;   // allocate.go:144-148: the job is abandoned; the host re-plans from here
      break;
    }
    const bool clean_wins = p.best == p.cand;
    if (!clean_wins) {
      // a dirty node wins: its owner applies NodeInfo.AddTask (api/node_info.go:172-212) to the LDS copy
#pragma unroll
      for (int j = 0; j < K5_SPT; j++) {
        uint32_t slot = tid + j * K5_EVAL;
        if (slot < nd && ck[j] == p.best) {
          const uint32_t n = KB_KEY_NODE(p.best);
          // Resreq == InitResreq unless an init container raised it (rare): a scalar branch, NOT a select between an LDS
          // and a global address (the compiler would turn that into flat loads with a full vmcnt+lgkmcnt drain)
          double res0 = cur.init0, res1 = cur.init1;
          asm volatile("" : "+v"(res0), "+v"(res1));   // materialise the LDS values here: no pointer phi -> no flat load
          if (!(__builtin_amdgcn_readfirstlane((int)cur.flags) & 1)) { const KbDev &d = *a.dev; res0 = d.t_res[cur.task]; res1 = d.t_res[(size_t)d.T + cur.task]; }
          uint32_t kind = 0;
          if (!a.backfill) {   // allocate.go:160: InitResreq.LessEqual(node.Idle) ? Allocate : Pipeline
            bool fi = le_eps(cur.init0, __longlong_as_double((long long)M.tab[K5F_IDLE0 * cap + slot]), EPS_CPU) &&
                      le_eps(cur.init1, __longlong_as_double((long long)M.tab[K5F_IDLE1 * cap + slot]), EPS_MEM);
            uint32_t act = cur.active >> 2, dd = 2;
            if (act) {
              const KbDev &d = *a.dev;
              while (act) {
                if (act & 1u) fi = fi && le_eps(d.t_init[(size_t)dd * d.T + cur.task], d.idle[(size_t)dd * d.NP + n], EPS_SCALAR);
                act >>= 1; dd++;
              }
            }
            kind = fi ? 0u : 1u;
          }
          const uint32_t f0 = kind ? K5F_REL0 : K5F_IDLE0;
          M.tab[(size_t)f0 * cap + slot] = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)M.tab[(size_t)f0 * cap + slot]) - res0);
          M.tab[(size_t)(f0 + 1) * cap + slot] = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)M.tab[(size_t)(f0 + 1) * cap + slot]) - res1);
          M.tab[(size_t)K5F_NZC * cap + slot] += (unsigned long long)cur.nzc;
          M.tab[(size_t)K5F_NZM * cap + slot] += (unsigned long long)cur.nzm;
          M.t_left[slot] -= 1;
          *reinterpret_cast<uint2 *>(&H.last_slot) = make_uint2(slot, kind);
          k5_commit_globals(a, cur, res0, res1, i, n, kind);
        }
      }
    }
    K5R_T3();
    __syncthreads();
    K5R_END(0, i, clean_wins, same);
    if (clean_wins) nd++;
    n_done = i + 1;
    const uint2 h2 = *reinterpret_cast<const uint2 *>(&H.last_slot);
    last_slot = h2.x;
    if (h2.y) { reason = KB_REASON_PIPELINED; break; }
  }
  nd_out = nd;
}

// ---- role: loader wave (every global load of the loop, a full row ahead, landed in LDS staging) -----------------
__device__ __forceinline__ void k5_load_role(const KbCommitArgs a, const K5Mem M, uint32_t &nd_out, uint32_t &n_done, uint32_t &reason) {
  const uint32_t lane = threadIdx.x & 63;
  K5Hdr &H = *M.H;
  // lane (g*16 + k) reads field k of one node with ONE load instruction per width: k < 10: 8-byte field k from its own
  // array; k = 10..12: cls / maxpods / podcnt.  Two lane groups -> two nodes per instruction.
  const uint32_t fld = lane & 15, grp = lane >> 4;
  const unsigned long long *g8 = nullptr;
  const uint32_t *g4 = nullptr;
  if (grp < 2) k5_field_ptrs(*a.dev, fld, g8, g4);
  unsigned long long pw_key = 0ull;   // candidate window being fetched, one entry per lane
  uint32_t pw_base = 0;
  unsigned long long ps8 = 0ull;      // node states being fetched: lane groups 0/1 -> staging slots 0/1
  uint32_t ps4 = 0;
  uint32_t ps_tag0 = 0xFFFFFFFFu, ps_tag1 = 0xFFFFFFFFu;   // nodes in flight for staging slot 0 / 1
  uint32_t ls_tag0 = 0xFFFFFFFFu, ls_tag1 = 0xFFFFFFFFu;   // what the staging slots hold / will hold
  uint32_t inv_node = 0xFFFFFFFFu, nd = 0;
  if (a.n_rows > 1) pw_key = (lane < a.L) ? a.keys[(size_t)M.desc[1].slot * a.L + lane] : 0ull;
  K5R_DECL(1)
  const uint32_t lane_id = lane, fld_id = fld, grp_id = grp;
  for (uint32_t i = 0; i < a.n_rows; i++) {
    K5R_T0();
    uint32_t lane = lane_id, fld = fld_id, grp = grp_id;   // opaque per row (see the candidate role)
    asm volatile("" : "+v"(lane), "+v"(fld), "+v"(grp));
    const uint32_t par = i & 1;
    // ---- phase 1: drop the copy of the node the previous row committed (it is dirty now), land the states requested in
    //      the previous row's phase 2 (consumed by this row's commit)
    if (inv_node != 0xFFFFFFFFu) {
      if (ls_tag0 == inv_node) { ls_tag0 = 0xFFFFFFFFu; if (ps_tag0 == inv_node) ps_tag0 = 0xFFFFFFFFu; if (lane == 0) H.cs_tag[0] = 0xFFFFFFFFu; }
      if (ls_tag1 == inv_node) { ls_tag1 = 0xFFFFFFFFu; if (ps_tag1 == inv_node) ps_tag1 = 0xFFFFFFFFu; if (lane == 0) H.cs_tag[1] = 0xFFFFFFFFu; }
    }
    if (ps_tag0 != 0xFFFFFFFFu || ps_tag1 != 0xFFFFFFFFu) {
      const bool on = (grp == 0 && ps_tag0 != 0xFFFFFFFFu) || (grp == 1 && ps_tag1 != 0xFFFFFFFFu);
      if (on && fld < K5_NF8) H.cs8[grp][fld] = ps8;
      if (on && fld >= 10 && fld < 13) H.cs4[grp][fld - 10] = ps4;
      if (lane == 0 && ps_tag0 != 0xFFFFFFFFu) H.cs_tag[0] = ps_tag0;
      if (lane == 16 && ps_tag1 != 0xFFFFFFFFu) H.cs_tag[1] = ps_tag1;
      ps_tag0 = 0xFFFFFFFFu; ps_tag1 = 0xFFFFFFFFu;
    }
    K5R_T1();
    __syncthreads();
    K5R_T2();
    // ---- phase 2
    K5Pick p = k5_pick(M, par);
    // (a) hand the next row's candidate window (requested a row ago) to the candidate wave and stage the states of its
    //     first two clean nodes: whichever of them survives this row's commit is the next row's clean candidate
    if (i + 1 < a.n_rows) {
      const uint32_t nshape = M.desc[i + 1].slot;
      H.win[par ^ 1][lane] = pw_key;
      if (lane == 0) H.win_base[par ^ 1] = pw_base;
      const uint32_t c0 = M.cursor[nshape];
      const bool nz = pw_key != 0ull;
      unsigned long long cb = __ballot(nz && (pw_base + lane) >= c0 && !bit_test(M.bitmap, KB_KEY_NODE(pw_key)));
      uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
      if (cb) {
        int f1 = __ffsll((unsigned long long)cb) - 1;
        m1 = 0xFFFFFFFFu - (uint32_t)__builtin_amdgcn_readlane((int)(pw_key & 0xFFFFFFFFull), f1);
        unsigned long long cb2 = cb & (cb - 1);
        if (cb2) {
          int f2 = __ffsll((unsigned long long)cb2) - 1;
          m2 = 0xFFFFFFFFu - (uint32_t)__builtin_amdgcn_readlane((int)(pw_key & 0xFFFFFFFFull), f2);
        }
      }
      const bool have1 = m1 == 0xFFFFFFFFu || ls_tag0 == m1 || ls_tag1 == m1;
      const bool have2 = m2 == 0xFFFFFFFFu || ls_tag0 == m2 || ls_tag1 == m2;
      if (!have1) {
        if (ls_tag0 != m2) { ls_tag0 = m1; ps_tag0 = m1; } else { ls_tag1 = m1; ps_tag1 = m1; }
      }
      if (!have2) {
        if (ls_tag0 != m1) { ls_tag0 = m2; ps_tag0 = m2; } else { ls_tag1 = m2; ps_tag1 = m2; }
      }
      const uint32_t want = grp == 0 ? ps_tag0 : (grp == 1 ? ps_tag1 : 0xFFFFFFFFu);
      if (want != 0xFFFFFFFFu) {
        if (g8) ps8 = g8[want];
        if (g4) ps4 = g4[want];
      }
    }
    // (b) request the window of the row after next
    if (i + 2 < a.n_rows) {
      const uint32_t n2shape = M.desc[i + 2].slot;
      pw_base = M.cursor[n2shape];   // may lag: stale entries are filtered by the dirty bitmap when the window is used
      uint32_t e = pw_base + lane;
      pw_key = (e < a.L) ? a.keys[(size_t)n2shape * a.L + e] : 0ull;
    }
    if (p.exhausted) { unsigned long long bc = k5_rescan(a, M, M.desc[i]); p.cand = bc; p.best = bc > p.best ? bc : p.best; }
    if (p.best == 0ull) {
      if (a.backfill) { __syncthreads(); n_done = i + 1; inv_node = 0xFFFFFFFFu; continue; }
      n_done = i; reason = KB_REASON_NO_FEASIBLE;
      break;
    }
    const bool clean_wins = p.best == p.cand;
    inv_node = clean_wins ? KB_KEY_NODE(p.best) : 0xFFFFFFFFu;
    K5R_T3();
    __syncthreads();
    K5R_END(1, i, clean_wins, false);
    if (clean_wins) nd++;
    n_done = i + 1;
    const uint2 h2 = *reinterpret_cast<const uint2 *>(&H.last_slot);
    if (h2.y) { reason = KB_REASON_PIPELINED; break; }
  }
  nd_out = nd;
}

// ---- role: candidate wave (walks the candidate lists, commits clean winners) -------------------------------------
__device__ __forceinline__ void k5_cand_role(const KbCommitArgs a, const K5Mem M, uint32_t &nd_out, uint32_t &n_done, uint32_t &reason) {
  const uint32_t lane0 = threadIdx.x & 63, cap = M.cap;
  K5Hdr &H = *M.H;
  const unsigned long long *g8 = nullptr;   // synchronous miss path: lanes 0..12 fetch the node's fields
  const uint32_t *g4 = nullptr;
  if (lane0 < 16) k5_field_ptrs(*a.dev, lane0, g8, g4);
  // the current shape's window (keys per lane), mask of its clean lanes, mask of its empty lanes
  unsigned long long wkey = 0ull, wb = 0ull, wzero = 0ull;
  uint32_t wbase = 0, prev_shape = 0xFFFFFFFFu, nd = 0;
  bool wvalid = false, prev_clean_win = false;
  K5R_DECL(2)
  const uint32_t lane_id = lane0;
  for (uint32_t i = 0; i < a.n_rows; i++) {
    K5R_T0();
    // re-materialise the lane id every row: otherwise every `lane == k` mask is hoisted out of the loop as an SGPR pair, the
    // kernel runs out of SGPRs and each mask is spilled to / restored from VGPR lanes (v_readlane pairs) at every use
    uint32_t lane = lane_id;
    asm volatile("" : "+v"(lane));
    const uint32_t par = i & 1;
    const KbRowDesc &cur = M.desc[i];
    const uint32_t shape = cur.slot;
    const bool same_tr = shape == prev_shape;
    (void)same_tr;
    // ---- phase 1: first list entry whose node is not dirty.  Inside a run of rows with the same shape the only node
    // that can have turned dirty since the previous row is the previous row's own clean winner, i.e. the lowest set bit
    // of the clean mask: clear it instead of re-reading the window and re-testing the bitmap.
    if (shape == prev_shape && wvalid) {
      if (prev_clean_win) wb &= wb - 1;
    } else {
      wkey = H.win[par][lane];
      wbase = H.win_base[par];
      const uint32_t c0 = M.cursor[shape];
      const bool nz = wkey != 0ull;
      wb = __ballot(nz && (wbase + lane) >= c0 && !bit_test(M.bitmap, KB_KEY_NODE(wkey)));
      wzero = __ballot(!nz);
      wvalid = true;
    }
    prev_shape = shape;
    unsigned long long cand = 0ull;
    bool list_end = false;
    uint32_t curs;
    for (;;) {
      if (wb) {
        int first = __ffsll((unsigned long long)wb) - 1;
        // lists are 0-terminated and sorted best-first, so the first clean entry is the best clean node
        cand = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(wkey >> 32), first) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)(wkey & 0xFFFFFFFFull), first);
        curs = wbase + first;
        break;
      }
      curs = wbase;
      if (wzero) { list_end = true; break; }            // ran past the last feasible node: no clean candidate exists
      wbase += 64;
      curs = wbase;
      if (wbase >= a.L) break;                          // list exhausted while still full: live rescan
      uint32_t e = wbase + lane;
      wkey = (e < a.L) ? a.keys[(size_t)shape * a.L + e] : 0ull;
      const bool nz = wkey != 0ull;
      wb = __ballot(nz && !bit_test(M.bitmap, KB_KEY_NODE(wkey)));
      wzero = __ballot(!nz);
      if (lane == 0) H.refills++;
    }
    if (lane == 0) {
      M.cursor[shape] = curs;
      *reinterpret_cast<uint4 *>(&H.cand) = make_uint4((uint32_t)(cand & 0xFFFFFFFFull), (uint32_t)(cand >> 32), (!cand && !list_end) ? 1u : 0u, 0u);
      if (cand) atomicMax(&H.best[par], cand);
    }
    K5R_T1();
    __syncthreads();
    K5R_T2();
    // ---- phase 2
    K5Pick p = k5_pick(M, par);
    if (p.exhausted) { unsigned long long bc = k5_rescan(a, M, cur); p.cand = bc; p.best = bc > p.best ? bc : p.best; wvalid = false; }
    if (p.best == 0ull) {
      if (a.backfill) { __syncthreads(); n_done = i + 1; prev_clean_win = false; continue; }
      n_done = i; reason = KB_REASON_NO_FEASIBLE;
      break;
    }
    const bool clean_wins = p.best == p.cand;
    if (clean_wins) {
      // commit: NodeInfo.AddTask (api/node_info.go:172-212) turns the clean node into dirty slot `nd`.  Its pristine state
      // was staged by the loader a row ahead (normal case) or is fetched synchronously.  Lanes 0..9 get the 8-byte
      // fields, lanes 10..12 cls / maxpods / podcnt.
      const uint32_t n = KB_KEY_NODE(p.best);
      double res0 = cur.init0, res1 = cur.init1;   // scalar branch, not an LDS/global address select (see the eval role)
      asm volatile("" : "+v"(res0), "+v"(res1));
      if (!(__builtin_amdgcn_readfirstlane((int)cur.flags) & 1)) { const KbDev &d = *a.dev; res0 = d.t_res[cur.task]; res1 = d.t_res[(size_t)d.T + cur.task]; }
      unsigned long long st8 = 0ull;
      uint32_t st4 = 0;
      const uint2 tags = *reinterpret_cast<const uint2 *>(&H.cs_tag[0]);
      const int src = (tags.x == n) ? 0 : ((tags.y == n) ? 1 : 2);
      if (lane == 0) { if (src < 2) H.refills += 1u << 16; else H.rescans += 1u << 16; }   // staging hits / misses (diagnostic)
      if (src < 2) {
        st8 = H.cs8[src][lane & 15];
        st4 = H.cs4[src][(lane - 10) & 3];
      } else {
        if (g8) st8 = g8[n];
        if (g4) st4 = g4[n];
      }
      double v = __longlong_as_double((long long)st8);
      uint32_t kind = 0;
      if (!a.backfill) {   // allocate.go:160: InitResreq.LessEqual(node.Idle) ? Allocate : Pipeline
        // lanes 0 / 1 hold Idle cpu / memory: straight-line compare with per-lane operands, other lanes pass
        const double ini = (lane == K5F_IDLE0) ? cur.init0 : cur.init1;
        const double eps = (lane == K5F_IDLE0) ? EPS_CPU : EPS_MEM;
        bool ok = (lane > K5F_IDLE1) || le_eps(ini, v, eps);
        if (lane == 63 && (cur.active >> 2)) {   // scalar dimensions (rare): compared against live global state
          const KbDev &d = *a.dev;
          uint32_t act = cur.active >> 2, dd = 2;
          while (act) {
            if (act & 1u) ok = ok && le_eps(d.t_init[(size_t)dd * d.T + cur.task], d.idle[(size_t)dd * d.NP + n], EPS_SCALAR);
            act >>= 1; dd++;
          }
        }
        kind = __ballot(!ok) ? 1u : 0u;
      }
      // apply the task: Allocated -> Idle.Sub(Resreq); Pipelined -> Releasing.Sub(Resreq); pod joins ni.Tasks
      const uint32_t f0 = kind ? K5F_REL0 : K5F_IDLE0;
      if (lane == f0) v -= res0;
      if (lane == f0 + 1) v -= res1;
      unsigned long long out = (unsigned long long)__double_as_longlong(v);
      if (lane == K5F_NZC) out = st8 + (unsigned long long)cur.nzc;
      if (lane == K5F_NZM) out = st8 + (unsigned long long)cur.nzm;
      if (lane >= K5F_INVAC && lane <= K5F_AM) out = st8;
      if (lane < K5_NF8) M.tab[(size_t)lane * cap + nd] = out;
      const int maxp = __builtin_amdgcn_readlane((int)st4, 11), pods = __builtin_amdgcn_readlane((int)st4, 12);
      // t_cls, t_node, t_left are consecutive [cap] arrays: one store with a per-lane offset (no pointer table)
      if (lane >= 10 && lane < 13) {
        const uint32_t which = (lane == 10) ? 0u : ((lane == 12) ? 1u : 2u);
        const uint32_t val = (lane == 10) ? st4 : ((lane == 12) ? n : (uint32_t)(maxp - pods - 1));
        M.t_cls[(size_t)which * cap + nd] = val;
      }
      if (lane == 0) {
        atomicOr(&M.bitmap[n >> 5], 1u << (n & 31));
        *reinterpret_cast<uint2 *>(&H.last_slot) = make_uint2(nd, kind);
        k5_commit_globals(a, cur, res0, res1, i, n, kind);
      }
    }
    K5R_T3();
    __syncthreads();
    K5R_END(2, i, clean_wins, same_tr);
    if (clean_wins) nd++;
    prev_clean_win = clean_wins;
    n_done = i + 1;
    const uint2 h2 = *reinterpret_cast<const uint2 *>(&H.last_slot);
    if (h2.y) { reason = KB_REASON_PIPELINED; break; }
  }
  nd_out = nd;
}

__global__ void __launch_bounds__(KB_K5_THREADS) k_commit(const KbCommitArgs a) {
  const KbDev &d = *a.dev;
  extern __shared__ __align__(16) unsigned char k5_smem[];
  K5Mem M;
  M.cap = a.cap;
  M.tab = reinterpret_cast<unsigned long long *>(k5_smem);
  M.t_cls = reinterpret_cast<uint32_t *>(M.tab + (size_t)K5_NF8 * M.cap);
  M.t_node = M.t_cls + M.cap;
  M.t_left = reinterpret_cast<int *>(M.t_node + M.cap);
  M.cursor = reinterpret_cast<uint32_t *>(M.t_left + M.cap);
  M.desc = reinterpret_cast<KbRowDesc *>(M.cursor + M.cap);
  M.bitmap = reinterpret_cast<uint32_t *>(M.desc + M.cap);
  M.H = reinterpret_cast<K5Hdr *>(M.bitmap + a.NP / 32);
  K5Hdr &H = *M.H;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cap = M.cap;

  for (uint32_t w = tid; w < d.NP / 32; w += KB_K5_THREADS) M.bitmap[w] = 0;
  for (uint32_t w = tid; w < cap; w += KB_K5_THREADS) M.cursor[w] = 0;
  {   // stage the window's row descriptors (coalesced 8-byte copies)
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.desc);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(M.desc);
    const uint32_t nq = a.n_rows * (uint32_t)(sizeof(KbRowDesc) / 8);
    for (uint32_t w = tid; w < nq; w += KB_K5_THREADS) dst[w] = src[w];
  }
  if (tid == 0) {
    H.best[0] = 0; H.best[1] = 0; H.cand = 0; H.exhausted = 0; H.pad0 = 0; H.stop = 0; H.last_slot = 0xFFFFFFFFu;
    H.refills = 0; H.rescans = 0; H.win_base[0] = 0; H.win_base[1] = 0; H.cs_tag[0] = 0xFFFFFFFFu; H.cs_tag[1] = 0xFFFFFFFFu;
  }
  {
    // The loop is latency-bound and this workgroup starts on a cold L2 (kernel boundary).  Touch every 128-byte line it
    // can need later — node state arrays and the candidate lists — once, with all threads.
    unsigned long long acc = 0;
    const uint32_t lines = d.NP / 16;   // 16 x 8 bytes per line
    const unsigned long long *arrs[10] = {
        reinterpret_cast<const unsigned long long *>(d.idle), reinterpret_cast<const unsigned long long *>(d.idle + d.NP),
        reinterpret_cast<const unsigned long long *>(d.rel), reinterpret_cast<const unsigned long long *>(d.rel + d.NP),
        reinterpret_cast<const unsigned long long *>(d.inv_acpu), reinterpret_cast<const unsigned long long *>(d.inv_amem),
        reinterpret_cast<const unsigned long long *>(d.acpu), reinterpret_cast<const unsigned long long *>(d.amem),
        reinterpret_cast<const unsigned long long *>(d.nzc), reinterpret_cast<const unsigned long long *>(d.nzm)};
#pragma unroll
    for (int a = 0; a < 10; a++)
      for (uint32_t l = tid; l < lines; l += KB_K5_THREADS) acc += arrs[a][(size_t)l * 16];
    const uint32_t *arr4[3] = {d.ncls, reinterpret_cast<const uint32_t *>(d.maxpods), reinterpret_cast<const uint32_t *>(d.podcnt)};
#pragma unroll
    for (int a = 0; a < 3; a++)
      for (uint32_t l = tid; l < d.NP / 32; l += KB_K5_THREADS) acc += arr4[a][(size_t)l * 32];
    const size_t klines = ((size_t)a.n_mrows * a.L + 15) / 16;
    for (size_t l = tid; l < klines; l += KB_K5_THREADS) acc += a.keys[l * 16];
    if (acc == 0x123456789abcdefull) H.pad0 = 1;   // keep the loads alive
  }
  __syncthreads();
  if (wave == K5_WAVES - 2) H.win[0][lane] = (lane < a.L) ? a.keys[(size_t)M.desc[0].slot * a.L + lane] : 0ull;   // row 0's window
  __syncthreads();

  // Three specialised loops that only meet at the two barriers per row: waves 0..W-3 own the dirty slots, wave W-2 is the
  // loader, wave W-1 walks the candidate lists and commits clean winners.  Every role derives the same control decisions
  // (continue / stop / rescan) from the same LDS words, so the barrier counts always match.
  uint32_t nd = 0, n_done = 0, reason = KB_REASON_DONE;
  if (wave == K5_WAVES - 1) k5_cand_role(a, M, nd, n_done, reason);
  else if (wave == K5_WAVES - 2) k5_load_role(a, M, nd, n_done, reason);
  else k5_eval_role(a, M, nd, n_done, reason);

  // ---- write the dirty nodes' live state back to HBM
  __syncthreads();
  for (uint32_t slot = tid; slot < nd; slot += KB_K5_THREADS) {
    uint32_t n = M.t_node[slot];
    d.idle[n] = __longlong_as_double((long long)M.tab[K5F_IDLE0 * cap + slot]);
    d.idle[(size_t)d.NP + n] = __longlong_as_double((long long)M.tab[K5F_IDLE1 * cap + slot]);
    d.rel[n] = __longlong_as_double((long long)M.tab[K5F_REL0 * cap + slot]);
    d.rel[(size_t)d.NP + n] = __longlong_as_double((long long)M.tab[K5F_REL1 * cap + slot]);
    d.nzc[n] = (long long)M.tab[K5F_NZC * cap + slot];
    d.nzm[n] = (long long)M.tab[K5F_NZM * cap + slot];
    d.podcnt[n] = d.maxpods[n] - M.t_left[slot];
  }
  if (tid == 0) { a.result[0] = n_done; a.result[1] = reason; a.result[2] = nd; a.result[3] = H.rescans & 0xFFFFu; a.result[4] = H.refills & 0xFFFFu; a.result[5] = H.refills >> 16; a.result[6] = H.rescans >> 16; }
}

// task-table side of ssn.Allocate / ssn.Pipeline for the rows the commit kernel processed (job.UpdateTaskStatus,
// task.NodeName: framework/session.go:243,205; api/node_info.go:206-209), applied in parallel after the round
__global__ void __launch_bounds__(256) k_apply(KbDev d, KbRound r) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= r.result[0]) return;     // n_done
  uint2 dc = *reinterpret_cast<const uint2 *>(&r.dec[i]);
  if (dc.x == KB_NONE_U32) return;
  uint32_t t = r.rows[i];
  d.t_status[t] = dc.y ? KB_TASK_PIPELINED : KB_TASK_ALLOCATED;
  d.t_node[t] = dc.x;
  d.t_counted[t] = 1;
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
  if (job_ready) {                         // session.go:277-285: every Allocated task of a ready job is dispatched
    for (uint32_t t = t0 + lane; t < t1; t += 64)
      if (d.t_status[t] == KB_TASK_ALLOCATED) { d.t_status[t] = KB_TASK_BINDING; d.t_bind[t] = d.t_node[t]; }
  }
  if (lane == 0) job_ready_cnt[j] = ready;
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
size_t kb_commit_smem_bytes(uint32_t cap, uint32_t NP) { return k5_smem_bytes(cap, NP); }
void kb_launch_gather(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  hipLaunchKernelGGL(k_gather, dim3((r.n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, r);
}
void kb_launch_matrix(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_mrows == 0) return;
  if ((size_t)r.n_mrows * d.NP >= (4u << 20)) {
    dim3 grid(d.NP / (256 * 4), (r.n_mrows + 31) / 32);
    hipLaunchKernelGGL((k_matrix<4, 32>), grid, dim3(256), 0, (hipStream_t)stream, d, r);
  } else {
    dim3 grid(d.NP / 256, (r.n_mrows + 3) / 4);
    hipLaunchKernelGGL((k_matrix<1, 4>), grid, dim3(256), 0, (hipStream_t)stream, d, r);
  }
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_argmax), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_argmax, dim3(r.n_mrows), dim3(256), sh, (hipStream_t)stream, d, r);
}
// the session / round views the commit kernel dereferences on its rare paths, written stream-ordered before the launch
__global__ void k_store_views(KbDev d, KbRound r, KbDev *dd, KbRound *dr) { *dd = d; *dr = r; }

void kb_launch_commit(const KbDev &d, const KbRound &r, KbDev *dev_copy, KbRound *round_copy, void *stream) {
  if (r.n_rows == 0) return;
  size_t sh = k5_smem_bytes(r.cap, d.NP);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_commit), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_store_views, dim3(1), dim3(1), 0, (hipStream_t)stream, d, r, dev_copy, round_copy);
  KbCommitArgs a;
  a.dev = dev_copy; a.round = round_copy;
  a.keys = r.keys; a.dec = r.dec; a.desc = r.desc; a.result = r.result; a.trace = r.trace;
  a.n_rows = r.n_rows; a.n_mrows = r.n_mrows; a.L = r.L; a.cap = r.cap; a.N = d.N; a.NP = d.NP;
  a.fit_mode = r.fit_mode; a.backfill = r.backfill; a.pred_enabled = d.pred_enabled; a.score_enabled = d.score_enabled;
  a.wL = d.wL; a.wM = d.wM; a.wB = d.wB;
  a.use_crow = (d.pred_enabled && d.crows != nullptr && d.n_nc <= 32) ? 1u : 0u;
  a.has_delta = r.delta != nullptr ? 1u : 0u;
  hipLaunchKernelGGL(k_commit, dim3(1), dim3(KB_K5_THREADS), sh, (hipStream_t)stream, a);
  hipLaunchKernelGGL(k_apply, dim3((r.n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, r);
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
