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
  nv.inv_ac = 1.0 / (double)nv.ac;
  nv.inv_am = 1.0 / (double)nv.am;
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
__device__ __forceinline__ uint32_t eval_pair(const KbDev &d, const TaskVals &t, const NodeVals &n, uint32_t node, int fit_mode) {
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
    if (d.compat) {
      uint32_t bit = t.cls * d.n_nc + n.cls;
      ok = ok && ((d.compat[bit >> 3] >> (bit & 7)) & 1);
    }
  }
  if (!ok) return 0;
  uint32_t score = 0;
  if (d.score_enabled) {
    long long rc = n.nzc + t.nzc, rm = n.nzm + t.nzm;   // resource_allocation.go:100-112
    int lc = 0, mc = 0, lm = 0, mm = 0, rem;
    if (!(n.ac == 0 || rc > n.ac)) { mc = div10(rc, n.ac, n.inv_ac, rem); lc = 10 - mc - rem; }   // most/least_requested.go
    if (!(n.am == 0 || rm > n.am)) { mm = div10(rm, n.am, n.inv_am, rem); lm = 10 - mm - rem; }
    int least = (lc + lm) / 2, most = (mc + mm) / 2;
    double cf = (n.ac == 0) ? 1.0 : (double)rc / (double)n.ac;      // balanced_resource_allocation.go:74-79
    double mf = (n.am == 0) ? 1.0 : (double)rm / (double)n.am;
    int bal = 0;
    if (!(cf >= 1.0 || mf >= 1.0)) bal = (int)(long long)((1.0 - fabs(cf - mf)) * 10.0);
    score = (uint32_t)(least * d.wL + most * d.wM + bal * d.wB);
  }
  return 0x10000u | (score & 0xFFFFu);
}

// ------------------------------------------------------------------------------------------------------------
// K1: mask + score matrix.  grid (NP / (256*NPT), ceil(n_rows/TR)); thread <-> NPT consecutive nodes kept in
// registers for all TR rows of the tile; the tile's task vectors are staged once in LDS.
// Stores: one 8-byte score vector per thread per row (512 B contiguous per wave), one 4-byte mask word per 8 lanes.
// ------------------------------------------------------------------------------------------------------------
#define K1_TR 32
#define K1_NPT 4
__global__ void __launch_bounds__(256) k_matrix(KbDev d, KbRound r) {
  __shared__ TaskVals srow[K1_TR];
  __shared__ uint8_t ssame[K1_TR];
  const uint32_t row0 = blockIdx.y * K1_TR;
  const uint32_t nr = min((uint32_t)K1_TR, r.n_rows - row0);
  if (threadIdx.x < nr) {
    uint32_t i = row0 + threadIdx.x;
    uint32_t t = r.rows ? r.rows[i] : r.row_task0 + i;
    srow[threadIdx.x] = load_task(d, t);
    ssame[threadIdx.x] = r.same_prev ? r.same_prev[i] : 0;
  }
  __syncthreads();
  const uint32_t n0 = (blockIdx.x * 256 + threadIdx.x) * K1_NPT;
  const uint32_t lane = threadIdx.x & 63;
  NodeVals nv[K1_NPT];
#pragma unroll
  for (int j = 0; j < K1_NPT; j++) nv[j] = load_node(d, n0 + j);
  uint32_t res[K1_NPT];
#pragma unroll
  for (int j = 0; j < K1_NPT; j++) res[j] = 0;
  const size_t mstride = d.NP / 32;
  for (uint32_t rr = 0; rr < nr; rr++) {
    const TaskVals tv = srow[rr];
    if (!(ssame[rr] && rr > 0)) {
#pragma unroll
      for (int j = 0; j < K1_NPT; j++) res[j] = eval_pair(d, tv, nv[j], n0 + j, r.fit_mode);
    }
    const size_t row = row0 + rr;
    uint2 pk;
    pk.x = (res[0] & 0xFFFFu) | (res[1] << 16);
    pk.y = (res[2] & 0xFFFFu) | (res[3] << 16);
    *reinterpret_cast<uint2 *>(r.score + row * d.NP + n0) = pk;
    uint32_t nib = ((res[0] >> 16) & 1u) | (((res[1] >> 16) & 1u) << 1) | (((res[2] >> 16) & 1u) << 2) | (((res[3] >> 16) & 1u) << 3);
    uint32_t w = nib << (4 * (lane & 7));
    w |= __shfl_xor(w, 1);
    w |= __shfl_xor(w, 2);
    w |= __shfl_xor(w, 4);
    if ((lane & 7) == 0) r.maskw[row * mstride + (n0 >> 5)] = w;
  }
}

// ------------------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long u = __shfl_xor(v, o);
    v = u > v ? u : v;
  }
  return v;
}
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
// K3: segmented arg-max / top-K.  One wave per row; the row (u16 scores + mask bits) is streamed with 16-byte loads.
// Pass A finds the highest score below the previous level, pass B collects that level's nodes in ascending index
// order by ballot + prefix popcount, stopping as soon as K candidates exist.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_argmax(KbDev d, KbRound r) {
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t row = blockIdx.x * 4 + wave;
  if (row >= r.n_rows) return;
  const uint4 *srow = reinterpret_cast<const uint4 *>(r.score + (size_t)row * d.NP);
  const uint8_t *mrow = reinterpret_cast<const uint8_t *>(r.maskw + (size_t)row * (d.NP / 32));
  const uint32_t K = r.topk;
  unsigned long long *out = r.keys + (size_t)row * K;
  const uint32_t nchunk = d.NP / 8;
  uint32_t found = 0;
  int cur = 0x10000;
  while (found < K) {
    int m = -1;
    for (uint32_t c = lane; c < nchunk; c += 64) {
      uint32_t mb = mrow[c];
      if (mb) {
        uint4 s = srow[c];
        uint32_t w[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
          int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
          if (((mb >> e) & 1u) && sc < cur && sc > m) m = sc;
        }
      }
    }
    m = wave_max_i32(m);
    if (m < 0) break;
    for (uint32_t base = 0; base < nchunk && found < K; base += 64) {
      uint32_t c = base + lane;
      uint32_t match = 0;
      if (c < nchunk) {
        uint32_t mb = mrow[c];
        if (mb) {
          uint4 s = srow[c];
          uint32_t w[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
          for (int e = 0; e < 8; e++) {
            int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
            if (((mb >> e) & 1u) && sc == m) match |= 1u << e;
          }
        }
      }
      uint32_t cnt = __popc(match);
      uint32_t pre = 0, total = 0;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        unsigned long long b = __ballot((match >> e) & 1u);
        pre += lane_prefix_popc(b, lane);
        total += __popcll(b);
      }
      (void)cnt;
      uint32_t pos = found + pre;
      while (match && pos < K) {
        int e = __ffs(match) - 1;
        match &= match - 1;
        out[pos] = KB_KEY(m, c * 8 + e);
        pos++;
      }
      found += total;
    }
    if (found > K) found = K;
    cur = m;
  }
  for (uint32_t i = found + lane; i < K; i += 64) out[i] = 0ull;
}

// ------------------------------------------------------------------------------------------------------------
// K5: sequential commit.  One 1024-thread workgroup walks the window in reference order.  The dirty set (nodes
// that received a task in this round) lives in LDS as a bitmap + list; clean columns come from the round's
// matrix (top-K keys, or the stored row when the candidates are exhausted), dirty columns are re-evaluated live.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool bit_test(const uint32_t *bm, uint32_t n) { return (bm[n >> 5] >> (n & 31)) & 1u; }

__global__ void __launch_bounds__(KB_K5_THREADS) k_commit(KbDev d, KbRound r) {
  extern __shared__ uint32_t dirty_bm[];   // NP/32 words
  __shared__ unsigned long long red[KB_K5_THREADS / 64];
  __shared__ unsigned long long s_best, s_cand;
  __shared__ uint32_t s_ndirty, s_stop, s_exh, s_fallbacks, s_rescans;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint32_t w = tid; w < d.NP / 32; w += KB_K5_THREADS) dirty_bm[w] = 0;
  if (tid == 0) { s_ndirty = 0; s_stop = 0; s_fallbacks = 0; s_rescans = 0; s_exh = 0; s_cand = 0; }
  __syncthreads();
  const uint32_t K = r.topk;
  const size_t mstride_b = (size_t)d.NP / 8;
  for (uint32_t i = 0; i < r.n_rows; i++) {
    const uint32_t t = r.rows ? r.rows[i] : r.row_task0 + i;
    const TaskVals tv = load_task(d, t);
    const uint32_t nd = s_ndirty;
    unsigned long long key = 0;
    uint32_t exhausted = 1;
    if (r.keys) {
      if (tid < 64) {
        unsigned long long ck = (tid < K) ? r.keys[(size_t)i * K + tid] : 0ull;
        bool nz = ck != 0ull;
        bool dirty = nz && bit_test(dirty_bm, KB_KEY_NODE(ck));
        unsigned long long clean = (nz && !dirty) ? ck : 0ull;
        clean = wave_max_u64(clean);
        uint32_t nnz = __popcll(__ballot(nz));
        if (tid == 0) { s_cand = clean; s_exh = (clean == 0ull && nnz == K) ? 1u : 0u; }
      }
      __syncthreads();
      exhausted = s_exh;
      if (tid == 0) key = s_cand;
    }
    bool live_all = false;
    if (exhausted) {
      if (r.use_rows) {
        // clean columns from the stored row: 8 nodes (16 B of scores + 1 mask byte + 1 dirty byte) per step
        const uint4 *srow = reinterpret_cast<const uint4 *>(r.score + (size_t)i * d.NP);
        const uint8_t *mrow = reinterpret_cast<const uint8_t *>(r.maskw) + (size_t)i * mstride_b;
        const uint8_t *dbytes = reinterpret_cast<const uint8_t *>(dirty_bm);
        for (uint32_t c = tid; c < d.NP / 8; c += KB_K5_THREADS) {
          uint32_t mb = mrow[c] & ~((uint32_t)dbytes[c]);
          if (mb) {
            uint4 s = srow[c];
            uint32_t w[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
            for (int e = 0; e < 8; e++) {
              if ((mb >> e) & 1u) {
                uint32_t sc = (w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                unsigned long long k2 = KB_KEY(sc, c * 8 + e);
                key = k2 > key ? k2 : key;
              }
            }
          }
        }
      } else {
        // rows live on another rank: re-evaluate every node against the live state
        live_all = true;
        for (uint32_t n = tid; n < d.N; n += KB_K5_THREADS) {
          NodeVals nv = load_node(d, n);
          uint32_t res = eval_pair(d, tv, nv, n, r.fit_mode);
          if (res) { unsigned long long k2 = KB_KEY(res & 0xFFFFu, n); key = k2 > key ? k2 : key; }
        }
      }
      if (tid == 0 && r.keys) { s_fallbacks++; if (live_all) s_rescans++; }
    }
    if (!live_all) {
      for (uint32_t dd = tid; dd < nd; dd += KB_K5_THREADS) {
        uint32_t n = r.dirty_list[dd];
        NodeVals nv = load_node(d, n);
        uint32_t res = eval_pair(d, tv, nv, n, r.fit_mode);
        if (res) { unsigned long long k2 = KB_KEY(res & 0xFFFFu, n); key = k2 > key ? k2 : key; }
      }
    }
    key = wave_max_u64(key);
    if (lane == 0) red[wave] = key;
    __syncthreads();
    if (tid < 64) {
      unsigned long long k2 = (tid < KB_K5_THREADS / 64) ? red[tid] : 0ull;
      k2 = wave_max_u64(k2);
      if (tid == 0) s_best = k2;
    }
    __syncthreads();
    const unsigned long long best = s_best;
    if (best == 0ull) {
      if (r.backfill) {   // backfill.go:50-66: no node passes the predicates -> the task simply stays Pending
        if (tid == 0) { r.dec_node[i] = KB_NONE_U32; r.dec_kind[i] = 0; }
        __syncthreads();
        continue;
      }
      // allocate.go:144-148: no feasible node -> the job is abandoned; the host re-plans from here
      if (tid == 0) { r.result[0] = i; r.result[1] = KB_REASON_NO_FEASIBLE; r.result[2] = nd; r.result[3] = s_fallbacks; r.result[4] = s_rescans; }
      return;
    }
    if (tid == 0) {
      const uint32_t n = KB_KEY_NODE(best);
      uint32_t kind = 0;
      if (!r.backfill) {   // allocate.go:160: InitResreq.LessEqual(node.Idle) ? Allocate : Pipeline
        bool fi = le_eps(tv.init0, d.idle[n], EPS_CPU) && le_eps(tv.init1, d.idle[(size_t)d.NP + n], EPS_MEM);
        uint32_t a = tv.active >> 2, dd = 2;
        while (a) {
          if (a & 1u) fi = fi && le_eps(d.t_init[(size_t)dd * d.T + t], d.idle[(size_t)dd * d.NP + n], EPS_SCALAR);
          a >>= 1; dd++;
        }
        kind = fi ? 0u : 1u;
      }
      // NodeInfo.AddTask (node_info.go:172-212): Allocated -> Idle.Sub(Resreq); Pipelined -> Releasing.Sub(Resreq)
      double *vec = kind ? d.rel : d.idle;
      const double r0 = d.t_res[t], r1 = d.t_res[(size_t)d.T + t];
      vec[n] -= r0;
      vec[(size_t)d.NP + n] -= r1;
      const uint32_t has_map = kind ? 1u : d.nmask[n];   // Sub returns early when the receiver's scalar map is nil (resource_info.go:148-153)
      uint32_t km = d.t_resmask[t];
      if (km && has_map) {
        uint32_t dd = 2;
        while (km) {
          if (km & 1u) vec[(size_t)dd * d.NP + n] -= d.t_res[(size_t)dd * d.T + t];
          km >>= 1; dd++;
        }
      }
      d.nzc[n] += tv.nzc;     // the pod is now in ni.Tasks: k8s NodeInfo rebuilt from it (nodeinfo/node_info.go:502-517)
      d.nzm[n] += tv.nzm;
      d.podcnt[n] += 1;
      d.t_status[t] = kind ? KB_TASK_PIPELINED : KB_TASK_ALLOCATED;
      d.t_node[t] = n;
      d.t_counted[t] = 1;
      r.dec_node[i] = n;
      r.dec_kind[i] = kind;
      if (r.delta && i >= r.own_row0 && i < r.own_row1) {
        // per-node committed deltas of the rows this rank owns: [dIdle R][dRel R][dnzc][dnzm][dpodcnt] x NP
        double *dv = r.delta + (size_t)(kind ? d.R : 0) * d.NP;
        dv[n] -= r0;
        dv[(size_t)d.NP + n] -= r1;
        uint32_t km2 = d.t_resmask[t];
        if (km2 && has_map) {
          uint32_t dd = 2;
          while (km2) {
            if (km2 & 1u) dv[(size_t)dd * d.NP + n] -= d.t_res[(size_t)dd * d.T + t];
            km2 >>= 1; dd++;
          }
        }
        double *tail = r.delta + (size_t)2 * d.R * d.NP;
        tail[n] += (double)tv.nzc;
        tail[(size_t)d.NP + n] += (double)tv.nzm;
        tail[(size_t)2 * d.NP + n] += 1.0;
      }
      if (!bit_test(dirty_bm, n)) {
        dirty_bm[n >> 5] |= 1u << (n & 31);
        r.dirty_list[nd] = n;
        s_ndirty = nd + 1;
      }
      if (kind == 1u) {   // the host speculated "Allocated": stop after a Pipeline so it can re-plan
        r.result[0] = i + 1; r.result[1] = KB_REASON_PIPELINED; r.result[2] = s_ndirty; r.result[3] = s_fallbacks; r.result[4] = s_rescans;
        s_stop = 1;
      }
    }
    __threadfence_block();
    __syncthreads();
    if (s_stop) return;
  }
  if (tid == 0) { r.result[0] = r.n_rows; r.result[1] = KB_REASON_DONE; r.result[2] = s_ndirty; r.result[3] = s_fallbacks; r.result[4] = s_rescans; }
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
void kb_launch_matrix(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  dim3 grid(d.NP / (256 * K1_NPT), (r.n_rows + K1_TR - 1) / K1_TR);
  hipLaunchKernelGGL(k_matrix, grid, dim3(256), 0, (hipStream_t)stream, d, r);
}
void kb_launch_argmax(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  hipLaunchKernelGGL(k_argmax, dim3((r.n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, d, r);
}
void kb_launch_commit(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  size_t sh = (size_t)(d.NP / 32) * sizeof(uint32_t);
  hipLaunchKernelGGL(k_commit, dim3(1), dim3(KB_K5_THREADS), sh, (hipStream_t)stream, d, r);
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
