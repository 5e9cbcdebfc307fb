// kb_eval.hpp — device-side arithmetic shared by the matrix kernel (kb_kernels.hip) and the commit kernel (kb_commit.hip):
// Resource.LessEqual's epsilon compare (api/resource_info.go:268-302) and nodeorder's three resource scorers
// (vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/{least_requested,most_requested,balanced_resource_allocation}.go).
#pragma once
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


// ---- the same scorers in float64 only --------------------------------------------------------------------------------------
// Every quantity on this path is an integer below 2^48 (kb_session_load rejects larger ones), so it is exact as a double, and
// so are 10 * req (< 2^52) and q * cap for q <= 11.  floor(10 * req / cap): q = trunc(10 req * RN(1/cap)) is off by at most
// one, and the remainder 10 req - q cap is an integer below 2^49 in magnitude, which one fused multiply-add delivers exactly
// (explicit fma: -ffp-contract=off only stops the compiler from forming them on its own).  No 64-bit integer multiply, no
// int64 <-> double conversion: about a third of the instructions of score_core, same results bit for bit
// (tests/test_gpu_parity.py compares every score of the 10k x 1k matrix with the oracle's integer arithmetic).
__device__ __forceinline__ int div10_f64(double req, double cap, double inv_cap, int &rem_nonzero) {
  const double a = req * 10.0;
  double q = trunc(a * inv_cap);
  double rem = __builtin_fma(-q, cap, a);
  if (rem < 0.0) { q -= 1.0; rem += cap; }
  else if (rem >= cap) { q += 1.0; rem -= cap; }
  rem_nonzero = rem != 0.0;
  return (int)q;
}
// n_* / t_*: nonzeroRequest sums of the node and of the pod, ac / am: nodeinfo.allocatableResource, all as (exact) doubles
__device__ __forceinline__ uint32_t score_core_f64(double t_nzc, double t_nzm, double n_nzc, double n_nzm, double ac, double am,
                                                   double inv_ac, double inv_am, int wL, int wM, int wB) {
  const double rc = n_nzc + t_nzc, rm = n_nzm + t_nzm;   // resource_allocation.go:100-112
  int lc = 0, mc = 0, lm = 0, mm = 0, rem;
  if (!(ac == 0.0 || rc > ac)) { mc = div10_f64(rc, ac, inv_ac, rem); lc = 10 - mc - rem; }   // most/least_requested.go
  if (!(am == 0.0 || rm > am)) { mm = div10_f64(rm, am, inv_am, rem); lm = 10 - mm - rem; }
  const int least = (lc + lm) / 2, most = (mc + mm) / 2;
  const double cf = (ac == 0.0) ? 1.0 : rc / ac;      // balanced_resource_allocation.go:74-79
  const double mf = (am == 0.0) ? 1.0 : rm / am;
  int bal = 0;
  if (!(cf >= 1.0 || mf >= 1.0)) bal = (int)((1.0 - fabs(cf - mf)) * 10.0);
  return (uint32_t)(least * wL + most * wM + bal * wB);
}
