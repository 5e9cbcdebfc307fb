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

// Resource.LessEqual's per-dimension test (resource_info.go:268-302): l < r || |l - r| < eps.  With s = RN(l - r): l < r gives s <= 0 < eps,
// and l >= r gives s = |l - r| rounded (rounding is monotone and sign-preserving), so the test is s < eps: one subtraction, one compare.
__device__ __forceinline__ bool le_eps(double l, double r, double eps) { return (l - r) < eps; }

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
// RN(a / b), the IEEE-754 correctly rounded quotient, for INTEGERS 0 <= a < b < 2^48 carried as doubles, given inv_b = RN(1 / b)
// (the session's per-node reciprocal, a true division on the host): three operations instead of the ~12 of the generic expansion
// (v_div_scale x2, v_rcp_f64, four Newton fmas, mul, fma, v_div_fmas, v_div_fixup), which BalancedResourceAllocation's two
// fractions (balanced_resource_allocation.go:74-79) paid per evaluation.  Proof, with q = a / b in [2^e, 2^(e+1)), ulp = 2^(e-52):
//   * inv_b = (1 + eps) / b, |eps| <= 2^-53; q0 = RN(a inv_b) = q (1 + eps)(1 + eps2), so |q - q0| < 2 ulp (1 + 2^-53);
//   * r0 = a - q0 b is an integer multiple of ulp(q0) >= ulp / 2 of magnitude < 2 ulp b (1 + ..), i.e. fewer than 2^51 units:
//     representable, so the fma delivers it exactly;
//   * the last fma rounds X = q0 + r0 inv_b = q + (q - q0) eps, |X - q| < 2^-52 ulp;
//   * a rounding boundary of that binade is m = (2M + 1) 2^(e-53); q - m = (a 2^(53-e) - b (2M + 1)) / (b 2^(53-e)) has a non-zero
//     integer numerator (b < 2^48 cannot supply 2^(53-e) >= 2^54), so |q - m| >= 2^(e-53) / b > 2^-49 ulp  >  |X - q|:
//     X and q lie on the same side of every boundary, RN(X) = RN(q).
// a = 0 gives 0; callers discard the value when a >= b or b == 0 (inv_b is then inf / the value is a NaN nobody selects).
// tests/test_eval_core_cpu.py runs the same three operations on the host against `/` (random, structured and near-boundary operands).
__device__ __forceinline__ double div_small_f64(double a, double b, double inv_b) {
  const double q0 = a * inv_b;
  const double r0 = __builtin_fma(-q0, b, a);
  return __builtin_fma(r0, inv_b, q0);
}

// n_* / t_*: nonzeroRequest sums of the node and of the pod, ac / am: nodeinfo.allocatableResource, all as (exact) doubles.
// Straight-line (selects, no branches: the matrix kernel is VALU-bound when task shapes are diverse and a divergent branch per
// guard cost more than the arithmetic it skipped).  Guards, in the reference's terms:
//   least / most:  capacity == 0 || requested > capacity -> 0                       (least_requested.go / most_requested.go)
//   balanced:      cpuFraction >= 1 || memoryFraction >= 1 -> 0, fraction = 1 when capacity == 0; RN(rc / ac) >= 1 <=> rc >= ac
//                  for integers below 2^48 (rc < ac gives rc / ac <= 1 - 2^-48, which does not round up to 1), so the guard is
//                  rc < ac && rm < am — and exactly then div_small_f64's precondition holds.
__device__ __forceinline__ uint32_t score_core_f64(double t_nzc, double t_nzm, double n_nzc, double n_nzm, double ac, double am,
                                                   double inv_ac, double inv_am, int wL, int wM, int wB) {
  const double rc = n_nzc + t_nzc, rm = n_nzm + t_nzm;   // resource_allocation.go:100-112
  // floor(10 rc / ac) and "the division left a remainder" (kb_eval.hpp: div10_f64, written out without its branches)
  const double a_c = rc * 10.0, a_m = rm * 10.0;
  const double qc = trunc(a_c * inv_ac), qm = trunc(a_m * inv_am);
  const double remc = __builtin_fma(-qc, ac, a_c), remm = __builtin_fma(-qm, am, a_m);
  const int okc = (ac > 0.0) & (rc <= ac), okm = (am > 0.0) & (rm <= am);
  int mc = (int)qc + (remc >= ac) - (remc < 0.0), mm = (int)qm + (remm >= am) - (remm < 0.0);
  const int exc = (remc == 0.0) | (remc == ac), exm = (remm == 0.0) | (remm == am);   // exact division: Least = 10 - Most
  int lc = 9 - mc + exc, lm = 9 - mm + exm;                                        // (cap - req) * 10 / cap = 10 - Most - (remainder != 0)
  mc = okc ? mc : 0; lc = okc ? lc : 0;
  mm = okm ? mm : 0; lm = okm ? lm : 0;
  const int least = (lc + lm) >> 1, most = (mc + mm) >> 1;     // non-negative: / 2
  const double cf = div_small_f64(rc, ac, inv_ac), mf = div_small_f64(rm, am, inv_am);   // balanced_resource_allocation.go:74-79
  int bal = (int)((1.0 - fabs(cf - mf)) * 10.0);
  bal = ((rc < ac) & (rm < am)) ? bal : 0;
  return (uint32_t)(least * wL + most * wM + bal * wB);
}
