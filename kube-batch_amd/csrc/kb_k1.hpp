// kb_k1.hpp — the matrix kernel's per-pair evaluation (K1): a task row against a node, as kb_kernels.hip's k_matrix / k_matrix_runs / k_probe
// run it and as the repair of overlapped candidate lists (k_repair) re-runs it for the predecessor's nodes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "kb_device.h"
#include "kb_eval.hpp"

// K1's views of a task row and of a node: the four quantities the scorers divide are carried as doubles (exact below 2^53),
// converted once per task row / once per node and thread, so that an evaluation is float64 arithmetic only (kb_eval.hpp:
// score_core_f64 — about a third fewer instructions than the int64 form; this kernel is VALU-bound when task shapes are diverse).
struct K1Task {
  double init0, init1, nzc, nzm;
  uint32_t cls, active, task, pad;
  unsigned long long conf;   // host-port bits that conflict with this pod's ports (0: none)
  uint32_t crow, pad2;
};
struct K1Node {
  double idle0, idle1, rel0, rel1, ac, am, nzc, nzm, inv_ac, inv_am;
  uint32_t cls;
  int slots;   // Allocatable.MaxTaskNum > len(pods)  (predicates.go:127 fails on <=)
  int valid;   // node index < N
  unsigned long long ports;
};
__device__ __forceinline__ K1Task k1_task(const KbDev &d, uint32_t t) {
  const TaskVals v = load_task(d, t);
  K1Task k;
  k.init0 = v.init0; k.init1 = v.init1; k.nzc = (double)v.nzc; k.nzm = (double)v.nzm;
  k.cls = v.cls; k.active = v.active; k.task = v.task; k.conf = v.conf;
  k.pad = (d.t_ip_checks && d.t_ip_checks[t]) ? 1u : 0u;   // bit 0: the pod has inter-pod predicate checks
  // bits 1..: unused.  crow: the task class's row of the static-predicate table (bit nc) when there are at most 32 node classes — the
  // class test is then a shift of a scalar, not a dependent load from the global bit table per pair
  k.crow = (d.crows != nullptr && d.n_nc <= 32) ? d.crows[(size_t)v.cls * 8] : 0u;
  return k;
}
// PodAffinityChecker.InterPodAffinityMatches (vendor/.../algorithm/predicates/predicates.go:1261-1290, meta == nil) on the kb_interpod
// counters: a positive count in the node's domain forbids (existing pods' anti-affinity :1400-1441, the pod's own :1535-1543) or is
// required (the pod's own affinity, unless no pod matches at all and the pod matches its own terms: :1519-1566)
__device__ __forceinline__ bool interpod_ok(const KbDev &d, uint32_t t, uint32_t node) {
  for (uint32_t w = 0; w < d.ip_Wc; w++) {
    unsigned long long fb = d.t_ip_forbid[(size_t)t * d.ip_Wc + w];
    while (fb) {
      const uint32_t c = 64u * w + (uint32_t)__ffsll((unsigned long long)fb) - 1u;
      fb &= fb - 1ull;
      const uint32_t dom = d.ip_ctr_dom[(size_t)c * d.NP + node];
      if (dom != KB_NONE_U32 && d.ip_ctr_count[(size_t)c * d.ip_D + dom] > 0) return false;
    }
  }
  const uint32_t r = d.t_ip_req[t];
  if (r != 0xFFFFu) {
    const uint32_t dom = d.ip_ctr_dom[(size_t)r * d.NP + node];
    if (!(dom != KB_NONE_U32 && d.ip_ctr_count[(size_t)r * d.ip_D + dom] > 0))
      if (d.ip_ctr_total[r] > 0 || !d.t_ip_self[t]) return false;
  }
  return true;
}
__device__ __forceinline__ K1Node k1_node(const KbDev &d, uint32_t n) {
  const NodeVals v = load_node(d, n);
  K1Node k;
  k.idle0 = v.idle0; k.idle1 = v.idle1; k.rel0 = v.rel0; k.rel1 = v.rel1;
  k.ac = (double)v.ac; k.am = (double)v.am; k.nzc = (double)v.nzc; k.nzm = (double)v.nzm; k.inv_ac = v.inv_ac; k.inv_am = v.inv_am;
  k.cls = v.cls; k.slots = v.slots; k.valid = v.valid; k.ports = v.ports;
  return k;
}

// A task row is the same for every lane of a workgroup, but it reaches the lanes through LDS, i.e. in VGPRs: every guard on it (active
// scalar dimensions, inter-pod checks) then compiles to a divergent exec-mask branch and every operand to a vector register.
// readfirstlane moves the row into SGPRs once per row: the guards become scalar branches, the row's doubles scalar operands.
__device__ __forceinline__ double k1_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t k1_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ K1Task k1_uniform(const K1Task &v) {
  K1Task k;
  k.init0 = k1_uniform(v.init0); k.init1 = k1_uniform(v.init1); k.nzc = k1_uniform(v.nzc); k.nzm = k1_uniform(v.nzm);
  k.cls = k1_uniform(v.cls); k.active = k1_uniform(v.active); k.task = k1_uniform(v.task); k.pad = k1_uniform(v.pad);
  k.conf = ((unsigned long long)k1_uniform((uint32_t)(v.conf >> 32)) << 32) | k1_uniform((uint32_t)(v.conf & 0xFFFFFFFFull));
  k.crow = k1_uniform(v.crow); k.pad2 = 0;
  return k;
}

// One task row against the NPT consecutive nodes of a thread: res[j] = 0 if infeasible, else 0x10000 | score.  `t` is wave-uniform
// (k1_uniform), so every `if` / `while` below tests a scalar (policy, or a field of the row) and is a scalar branch taken once for
// all NPT nodes; the per-lane part is straight-line (bitwise & | on the compare results, no short-circuit control flow).
template <int NPT>
__device__ __forceinline__ void eval_row(const KbDev &d, const K1Task &t, const K1Node (&n)[NPT], uint32_t n0, int fit_mode, uint32_t (&res)[NPT]) {
  int ok[NPT];
#pragma unroll
  for (int j = 0; j < NPT; j++) ok[j] = n[j].valid;
  if (fit_mode) {   // allocate.go:81: !InitResreq.LessEqual(Idle) && !InitResreq.LessEqual(Releasing) -> fail
    int fi[NPT], fr[NPT];
#pragma unroll
    for (int j = 0; j < NPT; j++) {
      fi[j] = (int)le_eps(t.init0, n[j].idle0, EPS_CPU) & (int)le_eps(t.init1, n[j].idle1, EPS_MEM);
      fr[j] = (int)le_eps(t.init0, n[j].rel0, EPS_CPU) & (int)le_eps(t.init1, n[j].rel1, EPS_MEM);
    }
    uint32_t a = t.active >> 2;     // scalar dims with InitResreq > 10 (resource_info.go:286-299); the others are skipped by LessEqual
    while (a) {
      const uint32_t dd = 2u + (uint32_t)__builtin_ctz(a);
      a &= a - 1u;
      const double l = d.t_init[(size_t)dd * d.T + t.task];
      const double *pi = d.idle + (size_t)dd * d.NP + n0, *pr = d.rel + (size_t)dd * d.NP + n0;   // rows are padded to NP: in bounds for every lane
      double vi[NPT], vr[NPT];
      if (NPT == 4) {   // 32 contiguous, 32-byte aligned bytes per thread: two 16-byte loads per vector
        const double2 i0 = reinterpret_cast<const double2 *>(pi)[0], i1 = reinterpret_cast<const double2 *>(pi)[1];
        const double2 r0 = reinterpret_cast<const double2 *>(pr)[0], r1 = reinterpret_cast<const double2 *>(pr)[1];
        vi[0] = i0.x; vi[1 % NPT] = i0.y; vi[2 % NPT] = i1.x; vi[3 % NPT] = i1.y;
        vr[0] = r0.x; vr[1 % NPT] = r0.y; vr[2 % NPT] = r1.x; vr[3 % NPT] = r1.y;
      } else {
#pragma unroll
        for (int j = 0; j < NPT; j++) { vi[j] = pi[j]; vr[j] = pr[j]; }
      }
#pragma unroll
      for (int j = 0; j < NPT; j++) { fi[j] &= (int)le_eps(l, vi[j], EPS_SCALAR); fr[j] &= (int)le_eps(l, vr[j], EPS_SCALAR); }
    }
#pragma unroll
    for (int j = 0; j < NPT; j++) ok[j] &= fi[j] | ((fit_mode != 2) & fr[j]);   // 2: backfill, AddTask's Resreq.LessEqual(Idle) only (node_info.go:161-167)
  }
  if (d.pred_enabled) {
#pragma unroll
    for (int j = 0; j < NPT; j++) ok[j] &= n[j].slots & (int)((n[j].ports & t.conf) == 0ull);   // pod count (predicates.go:127), PodFitsHostPorts (:181-190)
    if (d.port_xw) {   // host-port masks of several words: the words behind the first, straight from HBM ([port_xw][NP], rows padded to NP).  A row
                       // that stays inside word 0 carries zeros here (one scalar load per word, no node traffic)
      for (uint32_t w = 0; w < d.port_xw; w++) {
        const unsigned long long cx = d.t_conf_x[(size_t)t.task * d.port_xw + w];
        if (cx == 0ull) continue;
        const unsigned long long *px = d.ports_x + (size_t)w * d.NP + n0;
#pragma unroll
        for (int j = 0; j < NPT; j++) ok[j] &= (int)((px[j] & cx) == 0ull);
      }
    }
    if (d.crows != nullptr && d.n_nc <= 32) {   // the class row is a scalar word (K1Task::crow)
#pragma unroll
      for (int j = 0; j < NPT; j++) ok[j] &= (int)((t.crow >> n[j].cls) & 1u);
    } else if (d.compat) {
#pragma unroll
      for (int j = 0; j < NPT; j++) {
        const uint32_t bit = t.cls * d.n_nc + n[j].cls;
        ok[j] &= (d.compat[bit >> 3] >> (bit & 7)) & 1;
      }
    }
    if (d.t_ip_forbid != nullptr && t.pad) {   // predicates.go:249-262: subject rows of sessions with inter-pod terms only
#pragma unroll
      for (int j = 0; j < NPT; j++)
        if (ok[j]) ok[j] = interpod_ok(d, t.task, n0 + j);
    }
  }
  if (d.score_enabled) {
#pragma unroll
    for (int j = 0; j < NPT; j++) {
      const uint32_t sc = score_core_f64(t.nzc, t.nzm, n[j].nzc, n[j].nzm, n[j].ac, n[j].am, n[j].inv_ac, n[j].inv_am, d.wL, d.wM, d.wB);
      res[j] = ok[j] ? (0x10000u | (sc & 0xFFFFu)) : 0u;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NPT; j++) res[j] = ok[j] ? 0x10000u : 0u;
  }
}
