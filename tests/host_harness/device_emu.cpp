// device_emu.cpp — TEST INFRASTRUCTURE: a sequential CPU restatement of what the HIP kernels of kube-batch_amd/csrc compute, behind the
// very launch wrappers kb_device.h declares (kb_launch_matrix, kb_launch_argmax, kb_launch_commit, ...).  Linked with the engine's
// host sources — kb_engine.cpp, kb_session.cpp, kb_order.cpp, kb_preempt.cpp, compiled UNCHANGED with g++ against hip_mock/ — it gives
// tests/host_harness/build/libkbengine_emu.so: the complete C ABI of include/kb_engine.h on a CPU, so that everything the host side of the
// engine does (ActionRun's plan / absorb / dead shapes / probe, chained rounds and the pinned mailbox, kb_round_* for the sharded path,
// session reset, the aggregate cross-checks) runs against the oracle without a GPU (tests/test_emu_engine_cpu.py).
//
// What this is NOT: it is not part of the product (libkbengine.so has no CPU path and never will: kube-batch_amd/engine.py loads the
// hipcc-built library only), it is not a model of HOW the kernels work (no tiles, no LDS, no waves, no speculation inside the commit),
// and it says nothing about the kernels' own correctness — the `-m gpu` suite compares those with the oracle.  It restates their
// CONTRACT: the same inputs in the same buffers give the same outputs in the same buffers, including the output block, the
// reason codes and the chain word the host protocol depends on.  The per-pair arithmetic (epsilon compares, integer scorers) is the
// product's own kb_eval.hpp, compiled for the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../kube-batch_amd/csrc/kb_waterfill.hpp"   // before kb_eval.hpp: it brings include/kb_engine.h, whose task-status enum kb_eval.hpp names as macros
#include "../../kube-batch_amd/csrc/kb_device.h"
#include "../../kube-batch_amd/csrc/kb_eval.hpp"

namespace {

typedef unsigned long long u64;

inline u64 *out64(const KbRound &r) { return reinterpret_cast<u64 *>(r.result); }
inline bool mask_bit(const uint32_t *mw, uint32_t n) { return (mw[n >> 5] >> (n & 31)) & 1u; }

// PodAffinityChecker.InterPodAffinityMatches on the kb_interpod counters (kb_kernels.hip: interpod_ok)
bool interpod_ok(const KbDev &d, uint32_t t, uint32_t node) {
  for (uint32_t w = 0; w < d.ip_Wc; w++) {
    u64 fb = d.t_ip_forbid[(size_t)t * d.ip_Wc + w];
    for (uint32_t b = 0; b < 64; b++) {
      if (!((fb >> b) & 1ull)) continue;
      const uint32_t c = 64u * w + b;
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

// what an evaluation needs to know about the task side
struct Row {
  double init0, init1, nzc, nzm;
  uint32_t cls, active, task;
  u64 conf;
  bool ip_checks;      // the matrix kernel evaluates the inter-pod predicate for this row
  bool use_crow;       // the commit kernels' class test: the descriptor's row of the class table
  uint32_t crow;
};

Row row_of_task(const KbDev &d, uint32_t t) {
  const TaskVals v = load_task(d, t);
  Row r;
  r.init0 = v.init0; r.init1 = v.init1; r.nzc = (double)v.nzc; r.nzm = (double)v.nzm;
  r.cls = v.cls; r.active = v.active; r.task = t; r.conf = v.conf;
  r.ip_checks = d.t_ip_checks && d.t_ip_checks[t];
  r.use_crow = false; r.crow = 0;
  return r;
}

// One (task, node) evaluation against the LIVE node state: 0 if infeasible, else 0x10000 | score (kb_kernels.hip: eval_pair;
// kb_commit.hip: k9_eval_v, which has no inter-pod test and may use the descriptor's class row)
uint32_t eval_pair(const KbDev &d, const Row &t, uint32_t node, int fit_mode, bool with_interpod) {
  if (node >= d.N) return 0;
  const NodeVals n = load_node(d, node);
  bool ok = true;
  if (fit_mode) {   // allocate.go:81
    bool fi = le_eps(t.init0, n.idle0, EPS_CPU) && le_eps(t.init1, n.idle1, EPS_MEM);
    bool fr = le_eps(t.init0, n.rel0, EPS_CPU) && le_eps(t.init1, n.rel1, EPS_MEM);
    uint32_t a = t.active >> 2, dd = 2;
    while (a) {
      if (a & 1u) {
        const double l = d.t_init[(size_t)dd * d.T + t.task];
        fi = fi && le_eps(l, d.idle[(size_t)dd * d.NP + node], EPS_SCALAR);
        fr = fr && le_eps(l, d.rel[(size_t)dd * d.NP + node], EPS_SCALAR);
      }
      a >>= 1;
      dd++;
    }
    ok = fi || (fit_mode != 2 && fr);
  }
  if (d.pred_enabled) {
    ok = ok && n.slots && ((n.ports & t.conf) == 0ull);
    if (t.use_crow) ok = ok && ((t.crow >> (n.cls & 31)) & 1u);
    else if (d.compat) {
      const uint32_t bit = t.cls * d.n_nc + n.cls;
      ok = ok && ((d.compat[bit >> 3] >> (bit & 7)) & 1);
    }
    if (with_interpod && d.t_ip_forbid != nullptr && t.ip_checks && ok) ok = interpod_ok(d, t.task, node);
    // host-port masks of several words: only the matrix side (K1: kb_k1.hpp eval_row) reads the words behind the first; the commit kernels keep to word 0
    if (with_interpod)
      for (uint32_t w = 0; w < d.port_xw; w++)
        if (d.ports_x[(size_t)w * d.NP + node] & d.t_conf_x[(size_t)t.task * d.port_xw + w]) ok = false;
  }
  if (!ok) return 0;
  uint32_t score = 0;
  if (d.score_enabled)
    score = score_core_f64(t.nzc, t.nzm, (double)n.nzc, (double)n.nzm, (double)n.ac, (double)n.am, n.inv_ac, n.inv_am, d.wL, d.wM, d.wB);
  return 0x10000u | (score & 0xFFFFu);
}

// window row -> descriptor (kb_kernels.hip: gather_row)
void gather_row(const KbDev &d, const KbRound &r, uint32_t i) {
  const uint32_t t = r.rows[i];
  KbRowDesc k;
  std::memset(&k, 0, sizeof(k));
  k.init0 = d.t_init[t]; k.init1 = d.t_init[(size_t)d.T + t];
  k.nzc = d.t_nzc[t]; k.nzm = d.t_nzm[t];
  k.task = t; k.active = d.t_active[t]; k.resmask = d.t_resmask[t]; k.cls = d.t_cls[t];
  k.slot = (uint16_t)r.shape_slot[i];
  k.flags = (d.t_res[t] == k.init0 && d.t_res[(size_t)d.T + t] == k.init1) ? 1 : 0;
  if (d.aff_cls && d.aff_cls[k.cls]) k.flags |= 2;
  if (d.t_ip_subject && d.t_ip_subject[t]) k.flags |= 2;
  bool same = true;
  uint32_t m2 = k.resmask;
  for (int dd = 2; m2; dd++, m2 >>= 1)
    if ((m2 & 1u) && d.t_res[(size_t)dd * d.T + t] != d.t_init[(size_t)dd * d.T + t]) same = false;
  if (same) k.flags |= 4;
  k.crow = d.crows ? d.crows[(size_t)k.cls * 8] : 0xFFFFFFFFu;
  r.desc[i] = k;
}

uint32_t mrow_task(const KbRound &r, uint32_t m) { return r.mrows ? r.mrows[m] : r.mrow_task0 + m; }

// ---- the sequential commit of one window (kb_commit.hip / kb_commit_sel.hip: same decisions) ----
void remember_commit_nodes(const std::vector<uint32_t> &nodes);
unsigned long long g_selected_rows = 0;   // rows committed by the run selection (KB_EMU_RUN_SELECT)
unsigned long long g_select_runs = 0;     // ... the runs they came in
unsigned long long g_select_lanes = 0, g_select_steps = 0;   // ... the node sequences walked for them, and the evaluations those walks cost
void emu_commit(const KbDev &d, const KbRound &r) {
  if (r.n_rows == 0) return;
  u64 *o64 = out64(r);
  if (r.chain_expect != 0u && *r.chain != r.chain_expect) {   // chained to a round that stopped early: skip
    *r.chain = 0u;
    r.result[0] = 0; r.result[1] = KB_REASON_SKIPPED;
    if (r.host_out) {
      r.host_out[0] = (u64)KB_REASON_SKIPPED << 32;
      __atomic_store_n(&r.host_out[KB_OUT_SEQ], r.seq, __ATOMIC_RELEASE);
    }
    return;
  }
  const u64 t_start = kbemu_wall_clock();
  const bool has_aff = (d.aff != nullptr && d.score_enabled) || d.t_ip_subject != nullptr;
  const bool has_ports = d.ports != nullptr;
  const bool use_crow = d.pred_enabled && d.crows != nullptr && d.n_nc <= 32;
  std::vector<uint8_t> dirty(d.NP, 0);
  std::vector<uint32_t> dirty_nodes;
  std::vector<u64> dec(r.n_rows, 0);
  uint32_t n_done = 0, reason = KB_REASON_DONE, dirty_won = 0;
  KbDev dl = d;   // the commit kernels carry their own copies of the policy switches (KbCommitArgs); same values
  uint32_t i = 0;
  // KB_EMU_RUN_SELECT=1 (tests/run_selection_model.py): a run of consecutive rows that are the same in everything a decision
  // depends on is committed by ONE selection instead of row by row — every candidate node's own key sequence (its score after j placements of the
  // shape, while it fits), the prefix minimum of it, the first r of all (node, step) entries by (prefix minimum desc, node asc, step asc).  The
  // contract of the launch does not change: the whole emulated suite must still equal the oracle with the switch on.
  // KB_EMU_RUN_SELECT=2 is the negative control: the real key in the place of the prefix minimum (the premise "a node's keys only fall").
  static const int run_select_mode = [] { const char *v = getenv("KB_EMU_RUN_SELECT"); return v ? atoi(v) : 0; }();
  const bool run_select = run_select_mode != 0;
  auto make_row = [&](const KbRowDesc &k) {
    Row row;
    row.init0 = k.init0; row.init1 = k.init1; row.nzc = (double)k.nzc; row.nzm = (double)k.nzm;
    row.cls = k.cls; row.active = k.active; row.task = k.task;
    row.conf = has_ports ? d.t_conf[k.task] : 0ull;
    row.ip_checks = false;
    row.use_crow = use_crow; row.crow = k.crow;
    return row;
  };
  auto kind_of = [&](const KbRowDesc &k, uint32_t n) -> uint32_t {   // allocate.go:160: InitResreq.LessEqual(node.Idle) -> Allocate, else Pipeline
    if (r.backfill) return 0u;
    bool fi = le_eps(k.init0, d.idle[n], EPS_CPU) && le_eps(k.init1, d.idle[(size_t)d.NP + n], EPS_MEM);
    uint32_t a = k.active >> 2, dd = 2;
    while (a) {
      if (a & 1u) fi = fi && le_eps(d.t_init[(size_t)dd * d.T + k.task], d.idle[(size_t)dd * d.NP + n], EPS_SCALAR);
      a >>= 1;
      dd++;
    }
    return fi ? 0u : 1u;
  };
  // NodeInfo.AddTask (api/node_info.go:172-212): Idle (Allocated) or Releasing (Pipelined) -= Resreq, the pod joins ni.Tasks.
  // Resource.Sub leaves the scalars alone when the receiver's map is nil (resource_info.go:148-153).
  auto add_task = [&](const KbRowDesc &k, uint32_t n, uint32_t kind) {
    const uint32_t t = k.task;
    double *side = kind ? d.rel : d.idle;
    side[n] -= d.t_res[t];
    side[(size_t)d.NP + n] -= d.t_res[(size_t)d.T + t];
    const uint32_t nm = d.nmask[n];
    const bool has_map = kind ? (nm >> 31) != 0 : (nm & 0x7FFFFFFFu) != 0;
    if (k.resmask && has_map) {
      uint32_t m2 = k.resmask;
      for (uint32_t dd = 2; m2; dd++, m2 >>= 1)
        if (m2 & 1u) side[(size_t)dd * d.NP + n] -= d.t_res[(size_t)dd * d.T + t];
    }
    d.nzc[n] += k.nzc;
    d.nzm[n] += k.nzm;
    d.podcnt[n] += 1;
    if (has_ports) d.ports[n] |= d.t_want[t];
  };
  auto same_decision_inputs = [&](uint32_t a, uint32_t b) {   // rows a and b differ in the task id only
    const KbRowDesc &x = r.desc[a], &y = r.desc[b];
    if (x.slot != y.slot || x.flags != y.flags || (x.flags & 2u) || x.init0 != y.init0 || x.init1 != y.init1 || x.nzc != y.nzc || x.nzm != y.nzm ||
        x.active != y.active || x.resmask != y.resmask || x.cls != y.cls || x.crow != y.crow) return false;
    for (int dd = 0; dd < d.R; dd++)
      if (d.t_res[(size_t)dd * d.T + x.task] != d.t_res[(size_t)dd * d.T + y.task] || d.t_init[(size_t)dd * d.T + x.task] != d.t_init[(size_t)dd * d.T + y.task]) return false;
    if (has_ports && (d.t_conf[x.task] != d.t_conf[y.task] || d.t_want[x.task] != d.t_want[y.task])) return false;
    return true;
  };
  struct Entry { uint32_t eff, node, step, kind; };
  while (i < r.n_rows) {
    const KbRowDesc &k = r.desc[i];
    if (has_aff && !r.backfill && (k.flags & 2u) && i != 0) { reason = KB_REASON_RENORM; break; }
    uint32_t i1 = i + 1;
    if (run_select) while (i1 < r.n_rows && same_decision_inputs(i, i1)) i1++;
    if (i1 - i >= 2) {
      const uint32_t rlen = i1 - i;
      const Row row = make_row(k);
      std::vector<uint32_t> cand(dirty_nodes);
      const u64 *list = r.keys + (size_t)k.slot * r.L;
      for (uint32_t e = 0, got = 0; e < r.L && list[e] != 0ull && got < rlen; e++) {   // winners are among the rlen best initial keys
        const uint32_t n = KB_KEY_NODE(list[e]);
        if (!dirty[n]) { cand.push_back(n); got++; }
      }
      std::vector<Entry> ent;
      std::vector<double> sv_idle(d.R), sv_rel(d.R);
      // A lane may stop where its prefix minimum falls below the rlen-th best INITIAL key of the candidates: rlen entries (the first steps of the
      // rlen best) are at or above that key, so nothing below it is among the first rlen.  (Mode 2, the negative control, walks everything.)
      uint32_t floor_key = 0;
      if (run_select_mode == 1 && cand.size() >= rlen) {
        std::vector<uint32_t> k0;
        for (uint32_t n : cand) {
          const uint32_t e = eval_pair(dl, row, n, r.fit_mode, false);
          if (e) k0.push_back(e & 0xFFFFu);
        }
        if (k0.size() >= rlen) {
          std::nth_element(k0.begin(), k0.begin() + (rlen - 1), k0.end(), [](uint32_t a, uint32_t b) { return a > b; });
          floor_key = k0[rlen - 1];
        }
      }
      for (uint32_t n : cand) {   // one lane per node: its own sequence, from its own state
        for (int dd = 0; dd < d.R; dd++) { sv_idle[dd] = d.idle[(size_t)dd * d.NP + n]; sv_rel[dd] = d.rel[(size_t)dd * d.NP + n]; }
        const long long sv_nzc = d.nzc[n], sv_nzm = d.nzm[n];
        const int sv_pods = d.podcnt[n];
        const u64 sv_ports = has_ports ? d.ports[n] : 0ull;
        uint32_t eff = 0xFFFFFFFFu;
        __atomic_fetch_add(&g_select_lanes, 1ull, __ATOMIC_RELAXED);
        for (uint32_t j = 0; j < rlen; j++) {
          const uint32_t e = eval_pair(dl, row, n, r.fit_mode, false);
          __atomic_fetch_add(&g_select_steps, 1ull, __ATOMIC_RELAXED);
          if (!e) break;                                  // feasibility only shrinks inside a run: the sequence ends here
          eff = run_select_mode == 2 ? (e & 0xFFFFu) : std::min(eff, e & 0xFFFFu);
          if (eff < floor_key) break;                     // below the floor: neither this entry nor anything behind it can be picked
          const uint32_t kind = kind_of(k, n);
          ent.push_back(Entry{eff, n, j, kind});
          if (kind) break;                                // a Pipeline ends the round when it is picked: nothing behind it is ever reached
          add_task(k, n, 0u);
        }
        for (int dd = 0; dd < d.R; dd++) { d.idle[(size_t)dd * d.NP + n] = sv_idle[dd]; d.rel[(size_t)dd * d.NP + n] = sv_rel[dd]; }
        d.nzc[n] = sv_nzc; d.nzm[n] = sv_nzm; d.podcnt[n] = sv_pods;
        if (has_ports) d.ports[n] = sv_ports;
      }
      std::sort(ent.begin(), ent.end(), [](const Entry &a, const Entry &b) {
        if (a.eff != b.eff) return a.eff > b.eff;
        if (a.node != b.node) return a.node < b.node;
        return a.step < b.step;
      });
      bool stop = false;
      uint32_t q = 0;
      for (; q < rlen && q < ent.size(); q++) {
        const KbRowDesc &kq = r.desc[i + q];
        const uint32_t n = ent[q].node, kind = ent[q].kind;
        if (dirty[n]) dirty_won++;
        add_task(kq, n, kind);
        if (!dirty[n]) { dirty[n] = 1; dirty_nodes.push_back(n); }
        dec[i + q] = (u64)n | ((u64)kind << 32);
        if (kind) { q++; reason = KB_REASON_PIPELINED; stop = true; break; }
      }
      i += q;
      __atomic_fetch_add(&g_selected_rows, (unsigned long long)q, __ATOMIC_RELAXED);
      __atomic_fetch_add(&g_select_runs, 1ull, __ATOMIC_RELAXED);
      if (stop) break;
      if (q < rlen) {   // the run's shape has no feasible node left
        if (r.backfill) { for (; i < i1; i++) dec[i] = (u64)KB_NONE_U32; continue; }
        reason = KB_REASON_NO_FEASIBLE;
        break;
      }
      continue;
    }
    const Row row = make_row(k);
    // best node the round already changed: exact re-evaluation against its live state
    u64 best_dirty = 0;
    for (uint32_t n : dirty_nodes) {
      const uint32_t e = eval_pair(dl, row, n, r.fit_mode, false);
      if (e) best_dirty = std::max(best_dirty, ((u64)((e & 0xFFFFu) + 1u) << 32) | (u64)(0xFFFFFFFFu - n));
    }
    // best clean node: the first entry of the shape's sorted list (round-start state) whose node is still untouched
    u64 best_clean = 0;
    const u64 *list = r.keys + (size_t)k.slot * r.L;
    for (uint32_t e = 0; e < r.L && list[e] != 0ull; e++) {
      const uint32_t n = KB_KEY_NODE(list[e]);
      if (!dirty[n]) { best_clean = ((u64)(KB_KEY_SCORE(list[e]) + 1u) << 32) | (u64)(0xFFFFFFFFu - n); break; }
    }
    if (best_dirty == 0 && best_clean == 0) {
      if (r.backfill) { dec[i] = (u64)KB_NONE_U32; i++; continue; }   // backfill.go:50-66: the task stays Pending
      reason = KB_REASON_NO_FEASIBLE;                                 // allocate.go:144-148: the host re-plans from here
      break;
    }
    const u64 win = std::max(best_dirty, best_clean);
    const uint32_t n = 0xFFFFFFFFu - (uint32_t)(win & 0xFFFFFFFFull);
    if (best_dirty >= best_clean) dirty_won++;
    const uint32_t kind = kind_of(k, n);
    add_task(k, n, kind);
    if (!dirty[n]) { dirty[n] = 1; dirty_nodes.push_back(n); }
    dec[i] = (u64)n | ((u64)kind << 32);
    i++;
    if (kind) { reason = KB_REASON_PIPELINED; break; }   // a Pipeline ends the speculated order: the host re-plans
  }
  n_done = i;
  // ---- epilogue: decision records, the task-table side of ssn.Allocate / ssn.Pipeline, inter-pod counters, multi-GPU deltas
  for (uint32_t j = 0; j < n_done; j++) {
    const u64 rec = dec[j];
    r.dec[j] = rec;
    const uint32_t n = (uint32_t)rec, kind = (uint32_t)(rec >> 32);
    if (n == KB_NONE_U32) continue;
    const KbRowDesc &k = r.desc[j];
    const uint32_t t = k.task;
    d.t_status[t] = kind ? KB_TASK_PIPELINED : KB_TASK_ALLOCATED;
    d.t_node[t] = n;
    d.t_counted[t] = 1;
    if (!kind) d.j_allocated[d.t_job[t]] = 1;
    if (d.t_ip_cls_inc) {
      for (uint32_t w = 0; w < d.ip_Wp; w++) {
        const u64 cm = d.t_ip_cls_inc[(size_t)t * d.ip_Wp + w];
        for (uint32_t b = 0; b < 64; b++)
          if ((cm >> b) & 1ull) d.ip_cls_unbound[(size_t)(64u * w + b) * d.NP + n] += 1;
      }
      *d.ip_z = std::min(*d.ip_z, n);
      if (!kind)
        for (uint32_t w = 0; w < d.ip_Wc; w++) {
          const u64 im = d.t_ip_inc[(size_t)t * d.ip_Wc + w];
          for (uint32_t b = 0; b < 64; b++)
            if ((im >> b) & 1ull) {
              const uint32_t c = 64u * w + b;
              d.ip_ctr_total[c] += 1;
              const uint32_t dm = d.ip_ctr_dom[(size_t)c * d.NP + n];
              if (dm != KB_NONE_U32) d.ip_ctr_count[(size_t)c * d.ip_D + dm] += 1;
            }
        }
    }
    if (r.delta && j >= r.own_row0 && j < r.own_row1) {
      double *dv = r.delta + (size_t)(kind ? d.R : 0) * d.NP;
      dv[n] += -d.t_res[t];
      dv[(size_t)d.NP + n] += -d.t_res[(size_t)d.T + t];
      if (k.resmask) {
        const uint32_t nm = d.nmask[n];
        const bool has_map = kind ? (nm >> 31) != 0 : (nm & 0x7FFFFFFFu) != 0;
        if (has_map) {
          uint32_t m2 = k.resmask;
          for (uint32_t dd = 2; m2; dd++, m2 >>= 1)
            if (m2 & 1u) dv[(size_t)dd * d.NP + n] += -d.t_res[(size_t)dd * d.T + t];
        }
      }
      double *tail = r.delta + (size_t)2 * d.R * d.NP;
      tail[n] += (double)k.nzc;
      tail[(size_t)d.NP + n] += (double)k.nzm;
      tail[(size_t)2 * d.NP + n] += 1.0;
    }
  }
  // the output block: result words, (absent) trace words, stamps
  for (uint32_t w = 2; w < 8; w++) o64[w] = 0;          // words 4..15 of the 32-bit view
  for (uint32_t w = 13; w < KB_OUT_HDR; w++) o64[w] = 0;
  const uint32_t nd = (uint32_t)dirty_nodes.size();
  r.result[0] = n_done; r.result[1] = reason; r.result[2] = nd;
  r.result[3] = dirty_won; r.result[4] = n_done; r.result[5] = 0; r.result[6] = 0; r.result[7] = 0;
  if (r.chain) *r.chain = reason == KB_REASON_DONE ? r.chain_tag : 0u;
  remember_commit_nodes(dirty_nodes);   // what an overlapped matrix launch may have seen half-changed (kb_launch_matrix poisons it)
  o64[KB_OUT_STAMP0 + 2] = t_start;
  o64[KB_OUT_STAMP0 + 3] = kbemu_wall_clock();
  if (r.host_out) {   // fast rounds: header and decision records into the pinned mirror, the sequence number last
    for (uint32_t w = 0; w < KB_OUT_HDR; w++) if (w != KB_OUT_SEQ) r.host_out[w] = o64[w];
    for (uint32_t j = 0; j < n_done; j++) r.host_out[KB_OUT_HDR + j] = dec[j];
    __atomic_store_n(&r.host_out[KB_OUT_SEQ], r.seq, __ATOMIC_RELEASE);
  }
}

double share_of(double l, double r) { return (r == 0.0) ? ((l == 0.0) ? 0.0 : 1.0) : l / r; }

}  // namespace

// ---- LDS budgets: the window the engine plans depends on what the commit kernels can keep in their 160 KiB.  A model of the same
// shape (per-row slots and descriptors, per-shape candidate lists, the dirty bitmap) so that windows come out like the GPU build's
// for the cluster sizes the CPU suite uses; the exact byte counts live with the kernels.
// KB_EMU_LDS_PENALTY=bytes pretends the kernels need that much more: the engine then plans small windows with few shapes each, a
// regime the CPU suite's small clusters never reach otherwise (on the GPU it takes tens of thousands of nodes)
static size_t lds_penalty() {
  const char *v = getenv("KB_EMU_LDS_PENALTY");
  return v ? (size_t)atol(v) : 0;
}
size_t kb_commit_smem_bytes(uint32_t n_rows, uint32_t n_shapes, uint32_t NP, int R) {
  const size_t RS = R > 2 ? (size_t)(R - 2) : 0;
  size_t off = 0;
  off += ((size_t)n_rows + 64) * 13 * 8;
  off += 3 * (size_t)R * 8 + (size_t)n_shapes * RS * 8 + (size_t)n_shapes * 64 + (size_t)n_rows * sizeof(KbRowDesc);
  off = (off + 15) & ~(size_t)15;
  off += (size_t)n_rows * 16 + (size_t)n_rows * 8 + 48 + 256 * 4 + 64 * 4 + 64 * 4 + (size_t)n_shapes * 4 + (size_t)n_shapes * 4;
  off += (size_t)n_shapes * ((size_t)n_rows + 1) * 4 + (size_t)(NP / 32) * 4 + 10500 /* the selection kernel's block */ + lds_penalty();
  return (off + 15) & ~(size_t)15;
}

// ---- launch wrappers (kb_device.h) ----
void kb_launch_gather(const KbDev &d, const KbRound &r, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, r]() {
  if (r.n_rows == 0) return;
  if (KB_CHAIN_BROKEN(r)) return;
  out64(r)[KB_OUT_STAMP0] = kbemu_wall_clock();
  for (uint32_t i = 0; i < r.n_rows; i++) gather_row(d, r, i);
  });
}

// Overlapped rounds (KbRound::ready): on the device the matrix launch of such a round runs on a second stream while the predecessor's commit
// kernel is still changing nodes, so what it sees of THOSE nodes is arbitrary.  The emulated streams execute a launch where it is enqueued
// (or on one worker per stream), which would hide that: the emulated matrix launch therefore POISONS its result for every node the last
// commit changed (feasible with the top score, or infeasible, by a hash) — the repair launch must override exactly these, whatever they hold.
static std::vector<uint32_t> g_last_commit_nodes;
static std::mutex g_last_commit_mu;
namespace {
void remember_commit_nodes(const std::vector<uint32_t> &nodes) {
  std::lock_guard<std::mutex> lk(g_last_commit_mu);
  g_last_commit_nodes = nodes;
}
}

void kb_launch_matrix(const KbDev &d, const KbRound &r, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, r]() {
  if (r.n_mrows == 0) return;
  if (KB_CHAIN_BROKEN(r)) return;
  if (r.gather) {
    out64(r)[KB_OUT_STAMP0] = kbemu_wall_clock();
    for (uint32_t i = 0; i < r.n_rows; i++) gather_row(d, r, i);
  }
  const size_t mstride = d.NP / 32;
  for (uint32_t m = 0; m < r.n_mrows; m++) {
    const Row t = row_of_task(d, mrow_task(r, m));
    uint16_t *sc = r.score + (size_t)m * d.NP;
    uint32_t *mw = r.maskw + (size_t)m * mstride;
    std::memset(mw, 0, mstride * sizeof(uint32_t));
    for (uint32_t n = 0; n < d.NP; n++) {
      const uint32_t e = eval_pair(d, t, n, r.fit_mode, true);
      sc[n] = (uint16_t)(e & 0xFFFFu);
      if (e >> 16) mw[n >> 5] |= 1u << (n & 31);
    }
    // (with asynchronous emulated streams, KB_EMU_ASYNC=1, the launch really runs beside the predecessor and "the last commit" may be an
    // older round whose nodes are final and are NOT repaired: no poison then, the staleness is real)
    static const bool async_streams = getenv("KB_EMU_ASYNC") && getenv("KB_EMU_ASYNC")[0] == '1';
    if (r.ready != nullptr && !async_streams) {
      std::lock_guard<std::mutex> lk(g_last_commit_mu);
      for (uint32_t n : g_last_commit_nodes) {
        if (n >= d.N) continue;
        const uint32_t h = (n * 2654435761u) ^ (m * 40503u) ^ (uint32_t)r.ready_tag;
        if (h & 4u) { sc[n] = (uint16_t)(0xFFFFu - (h >> 20)); mw[n >> 5] |= 1u << (n & 31); }
        else { sc[n] = 0; mw[n >> 5] &= ~(1u << (n & 31)); }
      }
    }
  }
  });
}

void kb_launch_affinity(const KbDev &d, const KbRound &r, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, r]() {
  if (r.n_mrows == 0 || !d.aff || !d.score_enabled) return;
  if (KB_CHAIN_BROKEN(r)) return;
  for (uint32_t m = 0; m < r.n_mrows; m++) {
    const uint32_t tc = d.t_cls[mrow_task(r, m)];
    if (!d.aff_cls[tc]) continue;
    const int32_t *arow = d.aff + (size_t)tc * d.n_nc;
    const uint32_t *mw = r.maskw + (size_t)m * (d.NP / 32);
    uint16_t *sc = r.score + (size_t)m * d.NP;
    int mx = 0;
    for (uint32_t n = 0; n < d.N; n++)
      if (mask_bit(mw, n)) mx = std::max(mx, (int)arow[d.ncls[n]]);
    if (mx == 0) continue;   // reduce.go:28-63: max == 0 leaves zeros
    for (uint32_t n = 0; n < d.N; n++)
      if (mask_bit(mw, n)) sc[n] = (uint16_t)(sc[n] + (10 * arow[d.ncls[n]] / mx) * d.wNA);
  }
  });
}

void kb_launch_interpod(const KbDev &d, const KbRound &r, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, r]() {
  if (r.n_mrows == 0 || !d.t_ip_sig || !d.score_enabled || d.ip_P == 0) return;
  if (KB_CHAIN_BROKEN(r)) return;
  std::vector<long long> cnt(d.N), hist(d.N);
  for (uint32_t m = 0; m < r.n_mrows; m++) {
    const uint32_t sig = d.t_ip_sig[mrow_task(r, m)];
    if (sig == KB_NONE_U32) continue;
    const int32_t *w = d.ip_sig_w + (size_t)sig * d.ip_P;
    const uint32_t *mw = r.maskw + (size_t)m * (d.NP / 32);
    uint16_t *sc = r.score + (size_t)m * d.NP;
    const uint32_t Z = *d.ip_z;
    std::fill(cnt.begin(), cnt.end(), 0ll);
    for (uint32_t p = 0; p < d.ip_P; p++) {
      const int wp = w[p];
      if (wp == 0) continue;
      const uint32_t *dom = d.ip_cls_dom + (size_t)p * d.NP;
      const int32_t *cb = d.ip_cls_bound + (size_t)p * d.NP, *cu = d.ip_cls_unbound + (size_t)p * d.NP;
      std::fill(hist.begin(), hist.end(), 0ll);
      long long zs = 0;
      for (uint32_t n = 0; n < d.N; n++)
        if (mask_bit(mw, n)) {
          zs += cu[n];
          if (dom[n] != KB_NONE_U32) hist[dom[n]] += cb[n];
        }
      const uint32_t zdom = (Z != KB_NONE_U32) ? dom[Z] : KB_NONE_U32;
      for (uint32_t n = 0; n < d.N; n++)
        if (mask_bit(mw, n) && dom[n] != KB_NONE_U32) cnt[n] += (long long)wp * (hist[dom[n]] + (dom[n] == zdom ? zs : 0ll));
    }
    long long mx = 0, mn = 0;   // interpod_affinity.go:213-220: both start at 0
    for (uint32_t n = 0; n < d.N; n++)
      if (mask_bit(mw, n)) { mx = std::max(mx, cnt[n]); mn = std::min(mn, cnt[n]); }
    if (mx - mn <= 0) continue;
    for (uint32_t n = 0; n < d.N; n++)
      if (mask_bit(mw, n)) {
        const double f = 10.0 * ((double)(cnt[n] - mn) / (double)(mx - mn));
        sc[n] = (uint16_t)(sc[n] + (int)f * d.wPA);
      }
  }
  });
}

void kb_launch_or_ports_x(const KbDev &d, uint32_t task, uint32_t node, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, task, node]() {
    for (uint32_t w = 0; w < d.port_xw; w++) d.ports_x[(size_t)w * d.NP + node] |= d.t_want_x[(size_t)task * d.port_xw + w];
  });
}
void kb_launch_probe(const KbDev &d, const uint32_t *rows, uint32_t n_rows, uint32_t *alive, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, rows, n_rows, alive]() {
  KbDev dd = d;
  dd.score_enabled = 0;
  for (uint32_t i = 0; i < n_rows; i++) {
    const Row t = row_of_task(dd, rows[i]);
    for (uint32_t n = 0; n < d.N; n++)
      if (eval_pair(dd, t, n, 1, true)) { alive[i] |= 1u; break; }
  }
  });
}

void kb_launch_argmax(const KbDev &d, const KbRound &r, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, r]() {
  if (r.n_mrows == 0) return;
  if (KB_CHAIN_BROKEN(r)) return;
  if (r.mrow_task0 == 0 && r.mrows != nullptr) out64(r)[KB_OUT_STAMP0 + 1] = kbemu_wall_clock();
  std::vector<u64> keys;
  for (uint32_t m = 0; m < r.n_mrows; m++) {
    const uint32_t *mw = r.maskw + (size_t)m * (d.NP / 32);
    const uint16_t *sc = r.score + (size_t)m * d.NP;
    keys.clear();
    for (uint32_t n = 0; n < d.NP; n++)
      if (mask_bit(mw, n)) keys.push_back(KB_KEY(sc[n], n));
    std::sort(keys.begin(), keys.end(), [](u64 a, u64 b) { return a > b; });   // score descending, node ascending
    u64 *out = r.keys + (size_t)m * r.L;
    for (uint32_t i = 0; i < r.L; i++) out[i] = i < keys.size() ? keys[i] : 0ull;
    // KB_EMU_DROP_TAG=k: every k-th overlapped round never publishes its lists (what a launch that failed on the second stream would look like):
    // the repair launch must give up after its bounded wait and break the chain, the engine must run that round again on the plain path
    const int drop = getenv("KB_EMU_DROP_TAG") ? atoi(getenv("KB_EMU_DROP_TAG")) : 0;   // read per launch: tests switch it on and off inside one process
    const bool dropped = r.ready != nullptr && drop > 0 && (r.ready_tag % (uint32_t)drop) == 0;
    if (r.ready != nullptr && !dropped) __atomic_store_n(&r.ready[m], r.ready_tag, __ATOMIC_RELEASE);
  }
  });
}

// k_repair's contract: per matrix row, the stale list (r.stale, r.stale_L entries) without the predecessor's nodes (r.prev_dec, r.n_prev
// records), plus those nodes evaluated against the state the predecessor left, best first, r.L entries, 0-padded
static void emu_repair(const KbDev &d, const KbRound &r) {
  if (r.n_mrows == 0) return;
  if (KB_CHAIN_BROKEN(r)) return;
  out64(r)[KB_OUT_STAMP0] = kbemu_wall_clock();
  out64(r)[KB_OUT_STAMP0 + 1] = out64(r)[KB_OUT_STAMP0];
  std::vector<uint32_t> prev;
  for (uint32_t i = 0; i < r.n_prev; i++) {
    const uint32_t n = (uint32_t)(r.prev_dec[i] & 0xFFFFFFFFull);
    if (n != KB_NONE_U32) prev.push_back(n);
  }
  std::sort(prev.begin(), prev.end());
  prev.erase(std::unique(prev.begin(), prev.end()), prev.end());
  std::vector<u64> keys;
  for (uint32_t m = 0; m < r.n_mrows; m++) {
    const double t0 = (double)kbemu_wall_clock();
    while (__atomic_load_n(&r.ready[m], __ATOMIC_ACQUIRE) != r.ready_tag) {   // asynchronous emulated streams: the other worker is still on it
      std::this_thread::yield();
      const double bound = getenv("KB_EMU_REPAIR_WAIT_NS") ? atof(getenv("KB_EMU_REPAIR_WAIT_NS")) : 3.0e9;
      if ((double)kbemu_wall_clock() - t0 > bound) { if (r.chain) *r.chain = 0u; return; }   // the bounded wait of the kernel
    }
    const Row t = row_of_task(d, mrow_task(r, m));
    keys.clear();
    const u64 *st = r.stale + (size_t)m * r.stale_L;
    if (getenv("KB_EMU_REPAIR_OFF")) {   // negative control: the stale list as it is (tests/test_emu_engine_cpu.py expects wrong decisions)
      u64 *o = r.keys + (size_t)m * r.L;
      for (uint32_t i = 0; i < r.L; i++) o[i] = st[i];
      continue;
    }
    for (uint32_t i = 0; i < r.stale_L && st[i] != 0ull; i++)
      if (!std::binary_search(prev.begin(), prev.end(), KB_KEY_NODE(st[i]))) keys.push_back(st[i]);
    for (uint32_t n : prev) {
      const uint32_t e = eval_pair(d, t, n, r.fit_mode, true);
      if (e >> 16) keys.push_back(KB_KEY(e & 0xFFFFu, n));
    }
    std::sort(keys.begin(), keys.end(), [](u64 a, u64 b) { return a > b; });
    u64 *out = r.keys + (size_t)m * r.L;
    for (uint32_t i = 0; i < r.L; i++) out[i] = i < keys.size() ? keys[i] : 0ull;
    if (r.lists_ready != nullptr) __atomic_store_n(&r.lists_ready[m], r.lists_tag, __ATOMIC_RELEASE);   // the tag the commit workgroup of a fused launch waits for
  }
}
size_t kb_repair_smem_bytes(uint32_t NP) { return 8 * (1024 + 512) + 4 * (2 * 1025 + 512) + 4 * 36 + 4 * (size_t)(NP / 32); }   // kb_repair.hpp: kb_repair_lds_bytes
void kb_launch_repair(const KbDev &d, const KbRound &r, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, r]() { emu_repair(d, r); });
}

void kb_launch_expand(const KbDev &d, const uint16_t *s_score, const uint32_t *s_mask, const uint32_t *row_slot, const uint32_t *order, uint32_t n_rows,
                      uint16_t *score, uint32_t *maskw, void *stream, const KbXChunk *chunks, uint32_t n_chunks) {
  if (chunks) {   // the chunk table's contract: the chunks tile `order` (or, without it, the rows) exactly, every row of a chunk has the chunk's shape (the kernel takes the shape from the chunk)
    kbemu_enqueue((hipStream_t)stream, [row_slot, order, n_rows, chunks, n_chunks]() {
      uint32_t at = 0;
      for (uint32_t c = 0; c < n_chunks; c++) {
        if (chunks[c].first != at || chunks[c].count == 0 || chunks[c].count > KB_XCHUNK_ROWS) abort();
        for (uint32_t i = 0; i < chunks[c].count; i++)
          if (row_slot[order ? order[at + i] : at + i] != chunks[c].slot) abort();   // (no `order`: the chunks tile the rows themselves)
        at += chunks[c].count;
      }
      if (at != n_rows) abort();
    });
  }
  kbemu_enqueue((hipStream_t)stream, [d, s_score, s_mask, row_slot, order, n_rows, score, maskw]() {
  for (uint32_t at = 0; at < n_rows; at++) {   // the contract: `order` is a permutation of the rows (the kernel takes them in that order)
    const uint32_t row = order ? order[at] : at;
    const uint32_t slot = row_slot[row];
    std::memcpy(score + (size_t)row * d.NP, s_score + (size_t)slot * d.NP, sizeof(uint16_t) * d.NP);
    std::memcpy(maskw + (size_t)row * (d.NP / 32), s_mask + (size_t)slot * (d.NP / 32), sizeof(uint32_t) * (d.NP / 32));
  }
  });
}

void kb_launch_scatter_nodes(const KbDev &d, const unsigned long long *rec, uint32_t n, uint32_t *nmask, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, rec, n, nmask]() {
  const size_t words = 5 + 2 * (size_t)d.R;
  for (uint32_t i = 0; i < n; i++) {
    const unsigned long long *r = rec + (size_t)i * words;
    const uint32_t node = (uint32_t)r[0];
    nmask[node] = (uint32_t)(r[0] >> 32);
    d.podcnt[node] = (int)(uint32_t)r[1];
    d.nzc[node] = (long long)r[2];
    d.nzm[node] = (long long)r[3];
    if (d.ports) d.ports[node] = r[4];
    for (int dd = 0; dd < d.R; dd++) {
      std::memcpy(&d.idle[(size_t)dd * d.NP + node], &r[5 + dd], 8);
      std::memcpy(&d.rel[(size_t)dd * d.NP + node], &r[5 + d.R + dd], 8);
    }
  }
  });
}

void kb_launch_commit(const KbDev &d, const KbRound &r, void *stream) { kbemu_enqueue((hipStream_t)stream, [d, r]() { emu_commit(d, r); }); }
// the selection kernel (k_commit_run<true>): the run kernel's contract and statistics words
// A round whose launch carries its own repair workgroups (KbRound::lists_ready): the lists are repaired inside the launch, in front of everything
// the commit reads of them; a stale list that never arrives clears the chain word and the commit reports KB_REASON_SKIPPED
void kb_launch_commit_sel(const KbDev &d, const KbRound &r, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, r]() {
    if (r.lists_ready != nullptr) emu_repair(d, r);
    emu_commit(d, r);
  });
}

uint32_t kb_apply_deltas(const KbDev &d, const double *s_idle, const double *s_rel, const long long *s_nzc, const long long *s_nzm,
                         const int *s_podcnt, const double *delta, uint32_t *dev_counter, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [d, s_idle, s_rel, s_nzc, s_nzm, s_podcnt, delta, dev_counter]() {
  uint32_t bad = 0;
  for (uint32_t n = 0; n < d.NP; n++) {
    for (int dim = 0; dim < d.R; dim++) {
      const size_t o = (size_t)dim * d.NP + n;
      const double vi = s_idle[o] + delta[o];
      const double vr = s_rel[o] + delta[(size_t)d.R * d.NP + o];
      bad += (vi != d.idle[o]) + (vr != d.rel[o]);
      d.idle[o] = vi;
      d.rel[o] = vr;
    }
    const double *tail = delta + (size_t)2 * d.R * d.NP;
    const long long c = s_nzc[n] + (long long)tail[n], m = s_nzm[n] + (long long)tail[(size_t)d.NP + n];
    const int p = s_podcnt[n] + (int)tail[(size_t)2 * d.NP + n];
    bad += (c != d.nzc[n]) + (m != d.nzm[n]) + (p != d.podcnt[n]);
    d.nzc[n] = c; d.nzm[n] = m; d.podcnt[n] = p;
  }
  *dev_counter = bad;
  });
  kbemu_drain((hipStream_t)stream);   // the kernel's counter is copied back and the stream synchronised
  return *dev_counter;
}

void kb_check_deltas(const KbDev &d, const KbNodeCopy &s0, const KbNodeCopy &s1, const double *delta, uint32_t *dev_counter, void *stream) {
  const uint32_t NP = d.NP;
  const int R = d.R;
  kbemu_enqueue((hipStream_t)stream, [NP, R, s0, s1, delta, dev_counter]() {
    uint32_t bad = 0;
    for (uint32_t n = 0; n < NP; n++) {
      for (int dim = 0; dim < R; dim++) {
        const size_t o = (size_t)dim * NP + n;
        bad += (s0.idle[o] + delta[o] != s1.idle[o]) + (s0.rel[o] + delta[(size_t)R * NP + o] != s1.rel[o]);
      }
      const double *tail = delta + (size_t)2 * R * NP;
      bad += (s0.nzc[n] + (long long)tail[n] != s1.nzc[n]) + (s0.nzm[n] + (long long)tail[(size_t)NP + n] != s1.nzm[n]) +
             (s0.podcnt[n] + (int)tail[(size_t)2 * NP + n] != s1.podcnt[n]);
    }
    __atomic_fetch_add(dev_counter, bad, __ATOMIC_RELAXED);
  });
}

void kb_launch_finalize(const KbDev &d, const uint32_t *job_task_begin, const int *job_min_avail, const uint32_t *job_queue,
                        int gang_ready_enabled, const double *total, uint32_t total_mask, const double *deserved,
                        const uint32_t *deserved_mask, double *job_alloc, double *job_share, double *queue_alloc,
                        double *queue_share, int *job_ready_cnt, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [=]() {
  std::memset(queue_alloc, 0, sizeof(double) * (size_t)d.Q * d.R);
  for (uint32_t j = 0; j < d.J; j++) {
    const uint32_t t0 = job_task_begin[j], t1 = job_task_begin[j + 1];
    int ready = 0;
    for (uint32_t t = t0; t < t1; t++) {
      const int st = d.t_status[t];
      ready += st == KB_TASK_BOUND || st == KB_TASK_BINDING || st == KB_TASK_RUNNING || st == KB_TASK_ALLOCATED || st == KB_TASK_SUCCEEDED;
    }
    const bool job_ready = gang_ready_enabled ? (ready >= job_min_avail[j]) : true;
    if (job_ready && d.j_allocated[j])
      for (uint32_t t = t0; t < t1; t++)
        if (d.t_status[t] == KB_TASK_ALLOCATED) { d.t_status[t] = KB_TASK_BINDING; d.t_bind[t] = d.t_node[t]; }
    job_ready_cnt[j] = ready;
    d.j_allocated[j] = 0;
    const uint32_t q = job_queue[j];
    double share = 0.0;
    for (int dim = 0; dim < d.R; dim++) {
      double s = 0.0;   // integer-valued addends below 2^53: the sum is exact in any order
      for (uint32_t t = t0; t < t1; t++)
        if (d.t_counted[t]) s += d.t_res[(size_t)dim * d.T + t];
      job_alloc[(size_t)j * d.R + dim] = s;
      if (s != 0.0 && q < d.Q) queue_alloc[(size_t)q * d.R + dim] += s;
      if (dim < 2 || ((total_mask >> (dim - 2)) & 1u)) share = std::max(share, share_of(s, total[dim]));
    }
    job_share[j] = share;
  }
  for (uint32_t q = 0; q < d.Q; q++) {
    double share = 0.0;
    for (int dim = 0; dim < d.R; dim++) {
      if (dim >= 2 && !((deserved_mask[q] >> (dim - 2)) & 1u)) continue;
      share = std::max(share, share_of(queue_alloc[(size_t)q * d.R + dim], deserved[(size_t)dim * d.Q + q]));
    }
    queue_share[q] = share;
  }
  });
}

// k_waterfill's contract: proportion's OnSessionOpen loop over the queue records, in the steps the kernel's lanes run (kb_waterfill.hpp — the
// step functions are shared text; what the kernel adds, the placement of its barriers, is not emulated)
extern "C" unsigned long long kbemu_selected_rows() { return __atomic_load_n(&g_selected_rows, __ATOMIC_RELAXED); }
extern "C" unsigned long long kbemu_select_runs() { return __atomic_load_n(&g_select_runs, __ATOMIC_RELAXED); }
extern "C" unsigned long long kbemu_select_lanes() { return __atomic_load_n(&g_select_lanes, __ATOMIC_RELAXED); }
extern "C" unsigned long long kbemu_select_steps() { return __atomic_load_n(&g_select_steps, __ATOMIC_RELAXED); }
static unsigned long long g_waterfill_launches = 0;
extern "C" unsigned long long kbemu_waterfill_launches() { return __atomic_load_n(&g_waterfill_launches, __ATOMIC_RELAXED); }
void kb_launch_waterfill(kb::WfQueue *qs, uint32_t Q, kb::WfState *st, int R, double *des, uint32_t *desmask, void *stream) {
  kbemu_enqueue((hipStream_t)stream, [qs, Q, st, R, des, desmask]() {
    __atomic_fetch_add(&g_waterfill_launches, 1ull, __ATOMIC_RELAXED);
    kb::wf_run_sequential(qs, Q, *st, R);
    if (des)
      for (uint32_t q = 0; q < Q; q++) {
        desmask[q] = qs[q].deserved.mask;
        for (int d = 0; d < R; d++) des[(size_t)d * Q + q] = qs[q].deserved.get(d);
      }
  });
}
