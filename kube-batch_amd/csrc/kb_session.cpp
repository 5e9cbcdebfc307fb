// kb_session.cpp — the host half of kb_engine_create / kb_session_load: the conf.Tier lists compiled into a Policy, and the snapshot
// turned into the HostSession (validation of the exact envelope, task shapes, plugin OnSessionOpen state).  No device code: this file
// also builds with g++ into the CPU test harnesses (tests/host_harness), which drive the engine's own host logic without a GPU.
//
// This is the engine's own implementation, independent of oracle/kb_oracle.c (test infrastructure).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <unordered_map>

#include "kb_host.hpp"

namespace kb {

Policy compile_policy(const kb_config *cfg) {
  Policy p;
  if (!cfg->tier_begin && cfg->n_tiers) throw EngineError(KB_E_INVALID, "tier_begin is NULL");
  for (uint32_t t = 0; t < cfg->n_tiers; t++) {
    p.preempt_tiers.emplace_back();
    p.reclaim_tiers.emplace_back();
    for (uint32_t i = cfg->tier_begin[t]; i < cfg->tier_begin[t + 1]; i++) {
      const kb_plugin_option &o = cfg->plugins[i];
      // plugins that register a PreemptableFn: conformance.go:60, gang.go:93, priority.go:100, drf.go:111
      if ((o.enabled & KB_EN_PREEMPTABLE) && (o.plugin == KB_PLUGIN_CONFORMANCE || o.plugin == KB_PLUGIN_GANG || o.plugin == KB_PLUGIN_PRIORITY || o.plugin == KB_PLUGIN_DRF))
        p.preempt_tiers.back().push_back((uint8_t)o.plugin);
      // ... a ReclaimableFn: conformance.go:61, gang.go:92, proportion.go:171
      if ((o.enabled & KB_EN_RECLAIMABLE) && (o.plugin == KB_PLUGIN_CONFORMANCE || o.plugin == KB_PLUGIN_GANG || o.plugin == KB_PLUGIN_PROPORTION))
        p.reclaim_tiers.back().push_back((uint8_t)o.plugin);
      switch (o.plugin) {
        case KB_PLUGIN_PRIORITY:
          if (o.enabled & KB_EN_JOB_ORDER) p.job_chain.push_back(KB_PLUGIN_PRIORITY);
          if (o.enabled & KB_EN_TASK_ORDER) p.task_order_priority = true;
          break;
        case KB_PLUGIN_GANG:
          p.has_gang = true;
          if (o.enabled & KB_EN_JOB_ORDER) p.job_chain.push_back(KB_PLUGIN_GANG);
          if (o.enabled & KB_EN_JOB_READY) p.gang_job_ready = true;
          if (o.enabled & KB_EN_JOB_PIPELINED) p.gang_job_pipelined = true;
          break;
        case KB_PLUGIN_CONFORMANCE:   // registers evict filters only (conformance.go:41-63)
          break;
        case KB_PLUGIN_DRF:
          p.has_drf = true;
          if (o.enabled & KB_EN_JOB_ORDER) p.job_chain.push_back(KB_PLUGIN_DRF);
          break;
        case KB_PLUGIN_PREDICATES:
          if (o.enabled & KB_EN_PREDICATE) p.pred_enabled = true;
          if ((o.args_set & 7u) && (o.args[0] || o.args[1] || o.args[2]))
            throw EngineError(KB_E_UNSUPPORTED, "predicates pressure checks must be folded into node classes by the caller; flags not supported");
          break;
        case KB_PLUGIN_PROPORTION:
          p.has_proportion = true;
          if (o.enabled & KB_EN_QUEUE_ORDER) p.queue_order_proportion = true;
          break;
        case KB_PLUGIN_NODEORDER:
          if (o.enabled & KB_EN_NODE_ORDER) p.nodeorder_enabled = true;
          if (o.args_set & 1u) p.wL = o.args[KB_ARG_NODEORDER_LEAST];       // nodeorder.go:119-129
          if (o.args_set & 2u) p.wM = o.args[KB_ARG_NODEORDER_MOST];
          if (o.args_set & 4u) p.wNA = o.args[KB_ARG_NODEORDER_NODEAFF];
          if (o.args_set & 8u) p.wPA = o.args[KB_ARG_NODEORDER_PODAFF];
          if (o.args_set & 16u) p.wB = o.args[KB_ARG_NODEORDER_BALANCED];
          break;
        default:
          throw EngineError(KB_E_UNSUPPORTED, "unknown plugin id " + std::to_string(o.plugin));
      }
    }
  }
  // every scorer yields 0..10 (NodeAffinity after its NormalizeReduce too) and the matrix stores the weighted sum as u16
  if (p.wL < 0 || p.wM < 0 || p.wB < 0 || p.wNA < 0 || 10ll * ((long long)p.wL + p.wM + p.wB + p.wNA) > 65535)
    throw EngineError(KB_E_UNSUPPORTED, "nodeorder weights must be >= 0 with 10*(least+most+balanced+nodeaffinity) <= 65535 (u16 score)");
  return p;
}

namespace {
// shape interning: fixed-length double keys -> dense ids in first-appearance order (hashed: one lookup per task at session load)
struct Interner {
  size_t klen = 0;
  std::vector<double> keys;
  std::unordered_map<uint64_t, std::vector<uint32_t>> buckets;
  Interner() = default;
  explicit Interner(size_t len) : klen(len) {}
  uint32_t intern(const double *k) {
    uint64_t h = 0x9E3779B97F4A7C15ull;   // word-wise multiply-xorshift over the key's bit patterns
    for (size_t i = 0; i < klen; i++) {
      uint64_t w;
      std::memcpy(&w, &k[i], sizeof(w));
      h = (h ^ w) * 0xFF51AFD7ED558CCDull;
      h ^= h >> 32;
    }
    std::vector<uint32_t> &ids = buckets[h];
    for (uint32_t id : ids)
      if (std::memcmp(&keys[(size_t)id * klen], k, klen * sizeof(double)) == 0) return id;
    const uint32_t id = (uint32_t)(keys.size() / klen);
    keys.insert(keys.end(), k, k + klen);
    ids.push_back(id);
    return id;
  }
  uint32_t intern(const std::vector<double> &k) {
    if (!klen) klen = k.size();
    return intern(k.data());
  }
  const double *key(uint32_t id) const { return &keys[(size_t)id * klen]; }
  size_t size() const { return klen ? keys.size() / klen : 0; }
};
}  // namespace

// kb_session_load, host part.  t_active: per task the dimensions LessEqual compares (bits 0, 1 always; a scalar bit when InitResreq
// exceeds the epsilon); nmask: per node (padded to NP) the scalar keys of Idle, bit 31 <=> Releasing carries scalar keys.
// The Go action loads a session every scheduling cycle: at 1M tasks the O(T) passes below (copies, the task-major transpose, the
// "same as its predecessor" test that lets nearly every task skip shape interning) are memory traffic worth splitting over a few host
// threads.  fn(t0, t1) over disjoint ranges; small sessions stay on the calling thread.
// `weight`: 8-byte words a pass touches per task (a pass over 16 resource dimensions of 100k tasks is as much memory as one over 2 dimensions of 800k)
static uint32_t par_threads(uint32_t T, uint32_t weight) {
  const unsigned hw = std::thread::hardware_concurrency();
  // below ~2M words thread start-up and the remote cache lines it leaves behind cost more than the split saves (100k tasks x 2 dimensions on eight
  // threads: 3.7 -> 10 ms on the GPU box, round 4; 100k x 16 dimensions on one: 1.7 + 1.8 ms of the 11 ms load, round 5's first call)
  // KB_HOST_SPLIT_WORDS: the threshold, for the tests that hold the split passes to the one-thread passes on small sessions
  uint64_t min_words = 1ull << 21;
  if (const char *v = getenv("KB_HOST_SPLIT_WORDS")) min_words = strtoull(v, nullptr, 10);
  return (uint64_t)T * weight < min_words ? 1u : std::min<uint32_t>(8u, hw ? hw : 1u);
}
// fn(i, t0, t1): part i of nt, the same ranges for the same (T, nt)
template <typename F> static void par_parts(uint32_t T, uint32_t nt, F fn) {
  if (nt <= 1) { fn(0u, 0u, T); return; }
  std::vector<std::thread> th;
  const uint32_t step = (T + nt - 1) / nt;
  for (uint32_t i = 1; i < nt; i++) th.emplace_back([&, i] { fn(i, std::min(T, i * step), std::min(T, (i + 1) * step)); });
  fn(0u, 0u, std::min(T, step));
  for (auto &t : th) t.join();
}
template <typename F> static void par_for(uint32_t T, uint32_t weight, F fn) {
  par_parts(T, par_threads(T, weight), [&](uint32_t, uint32_t a, uint32_t b) { fn(a, b); });
}
template <typename V, typename S> static void par_copy(V &dst, const S *src, size_t n) {   // dst := src[0..n), the destination's storage reused
  dst.resize(n);
  par_for((uint32_t)n, 8u, [&](uint32_t a, uint32_t b) { std::copy(src + a, src + b, dst.begin() + a); });
}

void build_host_session(const kb_snapshot *sn, const Policy &pol, uint32_t NP, HostSession &hs, std::vector<uint32_t> &t_active, std::vector<uint32_t> &nmask) {
  const bool waterfill_on_device = hs.waterfill_on_device;   // the caller's choice, made before this call
  static const bool trace = [] { const char *v = getenv("KB_LOAD_TRACE"); return v && v[0] == '1'; }();
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto mark = [&](const char *what) { if (trace) { const double t = now(); fprintf(stderr, "  build_host_session: %-40s %8.3f ms\n", what, t - t_mark); t_mark = t; } };
  {   // a fresh session, but the T-sized arrays keep their storage from the last load (same cluster, next cycle: no allocation, no page faults)
    HostSession old = std::move(hs);
    hs = HostSession();
    hs.t_res = std::move(old.t_res); hs.t_init = std::move(old.t_init); hs.t_res_rows = std::move(old.t_res_rows); hs.t_resmask = std::move(old.t_resmask);
    hs.t_job = std::move(old.t_job); hs.t_cls = std::move(old.t_cls); hs.t_prio = std::move(old.t_prio); hs.t_creation = std::move(old.t_creation);
    hs.t_status = std::move(old.t_status); hs.t_node = std::move(old.t_node); hs.t_nzc = std::move(old.t_nzc); hs.t_nzm = std::move(old.t_nzm);
    hs.t_res_empty = std::move(old.t_res_empty); hs.t_init_empty = std::move(old.t_init_empty);
    hs.t_feas_shape = std::move(old.t_feas_shape); hs.t_row_shape = std::move(old.t_row_shape);
  }
  hs.waterfill_on_device = waterfill_on_device;
  const int R = hs.R = (int)sn->n_res;
  const uint32_t N = hs.N = sn->n_nodes, T = hs.T = sn->n_tasks, J = hs.J = sn->n_jobs, Q = hs.Q = sn->n_queues;
  // ---- validation of the exact envelope ----
  for (uint32_t n = 0; n < N; n++) {
    if (sn->node_alloc_cpu[n] < 0 || sn->node_alloc_mem[n] < 0 || sn->node_nz_cpu[n] < 0 || sn->node_nz_mem[n] < 0)
      throw EngineError(KB_E_INVALID, "negative node quantity");
    if (sn->node_alloc_cpu[n] >= (1ll << 48) || sn->node_alloc_mem[n] >= (1ll << 48) || sn->node_nz_cpu[n] >= (1ll << 48) || sn->node_nz_mem[n] >= (1ll << 48))
      throw EngineError(KB_E_UNSUPPORTED, "node quantity >= 2^48: exact integer scoring not guaranteed");
  }
  hs.t_res.resize((size_t)R * T); hs.t_init.resize((size_t)R * T); hs.t_resmask.resize(T); hs.t_res_rows.resize((size_t)T * R);
  // whole numbers below 2^47 (KbDev::whole): k8s quantities in milli-units / bytes always are; anything else only costs the selection kernel its
  // shots (the rows of a run then go one by one: one Sub per placement, in order, whatever the values)
  std::atomic<int> fractional{0};
  auto whole_run = [](const double *p, size_t n) {
    // |x| < 2^47 and (|x| + 2^52) - 2^52 == |x|: below 2^52 the sum rounds to a whole number (round-to-nearest), so the difference gives |x| back iff
    // it was one — no libm call (std::floor is one without SSE4.1), straight-line, vectorisable (the build never contracts or reassociates:
    // -ffp-contract=off -fno-fast-math); NaN and infinities fail the first compare
    int bad = 0;
    for (size_t i = 0; i < n; i++) {
      const double a = std::fabs(p[i]);
      const double big = 4503599627370496.0;   // 2^52 (never folded: the build does not reassociate)
      const double r = (a + big) - big;
      bad |= !(a < 140737488355328.0) | (r != a);
    }
    return bad == 0;
  };
  for (int d = 0; d < R; d++)
    if (!whole_run(sn->node_idle + (size_t)d * N, N) || !whole_run(sn->node_releasing + (size_t)d * N, N)) fractional.store(1, std::memory_order_relaxed);
  par_for(T, 4u * (uint32_t)R, [&](uint32_t t0, uint32_t t1) {
    for (uint32_t t = t0; t < t1; t++) hs.t_resmask[t] = sn->task_scalar_mask ? sn->task_scalar_mask[t] : 0u;
    for (int d = 0; d < R; d++) {   // one dimension's row at a time: sequential in the dimension-major layout
      std::copy(sn->task_resreq + (size_t)d * T + t0, sn->task_resreq + (size_t)d * T + t1, hs.t_res.begin() + (size_t)d * T + t0);
      std::copy(sn->task_init_resreq + (size_t)d * T + t0, sn->task_init_resreq + (size_t)d * T + t1, hs.t_init.begin() + (size_t)d * T + t0);
      if (!whole_run(sn->task_resreq + (size_t)d * T + t0, t1 - t0) || !whole_run(sn->task_init_resreq + (size_t)d * T + t0, t1 - t0)) fractional.store(1, std::memory_order_relaxed);
      if (d >= 2) {   // a dense value under an absent key reads 0 (Go map semantics)
        double *row = &hs.t_res[(size_t)d * T];
        for (uint32_t t = t0; t < t1; t++)
          if (!((hs.t_resmask[t] >> (d - 2)) & 1u)) row[t] = 0.0;
      }
    }
    // task-major copy for the order machine: one task's Resreq is read per scheduling step, and with the dimension-major
    // device layout that is R cache misses per step (written sequentially here, read from R streams)
    for (uint32_t t = t0; t < t1; t++)
      for (int d = 0; d < R; d++) hs.t_res_rows[(size_t)t * R + d] = hs.t_res[(size_t)d * T + t];
  });
  hs.whole = fractional.load() == 0;
  mark("request vectors (copy, absent keys, task-major copy)");
  hs.t_job.resize(T); hs.t_cls.resize(T); hs.t_prio.resize(T); hs.t_creation.resize(T); hs.t_status.resize(T); hs.t_node.resize(T); hs.t_nzc.resize(T); hs.t_nzm.resize(T);
  {
    std::atomic<int> bad_node{0};
    par_for(T, 8u, [&](uint32_t t0, uint32_t t1) {
      std::copy(sn->task_job + t0, sn->task_job + t1, hs.t_job.begin() + t0);
      if (sn->task_class) std::copy(sn->task_class + t0, sn->task_class + t1, hs.t_cls.begin() + t0); else std::fill(hs.t_cls.begin() + t0, hs.t_cls.begin() + t1, 0);
      std::copy(sn->task_priority + t0, sn->task_priority + t1, hs.t_prio.begin() + t0);
      std::copy(sn->task_creation + t0, sn->task_creation + t1, hs.t_creation.begin() + t0);
      std::copy(sn->task_status + t0, sn->task_status + t1, hs.t_status.begin() + t0);
      if (sn->task_node) std::copy(sn->task_node + t0, sn->task_node + t1, hs.t_node.begin() + t0); else std::fill(hs.t_node.begin() + t0, hs.t_node.begin() + t1, KB_NONE);
      std::copy(sn->task_nz_cpu + t0, sn->task_nz_cpu + t1, hs.t_nzc.begin() + t0);
      std::copy(sn->task_nz_mem + t0, sn->task_nz_mem + t1, hs.t_nzm.begin() + t0);
      for (uint32_t t = t0; t < t1; t++)
        if (hs.t_node[t] != KB_NONE && hs.t_node[t] >= N) bad_node.store(1, std::memory_order_relaxed);
    });
    if (bad_node.load()) throw EngineError(KB_E_INVALID, "task_node out of range");
  }
  hs.job_begin.assign(sn->job_task_begin, sn->job_task_begin + J + 1);
  hs.job_queue.assign(sn->job_queue, sn->job_queue + J);
  hs.job_min.assign(sn->job_min_available, sn->job_min_available + J);
  hs.job_prio.assign(sn->job_priority, sn->job_priority + J);
  hs.job_creation.assign(sn->job_creation, sn->job_creation + J);
  hs.queue_weight.assign(sn->queue_weight, sn->queue_weight + Q);
  hs.queue_creation.assign(Q, 0);
  if (sn->queue_creation) hs.queue_creation.assign(sn->queue_creation, sn->queue_creation + Q);
  if (hs.job_begin[0] != 0 || hs.job_begin[J] != T) throw EngineError(KB_E_INVALID, "job_task_begin must cover [0, n_tasks)");
  for (uint32_t j = 0; j < J; j++) {
    if (hs.job_begin[j] > hs.job_begin[j + 1]) throw EngineError(KB_E_INVALID, "job_task_begin not monotone");
    for (uint32_t t = hs.job_begin[j]; t < hs.job_begin[j + 1]; t++)
      if (hs.t_job[t] != j) throw EngineError(KB_E_INVALID, "tasks must be grouped by job in canonical order");
  }
  const uint32_t Wh = sn->port_words ? sn->port_words : 1;   // 64-bit words per host-port mask
  hs.port_xw = 0; hs.t_want_x.clear(); hs.t_conf_x.clear(); hs.t_wide.clear();
  if (sn->task_port_want || sn->task_port_conflict) {
    hs.t_want.assign(T, 0); hs.t_conf.assign(T, 0);
    for (uint32_t t = 0; t < T; t++) { if (sn->task_port_want) hs.t_want[t] = sn->task_port_want[(size_t)t * Wh]; if (sn->task_port_conflict) hs.t_conf[t] = sn->task_port_conflict[(size_t)t * Wh]; }
    if (Wh > 1) {
      const uint32_t X = Wh - 1;
      hs.port_xw = X;
      hs.t_want_x.assign((size_t)T * X, 0); hs.t_conf_x.assign((size_t)T * X, 0); hs.t_wide.assign(T, 0);
      for (uint32_t t = 0; t < T; t++)
        for (uint32_t w = 0; w < X; w++) {
          const uint64_t wt = sn->task_port_want ? sn->task_port_want[(size_t)t * Wh + 1 + w] : 0, cf = sn->task_port_conflict ? sn->task_port_conflict[(size_t)t * Wh + 1 + w] : 0;
          hs.t_want_x[(size_t)t * X + w] = wt; hs.t_conf_x[(size_t)t * X + w] = cf;
          if (wt & ~cf) throw EngineError(KB_E_INVALID, "a pod's host ports must conflict with themselves (want is not a subset of conflict)");
          if (wt | cf) hs.t_wide[t] = 1;
        }
      bool any_wide = false;
      for (uint32_t t = 0; t < T && !any_wide; t++) any_wide = hs.t_wide[t] != 0;
      if (!any_wide) { hs.port_xw = 0; hs.t_want_x.clear(); hs.t_conf_x.clear(); hs.t_wide.clear(); }   // what only nodes carry there never meets a pod's mask
    }
  }
  if (sn->task_evict_protected) hs.t_protected.assign(sn->task_evict_protected, sn->task_evict_protected + T);
  hs.n_ac.assign(sn->node_alloc_cpu, sn->node_alloc_cpu + N);
  hs.n_am.assign(sn->node_alloc_mem, sn->node_alloc_mem + N);
  hs.n_maxpods.assign(sn->node_max_pods, sn->node_max_pods + N);
  hs.n_cls.assign(N, 0);
  if (sn->node_class) hs.n_cls.assign(sn->node_class, sn->node_class + N);
  hs.n_idle_mask.assign(N, 0);
  if (sn->node_scalar_mask) hs.n_idle_mask.assign(sn->node_scalar_mask, sn->node_scalar_mask + N);
  hs.n_tc = sn->n_task_classes; hs.n_nc = sn->n_node_classes ? sn->n_node_classes : 1;
  if (sn->class_compat) hs.compat.assign(sn->class_compat, sn->class_compat + ((size_t)sn->n_task_classes * sn->n_node_classes + 7) / 8);
  mark("task / job / node arrays, validation");
  t_active.assign(T, 3u);
  hs.t_res_empty.resize(T);
  hs.t_init_empty.resize(T);
  // inter-pod (anti)affinity tables: validate what indexes device memory
  const kb_interpod *ip = sn->interpod;
  if (ip) {
    if (ip->n_counters > KB_INTERPOD_MAX || ip->n_classes > KB_INTERPOD_MAX) throw EngineError(KB_E_UNSUPPORTED, "more than KB_INTERPOD_MAX inter-pod counters / classes");
    const uint32_t Wc = ip->n_counters ? (ip->n_counters + 63) / 64 : 1, Wp = ip->n_classes ? (ip->n_classes + 63) / 64 : 1;
    if (ip->n_domains == 0 || ip->n_domains > std::max<uint32_t>(N, 1u)) throw EngineError(KB_E_INVALID, "inter-pod: n_domains outside 1..N");
    if (!ip->ctr_dom || !ip->ctr_count || !ip->ctr_total || !ip->task_inc || !ip->task_forbid || !ip->task_require || !ip->task_self ||
        !ip->cls_dom || !ip->cls_bound || !ip->cls_unbound || !ip->task_cls_inc || !ip->task_sig || !ip->sig_weight)
      throw EngineError(KB_E_INVALID, "inter-pod: missing table");
    if (ip->first_unbound_node != KB_NONE && ip->first_unbound_node >= N) throw EngineError(KB_E_INVALID, "inter-pod: first_unbound_node out of range");
    auto beyond = [](const uint64_t *row, uint32_t W, uint32_t nbits) {   // a mask bit at or beyond nbits
      for (uint32_t w = 0; w < W; w++) {
        const uint32_t lo = 64 * w;
        const uint64_t valid = nbits <= lo ? 0ull : (nbits - lo >= 64 ? ~0ull : ((1ull << (nbits - lo)) - 1ull));
        if (row[w] & ~valid) return true;
      }
      return false;
    };
    for (uint32_t c = 0; c < ip->n_counters; c++)
      for (uint32_t n = 0; n < N; n++) {
        const uint32_t dm = ip->ctr_dom[(size_t)c * N + n];
        if (dm != KB_NONE && dm >= ip->n_domains) throw EngineError(KB_E_INVALID, "inter-pod: counter domain id out of range");
      }
    for (uint32_t pc = 0; pc < ip->n_classes; pc++)
      for (uint32_t n = 0; n < N; n++) {
        const uint32_t dm = ip->cls_dom[(size_t)pc * N + n];
        if (dm != KB_NONE && dm >= N) throw EngineError(KB_E_INVALID, "inter-pod: class domain id out of range");
        if (ip->cls_bound[(size_t)pc * N + n] < 0 || ip->cls_unbound[(size_t)pc * N + n] < 0) throw EngineError(KB_E_INVALID, "inter-pod: negative pod count");
      }
    long long wsum = 0;
    for (uint32_t i = 0; i < ip->n_sigs * ip->n_classes; i++) wsum = std::max<long long>(wsum, std::llabs((long long)ip->sig_weight[i]));
    if (wsum > 1000000) throw EngineError(KB_E_UNSUPPORTED, "inter-pod: term weight beyond 1e6");
    for (uint32_t t = 0; t < T; t++) {
      if (beyond(ip->task_inc + (size_t)t * Wc, Wc, ip->n_counters) || beyond(ip->task_forbid + (size_t)t * Wc, Wc, ip->n_counters) ||
          beyond(ip->task_cls_inc + (size_t)t * Wp, Wp, ip->n_classes))
        throw EngineError(KB_E_INVALID, "inter-pod: mask names a missing counter / class");
      if (ip->task_require[t] != 0xFFFF && ip->task_require[t] >= ip->n_counters) throw EngineError(KB_E_INVALID, "inter-pod: task_require out of range");
      if (ip->task_sig[t] != KB_NONE && ip->task_sig[t] >= ip->n_sigs) throw EngineError(KB_E_INVALID, "inter-pod: task_sig out of range");
    }
    if (pol.nodeorder_enabled && (pol.wPA < 0 || 10ll * ((long long)pol.wL + pol.wM + pol.wB + pol.wNA + pol.wPA) > 65535))
      throw EngineError(KB_E_UNSUPPORTED, "nodeorder weights (with podaffinity.weight) exceed the 16-bit score range");
  }
  hs.t_feas_shape.resize(T);
  hs.t_row_shape.resize(T);
  // The tasks of a job are adjacent and nearly always identical in everything a shape depends on.  A task whose inputs equal its
  // predecessor's bit for bit (what the interner compares) takes over the predecessor's derived values; one whose key equals the
  // predecessor's takes its ids without a hash lookup.  Either way the ids are the ones a lookup would return.
  const uint32_t ipWc = ip ? (ip->n_counters ? (ip->n_counters + 63) / 64 : 1) : 0;
  auto same_bits = [](double a, double b) { return std::memcmp(&a, &b, sizeof(double)) == 0; };
  auto same_inputs_as_prev = [&](uint32_t t) {
    const uint32_t p = t - 1;
    for (int d = 0; d < R; d++)
      if (!same_bits(hs.t_res[(size_t)d * T + t], hs.t_res[(size_t)d * T + p]) || !same_bits(hs.t_init[(size_t)d * T + t], hs.t_init[(size_t)d * T + p])) return false;
    if (hs.t_resmask[t] != hs.t_resmask[p] || hs.t_cls[t] != hs.t_cls[p]) return false;
    if (sn->task_nz_cpu[t] != sn->task_nz_cpu[p] || sn->task_nz_mem[t] != sn->task_nz_mem[p]) return false;
    if (sn->task_port_conflict && std::memcmp(sn->task_port_conflict + (size_t)t * Wh, sn->task_port_conflict + (size_t)p * Wh, sizeof(uint64_t) * Wh) != 0) return false;
    if (sn->task_port_want && std::memcmp(sn->task_port_want + (size_t)t * Wh, sn->task_port_want + (size_t)p * Wh, sizeof(uint64_t) * Wh) != 0) return false;
    if (ip) {
      if (std::memcmp(ip->task_forbid + (size_t)t * ipWc, ip->task_forbid + (size_t)p * ipWc, sizeof(uint64_t) * ipWc) != 0) return false;
      if (ip->task_require[t] != ip->task_require[p] || ip->task_self[t] != ip->task_self[p] || ip->task_sig[t] != ip->task_sig[p]) return false;
    }
    return true;
  };
  // One pass over the tasks, split over the host threads: which tasks equal their predecessor bit for bit in everything a shape depends on (nearly
  // all: the tasks of a job are adjacent and alike), and for the others — the stretch heads, a few per job — validation, the derived flags and the
  // shape keys while their lines are in the cache.  ids are handed out in first-appearance order over the whole session, which reads sequential —
  // but only the ORDER of distinct keys is: every part of the task range interns its own heads (part-local ids in part-local first-appearance
  // order), the parts' distinct keys are then interned in part order (a key new to part i and absent from the parts before it meets the session's
  // table exactly when one pass over all tasks would have met it), and the fill pass below maps local ids to session ids.
  // One million tasks in 100k jobs: 8.5 ms of interning on one thread behind 2 ms of flags were the largest piece of a 21 ms load.
  std::vector<uint8_t> same_prev(T, 0);
  std::atomic<int> bad_status{0};
  const size_t feas_len = (size_t)R + 3 + 2 + 2 * (size_t)hs.port_xw + (ip ? 2 * (size_t)ipWc + 2 : 0);
  const size_t klen = feas_len + 2 + 2 * (size_t)hs.port_xw + 2 + (ip ? 1 : 0);
  struct HeadPart {
    std::vector<uint32_t> heads, lf, lr;   // the part's heads and their part-local feasibility / row shape ids
    Interner feas, row;
    std::vector<uint32_t> gf, gr;          // part-local id -> session id
    int err = 0;                           // the part's first refusal (its lowest task)
    const char *msg = nullptr;
  };
  const uint32_t nt_heads = par_threads(T, 2u * (uint32_t)R + 4u);
  std::vector<HeadPart> parts(nt_heads);
  par_parts(T, nt_heads, [&](uint32_t pi, uint32_t t0, uint32_t t1) {
    HeadPart &P = parts[pi];
    P.feas = Interner(feas_len); P.row = Interner(klen);
    std::vector<double> kbuf(klen), pbuf(klen);
    double *key = kbuf.data(), *prev = pbuf.data();
    bool prev_valid = false;   // `prev` holds the key of the part's last head
    uint32_t prev_f = 0, prev_r = 0;
    for (uint32_t t = t0; t < t1; t++) {
      if (hs.t_status[t] > KB_TASK_UNKNOWN) bad_status.store(1, std::memory_order_relaxed);
      same_prev[t] = (t > 0 && same_inputs_as_prev(t)) ? 1 : 0;
      if (same_prev[t] || P.err) continue;   // its head passed every check below with these very values (filled in behind this pass); a part reports its first refusal
      if (sn->task_nz_cpu[t] < 0 || sn->task_nz_mem[t] < 0 || sn->task_nz_cpu[t] >= (1ll << 48) || sn->task_nz_mem[t] >= (1ll << 48)) {
        P.err = KB_E_UNSUPPORTED; P.msg = "task non-zero request out of the exact range"; continue;
      }
      Res rq, in;
      rq.mask = hs.t_resmask[t];
      uint32_t active = 3u;
      for (int d = 0; d < R; d++) {
        rq.v[d] = hs.t_res[(size_t)d * T + t];
        in.v[d] = hs.t_init[(size_t)d * T + t];
        if (rq.v[d] < 0 || in.v[d] < 0) { P.err = KB_E_INVALID; P.msg = "negative request"; break; }
        // api/pod_info.go:53-62: InitResreq = max(sum of containers, every init container) >= Resreq per dimension;
        // without it ssn.Allocate's AddTask could fail after the status flip (session.go:243 vs :255)
        if (in.v[d] < rq.v[d]) { P.err = KB_E_UNSUPPORTED; P.msg = "InitResreq < Resreq"; break; }
        if (d >= 2 && in.v[d] != 0.0) in.setk(d);
        if (d >= 2 && in.v[d] > kMinMilliScalar) active |= 1u << d;
      }
      if (P.err) continue;
      t_active[t] = active;
      in.mask |= rq.mask;
      hs.t_res_empty[t] = res_is_empty(rq, R);
      hs.t_init_empty[t] = res_is_empty(in, R);
      double *k = key;
      auto put64 = [&k](uint64_t w) { *k++ = (double)(uint32_t)(w & 0xFFFFFFFFu); *k++ = (double)(uint32_t)(w >> 32); };
      for (int d = 0; d < R; d++) *k++ = in.v[d];
      // a BestEffort task is placed by backfill, whose only resource test is AddTask's Resreq.LessEqual(Idle)
      // (api/node_info.go:161-167): its fit vector is Resreq cpu / memory (non-zero below the epsilon at most), see t_fit below
      *k++ = hs.t_init_empty[t] ? rq.v[0] : -1.0;
      *k++ = hs.t_init_empty[t] ? rq.v[1] : -1.0;
      *k++ = (double)hs.t_cls[t];
      // host ports: the conflict mask is part of feasibility, the wanted bits of what a commit changes
      put64(sn->task_port_conflict ? sn->task_port_conflict[(size_t)t * Wh] : 0);
      for (uint32_t w = 0; w < hs.port_xw; w++) put64(hs.t_conf_x[(size_t)t * hs.port_xw + w]);   // the words behind the first
      if (ip) {   // inter-pod predicate checks are part of feasibility
        for (uint32_t w = 0; w < ipWc; w++) put64(ip->task_forbid[(size_t)t * ipWc + w]);
        *k++ = (double)ip->task_require[t];
        *k++ = (double)(ip->task_require[t] != 0xFFFF ? ip->task_self[t] : 0);
      }
      put64(sn->task_port_want ? sn->task_port_want[(size_t)t * Wh] : 0);
      for (uint32_t w = 0; w < hs.port_xw; w++) put64(hs.t_want_x[(size_t)t * hs.port_xw + w]);
      *k++ = (double)sn->task_nz_cpu[t];
      *k++ = (double)sn->task_nz_mem[t];
      if (ip) *k++ = (double)ip->task_sig[t];   // ... and the priority weights of the score row
      // a head whose key equals the last head's takes its ids without a hash lookup
      const bool same_feas = prev_valid && std::memcmp(prev, key, feas_len * sizeof(double)) == 0;
      const uint32_t f = same_feas ? prev_f : P.feas.intern(key);
      const bool same_row = same_feas && std::memcmp(prev + feas_len, key + feas_len, (klen - feas_len) * sizeof(double)) == 0;
      const uint32_t r = same_row ? prev_r : P.row.intern(key);
      P.heads.push_back(t); P.lf.push_back(f); P.lr.push_back(r);
      std::swap(key, prev);
      prev_valid = true; prev_f = f; prev_r = r;
    }
  });
  if (bad_status.load()) throw EngineError(KB_E_INVALID, "bad task status");
  for (const HeadPart &P : parts)   // parts in task order, each stopped at its first refusal: the lowest task's, as one pass over all tasks reports it
    if (P.err) throw EngineError(P.err, P.msg);
  Interner feas_ids(feas_len), row_ids(klen);
  for (HeadPart &P : parts) {
    P.gf.resize(P.feas.size()); P.gr.resize(P.row.size());
    for (uint32_t i = 0; i < P.gf.size(); i++) P.gf[i] = feas_ids.intern(P.feas.key(i));
    for (uint32_t i = 0; i < P.gr.size(); i++) P.gr[i] = row_ids.intern(P.row.key(i));
  }
  mark("shapes: equal-to-predecessor flags, stretch heads (validation, interning)");
  // session ids for the heads, and the tasks that equal their predecessor take the values of the head of their stretch: the same parts
  par_parts(T, nt_heads, [&](uint32_t pi, uint32_t t0, uint32_t t1) {
    const HeadPart &P = parts[pi];
    for (size_t i = 0; i < P.heads.size(); i++) { hs.t_feas_shape[P.heads[i]] = P.gf[P.lf[i]]; hs.t_row_shape[P.heads[i]] = P.gr[P.lr[i]]; }
    if (t0 >= t1) return;
    uint32_t h = t0, hf = 0, hr = 0;
    if (same_prev[t0]) {   // my first tasks continue a stretch whose head lies in a part before mine: its ids through that part's maps (its owner may not have stored them yet)
      uint32_t q = pi;
      while (q > 0 && parts[q - 1].heads.empty()) q--;
      const HeadPart &B = parts[q - 1];   // task 0 is a head: some part before mine holds one
      h = B.heads.back(); hf = B.gf[B.lf.back()]; hr = B.gr[B.lr.back()];
    }
    for (uint32_t t = t0; t < t1; t++) {
      if (!same_prev[t]) { h = t; hf = hs.t_feas_shape[t]; hr = hs.t_row_shape[t]; continue; }
      t_active[t] = t_active[h];
      hs.t_res_empty[t] = hs.t_res_empty[h];
      hs.t_init_empty[t] = hs.t_init_empty[h];
      hs.t_feas_shape[t] = hf;
      hs.t_row_shape[t] = hr;
    }
  });
  mark("shapes: stretches filled in");
  hs.n_feas_shapes = (uint32_t)feas_ids.size();
  hs.n_row_shapes = (uint32_t)row_ids.size();
  hs.init_empty_tasks.clear();   // backfill's candidates by request (backfill.go:47), ascending: the action filters them by status
  for (uint32_t t = 0; t < T; t++)
    if (hs.t_init_empty[t]) hs.init_empty_tasks.push_back(t);
  // per feasibility shape: the vector LessEqual actually compares (scalar dimensions at or below the epsilon are skipped,
  // resource_info.go:283-287) and the static class, for the dominance rule of ActionRun::mark_dead
  hs.feas_eff.assign((size_t)hs.n_feas_shapes * R, 0.0);
  hs.feas_cls.assign(hs.n_feas_shapes, 0);
  hs.feas_conf.assign(hs.n_feas_shapes, 0);
  hs.feas_rep.assign(hs.n_feas_shapes, 0);
  for (uint32_t t = T; t-- > 0;) hs.feas_rep[hs.t_feas_shape[t]] = t;
  hs.has_interpod = ip != nullptr;
  hs.t_ip_subject.clear(); hs.feas_ip_require.clear(); hs.feas_ip.clear();
  hs.ip_task_inc.clear(); hs.ip_task_forbid.clear(); hs.ip_task_cls_inc.clear(); hs.ip_task_require.clear(); hs.ip_task_self.clear(); hs.ip_ctr_dom.clear();
  hs.ip_C = 0; hs.ip_D = 1; hs.ip_P = 0; hs.ip_Wc = 1; hs.ip_Wp = 1;
  if (ip) {
    hs.ip_C = ip->n_counters; hs.ip_D = ip->n_domains ? ip->n_domains : 1; hs.ip_P = ip->n_classes;
    hs.ip_Wc = hs.ip_C ? (hs.ip_C + 63) / 64 : 1; hs.ip_Wp = hs.ip_P ? (hs.ip_P + 63) / 64 : 1;
    hs.ip_task_inc.assign(ip->task_inc, ip->task_inc + (size_t)T * hs.ip_Wc);
    hs.ip_task_forbid.assign(ip->task_forbid, ip->task_forbid + (size_t)T * hs.ip_Wc);
    hs.ip_task_cls_inc.assign(ip->task_cls_inc, ip->task_cls_inc + (size_t)T * hs.ip_Wp);
    hs.ip_task_require.assign(ip->task_require, ip->task_require + T);
    hs.ip_task_self.assign(ip->task_self, ip->task_self + T);
    hs.ip_ctr_dom.assign(ip->ctr_dom, ip->ctr_dom + (size_t)hs.ip_C * N);
    hs.t_ip_subject.assign(T, 0);
    hs.feas_ip_require.assign(hs.n_feas_shapes, 0);
    hs.feas_ip.assign(hs.n_feas_shapes, 0);
    Interner ipk;
    const uint32_t Wc = ip->n_counters ? (ip->n_counters + 63) / 64 : 1;
    std::vector<double> k3(2 * Wc + 2);
    hs.t_ip_checks.assign(T, 0);
    for (uint32_t t = 0; t < T; t++) {
      bool checks = ip->task_require[t] != 0xFFFF;
      for (uint32_t w = 0; w < Wc; w++) {
        const uint64_t fb = ip->task_forbid[(size_t)t * Wc + w];
        checks = checks || fb != 0;
        k3[2 * w] = (double)(uint32_t)(fb & 0xFFFFFFFFu); k3[2 * w + 1] = (double)(uint32_t)(fb >> 32);
      }
      hs.t_ip_checks[t] = checks ? 1 : 0;
      hs.t_ip_subject[t] = (checks || ip->task_sig[t] != KB_NONE) ? 1 : 0;
      const uint32_t f = hs.t_feas_shape[t];
      hs.feas_ip_require[f] = ip->task_require[t] != 0xFFFF ? 1 : 0;
      k3[2 * Wc] = (double)ip->task_require[t]; k3[2 * Wc + 1] = (double)(ip->task_require[t] != 0xFFFF ? ip->task_self[t] : 0);
      hs.feas_ip[f] = ipk.intern(k3);
    }
  }
  for (uint32_t f = 0; f < hs.n_feas_shapes; f++) {   // every task of a shape carries the same values (they are its key)
    const uint32_t t = hs.feas_rep[f];
    hs.feas_cls[f] = hs.t_cls[t];
    hs.feas_conf[f] = hs.t_conf.empty() ? 0 : hs.t_conf[t];   // word 0; the words behind it are compared through the shape's task (ActionRun::mark_dead)
    for (int d = 0; d < R; d++)
      hs.feas_eff[(size_t)f * R + d] = (d < 2 || ((t_active[t] >> d) & 1u)) ? hs.t_init[(size_t)d * T + t] : 0.0;
  }

  mark("shape tables, inter-pod");
  // ---- plugin OnSessionOpen state ----
  // drf.go:60-64 / proportion.go:58-62: total = sum of Allocatable over ssn.Nodes (ascending node name)
  hs.total = Res();
  nmask.assign(NP, 0);
  for (uint32_t n = 0; n < N; n++) {
    Res a;
    a.mask = sn->node_scalar_mask ? sn->node_scalar_mask[n] : 0;
    nmask[n] = a.mask & 0x3FFFFFFFu;
    for (int d = 2; d < R; d++)   // Releasing gains scalar keys only through Add of a Releasing task's Resreq: dense non-zero <=> key present
      if (sn->node_releasing[(size_t)d * N + n] != 0.0) nmask[n] |= 0x80000000u;
    for (int d = 0; d < R; d++) a.v[d] = (d < 2 || a.has(d)) ? sn->node_allocatable[(size_t)d * N + n] : 0.0;
    res_add(hs.total, a, R);
  }
  // proportion.go:65-154 water-filling over the queues that own a job, ascending QueueID
  hs.deserved.assign(Q, Res());
  hs.queue_has_attr.assign(Q, 0);
  std::vector<Res> request(Q);
  for (uint32_t j = 0; j < J; j++) {
    const uint32_t q = hs.job_queue[j];
    if (q >= Q) {
      // allocate / preempt / reclaim skip such a job ("queue not found", allocate.go:56-60) but proportion's OnSessionOpen reads
      // ssn.Queues[job.Queue].UID for every job (proportion.go:70-73): with the plugin loaded the reference panics on the nil queue
      if (pol.has_proportion) throw EngineError(KB_E_UNSUPPORTED, "a job names a queue the session does not hold (the proportion plugin would panic on it)");
      continue;
    }
    hs.queue_has_attr[q] = 1;
  }
  // Resource.Add (resource_info.go:128-140) of every allocated-status or Pending task's Resreq onto its queue's request, job by job, task by
  // task: cpu and memory always, a scalar where the task has the key.  `whole`: every addend a whole number below 2^53
  auto add_jobs = [&](uint32_t j0, uint32_t j1, std::vector<Res> &req, bool *whole) {
    auto is_whole = [](double v) { return v < 9007199254740992.0 && (double)(long long)v == v; };   // requests are >= 0 here (validated above); NaN fails the first test
    bool w = true;
    for (uint32_t j = j0; j < j1; j++) {
      const uint32_t q = hs.job_queue[j];
      if (q >= Q) continue;
      Res &rq = req[q];
      for (uint32_t t = hs.job_begin[j]; t < hs.job_begin[j + 1]; t++) {
        const int st = hs.t_status[t];
        const bool alloc_st = st == KB_TASK_BOUND || st == KB_TASK_BINDING || st == KB_TASK_RUNNING || st == KB_TASK_ALLOCATED;
        if (!alloc_st && st != KB_TASK_PENDING) continue;
        const double c = hs.t_res[t], mm = hs.t_res[(size_t)T + t];
        rq.v[0] += c;
        rq.v[1] += mm;
        w = w && is_whole(c) && is_whole(mm);
        const uint32_t m = hs.t_resmask[t];
        for (int d = 2; m != 0 && d < R; d++)
          if ((m >> (d - 2)) & 1u) { const double x = hs.t_res[(size_t)d * T + t]; rq.setk(d); rq.v[d] += x; w = w && is_whole(x); }
      }
    }
    if (whole) *whole = w;
  };
  // The sums are floating point and their order is the reference's — except when it cannot matter: whole numbers whose total stays below
  // 2^53 (milli-cpu, bytes, milli-units of extended resources) add without rounding in ANY order, so ranges of jobs are summed on the host
  // threads and combined; anything else (a fractional quantity, a total at or beyond 2^53) takes the one pass in order.
  const uint32_t nt_req = par_threads(T, 2u * (uint32_t)R + 2u);
  bool summed = false;
  if (nt_req > 1) {
    std::vector<std::vector<Res>> part(nt_req, std::vector<Res>(Q));
    std::vector<uint8_t> whole(nt_req, 0);
    par_parts(J, nt_req, [&](uint32_t i, uint32_t j0, uint32_t j1) { bool w = false; add_jobs(j0, j1, part[i], &w); whole[i] = w ? 1 : 0; });
    bool exact = true;
    for (uint32_t i = 0; i < nt_req; i++) exact = exact && whole[i];
    std::vector<Res> sum(Q);
    for (uint32_t q = 0; q < Q && exact; q++)
      for (uint32_t i = 0; i < nt_req && exact; i++) {
        sum[q].mask |= part[i][q].mask;
        for (int d = 0; d < R; d++) {
          exact = exact && part[i][q].v[d] < 9007199254740992.0;   // a part's total below 2^53: none of its partial sums was rounded
          sum[q].v[d] += part[i][q].v[d];
          exact = exact && sum[q].v[d] < 9007199254740992.0;
        }
      }
    if (exact) { request.swap(sum); summed = true; }
  }
  if (!summed) add_jobs(0, J, request, nullptr);
  mark("total, per-queue request");
  hs.queue_share_at_open = 1;
  hs.queue_request = request;
  if (pol.has_proportion && !hs.waterfill_on_device) {   // on the device: kb_session_load runs kb_launch_waterfill over hs.queue_request (KB_DEVICE_WATERFILL=1)
    Res remaining = hs.total;
    std::vector<uint8_t> meet(Q, 0);
    for (bool first = true;; first = false) {
      int32_t totalWeight = 0;
      for (uint32_t q = 0; q < Q; q++)
        if (hs.queue_has_attr[q] && !meet[q]) totalWeight += hs.queue_weight[q];
      if (totalWeight == 0) {
        if (first) hs.queue_share_at_open = 0;   // proportion.go:113-116: no updateShare ran, shares stay 0 until an event
        break;
      }
      Res increasedDeserved, decreasedDeserved;
      for (uint32_t q = 0; q < Q; q++) {
        if (!hs.queue_has_attr[q] || meet[q]) continue;
        Res oldDeserved = hs.deserved[q];
        Res part = remaining;
        res_multi(part, (double)hs.queue_weight[q] / (double)totalWeight, R);
        res_add(hs.deserved[q], part, R);
        if (res_less(request[q], hs.deserved[q], R)) {
          hs.deserved[q] = helpers_min(hs.deserved[q], request[q], R);
          meet[q] = 1;
        }
        Res inc, dec;
        res_diff(hs.deserved[q], oldDeserved, inc, dec, R);
        res_add(increasedDeserved, inc, R);
        res_add(decreasedDeserved, dec, R);
      }
      if (!res_sub(remaining, increasedDeserved, R))
        throw EngineError(KB_E_UNSUPPORTED, "proportion water-filling underflow (the reference would panic in Resource.Sub)");
      res_add(remaining, decreasedDeserved, R);
      if (res_is_empty(remaining, R)) break;
    }
  }
}

}  // namespace kb
