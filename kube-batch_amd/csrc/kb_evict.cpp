// kb_evict.cpp — kb_run_preempt / kb_run_reclaim: the bridge between the evict machine (kb_preempt.cpp: statements and victims on the host) and the
// device (PredicateNodes + PrioritizeNodes + SortNodes as complete sorted lists, node refreshes).  Split out of kb_engine.cpp in round 6 without a
// change of behaviour (kb_engine_int.hpp has the map).
#include "kb_engine_int.hpp"

extern "C" {

// ---- preempt (kb_preempt.hpp): statements and victims on the host, PredicateNodes + PrioritizeNodes + SortNodes on the device ----
namespace {
// node state of the given nodes: host mirror -> device (after Pipelines / evictions changed it)
// One packed record per node in a persistent pinned staging vector, one copy, one scatter kernel (kb_launch_scatter_nodes) — it used
// to be 2R + 5 tiny asynchronous copies per node out of loop-scoped stack locals (round-2 advisory: correct only because pageable
// sources are staged at the call, and ten driver calls per dirty node on the preempt refresh path).
void upload_live_nodes(kb_engine *e, const LiveNodes &ln, const std::vector<uint32_t> &nodes) {
  if (nodes.empty()) return;
  const int R = e->hs.R;
  const size_t rec = 5 + 2 * (size_t)R, words = rec * nodes.size();
  e->h_scatter.resize(words);
  unsigned long long *w = e->h_scatter.data();
  for (uint32_t n : nodes) {
    const uint32_t nm = (ln.idle[n].mask & 0x3FFFFFFFu) | (ln.rel[n].mask ? 0x80000000u : 0u);
    w[0] = (unsigned long long)n | ((unsigned long long)nm << 32);
    w[1] = (unsigned long long)(uint32_t)ln.podcnt[n];
    w[2] = (unsigned long long)ln.nzc[n];
    w[3] = (unsigned long long)ln.nzm[n];
    w[4] = e->dev.ports ? ln.ports[n] : 0ull;
    for (int d = 0; d < R; d++) {
      const double vi = ln.idle[n].get(d), vr = ln.rel[n].get(d);
      std::memcpy(&w[5 + d], &vi, 8);
      std::memcpy(&w[5 + R + d], &vr, 8);
    }
    w += rec;
  }
  if (e->b_scatter.bytes < sizeof(unsigned long long) * words) e->b_scatter.alloc(sizeof(unsigned long long) * words);   // grown, never shrunk: hipFree synchronises the device
  HIP_OK(hipMemcpyAsync(e->b_scatter.p, e->h_scatter.data(), sizeof(unsigned long long) * words, hipMemcpyHostToDevice, e->stream));
  kb_launch_scatter_nodes(e->dev, e->b_scatter.as<unsigned long long>(), (uint32_t)nodes.size(), e->b_nmask.as<uint32_t>(), e->stream);
  // host-port masks of several words: the words behind the first, one 8-byte copy each (a rare session; [port_xw][NP] on the device, [N][port_xw] here)
  for (uint32_t X = e->dev.port_xw, i = 0; X && i < nodes.size(); i++)
    for (uint32_t w = 0; w < X; w++)
      HIP_OK(hipMemcpyAsync(e->b_ports_x.as<unsigned long long>() + (size_t)w * e->dev.NP + nodes[i], &ln.ports_x[(size_t)nodes[i] * X + w], sizeof(unsigned long long),
                            hipMemcpyHostToDevice, e->stream));
  HIP_OK(hipStreamSynchronize(e->stream));   // the staging vector is reused by the next refresh
}
}  // namespace

static int run_evict_action(kb_engine *e, bool reclaim, kb_stmt_op *out, uint64_t cap, uint64_t *n_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "kb_session_load must precede kb_run_preempt / kb_run_reclaim");
    if (e->tainted) throw EngineError(KB_E_STATE, "a preempt / reclaim call failed after touching the session's state: kb_session_load or kb_session_reset first");
    quiesce(e);
    e->pristine = false;
    e->stale_checked = false;
    HostSession &hs = e->hs;
    // an eviction takes a pod OUT of the inter-pod predicate's pod list (Running -> Releasing leaves api.AllocatedStatus): modelled on the
    // host side of the evict machine (kb_preempt.cpp: ip_*), the lists rebuilt on the device after every change (round 3; on by default
    // since its first device run, round 4: profiles/round4/first_call)
    // preempt with preferred node-affinity terms: the lists of such preemptors carry the NormalizeReduce'd score and are rebuilt after
    // every Pipeline instead of repaired (kb_preempt.cpp: preempt_walk; tests/test_gpu_regressions.py: test_preempt_with_preferred_node_affinity).
    const double t_begin = now_ms();
    const int R = hs.R;
    const uint32_t N = hs.N, NP = e->dev.NP, T = hs.T, J = hs.J, Q = hs.Q;
    // ---- live state: device -> host
    LiveNodes ln;
    std::vector<uint8_t> counted_in;
    const double t_e0 = now_ms();
    double t_e_copy = t_e0, t_e_nodes = t_e0, t_e_tasks = t_e0;   // entry, in parts (KB_EVICT_TRACE)
    {
      // one pinned block, every copy asynchronous on the engine's stream, ONE synchronisation (round 3: seven blocking pageable copies)
      const size_t o_idle = 0, o_rel = o_idle + sizeof(double) * (size_t)R * NP, o_nzc = o_rel + sizeof(double) * (size_t)R * NP, o_nzm = o_nzc + sizeof(long long) * NP,
                   o_ports = o_nzm + sizeof(long long) * NP, o_nmask = o_ports + sizeof(unsigned long long) * NP, o_pod = o_nmask + sizeof(uint32_t) * NP,
                   o_cnt = o_pod + sizeof(int) * NP, o_end = o_cnt + (((size_t)T + 7) & ~(size_t)7);
      e->h_evict.resize(o_end + 16);
      unsigned char *hb = e->h_evict.data();
      HIP_OK(hipMemcpyAsync(hb + o_idle, e->b_idle.p, sizeof(double) * (size_t)R * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_rel, e->b_rel.p, sizeof(double) * (size_t)R * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_nzc, e->b_nzc.p, sizeof(long long) * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_nzm, e->b_nzm.p, sizeof(long long) * NP, hipMemcpyDeviceToHost, e->stream));
      if (e->dev.ports) HIP_OK(hipMemcpyAsync(hb + o_ports, e->b_ports.p, sizeof(unsigned long long) * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_nmask, e->b_nmask.p, sizeof(uint32_t) * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_pod, e->b_podcnt.p, sizeof(int) * NP, hipMemcpyDeviceToHost, e->stream));
      if (T) HIP_OK(hipMemcpyAsync(hb + o_cnt, e->b_tcounted.p, T, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipStreamSynchronize(e->stream));
      t_e_copy = now_ms();
      const double *idle = reinterpret_cast<const double *>(hb + o_idle), *rel = reinterpret_cast<const double *>(hb + o_rel);
      const uint32_t *nmask = reinterpret_cast<const uint32_t *>(hb + o_nmask);
      ln.nzc.assign(reinterpret_cast<const long long *>(hb + o_nzc), reinterpret_cast<const long long *>(hb + o_nzc) + NP);
      ln.nzm.assign(reinterpret_cast<const long long *>(hb + o_nzm), reinterpret_cast<const long long *>(hb + o_nzm) + NP);
      ln.podcnt.assign(reinterpret_cast<const int *>(hb + o_pod), reinterpret_cast<const int *>(hb + o_pod) + NP);
      if (e->dev.ports) ln.ports.assign(reinterpret_cast<const unsigned long long *>(hb + o_ports), reinterpret_cast<const unsigned long long *>(hb + o_ports) + NP);
      else ln.ports.assign(NP, 0);
      ln.ports_x.assign((size_t)N * hs.port_xw, 0);
      if (e->dev.port_xw) {   // the masks' words behind the first: [port_xw][NP] on the device, [N][port_xw] in the machine
        std::vector<unsigned long long> px((size_t)e->dev.port_xw * NP);
        HIP_OK(hipMemcpyAsync(px.data(), e->b_ports_x.p, sizeof(unsigned long long) * px.size(), hipMemcpyDeviceToHost, e->stream));
        HIP_OK(hipStreamSynchronize(e->stream));
        for (uint32_t n = 0; n < N; n++)
          for (uint32_t w = 0; w < hs.port_xw; w++) ln.ports_x[(size_t)n * hs.port_xw + w] = px[(size_t)w * NP + n];
      }
      counted_in.assign(hb + o_cnt, hb + o_cnt + T);
      ln.idle.assign(N, Res()); ln.rel.assign(N, Res());
      for (uint32_t n = 0; n < N; n++) {
        ln.idle[n].mask = nmask[n] & 0x3FFFFFFFu;
        for (int d = 0; d < R; d++) {
          ln.idle[n].v[d] = idle[(size_t)d * NP + n];
          ln.rel[n].v[d] = rel[(size_t)d * NP + n];
          // Releasing gains scalar keys only through Add: a dense non-zero value <=> the key is present
          if (d >= 2 && ln.rel[n].v[d] != 0.0) ln.rel[n].setk(d);
          // Idle: the device's mask holds the keys Allocatable had (plus what an earlier evict action uploaded).  Resource.Sub also
          // CREATES the keys of its operand in a non-nil map (resource_info.go:143-160: r.ScalarResources[name] -= quant), which is how
          // allocate / backfill leave a negative value under a key the node never advertised (sub-epsilon requests pass LessEqual
          // and add up).  Such a key reads non-zero, and a created key that reads 0 is indistinguishable from an absent one.
          if (d >= 2 && ln.idle[n].mask != 0 && ln.idle[n].v[d] != 0.0) ln.idle[n].setk(d);
        }
        // a non-nil Releasing map whose keys all read 0 (bit 31): which keys it holds does not matter, that Sub does not return early does
        if ((nmask[n] >> 31) && ln.rel[n].mask == 0 && R > 2) ln.rel[n].setk(2);
      }
      ln.ac.assign(hs.n_ac.begin(), hs.n_ac.end()); ln.am.assign(hs.n_am.begin(), hs.n_am.end());
      ln.maxpods = hs.n_maxpods; ln.cls = hs.n_cls;
    }
    t_e_nodes = now_ms();
    if (!e->evict_machine) e->evict_machine.reset(new PreemptMachine());   // one machine per engine: its tables keep their storage between actions and cycles
    PreemptMachine &pm = *e->evict_machine;
    pm.counted.assign(counted_in.begin(), counted_in.end());
    if (pm.counted.empty()) pm.counted.resize(1);
    pm.jalloc = hs.job_alloc; pm.jshare = hs.job_share; pm.qalloc = hs.queue_alloc; pm.qshare = hs.queue_share;
    pm.jmask.assign(J ? J : 1, 0); pm.qmask.assign(Q ? Q : 1, 0);
    for (uint32_t t = 0; t < T; t++)
      if (pm.counted[t] && hs.t_job[t] < J) {
        pm.jmask[hs.t_job[t]] |= hs.t_resmask[t];
        if (hs.job_queue[hs.t_job[t]] < Q) pm.qmask[hs.job_queue[hs.t_job[t]]] |= hs.t_resmask[t];
      }
    // ---- the device side: one complete sorted list per preemptor shape, on demand
    double tl_lists = 0.0, tl_lists_host = 0.0, tl_refresh = 0.0;
    uint64_t n_lists = 0, n_refresh = 0;
    auto lists = [&](uint32_t task, std::vector<uint64_t> &keys) {
      const double tl0 = now_ms();
      ensure_window_buffers(e, 1);
      ensure_matrix_buffers(e, 1, N + 1);
      HIP_OK(hipMemcpyAsync(e->b_mrows.p, &task, sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
      KbRound r = make_round(e, 0, 1, N + 1, 0 /* plugin predicates only */, false);
      r.mrows = e->b_mrows.as<uint32_t>();
      kb_launch_matrix(e->dev, r, e->stream);
      kb_launch_affinity(e->dev, r, e->stream);   // NodeAffinity priority over the row's feasible set (no-op without such terms)
      kb_launch_interpod(e->dev, r, e->stream);   // InterPodAffinityPriority over the same set, against the counts uploaded last (no-op without such terms)
      kb_launch_argmax(e->dev, r, e->stream);
      e->h_listkeys.resize((size_t)N + 1);   // pinned, persistent: the copy is a DMA into place instead of a staged pageable copy into a fresh vector
      const size_t raw_n = (size_t)N + 1;
      const unsigned long long *raw = e->h_listkeys.data();
      HIP_OK(hipMemcpyAsync(e->h_listkeys.data(), e->b_keys.p, sizeof(unsigned long long) * raw_n, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipStreamSynchronize(e->stream));
      HIP_OK(hipGetLastError());
      const double tl1 = now_ms();
      tl_lists += tl1 - tl0; n_lists++;
      e->stats.matrix_launches += 1;
      e->stats.matrix_evals += N;
      // K3 orders (score descending, node ASCENDING); SortNodes breaks score ties by DESCENDING host name: reverse every run
      keys.clear();
      size_t i = 0;
      keys.reserve(raw_n);
      while (i < raw_n && raw[i] != 0ull) {
        size_t k = i;
        const uint32_t sc = KB_KEY_SCORE(raw[i]);
        while (k < raw_n && raw[k] != 0ull && KB_KEY_SCORE(raw[k]) == sc) k++;
        for (size_t q = k; q-- > i;) keys.push_back(((uint64_t)sc << 32) | KB_KEY_NODE(raw[q]));
        i = k;
      }
      tl_lists_host += now_ms() - tl1;
    };
    auto refresh = [&](const std::vector<uint32_t> &nodes) { const double t0 = now_ms(); upload_live_nodes(e, ln, nodes); tl_refresh += now_ms() - t0; n_refresh++; };
    std::vector<uint8_t> status = hs.t_status;
    std::vector<uint32_t> tnode = hs.t_node;
    t_e_tasks = now_ms();
    pm.init(&hs, &e->pol, &ln, &status, &tnode, lists, refresh);
    // inter-pod terms: the live counts (allocate / backfill of this session may have advanced them) come to the host; the machine keeps them
    // current and puts them back on the device in front of every list it asks for, and once more when the action is over
    IpLive ipl;
    auto ip_upload = [&]() {
      if (!ipl.ccnt.empty()) HIP_OK(hipMemcpyAsync(e->b_ip_ccnt.p, ipl.ccnt.data(), sizeof(int32_t) * ipl.ccnt.size(), hipMemcpyHostToDevice, e->stream));
      if (!ipl.ctot.empty()) HIP_OK(hipMemcpyAsync(e->b_ip_ctot.p, ipl.ctot.data(), sizeof(int32_t) * ipl.ctot.size(), hipMemcpyHostToDevice, e->stream));
      if (!ipl.punb.empty()) HIP_OK(hipMemcpyAsync(e->b_ip_punb.p, ipl.punb.data(), sizeof(int32_t) * ipl.punb.size(), hipMemcpyHostToDevice, e->stream));
      HIP_OK(hipMemcpyAsync(e->b_ip_z.p, &ipl.z, sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
      HIP_OK(hipStreamSynchronize(e->stream));   // the sources are this frame's vectors, and they change again before the next call
    };
    if (hs.has_interpod) {
      ipl.NP = NP;
      ipl.ccnt.resize((size_t)std::max(hs.ip_C, 1u) * hs.ip_D); ipl.ctot.resize(std::max(hs.ip_C, 1u)); ipl.punb.resize((size_t)std::max(hs.ip_P, 1u) * NP);
      HIP_OK(hipMemcpy(ipl.ccnt.data(), e->b_ip_ccnt.p, sizeof(int32_t) * ipl.ccnt.size(), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(ipl.ctot.data(), e->b_ip_ctot.p, sizeof(int32_t) * ipl.ctot.size(), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(ipl.punb.data(), e->b_ip_punb.p, sizeof(int32_t) * ipl.punb.size(), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(&ipl.z, e->b_ip_z.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
      pm.set_interpod(&ipl, ip_upload);
    }
    // from here on a failure leaves state behind (a mid-action refresh may have updated nodes on the device; after the journal is out,
    // host and device state are committed): whatever throws below, the session is marked tainted and every kb_run_* answers KB_E_STATE
    // until kb_session_load / kb_session_reset (round-2 advisory: the cross-check at the end used to fail AFTER publishing results)
    struct Taint { kb_engine *e; bool armed = true; ~Taint() { if (armed) e->tainted = true; } } taint{e};
    const double t_e1 = now_ms();
    if (reclaim) pm.run_reclaim(); else pm.run();
    const double t_e2 = now_ms();
    // ---- results: journal out, state back to the device
    if (n_out) *n_out = pm.ops.size();
    if (pm.ops.size() > cap) throw EngineError(KB_E_CAPACITY, "journal buffer too small");   // no result was written; a refresh may have updated nodes on the device: load the session again before another action
    for (size_t i = 0; i < pm.ops.size(); i++) { out[i].op = pm.ops[i].op; out[i].task = pm.ops[i].task; out[i].node = pm.ops[i].node; out[i].stmt = pm.ops[i].stmt; }
    upload_live_nodes(e, ln, pm.touched_nodes);
    if (hs.has_interpod) ip_upload();   // what the next action's kernels read
    hs.t_status = status;
    hs.t_node = tnode;
    pm.off_node_tasks(hs.t_off_node);
    if (T) {   // the task table back: through the pinned block, asynchronous, ordered in front of the finalize launches on the same stream
      const size_t t8 = ((size_t)T + 7) & ~(size_t)7;
      e->h_evict.resize(2 * t8 + sizeof(uint32_t) * (size_t)T + 16);
      unsigned char *hb = e->h_evict.data();
      std::memcpy(hb, status.data(), T);
      std::memcpy(hb + t8, pm.counted.data(), T);
      std::memcpy(hb + 2 * t8, tnode.data(), sizeof(uint32_t) * (size_t)T);
      HIP_OK(hipMemcpyAsync(e->b_tstatus.p, hb, T, hipMemcpyHostToDevice, e->stream));
      HIP_OK(hipMemcpyAsync(e->b_tcounted.p, hb + t8, T, hipMemcpyHostToDevice, e->stream));
      HIP_OK(hipMemcpyAsync(e->b_tnode.p, hb + 2 * t8, sizeof(uint32_t) * (size_t)T, hipMemcpyHostToDevice, e->stream));
    }
    e->evictions.insert(e->evictions.end(), pm.evictions.begin(), pm.evictions.end());
    for (const StmtOp &op : pm.ops)   // Evict / Pipeline fire proportion's handlers -> updateShare for the task's queue
      if (op.task != KB_NONE && hs.job_queue[hs.t_job[op.task]] < Q) hs.queue_share_live[hs.job_queue[hs.t_job[op.task]]] = 1;
    const double t_x_fin = now_ms();
    run_finalize(e);
    // the host's running drf / proportion aggregates must equal the device reduction over the task table
    if (e->pol.has_drf)
      for (uint32_t j = 0; j < J; j++)
        if (pm.jshare[j] != hs.job_share[j]) throw EngineError(KB_E_INTERNAL, "evict action: drf share diverged from the device reduction at job " + std::to_string(j));
    e->stats.tasks_popped += pm.popped;
    e->stats.evals += pm.evals;
    e->stats.total_ms += now_ms() - t_begin;
    static const bool ev_trace = [] { const char *v = getenv("KB_EVICT_TRACE"); return v && v[0] == '1'; }();
    if (ev_trace)   // host timeline of the action (profiles/round4)
      fprintf(stderr, "[kb evict] %s: entry %.2f ms = copies %.2f + node mirror %.2f + task tables %.2f + machine tables %.2f; machine set-up %.2f ms (job / task queues) + run, of it %.2f ms collecting candidates (%llu queue nodes looked at); exit: journal + state back %.2f, finalize + checks %.2f\n",
              reclaim ? "reclaim" : "preempt", t_e1 - t_e0, t_e_copy - t_e0, t_e_nodes - t_e_copy, t_e_tasks - t_e_nodes, t_e1 - t_e_tasks, pm.tr_setup_ms, pm.tr_scan_ms,
              (unsigned long long)pm.tr_scan_nodes, t_x_fin - t_e2, now_ms() - t_x_fin);
    if (ev_trace)
      fprintf(stderr, "[kb evict] %s: entry (state to the host, machine set-up) %.2f ms; machine %.2f ms of which %llu lists %.2f ms on the device + %.2f ms host reorder, %llu node refreshes %.2f ms; exit (journal, state back, finalize, checks) %.2f ms; popped %llu (walked %llu: %.2f ms, %llu nodes tried; skipped with their job %llu, turned away one by one %llu, own-job preemptors the priority rule excludes %llu), journal %zu\n",
              reclaim ? "reclaim" : "preempt", t_e1 - t_e0, t_e2 - t_e1, (unsigned long long)n_lists, tl_lists, tl_lists_host, (unsigned long long)n_refresh, tl_refresh, now_ms() - t_e2,
              (unsigned long long)pm.popped, (unsigned long long)pm.tr_walks, pm.tr_walk_ms, (unsigned long long)pm.tr_tries, (unsigned long long)pm.tr_skipped, (unsigned long long)pm.tr_shortcut, (unsigned long long)pm.tr_pruned, pm.ops.size());
    taint.armed = false;
  });
}

int kb_run_preempt(kb_engine *e, kb_stmt_op *out, uint64_t cap, uint64_t *n_out) { return run_evict_action(e, false, out, cap, n_out); }
int kb_run_reclaim(kb_engine *e, kb_stmt_op *out, uint64_t cap, uint64_t *n_out) { return run_evict_action(e, true, out, cap, n_out); }

int kb_get_evictions(kb_engine *e, uint32_t *out, uint64_t cap, uint64_t *n_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    if (n_out) *n_out = e->evictions.size();
    if (e->evictions.size() > cap) throw EngineError(KB_E_CAPACITY, "eviction buffer too small");
    if (out && !e->evictions.empty()) std::memcpy(out, e->evictions.data(), sizeof(uint32_t) * e->evictions.size());
  });
}

}  // extern "C"
