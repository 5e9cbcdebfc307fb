// kb_load.cpp — kb_session_load (host session, every upload through one pinned area, proportion's water-fill as a launch, the load's one
// synchronisation) and kb_session_reset.  Split out of kb_engine.cpp in round 6 without a change of behaviour (kb_engine_int.hpp has the map).
#include "kb_engine_int.hpp"

extern "C" {

// proportion's OnSessionOpen water-fill as a launch (kb_waterfill.hip; default since its first device run, round 4; KB_DEVICE_WATERFILL=0: the host loop): the queues' requests, weights and the
// session's total go up, `deserved` comes back for the host's order machine (Overused, the queue order) and stays on the device for
// k_finalize_queues.  build_host_session left hs.deserved at zero.
// Queued on the engine's stream, nothing waited for: the launch leaves `deserved` where k_finalize_queues reads it (b_deserved / b_desmask,
// in their [R][Q] layout) and the queue records and flags travel back into the load's pinned area; waterfill_collect reads them behind
// the synchronisation that ends the load (run_finalize's).
struct WaterfillInFlight { WfQueue *qs = nullptr; WfState *st = nullptr; };
static WaterfillInFlight device_waterfill_queue(kb_engine *e) {
  HostSession &hs = e->hs;
  const uint32_t Q = hs.Q;
  const size_t nq = Q ? Q : 1;
  WaterfillInFlight w;
  w.qs = reinterpret_cast<WfQueue *>(e->load_arena.take(sizeof(WfQueue) * nq));
  w.st = reinterpret_cast<WfState *>(e->load_arena.take(sizeof(WfState)));
  for (size_t q = 0; q < nq; q++) new (&w.qs[q]) WfQueue();
  new (w.st) WfState();
  for (uint32_t q = 0; q < Q; q++) {
    w.qs[q].request = hs.queue_request[q];
    w.qs[q].weight = hs.queue_weight[q];
    w.qs[q].has_attr = hs.queue_has_attr[q];
    w.qs[q].meet = 0;
    w.qs[q].active = 0;
  }
  WfState &st = *w.st;
  st.remaining = hs.total;
  st.total_weight = 0; st.stop = 0; st.share_at_open = 1; st.underflow = 0; st.passes = 0;
  DevBuf &b_q = e->b_wf_queues, &b_st = e->b_wf_state;   // kept between loads (the Go action loads a session every cycle): grown, never shrunk
  b_q.alloc(sizeof(WfQueue) * nq);
  b_st.alloc(sizeof(WfState));
  e->b_deserved.alloc(sizeof(double) * (size_t)hs.R * nq);
  e->b_desmask.alloc(sizeof(uint32_t) * nq);
  HIP_OK(hipMemcpyAsync(b_q.p, w.qs, sizeof(WfQueue) * nq, hipMemcpyHostToDevice, e->stream));
  HIP_OK(hipMemcpyAsync(b_st.p, w.st, sizeof(WfState), hipMemcpyHostToDevice, e->stream));
  kb_launch_waterfill(b_q.as<WfQueue>(), Q, b_st.as<WfState>(), hs.R, e->b_deserved.as<double>(), e->b_desmask.as<uint32_t>(), e->stream);
  HIP_OK(hipMemcpyAsync(w.qs, b_q.p, sizeof(WfQueue) * nq, hipMemcpyDeviceToHost, e->stream));
  HIP_OK(hipMemcpyAsync(w.st, b_st.p, sizeof(WfState), hipMemcpyDeviceToHost, e->stream));
  return w;
}
static void waterfill_collect(kb_engine *e, const WaterfillInFlight &w) {
  HostSession &hs = e->hs;
  HIP_OK(hipGetLastError());
  if (w.st->underflow) throw EngineError(KB_E_UNSUPPORTED, "proportion water-filling underflow (the reference would panic in Resource.Sub)");
  for (uint32_t q = 0; q < hs.Q; q++) hs.deserved[q] = w.qs[q].deserved;
  hs.queue_share_at_open = w.st->share_at_open ? 1 : 0;
  e->waterfill_passes = w.st->passes;
}

int kb_session_load(kb_engine *e, const kb_snapshot *sn) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!sn) throw EngineError(KB_E_INVALID, "snapshot is NULL");
    if (sn->version != KB_ABI_VERSION) throw EngineError(KB_E_INVALID, "snapshot ABI version mismatch");
    if (sn->n_res < 2 || sn->n_res > KB_MAX_RES) throw EngineError(KB_E_INVALID, "n_res out of range");
    quiesce(e);
    e->loaded = false;
    e->fin0.valid = false;
    e->stale_checked = false; e->pristine = true; e->load_clean = false;
    mg_reset(e->mg);   // (its device buffers are grow-only like every other one: no hipFree / hipMalloc per cycle)
    HostSession &hs = e->hs;
    const uint32_t NP = ((sn->n_nodes + KB_NODE_PAD - 1) / KB_NODE_PAD) * KB_NODE_PAD + (sn->n_nodes == 0 ? KB_NODE_PAD : 0);
    std::vector<uint32_t> t_active, nmask;
    hs.waterfill_on_device = e->device_waterfill && e->pol.has_proportion;
    // KB_LOAD_TRACE=1: where a load's time goes, phase by phase, on stderr (the Go action loads a session every cycle)
    static const bool load_trace = [] { const char *v = getenv("KB_LOAD_TRACE"); return v && v[0] == '1'; }();
    double t_mark = now_ms();
    auto mark = [&](const char *what) {
      if (!load_trace) return;
      const double t = now_ms();
      fprintf(stderr, "kb_session_load: %-28s %8.3f ms\n", what, t - t_mark);
      t_mark = t;
    };
    build_host_session(sn, e->pol, NP, hs, t_active, nmask);   // kb_session.cpp: validation, shapes, plugin OnSessionOpen state
    mark("build_host_session");
    const int R = hs.R;
    const uint32_t N = hs.N, T = hs.T, J = hs.J, Q = hs.Q;
    const kb_interpod *ip = sn->interpod;
    hipStream_t s = e->stream;
    e->evictions.clear();

    // ---- device upload: every source goes through the load's pinned area (PinnedArena above), every copy is asynchronous on the engine's
    //      stream, and the ONE synchronisation of a load is the one that ends it (run_finalize's, below)
    e->load_arena.reset();
    e->async_pending = true;   // from here on copies out of (and the water-fill's answers into) the pinned area are queued: a validation that throws below
                               // leaves them in flight, and the next load's quiesce() must wait for them before the area is handed out again
    Uploader up(e->load_arena, s);
    WaterfillInFlight wf_flight;
    // proportion's water-fill first: it needs the host session only, and its (tiny, serial) launch runs while the host assembles the rest
    if (hs.waterfill_on_device) wf_flight = device_waterfill_queue(e);
    KbDev &d = e->dev;
    d = KbDev{};
    d.R = R; d.N = N; d.NP = NP; d.T = T; d.J = J; d.Q = Q;
    up.padded(e->b_idle, sn->node_idle, R, N, NP);
    e->idle_below_eps = false;   // NodeInfo keeps Idle above -epsilon (every Sub is guarded by LessEqual); a snapshot may not
    for (uint32_t n = 0; n < N; n++)
      if (sn->node_idle[n] <= -kMinMilliCPU || sn->node_idle[(size_t)N + n] <= -kMinMemory) e->idle_below_eps = true;
    up.padded(e->b_rel, sn->node_releasing, R, N, NP);
    up.padded(e->b_nzc, sn->node_nz_cpu, 1, N, NP);
    up.padded(e->b_nzm, sn->node_nz_mem, 1, N, NP);
    up.padded(e->b_podcnt, sn->node_pod_cnt, 1, N, NP);
    up.padded(e->b_acpu, sn->node_alloc_cpu, 1, N, NP);
    up.padded(e->b_amem, sn->node_alloc_mem, 1, N, NP);
    up.padded(e->b_maxpods, sn->node_max_pods, 1, N, NP);
    {   // reciprocals of the allocatable quantities for the exact integer-division estimate (IEEE division, same on host and device)
      double *ia = up.stage<double>(e->b_invac, NP);
      for (uint32_t n = 0; n < N; n++) ia[n] = 1.0 / (double)sn->node_alloc_cpu[n];
      std::fill(ia + N, ia + NP, 0.0);
      up.commit();
      double *im = up.stage<double>(e->b_invam, NP);
      for (uint32_t n = 0; n < N; n++) im[n] = 1.0 / (double)sn->node_alloc_mem[n];
      std::fill(im + N, im + NP, 0.0);
      up.commit();
    }
    // The commit kernel keeps the window in LDS (160 KiB per workgroup on gfx950): one dirty slot per row (one thread of the
    // 256-thread workgroup evaluates one slot), the row descriptors, and per distinct shape its candidate list.  Prefer the
    // largest window that still admits 64 shapes.
    {
      const uint32_t budget = 160u * 1024u;
      const uint32_t W = std::min<uint32_t>(e->window, KB_K5_MAX_ROWS);
      uint32_t best_w = 0, best_s = 0;
      for (uint32_t w = W; w >= 1; w = (w > 32 ? ((w - 1) / 32) * 32 : w - 1)) {
        uint32_t sc = std::min<uint32_t>(KB_K5_MAX_SHAPES, w);
        while (sc > 0 && kb_commit_smem_bytes(w, sc, NP, R) > budget) sc--;
        if (sc >= std::min<uint32_t>(64, w)) { best_w = w; best_s = sc; break; }
        if (sc > best_s) { best_w = w; best_s = sc; }
        if (w == 1) break;
      }
      if (best_w == 0 || best_s == 0) throw EngineError(KB_E_UNSUPPORTED, "too many nodes / resource dimensions for the commit kernel's LDS tables");
      e->eff_window = best_w;
      e->shape_cap = best_s;
    }
    {   // 32-bit keys: (score + 1) << node_bits | inverted node index
      const long long max_score = 10ll * ((long long)e->pol.wL + e->pol.wM + e->pol.wB + e->pol.wNA);
      if (((unsigned long long)(max_score + 2) << kb_node_bits(NP)) > (1ull << 32))
        throw EngineError(KB_E_UNSUPPORTED, "score range x node count exceeds the commit kernel's 32-bit keys");
    }
    mark("node arrays, window");
    const uint32_t *ncls;   // the staged copy stays readable for the range checks below (the area is only reset by the next load)
    {
      uint32_t *p = up.stage<uint32_t>(e->b_ncls, NP);
      std::fill(p, p + NP, 0u);
      if (sn->node_class) std::memcpy(p, sn->node_class, sizeof(uint32_t) * N);
      up.commit();
      ncls = p;
    }
    up.copy(e->b_nmask, nmask.data(), NP);
    up.copy_persistent(e->b_tinit, hs.t_init.data(), (size_t)R * T);
    {   // the backfill view of t_init: cpu / memory of a BestEffort task are its Resreq (scalar rows are never compared for
        // them: every InitResreq scalar is at or below the epsilon, resource_info.go:283-287)
      bool differs = false;
      for (uint32_t t = 0; t < T && !differs; t++)
        differs = hs.t_init_empty[t] && (hs.t_res[t] != hs.t_init[t] || hs.t_res[(size_t)T + t] != hs.t_init[(size_t)T + t]);
      if (differs) {
        std::vector<double> fit(hs.t_init);
        for (uint32_t t = 0; t < T; t++)
          if (hs.t_init_empty[t]) { fit[t] = hs.t_res[t]; fit[(size_t)T + t] = hs.t_res[(size_t)T + t]; }
        up.copy(e->b_tfit, fit.data(), (size_t)R * T);
        e->t_fit = e->b_tfit.as<double>();
      } else {
        e->t_fit = e->b_tinit.as<double>();
      }
    }
    up.copy_persistent(e->b_tres, hs.t_res.data(), (size_t)R * T);
    up.copy_persistent(e->b_tnzc, sn->task_nz_cpu, T);
    up.copy_persistent(e->b_tnzm, sn->task_nz_mem, T);
    up.copy_persistent(e->b_tcls, hs.t_cls.data(), T);
    up.copy_persistent(e->b_tactive, t_active.data(), T);
    up.copy_persistent(e->b_tresmask, hs.t_resmask.data(), T);
    up.copy_persistent(e->b_tjob, hs.t_job.data(), T);
    up.copy_persistent(e->b_tstatus, hs.t_status.data(), T);
    up.copy_persistent(e->b_tnode, hs.t_node.data(), T);
    e->b_tbind.alloc(sizeof(uint32_t) * (T ? T : 1));   // nothing is bound yet: KB_NONE everywhere, set on the device
    HIP_OK(hipMemsetAsync(e->b_tbind.p, 0xFF, sizeof(uint32_t) * (T ? T : 1), s));
    static_assert(KB_NONE == 0xFFFFFFFFu, "t_bind is cleared with a byte pattern");
    {
      uint8_t *counted = up.stage<uint8_t>(e->b_tcounted, T);
      for (uint32_t t = 0; t < T; t++) {
        const int st = hs.t_status[t];
        counted[t] = (st == KB_TASK_BOUND || st == KB_TASK_BINDING || st == KB_TASK_RUNNING || st == KB_TASK_ALLOCATED) ? 1 : 0;   // drf.go:71-77
      }
      up.commit();
    }
    e->b_jallocated.alloc(J ? J : 1);
    HIP_OK(hipMemsetAsync(e->b_jallocated.p, 0, J ? J : 1, s));
    mark("task arrays");
    d.compat = nullptr;
    d.n_nc = sn->n_node_classes ? sn->n_node_classes : 1;
    if (sn->class_compat) {
      size_t nb = ((size_t)sn->n_task_classes * sn->n_node_classes + 7) / 8;
      for (uint32_t t = 0; t < T; t++)
        if (hs.t_cls[t] >= sn->n_task_classes) throw EngineError(KB_E_INVALID, "task class out of range");
      for (uint32_t n = 0; n < N; n++)
        if (ncls[n] >= sn->n_node_classes) throw EngineError(KB_E_INVALID, "node class out of range");
      up.copy(e->b_compat, sn->class_compat, nb);
      d.compat = e->b_compat.as<uint8_t>();
      d.crows = nullptr;
      if (sn->n_node_classes <= 256) {   // word-aligned rows for the commit kernel (one 32-byte fetch per task class)
        std::vector<uint32_t> rows((size_t)sn->n_task_classes * 8, 0u);
        for (uint32_t tc = 0; tc < sn->n_task_classes; tc++)
          for (uint32_t nc = 0; nc < sn->n_node_classes; nc++) {
            size_t bit = (size_t)tc * sn->n_node_classes + nc;
            if ((sn->class_compat[bit >> 3] >> (bit & 7)) & 1) rows[(size_t)tc * 8 + (nc >> 5)] |= 1u << (nc & 31);
          }
        up.copy(e->b_crows, rows.data(), rows.size());
        d.crows = e->b_crows.as<uint32_t>();
      }
    }
    d.ports = nullptr; d.t_want = nullptr; d.t_conf = nullptr;
    d.ports_x = nullptr; d.t_want_x = nullptr; d.t_conf_x = nullptr; d.port_xw = 0;
    if (sn->node_ports || sn->task_port_want || sn->task_port_conflict) {
      const size_t Wh = sn->port_words ? sn->port_words : 1;   // 64-bit words per mask; word 0 here, the others below
      std::vector<unsigned long long> np_(NP, 0ull), tw(T ? T : 1, 0ull), tc(T ? T : 1, 0ull);
      bool any = false;
      for (uint32_t n = 0; n < N && sn->node_ports; n++) { np_[n] = sn->node_ports[(size_t)n * Wh]; any = any || np_[n]; }
      for (uint32_t t = 0; t < T; t++) {
        if (sn->task_port_want) tw[t] = sn->task_port_want[(size_t)t * Wh];
        if (sn->task_port_conflict) tc[t] = sn->task_port_conflict[(size_t)t * Wh];
        if ((tw[t] & ~tc[t]) != 0) throw EngineError(KB_E_INVALID, "a pod's host ports must conflict with themselves (want is not a subset of conflict)");
        any = any || tw[t] || tc[t];
      }
      if (any) {
        up.copy(e->b_ports, np_.data(), NP);
        up.copy(e->b_twant, tw.data(), tw.size());
        up.copy(e->b_tconf, tc.data(), tc.size());
        d.ports = e->b_ports.as<unsigned long long>();
        d.t_want = e->b_twant.as<unsigned long long>();
        d.t_conf = e->b_tconf.as<unsigned long long>();
      }
      if (hs.port_xw) {   // some pod reaches beyond word 0 (kb_host.hpp: t_wide): the words behind it, nodes word-major ([port_xw][NP]: K1 reads runs of nodes)
        const uint32_t X = hs.port_xw;
        std::vector<unsigned long long> nx((size_t)X * NP, 0ull);
        for (uint32_t n = 0; n < N && sn->node_ports; n++)
          for (uint32_t w = 0; w < X; w++) nx[(size_t)w * NP + n] = sn->node_ports[(size_t)n * Wh + 1 + w];
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "host-port words");
        up.copy(e->b_ports_x, nx.data(), nx.size());
        up.copy_persistent(e->b_twant_x, reinterpret_cast<const unsigned long long *>(hs.t_want_x.data()), hs.t_want_x.size());
        up.copy_persistent(e->b_tconf_x, reinterpret_cast<const unsigned long long *>(hs.t_conf_x.data()), hs.t_conf_x.size());
        d.ports_x = e->b_ports_x.as<unsigned long long>();
        d.t_want_x = e->b_twant_x.as<unsigned long long>();
        d.t_conf_x = e->b_tconf_x.as<unsigned long long>();
        d.port_xw = X;
        if (!d.ports) {   // word 0 empty everywhere: the kernels still take the host-port path by d.ports
          up.copy(e->b_ports, np_.data(), NP);
          up.copy(e->b_twant, tw.data(), tw.size());
          up.copy(e->b_tconf, tc.data(), tc.size());
          d.ports = e->b_ports.as<unsigned long long>();
          d.t_want = e->b_twant.as<unsigned long long>();
          d.t_conf = e->b_tconf.as<unsigned long long>();
        }
      }
    }
    d.aff = nullptr;
    d.aff_cls = nullptr;
    hs.cls_has_aff.clear();
    d.wNA = e->pol.wNA;
    if (sn->class_affinity && sn->n_task_classes && sn->n_node_classes) {
      for (uint32_t t = 0; t < T; t++)
        if (hs.t_cls[t] >= sn->n_task_classes) throw EngineError(KB_E_INVALID, "task class out of range");
      for (uint32_t n = 0; n < N; n++)
        if (ncls[n] >= sn->n_node_classes) throw EngineError(KB_E_INVALID, "node class out of range");
      const size_t na = (size_t)sn->n_task_classes * sn->n_node_classes;
      std::vector<uint8_t> has(sn->n_task_classes, 0);
      bool any = false;
      for (uint32_t tc = 0; tc < sn->n_task_classes; tc++)
        for (uint32_t nc = 0; nc < sn->n_node_classes; nc++) {
          const int32_t c = sn->class_affinity[(size_t)tc * sn->n_node_classes + nc];
          if (c < 0 || c > 100000) throw EngineError(KB_E_UNSUPPORTED, "node-affinity count outside 0..100000");
          if (c) { has[tc] = 1; any = true; }
        }
      if (any && e->pol.wNA != 0) {
        if (e->pol.wNA < 0 || 10 * (e->pol.wL + e->pol.wM + e->pol.wB + e->pol.wNA) > 65535)
          throw EngineError(KB_E_UNSUPPORTED, "nodeorder weights exceed the 16-bit score range");
        hs.has_affinity = true;
        hs.cls_has_aff = has;
        up.copy(e->b_aff, sn->class_affinity, na);
        up.copy(e->b_affcls, has.data(), has.size());
        d.aff = e->b_aff.as<int32_t>();
        d.aff_cls = e->b_affcls.as<uint8_t>();
      }
    }
    d.ip_ctr_dom = nullptr; d.ip_ctr_count = nullptr; d.ip_ctr_total = nullptr; d.t_ip_inc = nullptr; d.t_ip_forbid = nullptr;
    d.t_ip_req = nullptr; d.t_ip_self = nullptr; d.t_ip_subject = nullptr; d.ip_cls_dom = nullptr; d.ip_cls_bound = nullptr;
    d.ip_cls_unbound = nullptr; d.t_ip_cls_inc = nullptr; d.t_ip_sig = nullptr; d.ip_sig_w = nullptr; d.ip_z = nullptr;
    d.ip_scratch_cnt = nullptr; d.ip_scratch_hist = nullptr; d.ip_C = d.ip_D = d.ip_P = 0; d.ip_Wc = d.ip_Wp = 1; d.wPA = e->pol.wPA; d.t_ip_checks = nullptr;
    if (ip) {
      const uint32_t C = ip->n_counters, P = ip->n_classes, D = ip->n_domains;
      const uint32_t Wc = C ? (C + 63) / 64 : 1, Wp = P ? (P + 63) / 64 : 1;
      // [rows][N] -> [max(rows, 1)][NP], the pad (and the row of a table without rows) KB_NONE / 0
      auto pad_u32 = [&](DevBuf &b, const uint32_t *src, uint32_t rows) {
        if (rows) { up.padded<uint32_t>(b, src, rows, N, NP, KB_NONE); return; }
        uint32_t *p0 = up.stage<uint32_t>(b, NP);
        std::fill(p0, p0 + NP, KB_NONE);
        up.commit();
      };
      auto pad_i32 = [&](DevBuf &b, const int32_t *src, uint32_t rows) {
        if (rows) { up.padded<int32_t>(b, src, rows, N, NP, 0); return; }
        int32_t *p0 = up.stage<int32_t>(b, NP);
        std::fill(p0, p0 + NP, 0);
        up.commit();
      };
      pad_u32(e->b_ip_cdom, ip->ctr_dom, C);
      pad_u32(e->b_ip_pdom, ip->cls_dom, P);
      pad_i32(e->b_ip_pbound, ip->cls_bound, P);
      pad_i32(e->b_ip_punb, ip->cls_unbound, P);
      std::vector<int32_t> cc((size_t)std::max(C, 1u) * D, 0), ct(std::max(C, 1u), 0);
      if (C) { std::memcpy(cc.data(), ip->ctr_count, sizeof(int32_t) * (size_t)C * D); std::memcpy(ct.data(), ip->ctr_total, sizeof(int32_t) * C); }
      up.copy(e->b_ip_ccnt, cc.data(), cc.size());
      up.copy(e->b_ip_ctot, ct.data(), ct.size());
      up.copy_persistent(e->b_ip_tinc, ip->task_inc, (size_t)T * Wc);
      up.copy_persistent(e->b_ip_tforbid, ip->task_forbid, (size_t)T * Wc);
      up.copy(e->b_ip_tchk, hs.t_ip_checks.data(), T);
      up.copy(e->b_ip_treq, ip->task_require, T);
      up.copy(e->b_ip_tself, ip->task_self, T);
      up.copy(e->b_ip_tsubj, hs.t_ip_subject.data(), T);
      up.copy_persistent(e->b_ip_tcinc, ip->task_cls_inc, (size_t)T * Wp);
      up.copy(e->b_ip_tsig, ip->task_sig, T);
      std::vector<int32_t> sw((size_t)std::max(ip->n_sigs, 1u) * std::max(P, 1u), 0);
      if (ip->n_sigs && P) std::memcpy(sw.data(), ip->sig_weight, sizeof(int32_t) * (size_t)ip->n_sigs * P);
      up.copy(e->b_ip_sigw, sw.data(), sw.size());
      const uint32_t z0 = ip->first_unbound_node;
      up.copy(e->b_ip_z, &z0, 1);
      d.ip_ctr_dom = e->b_ip_cdom.as<uint32_t>(); d.ip_ctr_count = e->b_ip_ccnt.as<int32_t>(); d.ip_ctr_total = e->b_ip_ctot.as<int32_t>();
      d.t_ip_inc = e->b_ip_tinc.as<unsigned long long>(); d.t_ip_forbid = e->b_ip_tforbid.as<unsigned long long>();
      d.t_ip_req = e->b_ip_treq.as<uint16_t>(); d.t_ip_self = e->b_ip_tself.as<uint8_t>(); d.t_ip_subject = e->b_ip_tsubj.as<uint8_t>();
      d.t_ip_checks = e->b_ip_tchk.as<uint8_t>();
      d.ip_cls_dom = e->b_ip_pdom.as<uint32_t>(); d.ip_cls_bound = e->b_ip_pbound.as<int32_t>(); d.ip_cls_unbound = e->b_ip_punb.as<int32_t>();
      d.t_ip_cls_inc = e->b_ip_tcinc.as<unsigned long long>(); d.t_ip_sig = e->b_ip_tsig.as<uint32_t>(); d.ip_sig_w = e->b_ip_sigw.as<int32_t>();
      d.ip_z = e->b_ip_z.as<uint32_t>();
      d.ip_C = C; d.ip_D = D; d.ip_P = P; d.ip_Wc = Wc; d.ip_Wp = Wp;
    }
    mark("classes, ports, inter-pod");
    e->h_probe_alive.resize(std::max<uint32_t>(hs.n_feas_shapes, 1u));
    e->h_probe_rows.resize(std::max<uint32_t>(hs.n_feas_shapes, 1u));
    HIP_OK(hipHostGetDevicePointer((void **)&e->d_probe_alive, e->h_probe_alive.data(), 0));
    HIP_OK(hipHostGetDevicePointer((void **)&e->d_probe_rows, e->h_probe_rows.data(), 0));
    up.copy(e->b_jbegin, hs.job_begin.data(), J + 1);
    up.copy(e->b_jmin, hs.job_min.data(), J);
    up.copy(e->b_jqueue, hs.job_queue.data(), J);
    up.copy(e->b_total, hs.total.v, KB_MAX_RES);
    e->total_mask = hs.total.mask;
    if (!hs.waterfill_on_device) {   // the host loop of kb_session.cpp filled hs.deserved (the launch writes b_deserved / b_desmask itself)
      double *des = up.stage<double>(e->b_deserved, (size_t)R * (Q ? Q : 1));
      std::fill(des, des + (size_t)R * (Q ? Q : 1), 0.0);
      for (uint32_t q = 0; q < Q; q++)
        for (int dd = 0; dd < R; dd++) des[(size_t)dd * Q + q] = hs.deserved[q].get(dd);
      up.commit();
      uint32_t *desmask = up.stage<uint32_t>(e->b_desmask, Q ? Q : 1);
      desmask[0] = 0;
      for (uint32_t q = 0; q < Q; q++) desmask[q] = hs.deserved[q].mask;
      up.commit();
    }
    hs.job_alloc.assign((size_t)J * R, 0.0);
    hs.job_share.assign(J, 0.0);
    hs.queue_alloc.assign((size_t)Q * R, 0.0);
    hs.queue_share.assign(Q, 0.0);
    hs.job_ready.assign(J, 0);
    e->b_jalloc.alloc(sizeof(double) * (size_t)(J ? J : 1) * R);
    e->b_jshare.alloc(sizeof(double) * (J ? J : 1));
    e->b_qalloc.alloc(sizeof(double) * (size_t)(Q ? Q : 1) * R);
    e->b_qshare.alloc(sizeof(double) * (Q ? Q : 1));
    e->b_jready.alloc(sizeof(int) * (J ? J : 1));

    d.idle = e->b_idle.as<double>(); d.rel = e->b_rel.as<double>();
    d.nzc = e->b_nzc.as<long long>(); d.nzm = e->b_nzm.as<long long>(); d.podcnt = e->b_podcnt.as<int>();
    d.acpu = e->b_acpu.as<long long>(); d.amem = e->b_amem.as<long long>();
    d.maxpods = e->b_maxpods.as<int>(); d.ncls = e->b_ncls.as<uint32_t>(); d.nmask = e->b_nmask.as<uint32_t>();
    d.inv_acpu = e->b_invac.as<double>(); d.inv_amem = e->b_invam.as<double>();
    d.t_init = e->b_tinit.as<double>(); d.t_res = e->b_tres.as<double>();
    d.t_nzc = e->b_tnzc.as<long long>(); d.t_nzm = e->b_tnzm.as<long long>();
    d.t_cls = e->b_tcls.as<uint32_t>(); d.t_active = e->b_tactive.as<uint32_t>(); d.t_resmask = e->b_tresmask.as<uint32_t>();
    d.t_job = e->b_tjob.as<uint32_t>(); d.t_status = e->b_tstatus.as<uint8_t>(); d.t_node = e->b_tnode.as<uint32_t>();
    d.t_bind = e->b_tbind.as<uint32_t>(); d.t_counted = e->b_tcounted.as<uint8_t>(); d.j_allocated = e->b_jallocated.as<uint8_t>();
    d.wL = e->pol.wL; d.wM = e->pol.wM; d.wB = e->pol.wB;
    d.pred_enabled = e->pol.pred_enabled ? 1 : 0;
    d.score_enabled = e->pol.nodeorder_enabled ? 1 : 0;
    d.whole = hs.whole ? 1u : 0u;
    e->win_cap = 0; e->mat_cap = 0; e->keys_cap = 0;
    e->mat2_cap = 0; e->stale_cap = 0;   // the second stream's matrix rows are [rows][NP] too: a session with more nodes needs them again
    e->xs_cap = 0;   // kb_eval_matrix's per-shape rows are [shapes][NP] as well (found by tests/test_gpu_reload.py: fewer shapes over more nodes overran them)
    e->stats = kb_stats{};
    e->dirty_share = 0.0;
    e->commit_kernel = e->commit_pin >= 0 ? e->commit_pin : (int)KB_COMMIT_SELECT;
    e->round_no = 0;
    mark("jobs, queues, deserved");
    auto snap_copy = [&](DevBuf &dst, const DevBuf &src) {
      dst.alloc(src.bytes);
      HIP_OK(hipMemcpyAsync(dst.p, src.p, src.bytes, hipMemcpyDeviceToDevice, s));
    };
    snap_copy(e->p_idle, e->b_idle); snap_copy(e->p_rel, e->b_rel); snap_copy(e->p_nzc, e->b_nzc); snap_copy(e->p_nzm, e->b_nzm);
    snap_copy(e->p_podcnt, e->b_podcnt); snap_copy(e->p_tstatus, e->b_tstatus); snap_copy(e->p_tnode, e->b_tnode);
    if (d.ports) snap_copy(e->p_ports, e->b_ports);
    if (d.ports_x) snap_copy(e->p_ports_x, e->b_ports_x);
    snap_copy(e->p_tcounted, e->b_tcounted);
    snap_copy(e->p_nmask, e->b_nmask);   // the evict actions rewrite the key masks of the nodes they touch (upload_live_nodes)
    if (hs.has_interpod) { snap_copy(e->p_ip_ccnt, e->b_ip_ccnt); snap_copy(e->p_ip_ctot, e->b_ip_ctot); snap_copy(e->p_ip_punb, e->b_ip_punb); snap_copy(e->p_ip_z, e->b_ip_z); }
    // initial drf / proportion / gang aggregates come from the device reduction (K2+K4)
    mark("pristine copies (queued)");
    // the load's synchronisation; the water-fill's answer (deserved, "did a pass run") is read behind it, in front of the host-side
    // post-processing of the shares, which wants to know whether updateShare ran at open
    run_finalize(e, [&]() {
      if (hs.waterfill_on_device) waterfill_collect(e, wf_flight);
      hs.queue_share_live.assign(Q ? Q : 1, hs.queue_share_at_open);
    });
    mark("aggregates (device reduction, the load's one synchronisation)");
    e->fin0.job_alloc = hs.job_alloc; e->fin0.job_share = hs.job_share; e->fin0.queue_alloc = hs.queue_alloc; e->fin0.queue_share = hs.queue_share;
    e->fin0.job_ready = hs.job_ready; e->fin0.t_status = hs.t_status; e->fin0.t_node = hs.t_node; e->fin0.valid = true;
    e->stats.reduce_ms = 0;
    e->loaded = true;
    e->tainted = false;
  });
}

int kb_session_reset(kb_engine *e) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    const double t_reset0 = now_ms();
    e->tainted = false;
    e->pristine = true;
    e->stale_checked = e->load_clean;
    hipStream_t s = e->stream;
    auto restore = [&](DevBuf &dst, const DevBuf &src) { HIP_OK(hipMemcpyAsync(dst.p, src.p, src.bytes, hipMemcpyDeviceToDevice, s)); };
    restore(e->b_idle, e->p_idle); restore(e->b_rel, e->p_rel); restore(e->b_nzc, e->p_nzc); restore(e->b_nzm, e->p_nzm);
    restore(e->b_podcnt, e->p_podcnt); restore(e->b_tstatus, e->p_tstatus); restore(e->b_tnode, e->p_tnode);
    if (e->dev.ports) restore(e->b_ports, e->p_ports);
    if (e->dev.ports_x) restore(e->b_ports_x, e->p_ports_x);
    restore(e->b_tcounted, e->p_tcounted);
    restore(e->b_nmask, e->p_nmask);
    if (e->hs.has_interpod) { restore(e->b_ip_ccnt, e->p_ip_ccnt); restore(e->b_ip_ctot, e->p_ip_ctot); restore(e->b_ip_punb, e->p_ip_punb); restore(e->b_ip_z, e->p_ip_z); }
    HIP_OK(hipMemsetAsync(e->b_tbind.p, 0xFF, sizeof(uint32_t) * (e->hs.T ? e->hs.T : 1), s));
    HIP_OK(hipMemsetAsync(e->b_jallocated.p, 0, e->b_jallocated.bytes, s));
    mg_reset(e->mg);   // (its device buffers are grow-only like every other one: no hipFree / hipMalloc per cycle)
    std::fill(e->hs.queue_share_live.begin(), e->hs.queue_share_live.end(), e->hs.queue_share_at_open);
    e->evictions.clear();
    e->hs.t_off_node.clear();
    // The restored state is bit for bit the one kb_session_load reduced (the pristine copies were taken in front of that reduction,
    // which flips no status: no job has an Allocate yet): its results come back from the host copies made then.  The device-side result
    // buffers keep the previous reduction's values; nothing reads them before the next reduction rewrites them.
    if (e->fin0.valid) {
      e->async_pending = true;   // stream-ordered with everything run_allocate / run_backfill launch; quiesce() for the rest
      HostSession &hs = e->hs;
      hs.job_alloc = e->fin0.job_alloc; hs.job_share = e->fin0.job_share; hs.queue_alloc = e->fin0.queue_alloc; hs.queue_share = e->fin0.queue_share;
      hs.job_ready = e->fin0.job_ready; hs.t_status = e->fin0.t_status; hs.t_node = e->fin0.t_node;
    } else {
      double keep = e->stats.reduce_ms;
      run_finalize(e);
      e->stats.reduce_ms = keep;
    }
    e->tl_reset += now_ms() - t_reset0;
  });
}

}  // extern "C"
