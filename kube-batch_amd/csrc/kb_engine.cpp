// kb_engine.cpp — the C ABI of include/kb_engine.h over the HIP kernels of kb_kernels.hip: the engine object (create / destroy), its buffers and
// timers, the closing reduction, the getters.  The other entry points: kb_load.cpp, kb_rounds.cpp, kb_evict.cpp, kb_matrix.cpp (kb_engine_int.hpp).
//
// Round structure: the host order machine speculates the reference's task order for a window of W
// tasks (assuming each gets a node, which only ever fails when a whole feasibility class has died — and that is
// monotone inside one action), the device evaluates the window's mask+score matrix against the round-start node
// state (K1, once per distinct task shape), builds each shape's sorted candidate list (K3) and commits the window in the
// reference's order, a run of same-shape rows at a time (K5).  A mis-speculation (no feasible node / Pipeline instead of Allocate) stops the commit kernel at that row;
// the host rolls the order machine back to the round start, replays the confirmed prefix and re-plans.
#include "kb_engine_int.hpp"

thread_local std::string g_create_err;

namespace kbe {

// window buffers: one entry per task row of a round
void ensure_window_buffers(kb_engine *e, uint32_t rows) {
  if (rows <= e->win_cap) return;
  if (!e->b_desc.p) e->b_desc.alloc(sizeof(KbRowDesc) * 2 * KB_K5_MAX_WINDOW);   // two fixed halves (chained rounds alternate): a round in flight must not see them move
  e->h_rows.resize(rows);
  e->h_slot.resize(rows);
  e->h_decnode.resize(rows);
  e->h_deckind.resize(rows);
  e->win_cap = rows;
}
// scratch rows of the inter-pod priority kernel (per matrix row: per-node counts, per-domain sums)
void ensure_ip_scratch(kb_engine *e, size_t rows) {
  if (!e->hs.has_interpod) return;
  const size_t NP = e->dev.NP;
  if (e->b_ip_scnt.bytes < sizeof(long long) * rows * NP) {
    e->b_ip_scnt.alloc(sizeof(long long) * rows * NP);
    e->b_ip_shist.alloc(sizeof(int32_t) * rows * NP);
  }
  e->dev.ip_scratch_cnt = e->b_ip_scnt.as<long long>();
  e->dev.ip_scratch_hist = e->b_ip_shist.as<int32_t>();
}
// matrix buffers: one matrix row per distinct shape (or per task row for kb_eval_matrix), L candidate keys per row
void ensure_matrix_buffers(kb_engine *e, uint32_t mrows, uint32_t L) {
  const size_t NP = e->dev.NP;
  if (mrows > e->mat_cap) {
    e->b_mrows.alloc(sizeof(uint32_t) * mrows);
    e->b_same.alloc(mrows);
    e->b_score.alloc(sizeof(uint16_t) * (size_t)mrows * NP);
    e->b_maskw.alloc(sizeof(uint32_t) * (size_t)mrows * (NP / 32));
    e->h_mrows.resize(mrows);
    e->h_same.resize(mrows);
    e->mat_cap = mrows;
  }
  ensure_ip_scratch(e, std::max<uint32_t>(mrows, e->mat_cap));
  size_t need = (size_t)mrows * L;
  if (need > e->keys_cap) {
    e->b_keys.alloc(sizeof(unsigned long long) * need);
    e->keys_cap = need;
  }
}

Timer &get_timer(kb_engine *e, size_t i) {
  while (e->ev.size() <= i) {
    Timer t;
    t.init();
    e->ev.push_back(t);
  }
  return e->ev[i];
}

// the engine's own stream is non-blocking: a null-stream hipMemcpy does not wait for what kb_session_reset left queued on it
void quiesce(kb_engine *e) {
  if (!e->async_pending) return;
  HIP_OK(hipStreamSynchronize(e->stream));
  e->async_pending = false;
}

// gang ballot + share reduction on the device, results mirrored to the host session
void run_finalize(kb_engine *e, const std::function<void()> &after_sync) {
  Timer &tm = get_timer(e, 3);
  HIP_OK(hipEventRecord(tm.a, e->stream));
  kb_launch_finalize(e->dev, e->b_jbegin.as<uint32_t>(), e->b_jmin.as<int>(), e->b_jqueue.as<uint32_t>(), e->pol.gang_job_ready ? 1 : 0,
                     e->b_total.as<double>(), e->total_mask, e->b_deserved.as<double>(), e->b_desmask.as<uint32_t>(),
                     e->b_jalloc.as<double>(), e->b_jshare.as<double>(), e->b_qalloc.as<double>(), e->b_qshare.as<double>(),
                     e->b_jready.as<int>(), e->stream);
  HIP_OK(hipEventRecord(tm.b, e->stream));
  HostSession &hs = e->hs;
  // seven results, one pinned block: the copies queue behind the kernels as DMA commands (a pageable target makes every one of them a
  // staged, blocking copy), ONE synchronisation, then plain memcpys into the host session's vectors
  struct Part { void *dst; const void *src; size_t bytes, off; };
  Part parts[7] = {{hs.job_alloc.data(), e->b_jalloc.p, sizeof(double) * hs.job_alloc.size(), 0}, {hs.job_share.data(), e->b_jshare.p, sizeof(double) * hs.job_share.size(), 0},
                   {hs.queue_alloc.data(), e->b_qalloc.p, sizeof(double) * hs.queue_alloc.size(), 0}, {hs.queue_share.data(), e->b_qshare.p, sizeof(double) * hs.queue_share.size(), 0},
                   {hs.job_ready.data(), e->b_jready.p, sizeof(int32_t) * hs.job_ready.size(), 0}, {hs.t_status.data(), e->b_tstatus.p, hs.T, 0},
                   {hs.t_node.data(), e->b_tnode.p, sizeof(uint32_t) * hs.T, 0}};
  size_t total = 0;
  for (Part &pt : parts) { pt.off = total; total += (pt.bytes + 63) & ~(size_t)63; }
  e->h_fin.resize(total ? total : 64);
  for (const Part &pt : parts)
    if (pt.bytes) HIP_OK(hipMemcpyAsync(e->h_fin.data() + pt.off, pt.src, pt.bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_OK(hipStreamSynchronize(e->stream));
  e->async_pending = false;
  if (after_sync) after_sync();   // kb_session_load: what else came back behind this synchronisation (the water-fill's results)
  for (const Part &pt : parts)
    if (pt.bytes) std::memcpy(pt.dst, e->h_fin.data() + pt.off, pt.bytes);
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, tm.a, tm.b));
  e->stats.reduce_ms += ms;
  // queues without a job in the session keep share 0 (no queueOpts entry), and so do queues updateShare never ran for
  // (water-fill loop skipped on total weight 0 and no event yet: HostSession::queue_share_live)
  for (uint32_t q = 0; q < hs.Q; q++)
    if (!hs.queue_has_attr[q] || !hs.queue_share_live[q]) hs.queue_share[q] = 0.0;
}

int guarded(kb_engine *e, const std::function<void()> &fn) {
  try {
    if (e) HIP_OK(hipSetDevice(e->device));
    fn();
    return KB_OK;
  } catch (const EngineError &ex) {
    if (e) e->err = ex.what(); else g_create_err = ex.what();
    return ex.code;
  } catch (const std::bad_alloc &) {
    if (e) e->err = "out of host memory"; else g_create_err = "out of host memory";
    return KB_E_NOMEM;
  } catch (const std::exception &ex) {
    if (e) e->err = ex.what(); else g_create_err = ex.what();
    return KB_E_INTERNAL;
  } catch (...) {
    if (e) e->err = "unknown error"; else g_create_err = "unknown error";
    return KB_E_INTERNAL;
  }
}

}  // namespace kbe

extern "C" {

const char *kb_last_error(const kb_engine *e) { return e ? e->err.c_str() : g_create_err.c_str(); }

int kb_engine_create(const kb_config *cfg, kb_engine **out) {
  if (!cfg || !out) { g_create_err = "null argument"; return KB_E_INVALID; }
  *out = nullptr;
  kb_engine *e = nullptr;
  int rc = guarded(nullptr, [&]() {
    if (cfg->version != KB_ABI_VERSION) throw EngineError(KB_E_INVALID, "ABI version mismatch");
    std::unique_ptr<kb_engine> eng(new kb_engine());
    eng->pol = compile_policy(cfg);
    eng->device = cfg->device;
    if (cfg->window) eng->window = cfg->window;
    if (eng->window > KB_K5_MAX_ROWS) eng->window = KB_K5_MAX_ROWS;   // one thread of the commit kernel per dirty slot
    eng->commit_batch = cfg->commit_batch;
    eng->flags = cfg->flags;
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0)
      throw EngineError(KB_E_DEVICE, "no HIP device visible: the engine has no CPU fallback (the stock Go action must handle this cycle)");
    if (cfg->device < 0 || cfg->device >= ndev) throw EngineError(KB_E_INVALID, "device ordinal out of range");
    HIP_OK(hipSetDevice(cfg->device));
    HIP_OK(hipStreamCreateWithFlags(&eng->own_stream, hipStreamNonBlocking));
    eng->stream = eng->own_stream;
    eng->b_win.alloc(sizeof(uint32_t) * 3 * KB_K5_MAX_WINDOW);
    eng->h_win.flags = hipHostMallocMapped | hipHostMallocCoherent;   // read by the device directly in single-GPU fast rounds
    eng->h_win.resize(2 * 3 * KB_K5_MAX_WINDOW);   // two halves: a chained round is staged while its predecessor's copy may still be pending
    eng->b_out.alloc(sizeof(unsigned long long) * (KB_OUT_HDR + KB_K5_MAX_WINDOW));
    HIP_OK(hipMemset(eng->b_out.p, 0, eng->b_out.bytes));
    eng->h_out.flags = hipHostMallocMapped | hipHostMallocCoherent;   // written by the commit kernel while the host polls
    eng->h_probe_rows.flags = eng->h_probe_alive.flags = hipHostMallocMapped | hipHostMallocCoherent;   // the probe kernel's rows in / flags out
    eng->h_out.resize(2 * KB_OUT_STRIDE);
    std::memset(eng->h_out.data(), 0, sizeof(unsigned long long) * 2 * KB_OUT_STRIDE);
    eng->b_chain.alloc(sizeof(uint32_t));
    HIP_OK(hipMemsetAsync(eng->b_chain.p, 0, sizeof(uint32_t), eng->stream));
    HIP_OK(hipHostGetDevicePointer((void **)&eng->d_hout, eng->h_out.data(), 0));
    HIP_OK(hipHostGetDevicePointer((void **)&eng->d_hwin, eng->h_win.data(), 0));
    {
      int khz = 0;
      if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, eng->device) == hipSuccess && khz > 0) eng->wall_khz = (double)khz;
      if (const char *ck = getenv("KB_COMMIT_KERNEL")) {
        if (ck[0] == 'r') eng->commit_pin = KB_COMMIT_RUN;
        else if (ck[0] == 's') eng->commit_pin = KB_COMMIT_SELECT;
        if (eng->commit_pin >= 0) eng->commit_kernel = eng->commit_pin;
      }
      eng->fast_rounds = !(eng->flags & KB_FLAG_SYNC_ROUNDS);   // (the environment form KB_SYNC_ROUNDS went in round 6: the flag is the switch, tests/test_gpu_parity.py runs it)
      const char *cr = getenv("KB_CHAIN_ROUNDS");   // 0: launch every round only after the previous one was collected (A/B, debugging)
      eng->chain_rounds = !(cr && cr[0] == '0');
      const char *pb = getenv("KB_PROBE");
      eng->probe_enabled = !(pb && pb[0] == '0');
      const char *ov = getenv("KB_OVERLAP");
      eng->overlap = !(ov && ov[0] == '0');
      const char *fr = getenv("KB_FUSE_REPAIR");   // 0: the repair launch of its own between two commit kernels (A/B; the run kernel's rounds always take it)
      eng->fuse_repair = !(fr && fr[0] == '0');
      const char *wf = getenv("KB_DEVICE_WATERFILL");
      eng->device_waterfill = !(wf && wf[0] == '0');
    }
    e = eng.release();
  });
  if (rc == KB_OK) *out = e;
  return rc;
}

void kb_engine_destroy(kb_engine *e) {
  if (!e) return;
  if (getenv("KB_K5_STATS"))
    fprintf(stderr, "[kb K5] rounds on the run kernel %llu, on the selection kernel %llu, last dirty share %.3f\n",
            (unsigned long long)e->rounds_run, (unsigned long long)e->rounds_sel, e->dirty_share);
  if (getenv("KB_K5_STATS") && e->rounds_sel)
    fprintf(stderr, "[kb select] runs of >= 2 rows: all picks clean first placements %llu, committed by shots %llu (shots cut short by a table's end %llu; shots %llu)\n",
            (unsigned long long)e->sel_stat[0], (unsigned long long)e->sel_stat[1], (unsigned long long)e->sel_stat[2], (unsigned long long)e->sel_stat[3]);
  if (getenv("KB_K5_STATS"))
    fprintf(stderr, "[kb K5] rounds %llu, rows %llu, dirty slots %llu, dirty-won rows %llu, runs %llu, runs with a row-specific Resreq %llu (%llu)\n",
            (unsigned long long)e->stats.rounds, (unsigned long long)e->stats.decisions, (unsigned long long)e->k5_slots,
            (unsigned long long)e->stats.row_fallbacks, (unsigned long long)e->k5_walks, (unsigned long long)e->k5_rescans,
            (unsigned long long)e->k5_demand);
  if (getenv("KB_K5_STATS")) fprintf(stderr, "[kb overlap] rounds with candidate lists built beside the predecessor's commit %llu, lists that never arrived %llu; repair launches: %.3f ms from their start to the tag seen\n",
                                     (unsigned long long)e->overlapped_rounds, (unsigned long long)e->overlap_faults, e->tl_repair_tag);
  if (getenv("KB_K5_STATS"))
    fprintf(stderr, "[kb host] ms over the engine's life: reset %.2f, allocate up to its first launch %.2f, speculation breaks (answer -> re-planned launch) %.2f, "
            "closing reductions %.2f, waiting for rounds %.2f, backfill up to its first launch %.2f\n", e->tl_reset, e->tl_begin, e->tl_break, e->tl_finish, e->tl_wait, e->tl_backfill);
  if (getenv("KB_K5_STATS"))
    fprintf(stderr, "[kb host] of the start: order machine %.2f, first probe %.2f, first plan %.2f; of the breaks: probe launch + absorb %.2f, waiting for the probe %.2f, re-plan %.2f\n",
            e->tl_begin_parts[0], e->tl_begin_parts[1], e->tl_begin_parts[2], e->tl_break_parts[0], e->tl_break_parts[1], e->tl_break_parts[2]);
  if (getenv("KB_K5_STATS")) fprintf(stderr, "[kb probe] %llu probes, %llu shapes marked dead by them\n", (unsigned long long)e->probes, (unsigned long long)e->probe_deaths);
  if (getenv("KB_K5_STATS") && e->k5_trace[0] > 0) {
    static const char *ph[10] = {"barrier 1 (wave 0's wait)", "evaluate: candidates (wave 0)", "barrier 2", "rows", "prepare the next run",
                                "rows, of runs with scalar dimensions", "runs with scalar dimensions (count)", "evaluate, of runs with scalar dimensions",
                                "dirty-winner entries (count)", "loop top"};
    static const char *ph_sel[10] = {"wave 0 waits for candidates + dirty keys", "staging: lists, descriptors (10 ns units, not clocks)", "run tables (10 ns units)", "rows: serial loop, tail, publish", "wave 0's loop (10 ns units)",
                                    "selection: entries + first rank", "selection: deep passes", "selection: picks + AddTask", "selection: all picks clean", "shape tables (10 ns units)"};
    const double runs = (double)(e->k5_walks ? e->k5_walks : 1);
    for (int k = 0; k < 10; k++) fprintf(stderr, "[kb K5 trace] %-32s %14.0f clocks  (%.0f per run)\n", e->rounds_sel > e->rounds_run ? ph_sel[k] : ph[k], e->k5_trace[k], e->k5_trace[k] / runs);
    static const char *ph_role[4] = {"prep waves wait for their run's turn (sum of 3)", "prep waves wait for the walk's turn (sum of 3)", "wave 1 waits for the previous run", "wave 1 evaluates + publishes"};
    if (e->rounds_sel > e->rounds_run) for (int k = 10; k < 14; k++) fprintf(stderr, "[kb K5 trace] %-32s %14.0f clocks  (%.0f per run)\n", ph_role[k - 10], e->k5_trace[k], e->k5_trace[k] / runs);
  }
  (void)hipSetDevice(e->device);
  delete e;
}

int kb_get_binds(kb_engine *e, uint32_t *task_node_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    quiesce(e);
    HIP_OK(hipMemcpy(task_node_out, e->b_tbind.p, sizeof(uint32_t) * e->hs.T, hipMemcpyDeviceToHost));
    uint64_t nb = 0;
    for (uint32_t t = 0; t < e->hs.T; t++) nb += task_node_out[t] != KB_NONE;
    e->stats.binds = nb;
  });
}

int kb_get_task_state(kb_engine *e, uint8_t *status_out, uint32_t *node_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    quiesce(e);
    if (status_out) HIP_OK(hipMemcpy(status_out, e->b_tstatus.p, e->hs.T, hipMemcpyDeviceToHost));
    if (node_out) HIP_OK(hipMemcpy(node_out, e->b_tnode.p, sizeof(uint32_t) * e->hs.T, hipMemcpyDeviceToHost));
  });
}

int kb_get_node_state(kb_engine *e, double *idle, double *releasing, int64_t *nz_cpu, int64_t *nz_mem, int32_t *pod_cnt) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    quiesce(e);
    const uint32_t N = e->hs.N, NP = e->dev.NP;
    const int R = e->hs.R;
    if (idle) HIP_OK(hipMemcpy2D(idle, sizeof(double) * N, e->b_idle.p, sizeof(double) * NP, sizeof(double) * N, R, hipMemcpyDeviceToHost));
    if (releasing) HIP_OK(hipMemcpy2D(releasing, sizeof(double) * N, e->b_rel.p, sizeof(double) * NP, sizeof(double) * N, R, hipMemcpyDeviceToHost));
    if (nz_cpu) HIP_OK(hipMemcpy(nz_cpu, e->b_nzc.p, sizeof(int64_t) * N, hipMemcpyDeviceToHost));
    if (nz_mem) HIP_OK(hipMemcpy(nz_mem, e->b_nzm.p, sizeof(int64_t) * N, hipMemcpyDeviceToHost));
    if (pod_cnt) HIP_OK(hipMemcpy(pod_cnt, e->b_podcnt.p, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
  });
}

int kb_get_shares(kb_engine *e, double *job_share, double *queue_share, double *queue_deserved) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    const HostSession &hs = e->hs;
    if (job_share) std::memcpy(job_share, hs.job_share.data(), sizeof(double) * hs.J);
    if (queue_share) std::memcpy(queue_share, hs.queue_share.data(), sizeof(double) * hs.Q);
    if (queue_deserved)
      for (uint32_t q = 0; q < hs.Q; q++)
        for (int d = 0; d < hs.R; d++) queue_deserved[(size_t)d * hs.Q + q] = hs.deserved[q].get(d);
  });
}

int kb_get_stats(kb_engine *e, kb_stats *out) {
  if (!e || !out) return KB_E_INVALID;
  *out = e->stats;
  return KB_OK;
}

int kb_engine_use_stream(kb_engine *e, uint64_t hip_stream) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    HIP_OK(hipStreamSynchronize(e->stream));
    e->async_pending = false;
    e->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : e->own_stream;
  });
}

int kb_round_delta_doubles(const kb_engine *e, uint64_t *n_doubles) {
  if (!e || !n_doubles) return KB_E_INVALID;
  *n_doubles = (uint64_t)e->dev.NP * (2ull * e->hs.R + 3ull);
  return KB_OK;
}

}  // extern "C"
