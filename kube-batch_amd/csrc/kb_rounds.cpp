// kb_rounds.cpp — one device round in three host steps (prepare, candidate lists, commit + collect), chained and overlapped rounds, the action loop
// of allocate / backfill (run_action: plan, launch, absorb, re-plan) and the round-granular entry points of the task-row split (kb_round_*).
// Split out of kb_engine.cpp in round 6 without a change of behaviour (kb_engine_int.hpp has the map).
#include "kb_engine_int.hpp"

namespace kbe {

KbRound make_round(kb_engine *e, uint32_t n_rows, uint32_t n_mrows, uint32_t L, int fit_mode, bool backfill, uint32_t buf) {
  KbRound r{};
  r.rows = e->b_win.as<uint32_t>();
  r.shape_slot = e->b_win.as<uint32_t>() + KB_K5_MAX_WINDOW;
  r.n_rows = n_rows;
  r.desc = e->b_desc.as<KbRowDesc>() + (size_t)buf * KB_K5_MAX_WINDOW;
  r.trace = nullptr;
  r.cap = std::max<uint32_t>(64, ((n_rows + 63) / 64) * 64);
  r.mrows = e->b_win.as<uint32_t>() + 2 * KB_K5_MAX_WINDOW;
  r.mrow_task0 = 0;
  r.same_prev = nullptr;
  r.n_mrows = n_mrows;
  r.fit_mode = fit_mode;
  r.score = e->b_score.as<uint16_t>();
  r.maskw = e->b_maskw.as<uint32_t>();
  r.keys = e->b_keys.as<unsigned long long>();
  r.L = L;
  r.dec = e->b_out.as<unsigned long long>() + KB_OUT_HDR;   // header words, then the decision records
  r.result = e->b_out.as<uint32_t>();
  r.host_out = nullptr;
  r.seq = 0;
  r.backfill = backfill ? 1 : 0;
  r.batch = 0;
  r.gather = 0;
  r.delta = nullptr;
  r.own_row0 = r.own_row1 = 0;
  return r;
}

// distinct shapes of the window e->h_rows[0..n): fills h_slot (per row) and h_mrows (representative task per shape)
uint32_t assign_shapes(kb_engine *e, uint32_t n, const uint32_t *rows) {
  HostSession &hs = e->hs;
  if (!rows) rows = e->h_rows.data();
  if (e->shape_stamp.size() != hs.n_row_shapes) {
    e->shape_stamp.assign(hs.n_row_shapes, 0);
    e->shape_slot_of.assign(hs.n_row_shapes, 0);
    e->stamp = 0;
  }
  e->stamp++;
  uint32_t ns = 0;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t sh = hs.t_row_shape[rows[i]];
    if (e->shape_stamp[sh] != e->stamp) {
      e->shape_stamp[sh] = e->stamp;
      e->shape_slot_of[sh] = ns;
      e->h_mrows[ns] = rows[i];
      ns++;
    }
    e->h_slot[i] = e->shape_slot_of[sh];
  }
  return ns;
}

// upload the window e->h_rows[0..n) (task ids, shape slots, representative rows) and build the row descriptors
// `rows` (default e->h_rows) is the window; `buf` selects the half of the pinned staging blocks; a non-zero `chain_expect` queues
// the round behind a predecessor whose result the host has not seen yet (KbRound::chain)
RoundCtx round_prepare(kb_engine *e, uint32_t n, int fit_mode, bool backfill, bool gather_in_matrix, const uint32_t *rows, uint32_t buf, uint32_t chain_expect) {
  RoundCtx c;
  if (!rows) rows = e->h_rows.data();
  ensure_window_buffers(e, n);
  ensure_matrix_buffers(e, n, n + 1);
  c.n = n;
  c.ns = assign_shapes(e, n, rows);
  c.L = n + 1;   // more candidates than the round can dirty: a clean one always survives
  c.backfill = backfill;
  c.buf = buf;
  // one staging copy per round: [task rows | shape slots | representative rows], fixed offsets
  uint32_t *hw = e->h_win.data() + (size_t)buf * 3 * KB_K5_MAX_WINDOW;
  std::memcpy(hw, rows, sizeof(uint32_t) * n);
  std::memcpy(hw + KB_K5_MAX_WINDOW, e->h_slot.data(), sizeof(uint32_t) * n);
  std::memcpy(hw + 2 * KB_K5_MAX_WINDOW, e->h_mrows.data(), sizeof(uint32_t) * c.ns);
  // single-GPU fast rounds: the descriptor gather and the matrix kernel read the staged window straight from the pinned block
  // (a few hundred 4-byte reads over PCIe, overlapped with the matrix evaluation) instead of waiting for a 7 us copy command
  const bool direct = gather_in_matrix && e->fast_rounds;   // (the copied window is what the round API and KB_FLAG_SYNC_ROUNDS take: both tested)
  if (!direct) HIP_OK(hipMemcpyAsync(e->b_win.p, hw, sizeof(uint32_t) * (2 * KB_K5_MAX_WINDOW + c.ns), hipMemcpyHostToDevice, e->stream));
  c.d = e->dev;
  if (backfill) {
    c.d.score_enabled = 0;   // backfill.go:50-66 takes the first node that passes the predicates: all scores tie
    c.d.t_init = e->t_fit;   // ... and on which ssn.Allocate's AddTask succeeds: Resreq.LessEqual(Idle), fit_mode 2
  }
  c.r = make_round(e, n, c.ns, c.L, fit_mode, backfill, buf);
  if (direct) {
    const uint32_t *dw = e->d_hwin + (size_t)buf * 3 * KB_K5_MAX_WINDOW;
    c.r.rows = dw;
    c.r.shape_slot = dw + KB_K5_MAX_WINDOW;
    c.r.mrows = dw + 2 * KB_K5_MAX_WINDOW;
  }
  c.direct = direct;
  c.seq = ++e->seq;
  c.r.chain = e->b_chain.as<uint32_t>();
  c.r.chain_expect = chain_expect;
  c.r.chain_tag = (uint32_t)(c.seq & 0x7FFFFFFFull) + 1u;   // never 0
  c.r.gather = (gather_in_matrix && c.ns > 0) ? 1u : 0u;
  if (!c.r.gather) kb_launch_gather(c.d, c.r, e->stream);
  return c;
}

// K1 + K3 for matrix rows [m0, m1) of the round; keys go to `keys` (row m0 first)
void round_candidates(kb_engine *e, const RoundCtx &c, uint32_t m0, uint32_t m1, unsigned long long *keys) {
  if (m1 <= m0) return;
  KbRound r = c.r;
  r.mrows = c.r.mrows + m0;
  r.n_mrows = m1 - m0;
  r.keys = keys;
  if (e->fast_rounds) {   // kernel times come from the wall-clock stamps the kernels leave in the output block
    kb_launch_matrix(c.d, r, e->stream);
    kb_launch_affinity(c.d, r, e->stream);
    kb_launch_interpod(c.d, r, e->stream);
    kb_launch_argmax(c.d, r, e->stream);
  } else {
    Timer &t1 = get_timer(e, 0), &t3 = get_timer(e, 1);
    HIP_OK(hipEventRecord(t1.a, e->stream));
    kb_launch_matrix(c.d, r, e->stream);
    kb_launch_affinity(c.d, r, e->stream);
    kb_launch_interpod(c.d, r, e->stream);
    HIP_OK(hipEventRecord(t1.b, e->stream));
    HIP_OK(hipEventRecord(t3.a, e->stream));
    kb_launch_argmax(c.d, r, e->stream);
    HIP_OK(hipEventRecord(t3.b, e->stream));
  }
  e->stats.matrix_launches += 1;
  e->stats.matrix_evals += (uint64_t)(m1 - m0) * e->hs.N;
}

// The same for a chained round, overlapped with its predecessor (n_prev rows, running or queued on the first stream): matrix + arg-max on
// the second stream with lists of n_prev + L entries, then — first stream, i.e. behind the predecessor's commit kernel — the repair that
// waits for the lists, re-evaluates the predecessor's nodes and merges (kb_repair.hpp): workgroups of the selection kernel's own launch, or
// (the run kernel's rounds, KB_FUSE_REPAIR=0) a launch in front of it (kb_kernels.hip: k_repair).  The host's order guarantees that
// every round before the predecessor has been COLLECTED when this is called (run_action plans a window only after it has the answer of the
// round two in front of it), so the only nodes that can change under the second stream's launches are the predecessor's.
void ensure_overlap_buffers(kb_engine *e, uint32_t mrows, uint32_t stale_L) {
  const size_t NP = e->dev.NP;
  if (!e->stream_b) {
    HIP_OK(hipStreamCreateWithFlags(&e->stream_b, hipStreamNonBlocking));
  }
  if (mrows > e->mat2_cap) {
    HIP_OK(hipStreamSynchronize(e->stream_b));
    e->b_score2.alloc(sizeof(uint16_t) * (size_t)mrows * NP);
    e->b_maskw2.alloc(sizeof(uint32_t) * (size_t)mrows * (NP / 32));
    e->mat2_cap = mrows;
  }
  const size_t need = (size_t)mrows * stale_L;
  if (need > e->stale_cap) {
    HIP_OK(hipStreamSynchronize(e->stream_b));
    HIP_OK(hipStreamSynchronize(e->stream));
    e->b_stale.alloc(sizeof(unsigned long long) * 2 * need);
    e->stale_cap = need;
  }
  if (!e->b_ready.p) {
    e->b_ready.alloc(sizeof(uint32_t) * 2 * KB_K5_MAX_WINDOW);
    HIP_OK(hipMemset(e->b_ready.p, 0, e->b_ready.bytes));
    e->b_task_rows.alloc((size_t)64 * 2 * KB_K5_MAX_WINDOW);
    e->b_lready.alloc(sizeof(uint32_t) * 2 * KB_K5_MAX_WINDOW);
    HIP_OK(hipMemset(e->b_lready.p, 0, e->b_lready.bytes));
    e->h_cand_out.flags = hipHostMallocMapped | hipHostMallocCoherent;
    e->h_cand_out.resize(2 * KB_OUT_HDR);
    std::memset(e->h_cand_out.data(), 0, sizeof(unsigned long long) * 2 * KB_OUT_HDR);
    HIP_OK(hipHostGetDevicePointer((void **)&e->d_cand_out, e->h_cand_out.data(), 0));
  }
}
void round_candidates_overlapped(kb_engine *e, RoundCtx &c, uint32_t n_prev, unsigned long long *keys) {
  if (c.ns == 0) return;
  const uint32_t stale_L = n_prev + c.L;
  unsigned long long *stale = e->b_stale.as<unsigned long long>() + (size_t)c.buf * e->stale_cap;
  uint32_t *ready = e->b_ready.as<uint32_t>() + (size_t)c.buf * KB_K5_MAX_WINDOW;
  KbRound rb = c.r;   // the second stream's view: runs whatever the chain word says (the predecessor has not written it yet)
  rb.chain_expect = 0;
  rb.score = e->b_score2.as<uint16_t>();
  rb.maskw = e->b_maskw2.as<uint32_t>();
  rb.keys = stale;
  rb.L = stale_L;
  rb.result = reinterpret_cast<uint32_t *>(e->d_cand_out + (size_t)c.buf * KB_OUT_HDR);   // their time stamps, apart from the round's timeline (round_collect reads them)
  rb.ready = ready;
  rb.ready_tag = c.r.chain_tag;   // the round's own tag (sequence number folded to 31 bits, + 1): unique among the rounds in flight, never the 0 the words start from
  rb.task_rows = reinterpret_cast<unsigned char *>(e->b_task_rows.p) + (size_t)c.buf * 64 * KB_K5_MAX_WINDOW;
  kb_launch_matrix(c.d, rb, e->stream_b);        // also gathers the row descriptors into this round's half (gather == 1)
  kb_launch_argmax(c.d, rb, e->stream_b);
  KbRound ra = c.r;   // first stream: behind the predecessor's commit kernel
  ra.keys = keys;
  ra.ready = ready;
  ra.ready_tag = c.r.chain_tag;
  ra.task_rows = rb.task_rows;
  ra.stale = stale;
  ra.stale_L = stale_L;
  ra.prev_dec = e->b_out.as<unsigned long long>() + KB_OUT_HDR;   // the predecessor's decision records (it completed, or the chain is broken)
  ra.n_prev = n_prev;
  if (e->fuse_repair && e->commit_kernel == KB_COMMIT_SELECT) {
    // The selection kernel's launch carries the repair workgroups itself (kb_commit_sel.hip, kb_repair.hpp): they start with the commit
    // workgroup instead of a launch earlier — a kernel boundary, a launch latency and the commit prologue's staging off the dependent chain
    // of every round.  round_commit launches with these fields; this round's commit kernel overwrites prev_dec (the result block) in its
    // epilogue, i.e. behind its wait for the repaired lists.
    c.r.ready = ra.ready; c.r.ready_tag = ra.ready_tag; c.r.task_rows = ra.task_rows; c.r.stale = ra.stale; c.r.stale_L = ra.stale_L;
    c.r.prev_dec = ra.prev_dec; c.r.n_prev = ra.n_prev;
    c.r.lists_ready = e->b_lready.as<uint32_t>() + (size_t)c.buf * KB_K5_MAX_WINDOW;
    c.r.lists_tag = c.r.chain_tag;
  } else {
    kb_launch_repair(c.d, ra, e->stream);
  }
  e->stats.matrix_launches += 1;
  e->stats.matrix_evals += (uint64_t)c.ns * e->hs.N;
  e->overlapped_rounds += 1;
}

// K5 over the whole window with the complete candidate table `keys` [ns][L]
void round_commit(kb_engine *e, const RoundCtx &c, unsigned long long *keys, double *delta, uint32_t own0, uint32_t own1) {
  KbRound r = c.r;
  r.keys = keys;
  r.delta = delta;
  r.own_row0 = own0;
  r.own_row1 = own1;
  auto launch = [&]() {
    const int kern = e->commit_kernel;
    e->commit_kernel_of[c.buf] = kern;
    if (kern == KB_COMMIT_RUN) { kb_launch_commit(c.d, r, e->stream); e->rounds_run++; }
    else { kb_launch_commit_sel(c.d, r, e->stream); e->rounds_sel++; }
  };
  if (e->fast_rounds) {
    r.host_out = e->d_hout + (size_t)c.buf * KB_OUT_STRIDE;
    r.seq = c.seq;
    launch();
    // a launch the runtime refuses (too much LDS, a bad attribute) never publishes its round: say so now instead of after the watchdog's ten seconds
    HIP_OK(hipGetLastError());
    return;
  }
  Timer &t5 = get_timer(e, 2);
  HIP_OK(hipEventRecord(t5.a, e->stream));
  launch();
  HIP_OK(hipEventRecord(t5.b, e->stream));
  HIP_OK(hipMemcpyAsync(e->h_out.data(), e->b_out.p, sizeof(unsigned long long) * (KB_OUT_HDR + c.n), hipMemcpyDeviceToHost, e->stream));
}


// wait for the round, account the kernel times, unpack the decision records
void round_collect(kb_engine *e, const RoundCtx &c, bool had_candidates, uint32_t &n_done, uint32_t &reason) {
  const unsigned long long *ho = e->h_out.data() + (e->fast_rounds ? (size_t)c.buf * KB_OUT_STRIDE : 0);
  const uint32_t *h_result = reinterpret_cast<const uint32_t *>(ho);
  if (e->fast_rounds) {
    // the commit kernel publishes the round's sequence number into pinned host memory after everything else
    volatile const unsigned long long *seqw = ho + KB_OUT_SEQ;
    const double t0 = now_ms();
    uint32_t spins = 0;
    const bool spin_only = (e->flags & KB_FLAG_SPIN_WAIT) != 0;   // include/kb_engine.h: a short spin, then the core is given up between polls (default); or spin throughout
    bool yielding = false;
    while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != c.seq) {
      if (yielding) sched_yield(); else __builtin_ia32_pause();
      ++spins;
      if (!spin_only && !yielding && (spins & 0x3Fu) == 0 && now_ms() - t0 > KB_WAIT_SPIN_US * 1e-3) yielding = true;
      if ((spins & 0xFFFFu) == 0 && now_ms() - t0 > 10000.0) {   // a faulted kernel never publishes: surface the HIP error
        HIP_OK(hipStreamSynchronize(e->stream));
        HIP_OK(hipGetLastError());
        if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != c.seq) throw EngineError(KB_E_DEVICE, "commit kernel finished without publishing its round");
      }
    }
    n_done = h_result[0];
    reason = h_result[1];
    if (reason == KB_REASON_SKIPPED) return;   // queued behind a round that stopped early: nothing ran
    const unsigned long long *st = ho + KB_OUT_STAMP0;
    const double per_ms = 1.0 / e->wall_khz;
    if (had_candidates && c.overlapped) {
      // matrix: the launch itself, timed on the second stream (start of the matrix launch -> start of the arg-max launch behind it); it ran
      // beside the predecessor's commit kernel, i.e. NOT on the cycle's timeline.  arg-max: what the round waits for on the first stream
      // instead — the repair launch, from its start (its wait for the lists included) to the start of the commit kernel; a commit launch that
      // carries its own repair workgroups has that wait inside commit_ms
      const unsigned long long *cs = e->h_cand_out.data() + (size_t)c.buf * KB_OUT_HDR + KB_OUT_STAMP0;
      if (cs[1] > cs[0]) e->stats.matrix_ms += (double)(cs[1] - cs[0]) * per_ms;
      if (st[2] > st[0]) e->stats.argmax_ms += (double)(st[2] - st[0]) * per_ms;   // (a launch that carries its own repair workgroups: they start WITH the commit workgroup, the wait is inside commit_ms)
      if (st[1] > st[0]) e->tl_repair_tag += (double)(st[1] - st[0]) * per_ms;   // ... of which: until workgroup 0 had seen its list's tag
    } else if (had_candidates) {
      e->stats.matrix_ms += (double)(st[1] - st[0]) * per_ms;   // includes the descriptor gather
      e->stats.argmax_ms += (double)(st[2] - st[1]) * per_ms;
    }
    e->stats.commit_ms += (double)(st[3] - st[2]) * per_ms;
  } else {
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipGetLastError());
    float ms = 0;
    if (had_candidates) {
      HIP_OK(hipEventElapsedTime(&ms, get_timer(e, 0).a, get_timer(e, 0).b));
      e->stats.matrix_ms += ms;
      HIP_OK(hipEventElapsedTime(&ms, get_timer(e, 1).a, get_timer(e, 1).b));
      e->stats.argmax_ms += ms;
    }
    HIP_OK(hipEventElapsedTime(&ms, get_timer(e, 2).a, get_timer(e, 2).b));
    e->stats.commit_ms += ms;
  }
  for (uint32_t i = 0; i < c.n; i++) {
    e->h_decnode[i] = (uint32_t)(ho[KB_OUT_HDR + i] & 0xFFFFFFFFull);
    e->h_deckind[i] = (uint32_t)(ho[KB_OUT_HDR + i] >> 32);
  }
  n_done = h_result[0];
  reason = h_result[1];
  if (reason == KB_REASON_INTERNAL)   // only the selection kernel's bounded waits raise it (kb_commit_sel.hip: K9Sync::err): a hand-over between its waves never arrived
    throw EngineError(KB_E_INTERNAL, "selection commit kernel: a wave's bounded wait ran out (round " + std::to_string(e->round_no) + ", " + std::to_string(h_result[4]) +
                                     " runs and " + std::to_string(n_done) + " of " + std::to_string(c.n) + " rows committed before it)");
  const uint32_t dirty_won = h_result[3];   // rows won by a node the round had already changed
  e->stats.row_fallbacks += dirty_won;
  if (e->commit_kernel_of[c.buf] == KB_COMMIT_SELECT) {
    e->sel_stat[0] += h_result[6] & 0xFFFFu; e->sel_stat[1] += h_result[6] >> 16; e->sel_stat[2] += h_result[7] & 0xFFFFu; e->sel_stat[3] += h_result[7] >> 16;
    e->stats.rounds_select += 1;
    e->stats.select_runs_clean += h_result[6] & 0xFFFFu; e->stats.select_runs_shots += h_result[6] >> 16; e->stats.select_shots += h_result[7] >> 16;
  }
  {
    e->k5_slots += h_result[2];
    e->k5_walks += h_result[4];
    e->k5_rescans += h_result[5];
    for (int k = 0; k < 10; k++)   // zero unless built with -DKB_K9_TRACE
      e->k5_trace[k] += (double)(uint32_t)(ho[(k < 6 ? 5 + k / 2 : 13 + (k - 6) / 2)] >> (32 * (k & 1)));
    if (e->commit_kernel_of[c.buf] == KB_COMMIT_SELECT && e->k5_trace[0] > 0)   // the selection kernel's trace build: its other waves' evaluation phase
      for (int k = 10; k < 14; k++) e->k5_trace[k] += (double)(uint32_t)(ho[k < 12 ? 4 : 15] >> (32 * (k & 1)));
  }
  if (n_done) {
    const double share = (double)dirty_won / (double)n_done;
    e->dirty_share = e->stats.rounds == 0 ? share : 0.75 * e->dirty_share + 0.25 * share;
  }
  e->stats.rounds += 1;
  e->round_no += 1;
}

void check_aggregates(kb_engine *e, const OrderMachine &om) {
  // the host's running drf / proportion / gang aggregates must equal the device reduction bit for bit
  const HostSession &hs = e->hs;
  for (size_t i = 0; i < hs.job_ready.size(); i++)
    if (om.ready[i] != hs.job_ready[i]) throw EngineError(KB_E_INTERNAL, "gang ready count diverged from the device ballot at job " + std::to_string(i));
  if (e->pol.has_drf)
    for (size_t i = 0; i < hs.job_share.size(); i++)
      if (om.jshare[i] != hs.job_share[i]) throw EngineError(KB_E_INTERNAL, "drf share diverged from the device reduction at job " + std::to_string(i));
  if (e->pol.has_proportion)
    for (uint32_t q = 0; q < hs.Q; q++)
      if (hs.queue_has_attr[q] && om.qshare[q] != hs.queue_share[q])
        throw EngineError(KB_E_INTERNAL, "proportion share diverged from the device reduction at queue " + std::to_string(q));
}

}  // namespace kbe

void mg_free(MgState *m) { delete m; }
// kb_session_reset: the round-mode action state goes, its device buffers stay (ten node-state copies + the counter: a sharded cycle resets every
// step, and hipFree synchronises the device — inside the timed step, on the first rounds' critical path)
void mg_reset(MgState *m) {
  if (!m) return;
  m->run = ActionRun();
  m->in_round = false; m->committed = false; m->had_candidates = false;
  m->n_done = 0; m->reason = 0; m->rounds_begun = 0; m->chk_valid = false;
  m->last_decs.clear();
}

extern "C" {

// What allocate / backfill refuse before they touch anything — whichever way the action is entered (kb_run_allocate / kb_run_backfill, or the first
// kb_round_begin of an action on the task-row split: every rank would diverge alike there, so neither the delta cross-check nor the journal digest
// would notice).
static void action_entry_guards(kb_engine *e) {
  if (!e->loaded) throw EngineError(KB_E_STATE, "kb_session_load must precede kb_run_* / kb_round_begin");
  if (e->tainted) throw EngineError(KB_E_STATE, "a preempt / reclaim call failed after touching the session's state: kb_session_load or kb_session_reset first");
  // A Pending task that still carries a NodeName was un-pipelined by a discarded preempt statement (NodeInfo.RemoveTask never
  // clears it, api/node_info.go:217-243): the reference's AddTask then refuses every other node AFTER ssn.Allocate has flipped
  // the status (session.go:243 vs :255).  Not modelled: the stock action takes such a cycle (it cannot arise under the stock
  // action order, where preempt runs last).
  if (!e->stale_checked) {
    for (uint32_t t = 0; t < e->hs.T; t++)
      if (e->hs.t_status[t] == KB_TASK_PENDING && e->hs.t_node[t] != KB_NONE)
        throw EngineError(KB_E_UNSUPPORTED, "a Pending task carries a stale NodeName (un-pipelined by a discarded preempt statement)");
    e->stale_checked = true;   // allocate and backfill never create one
    if (e->pristine) e->load_clean = true;
  }
}

static int run_action(kb_engine *e, uint32_t action, kb_decision *out, uint64_t cap, uint64_t *n_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    action_entry_guards(e);
    const double t_act0 = now_ms();
    ActionRun run;
    run.begin(e, action);
    const double t_act1 = now_ms();
    run.probe_dead_shapes(e);   // shapes no node can take from the start (larger than every node, full classes) never cost a break
    const double t_act2 = now_ms();
    uint32_t n = run.plan(e);
    const double t_act3 = now_ms();
    if (action == 0) { e->tl_begin_parts[0] += t_act1 - t_act0; e->tl_begin_parts[1] += t_act2 - t_act1; e->tl_begin_parts[2] += t_act3 - t_act2; }
    ensure_matrix_buffers(e, e->eff_window, e->eff_window + 1);   // sized once: no reallocation under a round in flight
    // chained rounds of plain sessions (no score that is normalised over the feasible set, no inter-pod counters) build their candidate
    // lists beside the predecessor's commit kernel
    const bool overlap_ok = e->overlap && action == 0 && e->fast_rounds && !e->hs.has_affinity && !e->hs.has_interpod && 2 * e->eff_window + 1 <= 1024u &&
                            kb_repair_smem_bytes(e->dev.NP) <= 150u * 1024u;   // the repair launch's LDS (its node bitmap grows with the cluster)
    // The second stream is ordered behind nothing the first one holds: the copies kb_session_reset left queued there must have landed before
    // an overlapped launch reads the node state (the feasibility probe's read-back waits for them when it runs — it does not without the
    // predicates plugin, with KB_PROBE=0, or when every shape is dead; found on the emulated device with asynchronous streams)
    if (overlap_ok) quiesce(e);
    auto launch = [&](uint32_t rows_n, const uint32_t *rows, uint32_t buf, uint32_t chain_expect, uint32_t n_prev) {
      RoundCtx c = round_prepare(e, rows_n, action == 0 ? 1 : 2, action == 1, true, rows, buf, chain_expect);   // single GPU: every matrix row is local
      unsigned long long *keys = e->b_keys.as<unsigned long long>();
      if (overlap_ok && c.direct && chain_expect != 0 && !e->overlap_faults) {
        round_candidates_overlapped(e, c, n_prev, keys);
        c.overlapped = true;
      }
      else round_candidates(e, c, 0, c.ns, keys);
      round_commit(e, c, keys, nullptr, 0, 0);
      return c;
    };
    if (overlap_ok) ensure_overlap_buffers(e, e->eff_window, 2 * e->eff_window + 1);
    // Fast rounds return from the launch immediately.  The host uses the wait to speculate the NEXT window (assuming the one in
    // flight completes, which ~80 % do) and queues that round behind the running one right away: the device starts it the
    // moment the commit kernel ends instead of idling through a host round trip (~19 us per round).  A round that stops early
    // clears the chain word and the queued round skips itself (KbRound::chain).
    // (sessions with host-port masks of several words: one round at a time — a pod that reaches beyond word 0 changes node state from the host
    //  after its round, which a round already queued or overlapped would not see)
    const bool ahead = action == 0 && e->fast_rounds && !e->dev.port_xw;
    const bool chained = ahead && e->chain_rounds;
    uint32_t buf = 0;
    RoundCtx c{};
    if (n) c = launch(n, nullptr, buf, 0, 0);
    (action == 0 ? e->tl_begin : e->tl_backfill) += now_ms() - t_act0;
    // n_next: the window speculated behind the one in flight (planned: have_next; launched behind it: queued, its round cn).  n_next2: the
    // window behind THAT, planned while both are on the device (have_next2) and launched the moment the first one's answer is in — the
    // order machine's work for a window is then never between an answer and the launches that wait for it (ActionRun::plan_ahead).
    uint32_t n_next = 0, n_next2 = 0;
    bool have_next = false, have_next2 = false, queued = false;
    RoundCtx cn{};
    while (n) {
      uint32_t n_done = 0, reason = 0;
      if (ahead && !have_next) {
        n_next = run.plan_ahead(e);
        have_next = true;
        queued = chained && n_next > 0;
        if (queued) cn = launch(n_next, run.rows_next.data(), buf ^ 1u, c.r.chain_tag, n);
      }
      if (queued && !have_next2) { n_next2 = run.plan_ahead(e, true); have_next2 = true; }
      const double t_w0 = now_ms();
      round_collect(e, c, true, n_done, reason);
      const double t_b0 = now_ms();
      e->tl_wait += t_b0 - t_w0;
      // a break: the feasibility probe goes out before the host starts on the answer (ActionRun::probe_launch)
      const bool probe_early = ahead && reason != KB_REASON_DONE && reason != KB_REASON_RENORM;
      if (probe_early) run.probe_launch(e);
      try { run.absorb(e, n, n_done, reason); } catch (...) { run.probe_abandon(e); throw; }
      const double t_b1 = now_ms();
      if (ahead && reason == KB_REASON_DONE) {
        run.promote(e, n_next, have_next2);
        const uint32_t n_prev_rows = n;
        (void)n_prev_rows;
        n = n_next;
        if (queued) { c = cn; buf ^= 1u; }
        else if (n) c = launch(n, nullptr, buf, 0, 0);
        if (have_next2) {   // the window planned behind the queued one is the speculated one now: behind the round that has just become current
          n_next = n_next2;
          have_next2 = false;
          queued = chained && n_next > 0 && n > 0;
          if (queued) cn = launch(n_next, run.rows_next.data(), buf ^ 1u, c.r.chain_tag, n);
        } else {
          have_next = false;
          queued = false;
        }
      } else {
        if (probe_early) run.probe_collect(e);
        else if (reason != KB_REASON_RENORM) run.probe_dead_shapes(e);
        const double t_b2 = now_ms();
        n = run.plan(e);   // re-plan first: the queued round drains (three empty launches) while the host works
        if (action == 0) { e->tl_break_parts[0] += t_b1 - t_b0; e->tl_break_parts[1] += t_b2 - t_b1; e->tl_break_parts[2] += now_ms() - t_b2; }
        if (queued) {   // the queued round skipped itself: consume its publication before its staging half is reused
          uint32_t nd2 = 0, rs2 = 0;
          round_collect(e, cn, true, nd2, rs2);
          if (rs2 != KB_REASON_SKIPPED) throw EngineError(KB_E_INTERNAL, "a round queued behind a stopped round ran");
          e->stats.matrix_launches -= 1;
          e->stats.matrix_evals -= (uint64_t)cn.ns * e->hs.N;
          // its matrix / arg-max launches on the second stream (they run whatever the chain word says) read the descriptor and window
          // halves the re-planned rounds are about to rewrite from the first stream: nothing else orders the two
          if (cn.overlapped && e->stream_b) HIP_OK(hipStreamSynchronize(e->stream_b));
        }
        have_next = have_next2 = queued = false;   // (absorb rolled every window back)
        if (n) c = launch(n, nullptr, buf, 0, 0);
        if (action == 0) e->tl_break += now_ms() - t_b0;
      }
    }
    if (e->stream_b) HIP_OK(hipStreamSynchronize(e->stream_b));   // a candidate launch of a round that was skipped may still be running
    run.finish(e);
    if (n_out) *n_out = run.decs.size();
    if (run.decs.size() > cap) throw EngineError(KB_E_CAPACITY, "decision buffer too small");
    if (out && !run.decs.empty()) std::memcpy(out, run.decs.data(), sizeof(kb_decision) * run.decs.size());
  });
}

int kb_run_allocate(kb_engine *e, kb_decision *out, uint64_t cap, uint64_t *n_out) { return run_action(e, 0, out, cap, n_out); }
int kb_run_backfill(kb_engine *e, kb_decision *out, uint64_t cap, uint64_t *n_out) { return run_action(e, 1, out, cap, n_out); }

// ---- round-granular API for task-row sharding across GPUs ----
int kb_round_begin(kb_engine *e, uint32_t action, uint32_t *n_rows, uint32_t *n_mrows, uint32_t *list_len) {
  if (!e || !n_rows) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    if (action > 1) throw EngineError(KB_E_INVALID, "action must be 0 (allocate) or 1 (backfill)");
    if (!e->mg) e->mg = new MgState();
    MgState &m = *e->mg;
    if (!m.run.active) {
      action_entry_guards(e);   // (the same refusals as kb_run_allocate / kb_run_backfill: a tainted session, a stale NodeName behind a discarded statement)
      m.run.begin(e, action);
      m.run.probe_dead_shapes(e);   // as run_action does: shapes no node can take from the start never cost a round (every replica reads the same state: the same answer)
      m.in_round = false;
      m.rounds_begun = 0;
      m.chk_counter.alloc(sizeof(uint32_t));
      HIP_OK(hipMemsetAsync(m.chk_counter.p, 0, sizeof(uint32_t), e->stream));
      m.chk_valid = true;
    } else if (m.run.action != action) {
      throw EngineError(KB_E_STATE, "another action is still in progress");
    }
    if (m.in_round) throw EngineError(KB_E_STATE, "previous round not applied");
    uint32_t n = m.run.plan(e);
    *n_rows = n;
    if (n_mrows) *n_mrows = 0;
    if (list_len) *list_len = 0;
    if (n == 0) {   // action complete: gang ballot + share reduction, consistency checks
      m.last_decs = m.run.decs;
      m.run.finish(e);
      return;
    }
    m.ctx = round_prepare(e, n, action == 0 ? 1 : 2, action == 1);
    m.had_candidates = false;
    m.in_round = true;
    // round-start copy of the node state: the reduced deltas are applied to it.  The copy of the round before stays (kb_round_check compares
    // that round's reduced deltas against the two of them, one round late)
    const size_t NP = e->dev.NP;
    const int R = e->hs.R;
    m.q_idle.swap(m.s_idle); m.q_rel.swap(m.s_rel); m.q_nzc.swap(m.s_nzc); m.q_nzm.swap(m.s_nzm); m.q_podcnt.swap(m.s_podcnt);
    m.rounds_begun += 1;
    auto snap = [&](DevBuf &dst, const DevBuf &src) {
      if (dst.bytes != src.bytes) dst.alloc(src.bytes);
      HIP_OK(hipMemcpyAsync(dst.p, src.p, src.bytes, hipMemcpyDeviceToDevice, e->stream));
    };
    snap(m.s_idle, e->b_idle); snap(m.s_rel, e->b_rel); snap(m.s_nzc, e->b_nzc); snap(m.s_nzm, e->b_nzm); snap(m.s_podcnt, e->b_podcnt);
    (void)NP; (void)R;
    if (n_mrows) *n_mrows = m.ctx.ns;
    if (list_len) *list_len = m.ctx.L;
  });
}

int kb_round_candidates(kb_engine *e, uint32_t mrow0, uint32_t mrow1, uint64_t dev_keys_ptr) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !e->mg->in_round) throw EngineError(KB_E_STATE, "kb_round_begin must precede kb_round_candidates");
    MgState &m = *e->mg;
    if (mrow1 > m.ctx.ns) mrow1 = m.ctx.ns;
    if (mrow0 >= mrow1) return;   // this rank's shard is empty (fewer shapes than ranks)
    if (!dev_keys_ptr) throw EngineError(KB_E_INVALID, "null key buffer");
    round_candidates(e, m.ctx, mrow0, mrow1, reinterpret_cast<unsigned long long *>(dev_keys_ptr));
    m.had_candidates = true;
    if (e->stream == e->own_stream) {   // the caller's collective runs on another stream: it must see finished lists
      HIP_OK(hipStreamSynchronize(e->stream));
      HIP_OK(hipGetLastError());
    }
  });
}

int kb_round_commit(kb_engine *e, uint64_t dev_all_keys_ptr, uint32_t own_row0, uint32_t own_row1, uint64_t dev_delta_ptr) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !e->mg->in_round) throw EngineError(KB_E_STATE, "kb_round_begin must precede kb_round_commit");
    MgState &m = *e->mg;
    if (!dev_all_keys_ptr) throw EngineError(KB_E_INVALID, "null key table");
    double *delta = reinterpret_cast<double *>(dev_delta_ptr);
    if (delta) HIP_OK(hipMemsetAsync(delta, 0, sizeof(double) * (size_t)e->dev.NP * (2 * (size_t)e->hs.R + 3), e->stream));
    round_commit(e, m.ctx, reinterpret_cast<unsigned long long *>(dev_all_keys_ptr), delta, own_row0, own_row1);
    round_collect(e, m.ctx, m.had_candidates, m.n_done, m.reason);
    m.committed = true;
  });
}

int kb_round_apply(kb_engine *e, uint64_t dev_delta_ptr, uint32_t *done) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !e->mg->in_round || !e->mg->committed) throw EngineError(KB_E_STATE, "kb_round_commit must precede kb_round_apply");
    MgState &m = *e->mg;
    if (dev_delta_ptr) {
      // node state for the next round = round-start state + all-reduced deltas; it must equal this replica's own commit
      uint32_t mism = kb_apply_deltas(e->dev, m.s_idle.as<double>(), m.s_rel.as<double>(), m.s_nzc.as<long long>(), m.s_nzm.as<long long>(),
                                      m.s_podcnt.as<int>(), reinterpret_cast<const double *>(dev_delta_ptr), e->b_out.as<uint32_t>() + 8, e->stream);   // word [4] of the output block
      if (mism) throw EngineError(KB_E_INTERNAL, "replicas diverged: reduced per-node deltas differ from the local commit at " + std::to_string(mism) + " values");
    }
    m.run.absorb(e, m.ctx.n, m.n_done, m.reason);
    // a speculation break: the feasibility probe marks every shape that died with the one that broke the round (run_action's rule; without it the
    // split paid one round per dead shape — 61 breaks per 100k x 10k cycle against the single-GPU path's 14, round 5)
    if (m.reason != KB_REASON_DONE && m.reason != KB_REASON_RENORM) m.run.probe_dead_shapes(e);
    m.in_round = false;
    m.committed = false;
    if (done) *done = 0;
  });
}

int kb_round_check(kb_engine *e, uint64_t dev_delta_ptr, uint32_t against_live) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !dev_delta_ptr) throw EngineError(KB_E_STATE, "kb_round_check: no round-mode action / null delta buffer");
    MgState &m = *e->mg;
    const double *delta = reinterpret_cast<const double *>(dev_delta_ptr);
    const KbDev &d = e->dev;
    if (against_live) {   // the action's last round: its start copy + deltas == the live state (the kb_round_begin that ended the action took no copy)
      if (m.in_round || m.rounds_begun < 1) throw EngineError(KB_E_STATE, "kb_round_check(against_live): behind the kb_round_begin that ended the action");
      kb_check_deltas(d, m.cur(), KbNodeCopy{d.idle, d.rel, d.nzc, d.nzm, d.podcnt}, delta, m.chk_counter.as<uint32_t>(), e->stream);
    } else {              // round k's deltas, round k + 1 begun: the two start copies
      if (!m.in_round || m.rounds_begun < 2) throw EngineError(KB_E_STATE, "kb_round_check: behind the kb_round_begin of the NEXT round");
      kb_check_deltas(d, m.prev(), m.cur(), delta, m.chk_counter.as<uint32_t>(), e->stream);
    }
  });
}

int kb_round_check_result(kb_engine *e, uint32_t *mismatches) {
  if (!e || !mismatches) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !e->mg->chk_valid) throw EngineError(KB_E_STATE, "kb_round_check_result: no round-mode action has run");
    uint32_t h = 0;
    HIP_OK(hipMemcpyAsync(&h, e->mg->chk_counter.p, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    *mismatches = h;
  });
}

int kb_round_decisions(kb_engine *e, kb_decision *out, uint64_t cap, uint64_t *n_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg) throw EngineError(KB_E_STATE, "no round-mode action has run");
    const std::vector<kb_decision> &d = e->mg->run.active ? e->mg->run.decs : e->mg->last_decs;
    if (n_out) *n_out = d.size();
    if (d.size() > cap) throw EngineError(KB_E_CAPACITY, "decision buffer too small");
    if (out && !d.empty()) std::memcpy(out, d.data(), sizeof(kb_decision) * d.size());
  });
}

}  // extern "C"
