// kb_commit_sel.hip — K5, the run SELECTION commit of one window (gfx950 / CDNA4, wave64; KB_COMMIT_SELECT).
//
// Same contract and same decisions as kb_commit.hip (the reference's sequential placement, allocate.go:129-193 / backfill.go:44-67, a
// run of same-shape rows at a time), two differences in how a run is worked:
//
// 1. The rows of a run are ONE selection instead of a loop over the rows.  Every node has its own key sequence key(n, j): its key for the
//    shape after j placements of the shape, while the next placement still passes the predicates (api/node_info.go:172-212 applied j
//    times).  It depends on that node's state only.  With eff(n, j) = min key(n, 0..j), the serial loop's picks are the first r of all
//    (n, j) entries in (eff descending, j ascending) order — keys carry the node index, so entries of different nodes never tie: a node
//    picked at key s was the maximum with the lowest index among equals (util.SelectBestNode, scheduler_helper.go:188-208, with the
//    canonical tie-break); while its next keys stay >= s it wins again at once (nobody else changed) — the steps on which eff stays s —
//    and when its key falls below s the prefix minimum is the real key again.  A Pipeline entry (allocate.go:160: InitResreq no longer
//    fits Idle) ends its node's sequence; picked, it ends the round behind its row.  Most runs need no table at all: every pick is a clean
//    candidate's first placement (the all-clean test in wave 0's loop).  The others are committed by SHOTS (round 6): the rem best
//    contenders by current key — the distinct winners of rem rows are among them —, for each its next D keys evaluated in parallel by the
//    64 lanes (D = lanes / contenders; the best one gets sixteen), prefix minima by DPP, one rank through LDS, and of that order the
//    picks that are FINAL: above the floor (the best key outside the table) and in front of the first table that ends while its sequence
//    may go on.  What is not final yet is the next shot's, against the node states the committed picks left.  Placements are evaluated as
//    Idle - u * Resreq, which is u subtractions for whole numbers below 2^47 (KbDev::whole: kb_session_load looks at every request, Idle
//    and Releasing value; a session with anything else commits its runs row by row, like single rows, backfill and rows with their own
//    Resreq always do).  (tests/run_selection_model.py — selected_run, table_run — holds both claims to the serial loop on the CPU.)
//
// 2. The workgroup is a PIPELINE of waves that hand runs to each other through sequence words in LDS — no workgroup barrier inside the
//    loop, so that a wave with a long step does not hold up the others:
//      waves 5..7   prep, run m on wave 5 + m % 3: walk the shape's candidate list two runs ahead (r + r' + r'' clean entries: the two runs in
//                   front may still take theirs), fetch the entries' node state, NodeInfo.AddTask and the key after the placement for each
//                   (registers); when run m - 1 is committed: drop the entries it and its predecessor took (dirty bitmap), store the first r
//                   survivors as the run's candidates (their dirty slots included), publish seq_cand = m + 1;
//      waves 1..4   when run k - 1 is committed: key(shape k, dirty slot t) for every dirty slot, publish seq_dk = k + 1;
//      wave 0       when both are there: the selection (or the serial loop), decision records, cursors, dirty slots; publish seq_done = k + 1.
//    A run's critical path is one evaluation (dirty slots) and its selection.  The words only grow; every wait is bounded (a wave that
//    waits too long raises `err`, everybody leaves, the round reports KB_REASON_INTERNAL instead of hanging).
#include <string.h>

#include <algorithm>

#include "kb_k9.hpp"
#include "kb_repair.hpp"

#define K9S_PREP0 5u          // waves 5, 6, 7 prepare runs m % 3 == 0, 1, 2
#define K9S_PREPS 3u
#define K9S_SPIN_LIMIT (1u << 21)   // polls of ~100+ cycles each: a few hundred milliseconds

// the sequence words (K9Sync, in LDS): acquire loads / release stores at workgroup scope
__device__ __forceinline__ uint32_t k9s_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void k9s_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// wait until *p >= need; false when the round stopped (or somebody timed out) before that
__device__ __forceinline__ bool k9s_wait(K9Sync &Y, const uint32_t *p, uint32_t need, uint32_t nap = 1u) {
  uint32_t spins = 0;
  for (;;) {
    if (k9s_ld(p) >= need) return true;
    if ((++spins & 31u) == 0u) {   // the words a waiter is woken by move with seq_done; stop / err are looked at now and then
      if (k9s_ld(&Y.stop) | k9s_ld(&Y.err)) return k9s_ld(p) >= need;
      if (spins > K9S_SPIN_LIMIT) { k9s_st(&Y.err, 1u); return false; }
    }
    if (nap == 0u) { } else if (nap == 1u) __builtin_amdgcn_s_sleep(1); else if (nap <= 4u) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(16);   // (s_sleep takes an immediate)
  }
}

__global__ void __launch_bounds__(K9_THREADS) k_commit_select(const K9KernArgs ka) {
  KbCommitArgs a = ka.hot;
  {   // only `a` is named in the loops (SGPRs); the two views are read through the kernel-argument segment on rare paths
    const unsigned char __attribute__((address_space(4))) *kp = (const unsigned char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    a.dev = (const KbDev *)(kp + offsetof(K9KernArgs, dev));
    a.round = (const KbRound *)(kp + offsetof(K9KernArgs, round));
  }
  //@@ top
  extern __shared__ __align__(16) unsigned char k9_smem[];
  // A launch that carries its round's repair workgroups (KbRound::lists_ready): row j is workgroup number j among those whose index is not a
  // multiple of 8 (1 .. 7, 9 .. 15, ...; the multiples are the commit workgroup and its L2 helpers) — the FIRST workgroups behind workgroup 0, because
  // a launch's workgroups start in index order at ~0.13 us apiece (8 waves and a 160 KB LDS block each): behind the 64 others a row started
  // 8 us late and the commit workgroup waited 13 us for its lists (call 13)
  if ((blockIdx.x & 7u) != 0u && a.round->lists_ready != nullptr) {
    const uint32_t row = blockIdx.x - 1u - blockIdx.x / 8u;
    // (inlined, the views through the kernel-argument segment: as a function of its own the views arrive as VGPR pointers, every field becomes a
    // vector load repeated behind each store and a row takes 15 us instead of 7 — calls 13, 14; the segment pointer is not a callee's to ask for — call 15)
    if (row < a.n_mrows && !KB_CHAIN_BROKEN(*a.round)) kb_repair_row<K9_THREADS>(*a.dev, *a.round, k9_smem, row, threadIdx.x);
    return;
  }
  if (k9_preamble(a)) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t S = a.n_mrows, W = a.n_rows;
  const K9Layout lo = k9_layout(W, S, a.L, a.NP, a.R, true);
  unsigned char *k9_base_ = k9_smem;
  K9_LDS_VIEWS(lo)
  (void)shp; (void)S; (void)dk; (void)ckey; (void)cpos;
  K9Sel &X = *reinterpret_cast<K9Sel *>(k9_smem + lo.sel);
  K9Sync &Y = X.sync;
  const unsigned long long t_start = wall_clock64();
  // (run lengths: from the run-start bitmap below — k9_prologue's own rule walks forward row by row)
#ifdef KB_K9_TRACE
  unsigned long long t_stage[2] = {0ull, 0ull};   // (trace build: the fixed parts of a round in units of the 100 MHz clock — staging, shape tables, run tables, loop)
  const bool staged = k9_prologue(a, lo, k9_smem, tid, 0u, t_stage);
#else
  const bool staged = k9_prologue(a, lo, k9_smem, tid, 0u);
#endif
  if (!staged) {   // a candidate list never came: nothing was evaluated or committed
    if (tid == 0) k9_publish_skipped(a);
    return;
  }

  // the job of "my" row (the epilogue marks it as allocated): fetched now, wanted then
  uint32_t job_of_my_row = 0u;
  if (tid < W) job_of_my_row = a.dev->t_job[desc[tid].task];
  // ---- the runs of the window, by number: run k starts at row X.runs[k] (rinfo there holds its length, shape, flags, Resreq key mask).  A
  //      row starts a run iff it cannot join its predecessor (k9_prologue's rule) or its stretch of joinable rows has reached a multiple of
  //      the longest run.  W <= KB_K5_MAX_ROWS = 256: eight mask words.
  if (tid < 8) { X.brk[tid] = 0u; X.stm[tid] = 0u; }
  if (tid < 4) { X.stat[tid] = 0u; X.tr[tid] = 0u; }
  if (tid < sizeof(K9Sync) / 4) reinterpret_cast<uint32_t *>(&Y)[tid] = 0u;   // every sequence word starts at 0; nd_at[0] = 0: run 0 finds no dirty slot
  __syncthreads();
  if (tid < W) {
    bool brk = tid == 0;
    if (tid) {
      const KbRowDesc &p = desc[tid - 1], &q = desc[tid];
      const bool plain = (p.flags & 1u) && (p.resmask == 0u || (p.flags & 4u));
      brk = !(plain && !(p.flags & 2u) && q.slot == p.slot && q.flags == p.flags && q.resmask == p.resmask);
    }
    if (brk) atomicOr(&X.brk[tid >> 5], 1u << (tid & 31));
  }
  __syncthreads();
  if (tid < W) {
    uint32_t w = tid >> 5, mword = X.brk[w] & (0xFFFFFFFFu >> (31u - (tid & 31u)));   // break bits at or below my row
    while (mword == 0u) { w--; mword = X.brk[w]; }                                      // row 0 always breaks
    const uint32_t p = 32u * w + (31u - (uint32_t)__clz((int)mword));                   // first row of my stretch
    if ((tid - p) % K9_SEL_MAXRUN == 0u) atomicOr(&X.stm[tid >> 5], 1u << (tid & 31));
  }
  __syncthreads();
  if (tid < W && ((X.stm[tid >> 5] >> (tid & 31)) & 1u)) {
    uint32_t kidx = (uint32_t)__popc(X.stm[tid >> 5] & ((1u << (tid & 31)) - 1u));
    for (uint32_t w = 0; w < (tid >> 5); w++) kidx += (uint32_t)__popc(X.stm[w]);
    X.runs[kidx] = (uint16_t)tid;
  }
  __syncthreads();
  uint32_t K = 0;
  for (uint32_t w = 0; w < 8; w++) K += (uint32_t)__popc(X.stm[w]);
  {   // rinfo, compacted in place: entry k = run k's header {first row | rows << 16, shape, flags, Resreq key mask} — one load per run instead
      // of runs[k] -> rinfo[row] (k <= its first row, every read is done before the first write)
    uint4 rr = make_uint4(0u, 0u, 0u, 0u);
    uint32_t i0c = 0u;
    uint32_t rlen = 0u;
    if (tid < K) { i0c = X.runs[tid]; rr = rinfo[i0c]; rlen = ((tid + 1u < K) ? (uint32_t)X.runs[tid + 1u] : W) - i0c; }   // the run ends where the next one starts
    __syncthreads();
    if (tid < K) rinfo[tid] = make_uint4(i0c | (rlen << 16), rr.y, rr.z, rr.w);
    __syncthreads();
  }

#ifdef KB_K9_TRACE
  // make EXTRA=-DKB_K9_TRACE: wave 0's cycles — 0: waiting for a run's candidates and dirty keys, 3: rows (serial loop, tail, publish), 5: entries + first
  // rank, 6: deep passes, 7: picks + AddTask, 8: the all-clean path; X.tr: the prep waves' wait for their run's turn / for the walk's turn,
  // wave 1's wait for the previous run / its evaluation
  uint32_t tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_tab = wall_clock64();
  tacc[1] = (uint32_t)(t_stage[0] - t_start); tacc[9] = (uint32_t)(t_stage[1] - t_stage[0]); tacc[2] = (uint32_t)(t_tab - t_stage[1]);
  unsigned long long tlast = __builtin_readcyclecounter();
#define K9S_TR(k, t0) do { if (lane == 0) atomicAdd(&X.tr[k], (uint32_t)(__builtin_readcyclecounter() - (t0))); } while (0)
#define K9S_NOW() __builtin_readcyclecounter()
#else
#define K9S_TR(k, t0) do { (void)(t0); } while (0)
#define K9S_NOW() 0ull
#endif
  // how long a waiting prep / evaluating wave sleeps between two polls: s_sleep 1 (64 clocks).  Measured 0 .. 1 alike, longer slower
  // (profiles/round4/call18_19_20_polls_and_fence); the KB_SEL_SLEEP switch that swept it is gone
#ifndef KB_K9_NAP_PREP
#define KB_K9_NAP_PREP 1u
#endif
  constexpr uint32_t nap_prep = KB_K9_NAP_PREP, nap_dk = 1u;
  // issue priority: waves that share a SIMD — wave w and wave w + 4 — and the CU's one scalar unit are arbitrated by priority, then age
  // (MI355X_MICROARCH.md).  Wave 0 is the pipeline's critical path (3), the evaluating waves are what it waits for (2), the preparing waves work two
  // runs ahead (0).  Measured on one box, two runs each (profiles/round6/call11_issue_priority/): 100k x 10k 39.36 / 39.40 ms against 39.82 / 39.73,
  // survey nodes 54.66 / 55.00 against 55.11 / 55.49, R = 16 unchanged; a longer sleep of the preparing waves (KB_K9_NAP_PREP=2) gave nothing.
  // -DKB_K9_PRIO=0 builds without it
#if !defined(KB_K9_PRIO) || KB_K9_PRIO
#define K9S_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define K9S_SETPRIO(p) do { } while (0)
#endif
  const unsigned long long lt = (1ull << lane) - 1ull;
  gptrd gi, gr;
  { const KbDev &d = *a.dev; gi = (gptrd)d.idle; gr = (gptrd)d.rel; }
  // a run's header, from the tables built above (nothing in them changes inside the loop)
#define K9S_RUN_HEADER_OF(ri_)                                                                                         \
  const uint32_t hx_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ri_).x);                                          \
  const uint32_t i0 = hx_ & 0xFFFFu, r = hx_ >> 16, s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ri_).y);          \
  const uint32_t fl0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ri_).z), km0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ri_).w); \
  const bool plain0 = (fl0 & 1u) && (km0 == 0u || (fl0 & 4u));                                                         \
  const bool sel_run = !a.backfill && plain0 && r >= 2u && a.whole;   /* the run goes through the selection */          \
  (void)sel_run; (void)i0;
#define K9S_RUN_SHAPE(kk)                                                                                              \
  const K9Shape sh = shapes[s];                                                                                        \
  const double *si = sinit + (size_t)s * RS;                                                                           \
  const double *rres = rowres + ((kk) % K9S_PREPS) * (uint32_t)a.R;   /* the row's own Resreq (rows that are not plain) */ \
  const double *rqv = plain0 ? si : rres + 2;                         /* the rows' scalar Resreq */                    \
  (void)rqv; (void)rres;
#define K9S_RUN_HEADER(kk)                                                                                             \
  const uint4 ri_ = rinfo[(kk)];                                                                                       \
  K9S_RUN_HEADER_OF(ri_)                                                                                               \
  K9S_RUN_SHAPE(kk)

  //@@ w0_setup
  if (wave == 0) {
    K9S_SETPRIO(3);
    // =================================================== wave 0: the selection ===================================================
    uint32_t nd = 0, n_dirty_rows = 0, n_runs = 0, n_slow = 0, i_end = 0, reason_end = KB_REASON_DONE;
    uint32_t prev_nd0 = 0, prev_pc = 0;   // the run in front: dirty slots when it started, clean candidates it consumed
    bool prev_chg = true;                 // ... whether it changed slots in any other way (then the early dirty keys of this run are not final)
    bool prev_all = false;                // ... and whether it took every candidate of its own (then this run's candidates, settled ahead on that assumption, stand)
  //@@ w0_header_wait
    uint4 hd_next = rinfo[0];   // (K >= 1: a launch has rows)
    for (uint32_t k = 0; k < K; k++) {
      uint32_t lane_i = tid;   // wave 0: tid == lane.  Opaque per iteration: otherwise every `lane < c` of the loop is hoisted in front of it as an
      asm volatile("" : "+v"(lane_i));   // SGPR-pair mask, the pairs are spilled to VGPR lanes and read back here — dearer than the compare
      const uint4 hd = hd_next;
      if (k + 1u < K) hd_next = rinfo[k + 1u];   // the next run's header: in flight behind this run's work
      K9S_RUN_HEADER_OF(hd)
      K9_STAMP(3);
      {   // the run's candidates (seq_cand) and its dirty keys (seq_dk[0..3]): five neighbouring words, one load per poll
        uint32_t spins = 0;
        bool ok = true;
        for (;;) {
          // lane_i 0: the candidates — settled ahead (seq_spec) when the run in front took all of its own, else settled behind it (seq_cand);
          // lanes 1..4: the dirty keys — the early ones (seq_early) when the run in front changed nothing else, else the final ones (seq_dk)
          const uint32_t *wp = (lane_i == 0u) ? (prev_all ? &Y.seq_spec : &Y.seq_cand) : (prev_chg ? &Y.seq_dk[(lane_i - 1u) & 3u] : &Y.seq_early[(lane_i - 1u) & 3u]);
          const uint32_t v = (lane_i < 5u) ? k9s_ld(wp) : 0xFFFFFFFFu;
          if (__ballot(v < k + 1u) == 0ull) break;
          if ((++spins & 31u) == 0u && (k9s_ld(&Y.err) || spins > K9S_SPIN_LIMIT)) { k9s_st(&Y.err, 1u); ok = false; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) { reason_end = KB_REASON_INTERNAL; i_end = i0; break; }
      }
  //@@ w0_loads
      if (prev_all && lane_i == 0) k9s_st(&Y.seq_cand, k + 1u);   // the candidates settled ahead are the candidates: the evaluating waves may build on them
      // ONE batch of LDS reads behind the wait: the number of candidates, their fields (a lane_i beyond the last candidate reads a stale word and
      // zeroes it below), the evaluating waves' maxima
      const uint32_t ncand = Y.ncand_at[k & 3u];
      const K9Sel::Cand &CD = X.cand[k & 1u];
      const uint32_t *dk = X.dkb[k & 1u];
      uint32_t *chg = Y.chg[k & 1u];
      uint32_t ck = CD.ckey[lane_i], k1 = CD.ck1[lane_i], ckind = CD.ckind[lane_i];
      const uint32_t cpos = CD.cpos[lane_i];
      // best dirty key: the evaluating waves' maxima (clean winners update it in O(1)).  From the early evaluation: the maxima over the slots
      // that were dirty before the run in front, and the keys of the candidates it consumed
      uint32_t m = (lane_i < 4u) ? (prev_chg ? Y.mdk[k & 1u][lane_i] : Y.mdko[k & 1u][lane_i]) : 0u;
      if (!prev_chg && lane_i >= 4u && lane_i - 4u < prev_pc) m = dk[prev_nd0 + lane_i - 4u];   // prev_pc <= K9_SEL_MAXRUN
      if (lane_i < 10u) chg[lane_i] = 0u;
      K9_STAMP(0);
      if (lane_i >= ncand) { ck = 0u; k1 = 0u; ckind = 0u; }
      m = wave_max_u32(m);
      uint32_t pc = 0, j = 0, reason = KB_REASON_DONE, n_dirty = 0, sc_dirty = 0;
      bool chg_any = false;   // this run changed slots other than by consuming a clean candidate once (chg carries which; the next run's early keys are then not final)
      bool run_done = false;
  //@@ w0_clean
      if (!a.backfill && plain0) {
        // ---- every pick a clean candidate's first placement?  r candidates, no dirty key above the r-th of them, no candidate whose key
        //      after its placement is: row j takes candidate j (a Pipeline among them ends the round behind its row).  Single rows too:
        //      the serial loop below would do exactly this for a row whose best clean candidate beats every dirty key.
        // cmin: the r-th best clean candidate's key — r entries are at or above it, so no entry below it is among the picks (0: the list
        // holds fewer than r clean nodes)
        const uint32_t cmin = (ncand == r) ? rl32(ck, r - 1u) : 0u;
        const unsigned long long deeper = __ballot(lane_i + 1u < r && ckind == 0u && k1 > cmin);   // (lanes beyond the candidates hold zeros)
        if (ncand == r && m < cmin && !deeper) {
          const unsigned long long pipes = __ballot(lane_i < r && ckind != 0u);
          const uint32_t n_take = pipes ? (uint32_t)__ffsll((unsigned long long)pipes) : r;
          if (pipes) reason = KB_REASON_PIPELINED;
          if (lane_i < n_take) {
            const uint32_t n = nmaskbits - (ck & nmaskbits);
            ldec[i0 + lane_i] = (unsigned long long)n | ((unsigned long long)ckind << 32);
            atomicOr(&bitmap[n >> 5], 1u << (n & 31));
            if (km0) {   // the scalar dimensions Resreq names: Idle / Releasing in HBM
              const double *rqv = sinit + (size_t)s * RS;   // (plain rows: the shape's own request)
              const uint32_t rnm = CD.crnm[lane_i];
              atomicOr(&chg[(nd + lane_i) >> 5], 1u << ((nd + lane_i) & 31));   // an early evaluation for the next run read them before this Sub
              const bool has_map = ckind ? (rnm >> 31) : (rnm & 0x7FFFFFFFu);
              if (has_map)
                for (uint32_t mm = km0, dd = 0; mm; mm >>= 1, dd++)
                  if (mm & 1u) k9_sc_sub(ckind ? gr : gi, a.NP, dd, n, rqv[dd]);
            }
          }
          if (km0) { sc_dirty = 1; chg_any = true; }
          pc = n_take; j = n_take;
          run_done = true;
          if (lane_i == 0 && r >= 2u) atomicAdd(&X.stat[0], 1u);
          K9_STAMP(8);
        }
      }
  //@@ w0_slow_general
      if (!run_done) {
      // ================= everything else: the general selection, the serial loop (they need the shape and the candidates' other fields)
      K9S_RUN_SHAPE(k)
      uint32_t rnm = 0;
      if (lane_i < ncand) rnm = CD.crnm[lane_i];
      double res0 = sh.init0, res1 = sh.init1;
      if (!plain0) { res0 = rres[0]; res1 = rres[1]; }
      // dirty keys of the shape: old slot t in lane_i t & 63, register t >> 6 — fetched only where a dirty slot can be picked
      uint32_t d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u;
      bool have_d = false;
#define K9S_LOAD_D() do { if (!have_d) { d0 = (lane_i < nd) ? dk[lane_i] : 0u; d1 = (lane_i + 64 < nd) ? dk[lane_i + 64] : 0u; d2 = (lane_i + 128 < nd) ? dk[lane_i + 128] : 0u; \
                                        d3 = (lane_i + 192 < nd) ? dk[lane_i + 192] : 0u; have_d = true; } } while (0)
      bool sel_done = false;
      if (sel_run) {
        // ---- the SHOTS (round 6; tests/run_selection_model.py: table_run).  The contenders of the rows still open: everybody whose key is at or
        //      above the rem-th clean candidate in front (the distinct winners of rem rows are the rem best by current key) — clean candidates,
        //      candidates this run has consumed, dirty slots.  Lane g * D + u evaluates contender g after u further placements of the shape
        //      (D = the lanes' budget / contenders: two entries each for 32 contenders, 32 for two; api/node_info.go:172-212 applied u times as
        //      Idle - u * Resreq, exact for whole numbers below 2^47: KbDev::whole, the host's scan at kb_session_load — a session with anything
        //      else commits its runs row by row); prefix minima per contender; ONE rank
        //      of all entries by (eff desc, step asc): the serial loop's picks in order (the argument at the top of this file).  Of that order a
        //      shot commits what is FINAL: entries above the floor (nobody outside the table comes first) in front of the first table that ends
        //      while its sequence may go on — what lies behind a table's end ranks behind its last entry, nowhere else — ; that entry itself waits
        //      for the next shot, so a contender's key after its placements is always an entry of its table.  The best contender's first entry is
        //      always final: every shot commits a row.  A Pipeline (allocate.go:175-182) ends the round behind its row.
        K9S_LOAD_D();
        bool pooled = false;   // the pool's keys live in X.pool as well (from the second shot on)
        while (j < r) {
          const uint32_t rem = r - j;
          const uint32_t fidx = pc + rem - 1u;
          uint32_t thr = max(fidx < ncand ? rl32(ck, fidx) : 0u, 1u);   // the floor: the rem-th clean candidate in front (clean keys fall along the list); 1: fewer are left, everybody feasible contends
          // -- contenders: the pool's entries at or above the floor.  Lane i holds candidate i of this run (consumed: its key after the placements it took, slot
          //    nd + i; still clean: its list key, same slot, one placement ahead) and the dirty slots i, i + 64, i + 128, i + 192 (registers beyond nd: zeros)
          const uint32_t kc = lane_i < pc ? k1 : (lane_i < ncand ? ck : 0u);
          bool isc, is0, is1, is2, is3;
          unsigned long long bc, b0, b1, b2, b3;
          uint32_t nc_, n0_, n1_, n2_, n3_, nC;
          for (uint32_t again = 0;; again++) {
            isc = kc >= thr; is0 = d0 >= thr; is1 = d1 >= thr; is2 = d2 >= thr; is3 = d3 >= thr;
            bc = __ballot(isc); b0 = __ballot(is0);
            nc_ = (uint32_t)__popcll(bc); n0_ = (uint32_t)__popcll(b0);
            b1 = 0ull; b2 = 0ull; b3 = 0ull; n1_ = 0u; n2_ = 0u; n3_ = 0u;
            if (nd > 64u) { b1 = __ballot(is1); n1_ = (uint32_t)__popcll(b1); }
            if (nd > 128u) { b2 = __ballot(is2); b3 = __ballot(is3); n2_ = (uint32_t)__popcll(b2); n3_ = (uint32_t)__popcll(b3); }
            nC = nc_ + n0_ + n1_ + n2_ + n3_;
            if (nC <= 32u || again) break;
            // more contenders than the lanes hold two entries of (the clean list has run short: every feasible dirty slot contends): a higher floor —
            // the fifth best of the lanes' own maxima, i.e. the entries of five lanes at most (25); whoever stays outside has a key below it
            uint32_t lm = max(max(max(d0, d1), max(d2, d3)), kc);
            for (uint32_t t = 0; t < 5u; t++) {
              const uint32_t mx = wave_max_u32(lm);
              if (mx == 0u) break;
              thr = mx;
              if (lm == mx) lm = 0u;
            }
          }
          if (nC == 0u) { reason = KB_REASON_NO_FEASIBLE; break; }   // allocate.go:144-148
          // -- the rem best of them, best first: the distinct winners of rem rows are the rem best by current key, and every contender less is a deeper
          //    table for the others.  Records and keys go to LDS in pool order, contender lane p ranks its key among the nC (they are distinct: they carry
          //    the node), rank = its number; the ones behind the rem-th are dropped and the floor moves up to the rem-th key
          {
            const uint32_t p_c = (uint32_t)__popcll(bc & lt), p_0 = nc_ + (uint32_t)__popcll(b0 & lt);
            if (isc) { X.c_slot[p_c] = (nd + lane_i) | (lane_i >= pc ? 0x80000000u : 0u); X.c_key[p_c] = kc; }   // bit 31: the slot is one placement ahead (a clean candidate)
            if (is0) { X.c_slot[p_0] = lane_i; X.c_key[p_0] = d0; }
            if (nd > 64u && is1) { const uint32_t p_1 = nc_ + n0_ + (uint32_t)__popcll(b1 & lt); X.c_slot[p_1] = lane_i + 64u; X.c_key[p_1] = d1; }
            if (nd > 128u) {
              if (is2) { const uint32_t p_2 = nc_ + n0_ + n1_ + (uint32_t)__popcll(b2 & lt); X.c_slot[p_2] = lane_i + 128u; X.c_key[p_2] = d2; }
              if (is3) { const uint32_t p_3 = nc_ + n0_ + n1_ + n2_ + (uint32_t)__popcll(b3 & lt); X.c_slot[p_3] = lane_i + 192u; X.c_key[p_3] = d3; }
            }
            if (lane_i >= nC && lane_i < nC + 8u) X.c_key[lane_i] = 0u;   // (nC <= 32; the rank reads eight keys at a time)
            K9_WAVE_FENCE();
            const uint32_t myk = lane_i < nC ? X.c_key[lane_i] : 0u;
            uint32_t rk = 0u;
            const uint4 *kv = reinterpret_cast<const uint4 *>(X.c_key);
            for (uint32_t i = 0; i < nC; i += 8u) {
              const uint4 q0_ = kv[i >> 2], q1_ = kv[(i >> 2) + 1u];
              rk += (q0_.x > myk ? 1u : 0u) + (q0_.y > myk ? 1u : 0u) + (q0_.z > myk ? 1u : 0u) + (q0_.w > myk ? 1u : 0u);
              rk += (q1_.x > myk ? 1u : 0u) + (q1_.y > myk ? 1u : 0u) + (q1_.z > myk ? 1u : 0u) + (q1_.w > myk ? 1u : 0u);
            }
            // A packing session (a.wM > a.wL) keeps ten at most: its winners are few and take many rows each, so ten contenders with deeper tables (the two
            // best sixteen entries, the others four) resolve more rows per shot than rem contenders with two — BASELINE configs[3], shots per cycle: cap 6
            // 8 715, 8 8 135, 10 7 790, 12 7 813, none 9 106; 58.0 - 60.1 ms against 62.4 - 64.8 on that box (profiles/round6/call18_19_contender_cap/).
            // Every other session keeps sixteen at most (four entries each instead of two for seventeen and more): 100k x 10k 11 473 shots per five cycles against
            // 12 887 (cap 12: 13 566, 20: 13 363), survey nodes 27 111 against 34 398 and 51.8 - 52.1 ms against 54.6 - 54.8 (call20_21_contender_cap_every_session/).
            // -DKB_K9_PACK_CAP=n / -DKB_K9_SPREAD_CAP=n are the A/B builds
#ifndef KB_K9_PACK_CAP
#define KB_K9_PACK_CAP 10u
#endif
#ifndef KB_K9_SPREAD_CAP
#define KB_K9_SPREAD_CAP 16u
#endif
            const uint32_t keep = min(rem, (a.wM > a.wL) ? (uint32_t)KB_K9_PACK_CAP : (uint32_t)KB_K9_SPREAD_CAP);
            if (lane_i < nC && rk < keep) X.c_sorted[rk] = X.c_slot[lane_i];
            if (nC > keep) {
              thr = rl32(myk, (uint32_t)__ffsll((unsigned long long)__ballot(lane_i < nC && rk == keep - 1u)) - 1u);
              nC = keep;
            }
            K9_WAVE_FENCE();
          }
          // -- the lanes' layout.  Uniform: contender g's D entries in lanes g * D ...  With a DEEP table: the best contender (number 0) in lanes 0 .. 15 (sixteen
          //    entries: under MostRequested the node just used wins until it is full, most_requested.go:34-61), the others share lanes 16 .. 63 — wherever
          //    that costs them nothing (their D is what the uniform layout gives them) or the session packs (a.wM > a.wL)
          uint32_t lgD = nC <= 2u ? 5u : (nC <= 4u ? 4u : (nC <= 8u ? 3u : (nC <= 16u ? 2u : 1u)));
          uint32_t ndeep = 0u;   // contenders 0 .. ndeep - 1 (the best ones: they are numbered by key) get sixteen entries each
          if (nC >= 5u && nC <= 25u) {
            const uint32_t o_ = nC - 1u, lgO = o_ <= 6u ? 3u : (o_ <= 12u ? 2u : 1u);   // 48 lanes for the others
            if (lgO == lgD || a.wM > a.wL) { ndeep = 1u; lgD = lgO; }
          }
#if !defined(KB_K9_DEEP2) || KB_K9_DEEP2
          if (a.wM > a.wL && ((nC >= 5u && nC <= 10u) || (nC >= 14u && nC <= 18u))) {   // a packing session: the best TWO (BASELINE configs[3]: 61.6 / 62.1 -> 60.3 / 60.8 ms,
                                                                                          // 6 % fewer shots; profiles/round6/call17_last_row_and_two_deep_tables/; -DKB_K9_DEEP2=0 builds without)
            const uint32_t o_ = nC - 2u;                                                  // 32 lanes for the others, never fewer entries each than with one deep table
            ndeep = 2u; lgD = o_ <= 2u ? 4u : (o_ <= 4u ? 3u : (o_ <= 8u ? 2u : 1u));
          }
#endif
          const uint32_t D = 1u << lgD, dl = 16u * ndeep;   // dl: the deep tables' lanes
          K9_STAMP(5);
          // -- the table: contender tg's entry tu (its key after tu further placements); gst: the first lane of my contender, Dg: its entries
          uint32_t tg, tu, gst, Dg;
          if (lane_i < dl) { tg = lane_i >> 4; tu = lane_i & 15u; gst = lane_i & ~15u; Dg = 16u; }
          else { const uint32_t l_ = lane_i - dl; tg = ndeep + (l_ >> lgD); tu = l_ & (D - 1u); gst = lane_i - tu; Dg = D; }
          const bool act = tg < nC;
          const uint32_t n_e = dl + ((nC - ndeep) << lgD);   // lanes in use
          uint32_t key = 0u, kind = 0u, tnode = 0u;
          if (act) {
            const uint32_t cs = X.c_sorted[tg], slot = cs & 0x7FFFFFFFu, b = cs >> 31;
            const unsigned long long *st = slots + (size_t)slot * K9_NF;
            K9St v = k9_load(st);
            tnode = v.node;
            const uint32_t nm0 = (uint32_t)(st[F_NODE_NMASK] >> 32);
            const uint32_t adjm = (nm0 & 0x7FFFFFFFu) ? km0 : 0u;
            const K9Sc scx = k9_sc_preload(sh.active >> 2, gi, gr, a.NP, v.node);
            const uint32_t mpl = tu >= b ? tu - b : 0u;   // placements on top of the slot's state
            const double fm = (double)mpl;   // (whole numbers below 2^47: Idle - u * Resreq is what u subtractions give — resource_info.go:143-156, one Sub per placement)
            v.idle0 -= fm * sh.init0; v.idle1 -= fm * sh.init1; v.nzc += fm * sh.nzc; v.nzm += fm * sh.nzm;
            if (mpl) v.ports |= sh.want;
            v.left -= (int)mpl;
            bool fi = true;   // (sel_run: allocate, fit_mode 1)
            key = k9_eval_v(a, sh, v, scx, gi, gr, si, adjm, (double)tu, si, nb, nmaskbits, &fi);
            kind = fi ? 0u : 1u;   // allocate.go:160: InitResreq.LessEqual(node.Idle) -> Allocate, else Pipeline
            if (b && tu == 0u) {   // a clean candidate's own first placement: as its prep wave found it (key against the clean node, Allocate / Pipeline)
              const uint32_t ci = slot - nd;
              key = CD.ckey[ci]; kind = CD.ckind[ci];
            }
          }
          // an entry exists iff its key is not 0 and every entry of its contender in front of it exists and is an Allocate
          const unsigned long long ends = __ballot(act && (key == 0u || kind != 0u));
          const unsigned long long front = lt & ~((1ull << gst) - 1ull);   // my contender's lanes in front of me
          const bool valid = act && key != 0u && (ends & front) == 0ull;
          X.e_key[lane_i] = key;
          // prefix minimum per contender (its lanes are neighbours; up to 16 entries: inside one DPP row)
          uint32_t eff = valid ? key : 0xFFFFFFFFu;
#define K9S_SHR(sft) do { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)eff, 0x110 + (sft), 0xf, 0xf, false); if (tu >= (sft)) eff = min(eff, t_); } while (0)
          K9S_SHR(1); K9S_SHR(2); K9S_SHR(4); K9S_SHR(8);
#undef K9S_SHR
          if (Dg == 32u) { const uint32_t t_ = (uint32_t)__shfl((int)eff, (int)((lane_i & 0x30u) | 15u) - 16); if (tu >= 16u) eff = min(eff, t_); }   // the contender's first row
          if (!valid) eff = 0u;
          X.e_eff[lane_i] = eff;
          K9_WAVE_FENCE();
          K9_STAMP(6);
          // -- the rank: entries with a greater eff, and the same contender's earlier entries with the same eff (they are neighbours)
          uint32_t rank = 0u;
          {
            const uint4 *ev = reinterpret_cast<const uint4 *>(X.e_eff);   // (every lane stored its word: zeros behind the lanes in use)
            for (uint32_t i = 0; i < n_e; i += 16u) {   // four reads in flight
              const uint4 q0_ = ev[(i >> 2)], q1_ = ev[(i >> 2) + 1u], q2_ = ev[(i >> 2) + 2u], q3_ = ev[(i >> 2) + 3u];
              rank += (q0_.x > eff ? 1u : 0u) + (q0_.y > eff ? 1u : 0u) + (q0_.z > eff ? 1u : 0u) + (q0_.w > eff ? 1u : 0u);
              rank += (q1_.x > eff ? 1u : 0u) + (q1_.y > eff ? 1u : 0u) + (q1_.z > eff ? 1u : 0u) + (q1_.w > eff ? 1u : 0u);
              rank += (q2_.x > eff ? 1u : 0u) + (q2_.y > eff ? 1u : 0u) + (q2_.z > eff ? 1u : 0u) + (q2_.w > eff ? 1u : 0u);
              rank += (q3_.x > eff ? 1u : 0u) + (q3_.y > eff ? 1u : 0u) + (q3_.z > eff ? 1u : 0u) + (q3_.w > eff ? 1u : 0u);
            }
            const uint32_t prev = (uint32_t)__shfl_up((int)eff, 1);
            const unsigned long long starts = __ballot(!(tu > 0u && prev == eff));
            const unsigned long long upto = starts & (lt | (1ull << lane_i));
            rank += lane_i - (63u - (uint32_t)__clzll((long long)upto));
          }
          // -- how many of them are final
          const uint32_t nvalid = (uint32_t)__popcll(__ballot(valid));
          uint32_t cutv = 0xFFFFu;
          if (valid) {
            if (eff < thr) cutv = rank;                          // somebody outside the table may come first
            else if (tu == Dg - 1u && kind == 0u && rank + 1u < rem) cutv = rank;   // the table ends here, the sequence may not: the next shot — unless this
                                                                                    // is the run's last row: nobody asks for the contender's key behind it
            else if (kind != 0u) cutv = rank + 1u;                // a Pipeline ends the round behind its row
          }
          const uint32_t cut_at = 0xFFFFu - wave_max_u32(0xFFFFu - cutv);
          const uint32_t n_take = min(min(rem, nvalid), cut_at);
          if (cut_at < min(rem, nvalid) && __ballot(valid && rank == cut_at && eff >= thr && kind == 0u) != 0ull && lane_i == 0) atomicAdd(&X.stat[2], 1u);   // (statistics: a shot cut short by a table's end)
          if (n_take == 0u) { reason = KB_REASON_INTERNAL; k9s_st(&Y.err, 1u); break; }   // (never: the best contender's first entry is final)
          const bool picked = valid && rank < n_take;
          if (picked) ldec[i0 + j + rank] = (unsigned long long)tnode | ((unsigned long long)kind << 32);
          const unsigned long long pb = __ballot(picked), ppipe = __ballot(picked && kind != 0u);
          const bool pipe = ppipe != 0ull;
          uint32_t gp = 0xFFFFFFFFu;   // the contender whose last row is the Pipeline
          if (pipe) { const uint32_t lp_ = (uint32_t)__ffsll((unsigned long long)ppipe) - 1u; gp = lp_ < dl ? (lp_ >> 4) : ndeep + ((lp_ - dl) >> lgD); }
          // -- NodeInfo.AddTask (api/node_info.go:172-212) on every contender that took rows: lane g = contender g
          uint32_t T = 0u, c_x = 0u, c_b = 0u;
          const uint32_t cst = lane_i < ndeep ? 16u * lane_i : dl + ((lane_i - ndeep) << lgD);   // lane g: the first lane of contender g's table
          if (lane_i < nC) {
            T = (uint32_t)__popcll((pb >> cst) & ((1ull << (lane_i < ndeep ? 16u : D)) - 1ull));
            const uint32_t cs = X.c_sorted[lane_i];
            c_x = cs & 0x7FFFFFFFu; c_b = cs >> 31;
          }
          if (T) {
            unsigned long long *st = slots + (size_t)c_x * K9_NF;
            const uint32_t node = (uint32_t)st[F_NODE_NMASK], nm = (uint32_t)(st[F_NODE_NMASK] >> 32);
            const bool plast = lane_i == gp;
            const uint32_t extra = T - c_b;   // a clean candidate's slot already holds its first placement
            if (extra || km0 || c_x < nd) atomicOr(&chg[c_x >> 5], 1u << (c_x & 31));   // not what an early evaluation for the next run saw (scalar dimensions: in HBM only)
            if (extra) {
              const double fa = (double)(extra - (plast ? 1u : 0u));
              double idle0 = u2d(st[F_IDLE0]), idle1 = u2d(st[F_IDLE1]), rel0 = u2d(st[F_REL0]), rel1 = u2d(st[F_REL1]);
              idle0 -= fa * sh.init0; idle1 -= fa * sh.init1;
              if (plast) { rel0 -= sh.init0; rel1 -= sh.init1; }
              st[F_IDLE0] = d2u(idle0); st[F_IDLE1] = d2u(idle1); st[F_REL0] = d2u(rel0); st[F_REL1] = d2u(rel1);
              st[F_NZC] = d2u(u2d(st[F_NZC]) + (double)extra * sh.nzc); st[F_NZM] = d2u(u2d(st[F_NZM]) + (double)extra * sh.nzm);
              st[F_PORTS] |= sh.want;
              st[F_CLS_LEFT] -= ((unsigned long long)extra << 32);   // that many more pods on the node
            }
            if (km0)   // the scalar dimensions Resreq names, in HBM, one Sub per placement (Sub returns early when the receiver's map is nil)
              for (uint32_t t = 0; t < T; t++) {
                const bool pp = plast && t + 1u == T;
                const bool has_map = pp ? (nm >> 31) : (nm & 0x7FFFFFFFu);
                if (has_map)
                  for (uint32_t mm = km0, dd = 0; mm; mm >>= 1, dd++)
                    if (mm & 1u) k9_sc_sub(pp ? gr : gi, a.NP, dd, node, si[dd]);
              }
            if (c_b) atomicOr(&bitmap[node >> 5], 1u << (node & 31));
          }
          if (__ballot(T != 0u && (T - c_b != 0u || km0 != 0u || c_x < nd)) != 0ull) chg_any = true;
          if (km0) sc_dirty = 1;
          pc += (uint32_t)__popcll(__ballot(T != 0u && c_b != 0u));   // clean candidates leave the list in list order (their keys fall along it)
          j += n_take;
          if (lane_i == 0) atomicAdd(&X.stat[3], 1u);
          if (pipe) { reason = KB_REASON_PIPELINED; K9_STAMP(7); break; }
          if (j < r) {
            // -- another shot: the contenders' keys after their placements (entry T of their table: the last entry is never taken) go back into the pool
            if (!pooled) {
              X.pool[lane_i] = d0; X.pool[lane_i + 64u] = d1; X.pool[lane_i + 128u] = d2; X.pool[lane_i + 192u] = d3; X.pool[lane_i + 256u] = k1;
              pooled = true;
              K9_WAVE_FENCE();
            }
            if (T) X.pool[c_x < nd ? c_x : 256u + (c_x - nd)] = X.e_key[cst + T];
            K9_WAVE_FENCE();
            d0 = X.pool[lane_i]; d1 = X.pool[lane_i + 64u]; d2 = X.pool[lane_i + 128u]; d3 = X.pool[lane_i + 192u]; k1 = X.pool[lane_i + 256u];
          }
          K9_STAMP(7);
        }
        n_dirty = j - pc;
        sel_done = true;
        if (lane_i == 0) atomicAdd(&X.stat[1], 1u);
      }
      // backfill.go:50-66 takes the FIRST node that passes the predicates (all scores tie: the round runs with scores off), and nothing in the
      // placement of a plain BestEffort row — an empty request — changes what the predicates read of the node except its pod count (and its ports,
      // if the pod had any: then it conflicts with itself).  So the node that wins a row of such a run wins the following rows too until it is
      // full: they are committed at once.  (1M x 50k: 80 backfill rounds of 256 rows that all go for the same few nodes, ~20 ms row by row.)
  //@@ w0_slow_serial
      const bool bf_bulk = a.backfill && plain0 && km0 == 0u && !(fl0 & 2u) && !a.score_enabled && sh.init0 == 0.0 && sh.init1 == 0.0 && (sh.active >> 2) == 0u &&
                           res0 == 0.0 && res1 == 0.0 && (sh.want & sh.conf) == 0ull;
      if (!sel_done)
      for (; j < r; j++) {   // ---- the serial loop of kb_commit.hip (single rows, backfill, rows with their own Resreq, whatever the selection handed back)
        const uint32_t c = (pc < ncand) ? rl32(ck, pc) : 0u;
        if (m == 0u && c == 0u) {
          if (a.backfill) {   // backfill.go:50-66: no node passes the predicates -> the task stays Pending
            if (lane_i == 0) ldec[i0 + j] = (unsigned long long)KB_NONE_U32;
            continue;
          }
          reason = KB_REASON_NO_FEASIBLE;   // allocate.go:144-148: the job is abandoned; the host re-plans from here
          break;
        }
        uint32_t kind;
        if (c > m) {   // the clean candidate wins: its slot and its post-placement key are ready
          const uint32_t n = nmaskbits - (c & nmaskbits);
          kind = rl32(ckind, pc);
          m = max(m, rl32(k1, pc));
          if (lane_i == 0) {
            ldec[i0 + j] = (unsigned long long)n | ((unsigned long long)kind << 32);
            atomicOr(&bitmap[n >> 5], 1u << (n & 31));
          }
          if (km0) {   // the scalar dimensions Resreq names: Idle / Releasing in HBM (lane_i 16 + d' takes dimension d' + 2)
            if (lane_i == 0) atomicOr(&chg[(nd + pc) >> 5], 1u << ((nd + pc) & 31));   // an early evaluation for the next run read them before this Sub
            chg_any = true;
            const uint32_t nmc = rl32(rnm, pc);
            const bool has_map = kind ? (nmc >> 31) : (nmc & 0x7FFFFFFFu);
            if (has_map && lane_i >= 16 && lane_i < 16 + RS && ((km0 >> (lane_i - 16)) & 1u)) k9_sc_sub(kind ? gr : gi, a.NP, lane_i - 16, n, rqv[lane_i - 16]);
            sc_dirty = 1;
          }
          pc++;
        } else {       // a node this round already changed wins: AddTask on its slot, re-evaluate it
          K9S_LOAD_D();
          const uint32_t own = (lane_i < pc) ? k1 : 0u;
          const unsigned long long who = __ballot(d0 == m || d1 == m || d2 == m || d3 == m || own == m);
          const uint32_t L = (uint32_t)__ffsll((unsigned long long)who) - 1u;
          const bool is_new = L < pc && rl32(k1, L) == m;
          const uint32_t wsel = (rl32(d0, L) == m) ? 0u : (rl32(d1, L) == m) ? 1u : (rl32(d2, L) == m) ? 2u : 3u;
          const uint32_t x = is_new ? nd + L : L + 64u * wsel;
          if (lane_i == 0) atomicOr(&chg[x >> 5], 1u << (x & 31));   // the slot is not what an early evaluation for the next run saw
          chg_any = true;
          unsigned long long *st = slots + (size_t)x * K9_NF;
          // one row: lane_i f holds field f of the slot; lanes 16 + d' look at the scalar dimension d' + 2 in HBM when the shape or
          // the row names one
          const bool sc_lane = lane_i >= 16 && lane_i < 16 + RS;
          const uint32_t sd = sc_lane ? lane_i - 16 : 0;
          unsigned long long cur8 = 0ull;
          if (lane_i < K9_NF) cur8 = st[lane_i];
          const uint32_t nm = (uint32_t)(rl64(cur8, F_NODE_NMASK) >> 32);
          const uint32_t n = (uint32_t)rl64(cur8, F_NODE_NMASK);
          // the scalar dimensions the new key will read (only when a key is needed: not on the run's last row), in flight with the vote's loads
          const K9Sc scs = k9_sc_preload((j + 1u < r) ? (sh.active >> 2) : 0u, gi, gr, a.NP, n);
          kind = 0;
          if (!a.backfill) {
            bool ok = true;
            if (lane_i == F_IDLE0) ok = le_eps(sh.init0, u2d(cur8), EPS_CPU);
            else if (lane_i == F_IDLE1) ok = le_eps(sh.init1, u2d(cur8), EPS_MEM);
            else if (sc_lane && ((sh.active >> (2 + sd)) & 1u)) ok = le_eps(si[sd], k9_sc(gi, a.NP, sd, n), EPS_SCALAR);
            kind = __ballot(!ok) ? 1u : 0u;
          }
          const uint32_t has_map = kind ? (nm >> 31) : (nm & 0x7FFFFFFFu);
          const uint32_t f0 = kind ? F_REL0 : F_IDLE0;
          if (lane_i == f0) cur8 = d2u(u2d(cur8) - res0);
          else if (lane_i == f0 + 1) cur8 = d2u(u2d(cur8) - res1);
          else if (lane_i == F_NZC) cur8 = d2u(u2d(cur8) + sh.nzc);
          else if (lane_i == F_NZM) cur8 = d2u(u2d(cur8) + sh.nzm);
          else if (lane_i == F_PORTS) cur8 |= sh.want;
          else if (lane_i == F_CLS_LEFT) cur8 -= (1ull << 32);   // one more pod on the node
          uint32_t bulk = 1u;   // rows this node takes now
          if (bf_bulk) {
            const uint32_t left1 = (uint32_t)(rl64(cur8, F_CLS_LEFT) >> 32);   // pod slots left behind row j's placement: each takes one more row of the run
            bulk = 1u + min(r - j - 1u, left1);
            if (bulk > 1u) {
              if (lane_i == F_NZC) { double z = u2d(cur8); for (uint32_t t = 1; t < bulk; t++) z += sh.nzc; cur8 = d2u(z); }   // one addition per placement, as AddTask does them
              else if (lane_i == F_NZM) { double z = u2d(cur8); for (uint32_t t = 1; t < bulk; t++) z += sh.nzm; cur8 = d2u(z); }
              else if (lane_i == F_CLS_LEFT) cur8 -= ((unsigned long long)(bulk - 1u) << 32);
            }
          }
          if (lane_i < K9_NF) st[lane_i] = cur8;
          if (lane_i < bulk) ldec[i0 + j + lane_i] = (unsigned long long)n | ((unsigned long long)kind << 32);   // bulk > 1: backfill, kind == 0
          j += bulk - 1u;
          n_dirty += bulk - 1u;
          K9_WAVE_FENCE();
          // the new key first (scalar part of the Sub as an adjustment), then the Sub itself goes to HBM.  The last row of a run (and a
          // Pipeline, which ends the round) needs no new key: the next run evaluates every slot against ITS shape anyway
          const bool more = j + 1u < r && !kind;
          const uint32_t adjm1 = (!kind && has_map) ? km0 : 0u;
          uint32_t nk = 0u;
          if (more) nk = k9_eval_v(a, sh, k9_load(st), scs, gi, gr, si, adjm1, 1.0, rqv, nb, nmaskbits);   // uniform: every lane_i computes the same key
          if (km0 && has_map) {
            if (sc_lane && ((km0 >> sd) & 1u)) k9_sc_sub(kind ? gr : gi, a.NP, sd, n, rqv[sd]);
            sc_dirty = 1;
          }
          if (more) {
            if (lane_i == L) {
              if (is_new) k1 = nk;
              else if (wsel == 0) d0 = nk; else if (wsel == 1) d1 = nk; else if (wsel == 2) d2 = nk; else d3 = nk;
            }
            m = wave_max_u32(max(max(max(d0, d1), max(d2, d3)), (lane_i < pc) ? k1 : 0u));
          }
          n_dirty++;
        }
        if (kind) { j++; reason = KB_REASON_PIPELINED; break; }   // a Pipeline ends the speculated order: the host re-plans
      }
      }   // (!run_done)
  //@@ w0_tail
      if (sc_dirty) __threadfence();   // the scalar atomics have reached L2 before any other wave evaluates against these nodes (dropping it measured no gain: profiles/round4/call20)
      const uint32_t i_next = i0 + j;
      uint32_t stop = (reason != KB_REASON_DONE || k + 1u >= K) ? 1u : 0u;   // (a complete run ends the window iff it is the last one: i_next == W)
      if (!stop && a.has_aff && !a.backfill && i_next != 0u) {
        // a row whose score is normalised over its feasible set (preferred node affinity) is exact only against a fresh matrix: it may be
        // the first row of a round, nothing else
        // (the window goes on and this run is complete: row i_next heads run k + 1, whose header is already here)
        const uint32_t fln = (uint32_t)__builtin_amdgcn_readfirstlane((int)hd_next.z);
        if (fln & 2u) { reason = KB_REASON_RENORM; stop = 1u; }
      }
      if (pc) { const uint32_t cp = rl32(cpos, pc - 1u); if (lane_i == 0) cursor[s] = cp + 1u; }
      prev_nd0 = nd; prev_pc = pc; prev_all = pc == ncand;
      K9_WAVE_FENCE();
      prev_chg = chg_any;   // (chg itself is for the evaluating waves: which slots)
      nd += pc; n_dirty_rows += n_dirty; n_runs += 1u; n_slow += plain0 ? 0u : 1u;
      i_end = i_next; reason_end = reason;
      // publish: everything this run wrote (slots, bitmap, cursor, decision records) is in LDS before the word moves
      if (lane_i == 0) {
        Y.nd_at[(k + 1u) & 3u] = nd;
        Y.spec_ok[(k + 1u) & 3u] = prev_all ? 1u : 0u;
        if (stop) k9s_st(&Y.stop, 1u);
        k9s_st(&Y.seq_done, k + 1u);
      }
      if (stop) break;
    }
#ifdef KB_K9_TRACE
    tacc[4] = (uint32_t)(wall_clock64() - t_tab);
#endif
    if (lane == 0) {
      if (k9s_ld(&Y.err)) reason_end = KB_REASON_INTERNAL;
      k9s_st(&Y.stop, 1u);
      H.i = i_end; H.nd = nd; H.reason = reason_end; H.n_dirty_rows = n_dirty_rows; H.n_runs = n_runs; H.n_slow = n_slow;
    }
  //@@ dk_waves
  } else if (wave <= 4u) {
    K9S_SETPRIO(2);
    // =================================================== waves 1..4: the dirty slots ===================================================
    // Run q's dirty keys are evaluated EARLY, while run q - 1 is still being selected: against the slots as run q - 2 left them plus run
    // q - 1's candidates in their prepared state (one placement).  Most runs only consume clean candidates, once each: then the early
    // keys are the keys.  What run q - 1 changed otherwise (wave 0 lists those slots in chg) is evaluated again when it is committed.
    const uint32_t t = tid - 64u;
    for (uint32_t q = 0; q < K; q++) {
      K9S_RUN_HEADER(q)
      uint32_t *dko = X.dkb[q & 1u];
      uint32_t key = 0u, nd_early = 0u;
      const unsigned long long tw0 = K9S_NOW();
      if (q >= 1u) {
        if (!k9s_wait(Y, &Y.seq_done, q - 1u, nap_dk) || !k9s_wait(Y, &Y.seq_cand, q, nap_dk) || k9s_ld(&Y.stop)) break;
        nd_early = Y.nd_at[(q - 1u) & 3u] + Y.ncand_at[(q - 1u) & 3u];
        if (t < nd_early) {   // the shape against "their" dirty slot
          const K9St vs = k9_load(slots + (size_t)t * K9_NF);
          const K9Sc scp = k9_sc_preload(sh.active >> 2, gi, gr, a.NP, vs.node);
          key = k9_eval_v(a, sh, vs, scp, gi, gr, si, 0u, 0.0, si, nb, nmaskbits);
          dko[t] = key;
        }
        const uint32_t wmo = wave_max_u32(t < Y.nd_at[(q - 1u) & 3u] ? key : 0u);
        if (lane == 0) Y.mdko[q & 1u][wave - 1u] = wmo;
      }
      if (lane == 0) k9s_st(&Y.seq_early[wave - 1u], q + 1u);
      if (wave == 1u) K9S_TR(3, tw0);
      const unsigned long long tw1 = K9S_NOW();
      if (!k9s_wait(Y, &Y.seq_done, q, nap_dk) || k9s_ld(&Y.stop)) break;   // run q - 1 is committed: the slots are final
      if (wave == 1u) K9S_TR(2, tw1);
      const uint32_t nd = Y.nd_at[q & 3u];
      if (t < nd && (t >= nd_early || ((Y.chg[(q - 1u) & 1u][t >> 5] >> (t & 31)) & 1u))) {   // (q == 0: nd == 0)
        const K9St vs = k9_load(slots + (size_t)t * K9_NF);
        const K9Sc scp = k9_sc_preload(sh.active >> 2, gi, gr, a.NP, vs.node);
        key = k9_eval_v(a, sh, vs, scp, gi, gr, si, 0u, 0.0, si, nb, nmaskbits);
      }
      if (t >= nd) key = 0u;
      if (t < K9_MAXSLOTS) dko[t] = key;
      const uint32_t wm = wave_max_u32(key);
      if (lane == 0) { Y.mdk[q & 1u][wave - 1u] = wm; k9s_st(&Y.seq_dk[wave - 1u], q + 1u); }
    }
  //@@ prep_waves
  } else if (wave < K9S_PREP0 + K9S_PREPS) {
    // =================================================== waves 5..7: the candidates ===================================================
    typedef const unsigned long long __attribute__((address_space(1))) *gptr8;
    typedef const uint32_t __attribute__((address_space(1))) *gptr4;
    gptr8 g8[10];
    gptr8 gports;
    gptr4 gcls, gmaxp, gpodc, gnm;
    {
      const KbDev &d = *a.dev;
      g8[0] = (gptr8)reinterpret_cast<const unsigned long long *>(d.idle); g8[1] = (gptr8)reinterpret_cast<const unsigned long long *>(d.idle + d.NP);
      g8[2] = (gptr8)reinterpret_cast<const unsigned long long *>(d.rel); g8[3] = (gptr8)reinterpret_cast<const unsigned long long *>(d.rel + d.NP);
      g8[4] = (gptr8)reinterpret_cast<const unsigned long long *>(d.inv_acpu); g8[5] = (gptr8)reinterpret_cast<const unsigned long long *>(d.inv_amem);
      g8[6] = (gptr8)reinterpret_cast<const unsigned long long *>(d.acpu); g8[7] = (gptr8)reinterpret_cast<const unsigned long long *>(d.amem);
      g8[8] = (gptr8)reinterpret_cast<const unsigned long long *>(d.nzc); g8[9] = (gptr8)reinterpret_cast<const unsigned long long *>(d.nzm);
      gports = (gptr8)d.ports;
      gcls = (gptr4)d.ncls; gmaxp = (gptr4)reinterpret_cast<const uint32_t *>(d.maxpods); gpodc = (gptr4)reinterpret_cast<const uint32_t *>(d.podcnt); gnm = (gptr4)d.nmask;
    }
    K9Prep &P = X.prep[wave - K9S_PREP0];
    for (uint32_t m = wave - K9S_PREP0; m < K; m += K9S_PREPS) {
      K9S_RUN_HEADER(m)
      // ---- walk + fetch, two runs ahead: the runs in front that are not committed yet (m - 2 when the walk may start, and m - 1) can still
      //      take up to their own length of this shape's clean entries, so that many more are fetched: lane j = entry j (64 lanes).  With at
      //      most (rows committed so far) dirty nodes and a list of W + 1 entries the walk finds them unless the list ends (its 0 terminator):
      //      nf < want means every clean feasible node of the shape is among the nf.
      const uint32_t r1 = m >= 1u ? (uint32_t)X.runs[m] - (uint32_t)X.runs[m - 1u] : 0u, r2 = m >= 2u ? (uint32_t)X.runs[m - 1u] - (uint32_t)X.runs[m - 2u] : 0u;
      uint32_t want = r + r1 + r2, need = m >= 2u ? m - 2u : 0u;
      if (want > 64u) { want = r + r1; need = m >= 1u ? m - 1u : 0u; }   // r <= K9_SEL_MAXRUN = 32
      const unsigned long long tp0 = K9S_NOW();
      if (!k9s_wait(Y, &Y.seq_done, need, nap_prep) || k9s_ld(&Y.stop)) break;
      K9S_TR(1, tp0);
      uint32_t nf = 0;
      {
        uint32_t e_ = cursor[s];
        while (nf < want) {
          const uint32_t pos_ = e_ + lane;
          const uint32_t kk_ = (pos_ < Lp) ? lists[s * Lp + pos_] : 0u;
          const bool nz_ = kk_ != 0u;
          const uint32_t nn_ = nmaskbits - (kk_ & nmaskbits);
          const bool cl_ = nz_ && !((bitmap[(nz_ ? nn_ : 0u) >> 5] >> (nn_ & 31)) & 1u);
          const unsigned long long zeros_ = __ballot(!nz_);
          const uint32_t fz_ = zeros_ ? (uint32_t)__ffsll((unsigned long long)zeros_) - 1u : 64u;
          const unsigned long long clean_ = __ballot(cl_ && lane < fz_);   // entries behind the list's end do not count
          const uint32_t rank_ = (uint32_t)__popcll(clean_ & lt);
          if (((clean_ >> lane) & 1ull) && nf + rank_ < want) { P.ckey[nf + rank_] = kk_; P.cpos[nf + rank_] = pos_; }
          nf = min(want, nf + (uint32_t)__popcll(clean_));
          if (fz_ < 64u || e_ + 64u >= Lp) break;   // the list ended
          e_ += 64u;
        }
        K9_WAVE_FENCE();
      }
      const uint32_t key = (lane < nf) ? P.ckey[lane] : 0u, mypos = (lane < nf) ? P.cpos[lane] : 0u;
      const uint32_t n = nmaskbits - (key & nmaskbits);
      unsigned long long raw[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      uint32_t rcls = 0, rmaxp = 0, rpodc = 0, rnm = 0;
      if (lane < nf) {   // every field of entry `lane`, all loads in flight together
#pragma unroll
        for (int f = 0; f < 10; f++) raw[f] = g8[f][n];
        raw[F_PORTS] = gports ? gports[n] : 0ull;
        rcls = gcls[n]; rmaxp = gmaxp[n]; rpodc = gpodc[n]; rnm = gnm[n];
      }
      if (!plain0 && lane == 63) {   // an init container raised InitResreq above Resreq (rare): the row's own Resreq
        const KbDev &d_ = *a.dev;
        const uint32_t tk_ = desc[i0].task;
        for (int dd = 0; dd < a.R; dd++) rowres[(m % K9S_PREPS) * (uint32_t)a.R + (uint32_t)dd] = d_.t_res[(size_t)dd * d_.T + tk_];
      }
      K9_WAVE_FENCE();
      // ---- P2, for every fetched entry (which of them are the run's candidates is decided below): Allocate / Pipeline, NodeInfo.AddTask on
      //      the fetched state, key of the node after the placement
      K9St v;
      uint32_t kind = 0u, k1 = 0u;
      {
        double res0 = sh.init0, res1 = sh.init1;
        if (!plain0) { res0 = rres[0]; res1 = rres[1]; }
        double idle0 = u2d(raw[F_IDLE0]), idle1 = u2d(raw[F_IDLE1]), rel0 = u2d(raw[F_REL0]), rel1 = u2d(raw[F_REL1]);
        const K9Sc scn = k9_sc_preload(sh.active >> 2, gi, gr, a.NP, lane < nf ? n : 0u);   // the candidate's scalar dimensions: for the test below and the key
        if (!a.backfill) {   // allocate.go:160: InitResreq.LessEqual(node.Idle) -> Allocate, else Pipeline
          bool fi = le_eps(sh.init0, idle0, EPS_CPU) && le_eps(sh.init1, idle1, EPS_MEM);
          for (uint32_t aa = sh.active >> 2, dd = 0; aa; aa >>= 1, dd++)
            if (aa & 1u) fi = fi && le_eps(si[dd], k9_sci(scn, gi, a.NP, dd, lane < nf ? n : 0u), EPS_SCALAR);
          kind = fi ? 0u : 1u;
        }
        // NodeInfo.AddTask (api/node_info.go:172-212): Idle (Allocated) or Releasing (Pipelined) -= Resreq, pod joins ni.Tasks
        if (kind) { rel0 -= res0; rel1 -= res1; } else { idle0 -= res0; idle1 -= res1; }
        v.idle0 = idle0; v.idle1 = idle1; v.rel0 = rel0; v.rel1 = rel1;
        v.inv_ac = u2d(raw[F_INVAC]); v.inv_am = u2d(raw[F_INVAM]);
        v.ac = (double)(long long)raw[F_AC]; v.am = (double)(long long)raw[F_AM];
        v.nzc = (double)(long long)raw[F_NZC] + sh.nzc; v.nzm = (double)(long long)raw[F_NZM] + sh.nzm;
        v.ports = raw[F_PORTS] | sh.want;   // the pod's host ports join nodeinfo.UsedPorts()
        v.cls = rcls; v.node = lane < nf ? n : 0u; v.left = (int)rmaxp - (int)rpodc - 1;
        // the scalar part of the Sub reaches HBM when (and if) the candidate is consumed; the key is evaluated as if it had.
        // Sub returns early when the receiver's scalar map is nil (resource_info.go:148-153); a Pipeline ends the round, its
        // Releasing-side key is never read.
        const uint32_t adjm = (!kind && (rnm & 0x7FFFFFFFu)) ? km0 : 0u;
        if (lane < nf) {
          k1 = k9_eval_v(a, sh, v, scn, gi, gr, si, adjm, 1.0, rqv, nb, nmaskbits);   // the node's key once it is dirty
        }
      }
      // ---- the run's candidates: the first r entries whose node the runs in front leave alone, stored as its future dirty slots.  Settled
      //      AHEAD, while run m - 1 is being selected, on the assumption that it takes every candidate of its own (most runs do: then their nodes
      //      are exactly what gets dirty and the next slot is known); wave 0 says whether that held (spec_ok), and if not the same step runs
      //      again behind run m - 1 against the bitmap as it then is.
      K9Sel::Cand &CD = X.cand[m & 1u];
      auto settle = [&](bool ahead) {
        uint32_t nd0 = Y.nd_at[m & 3u];
        bool keep = lane < nf && !((bitmap[n >> 5] >> (n & 31)) & 1u);
        if (ahead) {   // m >= 1: run m - 1's candidates are final (seq_cand >= m); its nodes count as taken, its slots as used
          const K9Sel::Cand &CP = X.cand[(m - 1u) & 1u];
          const uint32_t ncp = Y.ncand_at[(m - 1u) & 3u];
          nd0 = Y.nd_at[(m - 1u) & 3u] + ncp;
          for (uint32_t i = 0; i < ncp; i++) keep = keep && (nmaskbits - (CP.ckey[i] & nmaskbits)) != n;
        }
        const unsigned long long kb = __ballot(keep);
        const uint32_t rho = (uint32_t)__popcll(kb & lt);
        const uint32_t ncand = min((uint32_t)__popcll(kb), r);
        if (keep && rho < r) {
          unsigned long long *st = slots + (size_t)(nd0 + rho) * K9_NF;
          st[F_IDLE0] = d2u(v.idle0); st[F_IDLE1] = d2u(v.idle1); st[F_REL0] = d2u(v.rel0); st[F_REL1] = d2u(v.rel1);
          st[F_INVAC] = raw[F_INVAC]; st[F_INVAM] = raw[F_INVAM]; st[F_AC] = d2u(v.ac); st[F_AM] = d2u(v.am);
          st[F_NZC] = d2u(v.nzc); st[F_NZM] = d2u(v.nzm); st[F_PORTS] = v.ports;
          st[F_CLS_LEFT] = (unsigned long long)rcls | ((unsigned long long)(uint32_t)v.left << 32);
          st[F_NODE_NMASK] = (unsigned long long)n | ((unsigned long long)rnm << 32);
          CD.ckey[rho] = key; CD.cpos[rho] = mypos;
          CD.ckind[rho] = kind; CD.ck1[rho] = k1; CD.crnm[rho] = rnm;
        }
        if (lane == 0) Y.ncand_at[m & 3u] = ncand;
      };
      if (m >= 1u) {
        if (!k9s_wait(Y, &Y.seq_cand, m, nap_prep) || k9s_ld(&Y.stop)) break;   // run m - 1's candidates stand (and run m - 2 is committed)
        settle(true);
        if (lane == 0) k9s_st(&Y.seq_spec, m + 1u);
      }
      const unsigned long long tp1 = K9S_NOW();
      if (!k9s_wait(Y, &Y.seq_done, m, nap_prep) || k9s_ld(&Y.stop)) break;
      K9S_TR(0, tp1);
      if (m == 0u || !Y.spec_ok[m & 3u]) {   // (otherwise wave 0 takes the candidates settled ahead and announces them itself)
        settle(false);
        if (lane == 0) k9s_st(&Y.seq_cand, m + 1u);
      }
    }
  }
  //@@ epilogue
#ifdef KB_K9_TRACE
  if (tid == 0) {
    unsigned long long *tw = reinterpret_cast<unsigned long long *>(a.result);
    tw[5] = (unsigned long long)tacc[0] | ((unsigned long long)tacc[1] << 32);
    tw[6] = (unsigned long long)tacc[2] | ((unsigned long long)tacc[3] << 32);
    tw[7] = (unsigned long long)tacc[4] | ((unsigned long long)tacc[5] << 32);
    tw[13] = (unsigned long long)tacc[6] | ((unsigned long long)tacc[7] << 32);
    tw[14] = (unsigned long long)tacc[8] | ((unsigned long long)tacc[9] << 32);
  }
#endif
  __syncthreads();   // wave 0 has left the loop: the header and the statistics words are final
#ifdef KB_K9_TRACE
  if (tid == 0) {
    unsigned long long *tw = reinterpret_cast<unsigned long long *>(a.result);
    tw[4] = (unsigned long long)X.tr[0] | ((unsigned long long)X.tr[1] << 32);
    tw[15] = (unsigned long long)X.tr[2] | ((unsigned long long)X.tr[3] << 32);
  }
#endif
  k9_epilogue(a, lo, k9_smem, tid, t_start, X.stat[0] | (X.stat[1] << 16), X.stat[2] | (X.stat[3] << 16), true, job_of_my_row);
}

void kb_launch_commit_sel(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  static bool lds_set[64] = {};
  k9_allow_full_lds(reinterpret_cast<const void *>(k_commit_select), lds_set);
  size_t sh = k9_layout(r.n_rows, r.n_mrows, r.L, d.NP, d.R, true).total;
  if (r.lists_ready != nullptr) sh = std::max(sh, kb_repair_lds_bytes(d.NP, K9_THREADS));   // the repair workgroups' own tables: more than a window of a few rows needs (call 20)
  K9KernArgs ka;
  k9_fill_args(ka, d, r);
  static_assert(KB_REPAIR_THREADS == K9_THREADS, "the repair workgroups of a fused launch are workgroups of this kernel");
  // a round whose lists are repaired beside its commit: one more workgroup per matrix row (they only need kb_repair_lds_bytes of the block)
  uint32_t grid = KB_WARM_GRID;
  if (r.lists_ready != nullptr) grid = std::max<uint32_t>(grid, r.n_mrows + (r.n_mrows - 1u) / 7u + 1u);   // row j: workgroup j + j / 7 + 1
  // a PROFILING aid (scripts/gpu_r6.sh profile; never set in operation): the commit workgroup alone in its launch, so that a rocprofv3 PMC pass
  // counts ITS waves' instructions and cycles — with the L2 helpers and the repair rows in the launch the per-dispatch sums are of ~30 workgroups
  // that mostly idle (round 5: "parked 0.992").  Needs the repair as a launch of its own (KB_FUSE_REPAIR=0); costs the warm L2.
  static const bool solo = [] { const char *v = getenv("KB_PROFILE_SOLO_COMMIT"); return v && v[0] == '1'; }();
  if (solo && r.lists_ready == nullptr) grid = 1u;
  hipLaunchKernelGGL(k_commit_select, dim3(grid), dim3(K9_THREADS), sh, (hipStream_t)stream, ka);
}
