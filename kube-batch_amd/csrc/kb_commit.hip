// kb_commit.hip — K5, the sequential commit of one window (gfx950 / CDNA4, wave64).
//
// The reference places one task after another (allocate.go:129-193 / backfill.go:44-67): task k+1 sees the node state
// task k left behind.  Per row the answer is
//     best = max( best CLEAN node of the row's shape  = first entry of the shape's sorted candidate list (K3) whose node
//                                                       this round has not touched yet,
//                 best DIRTY node                      = max over the nodes this round already changed of the key
//                                                       re-evaluated against their LIVE state ).
// A single wave executes one instruction every ~6 cycles, so a row-at-a-time loop pays ~6 cycles x (every instruction of
// fit + score + bookkeeping) per row.  Rows, however, arrive in RUNS of one shape (the tasks of a gang), and within a run
// everything that is expensive is data-parallel:
//
//   prepare   wave 0, between two runs: walks the shape's candidate list for the run's first r clean entries (ballot over the
//             dirty bitmap); lane j then issues every load of candidate j's node state (13 + 2(R-2) loads in flight per lane),
//             which travel while the workgroup passes the barrier;
//   evaluate  one evaluation deep, in parallel: waves 1..4: key(shape, dirty slot t), one slot per thread (<= 256 slots);
//             wave 0, lane j = candidate j: Allocate / Pipeline (allocate.go:160), NodeInfo.AddTask (api/node_info.go:172-212)
//             on the fetched state, and the key of the node AFTER the placement — the key it competes with once it is dirty;
//   rows      wave 0, sequential but O(1) per clean row: winner = max(best dirty key, next unconsumed clean candidate); the best
//             dirty key is a scalar that a clean winner updates with one max (its post-placement key is ready); only a dirty
//             winner costs an evaluation (AddTask on its slot in LDS, key re-evaluated, wave maximum rebuilt).
// Two workgroup barriers per run.
//
// No speculation, nothing to roll back: a candidate that is not consumed simply stays clean.  Everything the loop touches
// lives in LDS (slots, lists, row descriptors, dirty bitmap); HBM sees the dirty slots once, in the epilogue.
//
// Keys are 32-bit here: (score + 1) << node_bits | (2^node_bits - 1 - node), 0 = infeasible; integer max = highest score,
// then lowest node index = util.SelectBestNode with the canonical tie-break (scheduler_helper.go:188-208).
#include "kb_k9.hpp"

__global__ void __launch_bounds__(K9_THREADS) k_commit_run(const K9KernArgs ka) {
  KbCommitArgs a = ka.hot;
  {   // only `a` is named in the loops (SGPRs); the two views are read through the kernel-argument segment on rare paths
    const unsigned char __attribute__((address_space(4))) *kp = (const unsigned char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    a.dev = (const KbDev *)(kp + offsetof(K9KernArgs, dev));
    a.round = (const KbRound *)(kp + offsetof(K9KernArgs, round));
  }
  if (k9_preamble(a)) return;
  extern __shared__ __align__(16) unsigned char k9_smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t S = a.n_mrows, W = a.n_rows;
  const K9Layout lo = k9_layout(W, S, a.L, a.NP, a.R);
  unsigned char *k9_base_ = k9_smem;
  K9_LDS_VIEWS(lo)
  (void)shp;
  const unsigned long long t_start = wall_clock64();
  if (!k9_prologue(a, lo, k9_smem, tid, K9_MAXRUN)) {   // (only a launch that carries its own repair workgroups can fail here: the selection kernel's)
    if (tid == 0) k9_publish_skipped(a);
    return;
  }


#ifdef KB_K9_TRACE
  uint32_t tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  // wave 0's registers: the fetched (raw) node state of candidate `lane` of the run being prepared
  unsigned long long raw[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t rcls = 0, rmaxp = 0, rpodc = 0, rnm = 0;

  // wave 0, between two runs: the next run's parameters, its clean candidates (walk of the shape's list against the dirty
  // bitmap), and the fetch of the candidates' node state — lane j pulls every field of candidate j, all loads in flight together,
  // while the workgroup goes through the barrier and the other waves start on the dirty slots
#define K9_PREPARE_NEXT(i_next, nd_next, reason_in)                                                                    \
  do {                                                                                                                 \
    uint32_t rsn_ = (reason_in), stop_ = (rsn_ != KB_REASON_DONE) ? 1u : 0u, nc_ = 0, s_ = 0, r_ = 0, fl_ = 0, km_ = 0; \
    if (!stop_ && (i_next) < W) {                                                                                      \
      const uint4 ri_ = rinfo[(i_next)];                                                                               \
      r_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri_.x); s_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri_.y); \
      fl_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri_.z); km_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri_.w); \
      if (a.has_aff && !a.backfill && (fl_ & 2u) && (i_next) != 0) {                                                   \
        /* a row whose score is normalised over its feasible set (preferred node affinity) is exact only against a fresh */ \
        /* matrix: it may be the first row of a round, nothing else */                                                 \
        rsn_ = KB_REASON_RENORM; stop_ = 1;                                                                            \
      } else {                                                                                                         \
        uint32_t e_ = cursor[s_];                                                                                      \
        while (nc_ < r_) {                                                                                             \
          const uint32_t pos_ = e_ + lane;                                                                             \
          const uint32_t kk_ = (pos_ < Lp) ? lists[s_ * Lp + pos_] : 0u;                                               \
          const bool nz_ = kk_ != 0u;                                                                                  \
          const uint32_t nn_ = nmaskbits - (kk_ & nmaskbits);                                                          \
          const bool cl_ = nz_ && !((bitmap[(nz_ ? nn_ : 0u) >> 5] >> (nn_ & 31)) & 1u);                               \
          const unsigned long long zeros_ = __ballot(!nz_);                                                            \
          const uint32_t fz_ = zeros_ ? (uint32_t)__ffsll((unsigned long long)zeros_) - 1u : 64u;                      \
          const unsigned long long clean_ = __ballot(cl_ && lane < fz_);   /* entries behind the list's end do not count */ \
          const uint32_t rank_ = (uint32_t)__popcll(clean_ & ((1ull << lane) - 1ull));                                 \
          if (((clean_ >> lane) & 1ull) && nc_ + rank_ < r_) { ckey[nc_ + rank_] = kk_; cpos[nc_ + rank_] = pos_; }    \
          nc_ = min(r_, nc_ + (uint32_t)__popcll(clean_));                                                             \
          if (fz_ < 64u || e_ + 64u >= Lp) break;   /* the list ended (or, never with L = rows + 1, ran out) */         \
          e_ += 64u;                                                                                                   \
        }                                                                                                              \
        K9_WAVE_FENCE();                                                                                               \
        if (lane < nc_) {                                                                                              \
          const uint32_t n_ = nmaskbits - (ckey[lane] & nmaskbits);                                                    \
          _Pragma("unroll") for (int f = 0; f < 10; f++) raw[f] = g8[f][n_];                                           \
          raw[F_PORTS] = gports ? gports[n_] : 0ull;                                                                   \
          rcls = gcls[n_]; rmaxp = gmaxp[n_]; rpodc = gpodc[n_]; rnm = gnm[n_];                                        \
        }                                                                                                              \
        const bool plain_ = (fl_ & 1u) && (km_ == 0u || (fl_ & 4u));                                                   \
        if (!plain_ && lane == 63) {   /* an init container raised InitResreq above Resreq (rare): the row's own Resreq */ \
          const KbDev &d_ = *a.dev;                                                                                    \
          const uint32_t tk_ = desc[(i_next)].task;                                                                    \
          for (int dd = 0; dd < a.R; dd++) rowres[dd] = d_.t_res[(size_t)dd * d_.T + tk_];                             \
        }                                                                                                              \
      }                                                                                                                \
    } else {                                                                                                           \
      stop_ = 1;                                                                                                       \
    }                                                                                                                  \
    if (lane == 0) {                                                                                                   \
      H.i = (i_next); H.nd = (nd_next); H.reason = rsn_; H.stop = stop_; H.ncand = nc_;                                \
      H.cur_s = s_; H.cur_r = r_; H.cur_fl = fl_; H.cur_km = km_;                                                      \
    }                                                                                                                  \
  } while (0)

  typedef const unsigned long long __attribute__((address_space(1))) *gptr8;
  typedef const uint32_t __attribute__((address_space(1))) *gptr4;
  gptr8 g8[10];
  gptr8 gports;
  gptr4 gcls, gmaxp, gpodc, gnm;
  gptrd gi, gr;
  {
    const KbDev &d = *a.dev;
    g8[0] = (gptr8)reinterpret_cast<const unsigned long long *>(d.idle); g8[1] = (gptr8)reinterpret_cast<const unsigned long long *>(d.idle + d.NP);
    g8[2] = (gptr8)reinterpret_cast<const unsigned long long *>(d.rel); g8[3] = (gptr8)reinterpret_cast<const unsigned long long *>(d.rel + d.NP);
    g8[4] = (gptr8)reinterpret_cast<const unsigned long long *>(d.inv_acpu); g8[5] = (gptr8)reinterpret_cast<const unsigned long long *>(d.inv_amem);
    g8[6] = (gptr8)reinterpret_cast<const unsigned long long *>(d.acpu); g8[7] = (gptr8)reinterpret_cast<const unsigned long long *>(d.amem);
    g8[8] = (gptr8)reinterpret_cast<const unsigned long long *>(d.nzc); g8[9] = (gptr8)reinterpret_cast<const unsigned long long *>(d.nzm);
    gports = (gptr8)d.ports;
    gi = (gptrd)d.idle; gr = (gptrd)d.rel;
    gcls = (gptr4)d.ncls; gmaxp = (gptr4)reinterpret_cast<const uint32_t *>(d.maxpods); gpodc = (gptr4)reinterpret_cast<const uint32_t *>(d.podcnt); gnm = (gptr4)d.nmask;
  }
  if (wave == 0) K9_PREPARE_NEXT(0u, 0u, (uint32_t)KB_REASON_DONE);

  // ---------------- run loop (uniform across the workgroup): two barriers per run ----------------
  for (;;) {
    K9_STAMP(9);
    __syncthreads();   // B1: the run's parameters and candidates, and every slot the previous run wrote, are visible
    K9_STAMP(0);
    if (H.stop) break;
    const uint32_t i0 = H.i, nd = H.nd, ncand = H.ncand, s = H.cur_s, r = H.cur_r, fl0 = H.cur_fl, km0 = H.cur_km;
    const bool plain0 = (fl0 & 1u) && (km0 == 0u || (fl0 & 4u));
    const K9Shape sh = shapes[s];
    const double *si = sinit + (size_t)s * RS;
    const double *rqv = plain0 ? si : rowres + 2;   // the rows' scalar Resreq
    // ---- evaluation phase, one evaluation deep: waves 1..4 the shape against "their" dirty slot, wave 0 the candidates
    if (wave >= 1 && tid - 64u < nd) {
      const uint32_t t = tid - 64u;
      const K9St vs = k9_load(slots + (size_t)t * K9_NF);
      dk[t] = k9_eval_v(a, sh, vs, k9_sc_preload(sh.active >> 2, gi, gr, a.NP, vs.node), gi, gr, si, 0u, 0.0, si, nb, nmaskbits);
    }
    uint32_t ck = 0, k1 = 0, ckind = 0;
    double res0 = sh.init0, res1 = sh.init1;
    if (wave == 0) {
      if (!plain0) { res0 = rowres[0]; res1 = rowres[1]; }
      // P2: lane j = candidate j: Allocate / Pipeline, NodeInfo.AddTask on the fetched state, key of the node after the placement
      if (lane < ncand) {
        ck = ckey[lane];
        const uint32_t n = nmaskbits - (ck & nmaskbits);
        unsigned long long *st = slots + (size_t)(nd + lane) * K9_NF;
        double idle0 = u2d(raw[F_IDLE0]), idle1 = u2d(raw[F_IDLE1]), rel0 = u2d(raw[F_REL0]), rel1 = u2d(raw[F_REL1]);
        uint32_t kind = 0;
        const K9Sc scn = k9_sc_preload(sh.active >> 2, gi, gr, a.NP, n);   // the candidate's scalar dimensions: for the test below and the key
        if (!a.backfill) {   // allocate.go:160: InitResreq.LessEqual(node.Idle) -> Allocate, else Pipeline
          bool fi = le_eps(sh.init0, idle0, EPS_CPU) && le_eps(sh.init1, idle1, EPS_MEM);
          for (uint32_t aa = sh.active >> 2, dd = 0; aa; aa >>= 1, dd++)
            if (aa & 1u) fi = fi && le_eps(si[dd], k9_sci(scn, gi, a.NP, dd, n), EPS_SCALAR);
          kind = fi ? 0u : 1u;
        }
        // NodeInfo.AddTask (api/node_info.go:172-212): Idle (Allocated) or Releasing (Pipelined) -= Resreq, pod joins ni.Tasks
        if (kind) { rel0 -= res0; rel1 -= res1; } else { idle0 -= res0; idle1 -= res1; }
        K9St v;
        v.idle0 = idle0; v.idle1 = idle1; v.rel0 = rel0; v.rel1 = rel1;
        v.inv_ac = u2d(raw[F_INVAC]); v.inv_am = u2d(raw[F_INVAM]);
        v.ac = (double)(long long)raw[F_AC]; v.am = (double)(long long)raw[F_AM];
        v.nzc = (double)(long long)raw[F_NZC] + sh.nzc; v.nzm = (double)(long long)raw[F_NZM] + sh.nzm;
        v.ports = raw[F_PORTS] | sh.want;   // the pod's host ports join nodeinfo.UsedPorts()
        v.cls = rcls; v.node = n; v.left = (int)rmaxp - (int)rpodc - 1;
        st[F_IDLE0] = d2u(v.idle0); st[F_IDLE1] = d2u(v.idle1); st[F_REL0] = d2u(v.rel0); st[F_REL1] = d2u(v.rel1);
        st[F_INVAC] = raw[F_INVAC]; st[F_INVAM] = raw[F_INVAM]; st[F_AC] = d2u(v.ac); st[F_AM] = d2u(v.am);
        st[F_NZC] = d2u(v.nzc); st[F_NZM] = d2u(v.nzm); st[F_PORTS] = v.ports;
        st[F_CLS_LEFT] = (unsigned long long)rcls | ((unsigned long long)(uint32_t)v.left << 32);
        st[F_NODE_NMASK] = (unsigned long long)n | ((unsigned long long)rnm << 32);
        // the scalar part of the Sub reaches HBM when (and if) the candidate is consumed; the key is evaluated as if it had.
        // Sub returns early when the receiver's scalar map is nil (resource_info.go:148-153); a Pipeline ends the round, its
        // Releasing-side key is never read.
        const uint32_t adjm = (!kind && (rnm & 0x7FFFFFFFu)) ? km0 : 0u;
        ckind = kind;
        k1 = k9_eval_v(a, sh, v, scn, gi, gr, si, adjm, 1.0, rqv, nb, nmaskbits);   // the node's key once it is dirty
      }
    }
    K9_STAMP2(1, 7, (sh.active >> 2) != 0u || km0 != 0u);
    __syncthreads();   // B2: the dirty keys are in LDS
    K9_STAMP(2);

    // ---- P3 (wave 0): the rows of the run, in order; then the next run is prepared
    if (wave == 0) {
      // dirty keys of the shape: old slot t in lane t & 63, register t >> 6; the run's own new slots: k1 of lanes < pc
      uint32_t d0 = (lane < nd) ? dk[lane] : 0u, d1 = (lane + 64 < nd) ? dk[lane + 64] : 0u;
      uint32_t d2 = (lane + 128 < nd) ? dk[lane + 128] : 0u, d3 = (lane + 192 < nd) ? dk[lane + 192] : 0u;
      uint32_t m = wave_max_u32(max(max(d0, d1), max(d2, d3)));   // best dirty key; clean winners update it in O(1)
      uint32_t pc = 0, j = 0, reason = KB_REASON_DONE, n_dirty = 0, sc_dirty = 0;
      for (; j < r; j++) {
        const uint32_t c = (pc < ncand) ? rl32(ck, pc) : 0u;
        if (m == 0u && c == 0u) {
          if (a.backfill) {   // backfill.go:50-66: no node passes the predicates -> the task stays Pending
            if (lane == 0) ldec[i0 + j] = (unsigned long long)KB_NONE_U32;
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
          if (lane == 0) {
            ldec[i0 + j] = (unsigned long long)n | ((unsigned long long)kind << 32);
            atomicOr(&bitmap[n >> 5], 1u << (n & 31));
          }
          if (km0) {   // the scalar dimensions Resreq names: Idle / Releasing in HBM (lane 16 + d' takes dimension d' + 2)
            const uint32_t nmc = rl32(rnm, pc);
            const bool has_map = kind ? (nmc >> 31) : (nmc & 0x7FFFFFFFu);
            if (has_map && lane >= 16 && lane < 16 + RS && ((km0 >> (lane - 16)) & 1u)) k9_sc_sub(kind ? gr : gi, a.NP, lane - 16, n, rqv[lane - 16]);
            sc_dirty = 1;
          }
          pc++;
        } else {       // a node this round already changed wins: AddTask on its slot, re-evaluate it
          K9_COUNT(8, 1u);
          const uint32_t own = (lane < pc) ? k1 : 0u;
          const unsigned long long who = __ballot(d0 == m || d1 == m || d2 == m || d3 == m || own == m);
          const uint32_t L = (uint32_t)__ffsll((unsigned long long)who) - 1u;
          const bool is_new = L < pc && rl32(k1, L) == m;
          const uint32_t wsel = (rl32(d0, L) == m) ? 0u : (rl32(d1, L) == m) ? 1u : (rl32(d2, L) == m) ? 2u : 3u;
          const uint32_t x = is_new ? nd + L : L + 64u * wsel;
          unsigned long long *st = slots + (size_t)x * K9_NF;
          // ---- a chain: bin-packing scores keep the same node on top for several rows of the run.  Lane t evaluates the node
          // after t + 1 placements of this shape (one evaluation pass for the whole chain); the chain runs while the node's key
          // still beats every other dirty key and the next clean candidate and the placement is an Allocate.  Exact because the
          // quantities involved are integers (checked): Idle - (t + 1) * Resreq equals t + 1 successive subtractions.
          const uint32_t rem = r - j;
          if (rem >= 2u && plain0 && !a.backfill) {
            const K9St v0 = k9_load(st);
            const uint32_t nm0 = (uint32_t)(st[F_NODE_NMASK] >> 32);
            const uint32_t adjm = (nm0 & 0x7FFFFFFFu) ? km0 : 0u;   // Idle has a scalar map: Sub lowers the dimensions Resreq names
            bool exact = v0.idle0 == trunc(v0.idle0) && v0.idle1 == trunc(v0.idle1) && res0 == trunc(res0) && res1 == trunc(res1) &&
                         fabs(v0.idle0) < 4.0e15 && fabs(v0.idle1) < 4.0e15 && res0 < 3.0e13 && res1 < 3.0e13;
            // the node's scalar dimensions the chain reads: the ones the shape tests, the ones its Resreq lowers (loaded once, together)
            const K9Sc scc = k9_sc_preload((sh.active >> 2) | adjm, gi, gr, a.NP, v0.node);
            for (uint32_t mm = adjm, dd = 0; mm; mm >>= 1, dd++)
              if (mm & 1u) { const double id = k9_sci(scc, gi, a.NP, dd, v0.node), rq = si[dd]; exact = exact && id == trunc(id) && rq == trunc(rq) && fabs(id) < 4.0e15 && rq < 3.0e13; }
            if (exact) {
              const double tb = (double)lane, ta = (double)(lane + 1);   // placements before / after step `lane`
              // Allocate at step t needs InitResreq <= Idle after t placements (allocate.go:160)
              bool fit = le_eps(sh.init0, v0.idle0 - tb * res0, EPS_CPU) && le_eps(sh.init1, v0.idle1 - tb * res1, EPS_MEM);
              for (uint32_t aa = sh.active >> 2, dd = 0; aa; aa >>= 1, dd++)
                if (aa & 1u) { double id = k9_sci(scc, gi, a.NP, dd, v0.node); if ((adjm >> dd) & 1u) id -= tb * si[dd]; fit = fit && le_eps(si[dd], id, EPS_SCALAR); }
              K9St v = v0;
              v.idle0 = v0.idle0 - ta * res0; v.idle1 = v0.idle1 - ta * res1;
              v.nzc = v0.nzc + ta * sh.nzc; v.nzm = v0.nzm + ta * sh.nzm;
              v.ports = v0.ports | sh.want;
              v.left = v0.left - (int)(lane + 1);
              const uint32_t kt = k9_eval_v(a, sh, v, scc, gi, gr, si, adjm, ta, si, nb, nmaskbits);   // key after t + 1 placements
              // the best alternative: every other dirty key, and the next clean candidate
              const uint32_t others = max(max(max(lane == L && !is_new && wsel == 0 ? 0u : d0, lane == L && !is_new && wsel == 1 ? 0u : d1),
                                              max(lane == L && !is_new && wsel == 2 ? 0u : d2, lane == L && !is_new && wsel == 3 ? 0u : d3)),
                                          (lane < pc && !(is_new && lane == L)) ? k1 : 0u);
              const uint32_t alt = max(wave_max_u32(others), c);
              const uint32_t kprev = (uint32_t)__shfl_up((int)kt, 1);   // key before step t (t >= 1)
              const bool okt = lane < rem && fit && (lane == 0 || kprev > alt);
              const unsigned long long bad = ~__ballot(okt);
              const uint32_t Lc = bad ? (uint32_t)__ffsll((unsigned long long)bad) - 1u : 64u;
              if (Lc >= 1u) {
                const uint32_t n = v0.node;
                if (lane < Lc) ldec[i0 + j + lane] = (unsigned long long)n;   // kind 0
                if (lane == Lc - 1u) {   // the node after the chain
                  st[F_IDLE0] = d2u(v.idle0); st[F_IDLE1] = d2u(v.idle1); st[F_NZC] = d2u(v.nzc); st[F_NZM] = d2u(v.nzm);
                  st[F_PORTS] = v.ports;
                  st[F_CLS_LEFT] = (unsigned long long)v.cls | ((unsigned long long)(uint32_t)v.left << 32);
                  for (uint32_t mm = adjm, dd = 0; mm; mm >>= 1, dd++)
                    if (mm & 1u) k9_sc_sub(gi, a.NP, dd, v0.node, ta * si[dd]);   // after every evaluation above has read the old value
                }
                if (adjm) sc_dirty = 1;
                const uint32_t nk = rl32(kt, Lc - 1u);
                if (lane == L) {
                  if (is_new) k1 = nk;
                  else if (wsel == 0) d0 = nk; else if (wsel == 1) d1 = nk; else if (wsel == 2) d2 = nk; else d3 = nk;
                }
                K9_WAVE_FENCE();
                m = wave_max_u32(max(max(max(d0, d1), max(d2, d3)), (lane < pc) ? k1 : 0u));
                n_dirty += Lc;
                j += Lc - 1u;   // the loop header adds the last one
                continue;
              }
            }
          }
          // one row: lane f holds field f of the slot; lanes 16 + d' look at the scalar dimension d' + 2 in HBM when the shape or
          // the row names one
          const bool sc_lane = lane >= 16 && lane < 16 + RS;
          const uint32_t sd = sc_lane ? lane - 16 : 0;
          unsigned long long cur8 = 0ull;
          if (lane < K9_NF) cur8 = st[lane];
          const uint32_t nm = (uint32_t)(rl64(cur8, F_NODE_NMASK) >> 32);
          const uint32_t n = (uint32_t)rl64(cur8, F_NODE_NMASK);
          // the scalar dimensions the new key will read (only when a key is needed: not on the run's last row), in flight with the vote's loads
          const K9Sc scs = k9_sc_preload((j + 1u < r) ? (sh.active >> 2) : 0u, gi, gr, a.NP, n);
          kind = 0;
          if (!a.backfill) {
            bool ok = true;
            if (lane == F_IDLE0) ok = le_eps(sh.init0, u2d(cur8), EPS_CPU);
            else if (lane == F_IDLE1) ok = le_eps(sh.init1, u2d(cur8), EPS_MEM);
            else if (sc_lane && ((sh.active >> (2 + sd)) & 1u)) ok = le_eps(si[sd], k9_sc(gi, a.NP, sd, n), EPS_SCALAR);
            kind = __ballot(!ok) ? 1u : 0u;
          }
          const uint32_t has_map = kind ? (nm >> 31) : (nm & 0x7FFFFFFFu);
          const uint32_t f0 = kind ? F_REL0 : F_IDLE0;
          if (lane == f0) cur8 = d2u(u2d(cur8) - res0);
          else if (lane == f0 + 1) cur8 = d2u(u2d(cur8) - res1);
          else if (lane == F_NZC) cur8 = d2u(u2d(cur8) + sh.nzc);
          else if (lane == F_NZM) cur8 = d2u(u2d(cur8) + sh.nzm);
          else if (lane == F_PORTS) cur8 |= sh.want;
          else if (lane == F_CLS_LEFT) cur8 -= (1ull << 32);   // one more pod on the node
          if (lane < K9_NF) st[lane] = cur8;
          if (lane == 0) ldec[i0 + j] = (unsigned long long)n | ((unsigned long long)kind << 32);
          K9_WAVE_FENCE();
          // the new key first (scalar part of the Sub as an adjustment), then the Sub itself goes to HBM.  The last row of a run (and a
          // Pipeline, which ends the round) needs no new key: the next run evaluates every slot against ITS shape anyway
          const bool more = j + 1u < r && !kind;
          const uint32_t adjm1 = (!kind && has_map) ? km0 : 0u;
          uint32_t nk = 0u;
          if (more) nk = k9_eval_v(a, sh, k9_load(st), scs, gi, gr, si, adjm1, 1.0, rqv, nb, nmaskbits);   // uniform: every lane computes the same key
          if (km0 && has_map) {
            if (sc_lane && ((km0 >> sd) & 1u)) k9_sc_sub(kind ? gr : gi, a.NP, sd, n, rqv[sd]);
            sc_dirty = 1;
          }
          if (more) {
            if (lane == L) {
              if (is_new) k1 = nk;
              else if (wsel == 0) d0 = nk; else if (wsel == 1) d1 = nk; else if (wsel == 2) d2 = nk; else d3 = nk;
            }
            m = wave_max_u32(max(max(max(d0, d1), max(d2, d3)), (lane < pc) ? k1 : 0u));
          }
          n_dirty++;
        }
        if (kind) { j++; reason = KB_REASON_PIPELINED; break; }   // a Pipeline ends the speculated order: the host re-plans
      }
      K9_STAMP2(3, 5, (sh.active >> 2) != 0u || km0 != 0u);
      K9_COUNT(6, ((sh.active >> 2) != 0u || km0 != 0u) ? 1u : 0u);
      if (sc_dirty) __threadfence();   // the scalar atomics have reached L2 before any other wave evaluates against these nodes
      if (lane == 0) {
        if (pc) cursor[s] = cpos[pc - 1] + 1;
        H.n_dirty_rows += n_dirty; H.n_runs += 1; H.n_slow += plain0 ? 0u : 1u;
      }
      K9_WAVE_FENCE();
      K9_PREPARE_NEXT(i0 + j, nd + pc, reason);
      K9_STAMP(4);
    }
  }
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


  k9_epilogue(a, lo, k9_smem, tid, t_start, 0u, 0u);
}

size_t kb_commit_smem_bytes(uint32_t n_rows, uint32_t n_shapes, uint32_t NP, int R) {   // the window is planned for either of the two run-at-a-time kernels (the selection kernel's block included)
  return k9_layout(n_rows, n_shapes, n_rows + 1, NP, R, true).total;
}

void kb_launch_commit(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  static bool lds_set[64] = {};
  k9_allow_full_lds(reinterpret_cast<const void *>(k_commit_run), lds_set);
  const size_t sh = k9_layout(r.n_rows, r.n_mrows, r.L, d.NP, d.R).total;
  K9KernArgs ka;
  k9_fill_args(ka, d, r);
  hipLaunchKernelGGL(k_commit_run, dim3(KB_WARM_GRID), dim3(K9_THREADS), sh, (hipStream_t)stream, ka);
}
