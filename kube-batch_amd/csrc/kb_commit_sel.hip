// kb_commit_sel.hip — K5, the run SELECTION commit of one window (gfx950 / CDNA4, wave64; KB_COMMIT_SELECT).
//
// Same contract and same decisions as kb_commit.hip (the reference's sequential placement, allocate.go:129-193 / backfill.go:44-67, a
// run of same-shape rows at a time), two differences in how a run is worked:
//
// 1. The rows of a run are ONE selection instead of a loop over the rows.  Every candidate node of a run of r plain rows — the r best clean
//    entries of the shape's list and the dirty slots whose key is above the r-th of them — has its own key sequence key(n, j): its key
//    for the shape after j placements of the shape, while the next placement still passes the predicates (api/node_info.go:172-212
//    applied j times).  It depends on that node's state only.  With eff(n, j) = min key(n, 0..j), the serial loop's picks are the first
//    r of all (n, j) entries in (eff descending, j ascending) order — keys carry the node index, so entries of different nodes never tie:
//    a node picked at key s was the maximum with the lowest index among equals (util.SelectBestNode, scheduler_helper.go:188-208, with
//    the canonical tie-break); while its next keys stay >= s it wins again at once (nobody else changed) — the steps on which eff stays s —
//    and when its key falls below s the prefix minimum is the real key again.  A Pipeline entry (allocate.go:160: InitResreq no longer
//    fits Idle) ends its node's sequence; picked, it ends the round behind its row.  Step 0 of every candidate (and step 1 of the clean
//    ones) comes out of the parallel evaluation phase; deeper steps are walked by the lanes of wave 0, several steps per candidate in
//    one pass (most runs need none: every pick is a clean candidate's first placement); the order is a rank by count over at most 64 entries.  Nothing is written before the picks are known, so any limit of the tables simply hands
//    the run to the serial loop of kb_commit.hip, kept here verbatim.  (tests/run_selection_model.py and the emulated launch of
//    tests/host_harness/device_emu.cpp hold the claim to the serial loop on the CPU.)
//
// 2. A run is PREPARED while its predecessor is committed.  Two "prep" waves alternate: during the evaluation phase of run k one of them
//    walks run k + 1's candidate list and issues the fetch of its candidates' node state (r' + r of them: run k may still take up to r
//    of them), the other one — which did the same for run k one iteration earlier — drops the entries run k - 1 took (dirty bitmap),
//    applies NodeInfo.AddTask to the first r survivors and evaluates their keys after the placement.  Wave 0 only selects.  What is
//    left on the critical path of a run: one evaluation (dirty slots and candidates side by side), two barriers, the selection.
//
//   waves 1..4   key(shape, dirty slot t) and the kind of the slot's next placement
//   wave 5 / 6   alternating: candidates of the CURRENT run (filter, AddTask, key)  |  walk + fetch for the NEXT run
//   wave 0       selection (or the serial loop), decision records, cursors, the next run's header
#include "kb_k9.hpp"

#define K9S_PREP0 5u   // waves K9S_PREP0 and K9S_PREP0 + 1 prepare the runs of even / odd number

// make EXTRA=-DKB_K9_TRACE: cycles wave 0 spends per phase — 0: barrier 1, 1: the evaluation phase (wave 0 waits at barrier 2), 3: rows (serial
// loop, tail), 4: -, 5: entries + first rank, 6: deep passes (with their ranks), 7: picks + AddTask, 8: runs through the all-clean path, 9: loop top
__global__ void __launch_bounds__(K9_THREADS) k_commit_select(const K9KernArgs ka) {
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
  const K9Layout lo = k9_layout(W, S, a.L, a.NP, a.R, true);
  unsigned char *k9_base_ = k9_smem;
  K9_LDS_VIEWS(lo)
  (void)shp; (void)S;
  K9Sel &X = *reinterpret_cast<K9Sel *>(k9_smem + lo.sel);
  const unsigned long long t_start = wall_clock64();
  k9_prologue(a, lo, k9_smem, tid, K9_SEL_MAXRUN);
  if (tid < 4) { X.stat[tid] = 0u; X.tr[tid] = 0u; }   // first touched behind the loop's first barrier

#ifdef KB_K9_TRACE
  uint32_t tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#endif
  // a prep wave's registers: the fetched (raw) node state of entry `lane` of the run it prepared
  unsigned long long raw[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t rcls = 0, rmaxp = 0, rpodc = 0, rnm = 0;
  const unsigned long long lt = (1ull << lane) - 1ull;

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

  // A prep wave, one run ahead: the parameters of the run that starts at row i_next, the first r + r_prev clean entries of its shape's list
  // (walk against the dirty bitmap AS IT IS: the run in front of it, r_prev rows, is being committed and may take up to r_prev of them), and
  // the fetch of those entries' node state — lane j pulls every field of entry j, all loads in flight together; they land while the
  // predecessor is committed.  With at most i_prev dirty nodes and a list of W + 1 entries the walk finds r + r_prev clean entries unless
  // the list ends (its 0 terminator): nf < r + r_prev means every clean feasible node of the shape is among the nf.
#define K9S_PREP(i_next, r_prev, par)                                                                                  \
  do {                                                                                                                 \
    K9Prep &P_ = X.prep[(par)];                                                                                        \
    uint32_t nc_ = 0, s_ = 0, r_ = 0, fl_ = 0, km_ = 0, want_ = 0;                                                     \
    if ((i_next) < W) {                                                                                                \
      const uint4 ri_ = rinfo[(i_next)];                                                                               \
      r_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri_.x); s_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri_.y); \
      fl_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri_.z); km_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri_.w); \
      want_ = r_ + (r_prev);   /* <= 2 * K9_SEL_MAXRUN = 64: one entry per lane */                                     \
      uint32_t e_ = cursor[s_];                                                                                        \
      while (nc_ < want_) {                                                                                            \
        const uint32_t pos_ = e_ + lane;                                                                               \
        const uint32_t kk_ = (pos_ < Lp) ? lists[s_ * Lp + pos_] : 0u;                                                 \
        const bool nz_ = kk_ != 0u;                                                                                    \
        const uint32_t nn_ = nmaskbits - (kk_ & nmaskbits);                                                            \
        const bool cl_ = nz_ && !((bitmap[(nz_ ? nn_ : 0u) >> 5] >> (nn_ & 31)) & 1u);                                 \
        const unsigned long long zeros_ = __ballot(!nz_);                                                              \
        const uint32_t fz_ = zeros_ ? (uint32_t)__ffsll((unsigned long long)zeros_) - 1u : 64u;                        \
        const unsigned long long clean_ = __ballot(cl_ && lane < fz_);   /* entries behind the list's end do not count */ \
        const uint32_t rank_ = (uint32_t)__popcll(clean_ & lt);                                                        \
        if (((clean_ >> lane) & 1ull) && nc_ + rank_ < want_) { P_.ckey[nc_ + rank_] = kk_; P_.cpos[nc_ + rank_] = pos_; } \
        nc_ = min(want_, nc_ + (uint32_t)__popcll(clean_));                                                            \
        if (fz_ < 64u || e_ + 64u >= Lp) break;   /* the list ended */                                                 \
        e_ += 64u;                                                                                                     \
      }                                                                                                                \
      K9_WAVE_FENCE();                                                                                                 \
      if (lane < nc_) {                                                                                                \
        const uint32_t n_ = nmaskbits - (P_.ckey[lane] & nmaskbits);                                                   \
        _Pragma("unroll") for (int f = 0; f < 10; f++) raw[f] = g8[f][n_];                                             \
        raw[F_PORTS] = gports ? gports[n_] : 0ull;                                                                     \
        rcls = gcls[n_]; rmaxp = gmaxp[n_]; rpodc = gpodc[n_]; rnm = gnm[n_];                                          \
      }                                                                                                                \
      const bool plain_ = (fl_ & 1u) && (km_ == 0u || (fl_ & 4u));                                                     \
      if (!plain_ && lane == 63) {   /* an init container raised InitResreq above Resreq (rare): the row's own Resreq */ \
        const KbDev &d_ = *a.dev;                                                                                      \
        const uint32_t tk_ = desc[(i_next)].task;                                                                      \
        for (int dd = 0; dd < a.R; dd++) rowres[(par) * (uint32_t)a.R + (uint32_t)dd] = d_.t_res[(size_t)dd * d_.T + tk_]; \
      }                                                                                                                \
    }                                                                                                                  \
    if (lane == 0) { P_.i = (i_next); P_.s = s_; P_.r = r_; P_.fl = fl_; P_.km = km_; P_.nf = nc_; }                   \
    if (lane < (uint32_t)(sizeof(K9Shape) / 8) && (i_next) < W)                                                        \
      reinterpret_cast<unsigned long long *>(&P_.sh)[lane] = reinterpret_cast<const unsigned long long *>(&shapes[s_])[lane]; \
  } while (0)

  if (wave == K9S_PREP0) K9S_PREP(0u, 0u, 0u);

  // ---------------- run loop (uniform across the workgroup): two barriers per run ----------------
  for (uint32_t k = 0;; k++) {
    K9_STAMP(9);
    __syncthreads();   // B1: the run's header, its prepared entries, and every slot / bitmap bit the previous run wrote, are visible
    K9_STAMP(0);
    if (H.stop) break;
    const uint32_t par = k & 1u;
    const K9Prep &P = X.prep[par];
    const uint32_t i0 = H.i, nd = H.nd, s = P.s, r = P.r, fl0 = P.fl, km0 = P.km, nf = P.nf;
    const bool plain0 = (fl0 & 1u) && (km0 == 0u || (fl0 & 4u));
    const K9Shape sh = P.sh;   // == shapes[s]
    const double *si = sinit + (size_t)s * RS;
    const double *rres = rowres + par * (uint32_t)a.R;   // the row's own Resreq (rows that are not plain)
    const double *rqv = plain0 ? si : rres + 2;          // the rows' scalar Resreq
    // the run goes through the selection: plain rows of the allocate action, at least two of them
    const bool sel_run = !a.backfill && plain0 && r >= 2u;
#ifdef KB_K9_TRACE
    const unsigned long long tr0 = __builtin_readcyclecounter();
#define K9S_ROLE_END(k) do { if (lane == 0) atomicAdd(&X.tr[k], (uint32_t)(__builtin_readcyclecounter() - tr0)); } while (0)
#else
#define K9S_ROLE_END(k) do { } while (0)
#endif
    // ---- evaluation phase, one evaluation deep
    if (wave >= 1u && wave <= 4u) {
      if (tid - 64u < nd) {   // the shape against "their" dirty slot
        const uint32_t t = tid - 64u;
        const K9St vs = k9_load(slots + (size_t)t * K9_NF);
        const K9Sc scp = k9_sc_preload(sh.active >> 2, gi, gr, a.NP, vs.node);
        const uint32_t key0 = k9_eval_v(a, sh, vs, scp, gi, gr, si, 0u, 0.0, si, nb, nmaskbits);
        dk[t] = key0;
        // what the slot's next placement of the shape would be (allocate.go:160) — the selection reads it for the slots it considers
        if (sel_run) X.dkk[t] = k9_fits_idle(a, sh, vs.idle0, vs.idle1, scp, gi, si, vs.node, 0u, 0.0, si) ? 0u : 1u;
      }
      if (wave == 1u) K9S_ROLE_END(2);
    } else if (wave == K9S_PREP0 + par) {
      // ---- this run's candidates: of the entries fetched one iteration ago, the first r whose node the predecessor left alone; lane j holds
      //      entry j's node state.  P2: Allocate / Pipeline, NodeInfo.AddTask on the fetched state, key of the node after the placement
      const uint32_t key = (lane < nf) ? P.ckey[lane] : 0u;
      const uint32_t n = nmaskbits - (key & nmaskbits);
      const bool keep = lane < nf && !((bitmap[n >> 5] >> (n & 31)) & 1u);
      const unsigned long long kb = __ballot(keep);
      const uint32_t rho = (uint32_t)__popcll(kb & lt);
      const uint32_t ncand = min((uint32_t)__popcll(kb), r);
      if (keep && rho < r) {
        double res0 = sh.init0, res1 = sh.init1;
        if (!plain0) { res0 = rres[0]; res1 = rres[1]; }
        unsigned long long *st = slots + (size_t)(nd + rho) * K9_NF;
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
        const uint32_t k1 = k9_eval_v(a, sh, v, scn, gi, gr, si, adjm, 1.0, rqv, nb, nmaskbits);   // the node's key once it is dirty
        uint32_t kind1 = 0u;
        if (sel_run && !kind) kind1 = k9_fits_idle(a, sh, v.idle0, v.idle1, scn, gi, si, n, adjm, 1.0, rqv) ? 0u : 1u;   // a second placement on it
        ckey[rho] = key; cpos[rho] = P.cpos[lane];
        X.ckind[rho] = kind; X.ck1[rho] = k1; X.ckind1[rho] = kind1; X.crnm[rho] = rnm;
      }
      if (lane == 0) H.ncand = ncand;
      K9S_ROLE_END(0);
    } else if (wave == K9S_PREP0 + (par ^ 1u)) {
      // ---- the next run (it starts behind this one's last row if this one completes; if it does not, the round ends here)
      K9S_PREP(i0 + r, r, par ^ 1u);
      K9S_ROLE_END(1);
    }
    K9_STAMP(1);
    __syncthreads();   // B2: the dirty keys, the candidates and their slots are in LDS
    K9_STAMP(2);

    // ---- P3 (wave 0): the rows of the run; then the next run's header
    if (wave == 0) {
      const uint32_t ncand = H.ncand;
      uint32_t ck = 0, k1 = 0, ckind = 0, ckind1 = 0;
      rnm = 0;
      if (lane < ncand) { ck = ckey[lane]; k1 = X.ck1[lane]; ckind = X.ckind[lane]; ckind1 = X.ckind1[lane]; rnm = X.crnm[lane]; }
      double res0 = sh.init0, res1 = sh.init1;
      if (!plain0) { res0 = rres[0]; res1 = rres[1]; }
      // cmin: the r-th best clean candidate's key — r entries are at or above it, so no entry below it is among the picks (0: the list
      // holds fewer than r clean nodes)
      const uint32_t cmin = (sel_run && ncand == r) ? ckey[r - 1u] : 0u;
      // dirty keys of the shape: old slot t in lane t & 63, register t >> 6; the run's own new slots: k1 of lanes < pc
      uint32_t d0 = (lane < nd) ? dk[lane] : 0u, d1 = (lane + 64 < nd) ? dk[lane + 64] : 0u;
      uint32_t d2 = (lane + 128 < nd) ? dk[lane + 128] : 0u, d3 = (lane + 192 < nd) ? dk[lane + 192] : 0u;
      uint32_t m = wave_max_u32(max(max(d0, d1), max(d2, d3)));   // best dirty key; clean winners update it in O(1)
      uint32_t pc = 0, j = 0, reason = KB_REASON_DONE, n_dirty = 0, sc_dirty = 0;
      bool sel_done = false;
      if (sel_run) {
        // ---- every pick a clean candidate's first placement?  No dirty key above the r-th clean candidate, no clean candidate whose key
        //      after its placement is: row j takes candidate j (a Pipeline among them ends the round behind its row)
        const unsigned long long deeper = __ballot(lane + 1u < r && lane < ncand && ckind == 0u && k1 > cmin);
        if (ncand == r && m < cmin && !deeper) {
          const unsigned long long pipes = __ballot(lane < r && ckind != 0u);
          const uint32_t n_take = pipes ? (uint32_t)__ffsll((unsigned long long)pipes) : r;
          if (pipes) reason = KB_REASON_PIPELINED;
          if (lane < n_take) {
            const uint32_t n = nmaskbits - (ck & nmaskbits);
            ldec[i0 + lane] = (unsigned long long)n | ((unsigned long long)ckind << 32);
            atomicOr(&bitmap[n >> 5], 1u << (n & 31));
            if (km0) {   // the scalar dimensions Resreq names: Idle / Releasing in HBM
              const bool has_map = ckind ? (rnm >> 31) : (rnm & 0x7FFFFFFFu);
              if (has_map)
                for (uint32_t mm = km0, dd = 0; mm; mm >>= 1, dd++)
                  if (mm & 1u) k9_sc_sub(ckind ? gr : gi, a.NP, dd, n, rqv[dd]);
            }
          }
          if (km0) sc_dirty = 1;
          pc = n_take; j = n_take;
          sel_done = true;
          if (lane == 0) X.stat[0]++;
          K9_STAMP(8);
        } else {
          // ---- the general case.  Contenders: the clean candidates (lane = candidate) and the dirty slots whose key is above the floor;
          //      entries: steps 0 and 1 of each, as far as they exist and are above the floor
          const bool a0v = lane < ncand;
          const uint32_t ce1 = min(ck, k1);
          const bool a1v = a0v && ckind == 0u && k1 != 0u && ce1 > cmin;
          uint32_t dkk4[4], dd4[4] = {d0, d1, d2, d3};
          bool b0v[4];
          unsigned long long bb0[4];
          uint32_t nD = 0;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t t = lane + 64u * (uint32_t)u;
            dkk4[u] = (t < nd) ? X.dkk[t] : 0u;
            b0v[u] = dd4[u] > cmin;
            bb0[u] = __ballot(b0v[u]);
            nD += (uint32_t)__popcll(bb0[u]);
          }
          const unsigned long long ba1 = __ballot(a1v);
          const uint32_t nA1 = (uint32_t)__popcll(ba1);
          const uint32_t nC = ncand + nD;
          uint32_t n = ncand + nA1 + nD;
          bool bail = nC > 64u || n > 64u;
          if (!bail) {
            if (a0v) {
              X.e_comp[lane] = ((unsigned long long)ck << 8) | 255ull;
              X.e_info[lane] = lane | (ckind << 8);
              X.c_slot[lane] = nd + lane; X.c_next[lane] = a1v ? 2u : 1u; X.c_eff[lane] = a1v ? ce1 : ck;
              X.c_flag[lane] = 2u | ((!a1v || ckind1) ? 1u : 0u);   // ended: a Pipeline, no second placement, or one below the floor
              X.c_take[lane] = 0u;
            }
            uint32_t base = ncand;
            if (a1v) {
              const uint32_t pos = base + (uint32_t)__popcll(ba1 & lt);
              X.e_comp[pos] = ((unsigned long long)ce1 << 8) | 254ull;
              X.e_info[pos] = lane | (ckind1 << 8) | (1u << 16);
            }
            base += nA1;
#pragma unroll
            for (int u = 0; u < 4; u++) {   // the dirty contenders: step 0 (their key as the evaluation phase found it); the passes below walk them on
              if (b0v[u]) {
                const uint32_t pos = base + (uint32_t)__popcll(bb0[u] & lt), c = pos - nA1;
                X.e_comp[pos] = ((unsigned long long)dd4[u] << 8) | 255ull;
                X.e_info[pos] = c | ((dkk4[u] & 1u) << 8);
                X.c_slot[c] = lane + 64u * (uint32_t)u; X.c_next[c] = 1u; X.c_eff[c] = dd4[u];
                X.c_flag[c] = dkk4[u] & 1u;   // ended: its first placement is a Pipeline
                X.c_take[c] = 0u;
              }
              base += (uint32_t)__popcll(bb0[u]);
            }
            K9_WAVE_FENCE();
          }
          unsigned long long comp = 0ull;
          uint32_t info = 0u, rank = 0u;
          bool first = true;
          while (!bail) {
            // rank by count: entry e is picked as row #(entries in front of it)
            comp = lane < n ? X.e_comp[lane] : 0ull;
            info = lane < n ? X.e_info[lane] : 0u;
            rank = 0u;
            for (uint32_t i = 0; i < n; i++) { const unsigned long long si_ = rl64(comp, i); rank += (si_ > comp) ? 1u : 0u; }
            if (first) { K9_STAMP(5); first = false; }
            // a contender whose last known step would be picked in front of the last row may be picked again: walk it on
            const uint32_t c = info & 0xFFu, ej = info >> 16;
            const bool alive = lane < n && ej + 1u == X.c_next[c] && !(X.c_flag[c] & 1u) && rank + 1u < r;
            const unsigned long long ab = __ballot(alive);
            if (!ab) break;
            const uint32_t na = (uint32_t)__popcll(ab);
            uint32_t D = (64u - n) / na;
            if (D == 0u) { bail = true; break; }
            D = min(D, r - 1u);
            if (alive) X.al[(uint32_t)__popcll(ab & lt)] = c;
            K9_WAVE_FENCE();
            // lane -> (contender ai, step u of this pass): the contender's state after that many more placements, one subtraction at a time
            const bool act = lane < na * D;
            const uint32_t ai = lane / D, u = lane - ai * D;
            uint32_t cc = 0u, jj = 0u, kind = 0u, key = 0u, run = 0xFFFFFFFFu;
            bool inexact = false;
            if (act) {
              cc = X.al[ai];
              const uint32_t slot = X.c_slot[cc], b = (X.c_flag[cc] >> 1) & 1u;
              jj = X.c_next[cc] + u;
              run = X.c_eff[cc];
              const uint32_t mpl = jj - b;   // placements on top of the slot's state (a clean candidate's slot holds it after the first)
              const unsigned long long *st = slots + (size_t)slot * K9_NF;
              K9St v = k9_load(st);
              const uint32_t nm0 = (uint32_t)(st[F_NODE_NMASK] >> 32);
              const uint32_t adjm = (nm0 & 0x7FFFFFFFu) ? km0 : 0u;
              const K9Sc scx = k9_sc_preload(sh.active >> 2, gi, gr, a.NP, v.node);
              // scalar dimensions are evaluated as Idle - jj * Resreq: equal to jj subtractions when both are integers (checked)
              if (jj >= 2u)
                for (uint32_t mm = (sh.active >> 2) & adjm, dd = 0; mm; mm >>= 1, dd++)
                  if (mm & 1u) {
                    const double id = k9_sci(scx, gi, a.NP, dd, v.node), rq = si[dd];
                    if (!(id == trunc(id) && rq == trunc(rq) && fabs(id) < 4.0e15 && rq < 3.0e13)) inexact = true;
                  }
              for (uint32_t t = 0; t < mpl; t++) { v.idle0 -= sh.init0; v.idle1 -= sh.init1; v.nzc += sh.nzc; v.nzm += sh.nzm; }
              if (mpl) v.ports |= sh.want;
              v.left -= (int)mpl;
              key = k9_eval_v(a, sh, v, scx, gi, gr, si, adjm, (double)jj, si, nb, nmaskbits);
              kind = k9_fits_idle(a, sh, v.idle0, v.idle1, scx, gi, si, v.node, adjm, (double)jj, si) ? 0u : 1u;
            }
            if (__ballot(inexact)) { bail = true; break; }
            // step jj exists iff every step of the pass before it exists and is an Allocate, and its own key is not 0.  The lanes of a
            // contender are neighbours [g0, g0 + D): ballots for the existence, a shuffle per step for the prefix minimum
            const uint32_t g0 = ai * D;
            const unsigned long long ends = __ballot(act && (key == 0u || kind != 0u));   // nothing exists behind such a step
            const unsigned long long mine = lt & ~((1ull << (g0 & 63u)) - 1ull);          // my contender's lanes in front of me
            const bool valid = act && key != 0u && (ends & mine) == 0ull;
            for (uint32_t t = 0; t < D; t++) {
              const uint32_t kt = (uint32_t)__shfl((int)key, (int)(g0 + t));
              if (t <= u) run = min(run, kt);
            }
            const unsigned long long vb = __ballot(valid);
            if (valid) {
              const uint32_t pos = n + (uint32_t)__popcll(vb & lt);
              X.e_comp[pos] = ((unsigned long long)run << 8) | (unsigned long long)(255u - jj);
              X.e_info[pos] = cc | (kind << 8) | (jj << 16);
            }
            {   // the contender's record, by its first lane: g steps were found
              const uint32_t g = act ? (uint32_t)__popcll((vb >> (g0 & 63u)) & ((1ull << D) - 1ull)) : 0u;   // D <= K9_SEL_MAXRUN - 1
              const uint32_t lastl = g0 + (g ? g - 1u : 0u);
              const uint32_t run_last = (uint32_t)__shfl((int)run, (int)lastl), kind_last = (uint32_t)__shfl((int)kind, (int)lastl);
              if (act && u == 0u) {
                X.c_next[cc] += g;
                if (g) X.c_eff[cc] = run_last;
                if (g < D || kind_last != 0u) X.c_flag[cc] |= 1u;   // the sequence ended inside the pass, or its last step is a Pipeline
              }
            }
            n += (uint32_t)__popcll(vb);
            if (lane == 0) X.stat[3]++;
            K9_WAVE_FENCE();
          }
          K9_STAMP(6);
          if (!bail) {
            // ---- the picks: rows in rank order; a Pipeline ends the round behind its row; fewer entries than rows: no feasible node is left
            const bool have = lane < n;
            const uint32_t ekind = (info >> 8) & 1u, ec = info & 0xFFu;
            const uint32_t cnt = min(n, r);
            const uint32_t pr = (have && rank < r && ekind) ? rank : 0xFFFFFFFFu;
            const uint32_t minpipe = ~wave_max_u32(~pr);
            uint32_t n_take = cnt;
            if (minpipe < cnt) { n_take = minpipe + 1u; reason = KB_REASON_PIPELINED; }
            else if (cnt < r) reason = KB_REASON_NO_FEASIBLE;   // allocate.go:144-148
            if (have && rank < n_take) {
              const uint32_t node = (uint32_t)slots[(size_t)X.c_slot[ec] * K9_NF + F_NODE_NMASK];
              ldec[i0 + rank] = (unsigned long long)node | ((unsigned long long)ekind << 32);
              atomicAdd(&X.c_take[ec], 1u);
              if (ekind) atomicOr(&X.c_flag[ec], 4u);
            }
            K9_WAVE_FENCE();
            // ---- NodeInfo.AddTask (api/node_info.go:172-212), once per placement, on every contender that was picked: lane = contender
            const uint32_t T = (lane < nC) ? X.c_take[lane] : 0u;
            if (T) {
              const uint32_t fl = X.c_flag[lane], b = (fl >> 1) & 1u, pipe_last = (fl >> 2) & 1u;
              unsigned long long *st = slots + (size_t)X.c_slot[lane] * K9_NF;
              const uint32_t node = (uint32_t)st[F_NODE_NMASK], nm = (uint32_t)(st[F_NODE_NMASK] >> 32);
              const uint32_t extra = T - b;   // a clean candidate's slot already holds its first placement
              if (extra) {
                double idle0 = u2d(st[F_IDLE0]), idle1 = u2d(st[F_IDLE1]), rel0 = u2d(st[F_REL0]), rel1 = u2d(st[F_REL1]);
                double zc = u2d(st[F_NZC]), zm = u2d(st[F_NZM]);
                for (uint32_t t = 0; t < extra; t++) {
                  if (pipe_last && t + 1u == extra) { rel0 -= sh.init0; rel1 -= sh.init1; } else { idle0 -= sh.init0; idle1 -= sh.init1; }
                  zc += sh.nzc; zm += sh.nzm;
                }
                st[F_IDLE0] = d2u(idle0); st[F_IDLE1] = d2u(idle1); st[F_REL0] = d2u(rel0); st[F_REL1] = d2u(rel1);
                st[F_NZC] = d2u(zc); st[F_NZM] = d2u(zm);
                st[F_PORTS] |= sh.want;
                st[F_CLS_LEFT] -= ((unsigned long long)extra << 32);   // that many more pods on the node
              }
              if (km0)   // the scalar dimensions Resreq names, in HBM, one Sub per placement (Sub returns early when the receiver's map is nil)
                for (uint32_t t = 0; t < T; t++) {
                  const bool pp = pipe_last && t + 1u == T;
                  const bool has_map = pp ? (nm >> 31) : (nm & 0x7FFFFFFFu);
                  if (has_map)
                    for (uint32_t mm = km0, dd = 0; mm; mm >>= 1, dd++)
                      if (mm & 1u) k9_sc_sub(pp ? gr : gi, a.NP, dd, node, si[dd]);
                }
              if (b) atomicOr(&bitmap[node >> 5], 1u << (node & 31));
            }
            if (km0) sc_dirty = 1;
            pc = (uint32_t)__popcll(__ballot(lane < ncand && T != 0u));
            n_dirty = n_take - pc;
            j = n_take;
            sel_done = true;
            if (lane == 0) X.stat[1]++;
          } else if (lane == 0) {
            X.stat[2]++;
          }
          K9_STAMP(7);
        }
      }
      if (!sel_done)
      for (; j < r; j++) {   // ---- the serial loop of kb_commit.hip (single rows, backfill, rows with their own Resreq, whatever the selection handed back)
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
          const uint32_t own = (lane < pc) ? k1 : 0u;
          const unsigned long long who = __ballot(d0 == m || d1 == m || d2 == m || d3 == m || own == m);
          const uint32_t L = (uint32_t)__ffsll((unsigned long long)who) - 1u;
          const bool is_new = L < pc && rl32(k1, L) == m;
          const uint32_t wsel = (rl32(d0, L) == m) ? 0u : (rl32(d1, L) == m) ? 1u : (rl32(d2, L) == m) ? 2u : 3u;
          const uint32_t x = is_new ? nd + L : L + 64u * wsel;
          unsigned long long *st = slots + (size_t)x * K9_NF;
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
      if (sc_dirty) __threadfence();   // the scalar atomics have reached L2 before any other wave evaluates against these nodes
      // the next run's header (its candidates were walked while this one was evaluated)
      const uint32_t i_next = i0 + j;
      uint32_t stop = (reason != KB_REASON_DONE || i_next >= W) ? 1u : 0u;
      if (!stop && a.has_aff && !a.backfill && i_next != 0u) {
        // a row whose score is normalised over its feasible set (preferred node affinity) is exact only against a fresh matrix: it may be
        // the first row of a round, nothing else
        const uint32_t fln = (uint32_t)__builtin_amdgcn_readfirstlane((int)rinfo[i_next].z);
        if (fln & 2u) { reason = KB_REASON_RENORM; stop = 1u; }
      }
      if (lane == 0) {
        if (pc) cursor[s] = cpos[pc - 1] + 1;
        H.n_dirty_rows += n_dirty; H.n_runs += 1; H.n_slow += plain0 ? 0u : 1u;
        H.i = i_next; H.nd = nd + pc; H.reason = reason; H.stop = stop;
      }
#ifdef KB_K9_TRACE
      if (r == 1u && lane == 0) X.tr[3] += (uint32_t)(__builtin_readcyclecounter() - tlast);
#endif
      K9_STAMP(3);
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
    tw[4] = (unsigned long long)X.tr[0] | ((unsigned long long)X.tr[1] << 32);
    tw[15] = (unsigned long long)X.tr[2] | ((unsigned long long)X.tr[3] << 32);
  }
#endif
  __syncthreads();   // the statistics words (wave 0) before the epilogue reads them
  k9_epilogue(a, lo, k9_smem, tid, t_start, X.stat[0] | (X.stat[1] << 16), X.stat[2] | (X.stat[3] << 16));
}

void kb_launch_commit_sel(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_commit_select), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const size_t sh = k9_layout(r.n_rows, r.n_mrows, r.L, d.NP, d.R, true).total;
  K9KernArgs ka;
  k9_fill_args(ka, d, r);
  static const bool helpers_off = getenv("KB_WARM_HELPERS_OFF") && getenv("KB_WARM_HELPERS_OFF")[0] == '1';   // A/B switch
  hipLaunchKernelGGL(k_commit_select, dim3(helpers_off ? 1u : KB_WARM_GRID), dim3(K9_THREADS), sh, (hipStream_t)stream, ka);
}
