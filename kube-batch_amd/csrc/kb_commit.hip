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
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "kb_device.h"
#include "kb_eval.hpp"
#include "kb_warm.hpp"

#define K9_THREADS 512   // wave 0: the sequential part; waves 1..4: one dirty slot per thread; all: prologue / epilogue
#define K9_MAXRUN 64
#define K9_MAXSLOTS 256 // dirty slots (= rows) per round
#define K9_NF 13   // 8-byte fields per dirty slot
enum { F_IDLE0 = 0, F_IDLE1, F_REL0, F_REL1, F_INVAC, F_INVAM, F_AC, F_AM, F_NZC, F_NZM, F_PORTS, F_CLS_LEFT, F_NODE_NMASK };

struct K9Shape {   // what an evaluation needs to know about a task shape (64 bytes)
  double init0, init1, nzc, nzm;           // InitResreq cpu / memory, pod non-zero request (as doubles: exact below 2^53)
  unsigned long long conf, want;           // host-port conflict mask, host ports the pod occupies
  uint32_t cls, active, crow, pad;
};

struct K9Hdr {
  uint32_t i, nd, reason, stop;            // next row, dirty slots, KB_REASON_*, 1 = leave the run loop
  uint32_t ncand, n_dirty_rows, n_runs, n_slow;
  uint32_t cur_s, cur_r, cur_fl, cur_km;   // the run being processed: shape, rows, flags and Resreq key mask of its rows
};

// dynamic LDS layout for a round of n_rows rows and n_shapes distinct shapes
struct K9Layout {
  uint32_t slots, rowres, sinit, shapes, desc, rinfo, dec, hdr, dk, ckey, cpos, cursor, shp, lists, bitmap, total;   // byte offsets
  uint32_t sel;   // the selection kernel's scratch (K9Sel), 0 in the run kernel's layout
  uint32_t Lp, RS;
};
// Scratch of the run selection (k_commit_run<true>): what the evaluation phase leaves about every dirty slot one placement deep, and wave 0's
// entry / contender tables.
struct K9Sel {
  uint32_t dk1[K9_MAXSLOTS];          // key of dirty slot t after ONE more placement of the run's shape (0: infeasible; only where dk[t] is above the floor)
  uint32_t dkk[K9_MAXSLOTS];          // bit 0: that placement would be a Pipeline; bit 1: the one after it would be
  unsigned long long e_comp[64];      // entries: prefix-minimum key << 8 | 255 - step  (greater = picked earlier)
  uint32_t e_info[64];                // contender | kind << 8 | step << 16
  uint32_t c_slot[64], c_next[64], c_eff[64], c_flag[64], c_take[64];   // contenders: state slot, next unknown step, prefix minimum so far, bit 0 ended / bit 1 clean / bit 2 its last taken entry is a Pipeline
  uint32_t al[64], kt[64], kk[64];    // a deep pass: the contenders it walks, the keys and kinds its lanes found
  uint32_t stat[4];                   // runs committed with every pick a clean first placement / by the general selection / handed to the serial loop; deep passes
};
__host__ __device__ inline K9Layout k9_layout(uint32_t n_rows, uint32_t n_shapes, uint32_t L, uint32_t NP, int R, bool sel = false) {
  K9Layout o;
  o.RS = R > 2 ? (uint32_t)(R - 2) : 0u;
  o.Lp = L;
  uint32_t off = 0;
  o.slots = off;  off += (n_rows + K9_MAXRUN) * K9_NF * 8u;   // + a run's worth: P2 writes every candidate's post-placement state
  o.rowres = off; off += (uint32_t)R * 8u;
  o.sinit = off;  off += n_shapes * o.RS * 8u;
  o.shapes = off; off += n_shapes * (uint32_t)sizeof(K9Shape);
  o.desc = off;   off += n_rows * (uint32_t)sizeof(KbRowDesc);
  off = (off + 15u) & ~15u;
  o.rinfo = off;  off += n_rows * 16u;
  o.dec = off;    off += n_rows * 8u;
  o.hdr = off;    off += (uint32_t)sizeof(K9Hdr);
  o.dk = off;     off += K9_MAXSLOTS * 4u;
  o.ckey = off;   off += K9_MAXRUN * 4u;
  o.cpos = off;   off += K9_MAXRUN * 4u;
  o.cursor = off; off += n_shapes * 4u;
  o.shp = off;    off += n_shapes * 4u;
  o.lists = off;  off += n_shapes * o.Lp * 4u;
  o.bitmap = off; off += (NP / 32u) * 4u;
  off = (off + 15u) & ~15u;
  o.sel = 0;
  if (sel) { o.sel = off; off += (uint32_t)sizeof(K9Sel); }
  o.total = (off + 15u) & ~15u;
  return o;
}
size_t kb_commit_smem_bytes(uint32_t n_rows, uint32_t n_shapes, uint32_t NP, int R) {   // the window is planned for either of the two (the selection kernel's scratch included)
  return k9_layout(n_rows, n_shapes, n_rows + 1, NP, R, true).total;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define K9_UMAX(ctrl, row_mask) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, (ctrl), (row_mask), 0xf, false))
  K9_UMAX(0xB1, 0xf);    // quad_perm [1,0,3,2]
  K9_UMAX(0x4E, 0xf);    // quad_perm [2,3,0,1]
  K9_UMAX(0x141, 0xf);   // row_half_mirror
  K9_UMAX(0x140, 0xf);   // row_mirror
  K9_UMAX(0x142, 0xa);   // row_bcast:15
  K9_UMAX(0x143, 0xc);   // row_bcast:31 -> lane 63 holds the wave maximum
#undef K9_UMAX
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ unsigned long long rl64(unsigned long long v, uint32_t l) {
  return ((unsigned long long)rl32((uint32_t)(v >> 32), l) << 32) | rl32((uint32_t)v, l);
}
__device__ __forceinline__ double u2d(unsigned long long v) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ unsigned long long d2u(double v) { return (unsigned long long)__double_as_longlong(v); }
// make EXTRA=-DKB_K9_TRACE: cycles wave 0 spends in each phase of a run, summed per round into words 5..7 and 13..14 of the
// output block, printed by the host under KB_K5_STATS=1
#ifdef KB_K9_TRACE
#define K9_STAMP(k) do { if (wave == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += (uint32_t)(now_ - tlast); tlast = now_; } } while (0)
// the same interval also into slot k2 when `cond` holds (runs that touch scalar dimensions, accounted apart)
#define K9_STAMP2(k, k2, cond) do { if (wave == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += (uint32_t)(now_ - tlast); if (cond) tacc[k2] += (uint32_t)(now_ - tlast); tlast = now_; } } while (0)
#define K9_COUNT(k, v) do { tacc[k] += (v); } while (0)
#else
#define K9_STAMP(k) do { } while (0)
#define K9_STAMP2(k, k2, cond) do { } while (0)
#define K9_COUNT(k, v) do { } while (0)
#endif
#define K9_WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

struct K9KernArgs {
  KbCommitArgs hot;
  KbDev dev;
  KbRound round;
};

// a node's state as the evaluation reads it
struct K9St {
  double idle0, idle1, rel0, rel1, inv_ac, inv_am, ac, am, nzc, nzm;
  unsigned long long ports;
  uint32_t cls, node;
  int left;   // Allocatable.MaxTaskNum - len(pods): predicates.go:127 fails on <= 0
};
__device__ __forceinline__ K9St k9_load(const unsigned long long *st) {
  K9St v;
  v.idle0 = u2d(st[F_IDLE0]); v.idle1 = u2d(st[F_IDLE1]); v.rel0 = u2d(st[F_REL0]); v.rel1 = u2d(st[F_REL1]);
  v.inv_ac = u2d(st[F_INVAC]); v.inv_am = u2d(st[F_INVAM]); v.ac = u2d(st[F_AC]); v.am = u2d(st[F_AM]);
  v.nzc = u2d(st[F_NZC]); v.nzm = u2d(st[F_NZM]); v.ports = st[F_PORTS];
  v.cls = (uint32_t)st[F_CLS_LEFT]; v.left = (int)(uint32_t)(st[F_CLS_LEFT] >> 32); v.node = (uint32_t)st[F_NODE_NMASK];
  return v;
}
typedef double __attribute__((address_space(1))) *gptrd;
// Scalar resource dimensions stay in HBM (Idle / Releasing [R][NP]): only shapes that name a scalar read them, only rows whose
// Resreq names one change them (one float64 atomic add per dimension at L2, exact: a single IEEE addition), and the reads go to
// L2 as well (agent scope), so what one wave of the workgroup changed is what the others see after the barrier.
__device__ __forceinline__ double k9_sc(gptrd base, uint32_t NP, uint32_t dd, uint32_t node) {
  return __hip_atomic_load(base + (size_t)(dd + 2) * NP + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void k9_sc_sub(gptrd base, uint32_t NP, uint32_t dd, uint32_t node, double v) {
  (void)__hip_atomic_fetch_add(base + (size_t)(dd + 2) * NP + node, -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The scalar dimensions one evaluation reads: Idle and Releasing of up to two of them are loaded up front, the four loads in flight together.
// Read where they are used (a dependent L2 round trip per value, one after the other behind the `&&` of the fit test) they made an
// evaluation of a shape with scalar requests cost three times a plain one (profiles/round3/call18: config 4, runs with scalar dimensions
// are 29 % of the runs and 44 % of the rows phase).  Dimensions beyond two go to L2 where they are used, as before.
#define K9_NOD 0xFFFFFFFFu
struct K9Sc {
  uint32_t dA, dB;         // dimension indices (0 = the first one after cpu / memory); K9_NOD: none
  double iA, rA, iB, rB;   // Idle / Releasing of the node in dA / dB
};
__device__ __forceinline__ K9Sc k9_sc_preload(uint32_t mask, gptrd gi, gptrd gr, uint32_t NP, uint32_t node) {
  K9Sc c;
  c.dA = K9_NOD; c.dB = K9_NOD; c.iA = 0.0; c.rA = 0.0; c.iB = 0.0; c.rB = 0.0;
  if (mask) {
    const uint32_t m2 = mask & (mask - 1u);
    c.dA = (uint32_t)__ffs((int)mask) - 1u;
    if (m2) c.dB = (uint32_t)__ffs((int)m2) - 1u;
    const uint32_t b = m2 ? c.dB : c.dA;   // one dimension only: the second pair repeats the first (no branch around loads)
    c.iA = k9_sc(gi, NP, c.dA, node); c.rA = k9_sc(gr, NP, c.dA, node);
    c.iB = k9_sc(gi, NP, b, node); c.rB = k9_sc(gr, NP, b, node);
  }
  return c;
}
__device__ __forceinline__ double k9_sci(const K9Sc &c, gptrd gi, uint32_t NP, uint32_t dd, uint32_t node) {
  return dd == c.dA ? c.iA : (dd == c.dB ? c.iB : k9_sc(gi, NP, dd, node));
}
__device__ __forceinline__ double k9_scr(const K9Sc &c, gptrd gr, uint32_t NP, uint32_t dd, uint32_t node) {
  return dd == c.dA ? c.rA : (dd == c.dB ? c.rB : k9_sc(gr, NP, dd, node));
}
// key of shape sh against node state v (a dirty slot, or a candidate after its placement); si: the shape's scalar InitResreq;
// sc: the node's scalar dimensions as k9_sc_preload(sh.active >> 2, ...) returned them (nothing may have lowered them in between).
// adj_mask / adj_mul / rq: evaluate as if Idle of the scalar dimensions in adj_mask were lower by adj_mul * rq[d] — placements
// whose scalar part has not reached HBM yet (the caller has already lowered cpu / memory in v); 0 for a plain evaluation.
__device__ __forceinline__ uint32_t k9_eval_v(const KbCommitArgs &a, const K9Shape &sh, const K9St &v, const K9Sc &sc, gptrd gi, gptrd gr, const double *si,
                                              uint32_t adj_mask, double adj_mul, const double *rq, uint32_t nb, uint32_t nmaskbits) {
  bool ok = true;
  if (a.fit_mode) {   // allocate.go:81: !InitResreq.LessEqual(Idle) && !InitResreq.LessEqual(Releasing) -> fail
    bool fi = le_eps(sh.init0, v.idle0, EPS_CPU) && le_eps(sh.init1, v.idle1, EPS_MEM);
    bool fr = le_eps(sh.init0, v.rel0, EPS_CPU) && le_eps(sh.init1, v.rel1, EPS_MEM);
    uint32_t aa = sh.active >> 2, dd = 0;
    while (aa) {   // scalar dimensions with InitResreq > 10 (resource_info.go:286-299)
      if (aa & 1u) {
        const double l = si[dd];
        double id = k9_sci(sc, gi, a.NP, dd, v.node);
        if ((adj_mask >> dd) & 1u) id -= adj_mul * rq[dd];
        fi = fi && le_eps(l, id, EPS_SCALAR);
        fr = fr && le_eps(l, k9_scr(sc, gr, a.NP, dd, v.node), EPS_SCALAR);
      }
      aa >>= 1; dd++;
    }
    ok = fi || (a.fit_mode != 2 && fr);   // 2: backfill, AddTask's Resreq.LessEqual(Idle) only (node_info.go:161-167)
  }
  if (a.pred_enabled) {
    ok = ok && (v.left > 0) && ((v.ports & sh.conf) == 0ull);   // pod count (predicates.go:127), PodFitsHostPorts (predicates.go:181-190)
    if (a.use_crow) {
      ok = ok && ((sh.crow >> (v.cls & 31)) & 1u);
    } else {
      const KbDev &d = *a.dev;
      if (d.compat) {
        const uint32_t bit = sh.cls * d.n_nc + v.cls;
        ok = ok && ((d.compat[bit >> 3] >> (bit & 7)) & 1);
      }
    }
  }
  if (!ok) return 0u;
  uint32_t score = 0;
  if (a.score_enabled) score = score_core_f64(sh.nzc, sh.nzm, v.nzc, v.nzm, v.ac, v.am, v.inv_ac, v.inv_am, a.wL, a.wM, a.wB);
  return ((score + 1u) << nb) | (nmaskbits - v.node);
}

// allocate.go:160: InitResreq.LessEqual(node.Idle) -> Allocate, else Pipeline; idle0 / idle1: the node's Idle cpu / memory as the caller has it,
// the scalar dimensions as k9_eval_v reads them (adj_*: placements whose scalar part has not reached HBM)
__device__ __forceinline__ bool k9_fits_idle(const KbCommitArgs &a, const K9Shape &sh, double idle0, double idle1, const K9Sc &sc, gptrd gi, const double *si, uint32_t node,
                                             uint32_t adj_mask, double adj_mul, const double *rq) {
  bool fi = le_eps(sh.init0, idle0, EPS_CPU) && le_eps(sh.init1, idle1, EPS_MEM);
  for (uint32_t aa = sh.active >> 2, dd = 0; aa; aa >>= 1, dd++)
    if (aa & 1u) {
      double id = k9_sci(sc, gi, a.NP, dd, node);
      if ((adj_mask >> dd) & 1u) id -= adj_mul * rq[dd];
      fi = fi && le_eps(si[dd], id, EPS_SCALAR);
    }
  return fi;
}

// SEL (KB_COMMIT_SELECT, DESIGN.md section 4 "run selection"): the rows phase of a run of r >= 2 plain rows is ONE selection instead of a loop
// over the rows.  Every candidate node — the run's r best clean entries and the dirty slots whose key is above the r-th of them — has its own
// key sequence key(n, j) (its key for the shape after j placements of the shape, while the placement still fits), which depends on that
// node's state only; with eff(n, j) = min key(n, 0..j), the serial loop's picks are the first r of all (n, j) entries in (eff descending,
// j ascending) order (keys carry the node, so entries of different nodes never tie): a node picked at key s was the maximum; while its next
// keys stay >= s it wins again at once, and when its key falls below s the prefix minimum is the real key again.  A Pipeline entry ends
// its node's sequence (picked, it ends the round).  Steps 0 and 1 of every candidate come out of the parallel evaluation phase; deeper
// steps are walked by the lanes of wave 0, several steps per candidate in one pass; the order is a rank by count.  Nothing is written
// before the picks are known, so any limit of the tables (64 entries, 64 candidates) simply hands the run to the serial loop.
template <bool SEL>
__global__ void __launch_bounds__(K9_THREADS) k_commit_run(const K9KernArgs ka) {
  KbCommitArgs a = ka.hot;
  {   // only `a` is named in the loops (SGPRs); the two views are read through the kernel-argument segment on rare paths
    const unsigned char __attribute__((address_space(4))) *kp = (const unsigned char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    a.dev = (const KbDev *)(kp + offsetof(K9KernArgs, dev));
    a.round = (const KbRound *)(kp + offsetof(K9KernArgs, round));
  }
  if (a.round->chain_expect != 0u && *a.round->chain != a.round->chain_expect) {   // chained to a round that stopped early: skip
    if (threadIdx.x == 0 && blockIdx.x == 0) {
      *a.round->chain = 0u;
      a.result[0] = 0; a.result[1] = KB_REASON_SKIPPED;
      if (a.host_out) {
        a.host_out[0] = (unsigned long long)KB_REASON_SKIPPED << 32;
        __threadfence_system();
        __hip_atomic_store(&a.host_out[KB_OUT_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (blockIdx.x != 0) {   // helper workgroups (kb_warm.hpp): warm a slice of the node state into the XCD's L2 and leave
    if ((blockIdx.x & 7u) != 0u) return;
    const uint32_t h = blockIdx.x / 8u - 1u, lines = a.NP / 16;
    const uint32_t l0 = (uint32_t)(((unsigned long long)h * lines) / KB_WARM_HELPERS), l1 = (uint32_t)(((unsigned long long)(h + 1) * lines) / KB_WARM_HELPERS);
    const unsigned long long acc = kb_warm_lines(*a.dev, a.keys, 0, l0, l1, threadIdx.x, K9_THREADS);   // the lists are copied to LDS by the prologue itself
    if (acc == 0x123456789abcdefull) a.result[15] = 1;   // keep the loads alive (never true)
    return;
  }
  extern __shared__ __align__(16) unsigned char k9_smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t S = a.n_mrows, W = a.n_rows;
  const K9Layout lo = k9_layout(W, S, a.L, a.NP, a.R, SEL);
  K9Sel &X = *reinterpret_cast<K9Sel *>(k9_smem + lo.sel);   // SEL only
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(k9_smem + lo.slots);
  double *rowres = reinterpret_cast<double *>(k9_smem + lo.rowres);
  double *sinit = reinterpret_cast<double *>(k9_smem + lo.sinit);
  K9Shape *shapes = reinterpret_cast<K9Shape *>(k9_smem + lo.shapes);
  KbRowDesc *desc = reinterpret_cast<KbRowDesc *>(k9_smem + lo.desc);
  uint4 *rinfo = reinterpret_cast<uint4 *>(k9_smem + lo.rinfo);
  unsigned long long *ldec = reinterpret_cast<unsigned long long *>(k9_smem + lo.dec);
  K9Hdr &H = *reinterpret_cast<K9Hdr *>(k9_smem + lo.hdr);
  uint32_t *dk = reinterpret_cast<uint32_t *>(k9_smem + lo.dk);
  uint32_t *ckey = reinterpret_cast<uint32_t *>(k9_smem + lo.ckey);
  uint32_t *cpos = reinterpret_cast<uint32_t *>(k9_smem + lo.cpos);
  uint32_t *cursor = reinterpret_cast<uint32_t *>(k9_smem + lo.cursor);
  uint32_t *shp = reinterpret_cast<uint32_t *>(k9_smem + lo.shp);
  uint32_t *lists = reinterpret_cast<uint32_t *>(k9_smem + lo.lists);
  uint32_t *bitmap = reinterpret_cast<uint32_t *>(k9_smem + lo.bitmap);
  const uint32_t RS = lo.RS, Lp = lo.Lp;
  const uint32_t nb = a.node_bits, nmaskbits = (1u << nb) - 1u;
  const unsigned long long t_start = wall_clock64();

  // ---------------- prologue (all threads) ----------------
  for (uint32_t w = tid; w < a.NP / 32; w += K9_THREADS) bitmap[w] = 0;
  {   // candidate lists: 64-bit keys of K3 -> compact 32-bit keys
    const uint32_t tot = S * Lp;
    for (uint32_t idx = tid; idx < tot; idx += K9_THREADS) {
      const unsigned long long k64 = a.keys[idx];   // [S][L], Lp == L
      lists[idx] = k64 ? (((KB_KEY_SCORE(k64) + 1u) << nb) | (nmaskbits - KB_KEY_NODE(k64))) : 0u;
    }
  }
  {   // row descriptors
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.desc);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(desc);
    for (uint32_t w = tid; w < W * (uint32_t)(sizeof(KbRowDesc) / 8); w += K9_THREADS) dst[w] = src[w];
  }
  for (uint32_t s = tid; s < S; s += K9_THREADS) cursor[s] = 0;
  if constexpr (SEL) { if (tid < 4) X.stat[tid] = 0u; }
  if (tid == 0) { H.i = 0; H.nd = 0; H.reason = KB_REASON_DONE; H.stop = 0; H.ncand = 0; H.n_dirty_rows = 0; H.n_runs = 0; H.n_slow = 0; H.cur_s = 0; H.cur_r = 0; H.cur_fl = 0; H.cur_km = 0; }
  if (gridDim.x == 1) {   // no helper workgroups (KB_WARM_HELPERS_OFF=1): warm the XCD's L2 here, all threads
    const unsigned long long acc = kb_warm_lines(*a.dev, a.keys, 0, 0, a.NP / 16, tid, K9_THREADS);
    if (acc == 0x123456789abcdefull) H.n_slow = 0xFFFFFFFFu;   // keep the loads alive
  }
  __syncthreads();
  // shape -> one of its rows (any: rows of a shape agree on everything the evaluation reads)
  for (uint32_t i = tid; i < W; i += K9_THREADS) shp[desc[i].slot] = i;
  __syncthreads();
  for (uint32_t s = tid; s < S; s += K9_THREADS) {
    const KbRowDesc &k = desc[shp[s]];
    K9Shape sh;
    sh.init0 = k.init0; sh.init1 = k.init1; sh.nzc = (double)k.nzc; sh.nzm = (double)k.nzm;
    sh.conf = a.has_ports ? a.dev->t_conf[k.task] : 0ull;
    sh.want = a.has_ports ? a.dev->t_want[k.task] : 0ull;
    sh.cls = k.cls; sh.active = k.active; sh.crow = k.crow; sh.pad = 0;
    shapes[s] = sh;
  }
  if (RS) {
    const KbDev &d = *a.dev;
    for (uint32_t idx = tid; idx < S * RS; idx += K9_THREADS) {
      const uint32_t s = idx / RS, dd = idx % RS;
      sinit[idx] = d.t_init[(size_t)(dd + 2) * d.T + desc[shp[s]].task];
    }
  }
  __syncthreads();

  // run table: rows i .. i + r - 1 share a shape and take the shape's own request values (a row whose Resreq differs from its
  // InitResreq, or whose score needs renormalising, is a run of its own)
  for (uint32_t i = tid; i < W; i += K9_THREADS) {
    const KbRowDesc &k = desc[i];
    const uint32_t sl = k.slot, fl = k.flags, km = k.resmask;
    const bool plain = (fl & 1u) && (km == 0u || (fl & 4u));
    uint32_t r = 1;
    if (plain && !(fl & 2u))
      while (r < (SEL ? 32u : K9_MAXRUN) && i + r < W && desc[i + r].slot == sl && desc[i + r].flags == fl && desc[i + r].resmask == km) r++;
    rinfo[i] = make_uint4(r, sl, fl, km);
  }
  __syncthreads();

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
    // SEL: the run goes through the selection (plain rows of the allocate action, at least two of them); cmin: the r-th best clean candidate's
    // key — r entries are at or above it, so no entry below it is among the picks (0: the list holds fewer than r clean nodes)
    const bool sel_run = SEL && !a.backfill && plain0 && r >= 2u;
    const uint32_t cmin = (sel_run && ncand == r) ? ckey[r - 1u] : 0u;
    // ---- evaluation phase, one evaluation deep: waves 1..4 the shape against "their" dirty slot, wave 0 the candidates
    if (wave >= 1 && tid - 64u < nd) {
      const uint32_t t = tid - 64u;
      const K9St vs = k9_load(slots + (size_t)t * K9_NF);
      const K9Sc scp = k9_sc_preload(sh.active >> 2, gi, gr, a.NP, vs.node);
      const uint32_t key0 = k9_eval_v(a, sh, vs, scp, gi, gr, si, 0u, 0.0, si, nb, nmaskbits);
      dk[t] = key0;
      if constexpr (SEL) {
        // a slot that can be picked (its key is above the floor): what its first placement would be, and its key and kind one placement on
        uint32_t key1 = 0u, kk = 0u;
        if (sel_run && key0 > cmin) {
          const uint32_t nm0 = (uint32_t)(slots[(size_t)t * K9_NF + F_NODE_NMASK] >> 32);
          const uint32_t adjm = (nm0 & 0x7FFFFFFFu) ? km0 : 0u;   // Idle has a scalar map: Sub lowers the dimensions Resreq names
          if (k9_fits_idle(a, sh, vs.idle0, vs.idle1, scp, gi, si, vs.node, 0u, 0.0, si)) {
            K9St v1 = vs;
            v1.idle0 -= sh.init0; v1.idle1 -= sh.init1; v1.nzc += sh.nzc; v1.nzm += sh.nzm;
            v1.ports |= sh.want; v1.left -= 1;
            key1 = k9_eval_v(a, sh, v1, scp, gi, gr, si, adjm, 1.0, si, nb, nmaskbits);
            if (!k9_fits_idle(a, sh, v1.idle0, v1.idle1, scp, gi, si, vs.node, adjm, 1.0, si)) kk |= 2u;
          } else {
            kk = 1u;   // the first placement would be a Pipeline: it ends the slot's sequence
          }
        }
        X.dk1[t] = key1; X.dkk[t] = kk;
      }
    }
    uint32_t ck = 0, k1 = 0, ckind = 0, ckind1 = 0;
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
        if (SEL && sel_run && !kind) ckind1 = k9_fits_idle(a, sh, v.idle0, v.idle1, scn, gi, si, n, adjm, 1.0, rqv) ? 0u : 1u;   // a second placement on it: Allocate / Pipeline
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
      bool sel_done = false;
      if constexpr (SEL) if (sel_run) {
        const unsigned long long lt = (1ull << lane) - 1ull;
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
        } else {
          // ---- the general case.  Contenders: the clean candidates (lane = candidate) and the dirty slots whose key is above the floor;
          //      entries: steps 0 and 1 of each, as far as they exist and are above the floor
          const bool a0v = lane < ncand;
          const uint32_t ce1 = min(ck, k1);
          const bool a1v = a0v && ckind == 0u && k1 != 0u && ce1 > cmin;
          uint32_t dq[4], dkk4[4], dd4[4] = {d0, d1, d2, d3}, de1[4], myc[4] = {0, 0, 0, 0};
          bool b0v[4], b1v[4];
          unsigned long long bb0[4], bb1[4];
          uint32_t nD = 0, nB1 = 0;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t t = lane + 64u * (uint32_t)u;
            dq[u] = (t < nd) ? X.dk1[t] : 0u;
            dkk4[u] = (t < nd) ? X.dkk[t] : 0u;
            b0v[u] = dd4[u] > cmin;
            de1[u] = min(dd4[u], dq[u]);
            b1v[u] = b0v[u] && !(dkk4[u] & 1u) && dq[u] != 0u && de1[u] > cmin;
            bb0[u] = __ballot(b0v[u]);
            bb1[u] = __ballot(b1v[u]);
            nD += (uint32_t)__popcll(bb0[u]);
            nB1 += (uint32_t)__popcll(bb1[u]);
          }
          const unsigned long long ba1 = __ballot(a1v);
          const uint32_t nA1 = (uint32_t)__popcll(ba1);
          const uint32_t nC = ncand + nD;
          uint32_t n = ncand + nA1 + nD + nB1;
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
            uint32_t cbase = ncand;
#pragma unroll
            for (int u = 0; u < 4; u++) {
              if (b0v[u]) {
                const uint32_t c = cbase + (uint32_t)__popcll(bb0[u] & lt);
                const uint32_t pos = base + (c - ncand);
                myc[u] = c;
                X.e_comp[pos] = ((unsigned long long)dd4[u] << 8) | 255ull;
                X.e_info[pos] = c | ((dkk4[u] & 1u) << 8);
                X.c_slot[c] = lane + 64u * (uint32_t)u; X.c_next[c] = b1v[u] ? 2u : 1u; X.c_eff[c] = b1v[u] ? de1[u] : dd4[u];
                X.c_flag[c] = (!b1v[u] || (dkk4[u] & 2u)) ? 1u : 0u;
                X.c_take[c] = 0u;
              }
              cbase += (uint32_t)__popcll(bb0[u]);
            }
            base += nD;
#pragma unroll
            for (int u = 0; u < 4; u++) {
              if (b1v[u]) {
                const uint32_t pos = base + (uint32_t)__popcll(bb1[u] & lt);
                X.e_comp[pos] = ((unsigned long long)de1[u] << 8) | 254ull;
                X.e_info[pos] = myc[u] | (((dkk4[u] >> 1) & 1u) << 8) | (1u << 16);
              }
              base += (uint32_t)__popcll(bb1[u]);
            }
            K9_WAVE_FENCE();
          }
          unsigned long long comp = 0ull;
          uint32_t info = 0u, rank = 0u;
          while (!bail) {
            // rank by count: entry e is picked as row #(entries in front of it)
            comp = lane < n ? X.e_comp[lane] : 0ull;
            info = lane < n ? X.e_info[lane] : 0u;
            rank = 0u;
            for (uint32_t i = 0; i < n; i++) { const unsigned long long si_ = rl64(comp, i); rank += (si_ > comp) ? 1u : 0u; }
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
            uint32_t cc = 0u, jj = 0u, kind = 0u;
            bool inexact = false;
            if (act) {
              cc = X.al[ai];
              const uint32_t slot = X.c_slot[cc], b = (X.c_flag[cc] >> 1) & 1u;
              jj = X.c_next[cc] + u;
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
              const uint32_t key = k9_eval_v(a, sh, v, scx, gi, gr, si, adjm, (double)jj, si, nb, nmaskbits);
              kind = k9_fits_idle(a, sh, v.idle0, v.idle1, scx, gi, si, v.node, adjm, (double)jj, si) ? 0u : 1u;
              X.kt[lane] = key; X.kk[lane] = kind;
            }
            if (__ballot(inexact)) { bail = true; break; }
            K9_WAVE_FENCE();
            // step jj exists iff every step of the pass before it exists and is an Allocate, and its own key is not 0
            bool valid = act;
            uint32_t run = 0u;
            if (act) {
              run = X.c_eff[cc];
              for (uint32_t t = 0; t <= u; t++) {
                const uint32_t kt = X.kt[ai * D + t];
                valid = valid && kt != 0u && (t == u || X.kk[ai * D + t] == 0u);
                run = min(run, kt);
              }
            }
            const unsigned long long vb = __ballot(valid);
            if (valid) {
              const uint32_t pos = n + (uint32_t)__popcll(vb & lt);
              X.e_comp[pos] = ((unsigned long long)run << 8) | (unsigned long long)(255u - jj);
              X.e_info[pos] = cc | (kind << 8) | (jj << 16);
            }
            K9_WAVE_FENCE();
            if (act && u == 0u) {
              const unsigned long long gm = (vb >> (ai * D)) & (D >= 64u ? ~0ull : ((1ull << D) - 1ull));
              const uint32_t g = (uint32_t)__popcll(gm);
              uint32_t run2 = X.c_eff[cc];
              for (uint32_t t = 0; t < g; t++) run2 = min(run2, X.kt[ai * D + t]);
              const bool ended = g < D || X.kk[ai * D + g - 1u] != 0u;
              X.c_next[cc] += g; X.c_eff[cc] = run2;
              if (ended) X.c_flag[cc] |= 1u;
            }
            n += (uint32_t)__popcll(vb);
            if (lane == 0) X.stat[3]++;
            K9_WAVE_FENCE();
          }
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
        }
      }
      if (!sel_done)
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

  // ---------------- epilogue (all threads) ----------------
  const uint32_t n_done = H.i, nd = H.nd;
  {   // the dirty nodes' live state back to HBM
    const KbDev &d = *a.dev;
    for (uint32_t slot = tid; slot < nd; slot += K9_THREADS) {
      const unsigned long long *sl = slots + (size_t)slot * K9_NF;
      const uint32_t n = (uint32_t)sl[F_NODE_NMASK];
      d.idle[n] = u2d(sl[F_IDLE0]);
      d.idle[(size_t)d.NP + n] = u2d(sl[F_IDLE1]);
      d.rel[n] = u2d(sl[F_REL0]);
      d.rel[(size_t)d.NP + n] = u2d(sl[F_REL1]);
      d.nzc[n] = (long long)u2d(sl[F_NZC]);
      d.nzm[n] = (long long)u2d(sl[F_NZM]);
      d.podcnt[n] = d.maxpods[n] - (int)(uint32_t)(sl[F_CLS_LEFT] >> 32);
      if (a.has_ports) d.ports[n] = sl[F_PORTS];
    }
  }
  {
    // decision records; task-table side of ssn.Allocate / ssn.Pipeline for the committed rows (job.UpdateTaskStatus,
    // task.NodeName: framework/session.go:243,205; api/node_info.go:206-209); multi-GPU: per-node committed deltas of the
    // rows this rank owns [dIdle R][dRel R][dnzc][dnzm][dpodcnt] x NP (integer-valued: the float64 sums are exact)
    const KbDev &d = *a.dev;
    const KbRound &r = *a.round;
    for (uint32_t i = tid; i < n_done; i += K9_THREADS) {
      const unsigned long long rec = ldec[i];
      a.dec[i] = rec;
      const uint32_t n = (uint32_t)rec, kind = (uint32_t)(rec >> 32);
      if (n == KB_NONE_U32) continue;
      const KbRowDesc &k = desc[i];
      const uint32_t t = k.task;
      d.t_status[t] = kind ? KB_TASK_PIPELINED : KB_TASK_ALLOCATED;
      d.t_node[t] = n;
      d.t_counted[t] = 1;
      if (!kind) d.j_allocated[d.t_job[t]] = 1;   // ssn.Allocate ran for the job: its Allocated tasks are dispatched if it is ready
      if (d.t_ip_cls_inc) {   // inter-pod affinity: the pod joins ni.Tasks of its node and, when Allocated, the PodLister's allocated set
        for (uint32_t w = 0; w < d.ip_Wp; w++) {
          unsigned long long cm = d.t_ip_cls_inc[(size_t)t * d.ip_Wp + w];
          while (cm) {
            const uint32_t pcl = 64u * w + (uint32_t)__ffsll((unsigned long long)cm) - 1u;
            cm &= cm - 1ull;
            atomicAdd(&d.ip_cls_unbound[(size_t)pcl * d.NP + n], 1);
          }
        }
        atomicMin(d.ip_z, n);
        if (!kind)
          for (uint32_t w = 0; w < d.ip_Wc; w++) {
            unsigned long long im = d.t_ip_inc[(size_t)t * d.ip_Wc + w];
            while (im) {
              const uint32_t c = 64u * w + (uint32_t)__ffsll((unsigned long long)im) - 1u;
              im &= im - 1ull;
              atomicAdd(&d.ip_ctr_total[c], 1);
              const uint32_t dm = d.ip_ctr_dom[(size_t)c * d.NP + n];
              if (dm != KB_NONE_U32) atomicAdd(&d.ip_ctr_count[(size_t)c * d.ip_D + dm], 1);
            }
          }
      }
      if (a.has_delta && i >= r.own_row0 && i < r.own_row1) {
        double res0 = k.init0, res1 = k.init1;
        if (!(k.flags & 1)) { res0 = d.t_res[t]; res1 = d.t_res[(size_t)d.T + t]; }
        double *dv = r.delta + (size_t)(kind ? d.R : 0) * d.NP;
        atomicAdd(&dv[n], -res0);
        atomicAdd(&dv[(size_t)d.NP + n], -res1);
        const uint32_t km = k.resmask;
        if (km) {
          const uint32_t nm = d.nmask[n];
          const uint32_t has_map = kind ? (nm >> 31) : (nm & 0x7FFFFFFFu);
          if (has_map) {
            uint32_t dd = 2, m2 = km;
            while (m2) {
              if (m2 & 1u) atomicAdd(&dv[(size_t)dd * d.NP + n], -d.t_res[(size_t)dd * d.T + t]);
              m2 >>= 1; dd++;
            }
          }
        }
        double *tail = r.delta + (size_t)2 * d.R * d.NP;
        atomicAdd(&tail[n], (double)k.nzc);
        atomicAdd(&tail[(size_t)d.NP + n], (double)k.nzm);
        atomicAdd(&tail[(size_t)2 * d.NP + n], 1.0);
      }
    }
  }
  if (tid == 0) {
    a.result[0] = n_done; a.result[1] = H.reason; a.result[2] = nd; a.result[3] = H.n_dirty_rows;
    a.result[4] = H.n_runs; a.result[5] = H.n_slow; a.result[6] = 0; a.result[7] = 0;
    if constexpr (SEL) { a.result[6] = X.stat[0] | (X.stat[1] << 16); a.result[7] = X.stat[2] | (X.stat[3] << 16); }
    if (a.round->chain) *a.round->chain = H.reason == KB_REASON_DONE ? a.round->chain_tag : 0u;   // the round queued behind this one runs only then
    unsigned long long *st = reinterpret_cast<unsigned long long *>(a.result) + KB_OUT_STAMP0;
    st[2] = t_start;
    st[3] = wall_clock64();
  }
  // ---- fast rounds: mirror the header and the decision records into pinned host memory and publish the round's sequence
  //      number last; the host spins on that word instead of paying a stream synchronisation + D2H copy per round
  if (a.host_out) {
    __syncthreads();
    const unsigned long long *hdr = reinterpret_cast<const unsigned long long *>(a.result);
    for (uint32_t i = tid; i < KB_OUT_HDR; i += K9_THREADS) if (i != KB_OUT_SEQ) a.host_out[i] = hdr[i];
    for (uint32_t i = tid; i < n_done; i += K9_THREADS) a.host_out[KB_OUT_HDR + i] = ldec[i];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&a.host_out[KB_OUT_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <bool SEL>
static void k9_launch(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_commit_run<SEL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const size_t sh = k9_layout(r.n_rows, r.n_mrows, r.L, d.NP, d.R, SEL).total;
  K9KernArgs ka;
  ka.dev = d;
  ka.round = r;
  KbCommitArgs &a = ka.hot;
  a.dev = nullptr; a.round = nullptr;   // set from the kernel-argument segment inside the kernel
  a.keys = r.keys; a.dec = r.dec; a.desc = r.desc; a.result = r.result; a.trace = r.trace;
  a.n_rows = r.n_rows; a.n_mrows = r.n_mrows; a.L = r.L; a.cap = r.cap; a.N = d.N; a.NP = d.NP; a.T = d.T;
  a.fit_mode = r.fit_mode; a.backfill = r.backfill; a.pred_enabled = d.pred_enabled; a.score_enabled = d.score_enabled;
  a.wL = d.wL; a.wM = d.wM; a.wB = d.wB;
  a.use_crow = (d.pred_enabled && d.crows != nullptr && d.n_nc <= 32) ? 1u : 0u;
  a.has_delta = r.delta != nullptr ? 1u : 0u;
  a.has_aff = ((d.aff != nullptr && d.score_enabled) || d.t_ip_subject != nullptr) ? 1u : 0u;
  a.has_ports = d.ports != nullptr ? 1u : 0u;
  a.R = d.R;
  a.batch = 0;
  a.host_out = r.host_out;
  a.seq = r.seq;
  a.node_bits = kb_node_bits(d.NP);
  a.prewalk = 0;
  static const bool helpers_off = getenv("KB_WARM_HELPERS_OFF") && getenv("KB_WARM_HELPERS_OFF")[0] == '1';   // A/B switch
  hipLaunchKernelGGL(k_commit_run<SEL>, dim3(helpers_off ? 1u : KB_WARM_GRID), dim3(K9_THREADS), sh, (hipStream_t)stream, ka);
}
void kb_launch_commit(const KbDev &d, const KbRound &r, void *stream) { k9_launch<false>(d, r, stream); }       // KB_COMMIT_RUN
void kb_launch_commit_sel(const KbDev &d, const KbRound &r, void *stream) { k9_launch<true>(d, r, stream); }    // KB_COMMIT_SELECT
