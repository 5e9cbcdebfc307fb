// kb_k9.hpp — what the two run-at-a-time commit kernels share (kb_commit.hip: k_commit_run, kb_commit_sel.hip: k_commit_select): the LDS
// layout of a round, the dirty-slot record, the per-pair evaluation against a slot (k9_eval_v: kb_eval.hpp's arithmetic on LDS state), the
// prologue that stages a round into LDS and the epilogue that writes it back.  gfx950 / CDNA4, wave64.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "kb_device.h"
#include "kb_eval.hpp"
#include "kb_warm.hpp"

#define K9_THREADS 512   // wave 0: the sequential part; waves 1..4: one dirty slot per thread; all: prologue / epilogue
#define K9_MAXRUN 64
#define K9_SEL_MAXRUN 32   // the selection kernel: a run and the run in front of it share the 64 lanes of one candidate fetch
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

// ---- the selection kernel's extra LDS (kb_commit_sel.hip) ----
// A prep wave's scratch: the clean entries of the shape's list as of its walk (more than the run's rows: the runs in front of it may still
// take some; filtered against the dirty bitmap when the run's turn comes).
struct K9Prep {
  uint32_t ckey[64], cpos[64];                   // the fetched entries: key and list position, best first
};
// The sequence words the waves hand runs over with (they only grow; acquire / release at workgroup scope)
struct K9Sync {
  uint32_t seq_done;                  // runs committed (wave 0)
  uint32_t seq_cand;                  // runs whose candidates are in LDS (the run's prep wave)
  uint32_t seq_dk[4];                 // runs whose dirty keys are in LDS, per evaluating wave
  uint32_t seq_early[4];              // runs whose EARLY dirty keys are in LDS (final unless the run in front lists changed slots)
  uint32_t seq_spec;                  // runs whose candidates are in LDS ON THE ASSUMPTION that the run in front takes every candidate of its own
  uint32_t spec_ok[4];                // [m & 3]: that assumption held for run m (wave 0, in front of seq_done = m; four deep: a prep wave may read its run's word late)
  uint32_t stop, err;                 // the round is over (wave 0); a wait ran out of patience
  uint32_t nd_at[4];                  // [k & 3]: dirty slots when run k starts (wave 0, in front of seq_done = k)
  uint32_t ncand_at[4];               // [k & 3]: candidates of run k (its prep wave, in front of seq_cand = k + 1)
  uint32_t mdk[2][4];                 // [k & 1][w]: best dirty key of run k's shape among evaluating wave w's slots
  uint32_t mdko[2][4];                // ... among the slots that were dirty before run k - 1 (the early evaluation; run k - 1's consumed candidates: dkb)
  uint32_t chg[2][10];                // [k & 1]: slots run k changed other than by consuming a clean candidate once (what an early evaluation for run k + 1 got wrong)
};
struct K9Sel {
  K9Sync sync;
  K9Prep prep[3];                     // one per prep wave
  uint16_t runs[KB_K5_MAX_ROWS];      // first row of run k
  uint32_t brk[8], stm[8];            // row masks: the row cannot join its predecessor / the row starts a run
  struct Cand { uint32_t ckey[64], cpos[64], ckind[64], ck1[64], crnm[64]; } cand[2];   // [m & 1]: run m's candidates (its prep wave writes them — possibly ahead, see seq_spec —, wave 0 reads them)
  uint32_t dkb[2][K9_MAXSLOTS];       // [k & 1][t]: key(shape of run k, dirty slot t)
  // a shot of the selection (wave 0's own scratch): the contenders' slots (bit 31: a clean candidate, its slot one placement ahead), the table's
  // keys and prefix minima (lane g * D + u), the pool's keys between two shots ([t]: dirty slot t, [256 + i]: this run's candidate i once consumed)
  uint32_t c_slot[64], c_sorted[64];   // ... in pool order; the rem best by key, best first (what the table and the AddTask step read)
  alignas(16) uint32_t c_key[64];      // their keys (pool order)
  alignas(16) uint32_t e_eff[64];
  uint32_t e_key[64];
  uint32_t pool[K9_MAXSLOTS + 64];
  uint32_t tr[4];                     // KB_K9_TRACE: cycles the other waves wait / work (kb_commit_sel.hip)
  uint32_t stat[4];                   // runs committed with every pick a clean first placement / by look-ahead passes / handed to the serial loop; look-ahead passes
};

// dynamic LDS layout for a round of n_rows rows and n_shapes distinct shapes (sel: with the selection kernel's extra block)
struct K9Layout {
  uint32_t slots, rowres, sinit, shapes, desc, rinfo, dec, hdr, dk, ckey, cpos, cursor, shp, lists, bitmap, sel, total;   // byte offsets
  uint32_t Lp, RS;
};
__host__ __device__ inline K9Layout k9_layout(uint32_t n_rows, uint32_t n_shapes, uint32_t L, uint32_t NP, int R, bool sel = false) {
  K9Layout o;
  o.RS = R > 2 ? (uint32_t)(R - 2) : 0u;
  o.Lp = L;
  uint32_t off = 0;
  o.slots = off;  off += (n_rows + K9_MAXRUN) * K9_NF * 8u;   // + a run's worth: P2 writes every candidate's post-placement state
  o.rowres = off; off += 3u * (uint32_t)R * 8u;               // three thirds: the selection kernel prepares runs ahead of the one being committed
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
// fits_idle (when asked for; allocate's fit_mode only): InitResreq.LessEqual(Idle) of the same evaluation — allocate.go:160's Allocate / Pipeline test
__device__ __forceinline__ uint32_t k9_eval_v(const KbCommitArgs &a, const K9Shape &sh, const K9St &v, const K9Sc &sc, gptrd gi, gptrd gr, const double *si,
                                              uint32_t adj_mask, double adj_mul, const double *rq, uint32_t nb, uint32_t nmaskbits, bool *fits_idle = nullptr) {
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
    if (fits_idle) *fits_idle = fi;
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

// What every workgroup of a commit launch does first.  Returns true when this workgroup has nothing (more) to do: the round was chained to a
// predecessor that stopped early (workgroup 0 reports KB_REASON_SKIPPED), or this is a helper workgroup (kb_warm.hpp), which warms its slice
// of the node state into the XCD's L2 and leaves.
// the round does not run (queued behind a round that stopped early, or its candidate lists never arrived): nothing was evaluated or committed
__device__ __forceinline__ void k9_publish_skipped(const KbCommitArgs &a) {
  *a.round->chain = 0u;
  a.result[0] = 0; a.result[1] = KB_REASON_SKIPPED;
  if (a.host_out) {
    a.host_out[0] = (unsigned long long)KB_REASON_SKIPPED << 32;
    __threadfence_system();
    __hip_atomic_store(&a.host_out[KB_OUT_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__device__ __forceinline__ bool k9_preamble(const KbCommitArgs &a) {
  if (a.round->chain_expect != 0u && *a.round->chain != a.round->chain_expect) {   // chained to a round that stopped early: skip
    if (threadIdx.x == 0 && blockIdx.x == 0) k9_publish_skipped(a);
    return true;
  }
  if (blockIdx.x != 0) {
    if ((blockIdx.x & 7u) != 0u) return true;
    const uint32_t h = blockIdx.x / 8u - 1u, lines = a.NP / 16;
    if (h >= KB_WARM_HELPERS) return true;   // (a launch that carries repair workgroups may be longer than KB_WARM_GRID)
    const uint32_t l0 = (uint32_t)(((unsigned long long)h * lines) / KB_WARM_HELPERS), l1 = (uint32_t)(((unsigned long long)(h + 1) * lines) / KB_WARM_HELPERS);
    const unsigned long long acc = kb_warm_lines(*a.dev, a.keys, 0, l0, l1, threadIdx.x, K9_THREADS);   // the lists are copied to LDS by the prologue itself
    if (acc == 0x123456789abcdefull) a.result[15] = 1;   // keep the loads alive (never true)
    return true;
  }
  return false;
}

#define K9_LDS_VIEWS(lo)                                                                                           \
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(k9_base_ + lo.slots); \
  double *rowres = reinterpret_cast<double *>(k9_base_ + lo.rowres); \
  double *sinit = reinterpret_cast<double *>(k9_base_ + lo.sinit); \
  K9Shape *shapes = reinterpret_cast<K9Shape *>(k9_base_ + lo.shapes); \
  KbRowDesc *desc = reinterpret_cast<KbRowDesc *>(k9_base_ + lo.desc); \
  uint4 *rinfo = reinterpret_cast<uint4 *>(k9_base_ + lo.rinfo); \
  unsigned long long *ldec = reinterpret_cast<unsigned long long *>(k9_base_ + lo.dec); \
  K9Hdr &H = *reinterpret_cast<K9Hdr *>(k9_base_ + lo.hdr); \
  uint32_t *dk = reinterpret_cast<uint32_t *>(k9_base_ + lo.dk); \
  uint32_t *ckey = reinterpret_cast<uint32_t *>(k9_base_ + lo.ckey); \
  uint32_t *cpos = reinterpret_cast<uint32_t *>(k9_base_ + lo.cpos); \
  uint32_t *cursor = reinterpret_cast<uint32_t *>(k9_base_ + lo.cursor); \
  uint32_t *shp = reinterpret_cast<uint32_t *>(k9_base_ + lo.shp); \
  uint32_t *lists = reinterpret_cast<uint32_t *>(k9_base_ + lo.lists); \
  uint32_t *bitmap = reinterpret_cast<uint32_t *>(k9_base_ + lo.bitmap); \
  const uint32_t RS = lo.RS, Lp = lo.Lp; \
  const uint32_t nb = a.node_bits, nmaskbits = (1u << nb) - 1u;

// ---------------- prologue (all threads): the round staged into LDS ----------------
// One thread per tag waits until tags[i] == tag (i < n <= the workgroup); false when that never happens: the chain word was cleared meanwhile (a
// repair workgroup gave up on its stale list) or the bound ran out.  H.stop must read 0 in LDS when the first thread arrives; ends in a barrier
// behind which every thread of the workgroup may read what the tags cover (the waiting threads' acquire loads invalidate the CU's caches).
// ACQ: the tags cover plain stores of another launch (the arg-max launch's: an acquire per look); otherwise what they cover was stored through to
// memory and is read with agent-scope loads (the repair workgroups' lists), and the look is a relaxed one
template <bool ACQ>
__device__ __forceinline__ bool k9_wait_tags(const KbCommitArgs &a, K9Hdr &H, const uint32_t *tags, const uint32_t tag, const uint32_t n, const uint32_t tid) {
  if (tid < n) {
    bool gone = false;
    uint32_t spins = 0;
    while ((ACQ ? __hip_atomic_load(&tags[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(&tags[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != tag) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 15u) == 0u) {
        if (a.round->chain_expect != 0u && __hip_atomic_load(a.round->chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.round->chain_expect) { gone = true; break; }
        if (spins > (1u << 22)) { gone = true; break; }
      }
    }
    if (gone) H.stop = 1u;
  }
  __syncthreads();
  return H.stop == 0u;
}

// maxrun: longest run of rows that share one evaluation (K9_MAXRUN); 0: the caller derives the run lengths itself (the selection kernel: from its
// run-start bitmap).  `stamps` (trace builds): the 100 MHz clock behind the staging barrier and behind the shape tables.
// A launch that carries its own repair workgroups (KbRound::lists_ready): nothing the second stream wrote is read before the arg-max launch's
// tags are seen (the row descriptors come from its matrix launch), everything that does not need the lists is staged next, and the lists last,
// behind the repair workgroups' tags.  Returns false when a tag never came: the caller publishes KB_REASON_SKIPPED and leaves.
__device__ __forceinline__ bool k9_prologue(const KbCommitArgs &a, const K9Layout &lo, unsigned char *k9_base_, const uint32_t tid, const uint32_t maxrun,
                                            unsigned long long *stamps = nullptr) {
  const uint32_t S = a.n_mrows, W = a.n_rows;
  K9_LDS_VIEWS(lo)
  (void)slots; (void)rowres; (void)ldec; (void)dk; (void)ckey; (void)cpos; (void)RS;
  const bool fused = a.round->lists_ready != nullptr;
  for (uint32_t w = tid; w < a.NP / 32; w += K9_THREADS) bitmap[w] = 0;
  if (tid == 0) { H.i = 0; H.nd = 0; H.reason = KB_REASON_DONE; H.stop = 0; H.ncand = 0; H.n_dirty_rows = 0; H.n_runs = 0; H.n_slow = 0; H.cur_s = 0; H.cur_r = 0; H.cur_fl = 0; H.cur_km = 0; }
  if (fused) {
    __syncthreads();
    if (!k9_wait_tags<true>(a, H, a.round->ready, a.round->ready_tag, S, tid)) return false;
  }
  // Candidate lists (64-bit keys of K3 -> compact 32-bit keys) and row descriptors.  Every load of a thread is issued before the first one is
  // waited for: with one load per loop iteration, each waited for, a thread made 13 + 4 round trips to L2 / HBM one after the other at
  // 25 shapes x 257 entries and 256 rows — in front of every round, with nothing else on the CU to hide them
  constexpr uint32_t UD = 4, UL = 16;
  const uint32_t tot = S * Lp;   // [S][L], Lp == L
  auto stage_lists = [&](uint32_t lbase0) {
    for (uint32_t lbase = lbase0; lbase < tot; lbase += K9_THREADS * UL) {
      unsigned long long lv[UL];
#pragma unroll
      for (uint32_t u = 0; u < UL; u++) {   // (clamped, not predicated: S >= 1 with W >= 1)
        const unsigned long long *kp = &a.keys[min(lbase + u * K9_THREADS + tid, tot - 1u)];
        lv[u] = fused ? __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *kp;   // (fused: stored through by the repair workgroups of this launch)
      }
#pragma unroll
      for (uint32_t u = 0; u < UL; u++) {
        const uint32_t idx = lbase + u * K9_THREADS + tid;
        const unsigned long long k64 = lv[u];
        if (idx < tot) lists[idx] = k64 ? (((KB_KEY_SCORE(k64) + 1u) << nb) | (nmaskbits - KB_KEY_NODE(k64))) : 0u;
      }
    }
  };
  {
    const unsigned long long *dsrc = reinterpret_cast<const unsigned long long *>(a.desc);
    unsigned long long *ddst = reinterpret_cast<unsigned long long *>(desc);
    const uint32_t dwords = W * (uint32_t)(sizeof(KbRowDesc) / 8);
    static_assert(KB_K5_MAX_ROWS * (sizeof(KbRowDesc) / 8) <= K9_THREADS * UD, "the descriptors are staged in one batch");
    unsigned long long dv[UD], lv[UL] = {};
#pragma unroll
    for (uint32_t u = 0; u < UD; u++) dv[u] = dsrc[min(u * K9_THREADS + tid, dwords - 1u)];
    if (!fused && tot) {   // the first batch of the lists in the same round trip
#pragma unroll
      for (uint32_t u = 0; u < UL; u++) lv[u] = a.keys[min(u * K9_THREADS + tid, tot - 1u)];
    }
#pragma unroll
    for (uint32_t u = 0; u < UD; u++) { const uint32_t w = u * K9_THREADS + tid; if (w < dwords) ddst[w] = dv[u]; }
    if (!fused) {
#pragma unroll
      for (uint32_t u = 0; u < UL; u++) {
        const uint32_t idx = u * K9_THREADS + tid;
        const unsigned long long k64 = lv[u];
        if (idx < tot) lists[idx] = k64 ? (((KB_KEY_SCORE(k64) + 1u) << nb) | (nmaskbits - KB_KEY_NODE(k64))) : 0u;
      }
      stage_lists(K9_THREADS * UL);
    }
  }
  for (uint32_t s = tid; s < S; s += K9_THREADS) cursor[s] = 0;
  __syncthreads();
  if (stamps) stamps[0] = wall_clock64();
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
  if (stamps) stamps[1] = wall_clock64();

  // run table: rows i .. i + r - 1 share a shape and take the shape's own request values (a row whose Resreq differs from its
  // InitResreq, or whose score needs renormalising, is a run of its own)
  for (uint32_t i = tid; i < W; i += K9_THREADS) {
    const KbRowDesc &k = desc[i];
    const uint32_t sl = k.slot, fl = k.flags, km = k.resmask;
    const bool plain = (fl & 1u) && (km == 0u || (fl & 4u));
    uint32_t r = 1;
    if (plain && !(fl & 2u))   // (one LDS round trip per step, and the rows at the head of a stretch take maxrun - 1 of them: 3 us of every round at 32)
      while (r < maxrun && i + r < W && desc[i + r].slot == sl && desc[i + r].flags == fl && desc[i + r].resmask == km) r++;
    rinfo[i] = make_uint4(r, sl, fl, km);
  }
  if (fused) {   // the candidate lists last: everything above ran beside the workgroups that repair them
    if (!k9_wait_tags<false>(a, H, a.round->lists_ready, a.round->lists_tag, S, tid)) return false;
    stage_lists(0u);
  }
  __syncthreads();
  return true;
}

// ---------------- epilogue (all threads): dirty slots, decision records, task table, result words, host mirror ----------------
// w6 / w7: result words 6 and 7 (0 in the run kernel; the selection kernel's statistics)
// job_known / job_of_my_row: the caller fetched t_job of row `tid`'s task long ago (the selection kernel: behind its prologue)
__device__ __forceinline__ void k9_epilogue(const KbCommitArgs &a, const K9Layout &lo, unsigned char *k9_base_, const uint32_t tid, const unsigned long long t_start,
                                            const uint32_t w6, const uint32_t w7, const bool job_known = false, const uint32_t job_of_my_row = 0u) {
  K9_LDS_VIEWS(lo)
  (void)rowres; (void)sinit; (void)shapes; (void)rinfo; (void)dk; (void)ckey; (void)cpos; (void)cursor; (void)shp; (void)lists; (void)bitmap; (void)RS; (void)Lp; (void)nb; (void)nmaskbits;
  const uint32_t n_done = H.i, nd = H.nd;
  // Loads whose result is wanted further down, issued in front of the write-back so that their round trips run beside it: the header words
  // earlier launches of the round left in the result block (its matrix / arg-max stamps; the mirror below copies the block), and the job of
  // "my" committed row (n_done <= the window < the workgroup: one row per thread).  The words this kernel writes itself reach the mirror through LDS
  // (mh: the last two slots, beyond every dirty slot) instead of a read-back from global memory behind the barrier.
  const unsigned long long *hdr = reinterpret_cast<const unsigned long long *>(a.result);
  unsigned long long *mh = slots + (size_t)(a.n_rows + K9_MAXRUN - 2u) * K9_NF;   // (2 * K9_NF >= KB_OUT_HDR words)
  static_assert(2u * K9_NF >= KB_OUT_HDR, "the header's LDS copy lives in two slots");
  // the decision records' copies in pinned host memory first: the stores with the longest way to go, and the fence below waits for them
  if (a.host_out)
    for (uint32_t i = tid; i < n_done; i += K9_THREADS) a.host_out[KB_OUT_HDR + i] = ldec[i];
  unsigned long long hpre = 0ull;
  // (agent-scope loads: in a launch that carries its repair workgroups two of these words — the round's stamps — were stored through by repair row 0
  //  on another CU a moment ago; a plain load may be served an earlier round's value by this CU's caches)
  if (a.host_out && tid < KB_OUT_HDR) hpre = __hip_atomic_load(const_cast<unsigned long long *>(&hdr[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t my_job = job_of_my_row;
  if (!job_known && tid < n_done) my_job = a.dev->t_job[desc[tid].task];
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
      if (!kind) d.j_allocated[i == tid ? my_job : d.t_job[t]] = 1;   // ssn.Allocate ran for the job: its Allocated tasks are dispatched if it is ready
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
    a.result[4] = H.n_runs; a.result[5] = H.n_slow; a.result[6] = w6; a.result[7] = w7;
    if (a.round->chain) *a.round->chain = H.reason == KB_REASON_DONE ? a.round->chain_tag : 0u;   // the round queued behind this one runs only then
    unsigned long long *st = reinterpret_cast<unsigned long long *>(a.result) + KB_OUT_STAMP0;
    const unsigned long long t_end = wall_clock64();
    st[2] = t_start;
    st[3] = t_end;
    mh[0] = (unsigned long long)n_done | ((unsigned long long)H.reason << 32); mh[1] = (unsigned long long)nd | ((unsigned long long)H.n_dirty_rows << 32);
    mh[2] = (unsigned long long)H.n_runs | ((unsigned long long)H.n_slow << 32); mh[3] = (unsigned long long)w6 | ((unsigned long long)w7 << 32);
    mh[KB_OUT_STAMP0 + 2] = t_start; mh[KB_OUT_STAMP0 + 3] = t_end;
  }
  // ---- fast rounds: mirror the header and the decision records into pinned host memory and publish the round's sequence
  //      number last; the host spins on that word instead of paying a stream synchronisation + D2H copy per round
  if (a.host_out) {
    __syncthreads();
#ifdef KB_K9_TRACE
    (void)hpre;
    for (uint32_t i = tid; i < KB_OUT_HDR; i += K9_THREADS) if (i != KB_OUT_SEQ) a.host_out[i] = hdr[i];   // (the trace words were written by thread 0 a moment ago)
#else
    if (tid < KB_OUT_HDR && tid != KB_OUT_SEQ) a.host_out[tid] = (tid < 4u || tid == KB_OUT_STAMP0 + 2u || tid == KB_OUT_STAMP0 + 3u) ? mh[tid] : hpre;
#endif
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&a.host_out[KB_OUT_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// host side: 160 KiB of dynamic LDS for a commit kernel (kb_warm.hpp: kb_allow_lds)
static inline void k9_allow_full_lds(const void *fn, bool (&set_on)[64]) { kb_allow_lds(fn, 160 * 1024, set_on); }

// host side: the kernel-argument block of a commit launch
static inline void k9_fill_args(K9KernArgs &ka, const KbDev &d, const KbRound &r) {
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
  a.whole = d.whole;
}
