// kb_commit_batch.hip — K5, batched-speculative variant of the sequential commit (gfx950 / CDNA4, wave64).
//
// Best when CLEAN nodes win most rows (spreading scores: the default nodeorder weights): 16-32 rows, across shapes, are
// speculated at once and validated in parallel; a row won by a node the round already changed cuts the batch short.  When dirty
// nodes win most rows (bin-packing weights) half of the batches are cut at their first row and the run-at-a-time kernel of
// kb_commit.hip is about twice as fast; the engine picks per round from the measured share of dirty-won rows
// (kb_engine.cpp:choose_commit_kernel).  Both kernels compute the same decisions, bit for bit.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "kb_device.h"
#include "kb_eval.hpp"
#include "kb_warm.hpp"

#define KB_K5_THREADS 1024

// class_row: nullptr -> look the class pair up in the global bit table; otherwise the task class's row of the table (bit nc)
// ------------------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------------------
// max over keys (bit patterns of positive normal doubles, or 0) with DPP moves + v_max_f64
#define KB_DPP_STEP(v, ctrl, row_mask)                                                                    \
  do {                                                                                                    \
    int _lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), (ctrl), (row_mask), 0xf, false);          \
    int _hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), (ctrl), (row_mask), 0xf, false);          \
    v = fmax(v, __hiloint2double(_hi, _lo));                                                              \
  } while (0)
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k) {
  double v = __longlong_as_double((long long)k);
  KB_DPP_STEP(v, 0xB1, 0xf);    // quad_perm [1,0,3,2]
  KB_DPP_STEP(v, 0x4E, 0xf);    // quad_perm [2,3,0,1]
  KB_DPP_STEP(v, 0x141, 0xf);   // row_half_mirror
  KB_DPP_STEP(v, 0x140, 0xf);   // row_mirror: every lane of a 16-lane row holds the row maximum
  KB_DPP_STEP(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
  KB_DPP_STEP(v, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave maximum
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
// max over each aligned group of 8 lanes (every lane of the group gets it)
__device__ __forceinline__ unsigned long long oct_max_key(unsigned long long k) {
  double v = __longlong_as_double((long long)k);
  KB_DPP_STEP(v, 0xB1, 0xf);
  KB_DPP_STEP(v, 0x4E, 0xf);
  KB_DPP_STEP(v, 0x141, 0xf);
  return (unsigned long long)__double_as_longlong(v);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) { return wave_max_key(v); }
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ uint32_t lane_prefix_popc(unsigned long long ballot_mask, uint32_t lane) {
  return __popcll(ballot_mask & ((1ull << lane) - 1ull));
}

// inclusive scan inside the wave: DPP row shifts, then row broadcasts (the sequence LLVM's buildScan emits)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
  return v;
}


__device__ __forceinline__ bool bit_test(const uint32_t *bm, uint32_t n) { return (bm[n >> 5] >> (n & 31)) & 1u; }

// LDS layout of the commit kernel for slot capacity `cap`:
//   8-byte tables [10][cap]: idle0 idle1 rel0 rel1 inv_ac inv_am ac am nzc nzm   (field k of slot s at (k*cap + s)*8)
//   4-byte tables [4][cap]:  cls node left cursor
//   KbRowDesc [cap], dirty bitmap [NP/32], header
#define K5F_IDLE0 0
#define K5F_IDLE1 1
#define K5F_REL0 2
#define K5F_REL1 3
#define K5F_INVAC 4
#define K5F_INVAM 5
#define K5F_AC 6
#define K5F_AM 7
#define K5F_NZC 8
#define K5F_NZM 9
#define K5_NF8 10
#define K5_WAVES (KB_K5_THREADS / 64)

// Float64-only evaluation (kb_eval.hpp: score_core_f64): the k8s scorers' int64 quantities (allocatable, non-zero request sums) are
// kept as doubles in the LDS slot table and in the staged row descriptors (exact below 2^48, kb_session_load rejects more), so an
// evaluation has no 64-bit integer multiply and no int64 <-> double conversion.
static_assert(sizeof(KbRowDesc) == 56 && offsetof(KbRowDesc, nzc) == 16 && offsetof(KbRowDesc, nzm) == 24, "staging converts words 2 and 3 of a descriptor");
struct TaskValsD {
  double init0, init1, nzc, nzm;
  uint32_t cls, active, task, pad;
  unsigned long long conf;
};
struct NodeValsD {
  double idle0, idle1, rel0, rel1, ac, am, nzc, nzm, inv_ac, inv_am;
  uint32_t cls;
  int slots;
  unsigned long long ports;
};
__device__ __forceinline__ double u2d(unsigned long long v) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ unsigned long long d2u(double v) { return (unsigned long long)__double_as_longlong(v); }

__device__ __forceinline__ NodeValsD k5_slot_vals(const unsigned long long *tab, const uint32_t *t_cls, const int *t_left, uint32_t cap, uint32_t slot,
                                               const unsigned long long *ptab = nullptr) {
  NodeValsD nv;
  nv.idle0 = u2d(tab[K5F_IDLE0 * cap + slot]);
  nv.idle1 = u2d(tab[K5F_IDLE1 * cap + slot]);
  nv.rel0 = u2d(tab[K5F_REL0 * cap + slot]);
  nv.rel1 = u2d(tab[K5F_REL1 * cap + slot]);
  nv.inv_ac = u2d(tab[K5F_INVAC * cap + slot]);
  nv.inv_am = u2d(tab[K5F_INVAM * cap + slot]);
  nv.ac = u2d(tab[K5F_AC * cap + slot]);
  nv.am = u2d(tab[K5F_AM * cap + slot]);
  nv.nzc = u2d(tab[K5F_NZC * cap + slot]);
  nv.nzm = u2d(tab[K5F_NZM * cap + slot]);
  nv.ports = ptab ? ptab[slot] : 0ull;
  nv.cls = t_cls[slot];
  nv.slots = t_left[slot] > 0;
  return nv;
}

// eval_pair for the commit kernel: policy scalars from the by-value argument struct, session arrays (scalar resource
// dimensions, wide class tables) through the device-memory copy of KbDev on the rare paths only
__device__ __forceinline__ uint32_t eval_pair_k5(const KbCommitArgs &a, const TaskValsD &t, const NodeValsD &n, uint32_t node, const uint32_t *class_row) {
  bool ok = true;
  if (a.fit_mode) {   // allocate.go:81
    bool fi = le_eps(t.init0, n.idle0, EPS_CPU) && le_eps(t.init1, n.idle1, EPS_MEM);
    bool fr = le_eps(t.init0, n.rel0, EPS_CPU) && le_eps(t.init1, n.rel1, EPS_MEM);
    uint32_t act = t.active >> 2;
    if (act) {
      const KbDev &d = *a.dev;
      uint32_t dd = 2;
      while (act) {
        if (act & 1u) {
          double l = d.t_init[(size_t)dd * d.T + t.task];
          fi = fi && le_eps(l, d.idle[(size_t)dd * d.NP + node], EPS_SCALAR);
          fr = fr && le_eps(l, d.rel[(size_t)dd * d.NP + node], EPS_SCALAR);
        }
        act >>= 1;
        dd++;
      }
    }
    ok = fi || (a.fit_mode != 2 && fr);   // 2: backfill, AddTask's Resreq.LessEqual(Idle) only (node_info.go:161-167)
  }
  if (a.pred_enabled) {
    ok = ok && n.slots && ((n.ports & t.conf) == 0ull);   // pod count (predicates.go:127), PodFitsHostPorts (predicates.go:181-190)
    if (class_row) {
      ok = ok && ((class_row[n.cls >> 5] >> (n.cls & 31)) & 1u);
    } else {
      const KbDev &d = *a.dev;
      if (d.compat) {
        uint32_t bit = t.cls * d.n_nc + n.cls;
        ok = ok && ((d.compat[bit >> 3] >> (bit & 7)) & 1);
      }
    }
  }
  if (!ok) return 0;
  uint32_t score = 0;
  if (a.score_enabled) score = score_core_f64(t.nzc, t.nzm, n.nzc, n.nzm, n.ac, n.am, n.inv_ac, n.inv_am, a.wL, a.wM, a.wB);
  return 0x10000u | (score & 0xFFFFu);
}

// per-lane source arrays of the one-instruction node-state fetch: fld < 10 -> 8-byte field fld; 10..12 -> cls, maxpods, podcnt
__device__ __forceinline__ void k5_field_ptrs(const KbDev &d, uint32_t fld, const unsigned long long *&g8, const uint32_t *&g4) {
  g8 = nullptr; g4 = nullptr;
  switch (fld) {
    case K5F_IDLE0: g8 = reinterpret_cast<const unsigned long long *>(d.idle); break;
    case K5F_IDLE1: g8 = reinterpret_cast<const unsigned long long *>(d.idle + d.NP); break;
    case K5F_REL0: g8 = reinterpret_cast<const unsigned long long *>(d.rel); break;
    case K5F_REL1: g8 = reinterpret_cast<const unsigned long long *>(d.rel + d.NP); break;
    case K5F_INVAC: g8 = reinterpret_cast<const unsigned long long *>(d.inv_acpu); break;
    case K5F_INVAM: g8 = reinterpret_cast<const unsigned long long *>(d.inv_amem); break;
    case K5F_AC: g8 = reinterpret_cast<const unsigned long long *>(d.acpu); break;
    case K5F_AM: g8 = reinterpret_cast<const unsigned long long *>(d.amem); break;
    case K5F_NZC: g8 = reinterpret_cast<const unsigned long long *>(d.nzc); break;
    case K5F_NZM: g8 = reinterpret_cast<const unsigned long long *>(d.nzm); break;
    case 10: g4 = d.ncls; break;
    case 11: g4 = reinterpret_cast<const uint32_t *>(d.maxpods); break;
    case 12: g4 = reinterpret_cast<const uint32_t *>(d.podcnt); break;
    default: break;
  }
}

// ------------------------------------------------------------------------------------------------------------
// K5: commit.  One workgroup walks the window in the reference's task order (allocate.go:129-193 / backfill.go:44-67).
// The reference commits one task at a time, but ~90 % of the tasks take the best CLEAN node of their
// shape (the next untouched entry of the shape's sorted candidate list).  The kernel therefore speculates K7_B rows
// at once:
//   walk      one wave hands every row of the batch the next clean entry of its shape's list, in row order, marking
//             the nodes in the dirty bitmap as it goes (so a later row of another shape skips them);
//   fetch     16 threads per row pull the 13 state fields of the row's node into a NEW dirty slot (one load each);
//   apply     one thread per row decides Allocate / Pipeline (allocate.go:160) and applies NodeInfo.AddTask
//             (api/node_info.go:172-212) to its slot, i.e. the slot holds the state AFTER the row committed;
//   evaluate  all threads: key(shape q, slot x) for every distinct shape q of the batch and every dirty slot x, old
//             (-> dmax[q]) and new (-> kb[row][q]);
//   validate  row j really takes its clean candidate iff  c_j > dmax[q_j]  and  c_j > kb[l][q_j] for every earlier
//             batch row l: then no dirty node beats it, exactly the reference's arg-max.  The first row that fails is
//             the batch's "dirty row": its winner is the best dirty key, computed by the same evaluation;
//   commit    rows before the first failure are final (decision records, cursors); later rows are rolled back
//             (bitmap bits, speculative scalar-dimension writes) and re-speculated by the next batch; the dirty row is
//             applied to the slot that owns the winning node.
// A batch costs about as much as two rows of a row-at-a-time protocol, and commits ~10 rows on the benchmark snapshot.
// ------------------------------------------------------------------------------------------------------------
#define K7_B 32u          // most rows one batch can speculate
#define K7_B_DEFAULT 32u  // rows speculated per batch (kb_config.commit_batch / KB_K5_BATCH override; doubled after a fully valid batch, capped at K7_B).  16 until
                          // round 3: with the cheaper evaluation, the pre-walk and the look-ahead the sweep 8 / 12 / 16 / 24 / 32 gives C3 62.7 / 57.3 / 55.1 / 54.1 / 53.6 ms
#define K7_D 160u   // row descriptors staged per refill (five 32-row batches: four of them find their successor staged and can pre-walk it)
#define K7_PWIN_BYTES 32768u   // LDS for the per-shape candidate windows of a round
#define K7_KQ 5    // 64 * K7_KQ >= KB_K5_MAX_ROWS + K7_B dirty slots: the row-at-a-time mode keeps one key per slot in registers
#define K7_CH 16u   // depth of the on-demand chain table
#define K7_LA_INVALID 0xFFFFFFFFFFFFFFFFull   // look-ahead key: not available (the placement would be a Pipeline, or the slot's state moved on)

static_assert(64 * K7_KQ >= KB_K5_MAX_ROWS + K7_B, "row mode keeps one key per dirty slot in registers");
struct K7Hdr {
  unsigned long long c2[2][K7_B];       // clean candidate key of batch row j (0: the list has no clean feasible node left); two buffers:
                                        // wave 0 walks batch b + 1 into the other one while the workgroup fetches / evaluates batch b
  unsigned long long dmax[K7_B];        // per distinct shape q of the batch: best key over the pre-batch dirty slots
  unsigned long long kb[K7_B][K7_B];    // [row l][shape q]: key of row l's node in its post-commit state
  unsigned long long mrow[K7_B];        // per batch row j: max of kb[l][q_j] over the earlier batch rows l < j (built by the evaluate step)
  uint32_t rowmask[K7_B];               // per distinct shape q: the batch rows that carry it
  uint32_t rep[K7_B];                   // batch row whose descriptor represents shape q
  uint32_t q_of[K7_B];                  // shape index of batch row j
  uint32_t idx2[2][K7_B];               // list position of c[j]
  uint32_t kind[K7_B];                  // 0 Allocate, 1 Pipeline
  uint32_t has_map[K7_B];               // the row's speculative commit wrote scalar dimensions (saved values are valid)
  // evaluation work list of shape q: slots [e_start, nd), then dlog[e_log0 .. e_log0 + e_nlog), then the batch's new slots
  uint32_t e_off[K7_B], e_start[K7_B], e_nlog[K7_B], e_log0[K7_B];
  KbRowDesc dbuf[K7_D];                 // row descriptors [dbase, dbase + dcnt) of the window, staged ahead of the batches
  unsigned long long kstar;             // winner of the dirty row
  uint32_t nshapes, p, dirty_row, reason, exhausted, pad;
  uint32_t n_pairs, nlog, n_full, pad3;
  uint32_t seq_rows, seq_pc, n_seq_rows, pad4;
  uint32_t n_batches, n_dirty_rows, n_refills, pad2;
  uint32_t n_prewalks, n_prewalks_used, pad5, pad6;
};

struct K7Mem {
  unsigned long long *tab;              // [K5_NF8][cap2]
  uint32_t *t_cls, *t_node;             // [cap2]
  int *t_left;                          // [cap2]
  uint32_t *cursor;                     // [cap] per shape
  uint32_t *qstamp;                     // [cap] per shape: first batch row with that shape (0xFFFFFFFF between batches)
  // per shape: best key over the dirty slots [0, dc_nd) as of dirty-log position dc_log.  Still valid later for the slots
  // it covered unless the node of dc_key itself was changed by a dirty row since (the commit step then sets dc_nd to
  // 0xFFFFFFFF); newer slots and slots changed since are simply evaluated again and max'ed in.
  unsigned long long *dc_key;           // [cap]
  uint32_t *dc_nd, *dc_log;             // [cap]
  uint32_t *dlog;                       // [cap] slots changed by dirty rows, in order
  unsigned long long *keyq;             // [cap2] keys of one shape against every dirty slot (row-at-a-time mode)
  unsigned long long *keyq1, *keyq2;    // [cap2] the same after one / two MORE placements of the run's shape on the slot (K7_LA_INVALID: that
                                        //        placement would be a Pipeline): a dirty winner's new key without an evaluation in the serial loop
  uint32_t *wcnt;                       // [cap2] placements deferred on the slot (0, 1 or 2): the look-ahead level its next win reads
  uint32_t *wlist;                      // [64] slots that carry deferred placements, in the order of their first win
  unsigned long long *chain;            // [K7_CH] one slot's keys after 1 .. K7_CH more placements, built on demand when its look-ahead runs out
  unsigned long long *ptab;             // [cap2] host-port bits of the slot's node (sessions with host ports only)
  uint32_t *bitmap;                     // [NP/32]
  double *save;                         // [K7_B][R-2] scalar-dimension values overwritten by speculative commits
  // Candidate windows, one per shape of the ROUND and persistent across its batches: pwin[s][0..WL) = entries pbase[s] ..
  // pbase[s] + WL of the shape's sorted list (K3).  Invariant: every list entry in front of pbase[s] is dirty, so a walk only ever
  // looks at the window (consumed or foreign-taken entries inside it are dirty bits in the bitmap) and slides it when it holds
  // no clean entry any more.  WL = 64, 32 or 16 so that all windows of the round fit K7_PWIN_BYTES.
  unsigned long long *pwin;
  uint32_t *pbase;                      // [cap]
  K7Hdr *H;
  uint32_t cap2, WL;
};

__host__ __device__ inline size_t k7_smem_bytes(uint32_t cap, uint32_t NP, int R) {
  size_t cap2 = (size_t)cap + K7_B;
  return cap2 * (K5_NF8 * 8 + 8 + 8 + 3 * 4) + (size_t)cap * (8 + 8 + 12) + (size_t)(NP / 32) * 4 + (size_t)K7_B * (R > 2 ? R - 2 : 0) * 8 + sizeof(K7Hdr) + 64 +
         K7_PWIN_BYTES + (size_t)cap * 4 + cap2 * 20 + 64 * 4 + 16 + K7_CH * 8;
}

// The walk: rows [ja, jb) of a batch (descriptors bd) take, in row order, successive clean entries of their shape's persistent candidate
// window (runs of consecutive rows with the same shape share one scan), marking the nodes in the dirty bitmap so that later rows of other
// shapes skip them.  Wave 0 only.  c / idx: the batch's candidate buffers; slid: bit j0 set = the window of the run that starts at row j0
// slid during this walk (the roll-back re-anchors such windows at the shape's cursor); exh / refills: list exhaustion flag and statistics.
__device__ __forceinline__ void k7_walk(const K7Mem &M, const unsigned long long *keys, uint32_t L, const KbRowDesc *bd, uint32_t ja, uint32_t jb, uint32_t nb,
                                        uint32_t lane, unsigned long long *c, uint32_t *idx, uint32_t &slid, uint32_t &exh, uint32_t &refills) {
  const uint32_t WL = M.WL;
  const uint32_t s = (lane < nb) ? (uint32_t)bd[lane < nb ? lane : 0].slot : 0xFFFFFFFFu;
  uint32_t j = ja;
  while (j < jb) {
    const uint32_t sr = (uint32_t)__builtin_amdgcn_readlane((int)s, (int)j);
    const unsigned long long diff = __ballot(lane >= j && lane < jb && s != sr);
    const uint32_t j1 = diff ? (uint32_t)(__ffsll((unsigned long long)diff) - 1) : jb;
    const uint32_t jrun = j;
    uint32_t m = j1 - j;
    unsigned long long *wq = M.pwin + (size_t)sr * WL;
    uint32_t base = M.pbase[sr];
    for (;;) {
      const bool inw = lane < WL;
      const unsigned long long wkey = inw ? wq[lane] : 0ull;
      const bool nz = inw && wkey != 0ull;
      const uint32_t node = KB_KEY_NODE(wkey);
      const bool cl = nz && !bit_test(M.bitmap, nz ? node : 0u);
      const unsigned long long clean = __ballot(cl);
      const unsigned long long zeros = __ballot(inw && wkey == 0ull);
      const uint32_t cnt = (uint32_t)__popcll(clean);
      const uint32_t take = cnt < m ? cnt : m;
      const uint32_t rank = (uint32_t)__popcll(clean & ((1ull << lane) - 1ull));
      if (cl && rank < take) {
        c[j + rank] = wkey;
        idx[j + rank] = base + lane;
        atomicOr(&M.bitmap[node >> 5], 1u << (node & 31));
      }
      j += take;
      m -= take;
      if (m == 0) break;
      if (zeros) {   // the list ended: no clean feasible node is left for the remaining rows of the run
        if (lane < m) { c[j + lane] = 0ull; idx[j + lane] = 0; }
        j += m;
        break;
      }
      // every entry of the window is dirty: slide it.  Entries taken by THIS batch may be rolled back; the roll-back step
      // re-anchors the windows that slid during the batch (slid) at the shape's cursor
      const uint32_t nbase = base + WL;
      if (nbase >= L) {   // cannot happen while L > window: reported, never silently mis-scheduled
        exh = 1;
        if (lane < m) { c[j + lane] = 0ull; idx[j + lane] = 0; }
        j += m;
        break;
      }
      if (inw) wq[lane] = (nbase + lane < L) ? keys[(size_t)sr * L + nbase + lane] : 0ull;
      if (lane == 0) M.pbase[sr] = nbase;
      refills++;
      base = nbase;
      slid |= 1u << jrun;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// `k` is a STAGED descriptor (H.dbuf): its nzc / nzm words hold the pod's non-zero request as double bit patterns
__device__ __forceinline__ TaskValsD k7_task_vals(const KbCommitArgs &a, const KbRowDesc &k) {
  TaskValsD tv;
  tv.init0 = k.init0; tv.init1 = k.init1; tv.nzc = __longlong_as_double(k.nzc); tv.nzm = __longlong_as_double(k.nzm);
  tv.cls = k.cls; tv.active = k.active; tv.task = k.task; tv.pad = 0;
  tv.conf = a.has_ports ? a.dev->t_conf[k.task] : 0ull;   // host-port sessions only: straight from the task table
  return tv;
}

// TaskInfo.Resreq cpu / memory of a row: equal to InitResreq unless an init container raised the latter (flags bit 0 clear,
// rare).  The LDS values are materialised before the branch so that the compiler does not turn "LDS address or global
// address" into flat loads (which wait on both memory counters).
__device__ __forceinline__ void k7_resreq(const KbCommitArgs &a, const KbRowDesc &k, double &res0, double &res1) {
  res0 = k.init0;
  res1 = k.init1;
  asm volatile("" : "+v"(res0), "+v"(res1));
  if (!(k.flags & 1)) { const KbDev &d = *a.dev; res0 = d.t_res[k.task]; res1 = d.t_res[(size_t)d.T + k.task]; }
}

// decision record + multi-GPU deltas of one committed row; SUB: also apply the scalar dimensions of NodeInfo.AddTask
// (the batched clean rows did that speculatively in the apply step)
template <bool SUB>
__device__ __forceinline__ void k7_commit_globals(const KbCommitArgs &a, const KbRowDesc &k, uint32_t i, uint32_t n, uint32_t kind) {
  const uint32_t km = k.resmask;
  uint32_t has_map = 0;
  if (km) {
    const KbDev &d = *a.dev;
    has_map = kind ? (d.nmask[n] >> 31) : (d.nmask[n] & 0x7FFFFFFFu);   // Sub returns early when the receiver's scalar map is nil (resource_info.go:148-153)
    if (SUB && has_map) {
      double *vec = kind ? d.rel : d.idle;
      uint32_t dd = 2, m2 = km;
      while (m2) {
        if (m2 & 1u) vec[(size_t)dd * d.NP + n] -= d.t_res[(size_t)dd * d.T + k.task];
        m2 >>= 1; dd++;
      }
    }
  }
  *reinterpret_cast<uint2 *>(&a.dec[i]) = make_uint2(n, kind);
  if (a.has_delta) {
    const KbDev &d = *a.dev;
    const KbRound &r = *a.round;
    if (i >= r.own_row0 && i < r.own_row1) {
      double res0, res1;
      k7_resreq(a, k, res0, res1);
      // per-node committed deltas of the rows this rank owns: [dIdle R][dRel R][dnzc][dnzm][dpodcnt] x NP
      double *dv = r.delta + (size_t)(kind ? d.R : 0) * d.NP;
      dv[n] -= res0;
      dv[(size_t)d.NP + n] -= res1;
      if (km && has_map) {
        uint32_t dd = 2, m2 = km;
        while (m2) {
          if (m2 & 1u) dv[(size_t)dd * d.NP + n] -= d.t_res[(size_t)dd * d.T + k.task];
          m2 >>= 1; dd++;
        }
      }
      double *tail = r.delta + (size_t)2 * d.R * d.NP;
      tail[n] += __longlong_as_double(k.nzc);   // staged descriptor: double bits
      tail[(size_t)d.NP + n] += __longlong_as_double(k.nzm);
      tail[(size_t)2 * d.NP + n] += 1.0;
    }
  }
}

// The launch passes {hot scalars, KbDev, KbRound} as ONE by-value block.  Only `hot` is named in the code (-> SGPRs); the two
// views are reached through the kernel-argument segment pointer, i.e. read from constant memory where a rare path needs them.
// make EXTRA=-DKB_K7_TRACE OUT=../libkbengine_trace.so: clocks (100 MHz s_memtime) thread 0 spends in each phase of a batch,
// barrier waits included, summed per round into words 4..7 of the output block; printed by the host under KB_K5_STATS=1
#ifdef KB_K7_TRACE
#define K7_STAMP(k) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); tacc[k] += (uint32_t)(now_ - tlast); tlast = now_; } } while (0)
#else
#define K7_STAMP(k) do { } while (0)
#endif

struct K7KernArgs {
  KbCommitArgs hot;
  KbDev dev;
  KbRound round;
};

// what a round that must not run leaves behind: the chain stays broken for the round queued behind it, the host finds KB_REASON_SKIPPED
__device__ __forceinline__ void k7_report_skipped(const KbCommitArgs &a) {
  *a.round->chain = 0u;
  a.result[0] = 0; a.result[1] = KB_REASON_SKIPPED;
  if (a.host_out) {
    a.host_out[0] = (unsigned long long)KB_REASON_SKIPPED << 32;
    __threadfence_system();
    __hip_atomic_store(&a.host_out[KB_OUT_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void __launch_bounds__(KB_K5_THREADS) k_commit_batch(const K7KernArgs ka) {
  KbCommitArgs a = ka.hot;
  {
    const unsigned char __attribute__((address_space(4))) *kp = (const unsigned char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    a.dev = (const KbDev *)(kp + offsetof(K7KernArgs, dev));
    a.round = (const KbRound *)(kp + offsetof(K7KernArgs, round));
  }
  if (a.round->chain_expect != 0u && *a.round->chain != a.round->chain_expect) {   // chained to a round that stopped early: skip
    if (threadIdx.x == 0 && blockIdx.x == 0) k7_report_skipped(a);
    return;
  }
  extern __shared__ __align__(16) unsigned char k5_smem[];
  if (blockIdx.x != 0) {   // helper workgroups (kb_warm.hpp): warm a slice of the node state into the XCD's L2 and leave
    if ((blockIdx.x & 7u) != 0u) return;
    const uint32_t h = blockIdx.x / 8u - 1u, lines = a.NP / 16;
    const size_t klines = ((size_t)a.n_mrows * a.L + 15) / 16;
    const uint32_t l0 = (uint32_t)(((unsigned long long)h * lines) / KB_WARM_HELPERS), l1 = (uint32_t)(((unsigned long long)(h + 1) * lines) / KB_WARM_HELPERS);
    const unsigned long long acc = kb_warm_lines(*a.dev, a.keys, klines, l0, l1, threadIdx.x, KB_K5_THREADS);
    if (acc == 0x123456789abcdefull) a.result[15] = 1;   // keep the loads alive (never true)
    return;
  }
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cap = a.cap, cap2 = a.cap + K7_B;
  K7Mem M;
  M.cap2 = cap2;
  M.tab = reinterpret_cast<unsigned long long *>(k5_smem);
  M.keyq = M.tab + (size_t)K5_NF8 * cap2;   // 8-byte tables first
  M.ptab = M.keyq + cap2;
  M.dc_key = M.ptab + cap2;
  M.t_cls = reinterpret_cast<uint32_t *>(M.dc_key + cap);
  M.t_node = M.t_cls + cap2;
  M.t_left = reinterpret_cast<int *>(M.t_node + cap2);
  M.cursor = reinterpret_cast<uint32_t *>(M.t_left + cap2);
  M.qstamp = M.cursor + cap;
  M.dc_nd = M.qstamp + cap;
  M.dc_log = M.dc_nd + cap;
  M.dlog = M.dc_log + cap;
  M.bitmap = M.dlog + cap;
  {
    // byte offsets from the LDS base (pointer -> integer -> pointer round trips would lose the address space)
    size_t off = (size_t)cap2 * (K5_NF8 * 8 + 8 + 8 + 3 * 4) + (size_t)cap * (8 + 8 + 12) + (size_t)(a.NP / 32) * 4;
    off = (off + 15) & ~(size_t)15;
    M.H = reinterpret_cast<K7Hdr *>(k5_smem + off);
    M.save = reinterpret_cast<double *>(k5_smem + off + sizeof(K7Hdr));
    size_t off2 = off + sizeof(K7Hdr) + (size_t)K7_B * (a.R > 2 ? a.R - 2 : 0) * 8;
    off2 = (off2 + 15) & ~(size_t)15;
    M.pwin = reinterpret_cast<unsigned long long *>(k5_smem + off2);
    M.pbase = reinterpret_cast<uint32_t *>(k5_smem + off2 + K7_PWIN_BYTES);
    size_t off3 = off2 + K7_PWIN_BYTES + (size_t)cap * 4;
    off3 = (off3 + 15) & ~(size_t)15;
    M.keyq1 = reinterpret_cast<unsigned long long *>(k5_smem + off3);
    M.keyq2 = M.keyq1 + cap2;
    M.wcnt = reinterpret_cast<uint32_t *>(M.keyq2 + cap2);
    M.wlist = M.wcnt + cap2;
    M.chain = reinterpret_cast<unsigned long long *>(k5_smem + ((off3 + (size_t)cap2 * 20 + 64 * 4 + 15) & ~(size_t)15));
  }
  const uint32_t WL = a.n_mrows <= K7_PWIN_BYTES / (64u * 8u) ? 64u : (a.n_mrows <= K7_PWIN_BYTES / (32u * 8u) ? 32u : 16u);   // n_mrows <= KB_K5_MAX_SHAPES = 256
  M.WL = WL;
  K7Hdr &H = *M.H;
  const int RS = a.R > 2 ? a.R - 2 : 0;
  const unsigned long long t_start = wall_clock64();

  for (uint32_t w = tid; w < a.NP / 32; w += KB_K5_THREADS) M.bitmap[w] = 0;
  for (uint32_t w = tid; w < cap; w += KB_K5_THREADS) { M.cursor[w] = 0; M.qstamp[w] = 0xFFFFFFFFu; M.dc_key[w] = 0ull; M.dc_nd[w] = 0; M.dc_log[w] = 0; M.pbase[w] = 0; }
  for (uint32_t w = tid; w < a.n_mrows * WL; w += KB_K5_THREADS) {   // the first window of every shape's candidate list
    const uint32_t sh = w / WL, en = w % WL;
    M.pwin[w] = (en < a.L) ? a.keys[(size_t)sh * a.L + en] : 0ull;
  }
  if (tid == 0) { H.reason = KB_REASON_DONE; H.exhausted = 0; H.n_batches = 0; H.n_dirty_rows = 0; H.n_refills = 0; H.p = 0; H.dirty_row = 0; H.pad = 0; H.nlog = 0; H.n_full = 0; H.pad2 = 0; H.pad3 = 0; H.kstar = 0ull; H.n_seq_rows = 0; H.n_prewalks = 0; H.n_prewalks_used = 0; H.pad5 = 0; H.pad6 = 0; }
  // per-thread source array of the fetch step: thread (row*16 + f) reads field f of the row's node
  // (pointers read from the KbDev copy are generic; the fetch step wants global_load, not flat_load)
  typedef const unsigned long long __attribute__((address_space(1))) *gptr8;
  typedef const uint32_t __attribute__((address_space(1))) *gptr4;
  gptr8 g8;
  gptr4 g4;
  {
    const unsigned long long *f8 = nullptr;
    const uint32_t *f4 = nullptr;
    k5_field_ptrs(*a.dev, tid & 15, f8, f4);
    g8 = (gptr8)f8;
    g4 = (gptr4)f4;
  }
  if (gridDim.x == 1) {
    // no helper workgroups (KB_WARM_HELPERS_OFF=1): this workgroup warms the XCD's L2 itself, all threads, before the loop starts
    const size_t klines = ((size_t)a.n_mrows * a.L + 15) / 16;
    const unsigned long long acc = kb_warm_lines(*a.dev, a.keys, klines, 0, a.NP / 16, tid, KB_K5_THREADS);
    for (size_t l = (size_t)(a.NP / 16) + tid; l < klines; l += KB_K5_THREADS) if (a.keys[l * 16] == 0x123456789abcdefull) H.pad = 1;
    if (acc == 0x123456789abcdefull) H.pad = 1;   // keep the loads alive
  }
  __syncthreads();

  uint32_t i0 = 0, nd = 0, n_done = 0, reason = KB_REASON_DONE, dbase = 0, dcnt = 0, nb_cur = a.batch;
  // Pre-walk: while the workgroup fetches and evaluates batch b, wave 0 — whose part in those steps the other fifteen waves take over —
  // already walks the rows of batch b + 1 (first half during the fetch step, second half during the evaluation) into the other candidate
  // buffer, assuming batch b turns out fully valid (more than half of the batches do: every candidate consumed, the next batch starts
  // right behind it with the doubled size).  If it does not, the pre-walk is undone like any unconsumed candidate: bitmap bits cleared,
  // slid windows re-anchored.  par: candidate buffer of the current batch; pw_*: the pre-walk in flight / carried into the next batch.
  uint32_t par = 0, pw_slid = 0, pw_exh = 0, pw_ref = 0, pw_use = 0, carried_slid = 0;
  static_assert(K7_B <= 32, "slid windows are tracked by the batch row that starts their run (32-bit mask)");
#ifdef KB_K7_TRACE
  uint32_t tacc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = wall_clock64();
  if (tid == 0) tacc[13] = (uint32_t)(tlast - t_start);   // prologue
#endif
  while (i0 < a.n_rows) {
    uint32_t nb = min(nb_cur, a.n_rows - i0);
    // ---- stage row descriptors: three batches' worth per refill, so most batches find theirs already in LDS
    if (i0 < dbase || i0 + nb > dbase + dcnt) {
      __syncthreads();   // wave 0 may still be reading the staged descriptors of the previous batch (roll-back)
      dbase = i0;
      dcnt = min(K7_D, a.n_rows - i0);
      const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.desc + i0);
      unsigned long long *dst = reinterpret_cast<unsigned long long *>(H.dbuf);
      for (uint32_t w = tid; w < dcnt * (uint32_t)(sizeof(KbRowDesc) / 8); w += KB_K5_THREADS) {
        unsigned long long v = src[w];
        const uint32_t fw = w % (uint32_t)(sizeof(KbRowDesc) / 8);
        if (fw == 2 || fw == 3) v = d2u((double)(long long)v);   // nzc, nzm: staged as doubles (TaskValsD)
        dst[w] = v;
      }
      __syncthreads();
    }
    const KbRowDesc *bd = H.dbuf + (i0 - dbase);
    K7_STAMP(0);
    if (a.has_aff && !a.backfill) {
      // a row whose score is normalised over its feasible set (preferred node affinity) is exact only against a fresh matrix:
      // it may be the first row of a round, nothing else; the batch stops in front of it and the round ends there
      uint32_t ja = nb;
      for (uint32_t j = (i0 == 0) ? 1u : 0u; j < nb; j++)
        if (bd[j].flags & 2) { ja = j; break; }
      if (ja == 0) { reason = KB_REASON_RENORM; n_done = i0; break; }
      nb = ja;
    }
    unsigned long long *const Hc = H.c2[par];
    uint32_t *const Hidx = H.idx2[par];
    // ---- distinct shapes of the batch (wave 0): q_of[j] = rank of the first row with row j's shape
    uint32_t slid = carried_slid;   // wave 0: windows (by first row of their run) that slid during this batch's walk
    carried_slid = 0;
    if (wave == 0) {
      const bool in = lane < nb;
      const uint32_t s = in ? (uint32_t)bd[in ? lane : 0].slot : 0u;
      if (lane < K7_B) { H.rowmask[lane] = 0u; H.mrow[lane] = 0ull; }
      if (in) atomicMin(&M.qstamp[s], lane);
      const uint32_t first = in ? M.qstamp[s] : 0xFFFFFFFFu;
      const bool isrep = in && first == lane;
      const unsigned long long repmask = __ballot(isrep);
      // evaluation work list of every distinct shape: reuse the shape's cached dirty max when its node is untouched
      uint32_t cnt = 0, q = 0;
      if (in) {
        q = (uint32_t)__popcll(repmask & ((1ull << first) - 1ull));
        H.q_of[lane] = q;
        M.qstamp[s] = 0xFFFFFFFFu;
        atomicOr(&H.rowmask[q], 1u << lane);
      }
      if (isrep) {
        const uint32_t nlog = H.nlog;
        unsigned long long ck = M.dc_key[s];
        uint32_t start = M.dc_nd[s];
        const uint32_t log0 = M.dc_log[s];
        uint32_t nl = nlog - log0;
        const bool full = start == 0xFFFFFFFFu || nl > 16;   // invalidated by a dirty row on its arg-max node, or too stale
        if (full) { start = 0; nl = 0; ck = 0ull; }
        cnt = (nd - start) + nl + nb;
        H.rep[q] = lane; H.dmax[q] = ck;
        H.e_start[q] = start; H.e_nlog[q] = nl; H.e_log0[q] = log0;
        if (full) atomicAdd(&H.n_full, 1u);
      }
      // exclusive prefix of the counts over the representative lanes
      const uint32_t incl = wave_incl_scan_u32(cnt);
      if (isrep) H.e_off[q] = incl - cnt;
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      const uint32_t nsh = (uint32_t)__popcll(repmask);
      if (lane >= nsh && lane < K7_B) H.e_off[lane] = 0xFFFFFFFFu;
      if (lane == 0) { H.nshapes = nsh; H.n_pairs = total; H.n_batches++; }
#ifdef KB_K7_TRACE
      if (lane == 0) { tacc[2] += total; if (total > KB_K5_THREADS) tacc[12] += (1u << 16); }   // trace only: pairs per batch (slot 2); batches with more pairs than threads (high half of slot 12)
#endif
      K7_STAMP(1);
      // ---- walk (still wave 0, no barrier in between) — unless the previous batch's pre-walk already did it
      if (!pw_use) {
        uint32_t exh = 0, ref = 0;
        k7_walk(M, a.keys, a.L, bd, 0, nb, nb, lane, Hc, Hidx, slid, exh, ref);
        if (lane == 0) { if (exh) H.exhausted = 1; H.n_refills += ref; }
      }
    }
    // can the NEXT batch be pre-walked?  (uniform: every thread computes it)  Its rows must be staged already, and rows whose score is
    // renormalised over the feasible set stop a batch in front of them (has_aff): those sessions keep the plain protocol
    const uint32_t nb2 = min(min(2u * a.batch, K7_B), a.n_rows - (i0 + nb));
    const bool pw = (a.prewalk & 1u) && nb2 > 0 && !a.has_aff && (i0 + nb + nb2 <= dbase + dcnt);
    const KbRowDesc *bdn = bd + nb;
    const uint32_t pw_half = nb2 / 2u;
    pw_slid = 0; pw_exh = 0; pw_ref = 0;
    __syncthreads();
    K7_STAMP(3);
    // ---- fetch + apply: the 16 lanes of one DPP row handle one batch row.  Lane f reads field f of the row's candidate
    //      node (one load), the group votes Allocate / Pipeline (allocate.go:160) with a ballot, every lane applies
    //      NodeInfo.AddTask (api/node_info.go:172-212) to its own field and stores it into the row's NEW dirty slot: the slot
    //      holds the node's state AFTER the row committed.  Scalar dimensions (global memory, rare) are written
    //      speculatively by lane 15 and their old values saved for the rollback.
    if (pw && wave == 0) k7_walk(M, a.keys, a.L, bdn, 0, pw_half, nb2, lane, H.c2[par ^ 1u], H.idx2[par ^ 1u], pw_slid, pw_exh, pw_ref);
    for (uint32_t w = pw ? tid - 64u : tid; w < nb * 16 && !(pw && wave == 0); w += pw ? KB_K5_THREADS - 64u : KB_K5_THREADS) {
      const uint32_t j = w >> 4, f = w & 15;
      const unsigned long long cj = Hc[j];
      uint32_t kind = 0, has_map = 0;
      if (cj) {
        const KbRowDesc &k = bd[j];
        const uint32_t n = KB_KEY_NODE(cj), slot = nd + j;
        unsigned long long v8 = 0ull;
        uint32_t v4 = 0;
        if (f < K5_NF8) v8 = g8[n];
        else if (f <= 12) v4 = g4[n];
        if (f >= K5F_AC && f <= K5F_NZM) v8 = d2u((double)(long long)v8);   // int64 in HBM, double in the slot table
        double res0, res1;
        k7_resreq(a, k, res0, res1);
        bool ok = true;
        if (!a.backfill) {
          const double dv = __longlong_as_double((long long)v8);
          if (f == K5F_IDLE0) ok = le_eps(k.init0, dv, EPS_CPU);
          else if (f == K5F_IDLE1) ok = le_eps(k.init1, dv, EPS_MEM);
          else if (f == 15 && (k.active >> 2)) {
            const KbDev &d = *a.dev;
            uint32_t act = k.active >> 2, dd = 2;
            while (act) {
              if (act & 1u) ok = ok && le_eps(d.t_init[(size_t)dd * d.T + k.task], d.idle[(size_t)dd * d.NP + n], EPS_SCALAR);
              act >>= 1; dd++;
            }
          }
        }
        const unsigned long long bad = __ballot(!ok);
        kind = ((bad >> (lane & 48u)) & 0xFFFFull) ? 1u : 0u;
        const uint32_t f0 = kind ? K5F_REL0 : K5F_IDLE0;
        const double dv = __longlong_as_double((long long)v8);
        if (f == f0) v8 = (unsigned long long)__double_as_longlong(dv - res0);
        else if (f == f0 + 1) v8 = (unsigned long long)__double_as_longlong(dv - res1);
        else if (f == K5F_NZC) v8 = d2u(u2d(v8) + __longlong_as_double(k.nzc));
        else if (f == K5F_NZM) v8 = d2u(u2d(v8) + __longlong_as_double(k.nzm));
        if (f < K5_NF8) M.tab[(size_t)f * cap2 + slot] = v8;
        const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v4, 0x101, 0xf, 0xf, true);   // row_shl:1: lane 11 <- pods
        if (f == 10) M.t_cls[slot] = v4;
        else if (f == 11) M.t_left[slot] = (int)v4 - (int)nxt - 1;
        else if (f == 13) {
          M.t_node[slot] = n;
          if (a.has_ports) M.ptab[slot] = a.dev->ports[n] | a.dev->t_want[k.task];
        }
        else if (f == 15 && k.resmask) {
          const KbDev &d = *a.dev;
          has_map = kind ? (d.nmask[n] >> 31) : (d.nmask[n] & 0x7FFFFFFFu);   // Sub returns early when the receiver's scalar map is nil (resource_info.go:148-153)
          if (has_map) {
            double *vec = kind ? d.rel : d.idle;
            uint32_t dd = 2, m2 = k.resmask;
            while (m2) {
              if (m2 & 1u) {
                const double old = vec[(size_t)dd * d.NP + n];
                M.save[(size_t)j * RS + (dd - 2)] = old;
                vec[(size_t)dd * d.NP + n] = old - d.t_res[(size_t)dd * d.T + k.task];
              }
              m2 >>= 1; dd++;
            }
          }
        }
      }
      if (f == 14) H.kind[j] = kind;
      if (f == 15) H.has_map[j] = has_map;
    }
    __syncthreads();
    K7_STAMP(4);
    // ---- evaluate the work lists: (shape q, slot x) -> dmax[q] for slots older than the batch, kb[row][q] for its own
    {
      const uint32_t P = H.n_pairs;
      // pair e belongs to the shape q with e_off[q] <= e < e_off[q + 1]: lane l of every wave holds e_off[l], the search reads
      // them back as scalars, one compare per distinct shape of the batch (a handful)
      uint32_t voff = H.e_off[lane & (K7_B - 1u)];
      asm volatile("" : "+v"(voff));   // loaded by EVERY lane: the compiler must not sink the load under the `e < P` mask (readlane reads inactive lanes)
      const uint32_t nsh = (uint32_t)__builtin_amdgcn_readfirstlane((int)H.nshapes);
      if (pw && wave == 0) { k7_walk(M, a.keys, a.L, bdn, pw_half, nb2, nb2, lane, H.c2[par ^ 1u], H.idx2[par ^ 1u], pw_slid, pw_exh, pw_ref); }
      for (uint32_t e = pw ? tid - 64u : tid; e < P && !(pw && wave == 0); e += pw ? KB_K5_THREADS - 64u : KB_K5_THREADS) {
        uint32_t q = 0;
        for (uint32_t k = 1; k < nsh; k++) q += (e >= (uint32_t)__builtin_amdgcn_readlane((int)voff, (int)k)) ? 1u : 0u;
        uint32_t r = e - H.e_off[q];
        const uint32_t start = H.e_start[q], nn = nd - start, nl = H.e_nlog[q];
        uint32_t x;
        if (r < nn) x = start + r;
        else if (r < nn + nl) x = M.dlog[H.e_log0[q] + (r - nn)];
        else x = nd + (r - nn - nl);
        unsigned long long key = 0ull;
        if (x < nd || Hc[x - nd] != 0ull) {
          const KbRowDesc &k = bd[H.rep[q]];
          const TaskValsD tv = k7_task_vals(a, k);
          const NodeValsD nv = k5_slot_vals(M.tab, M.t_cls, M.t_left, cap2, x, a.has_ports ? M.ptab : nullptr);
          const uint32_t node = M.t_node[x];
          const uint32_t res = eval_pair_k5(a, tv, nv, node, a.use_crow ? &k.crow : nullptr);
          key = res ? KB_KEY(res & 0xFFFFu, node) : 0ull;
        }
        if (x < nd) { if (key) atomicMax(&H.dmax[q], key); }
        else {
          const uint32_t l = x - nd;
          H.kb[l][q] = key;
          if (key) {   // a later row of the same shape must beat this node in its post-commit state
            uint32_t later = H.rowmask[q] & ~((2u << l) - 1u);
            while (later) {
              const uint32_t j = (uint32_t)__ffs((int)later) - 1u;
              atomicMax(&H.mrow[j], key);
              later &= later - 1u;
            }
          }
        }
      }
    }
    __syncthreads();
    K7_STAMP(5);
    // ---- validate (wave 0)
    if (wave == 0) {
      const bool in = lane < nb;
      const unsigned long long cj = in ? Hc[lane] : 0ull;
      const uint32_t q = in ? H.q_of[lane] : 0u;
      unsigned long long m = 0ull;
      if (in) { const unsigned long long m1 = H.dmax[q], m2 = H.mrow[lane]; m = m1 > m2 ? m1 : m2; }
      if (in && H.rep[q] == lane) {   // the shape's dirty max as of this batch's start becomes its cache
        const uint32_t s = bd[lane].slot;
        M.dc_key[s] = H.dmax[q]; M.dc_nd[s] = nd; M.dc_log[s] = H.nlog;
      }
      const bool valid = in && cj != 0ull && cj > m;
      const unsigned long long inval = __ballot(in && !valid);
      const unsigned long long pipe = __ballot(valid && H.kind[in ? lane : 0] != 0u);
      uint32_t p = inval ? (uint32_t)(__ffsll((unsigned long long)inval) - 1) : nb;
      const unsigned long long pipe_before = pipe & ((1ull << p) - 1ull);
      uint32_t dirty_row = 0, rsn = KB_REASON_DONE;
      unsigned long long kstar = 0ull;
      if (pipe_before) {   // a Pipeline ends the speculation (the host re-plans): commit up to and including that row
        p = (uint32_t)__ffsll((unsigned long long)pipe_before);
        rsn = KB_REASON_PIPELINED;
      } else if (p < nb) {
        kstar = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(m >> 32), (int)p) << 32) |
                (uint32_t)__builtin_amdgcn_readlane((int)(m & 0xFFFFFFFFull), (int)p);
        if (kstar) dirty_row = 1;
        else if (a.backfill) dirty_row = 2;   // backfill.go:50-66: no node passes the predicates -> the task stays Pending
        else rsn = KB_REASON_NO_FEASIBLE;     // allocate.go:144-148: the job is abandoned; the host re-plans from here
      }
      if (lane == 0) {
        H.p = p; H.dirty_row = dirty_row; H.reason = rsn; H.kstar = kstar;
      }
    }
    __syncthreads();
    K7_STAMP(6);
    // ---- commit the valid prefix
    const uint32_t p = H.p, dirty_row = H.dirty_row;
    uint32_t pc = p, rows = p + (dirty_row == 2 ? 1u : 0u);   // candidates consumed, rows consumed
    if (tid < p) {
      const uint32_t j = tid;
      const KbRowDesc &k = bd[j];
      atomicMax(&M.cursor[k.slot], Hidx[j] + 1);
      k7_commit_globals<false>(a, k, i0 + j, KB_KEY_NODE(Hc[j]), H.kind[j]);
    }
    if (dirty_row == 2 && tid == 0) *reinterpret_cast<uint2 *>(&a.dec[i0 + p]) = make_uint2(KB_NONE_U32, 0u);
    K7_STAMP(7);
    if (dirty_row == 1) {
#ifdef KB_K7_TRACE
      tacc[12]++;
#endif
      // ---- a dirty node beats row p's clean candidate.  Dirty winners come in chains (a big node keeps the best score for
      // several tasks), so the rest of row p's run of same-shape rows is committed one row at a time by wave 0 alone, with no
      // workgroup barrier per row: every thread first evaluates the shape against all dirty slots (keyq), then per row
      //     winner = max( max(keyq) , next unconsumed clean candidate of the run )
      // a dirty winner's slot is updated in LDS and its key re-evaluated by one lane; a clean winner takes the slot the fetch /
      // apply steps already prepared for that candidate (same shape => same post-commit state) with the key the evaluate step
      // already computed (kb).
      const uint32_t q = H.q_of[p];
      // Look-ahead (round 3): a dirty winner used to cost wave 0 an AddTask and a whole evaluation on ONE lane (1.14 us, the largest single
      // item of the cycle: profiles/round3).  The all-thread pass below now also evaluates every slot — the batch's own prepared slots
      // included — after ONE and after TWO more placements of the run's shape (Allocate each; K7_LA_INVALID where the node would no longer
      // fit InitResreq, i.e. a Pipeline), so that the serial loop takes a winner's next key from a register and defers the AddTask on
      // the slot's LDS state to a lane-parallel flush.  Plain rows only (Resreq == InitResreq, no scalar dimensions, allocate, single GPU);
      // everything else, and a third win in a row on one slot, takes the evaluation on one lane as before (after a flush).
      const bool la_ok = (a.prewalk & 2u) && !a.backfill && !a.has_delta && (bd[p].flags & 1u) && bd[p].resmask == 0u && (bd[p].active >> 2) == 0u;
      {
        const KbRowDesc &k = bd[p];
        const TaskValsD tv = k7_task_vals(a, k);
        const unsigned long long want = (a.has_ports && la_ok) ? a.dev->t_want[k.task] : 0ull;
        const uint32_t x_end = la_ok ? nd + nb : nd + p;
        // only a slot that can still win needs a look-ahead: it wins at its current key k0 only against a clean candidate below k0, and the
        // run's clean candidates are successive entries of one sorted list — none of them is below the candidate of the LAST batch row
        // with the run's shape.  Most dirty slots lost long ago: their waves skip the two extra evaluations.
        const uint32_t rm_ = H.rowmask[q];
        const unsigned long long cmin = Hc[31u - (uint32_t)__clz((int)(rm_ | 1u))];
        for (uint32_t x = tid; x < x_end; x += KB_K5_THREADS) {
          unsigned long long k0 = 0ull, k1 = K7_LA_INVALID, k2 = K7_LA_INVALID;
          if (x < nd || Hc[x - nd] != 0ull) {
            NodeValsD nv = k5_slot_vals(M.tab, M.t_cls, M.t_left, cap2, x, a.has_ports ? M.ptab : nullptr);
            const uint32_t node = M.t_node[x];
            const uint32_t res = eval_pair_k5(a, tv, nv, node, a.use_crow ? &k.crow : nullptr);
            k0 = res ? KB_KEY(res & 0xFFFFu, node) : 0ull;
            if (la_ok && k0 != 0ull && k0 >= cmin) {
              int left = M.t_left[x];
              if (le_eps(k.init0, nv.idle0, EPS_CPU) && le_eps(k.init1, nv.idle1, EPS_MEM)) {   // allocate.go:160: Allocate
                nv.idle0 -= k.init0; nv.idle1 -= k.init1; nv.nzc += tv.nzc; nv.nzm += tv.nzm; nv.ports |= want; left -= 1; nv.slots = left > 0;
                const uint32_t r1 = eval_pair_k5(a, tv, nv, node, a.use_crow ? &k.crow : nullptr);
                k1 = r1 ? KB_KEY(r1 & 0xFFFFu, node) : 0ull;
                if (le_eps(k.init0, nv.idle0, EPS_CPU) && le_eps(k.init1, nv.idle1, EPS_MEM)) {
                  nv.idle0 -= k.init0; nv.idle1 -= k.init1; nv.nzc += tv.nzc; nv.nzm += tv.nzm; left -= 1; nv.slots = left > 0;
                  const uint32_t r2 = eval_pair_k5(a, tv, nv, node, a.use_crow ? &k.crow : nullptr);
                  k2 = r2 ? KB_KEY(r2 & 0xFFFFu, node) : 0ull;
                }
              }
            }
          }
          M.keyq[x] = k0;
          if (la_ok) { M.keyq1[x] = k1; M.keyq2[x] = k2; M.wcnt[x] = 0u; }
        }
      }
      __syncthreads();
      K7_STAMP(8);
      if (wave == 0) {
        const uint32_t myq = (lane < nb) ? H.q_of[lane] : 0xFFFFFFFFu;
        const unsigned long long diff = __ballot(lane > p && lane < nb && myq != q);
        const uint32_t run_end = diff ? (uint32_t)(__ffsll((unsigned long long)diff) - 1) : nb;
        const uint32_t shape = bd[p].slot;
        uint32_t r = p, ndc = nd + p, rsn = KB_REASON_DONE, last_n = 0xFFFFFFFFu, nlog = H.nlog;
        // the keys of the shape against every dirty slot live in registers for the whole run: lane l holds slots l, l + 64, ...
        // (K7_KQ of them cover cap + K7_B slots), and `best`, their maximum, is a wave-uniform scalar: a clean winner updates it
        // with one max (its post-placement key is ready: kb), only a dirty winner costs an evaluation and a new wave maximum
        unsigned long long kq[K7_KQ];
#pragma unroll
        for (int u = 0; u < K7_KQ; u++) { const uint32_t x = lane + 64u * (uint32_t)u; kq[u] = (x < ndc) ? M.keyq[x] : 0ull; }
        // look-ahead: keyq1 / keyq2 / wcnt stay in LDS (keeping them per lane in registers spilled the kernel to scratch); a dirty winner costs
        // three independent LDS reads on one lane.  npend (uniform): slots in wlist that carry deferred placements
        uint32_t npend = 0, chain_slot = 0xFFFFFFFFu;
        // (the run's task values are re-read from the staged descriptor inside the two helpers: held across the serial loop they pushed
        // the kernel over its 128-VGPR budget and into scratch)
        // apply the deferred placements to the slots' LDS state (one lane per slot, in parallel), drop the look-ahead of those slots, and
        // invalidate the cached dirty maxima of other shapes that sat on one of those nodes
        auto flush = [&]() {
          uint32_t mynode = 0xFFFFFFFFu;
          if (lane < npend) {
            const TaskValsD tvr = k7_task_vals(a, bd[p]);
            const unsigned long long wantr = a.has_ports ? a.dev->t_want[bd[p].task] : 0ull;
            const uint32_t x = M.wlist[lane], w = M.wcnt[x];
            double i0v = u2d(M.tab[(size_t)K5F_IDLE0 * cap2 + x]), i1v = u2d(M.tab[(size_t)K5F_IDLE1 * cap2 + x]);
            double zc = u2d(M.tab[(size_t)K5F_NZC * cap2 + x]), zm = u2d(M.tab[(size_t)K5F_NZM * cap2 + x]);
            for (uint32_t i = 0; i < w; i++) { i0v -= tvr.init0; i1v -= tvr.init1; zc += tvr.nzc; zm += tvr.nzm; }   // NodeInfo.AddTask, one task at a time
            M.tab[(size_t)K5F_IDLE0 * cap2 + x] = d2u(i0v); M.tab[(size_t)K5F_IDLE1 * cap2 + x] = d2u(i1v);
            M.tab[(size_t)K5F_NZC * cap2 + x] = d2u(zc); M.tab[(size_t)K5F_NZM * cap2 + x] = d2u(zm);
            M.t_left[x] -= (int)w;
            if (a.has_ports) M.ptab[x] |= wantr;
            M.wcnt[x] = 0u; M.keyq1[x] = K7_LA_INVALID; M.keyq2[x] = K7_LA_INVALID;
            mynode = M.t_node[x];
          }
          for (uint32_t sh = lane; sh < a.n_mrows; sh += 64) {
            const unsigned long long ck = M.dc_key[sh];
            const uint32_t cn = ck != 0ull ? KB_KEY_NODE(ck) : 0xFFFFFFFEu;
            bool hit = false;
            for (uint32_t i = 0; i < npend; i++) hit = hit || ((uint32_t)__builtin_amdgcn_readlane((int)mynode, (int)i) == cn);
            if (hit) M.dc_nd[sh] = 0xFFFFFFFFu;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          npend = 0;
          chain_slot = 0xFFFFFFFFu;
        };
        // a slot whose look-ahead has run out (a big node that keeps winning): lane t evaluates it after t + 1 more placements of the
        // run's shape, one at a time (NodeInfo.AddTask's own subtractions, Allocate only), in ONE pass — the next K7_CH wins on that
        // slot read their key from this table.  Called right after a flush: the slot's LDS state is current.
        auto chain_build = [&](uint32_t xs) {
          const TaskValsD tvr = k7_task_vals(a, bd[p]);
          const unsigned long long wantr = a.has_ports ? a.dev->t_want[bd[p].task] : 0ull;
          NodeValsD nv = k5_slot_vals(M.tab, M.t_cls, M.t_left, cap2, xs, a.has_ports ? M.ptab : nullptr);
          int left = M.t_left[xs];
          bool valid = true;
          const uint32_t steps = min(lane, K7_CH - 1u) + 1u;
          for (uint32_t st = 0; st < steps; st++) {
            valid = valid && le_eps(tvr.init0, nv.idle0, EPS_CPU) && le_eps(tvr.init1, nv.idle1, EPS_MEM);   // allocate.go:160
            nv.idle0 -= tvr.init0; nv.idle1 -= tvr.init1; nv.nzc += tvr.nzc; nv.nzm += tvr.nzm; left -= 1;
          }
          nv.ports |= wantr; nv.slots = left > 0;
          const uint32_t node = M.t_node[xs];
          const uint32_t res = eval_pair_k5(a, tvr, nv, node, a.use_crow ? &bd[p].crow : nullptr);
          if (lane < K7_CH) M.chain[lane] = valid ? (res ? KB_KEY(res & 0xFFFFu, node) : 0ull) : K7_LA_INVALID;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          chain_slot = xs;
        };
        // what the rows of the run need to know about the batch's prepared candidates, one per lane, read back with readlane
        const bool inb = lane < nb;
        unsigned long long myc = inb ? Hc[lane] : 0ull, mykb = inb ? H.kb[lane][q] : 0ull;
        uint32_t mykind = inb ? H.kind[lane] : 0u, myidx = inb ? Hidx[lane] : 0u;
        asm volatile("" : "+v"(myc), "+v"(mykb), "+v"(mykind), "+v"(myidx));   // loaded by every lane, here (readlane reads inactive lanes)
        const unsigned long long plainmask = __ballot(inb && (bd[inb ? lane : 0].flags & 1) && bd[inb ? lane : 0].resmask == 0);
        uint32_t dec_n = 0, dec_k = 0, dec_f = 0;   // lane j: the decision of row p + j when a clean candidate took it
        unsigned long long best;
        {
          unsigned long long kmax = kq[0];
#pragma unroll
          for (int u = 1; u < K7_KQ; u++) if (kq[u] > kmax) kmax = kq[u];
          best = wave_max_key(kmax);
        }
        const uint32_t pc0 = pc;
        while (r < run_end) {
          const unsigned long long cc = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(myc >> 32), (int)pc) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)(myc & 0xFFFFFFFFull), (int)pc);
#ifdef KB_K7_TRACE
          unsigned long long tr0 = 0;
          if (tid == 0) tr0 = wall_clock64();
#endif
          if (best == 0ull && cc == 0ull) {
            if (a.backfill) {   // backfill.go:50-66: the task stays Pending
              if (lane == 0) *reinterpret_cast<uint2 *>(&a.dec[i0 + r]) = make_uint2(KB_NONE_U32, 0u);
              r++;
              continue;
            }
            rsn = KB_REASON_NO_FEASIBLE;   // allocate.go:144-148
            break;
          }
          if (best > cc) {
            const KbRowDesc &k = bd[r];
            unsigned long long kmax = kq[0];
            uint32_t xmax = lane;
#pragma unroll
            for (int u = 1; u < K7_KQ; u++) if (kq[u] > kmax) { kmax = kq[u]; xmax = lane + 64u * (uint32_t)u; }
            const unsigned long long own = __ballot(kmax == best);
            const uint32_t ow = (uint32_t)__ffsll((unsigned long long)own) - 1u;
            const uint32_t xs = (uint32_t)__builtin_amdgcn_readlane((int)xmax, (int)ow);
            const uint32_t n = KB_KEY_NODE(best);
            if (la_ok && ((plainmask >> r) & 1ull)) {
              // the winner's key after this placement, from the look-ahead: level = placements already deferred on the slot
              unsigned long long nk = K7_LA_INVALID;
              uint32_t lvl = 0;
              if (lane == 0) {
                lvl = M.wcnt[xs];
                const unsigned long long c1 = M.keyq1[xs], c2 = M.keyq2[xs], cc2 = M.chain[lvl < K7_CH ? lvl : 0u];
                nk = (xs == chain_slot) ? (lvl < K7_CH ? cc2 : K7_LA_INVALID) : (lvl == 0u ? c1 : (lvl == 1u ? c2 : K7_LA_INVALID));
              }
              nk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(nk >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(nk & 0xFFFFFFFFull));
              lvl = (uint32_t)__builtin_amdgcn_readfirstlane((int)lvl);
              if (nk == K7_LA_INVALID && (a.prewalk & 4u)) {   // no look-ahead left for this slot (it keeps winning, or its state moved on): apply what is deferred, build its chain
                if (npend) flush();
                chain_build(xs);
                lvl = 0u;
                nk = M.chain[0];
                nk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(nk >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(nk & 0xFFFFFFFFull));
              }
              if (nk != K7_LA_INVALID) {
                if (lane == r - p) { dec_n = n; dec_k = 0u; dec_f = 1u; }
                if (lane == 0) {
                  M.wcnt[xs] = lvl + 1u;
                  if (lvl == 0u) M.wlist[npend] = xs;
                  M.dlog[nlog] = xs;
                }
                if (lvl == 0u) npend++;
                if (lane == ow) {
#pragma unroll
                  for (int u = 0; u < K7_KQ; u++) if ((xs >> 6) == (uint32_t)u) kq[u] = nk;
                }
                {
                  unsigned long long km2 = kq[0];
#pragma unroll
                  for (int u = 1; u < K7_KQ; u++) if (kq[u] > km2) km2 = kq[u];
                  best = wave_max_key(km2);
                }
                nlog++;
                r++;
#ifdef KB_K7_TRACE
                if (tid == 0) tacc[11] += (uint32_t)(wall_clock64() - tr0);
#endif
                continue;
              }
            }
            if (npend) flush();   // the evaluation below reads the slot's LDS state
            // shapes whose cached dirty max sits on the node that changes lose their cache (once per node of a chain)
            if (n != last_n) {
              for (uint32_t sh = lane; sh < a.n_mrows; sh += 64) {
                const unsigned long long ck = M.dc_key[sh];
                if (ck != 0ull && KB_KEY_NODE(ck) == n) M.dc_nd[sh] = 0xFFFFFFFFu;
              }
              last_n = n;
            }
            uint32_t kind = 0;
            unsigned long long nkey = 0ull;
            if (lane == 0) {
              // one batch of LDS loads, NodeInfo.AddTask (api/node_info.go:172-212) in registers, the changed fields stored back,
              // and the node's new key evaluated from the registers (no second trip through LDS)
              NodeValsD nv = k5_slot_vals(M.tab, M.t_cls, M.t_left, cap2, xs, a.has_ports ? M.ptab : nullptr);
              const int left = M.t_left[xs] - 1;
              double res0, res1;
              k7_resreq(a, k, res0, res1);
              if (!a.backfill) {
                bool fi = le_eps(k.init0, nv.idle0, EPS_CPU) && le_eps(k.init1, nv.idle1, EPS_MEM);
                uint32_t act = k.active >> 2;
                if (act) {
                  const KbDev &d = *a.dev;
                  uint32_t dd = 2;
                  while (act) {
                    if (act & 1u) fi = fi && le_eps(d.t_init[(size_t)dd * d.T + k.task], d.idle[(size_t)dd * d.NP + n], EPS_SCALAR);
                    act >>= 1; dd++;
                  }
                }
                kind = fi ? 0u : 1u;
              }
              if (kind) {
                nv.rel0 -= res0; nv.rel1 -= res1;
                M.tab[(size_t)K5F_REL0 * cap2 + xs] = d2u(nv.rel0);
                M.tab[(size_t)K5F_REL1 * cap2 + xs] = d2u(nv.rel1);
              } else {
                nv.idle0 -= res0; nv.idle1 -= res1;
                M.tab[(size_t)K5F_IDLE0 * cap2 + xs] = d2u(nv.idle0);
                M.tab[(size_t)K5F_IDLE1 * cap2 + xs] = d2u(nv.idle1);
              }
              nv.nzc += __longlong_as_double(k.nzc); nv.nzm += __longlong_as_double(k.nzm);
              M.tab[(size_t)K5F_NZC * cap2 + xs] = d2u(nv.nzc);
              M.tab[(size_t)K5F_NZM * cap2 + xs] = d2u(nv.nzm);
              if (a.has_ports) { nv.ports |= a.dev->t_want[k.task]; M.ptab[xs] = nv.ports; }   // the pod's host ports join nodeinfo.UsedPorts()
              M.t_left[xs] = left;
              nv.slots = left > 0;
              k7_commit_globals<true>(a, k, i0 + r, n, kind);
              M.dlog[nlog] = xs;
              const TaskValsD tv = k7_task_vals(a, k);
              const uint32_t res = eval_pair_k5(a, tv, nv, n, a.use_crow ? &k.crow : nullptr);
              nkey = res ? KB_KEY(res & 0xFFFFu, n) : 0ull;
            }
            if (lane == 0 && la_ok) { M.keyq1[xs] = K7_LA_INVALID; M.keyq2[xs] = K7_LA_INVALID; }   // the slot's state moved on: its look-ahead is stale
            kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)kind);
            nkey = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(nkey >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(nkey & 0xFFFFFFFFull));
            if (lane == (xs & 63u)) {
#pragma unroll
              for (int u = 0; u < K7_KQ; u++) if ((xs >> 6) == (uint32_t)u) kq[u] = nkey;
            }
            {   // the changed key may have been the maximum: a new wave maximum
              unsigned long long km2 = kq[0];
#pragma unroll
              for (int u = 1; u < K7_KQ; u++) if (kq[u] > km2) km2 = kq[u];
              best = wave_max_key(km2);
            }
            nlog++;
            r++;
#ifdef KB_K7_TRACE
            if (tid == 0) tacc[11] += (uint32_t)(wall_clock64() - tr0);
#endif
            if (kind) { rsn = KB_REASON_PIPELINED; break; }
          } else {
            // the prepared slot nd+pc holds candidate pc's node after ROW pc's task; identical for row r's task when both
            // rows carry plain requests (same shape; no init-container maximum, no scalar resources)
            if (!((plainmask >> r) & (plainmask >> pc) & 1ull)) break;
            const uint32_t kind = (uint32_t)__builtin_amdgcn_readlane((int)mykind, (int)pc);
            const unsigned long long ckey = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(mykb >> 32), (int)pc) << 32) |
                                            (uint32_t)__builtin_amdgcn_readlane((int)(mykb & 0xFFFFFFFFull), (int)pc);
            if (a.has_delta) {   // multi-GPU rounds: the per-node deltas are accumulated row by row
              if (lane == 0) k7_commit_globals<false>(a, bd[r], i0 + r, KB_KEY_NODE(cc), kind);
            } else if (lane == r - p) {
              dec_n = KB_KEY_NODE(cc); dec_k = kind; dec_f = 1u;
            }
            if (lane == (ndc & 63u)) {
#pragma unroll
              for (int u = 0; u < K7_KQ; u++) if ((ndc >> 6) == (uint32_t)u) kq[u] = ckey;
            }
            if (ckey > best) best = ckey;
            ndc++; pc++; r++;
            if (kind) { rsn = KB_REASON_PIPELINED; break; }
          }
        }
        if (npend) flush();
        // the clean winners of the run: their cursor and their decision records
        const uint32_t last_idx = (uint32_t)__builtin_amdgcn_readlane((int)myidx, (int)(pc > pc0 ? pc - 1u : 0u));
        if (pc > pc0 && lane == 0) atomicMax(&M.cursor[shape], last_idx + 1u);
        if (dec_f) *reinterpret_cast<uint2 *>(&a.dec[i0 + p + lane]) = make_uint2(dec_n, dec_k);
        // the shape's dirty max as of now becomes its cache
        const unsigned long long kmax = best;
        if (lane == 0) {
          M.dc_key[shape] = kmax; M.dc_nd[shape] = ndc; M.dc_log[shape] = nlog;
          H.n_dirty_rows += nlog - H.nlog;
          H.nlog = nlog;
          H.seq_rows = r; H.seq_pc = pc; H.reason = rsn; H.n_seq_rows += r - p;
        }
      }
      __syncthreads();
      K7_STAMP(9);
      rows = H.seq_rows;
      pc = H.seq_pc;
    }
    // ---- roll back the candidates nobody consumed: they are re-speculated by the next batch
    if (tid < nb && tid >= pc) {
      const uint32_t j = tid;
      const unsigned long long cj = Hc[j];
      if (cj) {
        const KbRowDesc &k = bd[j];
        const uint32_t n = KB_KEY_NODE(cj);
        atomicAnd(&M.bitmap[n >> 5], ~(1u << (n & 31)));
        if (H.has_map[j]) {
          const KbDev &d = *a.dev;
          double *vec = H.kind[j] ? d.rel : d.idle;
          uint32_t dd = 2, m2 = k.resmask;
          while (m2) {
            if (m2 & 1u) vec[(size_t)dd * d.NP + n] = M.save[(size_t)j * RS + (dd - 2)];
            m2 >>= 1; dd++;
          }
        }
      }
    }
    // the pre-walk stands iff this batch consumed every candidate and the next one starts right behind it with the size assumed
    const bool fully = dirty_row == 0 && p == nb && H.reason == KB_REASON_DONE;
    pw_use = (pw && fully) ? 1u : 0u;
    if (pw && !fully && tid < nb2) {   // undo it: its candidates are unconsumed (wave 0: nb2 <= 32 threads)
      const unsigned long long cj = H.c2[par ^ 1u][tid];
      if (cj) { const uint32_t n = KB_KEY_NODE(cj); atomicAnd(&M.bitmap[n >> 5], ~(1u << (n & 31))); }
    }
    if (wave == 0) {
      // a window that slid during this batch's walk (or during a pre-walk that is being undone) may have skipped entries whose bits the
      // roll-back has just cleared: re-anchor it at the shape's cursor (every entry in front of the cursor belongs to a committed row or
      // was dirty when that row passed)
      uint32_t sm = (pc < nb) ? slid : 0u;
      while (sm) {
        const uint32_t j0 = (uint32_t)__ffs((int)sm) - 1u;
        sm &= sm - 1u;
        const uint32_t sr = bd[j0].slot;
        const uint32_t nbase = M.cursor[sr];
        if (lane < WL) M.pwin[(size_t)sr * WL + lane] = (nbase + lane < a.L) ? a.keys[(size_t)sr * a.L + nbase + lane] : 0ull;
        if (lane == 0) M.pbase[sr] = nbase;
      }
      sm = (pw && !fully) ? pw_slid : 0u;
      while (sm) {
        const uint32_t j0 = (uint32_t)__ffs((int)sm) - 1u;
        sm &= sm - 1u;
        const uint32_t sr = bdn[j0].slot;
        const uint32_t nbase = M.cursor[sr];
        if (lane < WL) M.pwin[(size_t)sr * WL + lane] = (nbase + lane < a.L) ? a.keys[(size_t)sr * a.L + nbase + lane] : 0ull;
        if (lane == 0) M.pbase[sr] = nbase;
      }
      if (lane == 0 && pw) { H.n_prewalks++; if (fully) { H.n_prewalks_used++; H.n_refills += pw_ref; if (pw_exh) H.exhausted = 1; } }
    }
    if (pw_use) carried_slid = pw_slid;
    par ^= 1u;
    // no barrier here: the commit of the prefix and the roll-back above are wave 0's work (tid < K7_B <= 64) and so are the next
    // batch's shapes and walk steps; the other waves meet wave 0 again at the barrier in front of the fetch step
    K7_STAMP(10);
    nd += pc;
    i0 += rows;
    // clean streaks are long (56 % of the batches commit every row): speculate twice as many rows after a fully valid batch
    nb_cur = (dirty_row == 0 && p == nb) ? min(2u * a.batch, K7_B) : a.batch;
    n_done = i0;
    reason = H.reason;
    if (H.exhausted) reason = KB_REASON_INTERNAL;
    if (reason != KB_REASON_DONE) break;
  }

  // ---- write the dirty nodes' live state back to HBM
  __syncthreads();
  {
    const KbDev &d = *a.dev;
    for (uint32_t slot = tid; slot < nd; slot += KB_K5_THREADS) {
      const uint32_t n = M.t_node[slot];
      d.idle[n] = __longlong_as_double((long long)M.tab[K5F_IDLE0 * cap2 + slot]);
      d.idle[(size_t)d.NP + n] = __longlong_as_double((long long)M.tab[K5F_IDLE1 * cap2 + slot]);
      d.rel[n] = __longlong_as_double((long long)M.tab[K5F_REL0 * cap2 + slot]);
      d.rel[(size_t)d.NP + n] = __longlong_as_double((long long)M.tab[K5F_REL1 * cap2 + slot]);
      d.nzc[n] = (long long)u2d(M.tab[K5F_NZC * cap2 + slot]);
      d.nzm[n] = (long long)u2d(M.tab[K5F_NZM * cap2 + slot]);
      d.podcnt[n] = d.maxpods[n] - M.t_left[slot];
      if (a.has_ports) d.ports[n] = M.ptab[slot];
    }
  }
  // task-table side of ssn.Allocate / ssn.Pipeline for the committed rows (job.UpdateTaskStatus, task.NodeName:
  // framework/session.go:243,205; api/node_info.go:206-209)
  {
    const KbDev &d = *a.dev;
    for (uint32_t i = tid; i < n_done; i += KB_K5_THREADS) {
      const uint2 dc = *reinterpret_cast<const uint2 *>(&a.dec[i]);
      if (dc.x == KB_NONE_U32) continue;
      const uint32_t t = a.desc[i].task;
      d.t_status[t] = dc.y ? KB_TASK_PIPELINED : KB_TASK_ALLOCATED;
      d.t_node[t] = dc.x;
      d.t_counted[t] = 1;
      if (!dc.y) d.j_allocated[d.t_job[t]] = 1;   // ssn.Allocate ran for the job: its Allocated tasks are dispatched if it is ready
      if (d.t_ip_cls_inc) {   // inter-pod affinity: the pod joins ni.Tasks of its node and, when Allocated, the PodLister's allocated set
        for (uint32_t w = 0; w < d.ip_Wp; w++) {
          unsigned long long cm = d.t_ip_cls_inc[(size_t)t * d.ip_Wp + w];
          while (cm) {
            const uint32_t pcl = 64u * w + (uint32_t)__ffsll((unsigned long long)cm) - 1u;
            cm &= cm - 1ull;
            atomicAdd(&d.ip_cls_unbound[(size_t)pcl * d.NP + dc.x], 1);
          }
        }
        atomicMin(d.ip_z, dc.x);
        if (!dc.y)
          for (uint32_t w = 0; w < d.ip_Wc; w++) {
            unsigned long long im = d.t_ip_inc[(size_t)t * d.ip_Wc + w];
            while (im) {
              const uint32_t c = 64u * w + (uint32_t)__ffsll((unsigned long long)im) - 1u;
              im &= im - 1ull;
              atomicAdd(&d.ip_ctr_total[c], 1);
              const uint32_t dm = d.ip_ctr_dom[(size_t)c * d.NP + dc.x];
              if (dm != KB_NONE_U32) atomicAdd(&d.ip_ctr_count[(size_t)c * d.ip_D + dm], 1);
            }
          }
      }
    }
  }
  if (tid == 0) {
    a.result[0] = n_done; a.result[1] = reason; a.result[2] = nd; a.result[4] = H.n_refills; a.result[7] = H.n_full; a.result[3] = H.n_seq_rows;
    a.result[5] = H.n_batches; a.result[6] = H.n_dirty_rows;
#ifdef KB_K7_TRACE
    unsigned long long *tw = reinterpret_cast<unsigned long long *>(a.result);
    for (int k = 0; k < 4; k++) tw[4 + k] = (unsigned long long)tacc[2 * k] | ((unsigned long long)tacc[2 * k + 1] << 32);
    for (int k = 4; k < 7; k++) tw[9 + k] = (unsigned long long)tacc[2 * k] | ((unsigned long long)tacc[2 * k + 1] << 32);   // words 13..15
#endif
    if (a.round->chain) *a.round->chain = reason == KB_REASON_DONE ? a.round->chain_tag : 0u;   // the round queued behind this one runs only then
    unsigned long long *st = reinterpret_cast<unsigned long long *>(a.result) + KB_OUT_STAMP0;
    st[2] = t_start;
    st[3] = wall_clock64();
  }
  // ---- fast rounds: mirror the header and the decision records into pinned host memory and publish the round's sequence
  //      number last; the host spins on that word instead of paying a stream synchronisation + D2H copy per round
  if (a.host_out) {
    __syncthreads();
    const unsigned long long *hdr = reinterpret_cast<const unsigned long long *>(a.result);
    for (uint32_t i = tid; i < KB_OUT_HDR; i += KB_K5_THREADS) if (i != KB_OUT_SEQ) a.host_out[i] = hdr[i];
    for (uint32_t i = tid; i < n_done; i += KB_K5_THREADS) a.host_out[KB_OUT_HDR + i] = a.dec[i];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&a.host_out[KB_OUT_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


size_t kb_commit_batch_smem_bytes(uint32_t cap, uint32_t NP, int R) { return k7_smem_bytes(cap, NP, R); }
void kb_launch_commit_batch(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  static bool attr_set = false;
  static uint32_t env_batch = 0;
  static uint32_t env_prewalk = 7u;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_commit_batch), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const char *b = getenv("KB_K5_BATCH");   // tuning override of kb_config.commit_batch
    env_batch = b ? (uint32_t)atoi(b) : 0;
    const char *pwe = getenv("KB_K7_PREWALK"), *lae = getenv("KB_K7_LOOKAHEAD");   // A/B switches
    const char *che = getenv("KB_K7_CHAIN");
    env_prewalk = ((pwe && pwe[0] == '0') ? 0u : 1u) | ((lae && lae[0] == '0') ? 0u : 2u) | ((che && che[0] == '0') ? 0u : 4u);
    attr_set = true;
  }
  uint32_t batch = env_batch ? env_batch : (r.batch ? r.batch : K7_B_DEFAULT);
  if (batch > K7_B) batch = K7_B;
  const size_t sh = k7_smem_bytes(r.cap, d.NP, d.R);
  K7KernArgs ka;
  ka.dev = d;
  ka.round = r;
  KbCommitArgs &a = ka.hot;
  a.dev = nullptr; a.round = nullptr;   // set from the kernel-argument segment inside the kernel
  a.keys = r.keys; a.dec = r.dec; a.desc = r.desc; a.result = r.result; a.trace = nullptr;
  a.n_rows = r.n_rows; a.n_mrows = r.n_mrows; a.L = r.L; a.cap = r.cap; a.N = d.N; a.NP = d.NP;
  a.fit_mode = r.fit_mode; a.backfill = r.backfill; a.pred_enabled = d.pred_enabled; a.score_enabled = d.score_enabled;
  a.wL = d.wL; a.wM = d.wM; a.wB = d.wB;
  a.use_crow = (d.pred_enabled && d.crows != nullptr && d.n_nc <= 32) ? 1u : 0u;
  a.has_delta = r.delta != nullptr ? 1u : 0u;
  a.has_aff = ((d.aff != nullptr && d.score_enabled) || d.t_ip_subject != nullptr) ? 1u : 0u;
  a.has_ports = d.ports != nullptr ? 1u : 0u;
  a.R = d.R;
  a.batch = batch;
  a.T = d.T; a.node_bits = 0;
  a.prewalk = env_prewalk;
  a.host_out = r.host_out;
  a.seq = r.seq;
  static const bool helpers_off = getenv("KB_WARM_HELPERS_OFF") && getenv("KB_WARM_HELPERS_OFF")[0] == '1';   // A/B switch
  hipLaunchKernelGGL(k_commit_batch, dim3(helpers_off ? 1u : KB_WARM_GRID), dim3(KB_K5_THREADS), sh, (hipStream_t)stream, ka);
}
