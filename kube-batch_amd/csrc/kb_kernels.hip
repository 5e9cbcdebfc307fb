// kb_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for kube-batch's allocate/backfill hot path.
//
//   K1 k_matrix    mask + score matrix for a window of task rows x all nodes
//                  = allocate.go:73-87 predicate closure + plugins/predicates/predicates.go:123-265 (pod-count cap,
//                    static checks as a class bit table) + nodeorder's LeastRequested / MostRequested /
//                    BalancedResourceAllocation (vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/*.go)
//                    summed as util.PrioritizeNodes does (scheduler_helper.go:162-168).
//   K3 k_argmax    per matrix row, the first K entries of (score descending, node ascending) = util.SelectBestNode
//                  (scheduler_helper.go:188-208, first max in ascending node order) generalised to a sorted candidate list
//                  (also the order util.SortNodes gives preempt, scheduler_helper.go:174-185).
//   K5 k_commit    (kb_commit.hip) the sequential part of allocate.go:129-193 / backfill.go:44-67.
//   K2+K4 k_finalize   gang ready count by wavefront ballot (api/job_info.go:383-394, gang.go:122-125), gang-gated bind
//                  set (framework/session.go:277-285), drf / proportion share reduction (drf.go:157-171,
//                  proportion.go:241-253, api/helpers/helpers.go:47-60).
//
// Exactness: float64 compares with the reference's epsilons, IEEE double division (no fast-math, -ffp-contract=off),
// int64 truncating division reproduced exactly through a reciprocal estimate + integer remainder fix-up.
// The path is elementwise compare + integer scoring, HBM/latency bound: no MFMA.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "kb_device.h"
#include "kb_eval.hpp"
#include "kb_k1.hpp"
#include "kb_repair.hpp"
#include "kb_warm.hpp"

// ------------------------------------------------------------------------------------------------------------
// K1: mask + score matrix.  grid (NP / (256*NPT), ceil(n_rows/TR)); thread <-> NPT consecutive nodes kept in
// registers for all TR rows of the tile; the tile's task vectors are staged once in LDS.
// Stores: one 8-byte score vector per thread per row (512 B contiguous per wave), one 4-byte mask word per 8 lanes.
// ------------------------------------------------------------------------------------------------------------
// Two tile shapes: <4 nodes/thread, 32 rows/block> for matrix-sized launches (8-byte score stores, node state amortised
// over 32 rows) and <1 node/thread, 4 rows/block> for the small per-round launches (a few dozen distinct shapes), where
// the work has to be spread over all 256 CUs instead of being serialised inside few threads.
__device__ __forceinline__ void gather_row(const KbDev &d, const KbRound &r, uint32_t i) {
  if (i >= r.n_rows) return;
  const uint32_t t = r.rows[i];
  KbRowDesc k;
  k.init0 = d.t_init[t]; k.init1 = d.t_init[(size_t)d.T + t];
  k.nzc = d.t_nzc[t]; k.nzm = d.t_nzm[t];
  k.task = t; k.active = d.t_active[t]; k.resmask = d.t_resmask[t]; k.cls = d.t_cls[t];
  k.slot = (uint16_t)r.shape_slot[i];
  k.flags = (d.t_res[t] == k.init0 && d.t_res[(size_t)d.T + t] == k.init1) ? 1 : 0;
  if (d.aff_cls && d.aff_cls[k.cls]) k.flags |= 2;
  if (d.t_ip_subject && d.t_ip_subject[t]) k.flags |= 2;   // inter-pod subject: exact only against a fresh matrix, like a renormalised row
  {   // bit 2: in every scalar dimension Resreq names, Resreq == InitResreq (the commit kernel then takes the shape's value)
    bool same = true;
    uint32_t m2 = k.resmask;
    for (int dd = 2; m2; dd++, m2 >>= 1)
      if ((m2 & 1u) && d.t_res[(size_t)dd * d.T + t] != d.t_init[(size_t)dd * d.T + t]) same = false;
    if (same) k.flags |= 4;
  }
  k.crow = d.crows ? d.crows[(size_t)k.cls * 8] : 0xFFFFFFFFu;
  r.desc[i] = k;
}

template <int NPT, int TR>
__global__ void __launch_bounds__(256) k_matrix(KbDev d, KbRound r) {
  __shared__ K1Task srow[TR];
  __shared__ uint8_t ssame[TR];
  if (KB_CHAIN_BROKEN(r)) return;
  if (r.gather && blockIdx.y == gridDim.y - 1) {   // the extra block row of a single-GPU round: the window's row descriptors
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long *>(r.result)[KB_OUT_STAMP0] = wall_clock64();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < r.n_rows; i += gridDim.x * 256) gather_row(d, r, i);
    return;
  }
  const uint32_t row0 = blockIdx.y * TR;
  const uint32_t nr = min((uint32_t)TR, r.n_mrows - row0);
  if (threadIdx.x < nr) {
    uint32_t i = row0 + threadIdx.x;
    uint32_t t = r.mrows ? r.mrows[i] : r.mrow_task0 + i;
    srow[threadIdx.x] = k1_task(d, t);
    ssame[threadIdx.x] = r.same_prev ? r.same_prev[i] : 0;
  }
  __syncthreads();
  const uint32_t n0 = (blockIdx.x * 256 + threadIdx.x) * NPT;
  const uint32_t lane = threadIdx.x & 63;
  K1Node nv[NPT];
#pragma unroll
  for (int j = 0; j < NPT; j++) nv[j] = k1_node(d, n0 + j);
  uint32_t res[NPT];
#pragma unroll
  for (int j = 0; j < NPT; j++) res[j] = 0;
  const size_t mstride = d.NP / 32;
  uint2 pk = make_uint2(0u, 0u);   // packed scores / mask word of the last evaluated row (re-stored for identical rows)
  uint32_t mw = 0;
  for (uint32_t rr = 0; rr < nr; rr++) {
    const bool fresh = !(k1_uniform((uint32_t)ssame[rr]) && rr > 0);
    if (fresh) {
      const K1Task tv = k1_uniform(srow[rr]);
      eval_row<NPT>(d, tv, nv, n0, r.fit_mode, res);
    }
    const size_t row = row0 + rr;
    if (NPT == 4) {
      if (fresh) {
        pk.x = (res[0] & 0xFFFFu) | (res[1 % NPT] << 16);
        pk.y = (res[2 % NPT] & 0xFFFFu) | (res[3 % NPT] << 16);
        uint32_t nib = ((res[0] >> 16) & 1u) | (((res[1 % NPT] >> 16) & 1u) << 1) | (((res[2 % NPT] >> 16) & 1u) << 2) | (((res[3 % NPT] >> 16) & 1u) << 3);
        // OR the 8 lanes' nibbles into one mask word with DPP moves (no LDS crossbar): xor 1, xor 2, mirror within 8
        uint32_t w = nib << (4 * (lane & 7));
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x141, 0xf, 0xf, false);   // row_half_mirror
        mw = w;
      }
      // the matrix is written once and read by another kernel
      *reinterpret_cast<uint2 *>(r.score + row * d.NP + n0) = pk;
      if ((lane & 7) == 0) r.maskw[row * mstride + (n0 >> 5)] = mw;
    } else {
      r.score[row * d.NP + n0] = (uint16_t)(res[0] & 0xFFFFu);
      unsigned long long b = __ballot((res[0] >> 16) & 1u);   // one wave = 64 consecutive nodes = two mask words
      if (lane == 0) *reinterpret_cast<unsigned long long *>(r.maskw + row * mstride + (n0 >> 5)) = b;
    }
  }
}

// K1 for the launches that evaluate every task row themselves (kb_eval_matrix / kb_bench_matrix with more distinct shapes than the per-shape
// rows' cache residency allows): the tasks of a job are adjacent and equal, so the rows of a tile come in RUNS of identical rows.  The
// <4, 32> tile above stores every row of a run from each thread's own registers (8 bytes per lane per row, one row at a time); rocprofv3's
// SQ counters (profiles/round3/k1_profile) show what that costs: 141 VGPRs -> three waves per SIMD, each alive for 16 us of which 3 us issue
// VALU work: the stores of a run are issued one by one between LDS reads and scalar branches, with nothing to overlap them.
// Here a run is evaluated ONCE into LDS (2 KB of scores + 128 B of mask for the tile's 1 024 nodes) and then streamed out by the whole
// workgroup with 16-byte stores, every thread holding its 16 bytes in registers and issuing run_len / 2 independent stores back to back:
// the row expansion's store pattern without its load.
__global__ void __launch_bounds__(256) k_matrix_runs(KbDev d, KbRound r) {
  constexpr int NPT = 4, TR = 32;
  __shared__ K1Task srow[TR];
  __shared__ uint32_t ssame[TR];
  __shared__ __align__(16) uint2 sres[256];       // 1 024 u16 scores of the current run
  __shared__ __align__(16) uint32_t smask[32];    // their mask bits
  const uint32_t row0 = blockIdx.y * TR;
  const uint32_t nr = min((uint32_t)TR, r.n_mrows - row0);
  if (threadIdx.x < TR) {
    uint32_t same = 0;
    if (threadIdx.x < nr) {
      const uint32_t i = row0 + threadIdx.x;
      srow[threadIdx.x] = k1_task(d, r.mrows ? r.mrows[i] : r.mrow_task0 + i);
      same = (r.same_prev && r.same_prev[i] && threadIdx.x > 0) ? 1u : 0u;
    }
    ssame[threadIdx.x] = same;
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  // bit i: row i of the tile equals row i - 1 (never bit 0; rows past the tile's end read 0 and end the last run)
  const uint32_t samemask = (uint32_t)__ballot(lane < TR && ssame[lane & (TR - 1)] != 0u);
  const uint32_t nb = blockIdx.x * 1024u;                       // the tile's first node
  const uint32_t n0 = nb + threadIdx.x * NPT;
  K1Node nv[NPT];
#pragma unroll
  for (int j = 0; j < NPT; j++) nv[j] = k1_node(d, n0 + j);
  const size_t mstride = d.NP / 32;
  uint32_t rr = 0;
  while (rr < nr) {
    // rows rr+1 .. of the tile that repeat row rr: the first 0 bit above rr ends the run
    const uint32_t follow = (rr + 1u < 32u) ? ~(samemask >> (rr + 1u)) : 1u;
    const uint32_t run_len = min(1u + (uint32_t)__builtin_ctz(follow | 0x80000000u), nr - rr);
    uint32_t res[NPT];
    const K1Task tv = k1_uniform(srow[rr]);
    eval_row<NPT>(d, tv, nv, n0, r.fit_mode, res);
    uint2 pk;
    pk.x = (res[0] & 0xFFFFu) | (res[1] << 16);
    pk.y = (res[2] & 0xFFFFu) | (res[3] << 16);
    const uint32_t nib = ((res[0] >> 16) & 1u) | (((res[1] >> 16) & 1u) << 1) | (((res[2] >> 16) & 1u) << 2) | (((res[3] >> 16) & 1u) << 3);
    uint32_t w = nib << (4 * (lane & 7));                        // OR the 8 lanes' nibbles into one mask word (DPP, no LDS crossbar)
    w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x141, 0xf, 0xf, false);   // row_half_mirror
    const size_t row = row0 + rr;
    if (run_len == 1u) {   // nothing to share: straight from the registers
      *reinterpret_cast<uint2 *>(r.score + row * d.NP + n0) = pk;
      if ((lane & 7) == 0) r.maskw[row * mstride + (n0 >> 5)] = w;
    } else {
      sres[threadIdx.x] = pk;
      if ((lane & 7) == 0) smask[threadIdx.x >> 3] = w;
      __syncthreads();
      {
        const uint32_t c = threadIdx.x & 127u;                   // this thread's 16-byte column of the tile's 2 KB row segment
        const uint4 v = reinterpret_cast<const uint4 *>(sres)[c];
        uint16_t *dst = r.score + (row + (threadIdx.x >> 7)) * d.NP + nb + c * 8u;
        for (uint32_t k = threadIdx.x >> 7; k < run_len; k += 2u, dst += (size_t)2 * d.NP) *reinterpret_cast<uint4 *>(dst) = v;
      }
      if (threadIdx.x < 8u * run_len) {                          // 128 B of mask per row: eight 16-byte columns (run_len <= 32 rows)
        const uint32_t c = threadIdx.x & 7u;
        const uint4 v = reinterpret_cast<const uint4 *>(smask)[c];
        *reinterpret_cast<uint4 *>(r.maskw + (row + (threadIdx.x >> 3)) * mstride + (nb >> 5) + c * 4u) = v;
      }
      __syncthreads();   // the next run overwrites sres / smask
    }
    rr += run_len;
  }
}


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

// ------------------------------------------------------------------------------------------------------------
// K3: segmented arg-max generalised to a sorted candidate list (best first: descending score, ascending node index).
// One 256-thread workgroup per matrix row.  The row (u16 scores + mask bits) is staged once in LDS with 16-byte loads;
// each thread owns a contiguous run of NP/256 nodes, so "ascending node index" is thread order.  Per score level:
// pass A = block max of the scores below the previous level, pass B = count + block exclusive scan + ordered write.
// The first entry is util.SelectBestNode's choice (scheduler_helper.go:188-208, canonical first-max tie-break).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_max_i32_dpp(int v) {
#define KB_DPP_IMAX(ctrl, row_mask) v = max(v, __builtin_amdgcn_update_dpp(v, v, (ctrl), (row_mask), 0xf, false))
  KB_DPP_IMAX(0xB1, 0xf);
  KB_DPP_IMAX(0x4E, 0xf);
  KB_DPP_IMAX(0x141, 0xf);
  KB_DPP_IMAX(0x140, 0xf);
  KB_DPP_IMAX(0x142, 0xa);
  KB_DPP_IMAX(0x143, 0xc);
#undef KB_DPP_IMAX
  return __builtin_amdgcn_readlane(v, 63);
}
// Sorted candidate list of one matrix row: the first K entries of (score descending, node ascending).  Scores are integers,
// so the list is built one BAND of consecutive score values at a time (eight values; WIDE, rows of 65536 nodes or more where a
// 16-bit counter could overflow: four): one pass over the row (staged in LDS) counts each thread's nodes at every value of
// the band (16-bit counters packed two per word) and finds the best score below the band; packed wave scans turn the counts
// into output positions; one more pass scatters.  Spreading scores fill a window from the first band; bin-packing scores,
// whose best nodes are few per value, need several bands: NW = 4 words (eight values per band) when mostrequested carries weight,
// NW = 2 otherwise (config 4: 34 -> 26 us per launch; config 3 keeps its 11 us).
template <bool WIDE, int THREADS, int NW>
__global__ void __launch_bounds__(THREADS) k_argmax(KbDev d, KbRound r) {
  constexpr int WAVES = THREADS / 64;
  extern __shared__ __align__(16) unsigned char k3_smem[];
  uint16_t *ls = reinterpret_cast<uint16_t *>(k3_smem);                        // [NP] scores
  uint8_t *lm = reinterpret_cast<uint8_t *>(k3_smem) + (size_t)d.NP * 2;      // [NP/8] mask bytes
  __shared__ int s_wmax[WAVES];
  constexpr int VALS = WIDE ? NW : 2 * NW;   // consecutive score values one band covers
  __shared__ uint32_t s_wcnt[WAVES][NW];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t row = blockIdx.x;
  if (KB_CHAIN_BROKEN(r)) return;
  if (row == 0 && tid == 0 && r.mrow_task0 == 0 && r.mrows != nullptr)   // round launches only (not kb_eval_matrix's expanded rows)
    reinterpret_cast<unsigned long long *>(r.result)[KB_OUT_STAMP0 + 1] = wall_clock64();
  const uint32_t K = r.L;
  unsigned long long *out = r.keys + (size_t)row * K;
  // contiguous, balanced ranges of 8-node chunks: thread order == node order (NP/8 need not be a multiple of THREADS)
  const uint32_t nchunks = d.NP / 8;
  const uint32_t cbase = (uint32_t)(((unsigned long long)tid * nchunks) / THREADS);
  const uint32_t per8 = (uint32_t)(((unsigned long long)(tid + 1) * nchunks) / THREADS) - cbase;
  const uint4 *ls4 = reinterpret_cast<const uint4 *>(ls);
  int top = -1;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(r.score + (size_t)row * d.NP);
    uint4 *dst = reinterpret_cast<uint4 *>(ls);
    for (uint32_t c = tid; c < d.NP / 8; c += THREADS) dst[c] = src[c];
    const uint32_t *msrc = r.maskw + (size_t)row * (d.NP / 32);
    uint32_t *mdst = reinterpret_cast<uint32_t *>(lm);
    for (uint32_t c = tid; c < d.NP / 32; c += THREADS) mdst[c] = msrc[c];
  }
  __syncthreads();
  for (uint32_t c = 0; c < per8; c++) {   // best score of the row
    const uint32_t mb = lm[cbase + c];
    const uint4 sv = ls4[cbase + c];
    const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
      if (((mb >> e) & 1u) && sc > top) top = sc;
    }
  }
  top = wave_max_i32_dpp(top);
  if (lane == 0) s_wmax[wave] = top;
  __syncthreads();
  top = s_wmax[0];
#pragma unroll
  for (int w2 = 1; w2 < WAVES; w2++) top = max(top, s_wmax[w2]);
  uint32_t found = 0;
  while (top >= 0 && found < K) {
    __syncthreads();   // s_wmax / s_wcnt are reused
    // counts of this thread's nodes at the VALS scores of the band (16-bit counters, two per word; WIDE: one per word); best
    // score below the band
    uint32_t cw[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) cw[k] = 0;
    int below = -1;
    for (uint32_t c = 0; c < per8; c++) {
      const uint32_t mb = lm[cbase + c];
      const uint4 sv = ls4[cbase + c];
      const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
        if ((mb >> e) & 1u) {
          const int l = top - sc;   // 0 .. VALS-1 inside the band
          if (l < VALS) {
#pragma unroll
            for (int k = 0; k < NW; k++)
              if ((WIDE ? l : (l >> 1)) == k) cw[k] += WIDE ? 1u : (1u << (16 * (l & 1)));
          } else if (sc > below) below = sc;
        }
      }
    }
    uint32_t pw[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) pw[k] = wave_incl_scan_u32(cw[k]);
    below = wave_max_i32_dpp(below);
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < NW; k++) s_wcnt[wave][k] = pw[k];
    }
    if (lane == 0) s_wmax[wave] = below;
    __syncthreads();
    uint32_t offw[NW], totw[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) { offw[k] = 0; totw[k] = 0; }
#pragma unroll
    for (int w2 = 0; w2 < WAVES; w2++) {
#pragma unroll
      for (int k = 0; k < NW; k++) {
        if (w2 < (int)wave) offw[k] += s_wcnt[w2][k];
        totw[k] += s_wcnt[w2][k];
      }
    }
    below = s_wmax[0];
#pragma unroll
    for (int w2 = 1; w2 < WAVES; w2++) below = max(below, s_wmax[w2]);
    // first output position of this thread's nodes at each score of the band
    uint32_t pos[VALS], run = found, any = 0, first = 0xFFFFFFFFu;
#pragma unroll
    for (int v = 0; v < VALS; v++) {
      const int k = WIDE ? v : (v >> 1), sh = WIDE ? 0 : 16 * (v & 1);
      const uint32_t mine = WIDE ? cw[k] : ((cw[k] >> sh) & 0xFFFFu);
      const uint32_t before = WIDE ? (offw[k] + pw[k] - cw[k]) : (((offw[k] + pw[k] - cw[k]) >> sh) & 0xFFFFu);
      const uint32_t total = WIDE ? totw[k] : ((totw[k] >> sh) & 0xFFFFu);
      pos[v] = run + before;
      if (mine) { any = 1; first = min(first, pos[v]); }
      run += total;
    }
    if (any && first < K) {
      for (uint32_t c = 0; c < per8; c++) {
        const uint32_t mb = lm[cbase + c];
        const uint4 sv = ls4[cbase + c];
        const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int sc = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
          const int l = top - sc;
          if (((mb >> e) & 1u) && l >= 0 && l < VALS) {
            uint32_t at = 0;
#pragma unroll
            for (int v = 0; v < VALS; v++)
              if (l == v) at = pos[v]++;
            if (at < K) out[at] = KB_KEY(sc, (cbase + c) * 8 + e);
          }
        }
      }
    }
    found = run;
    top = below;
  }
  if (found > K) found = K;
  for (uint32_t i = found + tid; i < K; i += THREADS) out[i] = 0ull;
  if (r.ready != nullptr) {   // overlapped round: the list (and the row's task record) is complete and visible before its tag is
    static_assert(sizeof(K1Task) == 64, "KbRound::task_rows holds 64-byte records");
    if (tid == 0 && r.task_rows != nullptr) reinterpret_cast<K1Task *>(r.task_rows)[row] = k1_task(d, r.mrows ? r.mrows[row] : r.mrow_task0 + row);
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&r.ready[row], r.ready_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_repair: the candidate lists of an OVERLAPPED round as a launch of its own (kb_repair.hpp has the work; the selection kernel's launch
// carries the same workgroups itself: kb_commit_sel.hip).  One workgroup per matrix row, behind the predecessor's commit on the first stream.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(KB_REPAIR_THREADS) k_repair(KbDev d, KbRound r) {
  extern __shared__ __align__(16) unsigned char kr_smem[];
  if (KB_CHAIN_BROKEN(r)) return;
  kb_repair_row<KB_REPAIR_THREADS>(d, r, kr_smem, blockIdx.x, threadIdx.x);
}
size_t kb_repair_smem_bytes(uint32_t NP) { return kb_repair_lds_bytes(NP, KB_REPAIR_THREADS); }
void kb_launch_repair(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_mrows == 0) return;
  const size_t sh = kb_repair_smem_bytes(d.NP);   // the engine overlaps rounds only while this fits the attribute below (kb_engine.cpp: overlap_ok)
  static bool lds_set[64] = {};
  kb_allow_lds(reinterpret_cast<const void *>(k_repair), 150 * 1024, lds_set);
  hipLaunchKernelGGL(k_repair, dim3(r.n_mrows), dim3(KB_REPAIR_THREADS), sh, (hipStream_t)stream, d, r);
}

__device__ __forceinline__ bool bit_test(const uint32_t *bm, uint32_t n) { return (bm[n >> 5] >> (n & 31)) & 1u; }


// window rows -> contiguous descriptors (multi-GPU rounds launch it on its own; single-GPU rounds run it as one extra block
// row of the matrix launch)
__global__ void __launch_bounds__(256) k_gather(KbDev d, KbRound r) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (KB_CHAIN_BROKEN(r)) return;
  if (i == 0) reinterpret_cast<unsigned long long *>(r.result)[KB_OUT_STAMP0] = wall_clock64();   // round start (constant-rate clock)
  gather_row(d, r, i);
}

// nodeorder's NodeAffinity priority for the matrix rows whose task class has preferred terms (rare): Map = the class-pair count,
// Reduce = NormalizeReduce(10) over the row's FEASIBLE nodes (vendor/.../priorities/reduce.go:28-63: max == 0 leaves zeros,
// else 10 * count / max, integer division), then Score += score * weight (scheduler_helper.go:162-168).  One workgroup per row.
__global__ void __launch_bounds__(256) k_affinity(KbDev d, KbRound r) {
  __shared__ int s_max[4];
  const uint32_t row = blockIdx.x, tid = threadIdx.x;
  if (KB_CHAIN_BROKEN(r)) return;
  const uint32_t t = r.mrows ? r.mrows[row] : r.mrow_task0 + row;
  const uint32_t tc = d.t_cls[t];
  if (!d.aff_cls[tc]) return;   // uniform per block
  const int32_t *arow = d.aff + (size_t)tc * d.n_nc;
  const uint32_t *mw = r.maskw + (size_t)row * (d.NP / 32);
  uint16_t *sc = r.score + (size_t)row * d.NP;
  int mx = 0;
  for (uint32_t n = tid; n < d.N; n += 256)
    if ((mw[n >> 5] >> (n & 31)) & 1u) mx = max(mx, (int)arow[d.ncls[n]]);
  mx = wave_max_i32(mx);
  if ((tid & 63) == 0) s_max[tid >> 6] = mx;
  __syncthreads();
  mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
  if (mx == 0) return;
  for (uint32_t n = tid; n < d.N; n += 256)
    if ((mw[n >> 5] >> (n & 31)) & 1u) sc[n] = (uint16_t)(sc[n] + (10 * arow[d.ncls[n]] / mx) * d.wNA);
}

// ------------------------------------------------------------------------------------------------------------
// K2 + K4: one wave per job.  Ready count by ballot/popcount, gang-gated Allocated -> Binding flip, segmented sum
// of Resreq per job (drf) and per queue (proportion, f64 atomics: the addends are integer-valued milli-units /
// bytes below 2^53, so the sum is exact and order-independent), then the share maxima.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double share_of(double l, double r) { return (r == 0.0) ? ((l == 0.0) ? 0.0 : 1.0) : l / r; }

__global__ void __launch_bounds__(256) k_finalize_jobs(KbDev d, const uint32_t *job_task_begin, const int *job_min_avail,
                                                       const uint32_t *job_queue, int gang_ready_enabled, const double *total,
                                                       uint32_t total_mask, double *job_alloc, double *job_share, double *queue_alloc,
                                                       int *job_ready_cnt) {
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t j = blockIdx.x * 4 + wave;
  if (j >= d.J) return;
  const uint32_t t0 = job_task_begin[j], t1 = job_task_begin[j + 1];
  int ready = 0;
  for (uint32_t base = t0; base < t1; base += 64) {
    uint32_t t = base + lane;
    int st = (t < t1) ? (int)d.t_status[t] : -1;
    bool isr = st == KB_TASK_BOUND || st == KB_TASK_BINDING || st == KB_TASK_RUNNING || st == KB_TASK_ALLOCATED || st == KB_TASK_SUCCEEDED;
    ready += __popcll(__ballot(isr));     // JobInfo.ReadyTaskNum (job_info.go:383-394)
  }
  const bool job_ready = gang_ready_enabled ? (ready >= job_min_avail[j]) : true;   // gang.go:122-125 / session_plugins.go:182-200
  // session.go:277-285: the dispatch sits inside ssn.Allocate, so it needs an Allocate on this job in this action; the ready
  // count only changes through ssn.Allocate here, hence "ready after the job's last Allocate" == "ready now".  Every task in
  // TaskStatusIndex[Allocated] goes, including ones the snapshot already carried as Allocated.
  if (job_ready && d.j_allocated[j]) {
    for (uint32_t t = t0 + lane; t < t1; t += 64)
      if (d.t_status[t] == KB_TASK_ALLOCATED) { d.t_status[t] = KB_TASK_BINDING; d.t_bind[t] = d.t_node[t]; }
  }
  if (lane == 0) { job_ready_cnt[j] = ready; d.j_allocated[j] = 0; }
  const uint32_t q = job_queue[j];
  double share = 0.0;
  for (int dim = 0; dim < d.R; dim++) {
    double s = 0.0;
    for (uint32_t t = t0 + lane; t < t1; t += 64)
      if (d.t_counted[t]) s += d.t_res[(size_t)dim * d.T + t];
    s = wave_sum_f64(s);
    if (lane == 0) {
      job_alloc[(size_t)j * d.R + dim] = s;
      if (s != 0.0 && q < d.Q) atomicAdd(&queue_alloc[(size_t)q * d.R + dim], s);   // q >= Q: "queue not found" (allocate.go:56-60), no queue row
      if (dim < 2 || ((total_mask >> (dim - 2)) & 1u)) {   // totalResource.ResourceNames() (drf.go:161)
        double sh = share_of(s, total[dim]);
        if (sh > share) share = sh;
      }
    }
  }
  if (lane == 0) job_share[j] = share;
}

__global__ void k_finalize_queues(KbDev d, const double *deserved, const uint32_t *deserved_mask, const double *queue_alloc, double *queue_share) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= d.Q) return;
  double share = 0.0;
  for (int dim = 0; dim < d.R; dim++) {
    if (dim >= 2 && !((deserved_mask[q] >> (dim - 2)) & 1u)) continue;   // attr.deserved.ResourceNames() (proportion.go:245)
    double sh = share_of(queue_alloc[(size_t)q * d.R + dim], deserved[(size_t)dim * d.Q + q]);
    if (sh > share) share = sh;
  }
  queue_share[q] = share;
}

// ------------------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------------------
// multi-GPU: install round-start state + all-reduced deltas, counting values that differ from the replica's own commit
__global__ void __launch_bounds__(256) k_apply_deltas(KbDev d, const double *s_idle, const double *s_rel, const long long *s_nzc,
                                                      const long long *s_nzm, const int *s_podcnt, const double *delta, uint32_t *counter) {
  uint32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= d.NP) return;
  uint32_t bad = 0;
  for (int dim = 0; dim < d.R; dim++) {
    size_t o = (size_t)dim * d.NP + n;
    double vi = s_idle[o] + delta[o];
    double vr = s_rel[o] + delta[(size_t)d.R * d.NP + o];
    bad += (vi != d.idle[o]) + (vr != d.rel[o]);
    d.idle[o] = vi;
    d.rel[o] = vr;
  }
  const double *tail = delta + (size_t)2 * d.R * d.NP;
  long long c = s_nzc[n] + (long long)tail[n], m = s_nzm[n] + (long long)tail[(size_t)d.NP + n];
  int p = s_podcnt[n] + (int)tail[(size_t)2 * d.NP + n];
  bad += (c != d.nzc[n]) + (m != d.nzm[n]) + (p != d.podcnt[n]);
  d.nzc[n] = c; d.nzm[n] = m; d.podcnt[n] = p;
  if (bad) atomicAdd(counter, bad);
}
uint32_t kb_apply_deltas(const KbDev &d, const double *s_idle, const double *s_rel, const long long *s_nzc, const long long *s_nzm,
                         const int *s_podcnt, const double *delta, uint32_t *dev_counter, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(dev_counter, 0, sizeof(uint32_t), s);
  hipLaunchKernelGGL(k_apply_deltas, dim3((d.NP + 255) / 256), dim3(256), 0, s, d, s_idle, s_rel, s_nzc, s_nzm, s_podcnt, delta, dev_counter);
  uint32_t h = 0;
  (void)hipMemcpyAsync(&h, dev_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  return h;
}
__global__ void __launch_bounds__(256) k_check_deltas(uint32_t NP, int R, KbNodeCopy s0, KbNodeCopy s1, const double *delta, uint32_t *counter) {
  const uint32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= NP) return;
  uint32_t bad = 0;
  for (int dim = 0; dim < R; dim++) {
    const size_t o = (size_t)dim * NP + n;
    bad += (s0.idle[o] + delta[o] != s1.idle[o]) + (s0.rel[o] + delta[(size_t)R * NP + o] != s1.rel[o]);
  }
  const double *tail = delta + (size_t)2 * R * NP;
  bad += (s0.nzc[n] + (long long)tail[n] != s1.nzc[n]) + (s0.nzm[n] + (long long)tail[(size_t)NP + n] != s1.nzm[n]) +
         (s0.podcnt[n] + (int)tail[(size_t)2 * NP + n] != s1.podcnt[n]);
  if (bad) atomicAdd(counter, bad);
}
void kb_check_deltas(const KbDev &d, const KbNodeCopy &s0, const KbNodeCopy &s1, const double *delta, uint32_t *dev_counter, void *stream) {
  hipLaunchKernelGGL(k_check_deltas, dim3((d.NP + 255) / 256), dim3(256), 0, (hipStream_t)stream, d.NP, d.R, s0, s1, delta, dev_counter);
}
__global__ void __launch_bounds__(256) k_scatter_nodes(KbDev d, const unsigned long long *__restrict__ rec, uint32_t n, uint32_t *nmask) {
  const uint32_t words = 5u + 2u * (uint32_t)d.R;
  const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * words) return;
  const uint32_t i = idx / words, f = idx % words;
  const unsigned long long *r = rec + (size_t)i * words;
  const uint32_t node = (uint32_t)r[0];
  const unsigned long long v = r[f];
  if (f == 0) nmask[node] = (uint32_t)(v >> 32);
  else if (f == 1) d.podcnt[node] = (int)(uint32_t)v;
  else if (f == 2) d.nzc[node] = (long long)v;
  else if (f == 3) d.nzm[node] = (long long)v;
  else if (f == 4) { if (d.ports) d.ports[node] = v; }
  else if (f < 5u + (uint32_t)d.R) d.idle[(size_t)(f - 5u) * d.NP + node] = __longlong_as_double((long long)v);
  else d.rel[(size_t)(f - 5u - (uint32_t)d.R) * d.NP + node] = __longlong_as_double((long long)v);
}
void kb_launch_scatter_nodes(const KbDev &d, const unsigned long long *rec, uint32_t n, uint32_t *nmask, void *stream) {
  if (n == 0) return;
  const uint32_t total = n * (5u + 2u * (uint32_t)d.R);
  hipLaunchKernelGGL(k_scatter_nodes, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, rec, n, nmask);
}
void kb_launch_gather(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_rows == 0) return;
  hipLaunchKernelGGL(k_gather, dim3((r.n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, r);
}
void kb_launch_matrix(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_mrows == 0) return;
  if ((size_t)r.n_mrows * d.NP >= (4u << 20)) {
    // <4,32> amortises a thread's node state over 32 rows and stores 8 bytes per row, but a few hundred rows (the distinct shapes of
    // a whole session) make only ~100 such workgroups: below four per CU the one-node tile with 16 rows fills the chip instead
    const size_t blocks = (size_t)(d.NP / (256 * 4)) * ((r.n_mrows + 31) / 32);
    if (blocks >= 1024 && r.same_prev != nullptr && !r.gather && r.chain == nullptr) {
      hipLaunchKernelGGL(k_matrix_runs, dim3(d.NP / 1024, (r.n_mrows + 31) / 32), dim3(256), 0, (hipStream_t)stream, d, r);
    } else if (blocks >= 1024) {
      dim3 grid(d.NP / (256 * 4), (r.n_mrows + 31) / 32 + (r.gather ? 1 : 0));
      hipLaunchKernelGGL((k_matrix<4, 32>), grid, dim3(256), 0, (hipStream_t)stream, d, r);
    } else {
      dim3 grid(d.NP / 256, (r.n_mrows + 15) / 16 + (r.gather ? 1 : 0));
      hipLaunchKernelGGL((k_matrix<1, 16>), grid, dim3(256), 0, (hipStream_t)stream, d, r);
    }
  } else {
    dim3 grid(d.NP / 256, (r.n_mrows + 3) / 4 + (r.gather ? 1 : 0));
    hipLaunchKernelGGL((k_matrix<1, 4>), grid, dim3(256), 0, (hipStream_t)stream, d, r);
  }
}
// nodeorder's InterPodAffinityPriority (plugins/nodeorder/nodeorder.go:156-160 -> vendor/.../priorities/interpod_affinity.go:99-235) on
// the kb_interpod tables, for the matrix rows whose task carries weights.  One workgroup per row:
//   count(i) = sum_p w_p * ( sum over FEASIBLE n with dom_p(n) == dom_p(i) of bound_p[n]  +  [dom_p(i) == dom_p(Z)] * sum over feasible n of
//   unbound_p[n] ),  score(i) += int(10 * (count(i) - min) / (max - min)) * podaffinity.weight  with min / max over the feasible nodes and 0.
// Only the pods of the feasible nodes count (util/scheduler_helper.go:226-238); a pod whose Spec.NodeName is still empty is looked
// up through nodeorder's cachedNodeInfo and lands on Z, the first node holding any such pod (nodeorder.go:48-62).
__global__ void __launch_bounds__(256) k_interpod(KbDev d, KbRound r) {
  __shared__ long long s_red[2][4];
  const uint32_t row = blockIdx.x, tid = threadIdx.x;
  if (KB_CHAIN_BROKEN(r)) return;
  const uint32_t t = r.mrows ? r.mrows[row] : r.mrow_task0 + row;
  const uint32_t sig = d.t_ip_sig[t];
  if (sig == KB_NONE_U32) return;   // uniform per block
  const int32_t *w = d.ip_sig_w + (size_t)sig * d.ip_P;
  const uint32_t *mw = r.maskw + (size_t)row * (d.NP / 32);
  uint16_t *sc = r.score + (size_t)row * d.NP;
  long long *cnt = d.ip_scratch_cnt + (size_t)row * d.NP;
  int32_t *hist = d.ip_scratch_hist + (size_t)row * d.NP;
  const uint32_t Z = *d.ip_z;
  for (uint32_t n = tid; n < d.N; n += 256) cnt[n] = 0;
  for (uint32_t p = 0; p < d.ip_P; p++) {
    const int wp = w[p];
    if (wp == 0) continue;
    const uint32_t *dom = d.ip_cls_dom + (size_t)p * d.NP;
    const int32_t *cb = d.ip_cls_bound + (size_t)p * d.NP, *cu = d.ip_cls_unbound + (size_t)p * d.NP;
    for (uint32_t n = tid; n < d.N; n += 256) hist[n] = 0;   // domain ids are < N
    __syncthreads();
    long long zs = 0;
    for (uint32_t n = tid; n < d.N; n += 256)
      if ((mw[n >> 5] >> (n & 31)) & 1u) {
        zs += cu[n];
        const uint32_t dm = dom[n];
        if (dm != KB_NONE_U32 && cb[n] != 0) atomicAdd(&hist[dm], cb[n]);
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zs += __shfl_xor(zs, o);
    if ((tid & 63) == 0) s_red[0][tid >> 6] = zs;
    __threadfence_block();
    __syncthreads();
    zs = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    const uint32_t zdom = (Z != KB_NONE_U32) ? dom[Z] : KB_NONE_U32;
    for (uint32_t n = tid; n < d.N; n += 256)
      if ((mw[n >> 5] >> (n & 31)) & 1u) {
        const uint32_t dm = dom[n];
        if (dm != KB_NONE_U32) cnt[n] += (long long)wp * ((long long)__hip_atomic_load(&hist[dm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + (dm == zdom ? zs : 0ll));
      }
    __syncthreads();
  }
  long long mx = 0, mn = 0;   // interpod_affinity.go:213-220: both start at 0
  for (uint32_t n = tid; n < d.N; n += 256)
    if ((mw[n >> 5] >> (n & 31)) & 1u) { mx = max(mx, cnt[n]); mn = min(mn, cnt[n]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o)); mn = min(mn, __shfl_xor(mn, o)); }
  __syncthreads();
  if ((tid & 63) == 0) { s_red[0][tid >> 6] = mx; s_red[1][tid >> 6] = mn; }
  __syncthreads();
  mx = max(max(s_red[0][0], s_red[0][1]), max(s_red[0][2], s_red[0][3]));
  mn = min(min(s_red[1][0], s_red[1][1]), min(s_red[1][2], s_red[1][3]));
  if (mx - mn <= 0) return;
  for (uint32_t n = tid; n < d.N; n += 256)
    if ((mw[n >> 5] >> (n & 31)) & 1u) {
      const double f = 10.0 * ((double)(cnt[n] - mn) / (double)(mx - mn));   // :226-228, float64 like the reference
      sc[n] = (uint16_t)(sc[n] + (int)f * d.wPA);
    }
}

// Feasibility probe (kb_engine.cpp: ActionRun::probe_dead_shapes): thread <-> NPT nodes kept in registers for the TR task rows of the
// workgroup; a wave that finds a feasible node for a row sets the row's flag.  No scores, nothing stored per pair.
template <int NPT, int TR>
__global__ void __launch_bounds__(256) k_probe(KbDev d, const uint32_t *rows, uint32_t n_rows, uint32_t *alive) {
  __shared__ K1Task srow[TR];
  const uint32_t row0 = blockIdx.y * TR;
  const uint32_t nr = min((uint32_t)TR, n_rows - row0);
  if (threadIdx.x < nr) srow[threadIdx.x] = k1_task(d, rows[row0 + threadIdx.x]);
  __syncthreads();
  const uint32_t n0 = (blockIdx.x * 256 + threadIdx.x) * NPT;
  K1Node nv[NPT];
#pragma unroll
  for (int j = 0; j < NPT; j++) nv[j] = k1_node(d, n0 + j);
  // A row's flag is one word that every workgroup of the row's grid line may set: as an atomicOr per wave and row that was
  // (NP / 64) * n_rows agent-scope atomics on n_rows / 16 cache lines (100k x 10k: 160 per word, 82k per launch, most of the launch's 46 us).
  // Every writer stores the same 1 behind the launch's memset, so plain stores do — and the workgroup's waves merge theirs first
  // (bit rr of a wave-uniform word; TR <= 32).
  static_assert(TR <= 32, "one bit per row of the workgroup");
  __shared__ uint32_t s_any[4];
  uint32_t wave_any = 0;
  for (uint32_t rr = 0; rr < nr; rr++) {
    const K1Task tv = k1_uniform(srow[rr]);
    uint32_t pr[NPT], ok = 0;
    eval_row<NPT>(d, tv, nv, n0, 1, pr);
#pragma unroll
    for (int j = 0; j < NPT; j++) ok |= (pr[j] >> 16) & 1u;
    if (__ballot(ok)) wave_any |= 1u << rr;
  }
  if ((threadIdx.x & 63) == 0) s_any[threadIdx.x >> 6] = wave_any;
  __syncthreads();
  if (threadIdx.x < nr && (((s_any[0] | s_any[1] | s_any[2] | s_any[3]) >> threadIdx.x) & 1u)) alive[row0 + threadIdx.x] = 1u;
}
__global__ void k_or_ports_x(KbDev d, uint32_t task, uint32_t node) {
  const uint32_t w = threadIdx.x;
  if (w < d.port_xw) d.ports_x[(size_t)w * d.NP + node] |= d.t_want_x[(size_t)task * d.port_xw + w];
}
void kb_launch_or_ports_x(const KbDev &d, uint32_t task, uint32_t node, void *stream) {
  for (uint32_t w0 = 0; w0 < d.port_xw; w0 += 1024u) {   // one thread per word
    KbDev dd = d;
    dd.ports_x += (size_t)w0 * d.NP; dd.t_want_x += w0; dd.port_xw = d.port_xw;   // t_want_x keeps its row stride (port_xw)
    const uint32_t n = d.port_xw - w0 < 1024u ? d.port_xw - w0 : 1024u;
    hipLaunchKernelGGL(k_or_ports_x, dim3(1), dim3(n), 0, (hipStream_t)stream, dd, task, node);
  }
}
void kb_launch_probe(const KbDev &d, const uint32_t *rows, uint32_t n_rows, uint32_t *alive, void *stream) {
  if (n_rows == 0) return;
  KbDev dd = d;
  dd.score_enabled = 0;   // feasibility only
  if ((size_t)(d.NP / 1024) * ((n_rows + 31) / 32) >= 512)   // enough workgroups of the amortising tile to fill the chip
    hipLaunchKernelGGL((k_probe<4, 32>), dim3(d.NP / 1024, (n_rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, dd, rows, n_rows, alive);
  else
    hipLaunchKernelGGL((k_probe<1, 16>), dim3(d.NP / 256, (n_rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, dd, rows, n_rows, alive);
}
void kb_launch_interpod(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_mrows == 0 || !d.t_ip_sig || !d.score_enabled || d.ip_P == 0) return;
  hipLaunchKernelGGL(k_interpod, dim3(r.n_mrows), dim3(256), 0, (hipStream_t)stream, d, r);
}
void kb_launch_affinity(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_mrows == 0 || !d.aff || !d.score_enabled) return;
  hipLaunchKernelGGL(k_affinity, dim3(r.n_mrows), dim3(256), 0, (hipStream_t)stream, d, r);
}
// K1b: row expansion.  Tasks with the same shape (InitResreq, non-zero request, class) have identical matrix rows, so the materialised T x N matrix is
// produced by evaluating each distinct shape once (k_matrix over the representative rows) and streaming every task row out of its shape's row: the
// stores are the launch's HBM traffic (2 B score + 1 mask bit per evaluation).  The rows come in SHAPE order (order[]: the host's counting sort by slot),
// cut into chunks of at most 64 rows of one shape.
// Tiled: workgroup (tile, chunk) holds its tile of the chunk's shape row in registers — up to 16 384 u16 scores (eight 16-byte pieces per thread: a
// whole row at 10k nodes, so a row leaves as ONE contiguous 20 KB stream) and the tile's mask words — and streams them to every task row of
// the chunk: the launch's HBM traffic is its stores (2.125 B per pair) plus one read of a shape tile per 64 rows
#define KB_XTILE_NODES 16384u
// (non-temporal stores for the rows — written once, read by nobody in this launch — measured no consistent gain: 0.353 / 0.410 ms against 0.386 / 0.369 at
//  100k x 10k on one box, profiles/round6/call8_expand_tiles_whole_row/summary.txt)
__global__ void __launch_bounds__(256) k_expand_tiles(const uint16_t *__restrict__ s_score, const uint32_t *__restrict__ s_mask, const uint32_t *__restrict__ order,
                                                      const KbXChunk *__restrict__ chunks, uint32_t NP, uint16_t *__restrict__ score, uint32_t *__restrict__ maskw) {
  const KbXChunk c = chunks[blockIdx.y];
  const uint32_t node0 = blockIdx.x * KB_XTILE_NODES, tn = min(KB_XTILE_NODES, NP - node0);   // NP is a multiple of 2048
  const uint32_t n16 = tn / 8u, m16 = tn / 128u;   // 16-byte pieces of the tile's scores / mask words
  const size_t mstride = NP / 32;
  const uint4 *src = reinterpret_cast<const uint4 *>(s_score + (size_t)c.slot * NP + node0);
  uint4 v[8];
#pragma unroll
  for (uint32_t k = 0; k < 8u; k++) { const uint32_t p = threadIdx.x + 256u * k; v[k] = p < n16 ? src[p] : make_uint4(0u, 0u, 0u, 0u); }
  const bool ml = threadIdx.x < m16;   // (m16 <= 128)
  uint4 mv = make_uint4(0u, 0u, 0u, 0u);
  if (ml) mv = reinterpret_cast<const uint4 *>(s_mask + (size_t)c.slot * mstride + node0 / 32u)[threadIdx.x];
  // the chunk's rows: lane i of every wave holds row i (KB_XCHUNK_ROWS = 64 = a wave) — one vector load in front of the loop instead of a scalar
  // load, and the wait for it, in every iteration; the stores then leave back to back
  static_assert(KB_XCHUNK_ROWS <= 64u, "one lane per row of a chunk");
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t myrow = lane < c.count ? (order ? order[c.first + lane] : c.first + lane) : 0u;   // (no `order`: the chunk is rows first .. first + count - 1)
  for (uint32_t i = 0; i < c.count; i++) {
    const size_t row = (size_t)(uint32_t)__builtin_amdgcn_readlane((int)myrow, (int)i);
    uint4 *dst = reinterpret_cast<uint4 *>(score + row * NP + node0);
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) { const uint32_t p = threadIdx.x + 256u * k; if (p < n16) dst[p] = v[k]; }
    if (ml) reinterpret_cast<uint4 *>(maskw + row * mstride + node0 / 32u)[threadIdx.x] = mv;
  }
}

void kb_launch_expand(const KbDev &d, const uint16_t *s_score, const uint32_t *s_mask, const uint32_t *row_slot, const uint32_t *order, uint32_t n_rows,
                      uint16_t *score, uint32_t *maskw, void *stream, const KbXChunk *chunks, uint32_t n_chunks) {
  if (n_rows == 0) return;
  (void)row_slot;   // (the chunks carry their shape; the row -> shape map is the emulated launch's cross-check)
  if (!chunks || !n_chunks) return;
  // NP is a multiple of KB_NODE_PAD = 2048
  hipLaunchKernelGGL(k_expand_tiles, dim3((d.NP + KB_XTILE_NODES - 1u) / KB_XTILE_NODES, n_chunks), dim3(256), 0, (hipStream_t)stream, s_score, s_mask, order, chunks, d.NP, score, maskw);
}
template <bool WIDE, int THREADS, int NW> static void k3_launch(const KbDev &d, const KbRound &r, size_t sh, hipStream_t st) {
  static bool lds_set[64] = {};   // one per instantiation
  kb_allow_lds(reinterpret_cast<const void *>(k_argmax<WIDE, THREADS, NW>), 150 * 1024, lds_set);
  hipLaunchKernelGGL((k_argmax<WIDE, THREADS, NW>), dim3(r.n_mrows), dim3(THREADS), sh, st, d, r);
}
void kb_launch_argmax(const KbDev &d, const KbRound &r, void *stream) {
  if (r.n_mrows == 0) return;
  const size_t sh = (size_t)d.NP * 2 + d.NP / 8;
  hipStream_t st = (hipStream_t)stream;
  // few rows (a round's distinct shapes): the launch is latency-sized, 1024 threads shorten every pass over the row; many rows
  // (kb_argmax_rows over whole task ranges): 256 threads keep more rows resident per CU.  Wider bands when the bin-packing
  // scorer carries weight (its top scores are sparsely populated).
  const bool wide_bands = d.score_enabled && d.wM > 0;
  if (d.NP >= 65536u) { if (wide_bands) k3_launch<true, 1024, 4>(d, r, sh, st); else k3_launch<true, 1024, 2>(d, r, sh, st); }
  else if (r.n_mrows <= 512) { if (wide_bands) k3_launch<false, 1024, 4>(d, r, sh, st); else k3_launch<false, 1024, 2>(d, r, sh, st); }
  else { if (wide_bands) k3_launch<false, 256, 4>(d, r, sh, st); else k3_launch<false, 256, 2>(d, r, sh, st); }
}
void kb_launch_finalize(const KbDev &d, const uint32_t *job_task_begin, const int *job_min_avail, const uint32_t *job_queue,
                        int gang_ready_enabled, const double *total, uint32_t total_mask, const double *deserved,
                        const uint32_t *deserved_mask, double *job_alloc, double *job_share, double *queue_alloc,
                        double *queue_share, int *job_ready_cnt, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(queue_alloc, 0, sizeof(double) * (size_t)d.Q * d.R, s);
  if (d.J) hipLaunchKernelGGL(k_finalize_jobs, dim3((d.J + 3) / 4), dim3(256), 0, s, d, job_task_begin, job_min_avail, job_queue,
                              gang_ready_enabled, total, total_mask, job_alloc, job_share, queue_alloc, job_ready_cnt);
  if (d.Q) hipLaunchKernelGGL(k_finalize_queues, dim3((d.Q + 127) / 128), dim3(128), 0, s, d, deserved, deserved_mask, queue_alloc, queue_share);
}
