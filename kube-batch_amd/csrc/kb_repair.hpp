// kb_repair.hpp — the candidate lists of an OVERLAPPED round (KbRound::ready) made exact: one workgroup per matrix row.
//
// The round's matrix and arg-max launches ran on the second stream while the predecessor round's commit kernel was still changing nodes:
// their lists are exact for every node the predecessor left alone and arbitrary for the nodes it changed.  Behind the predecessor's commit:
//   1. the predecessor's nodes (its decision records; a node may have taken several rows) -> a bitmap in LDS, each node owned by one thread;
//   2. the owner evaluates the row's shape against the node's state as the predecessor LEFT it (eval_row<1>, K1's own arithmetic);
//   3. the stale list without the predecessor's nodes, merged with the new keys by rank.  A new key f finds its place lo(f) in the stale list (the
//      number of stale entries above it: binary search) and leaves a mark there; a survivor at index i then has exactly the new keys with
//      lo(f) <= i above it — a prefix sum of the marks, in the same block scan that counts the survivors in front of it; a new key has the
//      survivors in front of index lo(f) and the new keys above it (counted directly: at most n_prev of them, the work split over the whole
//      workgroup).  Keys are distinct (the node index is part of them; a stale entry equal to a new key is that node's own, and dropped), so the
//      ranks are a permutation.  (Every survivor counting the new keys above it by comparison — n_prev 64-bit compares in each of ~500
//      threads — was 3 of the launch's 14 us, on the dependent chain of every round.)  tests/test_repair_merge_cpu.py restates the arithmetic.
// A clean node of the true top L has at most L - 1 clean and n_prev changed nodes above it in the stale order: stale_L >= n_prev + L entries
// hold every one of them.
//
// Two callers: k_repair (kb_kernels.hip), a launch of its own between two commit kernels, and the selection kernel's launch
// (kb_commit_sel.hip), which carries these workgroups BESIDE its commit workgroup: that one stages everything else first and waits for the tag
// each row leaves in KbRound::lists_ready.  All LDS comes from the caller's dynamic block (a commit launch has no static LDS to spare).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kb_device.h"
#include "kb_eval.hpp"
#include "kb_k1.hpp"

#define KB_REPAIR_THREADS 512u    // the commit kernels' workgroup size (K9_THREADS): the same workgroups serve both launches
#define KB_REPAIR_ENTRIES 1024u   // stale entries a workgroup can merge (stale_L <= n_prev + L <= 256 + 257)
#define KB_REPAIR_FRESH 512u      // decision records of the predecessor (n_prev <= KB_K5_MAX_ROWS = 256)
static_assert(KB_K5_MAX_ROWS <= KB_REPAIR_FRESH / 2u && KB_REPAIR_FRESH <= KB_REPAIR_THREADS, "a thread per decision record of the predecessor, and one left over for the tag");

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

__host__ __device__ inline size_t kb_repair_lds_bytes(uint32_t NP, uint32_t threads) {
  return sizeof(unsigned long long) * (KB_REPAIR_ENTRIES + KB_REPAIR_FRESH) + sizeof(uint32_t) * (2u * (KB_REPAIR_ENTRIES + 1u) + KB_REPAIR_FRESH) +
         sizeof(uint32_t) * (2u * (KB_REPAIR_ENTRIES / 64u) + 4u) + sizeof(uint32_t) * (NP / 32u) + 0u * threads;
}

// One row's list.  Every thread of the workgroup calls it (barriers inside).  Returns false when the row's stale list never arrived: the chain
// word is cleared — the commit kernel of the round then reports KB_REASON_SKIPPED and the host launches the round again on the plain path.
// With r.lists_ready set (the selection kernel's launch) the row's tag is published behind the finished list.
template <uint32_t THREADS>
__device__ __forceinline__ bool kb_repair_row(const KbDev &d, const KbRound &r, unsigned char *smem, const uint32_t row, const uint32_t tid) {
  constexpr uint32_t EPT = KB_REPAIR_ENTRIES / THREADS, WAVES = THREADS / 64u;   // stale entries per thread: entry h * THREADS + tid, h < EPT
  static_assert(KB_REPAIR_ENTRIES % THREADS == 0u && THREADS % 64u == 0u && THREADS >= KB_REPAIR_FRESH, "workgroup shape");
  unsigned long long *stale = reinterpret_cast<unsigned long long *>(smem);                    // [KB_REPAIR_ENTRIES], 0-terminated, best first
  unsigned long long *fresh = stale + KB_REPAIR_ENTRIES;                                       // [KB_REPAIR_FRESH] keys of the predecessor's nodes (0: infeasible / not owned / behind n_prev)
  uint32_t *alive_before = reinterpret_cast<uint32_t *>(fresh + KB_REPAIR_FRESH);              // [KB_REPAIR_ENTRIES + 1] survivors in front of entry i
  uint32_t *marks = alive_before + KB_REPAIR_ENTRIES + 1;                                      // [KB_REPAIR_ENTRIES + 1] new keys whose place in the stale list is index i
  uint32_t *fresh_above = marks + KB_REPAIR_ENTRIES + 1;                                       // [KB_REPAIR_FRESH] new keys above new key j
  uint32_t *s_wtot = fresh_above + KB_REPAIR_FRESH;                                            // [KB_REPAIR_ENTRIES / 64] survivors per wave-sized piece of the list
  uint32_t *s_mtot = s_wtot + KB_REPAIR_ENTRIES / 64u;                                         // [KB_REPAIR_ENTRIES / 64] marks per piece
  uint32_t *s_late = s_mtot + KB_REPAIR_ENTRIES / 64u;                                         // [4]
  uint32_t *bitmap = s_late + 4;                                                               // [NP / 32]
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  if (row == 0 && tid == 0) {   // the round's matrix / arg-max stamps: the candidate launches ran beside the predecessor, this is what the round waits for
    // (stored through, like the lists: inside a commit launch the reader is the commit workgroup's epilogue on another CU, which mirrors the result
    //  block to the host — a plain store could sit in this XCD's L2 while that epilogue reads an earlier round's value; statistics only)
    unsigned long long *st = reinterpret_cast<unsigned long long *>(r.result);
    __hip_atomic_store(&st[KB_OUT_STAMP0], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // [+1] follows when row 0's tag has been seen
  }
  // Latency is what this costs (it sits on the dependent chain of every round): every load that does not depend on another is issued before the
  // first wait.  The predecessor's decision records are final (kernel boundary), so its nodes are fetched while another thread still looks for the tag.
  const uint32_t np = r.n_prev;
  uint32_t node = KB_NONE_U32;
  if (tid < np) node = (uint32_t)(r.prev_dec[tid] & 0xFFFFFFFFull);
  for (uint32_t w = tid; w < d.NP / 32; w += THREADS) bitmap[w] = 0u;
#pragma unroll
  for (uint32_t h = 0; h < EPT; h++) marks[h * THREADS + tid] = 0u;
  if (tid == 0) marks[KB_REPAIR_ENTRIES] = 0u;
  if (tid < KB_REPAIR_FRESH) fresh_above[tid] = 0u;
  if (tid == THREADS - 1u) {   // a thread without a decision record to fetch (n_prev <= 256 < the workgroup): the tag's round trip runs beside that fetch
    // the list was launched (second stream) before this launch (first stream) and had a whole commit kernel's time to finish: the wait is
    // normally over before it starts.  Bounded all the same: a list that never arrives breaks the chain instead of hanging the device.
    uint32_t spins = 0, late = 0;
    while (__hip_atomic_load(&r.ready[row], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != r.ready_tag) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 21)) { late = 1; break; }
    }
    s_late[0] = late;
    if (late && r.chain != nullptr) __hip_atomic_store(r.chain, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (row == 0) __hip_atomic_store(&reinterpret_cast<unsigned long long *>(r.result)[KB_OUT_STAMP0 + 1], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_late[0]) return false;
  // 1 + 2, one round trip for all of it: the predecessor's nodes (a node may have taken several rows: one owner each) and their state; behind
  // the tag, the row's task record and the stale list; then the owners evaluate (K1's own arithmetic)
  bool owner = false;
  K1Node nv[1];
  nv[0].valid = 0;
  if (node != KB_NONE_U32) {
    const uint32_t old = atomicOr(&bitmap[node >> 5], 1u << (node & 31));
    owner = !((old >> (node & 31)) & 1u);
    if (owner) nv[0] = k1_node(d, node);
  }
  const uint32_t Ls = r.stale_L;
  unsigned long long sk[EPT];
#pragma unroll
  for (uint32_t h = 0; h < EPT; h++) { const uint32_t i = h * THREADS + tid; sk[h] = (i < Ls) ? r.stale[(size_t)row * Ls + i] : 0ull; }
  unsigned long long fk = 0ull;
  if (owner) {
    const K1Task tv = k1_uniform(reinterpret_cast<const K1Task *>(r.task_rows)[row]);
    uint32_t res[1];
    eval_row<1>(d, tv, nv, node, r.fit_mode, res);
    if (res[0] >> 16) fk = KB_KEY(res[0] & 0xFFFFu, node);
  }
  if (tid < KB_REPAIR_FRESH) fresh[tid] = fk;
#pragma unroll
  for (uint32_t h = 0; h < EPT; h++) stale[h * THREADS + tid] = sk[h];
  __syncthreads();
  // 3: survivors and their counts per piece of 64 entries; the new keys' places
  bool alive[EPT];
  uint32_t in_wave[EPT];
#pragma unroll
  for (uint32_t h = 0; h < EPT; h++) {
    alive[h] = sk[h] != 0ull && !((bitmap[KB_KEY_NODE(sk[h]) >> 5] >> (KB_KEY_NODE(sk[h]) & 31)) & 1u);
    const unsigned long long bal = __ballot(alive[h]);
    in_wave[h] = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wtot[h * WAVES + wave] = (uint32_t)__popcll(bal);
  }
  uint32_t lo = 0;
  if (fk != 0ull) {
    // stale entries above fk: the list is descending; entries equal to 0 (behind its end) are never above
    uint32_t hi = KB_REPAIR_ENTRIES;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (stale[mid] > fk) lo = mid + 1; else hi = mid;
    }
    atomicAdd(&marks[lo], 1u);
  }
  {   // new keys above a new key, eight per step as four 16-byte LDS reads in flight together (fresh[] reads 0 behind n_prev).  Thread
      // (part, j) compares key j with one part of them: n_prev <= THREADS / PARTS
    auto count_above = [&](unsigned long long key, uint32_t i0, uint32_t i1) {
      uint32_t c = 0;
      for (uint32_t i = i0; i < i1; i += 8) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&fresh[i]), b = *reinterpret_cast<const ulonglong2 *>(&fresh[i + 2]);
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(&fresh[i + 4]), f = *reinterpret_cast<const ulonglong2 *>(&fresh[i + 6]);
        c += (a.x > key ? 1u : 0u) + (a.y > key ? 1u : 0u) + (b.x > key ? 1u : 0u) + (b.y > key ? 1u : 0u) +
             (e.x > key ? 1u : 0u) + (e.y > key ? 1u : 0u) + (f.x > key ? 1u : 0u) + (f.y > key ? 1u : 0u);
      }
      return c;
    };
    constexpr uint32_t PARTS = 2u;
    const uint32_t np8 = (np + 7u) & ~7u;
    if (np <= THREADS / PARTS) {
      const uint32_t j = tid % (THREADS / PARTS), part = tid / (THREADS / PARTS);
      const uint32_t per = (((np8 + PARTS - 1u) / PARTS) + 7u) & ~7u, i0 = part * per, i1 = min(np8, i0 + per);
      const unsigned long long key = fresh[j];
      if (key != 0ull && i0 < i1) {
        const uint32_t c = count_above(key, i0, i1);
        if (c) atomicAdd(&fresh_above[j], c);
      }
    } else if (fk != 0ull) {
      fresh_above[tid] = count_above(fk, 0u, np8);
    }
  }
  __syncthreads();
  uint32_t mscan[EPT];
#pragma unroll
  for (uint32_t h = 0; h < EPT; h++) {
    mscan[h] = wave_incl_scan_u32(marks[h * THREADS + tid]);   // new keys whose place is at or in front of my entry, inside its piece
    if (lane == 63) s_mtot[h * WAVES + wave] = mscan[h];
  }
  __syncthreads();
  uint32_t before[EPT], above[EPT];
#pragma unroll
  for (uint32_t h = 0; h < EPT; h++) {
    uint32_t bsum = in_wave[h], asum = mscan[h];
    for (uint32_t w = 0; w < h * WAVES + wave; w++) { bsum += s_wtot[w]; asum += s_mtot[w]; }
    before[h] = bsum; above[h] = asum;
    alive_before[h * THREADS + tid] = bsum;
  }
  if (tid == THREADS - 1u) alive_before[KB_REPAIR_ENTRIES] = before[EPT - 1u] + (alive[EPT - 1u] ? 1u : 0u);
  __syncthreads();
  const uint32_t K = r.L;
  unsigned long long *out = r.keys + (size_t)row * K;
  // A launch of its own ends in a kernel boundary, which publishes its list.  Inside the commit launch the list goes to the commit workgroup
  // — another CU, usually another XCD — through WRITE-THROUGH stores (agent-scope relaxed atomics) and a tag stored behind them: an agent-scope
  // release fence instead writes the XCD's whole L2 back, from every wave of every row, and the commit workgroup waited 14 us for lists that
  // take 6 (calls 13 - 16)
  const bool through = r.lists_ready != nullptr;
  auto put = [&](uint32_t at, unsigned long long v) {
    if (through) __hip_atomic_store(&out[at], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else out[at] = v;
  };
#pragma unroll
  for (uint32_t h = 0; h < EPT; h++)
    if (alive[h]) {
      const uint32_t rank = before[h] + above[h];
      if (rank < K) put(rank, sk[h]);
    }
  if (fk != 0ull) {
    const uint32_t rank = alive_before[lo] + fresh_above[tid];
    if (rank < K) put(rank, fk);
  }
  // the tail: entries behind the merged list read 0
  uint32_t cnt_fresh = 0;
  for (uint32_t i = lane; i < np; i += 64) cnt_fresh += fresh[i] != 0ull ? 1u : 0u;   // every wave counts for itself (no further barrier)
  for (int off = 32; off > 0; off >>= 1) cnt_fresh += (uint32_t)__shfl_xor((int)cnt_fresh, off);
  const uint32_t total = alive_before[KB_REPAIR_ENTRIES] + cnt_fresh;
  for (uint32_t i = total + tid; i < K; i += THREADS) put(i, 0ull);
  if (through) {   // the round's commit workgroup runs beside this one and waits for the tag (kb_k9.hpp: k9_prologue), then reads the list with agent-scope loads
    // every thread's stores above acknowledged (a barrier's workgroup-scope fence does not wait for them: without this wait the tag overtook the
    // list on small clusters — 10 of the first 57 adversarial cases, call 18), then the barrier, then the tag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&r.lists_ready[row], r.lists_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;
}
