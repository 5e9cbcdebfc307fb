// kb_repair.hpp — the repair of an overlapped round's candidate lists (round 3) as a device function, for the batch commit kernel's folded
// variant (kb_commit_batch.hip, KB_FOLD_REPAIR=1): workgroups of the commit launch itself run it, beside the commit workgroup's prologue.
// kb_kernels.hip's k_repair — the launch of its own between two commit kernels, the default — is the same text as a kernel.  It stays a
// separate copy until the folded variant has run on a device: as a wrapper around this function it compiles to different code (6572 instead
// of 5820 bytes, other unrolling), and the default path ships only code that has passed the GPU suite.  Change one, change both.
#pragma once
#include "kb_k1.hpp"

// ------------------------------------------------------------------------------------------------------------
// k_repair: the candidate lists of an OVERLAPPED round (KbRound::ready).  Its matrix and arg-max launches ran on the second stream while
// the predecessor round's commit kernel was still changing nodes: their lists are exact for every node the predecessor left alone and
// arbitrary for the nodes it changed.  One workgroup per matrix row, behind the predecessor's commit on the first stream:
//   1. the predecessor's nodes (its decision records; a node may have taken several rows) -> a bitmap in LDS, each node owned by one thread;
//   2. the owner evaluates the row's shape against the node's state as the predecessor LEFT it (eval_row<1>, K1's own arithmetic);
//   3. the stale list without the predecessor's nodes, merged with the new keys by rank: survivors keep their order (block scan of the
//      survivor flags) and count the new keys above them; a new key counts the survivors above it (binary search + the scan) and the new
//      keys above it.  Keys are distinct (the node index is part of them), so the ranks are a permutation.
// A clean node of the true top L has at most L - 1 clean and n_prev changed nodes above it in the stale order: stale_L >= n_prev + L entries
// hold every one of them.
// ------------------------------------------------------------------------------------------------------------
#define KB_REPAIR_THREADS 1024
// dynamic LDS a workgroup needs for it
__host__ __device__ inline size_t kb_repair_smem_bytes(uint32_t NP) {
  return sizeof(unsigned long long) * (KB_REPAIR_THREADS + KB_K5_MAX_WINDOW) + sizeof(uint32_t) * (KB_REPAIR_THREADS + 1) + sizeof(uint32_t) * (NP / 32) +
         sizeof(uint32_t) * (KB_REPAIR_THREADS / 64 + 1);   // no static LDS: the commit launch's dynamic block may be the whole 160 KiB
}
// One matrix row, by a workgroup of KB_REPAIR_THREADS threads with kb_repair_smem_bytes of LDS at kr_smem (16-byte aligned).  When the
// row's list never arrives the chain word is cleared and nothing is written.  The caller has checked the chain.
__device__ __forceinline__ void kb_repair_row(const KbDev &d, const KbRound &r, const uint32_t row, unsigned char *kr_smem) {
  unsigned long long *stale = reinterpret_cast<unsigned long long *>(kr_smem);                 // [KB_REPAIR_THREADS]
  unsigned long long *fresh = stale + KB_REPAIR_THREADS;                                       // [n_prev] keys of the predecessor's nodes (0: infeasible / not owned)
  uint32_t *alive_before = reinterpret_cast<uint32_t *>(fresh + KB_K5_MAX_WINDOW);             // [KB_REPAIR_THREADS + 1] survivors in front of entry i
  uint32_t *bitmap = alive_before + KB_REPAIR_THREADS + 1;                                     // [NP / 32]
  uint32_t *s_wtot = bitmap + d.NP / 32;                                                       // [KB_REPAIR_THREADS / 64] survivors per wave
  uint32_t &s_late = s_wtot[KB_REPAIR_THREADS / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (row == 0 && tid == 0) {   // the round's matrix / arg-max stamps: the candidate launches ran beside the predecessor, this is what the round waits for
    unsigned long long *st = reinterpret_cast<unsigned long long *>(r.result);
    st[KB_OUT_STAMP0] = wall_clock64();   // [+1] follows when row 0's tag has been seen: "matrix" time of such a round = what it waited for its lists
  }
  // Latency is what this launch costs (it sits between two commit kernels): every load that does not depend on another is issued before the
  // first wait.  The predecessor's decision records are final (kernel boundary), so its nodes are fetched while thread 0 still looks for the tag.
  const uint32_t np = r.n_prev;
  uint32_t node = KB_NONE_U32;
  if (tid < np) node = (uint32_t)(r.prev_dec[tid] & 0xFFFFFFFFull);
  for (uint32_t w = tid; w < d.NP / 32; w += KB_REPAIR_THREADS) bitmap[w] = 0u;
  if (tid == KB_REPAIR_THREADS - 1) {   // a thread without a decision record to fetch (n_prev <= the window < the workgroup): the tag's round trip runs beside that fetch
    // the list was launched (second stream) before this kernel (first stream) and had a whole commit kernel's time to finish: the wait is
    // normally over before it starts.  Bounded all the same: a list that never arrives breaks the chain — the commit kernel behind this one
    // then skips the round and the host launches it again on the plain path — instead of hanging the device.
    uint32_t spins = 0, late = 0;
    while (__hip_atomic_load(&r.ready[row], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != r.ready_tag) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 21)) { late = 1; break; }
    }
    s_late = late;
    if (late && r.chain != nullptr) *r.chain = 0u;
    if (row == 0) reinterpret_cast<unsigned long long *>(r.result)[KB_OUT_STAMP0 + 1] = wall_clock64();
  }
  __syncthreads();
  if (s_late) return;
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
  const unsigned long long sk = (tid < Ls) ? r.stale[(size_t)row * Ls + tid] : 0ull;
  unsigned long long fk = 0ull;
  if (owner) {
    const K1Task tv = k1_uniform(reinterpret_cast<const K1Task *>(r.task_rows)[row]);
    uint32_t res[1];
    eval_row<1>(d, tv, nv, node, r.fit_mode, res);
    if (res[0] >> 16) fk = KB_KEY(res[0] & 0xFFFFu, node);
  }
  if (tid < KB_K5_MAX_WINDOW) fresh[tid] = fk;
  stale[tid] = sk;   // 0-terminated, best first
  __syncthreads();
  // 3: survivors and their prefix counts
  const bool alive = sk != 0ull && !((bitmap[KB_KEY_NODE(sk) >> 5] >> (KB_KEY_NODE(sk) & 31)) & 1u);
  const unsigned long long bal = __ballot(alive);
  const uint32_t in_wave = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) s_wtot[wave] = (uint32_t)__popcll(bal);
  __syncthreads();
  uint32_t before = in_wave;
  for (uint32_t w = 0; w < wave; w++) before += s_wtot[w];
  alive_before[tid] = before;
  if (tid == KB_REPAIR_THREADS - 1) alive_before[KB_REPAIR_THREADS] = before + (alive ? 1u : 0u);
  __syncthreads();
  const uint32_t K = r.L;
  unsigned long long *out = r.keys + (size_t)row * K;
  // new keys above a key: eight per step as four 16-byte LDS reads in flight together (one 8-byte read per step, each waited for, made this
  // launch take 15 us; fresh[] reads 0 behind n_prev: every thread stored its key or 0)
  const uint32_t np8 = (np + 7u) & ~7u;
  auto fresh_above = [&](unsigned long long key) {
    uint32_t c = 0;
    for (uint32_t i = 0; i < np8; i += 8) {
      const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&fresh[i]), b = *reinterpret_cast<const ulonglong2 *>(&fresh[i + 2]);
      const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(&fresh[i + 4]), f = *reinterpret_cast<const ulonglong2 *>(&fresh[i + 6]);
      c += (a.x > key ? 1u : 0u) + (a.y > key ? 1u : 0u) + (b.x > key ? 1u : 0u) + (b.y > key ? 1u : 0u) +
           (e.x > key ? 1u : 0u) + (e.y > key ? 1u : 0u) + (f.x > key ? 1u : 0u) + (f.y > key ? 1u : 0u);
    }
    return c;
  };
  if (alive) {
    const uint32_t rank = before + fresh_above(sk);
    if (rank < K) out[rank] = sk;
  }
  if (fk != 0ull) {
    // stale entries above fk: the list is descending; entries equal to 0 (behind its end) are never above
    uint32_t lo = 0, hi = KB_REPAIR_THREADS;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (stale[mid] > fk) lo = mid + 1; else hi = mid;
    }
    const uint32_t rank = alive_before[lo] + fresh_above(fk);
    if (rank < K) out[rank] = fk;
  }
  // the tail: entries behind the merged list read 0
  uint32_t cnt_fresh = 0;
  for (uint32_t i = lane; i < np; i += 64) cnt_fresh += fresh[i] != 0ull ? 1u : 0u;   // every wave counts for itself (no further barrier)
  for (int off = 32; off > 0; off >>= 1) cnt_fresh += (uint32_t)__shfl_xor((int)cnt_fresh, off);
  const uint32_t total = alive_before[KB_REPAIR_THREADS] + cnt_fresh;
  for (uint32_t i = total + tid; i < K; i += KB_REPAIR_THREADS) out[i] = 0ull;
}
