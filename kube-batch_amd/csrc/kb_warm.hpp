// kb_warm.hpp — L2 warm-up of the commit kernels (kb_commit.hip, kb_commit_sel.hip), and the launchers' dynamic-LDS attribute.
//
// The commit loop is latency-bound and its workgroup starts on a cold L2 at every kernel boundary, so the prologue touches every
// 128-byte line of the node state once.  One CU pulls that at ~200 GB/s: 5.9 us per round at 10k nodes and
// 37 us per round at 50k nodes — 16 % of the 1M x 50k cycle (profiles/round3/call9).  The L2 belongs to the XCD, not to the CU, so
// the launch now carries HELPER workgroups whose only job is to touch a slice of those lines and exit: workgroup b lands on XCD b % 8
// (observed placement, MI355X_MICROARCH.md; used for speed only — a helper on another XCD warms the wrong L2 and nothing else), so
// workgroups 8, 16, ... share workgroup 0's L2 while running on other CUs of that XCD.  Nobody waits for them: a line that has not
// arrived yet is an ordinary miss for the commit workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kb_device.h"

#define KB_WARM_HELPERS 8u                         // helper workgroups on the commit workgroup's XCD
#define KB_WARM_GRID (1u + 8u * KB_WARM_HELPERS)   // workgroup 0 commits; 8, 16, ..., 8 * KB_WARM_HELPERS warm; the rest exit at once

// touch lines [l0, l1) of the ten 8-byte node arrays, the matching half range of the three 4-byte ones, and the same line range of the
// candidate lists; returns a value that depends on every load (the caller stores it somewhere dead so the loads stay)
__device__ __forceinline__ unsigned long long kb_warm_lines(const KbDev &d, const unsigned long long *keys, size_t klines, uint32_t l0, uint32_t l1,
                                                            uint32_t tid, uint32_t threads) {
  unsigned long long acc = 0;
  const unsigned long long *arrs[10] = {
      reinterpret_cast<const unsigned long long *>(d.idle), reinterpret_cast<const unsigned long long *>(d.idle + d.NP),
      reinterpret_cast<const unsigned long long *>(d.rel), reinterpret_cast<const unsigned long long *>(d.rel + d.NP),
      reinterpret_cast<const unsigned long long *>(d.inv_acpu), reinterpret_cast<const unsigned long long *>(d.inv_amem),
      reinterpret_cast<const unsigned long long *>(d.acpu), reinterpret_cast<const unsigned long long *>(d.amem),
      reinterpret_cast<const unsigned long long *>(d.nzc), reinterpret_cast<const unsigned long long *>(d.nzm)};
  const uint32_t *arr4[4] = {d.ncls, reinterpret_cast<const uint32_t *>(d.maxpods), reinterpret_cast<const uint32_t *>(d.podcnt), d.nmask};
  for (uint32_t lb = l0; lb < l1; lb += threads) {   // all loads of a pass in flight together
    const uint32_t l = lb + tid;
    unsigned long long v[11];
    uint32_t w[4];
#pragma unroll
    for (int f = 0; f < 10; f++) v[f] = (l < l1) ? arrs[f][(size_t)l * 16] : 0ull;
    v[10] = (l < l1 && l < klines) ? keys[(size_t)l * 16] : 0ull;
#pragma unroll
    for (int f = 0; f < 4; f++) w[f] = (l < l1 && (l & 1u) == 0u) ? arr4[f][(size_t)l * 16] : 0u;   // 32 4-byte elements per line: every other `l`
#pragma unroll
    for (int f = 0; f < 11; f++) acc += v[f];
#pragma unroll
    for (int f = 0; f < 4; f++) acc += w[f];
  }
  return acc;
}

// host side: let `fn` use up to `bytes` of dynamic LDS.  The attribute belongs to the (function, device) pair: each launcher remembers it per
// device ordinal (`set_on`, a static of its own), not per process — a second engine on another GPU of the same process needs it too
static inline void kb_allow_lds(const void *fn, int bytes, bool (&set_on)[64]) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && set_on[dev]) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (dev >= 0 && dev < 64) set_on[dev] = true;
}
