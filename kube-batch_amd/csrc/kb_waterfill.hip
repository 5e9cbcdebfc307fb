// kb_waterfill.hip — proportion's OnSessionOpen water-fill on the device (second half; KB_DEVICE_WATERFILL=1, off
// until its first device run).  One workgroup: a pass's queues are independent (one lane each, kb_waterfill.hpp: wf_queue), the pass's two
// float64 sums are ordered over the queues but independent per resource dimension (one lane per dimension walks the queues in ascending
// order, wf_reduce_dim), the bookkeeping between passes is one lane's (wf_weight, wf_tail).  Q is tens to a few hundred, the loop runs a
// handful of passes: this is load-time work of tens of microseconds (see the note at the kernel), on the device only so that `deserved` is born where k_finalize_queues
// reads it.  The arithmetic is kb_res.hpp's — the text the host loop in kb_session.cpp runs —, compiled for the device with the same
// -ffp-contract=off; what this file adds is the order of the steps and the barriers between them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kb_waterfill.hpp"

using namespace kb;

#define KB_WF_THREADS 256
#define KB_WF_CHUNK 64   // queues staged in LDS at a time for the ordered sums

// des / desmask (optional): `deserved` once more, in the layout k_finalize_queues reads ([R][Q] values, absent scalar keys as 0; [Q] key masks)
//
// Round 5: the first version walked the queue records in GLOBAL memory from single lanes — lane 0 over Q records for the total weight, one lane
// per dimension over Q records for the two ordered sums: Q dependent loads of ~0.5 us each, per pass: 408 us per launch at 128 queues
// (profiles/round5/rocprofv3_kernel_stats.csv), i.e. 0.4 of the 3 ms of a kb_session_load.  Now every thread loads its share side by side:
// the total weight is an integer sum (Go's int32 wraps, any order), and the sums' operands are staged in LDS 64 queues at a time, from where
// the dimension's lane adds them in queue order — the same additions in the same order (kb_waterfill.hpp: wf_reduce_dim is what both are held to).
__global__ void __launch_bounds__(KB_WF_THREADS) k_waterfill(WfQueue *qs, uint32_t Q, WfState *gst, int R, double *des, uint32_t *desmask) {
  __shared__ __align__(8) unsigned char s_raw[sizeof(WfState)];   // a Res has a constructor: no __shared__ object of it
  __shared__ double s_inc[KB_MAX_RES * KB_WF_CHUNK], s_dec[KB_MAX_RES * KB_WF_CHUNK];   // [dimension][queue of the chunk]
  __shared__ uint32_t s_act[KB_WF_CHUNK], s_imask[KB_WF_CHUNK], s_dmask[KB_WF_CHUNK];
  __shared__ unsigned s_weight;
  WfState &S = *reinterpret_cast<WfState *>(s_raw);
  const uint32_t tid = threadIdx.x;
  if (tid == 0) {
    S.remaining = gst->remaining;
    S.stop = 0; S.share_at_open = 1; S.underflow = 0; S.passes = 0; S.total_weight = 0;
  }
  __syncthreads();
  bool stop0 = false;   // lane 0's own copy of S.stop as the previous pass's wf_tail left it (the other lanes read S.stop behind the barrier only)
  for (bool first = true;; first = false) {
    if (tid == 0) s_weight = 0u;
    __syncthreads();
    {   // wf_weight, every thread its queues (unsigned addition: wraps exactly like the reference's int32)
      unsigned w = 0u;
      for (uint32_t q = tid; q < Q; q += KB_WF_THREADS)
        if (qs[q].has_attr && !qs[q].meet) w += (unsigned)qs[q].weight;
      if (w) atomicAdd(&s_weight, w);
    }
    __syncthreads();
    if (tid == 0 && !stop0) {
      S.total_weight = (int32_t)s_weight;
      if (S.total_weight == 0) {
        if (first) S.share_at_open = 0;
        S.stop = 1;
      }
      S.increased = Res();
      S.decreased = Res();
    }
    __syncthreads();
    if (S.stop) break;   // the same value for every lane: the next write to it (wf_tail) is several barriers away
    for (uint32_t q = tid; q < Q; q += KB_WF_THREADS) wf_queue(qs[q], S, R);
    __syncthreads();   // the queues' inc / dec (global memory, written by other waves of this workgroup) are visible behind it
    {   // wf_reduce_dim: lane d < R owns dimension d; the operands come through LDS a chunk of queues at a time
      double iv = 0.0, dv = 0.0;
      bool ih = false, dh = false;
      for (uint32_t c0 = 0; c0 < Q; c0 += KB_WF_CHUNK) {
        const uint32_t i = tid & (KB_WF_CHUNK - 1u), q = c0 + i;
        if (tid < KB_WF_CHUNK) {
          s_act[i] = q < Q ? qs[q].active : 0u;
          s_imask[i] = q < Q ? qs[q].inc.mask : 0u;
          s_dmask[i] = q < Q ? qs[q].dec.mask : 0u;
        }
        for (int d = (int)(tid / KB_WF_CHUNK); d < R; d += KB_WF_THREADS / KB_WF_CHUNK) {
          s_inc[d * KB_WF_CHUNK + i] = q < Q ? qs[q].inc.v[d] : 0.0;
          s_dec[d * KB_WF_CHUNK + i] = q < Q ? qs[q].dec.v[d] : 0.0;
        }
        __syncthreads();
        if (tid < (uint32_t)R) {
          const int d = (int)tid;
          for (uint32_t k = 0; k < KB_WF_CHUNK; k++) {
            if (!s_act[k]) continue;
            if (d < 2 || ((s_imask[k] >> (d - 2)) & 1u)) { iv += s_inc[d * KB_WF_CHUNK + k]; ih = true; }
            if (d < 2 || ((s_dmask[k] >> (d - 2)) & 1u)) { dv += s_dec[d * KB_WF_CHUNK + k]; dh = true; }
          }
        }
        __syncthreads();   // the chunk is consumed before the next one overwrites it
      }
      if (tid < (uint32_t)R) {
        S.increased.v[tid] = iv;
        S.decreased.v[tid] = dv;
        if (tid >= 2 && ih) atomicOr(&S.increased.mask, 1u << (tid - 2));
        if (tid >= 2 && dh) atomicOr(&S.decreased.mask, 1u << (tid - 2));
      }
    }
    __syncthreads();
    if (tid == 0) {
      wf_tail(S, R);
      stop0 = S.stop != 0;
    }
    __syncthreads();
  }
  if (des)   // behind the loop's last barrier: every queue record is final
    for (uint32_t q = tid; q < Q; q += KB_WF_THREADS) {
      desmask[q] = qs[q].deserved.mask;
      for (int d = 0; d < R; d++) des[(size_t)d * Q + q] = qs[q].deserved.get(d);
    }
  if (tid == 0) {
    gst->remaining = S.remaining;
    gst->total_weight = S.total_weight;
    gst->stop = S.stop; gst->share_at_open = S.share_at_open; gst->underflow = S.underflow; gst->passes = S.passes;
  }
}

void kb_launch_waterfill(WfQueue *qs, uint32_t Q, WfState *st, int R, double *des, uint32_t *desmask, void *stream) {
  hipLaunchKernelGGL(k_waterfill, dim3(1), dim3(KB_WF_THREADS), 0, (hipStream_t)stream, qs, Q, st, R, des, desmask);
}
