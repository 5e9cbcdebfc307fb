// kb_waterfill.hip — proportion's OnSessionOpen water-fill on the device (second half; KB_DEVICE_WATERFILL=1, off
// until its first device run).  One workgroup: a pass's queues are independent (one lane each, kb_waterfill.hpp: wf_queue), the pass's two
// float64 sums are ordered over the queues but independent per resource dimension (one lane per dimension walks the queues in ascending
// order, wf_reduce_dim), the bookkeeping between passes is one lane's (wf_weight, wf_tail).  Q is tens to a few hundred, the loop runs a
// handful of passes: this is load-time work of a few microseconds, on the device only so that `deserved` is born where k_finalize_queues
// reads it.  The arithmetic is kb_res.hpp's — the text the host loop in kb_session.cpp runs —, compiled for the device with the same
// -ffp-contract=off; what this file adds is the order of the steps and the barriers between them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kb_waterfill.hpp"

using namespace kb;

#define KB_WF_THREADS 256

// des / desmask (optional): `deserved` once more, in the layout k_finalize_queues reads ([R][Q] values, absent scalar keys as 0; [Q] key masks)
__global__ void __launch_bounds__(KB_WF_THREADS) k_waterfill(WfQueue *qs, uint32_t Q, WfState *gst, int R, double *des, uint32_t *desmask) {
  __shared__ __align__(8) unsigned char s_raw[sizeof(WfState)];   // a Res has a constructor: no __shared__ object of it
  WfState &S = *reinterpret_cast<WfState *>(s_raw);
  const uint32_t tid = threadIdx.x;
  if (tid == 0) {
    S.remaining = gst->remaining;
    S.stop = 0; S.share_at_open = 1; S.underflow = 0; S.passes = 0; S.total_weight = 0;
  }
  __syncthreads();
  bool stop0 = false;   // lane 0's own copy of S.stop as the previous pass's wf_tail left it (the other lanes read S.stop behind the barrier only)
  for (bool first = true;; first = false) {
    if (tid == 0 && !stop0) {
      S.total_weight = wf_weight(qs, Q);
      if (S.total_weight == 0) {
        if (first) S.share_at_open = 0;
        S.stop = 1;
      }
      S.increased = Res();
      S.decreased = Res();
    }
    __syncthreads();
    if (S.stop) break;   // the same value for every lane: the next write to it (wf_tail) is two barriers away
    for (uint32_t q = tid; q < Q; q += KB_WF_THREADS) wf_queue(qs[q], S, R);
    __syncthreads();   // the queues' inc / dec (global memory, written by other waves of this workgroup) are visible behind it
    if (tid < (uint32_t)R) {
      double iv, dv;
      bool ih, dh;
      wf_reduce_dim(qs, Q, (int)tid, iv, ih, dv, dh);
      S.increased.v[tid] = iv;
      S.decreased.v[tid] = dv;
      if (tid >= 2 && ih) atomicOr(&S.increased.mask, 1u << (tid - 2));
      if (tid >= 2 && dh) atomicOr(&S.decreased.mask, 1u << (tid - 2));
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
