// kb_waterfill.hpp — proportion's OnSessionOpen water-fill (plugins/proportion/proportion.go:101-154) cut into the steps a device launch
// can run side by side, as functions both sides compile: kb_waterfill.hip's kernel calls them from its lanes, the emulated device
// (tests/host_harness/device_emu.cpp) calls them from plain loops, and the host's own loop (kb_session.cpp, unchanged, the default) is what
// both are held to.  KB_DEVICE_WATERFILL=1 (off until its first device run) makes kb_session_load take `deserved` from the launch.
//
// One pass of the reference's loop:
//   totalWeight = sum of the weights of the queues that have not met their request             -> wf_weight   (integer: any order)
//   per such queue: deserved += remaining * weight / totalWeight; if request < deserved: deserved = min(deserved, request), met;
//                   (increased, decreased) = deserved.Diff(old deserved)                        -> wf_queue    (queues are independent)
//   increasedDeserved / decreasedDeserved = the sums of those, queue after queue                -> wf_reduce_dim (float64 sums are ordered: one
//                                                                                                  lane per resource dimension walks the queues
//                                                                                                  in ascending order; dimensions are independent)
//   remaining = remaining - increasedDeserved + decreasedDeserved; stop when it is empty        -> wf_tail
// The iteration order over queues is the one the host loop and the oracle use (ascending queue index; the reference ranges over a map).
#pragma once
#include "kb_res.hpp"

namespace kb {

struct WfQueue {   // one per queue, in device memory
  Res deserved, request, inc, dec;
  int32_t weight;
  uint32_t has_attr, meet, active;   // active: took part in the current pass (its inc / dec count)
};
struct WfState {   // one per launch, in device memory
  Res remaining, increased, decreased;
  int32_t total_weight;
  uint32_t stop;             // the loop is over
  uint32_t share_at_open;    // 0: the first pass found no weight at all (proportion.go:113-116: no updateShare ran)
  uint32_t underflow;        // remaining.Sub(increasedDeserved) would panic in the reference
  uint32_t passes;
};

KB_HD inline int32_t wf_weight(const WfQueue *qs, uint32_t Q) {
  int32_t w = 0;
  for (uint32_t q = 0; q < Q; q++)
    if (qs[q].has_attr && !qs[q].meet) w = (int32_t)((uint32_t)w + (uint32_t)qs[q].weight);   // Go's int32 wraps
  return w;
}
KB_HD inline void wf_queue(WfQueue &a, const WfState &st, int R) {   // proportion.go:121-141
  a.active = (a.has_attr && !a.meet) ? 1u : 0u;
  if (!a.active) return;
  const Res old = a.deserved;
  Res part = st.remaining;
  res_multi(part, (double)a.weight / (double)st.total_weight, R);
  res_add(a.deserved, part, R);
  if (res_less(a.request, a.deserved, R)) {
    a.deserved = helpers_min(a.deserved, a.request, R);
    a.meet = 1;
  }
  res_diff(a.deserved, old, a.inc, a.dec, R);
}
// dimension d of increasedDeserved.Add(inc) / decreasedDeserved.Add(dec) over the pass's queues, in queue order (resource_info.go:128-140:
// cpu and memory always, a scalar where the operand has the key — which also creates it in the sum)
KB_HD inline void wf_reduce_dim(const WfQueue *qs, uint32_t Q, int d, double &inc_v, bool &inc_has, double &dec_v, bool &dec_has) {
  inc_v = 0.0; dec_v = 0.0; inc_has = false; dec_has = false;
  for (uint32_t q = 0; q < Q; q++) {
    if (!qs[q].active) continue;
    if (d < 2 || qs[q].inc.has(d)) { inc_v += qs[q].inc.v[d]; inc_has = true; }
    if (d < 2 || qs[q].dec.has(d)) { dec_v += qs[q].dec.v[d]; dec_has = true; }
  }
}
KB_HD inline void wf_tail(WfState &st, int R) {   // proportion.go:143-153
  st.passes += 1;
  if (!res_sub(st.remaining, st.increased, R)) { st.underflow = 1; st.stop = 1; return; }
  res_add(st.remaining, st.decreased, R);
  if (res_is_empty(st.remaining, R)) st.stop = 1;
}

// the whole loop, one step after the other (the emulated device; the kernel runs the same steps with its lanes)
inline void wf_run_sequential(WfQueue *qs, uint32_t Q, WfState &st, int R) {
  st.stop = 0; st.share_at_open = 1; st.underflow = 0; st.passes = 0;
  for (bool first = true; !st.stop; first = false) {
    st.total_weight = wf_weight(qs, Q);
    if (st.total_weight == 0) {
      if (first) st.share_at_open = 0;
      break;
    }
    for (uint32_t q = 0; q < Q; q++) wf_queue(qs[q], st, R);
    st.increased = Res();
    st.decreased = Res();
    for (int d = 0; d < R; d++) {
      bool ih, dh;
      wf_reduce_dim(qs, Q, d, st.increased.v[d], ih, st.decreased.v[d], dh);
      if (d >= 2 && ih) st.increased.setk(d);
      if (d >= 2 && dh) st.decreased.setk(d);
    }
    wf_tail(st, R);
  }
}

}  // namespace kb

// kb_waterfill.hip (the emulated device: tests/host_harness/device_emu.cpp): the loop over `Q` queue records and one state record, both in
// device memory; st->remaining holds the session's total on entry; on return the records hold deserved / meet and st the flags above, and
// des [R][Q] / desmask [Q] (device memory, may be null) `deserved` in the layout k_finalize_queues reads
void kb_launch_waterfill(kb::WfQueue *qs, uint32_t Q, kb::WfState *st, int R, double *des, uint32_t *desmask, void *stream);
