// order_harness.cpp — test infrastructure: exposes the ENGINE'S OWN host order machine (kube-batch_amd/csrc/kb_order.cpp,
// compiled unchanged with g++) through a tiny C interface, so that the CPU suite can drive its next()/report()/checkpoint()/
// rollback() protocol against the restated reference loop without a GPU (tests/test_host_order_cpu.py).  The device's part
// (which node a task gets) is played by tests/pyref.py.  Nothing here is linked into libkbengine.so.
#include <time.h>
#include "../../kube-batch_amd/csrc/kb_host.hpp"

using namespace kb;

static int g_force_journal = -1;
struct HH {
  HostSession hs;
  Policy pol;
  OrderMachine om;
};

extern "C" {

HH *hh_create(int R, uint32_t T, uint32_t J, uint32_t Q,
              const double *t_res_rows, const uint32_t *t_resmask, const int32_t *t_prio, const int64_t *t_creation,
              const uint8_t *t_status, const uint8_t *t_res_empty,
              const uint32_t *job_begin, const uint32_t *job_queue, const int32_t *job_min, const int32_t *job_prio, const int64_t *job_creation,
              const int64_t *queue_creation,
              const double *total_v, uint32_t total_mask, const double *deserved_v, const uint32_t *deserved_mask,
              const double *job_alloc, const double *job_share, const double *queue_alloc, const double *queue_share, const int32_t *job_ready,
              const uint8_t *job_chain, int n_chain, int queue_order_proportion, int task_order_priority, int gang_job_ready,
              int has_gang, int has_drf, int has_proportion) {
  HH *h = new HH();
  HostSession &hs = h->hs;
  hs.R = R; hs.T = T; hs.J = J; hs.Q = Q;
  hs.t_res_rows.assign(t_res_rows, t_res_rows + (size_t)T * R);
  hs.t_resmask.assign(t_resmask, t_resmask + T);
  hs.t_prio.assign(t_prio, t_prio + T);
  hs.t_creation.assign(t_creation, t_creation + T);
  hs.t_status.assign(t_status, t_status + T);
  hs.t_res_empty.assign(t_res_empty, t_res_empty + T);
  hs.job_begin.assign(job_begin, job_begin + J + 1);
  hs.job_queue.assign(job_queue, job_queue + J);
  hs.job_min.assign(job_min, job_min + J);
  hs.job_prio.assign(job_prio, job_prio + J);
  hs.job_creation.assign(job_creation, job_creation + J);
  hs.queue_creation.assign(queue_creation, queue_creation + Q);
  for (int d = 0; d < R; d++) hs.total.v[d] = total_v[d];
  hs.total.mask = total_mask;
  hs.deserved.assign(Q, Res());
  for (uint32_t q = 0; q < Q; q++) {
    for (int d = 0; d < R; d++) hs.deserved[q].v[d] = deserved_v[(size_t)q * R + d];
    hs.deserved[q].mask = deserved_mask[q];
  }
  hs.job_alloc.assign(job_alloc, job_alloc + (size_t)J * R);
  hs.job_share.assign(job_share, job_share + J);
  hs.queue_alloc.assign(queue_alloc, queue_alloc + (size_t)Q * R);
  hs.queue_share.assign(queue_share, queue_share + Q);
  hs.job_ready.assign(job_ready, job_ready + J);
  Policy &p = h->pol;
  p.job_chain.assign(job_chain, job_chain + n_chain);
  p.queue_order_proportion = queue_order_proportion;
  p.task_order_priority = task_order_priority;
  p.gang_job_ready = gang_job_ready;
  p.has_gang = has_gang; p.has_drf = has_drf; p.has_proportion = has_proportion;
  h->om.force_journal = g_force_journal;
  h->om.init_allocate(&h->hs, &h->pol);
  return h;
}
void hh_destroy(HH *h) { delete h; }
// the allocate action's host loop with every round confirmed (ActionRun::plan / plan_ahead / promote, every row Allocated): seconds for the whole
// action, rows handed out in *rows — what the order machine costs per round when nothing breaks (scripts/time_order_machine.py)
double hh_bench(HH *h, uint32_t W, uint64_t *rows) {
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  OrderMachine &om = h->om;
  uint32_t t = 0;
  uint64_t total = 0;
  auto window = [&]() { uint32_t n = 0; while (n < W && om.next(t)) { om.report(Outcome::Allocated); n++; } total += n; return n; };
  om.checkpoint();
  uint32_t n = window();
  while (n) { om.push_checkpoint(); n = window(); om.pop_commit(); }
  clock_gettime(CLOCK_MONOTONIC, &b);
  if (rows) *rows = total;
  return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
// how the machines created from now on keep their heap arrays across roll-back points: -1 by size, 0 copies, 1 journals (OrderMachine::force_journal)
void hh_set_journal(int mode) { g_force_journal = mode; }
int hh_next(HH *h, uint32_t *task) { return h->om.next(*task) ? 1 : 0; }
void hh_report(HH *h, int outcome) { h->om.report(outcome == 0 ? Outcome::Allocated : outcome == 1 ? Outcome::Pipelined : Outcome::NoFeasibleNode); }
void hh_checkpoint(HH *h) { h->om.checkpoint(); }
void hh_push_checkpoint(HH *h) { h->om.push_checkpoint(); }
void hh_pop_commit(HH *h) { h->om.pop_commit(); }
void hh_rollback(HH *h) { h->om.rollback(); }
void hh_rollback_last_pop(HH *h) { h->om.rollback_last_pop(); }
uint64_t hh_steps(const HH *h) { return h->om.steps; }
void hh_state(const HH *h, double *jshare, double *qshare, int32_t *ready, double *jalloc, double *qalloc) {
  const OrderMachine &om = h->om;
  std::memcpy(jshare, om.jshare.data(), sizeof(double) * om.jshare.size());
  std::memcpy(qshare, om.qshare.data(), sizeof(double) * om.qshare.size());
  std::memcpy(ready, om.ready.data(), sizeof(int32_t) * om.ready.size());
  std::memcpy(jalloc, om.jalloc.data(), sizeof(double) * om.jalloc.size());
  std::memcpy(qalloc, om.qalloc.data(), sizeof(double) * om.qalloc.size());
}

}  // extern "C"
