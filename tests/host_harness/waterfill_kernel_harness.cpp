// waterfill_kernel_harness.cpp — TEST INFRASTRUCTURE: kube-batch_amd/csrc/kb_waterfill.hip's kernel TEXT compiled for the host and run by
// 256 real threads with a real barrier, so that what the kernel adds to kb_waterfill.hpp's steps — which lane runs which step, and where the
// barriers stand between them — is executed (and, under ThreadSanitizer, checked for unordered accesses) without a device.  The device
// vocabulary the kernel uses is mapped one to one: threadIdx.x = a thread-local index, __shared__ = one static object for the workgroup,
// __syncthreads() = pthread_barrier_wait over the 256 threads, atomicOr = an atomic OR.  What this cannot show is the compiler's device
// code; the first device run of the launch (round 4's first GPU call, profiles/round4/first_call/) did.
#include <pthread.h>
#include <stdint.h>

#include <thread>
#include <vector>

#include <hip/hip_runtime.h>   // tests/host_harness/hip_mock: __global__ / __device__ / __host__ as empty words

namespace {
struct Idx { uint32_t x; };
thread_local Idx threadIdx;
pthread_barrier_t g_bar;
inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
}  // namespace
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))
#define __launch_bounds__(n)
#define __syncthreads() pthread_barrier_wait(&g_bar)
#define hipLaunchKernelGGL(...) ((void)0)
#define kb_launch_waterfill kbwf_unused_launch

#include "../../kube-batch_amd/csrc/kb_waterfill.hip"

#undef kb_launch_waterfill

extern "C" {
// sizeof / offsets for the ctypes mirror in tests/test_waterfill_kernel_cpu.py
void kbwf_layout(uint32_t *out) {
  out[0] = sizeof(kb::Res); out[1] = sizeof(kb::WfQueue); out[2] = sizeof(kb::WfState); out[3] = KB_WF_THREADS;
  out[4] = offsetof(kb::WfQueue, weight); out[5] = offsetof(kb::WfState, total_weight); out[6] = offsetof(kb::Res, mask);
}
// the kernel, one workgroup of KB_WF_THREADS threads
void kbwf_run_kernel(kb::WfQueue *qs, uint32_t Q, kb::WfState *st, int R) {
  pthread_barrier_init(&g_bar, nullptr, KB_WF_THREADS);
  std::vector<std::thread> th;
  th.reserve(KB_WF_THREADS);
  for (uint32_t t = 0; t < KB_WF_THREADS; t++)
    th.emplace_back([=]() {
      threadIdx.x = t;
      k_waterfill(qs, Q, st, R, nullptr, nullptr);
    });
  for (auto &x : th) x.join();
  pthread_barrier_destroy(&g_bar);
}
// the steps one after the other (what the emulated device runs)
void kbwf_run_sequential(kb::WfQueue *qs, uint32_t Q, kb::WfState *st, int R) { kb::wf_run_sequential(qs, Q, *st, R); }
}
