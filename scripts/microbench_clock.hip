// What clock does a single busy workgroup get?  Compares s_memtime ticks and a fixed dependent-instruction count with
// wall time (HIP events), with and without "heater" workgroups keeping the other CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k(unsigned long long *out, volatile int *flag, int iters, int heaters) {
  if (blockIdx.x == 0) {
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long c0 = wall_clock64();
    float v = threadIdx.x;
    for (int i = 0; i < iters; i++) {   // 8 dependent v_fma per iteration
      v = fmaf(v, 1.0000001f, 0.5f); v = fmaf(v, 1.0000001f, 0.5f); v = fmaf(v, 1.0000001f, 0.5f); v = fmaf(v, 1.0000001f, 0.5f);
      v = fmaf(v, 1.0000001f, 0.5f); v = fmaf(v, 1.0000001f, 0.5f); v = fmaf(v, 1.0000001f, 0.5f); v = fmaf(v, 1.0000001f, 0.5f);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long c1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; out[2] = (unsigned long long)v; }
    __threadfence();
    if (threadIdx.x == 0) *flag = 1;
  } else if (heaters) {
    float v = threadIdx.x;
    while (*flag == 0) { for (int i = 0; i < 256; i++) v = fmaf(v, 1.0000001f, 0.5f); }
    if (v == 12345.f) out[8] = 1;
  }
}

int main() {
  unsigned long long *out; int *flag;
  CK(hipMalloc(&out, 256)); CK(hipMalloc(&flag, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 2000000;
  for (int heaters = 0; heaters < 2; heaters++) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(flag, 0, 4));
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k, dim3(heaters ? 256 : 1), dim3(512), 0, 0, out, flag, iters, heaters);
      CK(hipEventRecord(b));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      unsigned long long r[3]; CK(hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost));
      printf("heaters=%d rep=%d: %.3f ms wall; s_memtime ticks %llu (%.1f MHz); wall_clock64 ticks %llu (%.1f MHz); dependent fma: %.2f ns each\n",
             heaters, rep, ms, r[0], r[0] / (ms * 1e3), r[1], r[1] / (ms * 1e3), ms * 1e6 / (8.0 * iters));
    }
  }
  return 0;
}
