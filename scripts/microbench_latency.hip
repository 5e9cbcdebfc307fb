// Primitive latencies seen by ONE workgroup of 512 threads on an otherwise idle MI355X (the situation of the
// sequential commit kernel).  hipcc --offload-arch=gfx950 -O3 scripts/microbench_latency.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

__global__ void k(unsigned long long *out, const unsigned int *chain, const double *gd, unsigned int n_chain, int iters) {
  __shared__ unsigned int lds[4096];
  __shared__ unsigned long long red[8];
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (unsigned i = tid; i < 4096; i += blockDim.x) lds[i] = (i * 97 + 13) & 4095;
  __syncthreads();
  unsigned long long t0, t1;
  // 0: back-to-back s_memtime
  t0 = now(); for (int i = 0; i < 16; i++) { asm volatile("" ::: "memory"); t1 = now(); } 
  if (tid == 0) out[0] = (t1 - t0) / 16;
  // 1: dependent LDS read chain
  unsigned p = tid & 4095;
  t0 = now();
  for (int i = 0; i < iters; i++) p = lds[p];
  t1 = now();
  if (tid == 0) out[1] = (t1 - t0) / iters;
  out[20 + (p & 1)] = 0;
  // 2: barrier cost (8 waves, balanced)
  __syncthreads();
  t0 = now();
  for (int i = 0; i < iters; i++) __syncthreads();
  t1 = now();
  if (tid == 0) out[2] = (t1 - t0) / iters;
  // 3: dependent global load chain, cold (pointer chasing over a 64 MB array)
  unsigned q = tid * 4099u % n_chain;
  t0 = now();
  for (int i = 0; i < 64; i++) q = chain[q];
  t1 = now();
  if (tid == 0) out[3] = (t1 - t0) / 64;
  out[22 + (q & 1)] = 0;
  // 4: dependent global load chain, warm (re-walk the same 64 entries: L2/L1 hits)
  q = tid * 4099u % n_chain;
  t0 = now();
  for (int i = 0; i < 64; i++) q = chain[q];
  t1 = now();
  if (tid == 0) out[4] = (t1 - t0) / 64;
  out[24 + (q & 1)] = 0;
  // 5: single-lane dependent chain only (wave 0 lane 0), others idle at a barrier: cold region 2
  __syncthreads();
  if (tid == 0) {
    unsigned z = 12345u % n_chain;
    t0 = now();
    for (int i = 0; i < 64; i++) z = chain[(z + 7777777u) % n_chain];
    t1 = now();
    out[5] = (t1 - t0) / 64;
    out[26 + (z & 1)] = 0;
  }
  __syncthreads();
  // 6: DPP+fmax wave reduce of a double, dependent repeats
  double v = (double)tid;
  t0 = now();
  for (int i = 0; i < iters; i++) {
#define STEP(ctrl, rm) { int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rm, 0xf, false); int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rm, 0xf, false); v = fmax(v, __hiloint2double(hi, lo)); }
    STEP(0xB1, 0xf) STEP(0x4E, 0xf) STEP(0x141, 0xf) STEP(0x140, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    v = __hiloint2double(hi, lo) + (double)lane * 1e-9;
  }
  t1 = now();
  if (tid == 0) out[6] = (t1 - t0) / iters;
  out[28 + ((int)v & 1)] = 0;
  // 7: shuffle-based (ds_bpermute) 64-bit wave max, dependent repeats
  unsigned long long kk = tid * 1234567ull;
  t0 = now();
  for (int i = 0; i < iters; i++) {
    for (int o = 32; o > 0; o >>= 1) { unsigned long long u = __shfl_xor(kk, o); kk = u > kk ? u : kk; }
    kk += lane;
  }
  t1 = now();
  if (tid == 0) out[7] = (t1 - t0) / iters;
  out[30 + (kk & 1)] = 0;
  // 8: f64 division dependent chain
  double a = 1.0 + tid, b = 3.0 + lane;
  t0 = now();
  for (int i = 0; i < iters; i++) a = a / b + 1.0;
  t1 = now();
  if (tid == 0) out[8] = (t1 - t0) / iters;
  out[32 + ((int)a & 1)] = 0;
  // 9: cross-wave exchange: lane0 writes LDS, barrier, all read (one hop)
  t0 = now();
  for (int i = 0; i < iters; i++) {
    if (lane == 0) red[wave] = kk + i;
    __syncthreads();
    kk += red[lane & 7];
    __syncthreads();
  }
  t1 = now();
  if (tid == 0) out[9] = (t1 - t0) / iters;
  out[34 + (kk & 1)] = 0;
  // 10: one global store + dependent later load of different address (vmcnt in-order effect)
  t0 = now();
  for (int i = 0; i < 32; i++) { out[40 + tid % 8] = kk; q = chain[(q + i) % n_chain]; kk += q; }
  t1 = now();
  if (tid == 0) out[10] = (t1 - t0) / 32;
  out[36 + (kk & 1)] = 0;
  // 11: global load of an L2-resident small array (gd, 4 KB) independent loads, then use
  double s = 0;
  t0 = now();
  for (int i = 0; i < iters; i++) s += gd[(tid * 7 + i * 64 + (int)s) & 511];
  t1 = now();
  if (tid == 0) out[11] = (t1 - t0) / iters;
  out[38 + ((int)s & 1)] = 0;
}

int main() {
  const unsigned n = 16u << 20;
  std::vector<unsigned> h(n);
  unsigned x = 1;
  for (unsigned i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; h[i] = x % n; }
  unsigned *dc; double *gd; unsigned long long *out;
  CK(hipMalloc(&dc, n * 4)); CK(hipMalloc(&gd, 4096)); CK(hipMalloc(&out, 4096));
  CK(hipMemcpy(dc, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(gd, 0, 4096)); CK(hipMemset(out, 0, 4096));
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, dc, gd, n, 256);
    CK(hipDeviceSynchronize());
  }
  unsigned long long r[16];
  CK(hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost));
  const char *names[12] = {"s_memtime back-to-back", "dependent LDS read", "__syncthreads (8 waves)", "dependent global load, cold (HBM)",
                           "dependent global load, warm", "single-lane dependent global load, cold", "DPP+v_max_f64 wave reduce (+readlane)",
                           "ds_bpermute 64-bit wave max", "f64 division (dependent)", "LDS write + barrier + read + barrier",
                           "global store + dependent load", "global load L2-resident 4KB (dependent)"};
  for (int i = 0; i < 12; i++) printf("%-45s %6llu cycles\n", names[i], r[i]);
  return 0;
}
