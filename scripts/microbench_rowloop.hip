// Skeleton of the commit kernel's per-row protocol with synthetic data, pieces enabled one by one, to find the floor of
// each piece on MI355X.  512 threads (8 waves), one workgroup, R rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k) {
  double v = __longlong_as_double((long long)k);
#define STEP(ctrl, rm) { int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rm, 0xf, false); int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rm, 0xf, false); v = fmax(v, __hiloint2double(hi, lo)); }
  STEP(0xB1, 0xf) STEP(0x4E, 0xf) STEP(0x141, 0xf) STEP(0x140, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

struct Hdr { unsigned long long best[2]; unsigned long long cand; unsigned ex, pad; unsigned last, stop; };

template <int MODE>
__global__ void k(unsigned long long *out, int rows, unsigned long long *gdec) {
  __shared__ Hdr H;
  __shared__ unsigned long long tab[10 * 256];
  __shared__ unsigned desc[256 * 14];
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (unsigned i = tid; i < 256 * 14; i += 512) desc[i] = i * 2654435761u;
  for (unsigned i = tid; i < 10 * 256; i += 512) tab[i] = 0x4010000000000000ull + i;
  if (tid == 0) { H.best[0] = H.best[1] = 0; H.cand = 0; H.ex = 0; H.last = 0; H.stop = 0; }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long ck = 0x4000000100000000ull + tid, wmax = 0, acc = 0;
  unsigned nd = 0;
  for (int i = 0; i < rows; i++) {
    const unsigned par = i & 1;
    if (MODE >= 2) {
      if (wave == 7) {   // candidate role: publish cand + atomicMax
        unsigned long long cand = 0x4000000200000000ull + (unsigned)i;
        if (lane == 0) { *reinterpret_cast<uint4 *>(&H.cand) = make_uint4((unsigned)cand, (unsigned)(cand >> 32), 0u, 0u); atomicMax(&H.best[par], cand); }
      } else if (wave < 6) {
        if (MODE >= 3) {   // eval role: one wave re-reduces, everybody contributes
          if (wave == (unsigned)(i % 6)) { ck += desc[(i & 255) * 14]; wmax = wave_max_key(ck); }
          if (lane == 0 && wmax) atomicMax(&H.best[par], wmax);
        }
      }
    }
    __syncthreads();
    unsigned long long best = 0;
    if (MODE >= 2) {
      const uint4 h = *reinterpret_cast<const uint4 *>(&H.cand);
      best = H.best[par];
      acc += best + h.x;
      if (tid == 0) H.best[par ^ 1] = 0;
      if (MODE >= 4 && wave == 7) {   // clean commit: 13 lanes write a slot, lane 0 bookkeeping + one global store
        unsigned long long st8 = tab[(lane % 10) * 256 + ((i * 7) & 255)];
        double v = __longlong_as_double((long long)st8) - 1.0;
        if (lane < 10) tab[lane * 256 + (nd & 255)] = (unsigned long long)__double_as_longlong(v);
        if (lane == 0) { *reinterpret_cast<uint2 *>(&H.last) = make_uint2(nd, 0u); if (MODE >= 5) gdec[i & 1023] = best; }
      }
    }
    __syncthreads();
    if (MODE >= 2) { const uint2 h2 = *reinterpret_cast<const uint2 *>(&H.last); acc += h2.x; nd++; if (h2.y) break; }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[MODE] = (t1 - t0) / rows;
  if (acc == 1) out[20] = acc;
}

int main() {
  unsigned long long *out, *gdec;
  CK(hipMalloc(&out, 256)); CK(hipMalloc(&gdec, 8192)); CK(hipMemset(out, 0, 256));
  const int rows = 100000;
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, out, rows, gdec);
  hipLaunchKernelGGL(k<2>, dim3(1), dim3(512), 0, 0, out, rows, gdec);
  hipLaunchKernelGGL(k<3>, dim3(1), dim3(512), 0, 0, out, rows, gdec);
  hipLaunchKernelGGL(k<4>, dim3(1), dim3(512), 0, 0, out, rows, gdec);
  hipLaunchKernelGGL(k<5>, dim3(1), dim3(512), 0, 0, out, rows, gdec);
  CK(hipDeviceSynchronize());
  unsigned long long r[8]; CK(hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost));
  const char *names[6] = {"", "two barriers", "+ cand publish/atomicMax + pick + hdr2", "+ eval wave reduce + atomicMax", "+ clean commit (LDS)", "+ one global store per row"};
  for (int m = 1; m <= 5; m++) printf("mode %d %-45s %6llu cycles/row\n", m, names[m], r[m]);
  return 0;
}
