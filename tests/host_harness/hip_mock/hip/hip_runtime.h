// hip_runtime.h — TEST INFRASTRUCTURE, not a HIP implementation: the handful of runtime calls kube-batch_amd/csrc/kb_engine.cpp
// makes, answered by the host heap, so that the engine's host side can be compiled UNCHANGED with g++ and driven by the CPU
// restatement of the kernels in ../device_emu.cpp (tests/test_emu_engine_cpu.py).  Everything is synchronous: a "launch" has
// finished when it returns, a "stream" keeps no queue, a copy is a memcpy.  Nothing under tests/ is linked into libkbengine.so.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <cmath>

#define __device__
#define __host__
#define __global__
#ifndef __forceinline__
#define __forceinline__ inline __attribute__((always_inline))
#endif

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct kbemu_stream *hipStream_t;
typedef struct kbemu_event { double ms; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1 };
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1 };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "ok" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }   // kHz of kbemu_wall_clock

// "device" memory is host memory.  KB_EMU_POISON=1 fills fresh allocations with 0xA5 instead of zeros: a read of memory nobody wrote
// then shows up as a wrong answer instead of passing by luck.
static inline hipError_t kbemu_alloc(void **p, size_t bytes) {
  static const int poison = getenv("KB_EMU_POISON") && atoi(getenv("KB_EMU_POISON"));
  *p = malloc(bytes ? bytes : 1);
  if (!*p) return hipErrorOutOfMemory;
  memset(*p, poison ? 0xA5 : 0, bytes);
  return hipSuccess;
}
static inline hipError_t hipMalloc(void **p, size_t bytes) { return kbemu_alloc(p, bytes); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned = 0) { return kbemu_alloc((void **)p, bytes); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
template <typename T> static inline hipError_t hipHostGetDevicePointer(T **dev, void *host, unsigned) { *dev = (T *)host; return hipSuccess; }
static inline hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { if (n) memmove(dst, src, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(dst, src, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; r++) memmove((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t = nullptr) {
  return hipMemcpy2D(dst, dpitch, src, spitch, width, height, k);
}
static inline hipError_t hipMemset(void *dst, int v, size_t n) { if (n) memset(dst, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t = nullptr) { if (n) memset(dst, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline double kbemu_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, sizeof(kbemu_event)); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->ms = kbemu_now_ms(); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->ms - a->ms); return hipSuccess; }
// the constant-rate device clock the kernels stamp rounds with (100 MHz, like gfx950's)
static inline unsigned long long kbemu_wall_clock() { return (unsigned long long)(kbemu_now_ms() * 100000.0); }
