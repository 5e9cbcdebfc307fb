// hip_runtime.h — TEST INFRASTRUCTURE, not a HIP implementation: the handful of runtime calls kube-batch_amd/csrc/kb_engine.cpp
// makes, answered by the host heap (../hip_mock.cpp), so that the engine's host side can be compiled UNCHANGED with g++ and driven by
// the CPU restatement of the kernels in ../../device_emu.cpp (tests/test_emu_engine_cpu.py).  Nothing under tests/ is linked into
// libkbengine.so.
//
// Two modes.  Default: synchronous — a "launch" has finished when it returns, a copy is a memcpy.  KB_EMU_ASYNC=1: every stream is
// a worker thread with an in-order queue, and the calls keep HIP's ordering rules as the engine relies on them:
//   * kernel launches, hipMemsetAsync, device-to-device copies and copies from / to PINNED host memory are queued: they run later,
//     on the stream's thread, and read their source then;
//   * hipMemcpyAsync from PAGEABLE host memory stages the source before it returns (the caller may reuse it at once), to pageable
//     host memory it completes before it returns;
//   * hipMemcpy / hipMemset (the null stream) do NOT wait for a stream created with hipStreamNonBlocking — the engine's is;
//   * hipStreamSynchronize, hipFree, hipHostFree and hipStreamDestroy wait for the queue(s) to drain.
// Under ThreadSanitizer (scripts/sanitize_cpu.sh) that turns "the host touched a staging buffer, a mailbox word or a result before the
// device was done with it" into a reported data race instead of a rare wrong answer on the GPU.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <functional>

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
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t kbemu_host_alloc(void **p, size_t bytes);
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned = 0) { return kbemu_host_alloc((void **)p, bytes); }
hipError_t hipHostFree(void *p);
template <typename T> static inline hipError_t hipHostGetDevicePointer(T **dev, void *host, unsigned) { *dev = (T *)host; return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind);
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemset(void *dst, int v, size_t n);
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t words, const uint32_t *mask);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);

// for device_emu.cpp: run `f` as the next piece of work of stream `s` (at once in the synchronous mode), wait for a stream
void kbemu_enqueue(hipStream_t s, std::function<void()> f);
void kbemu_drain(hipStream_t s);
double kbemu_now_ms();
// the constant-rate device clock the kernels stamp rounds with (100 MHz, like gfx950's)
static inline unsigned long long kbemu_wall_clock() { return (unsigned long long)(kbemu_now_ms() * 100000.0); }
