// hip_mock.cpp — TEST INFRASTRUCTURE: the runtime behind hip/hip_runtime.h (see there).  Host heap for memory, one worker thread per
// stream in the asynchronous mode (KB_EMU_ASYNC=1), inline execution otherwise.
#include <hip/hip_runtime.h>

#include <stdio.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

namespace {

bool env_flag(const char *name) {
  const char *v = getenv(name);
  return v && atoi(v) != 0;
}
bool async_mode() {
  static const bool a = env_flag("KB_EMU_ASYNC");
  return a;
}

// KB_EMU_STATS=1: a census of the runtime calls (each one costs microseconds of host time on the real runtime), printed at exit
struct Census {
  unsigned long long copies[4] = {0, 0, 0, 0}, copy_bytes[4] = {0, 0, 0, 0}, small_copies = 0, memsets = 0, syncs = 0, tasks = 0;
  bool on = false;
  Census() { on = env_flag("KB_EMU_STATS"); }
  ~Census() {
    if (!on) return;
    fprintf(stderr, "[kbemu] stream tasks (launches + queued copies) %llu, synchronisations %llu, memsets %llu\n", tasks, syncs, memsets);
    const char *kn[4] = {"H2H", "H2D", "D2H", "D2D"};
    for (int k = 0; k < 4; k++)
      if (copies[k]) fprintf(stderr, "[kbemu] %s copies %llu, %.1f MB\n", kn[k], copies[k], copy_bytes[k] / 1e6);
    fprintf(stderr, "[kbemu] copies of <= 64 bytes: %llu\n", small_copies);
  }
  void copy(hipMemcpyKind k, size_t n) { if (on && (int)k < 4) { copies[k]++; copy_bytes[k] += n; if (n <= 64) small_copies++; } }
} g_census;

std::mutex g_mu;                              // registries below
std::map<const char *, size_t> g_pinned;      // hipHostMalloc'ed ranges
std::set<kbemu_stream *> g_streams;

bool is_pinned(const void *p) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_pinned.upper_bound((const char *)p);
  if (it == g_pinned.begin()) return false;
  --it;
  return (const char *)p < it->first + it->second;
}

hipError_t alloc(void **p, size_t bytes) {
  static const bool poison = env_flag("KB_EMU_POISON");
  *p = malloc(bytes ? bytes : 1);
  if (!*p) return hipErrorOutOfMemory;
  memset(*p, poison ? 0xA5 : 0, bytes);
  return hipSuccess;
}

}  // namespace

struct kbemu_stream {
  std::thread th;
  std::mutex m;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool stop = false, busy = false;
  long jitter_us = 0;   // set at creation (hipStreamCreateWithFlags)
  unsigned long long rng = (unsigned long long)(uintptr_t)this * 0x9E3779B97F4A7C15ull + (getenv("KB_EMU_JITTER_SEED") ? strtoull(getenv("KB_EMU_JITTER_SEED"), nullptr, 10) : 1ull);
  void run() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [&] { return stop || !q.empty(); });
      if (q.empty()) return;
      std::function<void()> f = std::move(q.front());
      q.pop_front();
      busy = true;
      lk.unlock();
      // KB_EMU_JITTER_US=n: every task starts up to n microseconds late, each stream on a pseudo-random sequence of its own (KB_EMU_JITTER_SEED):
      // the relative timing of the engine's two streams — which the device decides anew on every run — is swept instead of being whatever
      // two idle host threads make it
      if (jitter_us > 0) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        std::this_thread::sleep_for(std::chrono::microseconds((long)((rng >> 33) % (unsigned long long)(jitter_us + 1))));
      }
      f();
      lk.lock();
      busy = false;
      if (q.empty()) idle.notify_all();
    }
  }
};

double kbemu_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void kbemu_enqueue(hipStream_t s, std::function<void()> f) {
  if (g_census.on) g_census.tasks++;
  if (!async_mode() || s == nullptr) { f(); return; }   // the null stream: the engine never launches there
  std::lock_guard<std::mutex> lk(s->m);
  s->q.push_back(std::move(f));
  s->cv.notify_one();
}
void kbemu_drain(hipStream_t s) {
  if (!async_mode() || s == nullptr) return;
  std::unique_lock<std::mutex> lk(s->m);
  s->idle.wait(lk, [&] { return s->q.empty() && !s->busy; });
}
static void drain_all() {
  if (!async_mode()) return;
  std::vector<kbemu_stream *> all;
  { std::lock_guard<std::mutex> lk(g_mu); all.assign(g_streams.begin(), g_streams.end()); }
  for (kbemu_stream *s : all) kbemu_drain(s);
}

hipError_t hipMalloc(void **p, size_t bytes) { return alloc(p, bytes); }
hipError_t hipFree(void *p) { drain_all(); free(p); return hipSuccess; }   // hipFree synchronises the device
hipError_t kbemu_host_alloc(void **p, size_t bytes) {
  hipError_t e = alloc(p, bytes);
  if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_mu); g_pinned[(const char *)*p] = bytes ? bytes : 1; }
  return e;
}
hipError_t hipHostFree(void *p) {
  drain_all();
  { std::lock_guard<std::mutex> lk(g_mu); g_pinned.erase((const char *)p); }
  free(p);
  return hipSuccess;
}

hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind) { g_census.copy(kind, n); if (n) memmove(dst, src, n); return hipSuccess; }   // does not wait for non-blocking streams
hipError_t hipMemset(void *dst, int v, size_t n) { if (n) memset(dst, v, n); return hipSuccess; }
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; r++) memmove((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
  return hipSuccess;
}

hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s) {
  g_census.copy(kind, n);
  if (!n) return hipSuccess;
  if (!async_mode() || s == nullptr) { memmove(dst, src, n); return hipSuccess; }
  if (kind == hipMemcpyHostToDevice && !is_pinned(src)) {          // pageable source: staged before the call returns
    std::shared_ptr<std::vector<char>> tmp = std::make_shared<std::vector<char>>((const char *)src, (const char *)src + n);
    kbemu_enqueue(s, [dst, tmp]() { memcpy(dst, tmp->data(), tmp->size()); });
  } else if (kind == hipMemcpyDeviceToHost && !is_pinned(dst)) {   // pageable destination: complete when the call returns
    kbemu_drain(s);
    memmove(dst, src, n);
  } else {
    // KB_EMU_D2D_DELAY_US: a device-to-device copy takes that long on its stream (tests: work on another stream that needs the copy's result
    // must be ordered behind it by the engine, not by luck)
    const long delay_us = getenv("KB_EMU_D2D_DELAY_US") ? atol(getenv("KB_EMU_D2D_DELAY_US")) : 0;
    if (delay_us > 0 && kind == hipMemcpyDeviceToDevice)
      kbemu_enqueue(s, [dst, src, n, delay_us]() { std::this_thread::sleep_for(std::chrono::microseconds(delay_us)); memmove(dst, src, n); });
    else
      kbemu_enqueue(s, [dst, src, n]() { memmove(dst, src, n); });
  }
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s) {
  if (async_mode() && s != nullptr) {
    if (kind == hipMemcpyDeviceToHost && !is_pinned(dst)) kbemu_drain(s);
    else { kbemu_enqueue(s, [=]() { hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind); }); return hipSuccess; }
  }
  return hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind);
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s) {
  if (g_census.on) g_census.memsets++;
  if (n) kbemu_enqueue(s, [dst, v, n]() { memset(dst, v, n); });
  return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t *out, unsigned) {
  kbemu_stream *s = new kbemu_stream();
  {   // KB_EMU_JITTER_STREAMS: bit i set = the i-th stream an engine creates is jittered (the engine's own stream is its first, the overlapped
      // rounds' second stream its second; default: all); counted modulo 2 so that every engine of a process gets the same treatment
    static std::atomic<unsigned> created{0};
    const unsigned idx = created.fetch_add(1) & 1u;
    const unsigned long mask = getenv("KB_EMU_JITTER_STREAMS") ? strtoul(getenv("KB_EMU_JITTER_STREAMS"), nullptr, 0) : ~0ul;
    if ((mask >> idx) & 1ul) s->jitter_us = getenv("KB_EMU_JITTER_US") ? atol(getenv("KB_EMU_JITTER_US")) : 0;
  }
  if (async_mode()) s->th = std::thread([s] { s->run(); });
  { std::lock_guard<std::mutex> lk(g_mu); g_streams.insert(s); }
  *out = s;
  return hipSuccess;
}
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *out, uint32_t, const uint32_t *) { return hipStreamCreateWithFlags(out, 0); }
hipError_t hipStreamDestroy(hipStream_t s) {
  if (!s) return hipSuccess;
  kbemu_drain(s);
  { std::lock_guard<std::mutex> lk(g_mu); g_streams.erase(s); }
  if (s->th.joinable()) {
    { std::lock_guard<std::mutex> lk(s->m); s->stop = true; }
    s->cv.notify_one();
    s->th.join();
  }
  delete s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { if (g_census.on) g_census.syncs++; kbemu_drain(s); return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, sizeof(kbemu_event)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { drain_all(); free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { kbemu_enqueue(s, [e]() { e->ms = kbemu_now_ms(); }); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->ms - a->ms); return hipSuccess; }
