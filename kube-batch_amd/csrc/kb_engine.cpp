// kb_engine.cpp — the C ABI of include/kb_engine.h over the HIP kernels of kb_kernels.hip.
//
// Round structure: the host order machine speculates the reference's task order for a window of W
// tasks (assuming each gets a node, which only ever fails when a whole feasibility class has died — and that is
// monotone inside one action), the device evaluates the window's mask+score matrix against the round-start node
// state (K1, once per distinct task shape), builds each shape's sorted candidate list (K3) and commits the window in the
// reference's order, a run of same-shape rows at a time (K5).  A mis-speculation (no feasible node / Pipeline instead of Allocate) stops the commit kernel at that row;
// the host rolls the order machine back to the round start, replays the confirmed prefix and re-plans.
#include <hip/hip_runtime.h>

#include <sched.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/kb_engine.h"
#include "kb_device.h"
#include "kb_host.hpp"
#include "kb_waterfill.hpp"
#include "kb_preempt.hpp"

using namespace kb;

#define HIP_OK(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess) throw EngineError(KB_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

namespace {

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;   // the size asked for last
  size_t cap = 0;     // what is allocated: a session of the same size (the Go action loads one every cycle) or a smaller one reuses it —
                      // hipFree synchronises the device and hipMalloc is not cheap either
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    cap = 0;
  }
  void swap(DevBuf &o) { std::swap(p, o.p); std::swap(bytes, o.bytes); std::swap(cap, o.cap); }
  void alloc(size_t n) {
    n = n ? n : 16;
    if (p && n <= cap) { bytes = n; return; }
    release();
    bytes = cap = n;
    HIP_OK(hipMalloc(&p, cap));
  }
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// grow-only pinned host array: the per-round staging buffers (window rows in, decision records out) are copied with
// hipMemcpyAsync every round, which only is asynchronous (and cheap to issue) from page-locked memory
template <typename T> struct Pinned {
  T *p = nullptr;
  size_t n = 0;
  unsigned flags = hipHostMallocDefault;
  Pinned() = default;
  Pinned(const Pinned &) = delete;
  Pinned &operator=(const Pinned &) = delete;
  ~Pinned() { if (p) (void)hipHostFree(p); }
  void resize(size_t m) {
    if (m <= n) return;
    T *q = nullptr;
    HIP_OK(hipHostMalloc((void **)&q, sizeof(T) * m, flags));
    if (p) { std::memcpy(q, p, sizeof(T) * n); (void)hipHostFree(p); }
    p = q;
    n = m;
  }
  T *data() { return p; }
  const T *data() const { return p; }
  size_t size() const { return n; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};

// kb_session_load's staging: ONE pinned area, grow-only like the device buffers, in which every host-to-device source of a load is
// assembled (padding included) and from which it is copied asynchronously.  Round 4 copied from pageable memory — the caller's snapshot,
// std::vectors of this file —: the runtime pins such a source on the fly (or stages it, blocking) at every call, a per-call cost of
// tens to hundreds of microseconds with a long tail, about thirty times per load, and upload_padded synchronised the stream behind each
// of its eight temporaries.  A block stays valid until the next load resets the area, and a load ends behind a stream synchronisation.
// Sources of 8 MiB and more that outlive the load (the task vectors of a million-task session: Uploader::copy_persistent) skip the area: one pin
// per call is cheaper than the extra pass over them.
struct PinnedArena {
  struct Block { unsigned char *p; size_t cap; };
  std::vector<Block> blocks;
  size_t cur = 0, off = 0;
  PinnedArena() = default;
  PinnedArena(const PinnedArena &) = delete;
  PinnedArena &operator=(const PinnedArena &) = delete;
  ~PinnedArena() { for (Block &b : blocks) (void)hipHostFree(b.p); }
  void reset() { cur = 0; off = 0; }
  void *take(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    while (cur < blocks.size() && off + bytes > blocks[cur].cap) { cur++; off = 0; }
    if (cur == blocks.size()) {
      Block b{nullptr, std::max<size_t>(bytes, (size_t)8 << 20)};
      HIP_OK(hipHostMalloc((void **)&b.p, b.cap, hipHostMallocDefault));
      blocks.push_back(b);
      off = 0;
    }
    void *p = blocks[cur].p + off;
    off += bytes;
    return p;
  }
  size_t bytes_held() const { size_t t = 0; for (const Block &b : blocks) t += b.cap; return t; }
};
constexpr size_t kStageMaxBytes = (size_t)8 << 20;

struct Uploader {
  PinnedArena &arena;
  hipStream_t s;
  Uploader(PinnedArena &a, hipStream_t st) : arena(a), s(st) {}
  // b := n elements the caller writes through the returned pointer BEFORE the next take / copy (the copy is queued by commit())
  template <typename T> T *stage(DevBuf &b, size_t n) {
    b.alloc(n * sizeof(T));
    pending_dst = b.p; pending_bytes = n * sizeof(T);
    pending_src = arena.take(pending_bytes ? pending_bytes : 16);
    return reinterpret_cast<T *>(pending_src);
  }
  void commit() {
    if (pending_bytes) HIP_OK(hipMemcpyAsync(pending_dst, pending_src, pending_bytes, hipMemcpyHostToDevice, s));
    pending_bytes = 0;
  }
  // any source: copied into the area first (the source may die before the load's synchronisation: block-scoped temporaries)
  template <typename T> void copy(DevBuf &b, const T *src, size_t n) {
    T *p = stage<T>(b, n);
    if (n) std::memcpy(p, src, n * sizeof(T));
    commit();
  }
  // a source that outlives the load's synchronisation (the caller's snapshot, the host session's vectors): from 8 MiB on straight from where it
  // lies — the runtime pins it for the transfer; one pin per call is cheaper than an extra pass over a million-task vector
  template <typename T> void copy_persistent(DevBuf &b, const T *src, size_t n) {
    if (n * sizeof(T) >= kStageMaxBytes) {
      b.alloc(n * sizeof(T));
      HIP_OK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
      return;
    }
    copy(b, src, n);
  }
  // rows of a [rows][n] host matrix into a padded [rows][np] device matrix (pad value `fill`)
  template <typename T> void padded(DevBuf &b, const T *src, size_t rows, size_t n, size_t np, T fill = T(0)) {
    T *p = stage<T>(b, rows * np);
    for (size_t r = 0; r < rows; r++) {
      if (n) std::memcpy(p + r * np, src + r * n, n * sizeof(T));
      std::fill(p + r * np + n, p + (r + 1) * np, fill);
    }
    commit();
  }
 private:
  void *pending_dst = nullptr, *pending_src = nullptr;
  size_t pending_bytes = 0;
};

struct Timer {   // HIP-event pair on the engine stream
  hipEvent_t a = nullptr, b = nullptr;
  void init() {
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&b));
  }
  void destroy() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    a = b = nullptr;
  }
};

}  // namespace

struct MgState;
static void mg_free(MgState *m);
static void mg_reset(MgState *m);

struct kb_engine {
  std::string err;
  int device = 0;
  uint32_t window = 256, commit_batch = 0, flags = 0;   // 256: measured optimum on the 100k x 10k snapshots (small dirty sets vs per-round cost)
  Policy pol;
  hipStream_t stream = nullptr, own_stream = nullptr;   // stream: the one in use (own_stream unless kb_engine_use_stream gave another)
  bool loaded = false;
  bool tainted = false;   // an evict action failed after it had touched device / host state: kb_run_* answer KB_E_STATE until kb_session_load / kb_session_reset
  HostSession hs;
  KbDev dev{};
  kb_stats stats{};
  uint64_t round_no = 0;

  // session buffers
  DevBuf b_idle, b_rel, b_nzc, b_nzm, b_podcnt, b_acpu, b_amem, b_maxpods, b_ncls, b_nmask, b_invac, b_invam;
  uint64_t k5_walks = 0, k5_rescans = 0, k5_demand = 0, k5_slots = 0;   // commit kernel counters (KB_K5_STATS)
  double k5_trace[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double tl_repair_tag = 0;   // KB_K5_STATS: ms between the start of a repair launch and the moment its first workgroup had seen its list's tag
  uint32_t eff_window = 0;   // window actually used for this session (bounded by the commit kernel's LDS budget)
  // Which commit kernel a round runs on: the selection kernel (kb_commit_sel.hip), backfill rounds included; KB_COMMIT_KERNEL=run|select pins
  // one of the two — they compute the same decisions, and every -m gpu case runs under each (the run kernel, kb_commit.hip, is the plain
  // serial restatement the selection is held to).  Round 3's batch kernel (speculation across shapes) and the per-round rules that chose
  // between kernels lost to plain selection on every configuration (profiles/round4/call30_pinned_kernels) and are gone: HISTORY.md.
  int commit_kernel = KB_COMMIT_SELECT, commit_pin = -1;
  double dirty_share = 0.0;   // share of rows won by a node the round had already changed (exponential average; a statistic)
  uint64_t rounds_run = 0, rounds_sel = 0;
  uint64_t sel_stat[4] = {0, 0, 0, 0};   // selection kernel: runs with every pick a clean first placement / committed by shots; shots cut short by a table's end; shots
  uint32_t shape_cap = KB_K5_MAX_SHAPES;   // distinct shapes a window may hold (each keeps its candidate list in the commit kernel's LDS)
  std::vector<uint32_t> plan_stamp;   // per row-shape id: stamp of the window being planned
  uint32_t plan_epoch = 0;
  const double *t_fit = nullptr;   // backfill's view of t_init (BestEffort rows: Resreq cpu / memory), == b_tinit when they agree
  bool idle_below_eps = false;
  DevBuf b_tfit;
  DevBuf b_tinit, b_tres, b_tnzc, b_tnzm, b_tcls, b_tactive, b_tresmask, b_tjob, b_tstatus, b_tnode, b_tbind, b_tcounted, b_jallocated, b_compat, b_crows, b_aff, b_affcls;
  DevBuf p_idle, p_rel, p_nzc, p_nzm, p_podcnt, p_tstatus, p_tnode, p_tcounted, p_ports, p_nmask;   // pristine copies for kb_session_reset
  // inter-pod (anti)affinity tables (kb_interpod) and the pristine copies of their live parts
  DevBuf b_ip_cdom, b_ip_ccnt, b_ip_ctot, b_ip_tinc, b_ip_tforbid, b_ip_treq, b_ip_tself, b_ip_tsubj, b_ip_tchk, b_ip_pdom, b_ip_pbound, b_ip_punb, b_ip_tcinc,
      b_ip_tsig, b_ip_sigw, b_ip_z, b_ip_scnt, b_ip_shist, p_ip_ccnt, p_ip_ctot, p_ip_punb, p_ip_z;
  DevBuf b_ports, b_twant, b_tconf;   // host ports (only when the snapshot carries any)
  DevBuf b_ports_x, b_twant_x, b_tconf_x, p_ports_x;   // their words behind the first (kb_snapshot.port_words > 1 and some pod reaches there), pristine copy
  DevBuf b_jbegin, b_jmin, b_jqueue, b_total, b_deserved, b_desmask, b_jalloc, b_jshare, b_qalloc, b_qshare, b_jready;
  uint32_t total_mask = 0;
  // round buffers
  DevBuf b_desc;
  Pinned<unsigned long long> h_listkeys;  // one complete candidate list of an evict action's preemptor shape (D2H target)
  Pinned<unsigned char> h_evict;          // an evict action's entry / exit staging: node state and task table through ONE pinned block, one synchronisation each way
  DevBuf b_scatter;                       // packed node records of upload_live_nodes
  Pinned<unsigned long long> h_scatter;
  DevBuf b_sscore, b_smask, b_xslot, b_xorder, b_xchunks;   // per-shape rows, row->shape map, rows in shape order and its chunk table (kb_eval_matrix / kb_bench_matrix)
  std::vector<uint32_t> h_xorder;
  std::vector<KbXChunk> h_xchunks;
  size_t xs_cap = 0, xslot_cap = 0;
  DevBuf b_mrows, b_same, b_score, b_maskw, b_keys;
  // Overlapped candidate lists: the matrix and arg-max launches of a chained round run on a second stream
  // beside its predecessor's commit kernel, into buffers of their own (matrix rows, stale lists per staging half, one `ready` word per
  // list, a scratch block for their time stamps); kb_launch_repair on the first stream turns the stale lists into the round's lists
  hipStream_t stream_b = nullptr;
  DevBuf b_score2, b_maskw2, b_stale, b_ready, b_task_rows, b_lready;   // b_lready: one word per repaired list and staging half (KbRound::lists_ready)
  bool fuse_repair = true;   // the selection kernel's launch carries its round's repair workgroups (KB_FUSE_REPAIR=0: the launch of its own in front of it)
  Pinned<unsigned long long> h_cand_out;   // per staging half: the output-block words the second stream's launches stamp (start of the matrix launch, start of the arg-max launch)
  unsigned long long *d_cand_out = nullptr;
  uint32_t mat2_cap = 0;
  size_t stale_cap = 0;
  bool overlap = true;             // KB_OVERLAP=0: every round on the plain path (matrix -> arg-max -> commit on one stream)
  uint64_t overlapped_rounds = 0, overlap_faults = 0;
  bool device_waterfill = true;    // proportion's water-fill runs as a launch at kb_session_load (kb_waterfill.hip); KB_DEVICE_WATERFILL=0: the host loop of kb_session.cpp (A/B)
  uint32_t waterfill_passes = 0;
  DevBuf b_wf_queues, b_wf_state;
  DevBuf b_win, b_out;   // per-round upload / download blocks (see h_win / h_out)
  DevBuf b_chain;        // KbRound::chain: tag of the last round that committed its whole window
  // feasibility probe at speculation breaks (ActionRun::probe_launch / probe_collect): one representative task per feasibility shape still alive.
  // Rows in and flags out live in mapped pinned memory the kernel reads and writes directly (like h_win / h_out): the probe is ONE stream
  // operation — it was copy -> memset -> kernel -> copy, ~4.5 us each with a gap behind each, around a kernel of 8.6 us
  Pinned<uint32_t> h_probe_alive, h_probe_rows;
  uint32_t *d_probe_alive = nullptr, *d_probe_rows = nullptr;   // their device addresses (re-read when kb_session_load grew them)
  bool probe_enabled = true;          // KB_PROBE=0 disables
  uint64_t probes = 0, probe_deaths = 0;
  int commit_kernel_of[2] = {0, 0};   // the commit kernel launched for the round in each staging half
  uint32_t win_cap = 0, mat_cap = 0;
  size_t keys_cap = 0;
  Pinned<uint32_t> h_rows, h_slot, h_mrows;
  std::vector<uint32_t> h_decnode, h_deckind;
  Pinned<uint32_t> h_win;             // per-round upload  [rows | slots | mrows] at fixed offsets of KB_K5_MAX_WINDOW
  Pinned<unsigned long long> h_out;   // per-round download: KB_OUT_HDR header words (kb_device.h) + decision records
  const uint32_t *d_hwin = nullptr;       // device view of h_win
  bool direct_window = true;              // KB_DIRECT_WINDOW=0: always copy the window into b_win first
  unsigned long long *d_hout = nullptr;   // device view of h_out (fast rounds: the commit kernel writes it directly)
  bool fast_rounds = true;            // host spins on h_out[KB_OUT_SEQ] instead of synchronising the stream every round
  bool chain_rounds = true;           // queue the next speculated round behind the running one (KbRound::chain); KB_CHAIN_ROUNDS=0 disables
  unsigned long long seq = 0;
  double wall_khz = 100000.0;         // rate of the device's constant wall clock
  std::vector<uint8_t> h_same;
  std::vector<uint32_t> shape_stamp, shape_slot_of;   // per row-shape id: round stamp and slot inside the current round
  uint32_t stamp = 0;
  std::vector<Timer> ev;          // event pool for per-launch timing
  PinnedArena load_arena;         // kb_session_load's staging area (above)
  Pinned<unsigned char> h_fin;    // pinned D2H target of run_finalize (seven results in one block, copied out after ONE synchronisation)
  // the host mirrors of the device reduction as of kb_session_load: kb_session_reset restores them instead of reducing the restored
  // (identical) state again
  struct FinalMirror { std::vector<double> job_alloc, job_share, queue_alloc, queue_share; std::vector<int32_t> job_ready; std::vector<uint8_t> t_status; std::vector<uint32_t> t_node; bool valid = false; } fin0;
  // where the host's wall time of a cycle goes outside the device rounds (KB_K5_STATS=1 prints it): reset, the action's start up to its
  // first launch, the speculation breaks (from a stopped round's answer to the re-planned launch), the closing reduction, round waits
  bool async_pending = false;   // kb_session_reset queued device-to-device copies on `stream` and returned without waiting: whoever touches the
                                // buffers outside that stream (null-stream copies of the getters and of the evict actions, a stream switch) waits first
  // "a Pending task carries a NodeName" is looked for in front of an allocate / backfill only when it can have appeared: once per loaded
  // session (load_clean remembers that the load-time state passed; kb_session_reset returns to that state) and after every evict action
  // (a discarded statement is the one thing inside a session that creates such a task)
  bool stale_checked = false, pristine = true, load_clean = false;
  double tl_reset = 0, tl_begin = 0, tl_break = 0, tl_finish = 0, tl_wait = 0, tl_backfill = 0;
  double tl_begin_parts[3] = {0, 0, 0};   // of tl_begin: the order machine's set-up, the first feasibility probe, the first plan (the rest: buffers, the first launch)
  double tl_break_parts[3] = {0, 0, 0};   // of tl_break: the probe's launch + absorbing the answer (roll-back + replay) beside it, waiting for the probe, the re-plan (the rest: the skipped round, the launch)
  std::vector<kb_decision> decisions_all;   // decisions of the last multi-GPU round sequence
  std::vector<uint32_t> evictions;          // committed evictions of the session's preempt actions, in cache.Evict order

  // multi-GPU round state (kb_round_*), defined below
  struct MgState *mg = nullptr;

  ~kb_engine() {
    mg_free(mg);
    for (auto &t : ev) t.destroy();
    if (own_stream) (void)hipStreamDestroy(own_stream);
    if (stream_b) (void)hipStreamDestroy(stream_b);
  }
};

static thread_local std::string g_create_err;

namespace {


// window buffers: one entry per task row of a round
void ensure_window_buffers(kb_engine *e, uint32_t rows) {
  if (rows <= e->win_cap) return;
  if (!e->b_desc.p) e->b_desc.alloc(sizeof(KbRowDesc) * 2 * KB_K5_MAX_WINDOW);   // two fixed halves (chained rounds alternate): a round in flight must not see them move
  e->h_rows.resize(rows);
  e->h_slot.resize(rows);
  e->h_decnode.resize(rows);
  e->h_deckind.resize(rows);
  e->win_cap = rows;
}
// scratch rows of the inter-pod priority kernel (per matrix row: per-node counts, per-domain sums)
void ensure_ip_scratch(kb_engine *e, size_t rows) {
  if (!e->hs.has_interpod) return;
  const size_t NP = e->dev.NP;
  if (e->b_ip_scnt.bytes < sizeof(long long) * rows * NP) {
    e->b_ip_scnt.alloc(sizeof(long long) * rows * NP);
    e->b_ip_shist.alloc(sizeof(int32_t) * rows * NP);
  }
  e->dev.ip_scratch_cnt = e->b_ip_scnt.as<long long>();
  e->dev.ip_scratch_hist = e->b_ip_shist.as<int32_t>();
}
// matrix buffers: one matrix row per distinct shape (or per task row for kb_eval_matrix), L candidate keys per row
void ensure_matrix_buffers(kb_engine *e, uint32_t mrows, uint32_t L) {
  const size_t NP = e->dev.NP;
  if (mrows > e->mat_cap) {
    e->b_mrows.alloc(sizeof(uint32_t) * mrows);
    e->b_same.alloc(mrows);
    e->b_score.alloc(sizeof(uint16_t) * (size_t)mrows * NP);
    e->b_maskw.alloc(sizeof(uint32_t) * (size_t)mrows * (NP / 32));
    e->h_mrows.resize(mrows);
    e->h_same.resize(mrows);
    e->mat_cap = mrows;
  }
  ensure_ip_scratch(e, std::max<uint32_t>(mrows, e->mat_cap));
  size_t need = (size_t)mrows * L;
  if (need > e->keys_cap) {
    e->b_keys.alloc(sizeof(unsigned long long) * need);
    e->keys_cap = need;
  }
}

Timer &get_timer(kb_engine *e, size_t i) {
  while (e->ev.size() <= i) {
    Timer t;
    t.init();
    e->ev.push_back(t);
  }
  return e->ev[i];
}

// the engine's own stream is non-blocking: a null-stream hipMemcpy does not wait for what kb_session_reset left queued on it
void quiesce(kb_engine *e) {
  if (!e->async_pending) return;
  HIP_OK(hipStreamSynchronize(e->stream));
  e->async_pending = false;
}

// gang ballot + share reduction on the device, results mirrored to the host session
void run_finalize(kb_engine *e, const std::function<void()> &after_sync = nullptr) {
  Timer &tm = get_timer(e, 3);
  HIP_OK(hipEventRecord(tm.a, e->stream));
  kb_launch_finalize(e->dev, e->b_jbegin.as<uint32_t>(), e->b_jmin.as<int>(), e->b_jqueue.as<uint32_t>(), e->pol.gang_job_ready ? 1 : 0,
                     e->b_total.as<double>(), e->total_mask, e->b_deserved.as<double>(), e->b_desmask.as<uint32_t>(),
                     e->b_jalloc.as<double>(), e->b_jshare.as<double>(), e->b_qalloc.as<double>(), e->b_qshare.as<double>(),
                     e->b_jready.as<int>(), e->stream);
  HIP_OK(hipEventRecord(tm.b, e->stream));
  HostSession &hs = e->hs;
  // seven results, one pinned block: the copies queue behind the kernels as DMA commands (a pageable target makes every one of them a
  // staged, blocking copy), ONE synchronisation, then plain memcpys into the host session's vectors
  struct Part { void *dst; const void *src; size_t bytes, off; };
  Part parts[7] = {{hs.job_alloc.data(), e->b_jalloc.p, sizeof(double) * hs.job_alloc.size(), 0}, {hs.job_share.data(), e->b_jshare.p, sizeof(double) * hs.job_share.size(), 0},
                   {hs.queue_alloc.data(), e->b_qalloc.p, sizeof(double) * hs.queue_alloc.size(), 0}, {hs.queue_share.data(), e->b_qshare.p, sizeof(double) * hs.queue_share.size(), 0},
                   {hs.job_ready.data(), e->b_jready.p, sizeof(int32_t) * hs.job_ready.size(), 0}, {hs.t_status.data(), e->b_tstatus.p, hs.T, 0},
                   {hs.t_node.data(), e->b_tnode.p, sizeof(uint32_t) * hs.T, 0}};
  size_t total = 0;
  for (Part &pt : parts) { pt.off = total; total += (pt.bytes + 63) & ~(size_t)63; }
  e->h_fin.resize(total ? total : 64);
  for (const Part &pt : parts)
    if (pt.bytes) HIP_OK(hipMemcpyAsync(e->h_fin.data() + pt.off, pt.src, pt.bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_OK(hipStreamSynchronize(e->stream));
  e->async_pending = false;
  if (after_sync) after_sync();   // kb_session_load: what else came back behind this synchronisation (the water-fill's results)
  for (const Part &pt : parts)
    if (pt.bytes) std::memcpy(pt.dst, e->h_fin.data() + pt.off, pt.bytes);
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, tm.a, tm.b));
  e->stats.reduce_ms += ms;
  // queues without a job in the session keep share 0 (no queueOpts entry), and so do queues updateShare never ran for
  // (water-fill loop skipped on total weight 0 and no event yet: HostSession::queue_share_live)
  for (uint32_t q = 0; q < hs.Q; q++)
    if (!hs.queue_has_attr[q] || !hs.queue_share_live[q]) hs.queue_share[q] = 0.0;
}

KbRound make_round(kb_engine *e, uint32_t n_rows, uint32_t n_mrows, uint32_t L, int fit_mode, bool backfill, uint32_t buf = 0) {
  KbRound r{};
  r.rows = e->b_win.as<uint32_t>();
  r.shape_slot = e->b_win.as<uint32_t>() + KB_K5_MAX_WINDOW;
  r.n_rows = n_rows;
  r.desc = e->b_desc.as<KbRowDesc>() + (size_t)buf * KB_K5_MAX_WINDOW;
  r.trace = nullptr;
  r.cap = std::max<uint32_t>(64, ((n_rows + 63) / 64) * 64);
  r.mrows = e->b_win.as<uint32_t>() + 2 * KB_K5_MAX_WINDOW;
  r.mrow_task0 = 0;
  r.same_prev = nullptr;
  r.n_mrows = n_mrows;
  r.fit_mode = fit_mode;
  r.score = e->b_score.as<uint16_t>();
  r.maskw = e->b_maskw.as<uint32_t>();
  r.keys = e->b_keys.as<unsigned long long>();
  r.L = L;
  r.dec = e->b_out.as<unsigned long long>() + KB_OUT_HDR;   // header words, then the decision records
  r.result = e->b_out.as<uint32_t>();
  r.host_out = nullptr;
  r.seq = 0;
  r.backfill = backfill ? 1 : 0;
  r.batch = 0;
  r.gather = 0;
  r.delta = nullptr;
  r.own_row0 = r.own_row1 = 0;
  return r;
}

// distinct shapes of the window e->h_rows[0..n): fills h_slot (per row) and h_mrows (representative task per shape)
uint32_t assign_shapes(kb_engine *e, uint32_t n, const uint32_t *rows = nullptr) {
  HostSession &hs = e->hs;
  if (!rows) rows = e->h_rows.data();
  if (e->shape_stamp.size() != hs.n_row_shapes) {
    e->shape_stamp.assign(hs.n_row_shapes, 0);
    e->shape_slot_of.assign(hs.n_row_shapes, 0);
    e->stamp = 0;
  }
  e->stamp++;
  uint32_t ns = 0;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t sh = hs.t_row_shape[rows[i]];
    if (e->shape_stamp[sh] != e->stamp) {
      e->shape_stamp[sh] = e->stamp;
      e->shape_slot_of[sh] = ns;
      e->h_mrows[ns] = rows[i];
      ns++;
    }
    e->h_slot[i] = e->shape_slot_of[sh];
  }
  return ns;
}

// ---- one device round, in three host steps so the multi-GPU path can interleave its collectives ----
struct RoundCtx {
  KbRound r{};
  KbDev d{};
  uint32_t n = 0, ns = 0, L = 0;
  bool backfill = false;
  bool direct = false;           // the kernels read the window from the pinned staging block (no copy command)
  uint32_t buf = 0;              // which half of the pinned upload / download blocks the round uses (chained rounds alternate)
  unsigned long long seq = 0;    // the sequence number its commit kernel publishes
  bool overlapped = false;       // its candidate lists were built on the second stream and repaired (round_candidates_overlapped)
};

// upload the window e->h_rows[0..n) (task ids, shape slots, representative rows) and build the row descriptors
// `rows` (default e->h_rows) is the window; `buf` selects the half of the pinned staging blocks; a non-zero `chain_expect` queues
// the round behind a predecessor whose result the host has not seen yet (KbRound::chain)
RoundCtx round_prepare(kb_engine *e, uint32_t n, int fit_mode, bool backfill, bool gather_in_matrix = false, const uint32_t *rows = nullptr,
                       uint32_t buf = 0, uint32_t chain_expect = 0) {
  RoundCtx c;
  if (!rows) rows = e->h_rows.data();
  ensure_window_buffers(e, n);
  ensure_matrix_buffers(e, n, n + 1);
  c.n = n;
  c.ns = assign_shapes(e, n, rows);
  c.L = n + 1;   // more candidates than the round can dirty: a clean one always survives
  c.backfill = backfill;
  c.buf = buf;
  // one staging copy per round: [task rows | shape slots | representative rows], fixed offsets
  uint32_t *hw = e->h_win.data() + (size_t)buf * 3 * KB_K5_MAX_WINDOW;
  std::memcpy(hw, rows, sizeof(uint32_t) * n);
  std::memcpy(hw + KB_K5_MAX_WINDOW, e->h_slot.data(), sizeof(uint32_t) * n);
  std::memcpy(hw + 2 * KB_K5_MAX_WINDOW, e->h_mrows.data(), sizeof(uint32_t) * c.ns);
  // single-GPU fast rounds: the descriptor gather and the matrix kernel read the staged window straight from the pinned block
  // (a few hundred 4-byte reads over PCIe, overlapped with the matrix evaluation) instead of waiting for a 7 us copy command
  const bool direct = gather_in_matrix && e->fast_rounds && e->direct_window;
  if (!direct) HIP_OK(hipMemcpyAsync(e->b_win.p, hw, sizeof(uint32_t) * (2 * KB_K5_MAX_WINDOW + c.ns), hipMemcpyHostToDevice, e->stream));
  c.d = e->dev;
  if (backfill) {
    c.d.score_enabled = 0;   // backfill.go:50-66 takes the first node that passes the predicates: all scores tie
    c.d.t_init = e->t_fit;   // ... and on which ssn.Allocate's AddTask succeeds: Resreq.LessEqual(Idle), fit_mode 2
  }
  c.r = make_round(e, n, c.ns, c.L, fit_mode, backfill, buf);
  if (direct) {
    const uint32_t *dw = e->d_hwin + (size_t)buf * 3 * KB_K5_MAX_WINDOW;
    c.r.rows = dw;
    c.r.shape_slot = dw + KB_K5_MAX_WINDOW;
    c.r.mrows = dw + 2 * KB_K5_MAX_WINDOW;
  }
  c.direct = direct;
  c.seq = ++e->seq;
  c.r.chain = e->b_chain.as<uint32_t>();
  c.r.chain_expect = chain_expect;
  c.r.chain_tag = (uint32_t)(c.seq & 0x7FFFFFFFull) + 1u;   // never 0
  c.r.gather = (gather_in_matrix && c.ns > 0) ? 1u : 0u;
  if (!c.r.gather) kb_launch_gather(c.d, c.r, e->stream);
  return c;
}

// K1 + K3 for matrix rows [m0, m1) of the round; keys go to `keys` (row m0 first)
void round_candidates(kb_engine *e, const RoundCtx &c, uint32_t m0, uint32_t m1, unsigned long long *keys) {
  if (m1 <= m0) return;
  KbRound r = c.r;
  r.mrows = c.r.mrows + m0;
  r.n_mrows = m1 - m0;
  r.keys = keys;
  if (e->fast_rounds) {   // kernel times come from the wall-clock stamps the kernels leave in the output block
    kb_launch_matrix(c.d, r, e->stream);
    kb_launch_affinity(c.d, r, e->stream);
    kb_launch_interpod(c.d, r, e->stream);
    kb_launch_argmax(c.d, r, e->stream);
  } else {
    Timer &t1 = get_timer(e, 0), &t3 = get_timer(e, 1);
    HIP_OK(hipEventRecord(t1.a, e->stream));
    kb_launch_matrix(c.d, r, e->stream);
    kb_launch_affinity(c.d, r, e->stream);
    kb_launch_interpod(c.d, r, e->stream);
    HIP_OK(hipEventRecord(t1.b, e->stream));
    HIP_OK(hipEventRecord(t3.a, e->stream));
    kb_launch_argmax(c.d, r, e->stream);
    HIP_OK(hipEventRecord(t3.b, e->stream));
  }
  e->stats.matrix_launches += 1;
  e->stats.matrix_evals += (uint64_t)(m1 - m0) * e->hs.N;
}

// The same for a chained round, overlapped with its predecessor (n_prev rows, running or queued on the first stream): matrix + arg-max on
// the second stream with lists of n_prev + L entries, then — first stream, i.e. behind the predecessor's commit kernel — the repair that
// waits for the lists, re-evaluates the predecessor's nodes and merges (kb_repair.hpp): workgroups of the selection kernel's own launch, or
// (the run kernel's rounds, KB_FUSE_REPAIR=0) a launch in front of it (kb_kernels.hip: k_repair).  The host's order guarantees that
// every round before the predecessor has been COLLECTED when this is called (run_action plans a window only after it has the answer of the
// round two in front of it), so the only nodes that can change under the second stream's launches are the predecessor's.
void ensure_overlap_buffers(kb_engine *e, uint32_t mrows, uint32_t stale_L) {
  const size_t NP = e->dev.NP;
  if (!e->stream_b) {
    HIP_OK(hipStreamCreateWithFlags(&e->stream_b, hipStreamNonBlocking));
  }
  if (mrows > e->mat2_cap) {
    HIP_OK(hipStreamSynchronize(e->stream_b));
    e->b_score2.alloc(sizeof(uint16_t) * (size_t)mrows * NP);
    e->b_maskw2.alloc(sizeof(uint32_t) * (size_t)mrows * (NP / 32));
    e->mat2_cap = mrows;
  }
  const size_t need = (size_t)mrows * stale_L;
  if (need > e->stale_cap) {
    HIP_OK(hipStreamSynchronize(e->stream_b));
    HIP_OK(hipStreamSynchronize(e->stream));
    e->b_stale.alloc(sizeof(unsigned long long) * 2 * need);
    e->stale_cap = need;
  }
  if (!e->b_ready.p) {
    e->b_ready.alloc(sizeof(uint32_t) * 2 * KB_K5_MAX_WINDOW);
    HIP_OK(hipMemset(e->b_ready.p, 0, e->b_ready.bytes));
    e->b_task_rows.alloc((size_t)64 * 2 * KB_K5_MAX_WINDOW);
    e->b_lready.alloc(sizeof(uint32_t) * 2 * KB_K5_MAX_WINDOW);
    HIP_OK(hipMemset(e->b_lready.p, 0, e->b_lready.bytes));
    e->h_cand_out.flags = hipHostMallocMapped | hipHostMallocCoherent;
    e->h_cand_out.resize(2 * KB_OUT_HDR);
    std::memset(e->h_cand_out.data(), 0, sizeof(unsigned long long) * 2 * KB_OUT_HDR);
    HIP_OK(hipHostGetDevicePointer((void **)&e->d_cand_out, e->h_cand_out.data(), 0));
  }
}
void round_candidates_overlapped(kb_engine *e, RoundCtx &c, uint32_t n_prev, unsigned long long *keys) {
  if (c.ns == 0) return;
  const uint32_t stale_L = n_prev + c.L;
  unsigned long long *stale = e->b_stale.as<unsigned long long>() + (size_t)c.buf * e->stale_cap;
  uint32_t *ready = e->b_ready.as<uint32_t>() + (size_t)c.buf * KB_K5_MAX_WINDOW;
  KbRound rb = c.r;   // the second stream's view: runs whatever the chain word says (the predecessor has not written it yet)
  rb.chain_expect = 0;
  rb.score = e->b_score2.as<uint16_t>();
  rb.maskw = e->b_maskw2.as<uint32_t>();
  rb.keys = stale;
  rb.L = stale_L;
  rb.result = reinterpret_cast<uint32_t *>(e->d_cand_out + (size_t)c.buf * KB_OUT_HDR);   // their time stamps, apart from the round's timeline (round_collect reads them)
  rb.ready = ready;
  rb.ready_tag = c.r.chain_tag;   // the round's own tag (sequence number folded to 31 bits, + 1): unique among the rounds in flight, never the 0 the words start from
  rb.task_rows = reinterpret_cast<unsigned char *>(e->b_task_rows.p) + (size_t)c.buf * 64 * KB_K5_MAX_WINDOW;
  kb_launch_matrix(c.d, rb, e->stream_b);        // also gathers the row descriptors into this round's half (gather == 1)
  kb_launch_argmax(c.d, rb, e->stream_b);
  KbRound ra = c.r;   // first stream: behind the predecessor's commit kernel
  ra.keys = keys;
  ra.ready = ready;
  ra.ready_tag = c.r.chain_tag;
  ra.task_rows = rb.task_rows;
  ra.stale = stale;
  ra.stale_L = stale_L;
  ra.prev_dec = e->b_out.as<unsigned long long>() + KB_OUT_HDR;   // the predecessor's decision records (it completed, or the chain is broken)
  ra.n_prev = n_prev;
  if (e->fuse_repair && e->commit_kernel == KB_COMMIT_SELECT) {
    // The selection kernel's launch carries the repair workgroups itself (kb_commit_sel.hip, kb_repair.hpp): they start with the commit
    // workgroup instead of a launch earlier — a kernel boundary, a launch latency and the commit prologue's staging off the dependent chain
    // of every round.  round_commit launches with these fields; this round's commit kernel overwrites prev_dec (the result block) in its
    // epilogue, i.e. behind its wait for the repaired lists.
    c.r.ready = ra.ready; c.r.ready_tag = ra.ready_tag; c.r.task_rows = ra.task_rows; c.r.stale = ra.stale; c.r.stale_L = ra.stale_L;
    c.r.prev_dec = ra.prev_dec; c.r.n_prev = ra.n_prev;
    c.r.lists_ready = e->b_lready.as<uint32_t>() + (size_t)c.buf * KB_K5_MAX_WINDOW;
    c.r.lists_tag = c.r.chain_tag;
  } else {
    kb_launch_repair(c.d, ra, e->stream);
  }
  e->stats.matrix_launches += 1;
  e->stats.matrix_evals += (uint64_t)c.ns * e->hs.N;
  e->overlapped_rounds += 1;
}

// K5 over the whole window with the complete candidate table `keys` [ns][L]
void round_commit(kb_engine *e, const RoundCtx &c, unsigned long long *keys, double *delta, uint32_t own0, uint32_t own1) {
  KbRound r = c.r;
  r.keys = keys;
  r.delta = delta;
  r.own_row0 = own0;
  r.own_row1 = own1;
  auto launch = [&]() {
    const int kern = e->commit_kernel;
    e->commit_kernel_of[c.buf] = kern;
    if (kern == KB_COMMIT_RUN) { kb_launch_commit(c.d, r, e->stream); e->rounds_run++; }
    else { kb_launch_commit_sel(c.d, r, e->stream); e->rounds_sel++; }
  };
  if (e->fast_rounds) {
    r.host_out = e->d_hout + (size_t)c.buf * KB_OUT_STRIDE;
    r.seq = c.seq;
    launch();
    // a launch the runtime refuses (too much LDS, a bad attribute) never publishes its round: say so now instead of after the watchdog's ten seconds
    HIP_OK(hipGetLastError());
    return;
  }
  Timer &t5 = get_timer(e, 2);
  HIP_OK(hipEventRecord(t5.a, e->stream));
  launch();
  HIP_OK(hipEventRecord(t5.b, e->stream));
  HIP_OK(hipMemcpyAsync(e->h_out.data(), e->b_out.p, sizeof(unsigned long long) * (KB_OUT_HDR + c.n), hipMemcpyDeviceToHost, e->stream));
}


// wait for the round, account the kernel times, unpack the decision records
void round_collect(kb_engine *e, const RoundCtx &c, bool had_candidates, uint32_t &n_done, uint32_t &reason) {
  const unsigned long long *ho = e->h_out.data() + (e->fast_rounds ? (size_t)c.buf * KB_OUT_STRIDE : 0);
  const uint32_t *h_result = reinterpret_cast<const uint32_t *>(ho);
  if (e->fast_rounds) {
    // the commit kernel publishes the round's sequence number into pinned host memory after everything else
    volatile const unsigned long long *seqw = ho + KB_OUT_SEQ;
    const double t0 = now_ms();
    uint32_t spins = 0;
    const bool spin_only = (e->flags & KB_FLAG_SPIN_WAIT) != 0;   // include/kb_engine.h: a short spin, then the core is given up between polls (default); or spin throughout
    bool yielding = false;
    while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != c.seq) {
      if (yielding) sched_yield(); else __builtin_ia32_pause();
      ++spins;
      if (!spin_only && !yielding && (spins & 0x3Fu) == 0 && now_ms() - t0 > KB_WAIT_SPIN_US * 1e-3) yielding = true;
      if ((spins & 0xFFFFu) == 0 && now_ms() - t0 > 10000.0) {   // a faulted kernel never publishes: surface the HIP error
        HIP_OK(hipStreamSynchronize(e->stream));
        HIP_OK(hipGetLastError());
        if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != c.seq) throw EngineError(KB_E_DEVICE, "commit kernel finished without publishing its round");
      }
    }
    n_done = h_result[0];
    reason = h_result[1];
    if (reason == KB_REASON_SKIPPED) return;   // queued behind a round that stopped early: nothing ran
    const unsigned long long *st = ho + KB_OUT_STAMP0;
    const double per_ms = 1.0 / e->wall_khz;
    if (had_candidates && c.overlapped) {
      // matrix: the launch itself, timed on the second stream (start of the matrix launch -> start of the arg-max launch behind it); it ran
      // beside the predecessor's commit kernel, i.e. NOT on the cycle's timeline.  arg-max: what the round waits for on the first stream
      // instead — the repair launch, from its start (its wait for the lists included) to the start of the commit kernel; a commit launch that
      // carries its own repair workgroups has that wait inside commit_ms
      const unsigned long long *cs = e->h_cand_out.data() + (size_t)c.buf * KB_OUT_HDR + KB_OUT_STAMP0;
      if (cs[1] > cs[0]) e->stats.matrix_ms += (double)(cs[1] - cs[0]) * per_ms;
      if (st[2] > st[0]) e->stats.argmax_ms += (double)(st[2] - st[0]) * per_ms;   // (a launch that carries its own repair workgroups: they start WITH the commit workgroup, the wait is inside commit_ms)
      if (st[1] > st[0]) e->tl_repair_tag += (double)(st[1] - st[0]) * per_ms;   // ... of which: until workgroup 0 had seen its list's tag
    } else if (had_candidates) {
      e->stats.matrix_ms += (double)(st[1] - st[0]) * per_ms;   // includes the descriptor gather
      e->stats.argmax_ms += (double)(st[2] - st[1]) * per_ms;
    }
    e->stats.commit_ms += (double)(st[3] - st[2]) * per_ms;
  } else {
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipGetLastError());
    float ms = 0;
    if (had_candidates) {
      HIP_OK(hipEventElapsedTime(&ms, get_timer(e, 0).a, get_timer(e, 0).b));
      e->stats.matrix_ms += ms;
      HIP_OK(hipEventElapsedTime(&ms, get_timer(e, 1).a, get_timer(e, 1).b));
      e->stats.argmax_ms += ms;
    }
    HIP_OK(hipEventElapsedTime(&ms, get_timer(e, 2).a, get_timer(e, 2).b));
    e->stats.commit_ms += ms;
  }
  for (uint32_t i = 0; i < c.n; i++) {
    e->h_decnode[i] = (uint32_t)(ho[KB_OUT_HDR + i] & 0xFFFFFFFFull);
    e->h_deckind[i] = (uint32_t)(ho[KB_OUT_HDR + i] >> 32);
  }
  n_done = h_result[0];
  reason = h_result[1];
  if (reason == KB_REASON_INTERNAL)   // only the selection kernel's bounded waits raise it (kb_commit_sel.hip: K9Sync::err): a hand-over between its waves never arrived
    throw EngineError(KB_E_INTERNAL, "selection commit kernel: a wave's bounded wait ran out (round " + std::to_string(e->round_no) + ", " + std::to_string(h_result[4]) +
                                     " runs and " + std::to_string(n_done) + " of " + std::to_string(c.n) + " rows committed before it)");
  const uint32_t dirty_won = h_result[3];   // rows won by a node the round had already changed
  e->stats.row_fallbacks += dirty_won;
  if (e->commit_kernel_of[c.buf] == KB_COMMIT_SELECT) {
    e->sel_stat[0] += h_result[6] & 0xFFFFu; e->sel_stat[1] += h_result[6] >> 16; e->sel_stat[2] += h_result[7] & 0xFFFFu; e->sel_stat[3] += h_result[7] >> 16;
    e->stats.rounds_select += 1;
    e->stats.select_runs_clean += h_result[6] & 0xFFFFu; e->stats.select_runs_shots += h_result[6] >> 16; e->stats.select_shots += h_result[7] >> 16;
  }
  {
    e->k5_slots += h_result[2];
    e->k5_walks += h_result[4];
    e->k5_rescans += h_result[5];
    for (int k = 0; k < 10; k++)   // zero unless built with -DKB_K9_TRACE
      e->k5_trace[k] += (double)(uint32_t)(ho[(k < 6 ? 5 + k / 2 : 13 + (k - 6) / 2)] >> (32 * (k & 1)));
    if (e->commit_kernel_of[c.buf] == KB_COMMIT_SELECT && e->k5_trace[0] > 0)   // the selection kernel's trace build: its other waves' evaluation phase
      for (int k = 10; k < 14; k++) e->k5_trace[k] += (double)(uint32_t)(ho[k < 12 ? 4 : 15] >> (32 * (k & 1)));
  }
  if (n_done) {
    const double share = (double)dirty_won / (double)n_done;
    e->dirty_share = e->stats.rounds == 0 ? share : 0.75 * e->dirty_share + 0.25 * share;
  }
  e->stats.rounds += 1;
  e->round_no += 1;
}

void check_aggregates(kb_engine *e, const OrderMachine &om) {
  // the host's running drf / proportion / gang aggregates must equal the device reduction bit for bit
  const HostSession &hs = e->hs;
  for (size_t i = 0; i < hs.job_ready.size(); i++)
    if (om.ready[i] != hs.job_ready[i]) throw EngineError(KB_E_INTERNAL, "gang ready count diverged from the device ballot at job " + std::to_string(i));
  if (e->pol.has_drf)
    for (size_t i = 0; i < hs.job_share.size(); i++)
      if (om.jshare[i] != hs.job_share[i]) throw EngineError(KB_E_INTERNAL, "drf share diverged from the device reduction at job " + std::to_string(i));
  if (e->pol.has_proportion)
    for (uint32_t q = 0; q < hs.Q; q++)
      if (hs.queue_has_attr[q] && om.qshare[q] != hs.queue_share[q])
        throw EngineError(KB_E_INTERNAL, "proportion share diverged from the device reduction at queue " + std::to_string(q));
}

// Host side of one action as a resumable object: plan() fills e->h_rows with the next window, absorb() digests the
// device's answer (confirm, or roll back + replay on a mis-speculated round), finish() runs the gang/share reduction.
// probes of more (shape, node) pairs than this run at every fourth break only (ActionRun::probe_launch); -DKB_PROBE_SPARSE_ABOVE=... for an A/B build
#ifndef KB_PROBE_SPARSE_ABOVE
#define KB_PROBE_SPARSE_ABOVE (8ull << 20)
#endif
struct ActionRun {
  uint32_t action = 0;   // 0 allocate, 1 backfill
  bool bf_need_pred = false;
  std::vector<int> bf_podcnt;
  std::vector<unsigned long long> bf_ports, bf_ports_x;   // word 0 [NP]; the words behind it [port_xw][NP]
  OrderMachine om;
  std::vector<uint8_t> dead;
  std::vector<kb_decision> decs;
  std::vector<uint32_t> bf_list;
  size_t bf_pos = 0;
  uint64_t popped = 0, spec_pops = 0, spec_pops_next = 0;
  std::vector<uint32_t> rows_next;   // the window speculated behind the one in flight
  std::vector<uint32_t> probe_list;  // feasibility shapes the probe looks at (the ones still alive)
  uint32_t probe_calls = 0;
  double host_ms = 0, t_start = 0;
  bool active = false;

  // A task shape with no feasible node stays infeasible for the rest of the action (idle only shrinks, releasing does not
  // grow).  The same holds for every shape of the same static class whose compared InitResreq is >= in every dimension:
  // LessEqual is monotone in its left operand, so that shape's feasible set is a subset of an empty set.  Marking them
  // now saves the device round each would otherwise end.
  void mark_dead(const HostSession &hs, uint32_t x) {
    const int R = hs.R;
    // inter-pod affinity: a shape that REQUIRES a matching pod in the node's domain gains nodes as pods are placed: never dead.
    // Forbidding checks only shrink the feasible set (counts only grow inside allocate / backfill): dead stays dead, and a shape
    // with the same checks and a larger request is dominated as usual.
    if (!hs.feas_ip_require.empty() && hs.feas_ip_require[x]) return;
    const double *ex = &hs.feas_eff[(size_t)x * R];
    for (uint32_t y = 0; y < hs.n_feas_shapes; y++) {
      if (dead[y] || hs.feas_cls[y] != hs.feas_cls[x] || hs.feas_conf[y] != hs.feas_conf[x]) continue;
      if (hs.port_xw && std::memcmp(&hs.t_conf_x[(size_t)hs.feas_rep[y] * hs.port_xw], &hs.t_conf_x[(size_t)hs.feas_rep[x] * hs.port_xw], sizeof(uint64_t) * hs.port_xw) != 0) continue;
      if (!hs.feas_ip.empty() && hs.feas_ip[y] != hs.feas_ip[x]) continue;
      const double *ey = &hs.feas_eff[(size_t)y * R];
      bool ge = true;
      for (int d = 0; d < R && ge; d++) ge = ey[d] >= ex[d];
      if (ge) dead[y] = 1;
    }
    dead[x] = 1;
  }

  void begin(kb_engine *e, uint32_t act) {
    HostSession &hs = e->hs;
    action = act;
    decs.clear();
    popped = spec_pops = 0;
    host_ms = 0;
    t_start = now_ms();
    active = true;
    ensure_window_buffers(e, e->eff_window);
    if (action == 0) {
      double t0 = now_ms();
      om.init_allocate(&hs, &e->pol);
      host_ms += now_ms() - t0;
      dead.assign(hs.n_feas_shapes ? hs.n_feas_shapes : 1, 0);
    } else {
      // backfill.go:44-47: jobs ascending JobID, Pending tasks ascending UID with an empty InitResreq; the order does not
      // depend on outcomes, so there is nothing to speculate
      bf_list.clear();
      bf_pos = 0;
      for (uint32_t t : hs.init_empty_tasks)
        if (hs.t_status[t] == KB_TASK_PENDING && hs.t_job[t] < hs.J) bf_list.push_back(t);
      // Only a session with sub-epsilon BestEffort requests (or a node below -epsilon) can see AddTask refuse a node that passed
      // the predicates; absorb() then needs to tell "no node passes the predicates" (the task stays Pending) from "one did"
      // (outside the envelope).  Pod counts and used ports only grow during backfill, so the state as of now decides the former.
      bf_need_pred = e->idle_below_eps;
      if (hs.has_interpod)
        for (uint32_t t : bf_list)
          if (hs.t_res[t] != 0.0 || hs.t_res[(size_t)hs.T + t] != 0.0)
            throw EngineError(KB_E_UNSUPPORTED, "BestEffort task with a sub-epsilon request in a session with inter-pod affinity");
      for (uint32_t t : bf_list) bf_need_pred = bf_need_pred || hs.t_res[t] != 0.0 || hs.t_res[(size_t)hs.T + t] != 0.0;
      if (bf_need_pred) {
        const uint32_t NP = e->dev.NP;
        bf_podcnt.resize(NP); bf_ports.assign(NP, 0);
        HIP_OK(hipMemcpyAsync(bf_podcnt.data(), e->b_podcnt.p, sizeof(int) * NP, hipMemcpyDeviceToHost, e->stream));
        if (e->dev.ports) HIP_OK(hipMemcpyAsync(bf_ports.data(), e->b_ports.p, sizeof(unsigned long long) * NP, hipMemcpyDeviceToHost, e->stream));
        bf_ports_x.assign((size_t)e->dev.port_xw * NP, 0);
        if (e->dev.port_xw) HIP_OK(hipMemcpyAsync(bf_ports_x.data(), e->b_ports_x.p, sizeof(unsigned long long) * bf_ports_x.size(), hipMemcpyDeviceToHost, e->stream));
        HIP_OK(hipStreamSynchronize(e->stream));
      }
    }
  }

  // a window holds at most shape_cap distinct task shapes (one lane of the commit kernel's main wave each)
  static void new_window(kb_engine *e) {
    if (e->plan_stamp.size() != e->hs.n_row_shapes) { e->plan_stamp.assign(e->hs.n_row_shapes ? e->hs.n_row_shapes : 1, 0); e->plan_epoch = 0; }
    e->plan_epoch++;
  }
  static bool admit_shape(kb_engine *e, uint32_t shape, uint32_t &nshapes) {
    if (e->plan_stamp[shape] == e->plan_epoch) return true;
    if (nshapes >= e->shape_cap) return false;
    e->plan_stamp[shape] = e->plan_epoch;
    nshapes++;
    return true;
  }

  uint32_t plan(kb_engine *e) {
    HostSession &hs = e->hs;
    const uint32_t W = e->eff_window;
    if (action == 1) {
      uint32_t n = 0, nshapes = 0;
      new_window(e);
      while (n < W && bf_pos + n < bf_list.size() && !(n > 0 && !hs.t_ip_subject.empty() && hs.t_ip_subject[bf_list[bf_pos + n]]) &&
             !(n > 0 && hs.wide(bf_list[bf_pos + n])) && admit_shape(e, hs.t_row_shape[bf_list[bf_pos + n]], nshapes)) {   // an inter-pod subject heads its window
        e->h_rows[n] = bf_list[bf_pos + n];
        n++;
        if (hs.wide(e->h_rows[n - 1])) break;   // a pod whose host-port masks reach beyond word 0: a round of its own (kb_host.hpp)
      }
      return n;
    }
    double t0 = now_ms();
    om.checkpoint();   // roll-back point for a mis-speculated round
    uint32_t n = 0, t, nshapes = 0;
    spec_pops = 0;
    new_window(e);
    while (n < W && om.next(t)) {
      spec_pops++;
      if (dead[hs.t_feas_shape[t]]) { om.report(Outcome::NoFeasibleNode); continue; }   // known: feasibility only shrinks inside one action
      if (!admit_shape(e, hs.t_row_shape[t], nshapes) || (n > 0 && !hs.t_ip_subject.empty() && hs.t_ip_subject[t]) || (n > 0 && hs.wide(t))) {
        om.rollback_last_pop(); spec_pops--; break;   // the task heads the next window (shape budget, an inter-pod subject: fresh matrix, or host-port masks beyond word 0)
      }
      e->h_rows[n++] = t;
      om.report(Outcome::Allocated);
      if (hs.wide(t)) break;   // ... and is that window's only row: the commit kernels keep to word 0 of the masks (kb_host.hpp: t_wide)
    }
    host_ms += now_ms() - t0;
    if (n == 0) popped += spec_pops;
    return n;
  }

  // While the device works on the window just launched, speculate the one after it (assuming the one in flight completes,
  // which ~70 % do) behind a second roll-back point; promote() makes it the current window, a break rolls both back.
  uint32_t plan_ahead(kb_engine *e) {
    HostSession &hs = e->hs;
    const uint32_t W = e->eff_window;
    double t0 = now_ms();
    om.push_checkpoint();
    if (rows_next.size() < W) rows_next.resize(W);
    uint32_t n = 0, t, nshapes = 0;
    spec_pops_next = 0;
    new_window(e);
    while (n < W && om.next(t)) {
      spec_pops_next++;
      if (dead[hs.t_feas_shape[t]]) { om.report(Outcome::NoFeasibleNode); continue; }
      if (!admit_shape(e, hs.t_row_shape[t], nshapes) || (n > 0 && !hs.t_ip_subject.empty() && hs.t_ip_subject[t]) || (n > 0 && hs.wide(t))) { om.rollback_last_pop(); spec_pops_next--; break; }
      rows_next[n++] = t;
      om.report(Outcome::Allocated);
      if (hs.wide(t)) break;
    }
    host_ms += now_ms() - t0;
    return n;
  }
  void promote(kb_engine *e, uint32_t n_next) {
    om.pop_commit();
    if (n_next) std::memcpy(e->h_rows.data(), rows_next.data(), sizeof(uint32_t) * n_next);
    spec_pops = spec_pops_next;
    if (n_next == 0) popped += spec_pops;
  }

  // the plugin predicates of task t (predicates.go:127,181-190 and the static class table) against the pod counts / ports
  // backfill started from: a superset of the nodes that pass at any later point of the action
  bool passed_predicates_at_start(kb_engine *e, uint32_t t) const {
    const HostSession &hs = e->hs;
    if (!e->pol.pred_enabled) return hs.N > 0;
    const uint64_t conf = hs.t_conf.empty() ? 0 : hs.t_conf[t];
    for (uint32_t n = 0; n < hs.N; n++) {
      if (hs.n_maxpods[n] <= bf_podcnt[n]) continue;
      if (!hs.compat.empty()) {
        const uint32_t bit = hs.t_cls[t] * hs.n_nc + hs.n_cls[n];
        if (!((hs.compat[bit >> 3] >> (bit & 7)) & 1)) continue;
      }
      if (bf_ports[n] & conf) continue;
      bool clash = false;
      for (uint32_t w = 0; w < hs.port_xw && !clash; w++) clash = (bf_ports_x[(size_t)w * e->dev.NP + n] & hs.t_conf_x[(size_t)t * hs.port_xw + w]) != 0;
      if (clash) continue;
      return true;
    }
    return false;
  }

  // At a speculation break the device is idle and the host is about to re-plan anyway: every feasibility shape that is still
  // alive is evaluated against the current node state (one launch, feasibility only), and whatever has no node left is marked dead
  // NOW instead of costing a break of its own when its next task comes up.  Exact: inside the allocate action a shape without a
  // feasible node stays without one (the argument of mark_dead), so the reference's PredicateNodes will find none either when it
  // pops such a task.  In two halves: probe_launch() right behind the answer of the round that broke (same stream: behind that round's
  // commit kernel and the skipped round queued behind it; the node state it reads is final), probe_collect() in front of the re-plan —
  // the host absorbs the answer (roll-back + replay, ~12 us) while the kernel runs.  The list is built from `dead` as the broken round was
  // planned with; what absorb() marks meanwhile (the row that broke, the shapes it dominates) the probe finds dead again: not counted twice.
  // No planned window is outstanding between the two halves, and absorb() of the allocate action launches nothing (sessions with host-port
  // masks of several words, whose absorb() updates node words on the stream, probe behind it: run_action).
  uint32_t probe_S = 0;   // rows of the probe in flight (0: none)
  void probe_launch(kb_engine *e) {
    HostSession &hs = e->hs;
    probe_S = 0;
    if (!e->probe_enabled || action != 0 || hs.has_interpod || hs.n_feas_shapes == 0 || !e->pol.pred_enabled) return;
    // only the shapes that are still alive are looked at, and when that is a large matrix (many shapes x many nodes: a launch of
    // a few hundred microseconds) only every fourth break pays for it; the deaths of the breaks in between are found then
    probe_calls++;
    probe_list.clear();
    for (uint32_t f = 0; f < hs.n_feas_shapes; f++)
      if (!dead[f]) probe_list.push_back(f);
    const uint32_t S = (uint32_t)probe_list.size();
    if (S == 0) return;
    if ((uint64_t)S * hs.N > KB_PROBE_SPARSE_ABOVE && (probe_calls & 3u) != 1u) return;
    for (uint32_t i = 0; i < S; i++) { e->h_probe_rows[i] = hs.feas_rep[probe_list[i]]; e->h_probe_alive[i] = 0u; }
    kb_launch_probe(e->dev, e->d_probe_rows, S, e->d_probe_alive, e->stream);
    probe_S = S;
  }
  void probe_collect(kb_engine *e) {
    if (probe_S == 0) return;
    const uint32_t S = probe_S;
    probe_S = 0;
    HIP_OK(hipStreamSynchronize(e->stream));
    e->probes++;
    for (uint32_t i = 0; i < S; i++)   // no dominance scan needed: the probe looked at every live shape itself
      if (e->h_probe_alive[i] == 0 && !dead[probe_list[i]]) { dead[probe_list[i]] = 1; e->probe_deaths++; }
  }
  void probe_abandon(kb_engine *e) {   // an exception between the halves: the kernel must not outlive the call (it writes into h_probe_alive)
    if (probe_S) { (void)hipStreamSynchronize(e->stream); probe_S = 0; }
  }
  void probe_dead_shapes(kb_engine *e) { probe_launch(e); probe_collect(e); }

  // host-port masks of several words: the placed pod's words behind the first join the node's (both ssn.Allocate and ssn.Pipeline end in
  // NodeInfo.AddTask; the kernels advanced word 0).  On the action's stream, in front of whatever the next round launches.
  void absorb(kb_engine *e, uint32_t n, uint32_t n_done, uint32_t reason) {
    const size_t first = decs.size();
    absorb_round(e, n, n_done, reason);
    if (e->dev.port_xw)
      for (size_t i = first; i < decs.size(); i++)
        if (e->hs.wide(decs[i].task) && decs[i].node != KB_NONE) kb_launch_or_ports_x(e->dev, decs[i].task, decs[i].node, e->stream);
  }
  void absorb_round(kb_engine *e, uint32_t n, uint32_t n_done, uint32_t reason) {
    HostSession &hs = e->hs;
    const uint32_t round = (uint32_t)(e->round_no - 1);
    if (action == 1) {
      if (reason != KB_REASON_DONE || n_done != n) throw EngineError(KB_E_INTERNAL, "backfill round ended early");
      for (uint32_t i = 0; i < n; i++) {
        const uint32_t t = e->h_rows[i];
        if (e->h_decnode[i] != KB_NONE) { decs.push_back(kb_decision{t, e->h_decnode[i], 0u, round}); continue; }
        // No node took the task.  With a zero request that means no node passes the predicates and the task stays Pending
        // (unless a node's Idle sat at or below -epsilon in the snapshot).  With a non-zero sub-epsilon request a node may have passed
        // the predicates and failed AddTask: ssn.Allocate has then flipped the task to Allocated without a node
        // (session.go:243 before :255), and what a later dispatch of that job does with it depends on Go's map order.
        if ((hs.t_res[t] != 0.0 || hs.t_res[(size_t)hs.T + t] != 0.0 || e->idle_below_eps) && passed_predicates_at_start(e, t))
          throw EngineError(KB_E_UNSUPPORTED, "BestEffort task with a sub-epsilon request found no node (the reference may leave it Allocated without one)");
      }
      bf_pos += n;
      return;
    }
    double t0 = now_ms();
    if (reason == KB_REASON_DONE) {
      popped += spec_pops;
      for (uint32_t i = 0; i < n; i++) decs.push_back(kb_decision{e->h_rows[i], e->h_decnode[i], e->h_deckind[i], round});
    } else {
      // replay the confirmed prefix on the checkpoint, then feed the true outcome of the row that broke the speculation
      e->stats.spec_breaks += 1;
      om.rollback();
      uint32_t i = 0, t;
      for (;;) {
        if (!om.next(t)) throw EngineError(KB_E_INTERNAL, "order replay ran out of tasks");
        popped++;
        if (dead[hs.t_feas_shape[t]]) { om.report(Outcome::NoFeasibleNode); continue; }
        if (t != e->h_rows[i]) throw EngineError(KB_E_INTERNAL, "order replay diverged from the speculated sequence");
        if (reason == KB_REASON_NO_FEASIBLE && i == n_done) {
          mark_dead(hs, hs.t_feas_shape[t]);
          om.report(Outcome::NoFeasibleNode);
          break;
        }
        if (reason == KB_REASON_SKIPPED && i == n_done) {
          // only an overlapped round whose candidate lists never arrived skips itself behind a predecessor that completed (k_repair's
          // bounded wait): nothing was decided; the task heads the next window, which goes the plain way — and so does every round of this
          // engine from now on (overlap_faults is never cleared: a launch that got lost on the second stream is not expected to heal)
          e->overlap_faults += 1;
          om.rollback_last_pop();
          popped--;
          break;
        }
        if (reason == KB_REASON_RENORM && i == n_done) {
          // the device stopped in front of this task (its score must be normalised over a fresh feasible set): nothing was
          // decided for it; undo the pop so that it heads the next window
          om.rollback_last_pop();
          popped--;
          break;
        }
        decs.push_back(kb_decision{t, e->h_decnode[i], e->h_deckind[i], round});
        om.report(e->h_deckind[i] ? Outcome::Pipelined : Outcome::Allocated);
        i++;
        if (reason == KB_REASON_PIPELINED && i == n_done) break;
      }
    }
    host_ms += now_ms() - t0;
  }

  void finish(kb_engine *e) {
    HostSession &hs = e->hs;
    // every ssn.Allocate / ssn.Pipeline fires proportion's AllocateFunc -> updateShare for the task's queue (proportion.go:212-223)
    for (const kb_decision &dc : decs) {
      const uint32_t q = hs.job_queue[hs.t_job[dc.task]];
      if (q < hs.Q) hs.queue_share_live[q] = 1;
    }
    // an action that decided nothing left the task table as the last reduction saw it (every call that changes it ends with one):
    // the host mirrors are current, nothing to recount (a cycle's backfill usually finds no BestEffort task at all)
    const double t_fin0 = now_ms();
    if (!decs.empty()) run_finalize(e);
    e->tl_finish += now_ms() - t_fin0;
    if (action == 0) {
      check_aggregates(e, om);
      e->stats.tasks_popped += popped;
      e->stats.evals += popped * (uint64_t)hs.N;   // PredicateNodes visits every node for every popped task (allocate.go:143)
    } else {
      e->stats.tasks_popped += bf_list.size();
      // the reference stops at the first node that passes: count the nodes it actually visits
      uint64_t ev = 0;
      std::vector<uint8_t> placed(hs.T, 0);
      for (auto &dcs : decs) { placed[dcs.task] = 1; ev += (uint64_t)dcs.node + 1; }
      for (uint32_t t : bf_list) if (!placed[t]) ev += hs.N;
      e->stats.evals += ev;
    }
    e->stats.decisions += decs.size();
    e->stats.host_order_ms += host_ms;
    e->stats.total_ms += now_ms() - t_start;
    active = false;
  }
};

}  // namespace

struct MgState {
  ActionRun run;
  RoundCtx ctx;
  bool in_round = false, committed = false, had_candidates = false;
  uint32_t n_done = 0, reason = 0;
  std::vector<kb_decision> last_decs;
  DevBuf s_idle, s_rel, s_nzc, s_nzm, s_podcnt;   // node state at round start
  // the deferred cross-check (kb_round_check): the state at the start of the round BEFORE the current one, a device counter of differing
  // values that lives for the action, the rounds begun in it
  DevBuf q_idle, q_rel, q_nzc, q_nzm, q_podcnt, chk_counter;
  uint32_t rounds_begun = 0;
  bool chk_valid = false;   // chk_counter belongs to an action begun since the last load / reset
  KbNodeCopy cur() const { return KbNodeCopy{s_idle.as<double>(), s_rel.as<double>(), s_nzc.as<long long>(), s_nzm.as<long long>(), s_podcnt.as<int>()}; }
  KbNodeCopy prev() const { return KbNodeCopy{q_idle.as<double>(), q_rel.as<double>(), q_nzc.as<long long>(), q_nzm.as<long long>(), q_podcnt.as<int>()}; }
};
static void mg_free(MgState *m) { delete m; }
// kb_session_reset: the round-mode action state goes, its device buffers stay (ten node-state copies + the counter: a sharded cycle resets every
// step, and hipFree synchronises the device — inside the timed step, on the first rounds' critical path)
static void mg_reset(MgState *m) {
  if (!m) return;
  m->run = ActionRun();
  m->in_round = false; m->committed = false; m->had_candidates = false;
  m->n_done = 0; m->reason = 0; m->rounds_begun = 0; m->chk_valid = false;
  m->last_decs.clear();
}

namespace {
int guarded(kb_engine *e, const std::function<void()> &fn) {
  try {
    if (e) HIP_OK(hipSetDevice(e->device));
    fn();
    return KB_OK;
  } catch (const EngineError &ex) {
    if (e) e->err = ex.what(); else g_create_err = ex.what();
    return ex.code;
  } catch (const std::bad_alloc &) {
    if (e) e->err = "out of host memory"; else g_create_err = "out of host memory";
    return KB_E_NOMEM;
  } catch (const std::exception &ex) {
    if (e) e->err = ex.what(); else g_create_err = ex.what();
    return KB_E_INTERNAL;
  } catch (...) {
    if (e) e->err = "unknown error"; else g_create_err = "unknown error";
    return KB_E_INTERNAL;
  }
}
}  // namespace

extern "C" {

const char *kb_last_error(const kb_engine *e) { return e ? e->err.c_str() : g_create_err.c_str(); }

int kb_engine_create(const kb_config *cfg, kb_engine **out) {
  if (!cfg || !out) { g_create_err = "null argument"; return KB_E_INVALID; }
  *out = nullptr;
  kb_engine *e = nullptr;
  int rc = guarded(nullptr, [&]() {
    if (cfg->version != KB_ABI_VERSION) throw EngineError(KB_E_INVALID, "ABI version mismatch");
    std::unique_ptr<kb_engine> eng(new kb_engine());
    eng->pol = compile_policy(cfg);
    eng->device = cfg->device;
    if (cfg->window) eng->window = cfg->window;
    if (eng->window > KB_K5_MAX_ROWS) eng->window = KB_K5_MAX_ROWS;   // one thread of the commit kernel per dirty slot
    eng->commit_batch = cfg->commit_batch;
    eng->flags = cfg->flags;
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0)
      throw EngineError(KB_E_DEVICE, "no HIP device visible: the engine has no CPU fallback (the stock Go action must handle this cycle)");
    if (cfg->device < 0 || cfg->device >= ndev) throw EngineError(KB_E_INVALID, "device ordinal out of range");
    HIP_OK(hipSetDevice(cfg->device));
    HIP_OK(hipStreamCreateWithFlags(&eng->own_stream, hipStreamNonBlocking));
    eng->stream = eng->own_stream;
    eng->b_win.alloc(sizeof(uint32_t) * 3 * KB_K5_MAX_WINDOW);
    eng->h_win.flags = hipHostMallocMapped | hipHostMallocCoherent;   // read by the device directly in single-GPU fast rounds
    eng->h_win.resize(2 * 3 * KB_K5_MAX_WINDOW);   // two halves: a chained round is staged while its predecessor's copy may still be pending
    eng->b_out.alloc(sizeof(unsigned long long) * (KB_OUT_HDR + KB_K5_MAX_WINDOW));
    HIP_OK(hipMemset(eng->b_out.p, 0, eng->b_out.bytes));
    eng->h_out.flags = hipHostMallocMapped | hipHostMallocCoherent;   // written by the commit kernel while the host polls
    eng->h_probe_rows.flags = eng->h_probe_alive.flags = hipHostMallocMapped | hipHostMallocCoherent;   // the probe kernel's rows in / flags out
    eng->h_out.resize(2 * KB_OUT_STRIDE);
    std::memset(eng->h_out.data(), 0, sizeof(unsigned long long) * 2 * KB_OUT_STRIDE);
    eng->b_chain.alloc(sizeof(uint32_t));
    HIP_OK(hipMemsetAsync(eng->b_chain.p, 0, sizeof(uint32_t), eng->stream));
    HIP_OK(hipHostGetDevicePointer((void **)&eng->d_hout, eng->h_out.data(), 0));
    HIP_OK(hipHostGetDevicePointer((void **)&eng->d_hwin, eng->h_win.data(), 0));
    {
      int khz = 0;
      if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, eng->device) == hipSuccess && khz > 0) eng->wall_khz = (double)khz;
      if (const char *ck = getenv("KB_COMMIT_KERNEL")) {
        if (ck[0] == 'r') eng->commit_pin = KB_COMMIT_RUN;
        else if (ck[0] == 's') eng->commit_pin = KB_COMMIT_SELECT;
        if (eng->commit_pin >= 0) eng->commit_kernel = eng->commit_pin;
      }
      const char *sr = getenv("KB_SYNC_ROUNDS");
      eng->fast_rounds = !(eng->flags & KB_FLAG_SYNC_ROUNDS) && !(sr && sr[0] == '1');
      const char *cr = getenv("KB_CHAIN_ROUNDS");   // 0: launch every round only after the previous one was collected (A/B, debugging)
      eng->chain_rounds = !(cr && cr[0] == '0');
      const char *pb = getenv("KB_PROBE");
      eng->probe_enabled = !(pb && pb[0] == '0');
      const char *ov = getenv("KB_OVERLAP");
      eng->overlap = !(ov && ov[0] == '0');
      const char *fr = getenv("KB_FUSE_REPAIR");   // 0: the repair launch of its own between two commit kernels (A/B; the run kernel's rounds always take it)
      eng->fuse_repair = !(fr && fr[0] == '0');
      const char *wf = getenv("KB_DEVICE_WATERFILL");
      eng->device_waterfill = !(wf && wf[0] == '0');
      const char *dw = getenv("KB_DIRECT_WINDOW");
      eng->direct_window = !(dw && dw[0] == '0');
    }
    e = eng.release();
  });
  if (rc == KB_OK) *out = e;
  return rc;
}

void kb_engine_destroy(kb_engine *e) {
  if (!e) return;
  if (getenv("KB_K5_STATS"))
    fprintf(stderr, "[kb K5] rounds on the run kernel %llu, on the selection kernel %llu, last dirty share %.3f\n",
            (unsigned long long)e->rounds_run, (unsigned long long)e->rounds_sel, e->dirty_share);
  if (getenv("KB_K5_STATS") && e->rounds_sel)
    fprintf(stderr, "[kb select] runs of >= 2 rows: all picks clean first placements %llu, committed by shots %llu (shots cut short by a table's end %llu; shots %llu)\n",
            (unsigned long long)e->sel_stat[0], (unsigned long long)e->sel_stat[1], (unsigned long long)e->sel_stat[2], (unsigned long long)e->sel_stat[3]);
  if (getenv("KB_K5_STATS"))
    fprintf(stderr, "[kb K5] rounds %llu, rows %llu, dirty slots %llu, dirty-won rows %llu, runs %llu, runs with a row-specific Resreq %llu (%llu)\n",
            (unsigned long long)e->stats.rounds, (unsigned long long)e->stats.decisions, (unsigned long long)e->k5_slots,
            (unsigned long long)e->stats.row_fallbacks, (unsigned long long)e->k5_walks, (unsigned long long)e->k5_rescans,
            (unsigned long long)e->k5_demand);
  if (getenv("KB_K5_STATS")) fprintf(stderr, "[kb overlap] rounds with candidate lists built beside the predecessor's commit %llu, lists that never arrived %llu; repair launches: %.3f ms from their start to the tag seen\n",
                                     (unsigned long long)e->overlapped_rounds, (unsigned long long)e->overlap_faults, e->tl_repair_tag);
  if (getenv("KB_K5_STATS"))
    fprintf(stderr, "[kb host] ms over the engine's life: reset %.2f, allocate up to its first launch %.2f, speculation breaks (answer -> re-planned launch) %.2f, "
            "closing reductions %.2f, waiting for rounds %.2f, backfill up to its first launch %.2f\n", e->tl_reset, e->tl_begin, e->tl_break, e->tl_finish, e->tl_wait, e->tl_backfill);
  if (getenv("KB_K5_STATS"))
    fprintf(stderr, "[kb host] of the start: order machine %.2f, first probe %.2f, first plan %.2f; of the breaks: probe launch + absorb %.2f, waiting for the probe %.2f, re-plan %.2f\n",
            e->tl_begin_parts[0], e->tl_begin_parts[1], e->tl_begin_parts[2], e->tl_break_parts[0], e->tl_break_parts[1], e->tl_break_parts[2]);
  if (getenv("KB_K5_STATS")) fprintf(stderr, "[kb probe] %llu probes, %llu shapes marked dead by them\n", (unsigned long long)e->probes, (unsigned long long)e->probe_deaths);
  if (getenv("KB_K5_STATS") && e->k5_trace[0] > 0) {
    static const char *ph[10] = {"barrier 1 (wave 0's wait)", "evaluate: candidates (wave 0)", "barrier 2", "rows", "prepare the next run",
                                "rows, of runs with scalar dimensions", "runs with scalar dimensions (count)", "evaluate, of runs with scalar dimensions",
                                "dirty-winner entries (count)", "loop top"};
    static const char *ph_sel[10] = {"wave 0 waits for candidates + dirty keys", "staging: lists, descriptors (10 ns units, not clocks)", "run tables (10 ns units)", "rows: serial loop, tail, publish", "wave 0's loop (10 ns units)",
                                    "selection: entries + first rank", "selection: deep passes", "selection: picks + AddTask", "selection: all picks clean", "shape tables (10 ns units)"};
    const double runs = (double)(e->k5_walks ? e->k5_walks : 1);
    for (int k = 0; k < 10; k++) fprintf(stderr, "[kb K5 trace] %-32s %14.0f clocks  (%.0f per run)\n", e->rounds_sel > e->rounds_run ? ph_sel[k] : ph[k], e->k5_trace[k], e->k5_trace[k] / runs);
    static const char *ph_role[4] = {"prep waves wait for their run's turn (sum of 3)", "prep waves wait for the walk's turn (sum of 3)", "wave 1 waits for the previous run", "wave 1 evaluates + publishes"};
    if (e->rounds_sel > e->rounds_run) for (int k = 10; k < 14; k++) fprintf(stderr, "[kb K5 trace] %-32s %14.0f clocks  (%.0f per run)\n", ph_role[k - 10], e->k5_trace[k], e->k5_trace[k] / runs);
  }
  (void)hipSetDevice(e->device);
  delete e;
}

// proportion's OnSessionOpen water-fill as a launch (kb_waterfill.hip; default since its first device run, round 4; KB_DEVICE_WATERFILL=0: the host loop): the queues' requests, weights and the
// session's total go up, `deserved` comes back for the host's order machine (Overused, the queue order) and stays on the device for
// k_finalize_queues.  build_host_session left hs.deserved at zero.
// Queued on the engine's stream, nothing waited for: the launch leaves `deserved` where k_finalize_queues reads it (b_deserved / b_desmask,
// in their [R][Q] layout) and the queue records and flags travel back into the load's pinned area; waterfill_collect reads them behind
// the synchronisation that ends the load (run_finalize's).
struct WaterfillInFlight { WfQueue *qs = nullptr; WfState *st = nullptr; };
static WaterfillInFlight device_waterfill_queue(kb_engine *e) {
  HostSession &hs = e->hs;
  const uint32_t Q = hs.Q;
  const size_t nq = Q ? Q : 1;
  WaterfillInFlight w;
  w.qs = reinterpret_cast<WfQueue *>(e->load_arena.take(sizeof(WfQueue) * nq));
  w.st = reinterpret_cast<WfState *>(e->load_arena.take(sizeof(WfState)));
  for (size_t q = 0; q < nq; q++) new (&w.qs[q]) WfQueue();
  new (w.st) WfState();
  for (uint32_t q = 0; q < Q; q++) {
    w.qs[q].request = hs.queue_request[q];
    w.qs[q].weight = hs.queue_weight[q];
    w.qs[q].has_attr = hs.queue_has_attr[q];
    w.qs[q].meet = 0;
    w.qs[q].active = 0;
  }
  WfState &st = *w.st;
  st.remaining = hs.total;
  st.total_weight = 0; st.stop = 0; st.share_at_open = 1; st.underflow = 0; st.passes = 0;
  DevBuf &b_q = e->b_wf_queues, &b_st = e->b_wf_state;   // kept between loads (the Go action loads a session every cycle): grown, never shrunk
  b_q.alloc(sizeof(WfQueue) * nq);
  b_st.alloc(sizeof(WfState));
  e->b_deserved.alloc(sizeof(double) * (size_t)hs.R * nq);
  e->b_desmask.alloc(sizeof(uint32_t) * nq);
  HIP_OK(hipMemcpyAsync(b_q.p, w.qs, sizeof(WfQueue) * nq, hipMemcpyHostToDevice, e->stream));
  HIP_OK(hipMemcpyAsync(b_st.p, w.st, sizeof(WfState), hipMemcpyHostToDevice, e->stream));
  kb_launch_waterfill(b_q.as<WfQueue>(), Q, b_st.as<WfState>(), hs.R, e->b_deserved.as<double>(), e->b_desmask.as<uint32_t>(), e->stream);
  HIP_OK(hipMemcpyAsync(w.qs, b_q.p, sizeof(WfQueue) * nq, hipMemcpyDeviceToHost, e->stream));
  HIP_OK(hipMemcpyAsync(w.st, b_st.p, sizeof(WfState), hipMemcpyDeviceToHost, e->stream));
  return w;
}
static void waterfill_collect(kb_engine *e, const WaterfillInFlight &w) {
  HostSession &hs = e->hs;
  HIP_OK(hipGetLastError());
  if (w.st->underflow) throw EngineError(KB_E_UNSUPPORTED, "proportion water-filling underflow (the reference would panic in Resource.Sub)");
  for (uint32_t q = 0; q < hs.Q; q++) hs.deserved[q] = w.qs[q].deserved;
  hs.queue_share_at_open = w.st->share_at_open ? 1 : 0;
  e->waterfill_passes = w.st->passes;
}

int kb_session_load(kb_engine *e, const kb_snapshot *sn) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!sn) throw EngineError(KB_E_INVALID, "snapshot is NULL");
    if (sn->version != KB_ABI_VERSION) throw EngineError(KB_E_INVALID, "snapshot ABI version mismatch");
    if (sn->n_res < 2 || sn->n_res > KB_MAX_RES) throw EngineError(KB_E_INVALID, "n_res out of range");
    quiesce(e);
    e->loaded = false;
    e->fin0.valid = false;
    e->stale_checked = false; e->pristine = true; e->load_clean = false;
    mg_reset(e->mg);   // (its device buffers are grow-only like every other one: no hipFree / hipMalloc per cycle)
    HostSession &hs = e->hs;
    const uint32_t NP = ((sn->n_nodes + KB_NODE_PAD - 1) / KB_NODE_PAD) * KB_NODE_PAD + (sn->n_nodes == 0 ? KB_NODE_PAD : 0);
    std::vector<uint32_t> t_active, nmask;
    hs.waterfill_on_device = e->device_waterfill && e->pol.has_proportion;
    // KB_LOAD_TRACE=1: where a load's time goes, phase by phase, on stderr (the Go action loads a session every cycle)
    static const bool load_trace = [] { const char *v = getenv("KB_LOAD_TRACE"); return v && v[0] == '1'; }();
    double t_mark = now_ms();
    auto mark = [&](const char *what) {
      if (!load_trace) return;
      const double t = now_ms();
      fprintf(stderr, "kb_session_load: %-28s %8.3f ms\n", what, t - t_mark);
      t_mark = t;
    };
    build_host_session(sn, e->pol, NP, hs, t_active, nmask);   // kb_session.cpp: validation, shapes, plugin OnSessionOpen state
    mark("build_host_session");
    const int R = hs.R;
    const uint32_t N = hs.N, T = hs.T, J = hs.J, Q = hs.Q;
    const kb_interpod *ip = sn->interpod;
    hipStream_t s = e->stream;
    e->evictions.clear();

    // ---- device upload: every source goes through the load's pinned area (PinnedArena above), every copy is asynchronous on the engine's
    //      stream, and the ONE synchronisation of a load is the one that ends it (run_finalize's, below)
    e->load_arena.reset();
    e->async_pending = true;   // from here on copies out of (and the water-fill's answers into) the pinned area are queued: a validation that throws below
                               // leaves them in flight, and the next load's quiesce() must wait for them before the area is handed out again
    Uploader up(e->load_arena, s);
    WaterfillInFlight wf_flight;
    // proportion's water-fill first: it needs the host session only, and its (tiny, serial) launch runs while the host assembles the rest
    if (hs.waterfill_on_device) wf_flight = device_waterfill_queue(e);
    KbDev &d = e->dev;
    d = KbDev{};
    d.R = R; d.N = N; d.NP = NP; d.T = T; d.J = J; d.Q = Q;
    up.padded(e->b_idle, sn->node_idle, R, N, NP);
    e->idle_below_eps = false;   // NodeInfo keeps Idle above -epsilon (every Sub is guarded by LessEqual); a snapshot may not
    for (uint32_t n = 0; n < N; n++)
      if (sn->node_idle[n] <= -kMinMilliCPU || sn->node_idle[(size_t)N + n] <= -kMinMemory) e->idle_below_eps = true;
    up.padded(e->b_rel, sn->node_releasing, R, N, NP);
    up.padded(e->b_nzc, sn->node_nz_cpu, 1, N, NP);
    up.padded(e->b_nzm, sn->node_nz_mem, 1, N, NP);
    up.padded(e->b_podcnt, sn->node_pod_cnt, 1, N, NP);
    up.padded(e->b_acpu, sn->node_alloc_cpu, 1, N, NP);
    up.padded(e->b_amem, sn->node_alloc_mem, 1, N, NP);
    up.padded(e->b_maxpods, sn->node_max_pods, 1, N, NP);
    {   // reciprocals of the allocatable quantities for the exact integer-division estimate (IEEE division, same on host and device)
      double *ia = up.stage<double>(e->b_invac, NP);
      for (uint32_t n = 0; n < N; n++) ia[n] = 1.0 / (double)sn->node_alloc_cpu[n];
      std::fill(ia + N, ia + NP, 0.0);
      up.commit();
      double *im = up.stage<double>(e->b_invam, NP);
      for (uint32_t n = 0; n < N; n++) im[n] = 1.0 / (double)sn->node_alloc_mem[n];
      std::fill(im + N, im + NP, 0.0);
      up.commit();
    }
    // The commit kernel keeps the window in LDS (160 KiB per workgroup on gfx950): one dirty slot per row (one thread of the
    // 256-thread workgroup evaluates one slot), the row descriptors, and per distinct shape its candidate list.  Prefer the
    // largest window that still admits 64 shapes.
    {
      const uint32_t budget = 160u * 1024u;
      const uint32_t W = std::min<uint32_t>(e->window, KB_K5_MAX_ROWS);
      uint32_t best_w = 0, best_s = 0;
      for (uint32_t w = W; w >= 1; w = (w > 32 ? ((w - 1) / 32) * 32 : w - 1)) {
        uint32_t sc = std::min<uint32_t>(KB_K5_MAX_SHAPES, w);
        while (sc > 0 && kb_commit_smem_bytes(w, sc, NP, R) > budget) sc--;
        if (sc >= std::min<uint32_t>(64, w)) { best_w = w; best_s = sc; break; }
        if (sc > best_s) { best_w = w; best_s = sc; }
        if (w == 1) break;
      }
      if (best_w == 0 || best_s == 0) throw EngineError(KB_E_UNSUPPORTED, "too many nodes / resource dimensions for the commit kernel's LDS tables");
      e->eff_window = best_w;
      e->shape_cap = best_s;
    }
    {   // 32-bit keys: (score + 1) << node_bits | inverted node index
      const long long max_score = 10ll * ((long long)e->pol.wL + e->pol.wM + e->pol.wB + e->pol.wNA);
      if (((unsigned long long)(max_score + 2) << kb_node_bits(NP)) > (1ull << 32))
        throw EngineError(KB_E_UNSUPPORTED, "score range x node count exceeds the commit kernel's 32-bit keys");
    }
    mark("node arrays, window");
    const uint32_t *ncls;   // the staged copy stays readable for the range checks below (the area is only reset by the next load)
    {
      uint32_t *p = up.stage<uint32_t>(e->b_ncls, NP);
      std::fill(p, p + NP, 0u);
      if (sn->node_class) std::memcpy(p, sn->node_class, sizeof(uint32_t) * N);
      up.commit();
      ncls = p;
    }
    up.copy(e->b_nmask, nmask.data(), NP);
    up.copy_persistent(e->b_tinit, hs.t_init.data(), (size_t)R * T);
    {   // the backfill view of t_init: cpu / memory of a BestEffort task are its Resreq (scalar rows are never compared for
        // them: every InitResreq scalar is at or below the epsilon, resource_info.go:283-287)
      bool differs = false;
      for (uint32_t t = 0; t < T && !differs; t++)
        differs = hs.t_init_empty[t] && (hs.t_res[t] != hs.t_init[t] || hs.t_res[(size_t)T + t] != hs.t_init[(size_t)T + t]);
      if (differs) {
        std::vector<double> fit(hs.t_init);
        for (uint32_t t = 0; t < T; t++)
          if (hs.t_init_empty[t]) { fit[t] = hs.t_res[t]; fit[(size_t)T + t] = hs.t_res[(size_t)T + t]; }
        up.copy(e->b_tfit, fit.data(), (size_t)R * T);
        e->t_fit = e->b_tfit.as<double>();
      } else {
        e->t_fit = e->b_tinit.as<double>();
      }
    }
    up.copy_persistent(e->b_tres, hs.t_res.data(), (size_t)R * T);
    up.copy_persistent(e->b_tnzc, sn->task_nz_cpu, T);
    up.copy_persistent(e->b_tnzm, sn->task_nz_mem, T);
    up.copy_persistent(e->b_tcls, hs.t_cls.data(), T);
    up.copy_persistent(e->b_tactive, t_active.data(), T);
    up.copy_persistent(e->b_tresmask, hs.t_resmask.data(), T);
    up.copy_persistent(e->b_tjob, hs.t_job.data(), T);
    up.copy_persistent(e->b_tstatus, hs.t_status.data(), T);
    up.copy_persistent(e->b_tnode, hs.t_node.data(), T);
    e->b_tbind.alloc(sizeof(uint32_t) * (T ? T : 1));   // nothing is bound yet: KB_NONE everywhere, set on the device
    HIP_OK(hipMemsetAsync(e->b_tbind.p, 0xFF, sizeof(uint32_t) * (T ? T : 1), s));
    static_assert(KB_NONE == 0xFFFFFFFFu, "t_bind is cleared with a byte pattern");
    {
      uint8_t *counted = up.stage<uint8_t>(e->b_tcounted, T);
      for (uint32_t t = 0; t < T; t++) {
        const int st = hs.t_status[t];
        counted[t] = (st == KB_TASK_BOUND || st == KB_TASK_BINDING || st == KB_TASK_RUNNING || st == KB_TASK_ALLOCATED) ? 1 : 0;   // drf.go:71-77
      }
      up.commit();
    }
    e->b_jallocated.alloc(J ? J : 1);
    HIP_OK(hipMemsetAsync(e->b_jallocated.p, 0, J ? J : 1, s));
    mark("task arrays");
    d.compat = nullptr;
    d.n_nc = sn->n_node_classes ? sn->n_node_classes : 1;
    if (sn->class_compat) {
      size_t nb = ((size_t)sn->n_task_classes * sn->n_node_classes + 7) / 8;
      for (uint32_t t = 0; t < T; t++)
        if (hs.t_cls[t] >= sn->n_task_classes) throw EngineError(KB_E_INVALID, "task class out of range");
      for (uint32_t n = 0; n < N; n++)
        if (ncls[n] >= sn->n_node_classes) throw EngineError(KB_E_INVALID, "node class out of range");
      up.copy(e->b_compat, sn->class_compat, nb);
      d.compat = e->b_compat.as<uint8_t>();
      d.crows = nullptr;
      if (sn->n_node_classes <= 256) {   // word-aligned rows for the commit kernel (one 32-byte fetch per task class)
        std::vector<uint32_t> rows((size_t)sn->n_task_classes * 8, 0u);
        for (uint32_t tc = 0; tc < sn->n_task_classes; tc++)
          for (uint32_t nc = 0; nc < sn->n_node_classes; nc++) {
            size_t bit = (size_t)tc * sn->n_node_classes + nc;
            if ((sn->class_compat[bit >> 3] >> (bit & 7)) & 1) rows[(size_t)tc * 8 + (nc >> 5)] |= 1u << (nc & 31);
          }
        up.copy(e->b_crows, rows.data(), rows.size());
        d.crows = e->b_crows.as<uint32_t>();
      }
    }
    d.ports = nullptr; d.t_want = nullptr; d.t_conf = nullptr;
    d.ports_x = nullptr; d.t_want_x = nullptr; d.t_conf_x = nullptr; d.port_xw = 0;
    if (sn->node_ports || sn->task_port_want || sn->task_port_conflict) {
      const size_t Wh = sn->port_words ? sn->port_words : 1;   // 64-bit words per mask; word 0 here, the others below
      std::vector<unsigned long long> np_(NP, 0ull), tw(T ? T : 1, 0ull), tc(T ? T : 1, 0ull);
      bool any = false;
      for (uint32_t n = 0; n < N && sn->node_ports; n++) { np_[n] = sn->node_ports[(size_t)n * Wh]; any = any || np_[n]; }
      for (uint32_t t = 0; t < T; t++) {
        if (sn->task_port_want) tw[t] = sn->task_port_want[(size_t)t * Wh];
        if (sn->task_port_conflict) tc[t] = sn->task_port_conflict[(size_t)t * Wh];
        if ((tw[t] & ~tc[t]) != 0) throw EngineError(KB_E_INVALID, "a pod's host ports must conflict with themselves (want is not a subset of conflict)");
        any = any || tw[t] || tc[t];
      }
      if (any) {
        up.copy(e->b_ports, np_.data(), NP);
        up.copy(e->b_twant, tw.data(), tw.size());
        up.copy(e->b_tconf, tc.data(), tc.size());
        d.ports = e->b_ports.as<unsigned long long>();
        d.t_want = e->b_twant.as<unsigned long long>();
        d.t_conf = e->b_tconf.as<unsigned long long>();
      }
      if (hs.port_xw) {   // some pod reaches beyond word 0 (kb_host.hpp: t_wide): the words behind it, nodes word-major ([port_xw][NP]: K1 reads runs of nodes)
        const uint32_t X = hs.port_xw;
        std::vector<unsigned long long> nx((size_t)X * NP, 0ull);
        for (uint32_t n = 0; n < N && sn->node_ports; n++)
          for (uint32_t w = 0; w < X; w++) nx[(size_t)w * NP + n] = sn->node_ports[(size_t)n * Wh + 1 + w];
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "host-port words");
        up.copy(e->b_ports_x, nx.data(), nx.size());
        up.copy_persistent(e->b_twant_x, reinterpret_cast<const unsigned long long *>(hs.t_want_x.data()), hs.t_want_x.size());
        up.copy_persistent(e->b_tconf_x, reinterpret_cast<const unsigned long long *>(hs.t_conf_x.data()), hs.t_conf_x.size());
        d.ports_x = e->b_ports_x.as<unsigned long long>();
        d.t_want_x = e->b_twant_x.as<unsigned long long>();
        d.t_conf_x = e->b_tconf_x.as<unsigned long long>();
        d.port_xw = X;
        if (!d.ports) {   // word 0 empty everywhere: the kernels still take the host-port path by d.ports
          up.copy(e->b_ports, np_.data(), NP);
          up.copy(e->b_twant, tw.data(), tw.size());
          up.copy(e->b_tconf, tc.data(), tc.size());
          d.ports = e->b_ports.as<unsigned long long>();
          d.t_want = e->b_twant.as<unsigned long long>();
          d.t_conf = e->b_tconf.as<unsigned long long>();
        }
      }
    }
    d.aff = nullptr;
    d.aff_cls = nullptr;
    hs.cls_has_aff.clear();
    d.wNA = e->pol.wNA;
    if (sn->class_affinity && sn->n_task_classes && sn->n_node_classes) {
      for (uint32_t t = 0; t < T; t++)
        if (hs.t_cls[t] >= sn->n_task_classes) throw EngineError(KB_E_INVALID, "task class out of range");
      for (uint32_t n = 0; n < N; n++)
        if (ncls[n] >= sn->n_node_classes) throw EngineError(KB_E_INVALID, "node class out of range");
      const size_t na = (size_t)sn->n_task_classes * sn->n_node_classes;
      std::vector<uint8_t> has(sn->n_task_classes, 0);
      bool any = false;
      for (uint32_t tc = 0; tc < sn->n_task_classes; tc++)
        for (uint32_t nc = 0; nc < sn->n_node_classes; nc++) {
          const int32_t c = sn->class_affinity[(size_t)tc * sn->n_node_classes + nc];
          if (c < 0 || c > 100000) throw EngineError(KB_E_UNSUPPORTED, "node-affinity count outside 0..100000");
          if (c) { has[tc] = 1; any = true; }
        }
      if (any && e->pol.wNA != 0) {
        if (e->pol.wNA < 0 || 10 * (e->pol.wL + e->pol.wM + e->pol.wB + e->pol.wNA) > 65535)
          throw EngineError(KB_E_UNSUPPORTED, "nodeorder weights exceed the 16-bit score range");
        hs.has_affinity = true;
        hs.cls_has_aff = has;
        up.copy(e->b_aff, sn->class_affinity, na);
        up.copy(e->b_affcls, has.data(), has.size());
        d.aff = e->b_aff.as<int32_t>();
        d.aff_cls = e->b_affcls.as<uint8_t>();
      }
    }
    d.ip_ctr_dom = nullptr; d.ip_ctr_count = nullptr; d.ip_ctr_total = nullptr; d.t_ip_inc = nullptr; d.t_ip_forbid = nullptr;
    d.t_ip_req = nullptr; d.t_ip_self = nullptr; d.t_ip_subject = nullptr; d.ip_cls_dom = nullptr; d.ip_cls_bound = nullptr;
    d.ip_cls_unbound = nullptr; d.t_ip_cls_inc = nullptr; d.t_ip_sig = nullptr; d.ip_sig_w = nullptr; d.ip_z = nullptr;
    d.ip_scratch_cnt = nullptr; d.ip_scratch_hist = nullptr; d.ip_C = d.ip_D = d.ip_P = 0; d.ip_Wc = d.ip_Wp = 1; d.wPA = e->pol.wPA; d.t_ip_checks = nullptr;
    if (ip) {
      const uint32_t C = ip->n_counters, P = ip->n_classes, D = ip->n_domains;
      const uint32_t Wc = C ? (C + 63) / 64 : 1, Wp = P ? (P + 63) / 64 : 1;
      // [rows][N] -> [max(rows, 1)][NP], the pad (and the row of a table without rows) KB_NONE / 0
      auto pad_u32 = [&](DevBuf &b, const uint32_t *src, uint32_t rows) {
        if (rows) { up.padded<uint32_t>(b, src, rows, N, NP, KB_NONE); return; }
        uint32_t *p0 = up.stage<uint32_t>(b, NP);
        std::fill(p0, p0 + NP, KB_NONE);
        up.commit();
      };
      auto pad_i32 = [&](DevBuf &b, const int32_t *src, uint32_t rows) {
        if (rows) { up.padded<int32_t>(b, src, rows, N, NP, 0); return; }
        int32_t *p0 = up.stage<int32_t>(b, NP);
        std::fill(p0, p0 + NP, 0);
        up.commit();
      };
      pad_u32(e->b_ip_cdom, ip->ctr_dom, C);
      pad_u32(e->b_ip_pdom, ip->cls_dom, P);
      pad_i32(e->b_ip_pbound, ip->cls_bound, P);
      pad_i32(e->b_ip_punb, ip->cls_unbound, P);
      std::vector<int32_t> cc((size_t)std::max(C, 1u) * D, 0), ct(std::max(C, 1u), 0);
      if (C) { std::memcpy(cc.data(), ip->ctr_count, sizeof(int32_t) * (size_t)C * D); std::memcpy(ct.data(), ip->ctr_total, sizeof(int32_t) * C); }
      up.copy(e->b_ip_ccnt, cc.data(), cc.size());
      up.copy(e->b_ip_ctot, ct.data(), ct.size());
      up.copy_persistent(e->b_ip_tinc, ip->task_inc, (size_t)T * Wc);
      up.copy_persistent(e->b_ip_tforbid, ip->task_forbid, (size_t)T * Wc);
      up.copy(e->b_ip_tchk, hs.t_ip_checks.data(), T);
      up.copy(e->b_ip_treq, ip->task_require, T);
      up.copy(e->b_ip_tself, ip->task_self, T);
      up.copy(e->b_ip_tsubj, hs.t_ip_subject.data(), T);
      up.copy_persistent(e->b_ip_tcinc, ip->task_cls_inc, (size_t)T * Wp);
      up.copy(e->b_ip_tsig, ip->task_sig, T);
      std::vector<int32_t> sw((size_t)std::max(ip->n_sigs, 1u) * std::max(P, 1u), 0);
      if (ip->n_sigs && P) std::memcpy(sw.data(), ip->sig_weight, sizeof(int32_t) * (size_t)ip->n_sigs * P);
      up.copy(e->b_ip_sigw, sw.data(), sw.size());
      const uint32_t z0 = ip->first_unbound_node;
      up.copy(e->b_ip_z, &z0, 1);
      d.ip_ctr_dom = e->b_ip_cdom.as<uint32_t>(); d.ip_ctr_count = e->b_ip_ccnt.as<int32_t>(); d.ip_ctr_total = e->b_ip_ctot.as<int32_t>();
      d.t_ip_inc = e->b_ip_tinc.as<unsigned long long>(); d.t_ip_forbid = e->b_ip_tforbid.as<unsigned long long>();
      d.t_ip_req = e->b_ip_treq.as<uint16_t>(); d.t_ip_self = e->b_ip_tself.as<uint8_t>(); d.t_ip_subject = e->b_ip_tsubj.as<uint8_t>();
      d.t_ip_checks = e->b_ip_tchk.as<uint8_t>();
      d.ip_cls_dom = e->b_ip_pdom.as<uint32_t>(); d.ip_cls_bound = e->b_ip_pbound.as<int32_t>(); d.ip_cls_unbound = e->b_ip_punb.as<int32_t>();
      d.t_ip_cls_inc = e->b_ip_tcinc.as<unsigned long long>(); d.t_ip_sig = e->b_ip_tsig.as<uint32_t>(); d.ip_sig_w = e->b_ip_sigw.as<int32_t>();
      d.ip_z = e->b_ip_z.as<uint32_t>();
      d.ip_C = C; d.ip_D = D; d.ip_P = P; d.ip_Wc = Wc; d.ip_Wp = Wp;
    }
    mark("classes, ports, inter-pod");
    e->h_probe_alive.resize(std::max<uint32_t>(hs.n_feas_shapes, 1u));
    e->h_probe_rows.resize(std::max<uint32_t>(hs.n_feas_shapes, 1u));
    HIP_OK(hipHostGetDevicePointer((void **)&e->d_probe_alive, e->h_probe_alive.data(), 0));
    HIP_OK(hipHostGetDevicePointer((void **)&e->d_probe_rows, e->h_probe_rows.data(), 0));
    up.copy(e->b_jbegin, hs.job_begin.data(), J + 1);
    up.copy(e->b_jmin, hs.job_min.data(), J);
    up.copy(e->b_jqueue, hs.job_queue.data(), J);
    up.copy(e->b_total, hs.total.v, KB_MAX_RES);
    e->total_mask = hs.total.mask;
    if (!hs.waterfill_on_device) {   // the host loop of kb_session.cpp filled hs.deserved (the launch writes b_deserved / b_desmask itself)
      double *des = up.stage<double>(e->b_deserved, (size_t)R * (Q ? Q : 1));
      std::fill(des, des + (size_t)R * (Q ? Q : 1), 0.0);
      for (uint32_t q = 0; q < Q; q++)
        for (int dd = 0; dd < R; dd++) des[(size_t)dd * Q + q] = hs.deserved[q].get(dd);
      up.commit();
      uint32_t *desmask = up.stage<uint32_t>(e->b_desmask, Q ? Q : 1);
      desmask[0] = 0;
      for (uint32_t q = 0; q < Q; q++) desmask[q] = hs.deserved[q].mask;
      up.commit();
    }
    hs.job_alloc.assign((size_t)J * R, 0.0);
    hs.job_share.assign(J, 0.0);
    hs.queue_alloc.assign((size_t)Q * R, 0.0);
    hs.queue_share.assign(Q, 0.0);
    hs.job_ready.assign(J, 0);
    e->b_jalloc.alloc(sizeof(double) * (size_t)(J ? J : 1) * R);
    e->b_jshare.alloc(sizeof(double) * (J ? J : 1));
    e->b_qalloc.alloc(sizeof(double) * (size_t)(Q ? Q : 1) * R);
    e->b_qshare.alloc(sizeof(double) * (Q ? Q : 1));
    e->b_jready.alloc(sizeof(int) * (J ? J : 1));

    d.idle = e->b_idle.as<double>(); d.rel = e->b_rel.as<double>();
    d.nzc = e->b_nzc.as<long long>(); d.nzm = e->b_nzm.as<long long>(); d.podcnt = e->b_podcnt.as<int>();
    d.acpu = e->b_acpu.as<long long>(); d.amem = e->b_amem.as<long long>();
    d.maxpods = e->b_maxpods.as<int>(); d.ncls = e->b_ncls.as<uint32_t>(); d.nmask = e->b_nmask.as<uint32_t>();
    d.inv_acpu = e->b_invac.as<double>(); d.inv_amem = e->b_invam.as<double>();
    d.t_init = e->b_tinit.as<double>(); d.t_res = e->b_tres.as<double>();
    d.t_nzc = e->b_tnzc.as<long long>(); d.t_nzm = e->b_tnzm.as<long long>();
    d.t_cls = e->b_tcls.as<uint32_t>(); d.t_active = e->b_tactive.as<uint32_t>(); d.t_resmask = e->b_tresmask.as<uint32_t>();
    d.t_job = e->b_tjob.as<uint32_t>(); d.t_status = e->b_tstatus.as<uint8_t>(); d.t_node = e->b_tnode.as<uint32_t>();
    d.t_bind = e->b_tbind.as<uint32_t>(); d.t_counted = e->b_tcounted.as<uint8_t>(); d.j_allocated = e->b_jallocated.as<uint8_t>();
    d.wL = e->pol.wL; d.wM = e->pol.wM; d.wB = e->pol.wB;
    d.pred_enabled = e->pol.pred_enabled ? 1 : 0;
    d.score_enabled = e->pol.nodeorder_enabled ? 1 : 0;
    d.whole = hs.whole ? 1u : 0u;
    e->win_cap = 0; e->mat_cap = 0; e->keys_cap = 0;
    e->mat2_cap = 0; e->stale_cap = 0;   // the second stream's matrix rows are [rows][NP] too: a session with more nodes needs them again
    e->xs_cap = 0;   // kb_eval_matrix's per-shape rows are [shapes][NP] as well (found by tests/test_gpu_reload.py: fewer shapes over more nodes overran them)
    e->stats = kb_stats{};
    e->dirty_share = 0.0;
    e->commit_kernel = e->commit_pin >= 0 ? e->commit_pin : (int)KB_COMMIT_SELECT;
    e->round_no = 0;
    mark("jobs, queues, deserved");
    auto snap_copy = [&](DevBuf &dst, const DevBuf &src) {
      dst.alloc(src.bytes);
      HIP_OK(hipMemcpyAsync(dst.p, src.p, src.bytes, hipMemcpyDeviceToDevice, s));
    };
    snap_copy(e->p_idle, e->b_idle); snap_copy(e->p_rel, e->b_rel); snap_copy(e->p_nzc, e->b_nzc); snap_copy(e->p_nzm, e->b_nzm);
    snap_copy(e->p_podcnt, e->b_podcnt); snap_copy(e->p_tstatus, e->b_tstatus); snap_copy(e->p_tnode, e->b_tnode);
    if (d.ports) snap_copy(e->p_ports, e->b_ports);
    if (d.ports_x) snap_copy(e->p_ports_x, e->b_ports_x);
    snap_copy(e->p_tcounted, e->b_tcounted);
    snap_copy(e->p_nmask, e->b_nmask);   // the evict actions rewrite the key masks of the nodes they touch (upload_live_nodes)
    if (hs.has_interpod) { snap_copy(e->p_ip_ccnt, e->b_ip_ccnt); snap_copy(e->p_ip_ctot, e->b_ip_ctot); snap_copy(e->p_ip_punb, e->b_ip_punb); snap_copy(e->p_ip_z, e->b_ip_z); }
    // initial drf / proportion / gang aggregates come from the device reduction (K2+K4)
    mark("pristine copies (queued)");
    // the load's synchronisation; the water-fill's answer (deserved, "did a pass run") is read behind it, in front of the host-side
    // post-processing of the shares, which wants to know whether updateShare ran at open
    run_finalize(e, [&]() {
      if (hs.waterfill_on_device) waterfill_collect(e, wf_flight);
      hs.queue_share_live.assign(Q ? Q : 1, hs.queue_share_at_open);
    });
    mark("aggregates (device reduction, the load's one synchronisation)");
    e->fin0.job_alloc = hs.job_alloc; e->fin0.job_share = hs.job_share; e->fin0.queue_alloc = hs.queue_alloc; e->fin0.queue_share = hs.queue_share;
    e->fin0.job_ready = hs.job_ready; e->fin0.t_status = hs.t_status; e->fin0.t_node = hs.t_node; e->fin0.valid = true;
    e->stats.reduce_ms = 0;
    e->loaded = true;
    e->tainted = false;
  });
}

int kb_session_reset(kb_engine *e) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    const double t_reset0 = now_ms();
    e->tainted = false;
    e->pristine = true;
    e->stale_checked = e->load_clean;
    hipStream_t s = e->stream;
    auto restore = [&](DevBuf &dst, const DevBuf &src) { HIP_OK(hipMemcpyAsync(dst.p, src.p, src.bytes, hipMemcpyDeviceToDevice, s)); };
    restore(e->b_idle, e->p_idle); restore(e->b_rel, e->p_rel); restore(e->b_nzc, e->p_nzc); restore(e->b_nzm, e->p_nzm);
    restore(e->b_podcnt, e->p_podcnt); restore(e->b_tstatus, e->p_tstatus); restore(e->b_tnode, e->p_tnode);
    if (e->dev.ports) restore(e->b_ports, e->p_ports);
    if (e->dev.ports_x) restore(e->b_ports_x, e->p_ports_x);
    restore(e->b_tcounted, e->p_tcounted);
    restore(e->b_nmask, e->p_nmask);
    if (e->hs.has_interpod) { restore(e->b_ip_ccnt, e->p_ip_ccnt); restore(e->b_ip_ctot, e->p_ip_ctot); restore(e->b_ip_punb, e->p_ip_punb); restore(e->b_ip_z, e->p_ip_z); }
    HIP_OK(hipMemsetAsync(e->b_tbind.p, 0xFF, sizeof(uint32_t) * (e->hs.T ? e->hs.T : 1), s));
    HIP_OK(hipMemsetAsync(e->b_jallocated.p, 0, e->b_jallocated.bytes, s));
    mg_reset(e->mg);   // (its device buffers are grow-only like every other one: no hipFree / hipMalloc per cycle)
    std::fill(e->hs.queue_share_live.begin(), e->hs.queue_share_live.end(), e->hs.queue_share_at_open);
    e->evictions.clear();
    e->hs.t_off_node.clear();
    // The restored state is bit for bit the one kb_session_load reduced (the pristine copies were taken in front of that reduction,
    // which flips no status: no job has an Allocate yet): its results come back from the host copies made then.  The device-side result
    // buffers keep the previous reduction's values; nothing reads them before the next reduction rewrites them.
    if (e->fin0.valid) {
      e->async_pending = true;   // stream-ordered with everything run_allocate / run_backfill launch; quiesce() for the rest
      HostSession &hs = e->hs;
      hs.job_alloc = e->fin0.job_alloc; hs.job_share = e->fin0.job_share; hs.queue_alloc = e->fin0.queue_alloc; hs.queue_share = e->fin0.queue_share;
      hs.job_ready = e->fin0.job_ready; hs.t_status = e->fin0.t_status; hs.t_node = e->fin0.t_node;
    } else {
      double keep = e->stats.reduce_ms;
      run_finalize(e);
      e->stats.reduce_ms = keep;
    }
    e->tl_reset += now_ms() - t_reset0;
  });
}

// What allocate / backfill refuse before they touch anything — whichever way the action is entered (kb_run_allocate / kb_run_backfill, or the first
// kb_round_begin of an action on the task-row split: every rank would diverge alike there, so neither the delta cross-check nor the journal digest
// would notice).
static void action_entry_guards(kb_engine *e) {
  if (!e->loaded) throw EngineError(KB_E_STATE, "kb_session_load must precede kb_run_* / kb_round_begin");
  if (e->tainted) throw EngineError(KB_E_STATE, "a preempt / reclaim call failed after touching the session's state: kb_session_load or kb_session_reset first");
  // A Pending task that still carries a NodeName was un-pipelined by a discarded preempt statement (NodeInfo.RemoveTask never
  // clears it, api/node_info.go:217-243): the reference's AddTask then refuses every other node AFTER ssn.Allocate has flipped
  // the status (session.go:243 vs :255).  Not modelled: the stock action takes such a cycle (it cannot arise under the stock
  // action order, where preempt runs last).
  if (!e->stale_checked) {
    for (uint32_t t = 0; t < e->hs.T; t++)
      if (e->hs.t_status[t] == KB_TASK_PENDING && e->hs.t_node[t] != KB_NONE)
        throw EngineError(KB_E_UNSUPPORTED, "a Pending task carries a stale NodeName (un-pipelined by a discarded preempt statement)");
    e->stale_checked = true;   // allocate and backfill never create one
    if (e->pristine) e->load_clean = true;
  }
}

static int run_action(kb_engine *e, uint32_t action, kb_decision *out, uint64_t cap, uint64_t *n_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    action_entry_guards(e);
    const double t_act0 = now_ms();
    ActionRun run;
    run.begin(e, action);
    const double t_act1 = now_ms();
    run.probe_dead_shapes(e);   // shapes no node can take from the start (larger than every node, full classes) never cost a break
    const double t_act2 = now_ms();
    uint32_t n = run.plan(e);
    const double t_act3 = now_ms();
    if (action == 0) { e->tl_begin_parts[0] += t_act1 - t_act0; e->tl_begin_parts[1] += t_act2 - t_act1; e->tl_begin_parts[2] += t_act3 - t_act2; }
    ensure_matrix_buffers(e, e->eff_window, e->eff_window + 1);   // sized once: no reallocation under a round in flight
    // chained rounds of plain sessions (no score that is normalised over the feasible set, no inter-pod counters) build their candidate
    // lists beside the predecessor's commit kernel
    const bool overlap_ok = e->overlap && action == 0 && e->fast_rounds && !e->hs.has_affinity && !e->hs.has_interpod && 2 * e->eff_window + 1 <= 1024u &&
                            kb_repair_smem_bytes(e->dev.NP) <= 150u * 1024u;   // the repair launch's LDS (its node bitmap grows with the cluster)
    // The second stream is ordered behind nothing the first one holds: the copies kb_session_reset left queued there must have landed before
    // an overlapped launch reads the node state (the feasibility probe's read-back waits for them when it runs — it does not without the
    // predicates plugin, with KB_PROBE=0, or when every shape is dead; found on the emulated device with asynchronous streams)
    if (overlap_ok) quiesce(e);
    auto launch = [&](uint32_t rows_n, const uint32_t *rows, uint32_t buf, uint32_t chain_expect, uint32_t n_prev) {
      RoundCtx c = round_prepare(e, rows_n, action == 0 ? 1 : 2, action == 1, true, rows, buf, chain_expect);   // single GPU: every matrix row is local
      unsigned long long *keys = e->b_keys.as<unsigned long long>();
      if (overlap_ok && c.direct && chain_expect != 0 && !e->overlap_faults) {
        round_candidates_overlapped(e, c, n_prev, keys);
        c.overlapped = true;
      }
      else round_candidates(e, c, 0, c.ns, keys);
      round_commit(e, c, keys, nullptr, 0, 0);
      return c;
    };
    if (overlap_ok) ensure_overlap_buffers(e, e->eff_window, 2 * e->eff_window + 1);
    // Fast rounds return from the launch immediately.  The host uses the wait to speculate the NEXT window (assuming the one in
    // flight completes, which ~80 % do) and queues that round behind the running one right away: the device starts it the
    // moment the commit kernel ends instead of idling through a host round trip (~19 us per round).  A round that stops early
    // clears the chain word and the queued round skips itself (KbRound::chain).
    // (sessions with host-port masks of several words: one round at a time — a pod that reaches beyond word 0 changes node state from the host
    //  after its round, which a round already queued or overlapped would not see)
    const bool ahead = action == 0 && e->fast_rounds && !e->dev.port_xw;
    const bool chained = ahead && e->chain_rounds;
    uint32_t buf = 0;
    RoundCtx c{};
    if (n) c = launch(n, nullptr, buf, 0, 0);
    (action == 0 ? e->tl_begin : e->tl_backfill) += now_ms() - t_act0;
    while (n) {
      uint32_t n_done = 0, reason = 0;
      const uint32_t n_next = ahead ? run.plan_ahead(e) : 0;
      RoundCtx cn{};
      const bool queued = chained && n_next > 0;
      if (queued) cn = launch(n_next, run.rows_next.data(), buf ^ 1u, c.r.chain_tag, n);
      const double t_w0 = now_ms();
      round_collect(e, c, true, n_done, reason);
      const double t_b0 = now_ms();
      e->tl_wait += t_b0 - t_w0;
      // a break: the feasibility probe goes out before the host starts on the answer (ActionRun::probe_launch)
      const bool probe_early = ahead && reason != KB_REASON_DONE && reason != KB_REASON_RENORM;
      if (probe_early) run.probe_launch(e);
      try { run.absorb(e, n, n_done, reason); } catch (...) { run.probe_abandon(e); throw; }
      const double t_b1 = now_ms();
      if (ahead && reason == KB_REASON_DONE) {
        run.promote(e, n_next);
        n = n_next;
        if (queued) { c = cn; buf ^= 1u; }
        else if (n) c = launch(n, nullptr, buf, 0, 0);
      } else {
        if (probe_early) run.probe_collect(e);
        else if (reason != KB_REASON_RENORM) run.probe_dead_shapes(e);
        const double t_b2 = now_ms();
        n = run.plan(e);   // re-plan first: the queued round drains (three empty launches) while the host works
        if (action == 0) { e->tl_break_parts[0] += t_b1 - t_b0; e->tl_break_parts[1] += t_b2 - t_b1; e->tl_break_parts[2] += now_ms() - t_b2; }
        if (queued) {   // the queued round skipped itself: consume its publication before its staging half is reused
          uint32_t nd2 = 0, rs2 = 0;
          round_collect(e, cn, true, nd2, rs2);
          if (rs2 != KB_REASON_SKIPPED) throw EngineError(KB_E_INTERNAL, "a round queued behind a stopped round ran");
          e->stats.matrix_launches -= 1;
          e->stats.matrix_evals -= (uint64_t)cn.ns * e->hs.N;
          // its matrix / arg-max launches on the second stream (they run whatever the chain word says) read the descriptor and window
          // halves the re-planned rounds are about to rewrite from the first stream: nothing else orders the two
          if (cn.overlapped && e->stream_b) HIP_OK(hipStreamSynchronize(e->stream_b));
        }
        if (n) c = launch(n, nullptr, buf, 0, 0);
        if (action == 0) e->tl_break += now_ms() - t_b0;
      }
    }
    if (e->stream_b) HIP_OK(hipStreamSynchronize(e->stream_b));   // a candidate launch of a round that was skipped may still be running
    run.finish(e);
    if (n_out) *n_out = run.decs.size();
    if (run.decs.size() > cap) throw EngineError(KB_E_CAPACITY, "decision buffer too small");
    if (out && !run.decs.empty()) std::memcpy(out, run.decs.data(), sizeof(kb_decision) * run.decs.size());
  });
}

int kb_run_allocate(kb_engine *e, kb_decision *out, uint64_t cap, uint64_t *n_out) { return run_action(e, 0, out, cap, n_out); }
int kb_run_backfill(kb_engine *e, kb_decision *out, uint64_t cap, uint64_t *n_out) { return run_action(e, 1, out, cap, n_out); }

// ---- preempt (kb_preempt.hpp): statements and victims on the host, PredicateNodes + PrioritizeNodes + SortNodes on the device ----
namespace {
// node state of the given nodes: host mirror -> device (after Pipelines / evictions changed it)
// One packed record per node in a persistent pinned staging vector, one copy, one scatter kernel (kb_launch_scatter_nodes) — it used
// to be 2R + 5 tiny asynchronous copies per node out of loop-scoped stack locals (round-2 advisory: correct only because pageable
// sources are staged at the call, and ten driver calls per dirty node on the preempt refresh path).
void upload_live_nodes(kb_engine *e, const LiveNodes &ln, const std::vector<uint32_t> &nodes) {
  if (nodes.empty()) return;
  const int R = e->hs.R;
  const size_t rec = 5 + 2 * (size_t)R, words = rec * nodes.size();
  e->h_scatter.resize(words);
  unsigned long long *w = e->h_scatter.data();
  for (uint32_t n : nodes) {
    const uint32_t nm = (ln.idle[n].mask & 0x3FFFFFFFu) | (ln.rel[n].mask ? 0x80000000u : 0u);
    w[0] = (unsigned long long)n | ((unsigned long long)nm << 32);
    w[1] = (unsigned long long)(uint32_t)ln.podcnt[n];
    w[2] = (unsigned long long)ln.nzc[n];
    w[3] = (unsigned long long)ln.nzm[n];
    w[4] = e->dev.ports ? ln.ports[n] : 0ull;
    for (int d = 0; d < R; d++) {
      const double vi = ln.idle[n].get(d), vr = ln.rel[n].get(d);
      std::memcpy(&w[5 + d], &vi, 8);
      std::memcpy(&w[5 + R + d], &vr, 8);
    }
    w += rec;
  }
  if (e->b_scatter.bytes < sizeof(unsigned long long) * words) e->b_scatter.alloc(sizeof(unsigned long long) * words);   // grown, never shrunk: hipFree synchronises the device
  HIP_OK(hipMemcpyAsync(e->b_scatter.p, e->h_scatter.data(), sizeof(unsigned long long) * words, hipMemcpyHostToDevice, e->stream));
  kb_launch_scatter_nodes(e->dev, e->b_scatter.as<unsigned long long>(), (uint32_t)nodes.size(), e->b_nmask.as<uint32_t>(), e->stream);
  // host-port masks of several words: the words behind the first, one 8-byte copy each (a rare session; [port_xw][NP] on the device, [N][port_xw] here)
  for (uint32_t X = e->dev.port_xw, i = 0; X && i < nodes.size(); i++)
    for (uint32_t w = 0; w < X; w++)
      HIP_OK(hipMemcpyAsync(e->b_ports_x.as<unsigned long long>() + (size_t)w * e->dev.NP + nodes[i], &ln.ports_x[(size_t)nodes[i] * X + w], sizeof(unsigned long long),
                            hipMemcpyHostToDevice, e->stream));
  HIP_OK(hipStreamSynchronize(e->stream));   // the staging vector is reused by the next refresh
}
}  // namespace

static int run_evict_action(kb_engine *e, bool reclaim, kb_stmt_op *out, uint64_t cap, uint64_t *n_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "kb_session_load must precede kb_run_preempt / kb_run_reclaim");
    if (e->tainted) throw EngineError(KB_E_STATE, "a preempt / reclaim call failed after touching the session's state: kb_session_load or kb_session_reset first");
    quiesce(e);
    e->pristine = false;
    e->stale_checked = false;
    HostSession &hs = e->hs;
    // an eviction takes a pod OUT of the inter-pod predicate's pod list (Running -> Releasing leaves api.AllocatedStatus): modelled on the
    // host side of the evict machine (kb_preempt.cpp: ip_*), the lists rebuilt on the device after every change (round 3; on by default
    // since its first device run, round 4: profiles/round4/first_call)
    // preempt with preferred node-affinity terms: the lists of such preemptors carry the NormalizeReduce'd score and are rebuilt after
    // every Pipeline instead of repaired (kb_preempt.cpp: preempt_walk; tests/test_gpu_regressions.py: test_preempt_with_preferred_node_affinity).
    const double t_begin = now_ms();
    const int R = hs.R;
    const uint32_t N = hs.N, NP = e->dev.NP, T = hs.T, J = hs.J, Q = hs.Q;
    // ---- live state: device -> host
    LiveNodes ln;
    std::vector<uint8_t> counted_in;
    const double t_e0 = now_ms();
    double t_e_copy = t_e0, t_e_nodes = t_e0, t_e_tasks = t_e0;   // entry, in parts (KB_EVICT_TRACE)
    {
      // one pinned block, every copy asynchronous on the engine's stream, ONE synchronisation (round 3: seven blocking pageable copies)
      const size_t o_idle = 0, o_rel = o_idle + sizeof(double) * (size_t)R * NP, o_nzc = o_rel + sizeof(double) * (size_t)R * NP, o_nzm = o_nzc + sizeof(long long) * NP,
                   o_ports = o_nzm + sizeof(long long) * NP, o_nmask = o_ports + sizeof(unsigned long long) * NP, o_pod = o_nmask + sizeof(uint32_t) * NP,
                   o_cnt = o_pod + sizeof(int) * NP, o_end = o_cnt + (((size_t)T + 7) & ~(size_t)7);
      e->h_evict.resize(o_end + 16);
      unsigned char *hb = e->h_evict.data();
      HIP_OK(hipMemcpyAsync(hb + o_idle, e->b_idle.p, sizeof(double) * (size_t)R * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_rel, e->b_rel.p, sizeof(double) * (size_t)R * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_nzc, e->b_nzc.p, sizeof(long long) * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_nzm, e->b_nzm.p, sizeof(long long) * NP, hipMemcpyDeviceToHost, e->stream));
      if (e->dev.ports) HIP_OK(hipMemcpyAsync(hb + o_ports, e->b_ports.p, sizeof(unsigned long long) * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_nmask, e->b_nmask.p, sizeof(uint32_t) * NP, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipMemcpyAsync(hb + o_pod, e->b_podcnt.p, sizeof(int) * NP, hipMemcpyDeviceToHost, e->stream));
      if (T) HIP_OK(hipMemcpyAsync(hb + o_cnt, e->b_tcounted.p, T, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipStreamSynchronize(e->stream));
      t_e_copy = now_ms();
      const double *idle = reinterpret_cast<const double *>(hb + o_idle), *rel = reinterpret_cast<const double *>(hb + o_rel);
      const uint32_t *nmask = reinterpret_cast<const uint32_t *>(hb + o_nmask);
      ln.nzc.assign(reinterpret_cast<const long long *>(hb + o_nzc), reinterpret_cast<const long long *>(hb + o_nzc) + NP);
      ln.nzm.assign(reinterpret_cast<const long long *>(hb + o_nzm), reinterpret_cast<const long long *>(hb + o_nzm) + NP);
      ln.podcnt.assign(reinterpret_cast<const int *>(hb + o_pod), reinterpret_cast<const int *>(hb + o_pod) + NP);
      if (e->dev.ports) ln.ports.assign(reinterpret_cast<const unsigned long long *>(hb + o_ports), reinterpret_cast<const unsigned long long *>(hb + o_ports) + NP);
      else ln.ports.assign(NP, 0);
      ln.ports_x.assign((size_t)N * hs.port_xw, 0);
      if (e->dev.port_xw) {   // the masks' words behind the first: [port_xw][NP] on the device, [N][port_xw] in the machine
        std::vector<unsigned long long> px((size_t)e->dev.port_xw * NP);
        HIP_OK(hipMemcpyAsync(px.data(), e->b_ports_x.p, sizeof(unsigned long long) * px.size(), hipMemcpyDeviceToHost, e->stream));
        HIP_OK(hipStreamSynchronize(e->stream));
        for (uint32_t n = 0; n < N; n++)
          for (uint32_t w = 0; w < hs.port_xw; w++) ln.ports_x[(size_t)n * hs.port_xw + w] = px[(size_t)w * NP + n];
      }
      counted_in.assign(hb + o_cnt, hb + o_cnt + T);
      ln.idle.assign(N, Res()); ln.rel.assign(N, Res());
      for (uint32_t n = 0; n < N; n++) {
        ln.idle[n].mask = nmask[n] & 0x3FFFFFFFu;
        for (int d = 0; d < R; d++) {
          ln.idle[n].v[d] = idle[(size_t)d * NP + n];
          ln.rel[n].v[d] = rel[(size_t)d * NP + n];
          // Releasing gains scalar keys only through Add: a dense non-zero value <=> the key is present
          if (d >= 2 && ln.rel[n].v[d] != 0.0) ln.rel[n].setk(d);
          // Idle: the device's mask holds the keys Allocatable had (plus what an earlier evict action uploaded).  Resource.Sub also
          // CREATES the keys of its operand in a non-nil map (resource_info.go:143-160: r.ScalarResources[name] -= quant), which is how
          // allocate / backfill leave a negative value under a key the node never advertised (sub-epsilon requests pass LessEqual
          // and add up).  Such a key reads non-zero, and a created key that reads 0 is indistinguishable from an absent one.
          if (d >= 2 && ln.idle[n].mask != 0 && ln.idle[n].v[d] != 0.0) ln.idle[n].setk(d);
        }
        // a non-nil Releasing map whose keys all read 0 (bit 31): which keys it holds does not matter, that Sub does not return early does
        if ((nmask[n] >> 31) && ln.rel[n].mask == 0 && R > 2) ln.rel[n].setk(2);
      }
      ln.ac.assign(hs.n_ac.begin(), hs.n_ac.end()); ln.am.assign(hs.n_am.begin(), hs.n_am.end());
      ln.maxpods = hs.n_maxpods; ln.cls = hs.n_cls;
    }
    t_e_nodes = now_ms();
    PreemptMachine pm;
    pm.counted.assign(counted_in.begin(), counted_in.end());
    if (pm.counted.empty()) pm.counted.resize(1);
    pm.jalloc = hs.job_alloc; pm.jshare = hs.job_share; pm.qalloc = hs.queue_alloc; pm.qshare = hs.queue_share;
    pm.jmask.assign(J ? J : 1, 0); pm.qmask.assign(Q ? Q : 1, 0);
    for (uint32_t t = 0; t < T; t++)
      if (pm.counted[t] && hs.t_job[t] < J) {
        pm.jmask[hs.t_job[t]] |= hs.t_resmask[t];
        if (hs.job_queue[hs.t_job[t]] < Q) pm.qmask[hs.job_queue[hs.t_job[t]]] |= hs.t_resmask[t];
      }
    // ---- the device side: one complete sorted list per preemptor shape, on demand
    double tl_lists = 0.0, tl_lists_host = 0.0, tl_refresh = 0.0;
    uint64_t n_lists = 0, n_refresh = 0;
    auto lists = [&](uint32_t task, std::vector<uint64_t> &keys) {
      const double tl0 = now_ms();
      ensure_window_buffers(e, 1);
      ensure_matrix_buffers(e, 1, N + 1);
      HIP_OK(hipMemcpyAsync(e->b_mrows.p, &task, sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
      KbRound r = make_round(e, 0, 1, N + 1, 0 /* plugin predicates only */, false);
      r.mrows = e->b_mrows.as<uint32_t>();
      kb_launch_matrix(e->dev, r, e->stream);
      kb_launch_affinity(e->dev, r, e->stream);   // NodeAffinity priority over the row's feasible set (no-op without such terms)
      kb_launch_interpod(e->dev, r, e->stream);   // InterPodAffinityPriority over the same set, against the counts uploaded last (no-op without such terms)
      kb_launch_argmax(e->dev, r, e->stream);
      e->h_listkeys.resize((size_t)N + 1);   // pinned, persistent: the copy is a DMA into place instead of a staged pageable copy into a fresh vector
      const size_t raw_n = (size_t)N + 1;
      const unsigned long long *raw = e->h_listkeys.data();
      HIP_OK(hipMemcpyAsync(e->h_listkeys.data(), e->b_keys.p, sizeof(unsigned long long) * raw_n, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipStreamSynchronize(e->stream));
      HIP_OK(hipGetLastError());
      const double tl1 = now_ms();
      tl_lists += tl1 - tl0; n_lists++;
      e->stats.matrix_launches += 1;
      e->stats.matrix_evals += N;
      // K3 orders (score descending, node ASCENDING); SortNodes breaks score ties by DESCENDING host name: reverse every run
      keys.clear();
      size_t i = 0;
      keys.reserve(raw_n);
      while (i < raw_n && raw[i] != 0ull) {
        size_t k = i;
        const uint32_t sc = KB_KEY_SCORE(raw[i]);
        while (k < raw_n && raw[k] != 0ull && KB_KEY_SCORE(raw[k]) == sc) k++;
        for (size_t q = k; q-- > i;) keys.push_back(((uint64_t)sc << 32) | KB_KEY_NODE(raw[q]));
        i = k;
      }
      tl_lists_host += now_ms() - tl1;
    };
    auto refresh = [&](const std::vector<uint32_t> &nodes) { const double t0 = now_ms(); upload_live_nodes(e, ln, nodes); tl_refresh += now_ms() - t0; n_refresh++; };
    std::vector<uint8_t> status = hs.t_status;
    std::vector<uint32_t> tnode = hs.t_node;
    t_e_tasks = now_ms();
    pm.init(&hs, &e->pol, &ln, &status, &tnode, lists, refresh);
    // inter-pod terms: the live counts (allocate / backfill of this session may have advanced them) come to the host; the machine keeps them
    // current and puts them back on the device in front of every list it asks for, and once more when the action is over
    IpLive ipl;
    auto ip_upload = [&]() {
      if (!ipl.ccnt.empty()) HIP_OK(hipMemcpyAsync(e->b_ip_ccnt.p, ipl.ccnt.data(), sizeof(int32_t) * ipl.ccnt.size(), hipMemcpyHostToDevice, e->stream));
      if (!ipl.ctot.empty()) HIP_OK(hipMemcpyAsync(e->b_ip_ctot.p, ipl.ctot.data(), sizeof(int32_t) * ipl.ctot.size(), hipMemcpyHostToDevice, e->stream));
      if (!ipl.punb.empty()) HIP_OK(hipMemcpyAsync(e->b_ip_punb.p, ipl.punb.data(), sizeof(int32_t) * ipl.punb.size(), hipMemcpyHostToDevice, e->stream));
      HIP_OK(hipMemcpyAsync(e->b_ip_z.p, &ipl.z, sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
      HIP_OK(hipStreamSynchronize(e->stream));   // the sources are this frame's vectors, and they change again before the next call
    };
    if (hs.has_interpod) {
      ipl.NP = NP;
      ipl.ccnt.resize((size_t)std::max(hs.ip_C, 1u) * hs.ip_D); ipl.ctot.resize(std::max(hs.ip_C, 1u)); ipl.punb.resize((size_t)std::max(hs.ip_P, 1u) * NP);
      HIP_OK(hipMemcpy(ipl.ccnt.data(), e->b_ip_ccnt.p, sizeof(int32_t) * ipl.ccnt.size(), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(ipl.ctot.data(), e->b_ip_ctot.p, sizeof(int32_t) * ipl.ctot.size(), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(ipl.punb.data(), e->b_ip_punb.p, sizeof(int32_t) * ipl.punb.size(), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(&ipl.z, e->b_ip_z.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
      pm.set_interpod(&ipl, ip_upload);
    }
    // from here on a failure leaves state behind (a mid-action refresh may have updated nodes on the device; after the journal is out,
    // host and device state are committed): whatever throws below, the session is marked tainted and every kb_run_* answers KB_E_STATE
    // until kb_session_load / kb_session_reset (round-2 advisory: the cross-check at the end used to fail AFTER publishing results)
    struct Taint { kb_engine *e; bool armed = true; ~Taint() { if (armed) e->tainted = true; } } taint{e};
    const double t_e1 = now_ms();
    if (reclaim) pm.run_reclaim(); else pm.run();
    const double t_e2 = now_ms();
    // ---- results: journal out, state back to the device
    if (n_out) *n_out = pm.ops.size();
    if (pm.ops.size() > cap) throw EngineError(KB_E_CAPACITY, "journal buffer too small");   // no result was written; a refresh may have updated nodes on the device: load the session again before another action
    for (size_t i = 0; i < pm.ops.size(); i++) { out[i].op = pm.ops[i].op; out[i].task = pm.ops[i].task; out[i].node = pm.ops[i].node; out[i].stmt = pm.ops[i].stmt; }
    upload_live_nodes(e, ln, pm.touched_nodes);
    if (hs.has_interpod) ip_upload();   // what the next action's kernels read
    hs.t_status = status;
    hs.t_node = tnode;
    pm.off_node_tasks(hs.t_off_node);
    if (T) {   // the task table back: through the pinned block, asynchronous, ordered in front of the finalize launches on the same stream
      const size_t t8 = ((size_t)T + 7) & ~(size_t)7;
      e->h_evict.resize(2 * t8 + sizeof(uint32_t) * (size_t)T + 16);
      unsigned char *hb = e->h_evict.data();
      std::memcpy(hb, status.data(), T);
      std::memcpy(hb + t8, pm.counted.data(), T);
      std::memcpy(hb + 2 * t8, tnode.data(), sizeof(uint32_t) * (size_t)T);
      HIP_OK(hipMemcpyAsync(e->b_tstatus.p, hb, T, hipMemcpyHostToDevice, e->stream));
      HIP_OK(hipMemcpyAsync(e->b_tcounted.p, hb + t8, T, hipMemcpyHostToDevice, e->stream));
      HIP_OK(hipMemcpyAsync(e->b_tnode.p, hb + 2 * t8, sizeof(uint32_t) * (size_t)T, hipMemcpyHostToDevice, e->stream));
    }
    e->evictions.insert(e->evictions.end(), pm.evictions.begin(), pm.evictions.end());
    for (const StmtOp &op : pm.ops)   // Evict / Pipeline fire proportion's handlers -> updateShare for the task's queue
      if (op.task != KB_NONE && hs.job_queue[hs.t_job[op.task]] < Q) hs.queue_share_live[hs.job_queue[hs.t_job[op.task]]] = 1;
    const double t_x_fin = now_ms();
    run_finalize(e);
    // the host's running drf / proportion aggregates must equal the device reduction over the task table
    if (e->pol.has_drf)
      for (uint32_t j = 0; j < J; j++)
        if (pm.jshare[j] != hs.job_share[j]) throw EngineError(KB_E_INTERNAL, "evict action: drf share diverged from the device reduction at job " + std::to_string(j));
    e->stats.tasks_popped += pm.popped;
    e->stats.evals += pm.evals;
    e->stats.total_ms += now_ms() - t_begin;
    static const bool ev_trace = [] { const char *v = getenv("KB_EVICT_TRACE"); return v && v[0] == '1'; }();
    if (ev_trace)   // host timeline of the action (profiles/round4)
      fprintf(stderr, "[kb evict] %s: entry %.2f ms = copies %.2f + node mirror %.2f + task tables %.2f + machine tables %.2f; machine set-up %.2f ms (job / task queues) + run, of it %.2f ms collecting candidates (%llu queue nodes looked at); exit: journal + state back %.2f, finalize + checks %.2f\n",
              reclaim ? "reclaim" : "preempt", t_e1 - t_e0, t_e_copy - t_e0, t_e_nodes - t_e_copy, t_e_tasks - t_e_nodes, t_e1 - t_e_tasks, pm.tr_setup_ms, pm.tr_scan_ms,
              (unsigned long long)pm.tr_scan_nodes, t_x_fin - t_e2, now_ms() - t_x_fin);
    if (ev_trace)
      fprintf(stderr, "[kb evict] %s: entry (state to the host, machine set-up) %.2f ms; machine %.2f ms of which %llu lists %.2f ms on the device + %.2f ms host reorder, %llu node refreshes %.2f ms; exit (journal, state back, finalize, checks) %.2f ms; popped %llu (walked %llu: %.2f ms, %llu nodes tried; skipped with their job %llu, turned away one by one %llu, own-job preemptors the priority rule excludes %llu), journal %zu\n",
              reclaim ? "reclaim" : "preempt", t_e1 - t_e0, t_e2 - t_e1, (unsigned long long)n_lists, tl_lists, tl_lists_host, (unsigned long long)n_refresh, tl_refresh, now_ms() - t_e2,
              (unsigned long long)pm.popped, (unsigned long long)pm.tr_walks, pm.tr_walk_ms, (unsigned long long)pm.tr_tries, (unsigned long long)pm.tr_skipped, (unsigned long long)pm.tr_shortcut, (unsigned long long)pm.tr_pruned, pm.ops.size());
    taint.armed = false;
  });
}

int kb_run_preempt(kb_engine *e, kb_stmt_op *out, uint64_t cap, uint64_t *n_out) { return run_evict_action(e, false, out, cap, n_out); }
int kb_run_reclaim(kb_engine *e, kb_stmt_op *out, uint64_t cap, uint64_t *n_out) { return run_evict_action(e, true, out, cap, n_out); }

int kb_get_evictions(kb_engine *e, uint32_t *out, uint64_t cap, uint64_t *n_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    if (n_out) *n_out = e->evictions.size();
    if (e->evictions.size() > cap) throw EngineError(KB_E_CAPACITY, "eviction buffer too small");
    if (out && !e->evictions.empty()) std::memcpy(out, e->evictions.data(), sizeof(uint32_t) * e->evictions.size());
  });
}

// The materialised matrix for task rows [t0, t0+n): evaluate each distinct shape of the range once (K1), then stream every
// row out of its shape's row (K1b); optionally the sorted candidate lists of the expanded rows (K3, length k).
struct ChunkPlan {
  KbRound r{};        // describes the expanded rows (score / maskw / keys of n rows)
  KbRound rs{};       // the per-shape launch
  uint32_t ns = 0;
  uint32_t n_xchunks = 0;   // chunks of the tiled expansion (kb_device.h: KbXChunk)
  bool direct = false;   // evaluate every task row itself (no per-shape rows, no expansion)
};
static ChunkPlan matrix_plan(kb_engine *e, uint32_t t0, uint32_t n, uint32_t fit_flags, uint32_t k) {
  ChunkPlan p;
  const uint32_t fit_mode = fit_flags & 0xFFu;
  if (fit_mode > 2 || (fit_flags & ~(0xFFu | KB_MATRIX_DIRECT | KB_MATRIX_NO_DEDUP))) throw EngineError(KB_E_INVALID, "fit_mode: 0, 1 or 2, optionally with KB_MATRIX_DIRECT / KB_MATRIX_NO_DEDUP");
  const size_t NP = e->dev.NP;
  ensure_window_buffers(e, n);          // h_rows / h_slot staging (host side only matters here)
  ensure_matrix_buffers(e, n, k ? k : 1);
  for (uint32_t i = 0; i < n; i++) e->h_rows[i] = t0 + i;
  if (e->h_mrows.size() < n) e->h_mrows.resize(n);
  p.ns = assign_shapes(e, n);
  // Expansion streams every task row out of its shape's row.  The rows are expanded in SHAPE order (k_expand's `order`), so a shape
  // row is read from HBM once and copied to all its task rows out of the L2 however many shapes there are; what the per-shape pass
  // cannot avoid is evaluating and storing the shape rows themselves.  When (nearly) every job has its own request that is a second
  // matrix: then every row is evaluated by the matrix kernel itself, adjacent equal rows (the tasks of a job) sharing one evaluation
  // (k_matrix_runs).  Measured on one box (profiles/round3/call5): 9 386 shapes of 100k rows: direct 0.62 ms, expansion 0.89 ms;
  // 2 989 shapes (BASELINE configs[3]): direct 0.68, expansion 0.66; 1M x 50k with ~500 shapes: direct 28.0, expansion 24.0.
  // KB_MATRIX_DIRECT in fit_flags pins the direct path (bench.py's evaluator-only variants).
  p.direct = ((size_t)p.ns * e->dev.NP * 2 > (32u << 20) && (size_t)p.ns * 16 > n) || (fit_flags & KB_MATRIX_DIRECT);
  if (p.direct) {
    // the tasks of a job are adjacent and share a shape: a row equal to its predecessor re-stores the predecessor's result
    const bool dedup = !(fit_flags & KB_MATRIX_NO_DEDUP);
    for (uint32_t i = 0; i < n; i++) e->h_same[i] = (dedup && i > 0 && e->h_slot[i] == e->h_slot[i - 1]) ? 1 : 0;
    HIP_OK(hipMemcpyAsync(e->b_same.p, e->h_same.data(), n, hipMemcpyHostToDevice, e->stream));
    p.r = make_round(e, 0, n, k ? k : 1, (int)fit_mode, false);
    p.r.mrows = nullptr;
    p.r.mrow_task0 = t0;
    p.r.same_prev = e->b_same.as<uint8_t>();
    return p;
  }
  if (p.ns > e->xs_cap) {
    e->b_sscore.alloc(sizeof(uint16_t) * (size_t)p.ns * NP);
    e->b_smask.alloc(sizeof(uint32_t) * (size_t)p.ns * (NP / 32));
    e->xs_cap = p.ns;
  }
  if (n > e->xslot_cap) { e->b_xslot.alloc(sizeof(uint32_t) * n); e->b_xorder.alloc(sizeof(uint32_t) * n); e->xslot_cap = n; }
  HIP_OK(hipMemcpyAsync(e->b_xslot.p, e->h_slot.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, e->stream));
  {   // the rows in shape order (stable counting sort by slot): consecutive workgroups of k_expand then copy out of the same shape row
    e->h_xorder.resize(n);
    std::vector<uint32_t> start((size_t)p.ns + 1, 0u);
    for (uint32_t i = 0; i < n; i++) start[e->h_slot[i] + 1]++;
    for (uint32_t sidx = 0; sidx < p.ns; sidx++) start[sidx + 1] += start[sidx];
    for (uint32_t i = 0; i < n; i++) e->h_xorder[start[e->h_slot[i]]++] = i;
    HIP_OK(hipMemcpyAsync(e->b_xorder.p, e->h_xorder.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, e->stream));
    // ... and the chunk table of the tiled expansion: stretches of one shape, KB_XCHUNK_ROWS rows at most (start[s] is now the END of shape s's stretch)
    e->h_xchunks.clear();
    for (uint32_t sidx = 0, at = 0; sidx < p.ns; sidx++)
      while (at < start[sidx]) { const uint32_t cnt = std::min<uint32_t>(KB_XCHUNK_ROWS, start[sidx] - at); e->h_xchunks.push_back(KbXChunk{sidx, at, cnt, 0u}); at += cnt; }
    e->b_xchunks.alloc(sizeof(KbXChunk) * std::max<size_t>(e->h_xchunks.size(), 1));
    HIP_OK(hipMemcpyAsync(e->b_xchunks.p, e->h_xchunks.data(), sizeof(KbXChunk) * e->h_xchunks.size(), hipMemcpyHostToDevice, e->stream));
    p.n_xchunks = (uint32_t)e->h_xchunks.size();
    HIP_OK(hipStreamSynchronize(e->stream));   // h_xorder is pageable and reused by the next plan
  }
  HIP_OK(hipMemcpyAsync(e->b_mrows.p, e->h_mrows.data(), sizeof(uint32_t) * p.ns, hipMemcpyHostToDevice, e->stream));
  p.rs = make_round(e, 0, p.ns, 1, (int)fit_mode, false);
  p.rs.mrows = e->b_mrows.as<uint32_t>();   // this path stages its representative rows in its own buffer (can exceed a window)
  p.rs.score = e->b_sscore.as<uint16_t>();
  p.rs.maskw = e->b_smask.as<uint32_t>();
  p.r = make_round(e, 0, n, k ? k : 1, (int)fit_mode, false);
  p.r.mrows = nullptr;
  p.r.mrow_task0 = t0;
  return p;
}
static void matrix_launch(kb_engine *e, const ChunkPlan &p, uint32_t n, uint32_t k) {
  ensure_ip_scratch(e, p.direct ? n : p.ns);
  if (p.direct) {
    kb_launch_matrix(e->dev, p.r, e->stream);
    kb_launch_affinity(e->dev, p.r, e->stream);
    kb_launch_interpod(e->dev, p.r, e->stream);
    if (k) kb_launch_argmax(e->dev, p.r, e->stream);
    return;
  }
  kb_launch_matrix(e->dev, p.rs, e->stream);
  kb_launch_affinity(e->dev, p.rs, e->stream);
  kb_launch_interpod(e->dev, p.rs, e->stream);
  // shape order pays when the shape rows do not fit the L2s (C5: 23.0 -> 18.8 ms, BASELINE configs[3]: 0.66 -> 0.64 ms); while they do,
  // task order writes consecutive rows and is the faster one (C3, 509 shapes = 10 MB: 0.41 ms against 0.52; profiles/round3/call6)
  // round 6: the TILED expansion (a workgroup loads its tile of the shape row once for 64 task rows) takes the rows in shape order always; KB_EXPAND_TILES=0:
  // round 5's row-per-workgroup copy (the A/B switch of the traffic measurement)
  static const bool tiles = [] { const char *v = getenv("KB_EXPAND_TILES"); return !(v && v[0] == '0'); }();
  const bool by_shape = (size_t)p.ns * e->dev.NP * 2 > (16u << 20);
  if (tiles) kb_launch_expand(e->dev, p.rs.score, p.rs.maskw, e->b_xslot.as<uint32_t>(), e->b_xorder.as<uint32_t>(), n, p.r.score, p.r.maskw, e->stream, e->b_xchunks.as<KbXChunk>(), p.n_xchunks);
  else kb_launch_expand(e->dev, p.rs.score, p.rs.maskw, e->b_xslot.as<uint32_t>(), by_shape ? e->b_xorder.as<uint32_t>() : nullptr, n, p.r.score, p.r.maskw, e->stream);
  if (k) kb_launch_argmax(e->dev, p.r, e->stream);
}
static KbRound matrix_chunk(kb_engine *e, uint32_t t0, uint32_t n, uint32_t fit_mode, uint32_t k) {
  ChunkPlan p = matrix_plan(e, t0, n, fit_mode, k);
  matrix_launch(e, p, n, k);
  return p.r;
}

int kb_eval_matrix(kb_engine *e, uint32_t t0, uint32_t t1, uint32_t fit_mode, uint8_t *mask_bits, uint16_t *score) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    if (t0 > t1 || t1 > e->hs.T) throw EngineError(KB_E_INVALID, "row range out of bounds");
    const uint32_t N = e->hs.N, NP = e->dev.NP;
    const size_t rowb = ((size_t)N + 7) / 8;
    const uint32_t chunk = 4096;
    for (uint32_t a = t0; a < t1; a += chunk) {
      uint32_t n = std::min(chunk, t1 - a);
      matrix_chunk(e, a, n, fit_mode, 0);
      if (score)
        HIP_OK(hipMemcpy2DAsync(score + (size_t)(a - t0) * N, sizeof(uint16_t) * N, e->b_score.p, sizeof(uint16_t) * NP, sizeof(uint16_t) * N, n,
                                hipMemcpyDeviceToHost, e->stream));
      if (mask_bits)
        HIP_OK(hipMemcpy2DAsync(mask_bits + (size_t)(a - t0) * rowb, rowb, e->b_maskw.p, NP / 8, rowb, n, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipStreamSynchronize(e->stream));
    }
    HIP_OK(hipGetLastError());
  });
}

int kb_argmax_rows(kb_engine *e, uint32_t t0, uint32_t t1, uint32_t fit_mode, uint32_t k, uint32_t *out_node, uint16_t *out_score) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    if (t0 > t1 || t1 > e->hs.T) throw EngineError(KB_E_INVALID, "row range out of bounds");
    if (k == 0 || k > KB_MAX_TOPK) throw EngineError(KB_E_INVALID, "k must be in 1..4096");
    const uint32_t chunk = std::max<uint32_t>(1, std::min<uint32_t>(4096, (1u << 20) / k));
    std::vector<unsigned long long> keys((size_t)chunk * k);
    for (uint32_t a = t0; a < t1; a += chunk) {
      uint32_t n = std::min(chunk, t1 - a);
      matrix_chunk(e, a, n, fit_mode, k);
      HIP_OK(hipMemcpyAsync(keys.data(), e->b_keys.p, sizeof(unsigned long long) * (size_t)n * k, hipMemcpyDeviceToHost, e->stream));
      HIP_OK(hipStreamSynchronize(e->stream));
      for (size_t i = 0; i < (size_t)n * k; i++) {
        size_t o = (size_t)(a - t0) * k + i;
        if (keys[i] == 0ull) { out_node[o] = KB_NONE; if (out_score) out_score[o] = 0; }
        else { out_node[o] = KB_KEY_NODE(keys[i]); if (out_score) out_score[o] = (uint16_t)KB_KEY_SCORE(keys[i]); }
      }
    }
    HIP_OK(hipGetLastError());
  });
}

int kb_bench_matrix(kb_engine *e, uint32_t t0, uint32_t t1, uint32_t fit_mode, uint32_t reps, double *ms_avg) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    if (t0 >= t1 || t1 > e->hs.T || reps == 0) throw EngineError(KB_E_INVALID, "bad range / reps");
    uint32_t n = t1 - t0;
    ChunkPlan p = matrix_plan(e, t0, n, fit_mode, 0);
    matrix_launch(e, p, n, 0);   // warm-up
    HIP_OK(hipStreamSynchronize(e->stream));
    Timer &tm = get_timer(e, 4);
    HIP_OK(hipEventRecord(tm.a, e->stream));
    for (uint32_t i = 0; i < reps; i++) matrix_launch(e, p, n, 0);   // per-shape evaluation + row expansion: the whole matrix
    HIP_OK(hipEventRecord(tm.b, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipGetLastError());
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, tm.a, tm.b));
    if (ms_avg) *ms_avg = (double)ms / reps;
  });
}

int kb_get_binds(kb_engine *e, uint32_t *task_node_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    quiesce(e);
    HIP_OK(hipMemcpy(task_node_out, e->b_tbind.p, sizeof(uint32_t) * e->hs.T, hipMemcpyDeviceToHost));
    uint64_t nb = 0;
    for (uint32_t t = 0; t < e->hs.T; t++) nb += task_node_out[t] != KB_NONE;
    e->stats.binds = nb;
  });
}

int kb_get_task_state(kb_engine *e, uint8_t *status_out, uint32_t *node_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    quiesce(e);
    if (status_out) HIP_OK(hipMemcpy(status_out, e->b_tstatus.p, e->hs.T, hipMemcpyDeviceToHost));
    if (node_out) HIP_OK(hipMemcpy(node_out, e->b_tnode.p, sizeof(uint32_t) * e->hs.T, hipMemcpyDeviceToHost));
  });
}

int kb_get_node_state(kb_engine *e, double *idle, double *releasing, int64_t *nz_cpu, int64_t *nz_mem, int32_t *pod_cnt) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    quiesce(e);
    const uint32_t N = e->hs.N, NP = e->dev.NP;
    const int R = e->hs.R;
    if (idle) HIP_OK(hipMemcpy2D(idle, sizeof(double) * N, e->b_idle.p, sizeof(double) * NP, sizeof(double) * N, R, hipMemcpyDeviceToHost));
    if (releasing) HIP_OK(hipMemcpy2D(releasing, sizeof(double) * N, e->b_rel.p, sizeof(double) * NP, sizeof(double) * N, R, hipMemcpyDeviceToHost));
    if (nz_cpu) HIP_OK(hipMemcpy(nz_cpu, e->b_nzc.p, sizeof(int64_t) * N, hipMemcpyDeviceToHost));
    if (nz_mem) HIP_OK(hipMemcpy(nz_mem, e->b_nzm.p, sizeof(int64_t) * N, hipMemcpyDeviceToHost));
    if (pod_cnt) HIP_OK(hipMemcpy(pod_cnt, e->b_podcnt.p, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
  });
}

int kb_get_shares(kb_engine *e, double *job_share, double *queue_share, double *queue_deserved) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    const HostSession &hs = e->hs;
    if (job_share) std::memcpy(job_share, hs.job_share.data(), sizeof(double) * hs.J);
    if (queue_share) std::memcpy(queue_share, hs.queue_share.data(), sizeof(double) * hs.Q);
    if (queue_deserved)
      for (uint32_t q = 0; q < hs.Q; q++)
        for (int d = 0; d < hs.R; d++) queue_deserved[(size_t)d * hs.Q + q] = hs.deserved[q].get(d);
  });
}

int kb_get_stats(kb_engine *e, kb_stats *out) {
  if (!e || !out) return KB_E_INVALID;
  *out = e->stats;
  return KB_OK;
}

// ---- round-granular API for task-row sharding across GPUs ----
int kb_round_begin(kb_engine *e, uint32_t action, uint32_t *n_rows, uint32_t *n_mrows, uint32_t *list_len) {
  if (!e || !n_rows) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->loaded) throw EngineError(KB_E_STATE, "no session loaded");
    if (action > 1) throw EngineError(KB_E_INVALID, "action must be 0 (allocate) or 1 (backfill)");
    if (!e->mg) e->mg = new MgState();
    MgState &m = *e->mg;
    if (!m.run.active) {
      action_entry_guards(e);   // (the same refusals as kb_run_allocate / kb_run_backfill: a tainted session, a stale NodeName behind a discarded statement)
      m.run.begin(e, action);
      m.run.probe_dead_shapes(e);   // as run_action does: shapes no node can take from the start never cost a round (every replica reads the same state: the same answer)
      m.in_round = false;
      m.rounds_begun = 0;
      m.chk_counter.alloc(sizeof(uint32_t));
      HIP_OK(hipMemsetAsync(m.chk_counter.p, 0, sizeof(uint32_t), e->stream));
      m.chk_valid = true;
    } else if (m.run.action != action) {
      throw EngineError(KB_E_STATE, "another action is still in progress");
    }
    if (m.in_round) throw EngineError(KB_E_STATE, "previous round not applied");
    uint32_t n = m.run.plan(e);
    *n_rows = n;
    if (n_mrows) *n_mrows = 0;
    if (list_len) *list_len = 0;
    if (n == 0) {   // action complete: gang ballot + share reduction, consistency checks
      m.last_decs = m.run.decs;
      m.run.finish(e);
      return;
    }
    m.ctx = round_prepare(e, n, action == 0 ? 1 : 2, action == 1);
    m.had_candidates = false;
    m.in_round = true;
    // round-start copy of the node state: the reduced deltas are applied to it.  The copy of the round before stays (kb_round_check compares
    // that round's reduced deltas against the two of them, one round late)
    const size_t NP = e->dev.NP;
    const int R = e->hs.R;
    m.q_idle.swap(m.s_idle); m.q_rel.swap(m.s_rel); m.q_nzc.swap(m.s_nzc); m.q_nzm.swap(m.s_nzm); m.q_podcnt.swap(m.s_podcnt);
    m.rounds_begun += 1;
    auto snap = [&](DevBuf &dst, const DevBuf &src) {
      if (dst.bytes != src.bytes) dst.alloc(src.bytes);
      HIP_OK(hipMemcpyAsync(dst.p, src.p, src.bytes, hipMemcpyDeviceToDevice, e->stream));
    };
    snap(m.s_idle, e->b_idle); snap(m.s_rel, e->b_rel); snap(m.s_nzc, e->b_nzc); snap(m.s_nzm, e->b_nzm); snap(m.s_podcnt, e->b_podcnt);
    (void)NP; (void)R;
    if (n_mrows) *n_mrows = m.ctx.ns;
    if (list_len) *list_len = m.ctx.L;
  });
}

int kb_round_candidates(kb_engine *e, uint32_t mrow0, uint32_t mrow1, uint64_t dev_keys_ptr) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !e->mg->in_round) throw EngineError(KB_E_STATE, "kb_round_begin must precede kb_round_candidates");
    MgState &m = *e->mg;
    if (mrow1 > m.ctx.ns) mrow1 = m.ctx.ns;
    if (mrow0 >= mrow1) return;   // this rank's shard is empty (fewer shapes than ranks)
    if (!dev_keys_ptr) throw EngineError(KB_E_INVALID, "null key buffer");
    round_candidates(e, m.ctx, mrow0, mrow1, reinterpret_cast<unsigned long long *>(dev_keys_ptr));
    m.had_candidates = true;
    if (e->stream == e->own_stream) {   // the caller's collective runs on another stream: it must see finished lists
      HIP_OK(hipStreamSynchronize(e->stream));
      HIP_OK(hipGetLastError());
    }
  });
}

int kb_round_commit(kb_engine *e, uint64_t dev_all_keys_ptr, uint32_t own_row0, uint32_t own_row1, uint64_t dev_delta_ptr) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !e->mg->in_round) throw EngineError(KB_E_STATE, "kb_round_begin must precede kb_round_commit");
    MgState &m = *e->mg;
    if (!dev_all_keys_ptr) throw EngineError(KB_E_INVALID, "null key table");
    double *delta = reinterpret_cast<double *>(dev_delta_ptr);
    if (delta) HIP_OK(hipMemsetAsync(delta, 0, sizeof(double) * (size_t)e->dev.NP * (2 * (size_t)e->hs.R + 3), e->stream));
    round_commit(e, m.ctx, reinterpret_cast<unsigned long long *>(dev_all_keys_ptr), delta, own_row0, own_row1);
    round_collect(e, m.ctx, m.had_candidates, m.n_done, m.reason);
    m.committed = true;
  });
}

int kb_round_apply(kb_engine *e, uint64_t dev_delta_ptr, uint32_t *done) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !e->mg->in_round || !e->mg->committed) throw EngineError(KB_E_STATE, "kb_round_commit must precede kb_round_apply");
    MgState &m = *e->mg;
    if (dev_delta_ptr) {
      // node state for the next round = round-start state + all-reduced deltas; it must equal this replica's own commit
      uint32_t mism = kb_apply_deltas(e->dev, m.s_idle.as<double>(), m.s_rel.as<double>(), m.s_nzc.as<long long>(), m.s_nzm.as<long long>(),
                                      m.s_podcnt.as<int>(), reinterpret_cast<const double *>(dev_delta_ptr), e->b_out.as<uint32_t>() + 8, e->stream);   // word [4] of the output block
      if (mism) throw EngineError(KB_E_INTERNAL, "replicas diverged: reduced per-node deltas differ from the local commit at " + std::to_string(mism) + " values");
    }
    m.run.absorb(e, m.ctx.n, m.n_done, m.reason);
    // a speculation break: the feasibility probe marks every shape that died with the one that broke the round (run_action's rule; without it the
    // split paid one round per dead shape — 61 breaks per 100k x 10k cycle against the single-GPU path's 14, round 5)
    if (m.reason != KB_REASON_DONE && m.reason != KB_REASON_RENORM) m.run.probe_dead_shapes(e);
    m.in_round = false;
    m.committed = false;
    if (done) *done = 0;
  });
}

int kb_round_check(kb_engine *e, uint64_t dev_delta_ptr, uint32_t against_live) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !dev_delta_ptr) throw EngineError(KB_E_STATE, "kb_round_check: no round-mode action / null delta buffer");
    MgState &m = *e->mg;
    const double *delta = reinterpret_cast<const double *>(dev_delta_ptr);
    const KbDev &d = e->dev;
    if (against_live) {   // the action's last round: its start copy + deltas == the live state (the kb_round_begin that ended the action took no copy)
      if (m.in_round || m.rounds_begun < 1) throw EngineError(KB_E_STATE, "kb_round_check(against_live): behind the kb_round_begin that ended the action");
      kb_check_deltas(d, m.cur(), KbNodeCopy{d.idle, d.rel, d.nzc, d.nzm, d.podcnt}, delta, m.chk_counter.as<uint32_t>(), e->stream);
    } else {              // round k's deltas, round k + 1 begun: the two start copies
      if (!m.in_round || m.rounds_begun < 2) throw EngineError(KB_E_STATE, "kb_round_check: behind the kb_round_begin of the NEXT round");
      kb_check_deltas(d, m.prev(), m.cur(), delta, m.chk_counter.as<uint32_t>(), e->stream);
    }
  });
}

int kb_round_check_result(kb_engine *e, uint32_t *mismatches) {
  if (!e || !mismatches) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg || !e->mg->chk_valid) throw EngineError(KB_E_STATE, "kb_round_check_result: no round-mode action has run");
    uint32_t h = 0;
    HIP_OK(hipMemcpyAsync(&h, e->mg->chk_counter.p, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    *mismatches = h;
  });
}

int kb_round_decisions(kb_engine *e, kb_decision *out, uint64_t cap, uint64_t *n_out) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    if (!e->mg) throw EngineError(KB_E_STATE, "no round-mode action has run");
    const std::vector<kb_decision> &d = e->mg->run.active ? e->mg->run.decs : e->mg->last_decs;
    if (n_out) *n_out = d.size();
    if (d.size() > cap) throw EngineError(KB_E_CAPACITY, "decision buffer too small");
    if (out && !d.empty()) std::memcpy(out, d.data(), sizeof(kb_decision) * d.size());
  });
}

int kb_engine_use_stream(kb_engine *e, uint64_t hip_stream) {
  if (!e) return KB_E_INVALID;
  return guarded(e, [&]() {
    HIP_OK(hipStreamSynchronize(e->stream));
    e->async_pending = false;
    e->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : e->own_stream;
  });
}

int kb_round_delta_doubles(const kb_engine *e, uint64_t *n_doubles) {
  if (!e || !n_doubles) return KB_E_INVALID;
  *n_doubles = (uint64_t)e->dev.NP * (2ull * e->hs.R + 3ull);
  return KB_OK;
}

}  // extern "C"
