// kb_engine_int.hpp — what the translation units of the engine's host side share (round 6: kb_engine.cpp was one 2 400-line file): the engine
// object, its device / pinned buffers, a round's context, the action's host state (ActionRun) and the helpers' declarations.
//   kb_engine.cpp   create / destroy, buffers, timers, the closing reduction, the getters
//   kb_load.cpp     kb_session_load (host session, uploads, water-fill launch), kb_session_reset
//   kb_rounds.cpp   a round's three host steps, chaining and overlap, run_action (allocate / backfill), the round-granular API of the task-row split
//   kb_evict.cpp    the bridge between the evict machine (kb_preempt.cpp) and the device lists: kb_run_preempt / kb_run_reclaim
//   kb_matrix.cpp   kb_eval_matrix / kb_argmax_rows / kb_bench_matrix (the materialised matrix)
//
//
// Round structure: the host order machine speculates the reference's task order for a window of W
// tasks (assuming each gets a node, which only ever fails when a whole feasibility class has died — and that is
// monotone inside one action), the device evaluates the window's mask+score matrix against the round-start node
// state (K1, once per distinct task shape), builds each shape's sorted candidate list (K3) and commits the window in the
// reference's order, a run of same-shape rows at a time (K5).  A mis-speculation (no feasible node / Pipeline instead of Allocate) stops the commit kernel at that row;
// the host rolls the order machine back to the round start, replays the confirmed prefix and re-plans.
#pragma once
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

namespace kbe {


inline double now_ms() {
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

}  // namespace kbe
using namespace kbe;

struct MgState;
void mg_free(MgState *m);
void mg_reset(MgState *m);

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
  std::unique_ptr<PreemptMachine> evict_machine;   // kb_evict.cpp: kept for the engine's life (grow-only tables, like the device buffers)

  ~kb_engine() {
    mg_free(mg);
    for (auto &t : ev) t.destroy();
    if (own_stream) (void)hipStreamDestroy(own_stream);
    if (stream_b) (void)hipStreamDestroy(stream_b);
  }
};

extern thread_local std::string g_create_err;

namespace kbe {

// ---- buffers, timers, the closing reduction (kb_engine.cpp)
void ensure_window_buffers(kb_engine *e, uint32_t rows);
void ensure_ip_scratch(kb_engine *e, size_t rows);
void ensure_matrix_buffers(kb_engine *e, uint32_t mrows, uint32_t L);
Timer &get_timer(kb_engine *e, size_t i);
void quiesce(kb_engine *e);
void run_finalize(kb_engine *e, const std::function<void()> &after_sync = nullptr);
int guarded(kb_engine *e, const std::function<void()> &fn);

// ---- rounds (kb_rounds.cpp)
KbRound make_round(kb_engine *e, uint32_t n_rows, uint32_t n_mrows, uint32_t L, int fit_mode, bool backfill, uint32_t buf = 0);
uint32_t assign_shapes(kb_engine *e, uint32_t n, const uint32_t *rows = nullptr);
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
RoundCtx round_prepare(kb_engine *e, uint32_t n, int fit_mode, bool backfill, bool gather_in_matrix = false, const uint32_t *rows = nullptr,
                       uint32_t buf = 0, uint32_t chain_expect = 0);
void round_candidates(kb_engine *e, const RoundCtx &c, uint32_t m0, uint32_t m1, unsigned long long *keys);
void ensure_overlap_buffers(kb_engine *e, uint32_t mrows, uint32_t stale_L);
void round_candidates_overlapped(kb_engine *e, RoundCtx &c, uint32_t n_prev, unsigned long long *keys);
void round_commit(kb_engine *e, const RoundCtx &c, unsigned long long *keys, double *delta, uint32_t own0, uint32_t own1);
void round_collect(kb_engine *e, const RoundCtx &c, bool had_candidates, uint32_t &n_done, uint32_t &reason);
void check_aggregates(kb_engine *e, const OrderMachine &om);


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
  uint64_t popped = 0, spec_pops = 0, spec_pops_next = 0, spec_pops_next2 = 0;
  std::vector<uint32_t> rows_next;   // the window speculated behind the one in flight
  std::vector<uint32_t> rows_next2;  // ... and the one behind that: planned while the first two are on the device, launched when the first one's answer is in
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
  // `second`: the window behind the speculated one (rows_next2), behind a third roll-back point.  It is only PLANNED ahead — its matrix would be two
  // rounds stale; run_action launches it when the round in flight has answered, and then has its rows ready: the order machine's ~30 us per
  // window are no longer between a round's answer and the next launches (1M x 50k: the arg-max launch of the second stream was late for the
  // commit launch by ~3 us per round, and by more on a slower host)
  uint32_t plan_ahead(kb_engine *e, bool second = false) {
    HostSession &hs = e->hs;
    const uint32_t W = e->eff_window;
    double t0 = now_ms();
    om.push_checkpoint();
    std::vector<uint32_t> &rows = second ? rows_next2 : rows_next;
    uint64_t &pops = second ? spec_pops_next2 : spec_pops_next;
    if (rows.size() < W) rows.resize(W);
    uint32_t n = 0, t, nshapes = 0;
    pops = 0;
    new_window(e);
    while (n < W && om.next(t)) {
      pops++;
      if (dead[hs.t_feas_shape[t]]) { om.report(Outcome::NoFeasibleNode); continue; }
      if (!admit_shape(e, hs.t_row_shape[t], nshapes) || (n > 0 && !hs.t_ip_subject.empty() && hs.t_ip_subject[t]) || (n > 0 && hs.wide(t))) { om.rollback_last_pop(); pops--; break; }
      rows[n++] = t;
      om.report(Outcome::Allocated);
      if (hs.wide(t)) break;
    }
    host_ms += now_ms() - t0;
    return n;
  }
  // the window in flight is confirmed: the speculated one becomes the current one; with `have_second` the one planned behind it moves up
  void promote(kb_engine *e, uint32_t n_next, bool have_second = false) {
    om.pop_commit();
    if (n_next) std::memcpy(e->h_rows.data(), rows_next.data(), sizeof(uint32_t) * n_next);
    spec_pops = spec_pops_next;
    if (n_next == 0) popped += spec_pops;
    if (have_second) { rows_next.swap(rows_next2); spec_pops_next = spec_pops_next2; }
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

}  // namespace kbe

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
