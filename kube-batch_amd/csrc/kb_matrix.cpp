// kb_matrix.cpp — kb_eval_matrix / kb_argmax_rows / kb_bench_matrix: the materialised task x node matrix (K1 over the distinct shapes + the row
// expansion, or every row directly) behind the parity tests and the roofline measurement.  Split out of kb_engine.cpp in round 6 without a
// change of behaviour (kb_engine_int.hpp has the map).
#include "kb_engine_int.hpp"

extern "C" {

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
  // Expansion streams every task row out of its shape's row.  The rows are expanded in SHAPE order (`order`), so a shape
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
  {   // the rows in shape order (stable counting sort by slot): the chunks of k_expand_tiles are stretches of it
    e->h_xorder.resize(n);
    std::vector<uint32_t> start((size_t)p.ns + 1, 0u);
    for (uint32_t i = 0; i < n; i++) start[e->h_slot[i] + 1]++;
    for (uint32_t sidx = 0; sidx < p.ns; sidx++) start[sidx + 1] += start[sidx];
    for (uint32_t i = 0; i < n; i++) e->h_xorder[start[e->h_slot[i]]++] = i;
    HIP_OK(hipMemcpyAsync(e->b_xorder.p, e->h_xorder.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, e->stream));
    // ... and the chunk table of the tiled expansion: stretches of `order` that share a shape, KB_XCHUNK_ROWS rows at most (start[s] is now the END of
    // shape s's stretch): a shape row's tile comes from memory once per 64 of its task rows.  (Chunks of CONSECUTIVE task rows — the tasks of a job are
    // adjacent and share a shape, the matrix is then written front to back — measured no faster and read more: 0.447 - 0.451 ms and 1.11 x the algorithmic
    // bytes against 0.438 - 0.446 and 1.04 x at 100k x 10k, profiles/round6/call10_expand_row_order_chunks/, call14_profile_final/.)
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
  // round 6: the TILED expansion — a workgroup loads its tile of the shape row once for 64 task rows, the rows in shape order.  (Round 5's copy, one
  // workgroup per task row, re-fetched the shape rows through eight L2s — 1.25 x the algorithmic bytes against 1.04 x — and was slower at R = 16 and at
  // 1M x 50k on every box, within +- 6 % at 100k x 10k depending on the box: profiles/round6/call8..., call10..., call12...; retired with its switch.)
  kb_launch_expand(e->dev, p.rs.score, p.rs.maskw, e->b_xslot.as<uint32_t>(), e->b_xorder.as<uint32_t>(), n, p.r.score, p.r.maskw, e->stream, e->b_xchunks.as<KbXChunk>(), p.n_xchunks);
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

}  // extern "C"
