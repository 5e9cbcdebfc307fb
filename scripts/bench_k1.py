#!/usr/bin/env python
"""Times the materialised-matrix launch of K1 (kb_bench_matrix) on BASELINE config 3/4: GB/s vs the 8 TB/s HBM peak."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
cfg_idx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
conf = kbm.conf.load_scheduler_conf()
params = kbm.snapshot.synth_config(cfg_idx)
if len(sys.argv) > 2 and sys.argv[2] == "diverse":      # every job its own request: thousands of distinct task shapes
    params.diverse_requests = True
snap = kbm.snapshot.synth(params)
e = engine.Engine(conf)
e.load(snap)
T, N, R = snap.n_tasks, snap.n_nodes, snap.n_res
for fit in (1,):
    ms = min(e.bench_matrix(0, T, reps=5, fit_mode=fit) for _ in range(3))
    alg = T * N * 2.125 + N * (16 * R + 44) + T * (8 * R + 24)
    print(f"KB_K1_RUNS={os.environ.get('KB_K1_RUNS', '1')} config {cfg_idx}{' diverse' if params.diverse_requests else ''} KB_K1_DIRECT={os.environ.get('KB_K1_DIRECT', 'auto')} R={R} fit={fit}: {ms:.4f} ms  {alg / ms / 1e6:.1f} GB/s  {alg / ms / 1e6 / 8000:.3f} of peak  {T * N / ms / 1e6:.1f} Gevals/s")
