#!/usr/bin/env python
"""Times the three actions BASELINE configs[4] names — allocate, backfill, preempt — on a synthetic snapshot (default: config 5 at
full size, 1M tasks x 50k nodes) and prints what preempt did.  python scripts/time_preempt.py [config] [scale]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
CONF = """
actions: "allocate, backfill, preempt"
tiers:
- plugins:
  - name: priority
  - name: gang
  - name: conformance
- plugins:
  - name: drf
  - name: predicates
  - name: proportion
  - name: nodeorder
"""
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
snap = kbm.snapshot.synth(kbm.snapshot.synth_config(idx, scale))
e = engine.Engine(kbm.conf.load_scheduler_conf(CONF))
e.load(snap)
t0 = time.perf_counter(); e.run_allocate(); t1 = time.perf_counter(); e.run_backfill(); t2 = time.perf_counter(); e.run_preempt(); t3 = time.perf_counter()
j = e.last_journal
abi = kbm.abi
print(f"config {idx} x{scale}: {snap.n_tasks} tasks x {snap.n_nodes} nodes: allocate {1e3 * (t1 - t0):.1f} ms, backfill {1e3 * (t2 - t1):.1f} ms, "
      f"preempt {1e3 * (t3 - t2):.1f} ms; journal {len(j)} entries: {(j[:, 0] == abi.OP_EVICT).sum()} evict, {(j[:, 0] == abi.OP_PIPELINE).sum()} pipeline, "
      f"{(j[:, 0] == abi.OP_COMMIT).sum()} commit, {(j[:, 0] == abi.OP_DISCARD).sum()} discard; committed evictions {len(e.evictions())}; "
      f"preemptors popped {e.stats()['tasks_popped']}")
