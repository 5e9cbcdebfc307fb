#!/usr/bin/env python
"""One-off differential hunt on the CPU beyond the committed suite: preempt / reclaim in sessions with inter-pod (anti)affinity terms (round 3,
KB_EVICT_INTERPOD=1).  For every seed: the engine's host side on the emulated device against the C oracle (journal, evictions, statuses, node
state, shares), the oracle against tests/pyref.py (which recounts the inter-pod counts from the task statuses), and the oracle's predicate on the
session it leaves behind against the object-level restatement (tests/interpod_objref.py).
    python scripts/hunt_interpod_evict_cpu.py first_seed last_seed        -> prints every divergence, exit code 1 if any"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["KB_EVICT_INTERPOD"] = "1"
import pytest  # noqa: E402

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
import oracle  # noqa: E402
import test_emu_engine_cpu as emu  # noqa: E402
import test_gpu_preempt as gp  # noqa: E402
import test_interpod_oracle_cpu as ipo  # noqa: E402
from test_pyref_vs_oracle import _pyref_vs_oracle_evict  # noqa: E402

oracle.build()
engine.LIB_PATH, engine._LIB = emu.build_emulated_library(), None
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 1200)
bad = ran = skipped = 0
for seed in range(lo, hi):
    try:
        cfg, snap, order = ipo.interpod_evict_case(seed)
    except Exception:
        skipped += 1
        continue
    if snap.interpod is None:
        skipped += 1
        continue
    for name, fn in (("engine vs oracle", lambda: gp._run_both(oracle, cfg, snap, order, seed)),
                     ("oracle vs pyref", lambda: _pyref_vs_oracle_evict(oracle, cfg, snap, order, seed)),
                     ("oracle vs objects", lambda: ipo.test_predicate_after_evict_actions_equals_the_object_level_answer.__wrapped__(oracle, seed)
                      if hasattr(ipo.test_predicate_after_evict_actions_equals_the_object_level_answer, "__wrapped__")
                      else ipo.test_predicate_after_evict_actions_equals_the_object_level_answer(oracle, seed))):
        try:
            fn()
            ran += 1
        except pytest.skip.Exception:
            skipped += 1
        except BaseException as err:   # noqa: BLE001  (assertion or engine error: report and go on)
            bad += 1
            print(f"seed {seed} {order} {name}: {type(err).__name__}: {str(err)[:300]}", flush=True)
print(f"seeds [{lo}, {hi}): {ran} comparisons, {skipped} skipped, {bad} divergences")
sys.exit(1 if bad else 0)
