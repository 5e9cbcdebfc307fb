#!/usr/bin/env python
"""One-off differential hunt on CPU, beyond the committed suite: the engine's host code for session load + preempt / reclaim
(tests/host_harness/evict_harness.cpp around kb_session.cpp and kb_preempt.cpp) against the C oracle, on many more (cluster, evict
order, tier layout) combinations than tests/test_host_evict_cpu.py keeps.  No GPU.
python scripts/hunt_evict_cpu.py [first_seed] [last_seed]   -> prints every divergence, exit code 1 if any.
KB_HUNT_EMU=1: the same combinations through the WHOLE engine on the emulated device (tests/test_emu_engine_cpu.py): run_evict_action's
list path, state carried between actions by kb_engine.cpp itself, allocate / backfill mixed in."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pytest  # noqa: E402

import oracle as oracle_mod  # noqa: E402
import rawgen  # noqa: E402
import test_host_evict_cpu as T  # noqa: E402
import test_pyref_vs_oracle as cases  # noqa: E402

conf = importlib.import_module("kube-batch_amd").conf


def main():
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 200)
    emu = os.environ.get("KB_HUNT_EMU") == "1"
    if emu:
        import test_emu_engine_cpu as emu_mod
        import test_gpu_preempt as gp
        engine = importlib.import_module("kube-batch_amd.engine")
        engine.LIB_PATH, engine._LIB = emu_mod.build_emulated_library(), None
    L = T.load_harness()
    confs = [cases.CONF_FULL] + cases.EVICT_CONFS
    orders = cases.EVICT_ORDERS + [["preempt", "reclaim", "preempt"], ["preempt", "preempt", "preempt"], ["reclaim", "preempt", "reclaim", "preempt"]]
    if emu:
        orders = orders + [["allocate", "preempt"], ["reclaim", "allocate", "backfill", "preempt"], ["preempt", "allocate", "backfill", "reclaim"], ["allocate", "backfill", "preempt", "allocate"]]
    ok = skipped = bad = 0
    for seed in range(lo, hi):
        snaps = [("evict", cases._evict_case(seed)[1]), ("raw", rawgen.raw_snapshot(seed)), ("alloc", cases._case(seed)[1])]
        for kind, snap in snaps:
            for ci, ct in enumerate(confs):
                order = orders[(seed * 3 + ci) % len(orders)]
                cfg = conf.load_scheduler_conf(ct.format(actions=", ".join(order)))
                try:
                    if emu:
                        gp._run_both(oracle_mod, cfg, snap, order, (kind, seed, ci, order))
                    else:
                        T._run_both(L, oracle_mod, cfg, snap, order, (kind, seed, ci, order))
                    ok += 1
                except pytest.skip.Exception:
                    skipped += 1
                except AssertionError as e:
                    bad += 1
                    print("DIVERGES", kind, seed, "conf", ci, order, str(e)[:300])
    print(f"seeds [{lo},{hi}): {ok} equal, {skipped} outside the envelope, {bad} divergences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
