#!/usr/bin/env python3
"""The host order machine alone (kube-batch_amd/csrc/kb_order.cpp through tests/host_harness/order_harness.cpp), one allocate action with every round
confirmed: microseconds per round of `window` rows, with the roll-back points keeping the heap arrays as copies and as journals.
    python scripts/time_order_machine.py [config] [scale]        (config 3: 100k tasks / ~10k jobs; 5: 1M tasks / ~100k jobs)"""
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_host_order_cpu as T  # noqa: E402

kbm = importlib.import_module("kube-batch_amd")


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(config, scale))
    cfg = kbm.conf.load_scheduler_conf()
    L = T._build()
    L.hh_bench.restype = C.c_double
    t0 = time.time()
    p = T.pyref.Session(T.cases._tiers(cfg), snap)
    print(f"config {config} scale {scale}: {snap.n_tasks} tasks, {snap.n_jobs} jobs, {snap.n_queues} queues (reference session in Python: {time.time() - t0:.1f} s)")
    for mode, name in ((0, "copies"), (1, "journals")):
        for rep in range(3):
            L.hh_set_journal(C.c_int(mode))
            m = T.Machine(L, cfg, snap, p)
            rows = C.c_uint64()
            s = L.hh_bench(m.h, C.c_uint32(256), C.byref(rows))
            m.close()
            rounds = max(1, (rows.value + 255) // 256)
            print(f"  {name:9s} {s * 1e3:8.2f} ms, {rows.value} rows, {s * 1e6 / rounds:6.2f} us per round of 256, {s * 1e9 / max(1, rows.value):6.1f} ns per row")
    L.hh_set_journal(C.c_int(-1))


if __name__ == "__main__":
    main()
