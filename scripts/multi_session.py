#!/usr/bin/env python3
"""Several engines on ONE GPU in ONE process, a thread each (ctypes releases the GIL inside the C ABI): every engine schedules its own session
(the generator's seed + k, like bench.py's sessions mode across GPUs) cycle after cycle on its own streams.  One commit workgroup owns a cycle
and sits on one of 256 CUs, so sessions that do not depend on each other can share the device: this measures how far.
    python scripts/multi_session.py [engines=1,2,4,8] [config=3] [cycles=5]      -> one line per engine count: per-session ms, aggregate evals/s
Every engine's last cycle is held to the committed golden digest of its snapshot (tests/golden/bench_rank_digests.json) where one exists."""
import importlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
distmod = importlib.import_module("kube-batch_amd.dist")
import bench  # noqa: E402  (RANK_SEED_STRIDE)


def main():
    counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
    config = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    conf = kbm.conf.load_scheduler_conf()
    import dataclasses
    params = kbm.snapshot.synth_config(config, 1.0)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_rank_digests.json"))).get(f"config{config}_scale1", {})
    kmax = max(counts)
    snaps = [kbm.snapshot.synth(dataclasses.replace(params, seed=params.seed + bench.RANK_SEED_STRIDE * k)) for k in range(kmax)]
    engines = []
    for k in range(kmax):
        e = engine.Engine(conf, device=0)
        e.load(snaps[k])
        e.reset(); e.run(["allocate", "backfill"])          # warm-up: buffers, first-launch costs
        engines.append(e)
    for K in counts:
        out = [None] * K
        start = threading.Barrier(K + 1)

        def work(k):
            e = engines[k]
            start.wait()
            t0 = time.perf_counter()
            ev0 = e.stats()["evals"]
            dec = None
            for _ in range(cycles):
                e.reset()
                dec = e.run(["allocate", "backfill"])
            t1 = time.perf_counter()
            out[k] = (t1 - t0, e.stats()["evals"] - ev0, dec)

        th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        for t in th:
            t.start()
        start.wait()
        w0 = time.perf_counter()
        for t in th:
            t.join()
        wall = time.perf_counter() - w0
        ok = []
        for k in range(K):
            want = golden.get(str(k))
            mine = distmod.ReplicatedCycle.digest(out[k][2], engines[k].binds(), None, None)
            ok.append(None if want is None else int(want) == mine)
        per = [o[0] * 1e3 / cycles for o in out]
        agg = sum(o[1] for o in out) / wall
        print(json.dumps({"engines_on_one_gpu": K, "config": config, "cycles_each": cycles, "ms_per_cycle_min_max": [round(min(per), 2), round(max(per), 2)],
                          "aggregate_evals_per_s": agg, "sessions_per_s": K * cycles / wall, "verified": (all(ok) if all(v is not None for v in ok) else None)}), flush=True)
    for e in engines:
        e.close()


if __name__ == "__main__":
    main()
