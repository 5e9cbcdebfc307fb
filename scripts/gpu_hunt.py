#!/usr/bin/env python
"""One-off differential hunt on the GPU box, beyond the committed suite: engine vs oracle on many more seeds of the fuzz generator
(tests/test_gpu_fuzz.py), the adversarial raw snapshots (tests/rawgen.py) and the inter-pod affinity clusters, under both commit
kernels.  python scripts/gpu_hunt.py [seeds_fuzz] [seeds_raw] [seeds_interpod]   -> prints every divergence, exit code 1 if any.
KB_HUNT_OFFSET=k shifts every seed range by k (fresh cases).  KB_HUNT_EMU=1 runs the same hunt WITHOUT a GPU against the engine's
host side on the emulated device of tests/host_harness (tests/test_emu_engine_cpu.py): that hunts the host logic (speculation,
roll-back, dead shapes, probe, chained rounds), not the kernels."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
abi, conf, snapmod = kbm.abi, kbm.conf, kbm.snapshot
import oracle
import rawgen
import test_gpu_fuzz as fz
from test_interpod_oracle_cpu import interpod_case

oracle.build()
OFF = int(os.environ.get("KB_HUNT_OFFSET", "0"))
if os.environ.get("KB_HUNT_EMU") == "1":
    import test_emu_engine_cpu as emu
    engine.LIB_PATH, engine._LIB = emu.build_emulated_library(), None
n_fuzz, n_raw, n_ip = (int(x) for x in (sys.argv[1:4] + ["200", "600", "400"])[:3])
bad = 0
t0 = time.time()


def compare(tag, cfg, snap, **ekw):
    global bad
    try:
        o = oracle.Oracle(cfg, snap)
        o.run(["allocate", "backfill"])
    except RuntimeError:
        return "oracle-panic"
    for k in ("run", "select"):
        os.environ["KB_COMMIT_KERNEL"] = k
        try:
            e = engine.Engine(cfg, **ekw)
            e.load(snap)
            dec = e.run(["allocate", "backfill"])
        except engine.EngineError as err:
            if err.code in (abi.KB_E_UNSUPPORTED, abi.KB_E_INVALID):
                try:
                    e.close()
                except Exception:
                    pass
                continue
            raise
        od = o.decisions()
        ok = dec.shape == od.shape and np.array_equal(dec, od) and np.array_equal(e.binds(), o.binds())
        ok = ok and all(np.array_equal(a, b) for a, b in zip(e.node_state(), o.node_state()))
        ok = ok and all(np.array_equal(a, b) for a, b in zip(e.shares(), o.shares()))
        if not ok:
            bad += 1
            print("DIVERGENCE", tag, k, dec.shape, od.shape, flush=True)
        e.close()
    o.close()
    return "ok"


for seed in range(40 + OFF, 40 + OFF + n_fuzz):
    cfg, snap, window, batch = fz._case(seed)
    compare(f"fuzz {seed}", cfg, snap, window=window, commit_batch=batch)
print("fuzz done", n_fuzz, round(time.time() - t0, 1), "s", flush=True)
import test_pyref_vs_oracle as cases
for seed in range(200 + OFF, 200 + OFF + n_raw):
    snap = rawgen.raw_snapshot(seed)
    rng = np.random.RandomState(seed)
    wl, wm, wa, wb = [int(x) for x in rng.choice([0, 1, 1, 2, 5], size=4)]
    cfg = conf.load_scheduler_conf(cases.CONF_TMPL.format(wl=wl, wm=wm, wa=wa, wb=wb))
    compare(f"raw {seed}", cfg, snap, window=int(rng.choice([0, 1, 3, 64])), commit_batch=int(rng.choice([0, 1, 5, 16])))
print("raw done", round(time.time() - t0, 1), "s", flush=True)
n_sup = 0
for seed in range(60 + OFF, 60 + OFF + n_ip):
    try:
        cfg, snap = interpod_case(seed)
    except snapmod.UnsupportedSnapshot:
        continue
    if snap.interpod is None:
        continue
    n_sup += 1
    compare(f"interpod {seed}", cfg, snap, window=[0, 64, 16, 256][seed % 4])
print("interpod done", n_sup, "supported", round(time.time() - t0, 1), "s; divergences:", bad, flush=True)
sys.exit(1 if bad else 0)
