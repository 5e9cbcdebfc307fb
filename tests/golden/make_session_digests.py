#!/usr/bin/env python
"""Regenerates tests/golden/session_digests.json: FNV-1a digests (tests/host_harness/evict_harness.cpp: eh_digest) of everything the
engine's session build (kube-batch_amd/csrc/kb_session.cpp: build_host_session) derives from a snapshot — shape ids, the vectors
LessEqual compares, the task-major request copy, drf / proportion totals, the water-filled deserved.

These pin the session build against its own committed past: an optimisation of that code (it runs once per scheduling cycle, on the
host) must reproduce every derived array bit for bit.  The values themselves are checked against tests/pyref.py in
tests/test_host_evict_cpu.py.  Run from the repository root:  python tests/golden/make_session_digests.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]


def snapshots():
    import importlib
    import rawgen
    import test_interpod_oracle_cpu as ipc
    import test_pyref_vs_oracle as cases
    kbm = importlib.import_module("kube-batch_amd")
    for idx, scale in ((2, 0.2), (3, 0.1), (4, 0.05), (5, 0.01)):
        yield f"config{idx}_x{scale}", kbm.snapshot.synth(kbm.snapshot.synth_config(idx, scale))
    p = kbm.snapshot.synth_config(3, 0.05)
    p.diverse_requests = True
    yield "config3_x0.05_diverse", kbm.snapshot.synth(p)
    for seed in range(8):
        yield f"case{seed}", cases._case(seed)[1]
        yield f"evict{seed}", cases._evict_case(seed)[1]
        yield f"raw{seed}", rawgen.raw_snapshot(seed)
    for seed in range(4):
        yield f"interpod{seed}", ipc.interpod_case(seed, False)[1]


def digests():
    import importlib
    import test_host_evict_cpu as T
    import test_pyref_vs_oracle as cases
    conf = importlib.import_module("kube-batch_amd").conf
    L = T.load_harness()
    cfg = conf.load_scheduler_conf(cases.CONF_FULL.format(actions="allocate, backfill"))
    out = {}
    for name, snap in snapshots():
        try:
            e = T.HostEngine(L, cfg, snap)
        except T.HarnessError as err:
            out[name] = f"refused: {err.code}"
            continue
        except ArithmeticError:
            out[name] = "reference panics"
            continue
        out[name] = f"{e.digest():016x}"
        e.close()
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "session_digests.json"), "w") as f:
        json.dump(digests(), f, indent=1, sort_keys=True)
        f.write("\n")
