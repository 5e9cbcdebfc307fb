#!/usr/bin/env python
"""Regenerates tests/golden/oracle_digests.json: digests of what the ORACLE decides on a few deterministic synthetic snapshots.

The reference is Go and cannot be run here, so these are not reference outputs; they pin the oracle (the parity checker) against
accidental drift between rounds: the GPU parity tests compare the engine with the oracle, this file compares the oracle with
its own committed past.  Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [  # (name, config index, scale, actions)
    ("config2_x0.2", 2, 0.2, ["allocate", "backfill"]),
    ("config3_x0.02", 3, 0.02, ["allocate", "backfill"]),
    ("config4_r16_x0.01", 4, 0.01, ["allocate", "backfill"]),
    ("config3_x0.02_preempt_reclaim", 3, 0.02, ["allocate", "backfill", "preempt", "reclaim"]),
]


def digest(kbm, oracle, idx, scale, actions):
    import numpy as np
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(idx, scale))
    o = oracle.Oracle(kbm.conf.load_scheduler_conf(), snap)
    o.run(actions)
    dec = np.ascontiguousarray(o.decisions(), dtype=np.uint32)
    st, nd = o.task_state()
    h = hashlib.sha256()
    for a in (dec, np.ascontiguousarray(o.binds(), np.uint32), np.ascontiguousarray(st, np.uint8), np.ascontiguousarray(nd, np.uint32),
              np.ascontiguousarray(o.evictions(), np.uint32)):
        h.update(a.tobytes())
    return {"tasks": int(snap.n_tasks), "nodes": int(snap.n_nodes), "decisions": int(dec.shape[0]),
            "binds": int((o.binds() != kbm.abi.KB_NONE).sum()), "evictions": int(len(o.evictions())), "sha256": h.hexdigest()}


def main():
    kbm = importlib.import_module("kube-batch_amd")
    import oracle
    oracle.build()
    out = {name: digest(kbm, oracle, idx, scale, actions) for name, idx, scale, actions in CASES}
    with open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
