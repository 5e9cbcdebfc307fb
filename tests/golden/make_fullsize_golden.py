#!/usr/bin/env python
"""Regenerates tests/golden/fullsize_digests.json: what the ORACLE decides on BASELINE configs 3 and 4 at their full size
(100k tasks x 10k nodes; config 4 = 16 resource dimensions under the bin-packing weights BASELINE names: mostrequested 5,
leastrequested 0, balancedresource 1).  tests/test_gpu_fullsize.py compares the engine with the live oracle AND with these
digests, so full-size exactness is part of the driver's -m gpu record and an oracle that drifts between rounds is caught too.
Not reference outputs (the reference is Go and cannot run here).  Takes about five minutes:  python tests/golden/make_fullsize_golden.py
"""
import hashlib
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

BINPACK_CONF = """
actions: "allocate, backfill"
tiers:
- plugins:
  - name: priority
  - name: gang
- plugins:
  - name: drf
  - name: predicates
  - name: proportion
  - name: nodeorder
    arguments:
      leastrequested.weight: 0
      mostrequested.weight: 5
      balancedresource.weight: 1
"""

# BASELINE configs[4] as it is stated: allocate + backfill + preempt (default tiers plus conformance, bench.py --preempt)
PREEMPT_CONF = """
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

CASES = {  # name -> (config index, scheduler conf text or None for the default)
    "config3_full": (3, None),
    "config4_binpack_full": (4, BINPACK_CONF),
    "config5_full": (5, None),     # 1M tasks x 50k nodes, allocate + backfill
    "config5_full_preempt": (5, PREEMPT_CONF),   # 1M x 50k, allocate + backfill + preempt: decisions, binds, Statement journal, evictions
    "config3_diverse_full": (3, None),           # round 6: configs[2] with every job drawing its OWN request (bench.py --diverse: ~9 600 distinct task shapes)
}
# config 5 takes the faithful loop several minutes (and its preempt hours): its digests come from the oracle's fast modes
# (kbo_set_fast), which tests/test_oracle_fast_cpu.py holds to the faithful modes on every smaller snapshot, on every preempt /
# reclaim case of the suite, and on the two full-size digests above
FAST = {"config5_full", "config5_full_preempt", "config3_diverse_full"}
# held to the committed digest only on the GPU box (bench.py's `variants.diverse`, tests/test_gpu_fullsize.py): the oracle's incremental mode gains little from
# its per-shape cache when every job is a shape of its own and needs minutes for this one; the scaled-down diverse sessions of the suite are live-oracle cases
DIGEST_ONLY = {"config3_diverse_full"}


def case_actions(name):
    return ["allocate", "backfill", "preempt"] if name.endswith("_preempt") else ["allocate", "backfill"]


def case_inputs(kbm, name):
    idx, conf_text = CASES[name]
    conf = kbm.conf.load_scheduler_conf(conf_text) if conf_text else kbm.conf.load_scheduler_conf()
    params = kbm.snapshot.synth_config(idx, 1.0)
    if "_diverse" in name:
        params.diverse_requests = True
    snap = kbm.snapshot.synth(params)
    return conf, snap


def digest_of(np, decisions, binds, journal=None, evictions=None):
    """sha256 over the ordered decision list and the bind set; for the cases with an evict action also over the Statement
    journal (op, task, node, stmt rows) and the evictions in cache.Evict order"""
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(decisions, dtype=np.uint32).tobytes())
    h.update(np.ascontiguousarray(binds, dtype=np.uint32).tobytes())
    if journal is not None:
        h.update(np.ascontiguousarray(journal, dtype=np.uint32).tobytes())
        h.update(np.ascontiguousarray(evictions, dtype=np.uint32).tobytes())
    return h.hexdigest()


def main():
    import numpy as np
    kbm = importlib.import_module("kube-batch_amd")
    import oracle
    oracle.build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize_digests.json")
    only = sys.argv[1:]          # python tests/golden/make_fullsize_golden.py [case ...]: these cases only, merged into the committed file
    out = json.load(open(path)) if only else {}
    for name in (only or CASES):
        conf, snap = case_inputs(kbm, name)
        o = oracle.Oracle(conf, snap, threads=min(16, os.cpu_count() or 1))
        if name in FAST:
            o.set_fast(True)
        o.run(case_actions(name))
        evict = name.endswith("_preempt")
        out[name] = {"tasks": int(snap.n_tasks), "nodes": int(snap.n_nodes), "n_res": int(snap.n_res),
                     "decisions": int(o.decisions().shape[0]), "binds": int((o.binds() != kbm.abi.KB_NONE).sum()),
                     "evals": int(o.evals),
                     "sha256": digest_of(np, o.decisions(), o.binds(), o.journal() if evict else None, o.evictions() if evict else None),
                     "oracle_mode": "fast" if name in FAST else "faithful"}
        if evict:
            j = o.journal()
            out[name].update({"journal": int(len(j)), "evictions": int(len(o.evictions())), "popped": int(o.popped),
                              "journal_ops": {k: int((j[:, 0] == v).sum()) for k, v in (("evict", 0), ("pipeline", 1), ("commit", 2), ("discard", 3))}})
        print(name, out[name])
        o.close()
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
