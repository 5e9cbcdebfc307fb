#!/usr/bin/env python
"""Regenerates tests/golden/bench_rank_digests.json: what the ORACLE decides on the snapshots `bench.py --gpus N` hands its ranks (rank k:
the generator's seed + RANK_SEED_STRIDE * k; rank 0 is the N = 1 workload), as the 63-bit digest kube-batch_amd/dist.py's
ReplicatedCycle.digest takes of the ordered decision list and the bind set.  bench.py holds every rank's cycle to it after the timed
region (`sessions_verified_against_golden_digests`).  From the oracle's fast mode, which tests/test_oracle_fast_cpu.py holds to the
faithful loop.  About a minute:   python tests/golden/make_bench_rank_digests.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
RANKS = 8


def main():
    kbm = importlib.import_module("kube-batch_amd")
    distmod = importlib.import_module("kube-batch_amd.dist")
    import bench
    import oracle
    oracle.build()
    out = {"note": "see make_bench_rank_digests.py; key = bench.py's configuration, value = {rank: digest}"}
    for config, scale in ((3, 1.0), (3, 0.02)):     # the default bench configuration, and the scale the CPU suite's two-rank case uses
        conf = kbm.conf.load_scheduler_conf()
        per_rank = {}
        for rank in range(RANKS):
            params = kbm.snapshot.synth_config(config, scale)
            params.seed = params.seed + bench.RANK_SEED_STRIDE * rank
            snap = kbm.snapshot.synth(params)
            o = oracle.Oracle(conf, snap)
            o.set_fast(True)
            o.run(["allocate", "backfill"])
            per_rank[str(rank)] = distmod.ReplicatedCycle.digest(o.decisions(), o.binds())
            print(config, scale, rank, per_rank[str(rank)], flush=True)
            o.close()
        out[f"config{config}_scale{scale:g}"] = per_rank
    with open(os.path.join(ROOT, "tests", "golden", "bench_rank_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
