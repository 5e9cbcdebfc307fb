#!/usr/bin/env python
"""What sits between two commit kernels: from a rocprofv3 --kernel-trace CSV (gpurun_out/r5_profile/trace/bench_kernel_trace.csv, written by
`scripts/gpu_r5.sh profile` / `evict`), the commit kernels' busy time, the gaps between consecutive ones and which kernels ran inside them.
python scripts/trace_gaps.py <bench_kernel_trace.csv> [sessions]      (sessions: bench steps + warm-up in the traced command, to print per-session figures)"""
import collections
import csv
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    sessions = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ci = [i for i, r in enumerate(rows) if "k_commit" in r["Kernel_Name"]]
    busy = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in ci)
    span = int(rows[ci[-1]]["End_Timestamp"]) - int(rows[ci[0]]["Start_Timestamp"])
    kinds = collections.Counter(short(rows[i]["Kernel_Name"]) for i in ci)
    print(f"{len(ci)} commit launches {dict(kinds)}: busy {busy / 1e6:.2f} ms, first start to last end {span / 1e6:.2f} ms")
    inside = collections.Counter()
    hist = collections.Counter()
    n_hist = collections.Counter()
    for a, b in zip(ci, ci[1:]):
        ga, gb = int(rows[a]["End_Timestamp"]), int(rows[b]["Start_Timestamp"])
        g = gb - ga
        if g <= 0 or g > 20_000_000:      # > 20 ms: between two sessions / actions of the bench, not a gap of the cycle
            continue
        k = "< 20 us" if g < 20_000 else "< 50 us" if g < 50_000 else "< 200 us" if g < 200_000 else "< 1 ms" if g < 1_000_000 else ">= 1 ms"
        hist[k] += g
        n_hist[k] += 1
        for r in rows[a + 1:b]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            o = min(e, gb) - max(s, ga)
            if o > 0:
                inside[short(r["Kernel_Name"])] += o
    tot = sum(hist.values())
    print(f"gaps between consecutive commit kernels: {tot / 1e6:.2f} ms in all = {tot / 1e6 / sessions:.2f} ms per session ({sessions:g} sessions)")
    for k in ("< 20 us", "< 50 us", "< 200 us", "< 1 ms", ">= 1 ms"):
        if n_hist[k]:
            print(f"  gaps {k:9s}: {n_hist[k]:5d}, {hist[k] / 1e6:7.2f} ms ({hist[k] / 1e6 / sessions:.2f} per session), mean {hist[k] / n_hist[k] / 1e3:.1f} us")
    print("kernel time inside the gaps (ms): " + ", ".join(f"{k} {v / 1e6:.2f}" for k, v in inside.most_common()))


if __name__ == "__main__":
    main()
