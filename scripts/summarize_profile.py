#!/usr/bin/env python
"""gpurun_out/prof_<tag>/ (written by `scripts/gpu_r5.sh profile` on the GPU box) -> the summaries committed under profiles/.

  rocprofv3_kernel_stats.csv   the --kernel-trace --stats table as rocprofv3 wrote it
  rocprofv3_pmc_k_matrix.csv   per kernel: dispatches, mean / max KB per dispatch of FETCH_SIZE and WRITE_SIZE (separate passes)
"""
import csv
import shutil
import sys
from collections import defaultdict
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
dst = Path(sys.argv[2] if len(sys.argv) > 2 else "profiles/round1")
src = Path("gpurun_out") / f"prof_{tag}"
if not src.exists():
    src = Path("gpurun_out") / tag            # scripts/gpu_r5.sh profile writes gpurun_out/r5_profile
dst.mkdir(parents=True, exist_ok=True)
shutil.copy(src / "trace" / "bench_kernel_stats.csv", dst / "rocprofv3_kernel_stats.csv")
rows = []
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = defaultdict(list)
    with open(src / sub / "bench_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        rows.append((counter, k, len(v), sum(v) / len(v), max(v)))
with open(dst / "rocprofv3_pmc_k_matrix.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "dispatches", "mean_KB_per_dispatch", "max_KB_per_dispatch"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], f"{r[3]:.3f}", f"{r[4]:.3f}"])
# the commit kernels' SQ / GRBM counters (scripts/gpu_r5.sh profile: two passes), per kernel: dispatches and the mean per dispatch
crow = []
for sub in ("pmc_commit_a", "pmc_commit_b"):
    fn = src / sub / "bench_counter_collection.csv"
    if not fn.exists():
        continue
    acc = defaultdict(list)
    with open(fn) as f:
        for r in csv.DictReader(f):
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, cname), v in sorted(acc.items()):
        crow.append((k, cname, len(v), sum(v) / len(v), max(v)))
if crow:
    with open(dst / "rocprofv3_pmc_k_commit.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "max_per_dispatch"])
        for r in crow:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{r[4]:.1f}"])
# the device sources the numbers were measured on (written on the GPU box by scripts/gpu_r5.sh): bench.py quotes the CSVs only for these
if (src / "kernel_sources.sha256").exists():
    shutil.copy(src / "kernel_sources.sha256", dst / "kernel_sources.sha256")
# the bench line printed under the kernel trace
log = src / "bench_trace.log"
for line in log.read_text().splitlines():
    if line.startswith("{") and '"metric"' in line:
        (dst / "bench_under_rocprofv3_kernel_trace.json").write_text(line + "\n")
print("wrote", sorted(p.name for p in dst.iterdir()))
