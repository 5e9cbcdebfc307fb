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
    # which launch the commit kernel's counters are of (bench.py quotes it beside them): scripts/gpu_r6.sh profile runs these two passes with the
    # commit workgroup ALONE in its launch (KB_PROFILE_SOLO_COMMIT=1 KB_FUSE_REPAIR=0); round 5's passes summed ~30 workgroups per dispatch
    shape = src / "pmc_commit_launch_shape.txt"
    if shape.exists():
        shutil.copy(shape, dst / "rocprofv3_pmc_k_commit.launch_shape.txt")
    with open(dst / "rocprofv3_pmc_k_commit.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "max_per_dispatch"])
        for r in crow:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{r[4]:.1f}"])
# the device sources the numbers were measured on (written on the GPU box by scripts/gpu_r5.sh): bench.py quotes the CSVs only for these
if (src / "kernel_sources.sha256").exists():
    shutil.copy(src / "kernel_sources.sha256", dst / "kernel_sources.sha256")
# ... and per translation unit (scripts/kernel_sources_sha.py --tu): a call that measured the matrix launches only (scripts/gpu_r5.sh pmc_matrix)
# replaces kb_kernels.hip's line and leaves the commit kernels' as the call that measured THEM wrote it
if (src / "kernel_tu.sha256").exists():
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from kernel_sources_sha import read_tu_stamp
    fresh = read_tu_stamp(src / "kernel_tu.sha256")
    kept = read_tu_stamp(dst / "kernel_tu.sha256") if (dst / "kernel_tu.sha256").exists() else {}
    merged = dict(kept)
    merged.update(fresh if crow else {k: v for k, v in fresh.items() if k == "kb_kernels.hip"})
    with open(dst / "kernel_tu.sha256", "w") as f:
        f.write("# sha256 per translation unit (scripts/kernel_sources_sha.py --tu) of the sources the summaries of this directory were measured on:\n"
                "# rocprofv3_pmc_k_matrix.csv <-> kb_kernels.hip, rocprofv3_pmc_k_commit.csv <-> kb_commit_sel.hip (bench.py: stale()).\n"
                f"# last written from gpurun_out/{src.name} ({'all passes' if crow else 'matrix passes + kernel trace only: the other lines kept'})\n")
        for k in sorted(merged):
            f.write(f"{merged[k]}  {k}\n")
# the bench line printed under the kernel trace
log = src / "bench_trace.log"
for line in log.read_text().splitlines():
    if line.startswith("{") and '"metric"' in line:
        (dst / "bench_under_rocprofv3_kernel_trace.json").write_text(line + "\n")
print("wrote", sorted(p.name for p in dst.iterdir()))
