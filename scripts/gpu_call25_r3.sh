#!/usr/bin/env bash
# round 3, call 25: k_repair with the rank loops eight keys per step; wait + kernel trace, C3 / C5, parity module
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call25
mkdir -p "$out"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > "$out/pytest_subset.txt" 2>&1; echo "gpu subset rc=$? $(tail -1 $out/pytest_subset.txt)" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'])"; }
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3" | tee -a "$out/summary.txt"
timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line "c5" | tee -a "$out/summary.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d "$out/trace" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_under_trace.log" 2>&1
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/rocprofv3_kernel_stats.csv" \;
find "$out/trace" -name "*kernel_trace.csv" -exec cp {} "$out/kernel_trace.csv" \;
rm -rf "$out/trace"
python - "$out/kernel_trace.csv" <<'PY' | tee -a "$out/summary.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# overlap: for every k_repair, was the preceding k_argmax on the other queue finished before the repair started?  and what ran concurrently with commits
def nm(r): return r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
last_end = {}
import collections
gap = collections.defaultdict(list)
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = nm(r)
    if n == "k_repair":
        gap["repair_dur_us"].append((e - s) / 1e3)
        if "k_argmax" in last_end: gap["argmax_end_to_repair_start_us"].append((s - last_end["k_argmax"]) / 1e3)
        if "k_commit_batch" in last_end: gap["commit_end_to_repair_start_us"].append((s - last_end["k_commit_batch"]) / 1e3)
    if n == "k_commit_batch" and "k_repair" in last_end: gap["repair_end_to_commit_start_us"].append((s - last_end["k_repair"]) / 1e3)
    if n == "k_matrix" and "k_commit_batch" in last_end: gap["commit_END_to_matrix_start_us (negative = the matrix launch started while the commit was running)"].append((s - last_end["k_commit_batch"]) / 1e3)
    if n == "k_argmax": gap["argmax_dur_us"].append((e - s) / 1e3)
    if n == "k_matrix": gap["matrix_dur_us"].append((e - s) / 1e3)
    last_end[n] = e
for k, v in gap.items():
    v.sort()
    print(f"{k}: n={len(v)} median={v[len(v)//2]:.2f} mean={sum(v)/len(v):.2f} p10={v[len(v)//10]:.2f} p90={v[9*len(v)//10]:.2f}")
PY
