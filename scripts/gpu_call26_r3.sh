#!/usr/bin/env bash
# round 3, call 26: k_repair with the tag's round trip beside the decision records' (two dependent round trips instead of three): whole suite,
# same-box A/B against the library without overlapped lists, kernel trace of C3
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call26
mkdir -p "$out"
timeout 600 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 $out/pytest_gpu.txt)" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for rep in 1 2; do
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config3.json" | line "c3 new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3 prev" | tee -a "$out/summary.txt"
done
timeout 120 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "survey new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "survey prev" | tee -a "$out/summary.txt"
timeout 120 python bench.py --diverse --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "diverse new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --diverse --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "diverse prev" | tee -a "$out/summary.txt"
for rep in 1 2; do
timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | line "c5 new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line "c5 prev" | tee -a "$out/summary.txt"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d "$out/trace" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_under_trace.log" 2>&1
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/rocprofv3_kernel_stats.csv" \;
find "$out/trace" -name "*kernel_trace.csv" -exec cp {} "$out/kernel_trace.csv" \;
rm -rf "$out/trace"
python - "$out/kernel_trace.csv" <<'PY' | tee -a "$out/summary.txt"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
last_end, gap = {}, collections.defaultdict(list)
for r in rows:
    s, e, n = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r)
    if n == "k_repair":
        gap["repair_dur_us"].append((e - s) / 1e3)
        if "k_argmax" in last_end: gap["argmax_end_to_repair_start_us"].append((s - last_end["k_argmax"]) / 1e3)
    if n == "k_commit_batch": gap["commit_batch_dur_us"].append((e - s) / 1e3)
    if n == "k_argmax": gap["argmax_dur_us"].append((e - s) / 1e3)
    if n == "k_matrix": gap["matrix_dur_us"].append((e - s) / 1e3)
    last_end[n] = e
for k, v in gap.items():
    v.sort()
    print(f"{k}: n={len(v)} median={v[len(v)//2]:.2f} mean={sum(v)/len(v):.2f} p10={v[len(v)//10]:.2f} p90={v[9*len(v)//10]:.2f}")
PY
rm -f "$out/kernel_trace.csv"
