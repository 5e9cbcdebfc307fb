#!/usr/bin/env bash
# round 3, call 20: run kernel: per-slot LDS copy of the run's scalar dimensions (rows loop never waits for L2), candidates' scalars fetched with their state, fence behind the preparation.
# L2 round trip after the other; no closing evaluation on a run's last row.  Whole suite, same-box A/B against the previous commit's library
# (kube-batch_amd/libkbengine_prev.so, built from HEAD by the caller), phase trace of config 4
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call20
mkdir -p "$out"
python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 $out/pytest_gpu.txt)" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'], d['spec_breaks_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for rep in 1 2; do
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config4.json" | line "c4 new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "c4 previous commit" | tee -a "$out/summary.txt"
done
python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_survey_nodes.json" | line "survey new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "survey previous commit" | tee -a "$out/summary.txt"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config3.json" | line "c3 new" | tee -a "$out/summary.txt"
echo "== --config 4 trace" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_K5_STATS=1 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > "$out/t.json" 2> "$out/t.err"
grep "kb K5\|K5 trace" "$out/t.err" | tee -a "$out/summary.txt"
