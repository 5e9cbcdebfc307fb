#!/usr/bin/env bash
# round 3, call 16: run kernel with look-ahead tables (per-slot keys after 1..3 more placements, built during the evaluation phase) and no
# closing evaluation on a run's last row: whole suite, then A/B (KB_K9_LOOKAHEAD=0) on config 4 / survey nodes / config 3, pinned kernels
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call16
mkdir -p "$out"
python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 $out/pytest_gpu.txt)" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'], d['spec_breaks_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
KB_K5_STATS=1 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2> "$out/c4_stats.err" | tee "$out/bench_config4.json" | line "c4" | tee -a "$out/summary.txt"
grep "kb K5" "$out/c4_stats.err" | tee -a "$out/summary.txt"
KB_K9_LOOKAHEAD=0 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "c4 no look-ahead" | tee -a "$out/summary.txt"
KB_K5_STATS=1 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2> "$out/survey_stats.err" | tee "$out/bench_survey_nodes.json" | line "survey" | tee -a "$out/summary.txt"
grep "kb K5" "$out/survey_stats.err" | tee -a "$out/summary.txt"
KB_K9_LOOKAHEAD=0 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "survey no look-ahead" | tee -a "$out/summary.txt"
KB_COMMIT_KERNEL=run python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "survey pinned run kernel" | tee -a "$out/summary.txt"
KB_COMMIT_KERNEL=run KB_K9_LOOKAHEAD=0 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "survey pinned run kernel no look-ahead" | tee -a "$out/summary.txt"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config3.json" | line "c3" | tee -a "$out/summary.txt"
KB_COMMIT_KERNEL=run python bench.py --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "c3 pinned run kernel" | tee -a "$out/summary.txt"
KB_COMMIT_KERNEL=run KB_K9_LOOKAHEAD=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "c3 pinned run kernel no look-ahead" | tee -a "$out/summary.txt"
python bench.py --diverse --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_diverse.json" | line "diverse" | tee -a "$out/summary.txt"
python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config5.json" | line "c5" | tee -a "$out/summary.txt"
