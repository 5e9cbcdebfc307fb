#!/usr/bin/env bash
# round 3, call 14: batch 32 by default, look-ahead only for slots that can still win: whole suite + the round's bench lines + phase trace
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call14
mkdir -p "$out"
python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config3.json" | line "c3" | tee -a "$out/summary.txt"
python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_survey_nodes.json" | line "survey" | tee -a "$out/summary.txt"
python bench.py --diverse --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_diverse.json" | line "diverse" | tee -a "$out/summary.txt"
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config4.json" | line "c4" | tee -a "$out/summary.txt"
python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config5.json" | line "c5" | tee -a "$out/summary.txt"
python bench.py --config 5 --preempt --steps 2 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config5_three_actions.json" | line "c5 three actions" | tee -a "$out/summary.txt"
bash scripts/gpu_trace_k7.sh r3_call14_trace 3 2>&1 | grep -v "K5 trace" | tee "$out/phase_trace_c3.txt"
