#!/usr/bin/env bash
# round 3, call 28: the round's closing lines on the final code: smoke(), the default bench run exactly as the driver launches it (cpu baseline,
# variants, rooflines), config 5 with its three actions, config 2
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call28
mkdir -p "$out"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'], d.get('verified_bind_set_equals_oracle'), d.get('verified_evictions_equal_oracle'))"; }
timeout 300 python bench.py 2> "$out/bench_default.err" | tee "$out/bench_default_run.json" | line "default run" | tee -a "$out/summary.txt"
timeout 300 python bench.py --config 5 --preempt --steps 2 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config5_three_actions.json" | line "c5 three actions" | tee -a "$out/summary.txt"
timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config5.json" | line "c5" | tee -a "$out/summary.txt"
timeout 120 python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config2.json" | line "c2" | tee -a "$out/summary.txt"
timeout 120 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_survey_nodes.json" | line "survey" | tee -a "$out/summary.txt"
timeout 120 python bench.py --diverse --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_diverse.json" | line "diverse" | tee -a "$out/summary.txt"
