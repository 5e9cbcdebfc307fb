#!/usr/bin/env bash
# round 3, call 29: the candidate launches' time stamps through pinned memory (matrix_ms = the launch on the second stream, argmax_ms = repair + wait):
# parity module, the default bench line as the driver runs it
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call29
mkdir -p "$out"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_regressions.py -x -q > "$out/pytest_subset.txt" 2>&1; echo "gpu subset rc=$? $(tail -1 $out/pytest_subset.txt)" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'], d.get('verified_bind_set_equals_oracle'), d['scaling'], d['roofline_cycle']['frac'], d['roofline']['frac'])"; }
timeout 300 python bench.py 2> "$out/bench_default.err" | tee "$out/bench_default_run.json" | line "default run" | tee -a "$out/summary.txt"
