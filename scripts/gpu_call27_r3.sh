#!/usr/bin/env bash
# round 3, call 27: overlapped lists also for sessions with scalar dimensions when the predecessor ran on the run kernel (config 4): whole suite, A/B
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call27
mkdir -p "$out"
timeout 600 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 $out/pytest_gpu.txt)" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for rep in 1 2; do
KB_K5_STATS=1 timeout 120 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2> "$out/c4.err" | tee "$out/bench_config4.json" | line "c4 new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "c4 prev" | tee -a "$out/summary.txt"
done
grep "kb overlap" "$out/c4.err" | tee -a "$out/summary.txt"
