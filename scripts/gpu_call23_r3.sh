#!/usr/bin/env bash
# round 3, call 23: k_repair restructured for latency (independent loads issued before the first wait, the row's task record published by the
# arg-max launch): parity + full-size modules, same-box A/B against the library WITHOUT overlapped lists (libkbengine_prev.so)
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call23
mkdir -p "$out"
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q > "$out/pytest_subset.txt" 2>&1; echo "gpu subset rc=$? $(tail -1 $out/pytest_subset.txt)" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'], d['spec_breaks_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for rep in 1 2; do
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config3.json" | line "c3 new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3 prev" | tee -a "$out/summary.txt"
done
timeout 120 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "survey new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "survey prev" | tee -a "$out/summary.txt"
timeout 120 python bench.py --diverse --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "diverse new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --diverse --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "diverse prev" | tee -a "$out/summary.txt"
timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | line "c5 new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line "c5 prev" | tee -a "$out/summary.txt"
