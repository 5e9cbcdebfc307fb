#!/usr/bin/env bash
# round 3, call 22: overlapped candidate lists (matrix + arg-max of a chained round on a second stream beside the predecessor's commit kernel,
# k_repair behind it): whole suite, the parity module on the plain path (KB_OVERLAP=0), same-box A/B against the previous commit's library
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call22
mkdir -p "$out"
timeout 600 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 $out/pytest_gpu.txt)" | tee -a "$out/summary.txt"
KB_OVERLAP=0 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q > "$out/pytest_parity_plain.txt" 2>&1; echo "parity module, plain path rc=$? $(tail -1 $out/pytest_parity_plain.txt)" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'], d['spec_breaks_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
KB_K5_STATS=1 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2> "$out/c3.err" | tee "$out/bench_config3.json" | line "c3 new" | tee -a "$out/summary.txt"
grep "kb overlap\|kb host" "$out/c3.err" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3 prev" | tee -a "$out/summary.txt"
KB_OVERLAP=0 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3 new, plain path" | tee -a "$out/summary.txt"
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | line "c3 new" | tee -a "$out/summary.txt"
timeout 120 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_survey_nodes.json" | line "survey new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "survey prev" | tee -a "$out/summary.txt"
timeout 120 python bench.py --diverse --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_diverse.json" | line "diverse new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 120 python bench.py --diverse --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "diverse prev" | tee -a "$out/summary.txt"
timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config5.json" | line "c5 new" | tee -a "$out/summary.txt"
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line "c5 prev" | tee -a "$out/summary.txt"
timeout 120 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | tee "$out/bench_config4.json" | line "c4 new" | tee -a "$out/summary.txt"
