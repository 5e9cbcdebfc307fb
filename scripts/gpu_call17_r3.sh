#!/usr/bin/env bash
# round 3, call 17: per-phase clocks of the run kernel (trace build) on config 4 and survey nodes, look-ahead tables on / off
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call17
mkdir -p "$out"
for la in 1 0; do
  for cfg in "--config 4" "--survey-nodes"; do
    echo "== $cfg look-ahead=$la" | tee -a "$out/summary.txt"
    KB_K9_LOOKAHEAD=$la KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_K5_STATS=1 python bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline > "$out/t.json" 2> "$out/t.err"
    grep "kb K5\|K5 trace" "$out/t.err" | tee -a "$out/summary.txt"
    python -c "import json; d=json.loads(open('$out/t.json').read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],2), d['kernel_ms_per_step'])" | tee -a "$out/summary.txt"
  done
done
