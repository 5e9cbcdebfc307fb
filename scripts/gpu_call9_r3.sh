#!/usr/bin/env bash
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call9
mkdir -p "$out"
for v in "--config 4" "--survey-nodes" "--config 5"; do
  tag=$(echo $v | tr -d ' -')
  KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_K5_STATS=1 timeout 300 python bench.py $v --steps 3 --warmup 1 --no-cpu-baseline > "$out/trace_$tag.json" 2> "$out/trace_$tag.err"
  echo "== $v" | tee -a "$out/summary.txt"; grep "kb K" "$out/trace_$tag.err" | tee -a "$out/summary.txt"
done
