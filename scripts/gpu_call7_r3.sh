#!/usr/bin/env bash
set -uo pipefail
cd "$(dirname "$0")/.."
for pw in 1 0; do
  echo "== KB_K7_PREWALK=$pw"
  KB_K7_PREWALK=$pw bash scripts/gpu_trace_k7.sh r3_trace_pw$pw 3
done 2>&1 | tee gpurun_out/r3_trace_summary.txt
