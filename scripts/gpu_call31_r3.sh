#!/usr/bin/env bash
# round 3, call 31: the whole -m gpu suite on the round's final tree
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call31
mkdir -p "$out"
timeout 200 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 $out/pytest_gpu.txt)" | tee -a "$out/summary.txt"
