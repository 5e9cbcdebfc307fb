#!/usr/bin/env bash
# One short GPU call: the parity files one by one inside a fixed time budget (seconds, default 95), most recently touched areas first.
# Every file that finishes leaves its pytest summary in gpurun_out/val_<name>.txt.  Usage (on the GPU box): scripts/gpu_validate_subset.sh [budget]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
budget=${1:-95}
S=$SECONDS
for f in preempt parity interpod fuzz sharded adversarial fullsize; do
  left=$((budget - (SECONDS - S)))
  [ $left -le 3 ] && { echo "$f not run (budget)"; continue; }
  timeout $left python -m pytest tests/test_gpu_$f.py -x -q -p no:cacheprovider > gpurun_out/val_$f.txt 2>&1
  echo "$f rc=$? t=$((SECONDS - S)) $(tail -1 gpurun_out/val_$f.txt)"
done
