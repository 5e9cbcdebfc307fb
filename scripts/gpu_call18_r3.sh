#!/usr/bin/env bash
# round 3, call 18: run kernel, where the rows phase goes: runs that touch scalar dimensions apart, dirty-winner entries counted (trace build)
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call18
mkdir -p "$out"
for cfg in "--config 4" "--config 3"; do
  echo "== $cfg pinned run kernel" | tee -a "$out/summary.txt"
  KB_COMMIT_KERNEL=run KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_K5_STATS=1 python bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline > "$out/t.json" 2> "$out/t.err"
  grep "kb K5\|K5 trace" "$out/t.err" | tee -a "$out/summary.txt"
  python -c "import json; d=json.loads(open('$out/t.json').read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],2), d['kernel_ms_per_step'])" | tee -a "$out/summary.txt"
done
