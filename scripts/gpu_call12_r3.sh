#!/usr/bin/env bash
# round 3, call 12: helper workgroups warm the commit workgroup's L2 (kb_warm.hpp): whole suite, then A/B
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call12
mkdir -p "$out"
python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for rep in 1 2; do
  for f in 0 1; do
    KB_WARM_HELPERS_OFF=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | line "helpers_off=$f c3" | tee -a "$out/summary.txt"
    KB_WARM_HELPERS_OFF=$f python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "helpers_off=$f c4" | tee -a "$out/summary.txt"
  done
done
for f in 0 1; do KB_WARM_HELPERS_OFF=$f python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | line "helpers_off=$f c5" | tee -a "$out/summary.txt"; done
for f in 0 1; do KB_WARM_HELPERS_OFF=$f python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "helpers_off=$f survey" | tee -a "$out/summary.txt"; done
