#!/usr/bin/env bash
# same-box A/B of the working tree's library against kube-batch_amd/libkbengine_prev.so (built by the caller from the previous commit):
#   gpurun -- 'bash scripts/gpu_ab_quick.sh <tag> [reps]'
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/${1:-ab}
mkdir -p "$out"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for rep in $(seq 1 ${2:-2}); do
  python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "c4 new" | tee -a "$out/summary.txt"
  KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "c4 prev" | tee -a "$out/summary.txt"
  python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "survey new" | tee -a "$out/summary.txt"
  KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "survey prev" | tee -a "$out/summary.txt"
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | line "c3 new" | tee -a "$out/summary.txt"
  KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3 prev" | tee -a "$out/summary.txt"
done
KB_COMMIT_KERNEL=run python bench.py --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "c3 pinned run kernel new" | tee -a "$out/summary.txt"
KB_COMMIT_KERNEL=run KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_prev.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "c3 pinned run kernel prev" | tee -a "$out/summary.txt"
