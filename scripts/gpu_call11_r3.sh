#!/usr/bin/env bash
# round 3, call 11: on-demand chain tables in the batch kernel's row mode; which commit kernel now wins where
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call11
mkdir -p "$out"
python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for rep in 1 2; do
  for f in 1 0; do
    KB_K7_CHAIN=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | line "chain=$f c3" | tee -a "$out/summary.txt"
    KB_K7_CHAIN=$f python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "chain=$f survey" | tee -a "$out/summary.txt"
  done
done
for f in 1 0; do KB_K7_CHAIN=$f python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | line "chain=$f c5" | tee -a "$out/summary.txt"; done
for ck in batch run; do
  KB_COMMIT_KERNEL=$ck python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "pinned $ck c4" | tee -a "$out/summary.txt"
  KB_COMMIT_KERNEL=$ck python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "pinned $ck survey" | tee -a "$out/summary.txt"
  KB_COMMIT_KERNEL=$ck python bench.py --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "pinned $ck c3" | tee -a "$out/summary.txt"
done
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "auto c4" | tee -a "$out/summary.txt"
