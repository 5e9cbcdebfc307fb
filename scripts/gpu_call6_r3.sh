#!/usr/bin/env bash
# round 3, call 6: the batch commit kernel's pre-walk (wave 0 walks batch b + 1 during fetch / evaluate of batch b), k_expand in shape order:
# whole -m gpu suite, then A/B on one box
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call6
mkdir -p "$out"
python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
for rep in 1 2; do
  for pw in 1 0; do
    KB_K7_PREWALK=$pw python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prewalk=$pw c3', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['verified_bind_set_equals_oracle'])" | tee -a "$out/summary.txt"
    KB_K7_PREWALK=$pw python bench.py --diverse --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prewalk=$pw diverse', round(d['ms_per_step'],2), d['kernel_ms_per_step'])" | tee -a "$out/summary.txt"
  done
done
for cfg in "4" "5" "3"; do
  for rep in 1 2; do
    python scripts/bench_k1.py $cfg 2>&1 | grep config | tee -a "$out/summary.txt"
    KB_EXPAND_ORDER=0 python scripts/bench_k1.py $cfg 2>&1 | grep config | sed 's/^/task-order expand: /' | tee -a "$out/summary.txt"
  done
done
python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_config5.json" 2>/dev/null
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify > "$out/bench_config4.json" 2>/dev/null
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
