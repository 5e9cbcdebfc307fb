#!/usr/bin/env bash
# round 3, call 15: the host-path changes (reset without a reduction, no reduction after an action without decisions, pinned result block) on the
# device: parity + regression suites, the host timeline (KB_K5_STATS), A/B against the old path, and a window sweep for the three regimes
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call15
mkdir -p "$out"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_regressions.py tests/test_framework_actions.py -x -q > "$out/pytest_subset.txt" 2>&1; echo "gpu subset rc=$? $(tail -1 $out/pytest_subset.txt)" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d['rounds_per_step'], d['spec_breaks_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
KB_K5_STATS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2> "$out/c3_stats.err" | tee "$out/bench_config3.json" | line "c3" | tee -a "$out/summary.txt"
grep "kb host\|kb probe\|kb K5" "$out/c3_stats.err" | tee -a "$out/summary.txt"
KB_RESET_REDUCE=1 KB_ALWAYS_REDUCE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3 old reductions" | tee -a "$out/summary.txt"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3 again" | tee -a "$out/summary.txt"
for w in 128 192; do
  python bench.py --window $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3 window=$w" | tee -a "$out/summary.txt"
  python bench.py --window $w --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "survey window=$w" | tee -a "$out/summary.txt"
  python bench.py --window $w --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "c4 window=$w" | tee -a "$out/summary.txt"
done
python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "survey window=256" | tee -a "$out/summary.txt"
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "c4 window=256" | tee -a "$out/summary.txt"
KB_K5_STATS=1 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2> "$out/c5_stats.err" | line "c5" | tee -a "$out/summary.txt"
grep "kb host" "$out/c5_stats.err" | tee -a "$out/summary.txt"
