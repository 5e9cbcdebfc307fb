#!/usr/bin/env bash
# round 3, second GPU call: the suite with the promoted regression cases, the pinned skip set and the full-size preempt digest;
# BASELINE configs[4] verified at full size with all three actions; a rocprofv3 kernel trace of the three actions (the evict path's profile)
set -uo pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/r3_call2
mkdir -p "$out"
python -m pytest tests -x -q -m gpu -rs > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
python bench.py --config 5 --preempt --steps 2 --warmup 1 --no-cpu-baseline --verify > "$out/bench_config5_three_actions_verified.json" 2> "$out/bench_config5.err"; echo "bench config 5 rc=$?" | tee -a "$out/summary.txt"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/prof_preempt" -o preempt -- python scripts/time_preempt.py 5 1.0 > "$out/time_preempt_profiled.txt" 2>&1; echo "rocprof rc=$?" | tee -a "$out/summary.txt"
find "$out/prof_preempt" -name "*kernel_stats*" -exec cp {} "$out/preempt_kernel_stats.csv" \;
rm -rf "$out/prof_preempt"
for v in "--config 4" "--diverse" "--survey-nodes"; do
  python bench.py $v --steps 5 --warmup 2 --no-cpu-baseline --verify > "$out/bench_$(echo $v | tr -d ' -').json" 2>> "$out/bench_variants.err"
done
echo done | tee -a "$out/summary.txt"
