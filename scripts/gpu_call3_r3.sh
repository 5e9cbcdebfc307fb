#!/usr/bin/env bash
# round 3, third GPU call: the per-pair arithmetic rewritten (three-operation exact division, branch-free scorers, uniform task rows
# in K1): the whole -m gpu suite, then the headline and the two evaluation-bound matrix launches (--diverse, --config 4)
set -uo pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/r3_call3
mkdir -p "$out"
python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
for v in "--config 4" "--diverse" "--survey-nodes"; do
  python bench.py $v --steps 5 --warmup 2 --no-cpu-baseline --verify > "$out/bench_$(echo $v | tr -d ' -').json" 2>> "$out/bench_variants.err"
done
echo done | tee -a "$out/summary.txt"
