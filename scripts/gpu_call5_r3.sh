#!/usr/bin/env bash
# round 3, call 5: k_matrix_runs (runs of equal rows evaluated once, 16-byte cooperative stores) — parity, then A/B against the <4, 32> tile
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call5
mkdir -p "$out"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_interpod.py -x -q -m gpu > "$out/pytest_parity.txt" 2>&1; echo "parity rc=$?" | tee -a "$out/summary.txt"
for cfg in "3 diverse" "4" "3" "5"; do
  tag=$(echo $cfg | tr ' ' '_')
  for rep in 1 2; do
    python scripts/bench_k1.py $cfg >> "$out/k1_$tag.txt" 2>&1
    KB_K1_RUNS=0 KB_K1_DIRECT=1 python scripts/bench_k1.py $cfg >> "$out/k1_$tag.txt" 2>&1
    KB_K1_DIRECT=1 python scripts/bench_k1.py $cfg >> "$out/k1_$tag.txt" 2>&1
    KB_K1_DIRECT=0 python scripts/bench_k1.py $cfg >> "$out/k1_$tag.txt" 2>&1
  done
done
grep -h config "$out"/k1_*.txt | tee -a "$out/summary.txt"
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
