#!/usr/bin/env bash
# round 3, call 13: batch-size sweep, then the rocprofv3 evidence of the round's code (kernel trace + stats; FETCH / WRITE PMC passes of the
# materialised-matrix launches) and the per-phase trace of the batch kernel
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call13
mkdir -p "$out"
export TMPDIR=/tmp
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for b in 8 12 16 24 32; do
  KB_K5_BATCH=$b python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | line "batch=$b c3" | tee -a "$out/summary.txt"
  KB_K5_BATCH=$b python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "batch=$b survey" | tee -a "$out/summary.txt"
done
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -f csv -d "$out/trace" -o bench -- $CMD > "$out/bench_under_trace.log" 2>&1
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/rocprofv3_kernel_stats.csv" \;
rm -rf "$out/trace"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -f csv --kernel-include-regex "k_matrix|k_expand" -d "$out/pmc_$c" -o bench -- $CMD > "$out/bench_pmc_$c.log" 2>&1
  find "$out/pmc_$c" -name "*counter_collection.csv" -exec cp {} "$out/pmc_default_$c.csv" \;
  rm -rf "$out/pmc_$c"
  rocprofv3 --pmc $c -f csv --kernel-include-regex "k_matrix|k_expand" -d "$out/pmcd_$c" -o bench -- python scripts/bench_k1.py 3 diverse > "$out/k1_diverse_pmc_$c.log" 2>&1
  find "$out/pmcd_$c" -name "*counter_collection.csv" -exec cp {} "$out/pmc_diverse_$c.csv" \;
  rm -rf "$out/pmcd_$c"
  rocprofv3 --pmc $c -f csv --kernel-include-regex "k_matrix|k_expand" -d "$out/pmc4_$c" -o bench -- python scripts/bench_k1.py 4 > "$out/k1_c4_pmc_$c.log" 2>&1
  find "$out/pmc4_$c" -name "*counter_collection.csv" -exec cp {} "$out/pmc_config4_$c.csv" \;
  rm -rf "$out/pmc4_$c"
done
bash scripts/gpu_trace_k7.sh r3_call13_trace 3 2>&1 | tee "$out/phase_trace_c3.txt"
ls -la "$out"
