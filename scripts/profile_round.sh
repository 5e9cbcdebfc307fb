#!/bin/bash
# Collects the rocprofv3 evidence for the bench command on the GPU box (run through gpurun):
#   gpurun --timeout 900 -- 'bash scripts/profile_round.sh r1'
# Writes gpurun_out/prof_<tag>/... ; copy the *_stats.csv / *_counter_collection.csv summaries into profiles/.
set -u
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
# pass 1: kernel trace + stats (per-kernel time)
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o bench -- $CMD > "$OUT/bench_trace.log" 2>&1
# pass 2/3: HBM traffic counters, each in its own run, without any API trace domain
rocprofv3 --pmc FETCH_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$OUT/pmc_fetch" -o bench -- $CMD > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$OUT/pmc_write" -o bench -- $CMD > "$OUT/bench_pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
tail -2 "$OUT/bench_trace.log"
