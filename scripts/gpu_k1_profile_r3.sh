#!/usr/bin/env bash
# round 3: where does the matrix kernel's time go?  kernel trace (durations per kernel) and SQ instruction counters of the materialised-matrix
# launch on config 3 (diverse shapes: direct tile) and config 4 (R = 16: per-shape tile + expansion), each in its own rocprofv3 run
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_k1prof
mkdir -p "$out"
export TMPDIR=/tmp
for cfg in "3 diverse" "4" "3"; do
  tag=$(echo $cfg | tr ' ' '_')
  python scripts/bench_k1.py $cfg > "$out/k1_$tag.txt" 2>&1
  KB_K1_DIRECT=1 python scripts/bench_k1.py $cfg >> "$out/k1_$tag.txt" 2>&1
  rocprofv3 --kernel-trace --stats -f csv -d "$out/trace_$tag" -o k1 -- python scripts/bench_k1.py $cfg > "$out/trace_$tag.log" 2>&1
  cp "$out/trace_$tag"/*/k1_kernel_stats.csv "$out/kernel_stats_$tag.csv" 2>/dev/null || find "$out/trace_$tag" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_$tag.csv" \;
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv --kernel-include-regex "k_matrix|k_expand" -d "$out/pmc_$tag" -o k1 -- python scripts/bench_k1.py $cfg > "$out/pmc_$tag.log" 2>&1
  find "$out/pmc_$tag" -name "*counter_collection.csv" -exec cp {} "$out/pmc_$tag.csv" \;
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$out/pmc2_$tag" -o k1 -- python scripts/bench_k1.py $cfg > "$out/pmc2_$tag.log" 2>&1
  find "$out/pmc2_$tag" -name "*counter_collection.csv" -exec cp {} "$out/pmc2_$tag.csv" \;
  rm -rf "$out/trace_$tag" "$out/pmc_$tag" "$out/pmc2_$tag"
done
ls -la "$out"
