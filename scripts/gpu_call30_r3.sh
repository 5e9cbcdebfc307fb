#!/usr/bin/env bash
# round 3, call 30: the second stream at the device's least priority (KB_STREAM_PRIO=0: default priority), same box, alternating
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call30
mkdir -p "$out"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for rep in 1 2; do
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | line "c3 least priority" | tee -a "$out/summary.txt"
KB_STREAM_PRIO=0 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "c3 default priority" | tee -a "$out/summary.txt"
done
timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --verify 2>/dev/null | line "c5 least priority" | tee -a "$out/summary.txt"
KB_STREAM_PRIO=0 timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line "c5 default priority" | tee -a "$out/summary.txt"
timeout 120 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "c4 least priority" | tee -a "$out/summary.txt"
KB_STREAM_PRIO=0 timeout 120 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "c4 default priority" | tee -a "$out/summary.txt"
