#!/bin/bash
# per-phase clocks of the batch commit kernel (trace build) on config 3
#   make -C kube-batch_amd/csrc EXTRA=-DKB_K7_TRACE OUT=../libkbengine_trace.so ; gpurun --timeout 300 -- 'bash scripts/gpu_trace_k7.sh <tag>'
set -u
TAG=${1:-t}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
for c in ${2:-3}; do
KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_K5_STATS=1 timeout 200 python bench.py --config $c --steps 3 --warmup 1 --verify --no-cpu-baseline > "$OUT/trace_c$c.json" 2> "$OUT/trace_c$c.err"; echo "rc=$?"
grep "kb K" "$OUT/trace_c$c.err"
python - "$OUT/trace_c$c.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 2), "verified", d.get("verified_bind_set_equals_oracle"), d["kernel_ms_per_step"])
PY
done
