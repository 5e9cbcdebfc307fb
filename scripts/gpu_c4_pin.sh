#!/bin/bash
# config 4 with the commit kernel pinned (b = batch, r = run) against the adaptive choice
set -u
OUT=$PWD/gpurun_out/${1:-pin}
mkdir -p "$OUT"
for k in b r a; do
  if [ $k = a ]; then unset KB_COMMIT_KERNEL; else export KB_COMMIT_KERNEL=$k; fi
  KB_K5_STATS=1 timeout 200 python bench.py --config 4 --steps 3 --warmup 1 --verify --no-cpu-baseline > "$OUT/c4_$k.json" 2> "$OUT/c4_$k.err"
  python - "$OUT/c4_$k.json" $k <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step", round(d["ms_per_step"], 2), "verified", d.get("verified_bind_set_equals_oracle"), d["kernel_ms_per_step"], "breaks", d["spec_breaks_per_step"])
PY
done
