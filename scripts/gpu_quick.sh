#!/bin/bash
# quick check of a kernel change: parity + fuzz files, then configs 3 and 4 verified
#   gpurun --timeout 600 -- 'bash scripts/gpu_quick.sh <tag>'
set -u
TAG=${1:-q}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_adversarial.py -x -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -n 3 "$OUT/pytest.log"
for c in 3 4; do
  KB_K5_STATS=1 timeout 200 python bench.py --config $c --steps 5 --warmup 1 --verify --no-cpu-baseline > "$OUT/bench_c${c}.json" 2> "$OUT/bench_c${c}.err"; echo "bench c$c rc=$?"
done
python - "$OUT" <<'PY'
import json, sys, os, glob
for p in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(os.path.basename(p), "ms/step", round(d["ms_per_step"], 2), "rounds", d["rounds_per_step"], "breaks", d["spec_breaks_per_step"],
              "verified", d.get("verified_bind_set_equals_oracle"), d["kernel_ms_per_step"])
    except Exception as e:
        print(p, "unreadable:", e)
PY
