#!/bin/bash
# One GPU call: the -m gpu suite, then the two 100k x 10k configurations, verified against the oracle.
#   gpurun --timeout 900 -- 'bash scripts/gpu_check.sh <tag> [pytest|bench|all]'
set -u
TAG=${1:-x}
WHAT=${2:-all}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
if [ "$WHAT" != "bench" ]; then
  timeout 700 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest -m gpu rc=$?"
  tail -n 8 "$OUT/pytest_gpu.log"
fi
if [ "$WHAT" != "pytest" ]; then
  for c in 3 4; do
    KB_K5_STATS=1 timeout 300 python bench.py --config $c --steps 3 --warmup 1 --verify --no-cpu-baseline > "$OUT/bench_c$c.json" 2> "$OUT/bench_c$c.err"; echo "bench c$c rc=$?"
    grep "kb K5" "$OUT/bench_c$c.err"
  done
  KB_K5_STATS=1 timeout 300 python bench.py --config 3 --diverse --steps 3 --warmup 1 --verify --no-cpu-baseline > "$OUT/bench_c3_diverse.json" 2> "$OUT/bench_c3_diverse.err"; echo "bench c3 diverse rc=$?"
  python - "$OUT" <<'PY'
import json, sys, os
for f in ("bench_c3.json", "bench_c4.json", "bench_c3_diverse.json"):
    p = os.path.join(sys.argv[1], f)
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(f, "ms/step", round(d["ms_per_step"], 2), "evals/s %.3g" % d["value"], "binds", d["binds"], "rounds", d["rounds_per_step"],
              "dirty-won rows", d["row_fallbacks_per_step"], "verified", d.get("verified_bind_set_equals_oracle"), d["kernel_ms_per_step"],
              "K1 roofline frac", d["roofline"]["frac"], "avg ms", d["roofline"]["avg_launch_ms"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
fi
