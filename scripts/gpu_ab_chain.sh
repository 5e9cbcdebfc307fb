#!/bin/bash
# A/B of the per-round launch path on configs 3 and 4, verified: chained rounds (KB_CHAIN_ROUNDS) and the window read straight from
# the pinned staging block (KB_DIRECT_WINDOW); parity + fuzz files first; a rocprofv3 kernel trace of config 3 last.
#   gpurun --timeout 600 -- 'bash scripts/gpu_ab_chain.sh <tag>'
set -u
TAG=${1:-ch}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -n 3 "$OUT/pytest.log"
for c in 3 4; do
  for v in 11 10 00; do
    KB_CHAIN_ROUNDS=${v:0:1} KB_DIRECT_WINDOW=${v:1:1} KB_K5_STATS=1 timeout 200 python bench.py --config $c --steps 5 --warmup 1 --verify --no-cpu-baseline > "$OUT/bench_c${c}_v$v.json" 2> "$OUT/bench_c${c}_v$v.err"; echo "bench c$c chain,direct=$v rc=$?"
  done
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
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o bench -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_trace.log" 2>&1
tail -1 "$OUT/bench_trace.log" | cut -c1-300
