#!/bin/bash
# bench configs 3 and 4 (verified) with the regular build, then the commit kernel's per-phase cycle trace (-DKB_K8_TRACE build)
set -u
TAG=${1:-x}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
for c in 3 4; do
  KB_K5_STATS=1 timeout 300 python bench.py --config $c --steps 3 --warmup 1 --verify --no-cpu-baseline > "$OUT/bench_c$c.json" 2> "$OUT/bench_c$c.err"; echo "bench c$c rc=$?"
  grep "kb K5" "$OUT/bench_c$c.err"
  if [ -f kube-batch_amd/libkbengine_trace.so ]; then
    KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_K5_STATS=1 timeout 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/trace_c$c.json" 2> "$OUT/trace_c$c.err"
    grep "kb K5 trace" "$OUT/trace_c$c.err"
  fi
done
python - "$OUT" <<'PY'
import json, sys, os
for f in ("bench_c3.json", "bench_c4.json"):
    p = os.path.join(sys.argv[1], f)
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(f, "ms/step", round(d["ms_per_step"], 2), "evals/s %.3g" % d["value"], "binds", d["binds"], "rounds", d["rounds_per_step"],
              "dirty-won rows", d["row_fallbacks_per_step"], "verified", d.get("verified_bind_set_equals_oracle"), d["kernel_ms_per_step"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
