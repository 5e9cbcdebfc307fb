#!/bin/bash
# The first GPU call of a round (through gpurun): everything that was left unverified when the previous round's GPU budget ran out.
#   gpurun --timeout 1500 -- 'bash scripts/first_gpu_call.sh r2'
# 1. the regular -m gpu suite; 2. the opt-in adversarial differential test (never seen on a GPU yet); 3. the headline bench;
# 4. config 4 with its binpack weights, verified against the oracle (never timed at full size: DESIGN.md §7 correction, §9.1).
set -u
TAG=${1:-r2}
OUT=$PWD/gpurun_out/first_$TAG
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest -m gpu rc=$?" | tee -a "$OUT/summary.txt"
KB_GPU_ADVERSARIAL=1 timeout 600 python -m pytest tests/test_gpu_adversarial.py -q -m gpu > "$OUT/pytest_adversarial.log" 2>&1; echo "adversarial rc=$?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/pytest_gpu.log" "$OUT/pytest_adversarial.log" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --steps 3 --warmup 1 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"; echo "bench c3 rc=$?" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --config 4 --steps 3 --warmup 1 --verify --no-cpu-baseline > "$OUT/bench_c4_binpack.json" 2> "$OUT/bench_c4.err"; echo "bench c4 binpack rc=$?" | tee -a "$OUT/summary.txt"
python - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys, os
for f in ("bench_c3.json", "bench_c4_binpack.json"):
    p = os.path.join(sys.argv[1], f)
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(f, "ms/step", round(d["ms_per_step"], 2), "evals/s %.3g" % d["value"], "binds", d["binds"], "rounds", d["rounds_per_step"],
              "row-mode rows", d["row_fallbacks_per_step"], "verified", d.get("verified_bind_set_equals_oracle"), d["kernel_ms_per_step"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
