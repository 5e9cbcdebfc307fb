#!/bin/bash
# the materialised-matrix roofline (K1) of configs 3 and 4 + the matrix parity tests
set -u
OUT=$PWD/gpurun_out/${1:-k1}
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix or argmax" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
for c in 3 4; do
  timeout 200 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/b$c.json" 2> "$OUT/b$c.err"
  python - "$OUT/b$c.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 2), "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "GB/s", d["roofline"]["achieved"])
PY
done
