#!/usr/bin/env bash
# round 3, call 8: fused per-round matrix + candidate-list launch (KB_FUSE_K13), evict-path hygiene, replicas-only N > 1 bench
set -uo pipefail
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r3_call8
mkdir -p "$out"
python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['kernel_ms_per_step'], d.get('verified_bind_set_equals_oracle'))"; }
for rep in 1 2; do
  for f in 1 0; do
    KB_FUSE_K13=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 2>/dev/null | line "fuse=$f c3" | tee -a "$out/summary.txt"
    KB_FUSE_K13=$f python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --verify 2>/dev/null | line "fuse=$f c4" | tee -a "$out/summary.txt"
    KB_FUSE_K13=$f python bench.py --survey-nodes --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line "fuse=$f survey" | tee -a "$out/summary.txt"
  done
done
for f in 1 0; do KB_FUSE_K13=$f python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line "fuse=$f c5" | tee -a "$out/summary.txt"; done
python scripts/time_preempt.py 5 1.0 2>/dev/null | tee -a "$out/summary.txt"
KB_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > "$out/bench_gpus2_gloo_one_gpu.json" 2> "$out/bench_gpus2.err"; echo "2 ranks on one GPU rc=$?" | tee -a "$out/summary.txt"
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
