#!/usr/bin/env bash
# The N > 1 scaling curve, ready to run on a node with several MI355X:   bash scripts/scale_curve.sh [out-dir] [steps]
# For N in {1, 2, 4, 8} (as far as the node has GPUs) x mode in {sessions, sharded}: bench.py under torch.distributed.run, one rank per GPU
# over RCCL, exactly as the driver launches it; every line must report n_gpus == N, dist_backend == "nccl" (RCCL really carried N ranks)
# and a verified result.  Writes <out>/scale_<mode>_<N>.json and a table <out>/scale_curve.txt with the per-N value and value(N) / value(1).
#   sessions  one independent session per GPU, no data-path collective ("weak": value = the slowest rank's per-session rate)
#   sharded   north_star's task-row split of ONE session (KB_DIST_MODE=sharded; "strong")
# No node with more than one GPU was available to any round so far: this script has only been run with N = 1 (and with N = 2 ranks
# sharing one GPU over gloo: KB_SCALE_GLOO=1, which checks the plumbing, not the bandwidth).
set -uo pipefail
cd "$(dirname "$0")/.."
out="${1:-gpurun_out/scale}"; steps="${2:-5}"
mkdir -p "$out"
ngpu=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "GPUs on this node: $ngpu" | tee "$out/scale_curve.txt"
backend_env=(); max_n=$ngpu
if [ "${KB_SCALE_GLOO:-0}" = 1 ]; then backend_env=(KB_DIST_BACKEND=gloo); max_n=2; fi
rc=0
for mode in sessions sharded; do
  for n in 1 2 4 8; do
    [ "$n" -gt "$max_n" ] && continue
    f="$out/scale_${mode}_${n}.json"
    if [ "$n" = 1 ]; then
      timeout 900 python bench.py --gpus 1 --steps "$steps" --warmup 2 --no-cpu-baseline --verify > "$f" 2> "$f.err"
    else
      timeout 900 env "${backend_env[@]}" KB_DIST_MODE=$mode HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" \
        --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus "$n" --steps "$steps" --warmup 2 > "$f" 2> "$f.err"
    fi
    python - "$f" "$mode" "$n" "${KB_SCALE_GLOO:-0}" <<'PY' | tee -a "$out/scale_curve.txt" || rc=1
import json, sys
f, mode, n, gloo = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4] == "1"
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
except Exception as e:
    print(f"{mode:9s} N={n}: no bench line ({e})"); sys.exit(1)
ok = d["n_gpus"] == n
if n > 1:
    ok = ok and d.get("dist_backend") == ("gloo" if gloo else "nccl")
    ok = ok and (d.get("sessions_verified_against_golden_digests") is not False) and (d.get("replicas_agree") is not False)
else:
    ok = ok and d.get("verified_bind_set_equals_oracle") is True
print(f"{mode:9s} N={n}: value {d['value']:.4g} {d['unit']}, {d['ms_per_step']:.2f} ms/step, scaling {d['scaling']}, backend {d.get('dist_backend')}, aggregate {d.get('aggregate_evals_per_s')}, {'ok' if ok else 'CHECK FAILED'}")
sys.exit(0 if ok else 1)
PY
  done
done
exit $rc
