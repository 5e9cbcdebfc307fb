#!/usr/bin/env bash
# The N > 1 scaling curve, ready to run on a node with several MI355X:   bash scripts/scale_curve.sh [out-dir] [steps]
# For N in {1, 2, 4, 8} (as far as the node has GPUs): bench.py under torch.distributed.run, one rank per GPU over RCCL, exactly as the driver
# launches it (no environment: the line carries both multi-GPU answers); every line must report n_gpus == N, dist_backend == "nccl" (RCCL
# really carried N ranks) and verified results.  Writes <out>/scale_<N>.json and a table <out>/scale_curve.txt.
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
# round 6: with --gpus N and no KB_DIST_MODE bench.py prints BOTH answers in one line (`sharded`: north_star's task-row split, the line's value;
# `sessions`: one independent session per GPU) — one launch per N
for n in 1 2 4 8; do
  [ "$n" -gt "$max_n" ] && continue
  f="$out/scale_${n}.json"
  if [ "$n" = 1 ]; then
    timeout 900 python bench.py --gpus 1 --steps "$steps" --warmup 2 --no-cpu-baseline --verify > "$f" 2> "$f.err"
  else
    timeout 900 env "${backend_env[@]}" HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" \
      --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus "$n" --steps "$steps" --warmup 2 > "$f" 2> "$f.err"
  fi
  python - "$f" "$n" "${KB_SCALE_GLOO:-0}" <<'PY' | tee -a "$out/scale_curve.txt" || rc=1
import json, sys
f, n, gloo = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1"
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
except Exception as e:
    print(f"N={n}: no bench line ({e})"); sys.exit(1)
ok = d["n_gpus"] == n
if n == 1:
    ok = ok and d.get("verified_bind_set_equals_oracle") is True
    print(f"N=1: value {d['value']:.4g} {d['unit']}, {d['ms_per_step']:.2f} ms/step, {'ok' if ok else 'CHECK FAILED'}")
else:
    sh, se = d["sharded"], d["sessions"]
    want = "gloo" if gloo else "nccl"
    ok = ok and d.get("dist_backend") == want and sh["dist_backend"] == want and sh["ranks"] == n and (gloo or sh["ranks_seen_by_rccl"] == n)
    ok = ok and sh["verified"] is True and se["verified"] is True and d["scaling"] == "strong" and d["value"] == sh["value"]
    print(f"N={n} sharded : value {sh['value']:.4g} evals/s, {sh['ms_per_step']:.2f} ms/step, rounds {sh['rounds_per_step']}, breaks {sh['spec_breaks_per_step']}, "
          f"all-gather {sh['allgather_us_per_round']} us/round ({sh['rounds_that_exchanged_lists_per_step']} rounds per step), all-reduce {sh['allreduce_us_per_round']} us/round, backend {sh['dist_backend']}, verified {sh['verified']}")
    print(f"N={n} sessions: value {se['value']:.4g} evals/s per session, {se['ms_per_step']:.2f} ms/step, aggregate {se['aggregate_evals_per_s']:.4g}, verified {se['verified']}, {'ok' if ok else 'CHECK FAILED'}")
sys.exit(0 if ok else 1)
PY
done
exit $rc
