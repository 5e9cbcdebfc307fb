#!/usr/bin/env bash
# Round 6's GPU calls:   gpurun -- 'bash scripts/gpu_r6.sh <step> [args]'   (writes gpurun_out/r6_<step>/*; the r5 steps stay in scripts/gpu_r5.sh)
#   lookahead  the selection kernel's look-ahead passes: differential subset under the selection kernel first (a wrong kernel ends the call early),
#              then the build against kube-batch_amd/libkbengine_base.so (the tree of round 5) on configs 4, 3, survey nodes, 5, then the per-phase trace
#   ab         same-box A/B of libkbengine.so against libkbengine_base.so, most important line first:   ab [configs...]   (default: 4 3 survey 5)
#   trace      the selection kernel's per-phase cycle trace (libkbengine_trace.so), configs 4 and 3
#   subset     the differential modules under both commit kernels (parity, adversarial, fuzz, full size, regressions, reload)
#   suite      the whole -m gpu suite
set -uo pipefail
cd "$(dirname "$0")/.."
step="${1:-suite}"; shift || true
out="gpurun_out/r6_${step}${R6_TAG:+_$R6_TAG}"
mkdir -p "$out"
python scripts/kernel_sources_sha.py > "$out/kernel_sources.sha256"
python scripts/kernel_sources_sha.py --tu > "$out/kernel_tu.sha256"
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print(d.get('ms_per_step'), d.get('verified_bind_set_equals_oracle'), d.get('kernel_ms_per_step'), 'rounds', d.get('rounds_per_step'), 'breaks', d.get('spec_breaks_per_step'), 'fallbacks', d.get('row_fallbacks_per_step'))" 2>/dev/null; }
bench_ab() {   # name, env assignments..., -- bench args
  local name="$1"; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  timeout 400 env "${envs[@]}" KB_K5_STATS=1 python bench.py --no-cpu-baseline "$@" > "$out/bench_${name}.json" 2> "$out/bench_${name}.err"
  echo "bench $name rc=$? $(ms "$out/bench_${name}.json")" | tee -a "$out/summary.txt"
  grep -h "kb select\|kb host\|kb probe\|kb K5" "$out/bench_${name}.err" | tee -a "$out/summary.txt"
}
base="KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_base.so"
cfg_args() { case "$1" in 4) echo "--config 4 --steps 5 --warmup 2";; 3) echo "--config 3 --steps 5 --warmup 2";; survey) echo "--config 3 --survey-nodes --steps 5 --warmup 2";;
                          5) echo "--config 5 --steps 2 --warmup 1";; 2) echo "--config 2 --steps 10 --warmup 3";; diverse) echo "--config 3 --diverse --steps 3 --warmup 1";; esac; }
ab_cfgs() {
  for c in "$@"; do
    bench_ab "c${c}_new" -- $(cfg_args $c) --verify
    [ -f kube-batch_amd/libkbengine_base.so ] && bench_ab "c${c}_base" $base -- $(cfg_args $c)
  done
}
trace_cfgs() {
  for cfg in "$@"; do
    KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_K5_STATS=1 timeout 300 python bench.py --no-cpu-baseline --config ${cfg} --steps 2 --warmup 1 \
      > "$out/trace_c${cfg}.json" 2> "$out/trace_c${cfg}.err"
    echo "== trace c${cfg} $(ms "$out/trace_c${cfg}.json")" | tee -a "$out/summary.txt"; grep -h "kb K5 trace\|kb K5\] rounds [0-9]\|kb select" "$out/trace_c${cfg}.err" | tee -a "$out/summary.txt"
  done
}
case "$step" in
lookahead)
  KB_COMMIT_KERNEL=select timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adversarial.py tests/test_gpu_regressions.py tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider -k "select" --maxfail=5 \
    > "$out/pytest_select.txt" 2>&1; rc=$?; echo "differential subset under the selection kernel rc=$rc $(tail -1 "$out/pytest_select.txt")" | tee -a "$out/summary.txt"
  if [ $rc -ne 0 ]; then grep -h "^FAILED\|^ERROR\|Error\|assert" "$out/pytest_select.txt" | head -30 | tee -a "$out/summary.txt"; fi
  ab_cfgs 4 3
  trace_cfgs 4 3
  ab_cfgs survey 5
  timeout 400 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -k "select" > "$out/pytest_fullsize.txt" 2>&1; echo "full-size digests under the selection kernel rc=$? $(tail -1 "$out/pytest_fullsize.txt")" | tee -a "$out/summary.txt"
  ;;
ab)
  if [ $# -eq 0 ]; then set -- 4 3 survey 5; fi
  ab_cfgs "$@"
  ;;
trace)
  if [ $# -eq 0 ]; then set -- 4 3; fi
  trace_cfgs "$@"
  ;;
subset)
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adversarial.py tests/test_gpu_regressions.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_reload.py tests/test_gpu_interpod.py \
    -q -m gpu -p no:cacheprovider --maxfail=10 > "$out/pytest_subset.txt" 2>&1; echo "differential modules, both commit kernels rc=$? $(tail -1 "$out/pytest_subset.txt")" | tee -a "$out/summary.txt"
  ;;
suite)
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 "$out/pytest_gpu.txt")" | tee -a "$out/summary.txt"
  ;;
*) echo "unknown step $step"; exit 2 ;;
esac
cat "$out/summary.txt"
