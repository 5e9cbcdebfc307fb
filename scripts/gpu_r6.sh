#!/usr/bin/env bash
# Round 6's GPU calls:   gpurun -- 'bash scripts/gpu_r6.sh <step> [args]'   (writes gpurun_out/r6_<step>/*; the r5 steps stay in scripts/gpu_r5.sh)
#   lookahead  the selection kernel's look-ahead passes: differential subset under the selection kernel first (a wrong kernel ends the call early),
#              then the build against kube-batch_amd/libkbengine_base.so (the tree of round 5) on configs 4, 3, survey nodes, 5, then the per-phase trace
#   ab         same-box A/B of libkbengine.so against libkbengine_base.so, most important line first:   ab [configs...]   (default: 4 3 survey 5)
#   trace      the selection kernel's per-phase cycle trace (libkbengine_trace.so), configs 4 and 3
#   subset     the differential modules under both commit kernels (parity, adversarial, fuzz, full size, regressions, reload)
#   suite      the whole -m gpu suite
# (Steps that set KB_EXPAND_TILES=0 — mix, mix2, expand, xchunk — compared the tiled row expansion with round 5's copy; that kernel and its switch were retired
#  afterwards, the steps are kept as the record of how the numbers under profiles/round6/ were produced: they need the tree of commit b8bea44.)
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
gaps)   # where a cycle's time goes outside the commit kernels:   gaps [configs...]   (kernel trace of two timed sessions + one warm-up, the host's own timeline)
  if [ $# -eq 0 ]; then set -- 5; fi
  export TMPDIR=/tmp
  for cfg in "$@"; do
    ( cd /tmp; KB_K5_STATS=1 rocprofv3 --kernel-trace -f csv -d "$OLDPWD/$out/trace_c${cfg}" -o bench -- python "$OLDPWD/bench.py" --config ${cfg} --steps 2 --warmup 1 --no-cpu-baseline \
        > "$OLDPWD/$out/gaps_c${cfg}.json" 2> "$OLDPWD/$out/gaps_c${cfg}.err" )
    echo "== gaps c${cfg} $(ms "$out/gaps_c${cfg}.json")" | tee -a "$out/summary.txt"
    grep -h "kb host\|kb probe\|kb overlap" "$out/gaps_c${cfg}.err" | tee -a "$out/summary.txt"
    python scripts/trace_gaps.py "$(find "$out/trace_c${cfg}" -name '*kernel_trace.csv' | head -1)" 3 > "$out/gaps_config${cfg}.txt" 2>&1; head -14 "$out/gaps_config${cfg}.txt" | tee -a "$out/summary.txt"
    find "$out/trace_c${cfg}" -name '*.csv' -size +8M -delete   # the trace itself is too large to bring back
  done
  ;;
mix)   # one call: the differential modules under both commit kernels, the default bench line (variants, driver-style), then two A/Bs on the same box:
       # the tiled row expansion against round 5's copy (roofline of configs 3 and 4), the yielding wait against the pure spin
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adversarial.py tests/test_gpu_regressions.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_interpod.py tests/test_gpu_waterfill.py \
    -q -m gpu -p no:cacheprovider --maxfail=10 > "$out/pytest_subset.txt" 2>&1; echo "differential modules, both commit kernels rc=$? $(tail -1 "$out/pytest_subset.txt")" | tee -a "$out/summary.txt"
  grep -h "^FAILED\|^ERROR" "$out/pytest_subset.txt" | head -20 | tee -a "$out/summary.txt"
  timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench default rc=$? $(ms "$out/bench_default.json")" | tee -a "$out/summary.txt"
  python - "$out/bench_default.json" <<'PY' | tee -a "$out/summary.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "verified", d.get("verified_bind_set_equals_oracle"), d.get("verified_evals_equal_oracle"), "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_refused"))
for k, v in d.get("variants", {}).items():
    print(" variant", k, v["ms_per_step"], v["verified"], (v.get("roofline") or {}).get("frac"), v.get("evals_per_s"))
PY
  for cfg in 3 4; do
    for t in 1 0; do
      KB_EXPAND_TILES=$t timeout 300 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > "$out/expand_c${cfg}_tiles${t}.json" 2> "$out/expand_c${cfg}_tiles${t}.err"
      python -c "import json; d=json.loads(open('$out/expand_c${cfg}_tiles${t}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('expand config $cfg tiles=$t', r['kernel'], 'frac', r['frac'], 'ms', r['avg_launch_ms'], 'eval-only', d['roofline_eval']['frac'], d['roofline_eval_all_rows']['frac'])" | tee -a "$out/summary.txt"
    done
  done
  for rep in 1 2; do
    bench_ab "c3_yield_r${rep}" -- --config 3 --steps 8 --warmup 2
    bench_ab "c3_spin_r${rep}" -- --config 3 --steps 8 --warmup 2 --spin-wait
  done
  ;;
profile)   # rocprofv3 evidence of the default bench command on THIS tree: kernel stats, HBM bytes of the matrix launches (FETCH / WRITE in separate passes),
           # SQ counters of the commit kernel with the commit workgroup ALONE in its launch (KB_PROFILE_SOLO_COMMIT=1 KB_FUSE_REPAIR=0: attributable to
           # its eight waves; the default launch carries ~30 mostly idle workgroups) — then   python scripts/summarize_profile.py r6_profile profiles/round6
  export TMPDIR=/tmp
  CMD="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
  P="$PWD/$out"
  ( cd /tmp
    rocprofv3 --kernel-trace --stats -f csv -d "$P/trace" -o bench -- $CMD > "$P/bench_trace.log" 2>&1
    rocprofv3 --pmc FETCH_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_fetch" -o bench -- $CMD > "$P/bench_pmc_fetch.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_write" -o bench -- $CMD > "$P/bench_pmc_write.log" 2>&1
    KB_PROFILE_SOLO_COMMIT=1 KB_FUSE_REPAIR=0 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv --kernel-include-regex "k_commit" \
      -d "$P/pmc_commit_a" -o bench -- $CMD > "$P/bench_pmc_commit_a.log" 2>&1
    KB_PROFILE_SOLO_COMMIT=1 KB_FUSE_REPAIR=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv --kernel-include-regex "k_commit" \
      -d "$P/pmc_commit_b" -o bench -- $CMD > "$P/bench_pmc_commit_b.log" 2>&1
  )
  echo "the commit workgroup alone in its launch (KB_PROFILE_SOLO_COMMIT=1 KB_FUSE_REPAIR=0: no L2 helpers, the repair rows a launch of their own): the sums per dispatch are of its eight waves" > "$out/pmc_commit_launch_shape.txt"
  find "$out" -name "*.csv" | head -20 | tee -a "$out/summary.txt"
  python scripts/summarize_profile.py r6_profile profiles/round6 2>&1 | tail -2 | tee -a "$out/summary.txt"
  mkdir -p "$out/profiles_round6" && cp -r profiles/round6/. "$out/profiles_round6/"
  python scripts/trace_gaps.py "$out/trace/bench_kernel_trace.csv" 3 > "$out/gaps_config3.txt" 2>&1; head -12 "$out/gaps_config3.txt" | tee -a "$out/summary.txt"
  ;;
mix2)  # the tiled expansion again (rows preloaded), the evict action's host timeline at 1M x 50k, the profile passes, two gloo ranks on the one GPU
  for cfg in 3 4; do
    for t in 1 0; do
      KB_EXPAND_TILES=$t timeout 300 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > "$out/expand_c${cfg}_tiles${t}.json" 2> "$out/expand_c${cfg}_tiles${t}.err"
      python -c "import json; d=json.loads(open('$out/expand_c${cfg}_tiles${t}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('expand config $cfg tiles=$t', r['kernel'], 'frac', r['frac'], 'ms', r['avg_launch_ms'], 'eval-only', d['roofline_eval']['frac'], d['roofline_eval_all_rows']['frac'])" | tee -a "$out/summary.txt"
    done
  done
  KB_EVICT_TRACE=1 timeout 600 python bench.py --config 5 --preempt --steps 2 --warmup 1 --no-cpu-baseline --verify > "$out/bench_c5_preempt.json" 2> "$out/bench_c5_preempt.err"
  echo "config 5 + preempt rc=$? $(ms "$out/bench_c5_preempt.json")" | tee -a "$out/summary.txt"; grep -h "kb evict" "$out/bench_c5_preempt.err" | tail -2 | tee -a "$out/summary.txt"
  bash scripts/gpu_r6.sh profile > "$out/profile_step.txt" 2>&1; tail -16 "$out/profile_step.txt" | tee -a "$out/summary.txt"
  KB_SCALE_GLOO=1 timeout 900 bash scripts/scale_curve.sh "$out/scale" 3 > "$out/scale_curve_log.txt" 2>&1; cat "$out/scale/scale_curve.txt" | tee -a "$out/summary.txt"
  ;;
expand)  # the tiled expansion: bench's own event timing (normal run) for the default build, the non-temporal build (libkbengine_nt.so), round 5's copy
         # (KB_EXPAND_TILES=0), each twice; then the same under rocprofv3 --kernel-trace (the profiler's durations beside the bench's own)
  one() { # name, env..., -- args
    local name="$1"; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    timeout 300 env "${envs[@]}" python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$out/x_${name}.json" 2> "$out/x_${name}.err"
    python -c "import json; d=json.loads(open('$out/x_${name}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$name', r['kernel'], 'frac', r['frac'], 'ms', r['avg_launch_ms'])" | tee -a "$out/summary.txt"
  }
  base="KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_base.so"   # (here: an earlier build of the tiled kernel, if one is beside the default)
  for rep in 1 2 3; do
    for cfg in 3 4; do
      one "c${cfg}_tiles_r${rep}" -- --config $cfg
      [ -f kube-batch_amd/libkbengine_base.so ] && one "c${cfg}_base_r${rep}" $base -- --config $cfg
      one "c${cfg}_rowcopy_r${rep}" KB_EXPAND_TILES=0 -- --config $cfg
    done
  done
  one "c5_tiles" -- --config 5
  one "c5_rowcopy" KB_EXPAND_TILES=0 -- --config 5
  export TMPDIR=/tmp
  P="$PWD/$out"
  for v in tiles; do
    envs=()
    ( cd /tmp; env "${envs[@]}" rocprofv3 --kernel-trace --stats -f csv -d "$P/trace_$v" -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$P/trace_$v.log" 2>&1 ) || true
    grep -h "k_expand\|k_matrix<1, 16>" "$P/trace_$v/bench_kernel_stats.csv" 2>/dev/null | cut -c1-140 | tee -a "$out/summary.txt"
    grep -h '"metric"' "$P/trace_$v.log" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('under rocprofv3 ($v):', d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a "$out/summary.txt"
  done
  ;;
final)   # the closing evidence on the tree as it stands: the whole -m gpu suite, the rocprofv3 passes summarised ON THE BOX into profiles/round6 (so that the bench
         # line below quotes them), the default bench with its variants, the single-configuration lines, smoke, a fresh-seed hunt, two gloo ranks on the one GPU
  timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 "$out/pytest_gpu.txt")" | tee -a "$out/summary.txt"
  grep -h "^FAILED\|^ERROR" "$out/pytest_gpu.txt" | head -20 | tee -a "$out/summary.txt"
  R6_TAG= bash scripts/gpu_r6.sh profile > "$out/profile_step.txt" 2>&1; tail -14 "$out/profile_step.txt" | tee -a "$out/summary.txt"
  mkdir -p "$out/profiles_round6" && cp -r profiles/round6/. "$out/profiles_round6/"
  timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench default rc=$? $(ms "$out/bench_default.json")" | tee -a "$out/summary.txt"
  python - "$out/bench_default.json" <<'PY' | tee -a "$out/summary.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "verified", d.get("verified_bind_set_equals_oracle"), d.get("verified_evals_equal_oracle"), "roofline", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_refused"))
print("commit counters:", (d["roofline_commit"].get("counters_launch_shape") or d["roofline_commit"].get("counters_refused")))
for k, v in d.get("variants", {}).items():
    print(" variant", k, v["ms_per_step"], v["verified"], (v.get("roofline") or {}).get("frac"), v.get("evals_per_s"))
print("loads", d.get("session_load_ms_samples"))
PY
  bench_ab survey_nodes -- --config 3 --survey-nodes --steps 5 --warmup 2 --verify
  bench_ab config4 -- --config 4 --steps 5 --warmup 2 --verify
  bench_ab config2 -- --config 2 --steps 10 --warmup 3 --verify
  bench_ab config5 -- --config 5 --steps 3 --warmup 1 --verify
  KB_EVICT_TRACE=1 bench_ab config5_preempt -- --config 5 --preempt --steps 2 --warmup 1 --verify
  grep -h "kb evict" "$out/bench_config5_preempt.err" | tail -2 | tee -a "$out/summary.txt"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a "$out/summary.txt"
  KB_HUNT_OFFSET=${KB_HUNT_OFFSET:-80000} timeout 600 python scripts/gpu_hunt.py ${HUNT_ARGS:-200 600 300} > "$out/hunt.txt" 2>&1; echo "fresh-seed hunt (both kernels) rc=$? $(tail -1 "$out/hunt.txt")" | tee -a "$out/summary.txt"
  KB_SCALE_GLOO=1 timeout 900 bash scripts/scale_curve.sh "$out/scale" 3 > "$out/scale_curve_log.txt" 2>&1; cat "$out/scale/scale_curve.txt" | tee -a "$out/summary.txt"
  ;;
variants)  # same-box A/B of builds of the selection kernel: kube-batch_amd/libkbengine_<tag>.so beside the default one, alternating, each verified:   variants tag,tag,... [configs...]
  IFS=',' read -r -a tags <<< "${1:-prio}"; shift || true
  if [ $# -eq 0 ]; then set -- 3 4 survey; fi
  for rep in 1 2; do
    for c in "$@"; do
      bench_ab "c${c}_default_r${rep}" -- $(cfg_args $c) --verify
      for t in "${tags[@]}"; do bench_ab "c${c}_${t}_r${rep}" KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_$t.so -- $(cfg_args $c) --verify; done
    done
  done
  ;;
xchunk)  # rows per chunk of the tiled expansion: the default build (64) against libkbengine_x8.so / _x16.so and the row copy, configs 3 and 4, three times each
  one() { local name="$1"; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    timeout 300 env "${envs[@]}" python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$out/x_${name}.json" 2> "$out/x_${name}.err"
    python -c "import json; d=json.loads(open('$out/x_${name}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$name', 'frac', r['frac'], 'ms', r['avg_launch_ms'])" | tee -a "$out/summary.txt"; }
  for rep in 1 2 3; do
    for cfg in 3 4; do
      one "c${cfg}_x64_r${rep}" -- --config $cfg
      one "c${cfg}_x16_r${rep}" KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_x16.so -- --config $cfg
      one "c${cfg}_x8_r${rep}" KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_x8.so -- --config $cfg
      one "c${cfg}_rowcopy_r${rep}" KB_EXPAND_TILES=0 -- --config $cfg
    done
  done
  trace_cfgs 3
  ;;
profile_bench)  # the profile passes (summarised on the box into profiles/round6), then the default bench line, which quotes them
  R6_TAG= bash scripts/gpu_r6.sh profile > "$out/profile_step.txt" 2>&1; tail -14 "$out/profile_step.txt" | tee -a "$out/summary.txt"
  mkdir -p "$out/profiles_round6" && cp -r profiles/round6/. "$out/profiles_round6/"
  timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench default rc=$? $(ms "$out/bench_default.json")" | tee -a "$out/summary.txt"
  python - "$out/bench_default.json" <<'PY' | tee -a "$out/summary.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "verified", d.get("verified_bind_set_equals_oracle"), d.get("verified_evals_equal_oracle"), "roofline", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_refused"))
print("commit counters:", (d["roofline_commit"].get("counters_launch_shape") or d["roofline_commit"].get("counters_refused")))
for k, v in d.get("variants", {}).items():
    print(" variant", k, v["ms_per_step"], v["verified"], (v.get("roofline") or {}).get("frac"), v.get("evals_per_s"))
print("loads", d.get("session_load_ms_samples"))
PY
  ;;
pmc_matrix)  # the matrix launches' HBM bytes and the kernel stats again after a change that left the commit kernels' translation unit alone (host side of the
             # expansion, kb_kernels.hip): FETCH_SIZE, WRITE_SIZE, kernel trace; summarised on the box (the commit kernel's lines of profiles/round6 stay)
  export TMPDIR=/tmp
  CMD="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
  P="$PWD/gpurun_out/r6_profile"; mkdir -p "$P"
  python scripts/kernel_sources_sha.py > "$P/kernel_sources.sha256"; python scripts/kernel_sources_sha.py --tu > "$P/kernel_tu.sha256"
  ( cd /tmp
    rocprofv3 --pmc FETCH_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_fetch" -o bench -- $CMD > "$P/bench_pmc_fetch.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_write" -o bench -- $CMD > "$P/bench_pmc_write.log" 2>&1
    rocprofv3 --kernel-trace --stats -f csv -d "$P/trace" -o bench -- $CMD > "$P/bench_trace.log" 2>&1
  )
  python scripts/summarize_profile.py r6_profile profiles/round6 2>&1 | tail -2 | tee -a "$out/summary.txt"
  mkdir -p "$out/profiles_round6" && cp -r profiles/round6/rocprofv3_* profiles/round6/kernel_* profiles/round6/bench_under_rocprofv3_kernel_trace.json "$out/profiles_round6/"
  cat profiles/round6/rocprofv3_pmc_k_matrix.csv | cut -c1-150 | tee -a "$out/summary.txt"
  timeout 600 python bench.py --no-cpu-baseline --verify > "$out/bench_c3.json" 2> "$out/bench_c3.err"
  python -c "import json; d=json.loads(open('$out/bench_c3.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bench: ms', d['ms_per_step'], 'verified', d.get('verified_bind_set_equals_oracle'), 'roofline', r['frac'], r['avg_launch_ms'], 'traffic', r.get('traffic'), r.get('traffic_refused'), 'ratio', (r.get('traffic') or 0) / r['bytes_per_launch'])" | tee -a "$out/summary.txt"
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
