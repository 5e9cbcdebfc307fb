#!/usr/bin/env bash
# Round 5's GPU calls, one parameterised script:   gpurun -- 'bash scripts/gpu_r5.sh <step> [args]'   (writes gpurun_out/r5_<step>/*)
#   first    the reload module (one engine, many sessions; soak), the whole -m gpu suite, the default bench, kb_session_load traces of configs 3, 4, 5
#   ab       same-box A/B of engine builds: every kube-batch_amd/libkbengine_<tag>.so beside the default one (configs 3, survey, 4, 5), each
#            verified; optionally the differential suites on one of them:   ab [tag-to-test]
#   fuse     A/B of KB_FUSE_REPAIR (repair workgroups inside the selection kernel's launch, or a launch of their own) + the differential suites on both
#   abq      the default build against libkbengine_base.so on configs 3, 5, 4 and the survey's nodes, shortest useful order first (a call of a few minutes)
#   trace    the selection kernel's per-phase cycle trace (make EXTRA=-DKB_K9_TRACE OUT=../libkbengine_trace.so), configs 3 and 4
#   profile  rocprofv3 of the default command: kernel stats, HBM bytes of the matrix launches, SQ counters of the commit kernel
#            (scripts/summarize_profile.py r5_profile profiles/round5)
#   pmc_matrix  FETCH_SIZE / WRITE_SIZE passes of the matrix launches + the kernel trace only (a change to kb_kernels.hip alone)
#   suite    the whole -m gpu suite          bench   the default bench line and the single-configuration lines
#   loads    kb_session_load: KB_LOAD_TRACE of configs 3, 4, 5 and the bench's load statistics
#   scale    N in {1, 2, 4, 8} x {sessions, sharded} on whatever GPUs the box has (scripts/scale_curve.sh)
#   final    the closing evidence: suite, profile (summarised on the box), bench lines, smoke, N = 2 plumbing over gloo
set -uo pipefail
cd "$(dirname "$0")/.."
step="${1:-suite}"; shift || true
out="gpurun_out/r5_${step}"
mkdir -p "$out"
python scripts/kernel_sources_sha.py > "$out/kernel_sources.sha256"
python scripts/kernel_sources_sha.py --tu > "$out/kernel_tu.sha256"
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print(d.get('ms_per_step'), d.get('verified_bind_set_equals_oracle'), d.get('kernel_ms_per_step'), 'load', d.get('session_load_ms'), d.get('session_load_ms_max'))" 2>/dev/null; }
bench_ab() {   # name, env assignments..., -- bench args
  local name="$1"; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  timeout 400 env "${envs[@]}" KB_K5_STATS=1 python bench.py --no-cpu-baseline "$@" > "$out/bench_${name}.json" 2> "$out/bench_${name}.err"
  echo "bench $name rc=$? $(ms "$out/bench_${name}.json")" | tee -a "$out/summary.txt"
  grep -h "kb select\|kb host\|kb probe" "$out/bench_${name}.err" | tee -a "$out/summary.txt"
}
case "$step" in
first)
  timeout 900 python -m pytest tests/test_gpu_reload.py -q -m gpu -p no:cacheprovider -x -s > "$out/pytest_reload.txt" 2>&1; echo "reload module rc=$? $(tail -1 "$out/pytest_reload.txt")" | tee -a "$out/summary.txt"
  grep -h "^soak" "$out/pytest_reload.txt" | tee -a "$out/summary.txt"
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_reload.py > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite (without the reload module) rc=$? $(tail -1 "$out/pytest_gpu.txt")" | tee -a "$out/summary.txt"
  timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$? $(ms "$out/bench_default.json")" | tee -a "$out/summary.txt"
  python -c "import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('variants'), indent=1)); print('loads', d.get('session_load_ms_samples'))" | tee -a "$out/summary.txt"
  for cfg in 3 4 5; do
    KB_LOAD_TRACE=1 timeout 300 python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline > "$out/load_c${cfg}.json" 2> "$out/session_load_trace_config${cfg}.txt"
    echo "config $cfg load: $(ms "$out/load_c${cfg}.json")" | tee -a "$out/summary.txt"
  done
  ;;
loads)
  for cfg in 3 4 5; do
    KB_LOAD_TRACE=1 timeout 300 python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline > "$out/load_c${cfg}.json" 2> "$out/session_load_trace_config${cfg}.txt"
    echo "config $cfg load: $(ms "$out/load_c${cfg}.json")" | tee -a "$out/summary.txt"
  done
  ;;
ab)   # ab <tag,tag,...> [test]: kube-batch_amd/libkbengine_<tag>.so beside the default build, same box, alternating; `test`: the differential
      # suites (selection kernel, reload module) on the DEFAULT build afterwards
  IFS=',' read -r -a tags <<< "${1:-base}"
  libs=("default" "${tags[@]}")
  libpath() { if [ "$1" = default ]; then echo "$PWD/kube-batch_amd/libkbengine.so"; else echo "$PWD/kube-batch_amd/libkbengine_$1.so"; fi; }
  for rep in 1 2; do
    for tag in "${libs[@]}"; do bench_ab "c3_${tag}_r${rep}" KB_ENGINE_LIB=$(libpath $tag) -- --config 3 --steps 5 --warmup 2 --verify; done
  done
  for tag in "${libs[@]}"; do
    bench_ab "survey_${tag}" KB_ENGINE_LIB=$(libpath $tag) -- --config 3 --survey-nodes --steps 5 --warmup 2 --verify
    bench_ab "c4_${tag}" KB_ENGINE_LIB=$(libpath $tag) -- --config 4 --steps 5 --warmup 2 --verify
    bench_ab "c5_${tag}" KB_ENGINE_LIB=$(libpath $tag) -- --config 5 --steps 2 --warmup 1
    [ -n "${AB_SKIP_C2:-}" ] || bench_ab "c2_${tag}" KB_ENGINE_LIB=$(libpath $tag) -- --config 2 --steps 10 --warmup 3 --verify
  done
  if [ "${2:-}" = test ]; then
    timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "select or reload or fullsize or sharded or waterfill or framework" --maxfail=10 > "$out/pytest_default.txt" 2>&1
    echo "differential suites on the default build (selection kernel, reload, full size) rc=$? $(tail -1 "$out/pytest_default.txt")" | tee -a "$out/summary.txt"
  fi
  ;;
abq)   # the round's last GPU minutes: the default build against kube-batch_amd/libkbengine_base.so (the tree of call 26), alternating, most
       # important line first — the call may be cut anywhere: every line is written (and verified) on its own
  base="KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_base.so"
  bench_ab c3_new_r1 -- --config 3 --steps 5 --warmup 2 --verify
  bench_ab c3_base_r1 $base -- --config 3 --steps 5 --warmup 2 --verify
  bench_ab c3_new_r2 -- --config 3 --steps 5 --warmup 2 --verify
  bench_ab c3_base_r2 $base -- --config 3 --steps 5 --warmup 2 --verify
  bench_ab c5_new -- --config 5 --steps 2 --warmup 1 --verify
  bench_ab c5_base $base -- --config 5 --steps 2 --warmup 1
  bench_ab c4_new -- --config 4 --steps 3 --warmup 1 --verify
  bench_ab c4_base $base -- --config 4 --steps 3 --warmup 1
  bench_ab survey_new -- --config 3 --survey-nodes --steps 3 --warmup 1 --verify
  bench_ab survey_base $base -- --config 3 --survey-nodes --steps 3 --warmup 1
  timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x > "$out/pytest_subset.txt" 2>&1; echo "full-size + parity modules rc=$? $(tail -1 "$out/pytest_subset.txt")" | tee -a "$out/summary.txt"
  ;;
last)   # the round's last call, shortest useful order first (it may be cut anywhere): the host side of the probe (one stream operation, launched
        # before the answer is absorbed) against the build before it, the rocprofv3 passes on the final device sources, then probes at every break
        # for the large configurations (libkbengine_p64.so: -DKB_PROBE_SPARSE_ABOVE='(64ull<<20)')
  base="KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_base.so"; p64="KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_p64.so"
  bench_ab c3_new_r1 -- --config 3 --steps 5 --warmup 2 --verify
  bench_ab c3_base_r1 $base -- --config 3 --steps 5 --warmup 2 --verify
  bench_ab c3_new_r2 -- --config 3 --steps 5 --warmup 2 --verify
  bench_ab c3_base_r2 $base -- --config 3 --steps 5 --warmup 2 --verify
  bash scripts/gpu_r5.sh profile > "$out/profile_step.txt" 2>&1; tail -3 "$out/profile_step.txt" | tee -a "$out/summary.txt"
  bench_ab c5_new -- --config 5 --steps 2 --warmup 1 --verify
  bench_ab c5_p64 $p64 -- --config 5 --steps 2 --warmup 1
  bench_ab c4_new -- --config 4 --steps 3 --warmup 1 --verify
  bench_ab c4_p64 $p64 -- --config 4 --steps 3 --warmup 1 --verify
  bench_ab c5_p64v $p64 -- --config 5 --steps 2 --warmup 1 --verify
  ;;
fuse)   # the repair workgroups inside the selection kernel's launch (default) against the launch of their own (KB_FUSE_REPAIR=0), same box, same
        # library, alternating; then the differential suites on the default and the selection / full-size ones on the other path
  for rep in $(seq 1 ${AB_REPS:-2}); do
    bench_ab "c3_fused_r${rep}" -- --config 3 --steps 5 --warmup 2 --verify
    bench_ab "c3_unfused_r${rep}" KB_FUSE_REPAIR=0 -- --config 3 --steps 5 --warmup 2 --verify
  done
  for v in fused unfused; do
    envs=(); [ "$v" = unfused ] && envs=(KB_FUSE_REPAIR=0)
    bench_ab "survey_${v}" "${envs[@]}" -- --config 3 --survey-nodes --steps 5 --warmup 2 --verify
    bench_ab "c4_${v}" "${envs[@]}" -- --config 4 --steps 5 --warmup 2 --verify
    bench_ab "c5_${v}" "${envs[@]}" -- --config 5 --steps 2 --warmup 1 --verify
    [ -n "${AB_SKIP_C2:-}" ] || bench_ab "c2_${v}" "${envs[@]}" -- --config 2 --steps 10 --warmup 3 --verify
  done
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "select or reload or fullsize or sharded or waterfill or framework" --maxfail=10 > "$out/pytest_default.txt" 2>&1
  echo "differential suites, repair inside the launch rc=$? $(tail -1 "$out/pytest_default.txt")" | tee -a "$out/summary.txt"
  KB_FUSE_REPAIR=0 timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "fullsize or (select and (fuzz or parity or adversarial))" --maxfail=10 > "$out/pytest_unfused.txt" 2>&1
  echo "selection + full-size suites, repair as a launch of its own rc=$? $(tail -1 "$out/pytest_unfused.txt")" | tee -a "$out/summary.txt"
  ;;
trace)
  for cfg in "3" "4"; do
    KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_K5_STATS=1 python bench.py --no-cpu-baseline --config ${cfg} --steps 2 --warmup 1 \
      > "$out/trace_c${cfg}.json" 2> "$out/trace_c${cfg}.err"
    echo "== trace c${cfg} $(ms "$out/trace_c${cfg}.json")" | tee -a "$out/summary.txt"; grep -h "kb K5 trace\|kb K5\] rounds [0-9]\|kb select" "$out/trace_c${cfg}.err" | tee -a "$out/summary.txt"
  done
  ;;
profile)   # rocprofv3 evidence of the default bench command: kernel stats, HBM bytes of the matrix launches, SQ counters of the commit kernel
  export TMPDIR=/tmp
  CMD="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
  P="$PWD/$out"
  ( cd /tmp
    rocprofv3 --kernel-trace --stats -f csv -d "$P/trace" -o bench -- $CMD > "$P/bench_trace.log" 2>&1
    rocprofv3 --pmc FETCH_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_fetch" -o bench -- $CMD > "$P/bench_pmc_fetch.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_write" -o bench -- $CMD > "$P/bench_pmc_write.log" 2>&1
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv --kernel-include-regex "k_commit" \
      -d "$P/pmc_commit_a" -o bench -- $CMD > "$P/bench_pmc_commit_a.log" 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv --kernel-include-regex "k_commit" \
      -d "$P/pmc_commit_b" -o bench -- $CMD > "$P/bench_pmc_commit_b.log" 2>&1
  )
  find "$out" -name "*.csv" | head -20 | tee -a "$out/summary.txt"
  ;;
pmc_matrix)   # the matrix launches' HBM bytes and the kernel stats again, after a change to kb_kernels.hip that left the commit kernels' translation
              # unit alone (scripts/kernel_sources_sha.py --tu): FETCH_SIZE, WRITE_SIZE, then the kernel trace — in that order, the call may be cut
  export TMPDIR=/tmp
  CMD="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
  P="$PWD/$out"
  ( cd /tmp
    rocprofv3 --pmc FETCH_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_fetch" -o bench -- $CMD > "$P/bench_pmc_fetch.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_write" -o bench -- $CMD > "$P/bench_pmc_write.log" 2>&1
    rocprofv3 --kernel-trace --stats -f csv -d "$P/trace" -o bench -- $CMD > "$P/bench_trace.log" 2>&1
  )
  find "$out" -name "*.csv" | head -20 | tee -a "$out/summary.txt"
  ;;
suite)
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 "$out/pytest_gpu.txt")" | tee -a "$out/summary.txt"
  ;;
bench)
  timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$? $(ms "$out/bench_default.json")" | tee -a "$out/summary.txt"
  bench_ab survey_nodes -- --config 3 --survey-nodes --steps 5 --warmup 2 --verify
  bench_ab config4 -- --config 4 --steps 5 --warmup 2 --verify
  bench_ab config2 -- --config 2 --steps 10 --warmup 3 --verify
  bench_ab config5 -- --config 5 --steps 3 --warmup 1 --verify
  bench_ab config5_preempt -- --config 5 --preempt --steps 2 --warmup 1 --verify
  ;;
scale)
  bash scripts/scale_curve.sh "$out" 2>&1 | tee -a "$out/summary.txt"
  ;;
final)   # the round's closing evidence on the final tree: whole suite, rocprofv3 passes summarised ON THE BOX into profiles/round5 (so that the bench
         # line below quotes them), the default bench and the single-configuration lines, smoke, the N = 2 plumbing of scripts/scale_curve.sh
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 "$out/pytest_gpu.txt")" | tee -a "$out/summary.txt"
  bash scripts/gpu_r5.sh profile > "$out/profile_step.txt" 2>&1
  python scripts/summarize_profile.py r5_profile profiles/round5 >> "$out/profile_step.txt" 2>&1
  python scripts/trace_gaps.py gpurun_out/r5_profile/trace/bench_kernel_trace.csv 3 > "$out/gaps_config3.txt" 2>&1; head -12 "$out/gaps_config3.txt" | tee -a "$out/summary.txt"
  timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$? $(ms "$out/bench_default.json")" | tee -a "$out/summary.txt"
  python -c "import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print(json.dumps({k: (v['ms_per_step'], v['verified']) for k, v in d['variants'].items()})); print('roofline', d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_refused')); print('loads', d['session_load_ms_samples'])" | tee -a "$out/summary.txt"
  bench_ab survey_nodes -- --config 3 --survey-nodes --steps 5 --warmup 2 --verify
  bench_ab config4 -- --config 4 --steps 5 --warmup 2 --verify
  bench_ab config2 -- --config 2 --steps 10 --warmup 3 --verify
  bench_ab config5 -- --config 5 --steps 3 --warmup 1 --verify
  bench_ab config5_preempt -- --config 5 --preempt --steps 2 --warmup 1 --verify
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a "$out/summary.txt"
  KB_HUNT_OFFSET=${KB_HUNT_OFFSET:-50000} timeout 600 python scripts/gpu_hunt.py ${HUNT_ARGS:-200 600 300} > "$out/hunt.txt" 2>&1; echo "fresh-seed hunt (both kernels) rc=$? $(tail -1 "$out/hunt.txt")" | tee -a "$out/summary.txt"
  KB_SCALE_GLOO=1 timeout 900 bash scripts/scale_curve.sh "$out/scale" 3 > "$out/scale_curve_log.txt" 2>&1; cat "$out/scale/scale_curve.txt" | tee -a "$out/summary.txt"
  ;;
*) echo "unknown step $step"; exit 2 ;;
esac
cat "$out/summary.txt"
