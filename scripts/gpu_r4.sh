#!/usr/bin/env bash
# Round 4's GPU calls, one parameterised script:   gpurun -- 'bash scripts/gpu_r4.sh <step> [args]'   (writes gpurun_out/r4_<step>/*)
#   sel      the selection commit kernel (k_commit_run<true>): the differential suites pinned to it, then same-box A/B against the batch kernel
#   trace    the selection kernel's per-phase cycle trace (configs 3 and 4)
#   profile  rocprofv3: kernel stats, HBM bytes of the matrix launches, SQ counters of the commit kernels (scripts/summarize_profile.py r4_profile profiles/round4)
#   evict    1M x 50k allocate + backfill + preempt: host timeline of the evict action, session-load phases, kernel stats
#   hunt     fresh-seed differential hunt under the three commit kernels
#   wide     host-port masks of several words: their differential cases on the device, then the default bench (K1 gained a branch)
#   bf       backfill rows in bulk (selection kernel): every -m gpu case under the selection kernel, then configs 3, 5 and survey nodes
#   pin      configs 5 and 2 pinned to the selection / the batch kernel beside the per-round choice (what the policy costs or gains)
#   suite    the whole -m gpu suite
#   bench    the default bench line and the variants
set -uo pipefail
cd "$(dirname "$0")/.."
step="${1:-suite}"; shift || true
out="gpurun_out/r4_${step}"
mkdir -p "$out"
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print(d.get('ms_per_step'), d.get('verified_bind_set_equals_oracle'), d.get('kernel_ms_per_step'))" 2>/dev/null; }
bench_ab() {   # name, env assignments..., -- bench args
  local name="$1"; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  timeout 300 env "${envs[@]}" KB_K5_STATS=1 python bench.py --no-cpu-baseline "$@" > "$out/bench_${name}.json" 2> "$out/bench_${name}.err"
  echo "bench $name rc=$? $(ms "$out/bench_${name}.json")" | tee -a "$out/summary.txt"
  grep -h "kb select\|kb K5\] rounds on" "$out/bench_${name}.err" | tee -a "$out/summary.txt"
}
case "$step" in
sel)
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_adversarial.py tests/test_gpu_regressions.py tests/test_gpu_preempt.py tests/test_gpu_interpod.py \
    -q -m gpu -p no:cacheprovider -k "select" --maxfail=10 > "$out/pytest_select.txt" 2>&1; echo "differential suites on the selection kernel rc=$? $(tail -1 "$out/pytest_select.txt")" | tee -a "$out/summary.txt"
  for cfg in "3" "4" "3 --survey-nodes"; do
    tag="c${cfg// --survey-nodes/survey}"; tag="${tag// /}"
    bench_ab "${tag}_pinbatch" KB_COMMIT_KERNEL=batch -- --config ${cfg} --steps 5 --warmup 2 --verify
    bench_ab "${tag}_pinsel" KB_COMMIT_KERNEL=select -- --config ${cfg} --steps 5 --warmup 2 --verify
    if [ -f kube-batch_amd/libkbengine_trace.so ]; then   # make EXTRA=-DKB_K9_TRACE OUT=../libkbengine_trace.so: wave 0's cycles per phase
      KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_COMMIT_KERNEL=select KB_K5_STATS=1 python bench.py --no-cpu-baseline --config ${cfg} --steps 2 --warmup 1 \
        > "$out/trace_${tag}.json" 2> "$out/trace_${tag}.err"
      echo "== trace ${tag}" | tee -a "$out/summary.txt"; grep -h "kb K5 trace\|kb K5\] rounds [0-9]" "$out/trace_${tag}.err" | tee -a "$out/summary.txt"
    fi
  done
  ;;
wide)
  timeout 900 python -m pytest tests/test_gpu_wideports.py -q -m gpu -p no:cacheprovider --maxfail=10 > "$out/pytest_wide.txt" 2>&1
  echo "host-port masks of several words rc=$? $(tail -1 "$out/pytest_wide.txt")" | tee -a "$out/summary.txt"
  timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_preempt.py -q -m gpu -p no:cacheprovider --maxfail=10 -k "not select and not run" > "$out/pytest_ports.txt" 2>&1
  echo "fuzz + preempt suites (one-word host ports among them) rc=$? $(tail -1 "$out/pytest_ports.txt")" | tee -a "$out/summary.txt"
  bench_ab "c3" -- --config 3 --steps 5 --warmup 2 --verify
  ;;
trace)   # the selection kernel's per-phase trace only (make EXTRA=-DKB_K9_TRACE OUT=../libkbengine_trace.so)
  for cfg in "3" "4"; do
    KB_ENGINE_LIB=$PWD/kube-batch_amd/libkbengine_trace.so KB_COMMIT_KERNEL=select KB_K5_STATS=1 python bench.py --no-cpu-baseline --config ${cfg} --steps 2 --warmup 1 \
      > "$out/trace_c${cfg}.json" 2> "$out/trace_c${cfg}.err"
    echo "== trace c${cfg} $(ms "$out/trace_c${cfg}.json")" | tee -a "$out/summary.txt"; grep -h "kb K5 trace\|kb K5\] rounds [0-9]\|kb select" "$out/trace_c${cfg}.err" | tee -a "$out/summary.txt"
  done
  ;;
profile)   # rocprofv3 evidence of the default bench command: kernel stats, HBM bytes of the matrix launches, SQ counters of the commit kernels
  export TMPDIR=/tmp
  CMD="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
  P="$PWD/$out"
  ( cd /tmp
    rocprofv3 --kernel-trace --stats -f csv -d "$P/trace" -o bench -- $CMD > "$P/bench_trace.log" 2>&1
    rocprofv3 --pmc FETCH_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_fetch" -o bench -- $CMD > "$P/bench_pmc_fetch.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE -f csv --kernel-include-regex "k_matrix|k_expand" -d "$P/pmc_write" -o bench -- $CMD > "$P/bench_pmc_write.log" 2>&1
    # the commit kernels (one workgroup on one CU): waves, busy / wave cycles and how they split, instruction mix, LDS
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv --kernel-include-regex "k_commit" \
      -d "$P/pmc_commit_a" -o bench -- $CMD > "$P/bench_pmc_commit_a.log" 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv --kernel-include-regex "k_commit" \
      -d "$P/pmc_commit_b" -o bench -- $CMD > "$P/bench_pmc_commit_b.log" 2>&1
  )
  find "$out" -name "*.csv" | head -20 | tee -a "$out/summary.txt"
  for f in trace pmc_fetch pmc_write pmc_commit_a pmc_commit_b; do echo "$f: $(tail -1 "$out/bench_${f/trace/trace}.log" 2>/dev/null | cut -c1-160)" >> "$out/summary.txt"; done
  ;;
evict)   # 1M x 50k with its third action: host timeline of the evict action (KB_EVICT_TRACE), kb_session_load's phases (KB_LOAD_TRACE), kernel stats
  KB_EVICT_TRACE=1 KB_LOAD_TRACE=1 timeout 600 python bench.py --config 5 --preempt --steps 2 --warmup 1 --no-cpu-baseline --verify > "$out/bench_config5_preempt.json" 2> "$out/bench_config5_preempt.err"
  echo "config 5 three actions rc=$? $(ms "$out/bench_config5_preempt.json")" | tee -a "$out/summary.txt"
  grep -h "kb evict\|kb load" "$out/bench_config5_preempt.err" | tail -24 | tee -a "$out/summary.txt"
  export TMPDIR=/tmp; P="$PWD/$out"
  ( cd /tmp; rocprofv3 --kernel-trace --stats -f csv -d "$P/trace" -o bench -- python $OLDPWD/bench.py --config 5 --preempt --steps 1 --warmup 1 --no-cpu-baseline > "$P/bench_trace.log" 2>&1 )
  head -14 "$out/trace/bench_kernel_stats.csv" | cut -c1-150 | tee -a "$out/summary.txt"
  ;;
sleep)   # the selection kernel's polls: how long a waiting wave sleeps (KB_SEL_SLEEP=prep,dk), same box, config 3 pinned to the kernel
  for sl in ${SLEEPS:-1,1 0,0 0,1 1,0}; do
    bench_ab "c3_sleep_${sl/,/_}" KB_COMMIT_KERNEL=select KB_SEL_SLEEP=$sl -- --config 3 --steps 5 --warmup 2 --verify
  done
  bench_ab "survey_sleep_1_1" KB_COMMIT_KERNEL=select KB_SEL_SLEEP=1,1 -- --config 3 --survey-nodes --steps 5 --warmup 2 --verify
  bench_ab "survey_sleep_16_1" KB_COMMIT_KERNEL=select KB_SEL_SLEEP=16,1 -- --config 3 --survey-nodes --steps 5 --warmup 2 --verify
  ;;
c4)   # config 4 (R = 16: scalar dimensions) pinned to the selection kernel, with the R = 16 parity cases
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_regressions.py -q -m gpu -p no:cacheprovider -k "select" --maxfail=10 > "$out/pytest_select.txt" 2>&1
  echo "parity / fuzz / regressions on the selection kernel rc=$? $(tail -1 "$out/pytest_select.txt")" | tee -a "$out/summary.txt"
  bench_ab c4_pinsel KB_COMMIT_KERNEL=select -- --config 4 --steps 5 --warmup 2 --verify
  bench_ab c4_default -- --config 4 --steps 5 --warmup 2 --verify
  bench_ab c3_pinsel KB_COMMIT_KERNEL=select -- --config 3 --steps 5 --warmup 2 --verify
  ;;
hunt)   # fresh seeds beyond the committed suite, engine vs oracle under the three commit kernels (scripts/gpu_hunt.py; KB_HUNT_OFFSET shifts the seeds)
  KB_HUNT_OFFSET=${KB_HUNT_OFFSET:-40000} timeout 1200 python scripts/gpu_hunt.py ${1:-300} ${2:-900} ${3:-400} > "$out/hunt.txt" 2>&1; echo "hunt rc=$? $(tail -2 "$out/hunt.txt" | tr '\n' ' ')" | tee -a "$out/summary.txt"
  ;;
bf)
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "select" --maxfail=10 > "$out/pytest_select.txt" 2>&1; echo "every case on the selection kernel rc=$? $(tail -1 "$out/pytest_select.txt")" | tee -a "$out/summary.txt"
  bench_ab c3 -- --config 3 --steps 5 --warmup 2 --verify
  bench_ab c5 -- --config 5 --steps 3 --warmup 1 --verify
  bench_ab survey -- --config 3 --survey-nodes --steps 5 --warmup 2 --verify
  ;;
last)   # the lines the bf step did not measure
  python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$? $(ms "$out/bench_default.json")" | tee -a "$out/summary.txt"
  bench_ab config4 -- --config 4 --steps 5 --warmup 2 --verify
  bench_ab config2 -- --config 2 --steps 10 --warmup 3 --verify
  bench_ab config5_preempt -- --config 5 --preempt --steps 2 --warmup 1 --verify
  ;;
pin)
  for cfg in 5 2; do
    st=3; [ "$cfg" = 2 ] && st=10
    bench_ab "c${cfg}_auto" -- --config $cfg --steps $st --warmup 1 --verify
    bench_ab "c${cfg}_pinsel" KB_COMMIT_KERNEL=select -- --config $cfg --steps $st --warmup 1 --verify
    bench_ab "c${cfg}_pinbatch" KB_COMMIT_KERNEL=batch -- --config $cfg --steps $st --warmup 1 --verify
  done
  ;;
suite)
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$? $(tail -1 "$out/pytest_gpu.txt")" | tee -a "$out/summary.txt"
  ;;
bench)
  python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$? $(ms "$out/bench_default.json")" | tee -a "$out/summary.txt"
  bench_ab survey_nodes -- --config 3 --survey-nodes --steps 5 --warmup 2 --verify
  bench_ab config4 -- --config 4 --steps 5 --warmup 2 --verify
  bench_ab config2 -- --config 2 --steps 10 --warmup 3 --verify
  bench_ab config5 -- --config 5 --steps 3 --warmup 1 --verify
  bench_ab config5_preempt -- --config 5 --preempt --steps 2 --warmup 1 --verify
  ;;
*) echo "unknown step $step"; exit 2 ;;
esac
cat "$out/summary.txt"
