#!/usr/bin/env bash
# First GPU call of round 4: what round 3 wrote or fixed WITHOUT a device after its GPU budget was spent, against the real kernels, before anything
# else is built on it.  Usage on the GPU box:  bash scripts/first_gpu_call_r4.sh      (about 18 minutes; writes gpurun_out/r4_first/*)
#   1. the -m gpu suite (last device run: profiles/round3/call31; since then only host-side changes: kb_session_load resets the second stream's
#      buffer sizes, the overlapped rounds' list tag is the chain tag, an allocate waits for the copies kb_session_reset left queued before its
#      first launch, the order machine builds its heaps at their first pop)
#   2. the emulator-born cases of late round 3 with the PRODUCT library (KB_EMU_LIB names the library tests/test_emu_engine_cpu.py loads):
#      one engine through sessions of growing size (the reload fix), the launch-path variants
#   3. preempt / reclaim with inter-pod (anti)affinity terms behind KB_EVICT_INTERPOD=1: FIRST device run (every case sets the switch itself after
#      checking the refusal without it).  Green here => make it the default (kb_preempt.cpp: evict_interpod_enabled), move the cases into
#      tests/test_gpu_interpod.py, drop the line from DESIGN section 2
#   4. folded repair behind KB_FOLD_REPAIR=1: FIRST device run of k_commit_batch<true> (the batch commit launch repairs an overlapped round's
#      candidate lists in its own idle workgroups, kb_repair.hpp): the launch-path variants, the overlapped-list cases, the fuzz and full-size
#      suites with the switch on, then the same-box A/B on configs 3, 4, 5.  Green and not slower => default on, k_repair becomes a wrapper
#      around kb_repair_row (one copy of the text), DESIGN section 9 item 2
#   5. proportion's water-fill as a launch behind KB_DEVICE_WATERFILL=1: FIRST device run of k_waterfill (kb_waterfill.hip): the adversarial,
#      fuzz, regression and preempt suites with the switch on (they compare `deserved` and the shares with the oracle bit for bit), the
#      host-against-device cases of tests/test_emu_engine_cpu.py on the product library.  Green => default on, the host loop in
#      kb_session.cpp goes (kb_waterfill.hpp stays the one text), DESIGN section 9 item 4
#   6. the default bench line and the two variants, for the record of what the round starts from
set -uo pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/r4_first
mkdir -p "$out"
lib="$PWD/kube-batch_amd/libkbengine.so"
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
KB_EMU_LIB="$lib" timeout 600 python -m pytest tests/test_emu_engine_cpu.py -q -p no:cacheprovider \
  -k "growing_size or launch_path_variants" > "$out/pytest_emu_cases_on_device.txt" 2>&1   # (test_overlapped_candidate_lists_are_repaired needs the emulated launch's counters and negative control: CPU only)
echo "emulator-born cases on the device rc=$?" | tee -a "$out/summary.txt"
KB_EMU_LIB="$lib" timeout 900 python -m pytest tests/test_emu_engine_cpu.py -q -p no:cacheprovider \
  -k "evict_actions_with_interpod_terms" > "$out/pytest_evict_interpod_on_device.txt" 2>&1
echo "evict actions with inter-pod terms on the device rc=$?" | tee -a "$out/summary.txt"
KB_EVICT_INTERPOD=1 KB_EMU_LIB="$lib" timeout 600 python - > "$out/hunt_evict_interpod_on_device.txt" 2>&1 <<'EOF'
# the hunt's engine-against-oracle leg on fresh seeds, product library
import importlib, os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import pytest
engine = importlib.import_module("kube-batch_amd.engine")
import oracle, test_gpu_preempt as gp, test_interpod_oracle_cpu as ipo
oracle.build()
bad = ran = 0
for seed in range(3000, 3150):
    try:
        cfg, snap, order = ipo.interpod_evict_case(seed)
    except Exception:
        continue
    if snap.interpod is None:
        continue
    try:
        gp._run_both(oracle, cfg, snap, order, seed)
        ran += 1
    except pytest.skip.Exception:
        pass
    except BaseException as err:
        bad += 1
        print(f"seed {seed} {order}: {type(err).__name__}: {str(err)[:300]}", flush=True)
print(f"{ran} comparisons, {bad} divergences")
sys.exit(1 if bad else 0)
EOF
echo "inter-pod evict hunt on the device rc=$?" | tee -a "$out/summary.txt"
KB_FOLD_REPAIR=1 timeout 900 python -m pytest tests/test_gpu_regressions.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_adversarial.py -x -q -m gpu -p no:cacheprovider \
  > "$out/pytest_gpu_fold.txt" 2>&1; echo "folded repair: gpu suites rc=$?" | tee -a "$out/summary.txt"
KB_FOLD_REPAIR=1 KB_COMMIT_KERNEL=batch KB_EMU_LIB="$lib" timeout 600 python -m pytest tests/test_emu_engine_cpu.py -q -p no:cacheprovider \
  -k "growing_size or launch_path_variants or fuzz" > "$out/pytest_emu_cases_fold_on_device.txt" 2>&1
echo "folded repair: emulator-born cases on the device, batch kernel pinned rc=$?" | tee -a "$out/summary.txt"
for cfg in 3 4 5; do
  for fold in 0 1; do
    KB_FOLD_REPAIR=$fold KB_K5_STATS=1 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --verify > "$out/bench_config${cfg}_fold${fold}.json" 2> "$out/bench_config${cfg}_fold${fold}.err"
    echo "config $cfg fold $fold rc=$? $(python -c "import json,sys; d=json.loads(open('$out/bench_config${cfg}_fold${fold}.json').read().strip().splitlines()[-1]); print(d.get('ms_per_step'), d.get('verified_bind_set_equals_oracle'))" 2>/dev/null)" | tee -a "$out/summary.txt"
  done
done
KB_DEVICE_WATERFILL=1 timeout 900 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_fuzz.py tests/test_gpu_regressions.py tests/test_gpu_preempt.py -x -q -m gpu -p no:cacheprovider \
  > "$out/pytest_gpu_waterfill.txt" 2>&1; echo "device water-fill: gpu suites rc=$?" | tee -a "$out/summary.txt"
KB_EMU_LIB="$lib" timeout 600 python -m pytest tests/test_emu_engine_cpu.py -q -p no:cacheprovider -k "device_waterfill" \
  > "$out/pytest_waterfill_cases_on_device.txt" 2>&1; echo "device water-fill: host-against-device cases on the device rc=$?" | tee -a "$out/summary.txt"
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_config4.json" 2> "$out/bench_config4.err"; echo "bench config 4 rc=$?" | tee -a "$out/summary.txt"
python bench.py --config 5 --preempt --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_config5_three_actions.json" 2> "$out/bench_config5.err"; echo "bench config 5 three actions rc=$?" | tee -a "$out/summary.txt"
# 7. the CU-masked second stream (KB_STREAM_B_CUMASK): same-box A/B on config 3 — which bit is which CU of which XCD is unknown, so two readings:
#    "the first 32 bits are XCD 0" and "bit i is CU i/8 of XCD i%8"
for m in none 00000000,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff fefefefe,fefefefe,fefefefe,fefefefe,fefefefe,fefefefe,fefefefe,fefefefe; do
  if [ "$m" = none ]; then unset KB_STREAM_B_CUMASK; else export KB_STREAM_B_CUMASK=$m; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_cumask_${m:0:8}.json" 2> "$out/bench_cumask_${m:0:8}.err"
  echo "cumask $m rc=$? $(python -c "import json,sys; d=json.loads(open('$out/bench_cumask_${m:0:8}.json').read().strip().splitlines()[-1]); print(d.get('ms_per_step'), d.get('verified_bind_set_equals_oracle'))" 2>/dev/null)" | tee -a "$out/summary.txt"
done
unset KB_STREAM_B_CUMASK
cat "$out/summary.txt"
