#!/usr/bin/env bash
# First GPU call of the next round: everything that was written or fixed WITHOUT a device at the end of round 2 (the GPU budget was spent),
# against the real kernels, before anything else is built on it.  Usage on the GPU box: bash scripts/first_gpu_call_r3.sh
# Writes gpurun_out/r3_first/*.txt; copy what should be judged into profiles/round3/.
set -uo pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/r3_first
mkdir -p "$out"
# 1. the -m gpu suite on the host-side changes of late round 2 (session build, evict machine, Idle key masks, reset of the masks,
#    missing-queue guard in k_finalize_jobs)
python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.txt" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
# 2. the regression cases the emulated hunts produced (mixed action orders on adversarial snapshots, reset after evict actions,
#    launch-path variants) on the device: same test functions, product library
KB_EMU_USE_REAL=1 python -m pytest tests/test_emu_engine_cpu.py -q \
  -k "scalar_keys_created or mixed_action_orders or session_reset_after or launch_path_variants or missing_queue" > "$out/pytest_emu_cases_on_device.txt" 2>&1
echo "emulator-born cases on the device rc=$?" | tee -a "$out/summary.txt"
# 3. preempt with preferred node-affinity terms behind its switch: first device run; green here => make it the default
KB_EMU_USE_REAL=1 python -m pytest tests/test_emu_engine_cpu.py -q -k "preferred_node_affinity_behind" > "$out/pytest_preempt_affinity.txt" 2>&1
echo "preempt + node affinity (each case sets KB_PREEMPT_NODE_AFFINITY=1 itself after checking the refusal without it) rc=$?" | tee -a "$out/summary.txt"
# 4. the three actions of BASELINE configs[4] after the host work on the evict machine (not re-measured since)
python scripts/time_preempt.py 5 1.0 > "$out/time_preempt_config5.txt" 2>&1; echo "time_preempt rc=$?" | tee -a "$out/summary.txt"
python bench.py --config 5 --scale 0.1 --preempt --steps 2 --warmup 1 --verify --no-cpu-baseline > "$out/bench_config5_scale0.1_three_actions_verified.json" 2>> "$out/bench_config5_three_actions.err"
python bench.py --config 5 --preempt --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_config5_three_actions.json" 2> "$out/bench_config5_three_actions.err"; echo "bench config 5 with preempt rc=$?" | tee -a "$out/summary.txt"
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
