#!/usr/bin/env bash
# AddressSanitizer + UndefinedBehaviorSanitizer pass over the CPU-side C/C++ of the test infrastructure and the engine's host code:
# the oracle (oracle/kb_oracle.c), kube-batch_amd/csrc/kb_order.cpp (order machine) and kb_session.cpp + kb_preempt.cpp (policy
# compiler, session build, evict actions) behind tests/host_harness, and the WHOLE host side of the engine (kb_engine.cpp included) on the
# emulated device of tests/host_harness/device_emu.cpp.  The HIP kernels need a device and are not covered.
# Usage: scripts/sanitize_cpu.sh   (prints pytest's summary; any report aborts).
set -euo pipefail
cd "$(dirname "$0")/.."
out=$(mktemp -d)
san="-O1 -g -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer"
gcc -std=c11 $san -o "$out/libkboracle.so" oracle/kb_oracle.c -lm -lpthread
g++ -std=c++17 $san -o "$out/liborderharness.so" tests/host_harness/order_harness.cpp kube-batch_amd/csrc/kb_order.cpp
g++ -std=c++17 $san -o "$out/libevictharness.so" tests/host_harness/evict_harness.cpp kube-batch_amd/csrc/kb_session.cpp kube-batch_amd/csrc/kb_preempt.cpp
# libstdc++ is preloaded too: ASan resolves its __cxa_throw interceptor at start-up, and python itself does not link the C++ runtime
# (the harnesses throw EngineError where the engine answers KB_E_UNSUPPORTED)
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" \
KB_ORACLE_LIB="$out/libkboracle.so" KB_ORDER_HARNESS_LIB="$out/liborderharness.so" KB_EVICT_HARNESS_LIB="$out/libevictharness.so" \
  python -m pytest tests/test_oracle_kat.py tests/test_oracle_independent.py tests/test_pyref_vs_oracle.py tests/test_host_order_cpu.py tests/test_host_evict_cpu.py \
    tests/test_interpod_oracle_cpu.py tests/test_manifests_cpu.py \
    -x -q -p no:cacheprovider "$@"
# the complete C ABI on the emulated device: kb_engine.cpp's round protocol, session load / reset, evict actions, kb_round_* (about 3 minutes)
g++ -std=c++17 $san -Itests/host_harness/hip_mock -o "$out/libkbengine_emu.so" kube-batch_amd/csrc/kb_engine.cpp kube-batch_amd/csrc/kb_load.cpp kube-batch_amd/csrc/kb_rounds.cpp kube-batch_amd/csrc/kb_evict.cpp kube-batch_amd/csrc/kb_matrix.cpp kube-batch_amd/csrc/kb_session.cpp \
  kube-batch_amd/csrc/kb_order.cpp kube-batch_amd/csrc/kb_preempt.cpp tests/host_harness/device_emu.cpp tests/host_harness/hip_mock/hip_mock.cpp
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" KB_EMU_LIB="$out/libkbengine_emu.so" \
  python -m pytest tests/test_emu_engine_cpu.py -x -q -p no:cacheprovider -k "not two_gloo" "$@"
# ThreadSanitizer over the oracle's worker pool (the cpu_baseline leg's 16-way fan-out)
gcc -std=c11 -O1 -g -fPIC -shared -ffp-contract=off -fsanitize=thread -o "$out/libkboracle_tsan.so" oracle/kb_oracle.c -lm -lpthread
TSAN_OPTIONS="report_signal_unsafe=0 exitcode=66" LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" KB_ORACLE_LIB="$out/libkboracle_tsan.so" \
  python -m pytest tests/test_oracle_kat.py -q -k threaded -p no:cacheprovider
# ThreadSanitizer over kb_waterfill.hip's kernel text run by 256 host threads with a real barrier (tests/host_harness/waterfill_kernel_harness.cpp):
# an access pair the kernel's barriers do not order is a reported race (the first run reported one: lane 0's read of the stop word beside the
# other lanes' — a speculated load; lane 0 now keeps its own copy)
g++ -std=c++17 -O1 -g -fPIC -shared -ffp-contract=off -fsanitize=thread -pthread -Itests/host_harness/hip_mock -o "$out/libwaterfillkernel_tsan.so" \
  tests/host_harness/waterfill_kernel_harness.cpp
TSAN_OPTIONS="report_signal_unsafe=0 exitcode=66" LD_PRELOAD="$(gcc -print-file-name=libtsan.so) $(g++ -print-file-name=libstdc++.so.6)" \
KB_WATERFILL_HARNESS_LIB="$out/libwaterfillkernel_tsan.so" python -m pytest tests/test_waterfill_kernel_cpu.py -x -q -p no:cacheprovider
# the whole emulated suite with every run of identical rows committed by one selection (the emulated commit launch only)
KB_EMU_RUN_SELECT=1 python -m pytest tests/test_emu_engine_cpu.py -x -q -p no:cacheprovider -n 8 -k "not two_gloo and not fullsize"
# the whole emulated suite with proportion's water-fill on the host loop (the launch is the default)
KB_DEVICE_WATERFILL=0 python -m pytest tests/test_emu_engine_cpu.py -x -q -p no:cacheprovider -n 8 -k "not two_gloo and not fullsize"
# ThreadSanitizer over the host <-> device handshake: the emulated streams run asynchronously (KB_EMU_ASYNC=1: a worker thread per stream with
# HIP's ordering rules, tests/host_harness/hip_mock), so a staging half, a mailbox word or a result the host touches before the device is done
# with it is a reported race.  Chained / unchained rounds, pinned mailbox / synchronous rounds, direct window, both commit-kernel pins.
emu_src="kube-batch_amd/csrc/kb_engine.cpp kube-batch_amd/csrc/kb_load.cpp kube-batch_amd/csrc/kb_rounds.cpp kube-batch_amd/csrc/kb_evict.cpp kube-batch_amd/csrc/kb_matrix.cpp kube-batch_amd/csrc/kb_session.cpp kube-batch_amd/csrc/kb_order.cpp kube-batch_amd/csrc/kb_preempt.cpp tests/host_harness/hip_mock/hip_mock.cpp"
tsan="-std=c++17 -O1 -g -fPIC -shared -ffp-contract=off -fsanitize=thread -pthread -Itests/host_harness/hip_mock"
g++ $tsan -o "$out/libkbengine_emu_tsan.so" $emu_src tests/host_harness/device_emu.cpp
# KB_OVERLAP=0: an overlapped round's matrix launch (second stream) reads node state the predecessor's commit kernel (first stream) is writing
# — on purpose: k_repair overrides whatever it saw of those nodes — and ThreadSanitizer would report exactly that access
# pair on the emulated device.  The handshake this pass is about (staging halves, mailbox, chain word) is the same on both paths; the
# overlapped path's own protocol (tags, stale lists, repair) runs in the ASan / UBSan pass above with poisoned stale entries.
export KB_OVERLAP=0
KB_EMU_ASYNC=1 TSAN_OPTIONS="report_signal_unsafe=0 exitcode=66" LD_PRELOAD="$(gcc -print-file-name=libtsan.so) $(g++ -print-file-name=libstdc++.so.6)" \
KB_EMU_LIB="$out/libkbengine_emu_tsan.so" python -m pytest tests/test_emu_engine_cpu.py -x -q -p no:cacheprovider -k "variants or fuzz or reference_allocate or config2"
unset KB_OVERLAP
# the overlapped path with truly concurrent emulated streams (no sanitizer: the read / write pair above is intended): the repair launch's wait
# for its lists, the tags, the second stream's buffers; the whole emulated-engine suite must still equal the oracle
KB_EMU_ASYNC=1 python -m pytest tests/test_emu_engine_cpu.py -x -q -p no:cacheprovider -k "not two_gloo"
# the same with the relative timing of the two streams swept: every task of the first / the second / both streams starts up to 2 ms late
# (hip_mock: KB_EMU_JITTER_US, KB_EMU_JITTER_STREAMS) — two idle host threads otherwise always meet in the same order
for streams in 1 2 3; do
  KB_EMU_ASYNC=1 KB_EMU_JITTER_US=2000 KB_EMU_JITTER_STREAMS=$streams python -m pytest tests/test_emu_engine_cpu.py -x -q -p no:cacheprovider -n 8 -k "not two_gloo and not fullsize"
done
# negative control: the same build with the commit's mailbox publication weakened from release to relaxed MUST be reported (exit code 66)
sed 's/r.seq, __ATOMIC_RELEASE)/r.seq, __ATOMIC_RELAXED)/' tests/host_harness/device_emu.cpp > tests/host_harness/_device_emu_relaxed.cpp
g++ $tsan -o "$out/libkbengine_emu_neg.so" $emu_src tests/host_harness/_device_emu_relaxed.cpp; rm -f tests/host_harness/_device_emu_relaxed.cpp
set +e
KB_OVERLAP=0 KB_EMU_ASYNC=1 TSAN_OPTIONS="report_signal_unsafe=0 exitcode=66" LD_PRELOAD="$(gcc -print-file-name=libtsan.so) $(g++ -print-file-name=libstdc++.so.6)" \
KB_EMU_LIB="$out/libkbengine_emu_neg.so" python -m pytest tests/test_emu_engine_cpu.py -q -p no:cacheprovider -k "variants and 0" > "$out/neg.log" 2>&1
neg=$?
set -e
if [ "$neg" -ne 66 ]; then echo "negative control: ThreadSanitizer did not report the weakened mailbox publication (exit $neg)"; exit 1; fi
echo "negative control ok: the weakened publication is reported"
# the same host sources as libkbengine.so gets them: ROCm's clang at -O3 (the harness tests above use g++ -O2)
CL=/opt/rocm/lib/llvm/bin/clang++
if [ -x "$CL" ]; then
  fl="-O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math"
  $CL $fl -o "$out/liborderharness_clang.so" tests/host_harness/order_harness.cpp kube-batch_amd/csrc/kb_order.cpp
  $CL $fl -o "$out/libevictharness_clang.so" tests/host_harness/evict_harness.cpp kube-batch_amd/csrc/kb_session.cpp kube-batch_amd/csrc/kb_preempt.cpp
  KB_ORDER_HARNESS_LIB="$out/liborderharness_clang.so" KB_EVICT_HARNESS_LIB="$out/libevictharness_clang.so" \
    python -m pytest tests/test_host_order_cpu.py tests/test_host_evict_cpu.py -x -q -p no:cacheprovider
  # ... and the whole host side of the engine on the emulated device (the library's name is part of one test)
  mkdir -p "$out/clang"
  $CL $fl -pthread -Itests/host_harness/hip_mock -o "$out/clang/libkbengine_emu.so" kube-batch_amd/csrc/kb_engine.cpp kube-batch_amd/csrc/kb_load.cpp kube-batch_amd/csrc/kb_rounds.cpp kube-batch_amd/csrc/kb_evict.cpp kube-batch_amd/csrc/kb_matrix.cpp kube-batch_amd/csrc/kb_session.cpp \
    kube-batch_amd/csrc/kb_order.cpp kube-batch_amd/csrc/kb_preempt.cpp tests/host_harness/device_emu.cpp tests/host_harness/hip_mock/hip_mock.cpp
  KB_EMU_LIB="$out/clang/libkbengine_emu.so" python -m pytest tests/test_emu_engine_cpu.py -x -q -p no:cacheprovider -n 8 -k "not two_gloo"
fi
rm -rf "$out"
