#!/bin/bash
# A/B of K1 variants on ONE box: builds each EXTRA flag set into /tmp/kb_<i>.so and times the full-matrix launch, interleaved.
set -u
cd "$(dirname "$0")/.."
i=0
for x in "$@"; do
  make -C kube-batch_amd/csrc -s -B EXTRA="$x" OUT=/tmp/kb_$i.so 2>&1 | grep -i " error"
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for x in "$@"; do
    echo -n "[$x] "; KB_ENGINE_LIB=/tmp/kb_$i.so python scripts/bench_k1.py ${CFG:-3}
    i=$((i+1))
  done
done
