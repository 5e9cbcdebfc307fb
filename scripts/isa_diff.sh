#!/usr/bin/env bash
# Device ISA of the kernel sources at a commit against the working tree: the kernels of the default path must compile to the code that
# passed the last GPU suite when a change was meant to touch the host side (or code behind a switch) only.  No device needed.
#   scripts/isa_diff.sh <commit>      prints, per source file, the differing ISA lines that are not kernel-argument offsets / sizes / symbol names
set -euo pipefail
cd "$(dirname "$0")/.."
ref=${1:?commit}
tmp=$(mktemp -d)
mkdir -p "$tmp/old" "$tmp/a" "$tmp/b"
git archive "$ref" kube-batch_amd/csrc include | tar -x -C "$tmp/old"
fl="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --cuda-device-only -S"
files="kb_kernels kb_commit kb_commit_sel kb_waterfill"
for f in $files; do
  [ -f "$tmp/old/kube-batch_amd/csrc/$f.hip" ] && (cd "$tmp/old/kube-batch_amd/csrc" && /opt/rocm/bin/hipcc $fl -o "$tmp/a/$f.s" $f.hip 2>/dev/null)
  (cd kube-batch_amd/csrc && /opt/rocm/bin/hipcc $fl -o "$tmp/b/$f.s" $f.hip 2>/dev/null)
done
norm() {   # instructions only: no directives / metadata, kernel-argument offsets masked
  grep -E '^\s+[a-z_0-9]+(\s|$)' "$1" | grep -vE '^\s+\.' | sed -E 's/(s_load_dword[x0-9]* [^,]+, s\[[0-9:]+\], )0x[0-9a-f]+/\1OFF/; s/(s_add_u32 s[0-9]+, s[0-9]+, )0x[0-9a-f]+/\1OFF/'
}
rc=0
for f in $files; do
  if [ ! -f "$tmp/a/$f.s" ]; then echo "$f.hip: not in $ref"; continue; fi
  n=$(diff <(norm "$tmp/a/$f.s") <(norm "$tmp/b/$f.s") | grep -c '^[<>]' || true)
  echo "$f.hip: $(norm "$tmp/b/$f.s" | wc -l) instructions, $n differing"
  [ "$n" -eq 0 ] || rc=1
done
rm -rf "$tmp"
exit $rc
