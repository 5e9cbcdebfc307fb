#!/usr/bin/env bash
# Device ISA of the three kernel sources at a commit against the working tree: the kernels of the default path must compile to the code that
# passed the last GPU suite when a change was meant to touch the host side (or code behind a switch) only.  No device needed.
#   scripts/isa_diff.sh <commit>      prints, per source file, the differing ISA lines that are not kernel-argument offsets / sizes / symbol names
# (kb_commit_batch.hip: k_commit_batch became a template in round 3; its <false> instantiation is compared with the old kernel's body).
set -euo pipefail
cd "$(dirname "$0")/.."
ref=${1:?commit}
tmp=$(mktemp -d)
mkdir -p "$tmp/old" "$tmp/a" "$tmp/b"
git archive "$ref" kube-batch_amd/csrc include | tar -x -C "$tmp/old"
fl="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --cuda-device-only -S"
for f in kb_kernels kb_commit kb_commit_batch; do
  (cd "$tmp/old/kube-batch_amd/csrc" && /opt/rocm/bin/hipcc $fl -o "$tmp/a/$f.s" $f.hip 2>/dev/null)
  (cd kube-batch_amd/csrc && /opt/rocm/bin/hipcc $fl -o "$tmp/b/$f.s" $f.hip 2>/dev/null)
done
norm() {   # instructions only: no directives / metadata, kernel-argument offsets and the template's mangled name masked
  grep -E '^\s+[a-z_0-9]+(\s|$)' "$1" | grep -vE '^\s+\.' | sed -E 's/(s_load_dword[x0-9]* [^,]+, s\[[0-9:]+\], )0x[0-9a-f]+/\1OFF/; s/(s_add_u32 s[0-9]+, s[0-9]+, )0x[0-9a-f]+/\1OFF/'
}
rc=0
for f in kb_kernels kb_commit; do
  n=$(diff <(norm "$tmp/a/$f.s") <(norm "$tmp/b/$f.s") | grep -c '^[<>]' || true)
  echo "$f.hip: $(norm "$tmp/b/$f.s" | wc -l) instructions, $n differing"
  [ "$n" -eq 0 ] || rc=1
done
body() { awk -v pat="$2" '$0 ~ pat {on=1} on {print} on && /s_endpgm/ {exit}' "$1"; }
body "$tmp/a/kb_commit_batch.s" '^_Z14k_commit_batch(ILb0EEv)?10K7KernArgs:' > "$tmp/a/cb.s"
body "$tmp/b/kb_commit_batch.s" '^_Z14k_commit_batch(ILb0EEv)?10K7KernArgs:' > "$tmp/b/cb.s"
n=$(diff <(norm "$tmp/a/cb.s") <(norm "$tmp/b/cb.s") | grep -c '^[<>]' || true)
echo "kb_commit_batch.hip, default instantiation: $(norm "$tmp/b/cb.s" | wc -l) instructions, $n differing"
[ "$n" -eq 0 ] || rc=1
rm -rf "$tmp"
exit $rc
