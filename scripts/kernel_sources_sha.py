#!/usr/bin/env python
"""sha256 over the DEVICE sources of the engine, two granularities:

  kernel_sources_sha()      every file some translation unit below reaches (the four .hip files, the headers they include, the C ABI header, the
                            Makefile), in name order: the device sources as a whole.  (Until round 6's call 29 it took every *.hpp of csrc/, the
                            host-only ones too: a change to the host's order machine — kb_host.hpp — then voided a stamp no kernel depends on.)
  kernel_tu_sha(root, tu)   ONE translation unit: the .hip file, every header it reaches through `#include "..."` (transitively; the
                            C ABI header included) and the Makefile — what decides the ISA of the kernels that file defines

A rocprofv3 summary committed under profiles/ carries the values of the tree it was measured on (scripts/gpu_r5.sh writes them beside the
CSVs, on the GPU box: kernel_sources.sha256, kernel_tu.sha256); bench.py recomputes the translation unit's and refuses to quote counters
of kernels compiled from other sources (null + the reason in the line).  Per translation unit since the round's last GPU call: a change
to k_probe (kb_kernels.hip) does not touch the commit kernels' ISA (scripts/isa_diff.sh: 0 differing instructions) and must not void
their counters, while it does void the matrix kernels' — they share its file — until those are measured again.

  python scripts/kernel_sources_sha.py          the whole-tree value
  python scripts/kernel_sources_sha.py --tu     one line per translation unit: "<sha256>  <file>"
"""
import hashlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUS = ("kb_kernels.hip", "kb_commit_sel.hip", "kb_commit.hip", "kb_waterfill.hip")


def kernel_sources_sha(root=ROOT):
    h = hashlib.sha256()
    for rel in sorted({rel for tu in TUS for rel in tu_files(root, tu)}):
        h.update(rel.replace(os.sep, "/").encode() + b"\0")
        h.update(open(os.path.join(root, rel), "rb").read())
    return h.hexdigest()


def tu_files(root, tu):
    """the translation unit's own sources, repo-relative, in name order (system headers are the image's, not the tree's)"""
    src = os.path.join(root, "kube-batch_amd", "csrc")
    seen, todo = set(), [os.path.join(src, tu)]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen:
            continue
        seen.add(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f, encoding="utf-8", errors="replace").read(), flags=re.M):
            p = os.path.normpath(os.path.join(os.path.dirname(f), inc))
            if os.path.exists(p):
                todo.append(p)
    seen.add(os.path.join(src, "Makefile"))
    return sorted(os.path.relpath(f, root) for f in seen)


def kernel_tu_sha(root, tu):
    h = hashlib.sha256()
    for rel in tu_files(root, tu):
        h.update(rel.replace(os.sep, "/").encode() + b"\0")
        h.update(open(os.path.join(root, rel), "rb").read())
    return h.hexdigest()


def read_tu_stamp(path):
    """{file: sha256} of a kernel_tu.sha256 written by --tu (lines "<sha256>  <file>"; '#' comments)"""
    out = {}
    for line in open(path):
        parts = line.split("#", 1)[0].split()
        if len(parts) >= 2:
            out[parts[1]] = parts[0]
    return out


if __name__ == "__main__":
    if "--tu" in sys.argv[1:]:
        for tu in TUS:
            print(f"{kernel_tu_sha(ROOT, tu)}  {tu}")
    else:
        print(kernel_sources_sha())
