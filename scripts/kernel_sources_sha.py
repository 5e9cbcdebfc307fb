#!/usr/bin/env python
"""sha256 over the DEVICE sources of the engine (kube-batch_amd/csrc/*.hip, *.hpp, *.h and the Makefile's flags), in name order.
A rocprofv3 summary committed under profiles/ carries the value of the tree it was measured on (scripts/gpu_r5.sh writes it beside the
CSVs, on the GPU box); bench.py recomputes it and refuses to quote counters of other kernels (null + the reason in the line)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha(root=ROOT):
    h = hashlib.sha256()
    src = os.path.join(root, "kube-batch_amd", "csrc")
    files = sorted(f for pat in ("*.hip", "*.hpp", "*.h", "Makefile") for f in glob.glob(os.path.join(src, pat)))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_sha())
