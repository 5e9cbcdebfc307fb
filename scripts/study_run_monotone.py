#!/usr/bin/env python
"""Feasibility study for DESIGN section 9.2 (no device needed): in the spreading regime a run of identical rows is committed by a serial loop — arg-max,
place, re-score that node, arg-max again.  If every node's key sequence key(n, k) (its score for the run's shape after k placements of it, k = 0, 1, ...)
is NON-INCREASING in k, the loop's winners are the global top-r of all (node, k) keys in (score desc, node asc, k asc) order: a parallel selection.
This script measures on the synthetic clusters how often that premise holds, per shape, from the snapshot's initial node state:
    python scripts/study_run_monotone.py [--config 3] [--scale 0.1] [--survey-nodes] [--depth 16]
It prints, per nodeorder weight set, the share of (shape, node) sequences that are non-increasing over the feasible prefix, and the share of shapes for
which EVERY node's sequence is.  (Result, round 3: the premise fails for most shapes once Balanced is in the score — and is not needed: with the PREFIX MINIMUM of
every node's sequence in its place the selection equals the loop always; tests/run_selection_model.py has the argument and the executable model.)"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kbm = importlib.import_module("kube-batch_amd")


def scores(rc, rm, ac, am, wl, wm, wb):
    """vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/{least_requested,most_requested,balanced_resource_allocation}.go, int64 / float64"""
    ok_c, ok_m = (ac > 0) & (rc <= ac), (am > 0) & (rm <= am)
    lc = np.where(ok_c, (ac - rc) * 10 // np.maximum(ac, 1), 0)
    lm = np.where(ok_m, (am - rm) * 10 // np.maximum(am, 1), 0)
    mc = np.where(ok_c, rc * 10 // np.maximum(ac, 1), 0)
    mm = np.where(ok_m, rm * 10 // np.maximum(am, 1), 0)
    cf = np.where(ac == 0, 1.0, rc / np.maximum(ac, 1).astype(np.float64))
    mf = np.where(am == 0, 1.0, rm / np.maximum(am, 1).astype(np.float64))
    bal = np.where((cf >= 1) | (mf >= 1), 0, ((1 - np.abs(cf - mf)) * 10.0).astype(np.int64))
    return wl * ((lc + lm) // 2) + wm * ((mc + mm) // 2) + wb * bal


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--scale", type=float, default=0.1)
    ap.add_argument("--survey-nodes", action="store_true")
    ap.add_argument("--depth", type=int, default=16)
    args = ap.parse_args()
    params = kbm.snapshot.synth_config(args.config, args.scale)
    if args.survey_nodes:
        params.node_cpu_cores = (16, 32, 64, 96, 128)
        params.node_mem_gib = (64, 128, 256, 512)
    snap = kbm.snapshot.synth(params)
    N, T, R = snap.n_nodes, snap.n_tasks, snap.n_res
    ac, am = np.asarray(snap.node_alloc_cpu, np.int64), np.asarray(snap.node_alloc_mem, np.int64)
    nzc, nzm = np.asarray(snap.node_nz_cpu, np.int64), np.asarray(snap.node_nz_mem, np.int64)
    idle = np.asarray(snap.node_idle, np.float64).reshape(R, N)
    tn = np.stack([np.asarray(snap.task_nz_cpu, np.int64), np.asarray(snap.task_nz_mem, np.int64)], 1)
    init = np.asarray(snap.task_init_resreq, np.float64).reshape(R, T)[:2].T
    shapes, first = np.unique(np.concatenate([tn, init.astype(np.int64)], 1), axis=0, return_index=True)
    print(f"{T} tasks x {N} nodes, {len(shapes)} distinct (non-zero request, InitResreq cpu/mem) shapes, depth {args.depth}")
    for name, (wl, wm, wb) in (("least 1, balanced 1 (default)", (1, 0, 1)), ("least 1 only", (1, 0, 0)), ("most 5, balanced 1 (config 4)", (0, 5, 1))):
        seq_ok = seq_all = shapes_all_ok = 0
        viol_depth = []
        for sh in shapes:
            tc, tm, ic, im = [int(x) for x in sh]
            K = args.depth
            k = np.arange(K, dtype=np.int64)[:, None]
            sc = scores(nzc[None, :] + (k + 1) * tc, nzm[None, :] + (k + 1) * tm, ac[None, :], am[None, :], wl, wm, wb)      # [K][N]
            feas = (idle[0][None, :] - k * ic >= ic) & (idle[1][None, :] - k * im >= im)                                      # the (k+1)-th placement fits
            pair_ok = ~(feas[1:] & (sc[1:] > sc[:-1]))                                                                        # a feasible step that RAISES the key
            node_ok = pair_ok.all(0)
            has_seq = feas[1]                                                                                                 # the node can take at least two
            seq_all += int(has_seq.sum())
            seq_ok += int((node_ok & has_seq).sum())
            shapes_all_ok += bool(node_ok[has_seq].all())
            bad = ~pair_ok
            if bad.any():
                viol_depth.append(int(np.argmax(bad.any(1))) + 1)
        print(f"  {name:32s} non-increasing sequences {seq_ok}/{seq_all} = {seq_ok / max(seq_all, 1):.4f}; shapes with every node's sequence non-increasing "
              f"{shapes_all_ok}/{len(shapes)}; first raising step (median over shapes that have one): {int(np.median(viol_depth)) if viol_depth else '-'}")


if __name__ == "__main__":
    main()
