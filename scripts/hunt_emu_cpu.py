#!/usr/bin/env python
"""One-off differential hunts on the CPU through the WHOLE engine on the emulated device (tests/test_emu_engine_cpu.py), beyond the committed
suite; they look for host-side bugs (kb_engine.cpp), not kernel bugs.  python scripts/hunt_emu_cpu.py MODE first_seed last_seed
  reset    run a mixed action order, kb_session_reset, run it again: journals, evictions and all state identical
  reload   ONE engine per tier layout, hundreds of different sessions through it (kb_session_load every cycle, as the Go action does): == oracle
  sharded  the round-granular entry points (kb_round_*) through dist.ShardedCycle at world size 1, exchanging always / never: == oracle
  wideports  host-port masks of 2..5 words on raw / fuzz / evict snapshots under mixed action orders, reset + second run on some: == oracle
Exit code 1 on any difference.  (allocate + backfill and the evict orders against the oracle: scripts/gpu_hunt.py and
scripts/hunt_evict_cpu.py with KB_HUNT_EMU=1.)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
conf, abi = kbm.conf, kbm.abi
import oracle  # noqa: E402
import rawgen  # noqa: E402
import test_emu_engine_cpu as emu  # noqa: E402
import test_pyref_vs_oracle as cases  # noqa: E402
from test_interpod_oracle_cpu import interpod_case  # noqa: E402

engine.LIB_PATH, engine._LIB = emu.build_emulated_library(), None
CONFS = [cases.CONF_FULL] + cases.EVICT_CONFS


def hunt_reset(lo, hi):
    confs = CONFS
    orders = [["allocate","preempt"],["preempt","allocate","backfill"],["reclaim","allocate","backfill","preempt"],["allocate","backfill","preempt","reclaim"],["preempt"],["reclaim","preempt"]]
    bad=ok=skip=0
    def state(e):
        return [e.binds().copy(), *[x.copy() for x in e.task_state()], *[x.copy() for x in e.node_state()], *[x.copy() for x in e.shares()[:2]], np.array(e.evictions())]
    for seed in range(lo,hi):
        for kind in ("raw","evict"):
            snap = rawgen.raw_snapshot(seed) if kind=="raw" else cases._evict_case(seed)[1]
            ci = seed % len(confs); order = orders[(seed//len(confs))%len(orders)]
            cfg = conf.load_scheduler_conf(confs[ci].format(actions=", ".join(order)))
            try:
                e = engine.Engine(cfg); e.load(snap)
                outs1 = [np.array(e.run([a])) for a in order]; s1 = state(e)
                e.reset()
                outs2 = [np.array(e.run([a])) for a in order]; s2 = state(e)
            except engine.EngineError as err:
                skip+=1; e.close(); continue
            same = all(a.shape==b.shape and np.array_equal(a,b) for a,b in zip(outs1,outs2)) and all(np.array_equal(a,b) for a,b in zip(s1,s2))
            if not same:
                bad+=1; print("RESET DIVERGES", kind, seed, ci, order, flush=True)
            else: ok+=1
            e.close()
    print(f"[{lo},{hi}): {ok} same, {skip} skipped, {bad} differ")
    return bad


def hunt_reload(lo, hi):
    # ONE engine per tier layout, many sessions through it (what the Go action does: one engine per process, kb_session_load every cycle)
    confs = CONFS
    orders = [["allocate","backfill"],["allocate","backfill","preempt"],["reclaim","allocate","backfill","preempt"],["preempt","allocate","backfill"]]
    ok=bad=skip=0
    for ci, ct in enumerate(confs):
        cfg = conf.load_scheduler_conf(ct.format(actions="allocate, backfill, preempt, reclaim"))
        e = engine.Engine(cfg)
        for seed in range(lo,hi):
            rng=np.random.RandomState(seed*7+ci)
            kind = ["raw","evict","alloc","interpod"][rng.randint(4)]
            try:
                snap = {"raw": lambda: rawgen.raw_snapshot(seed), "evict": lambda: cases._evict_case(seed)[1], "alloc": lambda: cases._case(seed)[1], "interpod": lambda: interpod_case(seed)[1]}[kind]()
            except kbm.snapshot.UnsupportedSnapshot:
                continue
            order = orders[rng.randint(len(orders))]
            try:
                o = oracle.Oracle(cfg, snap); o.run(order)
            except RuntimeError:
                skip+=1; continue
            try:
                e.load(snap)
                for a in order: e.run([a])
            except engine.EngineError as err:
                if err.code in (abi.KB_E_UNSUPPORTED, abi.KB_E_INVALID): skip+=1; continue
                print("ERROR", ci, kind, seed, order, err, flush=True); bad+=1; continue
            same = np.array_equal(e.binds(), o.binds()) and all(np.array_equal(a,b) for a,b in zip(e.task_state(), o.task_state())) and all(np.array_equal(a,b) for a,b in zip(e.node_state(), o.node_state())) and [int(t) for t in e.evictions()]==[int(t) for t in o.evictions()]
            if not same: bad+=1; print("RELOAD DIVERGES conf", ci, kind, seed, order, flush=True)
            else: ok+=1
            o.close()
        e.close()
    print(f"[{lo},{hi}): {ok} equal, {skip} skipped, {bad} bad")
    return bad


def hunt_sharded(lo, hi):
    import torch
    import test_gpu_fuzz as fz
    distmod = importlib.import_module("kube-batch_amd.dist")
    cpu=torch.device("cpu")
    ok=bad=skip=0
    def run(tag, cfg, snap, window, batch):
        nonlocal ok, bad, skip
        try:
            o = oracle.Oracle(cfg, snap); o.run(["allocate","backfill"])
        except RuntimeError:
            skip+=1; return
        for mr in (0, 32):
            try:
                eng = engine.Engine(cfg, window=window, commit_batch=batch); eng.load(snap)
                cyc = distmod.ShardedCycle(cfg, snap, backend=distmod.EngineBackend(eng, cpu), buffer_device=cpu, min_rows_per_rank=mr)
                dec = cyc.step()
            except engine.EngineError as err:
                if err.code in (abi.KB_E_UNSUPPORTED, abi.KB_E_INVALID): skip+=1; eng.close(); continue
                print("ERROR", tag, mr, err, flush=True); bad+=1; continue
            od=o.decisions()
            same = dec.shape==od.shape and np.array_equal(dec,od) and np.array_equal(eng.binds(), o.binds()) and all(np.array_equal(a,b) for a,b in zip(eng.node_state(), o.node_state())) and all(np.array_equal(a,b) for a,b in zip(eng.shares(), o.shares()))
            if same:
                dec2 = cyc.step(); same = np.array_equal(dec2, dec)
            if not same: bad+=1; print("SHARDED DIVERGES", tag, "min_rows", mr, flush=True)
            else: ok+=1
            eng.close()
        o.close()
    for seed in range(lo,hi):
        cfg, snap, window, batch = fz._case(seed); run(("fuzz",seed), cfg, snap, window, batch)
        snap = rawgen.raw_snapshot(seed); rng=np.random.RandomState(seed)
        wl, wm, wa, wb = [int(x) for x in rng.choice([0, 1, 1, 2, 5], size=4)]
        cfg = conf.load_scheduler_conf(cases.CONF_TMPL.format(wl=wl, wm=wm, wa=wa, wb=wb)); run(("raw",seed), cfg, snap, int(rng.choice([0,1,3,64])), int(rng.choice([0,1,5,16])))
        try:
            cfg, snap = interpod_case(seed)
            if snap.interpod is not None: run(("interpod",seed), cfg, snap, [0,64,16,256][seed%4], 0)
        except kbm.snapshot.UnsupportedSnapshot:
            pass
    print(f"[{lo},{hi}): {ok} equal, {skip} skipped, {bad} bad")
    return bad


def hunt_wideports(lo, hi):
    import test_gpu_fuzz as fuzz
    import test_gpu_preempt as pre
    orders = [["allocate", "backfill"], ["allocate", "backfill", "preempt"], ["reclaim", "allocate", "backfill", "preempt"], ["preempt", "allocate", "backfill"], ["preempt"], ["reclaim", "preempt"]]
    bad = ok = skip = 0
    for seed in range(lo, hi):
        for kind in ("raw", "fuzz", "evict"):
            if kind == "raw":
                snap = rawgen.raw_snapshot(seed)
            elif kind == "fuzz":
                snap = fuzz._case(seed % 40)[1]
            else:
                snap = cases._evict_case(seed)[1]
            rawgen.widen_ports(snap, 70000 + seed, words=2 + seed % 4, p_task=[0.3, 0.6, 0.9][seed % 3], low_share=[0.0, 0.5, 0.9][(seed // 3) % 3])
            order = orders[(seed // 2) % len(orders)] if kind != "fuzz" else orders[0]
            cfg = conf.load_scheduler_conf(CONFS[seed % len(CONFS)].format(actions=", ".join(order)))
            o = oracle.Oracle(cfg, snap)
            try:
                o.run(order)
            except RuntimeError:
                skip += 1; o.close(); continue
            e = engine.Engine(cfg, window=[0, 64, 100, 256][seed % 4])
            try:
                e.load(snap); e.run(order)
                pre._compare(e, o, snap, (kind, seed), cfg)
                if seed % 5 == 0:
                    first = [e.binds().copy(), np.array(e.evictions())]
                    e.reset(); e.run(order)
                    assert np.array_equal(first[0], e.binds()) and np.array_equal(first[1], np.array(e.evictions())), "reset"
                ok += 1
            except engine.EngineError as err:
                if err.code in (abi.KB_E_UNSUPPORTED, abi.KB_E_INVALID):
                    skip += 1
                else:
                    bad += 1; print("ENGINE ERROR", kind, seed, order, err, flush=True)
            except AssertionError as err:
                bad += 1; print("WIDE PORTS DIVERGE", kind, seed, order, str(err)[:200], flush=True)
            e.close(); o.close()
    print(f"[{lo},{hi}): {ok} equal, {skip} skipped, {bad} differ")
    return bad


if __name__ == "__main__":
    mode, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    sys.exit(1 if {"reset": hunt_reset, "reload": hunt_reload, "sharded": hunt_sharded, "wideports": hunt_wideports}[mode](lo, hi) else 0)
