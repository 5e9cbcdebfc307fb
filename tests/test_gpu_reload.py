"""The engine's real operating mode: ONE engine, a DIFFERENT session every cycle.

The reference opens a session every scheduling period (pkg/scheduler/scheduler.go:85-101 -> framework/framework.go:30-52), so the Go
action calls kb_session_load on a live engine once a second, for the life of the process.  Every device buffer of the engine is
grow-only (kb_engine.cpp: DevBuf keeps its capacity, stale bytes stay behind `bytes`), so what a cycle finds in HBM, in the pinned
mailboxes and in the host session is whatever the cycles before it left there.  The cases here feed one engine per tier layout a
sequence of snapshots whose shape changes in every way the buffers care about — tasks / nodes up and down by orders of magnitude
(a handful -> 100k x 10k -> a handful), R in {2..6, 16}, host ports absent / one word / several words, inter-pod terms on and off,
class tables and preferred node affinity on and off, every action order incl. preempt / reclaim, a kb_session_reset now and then, loads
and runs the engine refuses (the envelope) in between — and hold every cycle to

  * the oracle (decisions in order, Statement journal, evictions, task table, float64 node state, shares, bind set), and
  * a FRESH engine given the same snapshot (same answers, same error codes where the engine refuses, same matrix rows),

under every commit kernel.  A soak (1 000 cycles of BASELINE configs[1]-sized sessions through one engine) checks SURVEY §8b's
"re-entrant across cycles without leaking device memory": hipMemGetInfo's free bytes stay constant after warm-up and the cycle
time does not creep.  The same sequences run on the emulated device in tests/test_emu_engine_cpu.py (host side only)."""
import ctypes
import importlib
import os
import time

import numpy as np
import pytest

import rawgen
import test_gpu_fuzz as fuzz
import test_pyref_vs_oracle as cases
from test_gpu_preempt import _compare
from test_interpod_oracle_cpu import interpod_case

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
abi, conf, snapmod = kbm.abi, kbm.conf, kbm.snapshot

pytestmark = [pytest.mark.gpu]

LAYOUTS = [cases.CONF_FULL] + cases.EVICT_CONFS          # five tier layouts: the stock one and the four that move the victim rules around
ORDERS = [["allocate", "backfill"], ["allocate", "backfill", "preempt"], ["reclaim", "allocate", "backfill", "preempt"], ["preempt", "allocate", "backfill"],
          ["allocate"], ["preempt", "reclaim"], ["backfill", "allocate"]]
N_STEPS = 72
MIN_COMPARED = 60                                         # cycles per sequence that must have been held to the oracle (the rest: envelope / reference panics)
ALL_ACTIONS = "allocate, backfill, preempt, reclaim"


def on_emulated_device():
    return engine.LIB_PATH.endswith("_emu.so")


def sequence(li, n_steps=N_STEPS):
    """-> [(tag, make_snapshot, order, extras)]: the cycles of tier layout `li`, deterministic.  extras: 'reset' (second run after
    kb_session_reset), 'matrix' (kb_eval_matrix / kb_argmax_rows of the loaded state against a fresh engine's)."""
    rng = np.random.RandomState(5150 + 31 * li)
    emu = on_emulated_device()
    big = 1.0 if li == 0 else 0.3                         # 100k x 10k through the stock layout, 30k x 3k through the others: the padded node
    if emu:                                               # count (kb_device.h: KB_NODE_PAD = 2048) must CHANGE along the way — every per-node row is sized by it
        big = 0.25
    steps = []
    kinds = ["raw", "evict", "alloc", "fuzz", "interpod", "wide", "r16", "raw", "evict", "alloc"]
    for k in range(n_steps):
        seed = int(rng.randint(0, 100000))
        kind = kinds[int(rng.randint(len(kinds)))]
        if k == 0:
            kind = "raw"                                  # a handful of tasks first: every buffer starts small
        elif k in (9, 33):
            kind = "fuzzbig"                              # thousands of tasks, hundreds of nodes, R up to 6, affinity / host ports
        elif k == 20:
            kind = "big"                                  # BASELINE configs[2]'s shape (scaled outside the stock layout)
        elif k in (21, 41):
            kind = "raw"                                  # ... and straight back down
        elif k == 40:
            kind = "r16big"                               # BASELINE configs[3]'s shape at 1/20: R = 16
        order = ORDERS[int(rng.randint(len(ORDERS)))]
        extras = set()
        if rng.uniform() < 0.2:
            extras.add("reset")
        if rng.uniform() < 0.25 or kind in ("big", "fuzzbig", "r16big") or k in (21, 41):
            extras.add("matrix")

        def make(kind=kind, seed=seed):
            if kind == "raw":
                return rawgen.raw_snapshot(seed)
            if kind == "evict":
                return cases._evict_case(seed % 400)[1]
            if kind == "alloc":
                return cases._case(seed % 200)[1]
            if kind == "fuzz":
                return fuzz._case(seed % 24)[1]
            if kind == "fuzzbig":
                return fuzz._case(24 + seed % 16)[1]
            if kind == "interpod":
                return interpod_case(seed % 300, wide=seed % 7 == 0)[1]
            if kind == "wide":                            # host-port masks of 2..4 words on a raw / evict snapshot
                s = rawgen.raw_snapshot(seed) if seed % 2 else cases._evict_case(seed % 400)[1]
                return rawgen.widen_ports(s, 4000 + seed, words=2 + seed % 3, p_task=[0.3, 0.6, 0.9][seed % 3], low_share=[0.0, 0.5, 0.9][(seed // 3) % 3])
            if kind == "r16":
                return snapmod.synth(snapmod.synth_config(4, 0.004 + 0.002 * (seed % 4)))
            if kind == "r16big":
                return snapmod.synth(snapmod.synth_config(4, 0.01 if emu else 0.05))
            if kind == "big":
                return snapmod.synth(snapmod.synth_config(3, big))
            raise AssertionError(kind)
        if kind in ("big", "r16big", "fuzzbig"):
            order = ["allocate", "backfill"] if kind != "r16big" else ["allocate", "backfill", "preempt"]
        steps.append((f"layout{li}/step{k}/{kind}/{seed}", make, order, extras))
    return steps


def _cycle(e, snap, order):
    """One cycle through engine `e`: -> ('ok', per-action outputs) or ('refused', where, code).  The engine stays usable either way."""
    try:
        e.load(snap)
    except engine.EngineError as err:
        if err.code in (abi.KB_E_UNSUPPORTED, abi.KB_E_INVALID):
            return ("refused", "load", err.code)
        raise
    outs = []
    for i, a in enumerate(order):
        try:
            outs.append(np.array(e.run([a])))
        except engine.EngineError as err:
            if err.code == abi.KB_E_UNSUPPORTED:
                return ("refused", f"action {i}", err.code)
            raise
    return ("ok", outs)


def _state(e):
    return [e.binds().copy(), *[x.copy() for x in e.task_state()], *[x.copy() for x in e.node_state()], *[x.copy() for x in e.shares()], np.array(e.evictions()),
            e.journal().copy()]


def _same(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


def run_sequence(oracle_mod, li, steps=None):
    cfg = conf.load_scheduler_conf(LAYOUTS[li].format(actions=ALL_ACTIONS))
    live = engine.Engine(cfg)                             # THE engine of this sequence
    compared = refused = panicked = 0
    for tag, make, order, extras in (steps if steps is not None else sequence(li)):
        try:
            snap = make()
        except snapmod.UnsupportedSnapshot:
            continue
        fresh = engine.Engine(cfg)
        got = _cycle(live, snap, order)
        want = _cycle(fresh, snap, order)
        assert got[0] == want[0] and (got[0] == "ok" or got[1:] == want[1:]), (tag, "live engine", got[:3] if got[0] != "ok" else "ok", "fresh engine", want[:3] if want[0] != "ok" else "ok")
        if got[0] == "refused":
            refused += 1
            fresh.close()
            continue
        assert _same(got[1], want[1]), (tag, "decisions differ from a fresh engine's")
        s_live = _state(live)
        assert _same(s_live, _state(fresh)), (tag, "final state differs from a fresh engine's")
        sl, sf = live.stats(), fresh.stats()
        assert all(sl[k] == sf[k] for k in ("evals", "decisions", "tasks_popped", "rounds", "spec_breaks")), (tag, sl, sf)
        o = oracle_mod.Oracle(cfg, snap, threads=min(16, os.cpu_count() or 1))
        if snap.n_tasks * snap.n_nodes >= 10**8:
            o.set_fast(True)                              # the incremental mode (tests/test_oracle_fast_cpu.py holds it to the faithful one)
        try:
            n0 = 0
            for a, dec in zip(order, got[1]):             # action by action: allocate / backfill answer with decisions, the evict actions with their journal
                o.run([a])
                od = o.decisions()[n0:]
                n0 += len(od)
                if a in ("allocate", "backfill"):         # (the oracle's list also carries the ssn.Pipeline calls of reclaim: those are journal entries here)
                    assert dec.shape == od.shape and np.array_equal(dec, od), (tag, a, "decisions differ from the oracle's")
        except RuntimeError:
            panicked += 1                                 # the reference would panic on this snapshot: live == fresh is all there is to say
        else:
            _compare(live, o, snap, tag, cfg)
            compared += 1
        o.close()
        if "reset" in extras:                             # back to the loaded state from the copy in HBM, same cycle again
            live.reset()
            again = [np.array(live.run([a])) for a in order]
            assert _same(again, got[1]) and _same(_state(live), s_live), (tag, "second cycle after kb_session_reset differs")
        if "matrix" in extras and snap.n_tasks:           # the materialised rows (their staging buffers are sized by nodes x shapes)
            live.reset(); fresh.reset()
            t1 = min(int(snap.n_tasks), 48)
            for a, b in zip(live.eval_matrix(0, t1), fresh.eval_matrix(0, t1)):
                assert np.array_equal(a, b), (tag, "kb_eval_matrix differs from a fresh engine's")
            k = min(8, max(1, int(snap.n_nodes)))
            for a, b in zip(live.argmax_rows(0, t1, k), fresh.argmax_rows(0, t1, k)):
                assert np.array_equal(a, b), (tag, "kb_argmax_rows differs from a fresh engine's")
        fresh.close()
    live.close()
    return compared, refused, panicked


@pytest.mark.parametrize("layout", range(len(LAYOUTS)))
def test_one_engine_many_sessions(oracle_mod, commit_kernel, layout):
    compared, refused, panicked = run_sequence(oracle_mod, layout)
    assert compared >= MIN_COMPARED, f"only {compared} of {N_STEPS} cycles were compared with the oracle ({refused} refused by the engine, {panicked} reference panics)"


def test_matrix_rows_after_the_cluster_grew(oracle_mod):
    """kb_eval_matrix's per-shape staging rows are [shapes][padded nodes]: the same shapes over MORE nodes must not reuse the smaller block
    (kb_engine.cpp kept `xs_cap` across loads: a row count, compared without the row length — found by reading while writing this module;
    on the emulated device under AddressSanitizer the unfixed build overruns b_sscore here)."""
    cfg = conf.load_scheduler_conf()
    small, large = [snapmod.synth(snapmod.SynthParams(n_tasks=400, n_nodes=n, n_queues=3, n_res=3, seed=snapmod.SEED_BASE + 4242, scalar_job_frac=0.4)) for n in (40, 2500)]
    live = engine.Engine(cfg)
    for snap in (small, large, small):
        live.load(snap)
        fresh = engine.Engine(cfg)
        fresh.load(snap)
        o = oracle_mod.Oracle(cfg, snap)
        got, want, ref = live.eval_matrix(0, 200), fresh.eval_matrix(0, 200), o.eval_matrix(0, 200)
        for a, b, c in zip(got, want, ref):
            assert np.array_equal(a, b) and np.array_equal(a, c), snap.n_nodes
        fresh.close(); o.close()
    live.close()


def _hip_free_bytes():
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_soak_1000_cycles_no_device_memory_growth(oracle_mod):
    """1 000 scheduling cycles through ONE engine — kb_session_load of a different BASELINE configs[1]-sized snapshot (10k x 1k and smaller),
    allocate, backfill — : free device memory is constant after warm-up, the cycle time does not creep, the answers stay the oracle's."""
    if on_emulated_device():
        pytest.skip("device memory accounting needs the device")
    cfg = conf.load_scheduler_conf()
    snaps = []
    for i, scale in enumerate([1.0, 0.4, 0.7, 1.0, 0.1, 0.9]):
        p = snapmod.synth_config(2, scale)
        p.seed += 11 * i
        snaps.append(snapmod.synth(p))
    want = []
    for s in snaps:
        o = oracle_mod.Oracle(cfg, s, threads=min(16, os.cpu_count() or 1))
        o.set_fast(True)
        o.run(["allocate", "backfill"])
        want.append((o.decisions().copy(), o.binds().copy()))
        o.close()
    e = engine.Engine(cfg)
    cycles, warm = 1000, 30
    free, ms = [], []
    for c in range(cycles):
        s = snaps[c % len(snaps)]
        t0 = time.perf_counter()
        e.load(s)
        dec = e.run(["allocate", "backfill"])
        ms.append((time.perf_counter() - t0) * 1e3)
        if c % 97 == 0 or c >= cycles - len(snaps):
            assert np.array_equal(dec, want[c % len(snaps)][0]) and np.array_equal(e.binds(), want[c % len(snaps)][1]), f"cycle {c} differs from the oracle"
        free.append(_hip_free_bytes())
    e.close()
    steady = free[warm:]
    assert max(steady) == min(steady), f"free device memory moved after warm-up: {min(steady)} .. {max(steady)} bytes (cycle {warm + int(np.argmin(steady))})"
    per = len(snaps)
    early = np.median(np.array(ms[warm:warm + 20 * per]).reshape(-1, per).sum(axis=1))
    late = np.median(np.array(ms[-20 * per:]).reshape(-1, per).sum(axis=1))
    print(f"soak: {cycles} cycles, free bytes {steady[0]}, ms per {per} cycles early {early:.2f} late {late:.2f}")
    assert late <= 1.25 * early + 1.0, f"cycle time crept: {early:.2f} -> {late:.2f} ms per {per} cycles"
