"""Regression cases that were BORN on the emulated device (tests/test_emu_engine_cpu.py, round 2: differential hunts through the engine's
whole host side on a CPU) and passed their first run on the MI355X at the start of round 3 (profiles/round3/first_gpu_call/): they are
part of the `-m gpu` suite now, against the real kernels, and tests/test_emu_engine_cpu.py re-collects them on the emulated device like
the other `-m gpu` modules.  Every case compares the engine with the oracle through the C ABI:
  * every launch-path variant of the host protocol (chained rounds, pinned mailbox, direct window, probe, commit-kernel pin);
  * a job whose queue is missing (allocate.go:56-60) without proportion;
  * preempt with preferred node-affinity terms (lists rebuilt after every Pipeline; default-on since this file exists);
  * scalar keys that allocate CREATED in a node's Idle map surviving an evict action (Resource.Sub on a non-nil map);
  * allocate / backfill mixed into evict orders on adversarial snapshots under every tier layout;
  * kb_session_reset after evict actions;
  * the journal capacity contract of kb_run_preempt."""
import ctypes as C
import importlib

import numpy as np
import pytest

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
abi = kbm.abi

pytestmark = [pytest.mark.gpu]


# ---- every launch-path variant of the host protocol gives the same cycle ------------------------------------------------------
_VARIANTS = [{}, {"KB_CHAIN_ROUNDS": "0"}, {"KB_PROBE": "0"},
             {"KB_COMMIT_KERNEL": "run"}, {"KB_COMMIT_KERNEL": "select"}, {"KB_CHAIN_ROUNDS": "0", "KB_PROBE": "0"},
             # (round 6: the environment switches KB_SYNC_ROUNDS and KB_DIRECT_WINDOW are gone — kb_config.flags & KB_FLAG_SYNC_ROUNDS is the former,
             #  run by tests/test_gpu_parity.py, and the copied window is what that flag and the round API of the task-row split take)
             # round 3: chained rounds build their candidate lists on a second stream beside the predecessor's commit and repair them
             # (the default, variant 0); KB_OVERLAP=0 keeps every round on one stream
             {"KB_OVERLAP": "0"}, {"KB_OVERLAP": "0", "KB_COMMIT_KERNEL": "run"},
             # round 5: the selection kernel's launch carries the repair workgroups of its round (the default, variant 0); KB_FUSE_REPAIR=0
             # keeps them a launch of their own in front of it (what the run kernel's rounds always do: variant 5)
             {"KB_FUSE_REPAIR": "0"}]


@pytest.mark.parametrize("variant", range(len(_VARIANTS)))
def test_launch_path_variants_agree_with_the_oracle(oracle_mod, variant, monkeypatch):
    """Chained rounds, the pinned mailbox, the direct window, the feasibility probe, the commit-kernel pin and the overlapped candidate lists only change HOW the
    host drives the device (kb_engine_create reads the switches): decisions, binds, node state and shares stay the oracle's."""
    import test_gpu_fuzz as fz
    for k, v in _VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    cases = [fz._case(seed) for seed in (3, 11, 19, 27)]
    cases.append((kbm.conf.load_scheduler_conf(), kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.03)), 0, 0))
    cases.append((kbm.conf.load_scheduler_conf(), kbm.snapshot.synth(kbm.snapshot.synth_config(4, 0.03)), 64, 0))
    # windows of one and three rows: the commit workgroup's LDS layout is then smaller than a repair workgroup's tables (round 5, call 20)
    cases.append((kbm.conf.load_scheduler_conf(), kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.01)), 3, 0))
    cases.append((kbm.conf.load_scheduler_conf(), kbm.snapshot.synth(kbm.snapshot.synth_config(2, 0.05)), 1, 0))
    for cfg, snap, window, batch in cases:
        o = oracle_mod.Oracle(cfg, snap)
        o.run(["allocate", "backfill"])
        e = engine.Engine(cfg, window=window, commit_batch=batch)
        e.load(snap)
        dec = e.run(["allocate", "backfill"])
        assert np.array_equal(dec, o.decisions())
        assert np.array_equal(e.binds(), o.binds())
        for a, b in zip(e.node_state(), o.node_state()):
            assert np.array_equal(a, b)
        assert e.stats()["evals"] == o.evals
        e.reset()                                            # a second cycle from the pristine copy: identical
        assert np.array_equal(e.run(["allocate", "backfill"]), dec)
        e.close()
        o.close()


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_windows_of_as_many_shapes_as_a_window_admits(oracle_mod, monkeypatch, fuse):
    """Single-task jobs that each draw their own request: every window is cut at the shape capacity (64 rows, 64 shapes), so a chained round
    repairs 64 lists — in the selection kernel's launch the repair workgroups then reach beyond its KB_WARM_GRID workgroups (rows 56 and up;
    round 5).  Equal to the oracle with the repair inside the launch and as a launch of its own."""
    monkeypatch.setenv("KB_FUSE_REPAIR", fuse)
    p = kbm.snapshot.synth_config(3, 0.03)
    p.gang_sizes, p.gang_probs, p.diverse_requests = (1,), (1.0,), True
    snap = kbm.snapshot.synth(p)
    cfg = kbm.conf.load_scheduler_conf()
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate", "backfill"])
    e = engine.Engine(cfg)
    e.load(snap)
    dec = e.run(["allocate", "backfill"])
    assert np.array_equal(dec, o.decisions()) and np.array_equal(e.binds(), o.binds())
    for a, b in zip(e.node_state(), o.node_state()):
        assert np.array_equal(a, b)
    st = e.stats()
    assert st["evals"] == o.evals
    assert st["matrix_evals"] / (snap.node_idle.shape[1] * st["matrix_launches"]) > 56.0, st      # shapes per round, on average
    e.close()
    o.close()


def test_job_with_a_missing_queue_without_proportion(oracle_mod):
    """"queue not found" (allocate.go:56-60) is legal when proportion is not loaded: allocate skips the job, drf still counts its
    running tasks; the share reduction must not look for a queue row (kb_kernels.hip: k_finalize_jobs guards q < Q)."""
    import copy
    import test_pyref_vs_oracle as cases
    conf_text = cases.CONF_FULL.format(actions="allocate, backfill").replace("  - name: proportion\n", "")
    cfg = kbm.conf.load_scheduler_conf(conf_text)
    assert not any(po.name == "proportion" for tier in cfg.tiers for po in tier)
    hit = 0
    for seed in range(12):
        base = cases._evict_case(seed)[1]                   # clusters with running tasks
        s = copy.copy(base)
        s.job_queue = base.job_queue.copy()
        s.job_queue[seed % s.n_jobs] = abi.KB_NONE
        try:
            o = oracle_mod.Oracle(cfg, s)
            o.run(["allocate", "backfill"])
        except RuntimeError:
            continue
        e = engine.Engine(cfg)
        e.load(s)
        dec = e.run(["allocate", "backfill"])
        assert np.array_equal(dec, o.decisions()), seed
        assert np.array_equal(e.binds(), o.binds()), seed
        ejs, _ = e.shares()[:2]
        assert np.array_equal(ejs, o.shares()[0]), seed
        e.close()
        o.close()
        hit += 1
    assert hit >= 6


@pytest.mark.parametrize("seed", range(60))
def test_preempt_with_preferred_node_affinity(oracle_mod, seed, monkeypatch):
    """The engine side of tests/test_host_evict_cpu.py's test of the same name: run_evict_action's list path with the NodeAffinity
    launch between matrix and arg-max, mixed action orders included."""
    import test_gpu_preempt as gp
    import test_host_evict_cpu as hev
    cfg, snap, order = hev.affinity_evict_case(seed)
    gp._run_both(oracle_mod, cfg, snap, order, ("affinity", seed))


@pytest.mark.parametrize("seed,ci,order", [(2096, 0, "allocate,preempt"), (2161, 3, "allocate,backfill,preempt,allocate"), (2216, 4, "allocate,preempt"),
                                           (2420, 3, "allocate,preempt"), (2720, 3, "allocate,backfill,preempt,allocate"), (2927, 3, "allocate,preempt"),
                                           (2983, 4, "allocate,preempt"), (3031, 3, "allocate,preempt")])
def test_scalar_keys_created_by_allocate_survive_an_evict_action(oracle_mod, seed, ci, order):
    """Found by scripts/hunt_evict_cpu.py on the emulated device (KB_HUNT_EMU=1), adversarial snapshots under mixed action orders:
    Resource.Sub creates the keys of its operand in a non-nil map, so sub-epsilon requests for a scalar a node never advertised leave a
    negative Idle value under a key its Allocatable does not have.  run_evict_action rebuilt the host mirror's key mask from the static
    mask alone, read that value as 0 and wrote 0 back for every node the action touched."""
    import rawgen
    import test_gpu_preempt as gp
    import test_pyref_vs_oracle as cases
    confs = [cases.CONF_FULL] + cases.EVICT_CONFS
    acts = order.split(",")
    cfg = kbm.conf.load_scheduler_conf(confs[ci].format(actions=", ".join(acts)))
    gp._run_both(oracle_mod, cfg, rawgen.raw_snapshot(seed), acts, (seed, ci, order))


@pytest.mark.parametrize("seed", range(3300, 3380))
def test_mixed_action_orders_on_adversarial_snapshots(oracle_mod, seed):
    """allocate / backfill between and around the evict actions, on the raw snapshots and every tier layout: the combination the
    committed suites did not have (evict-only orders on raw snapshots, mixed orders on synthetic clusters)."""
    import rawgen
    import test_gpu_preempt as gp
    import test_pyref_vs_oracle as cases
    confs = [cases.CONF_FULL] + cases.EVICT_CONFS
    orders = [["allocate", "preempt"], ["reclaim", "allocate", "backfill", "preempt"], ["preempt", "allocate", "backfill", "reclaim"],
              ["allocate", "backfill", "preempt", "allocate"], ["allocate", "reclaim", "preempt"]]
    ci = seed % len(confs)
    acts = orders[(seed // len(confs)) % len(orders)]
    cfg = kbm.conf.load_scheduler_conf(confs[ci].format(actions=", ".join(acts)))
    gp._run_both(oracle_mod, cfg, rawgen.raw_snapshot(seed), acts, (seed, ci, acts))


@pytest.mark.parametrize("seed", [13, 28, 288] + list(range(400, 440)))
def test_session_reset_after_evict_actions_reproduces_the_first_run(seed):
    """kb_session_reset restores the pristine session: the same actions then give the same journals, evictions and state.  The evict
    actions rewrite the key masks of the nodes they touch (upload_live_nodes), which the reset used to leave behind (seeds 13, 28, 288:
    a Releasing map that was nil at load stayed non-nil for the second run; found by a reset hunt on the emulated device)."""
    import rawgen
    import test_pyref_vs_oracle as cases
    confs = [cases.CONF_FULL] + cases.EVICT_CONFS
    orders = [["allocate", "preempt"], ["preempt", "allocate", "backfill"], ["reclaim", "allocate", "backfill", "preempt"],
              ["allocate", "backfill", "preempt", "reclaim"], ["preempt"], ["reclaim", "preempt"]]
    ci, order = seed % len(confs), orders[(seed // len(confs)) % len(orders)]
    cfg = kbm.conf.load_scheduler_conf(confs[ci].format(actions=", ".join(order)))

    def state(e):
        return [e.binds().copy(), *[x.copy() for x in e.task_state()], *[x.copy() for x in e.node_state()], *[x.copy() for x in e.shares()[:2]],
                np.array(e.evictions())]
    ran = 0
    for snap in (rawgen.raw_snapshot(seed), cases._evict_case(seed)[1]):
        e = engine.Engine(cfg)
        try:
            e.load(snap)
            first = [np.array(e.run([a])) for a in order] + state(e)
            e.reset()
            again = [np.array(e.run([a])) for a in order] + state(e)
        except engine.EngineError as err:
            assert err.code in (abi.KB_E_UNSUPPORTED, abi.KB_E_INVALID), err
            continue
        finally:
            e.close()
        for k, (a, b) in enumerate(zip(first, again)):
            assert a.shape == b.shape and np.array_equal(a, b), (seed, k)
        ran += 1
    if not ran:
        pytest.skip("both snapshots are outside the engine's envelope")


def test_journal_capacity_contract(oracle_mod):
    """kb_run_preempt with a journal buffer that is too small answers KB_E_CAPACITY with the required count and applies no result; after
    kb_session_load the same call with that count gives the journal a roomy first call gives (what the Go shim's runJournal does)."""
    import test_pyref_vs_oracle as cases
    done = 0
    for seed in range(40):
        cfg, snap, _ = cases._evict_case(seed)
        ref = engine.Engine(cfg)
        ref.load(snap)
        try:
            ref.run_preempt()
        except engine.EngineError:
            ref.close()
            continue
        want = ref.last_journal
        ref.close()
        if len(want) < 3:
            continue
        e = engine.Engine(cfg)
        e.load(snap)
        n = C.c_uint64()
        small = (abi.StmtOp * 2)()
        assert e.L.kb_run_preempt(e.h, small, 2, C.byref(n)) == abi.KB_E_CAPACITY
        assert n.value == len(want)
        assert len(e.evictions()) == 0                                  # no result was applied (the getters keep working)
        # ... but a mid-action refresh may have updated nodes on the device: every kb_run_* refuses until the session is loaded or reset
        with pytest.raises(engine.EngineError) as err:
            e.run_allocate()
        assert err.value.code == abi.KB_E_STATE
        if done % 2:
            e.reset()
        else:
            e.load(snap)
        exact = (abi.StmtOp * n.value)()
        n2 = C.c_uint64()
        assert e.L.kb_run_preempt(e.h, exact, n.value, C.byref(n2)) == abi.KB_OK and n2.value == n.value
        got = np.frombuffer(exact, dtype=np.uint32).reshape(n.value, 4)
        assert np.array_equal(got, want)
        e.close()
        done += 1
    assert done >= 10


# ---- round 6: the selection kernel's shots evaluate u placements as Idle - u * Resreq, which is u subtractions only for whole numbers -----------
@pytest.mark.parametrize("case", ["fractional", "bin_packing", "short_clean_lists"])
def test_runs_on_sessions_the_shots_cannot_take_and_on_the_ones_they_are_for(oracle_mod, commit_kernel, case):
    """kb_session_load scans requests, Idle and Releasing for whole numbers below 2^47 (KbDev::whole; k8s quantities in milli-units and bytes always
    are).  `fractional`: quarter-milli requests and Idle values — a run's rows then go one by one (one Sub per placement, in the reference's order of
    floating-point operations: api/node_info.go:172-212), and the cycle still equals the oracle bit for bit, node state included.
    `bin_packing`: most_requested.go:34-61 with weight 5 on few large nodes — one node takes many rows of a run (deep tables, two contenders).
    `short_clean_lists`: a handful of nodes and runs longer than that — every feasible dirty slot contends (the raised floor of the shots)."""
    S = kbm.snapshot
    binpack = kbm.conf.load_scheduler_conf('actions: "allocate, backfill"\ntiers:\n- plugins:\n  - name: priority\n  - name: gang\n- plugins:\n  - name: drf\n  - name: predicates\n'
                                           '  - name: proportion\n  - name: nodeorder\n    arguments:\n      leastrequested.weight: 0\n      mostrequested.weight: 5\n      balancedresource.weight: 1\n')
    if case == "fractional":
        cfg, snap = kbm.conf.load_scheduler_conf(), S.synth(S.synth_config(3, 0.03))
        T, N, R = snap.n_tasks, snap.n_nodes, snap.n_res
        res, init = np.asarray(snap.task_resreq).reshape(R, T), np.asarray(snap.task_init_resreq).reshape(R, T)
        jobs = np.asarray(snap.task_job)
        bump = np.where((jobs % 3 == 0) & (res[0] > 0), 0.25, 0.0)          # every task of a job alike: runs stay runs
        res[0] += bump
        init[0] += bump
        np.asarray(snap.node_idle).reshape(R, N)[0, ::4] += 0.5
    elif case == "bin_packing":
        cfg = binpack
        snap = S.synth(S.SynthParams(n_tasks=1500, n_nodes=12, n_queues=3, n_res=2, seed=S.SEED_BASE + 601, node_cpu_cores=(96, 128), node_mem_gib=(512,)))
    else:
        cfg = kbm.conf.load_scheduler_conf()
        snap = S.synth(S.SynthParams(n_tasks=1200, n_nodes=40, n_queues=2, n_res=2, seed=S.SEED_BASE + 602))
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate", "backfill"])
    e = engine.Engine(cfg)
    e.load(snap)
    dec = e.run(["allocate", "backfill"])
    assert len(dec) > 100 and np.array_equal(dec, o.decisions())
    assert np.array_equal(e.binds(), o.binds())
    for a, b in zip(e.node_state(), o.node_state()):
        assert np.array_equal(a, b)
    for a, b in zip(e.shares(), o.shares()):
        assert np.array_equal(a, b)
    e.close()
    o.close()
