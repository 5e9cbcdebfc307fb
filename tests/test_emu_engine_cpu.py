"""The engine's HOST side, whole, on a CPU: kb_engine.cpp + kb_session.cpp + kb_order.cpp + kb_preempt.cpp compiled unchanged with g++ against
tests/host_harness/hip_mock (a synchronous stand-in for the few HIP runtime calls they make) and linked with
tests/host_harness/device_emu.cpp, a sequential restatement of what each kernel launch computes.  The result exports the complete C ABI
of include/kb_engine.h, so the `-m gpu` suites themselves run here — same test functions, same oracle comparison — with the emulated
library loaded through kube-batch_amd/engine.py's own ctypes bindings.

What it covers that nothing else does without a GPU: ActionRun (speculated windows, roll-back and replay, dead shapes, the
feasibility probe), chained rounds and the pinned mailbox protocol, the commit-kernel choice, session load / reset, kb_eval_matrix /
kb_argmax_rows plumbing, kb_round_* (the sharded path's entry points), the evict actions end to end, the aggregate cross-checks.
What it does NOT cover: the kernels.  Their parity with the oracle is the `-m gpu` suite on the MI355X; the product has no CPU path
(tests/test_abi_cpu.py::test_create_without_gpu_fails_loudly) and this library is never loaded outside this file.

The full-size configurations 3 and 4 (100k x 10k) run here too, against the golden digests (about fifteen seconds each); 1M x 50k on request."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE]
CSRC = os.path.join(HERE, "..", "kube-batch_amd", "csrc")
HH = os.path.join(HERE, "host_harness")

engine = importlib.import_module("kube-batch_amd.engine")
kbm = importlib.import_module("kube-batch_amd")
abi = kbm.abi


def build_emulated_library():
    if os.environ.get("KB_EMU_LIB"):                     # an instrumented build (scripts/sanitize_cpu.sh)
        return os.environ["KB_EMU_LIB"]
    out_dir = os.path.join(HH, "build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libkbengine_emu.so")
    srcs = [os.path.join(CSRC, f) for f in ("kb_engine.cpp", "kb_session.cpp", "kb_order.cpp", "kb_preempt.cpp")] + [os.path.join(HH, "device_emu.cpp"), os.path.join(HH, "hip_mock", "hip_mock.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("kb_device.h", "kb_eval.hpp", "kb_host.hpp", "kb_preempt.hpp")] + \
        [os.path.join(HH, "hip_mock", "hip", "hip_runtime.h"), os.path.join(HERE, "..", "include", "kb_engine.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        tmp = f"{so}.{os.getpid()}"                      # atomic: parallel pytest workers may build at the same time
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result",
                               "-pthread", "-I" + os.path.join(HH, "hip_mock"), "-o", tmp] + srcs)
        os.replace(tmp, so)
    return so


REAL_DEVICE = os.environ.get("KB_EMU_USE_REAL") == "1"   # on the GPU box: the cases that were born here, against the real kernels


@pytest.fixture(autouse=True)
def emulated_engine():
    """engine.lib() binds whatever engine.LIB_PATH names: point it at the emulated build for the duration of a test of THIS module.
    KB_EMU_USE_REAL=1 leaves the product library in place instead (scripts/first_gpu_call_r3.sh): the regression cases the emulated
    hunts produced then run on the MI355X before they are promoted into the `-m gpu` suite."""
    if REAL_DEVICE:
        yield engine.LIB_PATH
        return
    so = build_emulated_library()
    saved = (engine.LIB_PATH, engine._LIB)
    engine.LIB_PATH, engine._LIB = so, None
    yield so
    engine.LIB_PATH, engine._LIB = saved


def test_the_emulated_library_exports_the_whole_c_abi(emulated_engine):
    L = C.CDLL(emulated_engine)
    for name in engine.EXPORTS:
        assert hasattr(L, name), name


def test_product_library_path_is_untouched_outside_this_module():
    if REAL_DEVICE:
        pytest.skip("running against the product library")
    assert engine.LIB_PATH.endswith("libkbengine_emu.so")          # inside a test of this module
    assert os.path.basename(os.path.dirname(engine.__file__)) == "kube-batch_amd"
    src = open(engine.__file__).read()
    assert "emu" not in src and "host_harness" not in src          # the product wrapper knows nothing about the emulation


# ---- the `-m gpu` suites, re-collected here without their marker -------------------------------------------------------------
# (module-level `pytestmark = pytest.mark.gpu` belongs to the module a function is collected FROM, so the copies below are plain
# CPU tests; their parametrisation travels with the function objects.)
_SKIP = {
    # sizes that only make sense on the device
    "test_gpu_parity": {"test_wide_cluster_more_than_64k_nodes", "test_full_size_properties_config3"},
}


def _adopt(module_name, only=None):
    mod = importlib.import_module(module_name)
    for name, obj in sorted(vars(mod).items()):
        if not (name.startswith("test_") and callable(obj)) or name in _SKIP.get(module_name, ()):
            continue
        if only is not None and name not in only:
            continue
        globals()[f"{name}__{module_name[5:]}"] = obj


for _m in ("test_gpu_parity", "test_gpu_fuzz", "test_gpu_adversarial", "test_gpu_interpod", "test_gpu_preempt", "test_framework_actions"):
    _adopt(_m)


# ---- the sharded path's entry points (kb_round_begin / candidates / commit / apply) through dist.py, CPU buffers ----------------
def _sharded_cycle(so, min_rows_per_rank):
    import torch
    distmod = importlib.import_module("kube-batch_amd.dist")
    engine.LIB_PATH, engine._LIB = so, None
    conf = kbm.conf.load_scheduler_conf()
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.05))
    eng = engine.Engine(conf, device=0, window=256)
    eng.load(snap)
    cpu = torch.device("cpu")       # the emulated "device" memory is host memory: torch's CPU tensors are its buffers
    return conf, snap, distmod.ShardedCycle(conf, snap, backend=distmod.EngineBackend(eng, cpu), buffer_device=cpu, min_rows_per_rank=min_rows_per_rank)


def test_sharded_rounds_world1_equal_the_oracle(emulated_engine, oracle_mod):
    conf, snap, cyc = _sharded_cycle(emulated_engine, 32)
    dec = cyc.step()
    o = oracle_mod.Oracle(conf, snap)
    o.run(["allocate", "backfill"])
    assert np.array_equal(dec, o.decisions())
    assert np.array_equal(cyc.engine.binds(), o.binds())
    for a, b in zip(cyc.engine.node_state(), o.node_state()):
        assert np.array_equal(a, b)
    assert np.array_equal(cyc.step(), dec)                 # reset + second cycle: identical


def _sharded_worker(rank, world, port, out_dir, so):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, _, cyc = _sharded_cycle(so, 0)                  # always exchange: every round all-gathers the lists and all-reduces the deltas
        dec = cyc.step()
        np.save(os.path.join(out_dir, f"dec{rank}.npy"), dec)
        np.save(os.path.join(out_dir, f"binds{rank}.npy"), cyc.engine.binds())
        st = cyc.engine.stats()
        np.save(os.path.join(out_dir, f"mevals{rank}.npy"), np.array([st["matrix_evals"], st["rounds"]]))
    finally:
        dist.destroy_process_group()


def test_sharded_rounds_two_gloo_ranks_equal_the_oracle(emulated_engine, oracle_mod, tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path), emulated_engine), nprocs=2, join=True)
    conf = kbm.conf.load_scheduler_conf()
    o = oracle_mod.Oracle(conf, kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.05)))
    o.run(["allocate", "backfill"])
    for r in (0, 1):
        assert np.array_equal(np.load(tmp_path / f"dec{r}.npy"), o.decisions()), f"rank {r}"
        assert np.array_equal(np.load(tmp_path / f"binds{r}.npy"), o.binds()), f"rank {r}"
    m0, m1 = np.load(tmp_path / "mevals0.npy"), np.load(tmp_path / "mevals1.npy")
    assert m0[1] == m1[1]                                   # same number of rounds on both ranks
    assert abs(int(m0[0]) - int(m1[0])) <= m0[1] * 10_000   # each rank evaluated about half of the matrix rows


# ---- every launch-path variant of the host protocol gives the same cycle ------------------------------------------------------
_VARIANTS = [{}, {"KB_CHAIN_ROUNDS": "0"}, {"KB_SYNC_ROUNDS": "1"}, {"KB_PROBE": "0"}, {"KB_DIRECT_WINDOW": "0"},
             {"KB_COMMIT_KERNEL": "run"}, {"KB_COMMIT_KERNEL": "batch"}, {"KB_CHAIN_ROUNDS": "0", "KB_PROBE": "0", "KB_DIRECT_WINDOW": "0"}]


@pytest.mark.parametrize("variant", range(len(_VARIANTS)))
def test_launch_path_variants_agree_with_the_oracle(oracle_mod, variant, monkeypatch):
    """Chained rounds, the pinned mailbox, the direct window, the feasibility probe and the commit-kernel pin only change HOW the
    host drives the device (kb_engine_create reads the switches): decisions, binds, node state and shares stay the oracle's."""
    import test_gpu_fuzz as fz
    for k, v in _VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    cases = [fz._case(seed) for seed in (3, 11, 19, 27)]
    cases.append((kbm.conf.load_scheduler_conf(), kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.03)), 0, 0))
    cases.append((kbm.conf.load_scheduler_conf(), kbm.snapshot.synth(kbm.snapshot.synth_config(4, 0.03)), 64, 0))
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
def test_preempt_with_preferred_node_affinity_behind_its_switch(oracle_mod, seed, monkeypatch):
    """The engine side of tests/test_host_evict_cpu.py's test of the same name: run_evict_action's list path with the NodeAffinity
    launch between matrix and arg-max, mixed action orders included.  Off by default (KB_E_UNSUPPORTED) until its first device run."""
    import test_gpu_preempt as gp
    import test_host_evict_cpu as hev
    cfg, snap, order = hev.affinity_evict_case(seed)
    e = engine.Engine(cfg)
    e.load(snap)
    if "preempt" in order:
        with pytest.raises(engine.EngineError) as err:
            e.run(order)
        assert err.value.code == abi.KB_E_UNSUPPORTED
    e.close()
    monkeypatch.setenv("KB_PREEMPT_NODE_AFFINITY", "1")
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
        assert len(e.evictions()) == 0                                  # no result was applied
        e.load(snap)
        exact = (abi.StmtOp * n.value)()
        n2 = C.c_uint64()
        assert e.L.kb_run_preempt(e.h, exact, n.value, C.byref(n2)) == abi.KB_OK and n2.value == n.value
        got = np.frombuffer(exact, dtype=np.uint32).reshape(n.value, 4)
        assert np.array_equal(got, want)
        e.close()
        done += 1
    assert done >= 10


@pytest.mark.parametrize("name", ["config3_full", "config4_binpack_full", "config5_full"])
def test_full_size_cycles_through_the_host_side(oracle_mod, name):
    """BASELINE configs[2] and [3] at full size (100k x 10k) through the engine's host side on the emulated device, against the oracle and
    the committed golden digests (tests/test_gpu_fullsize.py's own test function): about fifteen seconds each.  The 1M x 50k
    configuration takes two and a half minutes of sequential evaluation: KB_EMU_FULLSIZE_5=1 (passed when this test was written)."""
    import test_gpu_fullsize as fs
    if name in fs.mfg.FAST and os.environ.get("KB_EMU_FULLSIZE_5") != "1":
        pytest.skip("1M x 50k on the emulated device: set KB_EMU_FULLSIZE_5=1 (about 2.5 minutes)")
    fs.test_full_size_cycle_equals_oracle_and_golden_digest(oracle_mod, name)
