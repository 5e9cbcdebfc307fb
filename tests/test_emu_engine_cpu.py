"""The engine's HOST side, whole, on a CPU: kb_engine.cpp + kb_load.cpp + kb_rounds.cpp + kb_evict.cpp + kb_matrix.cpp + kb_session.cpp + kb_order.cpp + kb_preempt.cpp compiled unchanged with g++ against
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
    srcs = [os.path.join(CSRC, f) for f in ("kb_engine.cpp", "kb_load.cpp", "kb_rounds.cpp", "kb_evict.cpp", "kb_matrix.cpp", "kb_session.cpp", "kb_order.cpp", "kb_preempt.cpp")] + [os.path.join(HH, "device_emu.cpp"), os.path.join(HH, "hip_mock", "hip_mock.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("kb_engine_int.hpp", "kb_device.h", "kb_eval.hpp", "kb_host.hpp", "kb_res.hpp", "kb_waterfill.hpp", "kb_preempt.hpp")] + \
        [os.path.join(HH, "hip_mock", "hip", "hip_runtime.h"), os.path.join(HERE, "..", "include", "kb_engine.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        tmp = f"{so}.{os.getpid()}"                      # atomic: parallel pytest workers may build at the same time
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result",
                               "-pthread", "-I" + os.path.join(HH, "hip_mock"), "-o", tmp] + srcs)
        os.replace(tmp, so)
    return so


@pytest.fixture(autouse=True)
def emulated_engine():
    """engine.lib() binds whatever engine.LIB_PATH names: point it at the emulated build for the duration of a test of THIS module."""
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
    "test_gpu_reload": {"test_soak_1000_cycles_no_device_memory_growth"},      # device memory accounting
}


def _adopt(module_name, only=None):
    mod = importlib.import_module(module_name)
    for name, obj in sorted(vars(mod).items()):
        if not (name.startswith("test_") and callable(obj)) or name in _SKIP.get(module_name, ()):
            continue
        if only is not None and name not in only:
            continue
        globals()[f"{name}__{module_name[5:]}"] = obj


# test_gpu_regressions: the cases the hunts on this emulated device produced (round 2), since promoted into the `-m gpu` suite
for _m in ("test_gpu_parity", "test_gpu_fuzz", "test_gpu_adversarial", "test_gpu_interpod", "test_gpu_preempt", "test_framework_actions", "test_gpu_regressions",
           "test_gpu_wideports", "test_gpu_reload"):
    _adopt(_m)


# ---- the sharded path's entry points (kb_round_begin / candidates / commit / apply) through dist.py, CPU buffers ----------------
def _sharded_cycle(so, min_rows_per_rank, wide_ports=False):
    import torch
    distmod = importlib.import_module("kube-batch_amd.dist")
    engine.LIB_PATH, engine._LIB = so, None
    conf = kbm.conf.load_scheduler_conf()
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.05))
    if wide_ports:                         # host-port masks of three words: pods that reach beyond word 0 get rounds of their own on this path too
        import rawgen
        rawgen.widen_ports(snap, 77, words=3, p_task=0.3)
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


def test_sharded_rounds_cross_check_their_deltas_one_round_late(emulated_engine, oracle_mod):
    """defer_check (the default): every round's reduced deltas are compared on the device one round late (kb_round_check), the counter is read
    once per action; a reduced buffer that is off by one value fails the action on that read.  Lock-step (defer_check=False) gives the same cycle."""
    import torch
    distmod = importlib.import_module("kube-batch_amd.dist")
    conf, snap, cyc = _sharded_cycle(emulated_engine, 0)
    dec = cyc.step()
    assert cyc.defer_check and cyc.deferred_checks == cyc.rounds > 5
    eng = engine.Engine(conf, device=0, window=256)
    eng.load(snap)
    cpu = torch.device("cpu")
    lock = distmod.ShardedCycle(conf, snap, backend=distmod.EngineBackend(eng, cpu), buffer_device=cpu, min_rows_per_rank=0, defer_check=False)
    assert np.array_equal(lock.step(), dec) and lock.deferred_checks == 0

    class Poisoned(distmod.ShardedCycle):
        def _all_reduce_delta_async(self, buf):
            done = super()._all_reduce_delta_async(buf)

            def bad():
                done()
                if self.rounds == 3:
                    buf[5] += 1.0          # what a replica that committed something else would contribute
            return bad
    eng2 = engine.Engine(conf, device=0, window=256)
    eng2.load(snap)
    with pytest.raises(RuntimeError, match="replicas diverged"):
        Poisoned(conf, snap, backend=distmod.EngineBackend(eng2, cpu), buffer_device=cpu, min_rows_per_rank=0).step()


def test_sharded_rounds_with_host_port_masks_of_several_words(emulated_engine, oracle_mod):
    conf, snap, cyc = _sharded_cycle(emulated_engine, 32, wide_ports=True)
    dec = cyc.step()
    o = oracle_mod.Oracle(conf, snap)
    o.run(["allocate", "backfill"])
    assert np.array_equal(dec, o.decisions())
    assert np.array_equal(cyc.engine.binds(), o.binds())
    assert np.array_equal(cyc.step(), dec)


def _recording(cyc):
    """hooks on a ShardedCycle's two collectives: the all-gathered candidate table and the reduced per-node deltas of every round, as the cycle used them"""
    cyc.tables, cyc.deltas = [], []
    gather, reduce_async = cyc._all_gather_keys, cyc._all_reduce_delta_async

    def rec_gather(local, chunk, L):
        full = gather(local, chunk, L)
        cyc.tables.append(full.clone().numpy())
        return full

    def rec_reduce(buf):
        done = reduce_async(buf)

        def rec_done():
            done()
            cyc.deltas.append(buf.clone().numpy())
        return rec_done
    cyc._all_gather_keys, cyc._all_reduce_delta_async = rec_gather, rec_reduce
    return cyc


def _sharded_worker(rank, world, port, out_dir, so):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, _, cyc = _sharded_cycle(so, 0)                  # always exchange: every round all-gathers the lists and all-reduces the deltas
        _recording(cyc)
        dec = cyc.step()
        np.savez(os.path.join(out_dir, f"coll{rank}.npz"), **{f"t{i}": t for i, t in enumerate(cyc.tables)}, **{f"d{i}": d for i, d in enumerate(cyc.deltas)})
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
    # the collectives against REAL lists (round 5's review: tests/test_dist_cpu.py checks them on a stand-in backend only): every round's
    # all-gathered table and reduced deltas are the same on both ranks, and equal what ONE rank that evaluates every matrix row and owns every
    # window row gets without any exchange — rows of the table beyond the window's shapes are padding and stay 0
    _, _, ref = _sharded_cycle(emulated_engine, 0)
    _recording(ref)
    ref.step()
    c0, c1 = np.load(tmp_path / "coll0.npz"), np.load(tmp_path / "coll1.npz")
    n_t, n_d = len(ref.tables), int(m0[1])
    assert n_t > 5 and sorted(c0.files) == sorted(c1.files) == sorted([f"t{i}" for i in range(n_t)] + [f"d{i}" for i in range(n_d)])
    for i in range(n_t):
        t0, t1, tr = c0[f"t{i}"], c1[f"t{i}"], ref.tables[i]
        assert np.array_equal(t0, t1), f"round {i}: the ranks gathered different tables"
        assert np.array_equal(t0[: len(tr)], tr) and not t0[len(tr):].any(), f"round {i}: the gathered table differs from the lists of an unsharded round"
    ref_d = ref.engine.round_delta_doubles()
    for i in range(n_d):
        assert np.array_equal(c0[f"d{i}"], c1[f"d{i}"]), f"round {i}: the ranks hold different reduced deltas"
        assert c0[f"d{i}"].shape == (ref_d,) and c0[f"d{i}"].any()
    # the reduced deltas of all rounds sum to the cycle's whole effect on the node state (the device-side check compares them round by round)
    idle0 = np.asarray(kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.05)).node_idle, np.float64)
    total = sum(c0[f"d{i}"] for i in range(n_d))
    eidle = ref.engine.node_state()[0]
    N, R, NP = eidle.shape[1], eidle.shape[0], ref_d // (2 * eidle.shape[0] + 3)
    for dim in range(R):
        assert np.array_equal(idle0.reshape(R, N)[dim] + total[dim * NP: dim * NP + N], eidle[dim]), f"dimension {dim}"


# ---- BASELINE configs[4] names a third action: preempt (and reclaim) in the sharded mode — every replica runs the evict action, one all-reduce
#      compares the journals (kube-batch_amd/dist.py) — on the reference's own preempt cases and on a scaled 1M x 50k cycle
from test_gpu_sharded import sharded_evict_inputs as _sharded_evict_inputs, check_sharded_evict_outputs as _check_sharded_evict_outputs  # noqa: E402


def _sharded_evict_worker(rank, world, port, out_dir, so, case):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        distmod = importlib.import_module("kube-batch_amd.dist")
        engine.LIB_PATH, engine._LIB = so, None
        cfg, snap, order = _sharded_evict_inputs(case)
        eng = engine.Engine(cfg, device=0, window=256)
        eng.load(snap)
        cpu = torch.device("cpu")
        cyc = distmod.ShardedCycle(cfg, snap, backend=distmod.EngineBackend(eng, cpu), buffer_device=cpu, min_rows_per_rank=0, actions=order)
        dec = cyc.step()
        assert cyc.evict_actions == sum(a in ("preempt", "reclaim") for a in order)
        np.save(os.path.join(out_dir, f"dec{rank}.npy"), dec)
        np.save(os.path.join(out_dir, f"binds{rank}.npy"), eng.binds())
        np.save(os.path.join(out_dir, f"journal{rank}.npy"), eng.journal())
        np.save(os.path.join(out_dir, f"evict{rank}.npy"), np.array(eng.evictions(), np.uint32))
        for i, a in enumerate(eng.node_state()):
            np.save(os.path.join(out_dir, f"node{i}_{rank}.npy"), a)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", range(4))
def test_sharded_cycle_with_evict_actions_two_gloo_ranks(emulated_engine, oracle_mod, tmp_path, case):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sharded_evict_worker, args=(2, port, str(tmp_path), emulated_engine, case), nprocs=2, join=True)
    _check_sharded_evict_outputs(oracle_mod, tmp_path, case)


# ---- the default multi-GPU mode of bench.py: one session replica per rank, one digest all-reduce (dist.ReplicatedCycle) ---------
def _replica_worker(rank, world, port, out_dir, so, poison_rank):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        distmod = importlib.import_module("kube-batch_amd.dist")
        engine.LIB_PATH, engine._LIB = so, None
        conf = kbm.conf.load_scheduler_conf()
        snap = kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.03))
        cyc = distmod.ReplicatedCycle(conf, snap, device=0)
        dec = cyc.step(verify=False)
        if rank == poison_rank:           # a replica that went wrong: the cross-check must say so on EVERY rank
            dec = dec.copy()
            dec[len(dec) // 2, 1] ^= 1
        try:
            cyc.check(dec)
            verdict = "agree"
        except RuntimeError:
            verdict = "diverged"
        np.save(os.path.join(out_dir, f"dec{rank}.npy"), dec)
        open(os.path.join(out_dir, f"verdict{rank}.txt"), "w").write(verdict)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("poison_rank", [-1, 1])
def test_replicated_cycle_two_gloo_ranks(emulated_engine, oracle_mod, tmp_path, poison_rank):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_replica_worker, args=(2, port, str(tmp_path), emulated_engine, poison_rank), nprocs=2, join=True)
    verdicts = [open(tmp_path / f"verdict{r}.txt").read() for r in (0, 1)]
    if poison_rank < 0:
        assert verdicts == ["agree", "agree"]
        o = oracle_mod.Oracle(kbm.conf.load_scheduler_conf(), kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.03)))
        o.run(["allocate", "backfill"])
        for r in (0, 1):
            assert np.array_equal(np.load(tmp_path / f"dec{r}.npy"), o.decisions()), f"rank {r}"
    else:
        assert verdicts == ["diverged", "diverged"]


@pytest.mark.parametrize("name", ["config3_full", "config4_binpack_full", "config5_full", "config5_full_preempt"])
def test_full_size_cycles_through_the_host_side(oracle_mod, name):
    """BASELINE configs[2] and [3] at full size (100k x 10k) through the engine's host side on the emulated device, against the oracle and
    the committed golden digests (tests/test_gpu_fullsize.py's own test function): about fifteen seconds each.  The 1M x 50k
    configuration takes two and a half minutes of sequential evaluation: KB_EMU_FULLSIZE_5=1 (passed when this test was written)."""
    import test_gpu_fullsize as fs
    if name in fs.mfg.FAST and os.environ.get("KB_EMU_FULLSIZE_5") != "1":
        pytest.skip("1M x 50k on the emulated device: set KB_EMU_FULLSIZE_5=1 (about 2.5 minutes)")
    fs.test_full_size_cycle_equals_oracle_and_golden_digest(oracle_mod, name)


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_overlapped_candidate_lists_are_repaired(emulated_engine, monkeypatch, fuse):
    """(fuse: the repair inside the selection kernel's launch — the default — or as a launch of its own in front of it, KB_FUSE_REPAIR=0.)
    Chained rounds of plain sessions build their candidate lists on a second stream beside the predecessor's commit kernel and repair them
    behind it (kb_kernels.hip: k_repair).  The emulated matrix launch of such a round POISONS what it reports for the nodes
    the last commit changed, so the decisions only come out right if the repair launch overrides exactly those: equal to the oracle with it,
    different without it (KB_EMU_REPAIR_OFF=1 hands the stale lists on as they are — the negative control), and equal again on the plain
    path (KB_OVERLAP=0)."""
    monkeypatch.setenv("KB_FUSE_REPAIR", fuse)
    oracle_mod = importlib.import_module("oracle")
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.05))
    conf = kbm.conf.load_scheduler_conf()
    o = oracle_mod.Oracle(conf, snap)
    o.run(["allocate", "backfill"])

    def cycle():
        eng = engine.Engine(conf)
        eng.load(snap)
        dec = eng.run_allocate()
        eng.run_backfill()
        out = (np.array_equal(eng.binds(), o.binds()), eng.stats()["rounds"], len(dec))
        eng.close()
        return out

    ok, rounds, _ = cycle()
    assert ok and rounds > 10
    monkeypatch.setenv("KB_EMU_REPAIR_OFF", "1")
    broken, _, _ = cycle()
    assert not broken                                   # the overlapped path was taken, and its stale lists alone are wrong
    monkeypatch.delenv("KB_EMU_REPAIR_OFF")
    monkeypatch.setenv("KB_OVERLAP", "0")
    ok_plain, rounds_plain, _ = cycle()
    assert ok_plain and rounds_plain == rounds


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_a_candidate_list_that_never_arrives_breaks_the_chain_and_the_round_runs_again(emulated_engine, monkeypatch, fuse):
    """(fuse: as above.)  k_repair's wait for its lists is bounded: a list that never arrives (here: the emulated arg-max launch of every third overlapped round
    drops its tag) makes it break the chain, the commit kernel behind it skips the round, the host takes the skipped round back, counts the fault
    and keeps every later round of that engine on the plain path — same decisions as the oracle, no hang, no wrong bind."""
    oracle_mod = importlib.import_module("oracle")
    snap = kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.05))
    conf = kbm.conf.load_scheduler_conf()
    o = oracle_mod.Oracle(conf, snap)
    o.run(["allocate", "backfill"])
    monkeypatch.setenv("KB_FUSE_REPAIR", fuse)
    monkeypatch.setenv("KB_EMU_DROP_TAG", "3")
    monkeypatch.setenv("KB_EMU_REPAIR_WAIT_NS", "2e6")
    for _ in range(2):                                   # twice: the fault must not outlive the action that met it
        eng = engine.Engine(conf)
        eng.load(snap)
        dec = eng.run_allocate()
        eng.run_backfill()
        assert np.array_equal(dec, o.decisions()[: len(dec)]) and np.array_equal(eng.binds(), o.binds())
        eng.reset()
        dec2 = eng.run_allocate()
        eng.run_backfill()
        assert np.array_equal(dec2, dec) and np.array_equal(eng.binds(), o.binds())
        eng.close()


def test_one_engine_through_sessions_of_growing_size(emulated_engine):
    """The Go action keeps ONE engine and loads a new session every cycle; clusters grow.  Every buffer sized by the node count must follow
    (round 3: the second stream's matrix rows of the overlapped rounds kept the first session's size — found by review; with a small window and a
    cluster that grows from 8 to 20 000 nodes two matrix rows no longer fit, which scripts/sanitize_cpu.sh's AddressSanitizer pass reports)."""
    oracle_mod = importlib.import_module("oracle")
    conf = kbm.conf.load_scheduler_conf()
    S = kbm.snapshot
    eng = engine.Engine(conf, window=16)
    for n_tasks, n_nodes in ((64, 8), (200, 20000), (400, 300), (300, 45000)):
        snap = S.synth(S.SynthParams(n_tasks=n_tasks, n_nodes=n_nodes, n_queues=4, n_res=2, seed=S.SEED_BASE + 77 + n_nodes))
        o = oracle_mod.Oracle(conf, snap)
        o.run(["allocate", "backfill"])
        eng.load(snap)
        eng.run_allocate()
        eng.run_backfill()
        assert np.array_equal(eng.binds(), o.binds()), (n_tasks, n_nodes)
        o.close()
    eng.close()


@pytest.mark.parametrize("seed", [1, 5, 8, 11, 15])
def test_the_stale_node_name_check_runs_whenever_one_can_have_appeared(emulated_engine, seed):
    """allocate / backfill refuse a session in which a Pending task carries a NodeName (a discarded preempt statement leaves one behind).  The
    engine looks for one only when it can have appeared — once per loaded session, and after every evict action (round 3: it was a scan of all
    tasks in front of every action) — so: refused after the preempt that creates it, again on a second try, fine after kb_session_reset (the
    load-time state was checked by the first allocate), refused again when the preempt is repeated."""
    import test_pyref_vs_oracle as cases
    oracle_mod = importlib.import_module("oracle")
    cfg, snap, _ = cases._evict_case(seed)
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate"])
    e = engine.Engine(cfg)
    e.load(snap)
    for _ in range(2):
        first = e.run_allocate()
        assert np.array_equal(first, o.decisions()[: len(first)]) and np.array_equal(e.binds(), o.binds())
        e.reset()
        e.run(["preempt"])
        for _ in range(2):
            with pytest.raises(engine.EngineError) as err:
                e.run_allocate()
            assert err.value.code == abi.KB_E_UNSUPPORTED and "stale NodeName" in str(err.value)
        with pytest.raises(engine.EngineError):
            e.run_backfill()
        e.reset()
    e.close()
    o.close()


@pytest.mark.parametrize("seed", [1, 5, 8])
def test_the_round_api_refuses_what_the_actions_refuse(emulated_engine, seed):
    """round 5's advisor: kb_round_begin entered allocate / backfill without the guards of kb_run_allocate — a tainted session, a Pending task
    with a stale NodeName behind a discarded preempt statement.  On the task-row split (ShardedCycle with preempt AHEAD of allocate) every rank
    would have diverged from the reference alike, which neither the delta cross-check nor the journal digest can see.  Now the first
    kb_round_begin of the action answers KB_E_UNSUPPORTED like kb_run_allocate does, and goes on refusing until the session is reset."""
    import torch
    import test_pyref_vs_oracle as cases
    distmod = importlib.import_module("kube-batch_amd.dist")
    cfg, snap, _ = cases._evict_case(seed)
    e = engine.Engine(cfg)
    e.load(snap)
    cpu = torch.device("cpu")
    cyc = distmod.ShardedCycle(cfg, snap, backend=distmod.EngineBackend(e, cpu), buffer_device=cpu, min_rows_per_rank=0, actions=["preempt", "allocate", "backfill"])
    for _ in range(2):
        with pytest.raises(engine.EngineError) as err:
            cyc.step()
        assert err.value.code == abi.KB_E_UNSUPPORTED and "stale NodeName" in str(err.value)
        with pytest.raises(engine.EngineError) as err:     # ... and a second kb_round_begin on the same state says the same
            e.round_begin(0)
        assert err.value.code == abi.KB_E_UNSUPPORTED
    e.reset()
    plain = distmod.ShardedCycle(cfg, snap, backend=distmod.EngineBackend(e, cpu), buffer_device=cpu, min_rows_per_rank=0, actions=["allocate"])
    oracle_mod = importlib.import_module("oracle")
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate"])
    assert np.array_equal(plain.step(), o.decisions())      # the same engine, reset: the split runs and equals the oracle
    e.close()
    o.close()


_RESET_RACE_SCRIPT = r"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
oracle_mod = importlib.import_module("oracle")
engine.LIB_PATH, engine._LIB = {so!r}, None
snap = kbm.snapshot.synth(kbm.snapshot.synth_config(3, 0.03))
conf = kbm.conf.load_scheduler_conf()
o = oracle_mod.Oracle(conf, snap)
o.run(["allocate", "backfill"])
e = engine.Engine(conf)
e.load(snap)
dec = e.run(["allocate", "backfill"])
assert np.array_equal(dec, o.decisions())
for _ in range(3):
    e.reset()
    again = e.run(["allocate", "backfill"])
    assert np.array_equal(again, dec), "the cycle after kb_session_reset differs"
    assert np.array_equal(e.binds(), o.binds())
e.close()
print("ok")
"""


def test_overlapped_launches_wait_for_the_copies_kb_session_reset_left_queued(emulated_engine):
    """kb_session_reset restores the node state with device-to-device copies on the engine's stream and returns without waiting.  The first
    overlapped round of the next allocate evaluates on the SECOND stream, which nothing orders behind those copies unless the engine does
    (with the feasibility probe on, the probe's own read-back happened to; KB_PROBE=0 showed it — found on the emulated device with asynchronous
    streams, one run in twenty).  Here the copies are slow (20 ms each): without the wait the second cycle sees the first one's node state."""
    env = dict(os.environ, KB_EMU_ASYNC="1", KB_EMU_D2D_DELAY_US="20000", KB_PROBE="0")
    code = _RESET_RACE_SCRIPT.format(root=os.path.join(HERE, ".."), tests=HERE, so=emulated_engine)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


# ---- proportion's water-fill as a launch (round 3; the default since its first device run: tests/test_gpu_waterfill.py) ----
def _waterfill_counter(so):
    """launches of the emulated k_waterfill so far; the product library (KB_EMU_LIB pointing at it, as round 4's first device call did) has no such counter: None"""
    L = C.CDLL(so)
    if not hasattr(L, "kbemu_waterfill_launches"):
        return lambda: None
    f = L.kbemu_waterfill_launches
    f.restype = C.c_ulonglong
    return f


def _load_and_run(cfg, snap):
    """(error code, None) where kb_session_load refuses, else (None, (decisions, binds, node state, shares))"""
    e = engine.Engine(cfg)
    try:
        e.load(snap)
    except engine.EngineError as err:
        e.close()
        return err.code, None
    try:
        dec = e.run(["allocate", "backfill"])
    except engine.EngineError as err:
        e.close()
        return None, ("run", err.code)
    out = (dec, e.binds(), e.node_state(), e.shares())
    e.close()
    return None, out


def _same(a, b):
    if isinstance(a, tuple):
        return isinstance(b, tuple) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return np.array_equal(a, b)
    return a == b


@pytest.mark.parametrize("block", range(8))
def test_the_device_waterfill_equals_the_host_loop_on_adversarial_snapshots(emulated_engine, monkeypatch, block):
    """kb_waterfill.hpp's steps (the text kb_waterfill.hip's lanes run, here run one after the other by the emulated launch) against
    kb_session.cpp's loop — which tests/test_host_evict_cpu.py holds to tests/pyref.py and the adversarial `-m gpu` cases to the oracle — on
    tests/rawgen.py's snapshots (zero and huge queue weights, queues without jobs, nil scalar maps, requests above and below the cluster's total):
    the same `deserved` bit for bit, the same shares, the same decisions, and KB_E_UNSUPPORTED on exactly the snapshots whose water-fill
    underflows (the reference panics in Resource.Sub there)."""
    import rawgen
    import test_pyref_vs_oracle as cases
    oracle_mod = importlib.import_module("oracle")
    launches = _waterfill_counter(emulated_engine)
    refused = vs_oracle = 0
    for seed in range(block * 50, block * 50 + 50):
        snap = rawgen.raw_snapshot(seed)
        rng = np.random.RandomState(seed)
        wl, wm, wa, wb = [int(x) for x in rng.choice([0, 1, 1, 2, 5], size=4)]
        cfg = kbm.conf.load_scheduler_conf(cases.CONF_TMPL.format(wl=wl, wm=wm, wa=wa, wb=wb))
        monkeypatch.setenv("KB_DEVICE_WATERFILL", "0")     # the host loop (the A/B switch)
        n0 = launches()
        host = _load_and_run(cfg, snap)
        assert launches() == n0
        monkeypatch.delenv("KB_DEVICE_WATERFILL")          # the launch: the default since its first device run (round 4)
        dev = _load_and_run(cfg, snap)
        has_proportion = any(po.name == "proportion" for t in cfg.tiers for po in t)
        assert host[0] == dev[0], (seed, host[0], dev[0])
        if host[0] is None:
            assert n0 is None or launches() == n0 + (1 if has_proportion else 0), seed
            assert _same(host[1], dev[1]), seed
            # ... and the launch against the ORACLE directly (round 5's review: against the host loop alone the launch was compared with the
            # engine itself): `deserved` and both share vectors bit for bit, the decisions, the bind set, the node state
            if not (isinstance(dev[1][0], str) and dev[1][0] == "run"):     # ("run", code): refused at run time, by both alike
                try:
                    o = oracle_mod.Oracle(cfg, snap)
                    o.run(["allocate", "backfill"])
                except RuntimeError:
                    continue                               # the reference panics on this snapshot later in the cycle
                dec, binds, nodes, shares = dev[1]
                for got, want in zip(shares, o.shares()):
                    assert np.array_equal(got, want), seed
                assert np.array_equal(dec, o.decisions()) and np.array_equal(binds, o.binds()), seed
                for got, want in zip(nodes, o.node_state()):
                    assert np.array_equal(got, want), seed
                o.close()
                vs_oracle += 1
        else:
            refused += 1
    assert refused < 50 and vs_oracle >= 10                # the block compared something, with the host loop and with the oracle


def test_the_device_waterfill_on_the_tutorial_example_and_on_128_queues(emulated_engine, monkeypatch):
    """doc/usage/tutorial.md:297-330 (deserved = (3 cpu, 9 Gi) and (6 cpu, 18 Gi) for weights 2 and 4: the reference's own known answer for the
    loop) and BASELINE configs[2]'s 128 queues scaled down (several passes: queues meet their request one after the other), through the launch;
    the session then survives a reset and a second cycle."""
    snapmod = kbm.snapshot
    oracle_mod = importlib.import_module("oracle")
    monkeypatch.delenv("KB_DEVICE_WATERFILL", raising=False)
    launches = _waterfill_counter(emulated_engine)
    Gi = 1 << 30
    pods = [snapmod.Pod("q1", f"p{i}", [{"cpu": "1", "memory": "2Gi"}], group_name="j1") for i in range(5)]
    pods += [snapmod.Pod("q2", f"p{i}", [{"cpu": "1", "memory": "2Gi"}], group_name="j2") for i in range(10)]
    snap = snapmod.flatten(
        nodes=[snapmod.Node("n1", {"cpu": "6", "memory": "15Gi", "pods": "110"}), snapmod.Node("n2", {"cpu": "3", "memory": "12Gi", "pods": "110"})],
        pods=pods, pod_groups=[snapmod.PodGroup("q1", "j1", queue="queue1"), snapmod.PodGroup("q2", "j2", queue="queue2")],
        queues=[snapmod.Queue("queue1", 2), snapmod.Queue("queue2", 4)])
    cfg = kbm.conf.load_scheduler_conf()
    n0 = launches()
    e = engine.Engine(cfg)
    e.load(snap)
    assert n0 is None or launches() == n0 + 1
    des = e.shares()[2]
    assert [des[0, 0], des[1, 0]] == [3000.0, 9.0 * Gi] and [des[0, 1], des[1, 1]] == [6000.0, 18.0 * Gi]
    e.close()

    snap = snapmod.synth(snapmod.synth_config(3, 0.05))
    o = oracle_mod.Oracle(cfg, snap)
    o.run(["allocate", "backfill"])
    e = engine.Engine(cfg)
    e.load(snap)
    for _ in range(2):
        dec = e.run(["allocate", "backfill"])
        assert np.array_equal(dec, o.decisions()) and np.array_equal(e.binds(), o.binds())
        for a, b in zip(e.shares(), o.shares()):
            assert np.array_equal(a, b)
        e.reset()
    e.close()


# ---- kb_session_load's host passes split over threads (kb_session.cpp par_parts) against the same passes on one thread ------------------------
def _session_tables(cfg, snap):
    """(error code, None) or (None, everything a cycle shows of the session: decisions, bind set, node state, shares, counters — the counters
    carry the number of distinct shapes, the matrix rows evaluated per shape id)"""
    e = engine.Engine(cfg)
    try:
        e.load(snap)
    except engine.EngineError as err:
        e.close()
        return (err.code, str(err)), None
    try:
        dec = e.run(["allocate", "backfill"])
    except engine.EngineError as err:                    # outside the exact envelope at run time: the same refusal either way
        e.close()
        return None, ("run", err.code, str(err))
    st = e.stats()
    out = (dec, e.binds(), e.node_state(), e.shares(), tuple(sorted((k, v) for k, v in st.items() if isinstance(v, int))))
    e.close()
    return None, out


@pytest.mark.parametrize("block", range(4))
def test_session_build_split_over_host_threads_is_the_one_thread_build(emulated_engine, monkeypatch, block):
    """The stretch heads are interned per part of the task range and the parts' distinct keys merged in part order; the per-queue request is summed
    per range of jobs when every addend is a whole number.  KB_HOST_SPLIT_WORDS=1 forces eight parts on sessions of a few dozen tasks (parts
    without a head, stretches across part borders, parts of one task): the same refusal (code AND text: the lowest task's) or the same cycle
    bit for bit, counters included, as the one-thread passes — on tests/rawgen.py's snapshots (fractional quantities, nil scalar maps, zero
    weights, invalid inputs)."""
    import rawgen
    import test_pyref_vs_oracle as cases
    cfg = kbm.conf.load_scheduler_conf(cases.CONF_TMPL.format(wl=1, wm=0, wa=1, wb=1))
    compared = refused = 0
    for seed in range(block * 40, block * 40 + 40):
        snap = rawgen.raw_snapshot(seed)
        monkeypatch.setenv("KB_HOST_SPLIT_WORDS", str(1 << 40))
        one = _session_tables(cfg, snap)
        monkeypatch.setenv("KB_HOST_SPLIT_WORDS", "1")
        split = _session_tables(cfg, snap)
        monkeypatch.delenv("KB_HOST_SPLIT_WORDS")
        assert one[0] == split[0], (seed, one[0], split[0])
        if one[0] is None:
            assert _same(one[1], split[1]), seed
            compared += 0 if isinstance(one[1][0], str) else 1
        else:
            refused += 1
    assert compared >= 10, (compared, refused)


def test_split_session_build_on_scaled_configurations_and_the_lowest_refusal(emulated_engine, monkeypatch, oracle_mod):
    """Scaled BASELINE configurations (R = 2 and R = 16, inter-pod terms, host-port masks of several words) through the split build against the
    ORACLE; a fractional request and a total at 2^53 (both take the in-order sum); two invalid tasks in different parts (the lower one's
    refusal, as one pass over all tasks reports it)."""
    import test_gpu_interpod as gi
    import test_gpu_wideports as gw
    snapmod = kbm.snapshot
    monkeypatch.setenv("KB_HOST_SPLIT_WORDS", "1")
    cfg = kbm.conf.load_scheduler_conf()
    snaps = [snapmod.synth(snapmod.synth_config(3, 0.03)), snapmod.synth(snapmod.synth_config(4, 0.02))]
    frac = snapmod.synth(snapmod.synth_config(3, 0.02))
    frac.task_resreq[0, 5::7] += 0.5; frac.task_init_resreq[0, 5::7] += 0.5          # half a milli-cpu: not a whole number
    huge = snapmod.synth(snapmod.synth_config(3, 0.02))
    pend = np.nonzero(huge.task_status == 0)[0][:64]
    huge.task_resreq[1, pend] = float(1 << 47); huge.task_init_resreq[1, pend] = float(1 << 47)   # 64 x 2^47 = 2^53 bytes on the queues' requests
    snaps += [frac, huge]
    for snap in snaps:
        o = oracle_mod.Oracle(cfg, snap)
        o.run(["allocate", "backfill"])
        e = engine.Engine(cfg)
        e.load(snap)
        dec = e.run(["allocate", "backfill"])
        assert np.array_equal(dec, o.decisions()) and np.array_equal(e.binds(), o.binds())
        for a, b in zip(e.shares(), o.shares()):
            assert np.array_equal(a, b)
        for a, b in zip(e.node_state(), o.node_state()):
            assert np.array_equal(a, b)
        e.close()
    bad = snapmod.synth(snapmod.synth_config(3, 0.02))
    T = bad.task_resreq.shape[1]
    heads = np.nonzero(np.diff(bad.task_job.astype(np.int64), prepend=-1))[0]
    lo, hi = int(heads[1]), int(heads[-2])
    assert lo < T // 8 and hi > 7 * T // 8
    bad.task_init_resreq[0, lo] = bad.task_resreq[0, lo] - 1.0       # "InitResreq < Resreq" in the first part
    bad.task_resreq[1, hi] = -5.0                                    # "negative request" in the last
    for words in ("1", str(1 << 40)):
        monkeypatch.setenv("KB_HOST_SPLIT_WORDS", words)
        e = engine.Engine(cfg)
        with pytest.raises(engine.EngineError) as ei:
            e.load(bad)
        assert "InitResreq < Resreq" in str(ei.value), (words, str(ei.value))
        e.close()


# ---- a run of identical rows committed by ONE selection (the emulated commit launch, KB_EMU_RUN_SELECT=1) ------------------
_RUN_SELECT_SCRIPT = r"""
import importlib, os, sys
sys.path[:0] = [{root!r}, {tests!r}]
import ctypes as C
import numpy as np
engine = importlib.import_module("kube-batch_amd.engine")
kbm = importlib.import_module("kube-batch_amd")
import oracle
import bench
engine.LIB_PATH, engine._LIB = {so!r}, None
L = C.CDLL({so!r})
sel, lanes, steps = L.kbemu_selected_rows, L.kbemu_select_lanes, L.kbemu_select_steps
sel.restype = lanes.restype = steps.restype = C.c_ulonglong
wrong = total = 0
for idx, survey, scale in ((3, False, 0.05), (3, True, 0.05), (4, False, 0.03), (2, False, 1.0)):
    conf = kbm.conf.load_scheduler_conf(bench.BINPACK_CONF) if idx == 4 else kbm.conf.load_scheduler_conf()
    p = kbm.snapshot.synth_config(idx, scale)
    if survey:
        p.node_cpu_cores = (16, 32, 64, 96, 128)
        p.node_mem_gib = (64, 128, 256, 512)
    snap = kbm.snapshot.synth(p)
    o = oracle.Oracle(conf, snap)
    o.run(["allocate", "backfill"])
    e = engine.Engine(conf)
    e.load(snap)
    s0, l0, t0 = sel(), lanes(), steps()
    dec = e.run(["allocate", "backfill"])
    same = dec.shape == o.decisions().shape and np.array_equal(dec, o.decisions()) and np.array_equal(e.binds(), o.binds())
    if same:
        same = all(np.array_equal(a, b) for a, b in zip(e.node_state(), o.node_state()))
    wrong += 0 if same else 1
    total += 1
    print("case", idx, survey, "decisions", len(dec), "rows by selection", sel() - s0, "equal" if same else "DIFFERENT",
          "| node sequences walked", lanes() - l0, "evaluations", steps() - t0)
    e.close()
print("wrong", wrong, "of", total)
"""


@pytest.mark.parametrize("mode", ["1", "2"])
def test_runs_of_identical_rows_committed_by_one_selection(emulated_engine, mode):
    """tests/run_selection_model.py's claim inside the engine's own launch contract: the emulated commit launch takes every run of rows that differ in
    the task id only through ONE selection — each candidate node's own key sequence against its own state (the product's per-pair arithmetic, epsilon
    compares, scalar dimensions, ports, pod caps), prefix minima, the first r of all (node, step) entries — and the cycle still equals the oracle
    (decisions in order, binds, final node state) on the bench configurations, where most rows are committed that way.  (The whole emulated suite
    also passes with the switch on: scripts/sanitize_cpu.sh.)  Mode 2 is the negative control — the real key in the place of the prefix minimum,
    i.e. the premise 'a node's keys only fall', which Balanced breaks: at least one configuration must come out different."""
    env = dict(os.environ, KB_EMU_RUN_SELECT=mode)
    code = _RUN_SELECT_SCRIPT.format(root=os.path.join(HERE, ".."), tests=HERE, so=emulated_engine)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("case")]
    assert len(lines) == 4, r.stdout
    for l in lines:
        f = l.split()
        assert int(f[8]) * 2 > int(f[4]), l                     # most rows of the cycle went through the selection
    if mode == "1":
        assert "wrong 0 of 4" in r.stdout, r.stdout
    else:
        assert "wrong 0 of 4" not in r.stdout, r.stdout
