"""bench.py's own control flow — argument handling, the timed loop, the roofline / cpu_baseline / verification objects, the one JSON line the
driver parses — run on the CPU: the engine library is the emulated build of tests/test_emu_engine_cpu.py and torch.cuda is answered by a
stand-in.  The NUMBERS of such a run mean nothing (an emulated device has no bandwidth); what is checked is that the line has every
field of the bench contract, that the bind set is verified against the oracle in the same run, and that a flag cannot break the default."""
import importlib
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE)]
import test_emu_engine_cpu as emu  # noqa: E402

engine = importlib.import_module("kube-batch_amd.engine")


def _run_bench(monkeypatch, argv):
    import torch
    bench = importlib.import_module("bench")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(engine, "LIB_PATH", emu.build_emulated_library())
    monkeypatch.setattr(engine, "_LIB", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line
    return json.loads(lines[0])


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def test_default_line_has_every_field_and_verifies_itself(monkeypatch):
    out = _run_bench(monkeypatch, ["--scale", "0.02", "--steps", "2", "--warmup", "1"])
    for k in CONTRACT:
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["unit"] == "evals/s" and out["vs_baseline"] is None
    assert out["config"]["workload"].startswith("BASELINE configs[2]") and "allocate+backfill" in out["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"], k
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["traffic"] is None      # a scaled run matches no committed PMC profile
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"], k
    assert out["verified_bind_set_equals_oracle"] is True and out["verified_evals_equal_oracle"] is True
    assert out["binds"] > 0 and out["evals_per_step"] > 0
    rc = out["roofline_commit"]      # the commit kernels' own figure: live time per committed row + the committed PMC counters of the same command
    for k in ("commit_ms_per_step", "ns_per_committed_row", "cycles_per_committed_row_at_2.4GHz", "us_per_round", "streaming_equivalent_frac_of_hbm_peak", "counters"):
        assert k in rc, k
    assert rc["ns_per_committed_row"] > 0 and (rc["counters"] is None or any("k_commit" in k for k in rc["counters"]))


def test_default_run_carries_the_unfriendly_inputs(monkeypatch):
    """the driver's default line also times and verifies SURVEY 8d's literal node sizes and BASELINE configs[3] (variants), reports the
    oracle's incremental mode beside the 16-thread loop, and the matrix kernel's own evaluation rates (scaled here; full size on the GPU box)"""
    monkeypatch.setenv("KB_BENCH_VARIANTS", "1")
    out = _run_bench(monkeypatch, ["--scale", "0.02", "--steps", "1", "--warmup", "0"])
    assert set(out["variants"]) == {"survey_nodes", "config4_binpack", "config2"}      # (+ diverse, config5, config5_preempt at full size: their digests are the full size's)
    for v in out["variants"].values():
        assert v["verified"] is True and v["ms_per_step"] > 0 and v["binds"] > 0 and v["evals_per_s"] > 0
        assert v["roofline"]["frac"] >= 0 and v["roofline"]["bytes_per_launch"] > 0      # every variant carries its own roofline fraction (round 5's review)
    assert out["cpu_baseline_incremental"]["threads"] == 1 and out["cpu_baseline_incremental"]["value"] > 0
    for k in ("roofline_eval", "roofline_eval_all_rows"):
        assert out[k]["bound"] == "hbm" and out[k]["kernel"].startswith("k_matrix") and out[k]["frac"] >= 0


@pytest.mark.parametrize("argv", [["--config", "4", "--scale", "0.02", "--steps", "1", "--warmup", "0"],
                                  ["--config", "3", "--scale", "0.02", "--diverse", "--steps", "1", "--no-cpu-baseline", "--verify"],
                                  ["--config", "5", "--scale", "0.004", "--steps", "1", "--preempt", "--no-cpu-baseline", "--verify"],
                                  ["--config", "3", "--scale", "0.02", "--steps", "2", "--preempt"]])
def test_variants_produce_a_verified_line(monkeypatch, argv):
    out = _run_bench(monkeypatch, argv)
    assert out["verified_bind_set_equals_oracle"] is True and out["verified_evals_equal_oracle"] is True, out
    if "--preempt" in argv:
        assert out["verified_evictions_equal_oracle"] is True and out["verified_journal_equals_oracle"] is True and out["journal_entries"] >= 0
        assert "allocate+backfill+preempt" in out["config"]["workload"]


def test_smoke_entry_point_runs_end_to_end(monkeypatch, capsys):
    """__graft_entry__.smoke() is what the driver runs on the MI355X before the bench: its own Python must not be what fails there."""
    monkeypatch.setattr(engine, "LIB_PATH", emu.build_emulated_library())
    monkeypatch.setattr(engine, "_LIB", None)
    entry = importlib.import_module("__graft_entry__")
    entry.smoke()
    assert "smoke ok" in capsys.readouterr().out


def test_two_ranks_under_torch_distributed_run_print_one_line(tmp_path):
    """The driver's N > 1 launch — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...` — with two ranks here: gloo stands in for RCCL (KB_DIST_BACKEND), the emulated library for the engine
    (tests/host_harness/bench_emu_launcher.py).  Rank 0 prints ONE line with BOTH multi-GPU answers (round 6): north_star's task-row split of one
    session — the line's value, ms_per_step and scaling ("strong"), its decisions held to the golden digest of rank 0's snapshot, the collectives'
    own times — and, under `sessions`, one independent session per rank (each held to tests/golden/bench_rank_digests.json, the slowest rank's
    per-session rate and the aggregate)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, KB_EMU_LIB=emu.build_emulated_library(), KB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "KB_DIST_MODE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "host_harness", "bench_emu_launcher.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.02"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["multi_gpu_mode"].startswith("north_star's task-row split")
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None        # reported at N = 1 only
    assert d["roofline"]["bound"] == "hbm"
    sh, se = d["sharded"], d["sessions"]
    assert sh["verified"] is True and sh["scaling"] == "strong" and sh["dist_backend"] == "gloo" and sh["ranks"] == 2 and sh["ranks_seen_by_rccl"] == 0
    assert sh["value"] == d["value"] and abs(sh["ms_per_step"] - d["ms_per_step"]) < 1e-9 and sh["rounds_per_step"] > 0
    assert abs(d["value"] - d["evals_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]       # the split's rate: ONE session's evaluations over the job's time
    assert sh["allreduce_us_per_round"] > 0 and sh["deferred_delta_checks_per_step"] > 0
    assert abs(sh["rounds_that_exchanged_lists_per_step"] + sh["rounds_every_rank_evaluated_alone_per_step"] - sh["rounds_per_step"]) < 1e-9      # a round either exchanges its lists or every rank builds them all
    assert (sh["allgather_us_per_round"] is None) == (sh["rounds_that_exchanged_lists_per_step"] == 0)
    assert se["verified"] is True and se["scaling"] == "weak" and se["value"] > 0
    # sessions: one session's rate (the slowest rank's), never the ranks' sum; the sum is the aggregate
    assert se["aggregate_evals_per_s"] >= 1.5 * se["value"] and abs(se["sessions_per_s"] - 2 * 2 / (se["ms_per_step"] * 2e-3)) <= 1e-6 * se["sessions_per_s"]


def test_two_ranks_sessions_mode_alone(tmp_path):
    """KB_DIST_MODE=sessions: round 5's default line (every rank its own snapshot, value = the slowest rank's per-session rate, "weak")"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, KB_EMU_LIB=emu.build_emulated_library(), KB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", KB_DIST_MODE="sessions")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "host_harness", "bench_emu_launcher.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.02"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["sharded"] is None and d["sessions"] is None
    assert d["multi_gpu_mode"].startswith("one independent session per GPU") and d["sessions_verified_against_golden_digests"] is True
    assert d["value"] <= 1.05 * d["evals_per_step"] / (d["ms_per_step"] * 1e-3) * 1.5
    assert d["aggregate_evals_per_s"] >= 1.5 * d["value"] and abs(d["sessions_per_s"] - 2 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["sessions_per_s"]


def test_two_identical_replicas_still_agree(tmp_path):
    """KB_DIST_MODE=replicas (round 3's default): the same session on both ranks, digests compared across them; `value` is ONE session's rate."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, KB_EMU_LIB=emu.build_emulated_library(), KB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", KB_DIST_MODE="replicas")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "host_harness", "bench_emu_launcher.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.02"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["replicas_agree"] is True and d["multi_gpu_mode"].startswith("the same session on every GPU")
    assert abs(d["value"] - d["evals_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_two_ranks_when_the_split_cannot_be_set_up(tmp_path):
    """The task-row split has never run on more than one GPU (no such node was available): if its set-up fails on the driver's 8-GPU box, the line still
    carries the sessions mode — verified, "weak", its rate as `value` — and `sharded.error` says what went wrong; the exit code stays 0."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, KB_EMU_LIB=emu.build_emulated_library(), KB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", KB_EMU_BREAK_SHARDED="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "KB_DIST_MODE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "host_harness", "bench_emu_launcher.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.02"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "broken on purpose" in d["sharded"]["error"] and d["sharded"]["verified"] is None
    assert d["sessions"]["verified"] is True and d["sessions_verified_against_golden_digests"] is True
    assert d["multi_gpu_mode"].startswith("one independent session per GPU")


def test_two_ranks_when_the_split_never_answers(tmp_path):
    """A collective of the split that never completes (the one failure an exception handler cannot see) must not cost the sessions mode's finished answer:
    bench.py's watchdog prints the line with the sessions figures as its value and `sharded.error` saying that the phase was abandoned; every rank exits 0."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, KB_EMU_LIB=emu.build_emulated_library(), KB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", KB_EMU_HANG_SHARDED="1", KB_SHARDED_LIMIT_S="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "KB_DIST_MODE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "host_harness", "bench_emu_launcher.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.02"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["value"] == d["sessions"]["value"]
    assert "abandoned by bench.py's watchdog" in d["sharded"]["error"] and d["sharded"]["verified"] is None
    assert d["sessions"]["verified"] is True and d["sessions_verified_against_golden_digests"] is True
    for k in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
