#!/usr/bin/env python
"""bench.py — pod-node scoring evals/s (+ binds/s) of the allocate+backfill cycle on the BASELINE workload.

A "step" is one full scheduling cycle of the hot path over one synthetic session snapshot that is already
resident in HBM: kb_session_reset (device-to-device restore of the pristine node/task state) ->
kb_run_allocate -> kb_run_backfill.  The workload is BASELINE.json configs[2] — 100k tasks x 10k nodes, gang
minAvailable + DRF + proportion across 128 queues, R=2 — the configuration the headline metric is quoted on.

`value` = (task,node) predicate+score evaluations the reference algorithm performs for that cycle (N per popped
task in allocate, nodes visited until the first fit in backfill: SURVEY.md §8d) x steps / wall time.

One JSON line on rank 0; see DESIGN.md §7 for the roofline accounting (M: 2.125 B per evaluation written by the
matrix kernel + inputs once) and the cpu_baseline definition.
"""
import argparse
import importlib
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); keep whatever the launcher set
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
RANK_SEED_STRIDE = 7919   # --gpus N: rank k schedules the snapshot of seed + RANK_SEED_STRIDE * k (tests/golden/make_bench_rank_digests.py)

BINPACK_CONF = """
actions: "allocate, backfill"
tiers:
- plugins:
  - name: priority
  - name: gang
- plugins:
  - name: drf
  - name: predicates
  - name: proportion
  - name: nodeorder
    arguments:
      leastrequested.weight: 0
      mostrequested.weight: 5
      balancedresource.weight: 1
"""

PREEMPT_CONF = """
actions: "allocate, backfill, preempt"
tiers:
- plugins:
  - name: priority
  - name: gang
  - name: conformance
- plugins:
  - name: drf
  - name: predicates
  - name: proportion
  - name: nodeorder
"""


def _k(n):
    return f"{n // 1000}k" if n >= 1000 and n % 1000 == 0 else str(n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="BASELINE config index (2..5)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--diverse", action="store_true", help="stress variant: every job draws its own request (thousands of distinct task shapes)")
    ap.add_argument("--survey-nodes", action="store_true", help="variant: node sizes as SURVEY.md 8d lists them (16..128 cores, 64..512 GiB): capacity ~4x the "
                    "demand instead of the 1.3x pressure the same paragraph asks for (the default generator keeps the pressure)")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--commit-batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-tasks", type=int, default=0, help="0 = the whole allocate action (a few seconds on 16 threads)")
    ap.add_argument("--verify", action="store_true", help="with --no-cpu-baseline: still compare the bind set with the oracle after the timed region")
    ap.add_argument("--spin-wait", action="store_true", help="KB_FLAG_SPIN_WAIT: the calling thread spins for the whole wait for a round (round 5's default) instead of "
                    "giving the core up between polls after a short spin (N = 1 only)")
    ap.add_argument("--preempt", action="store_true", help="BASELINE configs[4] names allocate + backfill + preempt: a step becomes reset -> allocate -> backfill -> "
                    "preempt under the default tiers plus conformance (scripts/time_preempt.py's configuration); with --gpus N every replica runs the evict action and the journals are compared (kube-batch_amd/dist.py)")
    args = ap.parse_args()

    import numpy as np
    import torch
    kbm = importlib.import_module("kube-batch_amd")
    engine = importlib.import_module("kube-batch_amd.engine")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if args.gpus > 1 and world == 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the engine has no CPU path", file=sys.stderr)
        sys.exit(2)
    local_rank %= torch.cuda.device_count()      # a launcher that masks devices per rank leaves ordinal 0 only
    torch.cuda.set_device(local_rank)
    # KB_BENCH_SHARDED=1 (under torch.distributed.run with one process): the multi-GPU code path — round-granular engine calls,
    # RCCL collectives on the engine's stream — on a one-GPU box, with KB_DIST_ALWAYS_COLLECT=1 the collectives really run
    force_sharded = world == 1 and os.environ.get("KB_BENCH_SHARDED") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or force_sharded:
        import torch.distributed as dist
        # "nccl" = RCCL over xGMI.  KB_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses two ranks per device), which is how the
        # N > 1 code path is exercised on a one-GPU box
        dist.init_process_group(os.environ.get("KB_DIST_BACKEND", "nccl"))

    conf = kbm.conf.load_scheduler_conf()          # pkg/scheduler/util.go:31-42 default: allocate, backfill; all six plugins
    weights = "least 1, most 0, balanced 1"
    if args.config == 4:                           # BASELINE configs[3] "binpack weighted scoring" (SURVEY.md §8d synthetic inputs)
        conf = kbm.conf.load_scheduler_conf(BINPACK_CONF)
        weights = "least 0, most 5, balanced 1"
    if args.preempt:
        nodeorder_args = BINPACK_CONF.split("  - name: nodeorder\n")[1] if args.config == 4 else ""
        conf = kbm.conf.load_scheduler_conf(PREEMPT_CONF + nodeorder_args)
    params = kbm.snapshot.synth_config(args.config, args.scale)
    if args.diverse:
        params.diverse_requests = True
    if args.survey_nodes:
        params.node_cpu_cores = (16, 32, 64, 96, 128)
        params.node_mem_gib = (64, 128, 256, 512)
    snap = kbm.snapshot.synth(params)
    actions = ["allocate", "backfill"] + (["preempt"] if args.preempt else [])

    dist_mode = None
    rank_digest_expected = None
    sessions_block = None       # N > 1, both answers in one invocation: the sessions mode's figures beside the task-row split's
    if world > 1 or force_sharded:
        # N > 1, ONE command, BOTH answers (round 6; KB_DIST_MODE unset or "both"):
        #   sharded   north_star's task-row split of ONE session — every rank the same snapshot, the window's matrix rows sharded, candidate lists
        #             all-gathered, the commit replicated, per-node deltas all-reduced over RCCL per round.  Exact, "strong", and — because the commit
        #             (80 % of a round) is a sequential dependency every rank repeats — at best as fast as one GPU.  This is what north_star defines
        #             the multi-GPU metric on: it is the line's `value`, `ms_per_step` and `scaling`.
        #   sessions  rank k schedules its OWN snapshot (the generator's seed + k; rank 0's is the N = 1 workload) through the single-GPU fast path, no
        #             data-path collective; every rank's decisions are held to a committed golden digest (tests/golden/bench_rank_digests.json, the
        #             oracle's).  Reported beside it under `sessions`: the slowest rank's per-session rate and the aggregate over the ranks.
        # KB_DIST_MODE=sessions / sharded: that mode alone (round 5's lines); KB_DIST_MODE=replicas: round 3's mode, every rank the SAME session
        # through the fast path, digests compared across ranks.
        distmod = importlib.import_module("kube-batch_amd.dist")
        mode_env = os.environ.get("KB_DIST_MODE", "both")
        dist_mode = "sharded" if (force_sharded or mode_env in ("sharded", "both")) else ("replicas" if mode_env == "replicas" else "sessions")
        golden_key = f"config{args.config}_scale{args.scale:g}{'_survey' if args.survey_nodes else ''}{'_diverse' if args.diverse else ''}{'_preempt' if args.preempt else ''}"
        try:
            # the committed digests are the oracle's for the stock configuration of each config index under allocate + backfill: anything else
            # (another action list) has no entry and reports `verified` = null
            golden_ranks = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_rank_digests.json"))).get(golden_key, {})
        except OSError:
            golden_ranks = {}

        def sessions_phase(snap_mine):
            """W warm-up + K timed cycles of this rank's OWN session (barrier + synchronize on both sides, MAX over the ranks), one more cycle held to
            the golden digest -> (the `sessions` object, the runner)"""
            r_ = distmod.ReplicatedCycle(conf, snap_mine, device=local_rank, window=args.window, commit_batch=args.commit_batch, actions=actions)
            for _ in range(args.warmup):
                r_.step(verify=False)
            st0 = r_.engine.stats()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            q0 = time.perf_counter()
            for _ in range(args.steps):
                r_.step(verify=False)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el_local = time.perf_counter() - q0
            st1 = r_.engine.stats()
            ev = st1["evals"] - st0["evals"]
            dec_last = r_.step(verify=False)                 # outside the timed region: its digest against the committed one of THIS rank's snapshot
            nb = int((r_.engine.binds() != kbm.abi.KB_NONE).sum())
            mine = distmod.ReplicatedCycle.digest(dec_last, r_.engine.binds(), r_.engine.journal() if args.preempt else None, r_.engine.evictions() if args.preempt else None)
            want = golden_ranks.get(str(rank))
            ok_here = 1 if (want is not None and int(want) == mine) else (0 if want is not None else -1)
            if ok_here == 0:
                print(f"bench.py: rank {rank}: decisions digest {mine} differs from the golden digest {want}", file=sys.stderr)
            dev = "cuda" if (world > 1 and dist.get_backend() == "nccl") else "cpu"
            stat = torch.tensor([ev / el_local, -float(ok_here), float(ev), float(nb), -el_local], dtype=torch.float64, device=dev)
            if world > 1:
                mn = stat.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
                sm = stat.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
                mx = stat.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            else:
                mn = sm = mx = stat
            el = -float(mn[4].item())                        # MAX over the ranks of the timed region
            worst = -int(mx[1].item())                       # MIN over the ranks of ok_here
            verified = None if worst < 0 else bool(worst == 1)
            blk = {"mode": "one independent session per GPU (rank k: seed + k), no data-path collective", "ms_per_step": el * 1e3 / args.steps,
                   "value": None if verified is False else float(mn[0].item()), "value_is": "the slowest rank's per-session evals/s (comparable with the N = 1 line)",
                   "aggregate_evals_per_s": float(sm[2].item()) / el, "sessions_per_s": world * args.steps / el,
                   "aggregate_binds_per_s": float(sm[3].item()) * args.steps / el, "scaling": "weak", "verified": verified,
                   "verified_with": "tests/golden/bench_rank_digests.json (the oracle's digest of every rank's own snapshot)"}
            return blk, r_

        if mode_env == "both" and not force_sharded:
            import dataclasses
            mine_params = dataclasses.replace(params, seed=params.seed + RANK_SEED_STRIDE * rank)
            sessions_block, sess_runner = sessions_phase(kbm.snapshot.synth(mine_params) if rank > 0 else snap)
            sess_runner.engine.close()
            del sess_runner
        if dist_mode == "sessions" and rank > 0:
            params.seed = params.seed + RANK_SEED_STRIDE * rank
            snap = kbm.snapshot.synth(params)
        if dist_mode == "sessions":
            rank_digest_expected = golden_ranks.get(str(rank))
        sharded_setup_error = None
        sharded_watch = None
        if dist_mode == "sharded" and sessions_block is not None and world > 1:
            # A failure of the split that is NOT an exception — a collective that never completes — would cost the whole line, the sessions mode's
            # finished answer with it.  The sessions phase above is done: if the split's phase (set-up, warm-up, timed cycles, the verifying cycle)
            # is not through within a generous bound, rank 0 prints the line with the sessions mode's figures as its value and `sharded.error`
            # saying so, and every rank leaves with exit code 0 (a thread stuck in a collective cannot be interrupted: os._exit).
            # KB_SHARDED_LIMIT_S: the bound in seconds (default: 240 s + 400 x the time the sessions phase took per cycle, per cycle to run).
            import threading
            limit_s = float(os.environ.get("KB_SHARDED_LIMIT_S", "0")) or (240.0 + 0.4 * sessions_block["ms_per_step"] * (args.warmup + args.steps + 2))
            sharded_watch = threading.Event()

            def _abandon_the_split():
                if sharded_watch.wait(limit_s):
                    return
                why = (f"no answer within {limit_s:.0f} s of the split's phase (set-up + {args.warmup} warm-up + {args.steps} timed cycles + the verifying one): "
                       "abandoned by bench.py's watchdog; the line's value is the sessions mode's")
                print(f"bench.py: rank {rank}: {why}", file=sys.stderr, flush=True)
                if rank == 0:
                    print(json.dumps({
                        "metric": f"pod-node scoring evals/sec + binds/sec, {_k(snap.n_tasks)} tasks x {_k(snap.n_nodes)} nodes snapshot",
                        "value": sessions_block["value"], "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                        "ms_per_step": sessions_block["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                        "dtype": "f64+i64", "data": "synthetic",
                        "config": {"workload": f"BASELINE configs[{args.config - 1}]: {snap.n_tasks} tasks x {snap.n_nodes} nodes, {snap.n_jobs} gang jobs, "
                                               f"{snap.n_queues} queues, R={snap.n_res}, {'+'.join(actions)}, plugins priority,gang,drf,predicates,proportion,nodeorder ({weights})",
                                   "window": args.window or 256, "scale": args.scale},
                        "multi_gpu_mode": "one independent session per GPU (rank k: seed + k), no data-path collective; value = the slowest rank's per-session rate",
                        "sharded": {"error": why, "verified": None}, "sessions": sessions_block,
                        "sessions_verified_against_golden_digests": sessions_block["verified"],
                        "aggregate_evals_per_s": sessions_block["aggregate_evals_per_s"], "sessions_per_s": sessions_block["sessions_per_s"],
                        "dist_backend": dist.get_backend(), "roofline": None, "cpu_baseline": None,
                        "roofline_note": "N > 1: the roofline and the CPU baseline are the N = 1 line's"}), flush=True)
                sys.stdout.flush()
                os._exit(0)

            threading.Thread(target=_abandon_the_split, daemon=True, name="kb-sharded-watchdog").start()
        if dist_mode == "sharded":
            # The split's set-up (its streams, the engine on torch's stream, one whole cycle with its collectives) has only ever run at world size
            # <= 2 (no multi-GPU node was available to any round): if it fails HERE — symmetric failures: every rank runs the same code on the same
            # session — the line still carries the sessions mode's figures as its value, and `sharded` says what went wrong
            try:
                runner = distmod.ShardedCycle(conf, snap, device=local_rank, window=args.window, commit_batch=args.commit_batch, actions=actions)
                runner.step()
            except Exception as err:     # noqa: BLE001 — whatever it is, it is reported in the line
                sharded_setup_error = f"{type(err).__name__}: {err}"
                if sharded_watch is not None:
                    sharded_watch.set()          # an exception is an answer: the sessions mode runs again below as the line's own
                print(f"bench.py: rank {rank}: the task-row split failed during set-up ({sharded_setup_error}); reporting the sessions mode", file=sys.stderr)
                dist_mode = "sessions"
                if rank > 0:
                    import dataclasses
                    snap = kbm.snapshot.synth(dataclasses.replace(params, seed=params.seed + RANK_SEED_STRIDE * rank))
                rank_digest_expected = golden_ranks.get(str(rank))
        if dist_mode != "sharded":
            runner = distmod.ReplicatedCycle(conf, snap, device=local_rank, window=args.window, commit_batch=args.commit_batch, actions=actions)
        step = (lambda: runner.step(verify=False)) if dist_mode != "sharded" else runner.step
        eng = runner.engine
    else:
        eng = engine.Engine(conf, device=local_rank, window=args.window, commit_batch=args.commit_batch, flags=kbm.abi.FLAG_SPIN_WAIT if args.spin_wait else 0)
        eng.load(snap)
        # what the Go shim pays every cycle (host pre-processing + H2D into a live engine, warm buffers): kb_session_load returns behind its
        # own synchronisation, so the call is the cost.  Seven loads: the median is reported, the maximum beside it.
        load_samples = []
        for _ in range(7):
            tl0 = time.perf_counter()
            eng.load(snap)
            load_samples.append((time.perf_counter() - tl0) * 1e3)
        load_ms = sorted(load_samples)[len(load_samples) // 2]

        def step():
            eng.reset()
            for a in actions:
                getattr(eng, "run_" + a)()

    if world > 1 or force_sharded:
        load_ms, load_samples = None, []

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    s0 = eng.stats()
    coll0 = dict(runner.collective_times(), replicated=runner.replicated_rounds, checks=runner.deferred_checks) if dist_mode == "sharded" else None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    s1 = eng.stats()
    elapsed = t1 - t0
    elapsed_local = elapsed
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    binds = eng.binds()
    n_binds = int((binds != kbm.abi.KB_NONE).sum())
    d = {k: s1[k] - s0[k] for k in s1}
    evals = d["evals"]
    replicas_agree = None
    sessions_verified = None
    aggregate = None
    value = evals / elapsed                                   # one session's rate (N = 1, sharded: the job's)
    sharded_block = None
    if dist_mode == "sharded":
        # outside the timed region: one more cycle; every rank decided the SAME session, whose decisions + bind set are held to the committed digest of
        # rank 0's snapshot (the oracle's); the ranks' verdicts are reduced so that the line says what ALL of them found
        coll1 = dict(runner.collective_times(), replicated=runner.replicated_rounds, checks=runner.deferred_checks)
        coll = {k: (coll1[k] - coll0[k] if isinstance(coll1[k], (int, float)) else coll1[k]) for k in coll1}      # the timed region's own
        dec_last = runner.step()
        mine = distmod.ReplicatedCycle.digest(dec_last, eng.binds(), eng.journal() if args.preempt else None, eng.evictions() if args.preempt else None)
        want = golden_ranks.get("0")
        ok_here = 1 if (want is not None and int(want) == mine) else (0 if want is not None else -1)
        if ok_here == 0:
            print(f"bench.py: rank {rank}: the task-row split's decisions digest {mine} differs from the golden digest {want}", file=sys.stderr)
        if world > 1:
            tv = torch.tensor([float(ok_here)], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tv, op=dist.ReduceOp.MIN)
            ok_here = int(tv.item())
        sharded_verified = None if ok_here < 0 else bool(ok_here == 1)
        rounds_t = max(1.0, d["rounds"])
        sharded_block = {"mode": "north_star's task-row split of ONE session: matrix rows sharded, lists all-gathered, commit replicated, per-node deltas all-reduced per round",
                         "ms_per_step": elapsed * 1e3 / args.steps, "value": None if sharded_verified is False else value, "scaling": "strong",
                         "dist_backend": dist.get_backend(), "ranks_seen_by_rccl": dist.get_world_size() if dist.get_backend() == "nccl" else 0,
                         "ranks": dist.get_world_size(), "rounds_per_step": d["rounds"] / args.steps, "spec_breaks_per_step": d["spec_breaks"] / args.steps,
                         "rounds_that_exchanged_lists_per_step": coll["gathers"] / args.steps, "rounds_every_rank_evaluated_alone_per_step": coll["replicated"] / args.steps,
                         "allgather_us_per_round": None if not coll["gathers"] else round(coll["gather_s"] * 1e6 / coll["gathers"], 1),
                         "allreduce_us_per_round": None if not coll["reduces"] else round(coll["reduce_s"] * 1e6 / coll["reduces"], 1),
                         "collective_times_are": coll["clock"], "deferred_delta_checks_per_step": coll["checks"] / args.steps,
                         "verified": sharded_verified, "verified_with": "tests/golden/bench_rank_digests.json, rank 0's snapshot (the oracle's digest of decisions + bind set)"}
        if sharded_verified is False:
            value = None
        if sharded_watch is not None:
            sharded_watch.set()                  # the split's phase is through (its last collective was the verdict's reduction above)
    if dist_mode == "replicas" and world > 1:
        dec_last = runner.step(verify=False)                 # outside the timed region: one more cycle, its digest compared across ranks
        try:
            runner.check(dec_last)
            replicas_agree = True
        except RuntimeError as err:
            replicas_agree = False
            print(f"bench.py: {err}", file=sys.stderr)
    if dist_mode == "sessions":
        # outside the timed region: one more cycle per rank, its digest against the committed golden digest of THIS rank's snapshot;
        # the per-session rate of the slowest rank is the line's value, the sum over the ranks is the aggregate
        dec_last = runner.step(verify=False)
        mine = distmod.ReplicatedCycle.digest(dec_last, eng.binds(), eng.journal() if args.preempt else None, eng.evictions() if args.preempt else None)
        ok_here = 1 if (rank_digest_expected is not None and int(rank_digest_expected) == mine) else (0 if rank_digest_expected is not None else -1)
        dev = "cuda" if (world > 1 and dist.get_backend() == "nccl") else "cpu"
        stat = torch.tensor([evals / elapsed_local, -float(ok_here), float(evals), float(n_binds)], dtype=torch.float64, device=dev)
        if world > 1:
            mn = stat.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            sm = stat.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            mx = stat.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        else:
            mn = sm = mx = stat
        value = float(mn[0].item())                           # slowest rank's evals/s on its own snapshot
        worst = -int(mx[1].item())                            # min over ranks of ok_here
        sessions_verified = None if worst < 0 else bool(worst == 1)
        aggregate = {"aggregate_evals_per_s": float(sm[2].item()) / elapsed, "sessions_per_s": world * args.steps / elapsed,
                     "aggregate_binds_per_s": float(sm[3].item()) * args.steps / elapsed}
        if ok_here == 0:
            print(f"bench.py: rank {rank}: decisions digest {mine} differs from the golden digest {rank_digest_expected}", file=sys.stderr)
        if sessions_verified is False:
            value = None          # a session that was decided differently from the oracle has no rate

    # ---- roofline: the mask+score matrix (K1), HBM-bound by construction (SURVEY.md §8d accounting (M): 2 B score + 1/8 B
    # mask per evaluation written once, node and task vectors read once).  Two measurements, both with HIP events on the
    # engine's stream:
    #   "roofline"        the materialised T x N matrix of this workload (kb_bench_matrix, rows [0,T)): k_matrix over the
    #                     distinct task shapes + k_expand_tiles streaming every task row out — the size north_star quotes the
    #                     roofline target on; the time is for BOTH launches;
    #   "roofline_cycle"  k_matrix as the scheduling cycle launches it inside the timed region: one launch per round over the
    #                     DISTINCT task shapes of the window (a few dozen rows), i.e. latency-sized.
    R, N, T = snap.n_res, snap.n_nodes, snap.n_tasks
    b_node, b_task = 16 * R + 44, 8 * R + 24

    def roof(rows, ms, launches, label, kname="k_matrix", N=N, b_node=b_node, b_task=b_task):
        alg = rows * N * 2.125 + N * b_node + rows * b_task
        ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "kernel": kname, "launch": label, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": int(alg),
                "avg_launch_ms": round(ms, 5), "launches": int(launches), "rows_per_launch": round(rows, 1),
                "evals_per_s": round(rows * N / (ms * 1e-3), 1) if ms > 0 else 0.0}

    launches = max(1, d["matrix_launches"])
    roofline_cycle = roof(d["matrix_evals"] / N / launches, d["matrix_ms"] / launches, launches,
                          "per-round launch inside the timed region (distinct shapes of one window)")
    full_reps = 5
    full_ms = eng.bench_matrix(0, T, reps=full_reps) if world == 1 else 0.0
    if args.diverse:     # shapes > rows / 16: the engine evaluates every row directly instead of expanding shape rows (kb_engine.cpp)
        full_label, full_kernel = f"kb_bench_matrix rows [0,{T}) x {N} nodes: direct per-row evaluation, runs of adjacent equal rows evaluated once", "k_matrix_runs"
    else:
        full_label, full_kernel = f"kb_bench_matrix rows [0,{T}) x {N} nodes: per-shape evaluation + row expansion", "k_matrix+k_expand_tiles"
    roofline = roof(T, full_ms, full_reps, full_label, full_kernel) if full_ms > 0 else roofline_cycle
    # The launch above leans on the snapshot's shape redundancy (a few hundred distinct task shapes: evaluate once, copy out).  The same
    # T x N matrix through the matrix kernel's own evaluation, so that the driver's record shows the evaluator and not only the copy:
    #   roofline_eval           every row evaluated by k_matrix itself, rows equal to their predecessor (the tasks of a job) re-stored
    #   roofline_eval_all_rows  T x N evaluations, nothing shared (include/kb_engine.h: KB_MATRIX_DIRECT | KB_MATRIX_NO_DEDUP)
    roofline_eval = roofline_eval_all = None
    if world == 1 and full_ms > 0:
        abi = kbm.abi
        ms_d = eng.bench_matrix(0, T, reps=3, fit_mode=1 | abi.MATRIX_DIRECT)
        roofline_eval = roof(T, ms_d, 3, f"kb_bench_matrix rows [0,{T}) x {N} nodes: direct per-row evaluation, runs of adjacent equal rows evaluated once", "k_matrix_runs")
        ms_a = eng.bench_matrix(0, T, reps=2, fit_mode=1 | abi.MATRIX_DIRECT | abi.MATRIX_NO_DEDUP)
        roofline_eval_all = roof(T, ms_a, 2, f"kb_bench_matrix rows [0,{T}) x {N} nodes: {T} x {N} evaluations, nothing shared", "k_matrix<4,32>")
    # HBM bytes per launch from the PMC passes of scripts/gpu_r5.sh profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # runs; KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  Cannot be collected inside this
    # process, so the committed summary of the same command is read back; null when it is absent or for another config.
    import glob
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from kernel_sources_sha import kernel_sources_sha, kernel_tu_sha, read_tu_stamp
    live_sha = kernel_sources_sha(ROOT)

    def stale(csv_path, tu):
        """None when the committed summary was measured on kernels compiled from THESE sources, else the reason it may not be quoted.  The stamps
        are written on the GPU box beside the CSVs: kernel_tu.sha256 (one value per translation unit — `tu` is the .hip file that defines the
        kernels the CSV speaks of; scripts/kernel_sources_sha.py) or, for summaries older than that, kernel_sources.sha256 (the whole tree)"""
        here = os.path.dirname(csv_path)
        per_tu = os.path.join(here, "kernel_tu.sha256")
        if os.path.exists(per_tu):
            rec, live = read_tu_stamp(per_tu).get(tu), kernel_tu_sha(ROOT, tu)
            if rec is None:
                return f"{os.path.relpath(per_tu, ROOT)} has no entry for {tu}: {os.path.basename(csv_path)} cannot be tied to the kernels of this tree"
            return None if rec == live else f"{os.path.relpath(csv_path, ROOT)} was measured on another {tu} (sha256 of the translation unit {rec[:12]}, this tree {live[:12]})"
        stamp = os.path.join(here, "kernel_sources.sha256")
        if not os.path.exists(stamp):
            return f"{os.path.relpath(csv_path, ROOT)} carries no kernel_sources.sha256: it cannot be tied to the kernels of this tree"
        rec = open(stamp).read().split()[0]
        return None if rec == live_sha else f"{os.path.relpath(csv_path, ROOT)} was measured on other device sources (sha256 {rec[:12]}, this tree {live_sha[:12]})"
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "rocprofv3_pmc_k_matrix.csv")),
                   key=lambda f: int("".join(c for c in os.path.basename(os.path.dirname(f)) if c.isdigit()) or 0))
    pmc = found[-1] if found else ""                # the newest round's summary
    if full_ms > 0 and pmc and not args.diverse:     # the diverse-shape stress evaluates every row directly: another kernel mix
        # only a profile of THIS launch may speak for it: the bench line committed beside the CSV must carry the same algorithmic
        # bytes per launch (same configuration, scale and matrix layout); otherwise traffic stays null
        same = False
        for js in glob.glob(os.path.join(os.path.dirname(pmc), "*.json")):
            try:
                prof_line = json.loads(open(js).read().strip().splitlines()[-1])
                same = same or int(prof_line["roofline"]["bytes_per_launch"]) == int(roofline["bytes_per_launch"])
            except Exception:
                pass
        if same and stale(pmc, "kb_kernels.hip"):
            roofline["traffic_refused"] = stale(pmc, "kb_kernels.hip")
        elif same:
            import csv
            vals = {}
            for row in csv.DictReader(open(pmc)):
                # the materialised-matrix launch: k_expand_tiles + the matrix kernel over the distinct shapes (any tile but the per-round <1, 4>)
                if "k_expand" in row["kernel"] or ("k_matrix<" in row["kernel"] and "k_matrix<1, 4>" not in row["kernel"]):
                    vals[row["counter"]] = vals.get(row["counter"], 0.0) + float(row["mean_KB_per_dispatch"])
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                roofline["traffic"] = int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
                roofline["traffic_source"] = "committed rocprofv3 PMC passes of the same launch (not re-measured in this run): " + os.path.relpath(pmc, ROOT)

    # ---- the commit kernels: where the timed region actually goes (one workgroup on one of the 256 CUs; the HBM roofline above does not
    # bound it).  Live: the kernels' own wall-clock stamps over the timed region -> ns and shader cycles per committed row.  From the committed
    # rocprofv3 PMC passes of the same command (profiles/round*/rocprofv3_pmc_k_commit.csv, scripts/gpu_r5.sh profile; not re-measured here):
    # per kernel, the instruction mix per dispatch, the share of wave cycles spent issuing / parked, instructions per busy cycle of the CU.
    commit_ms_step = d["commit_ms"] / args.steps
    rows_step = max(1.0, d["decisions"] / args.steps)
    roofline_commit = {"bound": "issue of one CU (latency-bound serial dependency; not an HBM or MFMA roofline)",
                       "commit_ms_per_step": round(commit_ms_step, 3), "share_of_step": round(commit_ms_step / (elapsed * 1e3 / args.steps), 4),
                       "ns_per_committed_row": round(commit_ms_step * 1e6 / rows_step, 1),
                       "cycles_per_committed_row_at_2.4GHz": round(commit_ms_step * 1e6 / rows_step * 2.4, 0),
                       "rounds_per_step": d["rounds"] / args.steps, "us_per_round": round(commit_ms_step * 1e3 / max(1.0, d["rounds"] / args.steps), 1),
                       "streaming_equivalent_frac_of_hbm_peak": round(value * b_node / 1e9 / HBM_PEAK_GBS, 4),   # SURVEY 8d accounting (S), see streaming_equivalent
                       "counters": None}
    found_c = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "rocprofv3_pmc_k_commit.csv")),
                     key=lambda f: int("".join(c for c in os.path.basename(os.path.dirname(f)) if c.isdigit()) or 0))
    if found_c and stale(found_c[-1], "kb_commit_sel.hip"):
        roofline_commit["counters_refused"] = stale(found_c[-1], "kb_commit_sel.hip")
    elif found_c:
        import csv
        per = {}
        for row in csv.DictReader(open(found_c[-1])):
            per.setdefault(row["kernel"], {})[row["counter"]] = float(row["mean_per_dispatch"])
        cnt = {}
        for kname, c in per.items():
            insts = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
            wc = c.get("SQ_WAVE_CYCLES", 0.0)
            cnt[kname] = {"wave_insts_per_dispatch": {k[3:].lower(): int(c[k]) for k in c if k.startswith("SQ_INSTS_")},
                          "wave_cycle_shares": None if not wc else {"issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3), "parked_waitcnt_or_sleep": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                                                                    "issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3), "of_it_lds": round(c.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4)},
                          "wave_insts_per_cu_busy_cycle": None if not c.get("SQ_BUSY_CYCLES") else round(insts / c["SQ_BUSY_CYCLES"], 3),
                          "lds_bank_conflict_share": None if not c.get("SQ_LDS_IDX_ACTIVE") else round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3),
                          "waves_per_dispatch": int(c.get("SQ_WAVES", 0))}
        roofline_commit["counters"] = cnt
        shape_note = os.path.join(os.path.dirname(found_c[-1]), "rocprofv3_pmc_k_commit.launch_shape.txt")
        roofline_commit["counters_launch_shape"] = (open(shape_note).read().strip() if os.path.exists(shape_note)
                                                    else "the default launch: commit workgroup + L2 helpers + one repair workgroup per shape (sums over ~30 mostly idle workgroups)")
        roofline_commit["counters_source"] = "committed rocprofv3 PMC passes of the default command (not re-measured in this run): " + os.path.relpath(found_c[-1], ROOT)

    out = {
        "metric": f"pod-node scoring evals/sec + binds/sec, {_k(snap.n_tasks)} tasks x {_k(snap.n_nodes)} nodes snapshot",
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong" if dist_mode == "sharded" else "weak",   # the task-row split divides ONE session (total work fixed); sessions mode: one session per GPU
        "vs_baseline": None, "dtype": "f64+i64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{args.config - 1}]: {snap.n_tasks} tasks x {snap.n_nodes} nodes, "
                               f"{snap.n_jobs} gang jobs, {snap.n_queues} queues, R={R}, {'+'.join(actions)}, "
                               f"plugins priority,gang,drf,predicates,proportion,nodeorder ({weights})",
                   "window": args.window or 256, "scale": args.scale, "diverse_requests": bool(args.diverse), "host_wait": "spin" if args.spin_wait else "short spin, then sched_yield between polls",
                   "node_sizes": "SURVEY 8d list (no capacity pressure)" if args.survey_nodes else "sized for demand ~1.3x capacity"},
        "binds_per_s": n_binds * args.steps / elapsed, "binds": n_binds, "decisions": int(d["decisions"] / args.steps),
        "evals_per_step": int(evals / args.steps),
        # `value` counts the evaluations the REFERENCE performs for this cycle (N per popped task).  The engine
        # itself evaluates far fewer pairs (one matrix row per distinct task shape of a window + the dirty-node repairs):
        "value_counts": "reference-equivalent (task,node) evaluations",
        "matrix_evals_per_step": int(d["matrix_evals"] / args.steps),
        "kernel_ms_per_step": {k: round(d[k] / args.steps, 3) for k in ("matrix_ms", "argmax_ms", "commit_ms", "reduce_ms", "host_order_ms", "total_ms")},
        # (round 5: a chained round's repair workgroups ride in its commit launch — commit_ms contains the commit workgroup's wait for them,
        #  argmax_ms only what a round waited for its lists OUTSIDE that launch: first rounds, re-planned rounds, KB_FUSE_REPAIR=0; matrix_ms of
        #  chained rounds ran on the second stream beside the predecessor's commit and is not on the cycle's timeline)
        "kernel_ms_note": "commit_ms includes the wait for the repair workgroups of the same launch; matrix_ms of chained rounds is off the timeline",
        "rounds_per_step": d["rounds"] / args.steps, "spec_breaks_per_step": d["spec_breaks"] / args.steps,
        "row_fallbacks_per_step": d["row_fallbacks"] / args.steps,
        "roofline": roofline, "roofline_cycle": roofline_cycle, "roofline_eval": roofline_eval, "roofline_eval_all_rows": roofline_eval_all,
        "roofline_commit": roofline_commit,
        # SURVEY.md §8d accounting (S): the reference's own dataflow streams B_node(R) bytes per evaluation (every popped task
        # against every node's live state).  The engine never moves those bytes (shape dedup + dirty-node repair); this is the
        # end-to-end rate expressed in that currency, for comparison with the 8 TB/s a streaming pass would be bound by.
        "streaming_equivalent": {"bytes_per_eval": b_node, "equivalent_GBps": round(value * b_node / 1e9, 1),
                                 "frac_of_hbm_peak": round(value * b_node / 1e9 / HBM_PEAK_GBS, 4)},
        # not part of `value`: kb_session_load of the same snapshot (validation, shape interning, proportion water-fill, H2D)
        "multi_gpu_mode": None if dist_mode is None else {"sessions": "one independent session per GPU (rank k: seed + k), no data-path collective; value = the slowest rank's per-session rate",
                                                          "replicas": "the same session on every GPU (KB_DIST_MODE=replicas), digests compared; value = one session's rate",
                                                          "sharded": "north_star's task-row split of ONE session (value, ms_per_step, scaling are its figures)"
                                                                     + ("; the sessions mode of the same invocation under `sessions`" if sessions_block else "")}[dist_mode],
        "sharded": sharded_block if sharded_block is not None else ({"error": sharded_setup_error, "verified": None} if (world > 1 or force_sharded) and sharded_setup_error else None),
        "sessions": sessions_block,
        "dist_backend": None if dist_mode is None else dist.get_backend(),   # "nccl" = RCCL: what carried the N ranks (scripts/scale_curve.sh asserts it)
        "replicas_agree": replicas_agree, "sessions_verified_against_golden_digests": sessions_verified,
        **(aggregate or {}),
        "session_load_ms": None if load_ms is None else round(load_ms, 2),
        "session_load_ms_max": None if load_ms is None else round(max(load_samples), 2), "session_load_ms_samples": [round(x, 2) for x in load_samples],
        "evals_per_s_including_session_load": None if load_ms is None else evals / args.steps / (elapsed / args.steps + load_ms * 1e-3),
        "value_note": None if dist_mode != "sessions" else "N > 1, sessions mode: `value` is the slowest rank's per-session rate (round 3 printed the sum over the ranks: now `aggregate_evals_per_s`)",
    }

    if args.config == 5 and not args.cpu_sample_tasks:
        args.cpu_sample_tasks = 2000           # 1M x 50k: the faithful loop needs minutes; a bounded sample, as section 4 of the task allows
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU baseline is reported at N=1 only
        import oracle
        oracle.build()
        threads = min(16, os.cpu_count() or 1)    # util/scheduler_helper.go:84: 16 workers
        o = oracle.Oracle(conf, snap, threads=threads)
        if args.cpu_sample_tasks:
            o.set_task_limit(args.cpu_sample_tasks)
        c0 = time.perf_counter()
        o.allocate()
        c1 = time.perf_counter()
        out["cpu_baseline"] = {"value": o.evals / (c1 - c0), "unit": "evals/s", "cores": os.cpu_count() or 1, "threads": threads,
                               "kind": "port",
                               "sample": f"{'first ' if args.cpu_sample_tasks else 'all '}{o.popped} popped tasks of the same snapshot's allocate action "
                                         f"({o.evals} evals, {c1 - c0:.1f} s) on {threads} threads of the box's {os.cpu_count()} host cores; C restatement "
                                         "of the Go loop with the reference's 16-worker per-task fan-out, without its per-pair NodeInfo rebuilds"}
        if not args.cpu_sample_tasks:
            # the oracle has just run the whole allocate action for the baseline: let it finish the cycle (backfill, untimed) and
            # check the engine's bind set and evaluation count against it, so every headline line carries its own verification
            o.backfill()
            if args.preempt:
                o.preempt()
                out["verified_evictions_equal_oracle"] = bool([int(t) for t in eng.evictions()] == [int(t) for t in o.evictions()])
                ej, oj = eng.journal(), o.journal()
                out["verified_journal_equals_oracle"] = bool(ej.shape == oj.shape and np.array_equal(ej, oj))
                out["journal_entries"] = int(len(ej))
            out["verified_bind_set_equals_oracle"] = bool(np.array_equal(binds, o.binds()))
            out["verified_evals_equal_oracle"] = bool(o.evals == out["evals_per_step"])
        o.close()
        if not args.cpu_sample_tasks:
            # beside the reference-shaped loop on 16 threads: the same action by a smart CPU algorithm (the oracle's incremental mode: one
            # cached row per task shape, one-node repairs, a max-tree) on ONE thread — the engine's margin over THAT is the honest one
            o = oracle.Oracle(conf, snap, threads=1)
            o.set_fast(True)
            c0 = time.perf_counter()
            o.allocate()
            c1 = time.perf_counter()
            out["cpu_baseline_incremental"] = {"value": o.evals / (c1 - c0), "unit": "evals/s", "cores": os.cpu_count() or 1, "threads": 1, "kind": "port",
                                               "sample": f"the whole allocate action ({o.evals} reference-equivalent evals, {c1 - c0:.1f} s) by the oracle's "
                                                         "incremental mode (per-shape cached rows + one-node repairs + max-tree) on one thread"}
            o.close()
    default_run = (rank == 0 and world == 1 and args.config == 3 and not (args.diverse or args.survey_nodes or args.preempt) and not args.no_cpu_baseline
                   and (args.scale == 1.0 or os.environ.get("KB_BENCH_VARIANTS") == "1") and os.environ.get("KB_BENCH_VARIANTS") != "0")
    if default_run:
        # the unfriendly inputs, in the driver's own record: SURVEY 8d's literal node sizes (no capacity pressure: dirty nodes win most rows)
        # and BASELINE configs[3] (R = 16, bin-packing weights), each timed over 3 cycles and verified against the oracle's incremental mode
        import oracle

        def variant_roofline(ve, vsnap, direct=False):
            """the variant's OWN materialised T x N matrix through kb_bench_matrix (the launch `roofline` above times for the headline workload): every
            configuration's roofline fraction in the driver's record — R = 16's sits below R = 2's, and was invisible in round 5's line"""
            vR, vN, vT = vsnap.n_res, vsnap.n_nodes, vsnap.n_tasks
            vms = ve.bench_matrix(0, vT, reps=3)
            kn = "k_matrix_runs" if direct else "k_matrix+k_expand_tiles"
            rf = roof(vT, vms, 3, f"kb_bench_matrix rows [0,{vT}) x {vN} nodes, R={vR}", kn, N=vN, b_node=16 * vR + 44, b_task=8 * vR + 24)
            return {k: rf[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "avg_launch_ms")}

        variants = {}
        for name, idx, vconf, tweak in (("survey_nodes", 3, kbm.conf.load_scheduler_conf(), True), ("config4_binpack", 4, kbm.conf.load_scheduler_conf(BINPACK_CONF), False),
                                        ("config2", 2, kbm.conf.load_scheduler_conf(), False)):
            vp = kbm.snapshot.synth_config(idx, args.scale)
            if tweak:
                vp.node_cpu_cores = (16, 32, 64, 96, 128)
                vp.node_mem_gib = (64, 128, 256, 512)
            vsnap = kbm.snapshot.synth(vp)
            ve = engine.Engine(vconf, device=local_rank)
            ve.load(vsnap)
            ve.run(["allocate", "backfill"])
            torch.cuda.synchronize()
            v0 = time.perf_counter()
            for _ in range(3):
                ve.reset()
                vdec = ve.run(["allocate", "backfill"])
            torch.cuda.synchronize()
            vms = (time.perf_counter() - v0) * 1e3 / 3
            vo = oracle.Oracle(vconf, vsnap, threads=1)
            vo.set_fast(True)
            vo.run(["allocate", "backfill"])
            variants[name] = {"ms_per_step": round(vms, 2), "binds": int((ve.binds() != kbm.abi.KB_NONE).sum()),
                              "verified": bool(np.array_equal(vdec, vo.decisions()) and np.array_equal(ve.binds(), vo.binds())),
                              "evals_per_s": round(vo.evals / (vms * 1e-3), 1), "roofline": variant_roofline(ve, vsnap),
                              "workload": f"{vsnap.n_tasks} tasks x {vsnap.n_nodes} nodes, R={vsnap.n_res}"}
            ve.close()
            vo.close()
        # BASELINE configs[4] (1M tasks x 50k nodes), as allocate + backfill and with its third action: held to the digests committed under
        # tests/golden/fullsize_digests.json (the oracle's, tests/test_gpu_fullsize.py checks the live oracle against them) — no oracle minute here
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_fullsize_golden as mfg
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))
        # ... and configs[2] with every job drawing its OWN request (--diverse: ~9 600 distinct task shapes, the input that leans least on shape redundancy),
        # held to its committed digest likewise (the oracle's incremental mode needs minutes for it: make_fullsize_golden.py)
        for name, case in (("diverse", "config3_diverse_full"), ("config5", "config5_full"), ("config5_preempt", "config5_full_preempt")) if args.scale == 1.0 else ():   # the digests are the full size's
            vconf, vsnap = mfg.case_inputs(kbm, case)
            vact = mfg.case_actions(case)
            ve = engine.Engine(vconf, device=local_rank)
            ve.load(vsnap)
            ve.run(vact)
            vloads = []
            for _ in range(3):
                l0 = time.perf_counter()
                ve.load(vsnap)
                vloads.append((time.perf_counter() - l0) * 1e3)
            ve.run(vact)
            torch.cuda.synchronize()
            v0 = time.perf_counter()
            for _ in range(2):
                ve.reset()
                vdec = ve.run(vact)
            torch.cuda.synchronize()
            vms = (time.perf_counter() - v0) * 1e3 / 2
            evict = case.endswith("_preempt")
            digest = mfg.digest_of(np, vdec, ve.binds(), ve.last_journal if evict else None, ve.evictions() if evict else None)
            variants[name] = {"ms_per_step": round(vms, 2), "binds": int((ve.binds() != kbm.abi.KB_NONE).sum()),
                              "verified": bool(digest == golden[case]["sha256"]), "verified_with": "tests/golden/fullsize_digests.json (the oracle's digest of decisions, binds"
                              + (", Statement journal, evictions)" if evict else ")"), "session_load_ms": round(sorted(vloads)[1], 2),
                              "evals_per_s": None if evict else round(golden[case]["evals"] / (vms * 1e-3), 1),
                              # (1M x 50k: its materialised matrix is 109 GB — not allocated for a side figure; the 100k x 10k variants carry theirs)
                              "roofline": variant_roofline(ve, vsnap, direct=True) if name == "diverse" else None,
                              "workload": f"{vsnap.n_tasks} tasks x {vsnap.n_nodes} nodes, R={vsnap.n_res}, {'+'.join(vact)}" + (", one request per job" if name == "diverse" else "")}
            ve.close()
        out["variants"] = variants
    if rank == 0 and args.verify and "verified_bind_set_equals_oracle" not in out:
        import oracle
        o = oracle.Oracle(conf, snap, threads=min(16, os.cpu_count() or 1))
        if snap.n_tasks * snap.n_nodes > 4_000_000_000:   # config 5: the oracle's incremental mode (tests/test_oracle_fast_cpu.py)
            o.set_fast(True)
            out["verified_with"] = "oracle fast mode (allocate: cached rows + one-node repairs; preempt: per-queue node sets + cached SortNodes lists), held to the faithful mode in tests/test_oracle_fast_cpu.py"
        o.run(actions)
        if args.preempt:
            out["verified_evictions_equal_oracle"] = bool([int(t) for t in eng.evictions()] == [int(t) for t in o.evictions()])
            ej, oj = eng.journal(), o.journal()       # every Statement operation of the last step, with statement numbers and commit / discard markers
            out["verified_journal_equals_oracle"] = bool(ej.shape == oj.shape and np.array_equal(ej, oj))
            out["journal_entries"] = int(len(ej))
        out["verified_bind_set_equals_oracle"] = bool(np.array_equal(binds, o.binds()))
        out["verified_evals_equal_oracle"] = bool(o.evals == out["evals_per_step"])
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force_sharded:
        dist.destroy_process_group()
    if sessions_verified is False or replicas_agree is False or (sharded_block and sharded_block["verified"] is False) or (sessions_block and sessions_block["verified"] is False):
        sys.exit(1)


if __name__ == "__main__":
    main()
