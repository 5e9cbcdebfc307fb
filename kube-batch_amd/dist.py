"""Task-row sharding of the scheduling cycle across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank holds a full replica
of the session.  Per round:

    round_begin        identical speculated window on every rank (deterministic host code)
    round_candidates   mask+score matrix and sorted candidate lists for THIS RANK'S SHARD of the window's matrix rows
    all_gather         candidate lists -> full table on every rank              (the path's real exchange step)
    round_commit       identical sequential commit on every rank; per-node committed deltas of the rows this rank owns
    all_reduce(sum)    per-node deltas (integer-valued float64: exact, order-independent)
    round_apply        next round's node state := round start + reduced deltas; must equal the replica's own commit

Every replica commits the whole window itself, so the reduced deltas are a CROSS-CHECK, not data the next round waits for.  The default
(`defer_check=True`) therefore takes that all-reduce off the critical path: round k's buffer (two alternate) is reduced asynchronously — on a
side stream with RCCL — while round k + 1 is planned, evaluated and committed, and compared one round late on the device (kb_round_check, queued
behind round k + 1's commit: start of round k + deltas == start of round k + 1; a counter read once per action).  `defer_check=False` is the lock-step form above.

preempt / reclaim (BASELINE configs[4] names allocate + backfill + preempt) in this mode: the evict actions are host machines around a few
device lists (kb_preempt.cpp; 49 ms of host time at 1M x 50k, nothing in them shards), and every replica needs their result — the
Statement journal applied to its node and task state — before the next action.  So every rank runs the action on its own replica
(deterministic, like the host side of a round) and ONE all-reduce per action compares a digest of journal and evictions: a replica that
diverged is reported on every rank.  (Rank 0 running it alone and broadcasting the journal would leave the other replicas the same
host work — replaying the journal through the same machine — plus the broadcast.)

The transport is pluggable: with the "nccl" backend the collectives run on device buffers; with "gloo" (CPU tests, or
several ranks sharing one GPU) the same buffers are staged through host memory.

`RoundBackend` is the seam the tests use: the product backend is the engine (C ABI), the CPU tests substitute a small
deterministic stand-in to exercise the sharding arithmetic and the collectives without a GPU.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n items; every rank computes the same partition."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class EngineBackend:
    """The product backend: kb_round_* of the C ABI on this rank's GPU."""

    def __init__(self, engine, device: torch.device):
        self.e = engine
        self.device = device
        self.delta_len = engine.round_delta_doubles()

    def begin(self, action):
        return self.e.round_begin(action)

    def candidates(self, m0, m1, keys: torch.Tensor):
        self.e.round_candidates(m0, m1, keys.data_ptr())

    def commit(self, all_keys: torch.Tensor, r0, r1, delta: torch.Tensor):
        self.e.round_commit(all_keys.data_ptr(), r0, r1, delta.data_ptr())

    def apply(self, delta: Optional[torch.Tensor]):
        self.e.round_apply(delta.data_ptr() if delta is not None else 0)

    def check(self, delta: torch.Tensor, against_live: bool):
        self.e.round_check(delta.data_ptr(), against_live)

    def check_result(self) -> int:
        return self.e.round_check_result()

    def decisions(self):
        return self.e.round_decisions()


class ShardedCycle:
    """Runs allocate (+ backfill) with the window's matrix rows sharded across ranks."""

    def __init__(self, conf, snap, device: int = 0, window: int = 0, commit_batch: int = 0, backend=None,
                 buffer_device: Optional[torch.device] = None, actions=("allocate", "backfill"), min_rows_per_rank: int = 32, defer_check: bool = True):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.actions = list(actions)
        if backend is None:
            from .engine import Engine
            self.engine = Engine(conf, device=device, window=window, commit_batch=commit_batch)
            self.engine.load(snap)
            dev = torch.device("cuda", device)
            backend = EngineBackend(self.engine, dev)
            buffer_device = dev
        else:
            self.engine = getattr(backend, "e", None)
        self.backend = backend
        self.buf_dev = buffer_device or torch.device("cpu")
        # collectives on device buffers only with a device-capable backend (RCCL); otherwise stage through host memory
        self.stage_host = (not dist.is_initialized()) or dist.get_backend() != "nccl"
        # RCCL: the engine launches on torch's current stream, the stream the collectives are ordered on, so a round needs no host
        # synchronisation between its kernels and its collectives (the commit result still arrives through the pinned mailbox)
        self.stream_ordered = (not self.stage_host) and self.engine is not None and buffer_device is not None and buffer_device.type == "cuda"
        self._stream = None
        if self.stream_ordered:
            # a DEDICATED stream: torch's default stream has handle 0, which kb_engine_use_stream reads as "back to the engine's own
            # non-blocking stream" — the engine would then NOT be ordered with the collectives while every host synchronisation below
            # is skipped (round-2 review).  Collectives and torch's fills are issued under this stream (see _on_stream); if its
            # handle were ever 0 the ordering claim is dropped and the synchronising path is used instead.
            self._stream = torch.cuda.Stream(device=buffer_device)
            if self._stream.cuda_stream == 0:
                self.stream_ordered, self._stream = False, None
            else:
                self.engine.use_stream(self._stream.cuda_stream)
        with self._on_stream():
            self.delta = torch.zeros(backend.delta_len, dtype=torch.float64, device=self.buf_dev)
            self.delta2 = torch.zeros(backend.delta_len, dtype=torch.float64, device=self.buf_dev)
        # the reduced deltas as a deferred cross-check (module docstring); backends without the check entry points (the CPU stand-ins) run lock-step
        self.defer_check = bool(defer_check) and hasattr(backend, "check")
        self._side = torch.cuda.Stream(device=buffer_device) if (self.stream_ordered and self.defer_check) else None
        self.deferred_checks = 0
        self.rounds = 0
        self.replicated_rounds = 0
        self.evict_actions = 0
        # what the two collectives of a round cost (bench.py's `sharded` object): device time between two events on the stream the collective runs on
        # when it is stream-ordered (RCCL), else the host's wall clock around the call and its completion (gloo: staging through host memory included)
        self._gather_s = self._reduce_s = 0.0
        self._gathers = self._reduces = 0
        self._events = []          # (kind, start event, end event): summed by collective_times()
        # a single-rank group still goes through the collectives when asked to (exercises the RCCL path on a one-GPU box)
        import os as _os
        self.always_collect = dist.is_initialized() and _os.environ.get("KB_DIST_ALWAYS_COLLECT") == "1"
        # shard a round's matrix rows only when every rank gets at least this many (0 = always shard)
        self.min_rows_per_rank = min_rows_per_rank

    def _on_stream(self):
        """Context in which torch work (fills, collectives) is enqueued on the stream the engine's kernels run on."""
        import contextlib
        return torch.cuda.stream(self._stream) if self._stream is not None else contextlib.nullcontext()

    def _sync_torch(self):
        """The engine runs on its own non-blocking HIP stream: torch's fill kernels must have finished before the engine
        writes into a freshly zeroed buffer."""
        if self.buf_dev.type == "cuda" and not self.stream_ordered:
            torch.cuda.current_stream().synchronize()

    # ---- collectives
    def collective_times(self) -> dict:
        """seconds spent in the rounds' all-gathers / all-reduces since construction and how many there were (see __init__ for the clock)"""
        if self._events:
            torch.cuda.synchronize()
            for kind, a, b in self._events:
                dt = a.elapsed_time(b) * 1e-3
                if kind == 0:
                    self._gather_s += dt
                else:
                    self._reduce_s += dt
            self._events = []
        return {"gather_s": self._gather_s, "gathers": self._gathers, "reduce_s": self._reduce_s, "reduces": self._reduces,
                "clock": "device time between events on the collective's stream (RCCL)" if self.stream_ordered else "host wall clock around the call and its completion (staged through host memory)"}

    def _timed(self, kind: int, stream=None):
        """context manager: one collective of kind 0 (all-gather) / 1 (all-reduce)"""
        import contextlib
        import time as _time

        @contextlib.contextmanager
        def cm():
            if self.stream_ordered:
                st = stream if stream is not None else self._stream
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                yield
                b.record(st)
                self._events.append((kind, a, b))
            else:
                t0 = _time.perf_counter()
                yield
                if kind == 0:
                    self._gather_s += _time.perf_counter() - t0
                else:
                    self._reduce_s += _time.perf_counter() - t0
        return cm()

    def _all_gather_keys(self, local: torch.Tensor, chunk: int, L: int) -> torch.Tensor:
        if self.world == 1 and not self.always_collect:
            full = torch.empty((chunk * self.world, L), dtype=torch.int64, device=local.device)
            full.copy_(local)
            return full
        self._gathers += 1
        with self._timed(0):
            return self._all_gather_keys_impl(local, chunk, L)

    def _all_gather_keys_impl(self, local: torch.Tensor, chunk: int, L: int) -> torch.Tensor:
        full = torch.empty((chunk * self.world, L), dtype=torch.int64, device=local.device)
        if self.stage_host and local.device.type != "cpu":
            h_local = local.cpu()
            h_full = torch.empty((chunk * self.world, L), dtype=torch.int64)
            dist.all_gather_into_tensor(h_full, h_local)
            full.copy_(h_full)
        else:
            dist.all_gather_into_tensor(full, local)
            if full.device.type == "cuda" and not self.stream_ordered:
                torch.cuda.current_stream().synchronize()   # the engine's kernels run on their own stream
        return full

    def _all_reduce_delta(self):
        if self.world == 1 and not self.always_collect:
            return
        self._reduces += 1
        with self._timed(1):
            self._all_reduce_delta_impl()

    def _all_reduce_delta_impl(self):
        if self.stage_host and self.delta.device.type != "cpu":
            h = self.delta.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            self.delta.copy_(h)
        else:
            dist.all_reduce(self.delta, op=dist.ReduceOp.SUM)
            if self.delta.device.type == "cuda" and not self.stream_ordered:
                torch.cuda.current_stream().synchronize()

    def _all_reduce_delta_async(self, buf: torch.Tensor):
        """-> a callable that completes the reduction of `buf` (the commit that filled it has been collected: its kernel is over).  RCCL: issued on
        a side stream, the returned wait makes the engine's stream wait for it (no host wait); gloo: staged through host memory, asynchronous
        on the process group's own thread."""
        if self.world == 1 and not self.always_collect:
            return lambda: None
        self._reduces += 1
        import time as _time
        if self._side is not None:
            self._side.wait_stream(self._stream)
            with torch.cuda.stream(self._side):
                with self._timed(1, self._side):
                    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)

            def done():
                work.wait()                              # orders the current (the engine's) stream behind the collective
                self._stream.wait_stream(self._side)
            return done
        t0 = _time.perf_counter()
        if self.stage_host and buf.device.type != "cpu":
            h = buf.cpu()
            work = dist.all_reduce(h, op=dist.ReduceOp.SUM, async_op=True)
            self._reduce_s += _time.perf_counter() - t0

            def done():
                t1 = _time.perf_counter()
                work.wait()
                buf.copy_(h)
                if buf.device.type == "cuda":
                    torch.cuda.current_stream().synchronize()
                self._reduce_s += _time.perf_counter() - t1
            return done
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        self._reduce_s += _time.perf_counter() - t0

        def done():
            t1 = _time.perf_counter()
            work.wait()
            if buf.device.type == "cuda" and not self.stream_ordered:
                torch.cuda.current_stream().synchronize()
            self._reduce_s += _time.perf_counter() - t1
        return done

    # ---- one action
    def run_action(self, action: int) -> np.ndarray:
        with self._on_stream():
            return self._run_action(action)

    def _run_action(self, action: int) -> np.ndarray:
        b = self.backend
        pending = None      # (completion of the all-reduce of the round before, its buffer)
        k = 0
        while True:
            n_rows, n_mrows, L = b.begin(action)
            if n_rows == 0:
                if pending is not None:     # the last round's reduced deltas: its start + deltas == the live state (the begin that ended the action took no copy)
                    pending[0]()
                    b.check(pending[1], True)
                    self.deferred_checks += 1
                    pending = None
                break
            # equal-sized shards of the matrix rows (padded so all_gather_into_tensor applies); the sorted candidate
            # lists are 0-terminated, padding rows stay 0
            if n_mrows < self.min_rows_per_rank * self.world:
                # too few distinct shapes in this window for the exchange to pay (a matrix row costs ~1 us of kernel time, a
                # collective + its synchronisation tens of us): every rank evaluates all rows, no all-gather this round.
                # The decision depends on n_mrows only, which is identical on every rank.
                table = torch.zeros((n_mrows, L), dtype=torch.int64, device=self.buf_dev)
                self._sync_torch()
                b.candidates(0, n_mrows, table)
                self.replicated_rounds += 1
            else:
                chunk = (n_mrows + self.world - 1) // self.world
                m0 = min(self.rank * chunk, n_mrows)
                m1 = min(m0 + chunk, n_mrows)
                local = torch.zeros((chunk, L), dtype=torch.int64, device=self.buf_dev)
                self._sync_torch()
                b.candidates(m0, m1, local)
                table = self._all_gather_keys(local, chunk, L)
            # the gathered table is [world*chunk][L]; matrix row m lives at row m because shards are contiguous and equal
            r0, r1 = shard_bounds(n_rows, self.world, self.rank)
            if self.defer_check:
                buf = self.delta if (k & 1) == 0 else self.delta2
                b.commit(table, r0, r1, buf)
                if pending is not None:
                    # round k - 1's reduction had THIS round's plan, candidate lists and commit to hide behind (round 5's advisor: completed right
                    # behind begin() it only overlapped the host's plan): start of k - 1 + deltas == start of k, queued before the next begin
                    # rotates the two start copies; the other delta buffer is not written again before round k + 1's commit
                    pending[0]()
                    b.check(pending[1], False)
                    self.deferred_checks += 1
                pending = (self._all_reduce_delta_async(buf), buf)
                b.apply(None)                   # absorb the round's result; the cross-check follows one round late
            else:
                b.commit(table, r0, r1, self.delta)
                self._all_reduce_delta()
                b.apply(self.delta)
            self.rounds += 1
            k += 1
        if self.defer_check and k:
            bad = b.check_result()
            if bad:
                raise RuntimeError(f"replicas diverged: the reduced per-node deltas differ from the local commits at {bad} values (rank {self.rank})")
        return b.decisions()

    def run_evict(self, name: str) -> np.ndarray:
        """preempt / reclaim on every replica (module docstring), journals cross-checked with one all-reduce; -> the journal"""
        if self.engine is None:
            raise RuntimeError("the evict actions need the engine backend")
        with self._on_stream():
            getattr(self.engine, "run_" + name)()
        journal = self.engine.last_journal
        if self.world > 1 or self.always_collect:
            d = ReplicatedCycle.digest(journal, self.engine.evictions())
            dev = self.buf_dev if not self.stage_host else torch.device("cpu")
            with self._on_stream():
                t = torch.tensor([d, -d], dtype=torch.int64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                agree = int(t[0].item()) == -int(t[1].item())
            if not agree:
                raise RuntimeError(f"replicas diverged in {name}: rank {self.rank} journal digest {d:#x}")
        self.evict_actions += 1
        return journal

    def step(self):
        """One scheduling cycle from the pristine session state (bench step)."""
        if self.engine is not None:
            self.engine.reset()
        out = []
        for a in self.actions:
            if a in ("preempt", "reclaim"):
                self.run_evict(a)      # no ssn.Allocate / ssn.Pipeline decisions: the journal carries the Statement operations (engine.journal())
            else:
                out.append(self.run_action({"allocate": 0, "backfill": 1}[a]))
        return np.concatenate(out) if out else np.zeros((0, 3), np.uint32)



class ReplicatedCycle:
    """The default multi-GPU mode of bench.py: every rank runs the WHOLE cycle on its own replica through the
    engine's single-GPU fast path (chained rounds, pinned mailbox, plan-ahead) — the host code and the kernels are deterministic, so
    the replicas agree without talking — and the ranks compare a digest of what they decided with ONE all-reduce per cycle.

    Why not shard: the part of a round that shards (matrix + candidate lists of ~25 shapes: ~19 us on C3, ~33 us on 1M x 50k) is
    shorter than one small collective over xGMI, and the commit (80 % of the cycle) is a sequential dependency; ShardedCycle above is
    the exact task-row split north_star names and costs ~2x a single GPU for that reason.  Replicas cost nothing and buy a
    cross-check: a replica that diverged (bit flip, driver fault) is detected at the end of the cycle."""

    def __init__(self, conf, snap, device: int = 0, window: int = 0, commit_batch: int = 0, engine=None, actions=("allocate", "backfill")):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.actions = list(actions)
        if engine is None:
            from .engine import Engine
            engine = Engine(conf, device=device, window=window, commit_batch=commit_batch)
            engine.load(snap)
        self.engine = engine
        self.cycles = 0
        on_gpu = dist.is_initialized() and dist.get_backend() == "nccl"
        self._dev = torch.device("cuda", device) if on_gpu else torch.device("cpu")

    @staticmethod
    def digest(decisions: np.ndarray, binds: np.ndarray, journal: Optional[np.ndarray] = None, evictions: Optional[np.ndarray] = None) -> int:
        """63 bits of SHA-256 over the ordered decision list and the bind set (and, for cycles with an evict action, the Statement journal and
        the evictions in cache.Evict order)"""
        import hashlib
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(decisions, dtype=np.uint32).tobytes())
        h.update(np.ascontiguousarray(binds, dtype=np.uint32).tobytes())
        if journal is not None:
            h.update(np.ascontiguousarray(journal, dtype=np.uint32).tobytes())
            h.update(np.ascontiguousarray(evictions if evictions is not None else np.zeros(0, np.uint32), dtype=np.uint32).tobytes())
        return int.from_bytes(h.digest()[:8], "little") >> 1

    def step(self, verify: bool = True) -> np.ndarray:
        self.engine.reset()
        dec = self.engine.run(self.actions)
        self.cycles += 1
        if verify and self.world > 1:
            self.check(dec)
        return dec

    def check(self, dec: np.ndarray):
        """one all-reduce (MIN and MAX of the digest in one tensor): every replica took the same decisions"""
        evict = any(a in ("preempt", "reclaim") for a in self.actions)
        d = self.digest(dec, self.engine.binds(), self.engine.journal() if evict else None, self.engine.evictions() if evict else None)
        t = torch.tensor([d, -d], dtype=torch.int64, device=self._dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if int(t[0].item()) != -int(t[1].item()):
            raise RuntimeError(f"replicas diverged: rank {self.rank} digest {d:#x}, max {int(t[0].item()):#x}, min {-int(t[1].item()):#x}")
