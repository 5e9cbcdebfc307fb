"""ctypes binding of the C ABI (include/kb_engine.h -> libkbengine.so).

The library is built in-tree by `build()` (hipcc --offload-arch=gfx950) and loaded from
kube-batch_amd/libkbengine.so.  There is no CPU fallback: if the library is missing or no HIP device
is visible the calls raise (the Go side falls back to the stock allocate action in that case).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KB_ENGINE_LIB") or os.path.join(_HERE, "libkbengine.so")   # KB_ENGINE_LIB: A/B builds of the same source
_LIB = None

EXPORTS = ["kb_engine_create", "kb_engine_destroy", "kb_last_error", "kb_session_load", "kb_session_reset", "kb_run_allocate",
           "kb_run_backfill", "kb_run_preempt", "kb_run_reclaim", "kb_get_evictions", "kb_engine_use_stream", "kb_eval_matrix", "kb_argmax_rows", "kb_bench_matrix", "kb_get_binds",
           "kb_get_task_state", "kb_get_node_state", "kb_get_shares", "kb_get_stats", "kb_round_begin",
           "kb_round_candidates", "kb_round_commit", "kb_round_apply", "kb_round_check", "kb_round_check_result", "kb_round_delta_doubles",
           "kb_round_decisions"]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{abi.ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP engine for gfx950 in-tree (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    subprocess.check_call(["make", "-C", src_dir, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(abi.KB_E_DEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950). "
                                               "The engine has no CPU path.")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.kb_engine_create.argtypes = [C.POINTER(abi.Config), C.POINTER(vp)]
        L.kb_engine_destroy.argtypes = [vp]
        L.kb_engine_destroy.restype = None
        L.kb_last_error.argtypes = [vp]
        L.kb_last_error.restype = C.c_char_p
        L.kb_session_load.argtypes = [vp, C.POINTER(abi.Snapshot)]
        L.kb_session_reset.argtypes = [vp]
        for n in ("kb_run_allocate", "kb_run_backfill"):
            getattr(L, n).argtypes = [vp, C.POINTER(abi.Decision), C.c_uint64, C.POINTER(C.c_uint64)]
        L.kb_run_preempt.argtypes = [vp, C.POINTER(abi.StmtOp), C.c_uint64, C.POINTER(C.c_uint64)]
        L.kb_run_reclaim.argtypes = [vp, C.POINTER(abi.StmtOp), C.c_uint64, C.POINTER(C.c_uint64)]
        L.kb_get_evictions.argtypes = [vp, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint64)]
        L.kb_eval_matrix.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)]
        L.kb_argmax_rows.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]
        L.kb_bench_matrix.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        L.kb_get_binds.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.kb_get_task_state.argtypes = [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]
        L.kb_get_node_state.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.kb_get_shares.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.kb_get_stats.argtypes = [vp, C.POINTER(abi.Stats)]
        L.kb_engine_use_stream.argtypes = [vp, C.c_uint64]
        L.kb_round_begin.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.kb_round_candidates.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64]
        L.kb_round_commit.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]
        L.kb_round_apply.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint32)]
        L.kb_round_check.argtypes = [vp, C.c_uint64, C.c_uint32]
        L.kb_round_check_result.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.kb_round_delta_doubles.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.kb_round_decisions.argtypes = [vp, C.POINTER(abi.Decision), C.c_uint64, C.POINTER(C.c_uint64)]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else C.POINTER(t)()


class Engine:
    """One kb_engine handle.  Mirrors the life-cycle the Go action drives: create (once per process),
    load(snapshot) per scheduling cycle, run_allocate / run_backfill per configured action."""

    def __init__(self, conf, device: int = 0, window: int = 0, commit_batch: int = 0, flags: int = 0, pressure_folded: bool = False):
        self.L = lib()
        cfg, self._keep = conf.to_abi(device=device, window=window, commit_batch=commit_batch, flags=flags, pressure_folded=pressure_folded)
        h = C.c_void_p()
        rc = self.L.kb_engine_create(C.byref(cfg), C.byref(h))
        if rc != abi.KB_OK:
            raise EngineError(rc, (self.L.kb_last_error(None) or b"").decode())
        self.h = h
        self.snap = None

    def _ck(self, rc):
        if rc != abi.KB_OK:
            raise EngineError(rc, (self.L.kb_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.kb_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, snap):
        self.snap = snap
        self._journals = []
        s = snap.to_abi()
        self._ck(self.L.kb_session_load(self.h, C.byref(s)))

    def reset(self):
        """Back to the just-loaded state from the pristine copy resident in HBM (no host upload)."""
        self._journals = []
        self._ck(self.L.kb_session_reset(self.h))

    def _run(self, fn):
        cap = max(int(self.snap.n_tasks), 1)
        # the caller's decision buffer, kept between calls and never cleared (the engine writes the records it reports): a fresh ctypes array of T
        # records is zero-filled — 16 MB per action at 1M tasks, twice per cycle, charged to the actions by every timing of this wrapper; a Go caller
        # allocates once (the journal buffer of run_preempt had the same artefact)
        buf = getattr(self, "_dec_buf", None)
        if buf is None or buf.shape[0] < cap:
            buf = self._dec_buf = np.empty((cap, 4), np.uint32)
        n = C.c_uint64()
        self._ck(fn(self.h, C.cast(buf.ctypes.data, C.POINTER(abi.Decision)), cap, C.byref(n)))
        return buf[: n.value, :3].copy()

    def run_allocate(self):
        """-> uint32[n,3] (task, node, kind) in the order the reference loop places them."""
        return self._run(self.L.kb_run_allocate)

    def run_backfill(self):
        return self._run(self.L.kb_run_backfill)

    def run_reclaim(self):
        """The reclaim action (actions/reclaim/reclaim.go): ssn.Evict / ssn.Pipeline entries in last_journal, stmt 0."""
        return self.run_preempt(fn=self.L.kb_run_reclaim)

    def run_preempt(self, fn=None):
        """The preempt action (actions/preempt/preempt.go) -> uint32[n,4] journal (op, task, node, stmt), abi.OP_*, in order."""
        cap = 4 * max(int(self.snap.n_tasks), 1) + 16
        n = C.c_uint64()
        # the caller's journal buffer, kept between calls and never cleared (the engine writes the entries it reports): a fresh zero-filled ctypes
        # array of 4 T + 16 records was 64 MB — 10 ms — per call at 1M tasks, charged to the action by every timing of this wrapper (round 6)
        buf = getattr(self, "_journal_buf", None)
        if buf is None or buf.shape[0] < cap:
            buf = self._journal_buf = np.empty((cap, 4), np.uint32)
        rc = (fn or self.L.kb_run_preempt)(self.h, C.cast(buf.ctypes.data, C.POINTER(abi.StmtOp)), cap, C.byref(n))
        self._ck(rc)
        self.last_journal = buf[: n.value].copy()
        self._journals.append(self.last_journal)
        return np.zeros((0, 3), np.uint32)       # no ssn.Allocate / ssn.Pipeline decisions: the journal carries the Statement ops

    def journal(self):
        """The journals of every preempt / reclaim action since the session was loaded (or reset), concatenated in action order."""
        return np.concatenate(self._journals) if self._journals else np.zeros((0, 4), np.uint32)

    def evictions(self):
        """Task ids the committed statements handed to cache.Evict, in that order."""
        cap = max(int(self.snap.n_tasks), 1)
        out = np.empty(cap, np.uint32)
        n = C.c_uint64()
        self._ck(self.L.kb_get_evictions(self.h, _p(out, C.c_uint32), cap, C.byref(n)))
        return out[: n.value].copy()

    def run(self, actions):
        out = [getattr(self, "run_" + a)() for a in actions]
        return np.concatenate(out) if out else np.zeros((0, 3), np.uint32)

    def eval_matrix(self, t0, t1, fit_mode=1):
        N = self.snap.n_nodes
        mask = np.zeros((t1 - t0, (N + 7) // 8), np.uint8)
        score = np.zeros((t1 - t0, N), np.uint16)
        self._ck(self.L.kb_eval_matrix(self.h, t0, t1, fit_mode, _p(mask, C.c_uint8), _p(score, C.c_uint16)))
        return mask, score

    def argmax_rows(self, t0, t1, k, fit_mode=1):
        nodes = np.empty((t1 - t0, k), np.uint32)
        score = np.empty((t1 - t0, k), np.uint16)
        self._ck(self.L.kb_argmax_rows(self.h, t0, t1, fit_mode, k, _p(nodes, C.c_uint32), _p(score, C.c_uint16)))
        return nodes, score

    def bench_matrix(self, t0, t1, reps=10, fit_mode=1):
        ms = C.c_double()
        self._ck(self.L.kb_bench_matrix(self.h, t0, t1, fit_mode, reps, C.byref(ms)))
        return ms.value

    def binds(self):
        out = np.empty(self.snap.n_tasks, np.uint32)
        self._ck(self.L.kb_get_binds(self.h, _p(out, C.c_uint32)))
        return out

    def task_state(self):
        st = np.empty(self.snap.n_tasks, np.uint8)
        nd = np.empty(self.snap.n_tasks, np.uint32)
        self._ck(self.L.kb_get_task_state(self.h, _p(st, C.c_uint8), _p(nd, C.c_uint32)))
        return st, nd

    def node_state(self):
        R, N = self.snap.n_res, self.snap.n_nodes
        idle = np.empty((R, N)); rel = np.empty((R, N))
        nzc = np.empty(N, np.int64); nzm = np.empty(N, np.int64); cnt = np.empty(N, np.int32)
        self._ck(self.L.kb_get_node_state(self.h, _p(idle, C.c_double), _p(rel, C.c_double), _p(nzc, C.c_int64),
                                          _p(nzm, C.c_int64), _p(cnt, C.c_int32)))
        return idle, rel, nzc, nzm, cnt

    def shares(self):
        R, J, Q = self.snap.n_res, self.snap.n_jobs, self.snap.n_queues
        js = np.empty(J); qs = np.empty(Q); des = np.empty((R, Q))
        self._ck(self.L.kb_get_shares(self.h, _p(js, C.c_double), _p(qs, C.c_double), _p(des, C.c_double)))
        return js, qs, des

    def stats(self):
        st = abi.Stats()
        self._ck(self.L.kb_get_stats(self.h, C.byref(st)))
        return st.as_dict()

    # ---- round-granular API (task-row sharding across GPUs; see kube-batch_amd/dist.py)
    def use_stream(self, hip_stream: int):
        """Run the engine's kernels on the caller's HIP stream (0: back to the engine's own)."""
        self._ck(self.L.kb_engine_use_stream(self.h, int(hip_stream)))

    def round_begin(self, action: int):
        n, m, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self.L.kb_round_begin(self.h, action, C.byref(n), C.byref(m), C.byref(l)))
        return n.value, m.value, l.value

    def round_candidates(self, mrow0: int, mrow1: int, dev_keys_ptr: int):
        self._ck(self.L.kb_round_candidates(self.h, mrow0, mrow1, dev_keys_ptr))

    def round_commit(self, dev_all_keys_ptr: int, own_row0: int, own_row1: int, dev_delta_ptr: int):
        self._ck(self.L.kb_round_commit(self.h, dev_all_keys_ptr, own_row0, own_row1, dev_delta_ptr))

    def round_apply(self, dev_delta_ptr: int):
        done = C.c_uint32()
        self._ck(self.L.kb_round_apply(self.h, dev_delta_ptr, C.byref(done)))

    def round_check(self, dev_delta_ptr: int, against_live: bool = False):
        """deferred cross-check of a round's reduced deltas (include/kb_engine.h); queued, nothing waited for"""
        self._ck(self.L.kb_round_check(self.h, dev_delta_ptr, 1 if against_live else 0))

    def round_check_result(self) -> int:
        n = C.c_uint32()
        self._ck(self.L.kb_round_check_result(self.h, C.byref(n)))
        return n.value

    def round_delta_doubles(self) -> int:
        n = C.c_uint64()
        self._ck(self.L.kb_round_delta_doubles(self.h, C.byref(n)))
        return n.value

    def round_decisions(self):
        cap = max(int(self.snap.n_tasks), 1)
        buf = getattr(self, "_dec_buf", None)          # (the buffer of _run: kept between calls, never cleared)
        if buf is None or buf.shape[0] < cap:
            buf = self._dec_buf = np.empty((cap, 4), np.uint32)
        n = C.c_uint64()
        self._ck(self.L.kb_round_decisions(self.h, C.cast(buf.ctypes.data, C.POINTER(abi.Decision)), cap, C.byref(n)))
        return buf[: n.value, :3].copy()
