"""TEST INFRASTRUCTURE: bench.py's main() as one rank of `python -m torch.distributed.run`, with the engine library replaced by the emulated build
(tests/test_emu_engine_cpu.py) and torch.cuda answered by a stand-in — what tests/test_bench_cpu.py does in-process, for the N > 1 launch the
driver uses (`--gpus N` under torch.distributed.run, one rank per GPU).  KB_DIST_BACKEND=gloo stands in for RCCL.  The numbers of such a run
mean nothing; the control flow (rendezvous, barriers, MAX over ranks, the digest comparison of the replicas, ONE line from rank 0) is what runs."""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.dirname(HERE)]

import torch  # noqa: E402

torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 1
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None

engine = importlib.import_module("kube-batch_amd.engine")
engine.LIB_PATH, engine._LIB = os.environ["KB_EMU_LIB"], None

# the task-row split keeps its round buffers where the engine's "device" memory is: on the emulated device that is host memory
distmod = importlib.import_module("kube-batch_amd.dist")
_ShardedCycle = distmod.ShardedCycle


class _EmulatedShardedCycle(_ShardedCycle):
    def __init__(self, conf, snap, device=0, window=0, commit_batch=0, actions=("allocate", "backfill"), **kw):
        if os.environ.get("KB_EMU_BREAK_SHARDED") == "1":      # tests/test_bench_cpu.py: what bench.py prints when the split cannot be set up
            raise RuntimeError("the task-row split is broken on purpose")
        eng = engine.Engine(conf, device=device, window=window, commit_batch=commit_batch)
        eng.load(snap)
        cpu = torch.device("cpu")
        super().__init__(conf, snap, backend=distmod.EngineBackend(eng, cpu), buffer_device=cpu, actions=actions, **kw)

    def step(self, *a, **k):
        if os.environ.get("KB_EMU_HANG_SHARDED") == "1":       # tests/test_bench_cpu.py: a collective of the split that never completes
            import time
            while True:
                time.sleep(3600)
        return super().step(*a, **k)


distmod.ShardedCycle = _EmulatedShardedCycle

import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
