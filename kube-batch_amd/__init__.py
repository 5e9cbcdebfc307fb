"""kube-batch_amd — MI355X-native allocate/backfill engine behind kube-batch's Action interface.

The directory name carries a hyphen (it mirrors the reference's repository name), so import it with
    importlib.import_module("kube-batch_amd")
Sub-modules: abi (ctypes mirror of include/kb_engine.h), conf (scheduler YAML -> tiers),
snapshot (Session -> SoA flattener, synthetic clusters), engine (the C-ABI binding), framework
(Python mirror of framework.Session/Action used by the harness and the parity tests).
"""
from . import abi, conf, snapshot  # noqa: F401
