"""Scheduler configuration: YAML `actions:` + `tiers:` -> the engine's kb_config.

Mirrors pkg/scheduler/util.go:31-73 (defaultSchedulerConf, loadSchedulerConf),
pkg/scheduler/conf/scheduler_conf.go:20-56 (Tier / PluginOption) and
pkg/scheduler/plugins/defaults.go:22-52 (every Enabled* defaults to true when the YAML is loaded).
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import yaml

from . import abi

# pkg/scheduler/util.go:31-42
DEFAULT_SCHEDULER_CONF = """
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
"""

_ENABLE_KEYS = {
    "enableJobOrder": abi.EN_JOB_ORDER, "enableJobReady": abi.EN_JOB_READY,
    "enableJobPipelined": abi.EN_JOB_PIPELINED, "enableTaskOrder": abi.EN_TASK_ORDER,
    "enablePreemptable": abi.EN_PREEMPTABLE, "enableReclaimable": abi.EN_RECLAIMABLE,
    "enableQueueOrder": abi.EN_QUEUE_ORDER, "enablePredicate": abi.EN_PREDICATE,
    "enableNodeOrder": abi.EN_NODE_ORDER,
}
# plugins/nodeorder/nodeorder.go:30-39, plugins/predicates/predicates.go:33-40
_ARG_SLOTS = {
    "nodeorder": {"leastrequested.weight": 0, "mostrequested.weight": 1, "nodeaffinity.weight": 2,
                  "podaffinity.weight": 3, "balancedresource.weight": 4},
    "predicates": {"predicate.MemoryPressureEnable": 0, "predicate.DiskPressureEnable": 1,
                   "predicate.PIDPressureEnable": 2},
}


@dataclass
class PluginOption:
    """conf.PluginOption.  `enabled` holds the Enabled* pointers as a bitmask (nil == bit clear)."""
    name: str
    enabled: int = abi.EN_ALL
    arguments: Dict[str, str] = field(default_factory=dict)


@dataclass
class SchedulerConf:
    actions: List[str]
    tiers: List[List[PluginOption]]

    def pressure_flags(self):
        """(MemoryPressureEnable, DiskPressureEnable, PIDPressureEnable) of the predicates plugin (plugins/predicates/
        predicates.go:66-110: framework.Arguments.GetBool, default false) — what snapshot.flatten(pressure=...) folds into the
        static class table."""
        out = [False, False, False]
        for tier in self.tiers:
            for po in tier:
                if po.name != "predicates":
                    continue
                for key, slot in _ARG_SLOTS["predicates"].items():
                    sval = str((po.arguments or {}).get(key, ""))
                    if sval in ("1", "t", "T", "TRUE", "true", "True"):
                        out[slot] = True
                    elif sval in ("0", "f", "F", "FALSE", "false", "False"):
                        out[slot] = False
        return tuple(out)

    def to_abi(self, device: int = 0, window: int = 0, commit_batch: int = 0, flags: int = 0, pressure_folded: bool = False):
        """Returns (kb_config, keepalive) — keepalive owns the arrays the struct points to.  pressure_folded: the snapshot was
        flattened with pressure=self.pressure_flags(), so the predicates plugin's pressure arguments are not handed to the
        engine (which answers KB_E_UNSUPPORTED to them: it has no per-node condition state, kb_engine.h KB_ARG_PRED_*)."""
        n_p = sum(len(t) for t in self.tiers)
        tier_begin = (C.c_uint32 * (len(self.tiers) + 1))()
        plugins = (abi.PluginOption * max(n_p, 1))()
        k = 0
        for ti, tier in enumerate(self.tiers):
            tier_begin[ti] = k
            for po in tier:
                if po.name not in abi.PLUGIN_IDS:
                    raise ValueError(f"unknown plugin {po.name!r}")
                o = plugins[k]
                o.plugin = abi.PLUGIN_IDS[po.name]
                o.enabled = po.enabled
                o.args_set = 0
                for key, val in (po.arguments or {}).items():
                    slot = _ARG_SLOTS.get(po.name, {}).get(key)
                    if slot is None or val == "" or (pressure_folded and po.name == "predicates"):
                        continue
                    sval = str(val)
                    if po.name == "predicates":          # framework/arguments.go:49-66 strconv.ParseBool
                        if sval in ("1", "t", "T", "TRUE", "true", "True"):
                            iv = 1
                        elif sval in ("0", "f", "F", "FALSE", "false", "False"):
                            iv = 0
                        else:
                            continue
                    else:                                  # framework/arguments.go:29-46 strconv.Atoi
                        try:
                            iv = int(sval, 10)
                        except ValueError:
                            continue
                    o.args[slot] = iv
                    o.args_set |= 1 << slot
                k += 1
        tier_begin[len(self.tiers)] = k
        cfg = abi.Config()
        cfg.version = abi.KB_ABI_VERSION
        cfg.n_tiers = len(self.tiers)
        cfg.tier_begin = C.cast(tier_begin, C.POINTER(C.c_uint32))
        cfg.plugins = C.cast(plugins, C.POINTER(abi.PluginOption))
        cfg.device = device
        cfg.window = window
        cfg.commit_batch = commit_batch
        cfg.flags = flags
        return cfg, (tier_begin, plugins)


def load_scheduler_conf(conf_str: Optional[str] = None) -> SchedulerConf:
    """loadSchedulerConf (pkg/scheduler/util.go:44-73): parse, then ApplyPluginConfDefaults on every option."""
    doc = yaml.safe_load(conf_str if conf_str is not None else DEFAULT_SCHEDULER_CONF) or {}
    actions = [a.strip() for a in str(doc.get("actions", "")).split(",") if a.strip()]
    tiers = []
    for tier in doc.get("tiers") or []:
        opts = []
        for p in tier.get("plugins") or []:
            enabled = 0
            for key, bit in _ENABLE_KEYS.items():
                v = p.get(key)
                if v is None or bool(v):      # nil -> default true (plugins/defaults.go)
                    enabled |= bit
            args = {str(k): str(v) for k, v in (p.get("arguments") or {}).items()}
            opts.append(PluginOption(name=p["name"], enabled=enabled, arguments=args))
        tiers.append(opts)
    return SchedulerConf(actions=actions, tiers=tiers)


def tiers_literal(*tiers: List[PluginOption]) -> SchedulerConf:
    """Hand-written tier list as the reference's action tests build it (Enabled* nil unless set)."""
    return SchedulerConf(actions=["allocate"], tiers=[list(t) for t in tiers])
