"""The remaining known-answer tests SURVEY.md §8c lists for the path, restated against the pieces that carry them here:
util_test.go (conf parsing + defaults) on the conf loader, job_info_test.go (status index / Allocated / TotalRequest) on the
flattener's task statuses and the plugin-open sums, node_info_test.go's RemovePod on the AddTask / RemoveTask accounting."""
import importlib

import numpy as np

import pyref
from test_pyref_vs_oracle import _tiers

kbm = importlib.import_module("kube-batch_amd")
abi, conf, S, fx = kbm.abi, kbm.conf, kbm.snapshot, kbm.fixtures

G = 1e9


def test_load_scheduler_conf_defaults():
    """pkg/scheduler/util_test.go:27-146 TestLoadSchedulerConf: two tiers, every Enabled* pointer defaulted to true."""
    c = conf.load_scheduler_conf("""
actions: "allocate, backfill"
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
""")
    assert c.actions == ["allocate", "backfill"]
    assert [[p.name for p in t] for t in c.tiers] == [["priority", "gang", "conformance"], ["drf", "predicates", "proportion", "nodeorder"]]
    assert all(p.enabled == abi.EN_ALL for t in c.tiers for p in t)          # nine Enabled* fields, all &trueValue
    assert all(p.arguments == {} for t in c.tiers for p in t)
    # an explicit false survives the defaulting (plugins/defaults.go only fills nil pointers)
    c = conf.load_scheduler_conf('actions: "allocate"\ntiers:\n- plugins:\n  - name: gang\n    enableJobReady: false\n')
    assert c.tiers[0][0].enabled == abi.EN_ALL & ~abi.EN_JOB_READY


def test_job_info_add_task_info():
    """api/job_info_test.go:35-96 TestAddTaskInfo: a pending pod with a node name is Bound (api/helpers.go:35-61); Allocated sums
    the allocated-status tasks (4000m, 4G), TotalRequest every task (5000m, 5G).  drf's per-job `allocated` (drf.go:71-77) and
    proportion's per-queue `allocated` / `request` (proportion.go:85-97) are the same sums over the session's one job."""
    rl = fx.build_resource_list
    pods = [fx.build_pod("c1", "p1", "", "Pending", rl("1000m", "1G"), "pg"), fx.build_pod("c1", "p2", "n1", "Running", rl("2000m", "2G"), "pg"),
            fx.build_pod("c1", "p3", "n1", "Pending", rl("1000m", "1G"), "pg"), fx.build_pod("c1", "p4", "n1", "Pending", rl("1000m", "1G"), "pg")]
    snap = S.flatten(nodes=[S.Node("n1", rl("8000m", "10G"))], pods=pods, pod_groups=[S.PodGroup("c1", "pg", queue="q")], queues=[S.Queue("q", 1)])
    assert snap.names["tasks"] == ["c1/p1", "c1/p2", "c1/p3", "c1/p4"]
    assert snap.task_status.tolist() == [abi.TASK_PENDING, abi.TASK_RUNNING, abi.TASK_BOUND, abi.TASK_BOUND]
    p = pyref.Session(_tiers(conf.load_scheduler_conf()), snap)
    assert (p.jalloc[0].cpu, p.jalloc[0].mem) == (4000.0, 4 * G)
    assert (p.qattr[0]["allocated"].cpu, p.qattr[0]["allocated"].mem) == (4000.0, 4 * G)
    assert (p.qattr[0]["request"].cpu, p.qattr[0]["request"].mem) == (5000.0, 5 * G)
    assert p.ready_num(0) == 3


def test_job_info_delete_task_info():
    """api/job_info_test.go:98-197 TestDeleteTaskInfo, both cases: a job without the removed pod has Allocated (3000m, 3G) and
    TotalRequest (4000m, 4G).  On this path a deleted pod is simply absent from the next snapshot."""
    rl = fx.build_resource_list
    for removed, kept in (("p2", [("p1", "", "Pending", "1000m", "1G"), ("p3", "n1", "Running", "3000m", "3G")]),):
        pods = [fx.build_pod("c1", n, node, ph, rl(c, m), "pg") for n, node, ph, c, m in kept]
        snap = S.flatten(nodes=[S.Node("n1", rl("8000m", "10G"))], pods=pods, pod_groups=[S.PodGroup("c1", "pg", queue="q")], queues=[S.Queue("q", 1)])
        p = pyref.Session(_tiers(conf.load_scheduler_conf()), snap)
        assert (p.jalloc[0].cpu, p.jalloc[0].mem) == (3000.0, 3 * G)
        assert (p.qattr[0]["request"].cpu, p.qattr[0]["request"].mem) == (4000.0, 4 * G)


def test_node_info_remove_pod():
    """api/node_info_test.go:107-164 TestNodeInfo_RemovePod: node 8000m / 10G with running pods of 1, 2 and 3 cpu; removing the
    second leaves Idle (4000m, 6G).  The flattener does the three AddTask calls, pyref's RemoveTask the removal."""
    rl = fx.build_resource_list
    pods = [fx.build_pod("c1", f"p{i}", "n1", "Running", rl(f"{i}000m", f"{i}G"), "pg") for i in (1, 2, 3)]
    snap = S.flatten(nodes=[S.Node("n1", rl("8000m", "10G"))], pods=pods, pod_groups=[S.PodGroup("c1", "pg", queue="q")], queues=[S.Queue("q", 1)])
    assert snap.node_idle[:2, 0].tolist() == [2000.0, 4 * G] and snap.node_pod_cnt.tolist() == [3]
    p = pyref.Session(_tiers(conf.load_scheduler_conf()), snap)
    assert p.node_remove_task(snap.names["tasks"].index("c1/p2"))
    assert (p.idle[0].cpu, p.idle[0].mem) == (4000.0, 6 * G) and p.podcnt == [2]
    assert (p.rel[0].cpu, p.rel[0].mem) == (0.0, 0.0)


def test_arguments_get_int():
    """framework/arguments_test.go:29-76 TestArgumentsGetInt through the conf -> kb_config path: an absent key, an unparsable value
    and an empty value leave the default (args_set bit clear); "15" is taken."""
    def nodeorder_args(arguments):
        c = conf.SchedulerConf(actions=["allocate"], tiers=[[conf.PluginOption("nodeorder", arguments=arguments)]])
        cfg, _keep = c.to_abi()
        return cfg.plugins[0].args_set, cfg.plugins[0].args[0]   # KB_ARG_NODEORDER_LEAST
    assert nodeorder_args({"anotherkey": "12"}) == (0, 0)
    assert nodeorder_args({"leastrequested.weight": "15"}) == (1, 15)
    assert nodeorder_args({"leastrequested.weight": "errorvalue"}) == (0, 0)
    assert nodeorder_args({"leastrequested.weight": ""}) == (0, 0)


def test_add_task_on_a_node_without_room_changes_nothing():
    """cache/cache_test.go:366-436 (Bind with sufficient / insufficient resources) exercises NodeInfo.AddTask's guarantee
    (api/node_info.go:169-171: "If error occurs both task and node are guaranteed to be in the original state")."""
    rl = fx.build_resource_list
    pods = [fx.build_pod("c1", "small", "", "Pending", rl("1000m", "1G"), "pg"), fx.build_pod("c1", "huge", "", "Pending", rl("5000m", "50G"), "pg")]
    snap = S.flatten(nodes=[S.Node("n1", rl("2000m", "10G"))], pods=pods, pod_groups=[S.PodGroup("c1", "pg", queue="q")], queues=[S.Queue("q", 1)])
    p = pyref.Session(_tiers(conf.load_scheduler_conf()), snap)
    huge, small = snap.names["tasks"].index("c1/huge"), snap.names["tasks"].index("c1/small")
    before = (p.idle[0].cpu, p.idle[0].mem, p.podcnt[0], p.nzc[0], p.nzm[0])
    assert not p.node_add_task(huge, 0)
    assert (p.idle[0].cpu, p.idle[0].mem, p.podcnt[0], p.nzc[0], p.nzm[0]) == before and p.tnode[huge] == pyref.NONE and not p.onnode[huge]
    assert p.node_add_task(small, 0)
    assert (p.idle[0].cpu, p.idle[0].mem, p.podcnt[0]) == (1000.0, 9 * G, 1) and p.tnode[small] == 0
