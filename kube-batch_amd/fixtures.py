"""Fixtures restating the reference's own test inputs for the allocate path, as Kubernetes-shaped objects.

  * allocate_cases(): pkg/scheduler/actions/allocate/allocate_test.go:51-144 (two cases, drf+proportion tiers)
  * example_job(): BASELINE config 1 — example/job.yaml (Job qj-1, 6 pods x cpu "1", PodGroup minMember 6) on
    3 worker nodes (hack/e2e-kind-config.yaml); node size is not given by the reference: 4 cpu / 8 GiB / 110 pods
  * build_resource_list(): pkg/scheduler/util/test_utils.go:34-41 (every list carries nvidia.com/gpu: "0")
"""
from typing import Dict, List, Tuple

import yaml

from . import abi
from .conf import PluginOption, SchedulerConf, load_scheduler_conf, tiers_literal
from .snapshot import Node, Pod, PodGroup, Queue, SessionSnapshot, flatten

GPU = "nvidia.com/gpu"


def build_resource_list(cpu: str, memory: str) -> Dict[str, str]:
    return {"cpu": cpu, "memory": memory, GPU: "0"}


def build_pod(ns, name, node, phase, req, group) -> Pod:
    """util.BuildPod (test_utils.go:60-92): UID "<ns>-<name>", one container."""
    return Pod(namespace=ns, name=name, uid=f"{ns}-{name}", node_name=node, phase=phase, containers=[req], group_name=group)


def allocate_test_tiers() -> SchedulerConf:
    """allocate_test.go:183-198: drf {Preemptable, JobOrder}, proportion {QueueOrder, Reclaimable}; all else nil."""
    return tiers_literal([
        PluginOption("drf", enabled=abi.EN_PREEMPTABLE | abi.EN_JOB_ORDER),
        PluginOption("proportion", enabled=abi.EN_QUEUE_ORDER | abi.EN_RECLAIMABLE),
    ])


def allocate_cases() -> List[Tuple[str, SessionSnapshot, Dict[str, str]]]:
    cases = []
    # allocate_test.go:51-85
    snap = flatten(
        nodes=[Node("n1", build_resource_list("2", "4Gi"))],
        pods=[build_pod("c1", "p1", "", "Pending", build_resource_list("1", "1G"), "pg1"),
              build_pod("c1", "p2", "", "Pending", build_resource_list("1", "1G"), "pg1")],
        pod_groups=[PodGroup("c1", "pg1", queue="c1")],
        queues=[Queue("c1", 1)])
    cases.append(("one Job with two Pods on one node", snap, {"c1/p1": "n1", "c1/p2": "n1"}))
    # allocate_test.go:86-144
    snap = flatten(
        nodes=[Node("n1", build_resource_list("2", "4G"))],
        pods=[build_pod("c1", "p1", "", "Pending", build_resource_list("1", "1G"), "pg1"),
              build_pod("c1", "p2", "", "Pending", build_resource_list("1", "1G"), "pg1"),
              build_pod("c2", "p1", "", "Pending", build_resource_list("1", "1G"), "pg2"),
              build_pod("c2", "p2", "", "Pending", build_resource_list("1", "1G"), "pg2")],
        pod_groups=[PodGroup("c1", "pg1", queue="c1"), PodGroup("c2", "pg2", queue="c2")],
        queues=[Queue("c1", 1), Queue("c2", 1)])
    cases.append(("two Jobs on one node", snap, {"c2/p1": "n1", "c1/p1": "n1"}))
    return cases


# the content of example/job.yaml that the scheduler reads, restated (batch/v1 Job + PodGroup)
EXAMPLE_JOB_MANIFEST = """
apiVersion: batch/v1
kind: Job
metadata: {name: qj-1}
spec:
  parallelism: 6
  completions: 6
  template:
    metadata:
      annotations: {scheduling.k8s.io/group-name: qj-1}
    spec:
      schedulerName: kube-batch
      containers:
      - name: busybox
        resources: {requests: {cpu: "1"}}
---
apiVersion: scheduling.incubator.k8s.io/v1alpha1
kind: PodGroup
metadata: {name: qj-1}
spec: {minMember: 6}
"""

# example/kube-batch-conf.yaml restricted to BASELINE config 1 ("allocate-only")
EXAMPLE_CONF = """
actions: "allocate"
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


def load_manifests(text: str, namespace: str = "default", default_queue: str = "default"):
    """batch/v1 Job (expanded to `parallelism` pods named <job>-<i>) and PodGroup documents -> (pods, pod_groups)."""
    pods, pgs = [], []
    for doc in yaml.safe_load_all(text):
        if not doc:
            continue
        kind = doc.get("kind")
        meta = doc.get("metadata", {})
        ns = meta.get("namespace", namespace)
        if kind == "Job":
            tpl = doc["spec"]["template"]
            group = (tpl.get("metadata", {}).get("annotations", {}) or {}).get("scheduling.k8s.io/group-name", "")
            spec = tpl["spec"]
            conts = [dict((c.get("resources", {}) or {}).get("requests", {}) or {}) for c in spec.get("containers", [])]
            inits = [dict((c.get("resources", {}) or {}).get("requests", {}) or {}) for c in spec.get("initContainers", [])]
            conts = [{k: str(v) for k, v in c.items()} for c in conts]
            inits = [{k: str(v) for k, v in c.items()} for c in inits]
            for i in range(int(doc["spec"].get("parallelism", 1))):
                pods.append(Pod(namespace=ns, name=f"{meta['name']}-{i}", containers=[dict(c) for c in conts],
                                init_containers=[dict(c) for c in inits], group_name=group,
                                node_selector=dict(spec.get("nodeSelector", {}) or {})))
        elif kind == "PodGroup":
            sp = doc.get("spec", {})
            pgs.append(PodGroup(ns, meta["name"], min_member=int(sp.get("minMember", 0)), queue=sp.get("queue", "") or default_queue))
    return pods, pgs


def example_job() -> Tuple[SchedulerConf, SessionSnapshot]:
    pods, pgs = load_manifests(EXAMPLE_JOB_MANIFEST)
    nodes = [Node(f"kind-worker{i}", {"cpu": "4", "memory": "8Gi", "pods": "110"}) for i in (1, 2, 3)]
    snap = flatten(nodes, pods, pgs, [Queue("default", 1)])          # config/queue/default.yaml: weight 1
    return load_scheduler_conf(EXAMPLE_CONF), snap
