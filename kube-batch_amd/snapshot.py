"""Session snapshot in structure-of-arrays form, the flattener that produces it from
Kubernetes-shaped objects, and the deterministic synthetic cluster generator.

The flattener mirrors what the Go shim does inside `Execute(ssn)`:
  * cache.AddNode/AddPod/AddPodGroup/AddQueue + Snapshot()  (pkg/scheduler/cache/event_handlers.go,
    cache.go:627-683) -> api.NodeInfo / JobInfo / TaskInfo / QueueInfo
  * api.NewResource (pkg/scheduler/api/resource_info.go:73-90): cpu & scalars in milli units, memory bytes
  * api.NewTaskInfo / GetPodResourceRequest (api/job_info.go:69-93, api/pod_info.go:53-73)
  * k8s nodeinfo non-zero requests (vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/util/non_zero.go:32-61)
  * static predicates p2..p7 folded into a task-class x node-class bit table
and emits everything in the canonical order of SURVEY.md §8c.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from fractions import Fraction
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import abi

# ------------------------------------------------------------------------------------------------
# resource.Quantity (vendor/k8s.io/apimachinery/pkg/api/resource/quantity.go)
# ------------------------------------------------------------------------------------------------
_SUFFIX = {"n": Fraction(1, 10**9), "u": Fraction(1, 10**6), "m": Fraction(1, 1000), "": Fraction(1),
           "k": Fraction(10**3), "M": Fraction(10**6), "G": Fraction(10**9), "T": Fraction(10**12),
           "P": Fraction(10**15), "E": Fraction(10**18),
           "Ki": Fraction(2**10), "Mi": Fraction(2**20), "Gi": Fraction(2**30), "Ti": Fraction(2**40),
           "Pi": Fraction(2**50), "Ei": Fraction(2**60)}


def parse_quantity(q) -> Fraction:
    if isinstance(q, (int, Fraction)):
        return Fraction(q)
    s = str(q).strip()
    for suf in sorted(_SUFFIX, key=len, reverse=True):
        if suf and s.endswith(suf):
            return Fraction(s[: -len(suf)]) * _SUFFIX[suf]
    if "e" in s or "E" in s:
        mant, exp = s.replace("E", "e").split("e")
        return Fraction(mant) * Fraction(10) ** int(exp)
    return Fraction(s)


def quantity_value(q) -> int:
    """Quantity.Value(): rounds up (quantity.go:689-696)."""
    return math.ceil(parse_quantity(q))


def quantity_milli_value(q) -> int:
    """Quantity.MilliValue(): rounds up (quantity.go:699-711)."""
    return math.ceil(parse_quantity(q) * 1000)


def is_scalar_resource_name(name: str) -> bool:
    """v1helper.IsScalarResourceName (vendor/k8s.io/kubernetes/pkg/apis/core/v1/helper/helpers.go:38-103)."""
    native = ("/" not in name) or ("kubernetes.io/" in name)
    extended = (not native) and (not name.startswith("requests."))
    return (extended or name.startswith("hugepages-") or ("kubernetes.io/" in name)
            or name.startswith("attachable-volumes-"))


DEFAULT_MILLI_CPU_REQUEST = 100                 # non_zero.go:32-37
DEFAULT_MEMORY_REQUEST = 200 * 1024 * 1024


# ------------------------------------------------------------------------------------------------
# Kubernetes-shaped input objects (only the fields the hot path reads)
# ------------------------------------------------------------------------------------------------
@dataclass
class Node:
    name: str
    allocatable: Dict[str, str]
    labels: Dict[str, str] = field(default_factory=dict)
    taints: List[Tuple[str, str, str]] = field(default_factory=list)   # (key, value, effect)
    unschedulable: bool = False
    ready: bool = True                    # every NodeReady condition True (predicates.go:1675-1700)
    network_unavailable: bool = False
    # MemoryPressure / DiskPressure / PIDPressure condition status == "True"; read only when the predicates plugin's optional
    # checks are enabled (plugins/predicates/predicates.go:94-107,201-247)
    memory_pressure: bool = False
    disk_pressure: bool = False
    pid_pressure: bool = False


@dataclass
class Pod:
    namespace: str
    name: str
    containers: List[Dict[str, str]]                       # one requests dict per container
    group_name: str = ""
    node_name: str = ""
    phase: str = "Pending"
    uid: Optional[str] = None
    init_containers: List[Dict[str, str]] = field(default_factory=list)
    priority: Optional[int] = None
    creation: int = 0
    deleting: bool = False                                  # DeletionTimestamp != nil
    node_selector: Dict[str, str] = field(default_factory=dict)
    tolerations: List[Tuple[str, str, str, str]] = field(default_factory=list)  # (key, operator, value, effect)
    # nodeAffinity.preferredDuringSchedulingIgnoredDuringExecution: (weight, [(key, operator, (values...))])
    preferred_affinity: List[Tuple[int, List[Tuple[str, str, Tuple[str, ...]]]]] = field(default_factory=list)
    # nodeAffinity.requiredDuringSchedulingIgnoredDuringExecution.nodeSelectorTerms: None = field absent (selects every node);
    # a list of terms (matchExpressions, matchFields), each a list of (key, operator, (values...)); terms are ORed, an empty
    # list or an empty term selects nothing (vendor/.../apis/core/v1/helper/helpers.go:285-314)
    required_affinity: Optional[List[Tuple[List[Tuple[str, str, Tuple[str, ...]]], List[Tuple[str, str, Tuple[str, ...]]]]]] = None
    # container ports with a hostPort: (hostIP, protocol, hostPort); "" -> 0.0.0.0 / TCP (nodeinfo/host_ports.go:137-144)
    host_ports: List[Tuple[str, str, int]] = field(default_factory=list)
    # ---- inter-pod (anti)affinity (spec.affinity.podAffinity / podAntiAffinity).  A term is (namespaces, selector, topologyKey):
    # namespaces () = the pod's own; selector None = nil LabelSelector (matches nothing), else (matchLabels pairs,
    # matchExpressions [(key, operator, (values...))]) with an empty selector matching everything
    # (vendor/k8s.io/apimachinery/pkg/apis/meta/v1/helpers.go:34-70).  Preferred terms carry their weight first.
    labels: Dict[str, str] = field(default_factory=dict)
    pod_affinity_required: List[tuple] = field(default_factory=list)
    pod_affinity_preferred: List[tuple] = field(default_factory=list)        # (weight, term)
    pod_anti_affinity_required: List[tuple] = field(default_factory=list)
    pod_anti_affinity_preferred: List[tuple] = field(default_factory=list)   # (weight, term)
    # the cache already holds the pod on node_name (status Binding) while the pod object's Spec.NodeName is still empty
    spec_node_name_empty: bool = False
    priority_class_name: str = ""       # conformance plugin: system-cluster-critical / system-node-critical are never evicted
    # resources.limits of every container and init container (only read for the pod's QoS class: the memory-pressure check)
    limits: List[Dict[str, str]] = field(default_factory=list)
    # A pod WITHOUT the group-name annotation that kube-batch itself schedules gets a shadow PodGroup from the cache (cache/event_handlers.go:
    # 45-68 getOrCreateJob, cache/util.go:47-92 createShadowPodGroup): job id = the UID of its controller, or its own; minMember = the
    # group-min-member annotation or 1; the default queue.  shadow_job: that job id ("" = none: the pod is outside the session)
    shadow_job: str = ""
    shadow_min_member: int = 1
    # spec.volumes names a PersistentVolumeClaim.  Inside the session a claim cannot veto a placement (build_interpod's docstring has
    # the walk through AssumePodVolumes); it only matters together with inter-pod terms
    has_volume_claim: bool = False


@dataclass
class PodGroup:
    namespace: str
    name: str
    min_member: int = 0
    queue: str = "default"
    creation: int = 0
    priority: int = 0          # resolved PriorityClass value (cache.go:661-668)


@dataclass
class Queue:
    name: str
    weight: int = 1
    creation: int = 0


def _task_status(p: Pod) -> int:
    """getTaskStatus (pkg/scheduler/api/helpers.go:35-61)."""
    if p.phase == "Running":
        return abi.TASK_RELEASING if p.deleting else abi.TASK_RUNNING
    if p.phase == "Pending":
        if p.deleting:
            return abi.TASK_RELEASING
        return abi.TASK_PENDING if not p.node_name else abi.TASK_BOUND
    if p.phase == "Succeeded":
        return abi.TASK_SUCCEEDED
    if p.phase == "Failed":
        return abi.TASK_FAILED
    return abi.TASK_UNKNOWN


def _tolerates(tolerations, taint) -> bool:
    """v1.Toleration.ToleratesTaint (vendor/k8s.io/api/core/v1/toleration.go:37-56)."""
    tk, tv, te = taint
    for key, op, val, eff in tolerations:
        if eff and eff != te:
            continue
        if key and key != tk:
            continue
        op = op or "Equal"
        if op == "Exists":
            return True
        if op == "Equal" and val == tv:
            return True
    return False


def _requirement_matches(labels: Dict[str, str], key: str, op: str, values) -> bool:
    """labels.Requirement.Matches (vendor/k8s.io/apimachinery/pkg/labels/selector.go:194-236) for the node-selector operators."""
    has = key in labels
    if op == "In":
        return has and labels[key] in values
    if op == "NotIn":
        return (not has) or labels[key] not in values
    if op == "Exists":
        return has
    if op == "DoesNotExist":
        return not has
    if op in ("Gt", "Lt"):
        if not has or len(values) != 1:
            return False
        try:
            lv, rv = int(labels[key]), int(list(values)[0])
        except ValueError:
            return False
        return lv > rv if op == "Gt" else lv < rv
    raise ValueError(f"{op!r} is not a valid node selector operator")


def _affinity_count(preferred, node_labels) -> int:
    """CalculateNodeAffinityPriorityMap (vendor/.../priorities/node_affinity.go:34-77): sum of the weights of the preferred
    terms that select the node; a term without match expressions selects nothing (helpers.go:205-208), weight 0 is skipped."""
    labels = dict(node_labels)
    count = 0
    for weight, exprs in preferred:
        if weight == 0 or not exprs:
            continue
        if all(_requirement_matches(labels, k, op, vals) for k, op, vals in exprs):
            count += int(weight)
    return count


def _sanitize_port(hp) -> Tuple[str, str, int]:
    ip, proto, port = hp
    return (ip or "0.0.0.0", proto or "TCP", int(port))


def _ports_conflict(a, b) -> bool:
    """HostPortInfo.CheckConflict (vendor/.../nodeinfo/host_ports.go:107-135) between a wanted and a used (ip, protocol, port)."""
    return a[1] == b[1] and a[2] == b[2] and (a[0] == b[0] or a[0] == "0.0.0.0" or b[0] == "0.0.0.0")


def _requirement_valid(op: str, values) -> bool:
    """labels.NewRequirement's validation (vendor/k8s.io/apimachinery/pkg/labels/selector.go:130-176): a requirement it rejects
    makes NodeSelectorRequirementsAsSelector fail, and MatchNodeSelectorTerms then skips the whole term."""
    if op in ("In", "NotIn"):
        return len(values) > 0
    if op in ("Exists", "DoesNotExist"):
        return len(values) == 0
    if op in ("Gt", "Lt"):
        if len(values) != 1:
            return False
        try:
            int(list(values)[0])
        except ValueError:
            return False
        return True
    return False


def _node_selector_terms_match(terms, labels: Dict[str, str], node_name: str) -> bool:
    """v1helper.MatchNodeSelectorTerms (vendor/k8s.io/kubernetes/pkg/apis/core/v1/helper/helpers.go:285-314) with the node
    fields the scheduler offers (metadata.name, predicates.go:915-922); field requirements take In / NotIn with one value
    (helpers.go:248-273)."""
    for exprs, fields in terms:
        if not exprs and not fields:
            continue
        if exprs:
            if not all(_requirement_valid(op, vals) for _, op, vals in exprs):
                continue
            if not all(_requirement_matches(labels, k, op, vals) for k, op, vals in exprs):
                continue
        if fields:
            if not all(op in ("In", "NotIn") and len(vals) == 1 for _, op, vals in fields):
                continue
            have = {"metadata.name": node_name}
            if not all((have.get(k, "") == list(vals)[0]) == (op == "In") for k, op, vals in fields):
                continue
        return True
    return False


# ------------------------------------------------------------------------------------------------
# inter-pod (anti)affinity: predicate p8 and nodeorder's InterPodAffinityPriority
# ------------------------------------------------------------------------------------------------
IP_MAX = 65534       # inter-pod predicate counters / priority classes per session (include/kb_engine.h: KB_INTERPOD_MAX: the width of task_require)


class UnsupportedSnapshot(ValueError):
    """the session is outside the engine's envelope: the Go action hands the cycle to the stock action"""


def _canon_selector(sel):
    if sel is None:
        return None
    ml, exprs = sel
    ml = tuple(sorted(dict(ml).items())) if not isinstance(ml, tuple) else tuple(sorted(ml))
    exprs = tuple((k, op, tuple(vals)) for k, op, vals in exprs)
    for _, op, vals in exprs:   # metav1.LabelSelectorAsSelector / labels.NewRequirement reject these: the reference then errors per pair
        if op not in ("In", "NotIn", "Exists", "DoesNotExist") or not _requirement_valid(op, vals):
            raise UnsupportedSnapshot(f"pod selector requirement {op!r} {vals!r} is not valid")
    return (ml, exprs)


def _selector_matches(sel, labels: Dict[str, str]) -> bool:
    """metav1.LabelSelectorAsSelector(sel).Matches(labels) (helpers.go:34-70; labels/selector.go:194-236)."""
    if sel is None:
        return False
    ml, exprs = sel
    return all(labels.get(k) == v for k, v in ml) and all(_requirement_matches(labels, k, op, vals) for k, op, vals in exprs)


def _term_props(owner: "Pod", term):
    """(namespaces, selector) of a PodAffinityTerm as its owner resolves them (priorities/util/topologies.go:28-36)."""
    namespaces, sel, key = term
    return (tuple(sorted(set(namespaces))) if namespaces else (owner.namespace,), _canon_selector(sel), key)


def _pod_matches_props(p: "Pod", ns, sel) -> bool:
    """PodMatchesTermsNamespaceAndSelector (priorities/util/topologies.go:40-49)."""
    return p.namespace in ns and _selector_matches(sel, p.labels)


def has_interpod_terms(p: "Pod") -> bool:
    return bool(p.pod_affinity_required or p.pod_affinity_preferred or p.pod_anti_affinity_required or p.pod_anti_affinity_preferred)


def build_interpod(nodes, session_pods, other_pods_on_nodes, task_node, task_status):
    """The inter-pod tables of kb_interpod (include/kb_engine.h) from Kubernetes-shaped objects.

    nodes: sorted by name; session_pods: the session's tasks in canonical task order; task_node[t]: the node whose ni.Tasks holds
    task t at session open (KB_NONE: none); other_pods_on_nodes: [(pod, node index)] for pods outside the session that sit in some
    ni.Tasks.  Returns None when no pod of the cluster carries a pod-(anti)affinity term.

    Predicate (vendor/.../algorithm/predicates/predicates.go:1261-1575 without metadata, i.e. the slow path, over
    plugins/util/util.go:37-90's PodLister = the session's tasks in an allocated status):
      counter kind 'A' (one per distinct required anti-affinity term of any session pod, as resolved by its owner): counts the
        OWNERS in an allocated status per value of the term's topology key.  A pod that matches the term's namespaces + selector
        may not go where that count is positive (satisfiesExistingPodsAntiAffinity, :1400-1441).
      counter kind 'G' (one per distinct SET of required terms a pod carries as its affinity, or as its anti-affinity): counts
        the allocated-status pods that match ALL terms' properties, per tuple of the terms' topology values
        (podMatchesPodAffinityTerms, :1295-1320: the slow path ANDs the terms).  The pod's own affinity needs a positive count in
        the node's domain unless no pod at all matches and the pod matches its own terms (:1550-1565); its own anti-affinity
        needs a zero count (:1535-1543).
    Priority (vendor/.../priorities/interpod_affinity.go:99-235 behind plugins/nodeorder/nodeorder.go:48-62,156-160):
      class kind 'S' (a preferred term of the pod being scored): counts, per node, the pods in ni.Tasks that match it;
      class kind 'O' (a term some pod on a node owns: required affinity with hardPodAffinityWeight 1, preferred (anti)affinity
        with its signed weight): counts its owners per node; the scored pod's weight for it is that weight if it matches.
      count(node i) = sum over classes c of weight_c x (pods counted by c on the FEASIBLE nodes whose "node" shares c's topology
      value with i), where the "node" of a pod whose Spec.NodeName is still empty is the first node (ascending name) that holds
      any such pod (nodeorder.go:48-62)."""
    N = len(nodes)
    all_pods = list(session_pods) + [p for p, _ in other_pods_on_nodes]
    if not any(has_interpod_terms(p) for p in all_pods):
        return None
    # Volume claims.  ssn.Allocate opens with cache.AllocateVolumes (framework/session.go:236-238 -> cache/cache.go:206-211 ->
    # volumeBinder.AssumePodVolumes, vendor/.../controller/volume/scheduling/scheduler_binder.go:253-322).  At this commit nothing in
    # pkg/scheduler calls FindPodVolumes, so podBindingCache holds no decision for the pod: GetBindings / GetProvisionedPVCs return nil
    # (scheduler_binder_cache.go:124-154), both loops are empty and the call returns (false, nil) — or (true, nil) when every claim is
    # bound.  A claim can therefore never veto an Allocate; the one trace it leaves inside the session is `assumedPod.Spec.NodeName =
    # nodeName` (:269) on a pod whose claims are not all bound, and the only reader of that field is nodeorder's cachedNodeInfo
    # (plugins/nodeorder/nodeorder.go:55): with inter-pod terms in the session such a pod would stop counting for node Z.  Whether a
    # claim is bound is not in the snapshot, so that combination — and only it — stays with the stock action.
    for p, st in zip(session_pods, task_status):
        if p.has_volume_claim and st == abi.TASK_PENDING:
            raise UnsupportedSnapshot("pending pod with a PersistentVolumeClaim in a session with inter-pod (anti)affinity terms "
                                      "(AssumePodVolumes sets its Spec.NodeName, which nodeorder's inter-pod priority reads)")
    T = len(session_pods)
    KB_NONE = abi.KB_NONE
    allocated = (abi.TASK_ALLOCATED, abi.TASK_BINDING, abi.TASK_BOUND, abi.TASK_RUNNING)   # api.AllocatedStatus (api/helpers.go:63-72)

    # ---- predicate counters
    counters: Dict[tuple, int] = {}

    def counter(key):
        if key not in counters:
            counters[key] = len(counters)
        return counters[key]

    def group_key(owner, terms):
        return ("G", tuple(sorted((_term_props(owner, t) for t in terms), key=repr)))

    for p in session_pods:
        for t in p.pod_anti_affinity_required:
            ns, sel, key = _term_props(p, t)
            if not key:
                raise UnsupportedSnapshot("required pod anti-affinity term with an empty topologyKey")
            counter(("A", ns, sel, key))
        for terms in (p.pod_affinity_required, p.pod_anti_affinity_required):
            if terms:
                if any(not _term_props(p, t)[2] for t in terms):
                    raise UnsupportedSnapshot("required pod (anti)affinity term with an empty topologyKey")
                counter(group_key(p, terms))
    C = len(counters)
    if C > IP_MAX:
        raise UnsupportedSnapshot(f"more than {IP_MAX} distinct inter-pod predicate counters in one session")
    ckeys = sorted(counters, key=lambda k: counters[k])

    def domain_table(keys_of):
        """per counter/class: node -> interned id of the tuple of its topology values (KB_NONE when a label is missing)"""
        dom = np.full((len(keys_of), N), KB_NONE, np.uint32)
        width = 1
        for c, tk in enumerate(keys_of):
            vals = {}
            for n, node in enumerate(nodes):
                if tk and all(k and k in node.labels for k in tk):
                    vals.setdefault(tuple(node.labels[k] for k in tk), []).append(n)
            for i, v in enumerate(sorted(vals)):
                dom[c, vals[v]] = i
            width = max(width, len(vals))
        return dom, width

    def counter_topology(k):
        return (k[3],) if k[0] == "A" else tuple(t[2] for t in k[1])

    ctr_dom, D = domain_table([counter_topology(k) for k in ckeys])
    Wc, = (max(1, (C + 63) // 64),)           # 64-bit words per task mask: counter c is bit c % 64 of word c // 64
    t_inc = [0] * T; t_forbid = [0] * T        # Python integers here, split into words at the end
    t_req = np.full(T, 0xFFFF, np.uint16); t_self = np.zeros(T, np.uint8)
    for t, p in enumerate(session_pods):
        inc = forbid = 0
        own_a = {("A",) + _term_props(p, term) for term in p.pod_anti_affinity_required}
        for c, k in enumerate(ckeys):
            if k[0] == "A":
                if k in own_a:
                    inc |= 1 << c
                if _pod_matches_props(p, k[1], k[2]):
                    forbid |= 1 << c
            elif all(_pod_matches_props(p, ns, sel) for ns, sel, _ in k[1]):
                inc |= 1 << c
        if p.pod_anti_affinity_required:
            forbid |= 1 << counters[group_key(p, p.pod_anti_affinity_required)]
        if p.pod_affinity_required:
            g = group_key(p, p.pod_affinity_required)
            t_req[t] = counters[g]
            t_self[t] = int(all(_pod_matches_props(p, ns, sel) for ns, sel, _ in g[1]))   # targetPodMatchesAffinityOfPod(pod, pod)
        t_inc[t] = inc; t_forbid[t] = forbid
    ctr_count = np.zeros((C, D), np.int32); ctr_total = np.zeros(C, np.int32)
    for t, p in enumerate(session_pods):
        if int(task_status[t]) not in allocated:
            continue
        if int(task_node[t]) == KB_NONE:
            if t_inc[t]:
                # PodLister lists it under its NodeName although no ni.Tasks holds it: nodeInfo.Filter then hides it from that one
                # node only (predicates.go:1410-1413) -- not modelled
                raise UnsupportedSnapshot("an allocated-status task outside every ni.Tasks takes part in inter-pod affinity")
            continue
        for c in range(C):
            if (t_inc[t] >> c) & 1:
                ctr_total[c] += 1
                d = ctr_dom[c, int(task_node[t])]
                if d != KB_NONE:
                    ctr_count[c, d] += 1

    # ---- priority classes
    classes: Dict[tuple, int] = {}

    def klass(key):
        if key not in classes:
            classes[key] = len(classes)
        return classes[key]

    def owned_terms(p):
        """terms of p that score OTHER pods: required affinity with hardPodAffinityWeight 1, preferred affinity +w, preferred
        anti-affinity -w (interpod_affinity.go:163-196); identical terms of one pod add up"""
        acc: Dict[tuple, int] = {}
        for term in p.pod_affinity_required:
            k = _term_props(p, term); acc[k] = acc.get(k, 0) + 1
        for w, term in p.pod_affinity_preferred:
            k = _term_props(p, term); acc[k] = acc.get(k, 0) + int(w)
        for w, term in p.pod_anti_affinity_preferred:
            k = _term_props(p, term); acc[k] = acc.get(k, 0) - int(w)
        return {k: w for k, w in acc.items() if w != 0}

    def subject_terms(p):
        """preferred terms of p scoring p itself against the pods on the nodes (interpod_affinity.go:137-160)"""
        acc: Dict[tuple, int] = {}
        for w, term in p.pod_affinity_preferred:
            k = _term_props(p, term); acc[k] = acc.get(k, 0) + int(w)
        for w, term in p.pod_anti_affinity_preferred:
            k = _term_props(p, term); acc[k] = acc.get(k, 0) - int(w)
        return {k: w for k, w in acc.items() if w != 0}

    for p in all_pods:
        for k, w in owned_terms(p).items():
            klass(("O",) + k + (w,))
    for p in session_pods:
        for k in subject_terms(p):
            klass(("S",) + k)
    P = len(classes)
    if P > IP_MAX:
        raise UnsupportedSnapshot(f"more than {IP_MAX} distinct inter-pod priority classes in one session")
    pkeys = sorted(classes, key=lambda k: classes[k])
    cls_dom, _ = domain_table([(k[3],) for k in pkeys])

    def cls_inc_mask(p):
        m = 0
        own = owned_terms(p)
        for c, k in enumerate(pkeys):
            if k[0] == "O":
                if own.get(k[1:4]) == k[4]:
                    m |= 1 << c
            elif _pod_matches_props(p, k[1], k[2]):
                m |= 1 << c
        return m

    Wp = max(1, (P + 63) // 64)
    t_cinc = [0] * T; t_sig = np.full(T, KB_NONE, np.uint32)
    sigs: Dict[tuple, int] = {}
    for t, p in enumerate(session_pods):
        t_cinc[t] = cls_inc_mask(p)
        sub = subject_terms(p)
        w = [0] * P
        for c, k in enumerate(pkeys):
            if k[0] == "O":
                if _pod_matches_props(p, k[1], k[2]):
                    w[c] = k[4]
            else:
                w[c] = sub.get(k[1:4], 0)
        if any(w):
            t_sig[t] = sigs.setdefault(tuple(w), len(sigs))
    S = max(1, len(sigs))
    sig_w = np.zeros((S, max(1, P)), np.int32)
    for w, i in sigs.items():
        sig_w[i, :P] = w
    cls_bound = np.zeros((max(1, P), N), np.int32); cls_unbound = np.zeros((max(1, P), N), np.int32)
    first_unbound = KB_NONE
    on_nodes = [(p, int(task_node[t]), t_cinc[t]) for t, p in enumerate(session_pods) if int(task_node[t]) != KB_NONE]
    on_nodes += [(p, n, cls_inc_mask(p)) for p, n in other_pods_on_nodes]
    for p, n, m in on_nodes:
        if p.spec_node_name_empty:
            first_unbound = min(first_unbound, n)
        tab = cls_unbound if p.spec_node_name_empty else cls_bound
        for c in range(P):
            if (m >> c) & 1:
                tab[c, n] += 1
    def words(masks, W):
        out = np.zeros((max(1, len(masks)), W), np.uint64)
        for i, m in enumerate(masks):
            for w in range(W):
                out[i, w] = (m >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
        return out

    t_inc, t_forbid, t_cinc = words(t_inc, Wc), words(t_forbid, Wc), words(t_cinc, Wp)
    return dict(n_counters=C, n_domains=int(D), n_classes=P, n_sigs=len(sigs), first_unbound_node=int(first_unbound),
                ctr_dom=ctr_dom.reshape(C, N) if C else np.zeros((1, N), np.uint32) + KB_NONE,
                ctr_count=ctr_count if C else np.zeros((1, max(1, D)), np.int32), ctr_total=ctr_total if C else np.zeros(1, np.int32),
                task_inc=t_inc, task_forbid=t_forbid, task_require=t_req, task_self=t_self,
                cls_dom=cls_dom if P else np.zeros((1, N), np.uint32) + KB_NONE, cls_bound=cls_bound, cls_unbound=cls_unbound,
                task_cls_inc=t_cinc, task_sig=t_sig, sig_weight=sig_w)


def _best_effort_qos(p: "Pod") -> bool:
    """v1qos.GetPodQOS(pod) == BestEffort (vendor/k8s.io/kubernetes/pkg/apis/core/v1/helper/qos/qos.go:37-82): no container or init
    container has a cpu or memory request or limit above zero."""
    for c in list(p.containers) + list(p.init_containers) + list(p.limits):
        for name in ("cpu", "memory"):
            if name in c and parse_quantity(c[name]) > 0:
                return False
    return True


def _static_ok(pod_cls, node_cls, pressure=(False, False, False)) -> bool:
    """p2 CheckNodeCondition, p3 CheckNodeUnschedulable, p4 PodMatchNodeSelector (nodeSelector and required node affinity),
    p6 PodToleratesNodeTaints — vendor/.../algorithm/predicates/predicates.go:1675-1700,1576-1593,927-983,1596-1620."""
    selector, tolerations = pod_cls[0], pod_cls[1]
    required = pod_cls[3] if len(pod_cls) > 3 else (0,)
    labels, taints, unsched, ready, netun = node_cls[:5]
    node_name = node_cls[5] if len(node_cls) > 5 else ""
    if (not ready) or netun or unsched:
        return False
    labels = dict(labels)
    for k, v in selector:
        if labels.get(k) != v:
            return False
    if required[0] and not _node_selector_terms_match(required[1], labels, node_name):
        return False
    for t in taints:
        if t[2] in ("NoSchedule", "NoExecute") and not _tolerates(tolerations, t):
            return False
    # optional checks (vendor/.../predicates/predicates.go:1633-1672): memory pressure only turns BestEffort pods away
    mem_p, disk_p, pid_p = node_cls[6] if len(node_cls) > 6 else (False, False, False)
    best_effort = pod_cls[4] if len(pod_cls) > 4 else False
    if pressure[0] and mem_p and best_effort:
        return False
    if (pressure[1] and disk_p) or (pressure[2] and pid_p):
        return False
    return True


# ------------------------------------------------------------------------------------------------
# the SoA snapshot
# ------------------------------------------------------------------------------------------------
_DTYPES = {C.c_double: np.float64, C.c_uint32: np.uint32, C.c_int64: np.int64, C.c_int32: np.int32, C.c_uint8: np.uint8,
           C.c_uint64: np.uint64, C.c_uint16: np.uint16}


class SessionSnapshot:
    """Numpy-backed kb_snapshot.  Field names equal the C struct's."""

    def __init__(self, **kw):
        self.names: Dict[str, List[str]] = kw.pop("names", {})
        for k, v in kw.items():
            setattr(self, k, v)
        self._check()

    @property
    def R(self):
        return int(self.n_res)

    def _check(self):
        R, N, T, J, Q = self.n_res, self.n_nodes, self.n_tasks, self.n_jobs, self.n_queues
        shapes = {"node_idle": (R, N), "node_releasing": (R, N), "node_allocatable": (R, N),
                  "task_resreq": (R, T), "task_init_resreq": (R, T), "job_task_begin": (J + 1,)}
        for name, ctype in abi.SNAPSHOT_ARRAYS:
            a = getattr(self, name, None)
            if a is None:
                if name in ("class_compat", "class_affinity", "node_ports", "task_port_want", "task_port_conflict",
                            "task_evict_protected"):
                    continue
                raise ValueError(f"snapshot field {name} missing")
            a = np.ascontiguousarray(a, dtype=_DTYPES[ctype])
            if name in shapes and a.shape != shapes[name]:
                raise ValueError(f"{name}: shape {a.shape} != {shapes[name]}")
            setattr(self, name, a)
        for name in ("node_scalar_mask", "node_alloc_cpu", "node_alloc_mem", "node_nz_cpu", "node_nz_mem",
                     "node_max_pods", "node_pod_cnt", "node_class"):
            assert getattr(self, name).shape == (N,), name
        for name in ("task_scalar_mask", "task_nz_cpu", "task_nz_mem", "task_job", "task_class", "task_priority",
                     "task_creation", "task_status", "task_node"):
            assert getattr(self, name).shape == (T,), name
        for name in ("job_queue", "job_min_available", "job_priority", "job_creation"):
            assert getattr(self, name).shape == (J,), name
        for name in ("queue_weight", "queue_creation"):
            assert getattr(self, name).shape == (Q,), name
        # host-port masks: [n] (one 64-bit word) or [n][Wh]; all three of one width
        widths = set()
        for name, n in (("node_ports", N), ("task_port_want", T), ("task_port_conflict", T)):
            a = getattr(self, name, None)
            if a is None:
                continue
            if a.ndim == 1:
                a = a.reshape(n, 1) if getattr(self, "port_words", 1) in (0, 1) else a.reshape(n, int(self.port_words))
            assert a.ndim == 2 and a.shape[0] == n, name
            widths.add(int(a.shape[1]))
            setattr(self, name, np.ascontiguousarray(a) if a.shape[1] > 1 else np.ascontiguousarray(a.reshape(n)))
        if len(widths) > 1:
            raise ValueError(f"host-port masks of different widths: {sorted(widths)}")
        self.port_words = widths.pop() if widths else 1

    def to_abi(self) -> abi.Snapshot:
        s = abi.Snapshot()
        s.version = abi.KB_ABI_VERSION
        for k in ("n_res", "n_nodes", "n_tasks", "n_jobs", "n_queues", "n_task_classes", "n_node_classes"):
            setattr(s, k, int(getattr(self, k)))
        for name, ctype in abi.SNAPSHOT_ARRAYS:
            a = getattr(self, name, None)
            if a is None:
                setattr(s, name, C.POINTER(ctype)())
            else:
                setattr(s, name, a.ctypes.data_as(C.POINTER(ctype)))
        s.port_words = int(getattr(self, "port_words", 1))
        ip = getattr(self, "interpod", None)
        if ip is None:
            s.interpod = C.POINTER(abi.Interpod)()
        else:
            k = abi.Interpod()
            for name in ("n_counters", "n_domains", "n_classes", "n_sigs", "first_unbound_node"):
                setattr(k, name, int(ip[name]))
            for name, ctype in abi.INTERPOD_ARRAYS:
                a = np.ascontiguousarray(ip[name], dtype=_DTYPES[ctype])
                ip[name] = a                      # keep the buffer alive as long as the snapshot
                setattr(k, name, a.ctypes.data_as(C.POINTER(ctype)))
            self._interpod_abi = k                 # and the struct itself
            s.interpod = C.pointer(k)
        return s

    def task_name(self, t: int) -> str:
        n = self.names.get("tasks")
        return n[t] if n else f"t{t}"

    def node_name(self, n: int) -> str:
        nn = self.names.get("nodes")
        return nn[n] if nn else f"n{n:06d}"

    def bind_map(self, task_node: np.ndarray) -> Dict[str, str]:
        """{pod ns/name: node} — the object actions/allocate/allocate_test.go:208 compares."""
        return {self.task_name(int(t)): self.node_name(int(task_node[t])) for t in np.nonzero(task_node != abi.KB_NONE)[0]}


def _resource(rl: Dict[str, str], dims: Dict[str, int], R: int):
    """api.NewResource (resource_info.go:73-90): returns (vector[R], scalar key mask, MaxTaskNum)."""
    v = np.zeros(R, dtype=np.float64)
    mask = 0
    max_tasks = 0
    for name, q in rl.items():
        if name == "cpu":
            v[0] += float(quantity_milli_value(q))
        elif name == "memory":
            v[1] += float(quantity_value(q))
        elif name == "pods":
            max_tasks += quantity_value(q)
        elif is_scalar_resource_name(name):
            d = dims[name]
            v[d] += float(quantity_milli_value(q))
            mask |= 1 << (d - 2)
    return v, mask, max_tasks


def flatten(nodes: List[Node], pods: List[Pod], pod_groups: List[PodGroup], queues: List[Queue],
            default_queue: str = "default", pressure: Tuple[bool, bool, bool] = (False, False, False),
            prune_ports: bool = True) -> SessionSnapshot:
    """Kubernetes-shaped objects -> canonical SoA snapshot (what cache.Snapshot() + the Go shim's flatten produce).
    prune_ports=False interns every host port of the cluster (tests: the pruned and the unpruned table must decide alike).
    pressure = the predicates plugin's (MemoryPressureEnable, DiskPressureEnable, PIDPressureEnable) arguments
    (SchedulerConf.pressure_flags()): static per class pair, so they are folded into class_compat here and the engine never
    sees them."""
    pressure = tuple(bool(x) for x in pressure)
    scalar_names = set()
    for n in nodes:
        scalar_names.update(k for k in n.allocatable if is_scalar_resource_name(k))
    for p in pods:
        for c in list(p.containers) + list(p.init_containers):
            scalar_names.update(k for k in c if is_scalar_resource_name(k))
    dims = {name: 2 + i for i, name in enumerate(sorted(scalar_names))}
    R = 2 + len(dims)
    if R > abi.KB_MAX_RES:
        raise ValueError("too many scalar resources")

    nodes = sorted(nodes, key=lambda n: n.name)
    queues = sorted(queues, key=lambda q: q.name)
    qidx = {q.name: i for i, q in enumerate(queues)}
    nidx = {n.name: i for i, n in enumerate(nodes)}
    N, Q = len(nodes), len(queues)

    # jobs: PodGroup job id "<ns>/<name>" (cache/event_handlers.go:367-369); pods join by the
    # scheduling.k8s.io/group-name annotation (api/job_info.go:56-66); jobs without a PodGroup or whose
    # queue does not exist are skipped by Snapshot (cache.go:645-657)
    pgs = {}
    for pg in pod_groups:
        qname = pg.queue or default_queue
        if qname not in qidx:
            continue
        pgs[f"{pg.namespace}/{pg.name}"] = pg
    # shadow PodGroups (Pod.shadow_job): the job id is the bare UID, the group carries the first such pod's creation stamp, the default queue
    for p in pods:
        if not p.group_name and p.shadow_job and p.shadow_job not in pgs and default_queue in qidx:
            pgs[p.shadow_job] = PodGroup(p.namespace, p.shadow_job, min_member=p.shadow_min_member, queue=default_queue, creation=p.creation,
                                         priority=p.priority or 0)   # Spec.PriorityClassName = the pod's (cache/util.go:89): its resolved value
    job_ids = sorted(pgs)
    jidx = {j: i for i, j in enumerate(job_ids)}
    J = len(job_ids)

    job_tasks: List[List[Pod]] = [[] for _ in range(J)]
    other_pods: List[Pod] = []
    for p in pods:
        if p.uid is None:
            p.uid = f"{p.namespace}-{p.name}"          # util.BuildPod (pkg/scheduler/util/test_utils.go:66-69)
        jid = f"{p.namespace}/{p.group_name}" if p.group_name else p.shadow_job
        if jid in jidx:
            job_tasks[jidx[jid]].append(p)
        else:
            other_pods.append(p)
    for lst in job_tasks:
        lst.sort(key=lambda p: p.uid)

    T = sum(len(l) for l in job_tasks)
    node_idle = np.zeros((R, N)); node_rel = np.zeros((R, N)); node_alloc = np.zeros((R, N))
    node_mask = np.zeros(N, np.uint32); node_maxp = np.zeros(N, np.int32); node_cnt = np.zeros(N, np.int32)
    node_acpu = np.zeros(N, np.int64); node_amem = np.zeros(N, np.int64)
    node_nzc = np.zeros(N, np.int64); node_nzm = np.zeros(N, np.int64)
    node_cls_keys = []
    # a pod that selects nodes by field (metadata.name) makes the node name part of the static class
    by_name = any(fields for p in pods for _, fields in (p.required_affinity or []))
    for i, n in enumerate(nodes):
        v, m, mt = _resource(n.allocatable, dims, R)
        node_alloc[:, i] = v; node_idle[:, i] = v; node_mask[i] = m; node_maxp[i] = mt
        node_acpu[i] = quantity_milli_value(n.allocatable.get("cpu", 0))
        node_amem[i] = quantity_value(n.allocatable.get("memory", 0))
        node_cls_keys.append((tuple(sorted(n.labels.items())), tuple(n.taints), n.unschedulable, n.ready, n.network_unavailable,
                              n.name if by_name else "",
                              (n.memory_pressure and pressure[0], n.disk_pressure and pressure[1], n.pid_pressure and pressure[2])))

    def pod_vectors(p: Pod):
        res = np.zeros(R); mask = 0
        nzc = nzm = 0
        for c in p.containers:                           # GetPodResourceWithoutInitContainers (pod_info.go:66-73)
            v, m, _ = _resource(c, dims, R)
            res += v; mask |= m
            nzc += quantity_milli_value(c["cpu"]) if "cpu" in c else DEFAULT_MILLI_CPU_REQUEST
            nzm += quantity_value(c["memory"]) if "memory" in c else DEFAULT_MEMORY_REQUEST
        init = res.copy()
        for c in p.init_containers:                      # GetPodResourceRequest: SetMaxResource per init container
            v, m, _ = _resource(c, dims, R)
            init = np.maximum(init, v)
        return res, init, mask, nzc, nzm

    eps = np.array([10.0, 10.0 * 1024 * 1024] + [10.0] * (R - 2))

    def less_equal(l, r):
        """Resource.LessEqual on dense vectors (resource_info.go:268-302; an absent key reads 0, scalars <= 10 skipped)."""
        ok = (l < r) | (np.abs(l - r) < eps)
        ok[2:] |= l[2:] <= 10.0
        return bool(ok.all())

    def account_on_node(ni, st, res, nzc, nzm):
        """NodeInfo.AddTask (api/node_info.go:172-212) + k8s nodeinfo.AddPod for a pod already on a node.
        Returns False where AddTask errors ("Selected node NotReady": the pod does not enter ni.Tasks)."""
        if st == abi.TASK_PIPELINED:
            node_rel[:, ni] -= res
        else:
            if not less_equal(res, node_idle[:, ni]):
                return False
            node_idle[:, ni] -= res
            if st == abi.TASK_RELEASING:
                node_rel[:, ni] += res
        node_cnt[ni] += 1
        node_nzc[ni] += nzc; node_nzm[ni] += nzm
        return True

    # host ports: a distinct (ip, protocol, port > 0) of the cluster's pods is one bit — if it can ever take part in a conflict test.
    # PodFitsHostPorts is only ever asked for a Pending task (allocate, backfill, the preemptors of preempt / reclaim), and a placement
    # adds a Pending task's own ports: a triple that conflicts with no Pending task's port (what the daemons already running on the nodes
    # listen on, usually) can never decide anything and gets no bit.  Any number of them: masks of Wh = ceil(n / 64) words.  The engine
    # decides a Pending pod whose masks reach beyond word 0 in a device round of its own (include/kb_engine.h), so the triples the Pending
    # pods' ports conflict with most often take the low bits.
    every = sorted({_sanitize_port(hp) for p in pods for hp in p.host_ports if int(hp[2]) > 0})
    asked = sorted({_sanitize_port(hp) for p in pods if _task_status(p) == abi.TASK_PENDING for hp in p.host_ports if int(hp[2]) > 0})
    universe = [u for u in every if any(_ports_conflict(hp, u) for hp in asked)] if prune_ports else every
    if len(universe) > 64:
        pending = [[_sanitize_port(hp) for hp in p.host_ports if int(hp[2]) > 0] for p in pods if _task_status(p) == abi.TASK_PENDING]
        pending = [m for m in pending if m]
        weight = {u: sum(1 for m in pending if any(_ports_conflict(hp, u) for hp in m)) for u in universe}
        universe = sorted(universe, key=lambda u: (-weight[u], u))
    port_bit = {hp: i for i, hp in enumerate(universe)}
    Wh = max(1, (len(universe) + 63) // 64)
    node_ports = np.zeros((N, Wh), np.uint64)

    def mask_words(m: int):
        return np.array([(m >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(Wh)], np.uint64)

    def port_masks(p: Pod):
        mine = [_sanitize_port(hp) for hp in p.host_ports if int(hp[2]) > 0]
        want = 0
        conflict = 0
        for hp in mine:
            if hp in port_bit:
                want |= 1 << port_bit[hp]
        for other in universe:
            if any(_ports_conflict(hp, other) for hp in mine):
                conflict |= 1 << port_bit[other]
        return want, conflict

    others_on_nodes = []
    # cache/event_handlers.go:72-92 addTask: a pod with a NodeName enters that node's Tasks unless it is terminated (Succeeded / Failed):
    # a finished pod keeps its Spec.NodeName but holds nothing on the node — no Idle, no pod slot, no port (found by tests/objref_fit.py,
    # round 3: every terminated pod used to be accounted on its old node)
    terminated = (abi.TASK_SUCCEEDED, abi.TASK_FAILED)
    for p in other_pods:                                 # pods of other schedulers / jobs outside the session
        if p.node_name in nidx and _task_status(p) not in terminated:
            res, _, _, nzc, nzm = pod_vectors(p)
            if account_on_node(nidx[p.node_name], _task_status(p), res, nzc, nzm):
                node_ports[nidx[p.node_name]] |= mask_words(port_masks(p)[0])
                others_on_nodes.append((p, nidx[p.node_name]))

    t_res = np.zeros((R, T)); t_init = np.zeros((R, T)); t_mask = np.zeros(T, np.uint32)
    t_nzc = np.zeros(T, np.int64); t_nzm = np.zeros(T, np.int64); t_job = np.zeros(T, np.uint32)
    t_prio = np.zeros(T, np.int32); t_cre = np.zeros(T, np.int64); t_st = np.zeros(T, np.uint8)
    t_node = np.full(T, abi.KB_NONE, np.uint32)
    t_want = np.zeros((T, Wh), np.uint64); t_conf = np.zeros((T, Wh), np.uint64)
    t_prot = np.zeros(T, np.uint8)
    task_cls_keys = []
    names_tasks = []
    begin = np.zeros(J + 1, np.uint32)
    k = 0
    for j, lst in enumerate(job_tasks):
        begin[j] = k
        for p in lst:
            res, init, mask, nzc, nzm = pod_vectors(p)
            st = _task_status(p)
            t_res[:, k] = res; t_init[:, k] = init; t_mask[k] = mask
            t_nzc[k] = nzc; t_nzm[k] = nzm; t_job[k] = j
            t_prio[k] = 1 if p.priority is None else p.priority      # NewTaskInfo default (job_info.go:82)
            t_cre[k] = p.creation; t_st[k] = st
            want, conflict = port_masks(p)
            t_want[k] = mask_words(want); t_conf[k] = mask_words(conflict)
            # plugins/conformance/conformance.go:44-58
            t_prot[k] = int(p.namespace == "kube-system" or p.priority_class_name in ("system-cluster-critical", "system-node-critical"))
            if p.node_name in nidx and st not in terminated:
                if account_on_node(nidx[p.node_name], st, res, nzc, nzm):
                    t_node[k] = nidx[p.node_name]
                    node_ports[nidx[p.node_name]] |= mask_words(want)
            pref = tuple((int(w), tuple((k2, op, tuple(vals)) for k2, op, vals in exprs)) for w, exprs in p.preferred_affinity)
            req = (0,) if p.required_affinity is None else (1, tuple(
                (tuple((k2, op, tuple(vals)) for k2, op, vals in exprs), tuple((k2, op, tuple(vals)) for k2, op, vals in fields))
                for exprs, fields in p.required_affinity))
            task_cls_keys.append((tuple(sorted(p.node_selector.items())), tuple(p.tolerations), pref, req,
                                  pressure[0] and _best_effort_qos(p)))
            names_tasks.append(f"{p.namespace}/{p.name}")
            k += 1
    begin[J] = k

    interpod = build_interpod(nodes, [p for lst in job_tasks for p in lst], others_on_nodes, t_node, t_st)

    ucls_t = sorted(set(task_cls_keys)) or [((), (), (), (0,), False)]
    ucls_n = sorted(set(node_cls_keys)) or [((), (), False, True, False, "", (False, False, False))]
    tmap = {c: i for i, c in enumerate(ucls_t)}
    nmap = {c: i for i, c in enumerate(ucls_n)}
    compat = np.zeros((len(ucls_t) * len(ucls_n) + 7) // 8, np.uint8)
    for ti, tc in enumerate(ucls_t):
        for ni, nc in enumerate(ucls_n):
            if _static_ok(tc, nc, pressure):
                b = ti * len(ucls_n) + ni
                compat[b >> 3] |= 1 << (b & 7)
    affinity = None
    if any(tc[2] for tc in ucls_t):
        affinity = np.array([[_affinity_count(tc[2], nc[0]) for nc in ucls_n] for tc in ucls_t], np.int32)

    return SessionSnapshot(
        n_res=R, n_nodes=N, n_tasks=T, n_jobs=J, n_queues=Q,
        n_task_classes=len(ucls_t), n_node_classes=len(ucls_n),
        node_idle=node_idle, node_releasing=node_rel, node_allocatable=node_alloc, node_scalar_mask=node_mask,
        node_alloc_cpu=node_acpu, node_alloc_mem=node_amem, node_nz_cpu=node_nzc, node_nz_mem=node_nzm,
        node_max_pods=node_maxp, node_pod_cnt=node_cnt,
        node_class=np.array([nmap[c] for c in node_cls_keys], np.uint32).reshape(N),
        task_resreq=t_res, task_init_resreq=t_init, task_scalar_mask=t_mask, task_nz_cpu=t_nzc, task_nz_mem=t_nzm,
        task_job=t_job, task_class=np.array([tmap[c] for c in task_cls_keys], np.uint32).reshape(T),
        task_priority=t_prio, task_creation=t_cre, task_status=t_st, task_node=t_node,
        job_task_begin=begin,
        job_queue=np.array([qidx[pgs[j].queue or default_queue] for j in job_ids], np.uint32).reshape(J),
        job_min_available=np.array([pgs[j].min_member for j in job_ids], np.int32).reshape(J),
        job_priority=np.array([pgs[j].priority for j in job_ids], np.int32).reshape(J),
        job_creation=np.array([pgs[j].creation for j in job_ids], np.int64).reshape(J),
        queue_weight=np.array([q.weight for q in queues], np.int32).reshape(Q),
        queue_creation=np.array([q.creation for q in queues], np.int64).reshape(Q),
        class_compat=compat, class_affinity=affinity,
        node_ports=node_ports if universe else None, task_port_want=t_want if universe else None,
        task_port_conflict=t_conf if universe else None, task_evict_protected=t_prot if t_prot.any() else None,
        interpod=interpod,
        names={"nodes": [n.name for n in nodes], "tasks": names_tasks, "jobs": job_ids,
               "queues": [q.name for q in queues], "dims": ["cpu", "memory"] + sorted(scalar_names)},
    )


# ------------------------------------------------------------------------------------------------
# deterministic synthetic clusters: counter-based splitmix64, no global RNG state
# ------------------------------------------------------------------------------------------------
SEED_BASE = 0x6B756265


def _mix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


class _Stream:
    """u[i] = splitmix64(seed, stream, i): reproducible on any machine, order-independent."""

    def __init__(self, seed: int, stream: int):
        self.key = _mix(np.array([seed ^ (stream * 0xD1342543DE82EF95 & 0xFFFFFFFFFFFFFFFF)], np.uint64))[0]

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            return _mix(np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + self.key)

    def randint(self, n: int, hi: int) -> np.ndarray:
        return (self.u64(n) % np.uint64(hi)).astype(np.int64)

    def uniform(self, n: int) -> np.ndarray:
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)

    def choice(self, n: int, values, probs=None) -> np.ndarray:
        values = np.asarray(values)
        if probs is None:
            return values[self.randint(n, len(values))]
        cdf = np.cumsum(np.asarray(probs, np.float64))
        cdf /= cdf[-1]
        return values[np.searchsorted(cdf, self.uniform(n), side="right").clip(0, len(values) - 1)]


@dataclass
class SynthParams:
    n_tasks: int
    n_nodes: int
    n_queues: int = 1
    n_res: int = 2
    seed: int = SEED_BASE
    gang_sizes: Tuple[int, ...] = (1, 2, 4, 8, 16, 32, 64)
    gang_probs: Tuple[float, ...] = (0.15, 0.20, 0.20, 0.20, 0.13, 0.08, 0.04)   # mean ~10 tasks per job
    node_cpu_cores: Tuple[int, ...] = (4, 8, 16, 24, 32)          # sized so total demand ~1.3x capacity
    node_mem_gib: Tuple[int, ...] = (16, 32, 64, 128)
    task_cpu_milli: Tuple[int, ...] = (100, 250, 500, 1000, 2000, 4000, 8000)
    task_mem_mib: Tuple[int, ...] = (128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768)
    preload_node_frac: float = 0.10      # nodes carrying 1-8 pods of jobs outside the session
    running_job_frac: float = 0.05       # session jobs with some tasks already Running
    best_effort_frac: float = 0.02
    no_mem_key_frac: float = 0.05
    scalar_job_frac: float = 0.30        # C4: jobs requesting 1-2 extended resources
    n_zones: int = 8
    zone_selector_frac: float = 0.10     # jobs pinned to one zone by nodeSelector (static predicate classes)
    diverse_requests: bool = False       # stress variant: every job draws its OWN request (1 milli-cpu / 1 MiB steps), so the
                                         # session holds about as many distinct task shapes as jobs instead of ~70


def synth_config(idx: int, scale: float = 1.0) -> SynthParams:
    """The BASELINE.json configurations 2..5 (1 is the example/job.yaml fixture)."""
    def sc(x):
        return max(1, int(round(x * scale)))
    if idx == 2:
        return SynthParams(n_tasks=sc(10_000), n_nodes=sc(1_000), n_queues=1, n_res=2, seed=SEED_BASE + 2)
    if idx == 3:
        return SynthParams(n_tasks=sc(100_000), n_nodes=sc(10_000), n_queues=128, n_res=2, seed=SEED_BASE + 3)
    if idx == 4:
        return SynthParams(n_tasks=sc(100_000), n_nodes=sc(10_000), n_queues=128, n_res=16, seed=SEED_BASE + 4)
    if idx == 5:
        return SynthParams(n_tasks=sc(1_000_000), n_nodes=sc(50_000), n_queues=128, n_res=2,
                           seed=SEED_BASE + 5, preload_node_frac=0.6)
    raise ValueError("config index must be 2..5")


def synth(p: SynthParams) -> SessionSnapshot:
    R, N, Q = p.n_res, p.n_nodes, p.n_queues
    S = lambda k: _Stream(p.seed, k)

    # ---- nodes
    cpu_m = S(1).choice(N, np.array(p.node_cpu_cores, np.int64)) * 1000
    mem_b = S(2).choice(N, np.array(p.node_mem_gib, np.int64)) * (1 << 30)
    node_alloc = np.zeros((R, N)); node_alloc[0] = cpu_m; node_alloc[1] = mem_b
    node_mask = np.zeros(N, np.uint32)
    if R > 2:
        for d in range(2, R):
            units = S(100 + d).choice(N, np.array([0, 1, 2, 4, 8], np.int64))
            has = S(200 + d).uniform(N) < 0.6          # the node advertises the extended resource at all
            node_alloc[d] = np.where(has, units * 1000, 0)
            node_mask |= (has.astype(np.uint32) << np.uint32(d - 2))
    node_zone = S(3).randint(N, p.n_zones).astype(np.uint32)
    node_idle = node_alloc.copy()
    node_rel = np.zeros((R, N))
    node_nzc = np.zeros(N, np.int64); node_nzm = np.zeros(N, np.int64); node_cnt = np.zeros(N, np.int32)

    # pods of other tenants already on some nodes (only node aggregates see them)
    pre = S(4).uniform(N) < p.preload_node_frac
    npods = np.where(pre, 1 + S(5).randint(N, 8), 0)
    for k in range(8):
        on = npods > k
        c = S(10 + k).choice(N, np.array([100, 250, 500, 1000], np.int64))
        m = S(20 + k).choice(N, np.array([128, 256, 512, 1024], np.int64)) * (1 << 20)
        fits = on & (node_idle[0] - c >= 0) & (node_idle[1] - m >= 0)
        node_idle[0] -= np.where(fits, c, 0); node_idle[1] -= np.where(fits, m, 0)
        node_nzc += np.where(fits, c, 0); node_nzm += np.where(fits, m, 0); node_cnt += fits.astype(np.int32)

    # ---- jobs: draw gang sizes until the task budget is met
    est_jobs = max(int(p.n_tasks / 8) + 64, -(-p.n_tasks // min(p.gang_sizes)) + 1)   # (draws are indexed: asking for more leaves the first ones as they were)
    sizes = S(30).choice(est_jobs, np.array(p.gang_sizes, np.int64), p.gang_probs)
    csum = np.cumsum(sizes)
    J = int(np.searchsorted(csum, p.n_tasks, side="left")) + 1
    sizes = sizes[:J].copy()
    sizes[-1] -= int(csum[J - 1] - p.n_tasks)
    assert sizes.sum() == p.n_tasks and sizes.min() >= 1
    T = p.n_tasks
    begin = np.zeros(J + 1, np.uint32); begin[1:] = np.cumsum(sizes)
    full = S(31).uniform(J) < 0.7
    min_avail = np.where(full, sizes, (sizes + 1) // 2).astype(np.int32)
    job_queue = S(32).randint(J, Q).astype(np.uint32)
    job_prio = S(33).choice(J, np.array([0, 100, 1000], np.int32)).astype(np.int32)
    job_creation = (1_600_000_000 + np.arange(J, dtype=np.int64))      # strictly increasing seconds
    j_cpu = S(34).choice(J, np.array(p.task_cpu_milli, np.int64))
    j_mem = S(35).choice(J, np.array(p.task_mem_mib, np.int64)) * (1 << 20)
    if p.diverse_requests:               # same ranges, log-uniform-ish, but practically never two jobs alike
        j_cpu = (100 * np.exp2(S(34).uniform(J) * np.log2(80.0))).astype(np.int64)               # 100m .. 8000m
        j_mem = (128 * np.exp2(S(35).uniform(J) * 8.0)).astype(np.int64) * (1 << 20)             # 128Mi .. 32Gi
    j_nomem = S(36).uniform(J) < p.no_mem_key_frac
    j_be = S(37).uniform(J) < p.best_effort_frac
    j_zone_sel = S(38).uniform(J) < p.zone_selector_frac
    j_zone = S(39).randint(J, p.n_zones)

    task_job = np.repeat(np.arange(J, dtype=np.uint32), sizes)
    t_res = np.zeros((R, T))
    cpu_t = j_cpu[task_job].astype(np.float64); mem_t = np.where(j_nomem, 0, j_mem)[task_job].astype(np.float64)
    be_t = j_be[task_job]
    t_res[0] = np.where(be_t, 0.0, cpu_t); t_res[1] = np.where(be_t, 0.0, mem_t)
    t_mask = np.zeros(T, np.uint32)
    if R > 2:
        j_scalar = S(40).uniform(J) < p.scalar_job_frac
        d1 = 2 + S(41).randint(J, R - 2); d2 = 2 + S(42).randint(J, R - 2)
        two = S(43).uniform(J) < 0.5
        units = S(44).choice(J, np.array([1, 1, 2, 4], np.int64)) * 1000
        for d in range(2, R):
            hit = j_scalar & ((d1 == d) | (two & (d2 == d))) & ~j_be
            t_res[d] = np.where(hit, units, 0)[task_job].astype(np.float64)
            t_mask |= (hit[task_job].astype(np.uint32) << np.uint32(d - 2))
    t_init = t_res.copy()
    # non-zero requests: one container; cpu key always present unless BestEffort, mem key absent in a 5 % slice
    t_nzc = np.where(be_t, DEFAULT_MILLI_CPU_REQUEST, j_cpu[task_job]).astype(np.int64)
    t_nzm = np.where(be_t | j_nomem[task_job], DEFAULT_MEMORY_REQUEST, j_mem[task_job]).astype(np.int64)
    t_prio = np.ones(T, np.int32)
    t_cre = job_creation[task_job] + 1
    t_status = np.zeros(T, np.uint8)
    t_node = np.full(T, abi.KB_NONE, np.uint32)

    # some session jobs already have Running tasks (feeds drf/proportion/gang initial state)
    run_jobs = np.nonzero((S(45).uniform(J) < p.running_job_frac) & ~j_be)[0]
    if len(run_jobs):
        pick = S(46).randint(len(run_jobs), 1 << 30)
        cursor = 0
        for jj, j in enumerate(run_jobs):
            k = 1 + int(pick[jj] % max(1, int(sizes[j])))
            k = min(k, int(sizes[j]))
            for t in range(int(begin[j]), int(begin[j]) + k):
                for _ in range(8):                      # first node (deterministic probe sequence) with room
                    n = int((int(pick[jj]) + cursor * 7919) % N); cursor += 1
                    if np.all(node_idle[:, n] - t_res[:, t] >= 0) and node_cnt[n] < 100:
                        node_idle[:, n] -= t_res[:, t]
                        node_nzc[n] += t_nzc[t]; node_nzm[n] += t_nzm[t]; node_cnt[n] += 1
                        t_status[t] = abi.TASK_RUNNING; t_node[t] = n
                        break

    # static predicate classes: task class 0 = no selector, 1+z = nodeSelector zone=z ; node class = zone
    t_cls = np.where(j_zone_sel, 1 + j_zone, 0)[task_job].astype(np.uint32)
    n_tc, n_nc = 1 + p.n_zones, p.n_zones
    compat = np.zeros((n_tc * n_nc + 7) // 8, np.uint8)
    for tc in range(n_tc):
        for nc in range(n_nc):
            if tc == 0 or tc - 1 == nc:
                b = tc * n_nc + nc
                compat[b >> 3] |= 1 << (b & 7)

    queue_weight = S(50).choice(Q, np.array([1, 2, 4, 8], np.int32)).astype(np.int32)
    return SessionSnapshot(
        n_res=R, n_nodes=N, n_tasks=T, n_jobs=J, n_queues=Q, n_task_classes=n_tc, n_node_classes=n_nc,
        node_idle=node_idle, node_releasing=node_rel, node_allocatable=node_alloc, node_scalar_mask=node_mask,
        node_alloc_cpu=cpu_m.astype(np.int64), node_alloc_mem=mem_b.astype(np.int64),
        node_nz_cpu=node_nzc, node_nz_mem=node_nzm, node_max_pods=np.full(N, 110, np.int32), node_pod_cnt=node_cnt,
        node_class=node_zone,
        task_resreq=t_res, task_init_resreq=t_init, task_scalar_mask=t_mask, task_nz_cpu=t_nzc, task_nz_mem=t_nzm,
        task_job=task_job, task_class=t_cls, task_priority=t_prio, task_creation=t_cre, task_status=t_status,
        task_node=t_node, job_task_begin=begin, job_queue=job_queue, job_min_available=min_avail,
        job_priority=job_prio, job_creation=job_creation, queue_weight=queue_weight,
        queue_creation=np.zeros(Q, np.int64), class_compat=compat,
    )
