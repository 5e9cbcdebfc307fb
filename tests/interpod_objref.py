"""Inter-pod (anti)affinity restated ON KUBERNETES-SHAPED OBJECTS (labels, namespaces, selectors, topology keys), straight from the
Go sources — test infrastructure.  It is the independent check of what kube-batch_amd/snapshot.py:build_interpod folds into the
kb_interpod tables: tests/test_interpod_cpu.py walks random placement sequences and compares, pair by pair, this file's answers with
the table arithmetic that the oracle, tests/pyref.py and the engine share.

  predicate: plugins/predicates/predicates.go:249-262 -> vendor/.../algorithm/predicates/predicates.go:1261-1575 (meta == nil: the
             slow path) over plugins/util/util.go:37-90 (PodLister / CachedNodeInfo)
  priority:  plugins/nodeorder/nodeorder.go:48-62,156-160 -> vendor/.../priorities/interpod_affinity.go:99-235, fed by
             util/scheduler_helper.go:226-238 (only the feasible nodes' pods)
Maps are ranged in ascending key order."""
import importlib

snapmod = importlib.import_module("kube-batch_amd.snapshot")
_selector_matches = snapmod._selector_matches

ALLOCATED_STATUS = ("Allocated", "Binding", "Bound", "Running")      # api/helpers.go:63-72


class PodState:
    """one TaskInfo / pod: where the cache or the session holds it"""

    def __init__(self, pod, node=None, status="Pending", spec_node_name=""):
        self.pod, self.node, self.status, self.spec_node_name = pod, node, status, spec_node_name
        self.in_session = True          # part of ssn.Jobs (PodLister sees it)


class World:
    def __init__(self, nodes, states):
        self.nodes = sorted(nodes, key=lambda n: n.name)
        self.by_name = {n.name: n for n in self.nodes}
        self.states = states

    def pods_on(self, node_name):
        """NodeInfo.Pods(): every task in ni.Tasks"""
        return [s for s in self.states if s.node == node_name]

    # ---- plugins/util/util.go:37-90
    def pod_lister(self):
        """(pod, NodeName) of the session's tasks in an allocated status; NodeName = task.NodeName"""
        return [(s.pod, s.node) for s in self.states if s.in_session and s.status in ALLOCATED_STATUS]

    # ---- plugins/nodeorder/nodeorder.go:48-62
    def nodeorder_get_node(self, name):
        if name in self.by_name:
            return self.by_name[name]
        for n in self.nodes:                                # range c.session.Nodes, ascending
            if any(s.spec_node_name == "" for s in self.pods_on(n.name)):
                return n
        raise KeyError(name)


def namespaces_of(owner, term):                             # priorities/util/topologies.go:28-36
    ns, _, _ = term
    return set(ns) if ns else {owner.namespace}


def matches_term(pod, owner, term):                         # PodMatchesTermsNamespaceAndSelector with the owner's resolution
    _, sel, _ = term
    return pod.namespace in namespaces_of(owner, term) and _selector_matches(snapmod._canon_selector(sel), pod.labels)


def same_topology(a, b, key):                               # priorities/util/topologies.go:53-71
    return bool(key) and key in a.labels and key in b.labels and a.labels[key] == b.labels[key]


def predicate(w: World, pod, node) -> bool:
    """PodAffinityChecker.InterPodAffinityMatches(pod, nil, nodeInfo) (predicates.go:1261-1290)."""
    lister = w.pod_lister()
    # nodeInfo.Filter (vendor/.../nodeinfo/node_info.go: pods whose NodeName is this node but which the node does not hold are
    # dropped): every listed task sits in its node's ni.Tasks here, so the filter passes everything
    # satisfiesExistingPodsAntiAffinity (:1400-1441)
    pairs = set()
    for epod, enode in lister:
        en = w.by_name[enode]
        for term in epod.pod_anti_affinity_required:
            if matches_term(pod, epod, term) and term[2] in en.labels:
                pairs.add((term[2], en.labels[term[2]]))
    for k, v in node.labels.items():
        if (k, v) in pairs:
            return False
    if not pod.pod_affinity_required and not pod.pod_anti_affinity_required:
        return True
    # satisfiesPodsAffinityAntiAffinity, slow path (:1519-1566)
    aff, anti = pod.pod_affinity_required, pod.pod_anti_affinity_required
    match_found = terms_selector_match_found = False
    for tpod, tnode in lister:
        tn = w.by_name[tnode]
        if not match_found and aff:
            props_ok = all(matches_term(tpod, pod, t) for t in aff)            # podMatchesAllAffinityTermProperties
            if props_ok:
                terms_selector_match_found = True
                if all(same_topology(node, tn, t[2]) for t in aff):
                    match_found = True
        if anti:
            if all(matches_term(tpod, pod, t) for t in anti) and all(same_topology(node, tn, t[2]) for t in anti):
                return False
    if not match_found and aff:
        if terms_selector_match_found:
            return False
        if not all(matches_term(pod, pod, t) for t in aff):                    # targetPodMatchesAffinityOfPod(pod, pod)
            return False
    return True


def priority(w: World, pod, feasible) -> dict:
    """CalculateInterPodAffinityPriority(pod, nodeNameToInfo of the FEASIBLE nodes, feasible) -> {node name: 0..10}."""
    counts = {n.name: 0 for n in feasible}
    has_aff = bool(pod.pod_affinity_required or pod.pod_affinity_preferred)         # affinity.PodAffinity != nil
    has_anti = bool(pod.pod_anti_affinity_required or pod.pod_anti_affinity_preferred)

    def process_term(term, owner, to_check, fixed, weight):
        if matches_term(to_check, owner, term):
            for n in feasible:
                if same_topology(n, fixed, term[2]):
                    counts[n.name] += weight

    for holder in sorted(feasible, key=lambda n: n.name):
        for s in w.pods_on(holder.name):
            e = s.pod
            enode = w.nodeorder_get_node(s.spec_node_name)
            if has_aff:
                for wt, term in pod.pod_affinity_preferred:
                    process_term(term, pod, e, enode, int(wt))
            if has_anti:
                for wt, term in pod.pod_anti_affinity_preferred:
                    process_term(term, pod, e, enode, -int(wt))
            for term in e.pod_affinity_required:                                 # hardPodAffinityWeight = 1
                process_term(term, e, pod, enode, 1)
            for wt, term in e.pod_affinity_preferred:
                process_term(term, e, pod, enode, int(wt))
            for wt, term in e.pod_anti_affinity_preferred:
                process_term(term, e, pod, enode, -int(wt))
    mx = max([0] + list(counts.values()))
    mn = min([0] + list(counts.values()))
    out = {}
    for n in feasible:
        f = 0.0
        if mx - mn > 0:
            f = 10.0 * (float(counts[n.name] - mn) / float(mx - mn))
        out[n.name] = int(f)
    return out
