// flatten.go — framework.Session -> kb_snapshot (SOURCE ONLY, see gpuallocate.go).  The Python twin of this file is
// kube-batch_amd/snapshot.py:flatten; both produce the same canonical order (SURVEY.md §8c):
//
//	nodes  ascending name          (ssn.Nodes is a map: framework/session.go:43)
//	queues ascending QueueID       (ssn.Queues)
//	jobs   ascending JobID         (ssn.Jobs), tasks of a job ascending TaskID
//
// Every array is C.malloc'ed for the duration of one Execute and freed by free(); the engine copies what it needs in
// kb_session_load and keeps no caller pointer (cgo pointer rules).
package gpuallocate

/*
#include <stdlib.h>
#include <string.h>
#include "kb_engine.h"
*/
import "C"

import (
	"fmt"
	"sort"
	"unsafe"

	v1 "k8s.io/api/core/v1"
	v1qos "k8s.io/kubernetes/pkg/apis/core/v1/helper/qos"
	"k8s.io/kubernetes/pkg/scheduler/algorithm/predicates"
	"k8s.io/kubernetes/pkg/scheduler/algorithm/priorities"
	priorityutil "k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/util"
	"k8s.io/kubernetes/pkg/scheduler/nodeinfo"

	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/api"
	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/framework"
)

type flat struct {
	snap  C.kb_snapshot
	nodes []*api.NodeInfo // index -> object, for the replay
	tasks []*api.TaskInfo
	bufs  []unsafe.Pointer
}

func (f *flat) free() {
	for _, p := range f.bufs {
		C.free(p)
	}
	f.bufs = nil
}

// calloc-backed typed views -------------------------------------------------------------------------------
func (f *flat) f64(n int) []float64 {
	p := C.calloc(C.size_t(n+1), 8)
	f.bufs = append(f.bufs, p)
	return (*[1 << 30]float64)(p)[:n:n]
}
func (f *flat) i64(n int) []int64 {
	p := C.calloc(C.size_t(n+1), 8)
	f.bufs = append(f.bufs, p)
	return (*[1 << 30]int64)(p)[:n:n]
}
func (f *flat) u64(n int) []uint64 {
	p := C.calloc(C.size_t(n+1), 8)
	f.bufs = append(f.bufs, p)
	return (*[1 << 30]uint64)(p)[:n:n]
}
func (f *flat) u32(n int) []uint32 {
	p := C.calloc(C.size_t(n+1), 4)
	f.bufs = append(f.bufs, p)
	return (*[1 << 30]uint32)(p)[:n:n]
}
func (f *flat) i32(n int) []int32 {
	p := C.calloc(C.size_t(n+1), 4)
	f.bufs = append(f.bufs, p)
	return (*[1 << 30]int32)(p)[:n:n]
}
func (f *flat) u16(n int) []uint16 {
	p := C.calloc(C.size_t(n+1), 2)
	f.bufs = append(f.bufs, p)
	return (*[1 << 30]uint16)(p)[:n:n]
}
func (f *flat) u8(n int) []uint8 {
	p := C.calloc(C.size_t(n+1), 1)
	f.bufs = append(f.bufs, p)
	return (*[1 << 30]uint8)(p)[:n:n]
}

// errUnsupported makes Execute hand the cycle to the stock action (INTEGRATION.md §1)
type errUnsupported string

func (e errUnsupported) Error() string { return string(e) }

// scalar resource names of the session, sorted: dimension d = 2 + rank
func scalarDims(ssn *framework.Session) map[v1.ResourceName]int {
	set := map[v1.ResourceName]bool{}
	for _, n := range ssn.Nodes {
		for name := range n.Allocatable.ScalarResources {
			set[name] = true
		}
	}
	for _, j := range ssn.Jobs {
		for _, t := range j.Tasks {
			for name := range t.Resreq.ScalarResources {
				set[name] = true
			}
			for name := range t.InitResreq.ScalarResources {
				set[name] = true
			}
		}
	}
	names := make([]string, 0, len(set))
	for n := range set {
		names = append(names, string(n))
	}
	sort.Strings(names)
	dims := map[v1.ResourceName]int{}
	for i, n := range names {
		dims[v1.ResourceName(n)] = 2 + i
	}
	return dims
}

// put writes an api.Resource into column i of a dimension-major [R][n] matrix and returns its scalar-key mask
func put(dst []float64, n, i int, r *api.Resource, dims map[v1.ResourceName]int) uint32 {
	dst[0*n+i] = r.MilliCPU
	dst[1*n+i] = r.Memory
	var mask uint32
	for name, v := range r.ScalarResources {
		d := dims[name]
		dst[d*n+i] = v
		mask |= 1 << uint(d-2)
	}
	return mask
}

// podNonZero = sum over containers of GetNonzeroRequests (vendor/.../priorities/util/non_zero.go:48-61), what
// nodeinfo.calculateResource accumulates into nonzeroRequest (vendor/.../nodeinfo/node_info.go:502-517)
func podNonZero(pod *v1.Pod) (cpu, mem int64) {
	for i := range pod.Spec.Containers {
		c, m := priorityutil.GetNonzeroRequests(&pod.Spec.Containers[i].Resources.Requests)
		cpu += c
		mem += m
	}
	return
}

// static-predicate classes: everything the predicates plugin checks that does not change inside a cycle
// pressureFlags: the optional checks of the predicates plugin (plugins/predicates/predicates.go:66-110, default off).  The
// engine does not take them as arguments (kb_engine.h: KB_ARG_PRED_*): they are static per (pod class, node class) and are
// folded into the class table here, like the other node-condition predicates.
type pressureFlags struct{ mem, disk, pid bool }

func pressureArgs(ssn *framework.Session) pressureFlags {
	var pf pressureFlags
	for _, tier := range ssn.Tiers {
		for _, p := range tier.Plugins {
			if p.Name == "predicates" {
				args := framework.Arguments(p.Arguments) // conf.PluginOption.Arguments is a plain map[string]string
				args.GetBool(&pf.mem, "predicate.MemoryPressureEnable")
				args.GetBool(&pf.disk, "predicate.DiskPressureEnable")
				args.GetBool(&pf.pid, "predicate.PIDPressureEnable")
			}
		}
	}
	return pf
}

func nodeClassKey(n *api.NodeInfo) string {
	// only what the predicates read of a condition: its type and status (heartbeat times would make every node a class)
	conds := make([]string, 0, len(n.Node.Status.Conditions))
	for _, c := range n.Node.Status.Conditions {
		conds = append(conds, string(c.Type)+"="+string(c.Status))
	}
	taints := make([]string, 0, len(n.Node.Spec.Taints)) // not %v of the structs: TimeAdded is a pointer and would print as an address
	for _, t := range n.Node.Spec.Taints {
		taints = append(taints, t.Key+"="+t.Value+":"+string(t.Effect))
	}
	return fmt.Sprintf("%v|%v|%v|%v", n.Node.Labels, taints, n.Node.Spec.Unschedulable, conds)
}
// pendingWithClaim: a Pending pod whose spec names a PersistentVolumeClaim.  ssn.Allocate opens with cache.AllocateVolumes
// (framework/session.go:236-238 -> cache/cache.go:206-211 -> volumeBinder.AssumePodVolumes, vendor/.../controller/volume/scheduling/
// scheduler_binder.go:253-322).  Nothing in pkg/scheduler calls FindPodVolumes at this commit, so podBindingCache holds no decision for
// the pod: GetBindings / GetProvisionedPVCs return nil (scheduler_binder_cache.go:124-154), both loops are empty and the call returns
// (false, nil), or (true, nil) when every claim is bound.  A claim can therefore never veto an Allocate, and the replay goes through the
// real ssn.Allocate anyway (VolumeReady, BindVolumes at dispatch).  The one trace inside the session is `assumedPod.Spec.NodeName =
// nodeName` (:269) for a pod whose claims are not all bound; its only reader is nodeorder's cachedNodeInfo (plugins/nodeorder/
// nodeorder.go:55).  So the claim matters only together with inter-pod terms: buildInterpod refuses that combination, nothing else.
func pendingWithClaim(t *api.TaskInfo) bool {
	if t.Status != api.Pending {
		return false
	}
	for i := range t.Pod.Spec.Volumes {
		if t.Pod.Spec.Volumes[i].PersistentVolumeClaim != nil {
			return true
		}
	}
	return false
}

func taskClassKey(t *api.TaskInfo, pf pressureFlags) (string, error) {
	sp := &t.Pod.Spec
	bestEffort := pf.mem && v1qos.GetPodQOS(t.Pod) == v1.PodQOSBestEffort // memory pressure only turns BestEffort pods away
	tols := make([]string, 0, len(sp.Tolerations)) // TolerationSeconds is a pointer (and irrelevant to ToleratesTaint)
	for _, x := range sp.Tolerations {
		tols = append(tols, x.Key+"|"+string(x.Operator)+"|"+x.Value+"|"+string(x.Effect))
	}
	// *v1.Affinity is a Stringer (generated.pb.go): %v prints the whole tree, not pointers
	// inter-pod (anti)affinity is dynamic, not part of the static class: interpod.go folds it into kb_interpod
	var nodeAff *v1.NodeAffinity
	if sp.Affinity != nil {
		nodeAff = sp.Affinity.NodeAffinity
	}
	return fmt.Sprintf("%v|%v|%v|%v", sp.NodeSelector, nodeAff, tols, bestEffort), nil
}

// host ports: a distinct (hostIP, protocol, hostPort > 0) of the session's pods is one bit (nodeinfo/host_ports.go sanitises "" to
// 0.0.0.0 / TCP) if it can ever take part in a conflict test.  PodFitsHostPorts is only asked for a Pending task (allocate, backfill, the
// preemptors of preempt / reclaim) and a placement adds a Pending task's own ports: a triple that conflicts with no Pending task's port
// (what the daemons already on the nodes listen on, usually) can never decide anything and gets no bit.  Any number of the rest: masks of
// portTable.words() 64-bit words (kb_snapshot.port_words)
type hostPort struct {
	ip, proto string
	port      int32
}

func podHostPorts(pod *v1.Pod) []hostPort {
	var out []hostPort
	for i := range pod.Spec.Containers {
		for _, cp := range pod.Spec.Containers[i].Ports {
			if cp.HostPort <= 0 {
				continue
			}
			hp := hostPort{cp.HostIP, string(cp.Protocol), cp.HostPort}
			if hp.ip == "" {
				hp.ip = "0.0.0.0"
			}
			if hp.proto == "" {
				hp.proto = string(v1.ProtocolTCP)
			}
			out = append(out, hp)
		}
	}
	return out
}

// HostPortInfo.CheckConflict between a wanted and a used triple (vendor/.../nodeinfo/host_ports.go:107-135)
func portsConflict(a, b hostPort) bool {
	return a.proto == b.proto && a.port == b.port && (a.ip == b.ip || a.ip == "0.0.0.0" || b.ip == "0.0.0.0")
}

type portTable struct {
	all []hostPort
	bit map[hostPort]uint
}

func newPortTable(ssn *framework.Session) (*portTable, error) {
	pt := &portTable{bit: map[hostPort]uint{}}
	add := func(pod *v1.Pod) {
		for _, hp := range podHostPorts(pod) {
			if _, ok := pt.bit[hp]; !ok {
				pt.bit[hp] = 0
				pt.all = append(pt.all, hp)
			}
		}
	}
	var asked []hostPort // the ports of the Pending tasks
	for _, j := range ssn.Jobs {
		for _, t := range j.Tasks {
			add(t.Pod)
			if t.Status == api.Pending {
				asked = append(asked, podHostPorts(t.Pod)...)
			}
		}
	}
	for _, n := range ssn.Nodes {
		for _, t := range n.Tasks {
			add(t.Pod)
		}
	}
	kept := pt.all[:0]
	for _, hp := range pt.all {
		live := false
		for _, a := range asked {
			if portsConflict(a, hp) {
				live = true
				break
			}
		}
		if live {
			kept = append(kept, hp)
		} else {
			delete(pt.bit, hp)
		}
	}
	pt.all = kept
	// Any number of triples: masks of words() 64-bit words (kb_snapshot.port_words).  The engine decides a Pending pod whose masks reach
	// beyond word 0 in a device round of its own, so with more than 64 triples the ones the Pending pods' ports conflict with most often
	// take the low bits (the same rule as kube-batch_amd/snapshot.py; ties and the <= 64 case: ascending ip, protocol, port).
	weight := map[hostPort]int{}
	if len(pt.all) > 64 {
		for _, j := range ssn.Jobs {
			for _, t := range j.Tasks {
				if t.Status != api.Pending {
					continue
				}
				mine := podHostPorts(t.Pod)
				for _, hp := range pt.all {
					for _, a := range mine {
						if portsConflict(a, hp) {
							weight[hp]++
							break
						}
					}
				}
			}
		}
	}
	sort.Slice(pt.all, func(i, j int) bool {
		a, b := pt.all[i], pt.all[j]
		if weight[a] != weight[b] {
			return weight[a] > weight[b]
		}
		if a.ip != b.ip {
			return a.ip < b.ip
		}
		if a.proto != b.proto {
			return a.proto < b.proto
		}
		return a.port < b.port
	})
	for i, hp := range pt.all {
		pt.bit[hp] = uint(i)
	}
	return pt, nil
}

// 64-bit words per mask (kb_snapshot.port_words)
func (pt *portTable) words() int {
	if len(pt.all) <= 64 {
		return 1
	}
	return (len(pt.all) + 63) / 64
}

// masks ORs the pod's bits into want / conflict, two rows of words() words (triple i: bit i % 64 of word i / 64)
func (pt *portTable) masks(pod *v1.Pod, want, conflict []uint64) {
	mine := podHostPorts(pod)
	for _, hp := range mine {
		if b, ok := pt.bit[hp]; ok { // no bit: the triple conflicts with no pending pod's port
			want[b/64] |= 1 << (b % 64)
		}
	}
	if conflict == nil {
		return
	}
	for _, other := range pt.all {
		for _, hp := range mine {
			if portsConflict(hp, other) {
				b := pt.bit[other]
				conflict[b/64] |= 1 << (b % 64)
				break
			}
		}
	}
}

// NodeAffinity priority, Map step, for one (task class, node class) pair: the vendored function itself
// (vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/node_affinity.go:34-77); the engine applies NormalizeReduce(10)
// over each task's feasible nodes and the plugin weight
func affinityCount(t *api.TaskInfo, n *api.NodeInfo) int32 {
	ni := nodeinfo.NewNodeInfo()
	ni.SetNode(n.Node)
	hp, err := priorities.CalculateNodeAffinityPriorityMap(t.Pod, nil, ni)
	if err != nil {
		return 0
	}
	return int32(hp.Score)
}

// one (task class, node class) pair through the vendored predicates themselves, the ones
// plugins/predicates/predicates.go:123-157 and :181-207 call
func staticOK(t *api.TaskInfo, n *api.NodeInfo, pf pressureFlags) bool {
	ni := nodeinfo.NewNodeInfo()
	ni.SetNode(n.Node)
	if ok, _, _ := predicates.CheckNodeConditionPredicate(t.Pod, nil, ni); !ok {
		return false
	}
	if ok, _, _ := predicates.CheckNodeUnschedulablePredicate(t.Pod, nil, ni); !ok {
		return false
	}
	if ok, _, _ := predicates.PodMatchNodeSelector(t.Pod, nil, ni); !ok {
		return false
	}
	if ok, _, _ := predicates.PodToleratesNodeTaints(t.Pod, nil, ni); !ok {
		return false
	}
	if pf.mem { // plugins/predicates/predicates.go:201-215
		if ok, _, _ := predicates.CheckNodeMemoryPressurePredicate(t.Pod, nil, ni); !ok {
			return false
		}
	}
	if pf.disk { // :217-231
		if ok, _, _ := predicates.CheckNodeDiskPressurePredicate(t.Pod, nil, ni); !ok {
			return false
		}
	}
	if pf.pid { // :233-247
		if ok, _, _ := predicates.CheckNodePIDPressurePredicate(t.Pod, nil, ni); !ok {
			return false
		}
	}
	return true
}

func taskStatus(s api.TaskStatus) uint8 {
	switch s {
	case api.Pending:
		return C.KB_TASK_PENDING
	case api.Allocated:
		return C.KB_TASK_ALLOCATED
	case api.Pipelined:
		return C.KB_TASK_PIPELINED
	case api.Binding:
		return C.KB_TASK_BINDING
	case api.Bound:
		return C.KB_TASK_BOUND
	case api.Running:
		return C.KB_TASK_RUNNING
	case api.Releasing:
		return C.KB_TASK_RELEASING
	case api.Succeeded:
		return C.KB_TASK_SUCCEEDED
	case api.Failed:
		return C.KB_TASK_FAILED
	}
	return C.KB_TASK_UNKNOWN
}

func flatten(ssn *framework.Session) (*flat, error) {
	f := &flat{}
	pf := pressureArgs(ssn)
	dims := scalarDims(ssn)
	R := 2 + len(dims)
	if R > C.KB_MAX_RES {
		return nil, errUnsupported("too many scalar resource names")
	}

	// ---- canonical orders
	nodeNames := make([]string, 0, len(ssn.Nodes))
	for name := range ssn.Nodes {
		nodeNames = append(nodeNames, name)
	}
	sort.Strings(nodeNames)
	nodeIdx := map[string]uint32{}
	for i, name := range nodeNames {
		nodeIdx[name] = uint32(i)
		f.nodes = append(f.nodes, ssn.Nodes[name])
	}
	queueIDs := make([]string, 0, len(ssn.Queues))
	for id := range ssn.Queues {
		queueIDs = append(queueIDs, string(id))
	}
	sort.Strings(queueIDs)
	queueIdx := map[api.QueueID]uint32{}
	for i, id := range queueIDs {
		queueIdx[api.QueueID(id)] = uint32(i)
	}
	jobIDs := make([]string, 0, len(ssn.Jobs))
	for id := range ssn.Jobs {
		jobIDs = append(jobIDs, string(id))
	}
	sort.Strings(jobIDs)

	N, Q, J := len(nodeNames), len(queueIDs), len(jobIDs)
	T := 0
	for _, j := range ssn.Jobs {
		T += len(j.Tasks)
	}

	if T == 0 || N == 0 {
		// No task, or no node to put one on: no action has anything to decide, and the callers return before they touch the
		// engine (len(fl.tasks) == 0).  Taking &x[0] of the empty arrays below would panic.
		return f, nil
	}
	if Q == 0 {
		// allocate skips every job ("queue not found", allocate.go:56-60) but backfill.go:40-71 does not look at queues at all:
		// rather than model a session the reference itself handles inconsistently, the stock actions take the cycle
		return nil, errUnsupported("session without queues")
	}

	ports, err := newPortTable(ssn)
	if err != nil {
		return nil, err
	}

	// ---- nodes (api/node_info.go:28-47)
	idle, rel, alloc := f.f64(R*N), f.f64(R*N), f.f64(R*N)
	nmask := f.u32(N)
	acpu, amem, nzc, nzm := f.i64(N), f.i64(N), f.i64(N), f.i64(N)
	maxPods, podCnt := f.i32(N), f.i32(N)
	nclass := f.u32(N)
	Wh := ports.words()
	nports := f.u64(N * Wh) // [N][Wh]
	nodeClasses := map[string]uint32{}
	var nodeClassRep []*api.NodeInfo
	for i, n := range f.nodes {
		put(idle, N, i, n.Idle, dims)
		put(rel, N, i, n.Releasing, dims)
		nmask[i] = put(alloc, N, i, n.Allocatable, dims)
		acpu[i] = n.Node.Status.Allocatable.Cpu().MilliValue() // what nodeinfo.SetNode stores (vendor/.../nodeinfo/node_info.go:625-628)
		amem[i] = n.Node.Status.Allocatable.Memory().Value()
		for _, t := range n.Tasks { // every entry of ni.Tasks whatever its status (node_info.go:277-283)
			c, m := podNonZero(t.Pod)
			nzc[i] += c
			nzm[i] += m
			ports.masks(t.Pod, nports[i*Wh:(i+1)*Wh], nil) // nodeinfo.UsedPorts(): the ports of every pod in ni.Tasks
		}
		maxPods[i] = int32(n.Allocatable.MaxTaskNum)
		podCnt[i] = int32(len(n.Tasks))
		key := nodeClassKey(n)
		id, ok := nodeClasses[key]
		if !ok {
			id = uint32(len(nodeClasses))
			nodeClasses[key] = id
			nodeClassRep = append(nodeClassRep, n)
		}
		nclass[i] = id
	}

	// ---- jobs and tasks (api/job_info.go:36-54, :127-154)
	tres, tinit := f.f64(R*T), f.f64(R*T)
	tmask := f.u32(T)
	tnzc, tnzm, tcreate := f.i64(T), f.i64(T), f.i64(T)
	tjob, tclass, tnode := f.u32(T), f.u32(T), f.u32(T)
	tprio := f.i32(T)
	tstatus := f.u8(T)
	twant, tconf := f.u64(T*Wh), f.u64(T*Wh) // [T][Wh]
	tprot := f.u8(T)
	jbegin := f.u32(J + 1)
	jqueue := f.u32(J)
	jmin, jprio := f.i32(J), f.i32(J)
	jcreate := f.i64(J)
	taskClasses := map[string]uint32{}
	var taskClassRep []*api.TaskInfo
	t := 0
	for ji, id := range jobIDs {
		job := ssn.Jobs[api.JobID(id)]
		jbegin[ji] = uint32(t)
		if q, ok := queueIdx[job.Queue]; ok {
			jqueue[ji] = q
		} else {
			jqueue[ji] = C.KB_NONE // "queue not found": allocate.go:56-60 skips the job
		}
		jmin[ji] = job.MinAvailable
		jprio[ji] = job.Priority
		jcreate[ji] = job.CreationTimestamp.Unix()
		uids := make([]string, 0, len(job.Tasks))
		for uid := range job.Tasks {
			uids = append(uids, string(uid))
		}
		sort.Strings(uids)
		for _, uid := range uids {
			ti := job.Tasks[api.TaskID(uid)]
			f.tasks = append(f.tasks, ti)
			tmask[t] = put(tres, T, t, ti.Resreq, dims)
			put(tinit, T, t, ti.InitResreq, dims)
			tnzc[t], tnzm[t] = podNonZero(ti.Pod)
			tjob[t] = uint32(ji)
			tprio[t] = ti.Priority
			tcreate[t] = ti.Pod.CreationTimestamp.Unix()
			tstatus[t] = taskStatus(ti.Status)
			ports.masks(ti.Pod, twant[t*Wh:(t+1)*Wh], tconf[t*Wh:(t+1)*Wh])
			if ti.Namespace == "kube-system" || ti.Pod.Spec.PriorityClassName == "system-cluster-critical" || ti.Pod.Spec.PriorityClassName == "system-node-critical" {
				tprot[t] = 1 // plugins/conformance/conformance.go:44-58 (read by preempt / reclaim)
			}
			tnode[t] = C.KB_NONE
			// a terminated pod keeps its Spec.NodeName but was never added to that node's Tasks (cache/event_handlers.go:72-92 addTask:
			// !isTerminated): for the snapshot it is on no node.  (Round 3: such a task used to trip the "not in that node's Tasks" check
			// below, i.e. every session with a finished pod went to the stock action.)
			terminated := ti.Status == api.Succeeded || ti.Status == api.Failed
			if idx, ok := nodeIdx[ti.NodeName]; ok && ti.NodeName != "" && !terminated {
				tnode[t] = idx
			}
			if ti.Status == api.Pending && ti.NodeName != "" {
				// Only an earlier action of this cycle can leave this behind: a preempt statement pipelined the task and was
				// discarded (statement.go:152-187); NodeInfo.RemoveTask keeps task.NodeName (node_info.go:217-243), so AddTask on
				// any other node now fails after ssn.Allocate flipped the status (node_info.go:173-176, session.go:243).  The
				// engine's allocate does not model that failure: the stock action takes the cycle.
				f.free()
				return nil, errUnsupported("pending task with a stale NodeName (un-pipelined by a discarded statement)")
			}
			if ti.NodeName != "" && !terminated {
				// The snapshot says "task_node set <=> the task is in that node's Tasks".  A statement that pipelined a task carrying a
				// stale NodeName only logs the failed AddTask (statement.go:113-150): the task is then Pipelined, named after its old
				// host and on no node at all — a state the arrays cannot express (inside one loaded session the engine tracks it
				// itself: HostSession::t_off_node).
				if node, ok := ssn.Nodes[ti.NodeName]; ok {
					if _, on := node.Tasks[api.PodKey(ti.Pod)]; !on {
						f.free()
						return nil, errUnsupported("task carries a NodeName but is not in that node's Tasks (failed AddTask of an earlier statement)")
					}
				}
			}
			key, err := taskClassKey(ti, pf)
			if err != nil {
				f.free()
				return nil, err
			}
			cid, ok := taskClasses[key]
			if !ok {
				cid = uint32(len(taskClasses))
				taskClasses[key] = cid
				taskClassRep = append(taskClassRep, ti)
			}
			tclass[t] = cid
			t++
		}
	}
	jbegin[J] = uint32(t)

	// ---- queues (api/queue_info.go:74-93)
	qweight := f.i32(Q)
	qcreate := f.i64(Q)
	for i, id := range queueIDs {
		q := ssn.Queues[api.QueueID(id)]
		qweight[i] = q.Weight
		qcreate[i] = q.Queue.CreationTimestamp.Unix()
	}

	// ---- static predicates once per (task class, node class)
	ntc, nnc := len(taskClassRep), len(nodeClassRep)
	compat := f.u8((ntc*nnc + 7) / 8)
	affinity := f.i32(ntc * nnc)
	anyAffinity := false
	for a, tr := range taskClassRep {
		for b, nr := range nodeClassRep {
			if staticOK(tr, nr, pf) {
				bit := a*nnc + b
				compat[bit>>3] |= 1 << uint(bit&7)
			}
			if c := affinityCount(tr, nr); c != 0 {
				affinity[a*nnc+b] = c
				anyAffinity = true
			}
		}
	}

	s := &f.snap
	s.version = C.KB_ABI_VERSION
	s.n_res, s.n_nodes, s.n_tasks, s.n_jobs, s.n_queues = C.uint32_t(R), C.uint32_t(N), C.uint32_t(T), C.uint32_t(J), C.uint32_t(Q)
	s.n_task_classes, s.n_node_classes = C.uint32_t(ntc), C.uint32_t(nnc)
	s.node_idle = (*C.double)(unsafe.Pointer(&idle[0]))
	s.node_releasing = (*C.double)(unsafe.Pointer(&rel[0]))
	s.node_allocatable = (*C.double)(unsafe.Pointer(&alloc[0]))
	s.node_scalar_mask = (*C.uint32_t)(unsafe.Pointer(&nmask[0]))
	s.node_alloc_cpu = (*C.int64_t)(unsafe.Pointer(&acpu[0]))
	s.node_alloc_mem = (*C.int64_t)(unsafe.Pointer(&amem[0]))
	s.node_nz_cpu = (*C.int64_t)(unsafe.Pointer(&nzc[0]))
	s.node_nz_mem = (*C.int64_t)(unsafe.Pointer(&nzm[0]))
	s.node_max_pods = (*C.int32_t)(unsafe.Pointer(&maxPods[0]))
	s.node_pod_cnt = (*C.int32_t)(unsafe.Pointer(&podCnt[0]))
	s.node_class = (*C.uint32_t)(unsafe.Pointer(&nclass[0]))
	s.task_resreq = (*C.double)(unsafe.Pointer(&tres[0]))
	s.task_init_resreq = (*C.double)(unsafe.Pointer(&tinit[0]))
	s.task_scalar_mask = (*C.uint32_t)(unsafe.Pointer(&tmask[0]))
	s.task_nz_cpu = (*C.int64_t)(unsafe.Pointer(&tnzc[0]))
	s.task_nz_mem = (*C.int64_t)(unsafe.Pointer(&tnzm[0]))
	s.task_job = (*C.uint32_t)(unsafe.Pointer(&tjob[0]))
	s.task_class = (*C.uint32_t)(unsafe.Pointer(&tclass[0]))
	s.task_priority = (*C.int32_t)(unsafe.Pointer(&tprio[0]))
	s.task_creation = (*C.int64_t)(unsafe.Pointer(&tcreate[0]))
	s.task_status = (*C.uint8_t)(unsafe.Pointer(&tstatus[0]))
	s.task_node = (*C.uint32_t)(unsafe.Pointer(&tnode[0]))
	s.job_task_begin = (*C.uint32_t)(unsafe.Pointer(&jbegin[0]))
	s.job_queue = (*C.uint32_t)(unsafe.Pointer(&jqueue[0]))
	s.job_min_available = (*C.int32_t)(unsafe.Pointer(&jmin[0]))
	s.job_priority = (*C.int32_t)(unsafe.Pointer(&jprio[0]))
	s.job_creation = (*C.int64_t)(unsafe.Pointer(&jcreate[0]))
	s.queue_weight = (*C.int32_t)(unsafe.Pointer(&qweight[0]))
	s.queue_creation = (*C.int64_t)(unsafe.Pointer(&qcreate[0]))
	s.class_compat = (*C.uint8_t)(unsafe.Pointer(&compat[0]))
	if anyAffinity {
		s.class_affinity = (*C.int32_t)(unsafe.Pointer(&affinity[0]))
	}
	if len(ports.all) > 0 {
		s.node_ports = (*C.uint64_t)(unsafe.Pointer(&nports[0]))
		s.task_port_want = (*C.uint64_t)(unsafe.Pointer(&twant[0]))
		s.task_port_conflict = (*C.uint64_t)(unsafe.Pointer(&tconf[0]))
		s.port_words = C.uint32_t(Wh)
	}
	s.task_evict_protected = (*C.uint8_t)(unsafe.Pointer(&tprot[0]))
	if err := f.buildInterpod(tnode, tstatus); err != nil { // predicate p8 / priority a22: kb_interpod (nil when no pod has a term)
		f.free()
		return nil, err
	}
	return f, nil
}
