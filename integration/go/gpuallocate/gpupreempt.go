// gpupreempt.go — the preempt action on the engine (SOURCE ONLY, like gpuallocate.go: no Go toolchain in the build image; the
// identical C ABI is exercised through kube-batch_amd/engine.py and tests/test_gpu_preempt.py).
//
// Replaces the Execute body of pkg/scheduler/actions/preempt/preempt.go:45-168: flatten the Session as it stands (statuses
// Allocated / Pipelined / Releasing of this cycle's earlier actions included), run kb_run_preempt, replay the journal through
// framework.Statement — Evict / Pipeline / Commit / Discard in the engine's order, discarded statements too (their Pipeline
// leaves the sticky NodeName behind exactly as in the reference: statement.go:155-190, api/node_info.go:217-243).
package gpuallocate

/*
#include <stdlib.h>
#include "kb_engine.h"
*/
import "C"

import (
	"runtime"
	"unsafe"

	"github.com/golang/glog"

	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/actions/preempt"
	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/api"
	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/framework"
)

type gpuPreemptAction struct {
	alloc    *gpuAllocateAction // shares the engine (one per process) and ensureEngine
	fallback framework.Action   // the stock preempt action
}

// NewPreempt registers next to gpuallocate.New(): framework.RegisterAction(gpuallocate.NewPreempt(a)) with the same *gpuAllocateAction
func NewPreempt(a *gpuAllocateAction) *gpuPreemptAction {
	return &gpuPreemptAction{alloc: a, fallback: preempt.New()}
}

func (p *gpuPreemptAction) Name() string  { return "gpupreempt" } // or "preempt" to override the stock action
func (p *gpuPreemptAction) Initialize()   {}
func (p *gpuPreemptAction) UnInitialize() {}

func (p *gpuPreemptAction) Execute(ssn *framework.Session) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()

	if err := p.alloc.ensureEngine(ssn); err != nil {
		glog.Warningf("gpupreempt: %v; falling back to the stock preempt action", err)
		p.fallback.Execute(ssn)
		return
	}
	fl, err := flatten(ssn)
	if err != nil {
		glog.V(3).Infof("gpupreempt: %v; stock action takes this cycle", err)
		p.fallback.Execute(ssn)
		return
	}
	defer fl.free()
	if len(fl.tasks) == 0 {
		return
	}
	if rc := C.kb_session_load(p.alloc.engine, &fl.snap); rc != C.KB_OK {
		glog.Warningf("gpupreempt: load rc=%d (%s); stock action takes this cycle", rc, C.GoString(C.kb_last_error(p.alloc.engine)))
		p.fallback.Execute(ssn)
		return
	}
	ops, n, rc := runJournal(p.alloc.engine, &fl.snap, false, len(fl.tasks))
	if ops != nil {
		defer C.free(unsafe.Pointer(ops))
	}
	if rc != C.KB_OK { // nothing was applied: the stock action is still valid
		glog.Warningf("gpupreempt: run rc=%d (%s); stock action takes this cycle", int(rc), C.GoString(C.kb_last_error(p.alloc.engine)))
		p.fallback.Execute(ssn)
		return
	}
	replayPreemptJournal(ssn, fl, (*[1 << 28]C.kb_stmt_op)(unsafe.Pointer(ops))[:n:n])
}

// replayPreemptJournal applies kb_run_preempt's journal through framework.Statement in the engine's order; it returns how many entries the
// Session refused (0: the Session stands where the engine's session stands, cycle.go)
func replayPreemptJournal(ssn *framework.Session, fl *flat, journal []C.kb_stmt_op) int {
	failed := 0
	var stmt *framework.Statement
	cur := C.uint32_t(0)
	for i := range journal {
		op := journal[i]
		if op.stmt != cur { // preempt.go:93 / :153: stmt := ssn.Statement()
			stmt = ssn.Statement()
			cur = op.stmt
		}
		switch op.op {
		case C.KB_OP_EVICT: // preempt.go:232: the preemptee is the node's own clone of the task (preempt.go:203-209)
			node := fl.nodes[op.node]
			victim, found := node.Tasks[api.PodKey(fl.tasks[op.task].Pod)]
			if !found {
				glog.Errorf("gpupreempt: victim %s is not on node %s any more", fl.tasks[op.task].UID, node.Name)
				failed++
				continue
			}
			if err := stmt.Evict(victim.Clone(), "preempt"); err != nil {
				glog.Errorf("gpupreempt: evict %s: %v", victim.UID, err)
				failed++
			}
		case C.KB_OP_PIPELINE: // preempt.go:248
			if err := stmt.Pipeline(fl.tasks[op.task], fl.nodes[op.node].Name); err != nil {
				glog.Errorf("gpupreempt: pipeline %s on %s: %v", fl.tasks[op.task].UID, fl.nodes[op.node].Name, err)
				failed++
			}
		case C.KB_OP_COMMIT: // preempt.go:124 / :162
			stmt.Commit()
		case C.KB_OP_DISCARD: // preempt.go:131
			stmt.Discard()
		}
	}
	return failed
}

// runJournal calls kb_run_preempt / kb_run_reclaim with a journal buffer in C memory (the engine fills it, Go only reads it).  There is
// no useful a-priori bound on the journal: every discarded statement adds its Evict / Pipeline entries again, so a first guess of a few
// entries per task is used and, on KB_E_CAPACITY (no result was applied; *n_out holds the required count), the session is loaded again
// — the action may have refreshed the device's copy of some nodes while it ran — and the call repeated once with exactly that size.
// The caller frees the returned buffer.
func runJournal(eng *C.kb_engine, snap *C.kb_snapshot, reclaim bool, nTasks int) (*C.kb_stmt_op, int, C.int) {
	capOps := C.size_t(4*nTasks + 16)
	for attempt := 0; ; attempt++ {
		ops := (*C.kb_stmt_op)(C.calloc(capOps, C.size_t(unsafe.Sizeof(C.kb_stmt_op{}))))
		if ops == nil {
			return nil, 0, C.KB_E_INTERNAL
		}
		var n C.uint64_t
		var rc C.int
		if reclaim {
			rc = C.kb_run_reclaim(eng, ops, C.uint64_t(capOps), &n)
		} else {
			rc = C.kb_run_preempt(eng, ops, C.uint64_t(capOps), &n)
		}
		if rc == C.KB_E_CAPACITY && attempt == 0 && n > 0 {
			C.free(unsafe.Pointer(ops))
			if rl := C.kb_session_load(eng, snap); rl != C.KB_OK {
				return nil, 0, rl
			}
			capOps = C.size_t(n)
			continue
		}
		return ops, int(n), rc
	}
}

// ---- reclaim (pkg/scheduler/actions/reclaim/reclaim.go:41-193) on the engine: same flatten / load, kb_run_reclaim, and a replay
// through the Session itself — reclaim uses no Statement: ssn.Evict (framework/session.go:317-354) and ssn.Pipeline (:194-232).
type gpuReclaimAction struct {
	alloc    *gpuAllocateAction
	fallback framework.Action
}

// NewReclaim: framework.RegisterAction(gpuallocate.NewReclaim(a, reclaim.New())) — the stock action is passed in so that this
// file does not import a second actions package for one constructor.
func NewReclaim(a *gpuAllocateAction, stock framework.Action) *gpuReclaimAction {
	return &gpuReclaimAction{alloc: a, fallback: stock}
}

func (p *gpuReclaimAction) Name() string  { return "gpureclaim" } // or "reclaim" to override the stock action
func (p *gpuReclaimAction) Initialize()   {}
func (p *gpuReclaimAction) UnInitialize() {}

func (p *gpuReclaimAction) Execute(ssn *framework.Session) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()

	if err := p.alloc.ensureEngine(ssn); err != nil {
		p.fallback.Execute(ssn)
		return
	}
	fl, err := flatten(ssn)
	if err != nil {
		p.fallback.Execute(ssn)
		return
	}
	defer fl.free()
	if len(fl.tasks) == 0 {
		return
	}
	if rc := C.kb_session_load(p.alloc.engine, &fl.snap); rc != C.KB_OK {
		p.fallback.Execute(ssn)
		return
	}
	ops, n, rc := runJournal(p.alloc.engine, &fl.snap, true, len(fl.tasks))
	if ops != nil {
		defer C.free(unsafe.Pointer(ops))
	}
	if rc != C.KB_OK {
		glog.Warningf("gpureclaim: run rc=%d (%s); stock action takes this cycle", int(rc), C.GoString(C.kb_last_error(p.alloc.engine)))
		p.fallback.Execute(ssn)
		return
	}
	replayReclaimJournal(ssn, fl, (*[1 << 28]C.kb_stmt_op)(unsafe.Pointer(ops))[:n:n])
}

// replayReclaimJournal applies kb_run_reclaim's journal through the Session itself (reclaim uses no Statement); it returns how many entries
// the Session refused
func replayReclaimJournal(ssn *framework.Session, fl *flat, journal []C.kb_stmt_op) int {
	failed := 0
	for i := range journal {
		op := journal[i]
		switch op.op {
		case C.KB_OP_EVICT: // reclaim.go:163: the reclaimee is the node's own clone of the task (:136-139)
			node := fl.nodes[op.node]
			victim, found := node.Tasks[api.PodKey(fl.tasks[op.task].Pod)]
			if !found {
				glog.Errorf("gpureclaim: victim %s is not on node %s any more", fl.tasks[op.task].UID, node.Name)
				failed++
				continue
			}
			if err := ssn.Evict(victim.Clone(), "reclaim"); err != nil {
				glog.Errorf("gpureclaim: evict %s: %v", victim.UID, err)
				failed++
			}
		case C.KB_OP_PIPELINE: // reclaim.go:178
			if err := ssn.Pipeline(fl.tasks[op.task], fl.nodes[op.node].Name); err != nil {
				glog.Errorf("gpureclaim: pipeline %s on %s: %v", fl.tasks[op.task].UID, fl.nodes[op.node].Name, err)
				failed++
			}
		}
	}
	return failed
}
