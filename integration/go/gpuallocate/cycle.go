// cycle.go — ONE framework action for the engine actions of a cycle (SOURCE ONLY, like the rest of this package: no Go toolchain in the
// build image).
//
// gpuallocate / gpupreempt / gpureclaim registered one by one flatten the Session and kb_session_load it once PER ACTION: a cycle of
// "reclaim, allocate, backfill, preempt" pays three flattens and three loads (round-2 advisory; at 1M tasks a flatten + load is tens of
// milliseconds).  The engine does not need that: its session carries on from one action to the next exactly as the reference's Session does —
// tests/test_gpu_preempt.py and tests/test_gpu_regressions.py run such mixed orders on ONE loaded session against the oracle, sticky
// NodeNames of discarded statements included.  What a second load protects against is the Go side drifting from the engine: a replay call the
// Session refused, or another action running in between.  This action closes both: it is ONE action in the YAML's list (nothing runs in
// between), and it keeps the loaded session only while every replayed entry was accepted — after a refusal, an engine error or a stock
// fallback the next step flattens the Session as it then stands and loads again, which is what the single actions always do.
//
//	conf:  actions: "gpucycle"                      // instead of "reclaim, allocate, backfill, preempt"
//	code:  a := gpuallocate.New()
//	       framework.RegisterAction(gpuallocate.NewCycle(a, "reclaim, allocate, backfill, preempt", map[string]framework.Action{
//	           "reclaim": reclaim.New(), "allocate": allocate.New(), "backfill": backfill.New(), "preempt": preempt.New()}))
package gpuallocate

/*
#include <stdlib.h>
#include "kb_engine.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"strings"
	"unsafe"

	"github.com/golang/glog"

	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/framework"
)

type gpuCycleAction struct {
	alloc *gpuAllocateAction          // owns the engine (one per process) and ensureEngine
	steps []string                    // the reference's action names, in the order the YAML would list them
	stock map[string]framework.Action // the reference action behind every step, for the cycles (or steps) the engine hands back
}

// NewCycle: steps is the YAML's actions line ("reclaim, allocate, backfill, preempt"); stock must hold the reference action of every step.
func NewCycle(a *gpuAllocateAction, steps string, stock map[string]framework.Action) (*gpuCycleAction, error) {
	c := &gpuCycleAction{alloc: a, stock: stock}
	for _, s := range strings.Split(steps, ",") {
		s = strings.TrimSpace(s)
		switch s {
		case "allocate", "backfill", "preempt", "reclaim":
		default:
			return nil, fmt.Errorf("gpucycle: %q is not an action the engine runs", s)
		}
		if stock[s] == nil {
			return nil, fmt.Errorf("gpucycle: no stock action given for %q", s)
		}
		c.steps = append(c.steps, s)
	}
	return c, nil
}

func (c *gpuCycleAction) Name() string  { return "gpucycle" }
func (c *gpuCycleAction) Initialize()   {}
func (c *gpuCycleAction) UnInitialize() {}

// stepResult: rc is the engine's answer; need the journal size a KB_E_CAPACITY answer asks for; failed the replayed entries the Session refused
type stepResult struct {
	rc     C.int
	need   int
	failed int
}

func (c *gpuCycleAction) Execute(ssn *framework.Session) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()

	if err := c.alloc.ensureEngine(ssn); err != nil {
		glog.Warningf("gpucycle: %v; the stock actions take this cycle", err)
		c.runStock(ssn, 0)
		return
	}
	eng := c.alloc.engine
	var fl *flat
	defer func() {
		if fl != nil {
			fl.free()
		}
	}()
	loaded := false // the engine's session stands where ssn stands
	capOps := 0     // journal entries to offer the next evict step (0: the first guess)
	for i := 0; i < len(c.steps); {
		step := c.steps[i]
		if !loaded {
			if fl != nil {
				fl.free()
				fl = nil
			}
			var err error
			fl, err = flatten(ssn) // the Session as the earlier steps (engine or stock) left it
			if err != nil {        // outside the engine's envelope: so will the rest of the cycle be
				glog.V(3).Infof("gpucycle: %v; the stock actions take the cycle from %q on", err, step)
				c.runStock(ssn, i)
				return
			}
			if len(fl.tasks) == 0 { // idle cluster (or one without nodes): no action has anything to do
				return
			}
			if rc := C.kb_session_load(eng, &fl.snap); rc != C.KB_OK {
				glog.Warningf("gpucycle: load rc=%d (%s); the stock actions take the cycle from %q on", int(rc), C.GoString(C.kb_last_error(eng)), step)
				c.runStock(ssn, i)
				return
			}
			loaded = true
		}
		res := c.runStep(ssn, fl, step, capOps)
		if res.rc == C.KB_E_CAPACITY && capOps == 0 && res.need > 0 {
			// nothing of the step was applied, but the action may have refreshed the device's copy of some nodes while it ran
			// (include/kb_engine.h: any non-OK answer of an evict action asks for a load): same step again on a fresh load, with the size asked for
			capOps = res.need
			loaded = false
			continue
		}
		capOps = 0
		if res.rc != C.KB_OK { // the step applied nothing to ssn: the reference action runs it, and the engine's session is behind from here
			glog.Warningf("gpucycle: %s rc=%d (%s); the stock action takes this step", step, int(res.rc), C.GoString(C.kb_last_error(eng)))
			c.stock[step].Execute(ssn)
			loaded = false
		} else if res.failed > 0 { // the Session refused part of the replay: it no longer stands where the engine's session stands
			loaded = false
		}
		i++
	}
}

// runStock runs the reference actions of steps[from:]
func (c *gpuCycleAction) runStock(ssn *framework.Session, from int) {
	for _, s := range c.steps[from:] {
		c.stock[s].Execute(ssn)
	}
}

// runStep runs one action on the loaded session and replays its result through the Session
func (c *gpuCycleAction) runStep(ssn *framework.Session, fl *flat, step string, capOps int) stepResult {
	eng := c.alloc.engine
	switch step {
	case "allocate", "backfill":
		capDec := C.size_t(len(fl.tasks))
		decisions := (*C.kb_decision)(C.calloc(capDec, C.size_t(unsafe.Sizeof(C.kb_decision{}))))
		if decisions == nil {
			return stepResult{rc: C.KB_E_INTERNAL}
		}
		defer C.free(unsafe.Pointer(decisions))
		var n C.uint64_t
		var rc C.int
		if step == "allocate" {
			rc = C.kb_run_allocate(eng, decisions, C.uint64_t(capDec), &n)
		} else {
			rc = C.kb_run_backfill(eng, decisions, C.uint64_t(capDec), &n)
		}
		if rc != C.KB_OK {
			return stepResult{rc: rc}
		}
		dec := (*[1 << 28]C.kb_decision)(unsafe.Pointer(decisions))[:int(n):int(n)]
		return stepResult{rc: C.KB_OK, failed: c.alloc.replay(ssn, fl, dec)}
	default: // "preempt", "reclaim": a journal of statement operations
		if capOps == 0 {
			capOps = 4*len(fl.tasks) + 16
		}
		ops := (*C.kb_stmt_op)(C.calloc(C.size_t(capOps), C.size_t(unsafe.Sizeof(C.kb_stmt_op{}))))
		if ops == nil {
			return stepResult{rc: C.KB_E_INTERNAL}
		}
		defer C.free(unsafe.Pointer(ops))
		var n C.uint64_t
		var rc C.int
		if step == "reclaim" {
			rc = C.kb_run_reclaim(eng, ops, C.uint64_t(capOps), &n)
		} else {
			rc = C.kb_run_preempt(eng, ops, C.uint64_t(capOps), &n)
		}
		if rc != C.KB_OK {
			return stepResult{rc: rc, need: int(n)}
		}
		journal := (*[1 << 28]C.kb_stmt_op)(unsafe.Pointer(ops))[:int(n):int(n)]
		if step == "reclaim" {
			return stepResult{rc: C.KB_OK, failed: replayReclaimJournal(ssn, fl, journal)}
		}
		return stepResult{rc: C.KB_OK, failed: replayPreemptJournal(ssn, fl, journal)}
	}
}
