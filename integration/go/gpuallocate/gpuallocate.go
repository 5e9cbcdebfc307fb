// gpuallocate.go — the kube-batch side of the drop-in boundary (SOURCE ONLY: there is no Go toolchain in the build image, so
// this package is not compiled here; the identical C ABI is exercised through kube-batch_amd/engine.py).
//
// It goes to pkg/scheduler/actions/gpuallocate/ of the reference tree and is registered next to the stock actions
// (actions/factory.go: framework.RegisterAction(gpuallocate.New())).  It replaces the Execute body of
// actions/allocate/allocate.go:43-194 (and, optionally, actions/backfill/backfill.go:40-71): flatten the Session into the SoA
// snapshot of include/kb_engine.h, run the action on the GPU, replay the ordered decisions through ssn.Allocate / ssn.Pipeline.
package gpuallocate

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../third_party/kbengine/include
#cgo LDFLAGS: -L${SRCDIR}/../../../../third_party/kbengine/lib -lkbengine -Wl,-rpath,$ORIGIN/../lib
#include <stdlib.h>
#include "kb_engine.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"strconv"
	"unsafe"

	"github.com/golang/glog"

	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/actions/allocate"
	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/actions/backfill"
	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/framework"
)

type gpuAllocateAction struct {
	engine   *C.kb_engine      // one per process, created on first Execute from ssn.Tiers
	tiersKey string            // re-create the engine when the YAML tiers change
	fallback framework.Action  // the stock action, used when the engine says KB_E_UNSUPPORTED / KB_E_DEVICE
	backfill bool              // also run backfill.go's pass on the device (then drop "backfill" from the YAML actions list)
	fallbackBackfill framework.Action
}

func New() *gpuAllocateAction {
	return &gpuAllocateAction{fallback: allocate.New(), backfill: false, fallbackBackfill: backfill.New()}
}

// NewWithBackfill registers as one action that stands in for "allocate, backfill"
func NewWithBackfill() *gpuAllocateAction {
	a := New()
	a.backfill = true
	return a
}

func (a *gpuAllocateAction) Name() string  { return "gpuallocate" } // or "allocate" to override the stock action
func (a *gpuAllocateAction) Initialize()   {}
func (a *gpuAllocateAction) UnInitialize() { if a.engine != nil { C.kb_engine_destroy(a.engine); a.engine = nil } }

// stock runs the reference actions this one stands in for
func (a *gpuAllocateAction) stock(ssn *framework.Session) {
	a.fallback.Execute(ssn)
	if a.backfill {
		a.fallbackBackfill.Execute(ssn)
	}
}

func (a *gpuAllocateAction) Execute(ssn *framework.Session) {
	runtime.LockOSThread() // one HIP context per OS thread is simplest; runOnce is single-threaded anyway (scheduler.go:85-101)
	defer runtime.UnlockOSThread()

	if err := a.ensureEngine(ssn); err != nil {          // conf.Tier / conf.PluginOption -> kb_config
		glog.Warningf("gpuallocate: %v; falling back to the stock allocate action", err)
		a.stock(ssn)
		return
	}
	fl, err := flatten(ssn)                              // canonical order + SoA arrays in C memory (C.calloc), see flatten.go
	if err != nil {                                      // e.g. a PVC pod in a session with inter-pod terms: outside the engine's envelope
		glog.V(3).Infof("gpuallocate: %v; stock action takes this cycle", err)
		a.stock(ssn)
		return
	}
	defer fl.free()

	if len(fl.tasks) == 0 { // idle cluster (or one without nodes): nothing to place; flatten built no arrays for it
		return
	}
	if rc := C.kb_session_load(a.engine, &fl.snap); rc != C.KB_OK {
		glog.Warningf("gpuallocate: load rc=%d (%s); stock action takes this cycle", rc, C.GoString(C.kb_last_error(a.engine)))
		a.stock(ssn)
		return
	}
	// The decision buffer lives in C memory: the engine fills it, Go only reads it (no Go pointer crosses the boundary).
	capDec := C.size_t(len(fl.tasks))
	decisions := (*C.kb_decision)(C.calloc(capDec, C.size_t(unsafe.Sizeof(C.kb_decision{}))))
	if decisions == nil {
		a.stock(ssn)
		return
	}
	defer C.free(unsafe.Pointer(decisions))
	dec := (*[1 << 28]C.kb_decision)(unsafe.Pointer(decisions))[:len(fl.tasks):len(fl.tasks)]

	var n C.uint64_t
	rc := C.kb_run_allocate(a.engine, decisions, C.uint64_t(capDec), &n)
	if rc != C.KB_OK { // error conventions of SURVEY §8b: never abort; no decisions were applied, so the stock action is still valid
		glog.Warningf("gpuallocate: run rc=%d (%s); stock action takes this cycle", rc, C.GoString(C.kb_last_error(a.engine)))
		a.stock(ssn)
		return
	}
	// Replay in the engine's order through the Session, exactly what allocate.go:160-183 does per task:
	// status index, node accounting, plugin event handlers and the gang-gated cache.Bind all run in the reference code.
	a.replay(ssn, fl, dec[:int(n)])

	if a.backfill {
		// backfill.go:40-71 on the device: BestEffort tasks (empty InitResreq) take the first node, in canonical node order,
		// that passes the plugin predicates against the state the allocate pass left behind; every decision is an ssn.Allocate
		// (backfill.go:61).  The engine's session already holds that state, so no second flatten / load is needed.
		rc = C.kb_run_backfill(a.engine, decisions, C.uint64_t(capDec), &n)
		if rc != C.KB_OK {
			// allocate's decisions are already applied to ssn; the stock backfill works from that Session state
			glog.Warningf("gpuallocate: backfill rc=%d (%s); stock backfill takes over", rc, C.GoString(C.kb_last_error(a.engine)))
			a.fallbackBackfill.Execute(ssn)
			return
		}
		a.replay(ssn, fl, dec[:int(n)])
	}
}

// replay applies the engine's ordered decisions through the Session (framework/session.go:194-288); it returns how many of them the
// Session refused (0: the Session now stands where the engine's own session stands — what cycle.go needs to know before it lets the next
// action run on the loaded session)
func (a *gpuAllocateAction) replay(ssn *framework.Session, fl *flat, dec []C.kb_decision) int {
	failed := 0
	for i := range dec {
		task, node := fl.tasks[dec[i].task], fl.nodes[dec[i].node]
		var err error
		if dec[i].kind == 0 {
			err = ssn.Allocate(task, node.Name) // framework/session.go:235-288
		} else {
			err = ssn.Pipeline(task, node.Name) // framework/session.go:194-232
		}
		if err != nil {
			glog.Errorf("gpuallocate: replay of task %s on %s failed: %v", task.UID, node.Name, err)
			failed++
		}
	}
	return failed
}

var pluginIDs = map[string]C.uint32_t{
	"priority": C.KB_PLUGIN_PRIORITY, "gang": C.KB_PLUGIN_GANG, "conformance": C.KB_PLUGIN_CONFORMANCE, "drf": C.KB_PLUGIN_DRF,
	"predicates": C.KB_PLUGIN_PREDICATES, "proportion": C.KB_PLUGIN_PROPORTION, "nodeorder": C.KB_PLUGIN_NODEORDER,
}

func bit(p *bool, b C.uint32_t) C.uint32_t { // nil -> disabled, exactly isEnabled (framework/session_plugins.go:372-374)
	if p != nil && *p {
		return b
	}
	return 0
}

// ensureEngine (re)creates the engine when ssn.Tiers changed: conf.Tier / conf.PluginOption (conf/scheduler_conf.go:27-56) -> kb_config
func (a *gpuAllocateAction) ensureEngine(ssn *framework.Session) error {
	var opts []C.kb_plugin_option
	begin := []C.uint32_t{0}
	for _, tier := range ssn.Tiers {
		for _, p := range tier.Plugins {
			id, ok := pluginIDs[p.Name]
			if !ok {
				return fmt.Errorf("plugin %q has no built-in policy in the engine", p.Name)
			}
			var o C.kb_plugin_option
			o.plugin = id
			o.enabled = bit(p.EnabledJobOrder, C.KB_EN_JOB_ORDER) | bit(p.EnabledJobReady, C.KB_EN_JOB_READY) |
				bit(p.EnabledJobPipelined, C.KB_EN_JOB_PIPELINED) | bit(p.EnabledTaskOrder, C.KB_EN_TASK_ORDER) |
				bit(p.EnabledPreemptable, C.KB_EN_PREEMPTABLE) | bit(p.EnabledReclaimable, C.KB_EN_RECLAIMABLE) |
				bit(p.EnabledQueueOrder, C.KB_EN_QUEUE_ORDER) | bit(p.EnabledPredicate, C.KB_EN_PREDICATE) |
				bit(p.EnabledNodeOrder, C.KB_EN_NODE_ORDER)
			setArg := func(slot int, name string) { // framework.Arguments.GetInt: absent or unparsable -> the default stays
				if raw, given := p.Arguments[name]; given {
					if v, err := strconv.Atoi(raw); err == nil {
						o.args[slot] = C.int32_t(v)
						o.args_set |= 1 << uint(slot)
					}
				}
			}
			if p.Name == "nodeorder" { // plugins/nodeorder/nodeorder.go:65-117
				setArg(C.KB_ARG_NODEORDER_LEAST, "leastrequested.weight")
				setArg(C.KB_ARG_NODEORDER_MOST, "mostrequested.weight")
				setArg(C.KB_ARG_NODEORDER_NODEAFF, "nodeaffinity.weight")
				setArg(C.KB_ARG_NODEORDER_PODAFF, "podaffinity.weight")
				setArg(C.KB_ARG_NODEORDER_BALANCED, "balancedresource.weight")
			}
			opts = append(opts, o)
		}
		begin = append(begin, C.uint32_t(len(opts)))
	}
	// The key is the compiled policy itself (plain integers), not a print of ssn.Tiers: PluginOption holds *bool fields, whose
	// addresses say nothing about their values.
	key := fmt.Sprint(begin, opts)
	if a.engine != nil && key == a.tiersKey {
		return nil
	}
	if a.engine != nil {
		C.kb_engine_destroy(a.engine)
		a.engine = nil
	}
	// kb_config points at two arrays.  They must live in C memory: passing &cfg with fields that point into Go slices is a
	// "Go pointer to Go pointer" and panics under the default cgocheck=1.  Same rule flatten.go follows for the snapshot.
	cBegin := (*C.uint32_t)(C.calloc(C.size_t(len(begin)), 4))
	if cBegin == nil {
		return fmt.Errorf("out of memory")
	}
	defer C.free(unsafe.Pointer(cBegin))
	beginView := (*[1 << 20]C.uint32_t)(unsafe.Pointer(cBegin))[:len(begin):len(begin)]
	copy(beginView, begin)
	var cOpts *C.kb_plugin_option
	if len(opts) > 0 {
		cOpts = (*C.kb_plugin_option)(C.calloc(C.size_t(len(opts)), C.size_t(unsafe.Sizeof(C.kb_plugin_option{}))))
		if cOpts == nil {
			return fmt.Errorf("out of memory")
		}
		defer C.free(unsafe.Pointer(cOpts))
		optsView := (*[1 << 20]C.kb_plugin_option)(unsafe.Pointer(cOpts))[:len(opts):len(opts)]
		copy(optsView, opts)
	}
	cfg := (*C.kb_config)(C.calloc(1, C.size_t(unsafe.Sizeof(C.kb_config{}))))
	if cfg == nil {
		return fmt.Errorf("out of memory")
	}
	defer C.free(unsafe.Pointer(cfg))
	cfg.version = C.KB_ABI_VERSION
	cfg.n_tiers = C.uint32_t(len(ssn.Tiers))
	cfg.tier_begin = cBegin
	cfg.plugins = cOpts
	cfg.device = 0
	if rc := C.kb_engine_create(cfg, &a.engine); rc != C.KB_OK { // cfg and its arrays are only read during the call
		a.engine = nil
		return fmt.Errorf("kb_engine_create rc=%d: %s", int(rc), C.GoString(C.kb_last_error(nil)))
	}
	a.tiersKey = key
	return nil
}
