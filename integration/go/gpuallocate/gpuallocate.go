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
	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/api"
	"github.com/kubernetes-sigs/kube-batch/pkg/scheduler/framework"
)

type gpuAllocateAction struct {
	engine   *C.kb_engine      // one per process, created on first Execute from ssn.Tiers
	tiersKey string            // re-create the engine when the YAML tiers change
	fallback framework.Action  // the stock action, used when the engine says KB_E_UNSUPPORTED / KB_E_DEVICE
	backfill bool              // also run backfill.go's pass on the device
}

func New() *gpuAllocateAction { return &gpuAllocateAction{fallback: allocate.New(), backfill: false} }

func (a *gpuAllocateAction) Name() string  { return "gpuallocate" } // or "allocate" to override the stock action
func (a *gpuAllocateAction) Initialize()   {}
func (a *gpuAllocateAction) UnInitialize() { if a.engine != nil { C.kb_engine_destroy(a.engine); a.engine = nil } }

func (a *gpuAllocateAction) Execute(ssn *framework.Session) {
	runtime.LockOSThread() // one HIP context per OS thread is simplest; runOnce is single-threaded anyway (scheduler.go:85-101)
	defer runtime.UnlockOSThread()

	if err := a.ensureEngine(ssn); err != nil {          // conf.Tier / conf.PluginOption -> kb_config
		glog.Warningf("gpuallocate: %v; falling back to the stock allocate action", err)
		a.fallback.Execute(ssn)
		return
	}
	fl, err := flatten(ssn)                              // canonical order + SoA arrays in C memory (C.calloc), see flatten.go
	if err != nil {                                      // e.g. host ports / inter-pod affinity: not modelled by the engine
		glog.V(3).Infof("gpuallocate: %v; stock action takes this cycle", err)
		a.fallback.Execute(ssn)
		return
	}
	defer fl.free()

	if rc := C.kb_session_load(a.engine, &fl.snap); rc != C.KB_OK {
		glog.Warningf("gpuallocate: load rc=%d (%s); stock action takes this cycle", rc, C.GoString(C.kb_last_error(a.engine)))
		a.fallback.Execute(ssn)
		return
	}
	decisions := make([]C.kb_decision, len(fl.tasks))
	var n C.uint64_t
	rc := C.kb_run_allocate(a.engine, (*C.kb_decision)(unsafe.Pointer(&decisions[0])), C.uint64_t(len(decisions)), &n)
	if rc != C.KB_OK { // error conventions of SURVEY §8b: never abort; no decisions were applied, so the stock action is still valid
		glog.Warningf("gpuallocate: run rc=%d (%s); stock action takes this cycle", rc, C.GoString(C.kb_last_error(a.engine)))
		a.fallback.Execute(ssn)
		return
	}
	// Replay in the engine's order through the Session, exactly what allocate.go:160-183 does per task:
	// status index, node accounting, plugin event handlers and the gang-gated cache.Bind all run in the reference code.
	for i := 0; i < int(n); i++ {
		task, node := fl.tasks[decisions[i].task], fl.nodes[decisions[i].node]
		var err error
		if decisions[i].kind == 0 {
			err = ssn.Allocate(task, node.Name) // framework/session.go:235-288
		} else {
			err = ssn.Pipeline(task, node.Name) // framework/session.go:194-232
		}
		if err != nil {
			glog.Errorf("gpuallocate: replay of task %s on %s failed: %v", task.UID, node.Name, err)
		}
	}
	// (if a.backfill: C.kb_run_backfill + the same replay with ssn.Allocate, backfill.go:61)
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
	key := fmt.Sprintf("%+v", ssn.Tiers)
	if a.engine != nil && key == a.tiersKey {
		return nil
	}
	if a.engine != nil {
		C.kb_engine_destroy(a.engine)
		a.engine = nil
	}
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
	var cfg C.kb_config
	cfg.version = C.KB_ABI_VERSION
	cfg.n_tiers = C.uint32_t(len(ssn.Tiers))
	cfg.tier_begin = (*C.uint32_t)(unsafe.Pointer(&begin[0]))
	if len(opts) > 0 {
		cfg.plugins = (*C.kb_plugin_option)(unsafe.Pointer(&opts[0]))
	}
	cfg.device = 0
	if rc := C.kb_engine_create(&cfg, &a.engine); rc != C.KB_OK { // cfg is only read during the call
		return fmt.Errorf("kb_engine_create rc=%d: %s", int(rc), C.GoString(C.kb_last_error(nil)))
	}
	a.tiersKey = key
	return nil
}
