"""ctypes mirror of include/kb_engine.h (the C ABI of the engine).

Only plain C types cross the boundary; the same structs are what the Go shim fills through cgo
(INTEGRATION.md).  Field order and types must match include/kb_engine.h exactly.
"""
import ctypes as C

KB_ABI_VERSION = 8
KB_MAX_RES = 32
KB_NONE = 0xFFFFFFFF

KB_OK = 0
KB_E_INVALID = -1
KB_E_UNSUPPORTED = -2
KB_E_DEVICE = -3
KB_E_NOMEM = -4
KB_E_STATE = -5
KB_E_CAPACITY = -6
KB_E_INTERNAL = -7
ERR_NAMES = {0: "KB_OK", -1: "KB_E_INVALID", -2: "KB_E_UNSUPPORTED", -3: "KB_E_DEVICE", -4: "KB_E_NOMEM",
             -5: "KB_E_STATE", -6: "KB_E_CAPACITY", -7: "KB_E_INTERNAL"}

# task status, pkg/scheduler/api/types.go:27-61
(TASK_PENDING, TASK_ALLOCATED, TASK_PIPELINED, TASK_BINDING, TASK_BOUND, TASK_RUNNING, TASK_RELEASING,
 TASK_SUCCEEDED, TASK_FAILED, TASK_UNKNOWN) = range(10)

# plugins, pkg/scheduler/plugins/factory.go:31-42
PLUGIN_IDS = {"priority": 0, "gang": 1, "conformance": 2, "drf": 3, "predicates": 4, "proportion": 5, "nodeorder": 6}

EN_JOB_ORDER = 1 << 0
EN_JOB_READY = 1 << 1
EN_JOB_PIPELINED = 1 << 2
EN_TASK_ORDER = 1 << 3
EN_PREEMPTABLE = 1 << 4
EN_RECLAIMABLE = 1 << 5
EN_QUEUE_ORDER = 1 << 6
EN_PREDICATE = 1 << 7
EN_NODE_ORDER = 1 << 8
EN_ALL = 0x1FF

FLAG_SYNC_ROUNDS = 1
FLAG_YIELD_WAIT = 2     # round 5's opt-in, the default since round 6: a short spin, then the host gives its core up between two polls (include/kb_engine.h)
FLAG_SPIN_WAIT = 4      # spin for the whole wait (round 5's default)

# kb_stmt_op.op: the preempt action's journal (framework/statement.go)
OP_EVICT, OP_PIPELINE, OP_COMMIT, OP_DISCARD = range(4)
MATRIX_DIRECT, MATRIX_NO_DEDUP = 0x100, 0x200   # KB_MATRIX_* (OR'ed into fit_mode of kb_eval_matrix / kb_bench_matrix)


class PluginOption(C.Structure):
    _fields_ = [("plugin", C.c_uint32), ("enabled", C.c_uint32), ("args", C.c_int32 * 8), ("args_set", C.c_uint32)]


class StmtOp(C.Structure):
    _fields_ = [("op", C.c_uint32), ("task", C.c_uint32), ("node", C.c_uint32), ("stmt", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [("version", C.c_uint32), ("n_tiers", C.c_uint32),
                ("tier_begin", C.POINTER(C.c_uint32)), ("plugins", C.POINTER(PluginOption)),
                ("device", C.c_int32), ("window", C.c_uint32), ("commit_batch", C.c_uint32), ("flags", C.c_uint32)]


_P = C.POINTER
SNAPSHOT_ARRAYS = [
    ("node_idle", C.c_double), ("node_releasing", C.c_double), ("node_allocatable", C.c_double),
    ("node_scalar_mask", C.c_uint32), ("node_alloc_cpu", C.c_int64), ("node_alloc_mem", C.c_int64),
    ("node_nz_cpu", C.c_int64), ("node_nz_mem", C.c_int64), ("node_max_pods", C.c_int32),
    ("node_pod_cnt", C.c_int32), ("node_class", C.c_uint32),
    ("task_resreq", C.c_double), ("task_init_resreq", C.c_double), ("task_scalar_mask", C.c_uint32),
    ("task_nz_cpu", C.c_int64), ("task_nz_mem", C.c_int64), ("task_job", C.c_uint32), ("task_class", C.c_uint32),
    ("task_priority", C.c_int32), ("task_creation", C.c_int64), ("task_status", C.c_uint8), ("task_node", C.c_uint32),
    ("job_task_begin", C.c_uint32), ("job_queue", C.c_uint32), ("job_min_available", C.c_int32),
    ("job_priority", C.c_int32), ("job_creation", C.c_int64),
    ("queue_weight", C.c_int32), ("queue_creation", C.c_int64),
    ("class_compat", C.c_uint8), ("class_affinity", C.c_int32),
    ("node_ports", C.c_uint64), ("task_port_want", C.c_uint64), ("task_port_conflict", C.c_uint64),
    ("task_evict_protected", C.c_uint8),
]


# kb_interpod: inter-pod (anti)affinity tables (predicate p8 + nodeorder's InterPodAffinityPriority), NULL when no pod has a term
INTERPOD_ARRAYS = [
    ("ctr_dom", C.c_uint32), ("ctr_count", C.c_int32), ("ctr_total", C.c_int32),
    ("task_inc", C.c_uint64), ("task_forbid", C.c_uint64), ("task_require", C.c_uint16), ("task_self", C.c_uint8),
    ("cls_dom", C.c_uint32), ("cls_bound", C.c_int32), ("cls_unbound", C.c_int32),
    ("task_cls_inc", C.c_uint64), ("task_sig", C.c_uint32), ("sig_weight", C.c_int32),
]


class Interpod(C.Structure):
    _fields_ = [("n_counters", C.c_uint32), ("n_domains", C.c_uint32), ("n_classes", C.c_uint32), ("n_sigs", C.c_uint32),
                ("first_unbound_node", C.c_uint32), ("pad", C.c_uint32)] + [(n, _P(t)) for n, t in INTERPOD_ARRAYS]


class Snapshot(C.Structure):
    _fields_ = [("version", C.c_uint32), ("n_res", C.c_uint32), ("n_nodes", C.c_uint32), ("n_tasks", C.c_uint32),
                ("n_jobs", C.c_uint32), ("n_queues", C.c_uint32), ("n_task_classes", C.c_uint32),
                ("n_node_classes", C.c_uint32)] + [(n, _P(t)) for n, t in SNAPSHOT_ARRAYS] + [("interpod", _P(Interpod)),
                                                                                        ("port_words", C.c_uint32), ("pad", C.c_uint32)]


class Decision(C.Structure):
    _fields_ = [("task", C.c_uint32), ("node", C.c_uint32), ("kind", C.c_uint32), ("round", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("evals", C.c_uint64), ("tasks_popped", C.c_uint64), ("decisions", C.c_uint64), ("binds", C.c_uint64),
                ("rounds", C.c_uint64), ("spec_breaks", C.c_uint64), ("row_fallbacks", C.c_uint64),
                ("matrix_launches", C.c_uint64), ("matrix_evals", C.c_uint64),
                ("matrix_ms", C.c_double), ("argmax_ms", C.c_double), ("commit_ms", C.c_double),
                ("reduce_ms", C.c_double), ("host_order_ms", C.c_double), ("total_ms", C.c_double),
                ("rounds_select", C.c_uint64), ("select_runs_clean", C.c_uint64), ("select_runs_shots", C.c_uint64), ("select_shots", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}
