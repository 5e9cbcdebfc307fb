"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/kb_engine.h declares; no compute call is made (there is no GPU here)."""
import ctypes as C
import importlib
import os
import re

import pytest

kbm = importlib.import_module("kube-batch_amd")
engine = importlib.import_module("kube-batch_amd.engine")
abi = kbm.abi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    engine.build()
    return engine.lib()


def test_library_exports_every_header_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "kb_engine.h")).read()
    declared = set(re.findall(r"\b(kb_[a-z_]+)\s*\(", hdr))
    declared -= {"kb_engine"}
    assert declared == set(engine.EXPORTS), declared ^ set(engine.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_sizes_match_header():
    # include/kb_engine.h layouts (LP64): any drift between the header and the ctypes mirror shows up here
    assert C.sizeof(abi.PluginOption) == 4 + 4 + 32 + 4
    assert C.sizeof(abi.Decision) == 16
    assert C.sizeof(abi.Config) == 8 + 8 + 8 + 16
    assert C.sizeof(abi.Snapshot) == 32 + 8 * len(abi.SNAPSHOT_ARRAYS) + 8 + 8      # + the kb_interpod pointer + port_words, pad (ABI 8)
    assert C.sizeof(abi.Interpod) == 24 + 8 * len(abi.INTERPOD_ARRAYS)
    assert C.sizeof(abi.Stats) == 9 * 8 + 6 * 8 + 4 * 8          # + the selection kernel's four counters (round 4)


def test_create_without_gpu_fails_loudly(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    conf = kbm.conf.load_scheduler_conf()
    with pytest.raises(engine.EngineError) as ei:
        engine.Engine(conf)
    assert ei.value.code == abi.KB_E_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_weights_outside_the_u16_score_envelope_are_unsupported(L):
    """The matrix stores the weighted score as u16; every scorer (NodeAffinity after NormalizeReduce too) yields 0..10.  The policy
    check runs before any device call, so it answers without a GPU: such a conf must get KB_E_UNSUPPORTED (stock action), not a
    wrapped score."""
    tmpl = """
actions: "allocate"
tiers:
- plugins:
  - name: predicates
  - name: nodeorder
    arguments:
      leastrequested.weight: {wl}
      mostrequested.weight: {wm}
      balancedresource.weight: {wb}
      nodeaffinity.weight: {wa}
"""
    for wl, wm, wb, wa in ((3000, 3000, 500, 100), (7000, 0, 0, 1), (1, 0, 1, -1), (-1, 0, 1, 1), (0, 0, 0, 6554)):
        with pytest.raises(engine.EngineError) as ei:
            engine.Engine(kbm.conf.load_scheduler_conf(tmpl.format(wl=wl, wm=wm, wb=wb, wa=wa)))
        assert ei.value.code == abi.KB_E_UNSUPPORTED, (wl, wm, wb, wa)
    try:        # just inside the envelope: accepted by the policy check (then KB_E_DEVICE here, an engine on a GPU box)
        engine.Engine(kbm.conf.load_scheduler_conf(tmpl.format(wl=3000, wm=3000, wb=500, wa=53))).close()
    except engine.EngineError as e:
        assert e.code == abi.KB_E_DEVICE


def test_unknown_plugin_rejected():
    conf = kbm.conf.SchedulerConf(actions=["allocate"], tiers=[[kbm.conf.PluginOption("nosuch")]])
    with pytest.raises(ValueError):
        conf.to_abi()


def test_struct_layouts_match_the_header_as_gcc_sees_it(tmp_path):
    """Compile include/kb_engine.h with gcc and compare sizeof / offsetof of every field of the boundary structs with the ctypes
    mirror (kube-batch_amd/abi.py) — the Go side binds the same header through cgo, so the header is the authority."""
    import subprocess
    structs = {"kb_snapshot": abi.Snapshot, "kb_config": abi.Config, "kb_plugin_option": abi.PluginOption,
               "kb_decision": abi.Decision, "kb_stats": abi.Stats, "kb_interpod": abi.Interpod}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "kb_engine.h"', 'int main(void) {',
             '  printf("KB_ABI_VERSION %u\\n", (unsigned)KB_ABI_VERSION);']
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(got["KB_ABI_VERSION"]) == abi.KB_ABI_VERSION
    for cname, ct in structs.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"


def test_go_shim_sources_only_use_what_the_header_declares():
    """The Go action / flattener cannot be compiled here (no Go toolchain): at least every C.<name> they mention and every
    kb_snapshot field they assign must exist in include/kb_engine.h."""
    hdr = open(os.path.join(ROOT, "include", "kb_engine.h")).read()
    godir = os.path.join(ROOT, "integration", "go", "gpuallocate")
    for fn in sorted(os.listdir(godir)):
        src = open(os.path.join(godir, fn)).read()
        for name in set(re.findall(r"\bC\.((?:kb|KB)_[A-Za-z0-9_]+)", src)):
            assert re.search(r"\b" + re.escape(name) + r"\b", hdr), f"{fn}: C.{name} is not in the header"
        for field in set(re.findall(r"\bs\.((?:node|task|job|queue|class|n)_[a-z_]+)\b", src)):
            assert re.search(r"\b" + re.escape(field) + r"\b", hdr), f"{fn}: kb_snapshot.{field} is not in the header"
    # and the other way round: every snapshot array the header declares is filled by flatten.go (or documented as optional)
    flat = open(os.path.join(godir, "flatten.go")).read()
    for name, _ in abi.SNAPSHOT_ARRAYS:
        assert re.search(r"\bs\." + name + r"\b", flat), f"flatten.go never sets kb_snapshot.{name}"
    ipgo = open(os.path.join(godir, "interpod.go")).read()
    for name, _ in abi.Interpod._fields_:
        if name != "pad":
            assert re.search(r"\bip\." + name + r"\b", ipgo), f"interpod.go never sets kb_interpod.{name}"
    assert "f.snap.interpod = ip" in ipgo and "buildInterpod(" in flat


def test_go_shim_obeys_the_cgo_pointer_rules_statically():
    """No Go toolchain here, so the cgo rules the shim must obey are checked on the source text (kb_config used to
    point into Go slices, a 'Go pointer to Go pointer' panic under cgocheck=1; &decisions[0] panicked on an empty session):
      * no C struct field is ever assigned the address of a Go slice element or a Go variable;
      * no C.kb_* call receives &goSlice[0];
      * every array a C struct points at comes from C.calloc / C.malloc (or one of flatten.go's calloc-backed views);
      * the backfill pass is really replayed (kb_run_backfill is called, not merely mentioned in a comment)."""
    godir = os.path.join(ROOT, "integration", "go", "gpuallocate")
    for fn in sorted(os.listdir(godir)):
        code = []
        for line in open(os.path.join(godir, fn)).read().splitlines():
            code.append(re.sub(r"//.*$", "", line))          # comments may say anything
        src = "\n".join(code)
        # 1. struct-field = address-of-Go-slice-element / unsafe.Pointer(&x[...])
        #    (slices that ARE C memory are fine: flatten.go's calloc-backed views f.f64(n), f.u32(n), ...)
        cviews = set()
        for lhs, rhs in re.findall(r"^\s*([\w, ]+?)\s*:?=\s*(f\.(?:f64|i64|u64|u32|i32|u16|u8)\(.*)$", src, flags=re.M):
            names = [n.strip() for n in lhs.split(",")]
            if len(names) == len(re.findall(r"\bf\.(?:f64|i64|u64|u32|i32|u16|u8)\(", rhs)):
                cviews.update(names)
        taken = re.findall(r"^\s*[A-Za-z_][\w.]*\.[a-z_]+\s*=\s*\(\*C\.[\w]+\)\(unsafe\.Pointer\(&(\w+)\[", src, flags=re.M)
        bad = [v for v in taken if v not in cviews]
        assert not bad, f"{fn}: C struct field assigned the address of a Go slice element: {bad}"
        # 2. C.kb_* call with &slice[i] as an argument
        for call in re.findall(r"C\.kb_[a-z_]+\(([^\n]*)\)", src):
            assert not re.search(r"&\w+\[\d*\w*\]", call), f"{fn}: C.kb_* call takes a Go slice element address: {call}"
        # 3. no make([]C.kb_...) buffers handed to C
        assert not re.search(r"make\(\[\]C\.kb_", src), f"{fn}: C struct buffers must come from C.calloc"
    act = open(os.path.join(godir, "gpuallocate.go")).read()
    act_code = "\n".join(re.sub(r"//.*$", "", l) for l in act.splitlines())
    assert "C.kb_run_backfill(" in act_code and "fallbackBackfill" in act_code
    assert "len(fl.tasks) == 0" in act_code
    assert re.search(r"cfg\s*:=\s*\(\*C\.kb_config\)\(C\.calloc", act_code), "kb_config must be built in C memory"


def _go_code(src):
    """Go source with comments and string literals blanked (enough for the textual checks below)."""
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r'"(\\.|[^"\\\n])*"', '""', src)
    return re.sub(r"`[^`]*`", '""', src)


def test_go_shim_has_no_unused_imports():
    """`imported and not used` is a hard compile error in Go and nothing here compiles the shim: every import of every file must be
    referenced (as `name.`) in that file's code."""
    godir = os.path.join(ROOT, "integration", "go", "gpuallocate")
    for fn in sorted(os.listdir(godir)):
        src = open(os.path.join(godir, fn)).read()
        m = re.search(r"^import \((.*?)^\)", src, flags=re.S | re.M)
        assert m, fn
        code = _go_code(src[m.end():])
        for line in m.group(1).splitlines():
            mm = re.match(r'\s*(?:(\w+)\s+)?"([^"]+)"', line)
            if not mm:
                continue
            name = mm.group(1) or mm.group(2).rsplit("/", 1)[-1]
            assert re.search(r"\b" + re.escape(name) + r"\.", code), f"{fn}: import {mm.group(2)!r} is not used"


def test_go_shim_handles_empty_sessions_before_taking_slice_addresses():
    """&x[0] of an empty slice panics: flatten() must leave before its first &view[0] when the session has no task or no node,
    and every action must test len(fl.tasks) == 0 before it hands &fl.snap to the engine."""
    godir = os.path.join(ROOT, "integration", "go", "gpuallocate")
    flat = _go_code(open(os.path.join(godir, "flatten.go")).read())
    guard = re.search(r"if T == 0 \|\| N == 0 \{\s*return f, nil", flat)
    first_addr = re.search(r"unsafe\.Pointer\(&\w+\[0\]\)", flat)
    assert guard and first_addr and guard.start() < first_addr.start()
    for fn in ("gpuallocate.go", "gpupreempt.go", "cycle.go"):
        code = _go_code(open(os.path.join(godir, fn)).read())
        for body in re.split(r"\nfunc ", code):
            # the actions themselves; a helper that RE-loads an already loaded session (runJournal) is behind an action's check
            if "C.kb_session_load(" in body and re.match(r"\([^)]*\) Execute\(", body):
                assert "len(fl.tasks) == 0" in body and body.index("len(fl.tasks) == 0") < body.index("C.kb_session_load("), fn


def test_go_cycle_action_keeps_the_loaded_session_only_while_the_replay_was_clean():
    """cycle.go runs the engine actions of a cycle on ONE flatten + load (round-2 advisory: the single actions re-flatten per action).  Textual
    checks of what makes that safe: every replay helper reports refused entries, the cycle drops the loaded session after a refusal, after an
    engine error (the stock action then runs the step) and on KB_E_CAPACITY (same step again on a fresh load), and it calls all four actions."""
    godir = os.path.join(ROOT, "integration", "go", "gpuallocate")
    cyc = _go_code(open(os.path.join(godir, "cycle.go")).read())
    for call in ("C.kb_run_allocate(", "C.kb_run_backfill(", "C.kb_run_preempt(", "C.kb_run_reclaim(", "C.kb_session_load(", "flatten(ssn)"):
        assert call in cyc, call
    assert cyc.count("C.kb_session_load(") == 1                              # one place loads: the top of the step loop, behind `!loaded`
    body = cyc[cyc.index("func (c *gpuCycleAction) Execute("):cyc.index("func (c *gpuCycleAction) runStock(")]
    assert re.search(r"if !loaded \{.*?flatten\(ssn\).*?C\.kb_session_load\(.*?loaded = true", body, flags=re.S)
    assert re.search(r"res\.rc == C\.KB_E_CAPACITY.*?loaded = false\s*continue", body, flags=re.S)
    assert re.search(r"res\.rc != C\.KB_OK \{.*?c\.stock\[step\]\.Execute\(ssn\)\s*loaded = false", body, flags=re.S)
    assert re.search(r"res\.failed > 0 \{\s*loaded = false", body)
    act = _go_code(open(os.path.join(godir, "gpuallocate.go")).read())
    pre = _go_code(open(os.path.join(godir, "gpupreempt.go")).read())
    assert re.search(r"func \(a \*gpuAllocateAction\) replay\([^)]*\) int \{", act) and "failed++" in act
    for fn in ("replayPreemptJournal", "replayReclaimJournal"):
        assert re.search(r"func " + fn + r"\([^)]*\) int \{", pre), fn
        assert fn + "(" in cyc
    assert pre.count("failed++") >= 5


def test_integration_md_quotes_the_shipped_go_action():
    """INTEGRATION.md prints gpuallocate.go in full: the copy must be the file."""
    src = open(os.path.join(ROOT, "integration", "go", "gpuallocate", "gpuallocate.go")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = src[src.index("package gpuallocate"):].strip()
    assert body in doc, "INTEGRATION.md's listing of gpuallocate.go is out of date"
