"""The stamps that tie a committed rocprofv3 summary (profiles/roundN/rocprofv3_pmc_*.csv) to the device sources it was measured on
(scripts/kernel_sources_sha.py; bench.py quotes a summary only when the stamp of the kernels' translation unit equals the tree's).
No GPU, no engine: files and hashes only."""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_sources_sha as ks  # noqa: E402


def _copy_tree(tmp_path):
    for sub in (os.path.join("kube-batch_amd", "csrc"), "include"):
        shutil.copytree(os.path.join(ROOT, sub), os.path.join(tmp_path, sub))
    return str(tmp_path)


def _append(root, rel, text="\n// touched\n"):
    with open(os.path.join(root, rel), "a") as f:
        f.write(text)


def test_a_translation_unit_is_the_hip_file_and_the_headers_it_reaches():
    k = {os.path.basename(f) for f in ks.tu_files(ROOT, "kb_kernels.hip")}
    c = {os.path.basename(f) for f in ks.tu_files(ROOT, "kb_commit_sel.hip")}
    assert {"kb_kernels.hip", "kb_k1.hpp", "kb_eval.hpp", "kb_device.h", "kb_repair.hpp", "kb_warm.hpp", "Makefile"} <= k
    assert {"kb_commit_sel.hip", "kb_k9.hpp", "kb_repair.hpp", "kb_k1.hpp", "kb_eval.hpp", "kb_device.h", "Makefile"} <= c
    assert "kb_k9.hpp" not in k and "kb_commit_sel.hip" not in k and "kb_kernels.hip" not in c
    assert not any(f.endswith(".cpp") for f in k | c)          # host sources decide no kernel's ISA
    w = {os.path.basename(f) for f in ks.tu_files(ROOT, "kb_waterfill.hip")}
    assert {"kb_waterfill.hpp", "kb_res.hpp", "kb_engine.h"} <= w   # the C ABI header through kb_res.hpp


def test_a_change_voids_exactly_the_units_it_reaches(tmp_path):
    root = _copy_tree(tmp_path)
    before = {tu: ks.kernel_tu_sha(root, tu) for tu in ks.TUS}
    assert before == {tu: ks.kernel_tu_sha(ROOT, tu) for tu in ks.TUS}     # a copy hashes like the tree (paths are repo-relative)
    csrc = os.path.join("kube-batch_amd", "csrc")

    def changed():
        now = {tu: ks.kernel_tu_sha(root, tu) for tu in ks.TUS}
        return {tu for tu in ks.TUS if now[tu] != before[tu]}

    _append(root, os.path.join(csrc, "kb_engine.cpp"))
    assert changed() == set()                                               # the host side: every summary stays quotable
    _append(root, os.path.join(csrc, "kb_k9.hpp"))
    assert changed() == {"kb_commit_sel.hip", "kb_commit.hip"}              # the commit kernels' header: the matrix launches' summary survives
    _append(root, os.path.join(csrc, "kb_kernels.hip"))
    assert changed() == {"kb_commit_sel.hip", "kb_commit.hip", "kb_kernels.hip"}
    whole = ks.kernel_sources_sha(root)
    _append(root, os.path.join(csrc, "kb_eval.hpp"))                        # per-pair arithmetic: everything but the water-fill
    assert changed() == {"kb_commit_sel.hip", "kb_commit.hip", "kb_kernels.hip"} and ks.kernel_sources_sha(root) != whole
    _append(root, os.path.join(csrc, "Makefile"), "\n# flags\n")
    assert changed() == set(ks.TUS)


def test_stamp_file_round_trip(tmp_path):
    p = os.path.join(tmp_path, "kernel_tu.sha256")
    with open(p, "w") as f:
        f.write("# a comment\n" + "".join(f"{ks.kernel_tu_sha(ROOT, tu)}  {tu}\n" for tu in ks.TUS))
    assert ks.read_tu_stamp(p) == {tu: ks.kernel_tu_sha(ROOT, tu) for tu in ks.TUS}


def test_the_committed_summaries_were_measured_on_these_device_sources():
    """The newest profiles/roundN: the PMC summaries bench.py quotes (HBM bytes of the matrix launches, the commit kernel's counters) carry the
    stamps of THIS tree's translation units.  A change to a kernel without a new measurement makes the bench line say `traffic_refused` /
    `counters_refused` (tested above: bench.py never quotes a stale summary); this test says so first, as a SKIP with the reason — the tree is
    still correct, its profile evidence is one GPU call behind."""
    import glob
    import pytest
    rounds = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "kernel_tu.sha256")),
                    key=lambda f: int("".join(c for c in os.path.basename(os.path.dirname(f)) if c.isdigit()) or 0))
    assert rounds, "no profiles/roundN/kernel_tu.sha256"
    d = os.path.dirname(rounds[-1])
    stamp = ks.read_tu_stamp(os.path.join(d, "kernel_tu.sha256"))
    behind = []
    for csv_name, tu in (("rocprofv3_pmc_k_matrix.csv", "kb_kernels.hip"), ("rocprofv3_pmc_k_commit.csv", "kb_commit_sel.hip")):
        assert os.path.exists(os.path.join(d, csv_name))
        if stamp.get(tu) != ks.kernel_tu_sha(ROOT, tu):
            behind.append(f"{os.path.relpath(os.path.join(d, csv_name), ROOT)} was measured on another {tu}")
    if behind:
        pytest.skip("; ".join(behind) + " (bench.py reports traffic_refused / counters_refused until scripts/gpu_r6.sh profile has run on this tree)")
    assert open(os.path.join(d, "kernel_sources.sha256")).read().split()[0] == ks.kernel_sources_sha(ROOT)
