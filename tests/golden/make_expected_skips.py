#!/usr/bin/env python
"""Regenerates tests/golden/expected_skips.json: the differential cases that are skipped because the oracle reports a reference panic
or because the engine answers KB_E_UNSUPPORTED / KB_E_INVALID (its documented envelope).  tests/conftest.py turns
any OTHER skip of a differential case into a failure, so an envelope regression cannot hide as "more skips".  The envelope is host
logic: the list is recorded on the emulated device (tests/test_emu_engine_cpu.py re-collects every `-m gpu` module) and holds on the
MI355X, where each case runs once per commit kernel.   python tests/golden/make_expected_skips.py"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    with tempfile.NamedTemporaryFile("r", suffix=".txt") as f:
        env = dict(os.environ, KB_RECORD_SKIPS=f.name)
        subprocess.call([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_emu_engine_cpu.py"), "-q", "-p", "no:cacheprovider"], env=env, cwd=ROOT)
        rows = [line.rstrip("\n").split("\t", 1) for line in open(f.name)]
    skips = {}
    for key, reason in rows:
        reason = reason.replace("Skipped: ", "")
        skips[key] = "reference panics" if "panic" in reason else reason.split("KB_E_", 1)[-1][:150] if "KB_E_" in reason else reason[:150]
    out = {"note": "see make_expected_skips.py; keys are module::function[parameters without the commit-kernel axis]",
           "count": len(skips), "skips": dict(sorted(skips.items()))}
    with open(os.path.join(HERE, "expected_skips.json"), "w") as g:
        json.dump(out, g, indent=1)
        g.write("\n")
    print(len(skips), "expected skips")


if __name__ == "__main__":
    main()
