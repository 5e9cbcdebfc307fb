"""proportion's OnSessionOpen water-fill (plugins/proportion/proportion.go:101-154) as a launch — k_waterfill (kb_waterfill.hip), the default
since its first device run (round 4) — against the host loop it replaced (KB_DEVICE_WATERFILL=0, kb_session.cpp) on the MI355X: the same
`deserved` bit for bit, the same shares, decisions and refusals on tests/rawgen.py's adversarial snapshots — and, since round 6, the launch's
`deserved`, shares, decisions and node state against the ORACLE's directly in the same loop (the host loop is the engine's own) — and the
tutorial's known answer (doc/usage/tutorial.md:297-330).  The cases are tests/test_emu_engine_cpu.py's, here through the PRODUCT library; every other `-m gpu`
suite compares the launch's `deserved` and shares with the oracle as well (it is the default).  Needs a real MI355X: -m gpu."""
import importlib

import pytest

import test_emu_engine_cpu as emu_cases

engine = importlib.import_module("kube-batch_amd.engine")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(8))
def test_the_launch_equals_the_host_loop_on_adversarial_snapshots(monkeypatch, block):
    assert engine.LIB_PATH.endswith("libkbengine.so")
    emu_cases.test_the_device_waterfill_equals_the_host_loop_on_adversarial_snapshots(engine.LIB_PATH, monkeypatch, block)


def test_the_launch_on_the_tutorial_example_and_on_128_queues(monkeypatch):
    emu_cases.test_the_device_waterfill_on_the_tutorial_example_and_on_128_queues(engine.LIB_PATH, monkeypatch)
