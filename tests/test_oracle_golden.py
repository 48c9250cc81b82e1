"""Pins the CPU oracle against every golden vector the reference's own tests hold for the predicate hot path
(SURVEY.md §8c / Appendix B). Fixtures: tests/golden/*.json, transcribed by tests/golden/make_golden.py."""
import json
import os

import pytest

import _oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


PRED = load("predicate_cases.json")


@pytest.mark.parametrize("case", PRED, ids=[f"{c['test']}:{c['name']}" for c in PRED])
def test_predicate_cases(case):
    # newPredicateManagerInternal(handle, ep, ep, ep, ep): the same plugin set for all four lists
    # (predicate_manager_test.go:341), so the phase flag changes nothing in these tests.
    mask = orc.mask_of(case["plugins"])
    o = orc.Oracle({"nodes": [case["node"]], "pods": [case["pod"]]})
    fits, plugin, msg = o.predicates(0, 0, mask, mask)
    assert fits == case["fits"], f"{case['source']}: plugin={plugin!r} msg={msg!r}"
    if not fits:
        assert msg, "a failing Predicates() call carries a message (predicate_manager.go:210,216)"


@pytest.mark.parametrize("case", load("preemption_cases.json"), ids=lambda c: c["source"])
def test_preemption_cases(case):
    mask = orc.mask_of(case["plugins"])
    o = orc.Oracle({"nodes": [case["node"]], "pods": [case["pod"]]})
    assert o.preemption(0, 0, case["victims"], case["start_index"], mask, mask) == case["index"], case["source"]


@pytest.mark.parametrize("case", load("request_cases.json"), ids=lambda c: c["name"])
def test_request_vectors(case):
    o = orc.Oracle({"nodes": [], "pods": [case["pod"]]})
    got = {k: v for k, v in o.pod_request(0).items() if v != 0 or k in case["expect"]}
    assert got == case["expect"], case["source"]


@pytest.mark.parametrize("case", load("quantity_cases.json"), ids=lambda c: c["text"])
def test_quantities(case):
    L = orc.lib()
    assert L.orc_quantity_value(case["text"].encode()) == case["value"]
    assert L.orc_quantity_milli(case["text"].encode()) == case["milli"]


def test_default_manager_phases():
    """NewPredicateManager's phase lists (predicate_manager.go:321-373): reservation skips NodeResourcesFit."""
    node = {"metadata": {"name": "n0"}, "status": {"allocatable": {"cpu": "1", "memory": "1Gi", "pods": "10"}}}
    pod = {"metadata": {"name": "p"}, "spec": {"containers": [{"resources": {"requests": {"cpu": "2"}}}]}}
    o = orc.Oracle({"nodes": [node], "pods": [pod]})
    fits, plugin, msg = o.predicates(0, 0, orc.ALL, orc.ALL)
    assert not fits and plugin == "NodeResourcesFit" and "Insufficient cpu" in msg
    fits, plugin, _ = o.predicates(0, 0, orc.RESERVE_PRE, orc.RESERVE_FILT)
    assert fits and plugin == ""


def test_first_failing_plugin_order():
    """Filter order of predicate_manager.go:339-352: NodeUnschedulable before NodeName before TaintToleration..."""
    node = {"metadata": {"name": "n0"}, "spec": {"unschedulable": True, "taints": [{"key": "k", "value": "v", "effect": "NoSchedule"}]},
            "status": {"allocatable": {"cpu": "1", "memory": "1Gi", "pods": "10"}}}
    pod = {"metadata": {"name": "p"}, "spec": {"nodeName": "other", "containers": [{"resources": {"requests": {"cpu": "2"}}}]}}
    o = orc.Oracle({"nodes": [node], "pods": [pod]})
    assert o.predicates(0, 0)[1] == "NodeUnschedulable"
    nm = orc.ALL & ~orc.PLUGIN_BITS["NodeUnschedulable"]
    assert o.predicates(0, 0, nm, nm)[1] == "NodeName"
    nm &= ~orc.PLUGIN_BITS["NodeName"]
    fits, plugin, msg = o.predicates(0, 0, nm, nm)
    assert plugin == "TaintToleration" and "taint" in msg  # e2e regex `.*taint.*`, test/e2e/predicates/predicates_test.go:439
    nm &= ~orc.PLUGIN_BITS["TaintToleration"]
    assert o.predicates(0, 0, nm, nm)[1] == "NodeResourcesFit"
