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


@pytest.mark.parametrize("case", load("taint_cases.json"), ids=lambda c: c["name"][:60])
def test_taint_cases(case):
    """TaintToleration behaviour the reference holds outside unit tests: the two e2e scenarios
    (test/e2e/predicates/predicates_test.go:334-447) and the KWOK perf-test shapes — whole default plugin set."""
    import re
    o = orc.Oracle({"nodes": [case["node"]], "pods": [case["pod"]]})
    fits, plugin, msg = o.predicates(0, 0, orc.ALL, orc.ALL)
    assert fits == case["fits"], f"{case['source']}: plugin={plugin!r} msg={msg!r}"
    if not fits:
        assert plugin == case["plugin"] and re.match(case.get("message_regex", ".*taint.*"), msg), (plugin, msg)


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


# ---------------------------------------------------------------------------------------------------------------
# PodTopologySpread: PARITY UNPINNED in the reference (only instantiated with constraint-free pods,
# predicate_manager_test.go:120,2171). These cases restate the upstream documentation's own worked example and the
# rules of SURVEY.md A.6, so that the oracle's behaviour is at least written down and stable.
# ---------------------------------------------------------------------------------------------------------------
def _node(name, zone, pods_with_app=0, extra=None):
    labels = {"kubernetes.io/hostname": name}
    if zone is not None:
        labels["zone"] = zone
    if extra:
        labels.update(extra)
    pods = [{"metadata": {"name": f"{name}-{i}", "uid": f"{name}-{i}", "namespace": "default", "labels": {"foo": "bar"}}}
            for i in range(pods_with_app)]
    return {"metadata": {"name": name, "labels": labels}, "pods": pods}


def _spread_pod(max_skew=1, selector=None, labels=None, **kw):
    c = {"maxSkew": max_skew, "topologyKey": "zone", "whenUnsatisfiable": "DoNotSchedule"}
    if selector is not None:
        c["labelSelector"] = selector
    c.update(kw)
    return {"metadata": {"name": "mypod", "uid": "mypod", "namespace": "default", "labels": labels or {"foo": "bar"}},
            "spec": {"topologySpreadConstraints": [c]}}


SPREAD = orc.PLUGIN_BITS["PodTopologySpread"]


def test_spread_documentation_example():
    """kubernetes.io docs "Pod Topology Spread Constraints", example one constraint: zoneA holds 2 matching pods
    (node1, node2: 1 each), zoneB holds 1 (node3: 1, node4: 0); maxSkew=1 ⇒ the incoming pod may only land in zoneB."""
    nodes = [_node("node1", "zoneA", 1), _node("node2", "zoneA", 1), _node("node3", "zoneB", 1), _node("node4", "zoneB", 0)]
    o = orc.Oracle({"nodes": nodes, "pods": [_spread_pod(selector={"matchLabels": {"foo": "bar"}})]})
    assert [o.predicates(0, n, SPREAD, SPREAD)[0] for n in range(4)] == [False, False, True, True]
    o = orc.Oracle({"nodes": nodes, "pods": [_spread_pod(max_skew=2, selector={"matchLabels": {"foo": "bar"}})]})
    assert all(o.predicates(0, n, SPREAD, SPREAD)[0] for n in range(4))


def test_spread_rules():
    nodes = [_node("n1", "zoneA", 2), _node("n2", "zoneB", 0), _node("n3", None, 0)]
    sel = {"matchLabels": {"foo": "bar"}}
    # missing topology label on the candidate node ⇒ unresolvable failure; the node is not counted either
    o = orc.Oracle({"nodes": nodes, "pods": [_spread_pod(selector=sel)]})
    got = [o.predicates(0, n, SPREAD, SPREAD) for n in range(3)]
    assert [g[0] for g in got] == [False, True, False] and "missing required label" in got[2][2]
    # minDomains larger than the number of domains ⇒ global minimum treated as 0: zoneA (2+1-0 > 1) fails, zoneB (0+1-0) fits
    o = orc.Oracle({"nodes": nodes[:2], "pods": [_spread_pod(selector=sel, minDomains=5)]})
    assert [o.predicates(0, n, SPREAD, SPREAD)[0] for n in range(2)] == [False, True]
    # the pod does not match its own selector ⇒ selfMatch = 0: skew in zoneA = 2 - 0 = 2 > 1 still fails
    o = orc.Oracle({"nodes": nodes[:2], "pods": [_spread_pod(selector=sel, labels={"foo": "other"})]})
    assert [o.predicates(0, n, SPREAD, SPREAD)[0] for n in range(2)] == [False, True]
    # ScheduleAnyway constraints are not hard ⇒ PreFilter Skip ⇒ everything fits
    o = orc.Oracle({"nodes": nodes, "pods": [_spread_pod(selector=sel, whenUnsatisfiable="ScheduleAnyway")]})
    assert all(o.predicates(0, n, SPREAD, SPREAD)[0] for n in range(3))
    # nil / empty selectors count nothing; an empty selector still self-matches
    o = orc.Oracle({"nodes": nodes[:2], "pods": [_spread_pod(selector={}), _spread_pod()]})
    assert all(o.predicates(p, n, SPREAD, SPREAD)[0] for p in range(2) for n in range(2))
    # nodeAffinityPolicy=Honor (default): nodes outside the pod's nodeSelector are not counted, Ignore counts them
    pod = _spread_pod(selector=sel)
    pod["spec"]["nodeSelector"] = {"zone": "zoneB"}
    o = orc.Oracle({"nodes": nodes[:2], "pods": [pod]})
    assert o.predicates(0, 1, SPREAD, SPREAD)[0]  # only zoneB is a domain: min = 0, skew = 1
    pod2 = _spread_pod(selector=sel, nodeAffinityPolicy="Ignore", max_skew=1)
    pod2["spec"]["nodeSelector"] = {"zone": "zoneA"}
    o = orc.Oracle({"nodes": nodes[:2], "pods": [pod2]})
    assert not o.predicates(0, 0, SPREAD, SPREAD)[0]  # zoneA=2, zoneB=0 both counted: 2+1-0 > 1
    # Filter enabled without its PreFilter: Error status ⇒ does not fit
    o = orc.Oracle({"nodes": nodes[:2], "pods": [_spread_pod(selector=sel)]})
    assert not o.predicates(0, 1, 0, SPREAD)[0]


def test_node_resource_conversion():
    """TestNodeResource (pkg/common/resource_test.go:833-841): allocatable cpu "14500m" is 14500 milli-cores; memory and
    pods go through Value() (resource.go:273-285)."""
    node = {"metadata": {"name": "n"}, "status": {"allocatable": {"cpu": "14500m", "memory": "1Gi", "pods": "110"}}}
    info = orc.Oracle({"nodes": [node], "pods": []}).node_info(0)
    assert info["alloc"][:2] == [14500, 1 << 30] and info["allowed_pods"] == 110


def test_spread_state_is_kept_per_constraint():
    """Two hard constraints on the SAME topologyKey with different selectors (a shape API validation rejects, and the
    product's encoder refuses with that reason): the oracle restates k8s v1.36.1, where the PreFilter state is a slice
    indexed by constraint (TpValueToMatchNum) — up to 1.2x one map keyed by {topologyKey, value} was shared, the second
    constraint's per-node count overwrote the first's, and node n1 below would have been feasible."""
    def existing(uid, labels):
        return {"metadata": {"name": uid, "uid": uid, "namespace": "ns", "labels": labels}, "spec": {"containers": []}}
    nodes = [{"metadata": {"name": "n1", "labels": {"zone": "a"}}, "status": {"allocatable": {"pods": "10"}},
              "pods": [existing("e1", {"x": "1"}), existing("e2", {"x": "1"})]},
             {"metadata": {"name": "n2", "labels": {"zone": "b"}}, "status": {"allocatable": {"pods": "10"}},
              "pods": [existing("e3", {"y": "1"})]}]
    def constraint(key, value):
        return {"maxSkew": 1, "topologyKey": "zone", "whenUnsatisfiable": "DoNotSchedule", "labelSelector": {"matchLabels": {key: value}}}
    pod = {"metadata": {"name": "p", "uid": "p", "namespace": "ns", "labels": {"x": "1", "y": "1"}},
           "spec": {"containers": [], "topologySpreadConstraints": [constraint("x", "1"), constraint("y", "1")]}}
    o = orc.Oracle({"nodes": nodes, "pods": [pod]})
    # constraint 0: zone a holds 2 matches, zone b 0 → n1: 2 + 1 - 0 > 1; constraint 1: a 0, b 1 → n2: 1 + 1 - 0 > 1
    assert [o.predicates(0, n)[:2] for n in range(2)] == [(False, "PodTopologySpread"), (False, "PodTopologySpread")]
    # each constraint alone leaves the other zone's node feasible
    for keep, fits in ((0, [False, True]), (1, [True, False])):
        single = json.loads(json.dumps(pod))
        single["spec"]["topologySpreadConstraints"] = [pod["spec"]["topologySpreadConstraints"][keep]]
        o1 = orc.Oracle({"nodes": nodes, "pods": [single]})
        assert [o1.predicates(0, n)[0] for n in range(2)] == fits
    # the product refuses the two-constraint shape (ValidateTopologySpreadConstraints: duplicate {topologyKey, whenUnsatisfiable})
    import importlib
    m = importlib.import_module("yunikorn-k8shim_amd").GpuPredicateManager(device=-1)
    try:
        m.load_snapshot({"nodes": nodes, "pods": [pod]})
        ok, why = m.ask_supported(0)
        assert not ok and "duplicate topologyKey" in why
    finally:
        m.close()


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("allocate", [True, False])
def test_prefilter_once_form_equals_the_per_pair_form(seed, allocate):
    """orc_eval_rows runs a pod's PreFilter pass once for all its nodes; orc_eval_grid restates the reference line by line (a
    new CycleState and a full PreFilter pass per pair, predicate_manager.go:196,202,221-254). Same fit bit and same failing
    plugin for every pair — with hard spread constraints, inter-pod affinity, NodeNames PreFilter results and PreFilter
    rejections in the mix — is what lets the full-grid parity test afford BASELINE configs[4]."""
    import _gen
    snap = _gen.random_snapshot(9100 + seed, n_nodes=60 + 11 * seed, n_pods=50, spread=seed % 2 == 0, interpod=seed % 3 != 1)
    o = orc.Oracle(snap)
    pre, filt = (orc.ALL, orc.ALL) if allocate else (orc.RESERVE_PRE, orc.RESERVE_FILT)
    a, pa = o.eval_grid(pre_mask=pre, filt_mask=filt, threads=4, want_plugin=True)
    b, pb = o.eval_grid(pre_mask=pre, filt_mask=filt, threads=4, want_plugin=True, prefilter_once=True)
    assert (a == b).all() and (pa == pb).all()
    assert 0 < a.sum() < a.size
    for p in range(0, o.num_pods, 5):
        assert o.decide(p, pre, filt) == o.decide(p, pre, filt, prefilter_once=True)
    # plugin subsets: a Filter without its PreFilter (Error status), PreFilters alone
    for pre2, filt2 in ((0, orc.ALL), (orc.ALL, 0), (orc.PLUGIN_BITS["NodeAffinity"], orc.PLUGIN_BITS["NodeAffinity"] | orc.PLUGIN_BITS["PodTopologySpread"])):
        a, pa = o.eval_grid(pre_mask=pre2, filt_mask=filt2, threads=4, want_plugin=True)
        b, pb = o.eval_grid(pre_mask=pre2, filt_mask=filt2, threads=4, want_plugin=True, prefilter_once=True)
        assert (a == b).all() and (pa == pb).all()


@pytest.mark.parametrize("case", load("binpacking_cases.json"), ids=lambda c: c["name"])
def test_binpacking_node_order(case):
    """The reference's one behavioural pin of the DECISION ORDER (test/e2e/bin_packing/bin_packing_test.go:52-189, policy
    `binpacking`): padding pods stabilise the order, job A's three request-less pods all land on the node with the least
    available memory, job B's three pods — anti-affinity to that node's padding pod on kubernetes.io/hostname — all on the second.
    The oracle's sequential loop (orc_allocate_sequential: ascending bin-pack score, ties by NodeID, first fit, AssumePod) must
    place the eight asks exactly there — and must NOT under the reverse ("fair") order, so the pin really holds the direction."""
    import numpy as np
    o = orc.Oracle({"nodes": case["nodes"], "pods": case["pods"]})
    scores = o.binpack_scores()
    names = [n["metadata"]["name"] for n in case["nodes"]]
    got = o.allocate_sequential()
    assert [names[i] if i >= 0 else None for i in got] == case["expect"], case["source"]
    # the direction: the node the e2e calls nodeA is the one with the LOWEST score before any ask, and a most-available-first
    # walk would have put job A somewhere else
    assert names[int(np.argmin(scores))] == case["expect"][0]
    assert names[int(np.argmax(scores))] != case["expect"][2]
