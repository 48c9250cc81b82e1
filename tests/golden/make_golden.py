#!/usr/bin/env python3
"""Transcribes the reference's own table tests for the predicate hot path into tests/golden/*.json.

The reference is Go and cannot be executed in this image (no Go toolchain, k8s.io modules not vendored), so
these fixtures are a hand transcription of the *data* in the reference's tests — pod spec, node, enabled plugin
set, phase, expected result — each entry citing the `file:line` it was read from (paths relative to
/root/reference). Run `python tests/golden/make_golden.py` to regenerate the JSON files next to this script.

Every pod / node is written in Kubernetes JSON field names (what `json.Marshal(&v1.Pod{...})` would emit for the
fields the path reads), which is also the snapshot format the oracle and the host library parse.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PM = "pkg/plugin/predicates/predicate_manager_test.go"
RT = "pkg/common/resource_test.go"


def affinity(terms):
    return {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": terms}}}


def expr(key, op, values=None):
    r = {"key": key, "operator": op}
    if values is not None:
        r["values"] = values
    return r


def pod(spec=None, **meta):
    p = {"metadata": meta or {}}
    if spec is not None:
        p["spec"] = spec
    return p


def node(name="", labels=None, alloc=None, pods=None, unschedulable=False, taints=None):
    n = {"metadata": {"name": name}}
    if labels is not None:
        n["metadata"]["labels"] = labels
    if alloc is not None:
        n["status"] = {"allocatable": alloc}
    spec = {}
    if unschedulable:
        spec["unschedulable"] = True
    if taints:
        spec["taints"] = taints
    if spec:
        n["spec"] = spec
    if pods:
        n["pods"] = pods
    return n


def resource_pod(milli_cpu=0, memory=0, **meta):
    """newResourcePod (predicate_manager_test.go:1032-1060): one container, only the non-zero requests."""
    req = {}
    if milli_cpu > 0:
        req["cpu"] = f"{milli_cpu}m"
    if memory > 0:
        req["memory"] = str(memory)
    return pod({"containers": [{"resources": {"requests": req}}]}, **meta)


def make_resources(milli_cpu, memory, pods):
    """makeResources (predicate_manager_test.go:1062-1071) with the zero extended/storage/hugepage entries."""
    return {"cpu": f"{milli_cpu}m", "memory": str(memory), "pods": str(pods), "example.com/aaa": "0",
            "ephemeral-storage": "0", "hugepages-2Mi": "0"}


MATCH_NAME_NODE1 = expr("metadata.name", "In", ["node_1"])

# ---------------------------------------------------------------------------------------------------------
# TestPodFitsSelector — predicate_manager_test.go:338-1030; plugins NodePorts + NodeAffinity (:339); allocate
# phase (:1024); node = {name: nodeName, labels} with no allocatable (:1017-1022).
# ---------------------------------------------------------------------------------------------------------
SELECTOR = [
    (351, "no selector", pod(), None, "", True),
    (356, "missing labels", pod({"nodeSelector": {"foo": "bar"}}), None, "", False),
    (367, "same labels", pod({"nodeSelector": {"foo": "bar"}}), {"foo": "bar"}, "", True),
    (381, "node labels are superset", pod({"nodeSelector": {"foo": "bar"}}), {"foo": "bar", "baz": "blah"}, "", True),
    (396, "node labels are subset", pod({"nodeSelector": {"foo": "bar", "baz": "blah"}}), {"foo": "bar"}, "", False),
    (411, "matchExpressions In matches", pod({"affinity": affinity([{"matchExpressions": [expr("foo", "In", ["bar", "value2"])]}])}),
     {"foo": "bar"}, "", True),
    (439, "matchExpressions Gt matches", pod({"affinity": affinity([{"matchExpressions": [expr("kernel-version", "Gt", ["0204"])]}])}),
     {"kernel-version": "0206"}, "", True),
    (468, "matchExpressions NotIn matches", pod({"affinity": affinity([{"matchExpressions": [expr("mem-type", "NotIn", ["DDR", "DDR2"])]}])}),
     {"mem-type": "DDR3"}, "", True),
    (496, "matchExpressions Exists matches", pod({"affinity": affinity([{"matchExpressions": [expr("GPU", "Exists")]}])}),
     {"GPU": "NVIDIA-GRID-K1"}, "", True),
    (523, "affinity that doesn't match node's labels", pod({"affinity": affinity([{"matchExpressions": [expr("foo", "In", ["value1", "value2"])]}])}),
     {"foo": "bar"}, "", False),
    (551, "nil []NodeSelectorTerm", pod({"affinity": affinity(None)}), {"foo": "bar"}, "", False),
    (569, "empty []NodeSelectorTerm", pod({"affinity": affinity([])}), {"foo": "bar"}, "", False),
    (587, "empty MatchExpressions", pod({"affinity": affinity([{"matchExpressions": []}])}), {"foo": "bar"}, "", False),
    (609, "no Affinity", pod(), {"foo": "bar"}, "", True),
    (617, "Affinity but nil NodeSelector",
     pod({"affinity": {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": None}}}), {"foo": "bar"}, "", True),
    (633, "multiple matchExpressions ANDed, match",
     pod({"affinity": affinity([{"matchExpressions": [expr("GPU", "Exists"), expr("GPU", "NotIn", ["AMD", "INTER"])]}])}),
     {"GPU": "NVIDIA-GRID-K1"}, "", True),
    (664, "multiple matchExpressions ANDed, no match",
     pod({"affinity": affinity([{"matchExpressions": [expr("GPU", "Exists"), expr("GPU", "In", ["AMD", "INTER"])]}])}),
     {"GPU": "NVIDIA-GRID-K1"}, "", False),
    (695, "multiple NodeSelectorTerms ORed",
     pod({"affinity": affinity([{"matchExpressions": [expr("foo", "In", ["bar", "value2"])]},
                                {"matchExpressions": [expr("diffkey", "In", ["wrong", "value2"])]}])}),
     {"foo": "bar"}, "", True),
    (732, "Affinity and nodeSelector both satisfied",
     pod({"nodeSelector": {"foo": "bar"}, "affinity": affinity([{"matchExpressions": [expr("foo", "Exists")]}])}),
     {"foo": "bar"}, "", True),
    (763, "Affinity matches but nodeSelector does not",
     pod({"nodeSelector": {"foo": "bar"}, "affinity": affinity([{"matchExpressions": [expr("foo", "Exists")]}])}),
     {"foo": "barrrrrr"}, "", False),
    (794, "invalid value in Affinity term",
     pod({"affinity": affinity([{"matchExpressions": [expr("foo", "NotIn", ["invalid value: ___@#$%^"])]}])}),
     {"foo": "bar"}, "", False),
    (822, "matchFields In matches", pod({"affinity": affinity([{"matchFields": [MATCH_NAME_NODE1]}])}), None, "node_1", True),
    (848, "matchFields In does not match", pod({"affinity": affinity([{"matchFields": [MATCH_NAME_NODE1]}])}), None, "node_2", False),
    (874, "two terms: matchFields miss, matchExpressions hit",
     pod({"affinity": affinity([{"matchFields": [MATCH_NAME_NODE1]}, {"matchExpressions": [expr("foo", "In", ["bar"])]}])}),
     {"foo": "bar"}, "node_2", True),
    (910, "one term: matchFields miss AND matchExpressions hit",
     pod({"affinity": affinity([{"matchFields": [MATCH_NAME_NODE1], "matchExpressions": [expr("foo", "In", ["bar"])]}])}),
     {"foo": "bar"}, "node_2", False),
    (944, "one term: both match",
     pod({"affinity": affinity([{"matchFields": [MATCH_NAME_NODE1], "matchExpressions": [expr("foo", "In", ["bar"])]}])}),
     {"foo": "bar"}, "node_1", True),
    (978, "two terms: both miss",
     pod({"affinity": affinity([{"matchFields": [MATCH_NAME_NODE1]}, {"matchExpressions": [expr("foo", "In", ["not-match-to-bar"])]}])}),
     {"foo": "bar"}, "node_2", False),
]


def port_pod(host, *infos, **meta):
    """newPod(host, "PROTO/IP/PORT", ...) of predicate_manager_test.go:199-222."""
    ports = []
    for info in infos:
        proto, ip, port = info.split("/")
        ports.append({"hostIP": ip, "hostPort": int(port), "protocol": proto})
    return pod({"nodeName": host, "containers": [{"ports": ports}]}, **meta)


# TestPodFitsHostPorts — :224-335, plugin NodePorts (:225); NodeInfo without a Node object, only its pods (:237-321)
HOST_PORTS = [
    (235, "nothing running", pod(), [], True),
    (241, "other port", port_pod("m1", "UDP/127.0.0.1/8080"), ["UDP/127.0.0.1/9090"], True),
    (248, "same udp port", port_pod("m1", "UDP/127.0.0.1/8080"), ["UDP/127.0.0.1/8080"], False),
    (255, "same tcp port", port_pod("m1", "TCP/127.0.0.1/8080"), ["TCP/127.0.0.1/8080"], False),
    (262, "different host ip", port_pod("m1", "TCP/127.0.0.1/8080"), ["TCP/127.0.0.2/8080"], True),
    (269, "different protocol", port_pod("m1", "UDP/127.0.0.1/8080"), ["TCP/127.0.0.1/8080"], True),
    (276, "second udp port conflict", port_pod("m1", "UDP/127.0.0.1/8000", "UDP/127.0.0.1/8080"), ["UDP/127.0.0.1/8080"], False),
    (283, "first tcp port conflict", port_pod("m1", "TCP/127.0.0.1/8001", "UDP/127.0.0.1/8080"),
     ["TCP/127.0.0.1/8001", "UDP/127.0.0.1/8081"], False),
    (290, "first tcp port conflict due to 0.0.0.0 hostIP", port_pod("m1", "TCP/0.0.0.0/8001"), ["TCP/127.0.0.1/8001"], False),
    (297, "TCP hostPort conflict due to 0.0.0.0 hostIP", port_pod("m1", "TCP/10.0.10.10/8001", "TCP/0.0.0.0/8001"),
     ["TCP/127.0.0.1/8001"], False),
    (304, "second tcp port conflict to 0.0.0.0 hostIP", port_pod("m1", "TCP/127.0.0.1/8001"), ["TCP/0.0.0.0/8001"], False),
    (311, "second different protocol", port_pod("m1", "UDP/127.0.0.1/8001"), ["TCP/0.0.0.0/8001"], True),
    (318, "UDP hostPort conflict due to 0.0.0.0 hostIP", port_pod("m1", "UDP/127.0.0.1/8001"),
     ["TCP/0.0.0.0/8001", "UDP/0.0.0.0/8001"], False),
]


# ---------------------------------------------------------------------------------------------------------
# TestInterPodAffinity — predicate_manager_test.go:1171-2113; plugins InterPodAffinity + NodeAffinity (:1172); the node is
# machine1 {region: r1, zone: z11} (:1177-1182) and only the listed pods whose nodeName is machine1 are on it (:2097-2107).
# ---------------------------------------------------------------------------------------------------------
def sel(*exprs):
    return {"matchExpressions": list(exprs)}


def pterm(selector, key=None, namespaces=None):
    t = {"labelSelector": selector}
    if key is not None:
        t["topologyKey"] = key
    if namespaces is not None:
        t["namespaces"] = namespaces
    return t


def aff_spec(affinity=None, anti=None, node_name=None):
    a = {}
    if affinity is not None:
        a["podAffinity"] = {"requiredDuringSchedulingIgnoredDuringExecution": affinity}
    if anti is not None:
        a["podAntiAffinity"] = {"requiredDuringSchedulingIgnoredDuringExecution": anti}
    spec = {"affinity": a} if a else {}
    if node_name:
        spec["nodeName"] = node_name
    return spec


POD_LABEL = {"service": "securityscan"}
POD_LABEL2 = {"security": "S1"}
IN_SCAN = expr("service", "In", ["securityscan", "value2"])
IN_ANTIVIRUS = expr("service", "In", ["antivirusscan", "value2"])
NOTIN_SCAN = expr("service", "NotIn", ["securityscan", "value2"])
EX_SERVICE, EX_SECURITY = expr("service", "Exists"), expr("security", "Exists")
EX_ABC, EX_DEF = expr("abc", "Exists"), expr("def", "Exists")


def existing(labels, spec=None, ns=None, uid="existing"):
    meta = {"labels": labels, "uid": uid}
    if ns:
        meta["namespace"] = ns
    return {"metadata": meta, "spec": dict(spec or {}, nodeName="machine1")}


INTERPOD = [
    (1194, "no required pod affinity rules, no existing pods", pod(), [], True),
    (1225, "PodAffinity In matches the existing pod", pod(aff_spec([pterm(sel(IN_SCAN), "region")]), labels=POD_LABEL2),
     [existing(POD_LABEL)], True),
    (1256, "PodAffinity NotIn matches the existing pod",
     pod(aff_spec([pterm(sel(expr("service", "NotIn", ["securityscan3", "value3"])), "region")]), labels=POD_LABEL2),
     [existing(POD_LABEL)], True),
    (1287, "PodAffinity: different namespace", pod(aff_spec([pterm(sel(IN_SCAN), None, ["DiffNameSpace"])]), labels=POD_LABEL2),
     [existing(POD_LABEL, ns="ns")], False),
    (1317, "PodAffinity: unmatching labelSelector", pod(aff_spec([pterm(sel(IN_ANTIVIRUS))]), labels=POD_LABEL),
     [existing(POD_LABEL)], False),
    (1365, "PodAffinity: different operators in multiple terms",
     pod(aff_spec([pterm(sel(EX_SERVICE, expr("wrongkey", "DoesNotExist")), "region"),
                   pterm(sel(expr("service", "In", ["securityscan"]), expr("service", "NotIn", ["WrongValue"])), "region")]),
         labels=POD_LABEL2), [existing(POD_LABEL)], True),
    (1413, "PodAffinity: matchExpressions are ANDed, one does not match",
     pod(aff_spec([pterm(sel(EX_SERVICE, expr("wrongkey", "DoesNotExist")), "region"),
                   pterm(sel(expr("service", "In", ["securityscan2"]), expr("service", "NotIn", ["WrongValue"])), "region")]),
         labels=POD_LABEL2), [existing(POD_LABEL)], False),
    (1460, "PodAffinity and PodAntiAffinity satisfied",
     pod(aff_spec([pterm(sel(IN_SCAN), "region")], [pterm(sel(IN_ANTIVIRUS), "node")]), labels=POD_LABEL2),
     [existing(POD_LABEL)], True),
    (1532, "PodAffinity, PodAntiAffinity and symmetry satisfied",
     pod(aff_spec([pterm(sel(IN_SCAN), "region")], [pterm(sel(IN_ANTIVIRUS), "node")]), labels=POD_LABEL2),
     [existing(POD_LABEL, aff_spec(anti=[pterm(sel(IN_ANTIVIRUS), "node")]))], True),
    (1579, "PodAffinity satisfied, PodAntiAffinity not",
     pod(aff_spec([pterm(sel(IN_SCAN), "region")], [pterm(sel(IN_SCAN), "zone")]), labels=POD_LABEL2),
     [existing(POD_LABEL)], False),
    (1651, "PodAntiAffinity symmetry with the existing pod violated",
     pod(aff_spec([pterm(sel(IN_SCAN), "region")], [pterm(sel(IN_ANTIVIRUS), "node")]), labels=POD_LABEL),
     [existing(POD_LABEL, aff_spec(anti=[pterm(sel(IN_SCAN), "zone")]))], False),
    (1682, "pod does not match its own affinity term and no pod is on the node",
     pod(aff_spec([pterm(sel(NOTIN_SCAN), "region")]), labels=POD_LABEL), [], False),  # the listed pod sits on machine2
    (1717, "existing pod's anti-affinity respected (violated)", pod(labels=POD_LABEL),
     [existing(POD_LABEL, aff_spec(anti=[pterm(sel(IN_SCAN), "zone")]))], False),
    (1752, "existing pod's anti-affinity respected (satisfied)", pod(labels=POD_LABEL),
     [existing(POD_LABEL, aff_spec(anti=[pterm(sel(NOTIN_SCAN), "zone")]))], True),
    (1816, "incoming anti-affinity ok, symmetry with incoming pod violated",
     pod(aff_spec(anti=[pterm(sel(EX_SERVICE), "region"), pterm(sel(EX_SECURITY), "region")]), labels=POD_LABEL),
     [existing(POD_LABEL2, aff_spec(anti=[pterm(sel(EX_SECURITY), "zone")]))], False),
    (1879, "symmetry check a1", pod(aff_spec(anti=[pterm(sel(EX_SERVICE), "zone"), pterm(sel(EX_SECURITY), "zone")]), labels=POD_LABEL),
     [existing(POD_LABEL2, aff_spec(anti=[pterm(sel(EX_SECURITY), "zone")]))], False),
    (1942, "symmetry check a2", pod(aff_spec(anti=[pterm(sel(EX_SECURITY), "zone")]), labels=POD_LABEL2),
     [existing(POD_LABEL, aff_spec(anti=[pterm(sel(EX_SERVICE), "zone"), pterm(sel(EX_SECURITY), "zone")]))], False),
    (2016, "symmetry check b1", pod(aff_spec(anti=[pterm(sel(EX_ABC), "zone"), pterm(sel(EX_DEF), "zone")]), labels={"abc": "", "xyz": ""}),
     [existing({"def": "", "xyz": ""}, aff_spec(anti=[pterm(sel(EX_ABC), "zone"), pterm(sel(EX_DEF), "zone")]))], False),
    (2090, "symmetry check b2", pod(aff_spec(anti=[pterm(sel(EX_ABC), "zone"), pterm(sel(EX_DEF), "zone")]), labels={"def": "", "xyz": ""}),
     [existing({"abc": "", "xyz": ""}, aff_spec(anti=[pterm(sel(EX_ABC), "zone"), pterm(sel(EX_DEF), "zone")]))], False),
]


def predicate_cases():
    cases = []
    for line, name, p, on_node, fits in INTERPOD:
        cases.append({"test": "TestInterPodAffinity", "name": name, "source": f"{PM}:{line}",
                      "plugins": ["InterPodAffinity", "NodeAffinity"], "allocate": True, "pod": p,
                      "node": node("machine1", {"region": "r1", "zone": "z11"}, pods=on_node), "fits": fits})
    for line, name, p, existing, fits in HOST_PORTS:
        on_node = [port_pod("m1", *existing, uid="existing")] if existing else []
        cases.append({"test": "TestPodFitsHostPorts", "name": name, "source": f"{PM}:{line}", "plugins": ["NodePorts"],
                      "allocate": True, "pod": p, "node": node("", pods=on_node), "fits": fits})
    for line, name, p, labels, node_name, fits in SELECTOR:
        cases.append({"test": "TestPodFitsSelector", "name": name, "source": f"{PM}:{line}",
                      "plugins": ["NodePorts", "NodeAffinity"], "allocate": True,
                      "pod": p, "node": node(node_name, labels), "fits": fits})

    # TestPodFitsHost — :141-197, plugin NodeName (:142), allocate (:191)
    for line, name, p, n, fits in [
        (151, "no host specified", pod(), node(""), True),
        (157, "host matches", pod({"nodeName": "foo"}), node("foo"), True),
        (171, "host doesn't match", pod({"nodeName": "bar"}), node("foo"), False),
    ]:
        cases.append({"test": "TestPodFitsHost", "name": name, "source": f"{PM}:{line}", "plugins": ["NodeName"],
                      "allocate": True, "pod": p, "node": n, "fits": fits})

    # TestRunGeneralPredicates — :1095-1168, plugins NodeResourcesFit, NodeName, NodePorts, NodeVolumeLimits (:1096).
    alloc = make_resources(10, 20, 32)
    for line, name, p, existing, fits in [
        (1108, "no resources/port/host requested always fits", pod(), [resource_pod(9, 19)], True),
        (1120, "not enough cpu and memory resource", resource_pod(8, 10), [resource_pod(5, 19)], False),
        (1132, "host not match", pod({"nodeName": "machine2"}), [], False),
        (1147, "host port conflict", pod({"containers": [{"ports": [{"hostPort": 123}]}]}),
         [pod({"containers": [{"ports": [{"hostPort": 123}]}]}, uid="port-holder")], False),
    ]:
        cases.append({"test": "TestRunGeneralPredicates", "name": name, "source": f"{PM}:{line}",
                      "plugins": ["NodeResourcesFit", "NodeName", "NodePorts"], "allocate": True,
                      "pod": p, "node": node("machine1", alloc=alloc, pods=existing), "fits": fits})

    # TestReserveAlloc — :2115-2154, reservation phase (allocate=false), pod{nodeName: foo} on node foo that
    # already lists the pod (NewNodeInfo(pod) :2126).
    ra_pod = pod({"nodeName": "foo"}, uid="reserve-alloc")
    cases.append({"test": "TestReserveAlloc", "name": "no predicates configured", "source": f"{PM}:2129",
                  "plugins": [], "allocate": False, "pod": ra_pod, "node": node("foo", pods=[ra_pod]), "fits": True})
    cases.append({"test": "TestReserveAlloc", "name": "node is schedulable", "source": f"{PM}:2136",
                  "plugins": ["NodeUnschedulable"], "allocate": False, "pod": ra_pod, "node": node("foo", pods=[ra_pod]),
                  "fits": True})
    cases.append({"test": "TestReserveAlloc", "name": "node cordoned + unschedulable taint", "source": f"{PM}:2142",
                  "plugins": ["NodeUnschedulable"], "allocate": False, "pod": ra_pod,
                  "node": node("foo", pods=[ra_pod], unschedulable=True,
                               taints=[{"key": "node.kubernetes.io/unschedulable", "effect": "NoSchedule"}]),
                  "fits": False})

    # TestReserveNodeSelector — :2156-2199, reservation phase, plugins NodeName, NodePorts, PodTopologySpread,
    # NodeAffinity (:2171); node "node" holds the pod (:2168).
    for line, name, labels, selector, err in [
        (2181, "Match labels", {"foo": "bar"}, {"foo": "bar"}, False),
        (2182, "Missing labels", {"foo": "bar"}, {"foo2": "bar2"}, True),
        (2183, "empty node labels", {}, {"foo2": "bar2"}, True),
        (2184, "empty node selectors", {"foo": "bar"}, {}, False),
    ]:
        p = pod({"nodeSelector": selector}, uid="reserve-selector")
        cases.append({"test": "TestReserveNodeSelector", "name": name, "source": f"{PM}:{line}",
                      "plugins": ["NodeName", "NodePorts", "PodTopologySpread", "NodeAffinity"], "allocate": False,
                      "pod": p, "node": node("node", labels, pods=[pod({}, uid="reserve-selector")]), "fits": not err})
    return cases


def preemption_cases():
    """TestPreemptionPredicates :71-117 and TestPreemptionPredicatesEmpty :58-69; plugin NodeResourcesFit."""
    victims = [resource_pod(100, 1000000, name="pod0", uid="pod0"), resource_pod(100, 1000000, name="pod1", uid="pod1"),
               resource_pod(300, 3000000, name="pod2", uid="pod2"), resource_pod(500, 5000000, name="pod3", uid="pod3")]
    n0 = node("node0", alloc=make_resources(1000, 100000000, 10), pods=victims)
    return [
        {"test": "TestPreemptionPredicatesEmpty", "source": f"{PM}:58", "plugins": ["NodeResourcesFit"],
         "pod": pod(), "node": node(""), "victims": [], "start_index": 0, "index": -1},
        {"test": "TestPreemptionPredicates", "name": "smallpod", "source": f"{PM}:107", "plugins": ["NodeResourcesFit"],
         "pod": resource_pod(500, 5000000, name="smallpod", uid="smallpod"), "node": n0, "victims": [0, 1, 2, 3],
         "start_index": 1, "index": 2},
        {"test": "TestPreemptionPredicates", "name": "largepod", "source": f"{PM}:115", "plugins": ["NodeResourcesFit"],
         "pod": resource_pod(1500, 15000000, name="largepod", uid="largepod"), "node": n0, "victims": [0, 1, 2, 3],
         "start_index": 1, "index": -1},
    ]


def container(name, requests, restart=None):
    c = {"name": name, "resources": {"requests": requests}}
    if restart:
        c["restartPolicy"] = restart
    return c


def request_cases():
    """Request-vector known answers from pkg/common/resource_test.go. The expected maps drop YuniKorn's own
    "pods": 1 entry (resource.go:58), i.e. they are what the test asserts for upstream PodRequests too
    (resource_test.go:201-202 etc.)."""
    gpu = "nvidia.com/gpu"
    c1 = container("container-01", {"memory": "500M", "cpu": "1", gpu: "1"})
    c2 = container("container-02", {"memory": "1024M", "cpu": "2", gpu: "4"})
    overhead = {"memory": "500M", "cpu": "1", gpu: "1"}
    ic1 = container("initcontainer-01", {"memory": "4096M", "cpu": "0.5", gpu: "1"})
    ic2 = container("initcontainer-02", {"memory": "10000M", "cpu": "5.12", gpu: "4"})
    c1b = container("container-01", {"memory": "2000M", "cpu": "4.096", gpu: "2"})
    c2b = container("container-02", {"memory": "5000M", "cpu": "1.024", gpu: "2"})
    c1c = container("container-01", {"memory": "2000M", "cpu": "4.096"})
    c2c = container("container-02", {"memory": "5000M", "cpu": "1.024"})
    small = {"memory": "10M", "cpu": "1"}
    M = 1000 * 1000
    return [
        {"name": "two containers", "source": f"{RT}:197-200", "pod": pod({"containers": [c1, c2]}),
         "expect": {"cpu": 3000, "memory": 1524 * M, gpu: 5}},
        {"name": "two containers + overhead", "source": f"{RT}:214-217",
         "pod": pod({"containers": [c1, c2], "overhead": overhead}), "expect": {"cpu": 4000, "memory": 2024 * M, gpu: 6}},
        {"name": "init containers vs containers", "source": f"{RT}:266-269",
         "pod": pod({"containers": [c1b, c2b], "initContainers": [ic1, ic2]}),
         "expect": {"cpu": 5120, "memory": 10000 * M, gpu: 4}},
        {"name": "init container without cpu/gpu", "source": f"{RT}:287-290",
         "pod": pod({"containers": [c1c, c2c],
                     "initContainers": [ic1, container("initcontainer-02", {"memory": "10000M"})]}),
         "expect": {"cpu": 5120, "memory": 10000 * M, gpu: 1}},
        {"name": "single sidecar", "source": f"{RT}:314-316",
         "pod": pod({"containers": [c1c, c2c], "initContainers": [container("container-04", small, "Always")]}),
         "expect": {"cpu": 6120, "memory": 7010 * M}},
        {"name": "two sidecars + init container", "source": f"{RT}:357-359",
         "pod": pod({"containers": [c1c, c2c],
                     "initContainers": [container("container-05", small, "Always"), container("container-06", small, "Always"),
                                        container("container-07", {"memory": "4096M", "cpu": "10"})]}),
         "expect": {"cpu": 12000, "memory": 7020 * M}},
        {"name": "sidecar then init container (memory only)", "source": f"{RT}:424",
         "pod": pod({"containers": [container("container-main", {"memory": "512M"})],
                     "initContainers": [container("container-ic1", {"memory": "1024M"}, "Always"),
                                        container("container-ic2", {"memory": "256M"})]}),
         "expect": {"memory": 1536 * M}},
        {"name": "init container then sidecar", "source": f"{RT}:468-469",
         "pod": pod({"containers": [container("container-main", {"memory": "512M", "cpu": "1"})],
                     "initContainers": [container("container-ic1", {"memory": "1024M", "cpu": "1", gpu: "1"}),
                                        container("container-ic2", {"memory": "1024M", "cpu": "1"}, "Always")]}),
         "expect": {"cpu": 2000, "memory": 1536 * M, gpu: 1}},
        {"name": "sidecar, init, sidecar + two containers", "source": f"{RT}:539-541",
         "pod": pod({"containers": [container("container-main1", {"memory": "512M", "cpu": "1"}),
                                    container("container-main2", {"memory": "512M", "cpu": "1"})],
                     "initContainers": [container("container-ic1", {"memory": "1024M", "cpu": "1"}, "Always"),
                                        container("container-ic2", {"memory": "4096M", "cpu": "1", gpu: "1"}),
                                        container("container-ic3", {"memory": "512", "cpu": "100m"}, "Always")]}),
         "expect": {"cpu": 3100, "memory": 5120 * M, gpu: 1}},
        {"name": "pod-level requests override cpu/memory only", "source": f"{RT}:614-620",
         "pod": pod({"containers": [c1, c2], "resources": {"requests": {"memory": "128M", "cpu": "5", "invalid": "1"}}}),
         "expect": {"cpu": 5000, "memory": 128 * M, gpu: 5}},
    ] + resize_cases() + [
        # TestBestEffortPod (:746-788) and TestGPUOnlyResources (:790-832): zero-valued requests stay in the map
        {"name": "best effort pod", "source": f"{RT}:772-774", "pod": pod({"containers": [container("container-01", {})]}), "expect": {}},
        {"name": "cpu only", "source": f"{RT}:779-782", "pod": pod({"containers": [container("container-01", {"cpu": "1"})]}),
         "expect": {"cpu": 1000}},
        {"name": "explicit zero cpu and memory", "source": f"{RT}:784-791",
         "pod": pod({"containers": [container("container-01", {"memory": "0", "cpu": "0"})]}), "expect": {"cpu": 0, "memory": 0}},
        {"name": "gpu only", "source": f"{RT}:818-821", "pod": pod({"containers": [container("container-01", {gpu: "1"})]}),
         "expect": {gpu: 1}},
        {"name": "gpu only, zero", "source": f"{RT}:823-827", "pod": pod({"containers": [container("container-01", {gpu: "0"})]}),
         "expect": {gpu: 0}},
    ]


def resize_cases():
    """TestGetPodResourcesWithInPlacePodVerticalScaling (resource_test.go:624-744): KEP-1287 in-place resize. One pod walks
    through the states of a resize; every step asserts cpu (milli) and memory."""
    M = 1000 * 1000
    before = [{"memory": "1000M", "cpu": "1"}, {"memory": "2000M", "cpu": "2"}]
    after = [{"memory": "2000M", "cpu": "500m"}, {"memory": "4000M", "cpu": "1"}]
    names = ["container-01", "container-02"]

    def make(requests, statuses=None, resize=None, conditions=None):
        p = pod({"containers": [container(n, dict(r)) for n, r in zip(names, requests)]}, name="pod-resource-test-00001", uid="UID-00001")
        st = {}
        if statuses is not None:
            st["containerStatuses"] = statuses
        if resize is not None:
            st["resize"] = resize
        if conditions is not None:
            st["conditions"] = conditions
        if st:
            p["status"] = st
        return p

    def status(allocated, resources):
        out = []
        for n, a, r in zip(names, allocated, resources):
            cs = {"name": n}
            if a is not None:
                cs["allocatedResources"] = dict(a)
            if r is not None:
                cs["resources"] = {"requests": dict(r)}
            out.append(cs)
        return out

    running = status(before, before)
    pending = [{"type": "PodResizePending", "status": "True", "reason": "Infeasible"}]
    return [
        {"name": "resize: no container statuses", "source": f"{RT}:664-668", "pod": make(before), "expect": {"cpu": 3000, "memory": 3000 * M}},
        {"name": "resize: empty container statuses", "source": f"{RT}:670-675", "pod": make(before, []), "expect": {"cpu": 3000, "memory": 3000 * M}},
        {"name": "resize: statuses without resources", "source": f"{RT}:677-685", "pod": make(before, status([None, None], [None, None])),
         "expect": {"cpu": 3000, "memory": 3000 * M}},
        {"name": "resize: running, status equals spec", "source": f"{RT}:687-695", "pod": make(before, running),
         "expect": {"cpu": 3000, "memory": 3000 * M}},
        {"name": "resize: proposed (memory up, cpu down)", "source": f"{RT}:697-706", "pod": make(after, running, resize=""),
         "expect": {"cpu": 3000, "memory": 6000 * M}},
        {"name": "resize: infeasible (status.resize)", "source": f"{RT}:708-713", "pod": make(after, running, resize="Infeasible"),
         "expect": {"cpu": 3000, "memory": 3000 * M}},
        {"name": "resize: infeasible (PodResizePending condition)", "source": f"{RT}:715-721",
         "pod": make(after, running, resize="", conditions=pending), "expect": {"cpu": 3000, "memory": 3000 * M}},
        {"name": "resize: in progress (allocated = new spec)", "source": f"{RT}:723-731",
         "pod": make(after, status(after, before), resize="InProgress", conditions=[{"type": "PodResizing", "status": "True"}]),
         "expect": {"cpu": 3000, "memory": 6000 * M}},
        {"name": "resize: completed", "source": f"{RT}:733-741", "pod": make(after, status(after, after), resize="", conditions=[]),
         "expect": {"cpu": 1500, "memory": 6000 * M}},
    ]


QUANTITIES = [
    # (text, Value(), MilliValue()) — resource.Quantity semantics used by resource.go:273-285
    ("1", 1, 1000), ("2", 2, 2000), ("0.5", 1, 500), ("5.12", 6, 5120), ("500M", 500000000, 500000000000),
    ("1024M", 1024000000, 1024000000000), ("4096M", 4096000000, 4096000000000), ("100m", 1, 100), ("10m", 1, 10),
    ("1Gi", 1073741824, 1073741824000), ("256Gi", 274877906944, 274877906944000), ("128Mi", 134217728, 134217728000),
    ("1Ki", 1024, 1024000), ("1k", 1000, 1000000), ("1e3", 1000, 1000000), ("1E3", 1000, 1000000), ("1E", 10**18, 9223372036854775807),
    ("1.5Gi", 1610612736, 1610612736000), ("0", 0, 0), ("32", 32, 32000), ("110", 110, 110000), ("1u", 1, 1), ("1n", 1, 1),
    ("1500u", 1, 2), ("0.0001", 1, 1), ("12e-1", 2, 1200), ("14500m", 15, 14500), ("1G", 10**9, 10**12), ("1000M", 10**9, 10**12),
]


E2E = "test/e2e/predicates/predicates_test.go"
E2E_PREEMPT = "test/e2e/preemption/preemption_suite_test.go"
UPSTREAM_DOC = "k8s.io/api core/v1 Toleration.ToleratesTaint + kube-scheduler TaintToleration.Filter (upstream, not vendored)"
KWOK_SETUP = "deployments/kwok-perf-test/kwok-setup.sh"
KWOK_DEPLOY = "deployments/kwok-perf-test/deploy-tool.sh"


def taint_cases():
    """TaintToleration has no unit-level vector in the reference (the plugin is only enabled through "*"); what the reference
    DOES hold is behaviour: two e2e scenarios and the KWOK perf-test shapes. Transcribed here as single (pod, node) cases with
    the whole default plugin set ("*", NewPredicateManager) — `plugin` / `message_regex` where the scenario asserts them."""
    key, value = "kubernetes.io/e2e-taint-key-abcdefghij", "testing-taint-value"
    label_key, label_value = "kubernetes.io/e2e-label-key-klmnopqrst", "testing-label-value"
    alloc = {"cpu": "4", "memory": "8Gi", "pods": "110"}
    tainted = node("e2e-node", labels={label_key: label_value}, alloc=alloc, taints=[{"key": key, "value": value, "effect": "NoSchedule"}])
    untainted = node("e2e-node", labels={label_key: label_value}, alloc=alloc)
    sleep = [{"name": "sleepcontainer", "resources": {"requests": {"cpu": "100m", "memory": "100M"}}}]
    with_tol = pod({"containers": sleep, "nodeSelector": {label_key: label_value},
                    "tolerations": [{"key": key, "value": value, "effect": "NoSchedule"}]},  # operator unset = Equal
                   name="with-tolerations", uid="with-tolerations", namespace="ns", labels={"app": "tolerations-app", "applicationId": "app-1"})
    no_tol = pod({"containers": sleep, "nodeSelector": {label_key: label_value}},
                 name="with-no-tolerations", uid="with-no-tolerations", namespace="ns", labels={"app": "no-tolerations-app", "applicationId": "app-2"})
    kwok_node = node("kwok-node-0", labels={"beta.kubernetes.io/arch": "amd64", "beta.kubernetes.io/os": "linux", "kubernetes.io/arch": "amd64",
                                            "kubernetes.io/hostname": "kwok-node-0", "kubernetes.io/os": "linux", "kubernetes.io/role": "agent",
                                            "node-role.kubernetes.io/agent": "", "type": "kwok"},
                     alloc={"cpu": "32", "memory": "256Gi", "pods": "110"}, taints=[{"effect": "NoSchedule", "key": "kwok.x-k8s.io/node", "value": "fake"}])
    kwok_pod = pod({"containers": [{"name": "sleep300"}], "tolerations": [{"key": "kwok.x-k8s.io/node", "operator": "Exists", "effect": "NoSchedule"}]},
                   name="sleep-deployment-0-abc", uid="sleep-deployment-0-abc", namespace="default",
                   labels={"app": "nginx", "applicationId": "sleep-deployment-0", "queue": "root.default"})
    real_pod = pod({"containers": [{"name": "main"}]}, name="actual-pod", uid="actual-pod", namespace="default", labels={"app": "real"})

    no_tol_plain = pod({"containers": sleep}, name="sleepjob1", uid="sleepjob1", namespace="dev", labels={"app": "sleep", "applicationId": "app-3"})
    herd_tainted = node("worker-2", alloc=alloc, taints=[{"key": "e2e_test_preemption", "value": "value", "effect": "NoSchedule"}])
    herd_worker = node("worker-1", alloc=alloc)

    def tol_pod(tolerations):
        return pod({"containers": sleep, "tolerations": tolerations}, name="t", uid="t", namespace="ns", labels={"app": "t"})

    def tnode(taints):
        return node("tn", alloc=alloc, taints=[{"key": k, "value": v, "effect": e} for k, v, e in taints])

    def case(name, source, p, n, fits, plugin=None, regex=None):
        c = {"test": "TaintToleration", "name": name, "source": source, "plugins": ["*"], "allocate": True, "pod": p, "node": n, "fits": fits}
        if plugin is not None:
            c["plugin"] = plugin
        if regex is not None:
            c["message_regex"] = regex
        return c

    return [
        case("Verify_Matching_Taint_Tolerations_Respected: toleration {Key, Value, Effect} on the tainted, labelled node", f"{E2E}:334-381",
             with_tol, tainted, True),
        case("Verify_Not_Matching_Taint_Tolerations_Respected: no toleration, pod stays pending, log matches .*taint.*", f"{E2E}:384-439",
             no_tol, tainted, False, "TaintToleration", ".*taint.*"),
        case("Verify_Not_Matching_Taint_Tolerations_Respected: after UntaintNode the pod is scheduled on the node", f"{E2E}:441-447",
             no_tol, untainted, True),
        case("KWOK: deploy-tool pod (toleration Exists/NoSchedule, no requests) on a kwok-setup node", f"{KWOK_DEPLOY}:33-63 x {KWOK_SETUP}:31-58",
             kwok_pod, kwok_node, True),
        case("KWOK: a pod without the toleration is kept off the fake node ('Avoid scheduling actual running pods to fake Node')",
             f"{KWOK_SETUP}:50-53", real_pod, kwok_node, False, "TaintToleration", ".*taint.*"),
        # the preemption / simple_preemptor / recovery suites herd their sleep pods onto ONE worker by tainting every other node
        # (kClient.TaintNodes(nodesToTaint, taintKey, "value", NoSchedule)): pods without a toleration only fit the untainted worker
        case("preemption suite: sleep pod without toleration vs a node tainted e2e_test_preemption=value:NoSchedule", f"{E2E_PREEMPT}:85-99",
             no_tol_plain, herd_tainted, False, "TaintToleration", ".*taint.*"),
        case("preemption suite: the same pod on the one untainted worker", f"{E2E_PREEMPT}:85-99", no_tol_plain, herd_worker, True),
    ] + [dict(case(name, UPSTREAM_DOC, p, n, fits, plugin), pinned_by="upstream documentation of v1.Toleration.ToleratesTaint / TaintToleration.Filter — "
              "NOT held by any test, fixture or script of the reference (SURVEY.md §8c: parity unpinned)") for name, p, n, fits, plugin in [
        ("upstream rule: an untolerated NoExecute taint filters like NoSchedule", tol_pod([]), tnode([("maint", "true", "NoExecute")]), False, "TaintToleration"),
        ("upstream rule: a toleration without effect tolerates NoExecute too", tol_pod([{"key": "maint", "operator": "Exists"}]),
         tnode([("maint", "true", "NoExecute")]), True, None),
        ("upstream rule: PreferNoSchedule taints are ignored by the Filter", tol_pod([]), tnode([("soft", "x", "PreferNoSchedule")]), True, None),
        ("upstream rule: effect mismatch (toleration NoExecute, taint NoSchedule) does not tolerate",
         tol_pod([{"key": "k", "operator": "Equal", "value": "v", "effect": "NoExecute"}]), tnode([("k", "v", "NoSchedule")]), False, "TaintToleration"),
        ("upstream rule: empty key with Exists tolerates every taint", tol_pod([{"operator": "Exists"}]),
         tnode([("a", "1", "NoSchedule"), ("b", "2", "NoExecute")]), True, None),
        ("upstream rule: Equal with an empty value tolerates a taint whose value is empty", tol_pod([{"key": "k", "operator": "Equal", "effect": "NoSchedule"}]),
         tnode([("k", "", "NoSchedule")]), True, None),
        ("upstream rule: Equal with an empty value does NOT tolerate value 'value' (the getSchedulerPodTolerations shape, "
         "test/e2e/recovery_and_restart/recovery_and_restart_test.go:341-347)", tol_pod([{"key": "k", "operator": "Equal", "effect": "NoSchedule"}]),
         tnode([("k", "value", "NoSchedule")]), False, "TaintToleration"),
        ("upstream rule: one untolerated taint among tolerated ones fails", tol_pod([{"key": "a", "operator": "Exists"}]),
         tnode([("a", "1", "NoSchedule"), ("b", "2", "NoSchedule")]), False, "TaintToleration"),
    ]]


E2E_BINPACK = "test/e2e/bin_packing/bin_packing_test.go"


def binpacking_cases():
    """The reference's ONE behavioural pin of the bin-pack decision order (Verify_BinPacking_Node_Order_Memory,
    test/e2e/bin_packing/bin_packing_test.go:52-189, queue config `nodesortpolicy: binpacking`, bin_packing_suite_test.go:66-89):
      1. the worker nodes are sorted by increasing AVAILABLE memory: [nodeA, nodeB, ...]                              (:58-71)
      2. a padding pod of 20 % of nodeA's available memory (cpu 0m) is placed on nodeA through nodeSelector
         kubernetes.io/hostname=nodeA, labels node=nodeA; one of 10 % of nodeB's on nodeB                              (:77-110)
      3. job A: 3 pods without requests and without constraints — ALL land on nodeA (least available memory first)    (:127-139,169-188)
      4. job B: 3 pods with required pod anti-affinity to {node In [nodeA]} on kubernetes.io/hostname — ALL on nodeB  (:142-188)
    The e2e reads the numbers off a live kind cluster; a fixture has to choose them. Chosen here: identical allocatable on every
    worker (kind workers share the host), one system pod per node with equal cpu (kindnet's 100m / 50Mi) and a resident pod whose
    memory makes the available-memory order strict and different from the NodeID order. The asks are listed in the order the e2e
    submits them (it waits for each stage to run before the next: sequential assume) — one conflict-resolved round of 8 asks.
    `expect` = spec.nodeName the e2e asserts, for the padding pods by construction of their nodeSelector."""
    GI = 1 << 30
    MI = 1 << 20
    cases = []
    for workers, resident_gi, names in [
        (2, [3, 1], ["kind-worker", "kind-worker2"]),
        (3, [1, 3, 2], ["kind-worker", "kind-worker2", "kind-worker3"]),            # memory order != NodeID order
        (5, [2, 0, 4, 1, 3], ["kind-worker", "kind-worker2", "kind-worker3", "kind-worker4", "kind-worker5"]),
    ]:
        alloc_mem = 32 * GI
        nodes, avail = [], {}
        for name, gi in zip(names, resident_gi):
            system = [pod({"nodeName": name, "containers": [{"name": "kindnet-cni", "resources": {"requests": {"cpu": "100m", "memory": "50Mi"}}}]},
                          name=f"kindnet-{name}", uid=f"kindnet-{name}", namespace="kube-system", labels={"app": "kindnet"}),
                      pod({"nodeName": name, "containers": [{"name": "kube-proxy"}]},
                          name=f"kube-proxy-{name}", uid=f"kube-proxy-{name}", namespace="kube-system", labels={"k8s-app": "kube-proxy"})]
            if gi:
                system.append(pod({"nodeName": name, "containers": [{"name": "resident", "resources": {"requests": {"memory": f"{gi}Gi"}}}]},
                                  name=f"resident-{name}", uid=f"resident-{name}", namespace="default", labels={"app": "resident"}))
            nodes.append(node(name, labels={"kubernetes.io/hostname": name, "kubernetes.io/os": "linux"},
                              alloc={"cpu": "8", "memory": str(alloc_mem), "pods": "110"}, pods=system))
            avail[name] = alloc_mem - 50 * MI - gi * GI
        by_avail = sorted(names, key=lambda n: avail[n])                      # :60-66 (strict by construction)
        node_a, node_b = by_avail[0], by_avail[1]
        asks, expect = [], []
        for name, pct in ((node_a, 0.2), (node_b, 0.1)):                      # :77-110
            padding = int(float(avail[name]) * pct)                           # int64(float64(nodeAvailMem.Value()) * padPct[i])
            asks.append(pod({"nodeSelector": {"kubernetes.io/hostname": name},
                             "containers": [{"name": "sleepcontainer", "resources": {"requests": {"cpu": "0m", "memory": str(padding)}}}]},
                            name=f"{name}-padding", uid=f"{name}-padding", namespace="ns-binpack",
                            labels={"node": name, "app": "app-padding", "applicationId": f"appid-{name}-padding"}))
            expect.append(name)
            avail[name] -= padding
        after = sorted(names, key=lambda n: avail[n])                         # :113-124: the order the assertions use
        assert after[:2] == [node_a, node_b]
        for i in range(3):                                                    # job A :127-139
            asks.append(pod({"containers": [{"name": "sleepcontainer"}]}, name=f"joba-{i}", uid=f"joba-{i}", namespace="ns-binpack",
                            labels={"app": "sleep-joba", "applicationId": "appid-joba", "job-name": "joba"}))
            expect.append(after[0])
        anti = {"podAntiAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": [
            {"labelSelector": {"matchExpressions": [{"key": "node", "operator": "In", "values": [after[0]]}]}, "topologyKey": "kubernetes.io/hostname"}]}}
        for i in range(3):                                                    # job B :142-167
            asks.append(pod({"containers": [{"name": "sleepcontainer"}], "affinity": anti}, name=f"jobb-{i}", uid=f"jobb-{i}", namespace="ns-binpack",
                            labels={"app": "sleep-jobb", "applicationId": "appid-jobb", "job-name": "jobb"}))
            expect.append(after[1])
        cases.append({"test": "Verify_BinPacking_Node_Order_Memory", "name": f"{workers} workers", "source": f"{E2E_BINPACK}:52-189",
                      "nodes": nodes, "pods": asks, "expect": expect,
                      "sorted_by_available_memory_after_padding": after})
    return cases


def main():
    with open(os.path.join(HERE, "taint_cases.json"), "w") as f:
        json.dump(taint_cases(), f, indent=1)
    with open(os.path.join(HERE, "predicate_cases.json"), "w") as f:
        json.dump(predicate_cases(), f, indent=1)
    with open(os.path.join(HERE, "preemption_cases.json"), "w") as f:
        json.dump(preemption_cases(), f, indent=1)
    with open(os.path.join(HERE, "binpacking_cases.json"), "w") as f:
        json.dump(binpacking_cases(), f, indent=1)
    req = request_cases()
    with open(os.path.join(HERE, "request_cases.json"), "w") as f:
        json.dump(req, f, indent=1)
    with open(os.path.join(HERE, "quantity_cases.json"), "w") as f:
        json.dump([{"text": t, "value": v, "milli": m} for t, v, m in QUANTITIES], f, indent=1)


if __name__ == "__main__":
    main()
